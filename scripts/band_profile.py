#!/usr/bin/env python3
"""Where the time of a row band of config C goes (reference call being sharded: main.rs:171).

For the 3840x2160 frame and one INTERIOR band of its 2- / 4- / 8-way splits (rows + 7 halo rows either side, exactly what
one rank of `sr_upscale_sharded_*` computes after the exchange) this prints, per stage kernel: rows computed, tiles,
tiles per resident workgroup slot, fill of the last round, time (HIP events inside libsrhip, median), and the time the
same rows would take at the undivided frame's rate.  Device-resident, u8 in / RGBA8 out.

    python scripts/band_profile.py [prec] [reps] [--tails] [--once WAYS]     (--once: just run that band a few times, for rocprofv3;
    --tails: also two other lengths of the 4-row tail)
Per band three tile plans are timed: the automatic one (8-row tiles ended by 4-row tiles), 8-row tiles only, 4-row tiles only.
"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rusty_sr_amd as r  # noqa: E402
from bench import synth_u8  # noqa: E402

MARGIN = (5, 3, 2, 1, 0)
MAC = (2400, 25600, 34816, 44032, 23328)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    prec = args[0] if args else "f32"
    reps = int(args[1]) if len(args) > 1 else 7
    once = int(sys.argv[sys.argv.index("--once") + 1]) if "--once" in sys.argv else 0
    HC, WC = 2160, 3840
    eng = r.Engine(r.rsr.builtin("imagenet"), device=0, precision=prec)
    img = synth_u8(3, HC, WC)
    cus = eng.device_info()["compute_units"]

    def run(ways, th=None, tail=""):
        if th is not None:
            eng.set_experiment("th", th)
            eng.set_experiment("tail", tail)
        if ways == 1:
            ext, top, bot, rows = img, 0, 0, HC
        else:
            rows = HC // ways
            a = rows * (ways // 2)  # an interior band (2-way: the lower one, which has one halo)
            top = 7
            bot = 7 if ways > 2 else 0
            ext = img[a - top:a + rows + bot]
        x = torch.from_numpy(np.ascontiguousarray(ext)).cuda()
        out = torch.empty((3 * rows, 3 * WC, 4), dtype=torch.uint8, device="cuda")
        fn = (lambda: eng.upscale_band_rgba8_dev(x, top, bot, out=out)) if ways > 1 else (lambda: eng.upscale_rgba8_dev(x[None], out=out[None]))
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        if once:
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return None
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        eng.set_profiling(True)
        acc = []
        for _ in range(reps):
            fn()
            torch.cuda.synchronize()
            acc.append(eng.last_timing()["stage_ms"])
        eng.set_profiling(False)
        st = np.median(np.array(acc), axis=0)
        h_ext = ext.shape[0]
        stage_rows = [min(h_ext, rows + min(m, top) + min(m, bot)) for m in MARGIN]
        return {"ways": ways, "rows": rows, "wall_ms": round(wall, 4), "stage_ms": [round(float(v), 4) for v in st],
                "sum_stage_ms": round(float(st.sum()), 4), "stage_rows": stage_rows}

    if once:
        run(once)
        return
    full = run(1)
    per_row = [full["stage_ms"][s] / HC for s in range(5)]
    print(json.dumps({"full_frame": full}))
    variants = [("", ""), ("8", "")] + ([("", t) for t in ("0.75", "3")] if "--tails" in sys.argv else [])
    for ways in (2, 4, 8):
        for th, tail in variants:
            b = run(ways, th, tail)
            slots = 2 * cus
            rowsx = []
            for s in range(5):
                thh = 4 if th == "4" else 8
                tiles = ((b["stage_rows"][s] + thh - 1) // thh) * (WC // 32)  # (mixed launches: counted as 8-row tiles)
                ideal = per_row[s] * b["rows"]
                rowsx.append({"stage": s, "rows": b["stage_rows"][s], "tiles": tiles, "tiles_per_slot": round(tiles / slots, 2),
                              "ms": b["stage_ms"][s], "ms_at_frame_rate_own_rows": round(ideal, 4),
                              "ratio": round(b["stage_ms"][s] / ideal, 3),
                              "tflops": round(2 * MAC[s] * b["stage_rows"][s] * WC / (b["stage_ms"][s] / 1e3) / 1e12, 1)})
            b["th"] = th or "mixed"
            b["tail"] = tail or "auto"
            b["ideal_ms"] = round(full["wall_ms"] / ways, 4)
            b["wall_over_ideal"] = round(b["wall_ms"] / b["ideal_ms"], 4)
            b["useful_tflops"] = round(2 * sum(MAC) * b["rows"] * WC / (b["wall_ms"] / 1e3) / 1e12, 1)
            b["stages"] = rowsx
            print(json.dumps(b))
    eng.set_experiment("th", "")
    eng.set_experiment("tail", "")


if __name__ == "__main__":
    main()
