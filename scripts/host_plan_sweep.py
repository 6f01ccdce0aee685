#!/usr/bin/env python3
"""Host-pointer call (sr_upscale_rgba8, page-locked buffers) under explicit band plans: median ms per call and whether the
bytes equal the automatic plan's.  python scripts/host_plan_sweep.py [prec] [H W]
One JSON line per plan.  ("rows" experiment key: "a,b,c" = bands computed in order on one stream, "=a,b,c" on alternating streams.)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rusty_sr_amd as r  # noqa: E402
from rusty_sr_amd.engine import host_alloc  # noqa: E402
from bench import synth_u8  # noqa: E402


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
    H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1080, 1920)
    eng = r.Engine(r.rsr.builtin("imagenet"), device=0, precision=prec)
    f32io = os.environ.get("SWEEP_IO") == "f32"  # SWEEP_IO=f32: sr_upscale_f32 (f32 in, f32 out: three times the download)
    if f32io:
        pin_in, pin_out = host_alloc((H, W, 3), np.float32), host_alloc((3 * H, 3 * W, 3), np.float32)
        pin_in.array[...] = r.img_to_data(synth_u8(2, H, W))
        import ctypes as C
        fp = C.POINTER(C.c_float)
        call = lambda: r._lib.check(eng._L.sr_upscale_f32(eng._ctx, pin_in.array.ctypes.data_as(fp), 1, H, W, pin_out.array.ctypes.data_as(fp)), eng._ctx)
    else:
        pin_in, pin_out = host_alloc((H, W, 3)), host_alloc((3 * H, 3 * W, 4))
        pin_in.array[...] = synth_u8(2, H, W)
        call = lambda: eng.upscale_rgba8(pin_in.array, out=pin_out.array)
    plans = [""] + (sys.argv[4:] if len(sys.argv) > 4 else [])
    if len(plans) == 1:
        def cut(fracs):
            rows = [int(H * f) // 8 * 8 for f in fracs[:-1]]
            return rows + [H - sum(rows)]
        for fr in ([.25] * 4, [1 / 3] * 3, [.5, .5], [.2] * 5):
            plans.append("=" + ",".join(map(str, cut(fr))))
            plans.append(",".join(map(str, cut(fr))))
        for fr in ([.74, .26], [.70, .30], [.66, .34], [.70, .24, .06], [.68, .24, .08], [.64, .25, .11], [.60, .26, .14], [.55, .28, .17],
                   [.5, .3, .2], [.45, .32, .23], [.4, .33, .27], [.4, .3, .2, .1], [.35, .3, .2, .15], [.3, .3, .25, .15], [.3, .28, .24, .18],
                   [.45, .3, .15, .1], [.5, .25, .15, .1]):
            plans.append(",".join(map(str, cut(fr))))
        for fr in ([.3, .3, .25, .15], [.28, .28, .26, .18], [.3, .3, .3, .1], [.35, .35, .2, .1], [.4, .4, .2]):
            plans.append("=" + ",".join(map(str, cut(fr))))
    want = None
    for plan in plans:
        eng.set_experiment("rows", plan)
        for _ in range(3):
            call()
        per = []
        for _ in range(25):
            t0 = time.perf_counter()
            call()
            per.append((time.perf_counter() - t0) * 1e3)
        t = eng.last_timing()
        if want is None:
            want = pin_out.array.copy()
        print(json.dumps({"prec": prec, "io": "f32" if f32io else "rgba8", "image": [H, W], "rows": plan or "auto", "ms_median": round(float(np.median(per)), 4), "ms_min": round(min(per), 4),
                          "kernel_ms": round(t["total_ms"], 4), "d2h_ms": round(t["d2h_ms"], 4), "chunks": t.get("chunks"),
                          "same_bytes": bool(np.array_equal(want, pin_out.array))}), flush=True)
    eng.set_experiment("rows", "")


if __name__ == "__main__":
    main()
