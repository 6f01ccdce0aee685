#!/usr/bin/env python3
"""Chained launch (stages 1-4 in one persistent kernel) against one launch per stage: bit for bit, then timing.
    python scripts/chain_check.py [reps]"""
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


PERIODS = ("4", "32", "240")  # tiles a workgroup finishes between two publications


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for prec in ("f32", "split_f16"):
        eng = r.Engine(r.rsr.builtin("imagenet"), device=0, precision=prec)
        for name, n, h, w in (("600x1000", 1, 600, 1000), ("2x333x640", 2, 333, 640), ("1080p", 1, 1080, 1920), ("band8 284x3840", 1, 284, 3840), ("4K", 1, 2160, 3840)):
            px = torch.from_numpy(synth_u8(7, h, w, n=n) if n > 1 else synth_u8(7, h, w)[None]).cuda()
            band = name.startswith("band8")
            def call(out=None):
                if band:
                    return eng.upscale_band_rgba8_dev(px[0], 7, 7, out=out)
                return eng.upscale_rgba8_dev(px, out=out)
            res = {}
            for chain in ("0",) + PERIODS:
                eng.set_experiment("chain", chain)
                out = call()
                torch.cuda.synchronize()
                for _ in range(3):
                    call(out)
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        call(out)
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) / reps * 1e3)
                res[chain] = (out.cpu().numpy(), best)
            same = all(bool(np.array_equal(res["0"][0], res[p][0])) for p in PERIODS)
            print(json.dumps({"prec": prec, "shape": name, "identical": same, "ms_per_stage_launches": round(res["0"][1], 4),
                              "ms_chained": {p: round(res[p][1], 4) for p in PERIODS},
                              "gain": {p: round(1 - res[p][1] / res["0"][1], 4) for p in PERIODS}}), flush=True)
        eng.set_experiment("chain", "")
        eng.close()


if __name__ == "__main__":
    main()
