#!/usr/bin/env python3
"""Where does a split-f16 (or f32) stage kernel spend its non-matrix time?  Timing experiments that break the results on
purpose (sr_set_experiment "dbg"): 1 = contiguous gathers, 2 = no half-tile gathers, 4 = no epilogue stores.
    python scripts/dbg_exp.py [reps]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rusty_sr_amd as r  # noqa: E402
from bench import synth_u8  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
params = r.rsr.builtin("imagenet")
H, W = 1080, 1920
px = torch.from_numpy(synth_u8(2, H, W)).cuda()[None]
for prec in ("split_f16", "f32"):
    eng = r.Engine(params, device=0, precision=prec)
    out = eng.upscale_rgba8_dev(px)
    for dbg in (0, 1, 2, 4, 5, 6, 0):
        eng.set_experiment("dbg", str(dbg))
        for _ in range(3):
            eng.upscale_rgba8_dev(px, out=out)
        torch.cuda.synchronize()
        eng.set_profiling(True)
        acc = []
        for _ in range(reps):
            eng.upscale_rgba8_dev(px, out=out)
            torch.cuda.synchronize()
            acc.append(eng.last_timing()["stage_ms"])
        eng.set_profiling(False)
        st = np.median(np.array(acc), axis=0)
        print(f"{prec:9s} dbg={dbg}  stages {' '.join(f'{v:7.4f}' for v in st)}  sum {st.sum():.4f} ms", flush=True)
    eng.set_experiment("dbg", "0")
    eng.close()
