#!/usr/bin/env python3
"""A/B of the tile order (sr_set_experiment "bw": column-block width in tiles) on device-resident images:
per-stage kernel times (HIP events inside libsrhip) for both arithmetic modes at 1080p and 4K.
    python scripts/bw_exp.py [reps]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rusty_sr_amd as r  # noqa: E402
from bench import synth_u8  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
params = r.rsr.builtin("imagenet")
res = {}
for (H, W) in ((1080, 1920), (2160, 3840)):
    px = torch.from_numpy(synth_u8(2, H, W)).cuda()[None]
    for prec in ("f32", "split_f16"):
        eng = r.Engine(params, device=0, precision=prec)
        out = eng.upscale_rgba8_dev(px)
        for bw in ("0", "4", "8", "16", "32"):
            eng.set_experiment("bw", bw)
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
            res[f"{W}x{H} {prec} bw={bw}"] = [round(float(v), 4) for v in st] + [round(float(st.sum()), 4)]
            print(f"{W}x{H} {prec:9s} bw={bw:>2s}  stages {' '.join(f'{v:7.4f}' for v in st)}  sum {st.sum():.4f} ms", flush=True)
        eng.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "bw_exp.json"), "w"), indent=1)
