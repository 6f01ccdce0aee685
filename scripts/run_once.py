#!/usr/bin/env python3
"""Minimal driver for rocprofv3: a few device-resident passes of one workload.
    python scripts/run_once.py [prec] [HxW] [reps]      (SRHIP_LIB / SRHIP_BW / SRHIP_TH / SRHIP_TAIL select variants)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rusty_sr_amd as r  # noqa: E402
from bench import synth_u8  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "split_f16"
H, W = map(int, (sys.argv[2] if len(sys.argv) > 2 else "1080x1920").split("x"))
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
eng = r.Engine(r.rsr.builtin("imagenet"), device=0, precision=prec)
px = torch.from_numpy(synth_u8(2, H, W)).cuda()[None]
out = eng.upscale_rgba8_dev(px)
for _ in range(reps):
    eng.upscale_rgba8_dev(px, out=out)
torch.cuda.synchronize()
