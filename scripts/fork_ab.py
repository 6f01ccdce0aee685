#!/usr/bin/env python3
"""A/B of the forked device call (sr_run_stack_auto: one image as two row bands on two streams) against the undivided
call, in ONE process, interleaved rounds, bit-for-bit check of every variant.
    python scripts/fork_ab.py [--prec f32] [--sizes 1080x1920,2160x3840] [--variants 0,1,1:0.45,1:0.55] [--rounds 5] [--steps 20]
A variant is the value of sr_set_experiment("fork") with an optional ":share" (forkshare) and ":tail" (4-row tail tiles: "0" none, "1" ...).  One JSON line per (size, variant)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--prec", default="f32")
ap.add_argument("--sizes", default="1080x1920")
ap.add_argument("--variants", default="0,1")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--io", default="rgba8")
a = ap.parse_args()

import torch
import rusty_sr_amd as r
from bench import synth_u8, FLOP_PER_PX, PEAK_F32_MFMA_TFLOPS, PEAK_F16_MFMA_TFLOPS

eng = r.Engine(r.rsr.builtin("imagenet"), device=0, precision=a.prec)
peak = PEAK_F32_MFMA_TFLOPS if a.prec == "f32" else PEAK_F16_MFMA_TFLOPS
for size in a.sizes.split(","):
    H, W = map(int, size.split("x"))
    px = synth_u8(2, H, W)
    if a.io == "rgba8":
        x = torch.from_numpy(px).cuda()[None]
        fn = eng.upscale_rgba8_dev
    else:
        x = torch.from_numpy(r.img_to_data(px)).cuda()[None]
        fn = eng.upscale_f32_dev
    variants = a.variants.split(",")
    outs, times = {}, {v: [] for v in variants}

    def select(v):  # "fork[:share[:tail]]"
        fork, share, tail = (v.split(":") + ["", ""])[:3]
        eng.set_experiment("fork", fork)
        eng.set_experiment("forkshare", share)
        eng.set_experiment("tail", tail)

    for v in variants:
        select(v)
        outs[v] = fn(x)
        for _ in range(5):
            fn(x, out=outs[v])
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for v in variants:
            select(v)
            for _ in range(3):
                fn(x, out=outs[v])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                fn(x, out=outs[v])
            torch.cuda.synchronize()
            times[v].append((time.perf_counter() - t0) / a.steps * 1e3)
    base = outs[variants[0]]
    for v in variants:
        ms = float(np.median(times[v]))
        print(json.dumps({"prec": a.prec, "io": a.io, "image": [H, W], "fork": v, "ms_median": round(ms, 4), "ms_min": round(min(times[v]), 4),
                          "whole_call_frac": round(H * W * FLOP_PER_PX / (ms / 1e3) / 1e12 / peak, 4),
                          "same_bytes_as_first_variant": bool(torch.equal(outs[v], base))}), flush=True)
    del outs, x
    torch.cuda.empty_cache()
eng.set_experiment("fork", "")
eng.set_experiment("forkshare", "")
eng.set_experiment("tail", "")
