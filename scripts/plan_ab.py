#!/usr/bin/env python3
"""Interleaved A/B of whole tile PLANS for one device-resident image: every combination of the experiment switches named on the command
line (fork x pipe x th x tail), bit-for-bit check, ms per call.  One JSON line per shape with all variants.
    python scripts/plan_ab.py [--prec f32] [--sizes 448x448,...] [--plans "fork=0;fork=1;fork=0,pipe=all;fork=1,pipe=all"]"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--prec", default="f32")
ap.add_argument("--sizes", default="320x320,448x448,576x576,640x480")
ap.add_argument("--plans", default="fork=0;fork=1;fork=0,pipe=all;fork=1,pipe=all;fork=0,th=8;fork=1,th=8;fork=0,pipe=all,th=4;fork=1,pipe=all,th=4")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()

import torch  # noqa: E402
import rusty_sr_amd as r  # noqa: E402
from bench import synth_u8  # noqa: E402

KEYS = ("fork", "pipe", "th", "tail")
plans = [dict(kv.split("=") for kv in p.split(",")) for p in a.plans.split(";")]
eng = r.Engine(r.rsr.builtin("imagenet"), device=0, precision=a.prec)
eng.set_experiment("forktune", "0")


def select(p):
    for k in KEYS:
        eng.set_experiment(k, p.get(k, ""))


for size in a.sizes.split(","):
    H, W = map(int, size.split("x"))
    x = torch.from_numpy(synth_u8(2, H, W)).cuda()[None]
    outs, times = [], [[] for _ in plans]
    for p in plans:
        select(p)
        o = eng.upscale_rgba8_dev(x)
        for _ in range(5):
            eng.upscale_rgba8_dev(x, out=o)
        outs.append(o)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for i, p in enumerate(plans):
            select(p)
            for _ in range(3):
                eng.upscale_rgba8_dev(x, out=outs[i])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                eng.upscale_rgba8_dev(x, out=outs[i])
            torch.cuda.synchronize()
            times[i].append((time.perf_counter() - t0) / a.steps * 1e3)
    med = [float(np.median(t)) for t in times]
    print(json.dumps({"prec": a.prec, "image": [H, W], "rounds_of_tiles": round(math.ceil(W / 32) * math.ceil(H / 8) / 512, 2),
                      "ms": {a.plans.split(";")[i]: round(m, 4) for i, m in enumerate(med)}, "best": a.plans.split(";")[int(np.argmin(med))],
                      "same_bytes": all(bool(torch.equal(o, outs[0])) for o in outs)}), flush=True)
    del outs, x
    torch.cuda.empty_cache()
select({})
