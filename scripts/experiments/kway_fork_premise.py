#!/usr/bin/env python3
"""Would a K-way fork (K > 2 row bands of a lone mid-size frame on K streams) beat the library's two-way fork?  Priced without building it:
K contexts, K streams, each computes one band (with its 7-row halos) through sr_upscale_band_rgba8_dev into its rows of ONE output --
what a K-way sr_run_stack_auto would launch, minus its event fork / join (the K calls are simply queued on K streams; every burst is
fenced).  Against: the library's own undivided and two-way calls.  One JSON line per shape.
    python scripts/experiments/kway_fork_premise.py [--prec f32] [--sizes 320x320,...]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rusty_sr_amd as r  # noqa: E402
from bench import synth_u8  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--prec", default="f32")
ap.add_argument("--sizes", default="320x320,448x448,576x576,360x640,480x854,576x720")
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=40)
a = ap.parse_args()
params = r.rsr.builtin("imagenet")
KMAX = 4
engs = [r.Engine(params, device=0, precision=a.prec) for _ in range(KMAX)]
for e in engs:
    e.set_experiment("forktune", "0")
streams = [torch.cuda.Stream() for _ in range(KMAX)]

for size in a.sizes.split(","):
    H, W = map(int, size.split("x"))
    x = torch.from_numpy(synth_u8(2, H, W)).cuda()
    ref = None
    variants = {}

    def lib(fork):
        def run(out):
            engs[0].set_experiment("fork", fork)
            engs[0].upscale_rgba8_dev(x[None], out=out[None])
        return run

    def kway(K):
        cuts = [round(H * k / K) for k in range(K + 1)]
        pieces = []
        for k in range(K):
            top = 7 if k > 0 else 0
            bot = 7 if k < K - 1 else 0
            pieces.append((x[cuts[k] - top:cuts[k + 1] + bot].contiguous(), top, bot, cuts[k], cuts[k + 1]))

        def run(out):
            for k, (band, top, bot, y0, y1) in enumerate(pieces):
                engs[k].set_experiment("fork", "0")
                engs[k].upscale_band_rgba8_dev(band, top, bot, out=out[3 * y0:3 * y1], stream=streams[k])
        return run

    variants = {"undivided": lib("0"), "lib_2way": lib("1"), "k2": kway(2), "k3": kway(3), "k4": kway(4)}
    outs, times = {}, {v: [] for v in variants}
    for v, fn in variants.items():
        outs[v] = torch.empty((3 * H, 3 * W, 4), dtype=torch.uint8, device="cuda")
        for _ in range(5):
            fn(outs[v])
        torch.cuda.synchronize()
    for _ in range(a.rounds):
        for v, fn in variants.items():
            for _ in range(3):
                fn(outs[v])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                fn(outs[v])
            torch.cuda.synchronize()
            times[v].append((time.perf_counter() - t0) / a.steps * 1e3)
    print(json.dumps({"prec": a.prec, "image": [H, W], **{v: round(float(np.median(t)), 4) for v, t in times.items()},
                      "same_bytes": all(bool(torch.equal(o, outs["undivided"])) for o in outs.values())}), flush=True)
