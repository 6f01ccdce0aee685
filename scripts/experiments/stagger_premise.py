#!/usr/bin/env python3
"""Premise test for a staggered small-image schedule (config A, 256x256: every stage is ONE round of 512 four-row
tiles, both workgroups of a CU fill, stream and drain together).  Would a CU whose two workgroups are at DIFFERENT
phases do better?  Same total work three ways, exact f32, rgba8 in HBM:
  batch    one context, n = 2 images of (H/2) x W in one call: 512 tiles per stage launch, lockstep
  streams  two contexts on two streams, one (H/2) x W image each, calls queued back to back without a fence:
           256 tiles per launch, one workgroup per CU from either chain, the chains free to drift apart
  whole    the H x W image itself
One JSON line; `host_ms` is the time the submitting loop took (the streams leg is valid only where it is well below
the wall time).
    python scripts/experiments/stagger_premise.py [HxW] [reps]"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rusty_sr_amd as r  # noqa: E402
from bench import synth_u8  # noqa: E402

H, W = map(int, (sys.argv[1] if len(sys.argv) > 1 else "256x256").split("x"))
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
params = r.rsr.builtin("imagenet")
e1, e2 = r.Engine(params, device=0, precision="f32"), r.Engine(params, device=0, precision="f32")
for e in (e1, e2):
    e.set_experiment("fork", "0")
whole = torch.from_numpy(synth_u8(2, H, W)).cuda()[None]
halves = torch.stack([whole[0, :H // 2], whole[0, H // 2:]]).contiguous()
ha, hb = halves[0:1].contiguous(), halves[1:2].contiguous()
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, n, pre=None):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    best = None
    for _ in range(5):
        t0 = time.perf_counter()
        if pre:
            pre()
        for _ in range(n):
            fn()
        th = time.perf_counter() - t0
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        if best is None or t < best[0]:
            best = (t, th)
    return best[0] / n * 1e3, best[1] / n * 1e3


o_w = e1.upscale_rgba8_dev(whole)
o_b = e1.upscale_rgba8_dev(halves)
o_a, o_c = e1.upscale_rgba8_dev(ha), e2.upscale_rgba8_dev(hb)
res = {"image": [H, W], "reps": reps}
res["whole_ms"], res["whole_host_ms"] = timed(lambda: e1.upscale_rgba8_dev(whole, out=o_w), reps)
res["batch_ms"], res["batch_host_ms"] = timed(lambda: e1.upscale_rgba8_dev(halves, out=o_b), reps)


def two():
    e1.upscale_rgba8_dev(ha, out=o_a, stream=sa)
    e2.upscale_rgba8_dev(hb, out=o_c, stream=sb)


res["streams_ms"], res["streams_host_ms"] = timed(two, reps)
for off in (1000, 10000, 30000, 100000, 300000):  # the second chain starts late by a spin of `off` clock ticks (torch.cuda._sleep)
    def pre(off=off):
        with torch.cuda.stream(sb):
            torch.cuda._sleep(off)
    res[f"streams_offset_{off}_ms"], _ = timed(two, reps, pre)
res["one_half_alone_ms"], _ = timed(lambda: e1.upscale_rgba8_dev(ha, out=o_a, stream=sa), reps)
res["same_bytes"] = bool(torch.equal(o_b[0], o_a[0]) and torch.equal(o_b[1], o_c[0]))
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in res.items()}))
