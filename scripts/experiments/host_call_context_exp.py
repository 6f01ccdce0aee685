#!/usr/bin/env python3
"""What in a process's history changes the cost of the split-half 1080p host-pointer call?  (bench.py's `host_call` reads 3.37 ms where a
fresh process reads 2.17 with the same plan.)  Each variant in a FRESH context of this one process: some activity first, then the call.
    python scripts/experiments/host_call_context_exp.py [prec]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rusty_sr_amd as r  # noqa: E402
from rusty_sr_amd.engine import host_alloc  # noqa: E402
from bench import synth_u8  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "split_f16"
PLAN = sys.argv[2] if len(sys.argv) > 2 else ""   # sr_set_experiment("rows", ...): "" the automatic plan
H, W = 1080, 1920
px = synth_u8(2, H, W)
pin_in, pin_out = host_alloc((H, W, 3)), host_alloc((3 * H, 3 * W, 4))
pin_in.array[...] = px
dpx = torch.from_numpy(px).cuda()[None]
params = r.rsr.builtin("imagenet")


def host_ms(eng):
    for _ in range(4):
        eng.upscale_rgba8(pin_in.array, out=pin_out.array)
    per = []
    for _ in range(20):
        t0 = time.perf_counter()
        eng.upscale_rgba8(pin_in.array, out=pin_out.array)
        per.append((time.perf_counter() - t0) * 1e3)
    t = eng.last_timing()
    return {"ms": round(float(np.median(per)), 4), "kernel_ms": round(t["total_ms"], 3), "d2h_ms": round(t["d2h_ms"], 3)}


def variant(name, before):
    eng = r.Engine(params, device=0, precision=prec)
    before(eng)
    torch.cuda.synchronize()
    eng.set_experiment("rows", PLAN)
    print(json.dumps({"prec": prec, "plan": PLAN or "auto", "before": name, **host_ms(eng)}), flush=True)
    eng.close()


def dev_calls(eng):
    o = eng.upscale_rgba8_dev(dpx)
    for _ in range(20):
        eng.upscale_rgba8_dev(dpx, out=o)


def forked(eng):
    eng.set_experiment("fork", "1")
    dev_calls(eng)
    eng.set_experiment("fork", "")


def profiled(eng):
    eng.set_profiling(True)
    eng.upscale_rgba8_dev(dpx)
    torch.cuda.synchronize()
    eng.last_timing()
    eng.set_profiling(False)


def side_stream(eng):
    s = torch.cuda.Stream()
    o = eng.upscale_rgba8_dev(dpx)
    for _ in range(10):
        eng.upscale_rgba8_dev(dpx, out=o, stream=s)


def pageable_copy(eng):
    dpx.cpu()
    torch.from_numpy(px).cuda()


def one_chunk_first(eng):
    eng.set_pipeline(False)
    eng.upscale_rgba8(pin_in.array, out=pin_out.array)
    eng.set_pipeline(True)


def small_host_first(eng):
    a, b = host_alloc((480, 640, 3)), host_alloc((1440, 1920, 4))
    for _ in range(5):
        eng.upscale_rgba8(a.array, out=b.array)
    a.close(); b.close()


for name, fn in (("nothing", lambda e: None), ("device calls", dev_calls), ("forked device calls", forked), ("a profiled device call", profiled),
                 ("device calls on a side stream", side_stream), ("pageable torch copies", pageable_copy), ("a one-chunk host call", one_chunk_first),
                 ("small host calls", small_host_first), ("nothing again", lambda e: None)):
    variant(name, fn)
