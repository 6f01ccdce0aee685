#!/usr/bin/env python3
"""A few host-pointer calls (page-locked buffers) for a profiler to look at: which engine moves the pixels?"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rusty_sr_amd as r
from rusty_sr_amd.engine import host_alloc
from conftest import synth_u8
prec = sys.argv[1] if len(sys.argv) > 1 else "split_f16"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
eng = r.Engine(r.rsr.builtin("imagenet"), precision=prec)
px = synth_u8(9, 1, 1080, 1920)
pin_in = host_alloc(px.shape); pin_in.array[...] = px
pin_out = host_alloc((1, 3240, 5760, 4))
ts = []
for _ in range(reps):
    t0 = time.perf_counter(); eng.upscale_rgba8(pin_in.array, out=pin_out.array); ts.append(time.perf_counter() - t0)
print(prec, "wall min %.3f median %.3f ms" % (1e3 * min(ts[2:]), 1e3 * float(np.median(ts[2:]))), eng.last_timing())
