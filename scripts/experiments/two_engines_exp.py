#!/usr/bin/env python3
"""Host-pointer call after several contexts have come and gone in the process: does the stream -> hardware queue
mapping (and with it SDMA vs blit-kernel copies) change the wall time?"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rusty_sr_amd as r
from rusty_sr_amd.engine import host_alloc
from conftest import synth_u8
params = r.rsr.builtin("imagenet")
px = synth_u8(9, 1, 1080, 1920)
pin_in = host_alloc(px.shape); pin_in.array[...] = px
pin_out = host_alloc((1, 3240, 5760, 4))
def run(eng, reps=16):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); eng.upscale_rgba8(pin_in.array, out=pin_out.array); ts.append(time.perf_counter() - t0)
    return "min %.3f median %.3f" % (1e3 * min(ts[2:]), 1e3 * float(np.median(ts[2:])))
keep = []
for i in range(6):
    prec = "split_f16" if i % 2 == 0 else "f32"
    e = r.Engine(params, precision=prec)
    print(f"context {i} ({prec}, {len(keep)} older contexts alive): {run(e)}", flush=True)
    if i in (1, 2): keep.append(e)
    else: e.close()
if len(sys.argv) > 1:
    import torch
    torch.zeros(1, device="cuda"); s = [torch.cuda.Stream() for _ in range(3)]
    for i in range(3):
        e = r.Engine(params, precision="split_f16"); print(f"after torch + 3 torch streams, context {i}: {run(e)}", flush=True); e.close()
