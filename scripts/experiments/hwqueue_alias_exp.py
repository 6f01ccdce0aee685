#!/usr/bin/env python3
"""Does the number of HIP streams alive in the process change what a host-pointer call costs?  HIP maps streams onto a few hardware queues
(GPU_MAX_HW_QUEUES, 4 by default); streams that share one are serialised, so a download stream that lands on the compute stream's queue
hides nothing.  For k extra torch streams alive (each used once): a fresh context, the 1080p host call under the automatic plan and under
equal bands on alternating streams.  One JSON line per (precision, k).
    python scripts/experiments/hwqueue_alias_exp.py [prec] [k0,k1,...]"""
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
ks = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "0,2,4,6,9").split(",")]
H, W = 1080, 1920
pin_in, pin_out = host_alloc((H, W, 3)), host_alloc((3 * H, 3 * W, 4))
pin_in.array[...] = synth_u8(2, H, W)
keep = []
for k in ks:
    while len(keep) < k:
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            torch.zeros(16, device="cuda").add_(1)
        keep.append(s)
    torch.cuda.synchronize()
    eng = r.Engine(r.rsr.builtin("imagenet"), device=0, precision=prec)
    res = {"prec": prec, "extra_streams": k}
    nb = H // (176 if prec == "split_f16" else 256)
    eq = ",".join(str((H * (i + 1)) // nb - (H * i) // nb) for i in range(nb))
    for name, plan in (("auto", ""), ("equal_alternating", "=" + eq), ("auto_again", "")):
        eng.set_experiment("rows", plan)
        for _ in range(4):
            eng.upscale_rgba8(pin_in.array, out=pin_out.array)
        per = []
        for _ in range(20):
            t0 = time.perf_counter()
            eng.upscale_rgba8(pin_in.array, out=pin_out.array)
            per.append((time.perf_counter() - t0) * 1e3)
        t = eng.last_timing()
        res[name] = {"ms": round(float(np.median(per)), 4), "kernel_ms": round(t["total_ms"], 3), "d2h_ms": round(t["d2h_ms"], 3)}
    print(json.dumps(res), flush=True)
    eng.close()
