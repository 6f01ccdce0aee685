#!/usr/bin/env python3
"""Host-pointer call in a CLEAN context and after pageable torch copies in the process (which, on this stack, make some plans' calls ~1 ms
longer on the host side while every GPU-side time stays the same: scripts/experiments/host_call_context_exp.py).  One JSON line per case.
    python scripts/experiments/host_call_dirty_exp.py case [case ...]      case = prec:io:HxW:plan   (plan "" = automatic)"""
import ctypes as C
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

params = r.rsr.builtin("imagenet")
fp = C.POINTER(C.c_float)


def measure(prec, io, H, W, plan, dirty):
    px = synth_u8(2, H, W)
    if io == "f32":
        pin_in, pin_out = host_alloc((H, W, 3), np.float32), host_alloc((3 * H, 3 * W, 3), np.float32)
        pin_in.array[...] = r.img_to_data(px)
    else:
        pin_in, pin_out = host_alloc((H, W, 3)), host_alloc((3 * H, 3 * W, 4))
        pin_in.array[...] = px
    eng = r.Engine(params, device=0, precision=prec)
    if dirty:
        t = torch.from_numpy(px).cuda()
        t.cpu()
        torch.cuda.synchronize()
    eng.set_experiment("rows", plan)
    if io == "f32":
        call = lambda: r._lib.check(eng._L.sr_upscale_f32(eng._ctx, pin_in.array.ctypes.data_as(fp), 1, H, W, pin_out.array.ctypes.data_as(fp)), eng._ctx)
    else:
        call = lambda: eng.upscale_rgba8(pin_in.array, out=pin_out.array)
    for _ in range(4):
        call()
    per = []
    for _ in range(15):
        t0 = time.perf_counter()
        call()
        per.append((time.perf_counter() - t0) * 1e3)
    tm = eng.last_timing()
    eng.close()
    pin_in.close(); pin_out.close()
    return round(float(np.median(per)), 4), round(tm["total_ms"], 3), round(tm["d2h_ms"], 3)


for case in sys.argv[1:]:
    prec, io, size, plan = case.split(":")
    H, W = map(int, size.split("x"))
    clean = measure(prec, io, H, W, plan, False)
    dirty = measure(prec, io, H, W, plan, True)
    print(json.dumps({"prec": prec, "io": io, "image": [H, W], "plan": plan or "auto", "clean_ms": clean[0], "dirty_ms": dirty[0],
                      "kernel_ms": [clean[1], dirty[1]], "d2h_ms": [clean[2], dirty[2]]}), flush=True)
