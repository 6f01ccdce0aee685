#!/usr/bin/env python3
"""Host-pointer call (sr_upscale_rgba8, page-locked buffers): wall time against the number of row bands of the host pipeline."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import rusty_sr_amd as r
from rusty_sr_amd.engine import host_alloc
from conftest import synth_u8
params = r.rsr.builtin("imagenet")
for prec in ("f32", "split_f16"):
    eng = r.Engine(params, precision=prec)
    for (h, w) in ((1080, 1920), (2160, 3840)):
        px = synth_u8(9, 1, h, w)
        pin_in = host_alloc(px.shape); pin_in.array[...] = px
        pin_out = host_alloc((1, 3 * h, 3 * w, 4))
        for bands in ("1", "2", "3", "4", "5", "6", "8", "12", ""):
            eng.set_experiment("bands", bands)
            ts = []
            for it in range(12):
                t0 = time.perf_counter(); eng.upscale_rgba8(pin_in.array, out=pin_out.array); ts.append(time.perf_counter() - t0)
            t = eng.last_timing()
            print(f"{prec:9s} {w}x{h} bands={bands or 'auto':>4s}  wall min {1e3*min(ts[2:]):7.3f} median {1e3*np.median(ts[2:]):7.3f} ms   kernels {t['total_ms']:7.3f} h2d {t['h2d_ms']:.3f} d2h {t['d2h_ms']:.3f}", flush=True)
        pin_in.close(); pin_out.close()
    eng.close()
