#!/usr/bin/env python3
"""Host-pointer call (sr_upscale_rgba8, page-locked buffers): geometric band plan against equal bands, and the cost of a
geometry change (a context that alternates between two image sizes)."""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import rusty_sr_amd as r
from rusty_sr_amd.engine import host_alloc
from conftest import synth_u8
params = r.rsr.builtin("imagenet")
for prec in ("f32", "split_f16"):
    eng = r.Engine(params, precision=prec)
    for (h, w) in ((1080, 1920), (2160, 3840), (1440, 2560)):
        px = synth_u8(9, 1, h, w)
        pin_in = host_alloc(px.shape); pin_in.array[...] = px
        pin_out = host_alloc((1, 3 * h, 3 * w, 4))
        for label, bands, geo in (("equal4", "4", "1"), ("equal3", "3", "1"), ("equal2", "2", "1"), ("auto-geo", "", "1"), ("auto-equal", "", "0"), ("auto-geo", "", "1")):
            eng.set_experiment("bands", bands); eng.set_experiment("geo", geo)
            ts = []
            for it in range(14):
                t0 = time.perf_counter(); eng.upscale_rgba8(pin_in.array, out=pin_out.array); ts.append(time.perf_counter() - t0)
            t = eng.last_timing()
            print(f"{prec:9s} {w}x{h} {label:>10s}  wall min {1e3*min(ts[2:]):7.3f} median {1e3*np.median(ts[2:]):7.3f} ms   kernels {t['total_ms']:7.3f} h2d {t['h2d_ms']:.3f} d2h {t['d2h_ms']:.3f} chunks {t.get('chunks')}", flush=True)
        pin_in.close(); pin_out.close()
    eng.set_experiment("bands", ""); eng.set_experiment("geo", "1")
    # geometry change: device-resident calls alternating between a 4K frame and a thumbnail
    big = torch.from_numpy(synth_u8(1, 1, 2160, 3840)).cuda(); small = torch.from_numpy(synth_u8(2, 1, 256, 256)).cuda()
    ob, os_ = eng.upscale_rgba8_dev(big), eng.upscale_rgba8_dev(small)
    def run(seq, reps=10):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            for x, o in seq: eng.upscale_rgba8_dev(x, out=o)
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
    tb, ts_, both = run([(big, ob)]), run([(small, os_)], 50), run([(big, ob), (small, os_)])
    print(f"{prec:9s} geometry change: 4K alone {tb:.3f} ms, 256x256 alone {ts_:.3f} ms, alternating pair {both:.3f} ms (sum {tb+ts_:.3f})", flush=True)
    eng.close()
