#!/usr/bin/env python3
"""Device-resident 1080p / 4K frame as k row bands on two streams (two contexts, one workspace each) against the undivided
pass: does the overlap of one band's stage tails with the other band's stage heads pay for the 14 recomputed rows?"""
import sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import torch
import rusty_sr_amd as r
from conftest import synth_u8
params = r.rsr.builtin("imagenet")
for prec in ("f32", "split_f16"):
    e = [r.Engine(params, precision=prec) for _ in range(2)]
    st = [torch.cuda.Stream(), torch.cuda.Stream()]
    for (h, w) in ((1080, 1920), (2160, 3840)):
        px = torch.from_numpy(synth_u8(2, 1, h, w)).cuda()
        out = torch.empty((1, 3 * h, 3 * w, 4), dtype=torch.uint8, device="cuda")
        want = e[0].upscale_rgba8_dev(px).clone()
        def undivided():
            e[0].upscale_rgba8_dev(px, out=out)
        def banded(k):
            cuts = [(h * i // k) // 8 * 8 for i in range(k)] + [h]
            def run():
                for i in range(k):
                    y0, y1 = cuts[i], cuts[i + 1]
                    a, b = max(0, y0 - 7), min(h, y1 + 7)
                    with torch.cuda.stream(st[i % 2]):
                        e[i % 2].upscale_band_rgba8_dev(px[0, a:b], y0 - a, b - y1, out=out[0, 3 * y0:3 * y1], stream=st[i % 2])
            return run
        def timeit(fn, reps=30):
            for _ in range(5): fn()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(reps): fn()
            torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
        base = timeit(undivided)
        line = f"{prec:9s} {w}x{h}: undivided {base:.3f} ms"
        for k in (2, 3, 4, 6):
            fn = banded(k)
            out.zero_(); fn(); torch.cuda.synchronize()
            ok = bool(torch.equal(out, want))
            line += f" | {k} bands / 2 streams {timeit(fn):.3f} ms{'' if ok else ' (MISMATCH)'}"
        print(line, flush=True)
    for x in e: x.close()
