#!/usr/bin/env python3
"""Digest of what a build of libsrhip computes, for bit-for-bit A/B of two builds:
    SRHIP_LIB=/path/to/old/libsrhip.so python scripts/ab_digest.py > a.txt
    python scripts/ab_digest.py > b.txt ; diff a.txt b.txt
One line per (precision, factor, shape, plan): sha1 of the f32 output, of the RGBA8 output and of the four feature
maps.  Plans cover both kernel forms and both tile classes.  A kernel restructuring that claims "same arithmetic, same
order" must leave every line as it was."""
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rusty_sr_amd as r  # noqa: E402
from rusty_sr_amd import _lib  # noqa: E402
from bench import synth_u8  # noqa: E402


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def main():
    params = r.rsr.builtin("imagenet")
    shapes = [(1, 37, 53), (1, 250, 2080), (2, 333, 640), (1, 1080, 1920)]
    plans = [("", "", ""), ("8", "", "all"), ("4", "", "all"), ("", "0.3", "all"), ("", "", "none")]  # (th, tail, pipe)
    for prec in ("f32", "split_f16"):
        for factor in (3, 4, 2):
            if factor == 3:
                eng = r.Engine(params, device=0, precision=prec)
            else:
                rng = np.random.default_rng(factor)
                eng = r.Engine((rng.standard_normal(_lib.lib().sr_num_params_factor(factor)) * 0.05).astype(np.float32), device=0, precision=prec,
                               factor=factor)
            eng.set_experiment("fork", "0")  # (a forked call keeps its bands' maps in two workspaces: sr_read_feature refuses)
            for n, h, w in shapes[: 4 if factor == 3 else 2]:
                px = synth_u8(5, h, w, n=n) if n > 1 else synth_u8(5, h, w)[None]
                dpx = torch.from_numpy(px).cuda()
                dx = (dpx.float() / 255.0).contiguous()
                for th, tail, pipe in plans:
                    eng.set_experiment("th", th)
                    eng.set_experiment("tail", tail)
                    eng.set_experiment("pipe", pipe)
                    u8 = eng.upscale_rgba8_dev(dpx).cpu().numpy()
                    f = eng.upscale_f32_dev(dx).cpu().numpy()
                    feats = " ".join(sha(eng.read_feature(k, h, w)) for k in range(4)) if n == 1 and h * w <= 520000 else "-"
                    print(prec, factor, f"{n}x{h}x{w}", f"th={th or 'auto'} tail={tail or 'auto'} pipe={pipe or 'auto'}", sha(f), sha(u8), feats,
                          flush=True)
            del eng
    if "--time" in sys.argv:  # lines starting with "time" are measurements, not digests: grep -v ^time before diffing
        import time
        for prec in ("f32", "split_f16"):
            eng = r.Engine(params, device=0, precision=prec)
            for name, n, h, w in (("A", 1, 256, 256), ("B", 1, 1080, 1920), ("band8", 1, 270 + 14, 3840), ("D", 64, 512, 512)):
                dpx = torch.from_numpy(synth_u8(2, h, w, n=n) if n > 1 else synth_u8(2, h, w)[None]).cuda()
                out = eng.upscale_rgba8_dev(dpx)
                k = 400 if n * h * w < 1e6 else 30
                for _ in range(k // 4):
                    eng.upscale_rgba8_dev(dpx, out=out)
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(5):
                    t0 = time.perf_counter()
                    for _ in range(k):
                        eng.upscale_rgba8_dev(dpx, out=out)
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) / k * 1e3)
                eng.set_profiling(True)
                eng.upscale_rgba8_dev(dpx, out=out)
                torch.cuda.synchronize()
                st = eng.last_timing()["stage_ms"]
                eng.set_profiling(False)
                print("time", prec, name, round(best, 4), [round(v, 4) for v in st], flush=True)
            del eng


if __name__ == "__main__":
    main()
