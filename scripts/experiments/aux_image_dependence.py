#!/usr/bin/env python3
"""How much of the u8 bilinear graph's time depends on the image?  Its quantiser is a table look-up at a data-dependent bucket per
output sample (sr_aux.hip): lanes that hold different buckets of the same LDS bank serialise, lanes that hold the same bucket do not.
Times bilinear_net (u8 in, RGBA8 out) on: the bench's image (5x5 box-smoothed noise), a constant image (every look-up of a wave hits one
address: no conflict at all), a horizontal ramp (neighbouring lanes in neighbouring buckets), raw noise (the worst case).
    python scripts/experiments/aux_image_dependence.py [--size 1080x1920] [--reps 300]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bench import synth_u8  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", default="1080x1920")
    ap.add_argument("--reps", type=int, default=300)
    a = ap.parse_args()
    H, W = map(int, a.size.split("x"))
    import torch
    import rusty_sr_amd as r
    rng = np.random.default_rng(3)
    images = {
        "bench (5x5 box-smoothed noise)": synth_u8(7, H, W),
        "constant 128": np.full((H, W, 3), 128, np.uint8),
        "horizontal ramp": np.broadcast_to((np.arange(W) * 255 // max(W - 1, 1)).astype(np.uint8)[None, :, None], (H, W, 3)).copy(),
        "raw noise": rng.integers(0, 256, (H, W, 3), dtype=np.uint8),
    }
    for graph in ("bilinear", "downsample"):
        eng = r.Engine(graph=graph, device=0)
        for name, px in images.items():
            x = torch.from_numpy(px).cuda()[None]
            o = eng.upscale_rgba8_dev(x)
            for _ in range(10):
                eng.upscale_rgba8_dev(x, out=o)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(a.reps):
                    eng.upscale_rgba8_dev(x, out=o)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / a.reps * 1e6)
            print(json.dumps({"graph": graph, "image": name, "size": [H, W], "us": round(best, 2)}), flush=True)
        eng.close()
