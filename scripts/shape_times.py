#!/usr/bin/env python3
"""Wall time per device-resident call (u8 in, RGBA8 out) for the named configurations under several tile plans.
    python scripts/shape_times.py [prec] [reps]
Plans: "" automatic (pipe form, 8-row tiles ended by 4-row tiles; small launches: first form, 4-row tiles), th=8 / th=4 one
class only on the pipe form, tail=T other lengths of the 4-row tail.  One JSON line per (shape, plan)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rusty_sr_amd as r  # noqa: E402
from bench import synth_u8  # noqa: E402

FLOP_PER_PX = 260352


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    eng = r.Engine(r.rsr.builtin("imagenet"), device=0, precision=prec)
    shapes = [("A 256x256", 1, 256, 256), ("512x512", 1, 512, 512), ("B 1920x1080", 1, 1080, 1920), ("D 64x512x512", 64, 512, 512),
              ("2560x1440", 1, 1440, 2560)]
    plans = [("", "", ""), ("8", "", "all"), ("4", "", "all"), ("", "0", ""), ("", "3", "")]  # (th, tail, pipe)
    for name, n, h, w in shapes:
        px = torch.from_numpy(synth_u8(2, h, w, n=n) if n > 1 else synth_u8(2, h, w)[None]).cuda()
        out = eng.upscale_rgba8_dev(px)
        for th, tail, pipe in plans:
            eng.set_experiment("th", th)
            eng.set_experiment("tail", tail)
            eng.set_experiment("pipe", pipe)
            k = max(reps, int(40e6 / (n * h * w)) if n * h * w < 1e6 else reps)
            for _ in range(3):
                eng.upscale_rgba8_dev(px, out=out)
            torch.cuda.synchronize()
            best = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                for _ in range(k):
                    eng.upscale_rgba8_dev(px, out=out)
                torch.cuda.synchronize()
                best = min(best, (time.perf_counter() - t0) / k * 1e3)
            eng.set_profiling(True)
            eng.upscale_rgba8_dev(px, out=out)
            torch.cuda.synchronize()
            st = eng.last_timing()["stage_ms"]
            eng.set_profiling(False)
            print(json.dumps({"shape": name, "prec": prec, "th": th or "auto", "tail": tail or "auto", "pipe": pipe or "auto", "ms": round(best, 4),
                              "tflops": round(n * h * w * FLOP_PER_PX / (best / 1e3) / 1e12, 1),
                              "stage_ms": [round(v, 4) for v in st]}), flush=True)
        del out, px
    for key in ("th", "tail", "pipe"):
        eng.set_experiment(key, "")


if __name__ == "__main__":
    main()
