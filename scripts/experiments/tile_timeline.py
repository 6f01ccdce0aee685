#!/usr/bin/env python3
"""Per-tile phase times of the stage-3 and last-stage pipe kernels from a scratch build with -DSR_TIMELINE=1 (never the shipped build: it
stamps the shader clock at every phase boundary from wave 0 of the first 128 workgroups; the patch is in profiles/r6_tile_timeline.txt's header).
    SRHIP_LIB=.../libsrhip_timeline.so SRHIP_TIMELINE_OUT=/tmp/tl.bin python scripts/experiments/tile_timeline.py [f32|split_f16] [HxW]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import rusty_sr_amd as r  # noqa: E402
from bench import synth_u8  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
H, W = map(int, (sys.argv[2] if len(sys.argv) > 2 else "1080x1920").split("x"))
out_path = os.environ["SRHIP_TIMELINE_OUT"]
eng = r.Engine(r.rsr.builtin("imagenet"), device=0, precision=prec)
eng.set_experiment("fork", "0")
px = torch.from_numpy(synth_u8(2, H, W)).cuda()[None]
out = eng.upscale_rgba8_dev(px)
for _ in range(3):
    eng.upscale_rgba8_dev(px, out=out)
torch.cuda.synchronize()
eng.close()  # dumps the stamps of the LAST call
buf = np.fromfile(out_path, dtype=np.uint64).reshape(2, 128, 512)
for which, name in ((0, "stage 3 (f 5x5, l1 3x3, l2 3x3)"), (1, "last stage (l1, l2, l3 3x3 + residual taps + depth-to-space)")):
    tiles = []
    for wg in range(128):
        n = int(buf[which, wg, 0])
        if n < 18:
            continue
        st = buf[which, wg, 1:1 + n].astype(np.int64).reshape(-1, 9)  # start, after half 0..5, before epilogue (taps done), after epilogue
        tiles.append(st)
    if not tiles:
        print(name, ": no stamps")
        continue
    d = np.concatenate([np.diff(t, axis=1) for t in tiles])          # 8 phases per tile
    gap = np.concatenate([t[1:, 0] - t[:-1, 8] for t in tiles if len(t) > 1])  # end of a tile's epilogue -> the next tile's first stamp
    whole = np.concatenate([t[1:, 0] - t[:-1, 0] for t in tiles if len(t) > 1])
    lab = ["half 0", "half 1", "half 2", "half 3", "half 4", "half 5", "taps (last stage) / -", "epilogue"]
    print(f"== {name}: {sum(len(t) for t in tiles)} tiles of {len(tiles)} workgroups (8-row and 4-row tiles alike), shader-clock cycles, median [p10 .. p90]")
    for k in range(8):
        print(f"   {lab[k]:24s} {np.median(d[:, k]):9.0f}  [{np.percentile(d[:, k], 10):8.0f} .. {np.percentile(d[:, k], 90):8.0f}]")
    print(f"   {'tile to tile gap':24s} {np.median(gap):9.0f}  [{np.percentile(gap, 10):8.0f} .. {np.percentile(gap, 90):8.0f}]")
    print(f"   {'tile period':24s} {np.median(whole):9.0f}  [{np.percentile(whole, 10):8.0f} .. {np.percentile(whole, 90):8.0f}]")
