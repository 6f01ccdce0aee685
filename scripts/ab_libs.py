#!/usr/bin/env python3
"""Within-probe A/B of libsrhip builds (scripts/build_variant.sh): interleaved rounds, one fresh process per
(library, round), per-stage medians of HIP-event times on the device-resident 1080p (or HxW) workload.
    python scripts/ab_libs.py [--prec split_f16] [--rounds 3] [--hw 1080x1920] lib1.so lib2.so ..."""
import argparse
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, sys, numpy as np, torch
sys.path.insert(0, %r)
import rusty_sr_amd as r
from bench import synth_u8
prec, H, W, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
eng = r.Engine(r.rsr.builtin("imagenet"), device=0, precision=prec)
px = torch.from_numpy(synth_u8(2, H, W)).cuda()[None]
out = eng.upscale_rgba8_dev(px)
for _ in range(10):
    eng.upscale_rgba8_dev(px, out=out)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(reps):
    eng.upscale_rgba8_dev(px, out=out)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / reps * 1e3
eng.set_profiling(True)
acc = []
for _ in range(reps):
    eng.upscale_rgba8_dev(px, out=out); torch.cuda.synchronize(); acc.append(eng.last_timing()["stage_ms"])
print(json.dumps({"stages": np.median(np.array(acc), axis=0).tolist(), "wall": wall}))
''' % ROOT

ap = argparse.ArgumentParser()
ap.add_argument("--prec", default="split_f16")
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--hw", default="1080x1920")
ap.add_argument("libs", nargs="+")
a = ap.parse_args()
H, W = map(int, a.hw.split("x"))
res = {l: [] for l in a.libs}
for rnd in range(a.rounds):
    for lib in a.libs:
        path, *envs = lib.split("@")  # lib.so@SRHIP_TAIL=0@SRHIP_BW=8 ...
        env = dict(os.environ, SRHIP_LIB=os.path.abspath(path), **dict(e.split("=", 1) for e in envs))
        r = subprocess.run([sys.executable, "-c", CHILD, a.prec, str(H), str(W), str(a.reps)], env=env, capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            print(lib, "FAILED", r.stderr[-500:])
            continue
        d = json.loads(r.stdout.strip().splitlines()[-1])
        res[lib].append(d["stages"] + [sum(d["stages"]), d["wall"]])
for lib, v in res.items():
    if v:
        m = np.median(np.array(v), axis=0)
        mn = np.min(np.array(v), axis=0)
        print(f"{a.prec} {a.hw} {os.path.basename(lib):40s} stages {' '.join(f'{x:7.4f}' for x in m[:5])}  sum {m[5]:.4f}  wall {m[6]:.4f}  (min sum {mn[5]:.4f})", flush=True)
