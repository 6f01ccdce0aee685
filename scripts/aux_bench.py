#!/usr/bin/env python3
"""Timing of the two parameter-free graphs (bilinear_net / downsample_net, sr_aux.hip) on device-resident images:
ms per call, GB/s of compulsory I/O, fraction of the HBM roof.  One JSON line per (graph, io, size).
    python scripts/aux_bench.py [--sizes 1080x1920,2160x3840] [--reps 200]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


from bench import aux_entries  # noqa: E402  (bench.py's `aux_graphs` entry is the same measurement)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1080x1920,3240x5760")
    ap.add_argument("--reps", type=int, default=200)
    a = ap.parse_args()
    import torch
    import rusty_sr_amd as r
    for e in aux_entries(r, torch, [tuple(map(int, s.split("x"))) for s in a.sizes.split(",")], a.reps):
        print(json.dumps(e), flush=True)
