#!/usr/bin/env python3
"""Soak: random shapes, batches and band halos through the library's own plan AND through the pipe form under random tile
plans (8-row tiles, 4-row tiles, tails of any length, any tile order) against the first-form kernels (an
independent code path, bit-identical by construction), interleaved so that every call changes the workspace geometry, the
host-pointer entry point (chunks and bands on two streams) against the device call, plus
repeated 1080p calls that must reproduce themselves bit for bit (a missed vmcnt / barrier shows up as a flicker).
    python scripts/soak.py [seconds]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def soak(budget, seed=None):
    import torch
    import rusty_sr_amd as r
    params = r.rsr.builtin("imagenet")
    rng = np.random.default_rng(int(time.time()) & 0xffff if seed is None else seed)
    t_end = time.time() + budget
    stats = {"shapes": 0, "repeats": 0, "bands": 0, "host_calls": 0, "sharded": 0}
    bad = []
    for prec in ("f32", "split_f16"):
        a, b, c = r.Engine(params, precision=prec), r.Engine(params, precision=prec), r.Engine(params, precision=prec)
        b.set_experiment("pipe", "none")
        b.set_experiment("fork", "0")    # ... undivided: the reference every other path is compared with
        c.set_experiment("pipe", "all")  # the persistent pipe form on every launch, with a tile plan drawn per call
        group = [r.Engine(params, precision=prec) for _ in range(5)]   # one image sharded over k contexts (local transport)
        group_k = 0
        big = torch.from_numpy(rng.integers(0, 256, (1, 1080, 1920, 3), dtype=np.uint8)).cuda()
        first = a.upscale_rgba8_dev(big).clone()
        t_prec = time.time() + (t_end - time.time()) / (2 if prec == "f32" else 1)
        while time.time() < t_prec:
            n = int(rng.choice([1, 1, 1, 2, 3]))
            h, w = int(rng.integers(1, 700)), int(rng.integers(1, 1000))
            if rng.random() < (0.6 if os.environ.get("SOAK_BIG") else 0.15):  # SOAK_BIG=1: mostly large frames (many rounds of tiles, stealing)
                h, w = int(rng.integers(700, 2400 if os.environ.get("SOAK_BIG") else 1400)), int(rng.integers(1000, 4000 if os.environ.get("SOAK_BIG") else 2200))
            if rng.random() < 0.1:
                h, w = int(rng.integers(1, 12)), int(rng.integers(1, 4000))
            px = torch.from_numpy(rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)).cuda()
            ga, gb = a.upscale_rgba8_dev(px), b.upscale_rgba8_dev(px)
            if not torch.equal(ga, gb):
                bad.append((prec, "shape", n, h, w))
            plan = (str(rng.choice(["", "", "4", "8"])), str(rng.choice(["", "0", "0.01", "0.4", "2", "7"])), str(rng.choice(["", "0", "3", "16"])),
                    str(rng.choice(["", "0", "1", "1", "17", "40"])))  # ... and the device call undivided / as two bands on two streams, at any cut
            c.set_experiment("th", plan[0]); c.set_experiment("tail", plan[1]); c.set_experiment("bw", plan[2]); c.set_experiment("fork", plan[3])
            if not torch.equal(c.upscale_rgba8_dev(px), gb):
                bad.append((prec, "pipe form, plan th/tail/bw/fork", plan, n, h, w))
            stats["shapes"] += 1
            if n == 1 and h > 30:  # a band of it with halos must equal the same rows of the whole
                y0 = int(rng.integers(7, h - 15))
                y1 = int(rng.integers(y0 + 1, h - 7))
                band = a.upscale_band_rgba8_dev(px[0, y0 - 7:y1 + 7].contiguous(), 7, 7)
                if not torch.equal(band, ga[0, 3 * y0:3 * y1]):
                    bad.append((prec, "band", h, w, y0, y1))
                stats["bands"] += 1
            if stats["shapes"] % 4 == 0:  # the host-pointer entry point: chunks / bands on two streams and two workspaces
                if rng.random() < 0.5:
                    hh, ww = int(rng.integers(500, 1500)), int(rng.integers(1100, 2600))  # large enough to be cut into bands
                    hp = rng.integers(0, 256, (1, hh, ww, 3), dtype=np.uint8)
                else:
                    hp = rng.integers(0, 256, (int(rng.integers(2, 40)), int(rng.integers(8, 300)), int(rng.integers(8, 300)), 3), dtype=np.uint8)
                want = a.upscale_rgba8_dev(torch.from_numpy(hp).cuda()).cpu().numpy()
                if not np.array_equal(a.upscale_rgba8(hp), want):
                    bad.append((prec, "host", hp.shape))
                stats["host_calls"] += 1
            if stats["shapes"] % 5 == 0 and n == 1 and h >= 40:  # sr_upscale_sharded_*_all: random uneven bands, halos by peer copy
                k = int(rng.integers(2, min(5, h // 8) + 1))
                if k != group_k:
                    r.comm_init_all(group[:k], transport="local")
                    group_k = k
                cuts = np.sort(rng.choice(np.arange(1, h // 7), size=k - 1, replace=False)) * 7
                edges = [0] + [int(c) for c in cuts] + [h]
                if min(e1 - e0 for e0, e1 in zip(edges[:-1], edges[1:])) >= 7:
                    halo = str(rng.choice(["input", "layers"]))  # the recomputed overlap, or feature rows after every stage
                    for e in group[:k]:
                        e.set_experiment("halo", halo)
                    outs = r.upscale_sharded_all(group[:k], [px[0, e0:e1].contiguous() for e0, e1 in zip(edges[:-1], edges[1:])])
                    if not torch.equal(torch.cat(outs), ga[0]):
                        bad.append((prec, "sharded", halo, h, w, edges))
                    stats["sharded"] += 1
            for _ in range(3):
                if not torch.equal(a.upscale_rgba8_dev(big), first):
                    bad.append((prec, "repeat 1080p"))
                stats["repeats"] += 1
        a.close()
        b.close()
        c.close()
        for e in group:
            e.close()
    return stats, bad


if __name__ == "__main__":
    stats, bad = soak(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
    print("soak:", stats, "mismatches:", bad[:10], flush=True)
    sys.exit(1 if bad else 0)
