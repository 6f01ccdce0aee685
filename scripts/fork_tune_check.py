#!/usr/bin/env python3
"""What the fork tuner (sr_internal.h ForkTune; sr_set_experiment "forktune") decides per shape and what it is worth: for each shape
the steady-state time per call with the tuner off (the rule alone), forced undivided, forced forked, and with the tuner on after it has
settled -- interleaved rounds in one process, bit-for-bit check of every variant.  One JSON line per (precision, shape).
    python scripts/fork_tune_check.py [--prec f32,split_f16] [--sizes 320x320,...] [--rounds 5] [--steps 40]"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--prec", default="f32,split_f16")
ap.add_argument("--sizes", default="272x272,320x320,384x384,448x448,512x512,576x576,640x480,720x576,540x960,768x768,720x1280")
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--learn", default="fenced", choices=["fenced", "queued"], help="how the tuner meets the shape: a fence after every call, or bursts of 16 calls queued back to back")
a = ap.parse_args()

import torch  # noqa: E402
import rusty_sr_amd as r  # noqa: E402
from bench import synth_u8  # noqa: E402

for prec in a.prec.split(","):
    eng = r.Engine(r.rsr.builtin("imagenet"), device=0, precision=prec)
    for size in a.sizes.split(","):
        H, W = map(int, size.split("x"))
        x = torch.from_numpy(synth_u8(2, H, W)).cuda()[None]
        variants = {"rule": ("", "0"), "undivided": ("0", "0"), "forked": ("1", "0"), "tuned": ("", "1")}
        outs, times = {}, {v: [] for v in variants}

        def select(v):
            eng.set_experiment("fork", variants[v][0])
            if v != "tuned":
                eng.set_experiment("forktune", "0")

        # the tuner learns once, fenced calls, before anything is timed; selecting "tuned" later must not make it forget
        eng.set_experiment("fork", "")
        eng.set_experiment("forktune", "1")
        tuned_out = eng.upscale_rgba8_dev(x)
        if a.learn == "fenced":
            for _ in range(12):
                eng.upscale_rgba8_dev(x, out=tuned_out)
                torch.cuda.synchronize()
        else:
            for _ in range(12):  # (a sample is read at the first call that finds it finished: about one per burst)
                for _ in range(16):
                    eng.upscale_rgba8_dev(x, out=tuned_out)
                torch.cuda.synchronize()
        learned = eng.get_experiment("forktune").strip()
        decision = learned.split()[3] if learned else "rule"
        ms_u, ms_f = (float(learned.split()[4]), float(learned.split()[5])) if learned else (0.0, 0.0)
        # the other variants run with the tuner's memory intact: "forktune" is only touched through `fork` being forced (which the tuner
        # ignores) -- so "rule" is measured on a second context
        eng_rule = r.Engine(r.rsr.builtin("imagenet"), device=0, precision=prec)
        eng_rule.set_experiment("forktune", "0")
        engines = {"rule": eng_rule, "undivided": eng, "forked": eng, "tuned": eng}
        for v in variants:
            e = engines[v]
            e.set_experiment("fork", variants[v][0])
            outs[v] = e.upscale_rgba8_dev(x)
            for _ in range(5):
                e.upscale_rgba8_dev(x, out=outs[v])
        torch.cuda.synchronize()
        for _ in range(a.rounds):
            for v in variants:
                e = engines[v]
                e.set_experiment("fork", variants[v][0])
                for _ in range(3):
                    e.upscale_rgba8_dev(x, out=outs[v])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.steps):
                    e.upscale_rgba8_dev(x, out=outs[v])
                torch.cuda.synchronize()
                times[v].append((time.perf_counter() - t0) / a.steps * 1e3)
        eng.set_experiment("fork", "")
        med = {v: float(np.median(times[v])) for v in variants}
        print(json.dumps({"prec": prec, "image": [H, W], "rounds_of_tiles": round(math.ceil(W / 32) * math.ceil(H / 8) / 512, 2),
                          "learned": a.learn, "decision": decision, "tuner_ms": [ms_u, ms_f], **{v + "_ms": round(med[v], 4) for v in variants},
                          "tuned_vs_rule": round(med["tuned"] / med["rule"], 4), "best_fixed_vs_rule": round(min(med["undivided"], med["forked"]) / med["rule"], 4),
                          "same_bytes": all(bool(torch.equal(outs[v], outs["undivided"])) for v in variants)}), flush=True)
        eng_rule.close()
        del outs, x, tuned_out
        torch.cuda.empty_cache()
    eng.close()
