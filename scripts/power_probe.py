#!/usr/bin/env python3
"""Board power and shader clock while the 1080p frame loops, per arithmetic mode (round 5: is the split-half mode power-bound?).
    python scripts/power_probe.py [seconds per mode] [HxW]
Samples hwmon power1_average (uW) and pp_dpm_sclk ('*' level) every 10 ms on a thread; the second half of the samples is averaged."""
import glob
import json
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import rusty_sr_amd as r  # noqa: E402
from bench import synth_u8  # noqa: E402


def first(pats):
    for p in pats:
        g = sorted(glob.glob(p))
        if g:
            return g[0]
    return None


POWER = first(["/sys/class/drm/card*/device/hwmon/hwmon*/power1_average", "/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"])
SCLK = first(["/sys/class/drm/card*/device/pp_dpm_sclk"])


class Monitor(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.stop = False
        self.w, self.f = [], []

    def run(self):
        while not self.stop:
            try:
                if POWER:
                    self.w.append(float(open(POWER).read()) * 1e-6)
                if SCLK:
                    for ln in open(SCLK).read().splitlines():
                        if "*" in ln:
                            self.f.append(float(ln.split(":")[1].lower().replace("mhz", "").replace("*", "")))
            except Exception:
                pass
            time.sleep(0.01)

    def result(self):
        self.stop = True
        self.join()
        half = lambda v: sum(v[len(v) // 2:]) / max(1, len(v) - len(v) // 2) if v else None
        return half(self.w), half(self.f)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
    H, W = map(int, (sys.argv[2] if len(sys.argv) > 2 else "1080x1920").split("x"))
    px = torch.from_numpy(synth_u8(2, H, W)).cuda()[None]
    print(json.dumps({"power_path": POWER, "sclk_path": SCLK}))
    m = Monitor(); m.start(); time.sleep(1.0)
    w, f = m.result()
    print(json.dumps({"mode": "idle", "watts": w, "sclk_mhz": f}))
    for prec in ("f32", "split_f16"):
        eng = r.Engine(r.rsr.builtin("imagenet"), device=0, precision=prec)
        out = eng.upscale_rgba8_dev(px)
        for _ in range(20):
            eng.upscale_rgba8_dev(px, out=out)
        torch.cuda.synchronize()
        m = Monitor(); m.start()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < secs:
            for _ in range(50):
                eng.upscale_rgba8_dev(px, out=out)
            torch.cuda.synchronize()
            n += 50
        dt = time.perf_counter() - t0
        w, f = m.result()
        print(json.dumps({"mode": prec, "hw": [H, W], "ms_per_frame": dt / n * 1e3, "watts": w, "sclk_mhz": f, "frames": n}))


if __name__ == "__main__":
    main()
