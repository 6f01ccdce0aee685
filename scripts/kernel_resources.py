#!/usr/bin/env python3
"""Registers, scratch, LDS of every kernel in sr_kernels.hip (hipcc -Rpass-analysis=kernel-resource-usage).
    python scripts/kernel_resources.py [remarks.txt]     (without an argument: compiles the file, ~90 s)"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) > 1:
        text = open(sys.argv[1]).read()
    else:
        src = os.path.join(ROOT, "rusty_sr_amd", "csrc", "sr_kernels.hip")
        text = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-x", "hip", "-c",
                               src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    keys = [("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("SGPRs", "sgpr"), (r"ScratchSize \[bytes/lane\]", "scratch"),
            (r"Occupancy \[waves/SIMD\]", "occ"), (r"LDS Size \[bytes/block\]", "lds")]
    for b in re.split(r"remark: [^\n]*Function Name: ", text)[1:]:
        name = b.split("\n")[0]
        vals = []
        for k, label in keys:
            m = re.search(k + r": (\d+)", b)
            vals.append(f"{label} {m.group(1) if m else '?'}")
        print(f"{name[:110]:110s} " + "  ".join(vals))


if __name__ == "__main__":
    main()
