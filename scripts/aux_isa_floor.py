#!/usr/bin/env python3
"""Instruction-count floor of the u8 bilinear graph from its ISA ("or commit an instruction-count floor from the
ISA: look-ups per sample x cycles against bytes per sample").

Compiles rusty_sr_amd/csrc/sr_aux.hip to gfx950 assembly (no GPU needed), takes the main loop of bilinear_u8_kernel<3, true> -- one pass =
one wave's 64 work items of 12 output pixels each (4 pixels x the 3 output rows of an input row) -- drops the basic blocks of the
W < 3 path (the only ones with byte / short loads) and counts instructions by issue class.  A wave64 vector-ALU instruction occupies its
SIMD for 4 cycles (16 lanes per cycle; packed f32 and SDWA forms alike), so

    floor_time(pass) = VALU instructions x 4 cycles        bytes(pass) = 64 items x (48 B written + 4 B read)
    floor_rate = 1024 SIMDs x clock x bytes / floor_time

which is what the kernel would reach with its vector ALUs issuing every cycle and everything else hidden.
    python scripts/aux_isa_floor.py [measured_us_1080p measured_us_5760x3240]"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    path = os.path.join(tempfile.gettempdir(), "sr_aux_floor.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-x", "hip", "-S",
                           "--cuda-device-only", os.path.join(ROOT, "rusty_sr_amd", "csrc", "sr_aux.hip"), "-o", path], stderr=subprocess.DEVNULL)
    text = open(path).read()
    m = re.search(r"^(_ZN\S*bilinear_u8_kernelILi3ELb1EEEv7AuxArgs):.*?\.end_amdhsa_kernel", text, re.S | re.M)
    lines = m.group(0).split("\n")
    # the main loop = the innermost-depth-1 loop with the most instructions: from its header label to the last branch back to it
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    best = None
    for lab, i in labels.items():
        backs = [j for j, l in enumerate(lines) if j > i and re.search(r"s_cbranch\S*\s+" + re.escape(lab) + r"\b", l)]
        if backs and (best is None or backs[-1] - i > best[1] - best[0]):
            best = (i, backs[-1], lab)
    lo, hi, lab = best
    # also the latch blocks that sit BEFORE the header in layout order (".LBB_17: in Loop: Header=BB_18"): include every block marked as in this loop
    hdr = lab.replace(".L", "")
    first = min([i for i, l in enumerate(lines) if ("Header=" + hdr) in l] + [lo])
    blocks, cur = [], []
    for l in lines[first:hi + 1]:
        if re.match(r"^\.LBB\d+_\d+:", l) and cur:
            blocks.append(cur); cur = []
        cur.append(l)
    blocks.append(cur)
    def ops(block):
        out = []
        for l in block:
            t = l.split(";")[0].strip()
            if t and not t.endswith(":") and not t.startswith("."):
                out.append(t.split()[0])
        return out
    # The W < 3 path (windows assembled byte by byte) is one basic block per window row; the first row's sits in the loop header's block
    # together with main-path work (the SrgbToLinear look-ups).  Pure W < 3 blocks are dropped; from a mixed block as many instructions of
    # each class are taken off as a pure block has.
    is_slow = lambda o: any(x in ("global_load_ubyte", "global_load_ushort", "global_load_sbyte") for x in o)
    pure = [ops(b) for b in blocks if is_slow(ops(b)) and not any(x.startswith("ds_") for x in ops(b))]
    key = lambda op: "v" if op.startswith("v_") else "d" if op.startswith("ds_") else "m" if op.startswith(("global_", "flat_", "buffer_")) else "s"
    pure_counts = Counter(key(x) for x in pure[0]) if pure else Counter()
    main_ops, slow = [], 0
    for b in blocks:
        o = ops(b)
        if is_slow(o):
            if any(x.startswith("ds_") for x in o):  # mixed: remove a pure block's worth
                left = Counter(pure_counts)
                keep = []
                for x in o:
                    slowload = x in ("global_load_ubyte", "global_load_ushort", "global_load_sbyte")
                    if slowload or (left[key(x)] > 0 and key(x) != "d" and not x.startswith("v_pk_")):
                        left[key(x)] -= 1; slow += 1
                    else:
                        keep.append(x)
                main_ops += keep
            else:
                slow += len(o)
            continue
        main_ops += o
    cls = Counter()
    for op in main_ops:
        if op.startswith("v_"):
            cls["valu"] += 1
        elif op.startswith("ds_"):
            cls["lds"] += 1
        elif op.startswith(("global_", "flat_", "buffer_")):
            cls["vmem"] += 1
        elif op.startswith("s_"):
            cls["salu"] += 1
    valu = cls["valu"]
    pk = sum(1 for o in main_ops if o.startswith("v_pk_"))
    lut = sum(1 for o in main_ops if o == "ds_read_b32")
    bytes_pass = 64 * (48 + 4)
    ghz = 2.4
    rate = 1024 * ghz * 1e9 * bytes_pass / (valu * 4) / 1e12
    print(f"bilinear_u8_kernel<3, true>, main loop ({lab}), one pass = 64 work items x 12 output pixels; W < 3 blocks dropped ({slow} instructions)")
    print(f"  vector ALU {valu} (of which packed f32 {pk}), LDS {cls['lds']} (table look-ups ds_read_b32: {lut} = 27 SrgbToLinear + 36 quantiser),"
          f" vector memory {cls['vmem']}, scalar {cls['salu']}")
    print(f"  per work item (12 output pixels = 36 samples): {valu} vector instructions = {valu / 36:.1f} per sample, {lut} look-ups = {lut / 36:.2f} per sample")
    print(f"  vector-ALU floor: {valu} x 4 = {valu * 4} cycles per pass for {bytes_pass} B of compulsory I/O -> {rate:.2f} TB/s at {ghz} GHz"
          f" = {rate / 8:.3f} of 8 TB/s, {rate / 6.3:.3f} of the 6.3 TB/s a copy reaches")
    for name, px, arg in (("1920x1080", 1920 * 1080, 1), ("5760x3240", 5760 * 3240, 2)):
        floor_us = px * 39 / (rate * 1e12) * 1e6
        line = f"  {name}: {px * 39 / 1e6:.1f} MB -> floor {floor_us:.1f} us"
        if len(sys.argv) > arg:
            us = float(sys.argv[arg])
            line += f"; measured {us:.1f} us = {floor_us / us:.2f} of the floor's rate ({px * 39 / us / 1e6:.2f} TB/s = {px * 39 / us / 1e6 / 8:.3f} of 8 TB/s)"
        print(line)


if __name__ == "__main__":
    main()
