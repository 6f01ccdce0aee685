#!/usr/bin/env python3
"""Lint of the device assembly: a VGPR that an inline-asm VMEM instruction returns into asynchronously (the queue's
returning atomic, the head snapshot, the final stage's pixel prefetch) must not be READ before an s_waitcnt vmcnt -- the
compiler does not know those asm outputs land later and is free to copy them at once (it did, when such a register was
live across the merge of the two tile bodies: round 3, a hang).  Scan from each such instruction to the next s_waitcnt
vmcnt along the path the wave takes (unconditional branches followed, conditional ones fall through); flags any instruction
that names the register as a source.

    python scripts/check_async_regs.py [file.s]      (without an argument: compiles sr_kernels.hip to assembly, ~90 s)
Exit status 1 when something is flagged."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASYNC = re.compile(r"^\s*(global_atomic_add|global_load_dword|global_load_ubyte)\s+(v\d+),\s*v\[?\d+")


def regs_of(tok):
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def main():
    if len(sys.argv) > 1:
        path = sys.argv[1]
    else:
        path = os.path.join(tempfile.gettempdir(), "sr_kernels_lint.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-x", "hip", "-S",
                               "--cuda-device-only", os.path.join(ROOT, "rusty_sr_amd", "csrc", "sr_kernels.hip"), "-o", path],
                              stderr=subprocess.DEVNULL)
    lines = open(path).read().split("\n")
    labels = {}
    for i, l in enumerate(lines):  # (labels are unique within the file: .LBB<function>_<block>)
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    bad, checked, inasm, func = 0, 0, False, "?"
    for i, l in enumerate(lines):
        if l.startswith("_Z") and l.rstrip().endswith(("Args:", "Args")) or re.match(r"^_Z\w+:", l):
            func = l.split(":")[0]
        if "#ASMSTART" in l:
            inasm = True
        elif "#ASMEND" in l:
            inasm = False
        m = ASYNC.match(l)
        if not (m and inasm):
            continue
        checked += 1
        dst = int(m.group(2)[1:])
        j, steps = i, 0
        while steps < 4000 and j + 1 < len(lines):
            j += 1
            steps += 1
            t = lines[j].split(";")[0].strip()
            if not t or t.endswith(":") or t.startswith("."):
                continue
            if t.startswith("s_branch "):  # follow the path the wave takes (conditional branches: the fall-through)
                tgt = t.split()[1]
                if tgt in labels:
                    j = labels[tgt]
                continue
            if t.startswith("s_waitcnt") and "vmcnt" in t:
                break
            if t.startswith("s_endpgm"):
                break
            ops = t.replace(",", " ").split()
            srcs = set()
            for tok in ops[2:]:  # ops[1] is the destination of almost every VALU / VMEM-load form
                srcs |= regs_of(tok)
            if ops[0].startswith(("global_store", "ds_write", "buffer_store", "global_load_lds", "v_cmp", "v_readlane", "v_readfirstlane")):
                for tok in ops[1:]:
                    srcs |= regs_of(tok)
            if dst in srcs:
                print(f"{func}: line {j + 1}: `{t}` reads v{dst} before the wait for `{l.strip()}` (line {i + 1})")
                bad += 1
                break
    print(f"{checked} asynchronous asm results checked, {bad} read too early")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
