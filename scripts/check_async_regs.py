#!/usr/bin/env python3
"""Lint of the device assembly: a VGPR that an inline-asm VMEM instruction returns into asynchronously (the queue's
returning atomic, the head snapshot, the final stage's pixel prefetch) must not be READ before an s_waitcnt vmcnt that
COVERS it -- the compiler does not know those asm outputs land later and is free to copy them at once (it did, when such
a register was live across the merge of the two tile bodies: round 3, a hang).  From each such instruction every path the
wave can take is walked (both sides of a conditional branch) until a covering wait.  The kernels' OWN waits (inline asm:
wait_vm / wait_vm_barrier, whose immediate is picked at run time from the sequence numbers StepStream keeps) are trusted;
a wait the COMPILER inserted, `s_waitcnt vmcnt(N)` outside an asm block, covers the instruction only if at least N VMEM
instructions were issued after it on that path (memory instructions retire in order, so "at most N outstanding" then
includes it among the retired ones; with a larger N it does not and the walk goes on).  Any instruction on the way that
names the register as a source is flagged.

A second lint covers the two inline-asm idioms whose correctness depends on instruction ENCODING SIZES (advisor, round 4):
  * the run-time s_waitcnt table (sr_kernels.hip SR_WAIT_TABLE): `s_getpc_b64 vcc` must be followed by exactly `s_add_u32 vcc_lo, vcc_lo,
    sN` / `s_addc_u32 vcc_hi, vcc_hi, 0` / `s_setpc_b64 vcc` (three 4-byte encodings: a literal operand or an inserted s_nop would
    make the +12 in the index wrong) and then sixteen rows `s_waitcnt ...` / `s_branch ...` of 8 bytes each, nothing in between --
    a jump that lands mid-row would under-wait silently;
  * every `s_mov_b32 m0, ...` of an LDS-DMA statement must be followed by an s_nop before its global_load_lds (M0 written by the
    scalar ALU needs a wait state before a memory instruction reads it).

    python scripts/check_async_regs.py [file.s]      (without an argument: compiles sr_kernels.hip to assembly, ~90 s)
Exit status 1 when something is flagged."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VMEM = ("global_load", "global_store", "global_atomic", "buffer_load", "buffer_store", "buffer_atomic", "scratch_load", "scratch_store",
        "flat_load", "flat_store", "flat_atomic")
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
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-x", "hip", "-S",
                               "--cuda-device-only", os.path.join(ROOT, "rusty_sr_amd", "csrc", "sr_kernels.hip"), "-o", path],
                              stderr=subprocess.DEVNULL)
    lines = open(path).read().split("\n")
    labels = {}
    for i, l in enumerate(lines):  # (labels are unique within the file: .LBB<function>_<block>)
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    in_asm, flag = [], False
    for l in lines:
        if "#ASMSTART" in l:
            flag = True
        elif "#ASMEND" in l:
            flag = False
        in_asm.append(flag)
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
        # depth-first over the paths from here; best[j] = fewest VMEM instructions issued since the asm with which line j was reached
        # (fewer is the harder case for a wait to cover, so a revisit with more is pruned)
        stack, best, steps, flagged = [(i, 0)], {}, 0, False
        while stack and not flagged and steps < 200000:
            j, issued = stack.pop()
            while j + 1 < len(lines) and steps < 200000:
                j += 1
                steps += 1
                if j in best and best[j] <= issued:
                    break
                best[j] = issued
                t = lines[j].split(";")[0].strip()
                if not t or t.endswith(":") or t.startswith(".") or t.startswith("#"):
                    continue
                ops = t.replace(",", " ").split()
                if ops[0] == "s_branch":
                    if ops[1] in labels:
                        j = labels[ops[1]]
                        continue
                    break
                if ops[0].startswith("s_cbranch") and ops[-1] in labels:
                    stack.append((labels[ops[-1]], issued))  # ... and the fall-through below
                    continue
                if ops[0] in ("s_endpgm", "s_setpc_b64"):
                    break
                if ops[0] == "s_waitcnt":
                    mm = re.search(r"vmcnt\((\d+)\)", t)
                    if mm and (in_asm[j] or int(mm.group(1)) <= issued):
                        break  # covered on this path
                    continue
                srcs = set()
                for tok in ops[2:]:  # ops[1] is the destination of almost every VALU / VMEM-load form
                    srcs |= regs_of(tok)
                if ops[0].startswith(("global_store", "ds_write", "buffer_store", "scratch_store", "global_load_lds", "v_cmp", "v_readlane", "v_readfirstlane")):
                    for tok in ops[1:]:
                        srcs |= regs_of(tok)
                if dst in srcs:
                    print(f"{func}: line {j + 1}: `{t}` reads v{dst} before a wait that covers `{l.strip()}` (line {i + 1})")
                    bad += 1
                    flagged = True
                    break
                if ops[0].startswith(VMEM):
                    issued += 1
                    if ops[0].startswith(("global_load", "buffer_load", "scratch_load", "flat_load")) and "lds" not in ops[0] and dst in regs_of(ops[1]):
                        break  # the register is given a new value: the old result is dead on this path
    print(f"{checked} asynchronous asm results checked, {bad} read too early")
    tables, bad_layout = check_encoded_layouts(lines)
    print(f"{tables[0]} wait tables and {tables[1]} LDS-DMA m0 writes checked, {bad_layout} with an unexpected layout ({tables[2]} tables of 64 rows)")
    return 1 if bad or bad_layout or not tables[0] else 0


def check_encoded_layouts(lines):
    """See the module docstring: the computed jump into the s_waitcnt table, and the wait state behind a write of m0."""
    code = [(i, l.split(";")[0].strip()) for i, l in enumerate(lines)]
    code = [(i, t) for i, t in code if t and not t.endswith(":") and not t.startswith((".", "#"))]
    n_tab = n_m0 = bad = n_wide = 0
    for k, (i, t) in enumerate(code):
        if t.startswith("s_getpc_b64"):
            # (a kernel of more than 128 KB makes the COMPILER relax far branches into s_getpc_b64 sN / s_add_u32 ... (.LBB - .Lpost_getpc) /
            # s_setpc_b64: label arithmetic the assembler resolves, nothing of ours)
            if re.search(r"\.Lpost_getpc\d+\)", " ".join(x[1] for x in code[k + 1:k + 3])):
                continue
            n_tab += 1
            # the index arithmetic in front of it: clamp to the table's last row (15, or 63 in the 64-row table), << 3 (8-byte rows), + 12 (the
            # three instructions behind s_getpc)
            prev = [x[1] for x in code[max(0, k - 4):k]]
            clamp = [int(m.group(1)) for x in prev for m in [re.match(r"s_min_i32 s\d+, s\d+, (\d+)$", x)] if m]
            rows = clamp[0] + 1 if clamp and clamp[0] in (15, 63) else 0
            want = [r"s_add_u32 vcc_lo, vcc_lo, s\d+$", r"s_addc_u32 vcc_hi, vcc_hi, 0$", r"s_setpc_b64 vcc$"]
            for row in range(rows):
                want += [r"s_waitcnt vmcnt\(%d\)( lgkmcnt\(0\))?$" % row, r"s_branch \S+$"]
            got = [x[1] for x in code[k + 1:k + 1 + len(want)]]
            ok = rows and t == "s_getpc_b64 vcc" and len(got) == len(want) and all(re.match(w, g) for w, g in zip(want, got))
            # ... and the row behind the last one is not another row of a longer table
            nxt = code[k + 1 + len(want)][1] if k + 1 + len(want) < len(code) else ""
            ok = ok and not re.match(r"s_waitcnt vmcnt\(%d\)" % rows, nxt)
            n_wide += rows == 64
            ok = ok and any(re.match(r"s_lshl_b32 s\d+, s\d+, 3$", x) for x in prev) and any(re.match(r"s_add_u32 s\d+, s\d+, 12$", x) for x in prev)
            if not ok:
                print(f"line {i + 1}: the s_waitcnt table behind `{t}` does not have the layout its computed jump assumes")
                bad += 1
        if re.match(r"s_mov_b32 m0, ", t):
            nxt = [x[1] for x in code[k + 1:k + 3]]
            if nxt and nxt[0].startswith("s_nop") and len(nxt) > 1 and nxt[1].startswith("global_load_lds"):
                n_m0 += 1
            elif any(x.startswith("global_load_lds") for x in nxt):
                print(f"line {i + 1}: `{t}` is not followed by an s_nop before its global_load_lds")
                bad += 1
    return (n_tab, n_m0, n_wide), bad


if __name__ == "__main__":
    sys.exit(main())
