#!/usr/bin/env python3
"""Ask RCCL itself whether two ranks may share one device.

Two PROCESSES, both on device 0, each with its own context, join one communicator through libsrhip
(`sr_comm_init_rank(ctx, id, 128, rank, 2)` -> `ncclCommInitRank`).  The parent prints what each rank got back: the
sr_status, RCCL's own ncclResult_t (`sr_last_comm_error`) and whatever RCCL wrote with NCCL_DEBUG=WARN.  If RCCL accepts,
the two ranks shard one image (one `ncclSend` / `ncclRecv` pair each way) and compare their rows with the undivided call.

    python scripts/rccl_same_device.py > profiles/r6_rccl_same_device.txt
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys, time
sys.path.insert(0, %r)
import numpy as np, torch
import rusty_sr_amd as r
from rusty_sr_amd import _lib
rank, idfile = int(sys.argv[1]), sys.argv[2]
eng = r.Engine(r.rsr.builtin("imagenet"), device=0)
if rank == 0:
    uid = r.Engine.comm_unique_id()
    with open(idfile + ".tmp", "wb") as f: f.write(uid)
    os.replace(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 60: raise SystemExit("no id from rank 0")
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
res = {"rank": rank, "device": 0, "pid": os.getpid()}
try:
    eng.comm_init_rank(uid, rank, 2)
    res["init"] = "SR_OK"
except _lib.SrError as e:
    res["init"] = str(e)
    res["sr_status"] = e.code if hasattr(e, "code") else None
res["ncclResult_t"] = _lib.lib().sr_last_comm_error(eng._ctx)
if res["init"] == "SR_OK":
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (96, 160, 3), dtype=np.uint8)
    band = torch.from_numpy(img[:40] if rank == 0 else img[40:]).cuda()
    out = eng.upscale_sharded_dev(band)
    torch.cuda.synchronize()
    res["comm_ms"] = eng.last_comm_ms()
    single = r.Engine(r.rsr.builtin("imagenet"), device=0)
    whole = single.upscale_rgba8_dev(torch.from_numpy(img).cuda()[None])[0]
    torch.cuda.synchronize()
    rows = whole[:120] if rank == 0 else whole[120:]
    res["band_equals_whole"] = bool(torch.equal(rows, out))
print("RESULT " + json.dumps(res), flush=True)
''' % ROOT


def main():
    env = dict(os.environ, NCCL_DEBUG="WARN", HSA_ENABLE_IPC_MODE_LEGACY="0")
    with tempfile.TemporaryDirectory() as d:
        idfile = os.path.join(d, "uid")
        procs = [subprocess.Popen([sys.executable, "-c", CHILD, str(k), idfile], env=env, stdout=subprocess.PIPE,
                                  stderr=subprocess.STDOUT, text=True) for k in range(2)]
        outs = []
        for k, p in enumerate(procs):
            try:
                o, _ = p.communicate(timeout=240)
            except subprocess.TimeoutExpired:
                p.kill()
                o, _ = p.communicate()
                o += "\n[parent] rank %d killed after 240 s (RCCL never returned)\n" % k
            outs.append(o)
    print("question: may two ranks of one RCCL communicator share a device?  (two processes, both on device 0, nranks = 2)")
    for k, o in enumerate(outs):
        print("---- rank %d: exit %s" % (k, procs[k].returncode))
        for line in o.splitlines():
            at = line.find("RESULT {")  # (RCCL's warnings go to the same pipe unbuffered: the line may start in the middle of one)
            if at >= 0:
                try:
                    print("  result:", json.dumps(json.JSONDecoder().raw_decode(line[at + 7:])[0], sort_keys=True))
                    continue
                except ValueError:
                    pass
            if "alt_rsmi.cc" in line:  # RCCL's topology scan on a box that hides the other GPUs' sysfs nodes: dozens per rank
                continue
            if line.strip():
                print("  |", line[:300])


if __name__ == "__main__":
    main()
