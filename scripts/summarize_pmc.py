#!/usr/bin/env python3
"""Collapse rocprofv3 CSV output (kernel stats + counter_collection) into one
per-kernel table: average duration and average counter value per launch."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
rows = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv" in r["Name"]:
            rows.append((r["Name"], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print("== kernel-trace --stats (per launch) ==")
for name, calls, us, pct in sorted(rows, key=lambda r: -r[3]):
    print(f"{name:70s} calls={calls:4d} avg={us:10.1f} us  {pct:5.1f}%")

# per-dispatch durations -> median / min per kernel (the --stats table only has the average, which a few slow
# launches -- clock ramp, first touch -- pull up)
import statistics
durs = defaultdict(list)
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv" in r.get("Kernel_Name", ""):
            durs[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("\n== kernel-trace, per-dispatch durations ==")
for k in sorted(durs, key=lambda k: -sum(durs[k])):
    v = durs[k]
    print(f"{k:70s} n={len(v):4d} median={statistics.median(v):10.1f} us  min={min(v):10.1f}  max={max(v):10.1f}")

agg = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get("Kernel_Name", "")
        if "conv" not in k:
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("\n== PMC counters, average per launch ==")
for k in sorted(agg):
    print(k)
    for cn in sorted(agg[k]):
        v = agg[k][cn]
        print(f"    {cn:36s} {sum(v) / len(v):18.1f}   (n={len(v)})")
    c = {cn: sum(v) / len(v) for cn, v in agg[k].items()}
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
        print(f"    -> MFMA busy / SQ busy = {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f}")
    if "FETCH_SIZE" in c:
        print(f"    -> HBM read  ~ {2 * c['FETCH_SIZE'] / 1024:.1f} MB (FETCH_SIZE KB x2: gfx950 tallies 128-B requests at 64 B)")
    if "WRITE_SIZE" in c:
        print(f"    -> HBM write ~ {c['WRITE_SIZE'] / 1024:.1f} MB (uncalibrated)")

# machine-readable copy (bench.py reads the HBM traffic of the dominant kernel from it)
import json
js = {}
for name, calls, us, pct in rows:
    js.setdefault(name, {})["avg_us"] = us
    js[name]["calls"] = calls
for k, v in durs.items():
    d = js.setdefault(k, {})
    d["median_us"] = statistics.median(v)
    d["min_us"] = min(v)
    d["max_us"] = max(v)
for k in agg:
    c = {cn: sum(v) / len(v) for cn, v in agg[k].items()}
    d = js.setdefault(k, {})
    if "FETCH_SIZE" in c:
        d["hbm_read_bytes"] = 2 * c["FETCH_SIZE"] * 1024   # KB -> B, x2 gfx950 correction (MI355X_MICROARCH.md HBM section)
    if "WRITE_SIZE" in c:
        d["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
        d["mfma_util"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (c["GRBM_GUI_ACTIVE"] / 8 * 1024)  # 8 XCD GRBMs, 1024 SIMDs
    if "SQ_LDS_BANK_CONFLICT" in c:
        d["lds_bank_conflict_cycles"] = c["SQ_LDS_BANK_CONFLICT"]
json.dump(js, open(os.path.join(out, "summary.json"), "w"), indent=1)
