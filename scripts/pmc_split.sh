#!/bin/bash
# What binds the split-half stage kernels?  Two SQ counter passes (8 SQ slots each) over scripts/run_once.py, per-kernel averages.
#   scripts/pmc_split.sh OUTDIR [PREC] [HxW]       (round-4 review, item 3: LDS operand traffic or something else?)
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc_split}; PREC=${2:-split_f16}; HW=${3:-1080x1920}
mkdir -p "$OUT"
SRHIP_FORK=0 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/p1" -o p -- python scripts/run_once.py $PREC $HW 6 > "$OUT/p1.log" 2>&1
SRHIP_FORK=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/p2" -o p -- python scripts/run_once.py $PREC $HW 6 > "$OUT/p2.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, dd in sorted(agg.items()):
    c = {n: sum(v[1:]) / max(1, len(v[1:])) for n, v in dd.items()}   # (first launch dropped: cold)
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8          # cycles per XCD
    simd_cyc = cyc * 1024 / 8 * 8                  # 1024 SIMDs x cycles  (counters are summed over the chip)
    cu_cyc = cyc * 256
    d = {"cycles_per_xcd": round(cyc)}
    def put(name, val): d[name] = round(val, 4)
    if cyc:
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c: put("mfma_util", c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024))
        if "SQ_LDS_IDX_ACTIVE" in c: put("lds_array_active_per_cu_cycle", c["SQ_LDS_IDX_ACTIVE"] / cu_cyc)
        if "SQ_LDS_BANK_CONFLICT" in c: put("lds_bank_conflict_per_cu_cycle", c["SQ_LDS_BANK_CONFLICT"] / cu_cyc)
        if "SQ_INSTS_LDS" in c: put("lds_insts_per_cu_kcycle", 1e3 * c["SQ_INSTS_LDS"] / cu_cyc)
        if "SQ_INST_CYCLES_VMEM" in c: put("vmem_inst_cycles_per_simd_cycle", c["SQ_INST_CYCLES_VMEM"] / (cyc * 1024))
    wc = c.get("SQ_WAVE_CYCLES", 0)
    if wc:
        for nm, key in (("parked_waitcnt_or_barrier", "SQ_WAIT_ANY"), ("issue_stall", "SQ_WAIT_INST_ANY"), ("issuing", "SQ_ACTIVE_INST_ANY")):
            if key in c: put(nm + "_per_wave_cycle", c[key] / wc)
        if "SQ_ACTIVE_INST_VALU" in c: put("valu_active_per_wave_cycle", c["SQ_ACTIVE_INST_VALU"] / wc)
    bc = c.get("SQ_BUSY_CYCLES", 0)
    if bc:
        for nm, key in (("lds_issue_stall", "SQ_WAIT_INST_LDS"), ("lds_inst_active", "SQ_ACTIVE_INST_LDS")):
            if key in c: put(nm + "_per_sq_busy_cycle", c[key] / bc)
    d["raw"] = {n: round(v) for n, v in c.items()}
    out[k] = d
    print(k[:100]); print("   ", {kk: vv for kk, vv in d.items() if kk != "raw"})
json.dump(out, open(sys.argv[1] + "/summary.json", "w"), indent=1)
PY
