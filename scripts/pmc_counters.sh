#!/bin/bash
# One rocprofv3 counter pass over scripts/run_once.py, per-kernel averages of the named counters (first launch of each kernel dropped).
#   scripts/pmc_counters.sh OUTDIR PREC "ENV ASSIGNMENTS" COUNTER [COUNTER ...]
#   e.g. scripts/pmc_counters.sh gpurun_out/pmc_ic f32 "SRHIP_FORK=0" SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES GRBM_GUI_ACTIVE
export TMPDIR=/tmp
D=$1; PREC=$2; V=$3; shift 3
rm -rf "$D"
env $V rocprofv3 --pmc "$@" --output-format csv -d "$D" -o p -- python scripts/run_once.py $PREC > "$D.log" 2>&1
python - "$D" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, dd in sorted(agg.items()):
    c = {n: sum(v[1:]) / max(1, len(v[1:])) for n, v in dd.items()}
    print(f"{k[5:62]:58s} " + "  ".join(f"{n} {v:.4g}" for n, v in sorted(c.items())))
PY
