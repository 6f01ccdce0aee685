#!/bin/bash
# One SQ counter pass over scripts/run_once.py for a few variants (env assignments), per-kernel averages.
#   scripts/pmc_quick.sh OUTTAG PREC "ENV1" "ENV2" ...      e.g.  scripts/pmc_quick.sh tail f32 "SRHIP_TAIL=0" "SRHIP_TAIL=1.5"
export TMPDIR=/tmp
TAG=$1; PREC=$2; shift 2
for V in "$@"; do
  D=gpurun_out/pmcq_${TAG}_$(echo "$V" | tr -c 'A-Za-z0-9\n' '_')
  rm -rf "$D"
  env $V rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS \
      --output-format csv -d "$D" -o p -- python scripts/run_once.py $PREC > /dev/null 2>&1
  echo "== $V"
  python - "$D" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, dd in sorted(agg.items()):
    c = {n: sum(v[1:]) / max(1, len(v[1:])) for n, v in dd.items()}
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    wc = c["SQ_WAVE_CYCLES"]
    print(f"{k[5:60]:56s} cyc/XCD {cyc:9.0f}  mfma_util {c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}  parked {c['SQ_WAIT_ANY'] / wc:.3f}  "
          f"issue_stall {c['SQ_WAIT_INST_ANY'] / wc:.3f}  active {c['SQ_ACTIVE_INST_ANY'] / wc:.3f}  lds_stall {c['SQ_WAIT_INST_LDS'] / wc:.3f}")
PY
done
