#!/bin/bash
# usage: pmc_small.sh HxW "ENV" counters...
export TMPDIR=/tmp
SZ=$1; V=$2; shift 2
D=gpurun_out/pmc_small_$(echo "$SZ$V" | tr -c 'A-Za-z0-9\n' '_')
rm -rf "$D"
env $V rocprofv3 --pmc "$@" --output-format csv -d "$D" -o p -- python scripts/run_once.py f32 $SZ 20 > "$D.log" 2>&1
python - "$D" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv" in r["Kernel_Name"]:
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, dd in sorted(agg.items()):
    c = {n: sum(v[2:]) / max(1, len(v[2:])) for n, v in dd.items()}
    print(f"{k[5:62]:58s} " + "  ".join(f"{n} {v:.4g}" for n, v in sorted(c.items())))
PY
