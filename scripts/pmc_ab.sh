export TMPDIR=/tmp
for P in 1 0; do
  export SRHIP_PERSIST=$P
  echo "=== PERSIST=$P"
  python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | grep -o "\"stages.*hbm"
  rm -rf gpurun_out/pmcx; rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcx -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
  python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmcx/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "conv_stage_kernel<8, 1, 5" in r["Kernel_Name"] or "conv_stage_kernel<8, 3, 5" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    c = {n: sum(v) / len(v) for n, v in d.items()}
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(k, "cycles/XCD %.0f  waves %.0f  avg resident waves %.0f/2048  MFMA util %.3f  wait_any %.3f  wait_inst %.3f  valu_insts/wave %.0f  active_valu(quad)/wave %.0f" % (
        cyc, c["SQ_WAVES"], c["SQ_WAVE_CYCLES"] * 4 / cyc, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
        c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_INSTS_VALU"] / c["SQ_WAVES"], c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVES"]))
PY
done
