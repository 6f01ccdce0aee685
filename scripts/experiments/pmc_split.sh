export TMPDIR=/tmp
PREC=${1:-split_f16}
rm -rf gpurun_out/pmcx; rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcx -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --precision $PREC > /dev/null 2>&1
rm -rf gpurun_out/pmcy; rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/pmcy -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --precision $PREC > /dev/null 2>&1
rm -rf gpurun_out/pmcz; rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmcz -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --precision $PREC > /dev/null 2>&1
rm -rf gpurun_out/pmcw; rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d gpurun_out/pmcw -o p -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --precision $PREC > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("pmcx", "pmcy", "pmcz", "pmcw"):
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "conv" in r["Kernel_Name"]:
                agg[r["Kernel_Name"][:52]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, dd in sorted(agg.items()):
    c = {n: sum(v) / len(v) for n, v in dd.items()}
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print(k)
    print("   cycles/XCD %.0f (%.3f ms @2.4GHz) waves %.0f resident %.0f/2048  MFMA busy/SIMD-cycles %.3f  wait_any %.3f wait_inst %.3f active_valu %.3f  valu_insts/wave %.0f" % (
        cyc, cyc / 2.4e6, c["SQ_WAVES"], c["SQ_WAVE_CYCLES"] * 4 / cyc, c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024),
        c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], c["SQ_INSTS_VALU"] / c["SQ_WAVES"]))
    print("   LDS idx_active/CU-cycles %.3f  bank_conflict/idx_active %.3f  lds_insts/wave %.0f  wait_inst_lds/wave_cycles %.3f  HBM rd %.0f MB wr %.0f MB  L2 hit %.3f" % (
        c["SQ_LDS_IDX_ACTIVE"] / (cyc * 256), c["SQ_LDS_BANK_CONFLICT"] / max(c["SQ_LDS_IDX_ACTIVE"], 1), c["SQ_INSTS_LDS"] / c["SQ_WAVES"],
        c["SQ_WAIT_INST_LDS"] / c["SQ_WAVE_CYCLES"], 2 * c["FETCH_SIZE"] / 1024, c["WRITE_SIZE"] / 1024, c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])))
PY
