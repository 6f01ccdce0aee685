#!/bin/bash
# rocprofv3 recipe for bench.py (run on the GPU box via gpurun, from the repo root).
#   scripts/profile.sh <tag> [bench args...]
# Kernel-trace stats and each PMC group are SEPARATE runs (MI355X_MICROARCH.md:
# FETCH_SIZE needs 3 of the 4 TCC slots, WRITE_SIZE 2; never mixed with tracing).
# Raw output -> gpurun_out/prof_<tag>/, summaries -> gpurun_out/prof_<tag>/summary_*.csv
set -u
export TMPDIR=/tmp
# The device call may run one frame as two half-frame launches per stage (sr_run_stack_auto); rocprofv3 --stats aggregates by
# kernel NAME, so the profiled runs keep every launch a whole frame (what bench.py's `roofline` block is quoted on), and the
# sustained loop short.
export SRHIP_FORK=0 SRHIP_BENCH_SUSTAINED_S=0.3
TAG=${1:-r1}; shift || true
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
BENCH="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --no-configs $*"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- \
    python bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs "$@" > "$OUT/bench_trace.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE \
    --output-format csv -d "$OUT/pmc_sq" -o p -- $BENCH > "$OUT/bench_pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VALU SQ_LDS_UNALIGNED_STALL \
    --output-format csv -d "$OUT/pmc_sq2" -o p -- $BENCH > "$OUT/bench_pmc_sq2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o p -- $BENCH > "$OUT/bench_pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -o p -- $BENCH > "$OUT/bench_pmc_write.log" 2>&1
python scripts/summarize_pmc.py "$OUT" > "$OUT/summary.txt" 2>&1
cat "$OUT/summary.txt"
