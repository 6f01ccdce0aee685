set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3i
bash scripts/profile.sh r3_f32 --precision f32 > gpurun_out/r3i/prof_f32.txt 2>&1
bash scripts/profile.sh r3_split --precision split_f16 > gpurun_out/r3i/prof_split.txt 2>&1
python scripts/merge_pmc.py gpurun_out/prof_r3_f32/summary.json gpurun_out/prof_r3_split/summary.json > gpurun_out/r3i/merge.txt 2>&1
cp profiles/pmc_latest.json gpurun_out/r3i/pmc_latest.json
find gpurun_out/prof_r3_f32 gpurun_out/prof_r3_split -name "*kernel_trace.csv" -size +5M -delete
find gpurun_out/prof_r3_f32 gpurun_out/prof_r3_split -name "*counter_collection.csv" -size +5M -delete
timeout 400 python bench.py --steps 20 --warmup 3 > gpurun_out/r3i/bench.json 2> gpurun_out/r3i/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3i/trace_band8 -o t -- python $GRAFT_REPO_ROOT/scripts/band_profile.py f32 30 --once 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r3i/trace_band8 -name "*kernel_trace.csv" -delete
grep -A12 "kernel-trace --stats" gpurun_out/r3i/prof_f32.txt | head -14
grep "MFMA busy\|^void\|^conv" gpurun_out/r3i/prof_split.txt | head -30
