set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3m
bash scripts/profile.sh r3_f32 --precision f32 > gpurun_out/r3m/prof_f32.txt 2>&1
bash scripts/profile.sh r3_split --precision split_f16 > gpurun_out/r3m/prof_split.txt 2>&1
python scripts/merge_pmc.py gpurun_out/prof_r3_f32/summary.json gpurun_out/prof_r3_split/summary.json > gpurun_out/r3m/merge.txt 2>&1
cp profiles/pmc_latest.json gpurun_out/r3m/pmc_latest.json
find gpurun_out/prof_r3_f32 gpurun_out/prof_r3_split -name "*kernel_trace.csv" -size +5M -delete
find gpurun_out/prof_r3_f32 gpurun_out/prof_r3_split -name "*counter_collection.csv" -size +5M -delete
timeout 500 python bench.py --steps 20 --warmup 3 > gpurun_out/r3m/bench.json 2> gpurun_out/r3m/bench.err
timeout 500 python bench.py --steps 20 --warmup 3 --precision split_f16 --no-cpu-baseline > gpurun_out/r3m/bench_split.json 2> gpurun_out/r3m/bench_split.err
timeout 100 python scripts/band_profile.py f32 9 > gpurun_out/r3m/band_f32.jsonl 2> /dev/null
timeout 100 python scripts/band_profile.py split_f16 9 > gpurun_out/r3m/band_split.jsonl 2> /dev/null
timeout 150 python scripts/shape_times.py f32 20 > gpurun_out/r3m/shapes_f32.jsonl 2> /dev/null
timeout 150 python scripts/shape_times.py split_f16 20 > gpurun_out/r3m/shapes_split.jsonl 2> /dev/null
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3m/trace_band8 -o t -- python $GRAFT_REPO_ROOT/scripts/band_profile.py f32 30 --once 8 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r3m/trace_band8 -name "*kernel_trace.csv" -delete
wc -c gpurun_out/r3m/*.json gpurun_out/r3m/*.jsonl
