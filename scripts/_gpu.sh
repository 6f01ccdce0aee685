set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3e
ok=1
for cfg in "SRHIP_TH=4 SRHIP_PIPE=all" "SRHIP_TAIL=1.5"; do
  for prec in f32 split_f16; do
    echo "== $cfg $prec"
    env $cfg SRHIP_TRACE=2 timeout 40 python scripts/run_once.py $prec 1080x1920 2 2>&1 | grep -c "done" || ok=0
  done
done
echo "== 256 th4 pipe"; SRHIP_TH=4 SRHIP_PIPE=all timeout 40 python scripts/run_once.py f32 256x256 3 && echo fine || ok=0
if [ $ok = 1 ]; then
timeout 700 python -m pytest tests -m gpu -x -q --deselect tests/test_bench_contract.py > gpurun_out/r3e/pytest.log 2>&1
echo "pytest rc=$?"
tail -n 6 gpurun_out/r3e/pytest.log
timeout 150 python scripts/band_profile.py f32 7 --tails > gpurun_out/r3e/band_f32.jsonl 2> gpurun_out/r3e/band_f32.err
timeout 100 python scripts/band_profile.py split_f16 7 > gpurun_out/r3e/band_split.jsonl 2> gpurun_out/r3e/band_split.err
timeout 150 python scripts/shape_times.py f32 20 > gpurun_out/r3e/shapes_f32.jsonl 2> gpurun_out/r3e/shapes_f32.err
timeout 100 python scripts/shape_times.py split_f16 20 > gpurun_out/r3e/shapes_split.jsonl 2> gpurun_out/r3e/shapes_split.err
fi
