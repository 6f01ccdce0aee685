set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3k
timeout 150 python scripts/shape_times.py split_f16 20 > gpurun_out/r3k/shapes_split.jsonl 2> gpurun_out/r3k/shapes_split.err
python - <<'PY'
import json
for l in open("gpurun_out/r3k/shapes_split.jsonl"):
    d = json.loads(l)
    if d["th"] == "auto" and d["tail"] == "auto": print("split", d["shape"], d["ms"], d["tflops"], d["stage_ms"])
PY
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_multi.py -m gpu -x -q --durations=8 > gpurun_out/r3k/pytest.log 2>&1
echo "pytest rc=$?"
tail -n 16 gpurun_out/r3k/pytest.log
