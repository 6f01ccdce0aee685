set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3p
timeout 150 python scripts/shape_times.py split_f16 20 > gpurun_out/r3p/shapes_split.jsonl 2> /dev/null
python - <<'PY'
import json
for l in open("gpurun_out/r3p/shapes_split.jsonl"):
    d = json.loads(l)
    if d["th"] == "auto" and d["tail"] == "auto": print("split", d["shape"], d["ms"], d["tflops"], d["stage_ms"])
PY
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=5 -k "split_f16 and (small_shapes or per_stage or pipe_form or rgba8 or batch or band or cartoon or restatement)" > gpurun_out/r3p/pytest.log 2>&1
echo "pytest rc=$?"; tail -n 12 gpurun_out/r3p/pytest.log
