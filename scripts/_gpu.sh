set -u
export TMPDIR=/tmp
mkdir -p gpurun_out/r3z
timeout 500 python bench.py --steps 20 --warmup 3 > gpurun_out/r3z/bench.json 2> gpurun_out/r3z/bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3z/bench.json"))
print(d["ms_per_step"], d["whole_call_frac"], d.get("batch_of_4"), d.get("two_frames_in_flight"))
PY
