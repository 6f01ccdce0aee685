"""profiles/pmc_latest.json = per-kernel PMC averages of the kernels the engine runs on the headline workload,
merged from the two per-mode profile runs (scripts/profile.sh <tag>_f32 / <tag>_split).
    python scripts/merge_pmc.py gpurun_out/prof_r1c_f32/summary.json gpurun_out/prof_r1c_split/summary.json"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)

merged = {}
for path, precision in ((sys.argv[1], "f32"), (sys.argv[2], "split_f16")):
    d = json.load(open(path))
    for stage in range(5):
        names = [n for n in d if bench.kernel_matches(n, stage, precision)]
        names.sort(key=lambda n: 0 if "pipe" in n else 1)
        if names:
            merged[names[0]] = d[names[0]]
json.dump(merged, open(os.path.join(ROOT, "profiles", "pmc_latest.json"), "w"), indent=1)
for k, v in merged.items():
    print(k[:90].ljust(90), int(v.get("hbm_read_bytes", 0)), int(v.get("hbm_write_bytes", 0)))
