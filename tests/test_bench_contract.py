"""bench.py's one-line JSON contract, on the GPU: N=1 with roofline + cpu_baseline objects, and a
2-rank rehearsal of the sharded path (gloo, both ranks folded onto the one GPU of the test box --
RCCL refuses two ranks per device; the driver's real N>1 runs use RCCL)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _last_json(stdout):
    lines = [l for l in stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_contract():
    r = subprocess.run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--height", "360", "--width", "640"],
                       cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic" and d["dtype"] == "f32" and "workload" in d["config"]
    assert abs(d["value"] - 9 * 360 * 640 / 1e6 / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-3
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and 0 < rf["frac"] <= 1 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["source"].startswith("hip_events")
    # `frac` is quoted on the AVERAGE launch duration (the conservative figure), the median stands beside it
    assert 0 < rf["frac_at_median"] <= 1 and rf["avg_launch_ms"] > 0 and rf["median_launch_ms"] > 0
    assert abs(rf["frac_at_median"] * rf["median_launch_ms"] - rf["frac"] * rf["avg_launch_ms"]) < 1e-3 * rf["frac"] * rf["avg_launch_ms"] + 1e-4
    su = d["sustained"]  # the headline step looped for seconds, not for 3 steps
    assert "error" not in su and su["wall_s"] >= 2.5 and su["steps"] >= 100 and su["ms_per_step"] > 0 and su["event_ms_p95"] >= su["event_ms_median"]
    assert "error" not in d["fork_ab"] and d["fork_ab"]["undivided_ms"] > 0 and d["fork_ab"]["forked_ms"] > 0
    aux = d["aux_graphs"]["entries"]  # the two parameter-free graphs, u8 and f32, at the frame size and at 3x it
    assert {(e["graph"], e["io"]) for e in aux} == {(g, io) for g in ("bilinear_net", "downsample_net") for io in ("rgba8", "f32")}
    assert all(e["ms"] > 0 and 0 < e["frac_of_8TBps"] < 1 for e in aux)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert cb["leg"] in cb["legs"] and cb["value"] == max(cb["legs"][k]["value"] for k in ("c_oracle", "torch_cpu") if "value" in cb["legs"][k])
    # the other named configurations ride along as keyed entries
    for key in ("config_A", "config_C", "config_D", "host_call"):
        assert key in d and "error" not in d[key], (key, d.get(key))
    assert d["config_C"]["scaling"] == "strong" and len(d["config_C"]["roofline_frac_per_rank"]) == 1
    prev = d["config_C"]["band_preview"]  # what one rank of a 2- / 4- / 8-way split of config C does per frame, on this GPU
    for ways in (2, 4, 8):
        e = prev[f"{ways}_way"]
        assert e["rows"] == 2160 // ways and e["ms_per_band"] > 0 and 0 < e["useful_roofline_frac"] < 1
        # ... and projected with per-layer feature halos (nothing recomputed): the kernel work of the band's own rows
        assert 0 < e["layer_halos"]["kernel_ms_per_band"] <= e["ms_per_band"] * 1.02 and e["useful_roofline_frac"] * 0.98 <= e["layer_halos"]["useful_roofline_frac"] < 1
    assert 0 < d["config_A"]["whole_call_frac"] < 1
    # mid-size lone frames: the measured fork decision against the fixed rule -- same bytes, never meaningfully slower
    mid = d["mid_size_frames"]
    assert "error" not in mid and len(mid["entries"]) == 3
    assert all(e["same_bytes"] and e["plan"] in ("undivided", "forked") and 0 < e["tuned_ms"] <= 1.05 * e["rule_ms"] for e in mid["entries"]), mid
    assert 0 < mid["host_call"]["ms"] <= 1.05 * mid["host_call"]["one_chunk_ms"], mid["host_call"]
    # the split-half mode beside the headline: its dominant kernel against BOTH denominators (the nominal f16 peak and the rate a bare
    # random-operand stream of its instruction sustains on this part), on issued FLOPs (three f16 products per algorithmic one)
    orf = d["other_precision"]["roofline"]
    assert orf["peak_nominal"] == 2500.0 and 0 < orf["peak_measured_random_operands"] < orf["peak_nominal"] and "profiles/" in orf["source"]
    assert abs(orf["issued"] - 3 * orf["algorithmic"]) < 0.05 and 0 < orf["frac_of_nominal_issued"] < orf["frac_of_measured_issued"] < 1
    assert d["config_D"]["images_per_rank"] == 64 and d["config_D"]["data_path_collectives"] == 0 and "host_pipelined" in d["config_D"]


def test_bench_two_rank_rehearsal():
    env = dict(os.environ, SRHIP_DIST_BACKEND="gloo", SRHIP_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", "bench.py", "--gpus", "2", "--steps", "3",
                        "--warmup", "1", "--height", "360", "--width", "640"],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _last_json(r.stdout)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "rowband2" and "rehearsal" in d
    assert d["config"]["image"] == [720, 640]
    assert abs(d["value"] - 2 * 9 * 360 * 640 / 1e6 / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-3
    c, dd = d["config_C"], d["config_D"]
    assert "error" not in c and "error" not in dd, (c, dd)
    assert c["scaling"] == "strong" and len(c["roofline_frac_per_rank"]) == 2 and [p["rows"] for p in c["per_rank"]] == [1080, 1080]
    assert "WEAK scaling" in d["metric"] and "config_C" in d["metric"] and "speedup_vs_n1" in c
    assert abs(c["value"] - 9 * 2160 * 3840 / 1e6 / (c["ms_per_step"] / 1e3)) / c["value"] < 1e-3
    assert dd["images_per_rank"] == 32 and dd["data_path_collectives"] == 0
    assert abs(dd["value"] - 64 * 9 * 512 * 512 / 1e6 / (dd["ms_per_step"] / 1e3)) / dd["value"] < 1e-3
