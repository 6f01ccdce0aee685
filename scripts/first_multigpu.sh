#!/bin/bash
# The first lease that sees more than one MI355X: everything the multi-GPU paths still owe, in one command.
#   scripts/first_multigpu.sh [OUTDIR]            (from the repo root; ~10 min on an 8-GPU node)
# The RCCL halo exchange inside libsrhip (csrc/sr_comm.cpp) has run with ONE rank only -- the builder's lease is one device and RCCL
# admits one rank per device -- so this runs, in the order in which a failure is easiest to read:
#   1. tests/c/comm_smoke.c: plain C, no torch in the process; every device against device 0, one 3840-wide image sharded over all
#      devices through RCCL and through peer copies, bit for bit; prints config C's wall time and every rank's step / exchange time;
#   2. the two tests that skip on one device: contexts on every device (the > 64 KB LDS attribute is per device), and the RCCL-sharded
#      call + the host-memory multi-device forms over all devices;
#   3. bench.py --gpus 2 / 4 / 8 (whichever the node has), launched exactly as the driver does: the weak-scaling `value`, config C
#      (3840x2160 in N row bands: per-rank roofline fraction, halo-exchange time and how much of it the band's stream waited for) and
#      config D (64 x 512x512 dealt round-robin); for N > 1 a second line with SRHIP_HALO=layers (per-layer feature halos, nothing recomputed).
# Results: OUTDIR/comm_smoke.txt, OUTDIR/pytest_multi.txt, profiles/r6_scale_N.json (the bench line) and profiles/r6_scale_summary.json
# (N -> MP/s, ms per step, config C's ms, per-rank roofline_frac and comm_ms) -- commit the profiles/ files.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=${1:-gpurun_out/multigpu}; mkdir -p "$OUT" profiles
NDEV=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "devices: $NDEV" | tee "$OUT/devices.txt"
rc_all=0

# -- 1. plain-C smoke (also runs, with its single-device legs, on one GPU)
gcc -std=c99 -D_POSIX_C_SOURCE=199309L -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude tests/c/comm_smoke.c -Lrusty_sr_amd -lsrhip \
    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,"$PWD/rusty_sr_amd" -Wl,-rpath,/opt/rocm/lib -o "$OUT/comm_smoke" \
  && timeout 900 "$OUT/comm_smoke" rusty_sr_amd/res/imagenet.rsr 2>&1 | tee "$OUT/comm_smoke.txt"
rc=${PIPESTATUS[0]}; echo "comm_smoke rc=$rc" | tee -a "$OUT/comm_smoke.txt"; [ "$rc" = 0 ] || rc_all=1

# -- 2. the tests that need a second device (reported as skipped on one)
timeout 1800 python -m pytest tests/test_gpu_multi.py -q -m gpu -rs -k "other_devices or all_devices_with_rccl" 2>&1 | tail -15 | tee "$OUT/pytest_multi.txt"
[ "${PIPESTATUS[0]}" = 0 ] || rc_all=1

# -- 3. the scaling lines, launched as the driver launches them
for N in 1 2 4 8; do
    [ "$N" -le "$NDEV" ] || continue
    if [ "$N" = 1 ]; then
        timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/scale_$N.json" 2> "$OUT/scale_$N.err"
    else
        timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29500 + N)) \
            bench.py --gpus "$N" --steps 20 --warmup 5 > "$OUT/scale_$N.json" 2> "$OUT/scale_$N.err"
    fi
    echo "bench --gpus $N rc=$?"
    tail -1 "$OUT/scale_$N.json" > "profiles/r6_scale_$N.json"
    if [ "$N" -gt 1 ]; then  # the same line with feature rows exchanged after every stage instead of the recomputed overlap
        SRHIP_HALO=layers timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29600 + N)) \
            bench.py --gpus "$N" --steps 20 --warmup 5 > "$OUT/scale_layers_$N.json" 2> "$OUT/scale_layers_$N.err"
        echo "bench --gpus $N (layer halos) rc=$?"
        tail -1 "$OUT/scale_layers_$N.json" > "profiles/r6_scale_layers_$N.json"
    fi
done
python - "$OUT" <<'PY'
import glob, json, os, sys
out = {}
for f in sorted(glob.glob("profiles/r6_scale_[0-9]*.json")) + sorted(glob.glob("profiles/r6_scale_layers_[0-9]*.json")):
    try:
        d = json.loads(open(f).read())
    except Exception as e:  # noqa: BLE001
        out[os.path.basename(f)] = {"error": str(e)}
        continue
    c = d.get("config_C") or {}
    out[str(d.get("n_gpus")) + ("_layer_halos" if "layers" in f else "")] = {
        "value_mp_s": d.get("value"), "ms_per_step": d.get("ms_per_step"), "scaling": d.get("scaling"), "exchange": (d.get("config") or {}).get("exchange"),
        "exchange_check": d.get("exchange_check"), "comm_ms": d.get("comm_ms"), "comm_exposed_ms": d.get("comm_exposed_ms"),
        "config_C": {k: c.get(k) for k in ("ms_per_step", "value", "speedup_vs_n1", "efficiency_vs_n1", "roofline_frac_per_rank", "recompute_overhead")},
        "config_C_per_rank": [{k: p.get(k) for k in ("rank", "rows", "roofline_frac", "comm_ms", "comm_exposed_ms")} for p in (c.get("per_rank") or []) if p],
        "config_D": {k: (d.get("config_D") or {}).get(k) for k in ("ms_per_step", "value", "images_per_rank")},
    }
json.dump(out, open("profiles/r6_scale_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
echo "first_multigpu: $([ $rc_all = 0 ] && echo ok || echo FAILURES -- see $OUT)"
exit $rc_all
