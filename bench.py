#!/usr/bin/env python3
"""bench.py -- throughput of the upscale hot path (graph.forward, reference
src/main.rs:171) on MI355X, with the roofline of its dominant kernel and a CPU
baseline beside it.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of the whole conv stack over one image resident in HBM
(u8 RGB in -> u8 RGBA out by default, i.e. img_to_data + graph.forward +
data_to_img fused; --io f32 times the f32-in / f32-out form).

Workload of `value` at N=1: 1920x1080 RGB (BASELINE.json configs[2], the configuration
north_star quotes its target on), x3 upscale with the bundled imagenet.rsr.
BASELINE.json says "4x"; the reference is hard-wired to factor 3
(main.rs:31) and its weights only fit factor 3, so every number here is x3.
The other named configurations ride along as keyed entries of the same JSON line:
  config_A  256x256                       (N=1)
  config_C  3840x2160: at N=1 the whole image on one GPU; at N>1 STRONG scaling -- N row bands
            (2160/N rows each), 7-row halo exchange inside the timed region, per-rank roofline fraction
  config_D  64 x 512x512: image i on rank i mod N, no communication; device-resident and host-pipelined
            (H2D / kernels / D2H overlapped) rates

`value` at N>1 (weak scaling): the image grows to 1920 x (1080*N); rank r owns row band r,
exchanges 7-row halos with its neighbours over RCCL each step (inside the timed
region) and writes its own output rows.  value = all output pixels / max-rank time.
The exchange is libsrhip's own RCCL communicator (sr_comm_init_rank / sr_upscale_sharded_*_dev: what a
Rust host calls); torch.distributed only carries the 128-byte id, the barrier and the max-over-ranks.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# Algorithmic work per INPUT pixel (reference network.rs:33,60-72):
MAC_PER_PX = {  # stage -> MACs per input pixel
    0: 32 * 25 * 3,                    # conv0
    1: 32 * 25 * 32,                   # conv1
    2: 32 * 25 * 32 + 32 * 9 * 32,     # conv2 + conv5
    3: 32 * 25 * 32 + 2 * 32 * 9 * 32, # conv3 + conv6 + conv8
    4: 3 * 27 * 9 * 32,                # conv7 + conv9 + conv10
}
FLOP_PER_PX = 2 * sum(MAC_PER_PX.values())  # 260352
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 2.4 GHz x 256 FLOP/clk
PEAK_F16_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: dense bf16/f16 MFMA (not the 2:1-sparse headline)
MEASURED_F16_MFMA16_RANDOM_TFLOPS = 1967.0  # bare v_mfma_f32_16x16x32_f16 stream on random operands, 20 s: profiles/r6_power_clock_probe.txt
PEAK_HBM_GBPS = 8000.0
MARGIN = (5, 3, 2, 1, 0)  # extra rows stage s computes either side of a band (what later stages read)
# SRHIP_HALO=layers (include/srhip_experimental.h "halo"): the library's sharded calls exchange feature rows after every stage instead
# of recomputing that overlap -- a band's stages then compute its own rows only
LAYER_HALOS = os.environ.get("SRHIP_HALO", "") == "layers"


def synth_u8(seed, h, w, n=None):
    """seeded u8 noise, 5x5 box-smoothed (edge clamped, sum // 25).  n: a batch (n,h,w,3)."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, ((n or 1), h, w, 3), dtype=np.uint8).astype(np.int32)
    p = np.pad(a, ((0, 0), (2, 2), (2, 2), (0, 0)), mode="edge")
    s = np.zeros_like(a)
    for dy in range(5):
        for dx in range(5):
            s += p[:, dy:dy + h, dx:dx + w, :]
    s = (s // 25).astype(np.uint8)
    return s if n else s[0]


STAGE_SHAPE = {1: (1, 5), 2: (2, 5), 3: (3, 5), 4: (3, 3)}  # stage -> (sources, first kernel size)


def kernel_matches(name, stage, precision):
    """Does a rocprofv3 kernel name belong to this stage in this arithmetic mode (factor-3 instance)?
    pipe form : conv_stage_pipe_kernel<NSRC, KS0, FINAL, IMG_U8, OUT_U8, PREC, FACTOR>
    first form: conv_stage_kernel<TH, NSRC, KS0, FINAL, IMG_U8, OUT_U8, PREC, PERSIST, NW[, FACTOR]>"""
    prec = 0 if precision == "f32" else 1
    if stage == 0:
        if name.startswith("void conv0_split_kernel<8, "):  # the split-half mode's own stage-0 kernel (round 5)
            return prec == 1
        if name.startswith("void conv0_kernel<8, "):  # exact mode only since round 6 (<TH, IMG_U8>; rounds 1-5: <TH, IMG_U8, PREC>)
            args = name[name.index("<") + 1:name.rindex(">")].split(", ")
            return (int(args[2]) if len(args) > 2 else 0) == prec
        return False
    nsrc, ks = STAGE_SHAPE[stage]
    if f"conv_stage_pipe_kernel<{nsrc}, {ks}, " in name:
        args = name[name.index("<") + 1:name.rindex(">")].split(", ")
        return int(args[5]) == prec and int(args[6]) == 3
    if f"conv_stage_kernel<8, {nsrc}, {ks}, " in name:
        args = name[name.index("<") + 1:name.rindex(">")].split(", ")
        return int(args[6]) == prec and (len(args) < 10 or int(args[9]) == 3)
    return False


def _profile_json(name):
    try:
        return json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None


def pmc_entry(stage, H, W, precision):
    """The committed rocprofv3 record of one stage kernel on the headline workload (profiles/pmc_latest.json =
    scripts/profile.sh of this same command: kernel-trace durations incl. the median, and the PMC passes --
    FETCH_SIZE x2 gfx950 correction + WRITE_SIZE, collected in separate runs).  Only valid for 1920x1080."""
    if (H, W) != (1080, 1920):
        return None
    d = _profile_json("pmc_latest.json")
    if not d:
        return None
    names = [n for n in d if kernel_matches(n, stage, precision)]
    names.sort(key=lambda n: 0 if "pipe" in n else 1)  # the form the engine runs at this size
    return d[names[0]] if names else None


def pmc_traffic(stage, H, W, precision="f32"):
    v = pmc_entry(stage, H, W, precision)
    if not v or "hbm_read_bytes" not in v:
        return None
    return {"hbm_bytes_per_launch": int(v["hbm_read_bytes"] + v.get("hbm_write_bytes", 0)),
            "algorithmic_bytes_per_launch": int(H * W * 128 * (min(stage, 3) + 1)),
            "source": "profiles/pmc_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"}


def n1_reference(ms_c, world, precision, io):
    """config_C at N > 1: speed-up over the committed one-GPU time of the same 3840x2160 image (profiles/r5_bench.json, else
    an earlier round's record) -- measured on another box of the same kind, so good to the box-to-box spread (~1.5 %)."""
    if world == 1:
        return {}
    for name in ("r5_bench.json", "r4_bench.json", "r3_bench.json", "r2_bench.json"):
        d = _profile_json(name)
        ref = d and d.get("config_C", {})
        same = d and d.get("config", {}).get("precision") == precision and d.get("config", {}).get("io") == io
        if ref and same and "ms_per_step" in ref and "error" not in ref:
            return {"speedup_vs_n1": round(ref["ms_per_step"] / ms_c, 3), "efficiency_vs_n1": round(ref["ms_per_step"] / ms_c / world, 4),
                    "n1_reference": {"ms_per_step": ref["ms_per_step"], "source": f"profiles/{name} (N = 1 run of this command)"}}
    return {"speedup_vs_n1": None}


def stage_tflops(stage, rows, W, ms):
    return 2 * MAC_PER_PX[stage] * rows * W / (ms / 1e3) / 1e12


def roofline_of(stage_ms, rows, W, precision):
    """Roofline block of the dominant stage kernel from per-stage HIP-event times (ms) of one call."""
    k = int(np.argmax(stage_ms))
    ach = stage_tflops(k, rows[k], W, stage_ms[k])
    peak = PEAK_F32_MFMA_TFLOPS if precision == "f32" else PEAK_F16_MFMA_TFLOPS
    return k, ach, peak


def read_sclk_mhz(device=0):
    """Current shader clock as the driver reports it (sysfs pp_dpm_sclk: the level marked '*'), or None.  A GPU pod sees the sysfs
    nodes of every card of its host but owns one; the others sleep (level 'S', ~95 MHz), so the busiest card's clock is reported."""
    import glob
    best = None
    for path in glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"):
        try:
            for line in open(path):
                if "*" in line:
                    mhz = int("".join(ch for ch in line.split(":")[1] if ch.isdigit()))
                    best = mhz if best is None else max(best, mhz)
        except Exception:
            pass
    return best


def aux_entries(r, torch, sizes=((1080, 1920),), reps=200, device=0):
    """The two parameter-free graphs of the reference's upscale() (bilinear_net / downsample_net, network.rs:111-138; sr_aux.hip)
    on device-resident images: ms per call, GB/s of COMPULSORY I/O (input once + output once), fraction of the HBM roof (8 TB/s
    nominal; 6.3 TB/s is what MI355X_MICROARCH.md measures for a copy)."""
    out = []
    engines = {"bilinear": r.Engine(graph="bilinear", device=device), "downsample": r.Engine(graph="downsample", device=device)}
    for (H, W) in sizes:
        px = synth_u8(7, H, W)
        for graph, eng in engines.items():
            for io in ("rgba8", "f32"):
                if io == "rgba8":
                    x = torch.from_numpy(px).cuda(device)[None]
                    fn, in_b, out_b = eng.upscale_rgba8_dev, 3, 4
                else:
                    x = torch.from_numpy(r.img_to_data(px)).cuda(device)[None]
                    fn, in_b, out_b = eng.upscale_f32_dev, 12, 12
                o = fn(x)
                for _ in range(10):
                    fn(x, out=o)
                torch.cuda.synchronize()
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        fn(x, out=o)
                    torch.cuda.synchronize()
                    best = min(best, (time.perf_counter() - t0) / reps * 1e3)
                nbytes = H * W * in_b + o.shape[1] * o.shape[2] * out_b
                gbps = nbytes / (best / 1e3) / 1e9
                out.append({"graph": graph + "_net", "io": io, "image": [H, W], "ms": round(best, 5), "io_bytes": nbytes,
                            "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / PEAK_HBM_GBPS, 4),
                            "frac_of_6.3TBps_copy_roof": round(gbps / 6300.0, 4)})
                del o, x
    for e in engines.values():
        e.close()
    return out


def usable_cpus():
    """CPUs this process may actually use: the scheduler affinity, capped by the cgroup CPU quota (a GPU pod usually owns a
    fraction of its host: the round-2 box reports 256 logical CPUs and a quota of 16).  Oversubscribing the quota with 256
    OpenMP threads gets the process throttled, so the CPU baseline runs on this many threads and says so."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:
            pass
    if quota:
        n = max(1, min(n, int(quota + 0.999)))
    return n, quota


def cpu_baseline(params, px_u8, budget_s=10.0):
    """The reference's CPU path cannot be built here (Rust, un-vendored crates): two ports of it are timed on this
    host instead, on a bounded sample of the same workload -- (a) the C oracle (a correctness oracle: fixed summation
    order, no FMA; all threads and one thread), (b) the same graph on torch-CPU / oneDNN with all cores
    (oracle/torch_ref.py; checked against (a) in tests/test_cpu_baseline.py).  `value` is the FASTER of the two."""
    import ctypes
    import oracle
    h, w, _ = px_u8.shape
    cores, quota = usable_cpus()
    gomp = None
    try:
        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(cores)
    except Exception:
        pass
    x = oracle.img_to_data(px_u8)
    oracle.forward(params, x[:32], native=True)  # warm-up: builds the -march=native copy, spins up OpenMP
    t0 = time.perf_counter()
    oracle.forward(params, x[:64], native=True)
    t64 = time.perf_counter() - t0
    rows = int(min(h, max(64, 64 * budget_s / max(t64, 1e-6) / 2)))
    oracle.forward(params, x[:rows], native=True)  # first pass at this size grows / faults in the workspace
    t0 = time.perf_counter()
    oracle.forward(params, x[:rows], native=True)
    dt = time.perf_counter() - t0
    try:
        model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    legs = {"c_oracle": {"value": round(rows * w * 9 / 1e6 / dt, 3), "unit": "output MP/s", "cores": cores,
                         "sample": f"top {rows} rows of the {w}x{h} workload image, f32 in/out, second of two passes, OpenMP on {cores} "
                                   f"threads = the CPUs this container may use (oracle/sr_oracle.c, gcc -O3 -march=native -ffp-contract=off); "
                                   f"GFLOP/s={rows * w * FLOP_PER_PX / dt / 1e9:.1f}"}}
    try:  # also one thread (closest to what alumina 0.1.1 does for n = 1), on a ~4 s sample
        gomp.omp_set_num_threads(1)
        try:
            t0 = time.perf_counter()
            oracle.forward(params, x[:8], native=True)
            t8 = time.perf_counter() - t0
            r1 = int(min(h, max(8, 8 * 4.0 / max(t8, 1e-6))))
            t0 = time.perf_counter()
            oracle.forward(params, x[:r1], native=True)
            d1 = time.perf_counter() - t0
            legs["c_oracle_1thread"] = {"value": round(r1 * w * 9 / 1e6 / d1, 4), "unit": "output MP/s", "cores": 1,
                                        "sample": f"top {r1} rows, one OpenMP thread; GFLOP/s={r1 * w * FLOP_PER_PX / d1 / 1e9:.2f}"}
        finally:
            gomp.omp_set_num_threads(cores)
    except Exception:
        pass
    try:  # the tuned-library leg: torch-CPU conv2d (oneDNN), every core, whole frame
        import torch
        from oracle.torch_ref import TorchNet
        net = TorchNet(params)
        torch.set_num_threads(cores)
        nthr = torch.get_num_threads()
        net.forward(x[None, :128])
        t0 = time.perf_counter()
        net.forward(x[None, :256])
        t256 = time.perf_counter() - t0
        rows_t = int(min(h, max(256, 256 * budget_s / max(t256, 1e-6) / 3)))
        net.forward(x[None, :rows_t])
        best = 1e9
        for _ in range(2):
            t0 = time.perf_counter()
            net.forward(x[None, :rows_t])
            best = min(best, time.perf_counter() - t0)
        legs["torch_cpu"] = {"value": round(rows_t * w * 9 / 1e6 / best, 3), "unit": "output MP/s", "cores": nthr,
                             "sample": f"top {rows_t} rows, torch {torch.__version__} CPU conv2d (oneDNN, channels_last), {nthr} threads, "
                                       f"best of 2 after warm-up; GFLOP/s={rows_t * w * FLOP_PER_PX / best / 1e9:.1f}"}
    except Exception as ex:
        legs["torch_cpu"] = {"error": str(ex)[:200]}
    best_leg = max((k for k in ("c_oracle", "torch_cpu") if "value" in legs[k]), key=lambda k: legs[k]["value"])
    return {"value": legs[best_leg]["value"], "unit": "output MP/s", "cores": legs[best_leg]["cores"], "kind": "port",
            "leg": best_leg, "sample": legs[best_leg]["sample"], "legs": legs, "cpu": model,
            "host_logical_cpus": os.cpu_count(), "cgroup_cpu_quota": quota,
            "single_thread": legs.get("c_oracle_1thread"),
            "scaling_c_oracle": (round(legs["c_oracle"]["value"] / legs["c_oracle_1thread"]["value"], 1)
                                 if "c_oracle_1thread" in legs else None)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--io", choices=["rgba8", "f32"], default="rgba8")
    ap.add_argument("--weights", default="imagenet")
    ap.add_argument("--precision", choices=["f32", "split_f16"], default=os.environ.get("SRHIP_PRECISION", "f32"),
                    help="f32 (default, the headline `value`): exact-f32 MFMA, the reference's own arithmetic class.  "
                         "split_f16: hi/lo half pairs, 3 f16 MFMAs per product on the matrix cores -- 2x faster and inside "
                         "the same north-star parity bar (<= 1e-4; measured <= 2e-5, every GPU parity test runs in both "
                         "modes); at N=1 it is measured in the same run and reported as `other_precision`.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the config_A / config_C / config_D entries")
    args = ap.parse_args()

    # (before the HSA runtime comes up: the host driver only supports dmabuf IPC, RCCL's peer connections need this)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import rusty_sr_amd as r
    from rusty_sr_amd.shard import BandExchange, init_band_comm, round_robin, split_rows

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; rusty_sr_amd has no CPU fallback")
    # Rehearsal on a box with fewer GPUs than ranks (not a measurement): SRHIP_SHARE_GPU=1 folds the ranks
    # onto the devices present and SRHIP_DIST_BACKEND=gloo replaces RCCL, which refuses two ranks per device.
    backend = os.environ.get("SRHIP_DIST_BACKEND", "nccl")
    shared = os.environ.get("SRHIP_SHARE_GPU") == "1"
    if shared:
        local %= torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    H, W = args.height, args.width
    u8 = args.io == "rgba8"
    dt_in = torch.uint8 if u8 else torch.float32
    params = r.rsr.builtin(args.weights)
    eng = r.Engine(params, device=local, precision=args.precision)

    # ---- how halos travel: libsrhip's own RCCL communicator (the product path), or torch.distributed P2P
    # (rehearsals on a shared GPU / gloo, and the fallback should RCCL inside the library fail to come up)
    exchange = "none"
    if world > 1:
        exchange = "torch.distributed P2P (BandExchange)"
        if backend == "nccl" and not shared and os.environ.get("SRHIP_EXCHANGE", "lib") == "lib":
            try:
                init_band_comm(eng, rank, world)
                exchange = "libsrhip RCCL communicator (sr_upscale_sharded_*_dev: grouped ncclSend/ncclRecv)"
            except Exception as ex:  # noqa: BLE001 -- report and fall back rather than lose the run
                exchange += f" [libsrhip communicator failed: {str(ex)[:120]}]"
        flags = [exchange.startswith("libsrhip")]
        allf = [None] * world
        dist.all_gather_object(allf, flags[0])
        if not all(allf):  # every rank must take the same path
            if exchange.startswith("libsrhip"):
                eng.comm_init_rank(b"", 0, 1)
                exchange = "torch.distributed P2P (BandExchange) [another rank fell back]"
    use_lib = exchange.startswith("libsrhip")

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup):
        """W untimed steps, then exactly K steps between barrier + synchronize fences; max over ranks; ms per step."""
        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt / steps * 1e3

    class Band:
        """This rank's row band of an image sharded over the ranks, resident in HBM, and its step function."""

        def __init__(self, px_band, lib=None):
            lib = use_lib if lib is None else lib
            self.lib = lib
            hb, w = px_band.shape[:2]
            self.hb, self.w = hb, w
            src = torch.from_numpy(px_band).to(dev)
            if not u8:
                src = torch.from_numpy(r.img_to_data(px_band)).to(dev)
            self.out = torch.empty((3 * hb, 3 * w, 4 if u8 else 3), dtype=dt_in, device=dev)
            self.top = 7 if rank > 0 else 0
            self.bot = 7 if rank < world - 1 else 0
            if lib or world == 1:
                self.band = src.contiguous()
                self.xchg = None
            else:
                self.xchg = BandExchange(hb, w, 3, dt_in, dev, rank, world)
                self.xchg.band.copy_(src)

        def step(self):
            if self.xchg is None:
                if world == 1:
                    (eng.upscale_rgba8_dev if u8 else eng.upscale_f32_dev)(self.band[None], out=self.out[None])
                else:
                    eng.upscale_sharded_dev(self.band, out=self.out)
            else:
                ext = self.xchg.exchange()
                (eng.upscale_band_rgba8_dev if u8 else eng.upscale_band_f32_dev)(ext, self.top, self.bot, out=self.out)

        def stage_ms(self, reps):
            """Per-stage kernel times (HIP events on the launch stream, inside libsrhip); median over reps."""
            eng.set_profiling(True)
            acc, comm, expo = [], [], []
            for _ in range(reps):
                self.step()
                torch.cuda.synchronize()
                acc.append(eng.last_timing()["stage_ms"])
                comm.append(eng.last_comm_ms() if self.lib and world > 1 else 0.0)
                expo.append(eng.last_comm_exposed_ms() if self.lib and world > 1 else 0.0)
            eng.set_profiling(False)
            # (interior first: the exchange runs on its own stream beside the band copy and stage 0's interior rows; this is what the band's
            # stream still had to wait for it)
            self.comm_exposed_ms = float(np.median(expo))
            self.stage_mean = np.mean(np.array(acc), axis=0)   # beside the median: what `roofline.frac` is quoted on
            return np.median(np.array(acc), axis=0), float(np.median(comm))

        def rows(self):
            if LAYER_HALOS and self.lib:
                return [self.hb] * 5
            return [min(self.hb + self.top + self.bot, self.hb + (min(m, self.top) + min(m, self.bot))) for m in MARGIN]

    # ------------------------------------------------------------------ the `value` workload (weak scaling)
    px = synth_u8(2 + rank, H, W)  # seed 2 = SURVEY.md 8(d) config B; other ranks' bands differ
    exchange_check = None
    if use_lib:
        # The library's RCCL exchange is checked LIVE before it is timed: one step through it and one through torch.distributed's
        # point-to-point calls (the same band, the same kernels, only the halos travel differently) must give the same bytes on every
        # rank.  If not, every rank measures the torch path and the line says so.  A deadline stands behind the first exchange: ranks
        # that wait for each other for ever would otherwise cost the whole run its line.
        limit = float(os.environ.get("SRHIP_BENCH_EXCHANGE_LIMIT_S", "120"))

        def stuck():
            msg = f"rank {rank}: the first sharded step through libsrhip's RCCL communicator did not return within {limit:.0f} s"
            print(msg, file=sys.stderr, flush=True)
            if rank == 0:
                print(json.dumps({"metric": "output megapixels/sec at 3x upscale", "value": None, "unit": "output MP/s", "n_gpus": world,
                                  "error": msg + " (SRHIP_EXCHANGE=torch selects torch.distributed's point-to-point calls instead)"}), flush=True)
            os._exit(3)

        guard = threading.Timer(limit, stuck)
        guard.daemon = True
        guard.start()
        # The deadline stays armed until every rank has reported: a rank that throws before the gather must not leave its peers
        # waiting in the torch step or in the gather with nothing behind them.  Each rank contributes "identical", "different" or the
        # exception it met; anything but identical on every rank sends ALL ranks to the torch path together.
        try:
            try:
                via_lib, via_torch = Band(px, lib=True), Band(px, lib=False)
                via_lib.step()
                torch.cuda.synchronize()
                via_torch.step()
                torch.cuda.synchronize()
                mine = "identical" if torch.equal(via_lib.out, via_torch.out) else "different"
                del via_lib, via_torch
            except Exception as ex:  # noqa: BLE001 -- reported through the gather, so that every rank learns of it
                mine = "error: " + str(ex)[:160]
            same = [None] * world
            dist.all_gather_object(same, mine)
            ok = all(v == "identical" for v in same)
            exchange_check = {"identical_to_torch_p2p_on_every_rank": ok, "per_rank": same}
            if not ok:
                eng.comm_init_rank(b"", 0, 1)
                use_lib = False
                exchange = "torch.distributed P2P (BandExchange) [libsrhip's exchange failed its live check: see exchange_check]"
        except Exception as ex:  # noqa: BLE001 -- the gather itself failed: the check must not cost the run its line
            exchange_check = {"error": str(ex)[:200]}
        finally:
            guard.cancel()
        torch.cuda.empty_cache()
    main_band = Band(px)
    # set-up, not steps: the first calls allocate the workspace (4 feature maps), zero its borders
    # and bring the clocks up; with N > 1 RCCL's lazy P2P connection set-up also happens here
    for _ in range(3):
        main_band.step()
    torch.cuda.synchronize()
    ms_per_step = timed(main_band.step, args.steps, args.warmup)
    out_mp_total = world * (3 * H) * (3 * W) / 1e6
    value = out_mp_total / (ms_per_step / 1e3)

    result = {
        "metric": "output megapixels/sec at 3x upscale (BASELINE '4x'; reference factor is hard-wired 3)"
                  + ("; N > 1: `value` is WEAK scaling (one 1920x1080 band per GPU of a 1920x(1080 N) image) -- the strong-scaling "
                     "answer for one fixed image is `config_C` (3840x2160 over N GPUs, `speedup_vs_n1`)" if world > 1 else ""),
        "value": round(value, 2), "unit": "output MP/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None,
        **({"rehearsal": f"backend={backend}, ranks folded onto {torch.cuda.device_count()} device(s): not a measurement"}
           if world > 1 and (backend != "nccl" or shared) else {}),
        "dtype": "f32" if args.precision == "f32" else "f16x3 split (hi/lo half pairs, f32 accumulate)", "data": "synthetic",
        "config": {"workload": f"{W}x{H} RGB x3 upscale per GPU, {args.weights}.rsr, {args.io} in/out resident in HBM"
                               + (f"; WEAK scaling: {world} row bands of one {W}x{H * world} image, 7-row RCCL halo exchange per step "
                                  f"(strong scaling of a fixed 3840x2160 image: see config_C)" if world > 1 else ""),
                   "io": args.io, "image": [H * world, W], "factor": 3, "parallelism": f"rowband{world}",
                   "precision": args.precision, **({"exchange": exchange} if world > 1 else {})},
        "tflops": round(world * H * W * FLOP_PER_PX / (ms_per_step / 1e3) / 1e12, 2),
        "whole_call_frac": round(H * W * FLOP_PER_PX / (ms_per_step / 1e3) / 1e12 /
                                 (PEAK_F32_MFMA_TFLOPS if args.precision == "f32" else PEAK_F16_MFMA_TFLOPS), 4),
    }

    if world == 1 and not args.no_roofline:
        # ---- sustained: the same step looped for >= 3 s of wall time, every step bracketed by its own pair of events on the
        # launch stream (20 steps are 80 ms: too short for the clocks to settle, or for an SMI sampler to see the GPU busy)
        try:
            secs = float(os.environ.get("SRHIP_BENCH_SUSTAINED_S", "3.0"))
            sclk0 = read_sclk_mhz(local)
            evs, t_start, wall = [], time.perf_counter(), 0.0
            sclk_mid = []
            while time.perf_counter() - t_start < secs:
                # bursts of 100 steps, each burst fenced and timed on its own: reading the clock between bursts (a few file reads, ~1 ms)
                # is not part of any burst's wall time
                t0 = time.perf_counter()
                for _ in range(100):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    main_band.step()
                    e1.record()
                    evs.append((e0, e1))
                torch.cuda.synchronize()
                wall += time.perf_counter() - t0
                sclk_mid.append(read_sclk_mhz(local))
            per = np.array([a.elapsed_time(b) for a, b in evs])
            sclk_mid = [v for v in sclk_mid if v]
            result["sustained"] = {"ms_per_step": round(wall / len(per) * 1e3, 4), "steps": len(per), "wall_s": round(time.perf_counter() - t_start, 3), "busy_wall_s": round(wall, 3),
                                   "event_ms_median": round(float(np.median(per)), 4), "event_ms_mean": round(float(per.mean()), 4),
                                   "event_ms_p95": round(float(np.percentile(per, 95)), 4), "event_ms_min": round(float(per.min()), 4),
                                   "vs_headline": round(wall / len(per) * 1e3 / ms_per_step, 4),
                                   "sclk_mhz_before": sclk0, "sclk_mhz_under_load": (int(np.median(sclk_mid)) if sclk_mid else None),
                                   "sclk_mhz_under_load_min": (min(sclk_mid) if sclk_mid else None),
                                   "whole_call_frac": round(H * W * FLOP_PER_PX / (wall / len(per)) / 1e12 /
                                                            (PEAK_F32_MFMA_TFLOPS if args.precision == "f32" else PEAK_F16_MFMA_TFLOPS), 4),
                                   "note": "ms_per_step = wall / steps over fenced bursts of 100 back-to-back steps (what `value` measures, for >= 3 s instead of 80 ms); "
                                           "event_* = per-step durations from an event pair around each call on the launch stream; sclk from sysfs "
                                           "pp_dpm_sclk, read between bursts"}
            del evs
        except Exception as ex:  # noqa: BLE001
            result["sustained"] = {"error": str(ex)[:200]}
        # ---- what the fork inside the device call is worth on this box: the same 20 steps with it off, then on again
        # (not when the caller pins the mode with SRHIP_FORK, as scripts/profile.sh does to keep every profiled launch a whole frame)
        try:
            if "SRHIP_FORK" in os.environ:
                raise RuntimeError("skipped: SRHIP_FORK is set")
            ab = {}
            for key, val in (("undivided_ms", "0"), ("forked_ms", "1"), ("automatic_ms", "")):
                eng.set_experiment("fork", val)
                ab[key] = round(timed(main_band.step, args.steps, args.warmup), 4)
            eng.set_experiment("fork", "")
            result["config"]["device_call"] = ("sr_upscale_*_dev: one image as two row bands (+7 halo rows each, bit-identical) forked onto the context's "
                                               "second stream and joined back by events where the rounds of tiles allow (sr_run_stack_auto); asynchronous on the caller's stream")
            result["fork_ab"] = dict(ab, note="same step, sr_set_experiment('fork', '0' / '1' / ''): what overlapping the two bands' launch tails is worth here")
        except Exception as ex:  # noqa: BLE001
            result["fork_ab"] = {"error": str(ex)[:200]}

    stage_ms = None
    if not args.no_roofline:
        # dominant kernel = stage 3 (l3 node: conv3 5x5 + conv6 3x3 + conv8 3x3, K = 1376);
        # per-launch duration from HIP events recorded on the launch stream around each stage.
        stage_ms, comm_ms = main_band.stage_ms(max(10, min(2 * args.steps, 40)))
        stage_mean = main_band.stage_mean
        rows = main_band.rows() if world > 1 else [H] * 5
        k, ach, peak = roofline_of(stage_ms, rows, W, args.precision)
        if rank == 0:
            if args.precision == "f32":
                note = "v_mfma_f32_32x32x2_f32; algorithmic FLOPs = issued FLOPs"
            else:
                note = ("v_mfma_f32_16x16x32_f16 (stages 1-3; the last stage 32x32x16), 3 products per algorithmic product: achieved counts ALGORITHMIC FLOPs "
                        f"against the f16 dense peak (issued rate {3 * ach:.1f} TFLOP/s); the ceiling of this scheme is peak/3")
            pe = pmc_entry(k, H, W, args.precision) if world == 1 else None
            result["roofline"] = {
                "bound": "mfma",
                "kernel": (f"stage {k} (conv_stage_pipe_kernel<{STAGE_SHAPE[k][0]}, {STAGE_SHAPE[k][1]}, ...>)"
                           if k else "conv0_kernel"),
                # `achieved` / `frac`: algorithmic FLOPs of one launch / the AVERAGE launch duration of this run (the conservative
                # figure: a few slow launches -- clock dips -- pull the mean up); the median beside it
                "achieved": round(stage_tflops(k, rows[k], W, stage_mean[k]), 2), "peak": peak, "unit": "TFLOP/s",
                "frac": round(stage_tflops(k, rows[k], W, stage_mean[k]) / peak, 4),
                "frac_at_median": round(ach / peak, 4), "achieved_at_median": round(ach, 2),
                "traffic": pmc_traffic(k, H, W, args.precision) if world == 1 else None,
                "avg_launch_ms": round(float(stage_mean[k]), 4), "median_launch_ms": round(float(stage_ms[k]), 4),
                "launch": "one launch = the whole frame (the per-stage timing pass runs the call undivided; the headline call runs "
                          "the same kernel as two half-frame launches on two streams where that pays, see config.device_call)",
                "source": "hip_events on the launch stream, this run: mean and median over %d launches" % max(10, min(2 * args.steps, 40)),
                "rocprof": ({"median_ms": pe.get("median_us") and round(pe["median_us"] / 1e3, 4),
                             "avg_ms": pe.get("avg_us") and round(pe["avg_us"] / 1e3, 4),
                             "frac": pe.get("avg_us") and round(stage_tflops(k, H, W, pe["avg_us"] / 1e3) / peak, 4),
                             "frac_at_median": pe.get("median_us") and round(stage_tflops(k, H, W, pe["median_us"] / 1e3) / peak, 4),
                             "mfma_util_pmc": pe.get("mfma_util") and round(pe["mfma_util"], 4),
                             "source": "profiles/pmc_latest.json (rocprofv3 --kernel-trace of this command, committed)"}
                            if pe else None),
                "note": note}
            result["stages"] = [{"stage": s, "ms": round(float(stage_ms[s]), 4), "mean_ms": round(float(stage_mean[s]), 4),
                                 "tflops": round(stage_tflops(s, rows[s], W, stage_ms[s]), 2),
                                 "frac_of_peak": round(stage_tflops(s, rows[s], W, stage_ms[s]) / peak, 4)} for s in range(5)]
            if world > 1:
                result["comm_ms"] = round(comm_ms, 4)
                result["comm_exposed_ms"] = round(main_band.comm_exposed_ms, 4)
                if exchange_check is not None:
                    result["exchange_check"] = exchange_check
            io_bytes = H * W * (3 + 36 if u8 else 12 + 108)
            hbm = {"algorithmic_GBps": round(io_bytes / (ms_per_step / 1e3) / 1e9, 2), "peak_GBps": PEAK_HBM_GBPS,
                   "note": "algorithmic = compulsory image I/O only; the path is MFMA-bound (2170 FLOP/B)"}
            if world == 1:
                ents = [pmc_entry(s, H, W, args.precision) for s in range(5)]
                if all(e and "hbm_read_bytes" in e for e in ents):
                    tot = sum(e["hbm_read_bytes"] + e.get("hbm_write_bytes", 0) for e in ents)
                    hbm.update({"counter_bytes_per_call": int(tot), "counter_GBps": round(tot / (ms_per_step / 1e3) / 1e9, 1),
                                "counter_frac_of_peak": round(tot / (ms_per_step / 1e3) / 1e9 / PEAK_HBM_GBPS, 4),
                                "counter_source": "profiles/pmc_latest.json: FETCH_SIZE x2 + WRITE_SIZE summed over the five stage kernels"})
            result["hbm"] = hbm
            result["device"] = eng.device_info()

    if rank == 0 and world == 1 and not args.no_roofline:
        # the other arithmetic mode on the same resident image, same number of steps
        other = "f32" if args.precision == "split_f16" else "split_f16"
        eng.set_precision(other)
        for _ in range(args.warmup):
            main_band.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            main_band.step()
        torch.cuda.synchronize()
        oms = (time.perf_counter() - t0) / args.steps * 1e3
        ost, _ = main_band.stage_ms(3)
        eng.set_precision(args.precision)
        peak_o = PEAK_F32_MFMA_TFLOPS if other == "f32" else PEAK_F16_MFMA_TFLOPS
        ach_o = stage_tflops(3, H, W, ost[3])
        result["other_precision"] = {"precision": other, "value": round((3 * H) * (3 * W) / 1e6 / (oms / 1e3), 2),
                                     "unit": "output MP/s", "ms_per_step": round(oms, 4),
                                     "stage_ms": [round(float(v), 4) for v in ost],
                                     "stage3_tflops": round(ach_o, 2), "stage3_frac_of_peak": round(ach_o / peak_o, 4),
                                     "peak": peak_o}
        if other == "split_f16":
            # The split-half mode issues THREE f16 products per algorithmic one (hi.hi, hi.lo, lo.hi), and the f16 matrix pipe does not hold
            # its nominal clock on real operands: a bare stream of random-operand v_mfma_f32_16x16x32_f16 -- what stages 1-3 issue -- runs at
            # 1.96 GHz = 1 967 TFLOP/s on this part while the mode's own frames run at 1.72 GHz (profiles/r6_power_clock_probe.txt: clock from
            # a one-wave probe kernel beside 20 s of the real frames).  Both denominators, so that neither has to be taken on trust.
            issued = 3.0 * ach_o
            result["other_precision"]["roofline"] = {
                "bound": "mfma", "kernel": "stage 3 (conv_stage_pipe_kernel<3, 5, ..., 1, 3>, v_mfma_f32_16x16x32_f16)", "unit": "TFLOP/s",
                "algorithmic": round(ach_o, 2), "issued": round(issued, 2),
                "peak_nominal": PEAK_F16_MFMA_TFLOPS, "frac_of_nominal_issued": round(issued / PEAK_F16_MFMA_TFLOPS, 4),
                "peak_measured_random_operands": MEASURED_F16_MFMA16_RANDOM_TFLOPS,
                "frac_of_measured_issued": round(issued / MEASURED_F16_MFMA16_RANDOM_TFLOPS, 4),
                "source": "profiles/r6_power_clock_probe.txt (scripts/experiments/power_clock_probe.hip, third leg; round 5's boxes: 2 090, "
                          "profiles/r5_ubench_split_floor.txt)"}

    if rank == 0 and world == 1 and not args.no_roofline:
        # BASELINE.json labels its configs "4x".  The reference cannot do 4x (FACTOR = 3, no 4x weights);
        # sr_net(4) with seeded synthetic weights is timed here as an extra: 48 expand channels.
        try:
            n4 = r._lib.lib().sr_num_params_factor(4)
            p4 = (np.random.default_rng(4).standard_normal(n4) * 0.03).astype(np.float32)
            extra = {}
            for prec in ("f32", "split_f16"):
                e4 = r.Engine(p4, device=local, factor=4, precision=prec)
                xin = main_band.band.contiguous()[None]
                fn = e4.upscale_rgba8_dev if u8 else e4.upscale_f32_dev
                o4 = fn(xin)
                for _ in range(3):
                    fn(xin, out=o4)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(args.steps):
                    fn(xin, out=o4)
                torch.cuda.synchronize()
                ms4 = (time.perf_counter() - t0) / args.steps * 1e3
                extra[prec] = {"ms_per_step": round(ms4, 4), "value": round(16 * H * W / 1e6 / (ms4 / 1e3), 2),
                               "unit": "output MP/s at 4x"}
                e4.close()
                del o4
            result["x4_synthetic_weights"] = dict(extra, note="sr_net(4), seeded synthetic parameters (no 4x weights exist in the "
                                                  "reference); parity for this factor is against the CPU restatement only")
        except Exception as ex:  # never let the extra break the contract line
            result["x4_synthetic_weights"] = {"error": str(ex)}

    # ------------------------------------------------------------------ the other named configurations
    # N > 1: the entries below contain collectives of their own (exchanges, barriers).  Should one rank fail where the
    # others do not, they would wait for it for ever and the line above would be lost with them: a watchdog thread
    # prints what is complete and ends the process instead.
    watchdog = None
    if world > 1 and not args.no_configs:
        limit = float(os.environ.get("SRHIP_BENCH_EXTRAS_LIMIT_S", "150"))

        def give_up():
            if rank == 0:
                line = dict(result)
                line["watchdog"] = f"config_C / config_D did not finish within {limit:.0f} s; every other field is complete"
                print(json.dumps(line), flush=True)
            os._exit(0)

        watchdog = threading.Timer(limit, give_up)
        watchdog.daemon = True
        watchdog.start()
    del main_band
    torch.cuda.empty_cache()
    peak_here = PEAK_F32_MFMA_TFLOPS if args.precision == "f32" else PEAK_F16_MFMA_TFLOPS
    ksteps = max(3, min(args.steps, 10))
    if not args.no_configs:
        try:
            # config C: one 3840x2160 image.  N = 1: the whole image; N > 1: strong scaling, band r on rank r.
            HC, WC = 2160, 3840
            a, b = split_rows(HC, world)[rank]
            band = Band(synth_u8(3, HC, WC)[a:b])  # seed 3 = SURVEY.md 8(d) config C (every rank builds the image, keeps its rows)
            for _ in range(2):
                band.step()
            torch.cuda.synchronize()
            ms_c = timed(band.step, ksteps, 1)
            st_c, comm_c = band.stage_ms(3)
            rows_c = band.rows() if world > 1 else [HC] * 5
            kc, ach_c, _ = roofline_of(st_c, rows_c, WC, args.precision)
            mine = {"rank": rank, "rows": b - a, "stage_ms": [round(float(v), 4) for v in st_c], "comm_ms": round(comm_c, 4),
                    "comm_exposed_ms": round(band.comm_exposed_ms, 4), "roofline_frac": round(ach_c / peak_here, 4), "dominant_stage": kc}
            per_rank = [mine]
            if world > 1:
                per_rank = [None] * world
                dist.all_gather_object(per_rank, mine)
            preview = None
            if world == 1:
                # what ONE rank of an N-way split would do per frame: an interior band + 7 halo rows either side on this GPU
                # (the exchange itself needs N GPUs; everything else of a rank's step is measured here).  8-way = BASELINE configs[3].
                preview = {"note": "one interior band of the N-way split on this GPU, without the exchange: a projection of the per-rank "
                                   "step, not a measurement of N GPUs; useful_roofline_frac counts the band's OWN rows only (recomputed "
                                   "halo rows are overhead)"}
                img_c = synth_u8(3, HC, WC)
                for ways in (2, 4, 8):
                    rows_b = HC // ways
                    a0 = rows_b * (ways // 2)
                    topb, botb = 7, (7 if ways > 2 else 0)
                    ext = torch.from_numpy(np.ascontiguousarray(img_c[a0 - topb:a0 + rows_b + botb])).to(dev)
                    if not u8:
                        ext = torch.from_numpy(r.img_to_data(ext.cpu().numpy())).to(dev)
                    ob = torch.empty((3 * rows_b, 3 * WC, 4 if u8 else 3), dtype=dt_in, device=dev)
                    fnb = eng.upscale_band_rgba8_dev if u8 else eng.upscale_band_f32_dev
                    # (the plan a rank of the real split runs: its sharded call follows the fork RULE -- the halo gate takes a band out of the
                    # fork tuner's hands -- so the projection does too)
                    eng.set_experiment("forktune", "0")
                    for _ in range(3):
                        fnb(ext, topb, botb, out=ob)
                    torch.cuda.synchronize()
                    best = 1e9
                    for _ in range(3):
                        t0 = time.perf_counter()
                        for _ in range(ksteps):
                            fnb(ext, topb, botb, out=ob)
                        torch.cuda.synchronize()
                        best = min(best, (time.perf_counter() - t0) / ksteps * 1e3)
                    # ... and with per-layer feature halos (sr_set_experiment "halo" = "layers"): every stage computes the
                    # band's own rows only -- the kernel work of a whole image of that many rows, which is what is timed here (the four
                    # feature-row exchanges per frame, <= 1 MB each, need the neighbours and are not in it)
                    own = ext[topb:topb + rows_b].contiguous()[None]
                    fno = eng.upscale_rgba8_dev if u8 else eng.upscale_f32_dev
                    for _ in range(3):
                        fno(own, out=ob[None])
                    torch.cuda.synchronize()
                    best_l = 1e9
                    for _ in range(3):
                        t0 = time.perf_counter()
                        for _ in range(ksteps):
                            fno(own, out=ob[None])
                        torch.cuda.synchronize()
                        best_l = min(best_l, (time.perf_counter() - t0) / ksteps * 1e3)
                    preview[f"{ways}_way"] = {
                        "rows": rows_b, "halo_rows": topb + botb, "ms_per_band": round(best, 4),
                        "ideal_ms": round(ms_c / ways, 4), "over_ideal": round(best / (ms_c / ways), 4),
                        "useful_roofline_frac": round(rows_b * WC * FLOP_PER_PX / (best / 1e3) / 1e12 / peak_here, 4),
                        "n_bands_in_parallel_would_be": round(9 * HC * WC / 1e6 / (best / 1e3), 1),
                        "layer_halos": {"kernel_ms_per_band": round(best_l, 4),
                                        "useful_roofline_frac": round(rows_b * WC * FLOP_PER_PX / (best_l / 1e3) / 1e12 / peak_here, 4),
                                        "exchanged_bytes_per_neighbour": int((2 + 3) * (32 * ((WC + 31) // 32) + 4) * 128)}}
                    del ext, ob, own
                    eng.set_experiment("forktune", "")
                preview["rows"], preview["halo_rows"] = 270, 14  # (the 8-way entry under its round-2 keys)
                preview["ms_per_band"] = preview["8_way"]["ms_per_band"]
                preview["eight_bands_in_parallel_would_be"] = preview["8_way"]["n_bands_in_parallel_would_be"]
            if rank == 0:
                result["config_C"] = {
                    **({"band_preview": preview} if preview else {}),
                    "workload": f"3840x2160 RGB x3, {args.io}, resident in HBM" + (f", {world} row bands of {HC // world} rows + 7-row halos "
                                f"(strong scaling; exchange: {exchange})" if world > 1 else ", one GPU"),
                    "value": round(9 * HC * WC / 1e6 / (ms_c / 1e3), 2), "unit": "output MP/s", "ms_per_step": round(ms_c, 4),
                    "scaling": "strong", "steps": ksteps, "recompute_overhead": round(sum(p["rows"] + 14 for p in per_rank) / HC - 1, 4) if world > 1 and not (LAYER_HALOS and use_lib) else 0.0,
                    **({"halo": "per-layer feature rows (f 2, l1 / l2 / l3 one each way per neighbour), nothing recomputed"} if world > 1 and LAYER_HALOS and use_lib else {}),
                    **n1_reference(ms_c, world, args.precision, args.io),
                    "roofline_frac_per_rank": [p["roofline_frac"] for p in per_rank], "per_rank": per_rank}
            del band
            torch.cuda.empty_cache()
        except Exception as ex:  # noqa: BLE001
            if rank == 0:
                result["config_C"] = {"error": str(ex)[:300]}

        try:
            # config D: 64 x 512x512, image i -> rank i mod N (shard.round_robin), weights replicated, no communication
            ND, HD, WD = 64, 512, 512
            idx = round_robin(ND, rank, world)
            batch = synth_u8(4, HD, WD, n=ND)[idx]  # seed 4 = SURVEY.md 8(d) config D
            xin = torch.from_numpy(batch).to(dev) if u8 else torch.from_numpy(r.img_to_data(batch)).to(dev)
            outd = torch.empty((len(idx), 3 * HD, 3 * WD, 4 if u8 else 3), dtype=dt_in, device=dev)
            fn = eng.upscale_rgba8_dev if u8 else eng.upscale_f32_dev
            step_d = (lambda: fn(xin, out=outd)) if idx else (lambda: None)
            for _ in range(2):
                step_d()
            torch.cuda.synchronize()
            ms_d = timed(step_d, ksteps, 1)
            entry = {"workload": f"64 x 512x512 RGB x3, image i on rank i mod {world}, {args.io}",
                     "value": round(ND * 9 * HD * WD / 1e6 / (ms_d / 1e3), 2), "unit": "output MP/s", "ms_per_step": round(ms_d, 4),
                     "scaling": "strong", "steps": ksteps, "images_per_rank": len(idx), "data_path_collectives": 0,
                     "tflops": round(ND * HD * WD * FLOP_PER_PX / (ms_d / 1e3) / 1e12, 2)}
            del outd
            # ... and from / to host memory: upload, kernels and download overlapped in chunks of 4 images (sr_upscale_rgba8)
            if u8 and idx:
                from rusty_sr_amd.engine import host_alloc
                pin_in, pin_out = host_alloc(batch.shape), host_alloc((len(idx), 3 * HD, 3 * WD, 4))
                pin_in.array[...] = batch
                step_h = lambda: eng.upscale_rgba8(pin_in.array, out=pin_out.array)
                step_h()
                ms_h = timed(step_h, max(2, ksteps // 2), 1)
                t = eng.last_timing()
                entry["host_pipelined"] = {"value": round(ND * 9 * HD * WD / 1e6 / (ms_h / 1e3), 2), "unit": "output MP/s",
                                           "ms_per_step": round(ms_h, 4), "kernel_ms": round(t["total_ms"], 4),
                                           "h2d_ms": round(t["h2d_ms"], 4), "d2h_ms": round(t["d2h_ms"], 4),
                                           "note": "page-locked host buffers, PCIe-inclusive: never `value`"}
                pin_in.close(); pin_out.close()
            elif u8 and world > 1:
                timed(lambda: None, max(2, ksteps // 2), 1)  # keep the collective sequence of the timing helper in step
            if rank == 0:
                result["config_D"] = entry
            del xin
            torch.cuda.empty_cache()
        except Exception as ex:  # noqa: BLE001
            if rank == 0:
                result["config_D"] = {"error": str(ex)[:300]}

        if world == 1:
            try:
                # config A: 256x256, one GPU (4-row tiles of the pipe form: one round of workgroups)
                pa = torch.from_numpy(synth_u8(1, 256, 256)).to(dev)[None]  # seed 1 = config A
                xa = pa if u8 else torch.from_numpy(r.img_to_data(pa.cpu().numpy())).to(dev)
                fn = eng.upscale_rgba8_dev if u8 else eng.upscale_f32_dev
                oa = fn(xa)
                for _ in range(300):  # ~50 ms: a call this short must not be timed on clocks that are still coming up
                    fn(xa, out=oa)
                torch.cuda.synchronize()
                ms_a = 1e9
                for _ in range(3):  # best of three bursts of 400 calls, each burst fenced
                    t0 = time.perf_counter()
                    for _ in range(400):
                        fn(xa, out=oa)
                    torch.cuda.synchronize()
                    ms_a = min(ms_a, (time.perf_counter() - t0) / 400 * 1e3)
                eng.set_profiling(True)
                sa = []
                for _ in range(5):
                    fn(xa, out=oa); torch.cuda.synchronize(); sa.append(eng.last_timing()["stage_ms"])
                eng.set_profiling(False)
                sa = np.median(np.array(sa), axis=0)
                ka, ach_a, _ = roofline_of(sa, [256] * 5, 256, args.precision)
                result["config_A"] = {"workload": f"256x256 RGB x3, {args.io}, one GPU", "value": round(9 * 65536 / 1e6 / (ms_a / 1e3), 2),
                                      "unit": "output MP/s", "ms_per_step": round(ms_a, 4), "stage_ms": [round(float(v), 4) for v in sa],
                                      "roofline_frac": round(ach_a / peak_here, 4), "whole_call_frac": round(65536 * FLOP_PER_PX / (ms_a / 1e3) / 1e12 / peak_here, 4)}
            except Exception as ex:  # noqa: BLE001
                result["config_A"] = {"error": str(ex)[:300]}

            try:
                # Mid-size lone frames (0.6-3 rounds of tiles): whether one runs undivided or as two bands on two streams is measured by the
                # library on the caller's own calls (sr_ctx::ForkTune).  Per shape: a context with the tuner off (the rule alone) against
                # one that has met the shape under a queue, interleaved bursts, same bytes.
                if u8 and os.environ.get("SRHIP_BENCH_MID", "1") != "0":
                    e_rule = r.Engine(params, device=local, precision=args.precision)
                    e_rule.set_experiment("forktune", "0")
                    mids = []
                    for (mh, mw) in ((320, 320), (360, 640), (480, 854)):
                        xm = torch.from_numpy(synth_u8(3, mh, mw)).to(dev)[None]
                        o_rule, o_tuned = e_rule.upscale_rgba8_dev(xm), eng.upscale_rgba8_dev(xm)
                        for _ in range(12):  # the tuner reads about one sample per fenced burst
                            for _ in range(16):
                                eng.upscale_rgba8_dev(xm, out=o_tuned)
                                e_rule.upscale_rgba8_dev(xm, out=o_rule)
                            torch.cuda.synchronize()
                        t_m = {"rule": [], "tuned": []}
                        for _ in range(3):
                            for name, e, o in (("rule", e_rule, o_rule), ("tuned", eng, o_tuned)):
                                torch.cuda.synchronize()
                                t0 = time.perf_counter()
                                for _ in range(100):
                                    e.upscale_rgba8_dev(xm, out=o)
                                torch.cuda.synchronize()
                                t_m[name].append((time.perf_counter() - t0) / 100 * 1e3)
                        state = [l.split() for l in eng.get_experiment("forktune").splitlines() if l.startswith(f"{mh}x{mw}+")]
                        ms_r, ms_t = float(np.median(t_m["rule"])), float(np.median(t_m["tuned"]))
                        mids.append({"image": [mh, mw], "rule_ms": round(ms_r, 4), "tuned_ms": round(ms_t, 4), "plan": state[0][3] if state else "rule",
                                     "whole_call_frac": round(mh * mw * FLOP_PER_PX / (ms_t / 1e3) / 1e12 / peak_here, 4),
                                     "same_bytes": bool(torch.equal(o_rule, o_tuned))})
                        del xm, o_rule, o_tuned
                    e_rule.close()
                    # ... and the host-pointer call of such a frame (what a drop-in caller's 640x480 picture takes): two bands in order, the first
                    # band's download under the second band's kernels, against one chunk (upload, kernels, download in sequence)
                    from rusty_sr_amd.engine import host_alloc
                    hin, hout = host_alloc((480, 640, 3)), host_alloc((1440, 1920, 4))
                    hin.array[...] = synth_u8(3, 480, 640)
                    host_ms = {}
                    for rnd in range(3):
                        for name, on in (("one_chunk", False), ("two_bands", True)):
                            eng.set_pipeline(on)
                            for _ in range(3):
                                eng.upscale_rgba8(hin.array, out=hout.array)
                            per = []
                            for _ in range(20):
                                t0 = time.perf_counter()
                                eng.upscale_rgba8(hin.array, out=hout.array)
                                per.append((time.perf_counter() - t0) * 1e3)
                            host_ms.setdefault(name, []).append(float(np.median(per)))
                    eng.set_pipeline(True)
                    hin.close(); hout.close()
                    host_entry = {"image": [480, 640], "one_chunk_ms": round(float(np.median(host_ms["one_chunk"])), 4),
                                  "ms": round(float(np.median(host_ms["two_bands"])), 4), "note": "sr_upscale_rgba8, page-locked buffers, PCIe-inclusive"}
                    result["mid_size_frames"] = {"entries": mids, "host_call": host_entry, "note": "lone frames of 0.6-3 rounds of tiles: the fork decision measured on the caller's "
                                                 "own calls (DESIGN.md 4e, profiles/r6_fork_tune.txt) against the fixed rule; BASELINE's shapes are outside the tuner's range"}
            except Exception as ex:  # noqa: BLE001
                result["mid_size_frames"] = {"error": str(ex)[:300]}

            try:
                # the two parameter-free graphs (`-p bilinear`, `-d`), HBM-bound elementwise kernels
                result["aux_graphs"] = {"entries": aux_entries(r, torch, ((H, W), (3 * H, 3 * W)), 100, local),
                                        "note": "bilinear_net x3 / downsample_net /3 (network.rs:111-138), device-resident; GBps counts compulsory I/O only "
                                                "(u8: 3 B in + 4 B RGBA out per pixel; f32: 12 + 12); the (3H, 3W) size is what `-d` of an upscaled frame reads"}
            except Exception as ex:  # noqa: BLE001
                result["aux_graphs"] = {"error": str(ex)[:300]}

            try:
                # the same 1920x1080 frame as a batch of 4 in ONE call (n = 4): the per-launch costs of the five kernels -- fill,
                # drain, launch boundary -- are paid once per four frames; the difference to `value` is what they cost a lone frame
                nb4 = 4
                xb = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(px, (nb4,) + px.shape))).to(dev)
                if not u8:
                    xb = torch.from_numpy(r.img_to_data(xb.cpu().numpy())).to(dev)
                fn = eng.upscale_rgba8_dev if u8 else eng.upscale_f32_dev
                ob4 = fn(xb)
                for _ in range(2):
                    fn(xb, out=ob4)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(ksteps):
                    fn(xb, out=ob4)
                torch.cuda.synchronize()
                ms4f = (time.perf_counter() - t0) / ksteps / nb4 * 1e3
                result["batch_of_4"] = {"workload": f"4 x {W}x{H} in one call (n = 4), {args.io}, resident in HBM", "ms_per_frame": round(ms4f, 4),
                                        "value": round(9 * H * W / 1e6 / (ms4f / 1e3), 2), "unit": "output MP/s",
                                        "whole_call_frac": round(H * W * FLOP_PER_PX / (ms4f / 1e3) / 1e12 / peak_here, 4)}
                del xb, ob4
                torch.cuda.empty_cache()
            except Exception as ex:  # noqa: BLE001
                result["batch_of_4"] = {"error": str(ex)[:300]}

            try:
                # ... and two lone frames in flight: a second context on a second stream, frames dealt alternately.  Each stage
                # launch of one frame drains while the other frame's launch fills, so the per-launch cost is hidden without
                # batching (a video pipeline's mode of operation; latency per frame doubles, `value` stays the one-frame number)
                eng2 = r.Engine(params, device=local, precision=args.precision)
                s_a, s_b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
                fa = eng.upscale_rgba8_dev if u8 else eng.upscale_f32_dev
                fb = eng2.upscale_rgba8_dev if u8 else eng2.upscale_f32_dev
                x1 = torch.from_numpy(px).to(dev)[None] if u8 else torch.from_numpy(r.img_to_data(px)).to(dev)[None]
                oa2, ob2 = fa(x1), fb(x1)
                torch.cuda.synchronize()
                def two_frames():
                    fa(x1, out=oa2, stream=s_a)
                    fb(x1, out=ob2, stream=s_b)
                for _ in range(3):
                    two_frames()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(ksteps):
                    two_frames()
                torch.cuda.synchronize()
                ms2 = (time.perf_counter() - t0) / (2 * ksteps) * 1e3
                result["two_frames_in_flight"] = {"workload": f"{W}x{H}, two contexts on two streams, frames dealt alternately, {args.io}, resident in HBM",
                                                  "ms_per_frame": round(ms2, 4), "value": round(9 * H * W / 1e6 / (ms2 / 1e3), 2), "unit": "output MP/s",
                                                  "whole_call_frac": round(H * W * FLOP_PER_PX / (ms2 / 1e3) / 1e12 / peak_here, 4),
                                                  "identical_outputs": bool(torch.equal(oa2, ob2))}
                eng2.close()
                del oa2, ob2, x1
                torch.cuda.empty_cache()
            except Exception as ex:  # noqa: BLE001
                result["two_frames_in_flight"] = {"error": str(ex)[:300]}

            try:
                # what the drop-in user runs: host pointers in and out (sr_upscale_rgba8), 1080p, page-locked buffers
                if u8:
                    from rusty_sr_amd.engine import host_alloc
                    pin_in, pin_out = host_alloc((H, W, 3)), host_alloc((3 * H, 3 * W, 4))
                    pin_in.array[...] = px
                    call = lambda: eng.upscale_rgba8(pin_in.array, out=pin_out.array)
                    for _ in range(3):
                        call()
                    per_call = []  # every call is synchronous: timed one by one, the median reported (a stray slow call -- the
                    for _ in range(max(ksteps, 12)):  # host thread descheduled mid-pipeline -- would otherwise own the mean)
                        t0 = time.perf_counter()
                        call()
                        per_call.append((time.perf_counter() - t0) * 1e3)
                    ms_h = float(np.median(per_call))
                    t = eng.last_timing()
                    result["host_call"] = {"workload": f"{W}x{H} u8 RGB in host memory -> RGBA8 in host memory (sr_upscale_rgba8), page-locked",
                                           "t_e2e_device_ms": round(ms_h, 4), "t_e2e_device_ms_min": round(min(per_call), 4), "value": round(9 * H * W / 1e6 / (ms_h / 1e3), 2), "unit": "output MP/s",
                                           "t_kernel_ms": round(t["total_ms"], 4), "h2d_ms": round(t["h2d_ms"], 4), "d2h_ms": round(t["d2h_ms"], 4),
                                           "note": "PCIe-inclusive (6.2 MB up, 74.6 MB down): never `value`"}
                    # diagnostic (SRHIP_BENCH_HOST_PLANS="rows;rows;..."): the same call under explicit band plans IN THIS PROCESS -- a host call's
                    # wall time depends on the process's history in ways a fresh process does not show (profiles/r6_host_mid_plans.txt 7.)
                    if os.environ.get("SRHIP_BENCH_HOST_PLANS"):
                        alt = {}
                        for plan in os.environ["SRHIP_BENCH_HOST_PLANS"].split(";") + [""]:
                            eng.set_experiment("rows", plan)
                            for _ in range(3):
                                call()
                            per = []
                            for _ in range(12):
                                t0 = time.perf_counter()
                                call()
                                per.append((time.perf_counter() - t0) * 1e3)
                            alt[plan or "auto"] = round(float(np.median(per)), 4)
                        eng.set_experiment("rows", "")
                        result["host_call"]["plans"] = alt
                    if os.environ.get("SRHIP_BENCH_HOST_F32OUT"):  # diagnostic: the f32-output call (sr_upscale_f32), first use in this process, then explicit plans
                        import ctypes as C
                        fp = C.POINTER(C.c_float)
                        fin, fout = host_alloc((H, W, 3), np.float32), host_alloc((3 * H, 3 * W, 3), np.float32)
                        fin.array[...] = r.img_to_data(px)
                        fcall = lambda: r._lib.check(eng._L.sr_upscale_f32(eng._ctx, fin.array.ctypes.data_as(fp), 1, H, W, fout.array.ctypes.data_as(fp)), eng._ctx)
                        alt = {}
                        for plan in ["first use"] + os.environ["SRHIP_BENCH_HOST_F32OUT"].split(";") + [""]:
                            eng.set_experiment("rows", "" if plan == "first use" else plan)
                            for _ in range(3):
                                fcall()
                            per = []
                            for _ in range(12):
                                t0 = time.perf_counter()
                                fcall()
                                per.append((time.perf_counter() - t0) * 1e3)
                            alt[plan or "auto"] = round(float(np.median(per)), 4)
                        eng.set_experiment("rows", "")
                        result["host_call"]["f32_output_plans"] = alt
                        fin.close(); fout.close()
                    pin_in.close(); pin_out.close()
            except Exception as ex:  # noqa: BLE001
                result["host_call"] = {"error": str(ex)[:300]}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(params, px)
        result["speedup_vs_cpu_baseline"] = round(value / result["cpu_baseline"]["value"], 1)

    if world > 1:
        dist.barrier()
    if watchdog is not None:
        watchdog.cancel()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
