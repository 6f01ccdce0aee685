"""GPU tests of the multi-GPU / sharded side of the C ABI and of BASELINE.json's named configurations at their full
sizes: config B (1920x1080) whole frame against the oracle, config C (3840x2160 as 8 row bands, 270 rows + 7-row
halos), config D (64 x 512x512 dealt round-robin).  The test box has ONE GPU: several contexts then share it (same
code path, same host threads, no RCCL -- RCCL admits one rank per device); everything that needs distinct devices
runs whenever torch.cuda.device_count() > 1 (the driver's multi-GPU node)."""
import ctypes as C

import numpy as np
import pytest

import oracle
from conftest import synth_u8

pytestmark = pytest.mark.gpu

TOL = 1e-4
TIGHT = 2e-5


def _ndev():
    import torch
    return torch.cuda.device_count()


@pytest.fixture(scope="module", params=["f32", "split_f16"])
def eng(params, request):
    import rusty_sr_amd as r
    e = r.Engine(params["imagenet"], device=0, precision=request.param)
    yield e
    e.close()


# ---------------------------------------------------------------- config B: the whole 1080p frame, not crops
def test_config_b_whole_frame_against_the_oracle(eng, params):
    """1920x1080 -> 5760x3240, EVERY output sample against the CPU oracle (f32, pre-quantisation) and the fused u8
    path against the oracle's quantiser.  The oracle needs a few seconds for this frame on the GPU box's host."""
    px = synth_u8(2, 1, 1080, 1920)
    x = oracle.img_to_data(px)
    want = oracle.forward(params["imagenet"], x)
    got = eng.upscale_f32(x)
    assert got.shape == want.shape == (1, 3240, 5760, 3)
    err = np.abs(got - want)
    assert err.max() < TIGHT, err.max()
    got8 = eng.upscale_rgba8(px)
    want8 = oracle.data_to_rgba8(want)
    d = got8[..., :3].astype(np.int16) - want8[..., :3].astype(np.int16)
    assert (got8[..., 3] == 255).all() and np.abs(d).max() <= 1
    bad = d != 0
    assert bad.mean() < 1e-4
    frac = 255.0 * want.astype(np.float64) + 0.5
    if bad.any():
        assert np.abs(frac - np.round(frac))[bad].max() < 255 * TOL  # only at rounding knife-edges


def test_split_mode_is_as_close_to_exact_arithmetic_as_the_f32_cpu_path(params):
    """|split_f16 - f64 truth| <= 2 |oracle f32 - f64 truth| on a set of 1080p crops: the fast mode may be quoted
    beside the exact mode without a precision asterisk.  (Crops carry the 7-px halo; the comparison is on the
    interior, where crop == frame.)"""
    import rusty_sr_amd as r
    px = synth_u8(2, 1, 1080, 1920)[0]
    engs = {p: r.Engine(params["imagenet"], device=0, precision=p) for p in ("f32", "split_f16")}
    try:
        full = {p: e.upscale_f32(oracle.img_to_data(px)) for p, e in engs.items()}
        worst = {"f32": 0.0, "split_f16": 0.0, "cpu": 0.0}
        for (y, x) in ((0, 0), (100, 700), (500, 1200), (1080 - 142, 1920 - 142), (640, 0), (300, 1500)):
            crop = oracle.img_to_data(px[y:y + 142, x:x + 142])
            truth = oracle.forward(params["imagenet"], crop, f64=True)[0][21:-21, 21:-21].astype(np.float64)
            cpu = oracle.forward(params["imagenet"], crop)[0][21:-21, 21:-21]
            worst["cpu"] = max(worst["cpu"], float(np.abs(cpu - truth).max()))
            for p in engs:
                got = full[p][3 * (y + 7):3 * (y + 135), 3 * (x + 7):3 * (x + 135)]
                worst[p] = max(worst[p], float(np.abs(got - truth).max()))
        assert worst["split_f16"] <= 2 * worst["cpu"] + 1e-7, worst
        assert worst["f32"] <= 2 * worst["cpu"] + 1e-7, worst
        assert worst["split_f16"] < TIGHT and worst["f32"] < TIGHT
    finally:
        for e in engs.values():
            e.close()


# ---------------------------------------------------------------- config C: 3840x2160 as 8 row bands
def test_config_c_every_band_of_the_8_way_split(eng):
    """BASELINE configs[3]: each of the eight 270-row bands of the 3840x2160 image, extended by the 7 halo rows its
    neighbours would send, reproduces the undivided rows bit for bit (band 0 / 7 touch the true image edges)."""
    import torch
    from rusty_sr_amd.shard import split_rows
    H, W = 2160, 3840
    px = torch.from_numpy(synth_u8(3, 1, H, W)[0]).cuda()
    full = eng.upscale_rgba8_dev(px[None])[0]
    torch.cuda.synchronize()
    for a, b in split_rows(H, 8):
        assert b - a == 270
        top, bot = (0 if a == 0 else 7), (0 if b == H else 7)
        out = eng.upscale_band_rgba8_dev(px[a - top:b + bot].contiguous(), top, bot)
        torch.cuda.synchronize()
        assert torch.equal(out, full[3 * a:3 * b]), (a, b)


def test_config_c_through_eight_contexts(params, eng):
    """3840x2160 through sr_upscale_rgba8_multi with 8 contexts (the one-process form of config C; on this box all on
    one GPU, on a multi-GPU node one per device): equal to the single-context call."""
    import rusty_sr_amd as r
    import torch
    nd = _ndev()
    engs = [r.Engine(params["imagenet"], device=k % nd, precision=eng.precision) for k in range(8)]
    try:
        px = synth_u8(3, 1, 2160, 3840)[0]
        want = eng.upscale_rgba8(px)
        got = r.upscale_multi(engs, px)
        np.testing.assert_array_equal(got, want)
    finally:
        for e in engs:
            e.close()


# ---------------------------------------------------------------- config D: 64 x 512x512
def test_config_d_batch_of_64(eng):
    """BASELINE configs[4] at its full size on one GPU: the 64-image batch through sr_upscale_rgba8 (chunked host
    pipeline) equals 64 single-image calls."""
    px = synth_u8(4, 64, 512, 512)
    got = eng.upscale_rgba8(px)
    assert got.shape == (64, 1536, 1536, 4)
    for i in range(64):
        np.testing.assert_array_equal(got[i], eng.upscale_rgba8(px[i]), err_msg=f"image {i}")


@pytest.mark.parametrize("k", [2, 3, 8])
def test_config_d_dealt_round_robin_over_contexts(params, eng, k):
    """sr_upscale_*_batch_multi: image i -> context i mod k, one host thread per context; equal to one context's batch,
    image for image (u8 and f32, ragged: 13 images over k contexts)."""
    import rusty_sr_amd as r
    nd = _ndev()
    engs = [r.Engine(params["imagenet"], device=j % nd, precision=eng.precision) for j in range(k)]
    try:
        px = synth_u8(44, 13, 96, 160)
        np.testing.assert_array_equal(r.upscale_batch_multi(engs, px), eng.upscale_rgba8(px))
        x = oracle.img_to_data(px[:5])
        np.testing.assert_array_equal(r.upscale_batch_multi(engs, x), eng.upscale_f32(x))
        if k == 8:  # the named size: 64 x 512x512 over 8 contexts
            px = synth_u8(4, 64, 512, 512)
            np.testing.assert_array_equal(r.upscale_batch_multi(engs, px), eng.upscale_rgba8(px))
    finally:
        for e in engs:
            e.close()


def test_round_robin_driver_on_the_gpu(eng, params):
    """shard.upscale_batch_round_robin with the GPU engine as the per-rank compute: the shares of ranks 0..2 of a
    7-image batch, stitched together, are the single-rank batch."""
    from rusty_sr_amd.shard import upscale_batch_round_robin
    px = synth_u8(45, 7, 64, 100)
    want = eng.upscale_rgba8(px)
    got = np.empty_like(want)
    for rank in range(3):
        idx, outs = upscale_batch_round_robin(px, rank, 3, eng.upscale_rgba8)
        got[idx] = outs
    np.testing.assert_array_equal(got, want)


# ---------------------------------------------------------------- the context-set rules of the multi calls
def test_cooperating_contexts_must_match(params):
    import rusty_sr_amd as r
    a = r.Engine(params["imagenet"], device=0)
    b = r.Engine(params["imagenet"], device=0, precision="split_f16")
    c = r.Engine(params["anime"], device=0)
    d = r.Engine(params["imagenet"], device=0)
    px = synth_u8(1, 2, 40, 40)
    try:
        for bad in ([a, a], [a, b], [a, c]):  # the same context twice / another arithmetic mode / other parameters
            with pytest.raises(r.SrError):
                r.upscale_multi(bad, px[0])
            with pytest.raises(r.SrError):
                r.upscale_batch_multi(bad, px)
        np.testing.assert_array_equal(r.upscale_multi([a, d], px[0]), a.upscale_rgba8(px[0]))
    finally:
        for e in (a, b, c, d):
            e.close()


# ---------------------------------------------------------------- the RCCL communicator inside libsrhip
def test_comm_single_rank_and_argument_rules(eng, params):
    """sr_comm_*: a 1-rank communicator needs no RCCL object and makes sr_upscale_sharded_*_dev the plain call;
    the id is 128 bytes from ncclGetUniqueId; two contexts of one device cannot form a communicator (RCCL admits
    one rank per device)."""
    import torch
    import rusty_sr_amd as r
    from rusty_sr_amd import _lib
    L = _lib.lib()
    assert L.sr_comm_available() == 1
    uid = r.Engine.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    assert eng.comm_rank() == (0, 1)
    eng.comm_init_rank(b"", 0, 1)
    px = torch.from_numpy(synth_u8(9, 1, 50, 70)[0]).cuda()
    out = eng.upscale_sharded_dev(px)
    want = eng.upscale_rgba8_dev(px[None])[0]
    x = torch.from_numpy(oracle.img_to_data(px.cpu().numpy())).cuda()
    out32 = eng.upscale_sharded_dev(x)
    want32 = eng.upscale_f32_dev(x[None])[0]
    torch.cuda.synchronize()
    assert torch.equal(out, want) and torch.equal(out32, want32)
    with pytest.raises(r.SrError):
        eng.comm_init_rank(uid, 2, 2)   # rank out of range
    other = r.Engine(params["imagenet"], device=0, precision=eng.precision)
    try:
        with pytest.raises(r.SrError):
            r.comm_init_all([eng, other])  # same device twice
        r.comm_init_all([other])
        outs = r.upscale_sharded_all([other], [px])
        assert torch.equal(outs[0], want)
    finally:
        other.close()


@pytest.mark.parametrize("n,prec", [(2, "f32"), (3, "split_f16"), (8, "f32")])
def test_sharded_image_through_the_library_on_one_device(params, n, prec):
    """sr_comm_init_local + sr_upscale_sharded_*_all: the C code of the sharded path itself -- band geometry, halo
    offsets in the extended buffer, the band passes, the drain -- with n contexts of ONE device, halos by peer copy.
    Uneven bands (the last one short), u8 and f32, twice through the same contexts: bit-identical to the whole image.
    (The RCCL transport differs from this only in who moves the 7 rows.)"""
    import torch
    import rusty_sr_amd as r
    from rusty_sr_amd.shard import split_rows
    H, W = 37 * n + 5, 530
    px = synth_u8(21, 1, H, W)[0]
    engs = [r.Engine(params["imagenet"], device=0, precision=prec) for _ in range(n)]
    try:
        want = engs[0].upscale_rgba8(px)
        x = oracle.img_to_data(px)
        want32 = engs[0].upscale_f32(x)
        r.comm_init_all(engs, transport="local")
        assert [e.comm_rank() for e in engs] == [(k, n) for k in range(n)]
        cuts = split_rows(H, n)
        bands = [torch.from_numpy(px[a:b]).cuda() for a, b in cuts]
        for _ in range(2):
            outs = r.upscale_sharded_all(engs, bands)
            np.testing.assert_array_equal(np.concatenate([o.cpu().numpy() for o in outs]), want)
        bands32 = [torch.from_numpy(x[a:b]).cuda() for a, b in cuts]
        outs = r.upscale_sharded_all(engs, bands32)
        np.testing.assert_array_equal(np.concatenate([o.cpu().numpy() for o in outs]), want32)
        # very uneven bands: the smallest a neighbour can still read a halo from
        hs = [7] * (n - 1) + [H - 7 * (n - 1)]
        edges = np.cumsum([0] + hs)
        outs = r.upscale_sharded_all(engs, [torch.from_numpy(px[a:b]).cuda() for a, b in zip(edges[:-1], edges[1:])])
        np.testing.assert_array_equal(np.concatenate([o.cpu().numpy() for o in outs]), want)
        # SURVEY 8(e)(ii), per-layer feature halos: every stage computes the band's own rows, the neighbours' edge rows of its output
        # (f 2, l1 / l2 / l3 one each) are exchanged before the next stage -- nothing recomputed, the same bits
        for e in engs:
            e.set_experiment("halo", "layers")
        for bs, ref in ((bands, want), (bands32, want32), ([torch.from_numpy(px[a:b]).cuda() for a, b in zip(edges[:-1], edges[1:])], want)):
            for _ in range(2):
                outs = r.upscale_sharded_all(engs, bs)
                np.testing.assert_array_equal(np.concatenate([o.cpu().numpy() for o in outs]), ref)
        engs[-1].set_experiment("halo", "input")
        with pytest.raises(r.SrError):   # every context of a call must exchange the same way
            r.upscale_sharded_all(engs, bands)
        for e in engs:
            e.set_experiment("halo", "input")
        outs = r.upscale_sharded_all(engs, bands)  # ... and back: the recompute form on maps that held neighbours' rows
        np.testing.assert_array_equal(np.concatenate([o.cpu().numpy() for o in outs]), want)
        with pytest.raises(r.SrError):   # a band shorter than the halo its neighbour needs
            r.upscale_sharded_all(engs, [torch.from_numpy(px[:6]).cuda()] + bands[1:])
        with pytest.raises(r.SrError):   # a lone rank of a local communicator cannot see its neighbours
            engs[0].upscale_sharded_dev(bands[0])
        # every sharded call is timed by two event pairs on the band's stream, profiling or not: the context's whole step
        # (sr_last_timing total_ms, no per-stage times) and the halo exchange inside it (sr_last_comm_ms)
        r.upscale_sharded_all(engs, bands)
        for e in engs:
            t, c, x = e.last_timing(), e.last_comm_ms(), e.last_comm_exposed_ms()
            assert t["total_ms"] > 0 and sum(t["stage_ms"]) == 0 and 0 < c < t["total_ms"]
            # interior first: the band's stream waits for the exchange only behind the band copy and the first layer's interior rows
            # (sr_last_comm_exposed_ms: an event pair either side of that wait)
            assert 0 <= x < t["total_ms"]
        engs[0].set_profiling(True)
        r.upscale_sharded_all(engs, bands)
        assert engs[0].last_comm_ms() > 0 and sum(engs[0].last_timing()["stage_ms"]) > 0  # ... with it on, the conv stack's own per-stage events
        engs[0].set_profiling(False)
        engs[0].upscale_rgba8(px)  # an ordinary call afterwards reports its own timing again
        assert engs[0].last_timing()["h2d_ms"] > 0
    finally:
        for e in engs:
            e.close()


def test_contexts_are_independent_across_host_threads(params):
    """include/srhip.h: a context is single-caller, distinct contexts may be driven from distinct threads.  Four threads,
    each with its own context of the one device (two per arithmetic mode), hammer the host-pointer and the device-pointer
    entry points with different shapes at once (ctypes releases the GIL inside the library); every result must equal the one
    a lone context gives."""
    import threading
    import torch
    import rusty_sr_amd as r
    shapes = [(1, 300, 515), (2, 64, 64), (1, 523, 1100), (1, 40, 70), (3, 37, 129), (1, 700, 900)]
    rng = np.random.default_rng(23)
    imgs = [rng.integers(0, 256, sh + (3,), dtype=np.uint8) for sh in shapes]
    want = {}
    for prec in ("f32", "split_f16"):
        e = r.Engine(params["imagenet"], precision=prec)
        want[prec] = [e.upscale_rgba8(px) for px in imgs]
        e.close()
    errors = []

    def worker(k):
        prec = ("f32", "split_f16")[k % 2]
        try:
            e = r.Engine(params["imagenet"], precision=prec)
            stream = torch.cuda.Stream()
            for it in range(6):
                for j in np.random.default_rng(k * 10 + it).permutation(len(imgs)):
                    if (it + j + k) % 2:
                        got = e.upscale_rgba8(imgs[j])
                    else:
                        with torch.cuda.stream(stream):
                            got = e.upscale_rgba8_dev(torch.from_numpy(imgs[j]).cuda(), stream=stream)
                            stream.synchronize()
                        got = got.cpu().numpy()
                    if not np.array_equal(got, want[prec][j]):
                        errors.append((k, prec, it, shapes[j]))
            e.close()
        except Exception as ex:  # noqa: BLE001
            errors.append((k, repr(ex)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:5]


@pytest.mark.parametrize("prec", ["f32", "split_f16"])
def test_feature_maps_beyond_four_gib(params, prec):
    """Maximum sizes: a 6100 x 6200 image has 4.9 GB per 32-channel feature map, so the undivided device-resident pass
    addresses rows more than 2^32 bytes from the start of a map (no other test does: 4K is 1.06 GB, the 64-image batch
    2.1 GB).  Bands cut from below, across and above that line -- small workspaces, small offsets -- must equal the same
    rows of the whole, and so must the host-pointer call (bands through two other workspaces)."""
    import torch
    import rusty_sr_amd as r
    H, W = 6200, 6100
    rng = np.random.default_rng(61)
    px = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    eng = r.Engine(params["imagenet"], precision=prec)
    small = r.Engine(params["imagenet"], precision=prec)
    try:
        d = torch.from_numpy(px).cuda()
        whole = eng.upscale_rgba8_dev(d[None])[0]
        pitch = (W + 31) // 32 * 32 + 4
        line = (1 << 32) // 128 // pitch            # the row whose pixels sit 4 GiB into a map
        assert 100 < line < H - 100
        for y0, y1 in ((0, 40), (line - 30, line + 30), (H - 45, H), (H // 2, H // 2 + 16)):
            a, b = max(0, y0 - 7), min(H, y1 + 7)
            band = small.upscale_band_rgba8_dev(d[a:b].contiguous(), y0 - a, b - y1)
            assert torch.equal(band, whole[3 * y0:3 * y1]), (y0, y1)
        host = eng.upscale_rgba8(px)
        assert np.array_equal(host, whole.cpu().numpy())
    finally:
        eng.close(); small.close()


def test_output_beyond_four_gib(params):
    """... and an 11 000 x 11 000 image: 15.5 GB per feature map, a 4.36 GB RGBA output (rows beyond 2^32 bytes), 121 M
    pixels through the 32-bit tile arithmetic.  Device-resident, bands against the whole as above."""
    import torch
    import rusty_sr_amd as r
    H = W = 11000
    g = torch.Generator(device="cuda").manual_seed(7)
    d = torch.randint(0, 256, (H, W, 3), dtype=torch.uint8, device="cuda", generator=g)
    eng = r.Engine(params["imagenet"])
    small = r.Engine(params["imagenet"])
    try:
        whole = eng.upscale_rgba8_dev(d[None])[0]
        assert whole.numel() > (1 << 32)
        line = (1 << 32) // (3 * W * 4) // 3          # the input row whose output rows straddle 4 GiB
        for y0, y1 in ((0, 24), (line - 20, line + 20), (H - 40, H)):
            a, b = max(0, y0 - 7), min(H, y1 + 7)
            band = small.upscale_band_rgba8_dev(d[a:b].contiguous(), y0 - a, b - y1)
            assert torch.equal(band, whole[3 * y0:3 * y1]), (y0, y1)
        assert bool((whole[..., 3] == 255).all())
    finally:
        eng.close(); small.close()


@pytest.mark.skipif("_ndev() < 2")
def test_contexts_on_other_devices(params):
    """device != 0: the > 64 KB dynamic-LDS attribute is per (kernel, device); a context on every device of the node
    must run every stage kernel (both forms, both modes) and agree with device 0."""
    import rusty_sr_amd as r
    px = synth_u8(8, 1, 300, 1100)[0]   # pipe form
    small = synth_u8(8, 1, 40, 70)[0]   # first form
    for prec in ("f32", "split_f16"):
        ref = r.Engine(params["imagenet"], device=0, precision=prec)
        want, want_s = ref.upscale_rgba8(px), ref.upscale_rgba8(small)
        for dev in range(1, _ndev()):
            e = r.Engine(params["imagenet"], device=dev, precision=prec)
            np.testing.assert_array_equal(e.upscale_rgba8(px), want, err_msg=f"device {dev}")
            np.testing.assert_array_equal(e.upscale_rgba8(small), want_s, err_msg=f"device {dev}")
            e.close()
        ref.close()


@pytest.mark.skipif("_ndev() < 2")
def test_sharded_over_all_devices_with_rccl(params):
    """The real thing (runs on a multi-GPU node only): one context per device, ncclCommInitAll inside libsrhip, one
    3840-wide image in row bands resident on their devices, grouped ncclSend / ncclRecv halo exchange over xGMI,
    band passes -- bit-identical to the single-GPU call; and the host-memory forms across real devices."""
    import torch
    import rusty_sr_amd as r
    from rusty_sr_amd.shard import split_rows
    n = _ndev()
    H, W = 135 * n, 3840
    px = synth_u8(3, 1, H, W)[0]
    for prec in ("f32", "split_f16"):
        engs = [r.Engine(params["imagenet"], device=k, precision=prec) for k in range(n)]
        try:
            want = engs[0].upscale_rgba8(px)
            bands = [torch.from_numpy(px[a:b]).to(f"cuda:{k}") for k, (a, b) in enumerate(split_rows(H, n))]
            for transport in ("rccl", "local"):   # grouped ncclSend / ncclRecv, then peer copies over the same links
                r.comm_init_all(engs, transport=transport)
                for _ in range(2):
                    outs = r.upscale_sharded_all(engs, bands)
                got = np.concatenate([o.cpu().numpy() for o in outs])
                np.testing.assert_array_equal(got, want, err_msg=transport)
            np.testing.assert_array_equal(r.upscale_multi(engs, px), want)
            batch = synth_u8(4, 2 * n + 1, 128, 160)
            np.testing.assert_array_equal(r.upscale_batch_multi(engs, batch), engs[0].upscale_rgba8(batch))
        finally:
            for e in engs:
                e.close()
