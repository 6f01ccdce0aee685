"""GPU parity: libsrhip (hand-written gfx950 kernels, through the C ABI) against the
CPU oracle on identical inputs.  Bar (BASELINE.json north_star): every output channel
within 1e-4 f32 of the CPU path before quantisation; u8 outputs may differ only at
rounding knife-edges."""
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, load_png, synth_u8

pytestmark = pytest.mark.gpu

TOL = 1e-4          # north_star tolerance, pre-quantisation f32
TIGHT = 2e-5        # what exact-f32 MFMA actually achieves (rounding-order noise only)


@pytest.fixture(scope="module", params=["f32", "split_f16"])
def engines(params, request):
    """Every parity test runs in both arithmetic modes of the engine: exact-f32 MFMA
    and split-half (3 f16 MFMAs per product); the bar is the same."""
    import rusty_sr_amd as r
    e = {k: r.Engine(v, device=0, precision=request.param) for k, v in params.items()}
    yield e
    for x in e.values():
        x.close()


def _check_u8(got, v_ref):
    want = oracle.data_to_rgba8(v_ref)
    assert got.shape == want.shape
    assert (got[..., 3] == 255).all()
    d = got[..., :3].astype(int) - want[..., :3].astype(int)
    assert np.abs(d).max() <= 1
    if (d != 0).any():
        frac = 255.0 * v_ref.astype(np.float64) + 0.5
        edge = np.abs(frac - np.round(frac))
        assert edge[d != 0].max() < 255 * TOL, "u8 mismatch away from a rounding knife-edge"
        assert (d != 0).mean() < 1e-3


@pytest.mark.parametrize("h,w", [(1, 1), (2, 3), (5, 31), (8, 32), (9, 33), (40, 70), (37, 129)])
def test_f32_small_shapes(engines, params, h, w):
    """Empty-ish / ragged tiles: sizes below, at and just above the 8x32 and 4x32 tile."""
    x = oracle.img_to_data(synth_u8(10 + h, 1, h, w))
    want = oracle.forward(params["imagenet"], x)
    got = engines["imagenet"].upscale_f32(x)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < TIGHT


@pytest.mark.parametrize("name", ["imagenet", "imagenetlinear", "anime"])
def test_per_stage_features(engines, params, name):
    """Every node of the graph (f, l1, l2, l3 after BeLU) matches, not just the output."""
    h, w = 45, 77
    x = oracle.img_to_data(synth_u8(3, 1, h, w))
    want, taps = oracle.forward_taps(params[name], x)
    got = engines[name].upscale_f32(x)
    for k, key in enumerate(("f", "l1", "l2", "l3")):
        feat = engines[name].read_feature(k, h, w)
        err = np.abs(feat - taps[key]).max()
        assert err < TIGHT * max(1.0, np.abs(taps[key]).max()), (key, err)
    assert np.abs(got - want).max() < TIGHT


@pytest.mark.parametrize("case", ["crop", "border", "one", "twothree"])
def test_against_second_restatement_vectors(engines, case):
    """The committed torch-float64 node vectors (tests/golden/make_vectors.py), without the C oracle in
    the loop: every node of the graph on the GPU against an independent restatement."""
    v = np.load(os.path.join(GOLDEN, "vectors_torch_f64.npz"))
    eng = engines[str(v[f"{case}.weights"])]
    px = v[f"{case}.px"]
    h, w = px.shape[:2]
    got = eng.upscale_f32(oracle.img_to_data(px))
    assert np.abs(got - v[f"{case}.out"]).max() < TIGHT
    for k, key in enumerate(("f", "l1", "l2", "l3")):
        want = v[f"{case}.{key}"]
        feat = eng.read_feature(k, h, w)
        feat = feat[..., ::4] if want.shape[-1] == 8 else feat
        assert np.abs(feat - want).max() < TIGHT * max(1.0, np.abs(want).max()), (case, key)
    _check_u8(eng.upscale_rgba8(px), v[f"{case}.out"])


def test_white_noise_stress(engines, params):
    """White noise drives pre-activations to +-70; still inside 1e-4."""
    rng = np.random.default_rng(5)
    x = rng.random((1, 64, 96, 3), dtype=np.float32)
    want = oracle.forward(params["imagenet"], x)
    got = engines["imagenet"].upscale_f32(x)
    assert np.abs(got - want).max() < TOL
    # and against exact arithmetic the GPU is as close as the CPU f32 path is
    truth = oracle.forward(params["imagenet"], x, f64=True)
    assert np.abs(got - truth).max() < 2 * max(np.abs(want - truth).max(), 1e-6) + 1e-6


def test_out_of_range_inputs(engines, params):
    """graph.forward takes any f32, not only [0,1] images."""
    rng = np.random.default_rng(6)
    x = (rng.random((1, 20, 40, 3), dtype=np.float32) * 4 - 2)
    want = oracle.forward(params["anime"], x)
    got = engines["anime"].upscale_f32(x)
    assert np.abs(got - want).max() < TOL


def test_batch_equals_single(engines, params):
    xb = oracle.img_to_data(synth_u8(7, 3, 33, 50))
    got = engines["imagenet"].upscale_f32(xb)
    want = oracle.forward(params["imagenet"], xb)
    assert np.abs(got - want).max() < TIGHT
    for i in range(3):
        np.testing.assert_array_equal(got[i], engines["imagenet"].upscale_f32(xb[i]))


def test_config_A_256x256_both_tile_heights(engines, params):
    """BASELINE configs[1]: 256x256 RGB, bundled weights, single tile height 4 path;
    512x512 exercises the 8-row tile path (>= 2 workgroups per CU)."""
    for seed, n, h, w in ((1, 1, 256, 256), (4, 1, 512, 512)):
        px = synth_u8(seed, n, h, w)
        x = oracle.img_to_data(px)
        want = oracle.forward(params["imagenet"], x)
        got = engines["imagenet"].upscale_f32(x)
        assert np.abs(got - want).max() < TIGHT
        _check_u8(engines["imagenet"].upscale_rgba8(px), want)


def test_rgba8_fused_path(engines, params):
    """img_to_data + forward + data_to_img fused on device, RGB and RGBA inputs."""
    px3 = synth_u8(11, 2, 30, 45)
    want = oracle.forward(params["imagenet"], oracle.img_to_data(px3))
    _check_u8(engines["imagenet"].upscale_rgba8(px3), want)
    px4 = np.concatenate([px3, np.full(px3.shape[:-1] + (1,), 7, np.uint8)], axis=-1)
    got4 = engines["imagenet"].upscale_rgba8(px4)
    np.testing.assert_array_equal(got4, engines["imagenet"].upscale_rgba8(px3))  # alpha dropped


def test_cartoon_golden_end_to_end(engines):
    """The reference's own result pin, through the GPU: cartoon_lr + anime.rsr -> cartoon_rsa."""
    lr, gold = load_png("cartoon_lr.png"), load_png("cartoon_rsa.png")
    out = engines["anime"].upscale_rgba8(lr)
    d = out[..., :3].astype(int) - gold[..., :3].astype(int)
    assert (out[..., 3] == 255).all()
    assert np.abs(d).max() <= 1
    assert (d == 0).mean() >= 0.9999


def test_butterfly_sanity(engines, params):
    lr = load_png("butterfly_lr.png")
    out = engines["imagenet"].upscale_rgba8(lr)
    gold = load_png("butterfly_rs.png")
    mse = np.mean((out[..., :3].astype(float) - gold[..., :3].astype(float)) ** 2)
    assert 10 * np.log10(255 ** 2 / mse) >= 55.0


def test_band_equals_untiled_bit_exact(engines, params):
    """Row bands with a 7-row halo reproduce the un-sharded result bit for bit
    (SURVEY.md 8(e)), including bands that touch the true top / bottom edge."""
    import torch
    eng = engines["imagenet"]
    h, w = 96, 70
    x = oracle.img_to_data(synth_u8(12, 1, h, w))
    full = eng.upscale_f32(x)[0]
    xt = torch.from_numpy(x[0]).cuda()
    cuts = [0, 30, 41, 96]
    for a, b in zip(cuts[:-1], cuts[1:]):
        top = 0 if a == 0 else 7
        bot = 0 if b == h else 7
        ext = xt[a - top:b + bot].contiguous()
        out = eng.upscale_band_f32_dev(ext, top, bot)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(out.cpu().numpy(), full[3 * a:3 * b])
    # wider halos are accepted, too-narrow ones are refused
    ext = xt[30 - 9:41 + 8].contiguous()
    out = eng.upscale_band_f32_dev(ext, 9, 8)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), full[90:123])
    import rusty_sr_amd as r
    with pytest.raises(r.SrError):
        eng.upscale_band_f32_dev(xt[30 - 3:41 + 7].contiguous(), 3, 7)


def test_device_entry_points_match_host(engines, params):
    import torch
    eng = engines["imagenet"]
    px = synth_u8(13, 2, 24, 40)
    x = oracle.img_to_data(px)
    o1 = eng.upscale_f32_dev(torch.from_numpy(x).cuda())
    o2 = eng.upscale_rgba8_dev(torch.from_numpy(px).cuda())
    torch.cuda.synchronize()
    np.testing.assert_array_equal(o1.cpu().numpy(), eng.upscale_f32(x))
    np.testing.assert_array_equal(o2.cpu().numpy(), eng.upscale_rgba8(px))


def test_graph_forward_mirror(params):
    """The reference-shaped call: graph.forward(1, vec![input], &params) (main.rs:168-171)."""
    import rusty_sr_amd as r
    g = r.sr_net(r.FACTOR)
    p = params["imagenet"]
    assert len(p) == g.num_params()
    px = synth_u8(14, 1, 19, 23)[0]
    inp = r.NodeData.new_blank(r.DataShape(r.CHANNELS, [23, 19], 1))
    inp.values[:] = r.img_to_data(px).reshape(-1)
    out = g.forward(1, [inp], p)[0]
    assert list(out.shape.spatial_dimensions) == [69, 57]
    want = oracle.forward(p, oracle.img_to_data(px))[0]
    assert np.abs(out.values.reshape(57, 69, 3) - want).max() < TIGHT
    with pytest.raises(r.SrError):
        g.forward(1, [inp], p[:-1])


@pytest.mark.parametrize("H,W", [(1080, 1920), (2160, 3840)])
def test_full_size_properties_1080p(engines, params, H, W):
    """BASELINE configs[2] (1920x1080) and configs[3]'s image (3840x2160) on one GPU: too large for the oracle in seconds, so use
    size-independent properties: (a) a 96-row window recomputed on its own with a
    7-row halo is bit-identical; (b) a 64x64 crop matches the oracle run on the crop
    plus halo; (c) the u8 path equals quantising the f32 path."""
    import torch
    eng = engines["imagenet"]
    px = synth_u8(2 if H == 1080 else 3, 1, H, W)
    xt = torch.from_numpy(px).cuda()
    # true IEEE division on the host (torch divides by a scalar as x * (1/255): 1 ulp off)
    x32 = torch.from_numpy(oracle.img_to_data(px)).cuda()
    full = eng.upscale_f32_dev(x32)
    band = eng.upscale_band_f32_dev(x32[0, 500 - 7:596 + 7].contiguous(), 7, 7)
    torch.cuda.synchronize()
    assert torch.equal(band, full[0, 1500:1788])
    crop = oracle.img_to_data(px[0, 300 - 7:364 + 7, 900 - 7:964 + 7])
    want = oracle.forward(params["imagenet"], crop)[0][21:-21, 21:-21]
    got = full[0, 900:1092, 2700:2892].cpu().numpy()
    assert np.abs(got - want).max() < TIGHT
    # ... and one in the bottom-right corner (true image edges on two sides: zero padding there, halo on the others)
    crop = oracle.img_to_data(px[0, H - 71:, W - 71:])
    want = oracle.forward(params["imagenet"], crop)[0][21:, 21:]
    got = full[0, 3 * (H - 64):, 3 * (W - 64):].cpu().numpy()
    assert np.abs(got - want).max() < TIGHT
    out8 = eng.upscale_rgba8_dev(xt)
    torch.cuda.synchronize()
    q = torch.clamp(torch.floor(255.0 * full + 0.5), 0, 255).to(torch.uint8)
    assert torch.equal(out8[..., :3], q) and bool((out8[..., 3] == 255).all())


@pytest.mark.parametrize("h,w", [(1080, 1920), (577, 911), (523, 1100), (2000, 270), (1400, 2100)])
def test_host_pipeline_bands_are_bit_identical(engines, h, w):
    """sr_upscale_* on host pointers splits a large image into row bands (upload / kernels /
    download overlap; equal bands, or -- f32 arithmetic with u8 output -- two equal bands and a short tail on alternating
    streams (from 0.8 M px, here 1080x1920) or three geometrically shrinking ones computed in order (from ~2.9 M px, here
    1400x2100)); the result must equal the undivided pass bit for bit, f32 and u8,
    pageable and page-locked (sr_host_alloc) destinations alike."""
    import rusty_sr_amd as r
    from rusty_sr_amd.engine import host_alloc
    eng = engines["imagenet"]
    px = synth_u8(h + w, 1, h, w)[0]
    x = oracle.img_to_data(px)
    try:
        eng.set_pipeline(False)
        want32, want8 = eng.upscale_f32(x), eng.upscale_rgba8(px)
        assert eng.read_feature(0, h, w).shape == (h, w, 32)   # allowed after an undivided pass
        eng.set_pipeline(True)
        got8, got32 = eng.upscale_rgba8(px), eng.upscale_f32(x)   # f32 output is download-bound: banded at every one of these sizes
        t = eng.last_timing()
        assert t["total_ms"] > 0 and t["h2d_ms"] > 0 and t["d2h_ms"] > 0
        with pytest.raises(r.SrError):      # feature maps hold the last band only
            eng.read_feature(0, h, w)
        pin = host_alloc((3 * h, 3 * w, 4))
        got8p = eng.upscale_rgba8(px, out=pin.array).copy()
        pin.close()
    finally:
        eng.set_pipeline(True)
    np.testing.assert_array_equal(got32, want32)
    np.testing.assert_array_equal(got8, want8)
    np.testing.assert_array_equal(got8p, want8)


@pytest.mark.parametrize("h,w", [(480, 640), (448, 448), (540, 960), (450, 446), (333, 911)])
def test_host_call_of_a_mid_size_frame_runs_as_two_bands_bit_identically(engines, h, w):
    """sr_upscale_rgba8 on host pointers, 200K (split-half mode: 180K) ... 524K px: two bands in order on one stream (70 / 30 in exact
    f32, 60 / 40 in the split-half mode), the first band's download under the second band's kernels (450x446 = 200 700 px is just above
    the threshold); below it (300x515 here, 256x256 elsewhere) one chunk as before.  Bytes equal the undivided pass's; the call leaves
    a band's maps behind, so sr_read_feature refuses."""
    import rusty_sr_amd as r
    from rusty_sr_amd.engine import host_alloc
    eng = engines["imagenet"]
    px = synth_u8(h * w, 1, h, w)[0]
    try:
        x = oracle.img_to_data(px)
        eng.set_pipeline(False)
        want8, want32 = eng.upscale_rgba8(px), eng.upscale_f32(x)
        assert eng.read_feature(0, h, w).shape == (h, w, 32)
        eng.set_pipeline(True)
        got8 = eng.upscale_rgba8(px)
        with pytest.raises(r.SrError):
            eng.read_feature(0, h, w)
        got32 = eng.upscale_f32(x)   # f32 output: three times the download -- two bands (three from 400K / 300K px) in order
        with pytest.raises(r.SrError):
            eng.read_feature(0, h, w)
        np.testing.assert_array_equal(got32, want32)
        pin_in, pin_out = host_alloc((h, w, 3)), host_alloc((3 * h, 3 * w, 4))
        pin_in.array[...] = px
        got8p = eng.upscale_rgba8(pin_in.array, out=pin_out.array).copy()
        pin_in.close(); pin_out.close()
        small = synth_u8(7, 1, 300, 515)[0]   # 154 500 px: one chunk with u8 output
        eng.upscale_rgba8(small)
        assert eng.read_feature(0, 300, 515).shape == (300, 515, 32)
        tiny = oracle.img_to_data(synth_u8(8, 1, 200, 333)[0])   # 66 600 px: one chunk with f32 output too
        eng.upscale_f32(tiny)
        assert eng.read_feature(0, 200, 333).shape == (200, 333, 32)
    finally:
        eng.set_pipeline(True)
    np.testing.assert_array_equal(got8, want8)
    np.testing.assert_array_equal(got8p, want8)


def test_reserve_allocates_and_warms_without_touching_results(params):
    """sr_reserve_*: the allocations and one pass of the kernels of a later sr_upscale_* call of that shape, on whatever
    the staging buffers hold.  Results of the real calls are what a context that never reserved gives -- also when the
    reserved shape is not the one that comes, and in band form."""
    import rusty_sr_amd as r
    px = synth_u8(5, 1, 523, 1100)[0]
    x = oracle.img_to_data(px)
    plain = r.Engine(params["imagenet"])
    want8, want32 = plain.upscale_rgba8(px), plain.upscale_f32(x)
    plain.close()
    eng = r.Engine(params["imagenet"])
    try:
        eng.reserve(1, 523, 1100, io="rgba8", channels=3)
        np.testing.assert_array_equal(eng.upscale_rgba8(px), want8)
        eng.reserve(1, 523, 1100, io="f32")          # banded plan (f32 output is download-bound)
        np.testing.assert_array_equal(eng.upscale_f32(x), want32)
        eng.reserve(4, 64, 64)                       # another shape than the one that follows
        np.testing.assert_array_equal(eng.upscale_rgba8(px), want8)
        # a job no device can hold: the allocation fails loudly (SR_E_NOMEM) and the context keeps working
        with pytest.raises(r.SrError) as oom:
            eng.reserve(1, 400000, 400000)
        assert oom.value.status == r._lib.SR_E_NOMEM
        np.testing.assert_array_equal(eng.upscale_rgba8(px), want8)
        with pytest.raises(r.SrError):
            eng.reserve(1, 0, 10)
        with pytest.raises(r.SrError):
            eng.reserve(1, 16, 16, channels=5)
    finally:
        eng.close()


@pytest.mark.parametrize("precision", ["f32", "split_f16"])
def test_a_new_geometry_only_needs_its_border_cleared(params, precision):
    """One context meets images of many sizes (a folder of pictures): the feature maps keep their allocation and a new
    geometry re-zeroes just the cells outside the image interiors (clear_borders_kernel) -- what was interior data of
    the previous, larger or differently pitched image must not leak into the zero padding.  Every result equals the
    one a fresh context gives."""
    import rusty_sr_amd as r
    seq = [(1, 300, 515), (1, 40, 70), (3, 37, 129), (1, 16, 3000), (1, 2000, 40), (2, 64, 64), (1, 301, 514), (1, 8, 32), (1, 300, 515)]
    rng = np.random.default_rng(17)
    imgs = [rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8) for (n, h, w) in seq]
    eng = r.Engine(params["imagenet"], precision=precision)
    try:
        got = [eng.upscale_rgba8(px) for px in imgs]
        import torch
        ext = torch.from_numpy(imgs[0][0][100:160].copy()).cuda()         # a band in a workspace last used for another shape
        got_band = eng.upscale_band_rgba8_dev(ext, 7, 7).cpu().numpy()
    finally:
        eng.close()
    for px, g in zip(imgs, got):
        fresh = r.Engine(params["imagenet"], precision=precision)
        try:
            np.testing.assert_array_equal(g, fresh.upscale_rgba8(px), err_msg=str(px.shape))
        finally:
            fresh.close()
    np.testing.assert_array_equal(got_band, got[0][0][3 * 107:3 * 153])


@pytest.mark.parametrize("precision", ["f32", "split_f16"])
def test_multi_context_call_is_bit_identical(params, precision):
    """sr_upscale_*_multi: one image over several contexts (one per GPU in production; the test box has one GPU, so
    the contexts share it -- same code path, same threads).  Shares are whole 8-row tiles with halo rows taken from
    the caller's image; any number of contexts must reproduce the single-context result bit for bit."""
    import rusty_sr_amd as r
    engs = [r.Engine(params["imagenet"], device=0, precision=precision) for _ in range(3)]
    try:
        for (h, w) in ((37, 50), (100, 64), (600, 900)):
            px = synth_u8(h, 1, h, w)[0]
            x = oracle.img_to_data(px)
            want32, want8 = engs[0].upscale_f32(x), engs[0].upscale_rgba8(px)
            for k in (1, 2, 3):
                np.testing.assert_array_equal(r.upscale_multi(engs[:k], x), want32, err_msg=f"f32 {h}x{w} on {k}")
                np.testing.assert_array_equal(r.upscale_multi(engs[:k], px), want8, err_msg=f"u8 {h}x{w} on {k}")
        with pytest.raises(r.SrError):
            r.upscale_multi([engs[0], r.bilinear_net()], px)
    finally:
        for e in engs:
            e.close()


def test_host_pipeline_batch_chunks(engines, params):
    """A batch goes through the host pipeline in chunks of whole images (ragged last chunk);
    every image equals its own single-image call, and config D's shape (n x 512 x 512) works."""
    eng = engines["imagenet"]
    for (n, h, w) in ((25, 300, 300), (7, 64, 64), (9, 512, 512)):
        px = synth_u8(n * h, n, h, w)
        got = eng.upscale_rgba8(px)
        eng.set_pipeline(False)
        try:
            want = eng.upscale_rgba8(px)
            one = eng.upscale_rgba8(px[n - 1])
        finally:
            eng.set_pipeline(True)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(got[n - 1], one)
    x = oracle.img_to_data(synth_u8(3, 21, 200, 280))   # 56K px per image -> chunks of 18 + 3
    got = eng.upscale_f32(x)
    assert np.abs(got[20] - oracle.forward(params["imagenet"], x[20:21])[0]).max() < TOL


def test_pipe_form_equals_first_form_bit_for_bit(engines, params):
    """The two forms of the stage kernels (conv_stage_pipe_kernel: half tiles double-buffered, persistent, 8-row tiles and
    4-row tiles in one launch; conv_stage_kernel: whole tile resident, one tile class) share step order and weight chunks,
    so they must agree bit for bit -- on ragged shapes, on batches, with few tiles per workgroup, for either tile height
    alone, for every mix of the two (the "tail" of 4-row tiles), and for any tile order (column-block width).
    sr_set_experiment is the library's A/B switch."""
    eng = engines["imagenet"]
    rng = np.random.default_rng(5)
    shapes = [(1, 8, 32), (1, 9, 33), (2, 40, 70), (1, 64, 1024), (3, 37, 129), (1, 130, 700), (1, 300, 515), (1, 2000, 40), (1, 16, 3000),
              (1, 250, 2080), (2, 333, 640)]
    try:
        eng.set_pipeline(False)  # one chunk per host call: a lone frame of 100K px or more otherwise runs as bands, and a band's maps are not the image's (the feature reads below)
        for (n, h, w) in shapes:
            px = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
            x = oracle.img_to_data(px)
            eng.set_experiment("th", "8")
            eng.set_experiment("pipe", "all")
            pipe8, pipe32 = eng.upscale_rgba8(px), eng.upscale_f32(x)
            feats = [eng.read_feature(k, h, w) for k in range(4)]
            for bw in ("0", "3", "16"):
                eng.set_experiment("bw", bw)
                np.testing.assert_array_equal(eng.upscale_f32(x), pipe32, err_msg=f"tile order bw={bw} {(n, h, w)}")
            eng.set_experiment("bw", "")
            eng.set_experiment("th", "4")  # the pipe form on 4-row tiles only
            np.testing.assert_array_equal(eng.upscale_rgba8(px), pipe8, err_msg=f"pipe form, 4-row tiles, u8 {(n, h, w)}")
            np.testing.assert_array_equal(eng.upscale_f32(x), pipe32, err_msg=f"pipe form, 4-row tiles {(n, h, w)}")
            for k in range(4):
                np.testing.assert_array_equal(feats[k], eng.read_feature(k, h, w), err_msg=f"feature {k}, 4-row tiles {(n, h, w)}")
            eng.set_experiment("th", "")   # automatic: both classes in one launch, for several tail lengths
            for tail in ("", "0.01", "0.3", "4"):
                eng.set_experiment("tail", tail)
                np.testing.assert_array_equal(eng.upscale_f32(x), pipe32, err_msg=f"mixed tiles, tail={tail!r} {(n, h, w)}")
                np.testing.assert_array_equal(eng.upscale_rgba8(px), pipe8, err_msg=f"mixed tiles, u8, tail={tail!r} {(n, h, w)}")
                for bw in ("0", "3"):
                    eng.set_experiment("bw", bw)
                    np.testing.assert_array_equal(eng.upscale_f32(x), pipe32, err_msg=f"mixed tiles, tail={tail!r}, bw={bw} {(n, h, w)}")
                eng.set_experiment("bw", "")
            eng.set_experiment("tail", "")
            eng.set_experiment("th", "8")
            eng.set_experiment("pipe", "none")
            first32 = eng.upscale_f32(x)
            np.testing.assert_array_equal(pipe32, first32, err_msg=str((n, h, w)))
            for k in range(4):
                np.testing.assert_array_equal(feats[k], eng.read_feature(k, h, w), err_msg=f"feature {k} {(n, h, w)}")
            np.testing.assert_array_equal(pipe8, eng.upscale_rgba8(px))
            eng.set_experiment("th", "4")
            np.testing.assert_array_equal(eng.upscale_f32(x), first32, err_msg="tile height 4 vs 8")
            eng.set_experiment("th", "")
            eng.set_experiment("pipe", "")  # the library's own choice of form and tile plan
            np.testing.assert_array_equal(eng.upscale_f32(x), first32, err_msg=f"automatic plan {(n, h, w)}")
            np.testing.assert_array_equal(eng.upscale_rgba8(px), pipe8, err_msg=f"automatic plan, u8 {(n, h, w)}")
            if n * h * w <= 40 * 70 * 2:
                assert np.abs(pipe32 - oracle.forward(params["imagenet"], x)).max() < TIGHT
    finally:
        eng.set_pipeline(True)
        for key in ("th", "pipe", "bw", "tail"):
            eng.set_experiment(key, "")


def test_bilinear_and_downsample_graphs(params):
    """The two parameter-free graphs of upscale() (`-p bilinear`, `-d`; network.rs:111-138)."""
    import rusty_sr_amd as r
    from rusty_sr_amd import _lib
    assert _lib.lib().sr_num_params(_lib.SR_GRAPH_BILINEAR) == 0 and _lib.lib().sr_num_params(_lib.SR_GRAPH_SR_NET) == oracle.NPARAMS
    bl, ds = r.bilinear_net(r.FACTOR), r.downsample_net(r.FACTOR)
    for h, w in ((1, 1), (5, 7), (43, 43), (130, 257)):
        px = synth_u8(30 + h, 2, h, w)
        x = oracle.img_to_data(px)
        want = oracle.bilinear(x)
        got = bl.upscale_f32(x)
        # (the f32 entry points compute pow(x, p) as exp2(p * log2 x) on the hardware's v_log_f32 / v_exp_f32: measured <= 1e-6 here)
        assert got.shape == want.shape and np.abs(got - want).max() < 1e-5
        _check_u8(bl.upscale_rgba8(px), want)
        if h >= 3 and w >= 3:
            wd = oracle.downsample(x)
            gd = ds.upscale_f32(x)
            assert gd.shape == wd.shape == (2, h // 3, w // 3, 3) and np.abs(gd - wd).max() < 1e-5
            _check_u8(ds.upscale_rgba8(px), wd)
    with pytest.raises(r.SrError):
        r.Engine(params["imagenet"], graph="bilinear")  # main.rs:162: 130459 != 0
    with pytest.raises(r.SrError):
        ds.upscale_f32(np.zeros((2, 2, 3), np.float32))


def test_u8_bilinear_graph_on_images_wider_than_32767_pixels():
    """bilinear_u8_kernel divides output columns by 3 with a multiply-high; round 5's 16-bit form was exact below 98 304 output
    columns only (advisor): 33 001- and 40 000-pixel rows, 3- and 4-channel, against the oracle."""
    import rusty_sr_amd as r
    bl = r.bilinear_net(r.FACTOR)
    for k, (h, w, c) in enumerate(((2, 33001, 3), (3, 40000, 4))):
        px = synth_u8(970 + k, 1, h, w)
        want = oracle.bilinear(oracle.img_to_data(px))
        if c == 4:
            px = np.concatenate([px, synth_u8(980 + k, 1, h, w)[..., :1]], axis=-1)
        _check_u8(bl.upscale_rgba8(px), want)


def test_downsample_reads_any_channel_count_and_alignment():
    """downsample_net's threads read their 3x3 windows straight from global memory as the aligned words that hold them
    (sr_aux.hip): 3- and 4-channel u8 input, device pointers at every byte offset, widths with remainder columns, batches
    (so that rows start at every misalignment), against the oracle; the bilinear graph on the same views."""
    import torch
    import rusty_sr_amd as r
    bl, ds = r.bilinear_net(r.FACTOR), r.downsample_net(r.FACTOR)
    for k, (n, h, w, c) in enumerate(((1, 9, 9, 3), (2, 13, 71, 3), (3, 31, 200, 4), (1, 12, 193, 4), (2, 7, 65, 3))):
        px = synth_u8(900 + k, n, h, w)
        if c == 4:
            px = np.concatenate([px, synth_u8(950 + k, n, h, w)[..., :1]], axis=-1)  # an alpha channel nobody reads
        x = oracle.img_to_data(px[..., :3])
        want_ds, want_bl = oracle.downsample(x), oracle.bilinear(x)
        _check_u8(ds.upscale_rgba8(px), want_ds)
        for off in range(4):
            buf = torch.zeros(px.size + 8, dtype=torch.uint8, device="cuda")
            view = buf[off:off + px.size].view(n, h, w, c)
            view.copy_(torch.from_numpy(px))
            assert view.data_ptr() % 4 == (buf.data_ptr() + off) % 4
            _check_u8(ds.upscale_rgba8_dev(view).cpu().numpy(), want_ds)
            _check_u8(bl.upscale_rgba8_dev(view).cpu().numpy(), want_bl)
    # ragged sizes around every boundary of the two u8 kernels: W < 3 (the byte-wise window), W % 4 (16-byte or dword stores), blocks of
    # 64 chunks (3 W / 4 around 64, 128), one-row images, remainder rows / columns of the 3x3 mean
    rng = np.random.default_rng(1234)
    shapes = [(1, 1, 2, 3), (2, 2, 1, 4), (1, 3, 2, 3), (1, 1, 85, 3), (1, 2, 86, 3), (1, 1, 171, 4), (1, 5, 172, 3)]
    shapes += [(int(rng.integers(1, 4)), int(rng.integers(1, 40)), int(rng.integers(1, 300)), int(rng.integers(3, 5))) for _ in range(24)]
    for k, (n, h, w, c) in enumerate(shapes):
        px = synth_u8(2000 + k, n, h, w)
        if c == 4:
            px = np.concatenate([px, synth_u8(2100 + k, n, h, w)[..., :1]], axis=-1)
        x = oracle.img_to_data(px[..., :3])
        _check_u8(bl.upscale_rgba8(px), oracle.bilinear(x))
        if h >= 3 and w >= 3:
            _check_u8(ds.upscale_rgba8(px), oracle.downsample(x))
    ds.close()
    bl.close()


def test_parameter_free_graphs_soak():
    """A bounded soak of the two parameter-free graphs (the stage kernels have scripts/soak.py): random shapes, batches and channel
    counts against the oracle, u8 and f32; then whole frames repeated, which must reproduce themselves bit for bit (the u8 kernels
    prefetch the next item's pixels under sequence-counted waits: a missed wait shows up as a flicker)."""
    import time
    import torch
    import rusty_sr_amd as r
    bl, ds = r.bilinear_net(r.FACTOR), r.downsample_net(r.FACTOR)
    rng = np.random.default_rng(4321)
    t_end = time.time() + 12.0
    shapes = 0
    while time.time() < t_end and shapes < 60:
        n, h, w, c = int(rng.integers(1, 3)), int(rng.integers(1, 160)), int(rng.integers(1, 1200)), int(rng.integers(3, 5))
        px = rng.integers(0, 256, (n, h, w, c), dtype=np.uint8)
        x = oracle.img_to_data(px[..., :3])
        want_bl = oracle.bilinear(x)
        _check_u8(bl.upscale_rgba8(px), want_bl)
        assert np.abs(bl.upscale_f32(x) - want_bl).max() < 1e-5
        if h >= 3 and w >= 3:
            want_ds = oracle.downsample(x)
            _check_u8(ds.upscale_rgba8(px), want_ds)
            assert np.abs(ds.upscale_f32(x) - want_ds).max() < 1e-5
        shapes += 1
    assert shapes >= 10
    for (h, w) in ((1080, 1920), (2160, 3840), (1081, 1923)):
        px = torch.from_numpy(rng.integers(0, 256, (1, h, w, 3), dtype=np.uint8)).cuda()
        for eng in (bl, ds):
            first = eng.upscale_rgba8_dev(px).clone()
            out = torch.empty_like(first)
            for _ in range(25):
                eng.upscale_rgba8_dev(px, out=out)
                assert torch.equal(out, first), (eng.graph if hasattr(eng, "graph") else "", h, w)
    ds.close()
    bl.close()


def test_geometry_changes_keep_borders_clean(engines, params):
    """The feature maps carry their zero padding as a border in HBM that is only re-zeroed
    when (n, H, W) changes: interleave big / small / ragged / batched calls on ONE engine and
    require every result to equal a fresh engine's (stale data in a border would show up at
    the image edges)."""
    import rusty_sr_amd as r
    eng = engines["imagenet"]
    fresh = r.Engine(params["imagenet"], device=0, precision=eng.precision)
    shapes = [(1, 70, 130), (1, 9, 40), (2, 33, 65), (1, 70, 130), (1, 8, 32), (3, 5, 7), (1, 64, 96), (1, 9, 40)]
    for k, (n, h, w) in enumerate(shapes):
        x = oracle.img_to_data(synth_u8(50 + k, n, h, w))
        got = eng.upscale_f32(x)
        fresh.close()
        fresh = r.Engine(params["imagenet"], device=0, precision=eng.precision)
        np.testing.assert_array_equal(got, fresh.upscale_f32(x), err_msg=str((n, h, w)))
    fresh.close()


def test_random_shapes_sweep(engines, params):
    """Ragged sizes around every tile boundary (8x32 / 4x32 tiles, 36- and 34-wide halo tiles)."""
    rng = np.random.default_rng(99)
    for _ in range(12):
        h, w = int(rng.integers(1, 70)), int(rng.integers(1, 140))
        x = oracle.img_to_data(synth_u8(int(rng.integers(1 << 30)), 1, h, w))
        want = oracle.forward(params["anime"], x)
        got = engines["anime"].upscale_f32(x)
        assert np.abs(got - want).max() < TIGHT, (h, w)


def test_band_argument_validation(engines):
    import torch
    import rusty_sr_amd as r
    eng = engines["imagenet"]
    x = torch.zeros((30, 40, 3), dtype=torch.float32, device="cuda")
    for top, bot in ((3, 0), (0, 6), (16, 16), (-1, 0)):
        with pytest.raises(r.SrError):
            eng.upscale_band_f32_dev(x, top, bot)
    out = eng.upscale_band_f32_dev(x, 7, 7)
    assert out.shape == (48, 120, 3)


def _synthetic_params(factor, seed):
    """No 2x / 4x weights ship with the reference: seeded synthetic parameters with the bundled
    weights' scales (conv std from imagenet.rsr, small biases, BeLU betas in [-0.5, 1.5])."""
    rng = np.random.default_rng(seed)
    n = oracle.num_params(factor)
    p = (rng.standard_normal(n) * 0.03).astype(np.float32)
    e = 3 * factor * factor
    p[2400:2464 + e + 96] = (rng.standard_normal(64 + e + 96) * 0.05).astype(np.float32)       # biases
    p[2432:2464] = rng.uniform(-0.5, 1.5, 32).astype(np.float32)                                 # f_activ
    a0 = 2464 + e + 96
    p[a0:a0 + 96] = rng.uniform(-0.5, 1.5, 96).astype(np.float32)                                # l1..l3 activ
    return p


@pytest.mark.parametrize("factor", [2, 4])
@pytest.mark.parametrize("precision", ["f32", "split_f16"])
def test_other_factors_against_the_restatement(factor, precision):
    """sr_net(factor) for factor 2 and 4 (network.rs:16 takes it as an argument; main.rs:31 fixes 3).
    UNPINNED against the reference (no such weights exist): parity is against the oracle's
    generalised restatement, which is bit-identical to the pinned path at factor 3."""
    import torch
    import rusty_sr_amd as r
    p = _synthetic_params(factor, 100 + factor)
    eng = r.Engine(p, device=0, factor=factor, precision=precision)
    for n, h, w in ((1, 9, 33), (2, 40, 70), (1, 64, 96)):
        px = synth_u8(60 + h, n, h, w)
        x = oracle.img_to_data(px)
        want = oracle.forward_factor(p, x, factor)
        got = eng.upscale_f32(x)
        assert got.shape == (n, factor * h, factor * w, 3)
        assert np.abs(got - want).max() < TIGHT
        _check_u8(eng.upscale_rgba8(px), want)
    # row bands stay bit-identical at any factor
    x = oracle.img_to_data(synth_u8(77, 1, 50, 40))
    full = eng.upscale_f32(x)[0]
    xt = torch.from_numpy(x[0]).cuda()
    out = eng.upscale_band_f32_dev(xt[20 - 7:33 + 7].contiguous(), 7, 7)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(out.cpu().numpy(), full[factor * 20:factor * 33])
    # an image with more 8-row tiles than resident workgroups: persistent pipe-form stages (1-3, and the final
    # stage unless it needs two N-tiles) feeding / fed by first-form ones; whole output against the restatement
    px = synth_u8(91, 1, 296, 1100)
    x = oracle.img_to_data(px)
    got = eng.upscale_f32(x)
    assert np.abs(got - oracle.forward_factor(p, x, factor)).max() < TIGHT
    eng.close()


def test_factor3_general_path_is_the_pinned_path(params):
    x = oracle.img_to_data(synth_u8(5, 1, 21, 34))
    np.testing.assert_array_equal(oracle.forward_factor(params["anime"], x, 3), oracle.forward(params["anime"], x))


def test_forked_device_call_is_bit_identical(engines):
    """sr_upscale_*_dev / sr_upscale_band_*_dev may run one image as TWO row bands on two streams (sr_run_stack_auto: the drain of
    one band's launch under the fill of the other's): whatever the cut, the bytes must be those of the undivided pass -- u8 and
    f32, whole images and bands with halos, heights that leave every remainder of 8 rows, forced cuts and the automatic one --
    and the call must stay ordered on the caller's stream (the result is read on that stream without a device-wide sync)."""
    import torch
    eng = engines["imagenet"]
    rng = np.random.default_rng(77)
    side = torch.cuda.Stream()
    try:
        for (h, w, top, bot) in ((64, 96, 0, 0), (131, 257, 0, 0), (300, 515, 0, 0), (83, 640, 7, 0), (90, 333, 7, 7), (77, 1024, 0, 7), (1080, 1920, 0, 0)):
            px = torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).cuda()
            x = px.to(torch.float32) / 255.0
            band = top or bot
            def run8(stream=None):
                return eng.upscale_band_rgba8_dev(px, top, bot, stream=stream) if band else eng.upscale_rgba8_dev(px[None], stream=stream)[0]
            def run32(stream=None):
                return eng.upscale_band_f32_dev(x, top, bot, stream=stream) if band else eng.upscale_f32_dev(x[None], stream=stream)[0]
            eng.set_experiment("fork", "0")
            want8, want32 = run8(), run32()
            torch.cuda.synchronize()
            own = h - top - bot
            cuts = ["1"] + [str(c) for c in (14, 16, 21, own // 2 + 3, own - 14) if 14 <= c <= own - 14]
            for cut in cuts:
                eng.set_experiment("fork", cut)
                with torch.cuda.stream(side):
                    got8 = run8(side)
                    got32 = run32(side)
                    same8, same32 = torch.equal(got8, want8), torch.equal(got32, want32)  # on the caller's stream: ordered behind the join
                assert same8 and same32, (h, w, top, bot, cut)
    finally:
        eng.set_experiment("fork", "")


def test_fork_tuner_measures_both_plans_and_never_changes_a_byte(engines):
    """Mid-size shapes: whether a lone image runs undivided or as two bands is MEASURED on the caller's own calls (sr_internal.h ForkTune:
    four undivided calls, four forked, timed by event pairs that are queried, never waited for; then the faster plan stays).  Every call
    of the measurement and after it returns the undivided pass's bytes; the tuner settles after eight read samples when the caller fences
    between calls, goes on working under a deep queue of unfenced calls, leaves shapes outside its range to the rule (256x256, 1920x1080)
    and forgets on request."""
    import torch
    eng = engines["imagenet"]
    rng = np.random.default_rng(78)
    try:
        eng.set_experiment("forktune", "1")
        for (h, w) in ((448, 448), (300, 515)):
            px = torch.from_numpy(rng.integers(0, 256, (1, h, w, 3), dtype=np.uint8)).cuda()
            x = px.to(torch.float32) / 255.0
            seen = len(eng.get_experiment("forktune").splitlines())
            eng.set_experiment("fork", "0")
            want8, want32 = eng.upscale_rgba8_dev(px), eng.upscale_f32_dev(x)
            torch.cuda.synchronize()
            eng.set_experiment("fork", "")
            assert len(eng.get_experiment("forktune").splitlines()) == seen  # (a forced plan is not the tuner's business)
            for k in range(10):
                assert torch.equal(eng.upscale_rgba8_dev(px), want8), (h, w, k)
                torch.cuda.synchronize()
            lines = [l.split() for l in eng.get_experiment("forktune").splitlines()]
            mine = [l for l in lines if l[0] == f"{h}x{w}+0+0" and l[2] == "u8"]
            assert len(mine) == 1 and mine[0][3] in ("undivided", "forked"), lines
            assert float(mine[0][4]) > 0 and float(mine[0][5]) > 0
            for k in range(3):
                assert torch.equal(eng.upscale_rgba8_dev(px), want8)
            # the f32 entry point is a shape of its own, met here under a deep queue: no fence between the calls
            outs = [eng.upscale_f32_dev(x) for _ in range(24)]
            torch.cuda.synchronize()
            assert all(torch.equal(o, want32) for o in outs)
            del outs
        n_seen = len(eng.get_experiment("forktune").splitlines())
        assert n_seen == 4
        for (h, w) in ((256, 256), (1080, 1920)):  # 0.5 and 15.8 rounds of tiles: the rule's
            px = torch.from_numpy(rng.integers(0, 256, (1, h, w, 3), dtype=np.uint8)).cuda()
            eng.upscale_rgba8_dev(px)
        torch.cuda.synchronize()
        assert len(eng.get_experiment("forktune").splitlines()) == n_seen
        eng.set_experiment("forktune", "0")
        assert eng.get_experiment("forktune") == ""
        px = torch.from_numpy(rng.integers(0, 256, (1, 448, 448, 3), dtype=np.uint8)).cuda()
        eng.upscale_rgba8_dev(px)
        torch.cuda.synchronize()
        assert eng.get_experiment("forktune") == ""
        with pytest.raises(Exception):
            eng.get_experiment("no such key")
    finally:
        eng.set_experiment("fork", "")
        eng.set_experiment("forktune", "")


def test_reference_image_pins_through_the_gpu(engines, params):
    """The docs images the reference ships, fed to the ENGINE (tests/test_oracle_golden.py holds the same pins for the oracle, so
    that no fixture is seen by the oracle alone): `-p bilinear` against logo_lin.png (made by an older alumina whose data_to_img
    truncated: compared with a truncating quantiser, mismatches only at flat-region knife-edges), and imagenet.rsr on the 43 x 43
    logo and the butterfly against logo_rs.png / butterfly_rs.png (an earlier weight snapshot: sanity floors, as SURVEY.md 4 says)."""
    import rusty_sr_amd as r
    src = load_png("logo_nn.png")[1::3, 1::3]
    assert src.shape[:2] == (43, 43)
    # LinearInterp alignment + sRGB transfer (network.rs:111-123), f32 entry point of the bilinear graph
    bl = r.bilinear_net(r.FACTOR)
    v = bl.upscale_f32(oracle.img_to_data(src)[None])[0].astype(np.float64)
    gold = load_png("logo_lin.png")[..., :3].astype(int)
    d = np.clip(np.floor(255 * v), 0, 255).astype(int) - gold
    frac = 255 * v
    # The old quantiser cut at the integers, and on the logo's flat areas 255 v IS an integer up to rounding noise: whether a sample
    # reads k or k - 1 there is the sign of that noise (the oracle's f32 leg: 95 % equal, its f64 leg 99 %; the engine's v_log_f32 /
    # v_exp_f32 pow: 87 %).  What pins LinearInterp's alignment and the transfer curve is that EVERY sample lies in the interval the
    # golden byte stands for, give or take that noise -- and that every mismatch sits on such a knife-edge.
    assert ((frac >= gold - 1e-3) & (frac < gold + 1 + 1e-3)).all()
    assert np.abs(d).max() <= 1 and (d == 0).mean() >= 0.80
    assert np.abs(frac - np.round(frac))[d != 0].max() < 1e-3
    # ... and its u8 entry point (rounding quantiser) agrees with rounding the same values
    got8 = bl.upscale_rgba8(src[None])[0]
    want8 = np.clip(np.floor(255 * v + 0.5), 0, 255).astype(int)
    d8 = got8[..., :3].astype(int) - want8
    assert np.abs(d8).max() <= 1 and (d8 != 0).mean() < 1e-3
    # sr_net with imagenet.rsr, both arithmetic modes (the fixture)
    out = engines["imagenet"].upscale_rgba8(src[None])[0]
    gold = load_png("logo_rs.png")
    assert out.shape == gold.shape
    assert (out[..., :3] == gold[..., :3]).mean() >= 0.80
    assert np.abs(out[..., :3].astype(int) - gold[..., :3].astype(int)).max() <= 5
    out = engines["imagenet"].upscale_rgba8(load_png("butterfly_lr.png"))
    gold = load_png("butterfly_rs.png")
    mse = np.mean((out[..., :3].astype(float) - gold[..., :3].astype(float)) ** 2)
    assert 10 * np.log10(255 ** 2 / mse) >= 55.0
    assert (out[..., :3] == gold[..., :3]).mean() >= 0.80 and np.abs(out[..., :3].astype(int) - gold[..., :3].astype(int)).max() <= 5
