"""Pins the CPU oracle to the reference's own result images.
No GPU involved: this is the check that the checker is right."""
import hashlib
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, ROOT, load_png


def _psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 10 * np.log10(255.0 ** 2 / mse)


def test_fixture_hashes():
    """The fixtures are the files the reference ships (sha256 prefixes from SURVEY.md 8(a)/(c))."""
    want = {
        "rusty_sr_amd/res/imagenet.rsr": "5781fba8676d626b",
        "rusty_sr_amd/res/imagenetlinear.rsr": "4e37c3e1e4f0cf76",
        "rusty_sr_amd/res/anime.rsr": "aaf006212995ede0",
        "tests/golden/cartoon_lr.png": "308740cd3690e7c1",
        "tests/golden/cartoon_rsa.png": "422b36e7c22e48a8",
    }
    for rel, h in want.items():
        with open(os.path.join(ROOT, rel), "rb") as f:
            assert hashlib.sha256(f.read()).hexdigest()[:16] == h, rel


def test_rsr_format(params):
    for name, p in params.items():
        assert p.shape == (oracle.NPARAMS,) and p.dtype == np.float32
        assert np.isfinite(p).all()
    # segment table covers the blob exactly, in order
    pos = 0
    for name, (off, ln, shape) in oracle.SEGMENTS.items():
        assert off == pos and int(np.prod(shape)) == ln, name
        pos += ln
    assert pos == oracle.NPARAMS
    # expand_bias: 27 tight values
    eb = params["imagenet"][2464:2491]
    assert eb.std() < 0.05


def test_rsr_rejects_malformed():
    with pytest.raises(ValueError):
        oracle.rsr_decode(b"\x01\x00")
    with pytest.raises(ValueError):
        oracle.rsr_decode(b"\x02\x00\x00\x00" + b"\x04\x00\x00\x00" * 2 + b"\x00" * 4)  # short payload
    with pytest.raises(ValueError):
        oracle.rsr_decode(b"\x01\x00\x00\x00" + b"\x08\x00\x00\x00" + b"\x00" * 4)  # element size != 4


def test_cartoon_golden_pin(params):
    """docs/cartoon_lr.png + anime.rsr -> docs/cartoon_rsa.png: the reference's one
    bit-level result pin.  Gate: >= 99.99 % of u8 samples equal, max |d| = 1, and every
    mismatch sits on a rounding knife-edge of the f32 evaluation."""
    lr, gold = load_png("cartoon_lr.png"), load_png("cartoon_rsa.png")
    assert lr.shape == (120, 84, 4) and gold.shape == (360, 252, 4)
    out = oracle.upscale_rgba8(params["anime"], lr)[0]
    assert out.shape == gold.shape
    assert (out[..., 3] == 255).all() and (gold[..., 3] == 255).all()
    d = out[..., :3].astype(int) - gold[..., :3].astype(int)
    assert np.abs(d).max() <= 1
    assert (d == 0).mean() >= 0.9999
    v = oracle.forward(params["anime"], oracle.img_to_data(lr))[0]
    frac = 255.0 * v.astype(np.float64) + 0.5
    edge = np.abs(frac - np.round(frac))
    assert edge[d != 0].max() < 1e-3
    # the f64 evaluation of the same spec reproduces the golden exactly
    v64 = oracle.forward(params["anime"], oracle.img_to_data(lr), f64=True)[0]
    out64 = oracle.data_to_rgba8(v64.astype(np.float32))
    assert (out64 == gold).all()
    assert np.abs(v - v64).max() < 5e-6


def test_butterfly_and_logo_sanity_floor(params):
    """imagenet.rsr goldens were made with an earlier weight snapshot: PSNR floor only."""
    out = oracle.upscale_rgba8(params["imagenet"], load_png("butterfly_lr.png"))[0]
    assert _psnr(out[..., :3], load_png("butterfly_rs.png")[..., :3]) >= 55.0
    nn = load_png("logo_nn.png")
    src = nn[1::3, 1::3]
    out = oracle.upscale_rgba8(params["imagenet"], src)[0]
    gold = load_png("logo_rs.png")
    assert out.shape == gold.shape
    assert (out[..., :3] == gold[..., :3]).mean() >= 0.80
    assert np.abs(out[..., :3].astype(int) - gold[..., :3].astype(int)).max() <= 5


def test_degenerate_sizes_and_properties(params):
    p = params["imagenet"]
    rng = np.random.default_rng(0)
    for h, w in ((1, 1), (2, 3), (3, 1), (7, 9)):
        x = rng.random((h, w, 3), dtype=np.float32)
        y = oracle.forward(p, x)[0]
        assert y.shape == (3 * h, 3 * w, 3) and np.isfinite(y).all()
    # translation equivariance away from borders: crop with 7-px halo == crop of full
    x = rng.random((40, 44, 3), dtype=np.float32)
    full = oracle.forward(p, x)[0]
    sub = oracle.forward(p, x[5:35, 6:40])[0]
    np.testing.assert_array_equal(sub[21:-21, 21:-21], full[15 + 21:105 - 21, 18 + 21:120 - 21])
    # batch == per-image
    xb = rng.random((2, 9, 11, 3), dtype=np.float32)
    yb = oracle.forward(p, xb)
    np.testing.assert_array_equal(yb[1], oracle.forward(p, xb[1])[0])
    with pytest.raises(ValueError):
        oracle.forward(p[:-1], x)


def test_bilinear_net_pin_logo_lin():
    """`-p bilinear` (network.rs:111-123): docs/logo_lin.png was made from the 43x43 logo by an
    older alumina whose data_to_img truncated instead of rounding: with a
    truncating quantiser the f64 evaluation reproduces it to >= 99 %, every f32 mismatch is a
    flat-region knife-edge (255 v within 1e-3 of an integer), max |d| = 1.  Pins LinearInterp's
    half-pixel alignment and the sRGB transfer curve."""
    src = load_png("logo_nn.png")[1::3, 1::3]
    gold = load_png("logo_lin.png")[..., :3].astype(int)
    x = oracle.img_to_data(src)
    for f64, floor_exact in ((True, 0.99), (False, 0.95)):
        v = oracle.bilinear(x, f64=f64)[0].astype(np.float64)
        d = np.clip(np.floor(255 * v), 0, 255).astype(int) - gold
        assert np.abs(d).max() <= 1 and (d == 0).mean() >= floor_exact
        frac = 255 * v
        assert np.abs(frac - np.round(frac))[d != 0].max() < 1e-3
    assert np.abs(oracle.bilinear(x)[0] - oracle.bilinear(x, f64=True)[0]).max() < 1e-6


def test_downsample_net_properties():
    """`-d` (network.rs:125-138) has no reference image: UNPINNED.  Size-independent properties:
    a 3x nearest-neighbour upsample pools back to the original; flat images stay flat; the
    remainder rows / columns are dropped."""
    nn = load_png("logo_nn.png")
    src = oracle.img_to_data(nn[1::3, 1::3])
    np.testing.assert_allclose(oracle.downsample(oracle.img_to_data(nn))[0], src, atol=2e-7)
    flat = np.full((1, 9, 12, 3), 0.3, np.float32)
    np.testing.assert_allclose(oracle.downsample(flat), 0.3, atol=1e-6)
    rng = np.random.default_rng(3)
    x = rng.random((1, 11, 13, 3), dtype=np.float32)
    np.testing.assert_array_equal(oracle.downsample(x), oracle.downsample(x[:, :9, :12]))


VEC_CASES = ("crop", "border", "one", "twothree")


def load_vectors():
    return np.load(os.path.join(GOLDEN, "vectors_torch_f64.npz"))


@pytest.mark.parametrize("case", VEC_CASES)
def test_second_restatement_vectors(params, case):
    """tests/golden/vectors_torch_f64.npz (made by tests/golden/make_vectors.py) holds every node of
    the graph computed by an independent restatement -- torch conv2d / interpolate in float64.  The C
    oracle must agree node by node: in float64 to rounding of the f32 fixture, in float32 to
    accumulation noise.  Covers zero padding on all four sides of every layer and the 1x1 / 2x3 clamps."""
    v = load_vectors()
    p = params[str(v[f"{case}.weights"])]
    x = oracle.img_to_data(v[f"{case}.px"])
    for f64, tol in ((True, 4e-7), (False, 1e-5)):
        out, taps = oracle.forward_taps(p, x, f64=f64)
        for k in ("f", "l1", "l2", "l3", "e"):
            want = v[f"{case}.{k}"]
            got = taps[k][..., ::4] if want.shape[-1] == 8 else taps[k]
            assert np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max()), (case, k, f64)
        assert out[0].shape == v[f"{case}.out"].shape
        assert np.abs(out[0] - v[f"{case}.out"]).max() <= tol * 2, (case, "out", f64)
