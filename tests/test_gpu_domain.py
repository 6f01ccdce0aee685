"""Inputs outside the comfortable range: graph.forward takes ANY f32 (reference main.rs:171), so the exact mode must follow IEEE
arithmetic through large values, infinities and NaNs like the CPU path does, and the split-half mode -- whose values travel as pairs
of f16 halves -- must never hand out silently clamped pixels: it refuses weights it cannot carry, its synchronous entry points
recompute in exact f32 when an input or activation leaves the f16 range, and its asynchronous ones raise a fault (include/srhip.h,
sr_set_precision / sr_check_domain)."""
import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module", params=["f32", "split_f16"])
def eng(params, request):
    import rusty_sr_amd as r
    e = r.Engine(params["imagenet"], device=0, precision=request.param)
    yield e
    e.close()


@pytest.mark.parametrize("scale", [1e3, 6e4, 1e6])
def test_large_inputs_keep_relative_parity(eng, params, scale):
    rng = np.random.default_rng(int(scale))
    x = ((rng.random((1, 40, 72, 3), dtype=np.float32) * 2 - 1) * np.float32(scale)).astype(np.float32)
    want = oracle.forward(params["imagenet"], x)
    assert np.isfinite(want).all()
    got = eng.upscale_f32(x)
    assert np.isfinite(got).all()
    assert np.abs(got - want).max() <= TOL * np.abs(want).max()
    eng.check_domain()  # the synchronous call has dealt with it: no fault is left behind


def test_denormal_inputs(eng, params):
    rng = np.random.default_rng(11)
    x = (rng.random((1, 24, 40, 3), dtype=np.float32) * np.float32(1e-40)).astype(np.float32)
    assert (x[x != 0] < np.finfo(np.float32).tiny).all()
    want = oracle.forward(params["imagenet"], x)
    got = eng.upscale_f32(x)
    assert np.abs(got - want).max() < TOL


def test_nan_and_infinities_propagate_like_the_cpu_path(eng, params):
    """Where the CPU path's output is finite the engine's is, within the bar; where it is not, the engine's is not either
    (whether a poisoned pixel reads Inf or NaN depends on 0 x Inf products of zero weights and is not compared)."""
    rng = np.random.default_rng(12)
    x = rng.random((1, 48, 64, 3), dtype=np.float32)
    x[0, 10, 10, 1] = np.nan
    x[0, 30, 40, 0] = np.inf
    x[0, 5, 50, 2] = -np.inf
    with np.errstate(all="ignore"):
        want = oracle.forward(params["imagenet"], x)
    got = eng.upscale_f32(x)
    fin = np.isfinite(want)
    assert fin.any() and not fin.all()
    np.testing.assert_array_equal(np.isfinite(got), fin)
    assert np.abs(got[fin] - want[fin]).max() < TOL
    # a poisoned pixel poisons exactly its receptive field (radius 7 input pixels): one input row / column further on, nothing
    bad = ~fin.reshape(48, 3, 64, 3, 3).any(axis=(1, 3, 4))
    assert bad[10 - 7:10 + 8, 10 - 7:10 + 8].all() and not bad[10 + 8, 10] and not bad[10, 10 + 8]


def test_u8_quantiser_on_nan_and_infinities(params):
    """data_to_img (main.rs:175) clamps and casts: Rust's float -> u8 cast saturates and sends NaN to 0."""
    import rusty_sr_amd as r
    p = params["imagenet"].copy()
    p[2464:2467] = [np.nan, np.inf, -np.inf]  # expand_bias of sub-pixel (0, 0): R, G, B
    px = np.random.default_rng(13).integers(0, 256, (1, 20, 36, 3), dtype=np.uint8)
    with np.errstate(all="ignore"):
        want = oracle.upscale_rgba8(p, px)
    assert (want[0, 0::3, 0::3, 0] == 0).all() and (want[0, 0::3, 0::3, 1] == 255).all() and (want[0, 0::3, 0::3, 2] == 0).all()
    for precision in ("f32", "split_f16"):
        e = r.Engine(p, device=0, precision=precision)
        got = e.upscale_rgba8(px)
        e.close()
        np.testing.assert_array_equal(got[0, 0::3, 0::3], want[0, 0::3, 0::3])
        d = np.abs(got.astype(int) - want.astype(int))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3


def test_split_mode_refuses_weights_it_cannot_carry(params):
    import rusty_sr_amd as r
    from rusty_sr_amd import _lib
    for bad in (1e5, np.inf, np.nan):
        p = params["imagenet"].copy()
        p[2683 + 17] = bad  # a conv1 weight
        with pytest.raises(r.SrError) as e:
            r.Engine(p, device=0, precision="split_f16")
        assert e.value.status == _lib.SR_E_DOMAIN
        eng = r.Engine(p, device=0, precision="f32")  # the exact mode takes any f32
        with pytest.raises(r.SrError):
            eng.set_precision("split_f16")
        assert eng.precision == "f32"
        eng.close()
    p = params["imagenet"].copy()
    p[2683 + 17] = 6e4  # large, but a half carries it
    r.Engine(p, device=0, precision="split_f16").close()


def test_device_entry_point_raises_a_fault_instead_of_clamping(params):
    import torch
    import rusty_sr_amd as r
    from rusty_sr_amd import _lib
    eng = r.Engine(params["imagenet"], device=0, precision="split_f16")
    rng = np.random.default_rng(14)
    x = rng.random((1, 40, 72, 3), dtype=np.float32)
    eng.upscale_f32_dev(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    eng.check_domain()  # an ordinary image: no fault
    for poison in (1e6, np.nan, np.inf):
        y = x.copy()
        y[0, 20, 30, 1] = poison
        eng.upscale_f32_dev(torch.from_numpy(y).cuda())
        torch.cuda.synchronize()
        with pytest.raises(r.SrError) as e:
            eng.check_domain()
        assert e.value.status == _lib.SR_E_DOMAIN
        eng.check_domain()  # reported once
    # activations, not only inputs: every input value is small, the first layer's outputs are not
    p = params["imagenet"].copy()
    p[2400:2432] = 7e4  # f_bias
    big = r.Engine(p, device=0, precision="split_f16")
    big.upscale_f32_dev(torch.from_numpy(x).cuda())
    torch.cuda.synchronize()
    with pytest.raises(r.SrError):
        big.check_domain()
    want = oracle.forward(p, x)
    got = big.upscale_f32(x)  # the synchronous entry point: recomputed in exact f32
    assert np.abs(got - want).max() <= TOL * np.abs(want).max()
    big.close()
    f32 = r.Engine(params["imagenet"], device=0, precision="f32")
    f32.upscale_f32_dev(torch.from_numpy(x * 1e6).cuda())
    torch.cuda.synchronize()
    f32.check_domain()  # the exact mode has no such domain
    f32.close()
    eng.close()


def test_an_unchecked_device_fault_is_not_the_next_host_calls(params):
    """A fault an earlier *_dev call left behind belongs to sr_check_domain: the next host-pointer call, whose values are all in
    range, must run in the split-half mode (bit for bit what a fresh context returns), not recompute in f32 -- and the fault is
    still reported afterwards, once, also across a switch of mode (include/srhip.h sr_check_domain)."""
    import torch
    import rusty_sr_amd as r
    from rusty_sr_amd import _lib
    rng = np.random.default_rng(15)
    x = rng.random((1, 40, 72, 3), dtype=np.float32)
    clean = r.Engine(params["imagenet"], device=0, precision="split_f16")
    want_split = clean.upscale_f32(x)
    clean.set_precision("f32")
    want_f32 = clean.upscale_f32(x)
    clean.close()
    assert not np.array_equal(want_split, want_f32)  # (the two modes differ in the last bits: the comparison below means something)
    eng = r.Engine(params["imagenet"], device=0, precision="split_f16")
    y = x.copy()
    y[0, 20, 30, 1] = 1e6
    eng.upscale_f32_dev(torch.from_numpy(y).cuda())
    torch.cuda.synchronize()
    got = eng.upscale_f32(x)
    assert np.array_equal(got, want_split)
    eng.set_precision("f32")
    with pytest.raises(r.SrError) as e:
        eng.check_domain()
    assert e.value.status == _lib.SR_E_DOMAIN
    eng.check_domain()
    eng.close()
