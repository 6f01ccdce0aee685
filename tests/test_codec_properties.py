"""Property tests of the host codecs (rusty_sr_amd/host: the stand-ins for image::open / .save, reference main.rs:164,175)
against Pillow as an independent implementation: whatever Pillow writes in a lossless container we must read to the same
pixels, whatever we write Pillow must read back exactly (PNG, BMP, PPM) or closely (JPEG), for random sizes and contents --
flat, smooth and noisy, so that every PNG filter, long runs and incompressible data all occur."""
import ctypes as C
import io
import os

import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings, strategies as st

from conftest import ROOT

PNGLIB = os.path.join(ROOT, "rusty_sr_amd", "libsrpng.so")


@pytest.fixture(scope="module")
def codec():
    from rusty_sr_amd.build import build_host
    build_host()
    L = C.CDLL(PNGLIB)

    class Codec:
        @staticmethod
        def decode(path):
            W, H, P = C.c_int(), C.c_int(), C.POINTER(C.c_uint8)()
            if L.srpng_decode_any_rgba8(str(path).encode(), C.byref(W), C.byref(H), C.byref(P)):
                return None
            a = np.ctypeslib.as_array(P, shape=(H.value, W.value, 4)).copy()
            L.srpng_free(P)
            return a

        @staticmethod
        def encode(path, rgba):
            rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
            return L.srpng_encode_any_rgba8(str(path).encode(), rgba.ctypes.data_as(C.POINTER(C.c_uint8)), rgba.shape[1], rgba.shape[0])
    return Codec


@st.composite
def images(draw, max_side=96):
    h = draw(st.integers(1, max_side))
    w = draw(st.integers(1, max_side))
    kind = draw(st.sampled_from(["noise", "flat", "ramp", "blocks", "mixed"]))
    seed = draw(st.integers(0, 2 ** 31 - 1))
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    if kind == "noise":
        a = rng.integers(0, 256, (h, w, 4))
    elif kind == "flat":
        a = np.broadcast_to(rng.integers(0, 256, 4), (h, w, 4)).copy()
    elif kind == "ramp":
        a = np.stack([(xx * 3 + yy) % 256, (yy * 5) % 256, (xx + yy * 2) % 256, np.full((h, w), 255)], -1)
    elif kind == "blocks":
        a = rng.integers(0, 256, ((h + 7) // 8, (w + 7) // 8, 4)).repeat(8, 0).repeat(8, 1)[:h, :w]
    else:
        a = np.where((yy // 3 % 2 == 0)[..., None], rng.integers(0, 256, (h, w, 4)), np.broadcast_to(rng.integers(0, 256, 4), (h, w, 4)))
    return a.astype(np.uint8)


COMMON = dict(deadline=None, suppress_health_check=[HealthCheck.function_scoped_fixture, HealthCheck.too_slow])


@settings(max_examples=150, **COMMON)
@given(img=images(), mode=st.sampled_from(["RGBA", "RGB", "L", "LA", "P", "1"]), level=st.sampled_from([0, 1, 6, 9]))
def test_png_written_by_pillow_decodes_to_pillows_pixels(codec, tmp_path, img, mode, level):
    from PIL import Image
    p = tmp_path / "a.png"
    im = Image.fromarray(img).convert(mode)
    im.save(p, compress_level=level)
    want = np.array(Image.open(p).convert("RGBA"))
    got = codec.decode(p)
    assert got is not None and got.shape == want.shape
    np.testing.assert_array_equal(got, want)


@settings(max_examples=150, **COMMON)
@given(img=images(max_side=200))
def test_png_bmp_ppm_written_by_us_read_back_exactly(codec, tmp_path, img):
    from PIL import Image
    for ext, keep_alpha in (("png", True), ("bmp", False), ("ppm", False)):
        p = tmp_path / f"o.{ext}"
        assert codec.encode(p, img) == 0, ext
        back = np.array(Image.open(p).convert("RGBA"))
        if keep_alpha:
            np.testing.assert_array_equal(back, img, err_msg=ext)
        else:
            np.testing.assert_array_equal(back[..., :3], img[..., :3], err_msg=ext)
        ours = codec.decode(p)
        np.testing.assert_array_equal(ours[..., :3], img[..., :3], err_msg=ext)


@settings(max_examples=150, **COMMON)
@given(img=images(max_side=120), fmt=st.sampled_from([("TIFF", {"compression": "tiff_lzw"}), ("TIFF", {"compression": "packbits"}), ("TIFF", {"compression": "tiff_adobe_deflate"}),
                                                      ("TIFF", {"compression": "raw"}), ("GIF", {}), ("TGA", {}), ("TGA", {"compression": "tga_rle"}), ("BMP", {}), ("PPM", {})]))
def test_lossless_containers_written_by_pillow(codec, tmp_path, img, fmt):
    from PIL import Image
    name, kw = fmt
    ext = {"TIFF": "tif", "GIF": "gif", "TGA": "tga", "BMP": "bmp", "PPM": "ppm"}[name]
    p = tmp_path / f"a.{ext}"
    im = Image.fromarray(img[..., :3])
    if name == "GIF":
        im = im.convert("P")
    im.save(p, name, **kw)
    want = np.array(Image.open(p).convert("RGB"))
    got = codec.decode(p)
    assert got is not None and got.shape[:2] == want.shape[:2], fmt
    np.testing.assert_array_equal(got[..., :3], want, err_msg=str(fmt))


@settings(max_examples=100, **COMMON)
@given(img=images(max_side=150), quality=st.sampled_from([60, 85, 95]), sub=st.sampled_from([0, 1, 2]), prog=st.booleans())
def test_jpeg_both_ways_within_a_few_levels(codec, tmp_path, img, quality, sub, prog):
    """JPEG is lossy and decoders differ in IDCT and chroma upsampling: ours against libjpeg's on Pillow-written files, and
    Pillow against ours on the file we write, stay within a few levels on average."""
    from PIL import Image
    p = tmp_path / "a.jpg"
    Image.fromarray(img[..., :3]).save(p, quality=quality, subsampling=sub, progressive=prog)
    want = np.array(Image.open(p).convert("RGB")).astype(int)
    got = codec.decode(p)
    assert got is not None and got.shape[:2] == want.shape[:2]
    d = np.abs(got[..., :3].astype(int) - want)
    assert d.mean() < 2.5 and np.percentile(d, 99) <= 24, (d.mean(), d.max())
    q = tmp_path / "o.jpg"
    assert codec.encode(q, img) == 0
    theirs = np.array(Image.open(q).convert("RGB")).astype(int)
    ours = codec.decode(q)[..., :3].astype(int)
    assert np.abs(theirs - ours).mean() < 1.5


def test_highly_compressible_lzw_tiff_is_not_refused(codec, tmp_path):
    """A long run in ONE strip compresses ~1230:1 under LZW (a 4000x4000 solid grey image is 13 KB from libtiff): the
    "header the data cannot back" guard of the TIFF reader must use the codec's own best case, not deflate's 1032:1."""
    from PIL import Image
    p = tmp_path / "solid.tif"
    Image.new("L", (4000, 4000), 128).save(p, compression="tiff_lzw", tiffinfo={278: 4000})  # RowsPerStrip = the whole image
    assert os.path.getsize(p) < 40000
    rgba = codec.decode(str(p))
    assert rgba is not None and rgba.shape == (4000, 4000, 4)
    assert (rgba[..., :3] == 128).all() and (rgba[..., 3] == 255).all()
