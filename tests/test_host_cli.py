"""Host side of the drop-in: the C++ CLI with the reference's argv surface
(src/main.rs:33-178) and its PNG codec (stand-in for the `image` crate calls at
main.rs:164,175)."""
import ctypes as C
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, ROOT, load_png, synth_u8

CLI = os.path.join(ROOT, "rusty_sr_amd", "bin", "rusty_sr")
PNGLIB = os.path.join(ROOT, "rusty_sr_amd", "libsrpng.so")


@pytest.fixture(scope="module")
def png():
    from rusty_sr_amd.build import build_host
    build_host()
    L = C.CDLL(PNGLIB)
    L.srpng_decode_rgba8.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.POINTER(C.c_uint8))]
    L.srpng_encode_rgba8.argtypes = [C.c_char_p, C.POINTER(C.c_uint8), C.c_int, C.c_int]

    class Codec:
        @staticmethod
        def decode(path):
            w, h, p = C.c_int(), C.c_int(), C.POINTER(C.c_uint8)()
            if L.srpng_decode_rgba8(str(path).encode(), C.byref(w), C.byref(h), C.byref(p)) != 0:
                raise ValueError("decode failed")
            a = np.ctypeslib.as_array(p, (h.value, w.value, 4)).copy()
            L.srpng_free(p)
            return a

        @staticmethod
        def encode(path, a):
            a = np.ascontiguousarray(a, dtype=np.uint8)
            assert L.srpng_encode_rgba8(str(path).encode(), a.ctypes.data_as(C.POINTER(C.c_uint8)), a.shape[1], a.shape[0]) == 0
    return Codec


def test_png_decode_matches_pillow_on_reference_images(png):
    from PIL import Image
    for name in sorted(os.listdir(GOLDEN)):
        if name.endswith(".png"):
            want = np.array(Image.open(os.path.join(GOLDEN, name)).convert("RGBA"))
            np.testing.assert_array_equal(png.decode(os.path.join(GOLDEN, name)), want, err_msg=name)


def test_png_colour_types_depths_and_roundtrip(png, tmp_path):
    from PIL import Image
    rng = np.random.default_rng(0)
    base = rng.integers(0, 256, (13, 17, 4), dtype=np.uint8)
    for mode in ("L", "LA", "RGB", "RGBA", "P", "1"):
        p = tmp_path / f"{mode}.png"
        Image.fromarray(base).convert(mode).save(p)
        np.testing.assert_array_equal(png.decode(p), np.array(Image.open(p).convert("RGBA")), err_msg=mode)
    g16 = rng.integers(0, 65536, (9, 11)).astype(np.uint16)
    Image.fromarray(g16).save(tmp_path / "g16.png")
    np.testing.assert_array_equal(png.decode(tmp_path / "g16.png")[..., 0], (g16 >> 8).astype(np.uint8))
    # encoder: RGBA8, readable by Pillow, exact
    for shape in ((1, 1, 4), (7, 5, 4), (360, 252, 4)):
        a = rng.integers(0, 256, shape, dtype=np.uint8)
        png.encode(tmp_path / "o.png", a)
        im = Image.open(tmp_path / "o.png")
        assert im.mode == "RGBA"
        np.testing.assert_array_equal(np.array(im), a)
        np.testing.assert_array_equal(png.decode(tmp_path / "o.png"), a)
    with pytest.raises(ValueError):
        png.decode(os.path.join(ROOT, "rusty_sr_amd", "res", "anime.rsr"))


def _chunk(t, d):
    return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))


def test_png_adam7_interlaced(png, tmp_path):
    """Interlaced PNGs (Pillow cannot write them): build one by hand, filter type 0."""
    rng = np.random.default_rng(1)
    h, w = 11, 19
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    xs, ys, dx, dy = (0, 4, 0, 2, 0, 1, 0), (0, 0, 4, 0, 2, 0, 1), (8, 8, 4, 4, 2, 2, 1), (8, 8, 8, 4, 4, 2, 2)
    raw = b""
    for k in range(7):
        sub = img[ys[k]::dy[k], xs[k]::dx[k]]
        if sub.size:
            raw += b"".join(b"\x00" + row.tobytes() for row in sub)
    data = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) + \
        _chunk(b"IDAT", zlib.compress(raw)) + _chunk(b"IEND", b"")
    (tmp_path / "a7.png").write_bytes(data)
    got = png.decode(tmp_path / "a7.png")
    np.testing.assert_array_equal(got[..., :3], img)
    assert (got[..., 3] == 255).all()


def _run(*args):
    return subprocess.run([CLI, *args], capture_output=True, text=True, timeout=120)


def test_cli_argv_surface(png):
    """clap rules of the reference (main.rs:34-116): required positionals, possible values,
    conflicts; `train` is declined; nothing here needs a GPU."""
    r = _run("--version")
    assert r.returncode == 0 and "Rusty SR v0.1.1" in r.stdout
    r = _run("--help")
    assert r.returncode == 0 and "<INPUT_FILE>" in r.stdout and "--parameters" in r.stdout and "--downsample" in r.stdout
    r = _run()
    assert r.returncode == 2 and "required arguments were not provided" in r.stderr
    r = _run("in.png")
    assert r.returncode == 2
    r = _run("a.png", "b.png", "-p", "nonsense")
    assert r.returncode == 2 and "isn't a valid value" in r.stderr
    r = _run("a.png", "b.png", "-p", "anime", "-c", "x.rsr")
    assert r.returncode == 2 and "cannot be used with" in r.stderr
    r = _run("a.png", "b.png", "-d", "-p", "anime")
    assert r.returncode == 2 and "cannot be used with" in r.stderr
    r = _run("a.png", "b.png", "--bogus")
    assert r.returncode == 2
    r = _run("train", "p.rsr", "folder")
    assert r.returncode == 2 and "train" in r.stderr
    r = _run("a.png", "b.png", "-c", "/nonexistent/x.rsr")
    assert r.returncode == 1 and "Error opening parameter file" in r.stderr  # main.rs:134


@pytest.mark.gpu
def test_cli_end_to_end(png, tmp_path, params):
    """rusty_sr IN OUT [-p ..|-c ..|-d] on a GPU: same stdout text as the reference
    (main.rs:137-177), cartoon golden >= 99.99 %, -c == -p, bilinear / downsample vs oracle."""
    out = tmp_path / "o.png"
    r = _run(os.path.join(GOLDEN, "cartoon_lr.png"), str(out), "-p", "anime")
    assert r.returncode == 0, r.stderr
    assert r.stdout == "Upscaling using anime neural net parameters... Writing file... Done\n"
    got, gold = png.decode(out), load_png("cartoon_rsa.png")
    d = got[..., :3].astype(int) - gold[..., :3].astype(int)
    assert np.abs(d).max() <= 1 and (d == 0).mean() >= 0.9999 and (got[..., 3] == 255).all()
    # default parameters = imagenet (main.rs:144), split-half mode agrees to the knife-edge
    r = _run(os.path.join(GOLDEN, "butterfly_lr.png"), str(out), "--precision", "split_f16", "--timing")
    assert r.returncode == 0 and r.stdout.startswith("Upscaling using imagenet neural net parameters...") and "[timing]" in r.stderr
    want = oracle.upscale_rgba8(params["imagenet"], load_png("butterfly_lr.png"))[0]
    d = png.decode(out)[..., :3].astype(int) - want[..., :3].astype(int)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 1e-3
    # -c FILE (main.rs:133-138) with the anime blob == -p anime
    out2 = tmp_path / "o2.png"
    r = _run(os.path.join(GOLDEN, "cartoon_lr.png"), str(out2), "-c", os.path.join(ROOT, "rusty_sr_amd", "res", "anime.rsr"))
    assert r.returncode == 0 and r.stdout.startswith("Upscaling using custom neural net parameters...")
    np.testing.assert_array_equal(png.decode(out2), got)
    bad = tmp_path / "bad.rsr"
    bad.write_bytes(b"\x03\x00\x00\x00" + b"\x04\x00\x00\x00" * 3 + b"\x00" * 12)  # 3 params: count mismatch
    r = _run(os.path.join(GOLDEN, "cartoon_lr.png"), str(out2), "-c", str(bad))
    assert r.returncode == 1 and "Parameters selected do not have the size required" in r.stderr  # main.rs:162
    # -p bilinear and -d
    src = tmp_path / "src.png"
    px = synth_u8(40, 1, 33, 47)[0]
    png.encode(src, np.concatenate([px, np.full(px.shape[:2] + (1,), 255, np.uint8)], -1))
    r = _run(str(src), str(out), "-p", "bilinear")
    assert r.returncode == 0 and r.stdout.startswith("Upscaling using bilinear interpolation...")
    want = oracle.data_to_rgba8(oracle.bilinear(oracle.img_to_data(px))[0])
    d = png.decode(out).astype(int) - want.astype(int)
    assert np.abs(d).max() <= 1 and (d != 0).mean() < 1e-3
    r = _run(str(src), str(out), "-d")
    assert r.returncode == 0 and r.stdout.startswith("Downsampling using average pooling of linear RGB values...")
    want = oracle.data_to_rgba8(oracle.downsample(oracle.img_to_data(px))[0])
    d = png.decode(out).astype(int) - want.astype(int)
    assert want.shape == (11, 15, 4) and np.abs(d).max() <= 1 and (d != 0).mean() < 1e-2
    r = _run("/nonexistent.png", str(out))
    assert r.returncode == 1 and "Error opening input image file." in r.stderr  # main.rs:164
