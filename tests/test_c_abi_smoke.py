"""The C ABI from plain C99 (tests/c/abi_smoke.c): gcc, no C++ / Python / torch in the process -- what a maintainer's FFI
stub does.  Its RGBA8 output must equal the Python host's for the same pixels."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT


def _build(tmp_path):
    exe = tmp_path / "abi_smoke"
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "abi_smoke.c"),
           "-L", os.path.join(ROOT, "rusty_sr_amd"), "-lsrhip", "-Wl,-rpath," + os.path.join(ROOT, "rusty_sr_amd"),
           "-Wl,-rpath-link,/opt/rocm/lib", "-o", str(exe)]
    subprocess.check_call(cmd)
    return exe


def test_c_smoke_compiles_as_c99(tmp_path):
    from rusty_sr_amd.build import build_lib
    build_lib()
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_c_smoke_runs_and_matches_the_python_host(tmp_path):
    import rusty_sr_amd as r
    exe = _build(tmp_path)
    out = tmp_path / "out.bin"
    res = subprocess.run([str(exe), os.path.join(ROOT, "rusty_sr_amd", "res", "imagenet.rsr"), str(out)], capture_output=True, text=True, timeout=300)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.startswith("abi_smoke ok")
    # the same xorshift32 pixels, through the Python mirror
    n, h, w = 3, 40, 70
    z, px = 2463534242, np.empty(n * h * w * 3, np.uint8)
    for i in range(px.size):
        z ^= (z << 13) & 0xffffffff; z ^= z >> 17; z ^= (z << 5) & 0xffffffff
        px[i] = z >> 24
    eng = r.Engine(r.rsr.builtin("imagenet"), device=0)
    want = eng.upscale_rgba8(px.reshape(n, h, w, 3))[0]
    got = np.fromfile(out, np.uint8).reshape(3 * h, 3 * w, 4)
    np.testing.assert_array_equal(got, want)
    eng.close()
