"""One image over every visible GPU from a plain C99 process (tests/c/comm_smoke.c): no torch, no Python in the process that
owns the RCCL communicator.  On a one-GPU box the RCCL leg prints "skipped: 1 device" and the peer-copy leg runs with two
contexts of device 0; on a multi-GPU box this is the first thing to look at when a sharded run misbehaves, because nothing
here depends on torch's RCCL instance."""
import os
import subprocess

import pytest

from conftest import ROOT


def _build(tmp_path):
    exe = tmp_path / "comm_smoke"
    cmd = ["gcc", "-std=c99", "-D_POSIX_C_SOURCE=199309L", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "c", "comm_smoke.c"), "-L", os.path.join(ROOT, "rusty_sr_amd"), "-lsrhip", "-L", "/opt/rocm/lib",
           "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "rusty_sr_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)]
    subprocess.check_call(cmd)
    return exe


def test_comm_smoke_compiles_as_c99(tmp_path):
    from rusty_sr_amd.build import build_lib
    build_lib()
    assert os.path.exists(_build(tmp_path))


@pytest.mark.gpu
def test_comm_smoke_runs(tmp_path):
    exe = _build(tmp_path)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([str(exe), os.path.join(ROOT, "rusty_sr_amd", "res", "imagenet.rsr")], capture_output=True, text=True,
                         timeout=600, env=env)
    print(res.stdout)
    assert res.returncode == 0, res.stdout + res.stderr
    assert res.stdout.rstrip().endswith("comm_smoke ok")
    assert "bit-identical" in res.stdout
    assert "config C, 3840x2160 over" in res.stdout and "halo exchange" in res.stdout  # the per-rank numbers a multi-GPU lease should yield


def test_first_multigpu_script_is_runnable():
    """scripts/first_multigpu.sh is what the first multi-GPU lease runs: it must at least parse, and name files that exist."""
    script = os.path.join(ROOT, "scripts", "first_multigpu.sh")
    subprocess.check_call(["bash", "-n", script])
    text = open(script).read()
    for path in ("tests/c/comm_smoke.c", "tests/test_gpu_multi.py", "bench.py", "rusty_sr_amd/res/imagenet.rsr"):
        assert path in text and os.path.exists(os.path.join(ROOT, path)), path
    for test in ("other_devices", "all_devices_with_rccl"):
        assert test in text and test in open(os.path.join(ROOT, "tests", "test_gpu_multi.py")).read()
