"""Build libsrhip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsrhip.so")
SOURCES = ["sr_kernels.hip", "sr_api.cpp"]
HEADERS = ["sr_kernels.h", os.path.join("..", "..", "include", "srhip.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, f)) > t for f in SOURCES + HEADERS)


def build_lib(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> rusty_sr_amd/libsrhip.so.  Cross-compiles
    without a GPU.  Returns the library path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    tmp = LIB + f".{os.getpid()}.tmp"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
           "-x", "hip", *[os.path.join(CSRC, f) for f in SOURCES], "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
