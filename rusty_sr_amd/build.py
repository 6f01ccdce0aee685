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
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread",
           "-x", "hip", *[os.path.join(CSRC, f) for f in SOURCES], "-o", tmp]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(tmp, LIB)
    return LIB


HOST = os.path.join(HERE, "host")
CLI = os.path.join(HERE, "bin", "rusty_sr")
PNGLIB = os.path.join(HERE, "libsrpng.so")


def build_host(force=False, verbose=False):
    """g++ -> rusty_sr_amd/bin/rusty_sr (the CLI with the reference's argv surface; links
    libsrhip.so via $ORIGIN/..) and rusty_sr_amd/libsrpng.so (the PNG codec alone, for tests)."""
    srcs = [os.path.join(HOST, f) for f in ("main.cpp", "png.cpp", "jpeg.cpp")]
    deps = srcs + [os.path.join(HOST, "png.hpp"), os.path.join(HERE, "..", "include", "srhip.h"), LIB]
    if not force and os.path.exists(CLI) and os.path.exists(PNGLIB) and \
            all(os.path.getmtime(d) <= min(os.path.getmtime(CLI), os.path.getmtime(PNGLIB)) for d in deps):
        return CLI
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    res = os.path.join(HERE, "res")
    cmds = [
        ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", srcs[1], srcs[2], "-lz", "-o", PNGLIB],
        ["g++", "-O2", "-std=c++17", "-pthread", f'-DSR_RES_DIR="{res}"', *srcs, "-L", HERE, "-lsrhip", "-lz",
         "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath-link," + "/opt/rocm/lib", "-o", CLI],
    ]
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
