"""Build libsrhip.so (HIP kernels + C ABI) in-tree with hipcc for gfx950."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsrhip.so")
SOURCES = ["sr_kernels.hip", "sr_aux.hip", "sr_api.cpp", "sr_comm.cpp"]
DEVICE_ASM = os.path.join(HERE, "build", "sr_kernels.gfx950.s")  # device assembly of the stage kernels, kept for the ISA lint
HEADERS = ["sr_kernels.h", "sr_internal.h", os.path.join("..", "..", "include", "srhip.h")]


FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-pthread"]
# the stage kernels only: the split-half epilogues spell their pair arithmetic as scalar instructions on purpose (sr_kernels.hip f32p) and
# must not be packed again; the exact mode's packed instructions are explicit vector types and stay
KERNEL_FLAGS = ["-fno-slp-vectorize"]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> rusty_sr_amd/libsrhip.so.  Cross-compiles without a GPU.  Each source is
    compiled to its own object (build/, git-ignored) so that a change to the host side does not recompile the
    kernels.  Returns the library path."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs, relink = [], force or not os.path.exists(LIB)
    for f in SOURCES:
        src, obj = os.path.join(CSRC, f), os.path.join(objdir, f + ".o")
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            tmp = obj + f".{os.getpid()}.tmp"
            cmd = [hipcc, *FLAGS, "-x", "hip", "-c", src, "-o", tmp]
            temps = None
            if f == "sr_kernels.hip":
                # keep the device assembly of the stage kernels: scripts/check_async_regs.py lints it (tests/test_abi.py).  The
                # compiler's temporaries go to a directory of this process's own (two builds at once -- parallel tests, several
                # ranks on a fresh checkout -- must not write each other's files); object and assembly are then moved into place
                temps = os.path.join(objdir, f"temps.{os.getpid()}")
                os.makedirs(temps, exist_ok=True)
                tmp = os.path.join(temps, f + ".o")
                cmd = [hipcc, "-save-temps=obj", *FLAGS, *KERNEL_FLAGS, "-x", "hip", "-c", src, "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd)
                os.replace(tmp, obj)
                if temps:
                    # (the temporary's name is the compiler's: <stem>-hip-amdgcn-amd-amdhsa-gfx950.s today; take whichever device .s is there)
                    asm = [n for n in os.listdir(temps) if n.endswith(".s") and "amdgcn" in n]
                    if asm:
                        os.replace(os.path.join(temps, asm[0]), DEVICE_ASM)
            finally:
                if temps:
                    shutil.rmtree(temps, ignore_errors=True)
            relink = True
    if relink or _newer(LIB, objs):
        tmp = LIB + f".{os.getpid()}.tmp"
        cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread", *objs, "-ldl", "-o", tmp]
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
    srcs = [os.path.join(HOST, f) for f in ("main.cpp", "png.cpp", "jpeg.cpp", "formats.cpp", "rle_deflate.cpp")]
    deps = srcs + [os.path.join(HOST, "png.hpp"), os.path.join(HERE, "..", "include", "srhip.h"), LIB]
    if not force and os.path.exists(CLI) and os.path.exists(PNGLIB) and \
            all(os.path.getmtime(d) <= min(os.path.getmtime(CLI), os.path.getmtime(PNGLIB)) for d in deps):
        return CLI
    os.makedirs(os.path.dirname(CLI), exist_ok=True)
    res = os.path.join(HERE, "res")
    cmds = [
        ["g++", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread", *srcs[1:], "-lz", "-o", PNGLIB],
        ["g++", "-O3", "-std=c++17", "-pthread", f'-DSR_RES_DIR="{res}"', *srcs, "-L", HERE, "-lsrhip", "-lz",
         "-Wl,-rpath,$ORIGIN/..", "-Wl,-rpath-link," + "/opt/rocm/lib", "-o", CLI],
    ]
    for cmd in cmds:
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return CLI


FUZZ = os.path.join(HERE, "bin", "srcodec_asan")


def build_sanitized(force=False, verbose=False):
    """g++ -fsanitize=address,undefined -> rusty_sr_amd/bin/srcodec_asan: the hand-written PNG / JPEG / PNM / BMP
    decoders under AddressSanitizer + UBSan, driven by tests/test_decoder_robustness.py with truncated and
    bit-flipped files."""
    srcs = [os.path.join(HOST, f) for f in ("fuzz_main.cpp", "png.cpp", "jpeg.cpp", "formats.cpp", "rle_deflate.cpp")]
    deps = srcs + [os.path.join(HOST, "png.hpp")]
    if not force and os.path.exists(FUZZ) and all(os.path.getmtime(d) <= os.path.getmtime(FUZZ) for d in deps):
        return FUZZ
    os.makedirs(os.path.dirname(FUZZ), exist_ok=True)
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-fno-omit-frame-pointer", *srcs, "-lz", "-o", FUZZ]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return FUZZ


if __name__ == "__main__":
    print(build_lib(force=True, verbose=True))
    print(build_host(force=True, verbose=True))
    print(build_sanitized(force=True, verbose=True))
