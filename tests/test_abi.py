"""The C-ABI boundary (include/srhip.h <-> libsrhip.so) and the host-side logic that
needs no GPU.  Compute entry points are exercised in test_gpu_parity.py."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle
from conftest import ROOT, gpu_available


def _header_functions(header="srhip.h"):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sr_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    """Every function any header under include/ declares is exported; the drop-in header (srhip.h) and the tuning header
    (srhip_experimental.h) are bound by separate tables, so that a switch cannot drift into the stable ABI unnoticed."""
    from rusty_sr_amd import _lib
    L = _lib.lib()
    assert sorted(os.listdir(os.path.join(ROOT, "include"))) == ["srhip.h", "srhip_experimental.h"]
    declared = _header_functions()
    assert len(declared) >= 16
    for name in declared:
        assert hasattr(L, name), f"{name} declared in include/srhip.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared, "python binding table out of sync with the header"
    experimental = _header_functions("srhip_experimental.h")
    assert experimental == sorted(_lib.EXPERIMENTAL) == ["sr_get_experiment", "sr_set_experiment"]
    for name in experimental:
        assert hasattr(L, name), f"{name} declared in include/srhip_experimental.h but not exported"
    assert not set(experimental) & set(declared)


def test_drop_in_header_has_no_tuning_surface():
    """include/srhip.h is what a reference maintainer binds: no experiment switch, no SRHIP_* environment knob."""
    src = open(os.path.join(ROOT, "include", "srhip.h")).read()
    src = src.replace("SRHIP_H", "")  # the include guard
    assert "sr_set_experiment" not in src and "SRHIP_" not in src


def test_header_constants_match_binding():
    from rusty_sr_amd import _lib
    src = open(os.path.join(ROOT, "include", "srhip.h")).read()
    assert int(re.search(r"#define SR_NUM_PARAMS (\d+)", src).group(1)) == _lib.SR_NUM_PARAMS == oracle.NPARAMS
    assert int(re.search(r"#define SR_HALO (\d+)", src).group(1)) == _lib.SR_HALO == 7
    assert int(re.search(r"#define SR_FACTOR (\d+)", src).group(1)) == _lib.SR_FACTOR == 3
    for name, val in re.findall(r"(SR_E_[A-Z_]+) = (-\d+)", src):
        assert getattr(_lib, name) == int(val)


def test_rsr_decode_matches_oracle_and_roundtrips(params):
    import rusty_sr_amd as r
    for name in r.rsr.BUILTIN:
        blob = open(os.path.join(ROOT, "rusty_sr_amd", "res", name + ".rsr"), "rb").read()
        p = r.rsr.decode(blob)
        np.testing.assert_array_equal(p, params[name])
        assert r.rsr.encode(p) == blob  # bytevec encode::<u32> is the exact inverse (main.rs:213)
        np.testing.assert_array_equal(r.rsr.builtin(name), p)
    assert r.rsr.encode(np.zeros(0, np.float32)) == b"\x00\x00\x00\x00"
    assert r.rsr.decode(b"\x00\x00\x00\x00").size == 0


@pytest.mark.parametrize("blob", [b"", b"\x01", b"\x01\x00\x00\x00",
                                  b"\x01\x00\x00\x00\x08\x00\x00\x00\x00\x00\x00\x00",
                                  b"\x02\x00\x00\x00" + b"\x04\x00\x00\x00" * 2 + b"\x00" * 7])
def test_rsr_decode_rejects_malformed(blob):
    import rusty_sr_amd as r
    with pytest.raises(r.SrError) as e:
        r.rsr.decode(blob)
    assert "ByteVec conversion failed" in str(e.value)  # reference main.rs:138 message


def test_create_validates_before_touching_the_gpu(params):
    import rusty_sr_amd as r
    from rusty_sr_amd import _lib
    p = params["imagenet"]
    with pytest.raises(r.SrError) as e:
        r.Engine(p[:-1])
    assert e.value.status == _lib.SR_E_PARAM_COUNT
    assert "Parameters selected do not have the size required by the neural net" in str(e.value)  # main.rs:162
    with pytest.raises(r.SrError) as e:
        r.Engine(p, factor=5)
    assert e.value.status == _lib.SR_E_FACTOR
    with pytest.raises(r.SrError) as e:  # factor 4 needs sr_net(4)'s 148624 parameters, not the bundled 130459
        r.Engine(p, factor=4)
    assert e.value.status == _lib.SR_E_PARAM_COUNT
    L = _lib.lib()
    assert [L.sr_num_params_factor(f) for f in (1, 2, 3, 4, 5)] == [-1, oracle.num_params(2), oracle.NPARAMS, oracle.num_params(4), -1]
    with pytest.raises(r.SrError) as e:  # parameter-free graphs take exactly zero parameters
        r.Engine(p, graph="bilinear")
    assert e.value.status == _lib.SR_E_PARAM_COUNT
    with pytest.raises(r.SrError) as e:
        r.Engine((), graph="sr_net")
    assert e.value.status == _lib.SR_E_PARAM_COUNT
    with pytest.raises(r.SrError):
        r.sr_net(5)
    assert r.sr_net(4).num_params() == 148624 and r.sr_net(2).num_params() == 117484
    with pytest.raises(NotImplementedError):
        r.sr_net(3, training=(1e-6, False))


@pytest.mark.skipif(gpu_available(), reason="checks the no-GPU failure mode")
def test_no_cpu_fallback(params):
    """Without a HIP device the engine refuses loudly instead of computing on the CPU."""
    import rusty_sr_amd as r
    from rusty_sr_amd import _lib
    with pytest.raises(r.SrError) as e:
        r.Engine(params["imagenet"])
    assert e.value.status == _lib.SR_E_NO_DEVICE
    g = r.sr_net(3)
    inp = r.NodeData.new_blank(r.DataShape(3, [4, 4], 1))
    with pytest.raises(r.SrError):
        g.forward(1, [inp], params["imagenet"])


def test_null_arguments_are_rejected():
    from rusty_sr_amd import _lib
    L = _lib.lib()
    assert L.sr_create(None, None, 0, 3, 0) == _lib.SR_E_INVALID
    assert L.sr_upscale_f32(None, None, 1, 1, 1, None) == _lib.SR_E_INVALID
    assert L.sr_set_profiling(None, 1) == _lib.SR_E_INVALID
    assert L.sr_strerror(_lib.SR_E_HALO).decode().startswith("band halo")
    L.sr_destroy(None)  # no-op


def test_img_to_data_matches_oracle():
    import rusty_sr_amd as r
    rng = np.random.default_rng(1)
    px = rng.integers(0, 256, (5, 7, 4), dtype=np.uint8)
    np.testing.assert_array_equal(r.img_to_data(px), oracle.img_to_data(px))


def test_product_never_touches_the_oracle():
    """rusty_sr_amd (and bench/entry glue outside their checker legs) must not route
    compute through oracle/: the package may not even mention it."""
    pkg = os.path.join(ROOT, "rusty_sr_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(import|from)\s+oracle", text, flags=re.M), f
                assert "sr_oracle" not in text, f


def test_header_is_plain_c(tmp_path):
    """include/srhip.h must bind from C (and therefore from Rust bindgen / cgo / ctypes):
    compile a C99 translation unit that takes the address of every declared function."""
    import subprocess
    names = _header_functions() + _header_functions("srhip_experimental.h")
    src = "#include \"srhip.h\"\n#include \"srhip_experimental.h\"\ntypedef void (*fn)(void);\nfn table[] = {" + ", ".join(f"(fn){n}" for n in names) + "};\nint main(void) { return sizeof(table) ? 0 : 1; }\n"
    c = tmp_path / "abi.c"
    c.write_text(src)
    lib_dir = os.path.join(ROOT, "rusty_sr_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"), str(c),
                           "-L", lib_dir, "-lsrhip", "-Wl,-rpath," + lib_dir, "-Wl,-rpath-link,/opt/rocm/lib",
                           "-o", str(tmp_path / "abi")])


def test_async_asm_results_are_not_read_before_their_wait():
    """The kernels issue a few VMEM instructions from inline asm whose results arrive later (the tile queue's returning
    atomic, the head snapshot, the last stage's pixel prefetch); the compiler cannot know and may read such a register at
    once.  build_lib keeps the device assembly; scripts/check_async_regs.py lints it."""
    import subprocess
    import sys
    from rusty_sr_amd.build import DEVICE_ASM as asm, build_lib
    build_lib()
    if not os.path.exists(asm):  # an object built before this check existed
        build_lib(force=True)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_async_regs.py"), asm], capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "asynchronous asm results checked, 0 read too early" in res.stdout and not res.stdout.startswith("0 ")
    # ... and the two idioms that depend on encoding sizes: the computed jump into the s_waitcnt table, the wait state behind an m0 write
    assert re.search(r"[1-9]\d* wait tables and [1-9]\d* LDS-DMA m0 writes checked, 0 with an unexpected layout", res.stdout), res.stdout


def test_last_stage_of_the_exact_mode_is_on_4x4x1_mfmas():
    """DESIGN.md 4a': the exact mode's last stage runs its 8-row tiles on v_mfma_f32_4x4x1_16B_f32 with CBSZ = 4 (27 channels in 28 slots:
    seven 4-slot groups per k-quad, ABID 0-6 and 8-14), its 4-row tiles on 32x32x2; no other kernel uses the instruction; nothing of it
    spills; and one tile's unrolled body still fits a 64 KB instruction cache with the rest of the loop."""
    from rusty_sr_amd.build import DEVICE_ASM as asm, build_lib
    build_lib()
    text = open(asm).read()
    kernels = dict(re.findall(r"^(_Z22conv_stage_pipe_kernel\w+):.*?\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M))
    finals = {k: v for k, v in kernels.items() if re.search(r"ILi3ELi3ELb1ELb[01]ELb[01]ELi0ELi3E", k)}   # <3, 3, FINAL, *, *, PREC 0, FACTOR 3>
    assert len(finals) == 2, sorted(kernels)
    for name, body in finals.items():
        quad = re.findall(r"v_mfma_f32_4x4x1_16b_f32 .*", body)
        assert len(quad) == (864 + 27) * 7, (name, len(quad))            # 27 taps x 32 channels + the residual's 27 k-steps, seven groups each
        assert all("cbsz:4" in q for q in quad)
        abids = {int(m) for q in quad for m in re.findall(r"abid:(\d+)", q)} | ({0} if any("abid" not in q for q in quad) else set())
        assert abids == set(range(0, 7)) | set(range(8, 15)), abids
        assert len(re.findall(r"v_mfma_f32_32x32x2_f32", body)) == 27 * 16 + 9 * 2   # the 4-row tile body: 27 taps x 16 + the residual's 18
        assert "scratch_" not in body and re.search(r"\.amdhsa_private_segment_fixed_size 0", body)
        assert len(quad) * 8 < 56 * 1024
    others = [k for k, v in kernels.items() if k not in finals and "v_mfma_f32_4x4x1" in v and not re.search(r"Lb1ELb[01]ELb[01]ELi0ELi[24]E", k)]
    assert not others, others   # (factor 2 / 4 instances of the same stage use it too: 3 and 8 + 5 groups)


def test_split_half_epilogues_are_lane_local_and_scalar():
    """DESIGN.md 4b (round 6): every split-half kernel computes the transposed tile (weights as the MFMA's A operand), so its epilogue packs
    channel pairs of ONE pixel -- no DPP lane exchange, no v_perm_b32 in the stage kernels' epilogues -- and spells its pair arithmetic as
    scalar instructions (packed f32 instructions beside the partner wave's MFMAs cost more than two scalar ones; the file is built with
    -fno-slp-vectorize so that they stay scalar).  The exact mode keeps its packed epilogue.  Nothing spills."""
    from rusty_sr_amd.build import DEVICE_ASM as asm, build_lib
    build_lib()
    text = open(asm).read()
    kernels = dict(re.findall(r"^(_Z\w+):.*?\n(.*?)\.end_amdhsa_kernel", text, re.S | re.M))
    split = {k: v for k, v in kernels.items() if "conv0_split_kernel" in k or re.search(r"conv_stage(_pipe)?_kernelI.*Lb[01]ELb[01]ELb[01]ELi1ELi[234]E", k)}
    exact = {k: v for k, v in kernels.items() if re.search(r"conv_stage(_pipe)?_kernelI.*Lb[01]ELb[01]ELb[01]ELi0ELi3E", k)}
    assert len(split) >= 4 + 6 and len(exact) >= 8, (len(split), len(exact))
    for name, body in split.items():
        assert not re.search(r"v_pk_(fma|mul|add)_f32", body), name
        assert "_dpp" not in body and "row_shl" not in body and "quad_perm" not in body, name
        assert re.search(r"\.amdhsa_private_segment_fixed_size 0", body) and "scratch_" not in body, name
    # the last stage at factor 3: the lane's sub-pixels of an output row leave as runs (RGBA8: 3 + 2 or 1 + 3 dwords per lane and tile row)
    finals_u8 = [v for k, v in split.items() if re.search(r"conv_stage_pipe_kernelILi3ELi3ELb1ELb1ELb1ELi1ELi3E", k)]
    assert len(finals_u8) == 1 and "global_store_dwordx3" in finals_u8[0] and "global_store_dwordx2" in finals_u8[0]
    assert any(re.search(r"v_pk_(fma|mul|add)_f32", body) for body in exact.values())


def test_parameter_free_kernels_keep_their_loads_global():
    """What made the round-5 kernels of sr_aux.hip fast is visible in their assembly, and easy to lose in an edit: the pixel
    windows must be read with global_load (an integer cast back to a pointer turns them into flat_load, and then every wait the
    compiler places is vmcnt(0): each pass sits out the previous pass's write acknowledgements), nothing may spill, and the
    prefetched window of the aligned u8 bilinear kernel must be consumed behind its three stores with `vmcnt(3)`."""
    import subprocess
    from rusty_sr_amd.build import CSRC, FLAGS
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    res = subprocess.run([hipcc, *FLAGS, "--cuda-device-only", "-S", "-x", "hip", os.path.join(CSRC, "sr_aux.hip"), "-o", "-"],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.split("\n")
    bodies, cur = {}, None
    for l in lines:
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur = m.group(1)
            bodies[cur] = []
        elif l.startswith(".Lfunc_end"):
            cur = None
        elif cur:
            bodies[cur].append(l.split(";")[0].strip() if not l.strip().startswith(";;#ASM") else l.strip())
    direct = {k: v for k, v in bodies.items() if "bilinear_u8_kernel" in k or "downsample_tile_kernel" in k}
    assert len(direct) == 7, sorted(bodies)   # 4 u8 bilinear (3 / 4 channels x aligned or not) + 3 downsample (u8 3 / 4 channels, f32)
    for name, body in direct.items():
        assert not [t for t in body if t.startswith(("flat_load", "flat_store", "scratch_"))], name
        assert any(t.startswith("global_load_dword") for t in body), name
    assert all(int(m) == 0 for m in re.findall(r"\.vgpr_spill_count:\s*(\d+)", res.stdout))
    aligned3 = [v for k, v in direct.items() if "bilinear_u8_kernelILi3ELb1E" in k]
    assert len(aligned3) == 1
    body = [t for t in aligned3[0] if t]
    waits = [body[i - 1] for i, t in enumerate(body) if t.startswith(";;#ASMSTART") and body[i - 1].startswith("s_waitcnt")]
    assert "s_waitcnt vmcnt(3)" in waits, waits
