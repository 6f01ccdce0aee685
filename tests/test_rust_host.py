"""rust_host/ is the Rust host the north star asks for; no Rust toolchain exists in this image,
so it cannot be compiled here.  These checks keep it honest anyway: its extern block names
exactly the C ABI (same symbols, same arity as include/srhip.h and the ctypes table), its
constants equal the header's, and its user-visible strings equal the C++ twin's, which the
GPU tests drive end to end."""
import os
import re

from conftest import ROOT
from rusty_sr_amd._lib import SYMBOLS

RS = os.path.join(ROOT, "rust_host", "src")


def _read(*p):
    with open(os.path.join(*p)) as f:
        return f.read()


def _rust_externs():
    src = _read(RS, "srhip.rs")
    block = src[src.index('extern "C" {'):]
    block = block[:block.index("\n}\n")]
    out = {}
    for m in re.finditer(r"pub fn (sr_\w+)\(([^)]*)\)\s*(->\s*[^;]+)?;", block, re.S):
        args = [a for a in m.group(2).split(",") if a.strip()]
        out[m.group(1)] = (len(args), (m.group(3) or "").replace("->", "").strip())
    return out


def test_rust_extern_block_is_the_c_abi():
    ext = _rust_externs()
    assert set(ext) == set(SYMBOLS), set(ext) ^ set(SYMBOLS)
    for name, (nargs, ret) in ext.items():
        res, args = SYMBOLS[name]
        assert nargs == len(args), name
        assert (ret == "") == (res is None), name
    header = re.sub(r"/\*.*?\*/", "", _read(ROOT, "include", "srhip.h"), flags=re.S)
    for name, (nargs, _) in ext.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, header, re.S)
        assert m, name
        assert len([a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"]) == nargs, name


def test_rust_constants_match_header():
    src, header = _read(RS, "srhip.rs"), _read(ROOT, "include", "srhip.h")
    consts = dict(re.findall(r"pub const (SR_\w+): c_int = (-?\d+);", src))
    assert {"SR_OK", "SR_E_HIP", "SR_GRAPH_SR_NET", "SR_GRAPH_BILINEAR", "SR_GRAPH_DOWNSAMPLE", "SR_PRECISION_F32",
            "SR_PRECISION_SPLIT_F16", "SR_FACTOR"} <= set(consts)
    for name, val in consts.items():
        m = re.search(r"\b%s\s*=\s*(-?\d+)" % name, header) or re.search(r"#define\s+%s\s+(-?\d+)" % name, header)
        assert m and int(m.group(1)) == int(val), name


def test_rust_host_speaks_like_the_cpp_host():
    rs, cpp = _read(RS, "main.rs"), _read(ROOT, "rusty_sr_amd", "host", "main.cpp")
    for text in ("Upscaling using custom neural net parameters...", "Downsampling using average pooling of linear RGB values...",
                 "Upscaling using imagenet neural net parameters...", "Upscaling using linear loss imagenet neural net parameters...",
                 "Upscaling using anime neural net parameters...", "Upscaling using bilinear interpolation...",
                 " Writing file...", " Done", "Error opening parameter file", "ByteVec conversion failed",
                 "Error opening input image file.", "Could not write output file", "Rusty SR v0.1.1",
                 "cannot be used with '--parameters <PARAMETERS>'", "The following required arguments were not provided:",
                 "isn't a valid value for '--parameters <PARAMETERS>'"):
        assert text in rs and text in cpp, text
    for opt in ("--parameters", "--custom", "--downsample", "--device", "--precision", "--timing", '"-p"', '"-c"', '"-d"'):
        assert opt in rs and opt in cpp, opt
    for blob in ("imagenet.rsr", "imagenetlinear.rsr", "anime.rsr"):
        assert f'include_bytes!("../../rusty_sr_amd/res/{blob}")' in rs
        assert os.path.exists(os.path.join(ROOT, "rusty_sr_amd", "res", blob))


def _extern_decls(text):
    """The declarations of the first `extern "C" { ... }` block of `text`, comments and layout removed."""
    block = text[text.index('extern "C" {') + len('extern "C" {'):]
    block = block[:block.index("\n}")]
    block = re.sub(r"//[^\n]*", "", block)
    decls = [re.sub(r"\s+", " ", d).strip() for d in block.split(";")]
    return [d for d in decls if d]


def test_integration_md_shows_the_same_extern_block():
    """The Rust block INTEGRATION.md shows a maintainer must parse and must BE rust_host/src/srhip.rs's block
    (round 2 shipped it with two declarations pasted into the middle of a third)."""
    md = _read(ROOT, "INTEGRATION.md")
    rust_block = md[md.index("```rust") + len("```rust"):]
    rust_block = rust_block[:rust_block.index("```")]
    shown, real = _extern_decls(rust_block), _extern_decls(_read(RS, "srhip.rs"))
    for d in shown:  # every declaration is one well-formed `pub fn name(args) [-> ret]`
        assert re.fullmatch(r"pub fn sr_\w+\([^()]*\)( -> [\w*: ]+)?", d), d
    assert shown == real
