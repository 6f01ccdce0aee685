"""`.rsr` parameter files = bytevec 0.2.0 `Vec<f32>` blobs with u32 size
prefixes (reference main.rs:138,146,149,152 decode; main.rs:213 encode).
Thin host-side wrappers over libsrhip's sr_rsr_decode / sr_rsr_encode."""
import ctypes as C
import os

import numpy as np

from . import _lib

RES_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "res")
# reference main.rs:26-28 include_bytes!("res/*.rsr")
BUILTIN = ("imagenet", "imagenetlinear", "anime")


def decode(blob: bytes) -> np.ndarray:
    """`<Vec<f32>>::decode::<u32>(&data).expect("ByteVec conversion failed")`."""
    L = _lib.lib()
    buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob) if len(blob) else (C.c_uint8 * 1)()
    n = C.c_size_t(0)
    _lib.check(L.sr_rsr_decode(buf, len(blob), None, 0, C.byref(n)))
    out = np.empty(n.value, dtype=np.float32)
    _lib.check(L.sr_rsr_decode(buf, len(blob), out.ctypes.data_as(C.POINTER(C.c_float)), n.value, C.byref(n)))
    return out


def encode(params) -> bytes:
    """`data.params.encode::<u32>()` (reference main.rs:213)."""
    L = _lib.lib()
    p = np.ascontiguousarray(params, dtype=np.float32)
    ln = C.c_size_t(0)
    fp = p.ctypes.data_as(C.POINTER(C.c_float))
    _lib.check(L.sr_rsr_encode(fp, p.size, None, 0, C.byref(ln)))
    out = (C.c_uint8 * ln.value)()
    _lib.check(L.sr_rsr_encode(fp, p.size, out, ln.value, C.byref(ln)))
    return bytes(out)


def builtin(name: str = "imagenet") -> np.ndarray:
    """The parameter sets the reference embeds (main.rs:26-28, 144-152)."""
    if name not in BUILTIN:
        raise ValueError(f"unknown built-in parameters {name!r}; expected one of {BUILTIN}")
    with open(os.path.join(RES_DIR, name + ".rsr"), "rb") as f:
        return decode(f.read())
