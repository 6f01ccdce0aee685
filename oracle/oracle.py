"""ctypes loader for oracle/libsr_oracle.so (see sr_oracle.c header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
NPARAMS = 130459

# name -> (offset, length, shape [O][KH][KW][I] or [C]); SURVEY.md 8(a) row W,
# op insertion order of reference src/network.rs:33-72.
SEGMENTS = {
    "conv0": (0, 2400, (32, 5, 5, 3)),
    "f_bias": (2400, 32, (32,)),
    "f_activ": (2432, 32, (32,)),
    "expand_bias": (2464, 27, (27,)),
    "l1_bias": (2491, 32, (32,)),
    "l2_bias": (2523, 32, (32,)),
    "l3_bias": (2555, 32, (32,)),
    "l1_activ": (2587, 32, (32,)),
    "l2_activ": (2619, 32, (32,)),
    "l3_activ": (2651, 32, (32,)),
    "conv1": (2683, 25600, (32, 5, 5, 32)),
    "conv2": (28283, 25600, (32, 5, 5, 32)),
    "conv3": (53883, 25600, (32, 5, 5, 32)),
    "conv5": (79483, 9216, (32, 3, 3, 32)),
    "conv6": (88699, 9216, (32, 3, 3, 32)),
    "conv7": (97915, 7776, (27, 3, 3, 32)),
    "conv8": (105691, 9216, (32, 3, 3, 32)),
    "conv9": (114907, 7776, (27, 3, 3, 32)),
    "conv10": (122683, 7776, (27, 3, 3, 32)),
}

_lib = None


def _cpu_tag():
    import hashlib
    try:
        with open("/proc/cpuinfo") as f:
            lines = [l for l in f if l.startswith(("model name", "flags"))][:2]
    except OSError:
        lines = []
    return hashlib.sha1("".join(lines).encode()).hexdigest()[:10]


def build(native=False, force=False):
    """Compile the oracle with gcc.  native=True builds a -march=native copy
    (used for the cpu_baseline timing on the GPU box's host CPU)."""
    # the -march=native copy is only valid on the CPU it was built on: key its name on the host's CPU model so a copy
    # built in one container is never loaded on another machine (the GPU box rebuilds its own on first use)
    name = f"libsr_oracle_native_{_cpu_tag()}.so" if native else "libsr_oracle.so"
    out = os.path.join(_HERE, name)
    src = os.path.join(_HERE, "sr_oracle.c")
    if not force and os.path.exists(out) and os.path.getmtime(out) >= os.path.getmtime(src):
        return out
    march = "native" if native else "x86-64-v3"
    flags = ["-O3", f"-march={march}", "-fopenmp", "-fPIC", "-std=c11", "-ffp-contract=off"]
    tmp = out + f".{os.getpid()}.tmp"
    objs = []
    for tag, extra in (("f32", []), ("f64", ["-DSR_REAL_DOUBLE"])):
        obj = f"{tmp}.{tag}.o"
        subprocess.check_call(["gcc", *flags, *extra, "-c", src, "-o", obj])
        objs.append(obj)
    subprocess.check_call(["gcc", "-shared", "-fopenmp", "-o", tmp, *objs, "-lm"])
    for o in objs:
        os.remove(o)
    os.replace(tmp, out)
    return out


def lib(native=False):
    global _lib
    if native:
        return _bind(ctypes.CDLL(build(native=True)))
    if _lib is None:
        path = os.path.join(_HERE, "libsr_oracle.so")
        try:
            if not os.path.exists(path):
                raise OSError("missing")
            _lib = _bind(ctypes.CDLL(path))
        except OSError:
            _lib = _bind(ctypes.CDLL(build(force=True)))
    return _lib


def _bind(L):
    c = ctypes
    fp, dp, u8p = c.POINTER(c.c_float), c.POINTER(c.c_double), c.POINTER(c.c_uint8)
    L.sr_oracle_rsr_decode.restype = c.c_long
    L.sr_oracle_rsr_decode.argtypes = [u8p, c.c_size_t, fp, c.c_size_t]
    for suf, rp in (("", fp), ("_f64", dp)):
        f = getattr(L, "sr_oracle_forward" + suf)
        f.restype = c.c_int
        f.argtypes = [fp, c.c_size_t, rp, c.c_int, c.c_int, c.c_int, rp, rp]
        ff = getattr(L, "sr_oracle_forward_factor" + suf)
        ff.restype = c.c_int
        ff.argtypes = [fp, c.c_size_t, c.c_int, rp, c.c_int, c.c_int, c.c_int, rp]
        getattr(L, "sr_oracle_num_params_factor" + suf).restype = c.c_int
        getattr(L, "sr_oracle_num_params_factor" + suf).argtypes = [c.c_int]
        for nm in ("sr_oracle_bilinear", "sr_oracle_downsample"):
            q = getattr(L, nm + suf)
            q.restype = c.c_int
            q.argtypes = [rp, c.c_int, c.c_int, c.c_int, rp]
        g = getattr(L, "sr_oracle_img_to_data" + suf)
        g.restype = None
        g.argtypes = [u8p, c.c_int, c.c_size_t, rp]
        h = getattr(L, "sr_oracle_data_to_rgba8" + suf)
        h.restype = None
        h.argtypes = [rp, c.c_size_t, u8p]
    return L


def _ptr(a, ct):
    return a.ctypes.data_as(ctypes.POINTER(ct))


def rsr_decode(blob: bytes) -> np.ndarray:
    """bytevec `<Vec<f32>>::decode::<u32>` (reference main.rs:146)."""
    L = lib()
    buf = np.frombuffer(blob, dtype=np.uint8)
    n = L.sr_oracle_rsr_decode(_ptr(buf, ctypes.c_uint8), len(blob), None, 0)
    if n < 0:
        raise ValueError(f"ByteVec conversion failed (code {n})")
    out = np.empty(n, dtype=np.float32)
    L.sr_oracle_rsr_decode(_ptr(buf, ctypes.c_uint8), len(blob), _ptr(out, ctypes.c_float), n)
    return out


def _forward(params, x, f64, taps, native):
    L = lib(native)
    dt, ct, suf = (np.float64, ctypes.c_double, "_f64") if f64 else (np.float32, ctypes.c_float, "")
    params = np.ascontiguousarray(params, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=dt)
    if x.ndim == 3:
        x = x[None]
    n, H, W, C = x.shape
    assert C == 3
    out = np.empty((n, 3 * H, 3 * W, 3), dtype=dt)
    tp = np.empty(H * W * (4 * 32 + 27), dtype=dt) if taps else None
    rc = getattr(L, "sr_oracle_forward" + suf)(
        _ptr(params, ctypes.c_float), params.size, _ptr(x, ct), n, H, W, _ptr(out, ct),
        _ptr(tp, ct) if taps else None)
    if rc != 0:
        raise ValueError(
            "Parameters selected do not have the size required by the neural net" if rc == -1
            else f"oracle forward failed ({rc})")
    if not taps:
        return out
    npx = H * W
    d = {k: tp[i * npx * 32:(i + 1) * npx * 32].reshape(H, W, 32)
         for i, k in enumerate(("f", "l1", "l2", "l3"))}
    d["e"] = tp[4 * npx * 32:].reshape(H, W, 27)
    return out, d


def forward(params, x, f64=False, native=False):
    """graph.forward (reference main.rs:171): x (n,H,W,3) or (H,W,3) in [0,1] ->
    (n,3H,3W,3) pre-quantisation."""
    return _forward(params, x, f64, False, native)


def num_params(factor=3):
    return lib().sr_oracle_num_params_factor(factor)


def forward_factor(params, x, factor, f64=False):
    """sr_net(factor, None) for factor 1..4 (UNPINNED for factor != 3; see sr_oracle.c)."""
    L = lib()
    dt, ct, suf = (np.float64, ctypes.c_double, "_f64") if f64 else (np.float32, ctypes.c_float, "")
    params = np.ascontiguousarray(params, dtype=np.float32)
    x = np.ascontiguousarray(x, dtype=dt)
    if x.ndim == 3:
        x = x[None]
    n, H, W, _ = x.shape
    out = np.empty((n, factor * H, factor * W, 3), dtype=dt)
    rc = getattr(L, "sr_oracle_forward_factor" + suf)(_ptr(params, ctypes.c_float), params.size, factor, _ptr(x, ct),
                                                      n, H, W, _ptr(out, ct))
    if rc != 0:
        raise ValueError(f"forward_factor failed ({rc})")
    return out


def forward_taps(params, x, f64=False):
    return _forward(params, x, f64, True, False)


def _aux(name, x, f64, out_shape):
    L = lib()
    dt, ct, suf = (np.float64, ctypes.c_double, "_f64") if f64 else (np.float32, ctypes.c_float, "")
    x = np.ascontiguousarray(x, dtype=dt)
    if x.ndim == 3:
        x = x[None]
    n, H, W, _ = x.shape
    out = np.empty((n,) + out_shape(H, W) + (3,), dtype=dt)
    rc = getattr(L, name + suf)(_ptr(x, ct), n, H, W, _ptr(out, ct))
    if rc != 0:
        raise ValueError(f"{name} failed ({rc})")
    return out


def bilinear(x, f64=False):
    """bilinear_net(3) (reference network.rs:111-123): sRGB->linear, x3 bilinear, ->sRGB."""
    return _aux("sr_oracle_bilinear", x, f64, lambda H, W: (3 * H, 3 * W))


def downsample(x, f64=False):
    """downsample_net(3) (reference network.rs:125-138): sRGB->linear, 3x3 mean pool, ->sRGB."""
    return _aux("sr_oracle_downsample", x, f64, lambda H, W: (H // 3, W // 3))


def img_to_data(px: np.ndarray) -> np.ndarray:
    """u8 (...,H,W,3|4) -> f32 (...,H,W,3) = u8/255, alpha dropped (main.rs:170)."""
    px = np.ascontiguousarray(px, dtype=np.uint8)
    out = np.empty(px.shape[:-1] + (3,), dtype=np.float32)
    lib().sr_oracle_img_to_data(_ptr(px, ctypes.c_uint8), px.shape[-1], out.size // 3,
                                _ptr(out, ctypes.c_float))
    return out


def data_to_rgba8(v: np.ndarray) -> np.ndarray:
    """f32 (...,3) -> u8 (...,4): clamp(floor(255v+0.5)), alpha 255 (main.rs:175)."""
    v = np.ascontiguousarray(v, dtype=np.float32)
    out = np.empty(v.shape[:-1] + (4,), dtype=np.uint8)
    lib().sr_oracle_data_to_rgba8(_ptr(v, ctypes.c_float), v.size // 3, _ptr(out, ctypes.c_uint8))
    return out


def upscale_rgba8(params, px):
    """The whole of upscale() between image::open and .save (main.rs:168-175)."""
    return data_to_rgba8(forward(params, img_to_data(px)))
