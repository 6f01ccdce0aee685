"""Host-side mirror of the reference's engine seam for the upscale path.

Reference call site (src/main.rs:161-171):
    assert_eq!(params.len(), graph.num_params(), ...);
    let mut input = NodeData::new_blank(DataShape::new(CHANNELS, &[W, H], 1));
    img_to_data(&mut input.values, &input_image);
    let output = graph.forward(1, vec![input], &params).remove(0);
Here `sr_net(factor)` returns a Graph whose forward() runs the hand-written
gfx950 kernels through libsrhip's C ABI.  torch is used only for device memory
and streams."""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib

FACTOR = 3    # reference main.rs:31
CHANNELS = 3  # reference network.rs:13


@dataclass
class DataShape:
    """alumina DataShape::new(channels, &[W, H], n) (reference main.rs:168)."""
    channels: int
    spatial_dimensions: Sequence[int]  # [W, H]
    n: int = 1

    def flat_size_single(self):
        w, h = self.spatial_dimensions
        return self.channels * w * h


@dataclass
class NodeData:
    """alumina NodeData { shape, values } -- values are n x [y][x][c] f32."""
    shape: DataShape
    values: np.ndarray = field(default=None)

    @staticmethod
    def new_blank(shape: DataShape) -> "NodeData":
        return NodeData(shape, np.zeros(shape.n * shape.flat_size_single(), dtype=np.float32))


def img_to_data(pixels: np.ndarray) -> np.ndarray:
    """alumina supplier::imagefolder::img_to_data (reference main.rs:170):
    u8 RGB(A) -> f32 = u8/255, alpha dropped.  Host-side convenience for the
    f32 entry points; the rgba8 entry points fuse this into the first kernel."""
    px = np.asarray(pixels)
    if px.dtype != np.uint8 or px.shape[-1] not in (3, 4):
        raise ValueError("expected u8 pixels with 3 or 4 channels")
    return px[..., :3].astype(np.float32) / np.float32(255.0)


class Engine:
    """One sr_ctx (= one GPU, one parameter set)."""

    PRECISIONS = {"f32": _lib.SR_PRECISION_F32, "split_f16": _lib.SR_PRECISION_SPLIT_F16}

    GRAPHS = {"sr_net": _lib.SR_GRAPH_SR_NET, "bilinear": _lib.SR_GRAPH_BILINEAR, "downsample": _lib.SR_GRAPH_DOWNSAMPLE}

    def __init__(self, params=(), device: int = 0, factor: int = FACTOR, precision: str = "f32", graph: str = "sr_net"):
        L = _lib.lib()
        p = np.ascontiguousarray(params, dtype=np.float32)
        self._ctx = C.c_void_p()
        self.graph = graph
        self.factor = factor
        _lib.check(L.sr_create_graph(C.byref(self._ctx), self.GRAPHS[graph],
                                     p.ctypes.data_as(C.POINTER(C.c_float)) if p.size else None, p.size, factor, device))
        self.device = device
        self._L = L
        self.set_precision(precision)

    def set_precision(self, precision: str):
        """"f32": exact-f32 MFMA (default).  "split_f16": hi/lo half pairs, 3 f16 MFMAs per product."""
        _lib.check(self._L.sr_set_precision(self._ctx, self.PRECISIONS[precision]))
        self.precision = precision

    def check_domain(self):
        """sr_check_domain: after the stream of earlier *_dev calls has been synchronised, raises SrError(SR_E_DOMAIN) if one of them
        left the domain of "split_f16" (a non-finite value, or one of 65504 and beyond); the host-pointer calls recompute in f32 themselves."""
        _lib.check(self._L.sr_check_domain(self._ctx))

    def _out_hw(self, h, w):
        return (h // 3, w // 3) if self.graph == "downsample" else (self.factor * h, self.factor * w)

    def close(self):
        if getattr(self, "_ctx", None):
            self._L.sr_destroy(self._ctx)
            self._ctx = None

    __del__ = close

    # ---- host-memory entry points (numpy in, numpy out) --------------------
    def upscale_f32(self, x: np.ndarray) -> np.ndarray:
        """(n,H,W,3) or (H,W,3) f32 in [0,1] -> (n,3H,3W,3) f32 pre-quantisation."""
        x = np.ascontiguousarray(x, dtype=np.float32)
        squeeze = x.ndim == 3
        if squeeze:
            x = x[None]
        n, h, w, c = x.shape
        if c != 3:
            raise ValueError("expected 3 channels")
        oh, ow = self._out_hw(h, w)
        out = np.empty((n, oh, ow, 3), dtype=np.float32)
        fp = C.POINTER(C.c_float)
        _lib.check(self._L.sr_upscale_f32(self._ctx, x.ctypes.data_as(fp), n, h, w, out.ctypes.data_as(fp)), self._ctx)
        return out[0] if squeeze else out

    def upscale_rgba8(self, px: np.ndarray, out: np.ndarray = None) -> np.ndarray:
        """(n,H,W,3|4) or (H,W,3|4) u8 -> (n,3H,3W,4) u8 RGBA (alpha 255).  `out` may be a
        page-locked array from host_alloc (copies then overlap the kernels at PCIe rate)."""
        px = np.ascontiguousarray(px, dtype=np.uint8)
        squeeze = px.ndim == 3
        if squeeze:
            px = px[None]
        n, h, w, c = px.shape
        oh, ow = self._out_hw(h, w)
        if out is None:
            out = np.empty((n, oh, ow, 4), dtype=np.uint8)
        elif out.dtype != np.uint8 or out.size != n * oh * ow * 4 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous u8 array of n*oh*ow*4 elements")
        else:
            out = out.reshape(n, oh, ow, 4)
        u8p = C.POINTER(C.c_uint8)
        _lib.check(self._L.sr_upscale_rgba8(self._ctx, px.ctypes.data_as(u8p), c, n, h, w, out.ctypes.data_as(u8p)), self._ctx)
        return out[0] if squeeze else out

    def reserve(self, n: int, h: int, w: int, io: str = "rgba8", channels: int = 3):
        """Allocate and warm everything upscale_rgba8 / upscale_f32 of that shape needs (sr_reserve_*): optional."""
        if io == "rgba8":
            _lib.check(self._L.sr_reserve_rgba8(self._ctx, channels, n, h, w), self._ctx)
        elif io == "f32":
            _lib.check(self._L.sr_reserve_f32(self._ctx, n, h, w), self._ctx)
        else:
            raise ValueError("io must be 'rgba8' or 'f32'")

    # ---- device-memory entry points (torch tensors on this GPU) ------------
    @staticmethod
    def _stream_ptr(stream=None, device=None):
        """The stream to launch on: the caller's, else torch's current stream OF THE TENSOR'S DEVICE (not of whatever device
        happens to be current in this thread: a process that drives several GPUs may be elsewhere)."""
        import torch
        s = stream if stream is not None else torch.cuda.current_stream(device)
        return C.c_void_p(s.cuda_stream)

    def upscale_f32_dev(self, x, out=None, stream=None):
        import torch
        assert x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.shape[-1] == 3
        n, h, w, _ = x.shape
        if out is None:
            out = torch.empty((n,) + self._out_hw(h, w) + (3,), dtype=torch.float32, device=x.device)
        _lib.check(self._L.sr_upscale_f32_dev(self._ctx, C.c_void_p(x.data_ptr()), n, h, w,
                                              C.c_void_p(out.data_ptr()), self._stream_ptr(stream, x.device)), self._ctx)
        return out

    def upscale_rgba8_dev(self, px, out=None, stream=None):
        import torch
        assert px.is_cuda and px.dtype == torch.uint8 and px.is_contiguous() and px.dim() == 4
        n, h, w, c = px.shape
        if out is None:
            out = torch.empty((n,) + self._out_hw(h, w) + (4,), dtype=torch.uint8, device=px.device)
        _lib.check(self._L.sr_upscale_rgba8_dev(self._ctx, C.c_void_p(px.data_ptr()), c, n, h, w,
                                                C.c_void_p(out.data_ptr()), self._stream_ptr(stream, px.device)), self._ctx)
        return out

    def upscale_band_f32_dev(self, x_ext, halo_top, halo_bot, out=None, stream=None):
        """x_ext: (h_ext,W,3) f32 rows = halo_top + band + halo_bot -> (3*band,3W,3)."""
        import torch
        assert x_ext.is_cuda and x_ext.dtype == torch.float32 and x_ext.is_contiguous() and x_ext.dim() == 3
        h_ext, w, _ = x_ext.shape
        hb = h_ext - halo_top - halo_bot
        if hb <= 0 or halo_top < 0 or halo_bot < 0:
            raise _lib.SrError(_lib.SR_E_INVALID)
        if out is None:
            out = torch.empty((self.factor * hb, self.factor * w, 3), dtype=torch.float32, device=x_ext.device)
        _lib.check(self._L.sr_upscale_band_f32_dev(self._ctx, C.c_void_p(x_ext.data_ptr()), h_ext, w, halo_top,
                                                   halo_bot, C.c_void_p(out.data_ptr()), self._stream_ptr(stream, x_ext.device)),
                   self._ctx)
        return out

    def upscale_band_rgba8_dev(self, px_ext, halo_top, halo_bot, out=None, stream=None):
        import torch
        assert px_ext.is_cuda and px_ext.dtype == torch.uint8 and px_ext.is_contiguous() and px_ext.dim() == 3
        h_ext, w, c = px_ext.shape
        hb = h_ext - halo_top - halo_bot
        if hb <= 0 or halo_top < 0 or halo_bot < 0:
            raise _lib.SrError(_lib.SR_E_INVALID)
        if out is None:
            out = torch.empty((self.factor * hb, self.factor * w, 4), dtype=torch.uint8, device=px_ext.device)
        _lib.check(self._L.sr_upscale_band_rgba8_dev(self._ctx, C.c_void_p(px_ext.data_ptr()), c, h_ext, w,
                                                     halo_top, halo_bot, C.c_void_p(out.data_ptr()),
                                                     self._stream_ptr(stream, px_ext.device)), self._ctx)
        return out

    # ---- sharded image: RCCL communicator inside libsrhip (include/srhip.h sr_comm_*) ----------
    @staticmethod
    def comm_unique_id() -> bytes:
        """128 opaque bytes from rank 0 (ncclGetUniqueId) that every rank passes to comm_init_rank."""
        buf = (C.c_uint8 * _lib.SR_COMM_ID_BYTES)()
        _lib.check(_lib.lib().sr_comm_unique_id(buf, _lib.SR_COMM_ID_BYTES))
        return bytes(buf)

    def comm_init_rank(self, uid: bytes, rank: int, nranks: int):
        """Join the band communicator as `rank` of `nranks` (collective; one process per GPU)."""
        buf = (C.c_uint8 * _lib.SR_COMM_ID_BYTES).from_buffer_copy(uid) if nranks > 1 else None
        _lib.check(self._L.sr_comm_init_rank(self._ctx, buf, _lib.SR_COMM_ID_BYTES if nranks > 1 else 0, rank, nranks), self._ctx)

    def comm_rank(self):
        r, n = C.c_int(), C.c_int()
        _lib.check(self._L.sr_comm_rank(self._ctx, C.byref(r), C.byref(n)))
        return r.value, n.value

    def last_comm_ms(self) -> float:
        v = C.c_double()
        _lib.check(self._L.sr_last_comm_ms(self._ctx, C.byref(v)))
        return v.value

    def last_comm_exposed_ms(self) -> float:
        """What of the last sharded call's halo exchange the band's stream had to wait for (sr_last_comm_exposed_ms)."""
        v = C.c_double()
        _lib.check(self._L.sr_last_comm_exposed_ms(self._ctx, C.byref(v)))
        return v.value

    def upscale_sharded_dev(self, band, out=None, stream=None):
        """This rank's band (rows, W, 3 f32 | 3-4 u8) of an image sharded in rank order: halo exchange over the
        context's RCCL communicator + band pass, asynchronous on the stream -> (3 rows, 3 W, 3 f32 | 4 u8)."""
        import torch
        assert band.is_cuda and band.is_contiguous() and band.dim() == 3
        hb, w, c = band.shape
        u8 = band.dtype == torch.uint8
        assert u8 or (band.dtype == torch.float32 and c == 3)
        if out is None:
            out = torch.empty((self.factor * hb, self.factor * w, 4 if u8 else 3), dtype=band.dtype, device=band.device)
        if u8:
            st = self._L.sr_upscale_sharded_rgba8_dev(self._ctx, C.c_void_p(band.data_ptr()), c, hb, w, C.c_void_p(out.data_ptr()),
                                                      self._stream_ptr(stream, band.device))
        else:
            st = self._L.sr_upscale_sharded_f32_dev(self._ctx, C.c_void_p(band.data_ptr()), hb, w, C.c_void_p(out.data_ptr()),
                                                    self._stream_ptr(stream, band.device))
        _lib.check(st, self._ctx)
        return out

    # ---- introspection ------------------------------------------------------
    def read_feature(self, which: int, h: int, w: int) -> np.ndarray:
        """Post-activation node data of the last call: 0..3 = f, l1, l2, l3."""
        out = np.empty((h, w, 32), dtype=np.float32)
        _lib.check(self._L.sr_read_feature(self._ctx, which, out.ctypes.data_as(C.POINTER(C.c_float)), out.size), self._ctx)
        return out

    def set_pipeline(self, on: bool):
        """Host-pointer entry points: chunked upload / compute / download overlap (default on);
        results do not depend on it."""
        _lib.check(self._L.sr_set_pipeline(self._ctx, 1 if on else 0))

    def set_experiment(self, key: str, value: str = ""):
        """sr_set_experiment: "th" / "pipe" / "bw" A/B switches (results do not depend on them)."""
        _lib.check(self._L.sr_set_experiment(self._ctx, key.encode(), value.encode()))

    def get_experiment(self, key: str) -> str:
        """sr_get_experiment: what a switch has learned ("forktune": one line per shape the fork tuner has met)."""
        buf = C.create_string_buffer(4096)
        _lib.check(self._L.sr_get_experiment(self._ctx, key.encode(), buf, len(buf)))
        return buf.value.decode()

    def set_profiling(self, on: bool):
        _lib.check(self._L.sr_set_profiling(self._ctx, int(on)))

    def last_timing(self):
        tot, h2d, d2h = C.c_double(), C.c_double(), C.c_double()
        st = (C.c_double * 5)()
        _lib.check(self._L.sr_last_timing(self._ctx, C.byref(tot), st, C.byref(h2d), C.byref(d2h)))
        return {"total_ms": tot.value, "stage_ms": list(st), "h2d_ms": h2d.value, "d2h_ms": d2h.value}

    def device_info(self):
        name = C.create_string_buffer(128)
        cus, mhz = C.c_int(), C.c_int()
        _lib.check(self._L.sr_device_info(self._ctx, name, 128, C.byref(cus), C.byref(mhz)))
        return {"name": name.value.decode(), "compute_units": cus.value, "clock_mhz": mhz.value}


def upscale_multi(engines, px: np.ndarray, out: np.ndarray = None) -> np.ndarray:
    """One image over several engines (normally one per GPU) from this process: sr_upscale_*_multi.  px (H,W,3|4) u8
    -> (3H,3W,4) u8 RGBA, or (H,W,3) f32 -> (3H,3W,3) f32; bit-identical to the single-engine call."""
    L = _lib.lib()
    arr = (C.c_void_p * len(engines))(*[e._ctx for e in engines])
    f = engines[0].factor
    if px.dtype == np.uint8:
        px = np.ascontiguousarray(px)
        h, w, c = px.shape
        if out is None:
            out = np.empty((f * h, f * w, 4), dtype=np.uint8)
        u8p = C.POINTER(C.c_uint8)
        _lib.check(L.sr_upscale_rgba8_multi(arr, len(engines), px.ctypes.data_as(u8p), c, h, w, out.ctypes.data_as(u8p)))
    else:
        px = np.ascontiguousarray(px, dtype=np.float32)
        h, w, c = px.shape
        if c != 3:
            raise ValueError("expected 3 channels")
        if out is None:
            out = np.empty((f * h, f * w, 3), dtype=np.float32)
        fp = C.POINTER(C.c_float)
        _lib.check(L.sr_upscale_f32_multi(arr, len(engines), px.ctypes.data_as(fp), h, w, out.ctypes.data_as(fp)))
    return out


def upscale_batch_multi(engines, px: np.ndarray, out: np.ndarray = None) -> np.ndarray:
    """A batch dealt over several engines from this process (sr_upscale_*_batch_multi): image i -> engines[i mod N].
    px (n,H,W,3|4) u8 -> (n,3H,3W,4) u8 RGBA, or (n,H,W,3) f32 -> (n,3H,3W,3) f32; identical to one engine's result."""
    L = _lib.lib()
    arr = (C.c_void_p * len(engines))(*[e._ctx for e in engines])
    f = engines[0].factor
    if px.dtype == np.uint8:
        px = np.ascontiguousarray(px)
        n, h, w, c = px.shape
        if out is None:
            out = np.empty((n, f * h, f * w, 4), dtype=np.uint8)
        u8p = C.POINTER(C.c_uint8)
        _lib.check(L.sr_upscale_rgba8_batch_multi(arr, len(engines), px.ctypes.data_as(u8p), c, n, h, w, out.ctypes.data_as(u8p)))
    else:
        px = np.ascontiguousarray(px, dtype=np.float32)
        n, h, w, c = px.shape
        if c != 3:
            raise ValueError("expected 3 channels")
        if out is None:
            out = np.empty((n, f * h, f * w, 3), dtype=np.float32)
        fp = C.POINTER(C.c_float)
        _lib.check(L.sr_upscale_f32_batch_multi(arr, len(engines), px.ctypes.data_as(fp), n, h, w, out.ctypes.data_as(fp)))
    return out


def comm_init_all(engines, transport: str = "rccl"):
    """One process, several engines; engine k becomes rank k.  transport "rccl": ncclCommInitAll inside libsrhip, one
    engine per device.  "local": sr_comm_init_local -- halos by peer copy on each engine's own stream, no RCCL, and the
    same device may carry several engines."""
    if transport not in ("rccl", "local"):
        raise ValueError("transport must be 'rccl' or 'local'")
    arr = (C.c_void_p * len(engines))(*[e._ctx for e in engines])
    fn = _lib.lib().sr_comm_init_all if transport == "rccl" else _lib.lib().sr_comm_init_local
    _lib.check(fn(arr, len(engines)), engines[0]._ctx)


def upscale_sharded_all(engines, bands, outs=None):
    """Bands (torch tensors, band k on engine k's device, rank order) of ONE image -> their output rows, through
    sr_upscale_sharded_*_all (halo exchange over the transport comm_init_all chose + band passes, synchronous)."""
    import torch
    n = len(engines)
    f = engines[0].factor
    u8 = bands[0].dtype == torch.uint8
    w, c = bands[0].shape[1], bands[0].shape[2]
    if outs is None:
        outs = [torch.empty((f * b.shape[0], f * w, 4 if u8 else 3), dtype=b.dtype, device=b.device) for b in bands]
    ctxs = (C.c_void_p * n)(*[e._ctx for e in engines])
    bp = (C.c_void_p * n)(*[b.data_ptr() for b in bands])
    op = (C.c_void_p * n)(*[o.data_ptr() for o in outs])
    hb = (C.c_int * n)(*[b.shape[0] for b in bands])
    L = _lib.lib()
    for b in bands:
        torch.cuda.synchronize(b.device)  # the library runs on the contexts' own streams
    if u8:
        _lib.check(L.sr_upscale_sharded_rgba8_all(ctxs, n, bp, c, hb, w, op), engines[0]._ctx)
    else:
        _lib.check(L.sr_upscale_sharded_f32_all(ctxs, n, bp, hb, w, op), engines[0]._ctx)
    return outs


class PinnedBuffer:
    """Page-locked host memory from sr_host_alloc, exposed as a numpy array (`.array`)."""

    def __init__(self, shape, dtype=np.uint8):
        self._L = _lib.lib()
        self._p = C.c_void_p()
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        _lib.check(self._L.sr_host_alloc(C.byref(self._p), nbytes))
        self.array = np.frombuffer((C.c_char * nbytes).from_address(self._p.value), dtype=dtype).reshape(shape)

    def close(self):
        if self._p:
            self.array = None
            self._L.sr_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def host_alloc(shape, dtype=np.uint8) -> PinnedBuffer:
    return PinnedBuffer(shape, dtype)


class Graph:
    """What `sr_net(FACTOR, None)` returns in the reference (network.rs:16-109),
    reduced to the two methods upscale() uses: num_params() (main.rs:162) and
    forward() (main.rs:171)."""

    def __init__(self, factor: int, device: int = 0):
        if factor not in (2, 3, 4):
            raise _lib.SrError(_lib.SR_E_FACTOR)
        self.factor = factor
        self.device = device
        self._engine: Optional[Engine] = None
        self._params_key = None

    def num_params(self) -> int:
        return _lib.lib().sr_num_params_factor(self.factor)

    def forward(self, n: int, inputs: List[NodeData], params) -> List[NodeData]:
        if len(inputs) != 1:
            raise ValueError("sr_net has exactly one input node")
        p = np.ascontiguousarray(params, dtype=np.float32)
        if p.size != self.num_params():
            raise _lib.SrError(_lib.SR_E_PARAM_COUNT)
        key = (p.size, hash(p.tobytes()))
        if self._engine is None or key != self._params_key:
            if self._engine is not None:
                self._engine.close()
            self._engine = Engine(p, self.device, self.factor)
            self._params_key = key
        inp = inputs[0]
        w, h = inp.shape.spatial_dimensions
        if inp.shape.channels != CHANNELS or inp.shape.n != n:
            raise ValueError("input shape does not match the graph's input node")
        x = np.asarray(inp.values, dtype=np.float32).reshape(n, h, w, CHANNELS)
        out = self._engine.upscale_f32(x)
        return [NodeData(DataShape(CHANNELS, [w * self.factor, h * self.factor], n), out.reshape(-1))]


def bilinear_net(factor: int = FACTOR, device: int = 0) -> "Engine":
    """reference network.rs:111 `pub fn bilinear_net(factor)` (`-p bilinear`): parameter-free."""
    return Engine((), device, factor, graph="bilinear")


def downsample_net(factor: int = FACTOR, device: int = 0) -> "Engine":
    """reference network.rs:125 `pub fn downsample_net(factor)` (`-d`): parameter-free."""
    return Engine((), device, factor, graph="downsample")


def sr_net(factor: int = FACTOR, training=None, device: int = 0) -> Graph:
    """reference network.rs:16 `pub fn sr_net(factor, training)`; inference branch only."""
    if training is not None:
        raise NotImplementedError("training graphs are outside the upscale hot path (SURVEY.md section 8)")
    return Graph(factor, device)
