"""ctypes binding of libsrhip.so (include/srhip.h).  There is no fallback: if
the HIP library is missing or cannot be loaded every engine entry point raises."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SRHIP_LIB", os.path.join(HERE, "libsrhip.so"))  # override: A/B kernel experiments

SR_FACTOR = 3
SR_NUM_PARAMS = 130459
SR_HALO = 7

SR_OK, SR_E_INVALID, SR_E_PARAM_COUNT, SR_E_FACTOR, SR_E_NO_DEVICE = 0, -1, -2, -3, -4
SR_E_HIP, SR_E_NOMEM, SR_E_BYTEVEC, SR_E_HALO, SR_E_COMM, SR_E_DOMAIN = -5, -6, -7, -8, -9, -10
SR_COMM_ID_BYTES = 128
SR_PRECISION_F32, SR_PRECISION_SPLIT_F16 = 0, 1
SR_GRAPH_SR_NET, SR_GRAPH_BILINEAR, SR_GRAPH_DOWNSAMPLE = 0, 1, 2

# every symbol include/srhip.h declares: (restype, argtypes)
_vp, _fp, _u8p, _dp = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
_sz, _i = C.c_size_t, C.c_int
SYMBOLS = {
    "sr_rsr_decode": (_i, [_u8p, _sz, _fp, _sz, C.POINTER(_sz)]),
    "sr_rsr_encode": (_i, [_fp, _sz, _u8p, _sz, C.POINTER(_sz)]),
    "sr_create": (_i, [C.POINTER(_vp), _fp, _sz, _i, _i]),
    "sr_create_graph": (_i, [C.POINTER(_vp), _i, _fp, _sz, _i, _i]),
    "sr_num_params": (_i, [_i]),
    "sr_num_params_factor": (_i, [_i]),
    "sr_destroy": (None, [_vp]),
    "sr_upscale_f32": (_i, [_vp, _fp, _i, _i, _i, _fp]),
    "sr_upscale_rgba8": (_i, [_vp, _u8p, _i, _i, _i, _i, _u8p]),
    "sr_reserve_f32": (_i, [_vp, _i, _i, _i]),
    "sr_reserve_rgba8": (_i, [_vp, _i, _i, _i, _i]),
    "sr_upscale_f32_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "sr_upscale_rgba8_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sr_upscale_band_f32_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "sr_upscale_band_rgba8_dev": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "sr_read_feature": (_i, [_vp, _i, _fp, _sz]),
    "sr_set_precision": (_i, [_vp, _i]),
    "sr_check_domain": (_i, [_vp]),
    "sr_upscale_f32_multi": (_i, [C.POINTER(_vp), _i, _fp, _i, _i, _fp]),
    "sr_upscale_rgba8_multi": (_i, [C.POINTER(_vp), _i, _u8p, _i, _i, _i, _u8p]),
    "sr_upscale_f32_batch_multi": (_i, [C.POINTER(_vp), _i, _fp, _i, _i, _i, _fp]),
    "sr_upscale_rgba8_batch_multi": (_i, [C.POINTER(_vp), _i, _u8p, _i, _i, _i, _i, _u8p]),
    "sr_comm_available": (_i, []),
    "sr_comm_unique_id": (_i, [_u8p, _sz]),
    "sr_comm_init_rank": (_i, [_vp, _u8p, _sz, _i, _i]),
    "sr_comm_init_all": (_i, [C.POINTER(_vp), _i]),
    "sr_comm_init_local": (_i, [C.POINTER(_vp), _i]),
    "sr_comm_destroy": (None, [_vp]),
    "sr_comm_rank": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "sr_last_comm_error": (_i, [_vp]),
    "sr_last_comm_ms": (_i, [_vp, _dp]),
    "sr_last_comm_exposed_ms": (_i, [_vp, _dp]),
    "sr_upscale_sharded_f32_dev": (_i, [_vp, _vp, _i, _i, _vp, _vp]),
    "sr_upscale_sharded_rgba8_dev": (_i, [_vp, _vp, _i, _i, _i, _vp, _vp]),
    "sr_upscale_sharded_f32_all": (_i, [C.POINTER(_vp), _i, C.POINTER(_vp), C.POINTER(_i), _i, C.POINTER(_vp)]),
    "sr_upscale_sharded_rgba8_all": (_i, [C.POINTER(_vp), _i, C.POINTER(_vp), _i, C.POINTER(_i), _i, C.POINTER(_vp)]),
    "sr_set_pipeline": (_i, [_vp, _i]),
    "sr_host_alloc": (_i, [C.POINTER(_vp), _sz]),
    "sr_host_free": (None, [_vp]),
    "sr_set_profiling": (_i, [_vp, _i]),
    "sr_last_timing": (_i, [_vp, _dp, _dp, _dp, _dp]),
    "sr_device_info": (_i, [_vp, C.c_char_p, _sz, C.POINTER(_i), C.POINTER(_i)]),
    "sr_last_hip_error": (_i, [_vp]),
    "sr_strerror": (C.c_char_p, [_i]),
}

# include/srhip_experimental.h: A/B tuning switches (no result bit depends on them), outside the drop-in ABI
EXPERIMENTAL = {
    "sr_set_experiment": (_i, [_vp, C.c_char_p, C.c_char_p]),
    "sr_get_experiment": (_i, [_vp, C.c_char_p, C.c_char_p, _sz]),
}

_lib = None


class SrError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        msg = lib().sr_strerror(status).decode() if _lib is not None else f"status {status}"
        super().__init__(f"{msg}{(' (' + detail + ')') if detail else ''}")


def lib():
    """Load libsrhip.so (built in-tree by rusty_sr_amd.build / __graft_entry__.build)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build it with `python -m rusty_sr_amd.build` "
                "(hipcc --offload-arch=gfx950). rusty_sr_amd has no CPU fallback.")
        # torch ships its own copy of the HIP runtime (SONAME libamdhip64.so.7, found
        # through its RPATH).  Load it FIRST so libsrhip's NEEDED libamdhip64.so.7
        # resolves to that same instance; two HIP runtimes in one process cannot
        # both own the GPU ("No HIP GPUs are available" from whichever comes second).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in list(SYMBOLS.items()) + list(EXPERIMENTAL.items()):
            f = getattr(L, name)  # AttributeError if the ABI is incomplete
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def check(status, ctx=None):
    if status != SR_OK:
        detail = ""
        if ctx and status == SR_E_HIP:
            detail = f"hipError {lib().sr_last_hip_error(ctx)}"
        if ctx and status == SR_E_COMM:
            detail = f"ncclResult {lib().sr_last_comm_error(ctx)}"
        raise SrError(status, detail)
