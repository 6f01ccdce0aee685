#!/usr/bin/env python3
"""Host codec timing (no GPU): PNG encode / decode of an upscaled-looking 5760x3240 RGBA image through libsrpng."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from conftest import synth_u8
from scipy import ndimage
lib = C.CDLL(os.path.join("rusty_sr_amd", "libsrpng.so"))
h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
src = synth_u8(2, 1, h, w)[0].astype(np.float32)
up = np.clip(ndimage.zoom(src, (3, 3, 1), order=3), 0, 255).astype(np.uint8)
rgba = np.concatenate([up, np.full(up.shape[:2] + (1,), 255, np.uint8)], axis=2).copy()
print("image", rgba.shape, rgba.nbytes / 1e6, "MB")
path = b"/tmp/enc/out.png"
for rep in range(3):
    t0 = time.perf_counter()
    rc = lib.srpng_encode_rgba8(path, rgba.ctypes.data_as(C.c_void_p), rgba.shape[1], rgba.shape[0])
    dt = time.perf_counter() - t0
    print(f"encode rc={rc} {dt*1e3:.1f} ms  {os.path.getsize(path)/1e6:.1f} MB  ({rgba.nbytes/1e6/dt:.0f} MB/s raw)")
W, H, P = C.c_int(), C.c_int(), C.POINTER(C.c_uint8)()
for rep in range(2):
    t0 = time.perf_counter()
    rc = lib.srpng_decode_rgba8(path, C.byref(W), C.byref(H), C.byref(P))
    dt = time.perf_counter() - t0
    back = np.ctypeslib.as_array(P, shape=(H.value, W.value, 4))
    same = bool((back == rgba).all())
    lib.srpng_free(P)
    print(f"decode rc={rc} {dt*1e3:.1f} ms  identical={same}")
