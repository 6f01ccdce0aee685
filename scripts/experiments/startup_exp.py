#!/usr/bin/env python3
"""Where does a fresh process spend its start-up?  First HIP call (runtime + driver), sr_create (streams, weight packing and
upload), first kernel launches (code object load), in a process without torch."""
import ctypes as C, sys, time
import numpy as np
sys.path.insert(0, ".")
t0 = time.perf_counter()
from rusty_sr_amd import _lib, rsr
L = _lib.lib()
t1 = time.perf_counter()
p = C.c_void_p()
L.sr_host_alloc(C.byref(p), 4096)
t2 = time.perf_counter()
params = np.ascontiguousarray(rsr.builtin("imagenet"), dtype=np.float32)
t3 = time.perf_counter()
ctx = C.c_void_p()
rc = L.sr_create(C.byref(ctx), params.ctypes.data_as(C.POINTER(C.c_float)), params.size, 3, 0)
t4 = time.perf_counter()
L.sr_reserve_rgba8(ctx, 3, 1, 64, 64)
t5 = time.perf_counter()
L.sr_reserve_rgba8(ctx, 3, 1, 1080, 1920)
t6 = time.perf_counter()
L.sr_reserve_rgba8(ctx, 3, 1, 1080, 1920)
t7 = time.perf_counter()
print(f"dlopen libsrhip {1e3*(t1-t0):.1f} ms | first HIP call (sr_host_alloc) {1e3*(t2-t1):.1f} ms | .rsr decode {1e3*(t3-t2):.1f} ms | sr_create rc={rc} {1e3*(t4-t3):.1f} ms | "
      f"first tiny pass (first-form kernels load) {1e3*(t5-t4):.1f} ms | first 1080p pass (pipe kernels load, workspace) {1e3*(t6-t5):.1f} ms | second {1e3*(t7-t6):.1f} ms")
