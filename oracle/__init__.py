"""CPU oracle for the rusty_sr upscale hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; rusty_sr_amd never does (tests/test_abi.py::test_product_never_touches_the_oracle
enforces it).  See sr_oracle.c for what is restated and how it is pinned.
"""
from .oracle import (  # noqa: F401
    NPARAMS, SEGMENTS, build, lib, rsr_decode, forward, forward_taps, img_to_data,
    data_to_rgba8, upscale_rgba8, bilinear, downsample, forward_factor, num_params,
)
