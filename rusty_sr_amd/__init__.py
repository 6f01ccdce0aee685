"""rusty_sr_amd -- MI355X (gfx950) engine for the upscale hot path of
millardjn/rusty_sr v1.  The compute lives in libsrhip.so (hand-written HIP,
C ABI in include/srhip.h); this package is the thin host side."""
from . import rsr  # noqa: F401
from .engine import (  # noqa: F401
    CHANNELS, FACTOR, DataShape, Engine, Graph, NodeData, bilinear_net, downsample_net, img_to_data, sr_net,
    comm_init_all, upscale_batch_multi, upscale_multi, upscale_sharded_all,
)
from ._lib import SrError  # noqa: F401
