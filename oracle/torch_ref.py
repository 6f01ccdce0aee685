"""Third restatement of the upscale path, on torch-CPU (oneDNN) -- TEST INFRASTRUCTURE / CPU BASELINE ONLY.

Same graph as oracle/sr_oracle.c (reference src/network.rs:27-72, semantics SURVEY.md 8(a)), written with
torch.nn.functional so that the host's tuned convolution library does the work: this is the "honest" CPU baseline
leg of bench.py (the C oracle is a correctness oracle: -ffp-contract=off, fixed summation order, untuned threading).
tests/test_cpu_baseline.py checks it against the C oracle (<= 1e-5; oneDNN sums in another order).  Never imported
by rusty_sr_amd."""
import numpy as np
import torch
import torch.nn.functional as F

from .oracle import SEGMENTS


def _seg(params, name):
    off, n, shape = SEGMENTS[name]
    return torch.from_numpy(np.ascontiguousarray(params[off:off + n], dtype=np.float32)).reshape(shape)


class TorchNet:
    """Weights converted once: alumina [O][KH][KW][I] -> torch [O][I][KH][KW] (cross-correlation in both)."""

    def __init__(self, params):
        p = np.asarray(params, dtype=np.float32)
        self.w = {k: _seg(p, k).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
                  for k in ("conv0", "conv1", "conv2", "conv3", "conv5", "conv6", "conv7", "conv8", "conv9", "conv10")}
        self.b = {k: _seg(p, k).reshape(-1) for k in ("f_bias", "l1_bias", "l2_bias", "l3_bias", "expand_bias")}
        self.beta = {k: _seg(p, k).reshape(1, -1, 1, 1) for k in ("f_activ", "l1_activ", "l2_activ", "l3_activ")}

    @staticmethod
    def _belu(v, beta):  # network.rs:35,54-56: beta*x + sqrt(x*x+1) - 1
        return beta * v + torch.sqrt(v * v + 1.0) - 1.0

    @torch.no_grad()
    def forward(self, x_nhwc: np.ndarray) -> np.ndarray:
        """(n,H,W,3) f32 -> (n,3H,3W,3) f32, pre-quantisation (graph.forward, main.rs:171)."""
        x = torch.from_numpy(np.ascontiguousarray(x_nhwc, dtype=np.float32)).permute(0, 3, 1, 2).contiguous(memory_format=torch.channels_last)
        w, b, a = self.w, self.b, self.beta
        f = self._belu(F.conv2d(x, w["conv0"], b["f_bias"], padding=2), a["f_activ"])
        l1 = self._belu(F.conv2d(f, w["conv1"], b["l1_bias"], padding=2), a["l1_activ"])
        l2 = self._belu(F.conv2d(f, w["conv2"], b["l2_bias"], padding=2) + F.conv2d(l1, w["conv5"], None, padding=1), a["l2_activ"])
        l3 = self._belu(F.conv2d(f, w["conv3"], b["l3_bias"], padding=2) + F.conv2d(l1, w["conv6"], None, padding=1) +
                        F.conv2d(l2, w["conv8"], None, padding=1), a["l3_activ"])
        e = (F.conv2d(l1, w["conv7"], b["expand_bias"], padding=1) + F.conv2d(l2, w["conv9"], None, padding=1) +
             F.conv2d(l3, w["conv10"], None, padding=1))
        n, _, H, W = e.shape
        # Expand (network.rs:39): channel (dy*3+dx)*3+c -> out[3y+dy][3x+dx][c]  (NOT torch's pixel_shuffle order)
        d2s = e.reshape(n, 3, 3, 3, H, W).permute(0, 4, 1, 5, 2, 3).reshape(n, 3 * H, 3 * W, 3)
        lin = F.interpolate(x, scale_factor=3, mode="bilinear", align_corners=False).permute(0, 2, 3, 1)  # LinearInterp, :27
        return (lin + d2s).contiguous().numpy()

    @torch.no_grad()
    def band_stages(self):
        """The five stages as callables on ROW-EXTENDED (rows, W, C) maps that return a band's own rows (rusty_sr_amd.shard
        upscale_sharded_layers: the per-layer feature-halo protocol, tests/test_shard_gloo.py): horizontal zero padding as ever, NO vertical
        padding -- the extension rows are the neighbours' rows, or the zeros of a true image edge."""
        w, b, a = self.w, self.b, self.beta
        nchw = lambda t: t.permute(2, 0, 1)[None].contiguous()
        nhwc = lambda t: t[0].permute(1, 2, 0).contiguous()
        conv = lambda t, key, bias, k: F.conv2d(nchw(t), w[key], None if bias is None else b[bias], padding=(0, k // 2))
        crop = lambda t, d: t[d:t.shape[0] - d]  # an extension of 2 rows seen by a 3x3 conv
        s0 = lambda x2: nhwc(self._belu(conv(x2, "conv0", "f_bias", 5), a["f_activ"]))
        s1 = lambda f2: nhwc(self._belu(conv(f2, "conv1", "l1_bias", 5), a["l1_activ"]))
        s2 = lambda f2, l1: nhwc(self._belu(conv(f2, "conv2", "l2_bias", 5) + conv(l1, "conv5", None, 3), a["l2_activ"]))
        s3 = lambda f2, l1, l2: nhwc(self._belu(conv(f2, "conv3", "l3_bias", 5) + conv(l1, "conv6", None, 3) + conv(l2, "conv8", None, 3), a["l3_activ"]))

        def s4(l1, l2, l3, x1, top_edge, bot_edge):
            e = conv(l1, "conv7", "expand_bias", 3) + conv(l2, "conv9", None, 3) + conv(l3, "conv10", None, 3)
            n, _, H, W = e.shape
            d2s = e.reshape(n, 3, 3, 3, H, W).permute(0, 4, 1, 5, 2, 3).reshape(3 * H, 3 * W, 3)
            lin = F.interpolate(nchw(x1), scale_factor=3, mode="bilinear", align_corners=False)[0].permute(1, 2, 0)
            lin = lin[(0 if top_edge else 3):lin.shape[0] - (0 if bot_edge else 3)]  # the neighbours' rows were only there to interpolate towards
            return (lin + d2s).contiguous()
        return [s0, s1, s2, s3, s4]
