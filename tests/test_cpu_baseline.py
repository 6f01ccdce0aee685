"""bench.py's second cpu_baseline leg (oracle/torch_ref.py: torch-CPU / oneDNN) computes the same function as the
pinned C oracle -- otherwise timing it beside the GPU would be meaningless."""
import numpy as np

import oracle
from conftest import synth_u8
from oracle.torch_ref import TorchNet


def test_torch_cpu_leg_matches_the_oracle(params):
    for name, (n, h, w) in (("imagenet", (2, 37, 53)), ("anime", (1, 8, 9)), ("imagenetlinear", (1, 1, 1))):
        x = oracle.img_to_data(synth_u8(70 + h, n, h, w))
        want = oracle.forward(params[name], x)
        got = TorchNet(params[name]).forward(x)
        assert got.shape == want.shape
        assert np.abs(got - want).max() < 1e-5, name
