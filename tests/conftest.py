import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def synth_u8(seed, n, h, w):
    """SURVEY.md 8(d) synthetic input: seeded u8 noise smoothed with a 5x5 box
    filter (edge-clamped, integer sum / 25) so activations stay in the trained range."""
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8).astype(np.int32)
    p = np.pad(a, ((0, 0), (2, 2), (2, 2), (0, 0)), mode="edge")
    s = np.zeros_like(a)
    for dy in range(5):
        for dx in range(5):
            s += p[:, dy:dy + h, dx:dx + w, :]
    return (s // 25).astype(np.uint8)


@pytest.fixture(scope="session")
def params():
    import oracle
    out = {}
    for name in ("imagenet", "imagenetlinear", "anime"):
        with open(os.path.join(ROOT, "rusty_sr_amd", "res", name + ".rsr"), "rb") as f:
            out[name] = oracle.rsr_decode(f.read())
    return out


def load_png(name):
    from PIL import Image
    return np.array(Image.open(os.path.join(GOLDEN, name)))


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
