"""A short soak of the stage kernels (scripts/soak.py): random shapes, batches and bands through the library's own plan and
through the pipe form under random tile plans (8-row / 4-row tiles, tails, tile orders) against the first form -- code
paths that must agree bit for bit -- with a geometry change on every call, and repeated 1080p frames that must reproduce
themselves (a missed wait or barrier in the LDS-DMA pipeline flickers).  340 s of the same loop on the round-3 kernels
(14 800 shapes, 7 700 bands, 3 700 host calls, 1 500 sharded calls, 44 000 repeats, no mismatch): profiles/r3_soak.log."""
import os
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_short_soak():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import soak
    stats, bad = soak.soak(10.0, seed=123)
    assert not bad, bad[:5]
    assert stats["shapes"] > 40 and stats["repeats"] > 120 and stats["bands"] > 15 and stats["host_calls"] > 8 and stats["sharded"] > 5, stats
