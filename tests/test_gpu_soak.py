"""A short soak of the stage kernels (scripts/soak.py): random shapes, batches and bands through the pipe / column form
against the first form -- two code paths that must agree bit for bit -- with a geometry change on every call, and
repeated 1080p frames that must reproduce themselves (a missed wait or barrier in the LDS-DMA pipeline flickers).
A 100 s run of the same loop (9 000 shapes, 4 700 bands, 27 000 repeats, no mismatch) is in profiles/r2_soak.log."""
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
