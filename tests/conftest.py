import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")
YSD1 = os.path.join(GOLDEN, "ysd1_lag_5_file_0_preshuf.tsv")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def ysd1():
    import bear_oracle as o
    kmers, counts = o.parse_counts_tsv(YSD1, 3)
    return kmers, counts


def pytest_collection_modifyitems(config, items):
    """`gpu` tests need a device: on a box without one they are skipped (never run on a fallback -- there is none)."""
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="needs an MI355X (no CPU fallback exists)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
