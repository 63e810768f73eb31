import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); runs through the C ABI of libzkpor.so")


@pytest.fixture(scope="session")
def zk():
    """one libzkpor context on cuda:0 — fails loudly (no fallback) when the HIP library or the GPU is missing"""
    import zkpor
    ctx = zkpor.Context(0)
    yield ctx
    ctx.close()
