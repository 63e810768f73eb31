import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    os.environ.setdefault("ZKPOR_TESTING", "1")   # enables the library's test-only hooks (zkpor_set_param "debug_ntt_fault")
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950); runs through the C ABI of libzkpor.so")
    # a fresh checkout has no built artefacts (they are git-ignored): build them once (hipcc cross-compiles without a GPU)
    lib = os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd", "libzkpor.so")
    drivers = [os.path.join(ROOT, "tests", "hostlib", n) for n in ("tree_driver", "witness_driver", "libhostmath.so", "libsolverlogic.so", "libdispatch_gpu.so")]
    if not os.path.exists(lib) or not all(os.path.exists(d) for d in drivers):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def zk():
    """one libzkpor context on cuda:0 — fails loudly (no fallback) when the HIP library or the GPU is missing"""
    import zkpor
    ctx = zkpor.Context(0)
    yield ctx
    ctx.close()
