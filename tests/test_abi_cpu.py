"""CPU suite: the C-ABI shared library loads without a GPU and exports every symbol include/zkpor.h declares; with no
usable device the product fails loudly (no CPU fallback)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd", "libzkpor.so")


def _declared():
    hdr = open(os.path.join(ROOT, "include", "zkpor.h")).read()
    return sorted(set(re.findall(r"\b(zkpor_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(LIB), "run __graft_entry__.build() first"
    lib = ctypes.CDLL(LIB)
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    import zkpor
    with pytest.raises(zkpor.ZkporError):
        zkpor.Context(0)


def test_proof_write_raw_is_host_only_and_sized():
    import numpy as np
    import zkpor
    proof = np.zeros(256, dtype=np.uint8)
    raw = zkpor.proof_write_raw(proof)
    assert raw.size == 324
    raw = zkpor.proof_write_raw(proof, np.zeros((1, 8), np.uint64), np.zeros(8, np.uint64))
    assert raw.size == 388 and raw[259] == 1   # the reference's 388-byte proof (one commitment), SURVEY.md a6.7
