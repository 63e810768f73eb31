"""-m gpu: the in-process dispatcher of host/prover_host.hpp (the replacement of the reference's Redis task queue,
src/prover/prover/prover.go:72-247) driving REAL proofs on the device through the host-pointer C ABI with several contexts:
every batch proven exactly once, every proof bit-exact with the same call made from one context, and correct against the
synthetic key's trapdoor.  On a one-GPU box the contexts share device 0 (two proofs in flight per GPU); with more GPUs visible
they are spread round-robin — the worker threads never touch hipSetDevice, the library binds each call to its handle's GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

import oracle as O
import trapdoor as T
import zkpor

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd")


@pytest.fixture(scope="module")
def drv():
    so = os.path.join(ROOT, "tests", "hostlib", "libdispatch_gpu.so")
    src = os.path.join(ROOT, "tests", "hostlib", "dispatch_gpu.cpp")
    if not os.path.exists(so) or os.path.getmtime(src) > os.path.getmtime(so):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-o", so, src, "-L", PKG, "-lzkpor",
                               "-Wl,-rpath,$ORIGIN/../../zkmerkle-proof-of-solvency_amd"])
    return ctypes.CDLL(so)


@pytest.mark.isolated
@pytest.mark.parametrize("n_workers", [2, 3])
def test_dispatcher_drives_real_proofs_on_several_contexts(zk, drv, n_workers):
    import torch
    n_dev = min(torch.cuda.device_count(), n_workers)
    log2, n_batches, seed = 12, 7, 0xD15C
    n = 1 << log2
    ncons = n - 5                                   # ragged: the padding rows are the library's job
    vec = np.empty((n_batches, n + 3 * ncons, 4), dtype=np.uint64)
    rs = np.empty((n_batches, 8), dtype=np.uint64)
    for h in range(n_batches):
        w = O.fr_random(100 + h, n); a = O.fr_random(200 + h, ncons); b = O.fr_random(300 + h, ncons)
        vec[h] = np.concatenate([w, a, b, O.fr_mul(a, b)])
        rs[h, :4] = O.fr_random(400 + h, 1)[0]; rs[h, 4:] = O.fr_random(500 + h, 1)[0]
    proofs = np.zeros((n_batches, 256), dtype=np.uint8)
    worker_of = np.full(n_batches, -1, dtype=np.int32)
    made = np.zeros(n_workers, dtype=np.int32)
    err = ctypes.create_string_buffer(512)
    rc = drv.dispatch_gpu_run(n_workers, n_dev, ctypes.c_int64(n_batches), log2, ctypes.c_size_t(n), ctypes.c_size_t(ncons), ctypes.c_uint64(seed),
                              zkpor._p(vec), zkpor._p(rs), zkpor._p(proofs), zkpor._p(worker_of), zkpor._p(made), err, ctypes.c_size_t(512))
    assert rc == 0, err.value.decode()
    assert made.sum() == n_batches and (made >= 0).all() and (worker_of >= 0).all()
    # the same proofs from ONE context, one after the other, and from the discrete logs of the key
    pk = zkpor.ProvingKey(zk)
    try:
        pk.synth(log2, n, 3, 0, seed)
        for h in range(n_batches):
            w, a, b, c = vec[h, :n], vec[h, n:n + ncons], vec[h, n + ncons:n + 2 * ncons], vec[h, n + 2 * ncons:]
            ref = zk.prove_tail(pk, w, a, b, c, rs[h, :4], rs[h, 4:])
            assert np.array_equal(proofs[h], ref), f"batch {h} (worker {worker_of[h]})"
            hh = zk.compute_h(a, b, c, log2)
            assert T.SynthKeyTrapdoor(seed, 3, w, hh[: n - 1]).check(proofs[h], rs[h, :4], rs[h, 4:])
    finally:
        pk.close()
