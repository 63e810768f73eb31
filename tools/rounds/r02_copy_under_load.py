#!/usr/bin/env python3
"""Does a host-to-device copy keep its rate while another context's VALU-bound kernels fill the GPU?  (Why two callers of the
host-pointer ABI overlap worse than two callers with resident inputs: DESIGN.md §5 `boundary`.)
Load: Poseidon tree builds (2^26 leaves, VALU-bound) in a loop on context A.  Probe: 1 GiB uploads on context B, from page-locked
and from pageable memory, idle and under load.  Usage: python tools/r02_copy_under_load.py  (env HSA_ENABLE_SDMA etc. from the caller)"""
import ctypes
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch  # noqa: E402,F401  (device runtime)
import zkpor  # noqa: E402
import oracle as O  # noqa: E402

torch.cuda.set_device(0)
A = zkpor.Context(0)
B = zkpor.Context(0)
n = 1 << 26
leaves = A.alloc(32 * n)
A.fill_fr(leaves, n, 5, 0)
nil = O.poseidon_hash(O.fr_from_ints([0, 0, 0, 0, 0]))
A.merkle_build_dev(leaves.ptr, 1 << 16, 28, nil)
A.sync()
GB = 1 << 30
host = np.random.default_rng(1).integers(0, 1 << 62, size=GB // 8, dtype=np.uint64)
dst = B.alloc(GB)


def probe(label, reps=6):
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        B._ck(B.lib.zkpor_dev_upload(B.h, ctypes.c_void_p(dst.ptr), zkpor._p(host), ctypes.c_size_t(GB)))
        t.append(time.perf_counter() - t0)
    t.sort()
    print(f"{label:44s} median {GB / t[len(t) // 2] / 1e9:6.1f} GB/s   best {GB / t[0] / 1e9:6.1f}   worst {GB / t[-1] / 1e9:6.1f}", flush=True)


stop = [False]
builds = [0]


def load():
    torch.cuda.set_device(0)
    while not stop[0]:
        A.merkle_build_dev(leaves.ptr, n, 28, nil)
        builds[0] += 1


print({k: os.environ.get(k) for k in ("HSA_ENABLE_SDMA", "GPU_MAX_HW_QUEUES", "HIP_FORCE_DEV_KERNARG")})
t0 = time.perf_counter()
for _ in range(4):
    A.merkle_build_dev(leaves.ptr, n, 28, nil)
print(f"load alone: {(time.perf_counter() - t0) / 4 * 1e3:.0f} ms per tree build of 2^26 leaves", flush=True)
probe("pageable, GPU idle")
B._ck(B.lib.zkpor_host_register(B.h, zkpor._p(host), ctypes.c_size_t(GB)))
probe("page-locked, GPU idle")
th = threading.Thread(target=load)
t0 = time.perf_counter()
th.start()
time.sleep(0.5)
b0 = builds[0]; t1 = time.perf_counter()
probe("page-locked, other context VALU-bound", reps=60)
b1 = builds[0]; t2 = time.perf_counter()
print(f"   load meanwhile: {(t2 - t1) / max(1, b1 - b0) * 1e3:.0f} ms per tree build ({b1 - b0} builds)")
B._ck(B.lib.zkpor_host_unregister(B.h, zkpor._p(host)))
probe("pageable, other context VALU-bound", reps=60)
b2 = builds[0]; t3 = time.perf_counter()
print(f"   load meanwhile: {(t3 - t2) / max(1, b2 - b1) * 1e3:.0f} ms per tree build ({b2 - b1} builds)")
stop[0] = True
th.join()

# ---- the same question with the prover's own kernels as the load: device-resident proofs at 2^24 on context A ----
log2 = int(os.environ.get("LOAD_LOG2", "24"))
D = 1 << log2
A.set_param("msm_tables", 4)
pk = zkpor.ProvingKey(A)
pk.synth(log2, D, 3, 0, seed=0x5A4B504F52)
bufs = [A.alloc(32 * D) for _ in range(4)]
for i, b_ in enumerate(bufs):
    A.fill_fr(b_, D, 11 + i, 2 if i == 0 else 0)
r_ = O.fr_random(5, 1)[0]; s_ = O.fr_random(6, 1)[0]


def proofs(nn):
    t0 = time.perf_counter()
    for _ in range(nn):
        A.prove_tail_dev(pk, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, r_, s_)
    return (time.perf_counter() - t0) / nn * 1e3


proofs(2)
print(f"prove tail 2^{log2}, resident, alone: {proofs(8):.1f} ms per proof", flush=True)
B._ck(B.lib.zkpor_host_register(B.h, zkpor._p(host), ctypes.c_size_t(GB)))
for chunk_mb in (1024, 64, 8):
    stop = [False]
    copied = [0]

    def copier():
        torch.cuda.set_device(0)
        step = chunk_mb << 20
        while not stop[0]:
            for off in range(0, GB, step):
                B._ck(B.lib.zkpor_dev_upload(B.h, ctypes.c_void_p(dst.ptr + off), ctypes.c_void_p(host.ctypes.data + off), ctypes.c_size_t(step)))
                copied[0] += step
                if stop[0]:
                    break

    th = threading.Thread(target=copier)
    t0 = time.perf_counter()
    th.start()
    time.sleep(0.2)
    ms = proofs(8)
    dt = time.perf_counter() - t0
    stop[0] = True
    th.join()
    print(f"   with page-locked uploads of {chunk_mb:4d} MiB running on the other context ({copied[0] / dt / 1e9:5.1f} GB/s): {ms:.1f} ms per proof", flush=True)
B._ck(B.lib.zkpor_host_unregister(B.h, zkpor._p(host)))
