#!/usr/bin/env python3
"""Secondary measurement (Poseidon rows): account-tree build at the reference's BenchmarkBuild size
(2^27 leaves, src/utils/merkletree/merkletree_test.go:287-298) and leaf hashing for synthetic tier-50 accounts.
Prints one JSON line; not the headline metric (bench.py is)."""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd"))
import numpy as np
import zkpor


def main():
    log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 27
    n_acc = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 18
    ctx = zkpor.Context(0)
    n = 1 << log2
    buf = ctx.alloc(32 * n)
    ctx.fill_fr(buf, n, 5, 0)
    nil = np.array([1, 2, 3, 4], dtype=np.uint64)
    ctx.merkle_build_dev(buf.ptr, 1 << 16, 28, nil)  # warm-up (tables, workspace)
    ctx.phase_reset()
    t0 = time.perf_counter()
    root = ctx.merkle_build_dev(buf.ptr, n, 28, nil)
    dt = time.perf_counter() - t0
    tree_ms, _ = ctx.phase_ms("poseidon_tree")
    buf.free()
    # leaves: synthetic accounts with 4..50 assets
    rng = np.random.default_rng(1)
    acc = np.zeros(n_acc, dtype=zkpor.ACCOUNT_DTYPE)
    k = rng.integers(4, 51, size=n_acc)
    off = np.concatenate([[0], np.cumsum(k)[:-1]])
    acc["n_assets"] = k; acc["asset_off"] = off
    acc["id_be"][:, 24:] = rng.integers(0, 256, size=(n_acc, 8), dtype=np.uint8)
    acc["equity"][:, 0] = rng.integers(0, 1 << 40, size=n_acc, dtype=np.uint64)
    tot = int(k.sum())
    assets = np.zeros(tot, dtype=zkpor.ASSET_DTYPE)
    for name in ("equity", "debt", "loan", "margin", "portfolio_margin"):
        assets[name] = rng.integers(0, 1 << 40, size=tot, dtype=np.uint64)
    idx = np.empty(tot, dtype=np.uint32)
    for i in range(n_acc):  # sorted distinct indices per account
        idx[off[i]:off[i] + k[i]] = np.sort(rng.choice(350, size=k[i], replace=False))
    assets["index"] = idx
    ctx.poseidon_leaves(acc[:1024], assets, 50)
    ctx.phase_reset()
    t1 = time.perf_counter()
    ctx.poseidon_leaves(acc, assets, 50)
    dl = time.perf_counter() - t1
    leaf_ms, _ = ctx.phase_ms("poseidon_leaf")
    print(json.dumps({"merkle_leaves": n, "merkle_build_ms_gpu": tree_ms, "merkle_build_ms_wall": dt * 1e3,
                      "merkle_hashes_per_s": (n - 1) / (tree_ms * 1e-3), "merkle_algorithmic_GBps": 64.0 * n / (tree_ms * 1e-3) / 1e9,
                      "leaf_accounts": n_acc, "leaf_kernel_ms": leaf_ms, "leaf_wall_ms_incl_pcie": dl * 1e3,
                      "leaf_accounts_per_s_kernel": n_acc / (leaf_ms * 1e-3)}))
    ctx.close()


if __name__ == "__main__":
    main()
