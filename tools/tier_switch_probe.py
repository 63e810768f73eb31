#!/usr/bin/env python3
"""What does a tier switch cost on one MI355X?  Times the pieces of replacing a resident 2^26 key: freeing the old key, building
a key as plain arrays (msm_tables 1: the synthetic generator stands in for the upload — 28 GB from host memory cross PCIe in ~0.6 s at the measured
50 GB/s), and building the fixed-base tables on top (msm_tables 2 and 4).  Also reports the HBM each layout holds."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd"))
import torch
import zkpor


def main():
    log2 = int(sys.argv[1]) if len(sys.argv) > 1 else 26
    ctx = zkpor.Context(0)
    out = {}
    free0 = torch.cuda.mem_get_info(0)[0]
    for m in (1, 2, 4):
        ctx.set_param("msm_tables", m)
        pk = zkpor.ProvingKey(ctx)
        t0 = time.perf_counter()
        pk.synth(log2, 1 << log2, 3, 1 << (log2 - 2), seed=7)
        ctx.sync()
        t_load = time.perf_counter() - t0
        used = free0 - torch.cuda.mem_get_info(0)[0]
        t0 = time.perf_counter()
        pk.close()
        ctx.sync()
        t_free = time.perf_counter() - t0
        out[f"tables_{m}"] = {"build_seconds": round(t_load, 2), "free_seconds": round(t_free, 3), "hbm_gb": round(used / 1e9, 1)}
    out["note"] = ("build_seconds of tables_1 = generating the synthetic points on the device (stands in for the one-time upload or the device "
                   "decompression of a .pk); the difference to tables_2 / tables_4 is the fixed-base table build")
    print(json.dumps(out))
    ctx.close()


if __name__ == "__main__":
    main()
