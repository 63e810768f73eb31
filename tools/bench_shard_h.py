#!/usr/bin/env python3
"""Per-rank device time of the sharded computeH (zkpor_compute_h_shard_dev) on ONE GPU: rank 0's four steps and its
seven local transposes on arrays of 2^(log2 - wlog) elements — every rank does the same work, so this is the compute side of
one proof's computeH on 2^wlog GPUs; the all-to-alls between the steps are not part of it (this box has one GPU).
usage: python tools/bench_shard_h.py [log2 ...]      (default 26 28, 8 ranks)"""
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd"))
import zkpor  # noqa: E402

wlog = 3
ctx = zkpor.Context(0)
out = {}
for log2 in [int(x) for x in sys.argv[1:]] or [26, 28]:
    nl = log2 - wlog
    bufs = [ctx.alloc(32 << nl) for _ in range(4)]
    a, b, c, tmp = bufs
    try:
        for i, x in enumerate((a, b, c)):
            ctx.fill_fr(x, 1 << nl, 21 + i, 0)

        def proof():
            ctx.compute_h_shard_dev(log2, wlog, 0, a.ptr, b.ptr, c.ptr, 0)
            for x in (a, b, c):
                ctx.shard_transpose_dev(x.ptr, tmp.ptr, nl, wlog, True)
            ctx.compute_h_shard_dev(log2, wlog, 0, a.ptr, b.ptr, c.ptr, 1)
            for x in (a, b):                    # "ntt_h" 1 (the default): c stays where step 1 left it, step 3 subtracts it
                ctx.shard_transpose_dev(tmp.ptr, x.ptr, nl, wlog, False)
            ctx.compute_h_shard_dev(log2, wlog, 0, a.ptr, b.ptr, None, 2)
            ctx.shard_transpose_dev(a.ptr, tmp.ptr, nl, wlog, True)
            ctx.compute_h_shard_dev(log2, wlog, 0, a.ptr, None, c.ptr, 3)

        proof(); ctx.sync()                     # builds the 2^log2 domain tables
        ctx.phase_reset()
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            proof()
        ctx.sync()
        wall = (time.perf_counter() - t0) / reps * 1e3
        out[f"2^{log2} over {1 << wlog} ranks"] = {
            "local_elements": 1 << nl, "wall_ms_per_proof": round(wall, 2),
            "ntt_ms": round(ctx.phase_ms("ntt")[0] / reps, 2), "pointwise_ms": round(ctx.phase_ms("pointwise")[0] / reps, 3),
            "transposes_ms": round(wall - (ctx.phase_ms("ntt")[0] + ctx.phase_ms("pointwise")[0]) / reps, 2),
            "all_to_all_bytes_per_rank": 7 * (32 << nl) * ((1 << wlog) - 1) // (1 << wlog)}
    finally:
        for x in bufs:
            x.free()
print(json.dumps(out))
