"""Throughput of the rows either side of the prove tail, on one MI355X:
decompression of compressed key points, device-side constraint evaluation, and the device-resident account tree.
usage: python tests/bench_aux.py   -> one JSON line.  Lives under tests/ because it uses the oracle (to produce compressed
inputs and to check outputs), which only test code may touch."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle as O
import zkpor

ctx = zkpor.Context(0)
out = {}

# ---- f2: decompression (points from the synthetic key generator, compressed by the oracle) ----
pk = zkpor.ProvingKey(ctx)
LOG = 22
pk.synth(LOG, 1 << LOG, 3, 1 << (LOG - 2), seed=7)
ptr, cnt = pk.g1_dev(zkpor.G1_A)
buf = zkpor.DevBuf.__new__(zkpor.DevBuf); buf.ctx = ctx; buf.nbytes = cnt * 64; buf.ptr = ptr
pts = buf.download(np.uint64, (cnt, 8))
comp = O.g1_compress(pts)
ctx.g1_decompress(comp[:1024])
ctx.phase_reset()
got = ctx.g1_decompress(comp)
ms, _ = ctx.phase_ms("decompress")
assert np.array_equal(got, pts)
out["g1_decompress"] = {"points": int(cnt), "kernel_ms": round(ms, 2), "points_per_s": round(cnt / (ms * 1e-3))}
ptr2, cnt2 = pk.g2_dev(zkpor.G2_B)
buf2 = zkpor.DevBuf.__new__(zkpor.DevBuf); buf2.ctx = ctx; buf2.nbytes = cnt2 * 128; buf2.ptr = ptr2
n2 = min(cnt2, 1 << 20)
pts2 = buf2.download(np.uint64, (cnt2, 16))[:n2]
comp2 = O.g2_compress(pts2)
ctx.phase_reset()
got2 = ctx.g2_decompress(comp2)
ms, _ = ctx.phase_ms("decompress")
assert np.array_equal(got2, pts2)
out["g2_decompress"] = {"points": int(n2), "kernel_ms": round(ms, 2), "points_per_s": round(n2 / (ms * 1e-3))}
pk.close()

# ---- f1: constraint evaluation, 2^24 constraints x 3 matrices, 1-4 terms per row, gnark-like coefficient mix ----
nC = 1 << 24; nW = 1 << 24
rng = np.random.default_rng(1)
table = O.fr_from_ints([0, 1, O.R_MOD - 1, 2, O.R_MOD - 2] + [int(x) for x in rng.integers(3, 1 << 62, size=59)])
r = zkpor.R1CS(ctx, nC, nW, table)
total_terms = 0
for which in range(3):
    lens = rng.integers(1, 5, size=nC).astype(np.uint64)
    row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    nnz = int(row_ptr[-1]); total_terms += nnz
    cid = rng.choice(np.arange(64, dtype=np.uint32), size=nnz, p=[0.0] + [0.5, 0.2, 0.05, 0.05] + [0.2 / 59] * 59).astype(np.uint32)
    wid = rng.integers(0, nW, size=nnz, dtype=np.uint32)
    r.set_matrix(which, row_ptr, cid, wid)
dw = ctx.alloc(32 * nW); ctx.fill_fr(dw, nW, 3, 1)
da, db, dc = (ctx.alloc(32 * nC) for _ in range(3))
r.eval_dev(dw.ptr, da.ptr, db.ptr, dc.ptr, nC); ctx.sync()
ctx.phase_reset()
for _ in range(3):
    r.eval_dev(dw.ptr, da.ptr, db.ptr, dc.ptr, nC)
ctx.sync()
ms, calls = ctx.phase_ms("r1cs_eval")
out["r1cs_eval"] = {"constraints": nC, "terms": total_terms, "kernel_ms": round(ms / calls, 2),
                    "constraints_per_s": round(nC / (ms / calls * 1e-3)), "host_bytes_saved_per_proof_at_2p26": 3 * 32 * (1 << 26)}
r.close()
for b in (dw, da, db, dc):
    b.free()

# ---- a11: device-resident tree, 2^27 leaves set from device memory, Build, 1380 proofs ----
N = 1 << 27
leaves = ctx.alloc(32 * N); ctx.fill_fr(leaves, N, 9, 0)
nil = O.fr_to_be(O.poseidon_hash(O.fr_from_ints([0, 0, 0, 0, 0])))[0].tobytes()
t = zkpor.FixedDepthMerkleTree(ctx, 28, nil, N)
t.set_range_dev(0, leaves.ptr, N); ctx.sync()
t0 = time.time(); t.build(); dt_build = time.time() - t0
keys = np.random.default_rng(2).integers(0, N, size=1380, dtype=np.uint32)
t0 = time.time(); proofs = t.get_proofs(keys); dt_proofs = time.time() - t0
ok = zkpor.verify_proofs(ctx, t.root(), keys, proofs, t.get_many(keys), 28)
assert ok.all()
out["account_tree"] = {"leaves": N, "build_ms": round(dt_build * 1e3, 1), "leaves_per_s": round(N / dt_build),
                       "get_proofs_1380_ms": round(dt_proofs * 1e3, 2)}
t.close(); leaves.free()
# ---- a12 / f3: CEX asset-list commitments, 4096 boundary states x 500 assets (one state per thread) ----
import cex_cases as C
consts = C.make_assets(500, seed=3)
nst = 4096
totals = np.zeros((nst, 500), dtype=O.CEX_TOTALS_DTYPE)
for name in O.CEX_TOTALS_DTYPE.names:
    totals[name] = np.random.default_rng(5).integers(0, 1 << 62, size=(nst, 500), dtype=np.uint64)
ctx.cex_commitments(consts, totals[:64])
ctx.phase_reset()
com = ctx.cex_commitments(consts, totals)
ms, _ = ctx.phase_ms("cex_commitments")
assert np.array_equal(com[:2], O.fr_to_be(O.cex_commitments(consts, totals[:2])))
out["cex_commitments"] = {"states": nst, "assets": 500, "kernel_ms": round(ms, 1), "states_per_s": round(nst / (ms * 1e-3)),
                          "width13_permutations_per_s": round(nst * 834 / (ms * 1e-3))}
print(json.dumps(out))
