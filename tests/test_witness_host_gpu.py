"""-m gpu: host/witness_host.hpp (the C++ mirror of Witness.Run, src/witness/witness/witness.go:138-206) driven by
tests/hostlib/witness_driver.cpp: running CEX totals on the host, then ONE launch each for all boundary-state commitments,
all Merkle proofs and all batch commitments — compared per batch with the oracle computing the same quantities the serial
way the reference does."""
import os
import struct
import subprocess

import numpy as np
import pytest

import cex_cases as C
import oracle as O
import zkpor

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_witness_run_matches_serial_restatement(zk, tmp_path):
    exe = os.path.join(HERE, "hostlib", "witness_driver")
    assert os.path.exists(exe), "tests/hostlib/witness_driver is not built (run __graft_entry__.build())"
    rng = np.random.default_rng(11)
    n_cex, per_batch, n_batches = 23, 4, 5
    n_ops = per_batch * n_batches
    consts = C.make_assets(n_cex, seed=6)
    totals = np.zeros(n_cex, dtype=O.CEX_TOTALS_DTYPE)
    for name in O.CEX_TOTALS_DTYPE.names:
        totals[name] = rng.integers(0, 1 << 40, size=n_cex, dtype=np.uint64)
    leaves = O.fr_to_be(O.fr_from_ints(list(range(1, n_ops + 1))))
    nil = O.fr_to_be(O.poseidon_hash(O.fr_from_ints([0, 0, 0, 0, 0])))[0]
    ops = []
    for i in range(n_ops):
        k = int(rng.integers(0, 6))
        a = np.zeros(k, dtype=zkpor.ASSET_DTYPE)
        a["index"] = np.sort(rng.choice(n_cex, size=k, replace=False))
        for name in ("equity", "debt", "loan", "margin", "portfolio_margin"):
            a[name] = rng.integers(0, 1 << 50, size=k, dtype=np.uint64)
        ops.append((i, a))
    path = tmp_path / "witness_input.bin"
    with open(path, "wb") as f:
        f.write(struct.pack("<4I", n_cex, n_ops, per_batch, n_ops))
        f.write(nil.tobytes()); f.write(consts.tobytes()); f.write(totals.tobytes()); f.write(leaves.tobytes())
        for idx, a in ops:
            f.write(struct.pack("<2I", idx, a.shape[0])); f.write(a.tobytes())
    res = subprocess.run([exe, str(path)], capture_output=True, text=True, timeout=180)
    assert res.returncode == 0, res.stderr
    all_lines = res.stdout.strip().splitlines()
    assert all_lines[-1] == "overflow 1"                           # SafeAdd's panic surfaces as std::overflow_error
    lines = [l for l in all_lines if l.startswith("batch ")]
    rows = [l.split(" ", 3) for l in all_lines if l.startswith("row ")]
    assert [l for l in all_lines if l.startswith("rt ")] == ["rt 1"] * n_batches and len(rows) == n_batches
    # the serial restatement: walk the batches, update the totals op by op, hash before / after
    root, proofs = O.sparse_tree(np.arange(n_ops, dtype=np.uint32), O.fr_from_be(leaves), 28, O.fr_from_be(nil[None, :])[0],
                                 np.arange(n_ops, dtype=np.uint32))
    run = totals.copy()
    for b in range(n_batches):
        before = O.cex_commitments(consts, run)[0]
        for idx, a in ops[b * per_batch:(b + 1) * per_batch]:
            for r in a:
                t = run[int(r["index"])]
                t["total_equity"] += r["equity"]; t["total_debt"] += r["debt"]; t["loan_collateral"] += r["loan"]
                t["margin_collateral"] += r["margin"]; t["portfolio_margin_collateral"] += r["portfolio_margin"]
        after = O.cex_commitments(consts, run)[0]
        mn, mx = b * per_batch, (b + 1) * per_batch - 1
        bc = O.poseidon_hash(np.concatenate([root[None, :], before[None, :], after[None, :], O.fr_from_ints([mn, mx])]))
        parts = lines[b].split()
        assert parts[0] == "batch" and int(parts[1]) == b
        assert parts[2] == O.fr_to_be(bc[None, :])[0].tobytes().hex()
        assert parts[3] == O.fr_to_be(before[None, :])[0].tobytes().hex()
        assert parts[4] == O.fr_to_be(after[None, :])[0].tobytes().hex()
        assert (int(parts[5]), int(parts[6])) == (mn, mx)
        assert parts[7] == O.fr_to_be(proofs[mn, 0][None, :])[0].tobytes().hex()      # leaf-level sibling of the batch's first user

        # the witness table row of this batch, in the reference's own encoding (base64(s2(gob(...))), witness.go:215-232), read with the
        # test-side independent decoder: what utils.DecodeBatchWitness would see
        import base64
        import gobs2 as G
        assert rows[b][1] == str(100 + b) and rows[b][2] == "0"
        v, _ = G.gob_decode(G.s2_decode(base64.b64decode(rows[b][3], validate=True)))
        be = lambda x: O.fr_to_be(x[None, :])[0].tobytes()
        assert v["BatchCommitment"] == be(bc) and v["AccountTreeRoot"] == be(root)
        assert v["BeforeCEXAssetsCommitment"] == be(before) and v["AfterCEXAssetsCommitment"] == be(after)
        assert v.get("MinAccountIndex", 0) == mn and v["MaxAccountIndex"] == mx
        assert len(v["BeforeCexAssets"]) == n_cex and len(v["CreateUserOps"]) == per_batch
        for j, (idx, a) in enumerate(ops[b * per_batch:(b + 1) * per_batch]):
            op = v["CreateUserOps"][j]
            assert op.get("AccountIndex", 0) == idx and len(op["AccountProof"]) == 28
            assert op["AccountProof"] == [be(proofs[idx, k]) for k in range(28)]
            sent = op.get("Assets", [])
            assert [x.get("Index", 0) for x in sent] == [int(i) for i in a["index"]]
            assert [x.get("Equity", 0) for x in sent] == [int(e) for e in a["equity"]]
        # the CEX asset list carries the totals that ENTERED the batch and the tier tables with CalculatePrecomputedValue's sums
        c0 = v["BeforeCexAssets"][3]
        assert c0["Symbol"] == b"a3" and c0["Index"] == 3 and c0.get("BasePrice", 0) == int(consts[3]["base_price"])
        pre, prev = 0, 0
        for t in range(12):
            bnd = int(consts[3]["loan"][t]["boundary"][0]) | (int(consts[3]["loan"][t]["boundary"][1]) << 64)
            pre += (bnd - prev) * int(consts[3]["loan"][t]["ratio"]) // 100; prev = bnd
            tr = c0["LoanRatios"][t]
            assert int.from_bytes(tr["BoundaryValue"][1:], "big") == bnd and tr["BoundaryValue"][0] == 2 and tr.get("Ratio", 0) == int(consts[3]["loan"][t]["ratio"])
            assert int.from_bytes(tr["PrecomputedValue"][1:], "big") == pre
