#!/usr/bin/env python3
"""End-to-end walk through every row of the accelerated path on one MI355X, in the order the reference's services run them
(witness service -> prover service -> verifier), with the oracle checking each step:

  accounts --zkpor_poseidon_leaves--> leaf hashes --zkpor_tree_*--> account tree, root, per-user Merkle proofs
  CEX running totals --zkpor_cex_commitments / zkpor_batch_commitments--> the batch's public commitments
  compressed proving key --zkpor_pk_set_*_compressed--> key resident in HBM
  wire vector w --zkpor_r1cs_eval_dev--> a, b, c in HBM --zkpor_prove_tail_dev--> proof --oracle pairing verifier--> accept

The circuit is the oracle's small synthetic R1CS (the real BatchCreateUserCircuit needs gnark to compile it);
every interface used is the one the real circuit would go through.  usage: python tests/pipeline_demo.py [n_accounts]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):  # lives under tests/: it uses the oracle as checker
    sys.path.insert(0, p)
import numpy as np
import oracle as O
import zkpor


def main(n_acc=5000, verbose=True):
    say = print if verbose else (lambda *a, **k: None)
    ctx = zkpor.Context(0)
    rng = np.random.default_rng(7)
    TIER, DEPTH = 50, 28

    # ---- witness service: leaves, tree, proofs (src/witness/main.go:130-199, witness.go:319-340)
    acc = np.zeros(n_acc, dtype=zkpor.ACCOUNT_DTYPE)
    k = rng.integers(1, TIER + 1, size=n_acc)
    off = np.concatenate([[0], np.cumsum(k)[:-1]])
    acc["n_assets"] = k; acc["asset_off"] = off
    acc["id_be"] = rng.integers(0, 256, size=(n_acc, 32), dtype=np.uint8); acc["id_be"][:, 0] &= 0x0F
    for name in ("equity", "debt", "collateral"):
        acc[name][:, 0] = rng.integers(0, 1 << 50, size=n_acc, dtype=np.uint64)
    assets = np.zeros(int(k.sum()), dtype=zkpor.ASSET_DTYPE)
    for name in ("equity", "debt", "loan", "margin", "portfolio_margin"):
        assets[name] = rng.integers(0, 1 << 40, size=assets.shape[0], dtype=np.uint64)
    for i in range(n_acc):
        assets["index"][off[i]:off[i] + k[i]] = np.sort(rng.choice(500, size=k[i], replace=False))
    leaves = ctx.poseidon_leaves(acc, assets, TIER)
    sample = rng.choice(n_acc, size=min(n_acc, 16), replace=False)
    sample_assets = np.concatenate([assets[off[i]:off[i] + k[i]] for i in sample])
    sample_acc = acc[sample].copy(); sample_acc["asset_off"] = np.concatenate([[0], np.cumsum(k[sample])[:-1]])
    assert np.array_equal(leaves[sample], O.fr_to_be(O.account_leaves(sample_acc, sample_assets, TIER))), "leaf hashes differ from the oracle"
    nil = O.fr_to_be(O.poseidon_hash(O.fr_from_ints([0, 0, 0, 0, 0])))[0].tobytes()
    tree = zkpor.FixedDepthMerkleTree(ctx, DEPTH, nil, n_acc)
    tree.set_many(np.arange(n_acc, dtype=np.uint32), leaves)
    tree.build()
    root = tree.root()
    want_root, _, _ = O.merkle_build(O.fr_from_be(leaves), DEPTH, O.fr_from_be(np.frombuffer(nil, np.uint8))[0])
    assert root == O.fr_to_be(want_root)[0].tobytes(), "tree root differs from the oracle"
    users = np.arange(0, min(n_acc, 1380), dtype=np.uint32)                   # one batch of users
    proofs = tree.get_proofs(users)
    assert zkpor.verify_proofs(ctx, root, users, proofs, leaves[users], DEPTH).all()
    say(f"account tree: {n_acc} leaves, root {root.hex()[:16]}…, {users.size} Merkle proofs verified")

    # ---- the batch's public commitments (witness.go:159-198)
    import cex_cases as C
    consts = C.make_assets(500, seed=3)
    totals = C.make_totals(2, 500, seed=4)                                    # CEX state before / after the batch
    com = ctx.cex_commitments(consts, totals)
    assert np.array_equal(com, O.fr_to_be(O.cex_commitments(consts, totals))), "CEX commitments differ from the oracle"
    batch = ctx.batch_commitments(np.frombuffer(root, np.uint8), com[0], com[1], [int(users[0])], [int(users[-1])])
    say(f"batch commitment {batch[0].tobytes().hex()[:16]}… over before/after CEX commitments")
    tree.close()

    # ---- prover service: key from its compressed form, R1CS resident, w -> proof (prover.go:250-283)
    S = O.Synth(8, 3000, n_public=2, seed=19)
    pk = zkpor.ProvingKey(ctx)
    pk.set_g1_compressed(zkpor.G1_A, O.g1_compress(S.A)); pk.set_g1_compressed(zkpor.G1_B, O.g1_compress(S.B1))
    pk.set_g2_compressed(zkpor.G2_B, O.g2_compress(S.B2))
    pk.set_g1_compressed(zkpor.G1_K, O.g1_compress(S.K[S.n_public:])); pk.set_g1_compressed(zkpor.G1_Z, O.g1_compress(S.Z))
    z = np.zeros(S.n_wires, dtype=np.uint8)
    pk.set_consts(S.abd1[0], S.abd1[1], S.abd1[2], S.bd2[0], S.bd2[1], S.log2d, z, z, S.n_wires, S.n_public)
    table, mats = S.r1cs()
    r1cs = zkpor.R1CS(ctx, S.n_cons, S.n_wires, table)
    for which, (row_ptr, cid, wid) in enumerate(mats):
        r1cs.set_matrix(which, row_ptr, cid, wid)
    D = 1 << S.log2d
    dw = ctx.alloc(32 * S.n_wires).upload(S.w)
    da, db, dc = (ctx.alloc(32 * D) for _ in range(3))
    r1cs.eval_dev(dw.ptr, da.ptr, db.ptr, dc.ptr, D)
    r = O.fr_random(1, 1)[0]; s = O.fr_random(2, 1)[0]
    proof = ctx.prove_tail_dev(pk, dw.ptr, da.ptr, db.ptr, dc.ptr, r, s)
    assert np.array_equal(proof, S.prove_tail(r, s)), "proof differs from the oracle's"
    # ---- verifier (prover.go:276, src/verifier): the pairing equation, from vk + public wires + proof only
    assert S.verify_pairing(proof), "proof rejected"
    raw = zkpor.proof_write_raw(proof)
    say(f"proof of {S.n_cons} constraints: {raw.size} raw bytes, bit-exact with the oracle, accepted by the pairing verifier")
    for b in (dw, da, db, dc):
        b.free()
    r1cs.close(); pk.close(); ctx.close()
    return True


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 5000)
