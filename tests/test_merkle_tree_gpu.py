"""-m gpu: the device-resident FixedDepthMerkleTree through the C ABI (zkpor_tree_*), test for test against the
reference's src/utils/merkletree/merkletree_test.go (TestNewFixedDepthMerkleTree :39, TestSetBuildAndRoot :56,
TestSetMultipleKeys :72, TestGetProof :95, TestGetProofVerify :114, TestConcurrentSet :151,
TestSequentialKeysConcurrent :220, TestCapacityOverflowCheck :253, TestCapacityAtMax :267), plus bit-exact comparison
of roots and proofs with the oracle's restatement of the same tree (oracle/poseidon.hpp SparseMerkleTree)."""
import numpy as np
import pytest

import oracle as O
import zkpor

pytestmark = pytest.mark.gpu
DEPTH = 28


def nil_account_hash():  # merkletree_test.go:14-18
    return O.poseidon_hash(O.fr_from_ints([0, 0, 0, 0, 0]))


def make_leaf_values(ks):  # merkletree_test.go:32-37: Fr(k+1), 32 bytes big-endian
    return O.fr_to_be(O.fr_from_ints([int(k) + 1 for k in ks]))


def new_test_tree(zk, capacity, depth=DEPTH):
    return zkpor.FixedDepthMerkleTree(zk, depth, O.fr_to_be(nil_account_hash())[0].tobytes(), capacity)


def oracle_tree(keys, leaves_be, query, depth=DEPTH):
    root, proofs = O.sparse_tree(keys, O.fr_from_be(leaves_be), depth, nil_account_hash(), query)
    return O.fr_to_be(root)[0].tobytes(), O.fr_to_be(proofs.reshape(-1, 4)).reshape(len(query), depth, 32)


def test_new_tree_root_is_nil_hash(zk):
    t = new_test_tree(zk, 100)
    try:
        root = t.root()
        assert len(root) == 32
        assert root == t.nil_hash(28)
        # the whole nil chain against the oracle
        _, nilh, _ = O.merkle_build(np.zeros((0, 4), np.uint64), DEPTH, nil_account_hash())
        for l in range(DEPTH + 1):
            assert t.nil_hash(l) == O.fr_to_be(nilh[l:l + 1])[0].tobytes()
    finally:
        t.close()


def test_set_build_and_root(zk):
    t = new_test_tree(zk, 100)
    try:
        empty = t.root()
        leaf = make_leaf_values([0])[0]
        t.set(0, leaf.tobytes())
        assert t.root() == empty          # Root only reflects Sets followed by a Build (merkletree.go:24)
        t.build()
        new_root = t.root()
        assert new_root != empty and len(new_root) == 32
        assert new_root == oracle_tree([0], leaf[None, :], [0])[0]
    finally:
        t.close()


def test_set_multiple_keys_and_get(zk):
    t = new_test_tree(zk, 1000)
    try:
        vals = make_leaf_values(range(3))
        for i in range(3):
            t.set(i, vals[i].tobytes())
        t.build()
        for i in range(3):
            assert t.get(i) == vals[i].tobytes()
        assert t.get(999) == t.nil_hash(0)                      # unset key
        assert t.get(5000) == t.nil_hash(0)                     # beyond capacity (merkletree.go:288-290)
    finally:
        t.close()


def test_get_proof_shape(zk):
    t = new_test_tree(zk, 100)
    try:
        t.set(5, make_leaf_values([42])[0].tobytes())
        t.build()
        proof = t.get_proof(5)
        assert len(proof) == 28 and all(len(p) == 32 for p in proof)
    finally:
        t.close()


def test_get_proof_verify_sparse_keys(zk):
    t = new_test_tree(zk, 100000)
    try:
        keys = [0, 5, 100, 50000]
        vals = make_leaf_values(keys)
        for k, v in zip(keys, vals):
            t.set(k, v.tobytes())
        t.build()
        root = t.root()
        query = keys + [999, 99999]                              # two empty keys as in the reference test
        want_root, want_proofs = oracle_tree(keys, vals, query)
        assert root == want_root
        got = t.get_proofs(query)
        assert np.array_equal(got, want_proofs)
        leaves = t.get_many(query)
        assert zkpor.verify_proofs(zk, root, query, got, leaves, DEPTH).all()
        for k in query:
            assert zkpor.verify_proof(zk, root, k, t.get_proof(k), t.get(k), DEPTH)
            # the oracle's VerifyProof accepts the device's proof too
            assert O.merkle_verify(O.fr_from_be(np.frombuffer(root, np.uint8)), k, O.fr_from_be(t.get_proofs([k])[0]),
                                   O.fr_from_be(np.frombuffer(t.get(k), np.uint8)))
        # negative cases of VerifyProof (:335-337 and a wrong leaf / wrong key)
        assert not zkpor.verify_proof(zk, root, 5, t.get_proof(5)[:-1], t.get(5), DEPTH)
        assert not zkpor.verify_proof(zk, root, 5, t.get_proof(5), t.get(100), DEPTH)
        assert not zkpor.verify_proof(zk, root, 4, t.get_proof(5), t.get(5), DEPTH)
    finally:
        t.close()


@pytest.mark.parametrize("num_keys", [1000, 10000])
def test_dense_keys_all_proofs_verify(zk, num_keys):
    """TestConcurrentSet / TestSequentialKeysConcurrent: every key set (here: one batched Set), all proofs verify"""
    t = new_test_tree(zk, num_keys)
    try:
        keys = np.arange(num_keys, dtype=np.uint32)
        vals = make_leaf_values(keys)
        order = np.random.default_rng(3).permutation(num_keys)  # Set order must not matter
        t.set_many(keys[order], vals[order])
        t.build()
        assert np.array_equal(t.get_many(keys), vals)
        root = t.root()
        # same tree through the dense one-shot entry point and through the oracle
        dense_root, _ = zk.merkle_build(vals, DEPTH, O.fr_to_be(nil_account_hash())[0])
        assert root == dense_root.tobytes()
        proofs = t.get_proofs(keys)
        assert zkpor.verify_proofs(zk, root, keys, proofs, vals, DEPTH).all()
        sample = [0, 1, num_keys // 2, num_keys - 1]
        want_root, want_proofs = oracle_tree(keys, vals, sample)
        assert root == want_root
        assert np.array_equal(proofs[sample], want_proofs)
    finally:
        t.close()


def test_rebuild_after_more_sets(zk):
    """Set -> Build -> Set -> Build: correct though not incremental in the reference (merkletree.go:21-23)"""
    t = new_test_tree(zk, 5000)
    try:
        k1 = [3, 4, 70, 4099]
        v1 = make_leaf_values(k1)
        t.set_many(k1, v1)
        t.build()
        assert t.root() == oracle_tree(k1, v1, [0])[0]
        k2 = [4, 71, 2048]                                       # overwrite key 4, add neighbours
        v2 = make_leaf_values([1000, 1001, 1002])
        t.set_many(k2, v2)
        t.build()
        keys = [3, 70, 4099, 4, 71, 2048]
        vals = np.concatenate([v1[[0, 2, 3]], v2])
        want_root, want_proofs = oracle_tree(keys, vals, keys)
        assert t.root() == want_root
        assert np.array_equal(t.get_proofs(keys), want_proofs)
    finally:
        t.close()


def test_capacity_checks(zk):
    nil = bytes(32)
    with pytest.raises(zkpor.ZkporError):                        # TestCapacityOverflowCheck: depth 4, capacity 17
        zkpor.FixedDepthMerkleTree(zk, 4, nil, 17)
    with pytest.raises(zkpor.ZkporError):                        # depth > 32 / depth <= 0 panic in the reference
        zkpor.FixedDepthMerkleTree(zk, 33, nil, 1)
    with pytest.raises(zkpor.ZkporError):
        zkpor.FixedDepthMerkleTree(zk, 0, nil, 1)
    t = zkpor.FixedDepthMerkleTree(zk, 4, nil, 16)               # TestCapacityAtMax
    try:
        vals = make_leaf_values(range(16))
        t.set_many(np.arange(16, dtype=np.uint32), vals)
        with pytest.raises(zkpor.ZkporError):                    # Set's error: key out of range for capacity
            t.set(16, vals[0].tobytes())
        t.build()
        root, _, _ = O.merkle_build(O.fr_from_be(vals), 4, O.fr_from_be(np.zeros((1, 32), np.uint8))[0])
        assert t.root() == O.fr_to_be(root)[0].tobytes()
        with pytest.raises(zkpor.ZkporError):                    # GetProof's error: key out of range for depth
            t.get_proof(16)
    finally:
        t.close()


def test_set_range_from_device_leaves(zk):
    """the witness service's pattern (src/witness/main.go:130-199): leaves produced on the device go straight into the tree"""
    n = 3000
    leaves = O.fr_from_ints(list(range(1, n + 1)))
    buf = zk.alloc(n * 32).upload(leaves)
    t = new_test_tree(zk, 4096)
    try:
        t.set_range_dev(0, buf.ptr, n)
        t.build()
        root, _, _ = O.merkle_build(leaves, DEPTH, nil_account_hash())
        assert t.root() == O.fr_to_be(root)[0].tobytes()
        assert t.get(n - 1) == O.fr_to_be(leaves[n - 1:n])[0].tobytes()
        assert t.get(n) == t.nil_hash(0)
    finally:
        t.close()
        buf.free()


def test_cpp_host_mirror_driver(zk):
    """host/merkle_tree.hpp (the C++ mirror of merkletree.FixedDepthMerkleTree / utils.NewAccountTree) driven by
    tests/hostlib/tree_driver.cpp: NewAccountTree -> Set -> Build -> Root/GetProof/VerifyMerkleProof"""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostlib", "tree_driver")
    assert os.path.exists(exe), "tests/hostlib/tree_driver is not built (run __graft_entry__.build())"
    keys = [0, 5, 100, 50000]
    nil_hex = O.fr_to_be(nil_account_hash())[0].tobytes().hex()
    res = subprocess.run([exe, nil_hex, "100000"] + [str(k) for k in keys], capture_output=True, text=True, timeout=120)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.strip().splitlines()
    want_root, want_proofs = oracle_tree(keys, make_leaf_values(keys), keys)
    assert lines[0] == "root " + want_root.hex()
    for i, k in enumerate(keys):
        parts = lines[1 + i].split()
        assert parts[0] == "proof" and int(parts[1]) == k
        assert [bytes.fromhex(h) for h in parts[2:]] == [p.tobytes() for p in want_proofs[i]]
    assert lines[-1] == "verify 1"


def test_accounts_straight_into_the_tree(zk):
    """zkpor_tree_set_accounts: totals -> leaf hashes -> Set without the leaves leaving the device (src/witness/main.go:130-199),
    in two chunks, equal to the step-by-step path and to the oracle"""
    import cex_cases as C
    rng = np.random.default_rng(21)
    n_acc, n_cex, tier = 700, 60, 50
    consts = C.make_assets(n_cex, seed=9)
    consts["base_price"] = rng.integers(1, 1 << 50, size=n_cex, dtype=np.uint64)
    acc = np.zeros(n_acc, dtype=zkpor.ACCOUNT_DTYPE)
    k = rng.integers(0, 21, size=n_acc)
    off = np.concatenate([[0], np.cumsum(k)[:-1]])
    acc["n_assets"] = k; acc["asset_off"] = off
    acc["id_be"] = rng.integers(0, 256, size=(n_acc, 32), dtype=np.uint8); acc["id_be"][:, 0] &= 0x0F
    assets = np.zeros(int(k.sum()), dtype=zkpor.ASSET_DTYPE)
    for i in range(n_acc):
        assets["index"][off[i]:off[i] + k[i]] = np.sort(rng.choice(n_cex, size=k[i], replace=False))
    eq = rng.integers(0, 1 << 44, size=assets.shape[0], dtype=np.uint64)
    assets["equity"] = eq; assets["debt"] = rng.integers(0, 1 << 30, size=assets.shape[0], dtype=np.uint64)
    assets["loan"] = eq // np.uint64(3); assets["margin"] = eq // np.uint64(6); assets["portfolio_margin"] = eq // np.uint64(12)
    t = new_test_tree(zk, 1024)
    try:
        half = n_acc // 2
        a0, v0 = t.set_accounts(0, acc[:half], assets, tier, consts)
        a1, v1 = t.set_accounts(half, acc[half:], assets, tier, consts)
        t.build()
        filled = np.concatenate([a0, a1])
        ref, ref_valid = O.account_totals(acc, assets, consts)
        for field in ("equity", "debt", "collateral"):
            assert np.array_equal(filled[field], ref[field])
        assert np.array_equal(np.concatenate([v0, v1]), ref_valid)
        leaves = O.account_leaves(ref, assets, tier)
        want_root, _, _ = O.merkle_build(leaves, DEPTH, nil_account_hash())
        assert t.root() == O.fr_to_be(want_root)[0].tobytes()
        assert np.array_equal(t.get_many(np.arange(n_acc, dtype=np.uint32)), O.fr_to_be(leaves))
    finally:
        t.close()


def test_empty_capacity_tree(zk):
    """capacity 0: nothing can be Set; root, Get and GetProof all read nil hashes"""
    t = new_test_tree(zk, 0)
    try:
        with pytest.raises(zkpor.ZkporError):
            t.set(0, make_leaf_values([0])[0].tobytes())
        t.build()
        assert t.root() == t.nil_hash(28)
        assert t.get(0) == t.nil_hash(0)
        proof = t.get_proof(12345)
        assert proof == [t.nil_hash(l) for l in range(28)]
    finally:
        t.close()
