"""-m gpu: the Groth16 prove tail on the device against the oracle's trapdoor-known setup.
 - bit-exact: device proof points == oracle proof points for pinned (r, s)
 - semantic: the device proof satisfies the Groth16 verification equation, checked on discrete logs
   (the stand-in for `groth16.Verify`, src/prover/prover/prover.go:276, until a box with Go runs the real one)
 - drop-in details: gnark's compacted A/B/K arrays + infinity masks, Z in natural or bit-reversed order,
   raw proof encoding (proof.WriteRawTo, prover.go:201)."""
import numpy as np
import pytest

import oracle as O
import zkpor

pytestmark = pytest.mark.gpu


def _load_pk(zk, S, z_order, knock_out=()):
    """feed the oracle's wire-indexed key through the C ABI the way gnark holds it (compacted + masks)"""
    pk = zkpor.ProvingKey(zk)
    nw = S.n_wires
    A = S.A.copy(); B1 = S.B1.copy(); B2 = S.B2.copy()
    inf_a = np.array([not A[i].any() for i in range(nw)], dtype=np.uint8)
    inf_b = np.array([not B1[i].any() for i in range(nw)], dtype=np.uint8)
    pk.set_g1(zkpor.G1_A, A[inf_a == 0])
    pk.set_g1(zkpor.G1_B, B1[inf_b == 0])
    pk.set_g2(zkpor.G2_B, B2[inf_b == 0])
    pk.set_g1(zkpor.G1_K, S.K[S.n_public:])
    pk.set_g1(zkpor.G1_Z, S.Z)
    pk.set_g1(zkpor.G1_COMMIT_BASIS, np.zeros((0, 8), np.uint64))
    pk.set_g1(zkpor.G1_COMMIT_BASIS_SIGMA, np.zeros((0, 8), np.uint64))
    pk.set_consts(S.abd1[0], S.abd1[1], S.abd1[2], S.bd2[0], S.bd2[1], S.log2d, inf_a, inf_b, nw, S.n_public, None, z_order)
    return pk


@pytest.mark.parametrize("n_cons,z_bitrev", [(7, True), (300, True), (300, False), (2000, True)])
def test_prove_tail_matches_oracle_and_verifies(zk, n_cons, z_bitrev):
    S = O.Synth(6, n_cons, n_public=2, seed=11 + n_cons, z_bitrev=z_bitrev)
    pk = _load_pk(zk, S, zkpor.Z_ORDER_BITREV if z_bitrev else zkpor.Z_ORDER_NATURAL)
    try:
        r = O.fr_random(5, 1)[0]; s = O.fr_random(6, 1)[0]
        got = zk.prove_tail(pk, S.w, S.a, S.b, S.c, r, s)
        ref = S.prove_tail(r, s)
        assert np.array_equal(got, ref)            # bit-exact with the CPU restatement
        assert S.check(r, s, got)                  # Groth16 equation holds (in the exponent)
        assert S.verify_pairing(got)               # ... and under a real pairing, from vk + public wires + proof only
        bad = got.copy(); bad[5] ^= 1
        assert not S.check(r, s, bad)
        swapped = got.copy(); swapped[192:256] = got[0:64]   # Krs := Ar (a point on the curve, wrong value)
        assert not S.verify_pairing(swapped)
        # a different blinding gives a different but still valid proof
        r2 = O.fr_random(7, 1)[0]
        got2 = zk.prove_tail(pk, S.w, S.a, S.b, S.c, r2, s)
        assert not np.array_equal(got2, got) and S.check(r2, s, got2) and S.verify_pairing(got2)
        # raw encoding = oracle's restatement of WriteRawTo (first 256 bytes), then u32 0 commitments + empty pok
        raw = zkpor.proof_write_raw(got)
        assert np.array_equal(raw[:256], O.proof_raw(got)) and raw.size == 324 and not raw[256:260].any()
    finally:
        pk.close()


def test_pk_rejects_inconsistent_lengths(zk):
    S = O.Synth(4, 20, n_public=2, seed=3)
    pk = zkpor.ProvingKey(zk)
    try:
        pk.set_g1(zkpor.G1_A, S.A[:-1])
        pk.set_g1(zkpor.G1_B, S.B1); pk.set_g2(zkpor.G2_B, S.B2)
        pk.set_g1(zkpor.G1_K, S.K[S.n_public:]); pk.set_g1(zkpor.G1_Z, S.Z)
        with pytest.raises(zkpor.ZkporError):
            pk.set_consts(S.abd1[0], S.abd1[1], S.abd1[2], S.bd2[0], S.bd2[1], S.log2d, None, None, S.n_wires, S.n_public)
        with pytest.raises(zkpor.ZkporError):
            zk.prove_tail(pk, S.w, S.a, S.b, S.c, O.fr_random(1, 1)[0], O.fr_random(2, 1)[0])
    finally:
        pk.close()


def test_commit_matches_two_msms(zk):
    n = 500
    S = O.Synth(4, 20, n_public=2, seed=4)
    pk = zkpor.ProvingKey(zk)
    try:
        basis = O.g1_from_scalars(O.fr_random(21, n)); sigma = O.g1_from_scalars(O.fr_random(22, n))
        pk.set_g1(zkpor.G1_A, S.A); pk.set_g1(zkpor.G1_B, S.B1); pk.set_g2(zkpor.G2_B, S.B2)
        pk.set_g1(zkpor.G1_K, S.K[S.n_public:]); pk.set_g1(zkpor.G1_Z, S.Z)
        pk.set_g1(zkpor.G1_COMMIT_BASIS, basis); pk.set_g1(zkpor.G1_COMMIT_BASIS_SIGMA, sigma)
        inf_a = np.array([not S.A[i].any() for i in range(S.n_wires)], dtype=np.uint8)
        # the oracle key has no infinity points in A/B except by chance; pass explicit all-false masks
        pk.set_consts(S.abd1[0], S.abd1[1], S.abd1[2], S.bd2[0], S.bd2[1], S.log2d, inf_a * 0, inf_a * 0, S.n_wires, S.n_public)
        vals = O.fr_random(23, n)
        c, k = zk.commit(pk, vals)
        assert np.array_equal(c, O.g1_msm(basis, vals)) and np.array_equal(k, O.g1_msm(sigma, vals))
    finally:
        pk.close()


def test_commitment_proof_of_knowledge_verifies_under_pairing(zk):
    """BSB22 commitment as the verifier sees it (gnark-crypto fr/pedersen VerifyingKey.Verify, run inside groth16.Verify,
    prover.go:276): BasisExpSigma_i = sigma * Basis_i, and e(commitment, sigma*G2) == e(pok, G2)"""
    n = 300
    S = O.Synth(4, 20, n_public=2, seed=4)
    pk = zkpor.ProvingKey(zk)
    try:
        bs = O.fr_random(31, n)
        sig = O.fr_random(32, 1)[0]
        basis = O.g1_from_scalars(bs)
        basis_sigma = O.g1_from_scalars(O.fr_mul(bs, np.repeat(sig[None, :], n, axis=0)))
        pk.set_g1(zkpor.G1_A, S.A); pk.set_g1(zkpor.G1_B, S.B1); pk.set_g2(zkpor.G2_B, S.B2)
        pk.set_g1(zkpor.G1_K, S.K[S.n_public:]); pk.set_g1(zkpor.G1_Z, S.Z)
        pk.set_g1(zkpor.G1_COMMIT_BASIS, basis); pk.set_g1(zkpor.G1_COMMIT_BASIS_SIGMA, basis_sigma)
        z = np.zeros(S.n_wires, dtype=np.uint8)
        pk.set_consts(S.abd1[0], S.abd1[1], S.abd1[2], S.bd2[0], S.bd2[1], S.log2d, z, z, S.n_wires, S.n_public)
        vals = O.fr_random(33, n)
        c, k = zk.commit(pk, vals)
        g2s = O.g2_mul_gen(sig)
        assert O.pedersen_verify_pairing(c, k, g2s)
        assert not O.pedersen_verify_pairing(c, c, g2s)
        # the 388-byte wire form with one commitment round-trips the same points
        proof = zk.prove_tail(pk, S.w, S.a, S.b, S.c, O.fr_random(5, 1)[0], O.fr_random(6, 1)[0])
        raw = zkpor.proof_write_raw(proof, commitments=c[None, :], pok=k)
        assert raw.size == 388 and raw[256:260].tolist() == [0, 0, 0, 1]
    finally:
        pk.close()


def test_synth_key_trapdoor(zk):
    """ProvingKey.synth: points are on the curve and equal (k + j*q)*G; infinity pattern as documented"""
    pk = zkpor.ProvingKey(zk)
    try:
        n = 3000
        pk.synth(10, n, 3, 100, seed=99)
        for which in (zkpor.G1_A, zkpor.G1_K, zkpor.G1_Z):
            ptr, cnt = pk.g1_dev(which)
            buf = zkpor.DevBuf.__new__(zkpor.DevBuf); buf.ctx = zk; buf.nbytes = cnt * 64; buf.ptr = ptr
            pts = buf.download(np.uint64, (cnt, 8))
            assert O.g1_on_curve(pts)
            sc = O.fr_from_ints([zkpor.synth_scalar(99, which, i) for i in range(cnt)])
            exp = O.g1_from_scalars(sc)
            for i in range(cnt):
                if (which == zkpor.G1_K and i < 3) or zkpor.synth_is_inf(i, zkpor.SYNTH_INF_MOD[which]):
                    exp[i] = 0
            assert np.array_equal(pts, exp)
        ptr, cnt = pk.g2_dev()
        buf = zkpor.DevBuf.__new__(zkpor.DevBuf); buf.ctx = zk; buf.nbytes = cnt * 128; buf.ptr = ptr
        p2 = buf.download(np.uint64, (cnt, 16))
        assert O.g2_on_curve(p2)
        sc = O.fr_from_ints([zkpor.synth_scalar(99, zkpor.G1_B, i) for i in range(cnt)])
        exp2 = O.g2_from_scalars(sc)
        for i in range(cnt):
            if zkpor.synth_is_inf(i, 10):
                exp2[i] = 0
        assert np.array_equal(p2, exp2)
    finally:
        pk.close()
