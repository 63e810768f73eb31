"""CPU: the checker of the full-size proofs (oracle/trapdoor.py) is itself checked — its vectorised scalar generator against
the scalar Python form in zkpor.py, its dot products against the oracle's multi-exponentiation over the points those
scalars generate, and its proof prediction against the Groth16 formulas evaluated on group elements by the oracle."""
import numpy as np

import oracle as O
import trapdoor as T
import zkpor


def _mont(canon):
    out = np.empty_like(canon)
    O.lib().orc_fr_from_canon(O._p(canon), O._p(out), canon.shape[0])
    return out


def test_scalar_generator_matches_scalar_form():
    for arr, below in ((T.G1_A, 0), (T.G1_B, 0), (T.G1_K, 3), (T.G1_Z, 0)):
        c = T.synth_scalars_canon(99, arr, 0, 3000, None, below)
        ints = O.limbs_to_ints(c)
        for i in (0, 1, 2, 3, 31, 32, 33, 777, 2999):
            inf = i < below or zkpor.synth_is_inf(i, zkpor.SYNTH_INF_MOD[arr])
            assert ints[i] == (0 if inf else zkpor.synth_scalar(99, arr, i))
    # a window that does not start at 0 is the same slice
    assert np.array_equal(T.synth_scalars_canon(5, T.G1_A, 100, 900), T.synth_scalars_canon(5, T.G1_A, 0, 900)[100:])
    assert T.synth_k(7, 100, 0) == zkpor.synth_scalar(7, 100, 0)


def test_dot_equals_msm_over_generated_points():
    n, seed = 300, 1234
    w = O.fr_random(3, n)
    for arr in (T.G1_A, T.G1_K):
        pts = O.g1_from_scalars(_mont(T.synth_scalars_canon(seed, arr, 0, n, None, 2 if arr == T.G1_K else 0)))
        d = T.synth_dot(seed, arr, w, inf_below=2 if arr == T.G1_K else 0, chunk=64)     # several chunks
        assert np.array_equal(O.g1_msm(pts, w), O.g1_from_scalars(d.reshape(1, 4))[0])


def test_fused_dot_equals_the_numpy_generator():
    """orc_synth_dot (generator + dot product in one OpenMP pass, what the 2^26 checks use) against the numpy restatement, array by array,
    with and without the infinity patterns, across run boundaries and chunk edges"""
    n, seed = 100_000, 0x5A4B504F52 + 3
    x = O.fr_random(11, n)
    for arr, below in ((T.G1_A, 0), (T.G1_B, 0), (T.G1_K, 3), (T.G1_Z, 0), (T.G1_COMMIT_BASIS, 0), (T.G1_COMMIT_BASIS_SIGMA, 0)):
        for m in (n, 33, 1):
            a = T.synth_dot(seed, arr, x[:m], inf_below=below, fused=True)
            b = T.synth_dot(seed, arr, x[:m], inf_below=below, fused=False, chunk=4096)
            assert np.array_equal(a, b)
    assert np.array_equal(T.synth_dot(seed, T.G1_A, x, inf_mod=7, fused=True), T.synth_dot(seed, T.G1_A, x, inf_mod=7, fused=False))


def test_proof_prediction_matches_group_formulas():
    n, seed, npub = 256, 77, 3
    w = O.fr_random(5, n); h = O.fr_random(6, n - 1)
    r = O.fr_random(8, 1)[0]; s = O.fr_random(9, 1)[0]
    td = T.SynthKeyTrapdoor(seed, npub, w, h)
    g = lambda arr, m, below=0: O.g1_from_scalars(_mont(T.synth_scalars_canon(seed, arr, 0, m, None, below)))
    A, B1, K, Z = g(T.G1_A, n), g(T.G1_B, n), g(T.G1_K, n, npub), g(T.G1_Z, n - 1)
    B2 = O.g2_from_scalars(_mont(T.synth_scalars_canon(seed, T.G1_B, 0, n)))
    k = lambda a: O.fr_from_ints([T.synth_k(seed, a, 0)])
    alpha, beta, delta = (O.g1_from_scalars(k(a))[0] for a in (100, 101, 102))
    beta2, delta2 = (O.g2_from_scalars(k(a))[0] for a in (101, 102))
    mul1 = lambda p, x: O.g1_scalar_mul(p.reshape(1, 8), x.reshape(1, 4))[0]
    add1 = lambda p, q: O.g1_add(p.reshape(1, 8), q.reshape(1, 8))[0]
    ar = add1(add1(alpha, O.g1_msm(A, w)), mul1(delta, r))
    bs1 = add1(add1(beta, O.g1_msm(B1, w)), mul1(delta, s))
    # G2: s*delta2 through the discrete log (the oracle has no G2 scalar multiplication export)
    sd2 = O.g2_from_scalars(O.fr_mul(s.reshape(1, 4), k(102)))[0]
    bs2 = O.g2_add(O.g2_add(beta2.reshape(1, 16), O.g2_msm(B2, w).reshape(1, 16)), sd2.reshape(1, 16))[0]
    rs = O.fr_mul(r.reshape(1, 4), s.reshape(1, 4))[0]
    neg_rs = O.fr_sub(O.fr_from_ints([0]), rs.reshape(1, 4))[0]
    krs = add1(add1(O.g1_msm(K, w), O.g1_msm(Z, h)), add1(add1(mul1(ar, s), mul1(bs1, r)), mul1(delta, neg_rs)))
    ear, ebs, ekrs = td.expected(r, s)
    assert np.array_equal(ear, ar) and np.array_equal(ebs, bs2) and np.array_equal(ekrs, krs)
    proof = np.concatenate([ar, bs2, krs]).view(np.uint8)
    assert td.check(proof, r, s) and not td.check(proof, s, r)
    v = O.fr_random(10, 64)
    ec, ek = T.expected_commitment(seed, v)
    assert np.array_equal(ec, O.g1_msm(g(T.G1_COMMIT_BASIS, 64), v)) and np.array_equal(ek, O.g1_msm(g(T.G1_COMMIT_BASIS_SIGMA, 64), v))
