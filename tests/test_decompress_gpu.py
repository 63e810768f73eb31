"""-m gpu: device-side decompression of gnark-crypto's compressed BN254 points (the work
pk.UnsafeReadFrom does on the CPU at src/prover/prover/prover.go:336-349) against the oracle's restatement of
ecc/bn254/marshal.go: bit-exact affine points for both root choices and infinity, rejection of malformed input, and a
proving key loaded from compressed arrays giving the same proof as one loaded from affine arrays."""
import numpy as np
import pytest

import oracle as O
import zkpor

pytestmark = pytest.mark.gpu


def test_generator_bytes():
    """hand-checkable: G1 generator (1, 2): y = 2 is the smaller root -> flag 10, X = 1"""
    g = O.g1_from_scalars(O.fr_from_ints([1]))
    assert O.g1_compress(g)[0].tobytes() == bytes([0x80] + [0] * 30 + [1])
    neg = O.g1_from_scalars(O.fr_from_ints([O.R_MOD - 1]))          # -G = (1, p - 2): the larger root
    assert O.g1_compress(neg)[0].tobytes() == bytes([0xC0] + [0] * 30 + [1])


@pytest.mark.parametrize("n", [1, 2, 257, 5000])
def test_g1_roundtrip_matches_oracle(zk, n):
    pts = O.g1_from_scalars(O.fr_random(100 + n, n))
    if n > 2:
        pts[n // 3] = 0                                              # infinity in the middle
    comp = O.g1_compress(pts)
    assert set(np.unique(comp[:, 0] & 0xC0)) <= {0x40, 0x80, 0xC0}
    got = zk.g1_decompress(comp)
    rc, ref = O.g1_decompress(comp)
    assert rc == 0 and np.array_equal(ref, pts)
    assert np.array_equal(got, pts)


@pytest.mark.parametrize("n", [1, 2, 300, 3000])
def test_g2_roundtrip_matches_oracle(zk, n):
    pts = O.g2_from_scalars(O.fr_random(200 + n, n))
    if n > 2:
        pts[n // 2] = 0
    comp = O.g2_compress(pts)
    got = zk.g2_decompress(comp)
    rc, ref = O.g2_decompress(comp)
    assert rc == 0 and np.array_equal(ref, pts)
    assert np.array_equal(got, pts)
    if n >= 300:                                                     # both flag values occur in a sample this size
        assert {0x80, 0xC0} <= set(np.unique(comp[:, 0] & 0xC0))


def test_rejects_malformed_input(zk):
    pts = O.g1_from_scalars(O.fr_random(7, 8))
    comp = O.g1_compress(pts)
    # an X that is not on the curve: walk the low byte until the oracle says so
    bad = comp.copy()
    for d in range(1, 200):
        bad[3, 31] = (comp[3, 31] + d) & 0xFF
        if O.g1_decompress(bad[3:4])[0] == 3:
            break
    else:
        pytest.fail("no off-curve X found")
    with pytest.raises(zkpor.ZkporError, match="element 3 is not on the curve"):
        zk.g1_decompress(bad)
    # X >= p
    big = comp.copy(); big[5, :] = 0xFF; big[5, 0] = 0xBF
    assert O.g1_decompress(big[5:6])[0] == 2
    with pytest.raises(zkpor.ZkporError, match="element 5 has a coordinate >= p"):
        zk.g1_decompress(big)
    # flag bits 00 = an uncompressed point, which is not what this stream holds
    unc = comp.copy(); unc[0, 0] &= 0x3F
    assert O.g1_decompress(unc[0:1])[0] == 1
    with pytest.raises(zkpor.ZkporError, match="element 0 is not a compressed point"):
        zk.g1_decompress(unc)
    # G2: off-curve
    p2 = O.g2_compress(O.g2_from_scalars(O.fr_random(8, 4)))
    bad2 = p2.copy()
    for d in range(1, 200):
        bad2[1, 63] = (p2[1, 63] + d) & 0xFF
        if O.g2_decompress(bad2[1:2])[0] == 3:
            break
    with pytest.raises(zkpor.ZkporError, match="element 1 is not on the curve"):
        zk.g2_decompress(bad2)


def test_proving_key_from_compressed_arrays(zk):
    S = O.Synth(6, 300, n_public=2, seed=17)
    nw = S.n_wires
    z = np.zeros(nw, dtype=np.uint8)
    r = O.fr_random(5, 1)[0]; s = O.fr_random(6, 1)[0]
    proofs = []
    for compressed in (False, True):
        pk = zkpor.ProvingKey(zk)
        try:
            if compressed:
                pk.set_g1_compressed(zkpor.G1_A, O.g1_compress(S.A)); pk.set_g1_compressed(zkpor.G1_B, O.g1_compress(S.B1))
                pk.set_g2_compressed(zkpor.G2_B, O.g2_compress(S.B2))
                pk.set_g1_compressed(zkpor.G1_K, O.g1_compress(S.K[S.n_public:])); pk.set_g1_compressed(zkpor.G1_Z, O.g1_compress(S.Z))
            else:
                pk.set_g1(zkpor.G1_A, S.A); pk.set_g1(zkpor.G1_B, S.B1); pk.set_g2(zkpor.G2_B, S.B2)
                pk.set_g1(zkpor.G1_K, S.K[S.n_public:]); pk.set_g1(zkpor.G1_Z, S.Z)
            pk.set_consts(S.abd1[0], S.abd1[1], S.abd1[2], S.bd2[0], S.bd2[1], S.log2d, z, z, nw, S.n_public)
            proofs.append(zk.prove_tail(pk, S.w, S.a, S.b, S.c, r, s))
        finally:
            pk.close()
    assert np.array_equal(proofs[0], proofs[1])
    assert np.array_equal(proofs[1], S.prove_tail(r, s)) and S.verify_pairing(proofs[1])
