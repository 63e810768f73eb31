"""Test infrastructure: WRITERS for gnark's Groth16 key containers, used to produce streams the product's reader
(csrc/keyfile.hip, zkpor_pk_load_gnark*) is tested on.  The product only reads these containers; writing them is what
gnark's keygen does (pk.WriteTo / vk.WriteTo, src/keygen/main.go:46,55).

Layout: gnark v0.10 backend/groth16/bn254/marshal.go + gnark-crypto v0.14 (fft.Domain.WriteTo, pedersen keys, Encoder), the
versions pinned at go.mod:57-60 — third-party code absent from /root/reference, restated from the published sources.
The one fact the reference does hold: a verifying key of this circuit is 524 bytes (README.md:54), which pins the vk
framing (see vk_bytes)."""
import numpy as np

import oracle as O

ROOT_2_28 = 19103219067921713944291392827692070036145651957329286315305642004821462161904  # gnark-crypto fr: 2^28-th root of unity
COSET_GEN = 5                                                                                # fr multiplicative generator


def _be(v, n):
    return int(v).to_bytes(n, "big")


def _g1s(pts):
    pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, 8)
    return _be(pts.shape[0], 4) + (O.g1_compress(pts).tobytes() if pts.shape[0] else b"")


def _g2s(pts):
    pts = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1, 16)
    return _be(pts.shape[0], 4) + (O.g2_compress(pts).tobytes() if pts.shape[0] else b"")


def domain_bytes(log2d, with_precompute_byte=True):
    """fft.Domain.WriteTo: Cardinality u64, then CardinalityInv, Generator, GeneratorInv, FrMultiplicativeGen,
    FrMultiplicativeGenInv as 32-byte big-endian canonical values (+ the withPrecompute bool of gnark-crypto >= v0.12)"""
    r = O.R_MOD
    card = 1 << log2d
    gen = pow(ROOT_2_28, 1 << (28 - log2d), r)
    out = _be(card, 8)
    for v in (pow(card, -1, r), gen, pow(gen, -1, r), COSET_GEN, pow(COSET_GEN, -1, r)):
        out += _be(v, 32)
    return out + (b"\x01" if with_precompute_byte else b"")


def pk_bytes(log2d, alpha, beta, delta, A, B1, Z, K, beta2, delta2, B2, inf_a, inf_b, commitment_keys=(), with_precompute_byte=True):
    """pk.WriteTo: A/B1/B2 compacted (infinity wires absent), K without public/committed wires, masks one byte per wire"""
    inf_a = np.ascontiguousarray(inf_a, dtype=np.uint8); inf_b = np.ascontiguousarray(inf_b, dtype=np.uint8)
    one = lambda p: O.g1_compress(np.asarray(p, dtype=np.uint64).reshape(1, 8)).tobytes()
    two = lambda p: O.g2_compress(np.asarray(p, dtype=np.uint64).reshape(1, 16)).tobytes()
    out = domain_bytes(log2d, with_precompute_byte)
    out += one(alpha) + one(beta) + one(delta)
    out += _g1s(A) + _g1s(B1) + _g1s(Z) + _g1s(K)
    out += two(beta2) + two(delta2) + _g2s(B2)
    out += _be(inf_a.size, 8) + _be(int(inf_a.sum()), 8) + _be(int(inf_b.sum()), 8)
    out += inf_a.tobytes() + inf_b.tobytes()
    out += _be(len(commitment_keys), 4)
    for basis, basis_sigma in commitment_keys:
        out += _g1s(basis) + _g1s(basis_sigma)
    return out


def pk_bytes_from_synth(S, commitment_keys=(), with_precompute_byte=True, z_full_domain=True, seed=0):
    """the oracle's wire-indexed synthetic key as gnark would have written it; returns (stream, inf_a, inf_b)"""
    nw = S.n_wires
    inf_a = np.array([not S.A[i].any() for i in range(nw)], dtype=np.uint8)
    inf_b = np.array([not S.B1[i].any() for i in range(nw)], dtype=np.uint8)
    Z = S.Z
    if z_full_domain:   # gnark keeps Cardinality points in the slice; the prover uses the first Cardinality - 1
        Z = np.concatenate([S.Z, O.g1_from_scalars(O.fr_random(900 + seed, 1))])
    data = pk_bytes(S.log2d, S.abd1[0], S.abd1[1], S.abd1[2], S.A[inf_a == 0], S.B1[inf_b == 0], Z, S.K[S.n_public:],
                    S.bd2[0], S.bd2[1], S.B2[inf_b == 0], inf_a, inf_b, commitment_keys, with_precompute_byte)
    return data, inf_a, inf_b


def vk_bytes(alpha1, beta1, beta2, gamma2, delta1, delta2, K, public_and_commitment_committed, pedersen_g, pedersen_g_sigma_neg):
    """vk.WriteTo (gnark v0.10): G1.Alpha, G1.Beta, G2.Beta, G2.Gamma, G1.Delta, G2.Delta, G1.K (u32 + points),
    PublicAndCommitmentCommitted ([][]uint64: u32 outer length, per entry u32 length + u64s), then the single Pedersen
    verifying key (G, GSigmaNeg in G2).  For the reference's circuit — 3 points in K (ONE wire, BatchCommitment, the commitment
    wire), one commitment with no public committed wires — that is 288 + (4 + 3*32) + (4 + 4) + 128 = 524 bytes, the size
    README.md:54 lists for zkpor500_200.vk and zkpor50_1380.vk."""
    one = lambda p: O.g1_compress(np.asarray(p, dtype=np.uint64).reshape(1, 8)).tobytes()
    two = lambda p: O.g2_compress(np.asarray(p, dtype=np.uint64).reshape(1, 16)).tobytes()
    out = one(alpha1) + one(beta1) + two(beta2) + two(gamma2) + one(delta1) + two(delta2)
    out += _g1s(K)
    out += _be(len(public_and_commitment_committed), 4)
    for row in public_and_commitment_committed:
        out += _be(len(row), 4) + b"".join(_be(v, 8) for v in row)
    out += two(pedersen_g) + two(pedersen_g_sigma_neg)
    return out
