"""CPU suite: the host-side walk of gnark's proving-key container (zkpor_pk_gnark_layout, csrc/keyfile.hip)
on streams produced by tests/gnark_keyfile.py from the oracle's synthetic key — section counts and offsets, both domain-header
variants, and the rejections (every count in the stream is cross-checked).  No device is needed to walk the headers; loading
the arrays is the -m gpu test (test_keyfile_gpu.py).  Also the one size the reference publishes: the 524-byte verifying key."""
import numpy as np
import pytest

import gnark_keyfile as GK
import oracle as O
import zkpor


@pytest.fixture(scope="module")
def synth():
    return O.Synth(5, 40, n_public=2, seed=23)


@pytest.mark.parametrize("flag_byte,z_full", [(True, True), (True, False), (False, True)])
def test_layout_counts_and_offsets(synth, flag_byte, z_full):
    S = synth
    basis = O.g1_from_scalars(O.fr_random(1, 7)); sigma = O.g1_from_scalars(O.fr_random(2, 7))
    data, inf_a, inf_b = GK.pk_bytes_from_synth(S, [(basis, sigma)], with_precompute_byte=flag_byte, z_full_domain=z_full)
    L = zkpor.pk_gnark_layout(data)
    D = 1 << S.log2d
    assert L["domain_cardinality"] == D and L["domain_header_bytes"] == (169 if flag_byte else 168)
    assert L["n_wires"] == S.n_wires and L["n_inf_a"] == int(inf_a.sum()) and L["n_inf_b"] == int(inf_b.sum())
    assert L["n_a"] == S.n_wires - int(inf_a.sum()) and L["n_b1"] == L["n_b2"] == S.n_wires - int(inf_b.sum())
    assert L["n_z"] == (D if z_full else D - 1) and L["n_k"] == S.n_wires - S.n_public
    assert L["n_commitment_keys"] == 1 and L["n_basis"] == L["n_basis_sigma"] == 7
    assert L["bytes_total"] == len(data)
    buf = np.frombuffer(data, dtype=np.uint8)
    sect = lambda off, n, w: buf[off:off + n * w].reshape(n, w)
    assert np.array_equal(sect(L["off_alpha"], 3, 32), O.g1_compress(S.abd1))
    assert np.array_equal(sect(L["off_a"], L["n_a"], 32), O.g1_compress(S.A[inf_a == 0]))
    assert np.array_equal(sect(L["off_b1"], L["n_b1"], 32), O.g1_compress(S.B1[inf_b == 0]))
    assert np.array_equal(sect(L["off_z"], D - 1, 32), O.g1_compress(S.Z))
    assert np.array_equal(sect(L["off_k"], L["n_k"], 32), O.g1_compress(S.K[S.n_public:]))
    assert np.array_equal(sect(L["off_beta2"], 2, 64), O.g2_compress(S.bd2))
    assert np.array_equal(sect(L["off_b2"], L["n_b2"], 64), O.g2_compress(S.B2[inf_b == 0]))
    assert np.array_equal(buf[L["off_inf_a"]:L["off_inf_a"] + S.n_wires], inf_a)
    assert np.array_equal(buf[L["off_inf_b"]:L["off_inf_b"] + S.n_wires], inf_b)
    assert np.array_equal(sect(L["off_basis"], 7, 32), O.g1_compress(basis))
    assert np.array_equal(sect(L["off_basis_sigma"], 7, 32), O.g1_compress(sigma))
    # the section offsets follow from the framing alone: header, 3 points, then u32-prefixed slices back to back
    assert L["off_alpha"] == L["domain_header_bytes"] and L["off_a"] == L["off_alpha"] + 96 + 4
    assert L["off_b1"] == L["off_a"] + 32 * L["n_a"] + 4 and L["off_beta2"] == L["off_k"] + 32 * L["n_k"]
    assert L["off_inf_a"] == L["off_b2"] + 64 * L["n_b2"] + 24 and L["off_inf_b"] == L["off_inf_a"] + S.n_wires


def test_no_commitment_key(synth):
    data, _, _ = GK.pk_bytes_from_synth(synth)
    L = zkpor.pk_gnark_layout(data)
    assert L["n_commitment_keys"] == 0 and L["n_basis"] == 0 and L["bytes_total"] == len(data)


def test_rejections(synth):
    S = synth
    data, inf_a, inf_b = GK.pk_bytes_from_synth(S, [(O.g1_from_scalars(O.fr_random(1, 3)), O.g1_from_scalars(O.fr_random(2, 3)))])
    L = zkpor.pk_gnark_layout(data)
    with pytest.raises(zkpor.ZkporError, match="not a gnark bn254 Groth16 proving key"):
        zkpor.pk_gnark_layout(data[:-1])                                    # truncated
    with pytest.raises(zkpor.ZkporError, match="does not end after the last commitment key"):
        zkpor.pk_gnark_layout(data + b"\x00")                               # trailing bytes
    with pytest.raises(zkpor.ZkporError, match="truncated"):
        zkpor.pk_gnark_layout(data[:100])
    m = bytearray(data); m[L["off_inf_a"] + 3] ^= 1                         # mask no longer adds up to NbInfinityA
    with pytest.raises(zkpor.ZkporError, match="infinity masks do not add up"):
        zkpor.pk_gnark_layout(bytes(m))
    m = bytearray(data); m[L["off_inf_b"]] = 2
    with pytest.raises(zkpor.ZkporError, match="not 0/1"):
        zkpor.pk_gnark_layout(bytes(m))
    m = bytearray(data); m[0:8] = (3 << 10).to_bytes(8, "big")              # cardinality not a power of two
    with pytest.raises(zkpor.ZkporError, match="power of two"):
        zkpor.pk_gnark_layout(bytes(m))
    # NbInfinityA off by one (field sits 16 bytes before the masks)
    m = bytearray(data); o = L["off_inf_a"] - 16; m[o:o + 8] = (L["n_inf_a"] + 1).to_bytes(8, "big")
    with pytest.raises(zkpor.ZkporError, match=r"len\(A\) \+ NbInfinityA != nbWires"):
        zkpor.pk_gnark_layout(bytes(m))
    # G2.B shorter than G1.B: re-frame with one point fewer
    short = GK.pk_bytes(S.log2d, S.abd1[0], S.abd1[1], S.abd1[2], S.A[inf_a == 0], S.B1[inf_b == 0], S.Z, S.K[S.n_public:],
                        S.bd2[0], S.bd2[1], S.B2[inf_b == 0][:-1], inf_a, inf_b)
    with pytest.raises(zkpor.ZkporError, match="G1.B and G2.B differ"):
        zkpor.pk_gnark_layout(short)
    with pytest.raises(zkpor.ZkporError):
        zkpor.pk_gnark_layout(b"")


def test_verifying_key_is_524_bytes_for_the_reference_shape():
    """README.md:54: `524 ... zkpor500_200.vk`, `524 ... zkpor50_1380.vk` — the only byte count of a key container the
    reference publishes; it pins the vk framing (fixed points, K with 3 entries, one commitment, one Pedersen key)"""
    g1 = O.g1_from_scalars(O.fr_random(3, 6)); g2 = O.g2_from_scalars(O.fr_random(4, 5))
    vk = GK.vk_bytes(g1[0], g1[1], g2[0], g2[1], g1[2], g2[2], g1[3:6], [[]], g2[3], g2[4])
    assert len(vk) == 524


def test_layout_walk_survives_corrupted_streams(synth):
    """random truncations, byte flips and spliced length fields: the walk either accepts a stream whose counts are all
    consistent or rejects it with a message — it never reads outside the buffer (the process would not survive that)"""
    rng = np.random.default_rng(11)
    data, _, _ = GK.pk_bytes_from_synth(synth, [(O.g1_from_scalars(O.fr_random(1, 3)), O.g1_from_scalars(O.fr_random(2, 3)))])
    base = zkpor.pk_gnark_layout(data)
    accepted = rejected = 0
    for trial in range(1500):
        m = bytearray(data)
        kind = trial % 4
        if kind == 0:
            m = m[:int(rng.integers(0, len(m)))]
        elif kind == 1:
            for _ in range(int(rng.integers(1, 4))):
                m[int(rng.integers(0, len(m)))] ^= 1 << int(rng.integers(0, 8))
        elif kind == 2:                                   # overwrite one of the length / count fields with a huge value
            off = [8 * 0, base["off_a"] - 4, base["off_b1"] - 4, base["off_z"] - 4, base["off_k"] - 4, base["off_b2"] - 4,
                   base["off_inf_a"] - 24, base["off_inf_a"] - 16, base["off_inf_a"] - 8][int(rng.integers(0, 9))]
            m[off:off + 4] = int(rng.integers(0, 1 << 32)).to_bytes(4, "big")
        else:
            m += bytes(int(rng.integers(1, 64)))
        try:
            L = zkpor.pk_gnark_layout(bytes(m))
            accepted += 1
            assert L["bytes_total"] == len(m) and L["n_a"] + L["n_inf_a"] == L["n_wires"]
        except zkpor.ZkporError:
            rejected += 1
    assert rejected > 1000 and accepted + rejected == 1500
