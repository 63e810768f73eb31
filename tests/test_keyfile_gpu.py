"""-m gpu: a proving key loaded from gnark's on-disk container (pk.WriteTo, src/keygen/main.go:46 — what pk.UnsafeReadFrom
reads at src/prover/prover/prover.go:343) proves exactly like the same key fed array by array: the reader
locates the compressed arrays in the stream, the device decompresses them, the infinity masks re-expand A / B to wire order."""
import numpy as np
import pytest

import gnark_keyfile as GK
import oracle as O
import zkpor

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("flag_byte,z_full,from_file", [(True, True, True), (False, False, False)])
def test_key_loaded_from_container_proves_like_the_oracle(zk, tmp_path, flag_byte, z_full, from_file):
    S = O.Synth(6, 300, n_public=2, seed=29)
    n = 40
    bs = O.fr_random(31, n); sig = O.fr_random(32, 1)[0]
    basis = O.g1_from_scalars(bs); basis_sigma = O.g1_from_scalars(O.fr_mul(bs, np.repeat(sig[None, :], n, axis=0)))
    data, inf_a, inf_b = GK.pk_bytes_from_synth(S, [(basis, basis_sigma)], with_precompute_byte=flag_byte, z_full_domain=z_full)
    pk = zkpor.ProvingKey(zk)
    try:
        if from_file:
            path = tmp_path / "zkpor_test.pk"
            path.write_bytes(data)
            L = pk.load_gnark(str(path), S.n_public)
        else:
            L = pk.load_gnark(data, S.n_public)
        assert L["n_wires"] == S.n_wires and L["bytes_total"] == len(data) and L["n_basis"] == n
        r = O.fr_random(5, 1)[0]; s = O.fr_random(6, 1)[0]
        proof = zk.prove_tail(pk, S.w, S.a, S.b, S.c, r, s)
        assert np.array_equal(proof, S.prove_tail(r, s)) and S.verify_pairing(proof)
        vals = O.fr_random(33, n)
        c, k = zk.commit(pk, vals)
        assert O.pedersen_verify_pairing(c, k, O.g2_mul_gen(sig))
    finally:
        pk.close()


@pytest.mark.parametrize("z_full", [True, False])
def test_z_order_of_the_file_is_the_callers_statement(zk, z_full):
    """The stream does not record whether G1.Z was bit-reversed at setup: both conventions load cleanly, only one verifies.  A key
    written in natural order proves correctly when loaded as ZKPOR_Z_ORDER_NATURAL, and the SAME bytes loaded as bit-reversed give
    a proof the pairing check rejects (why z_order is an argument of the loaders, not a constant)."""
    S = O.Synth(6, 300, n_public=2, seed=31, z_bitrev=False)
    data, _, _ = GK.pk_bytes_from_synth(S, z_full_domain=z_full)
    r = O.fr_random(5, 1)[0]; s = O.fr_random(6, 1)[0]
    pk = zkpor.ProvingKey(zk)
    try:
        pk.load_gnark(data, S.n_public, z_order=zkpor.Z_ORDER_NATURAL)
        good = zk.prove_tail(pk, S.w, S.a, S.b, S.c, r, s)
        assert np.array_equal(good, S.prove_tail(r, s)) and S.verify_pairing(good)
        pk.load_gnark(data, S.n_public, z_order=zkpor.Z_ORDER_BITREV)
        bad = zk.prove_tail(pk, S.w, S.a, S.b, S.c, r, s)
        assert not S.verify_pairing(bad)
        with pytest.raises(zkpor.ZkporError, match="unknown z_order"):
            pk.load_gnark(data, S.n_public, z_order=7)
        with pytest.raises(zkpor.ZkporError, match="bit-reversed"):
            pk.load_gnark_shard(data, S.n_public, 0, S.n_wires, 0, 4, z_order=zkpor.Z_ORDER_NATURAL)
    finally:
        pk.close()


def test_container_errors_are_reported(zk, tmp_path):
    S = O.Synth(4, 20, n_public=2, seed=4)
    data, inf_a, inf_b = GK.pk_bytes_from_synth(S)
    pk = zkpor.ProvingKey(zk)
    try:
        with pytest.raises(zkpor.ZkporError, match=r"len\(K\) = .* does not equal nbWires - n_public - n_committed"):
            pk.load_gnark(data, S.n_public + 1)
        with pytest.raises(zkpor.ZkporError, match="not a gnark bn254 Groth16 proving key"):
            pk.load_gnark(data[:-7], S.n_public)
        with pytest.raises(zkpor.ZkporError, match="cannot open"):
            pk.load_gnark(str(tmp_path / "missing.pk"), S.n_public)
        # a point of A knocked off the curve is named by section and index
        L = zkpor.pk_gnark_layout(data)
        m = bytearray(data)
        for d in range(1, 200):
            m[L["off_a"] + 2 * 32 + 31] = (data[L["off_a"] + 2 * 32 + 31] + d) & 0xFF
            if O.g1_decompress(np.frombuffer(bytes(m[L["off_a"] + 64:L["off_a"] + 96]), dtype=np.uint8).reshape(1, 32))[0] == 3:
                break
        with pytest.raises(zkpor.ZkporError, match=r"G1\.A: .*element 2 is not on the curve"):
            pk.load_gnark(bytes(m), S.n_public)
        # and the key still loads from the good stream afterwards
        pk.load_gnark(data, S.n_public)
        proof = zk.prove_tail(pk, S.w, S.a, S.b, S.c, O.fr_random(5, 1)[0], O.fr_random(6, 1)[0])
        assert S.verify_pairing(proof)
    finally:
        pk.close()


def test_key_file_loaded_as_fixed_base_tables():
    """the production shape: "msm_tables" = 4 set before zkpor_pk_load_gnark — the file's compacted, compressed arrays are decompressed,
    re-laid out wire-indexed AND turned into tables on the device; the proof is the oracle's"""
    ctx = zkpor.Context(0)
    try:
        ctx.set_param("msm_tables", 4)
        S = O.Synth(6, 700, n_public=2, seed=33)
        nb = 24
        bs = O.fr_random(31, nb); sig = O.fr_random(32, 1)[0]
        basis = O.g1_from_scalars(bs); basis_sigma = O.g1_from_scalars(O.fr_mul(bs, np.repeat(sig[None, :], nb, axis=0)))
        data, _, _ = GK.pk_bytes_from_synth(S, [(basis, basis_sigma)])
        pk = zkpor.ProvingKey(ctx)
        try:
            pk.load_gnark(data, S.n_public)
            r = O.fr_random(5, 1)[0]; s = O.fr_random(6, 1)[0]
            proof = ctx.prove_tail(pk, S.w, S.a, S.b, S.c, r, s)
            assert np.array_equal(proof, S.prove_tail(r, s)) and S.verify_pairing(proof)
            vals = O.fr_random(33, nb)
            c, k = ctx.commit(pk, vals)
            assert np.array_equal(c, O.g1_msm(basis, vals)) and O.pedersen_verify_pairing(c, k, O.g2_mul_gen(sig))
            with pytest.raises(zkpor.ZkporError, match="fixed-base tables"):
                pk.g1_dev(zkpor.G1_Z)
        finally:
            pk.close()
    finally:
        ctx.close()
