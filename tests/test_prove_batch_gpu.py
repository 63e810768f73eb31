"""-m gpu: Prover.GenerateAndVerifyProof end to end over the C ABI (host/prove_batch.hpp; src/prover/prover/prover.go:161-283): a
witness-table row in the reference's encoding is decoded, assigned to the circuit's input vector, handed to the solver (here a stub
that returns the test circuit's own solution after checking what it was given — gnark's solver is not part of this repo), the
BSB22 commitment and the prove tail run on the device, and the proof-table row comes out as the CSV line the unmodified verifier
reads.  Checked: proof bytes = the oracle's proof, commitment = the library's, row fields = the witness's, failures name their stage."""
import base64
import csv
import ctypes
import io
import json
import os

import numpy as np
import pytest

import gobs2 as G
import oracle as O
import zkpor

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_row_to_row(zk):
    from test_dispatcher_gpu import drv as _f  # noqa: F401  (builds the driver library if needed)
    host = ctypes.CDLL(os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd", "libzkpor_host.so"))
    drv = ctypes.CDLL(os.path.join(ROOT, "tests", "hostlib", "libdispatch_gpu.so"))
    users, assets, cex, seed = 3, 4, 500, 77
    buf = ctypes.create_string_buffer(1 << 24)
    host.zkh_witness_synth_encode.restype = ctypes.c_long
    n = host.zkh_witness_synth_encode(ctypes.c_uint64(seed), users, assets, cex, 1, 2, buf, ctypes.c_size_t(1 << 24))
    assert n > 0
    column = buf.raw[:n]
    full = G.synth_witness(seed, users, assets, cex)
    # the circuit behind the key: the oracle's synthetic R1CS plus one Pedersen commitment key of 40 points
    S = O.Synth(6, 300, n_public=2, seed=29)
    nb = 40
    bs = O.fr_random(31, nb); sig = O.fr_random(32, 1)[0]
    pk = zkpor.ProvingKey(zk)
    try:
        z = np.zeros(S.n_wires, dtype=np.uint8)
        pk.set_g1(zkpor.G1_A, S.A); pk.set_g1(zkpor.G1_B, S.B1); pk.set_g2(zkpor.G2_B, S.B2)
        pk.set_g1(zkpor.G1_K, S.K[S.n_public:]); pk.set_g1(zkpor.G1_Z, S.Z)
        pk.set_g1(zkpor.G1_COMMIT_BASIS, O.g1_from_scalars(bs))
        pk.set_g1(zkpor.G1_COMMIT_BASIS_SIGMA, O.g1_from_scalars(O.fr_mul(bs, np.repeat(sig[None, :], nb, axis=0))))
        pk.set_consts(S.abd1[0], S.abd1[1], S.abd1[2], S.bd2[0], S.bd2[1], S.log2d, z, z, S.n_wires, S.n_public)
        vals = O.fr_random(33, nb)
        r = O.fr_random(5, 1)[0]; s = O.fr_random(6, 1)[0]
        expect_inputs = 1 + 5 + 114 * cex + users * (7 * 50 + 5 * cex + 30)

        def run(fail_stage=0):
            out = ctypes.create_string_buffer(8192); err = ctypes.create_string_buffer(256)
            raw = np.zeros(512, dtype=np.uint8); raw_len = ctypes.c_size_t(); tier = ctypes.c_int()
            drv.prove_batch_row.restype = ctypes.c_long
            w, a, b, c = (np.ascontiguousarray(x) for x in (S.w, S.a, S.b, S.c))
            rc = drv.prove_batch_row(zk.h, pk.h, column, ctypes.c_size_t(len(column)), ctypes.c_int64(7), zkpor._p(w), zkpor._p(a), zkpor._p(b), zkpor._p(c),
                                     ctypes.c_size_t(S.n_wires), ctypes.c_size_t(S.n_cons), zkpor._p(vals), ctypes.c_size_t(nb), zkpor._p(r), zkpor._p(s),
                                     ctypes.c_uint64(expect_inputs), fail_stage, out, ctypes.c_size_t(8192), ctypes.byref(tier), zkpor._p(raw),
                                     ctypes.byref(raw_len), err, ctypes.c_size_t(256))
            return rc, out.raw[:max(rc, 0)].decode(), raw[:raw_len.value].tobytes(), tier.value, err.value.decode()

        rc, text, raw, tier, err = run()
        assert rc > 0, err
        assert tier == 50 and len(raw) == 388
        proof = S.prove_tail(r, s)
        assert raw[:256] == O.proof_raw(proof).tobytes() and raw[256:260] == b"\x00\x00\x00\x01"
        cm, kp = zk.commit(pk, vals)
        assert raw[260:388] == zkpor.proof_write_raw(proof, cm[None, :], kp).tobytes()[260:388]
        assert O.pedersen_verify_pairing(cm, kp, O.g2_mul_gen(sig)) and S.verify_pairing(proof)
        # the solver got the hint's output: hash_to_field(commitment), DST "bsb22-commitment" (host/bsb22_challenge.hpp)
        from test_bsb22_challenge_cpu import fr_hash_py
        ch = (ctypes.c_uint8 * 32)()
        drv.prove_batch_last_challenge(ch)
        assert int.from_bytes(bytes(ch), "big") == fr_hash_py(raw[260:324], b"bsb22-commitment", 1)[0]
        rows = list(csv.DictReader(io.StringIO(text)))
        assert len(rows) == 1
        row = rows[0]
        assert base64.b64decode(row["proof_info"]) == raw and int(row["batch_number"]) == 7 and int(row["assets_count"]) == 50
        assert base64.b64decode(row["batch_commitment"]) == full["BatchCommitment"]
        assert [base64.b64decode(x) for x in json.loads(row["cex_asset_list_commitments"])] == [full["BeforeCEXAssetsCommitment"], full["AfterCEXAssetsCommitment"]]
        assert [base64.b64decode(x) for x in json.loads(row["account_tree_roots"])] == [full["AccountTreeRoot"]]
        assert (int(row["min_account_index"]), int(row["max_account_index"])) == (full["MinAccountIndex"], full["MaxAccountIndex"])
        # failures name their stage: 3 = the solver, 5 = the verifier (prover.go:270-279 returns the error, Run stops)
        assert run(3)[0] == -3 and "solve" in run(3)[4]
        assert run(5)[0] == -5 and "rejected" in run(5)[4]
        bad = column[:-8] + b"AAAAAAA="
        out = ctypes.create_string_buffer(64); e2 = ctypes.create_string_buffer(256)
    finally:
        pk.close()
