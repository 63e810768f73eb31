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


@pytest.mark.isolated
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


def _input_circuit(values, picks, challenge):
    """a circuit over the ASSIGNED INPUT VECTOR of a batch (wire 0 = ONE, 1 public, the rest secret — circuit/types.go order): 64-bit range checks
    on `picks` (their 16-bit limbs are the committed wires), gnark's commitment placeholder, then one inverse wire per limb and a running product"""
    import solver_circuit as SC
    b = SC.Builder(values[:1], values[1:])
    limbs = []
    for i in picks:
        limbs += b.range_check(b.wire(1 + i), 64, 16)
    hint_at = len(b.instr)
    (ch,) = b.hint("bsb22CommitmentComputePlaceholder", [b.const(0)] + [b.wire(l) for l in limbs], [challenge])
    acc = b.mul(b.wire(ch), b.wire(1))
    for i, l in enumerate(limbs):
        inv = b.inverse(b.sub(b.wire(ch), b.wire(l)))
        if i % 16 == 0:
            acc = b.mul(b.wire(acc), b.wire(inv))
    return b, limbs, hint_at


@pytest.mark.isolated
def test_row_to_row_with_the_solver_on_the_device(zk):
    """Prover.GenerateAndVerifyProof with NO host solver (host/prove_on_device.hpp): the witness-table row is decoded and assigned on the host,
    the inputs cross PCIe, the solver program (with its BSB22 commitment), a / b / c and the prove tail run on the device, the proof-table row
    comes out.  Checked: the solved wire vector against the circuit builder's integers, the commitment / knowledge proof / challenge, the proof
    in the exponent (synthetic key's trapdoor) with h verified from its definition, the row's fields, a rejected proof names its stage."""
    from test_dispatcher_gpu import drv as _f  # noqa: F401  (builds the driver library if needed)
    from test_bsb22_challenge_cpu import fr_hash_py
    from test_solver_gpu import device_system
    import solver_circuit as SC
    import trapdoor as T
    host = ctypes.CDLL(os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd", "libzkpor_host.so"))
    drv = ctypes.CDLL(os.path.join(ROOT, "tests", "hostlib", "libdispatch_gpu.so"))
    users, assets, cex, wseed, seed, log2 = 3, 4, 500, 78, 0x5A4B504F52, 12
    buf = ctypes.create_string_buffer(1 << 24)
    host.zkh_witness_synth_encode.restype = ctypes.c_long
    n = host.zkh_witness_synth_encode(ctypes.c_uint64(wseed), users, assets, cex, 1, 2, buf, ctypes.c_size_t(1 << 24))
    assert n > 0
    column = buf.raw[:n]
    full = G.synth_witness(wseed, users, assets, cex)
    cap = 1 + 5 + 114 * cex + users * (7 * 50 + 5 * cex + 30)
    vals = np.zeros((cap, 4), np.uint64); counts = (ctypes.c_uint64 * 3)(); err = ctypes.create_string_buffer(256)
    host.zkh_witness_assign.restype = ctypes.c_long
    tiers = (ctypes.c_int * 2)(50, 500)
    assert host.zkh_witness_assign(column, ctypes.c_size_t(len(column)), tiers, 2, zkpor._p(vals), ctypes.c_size_t(cap), counts, err, ctypes.c_size_t(256)) == cap, err.value
    values = [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in vals]
    picks = [i for i, v in enumerate(values[1:], start=1) if 1 < v < (1 << 64)][:48]           # 48 inputs that fit 64 bits -> 192 committed limbs
    assert len(picks) == 48
    b0, limbs, hint_at = _input_circuit(values, picks, 0)
    want_com, want_pok = T.expected_commitment(seed, SC.to_mont_limbs([b0.val[l] for l in limbs]))
    be = np.zeros(64, np.uint8)
    zk._ck(zk.lib.zkpor_g1_marshal(zkpor._p(want_com), zkpor._p(be)))
    challenge = fr_hash_py(bytes(be), b"bsb22-commitment", 1)[0]
    b, limbs, hint_at = _input_circuit(values, picks, challenge)
    n_wires, D = len(b.val), 1 << log2
    assert len(b.rows) <= D and b.n_public == 2 and b.n_public + b.n_secret == 1 + cap
    r1, sp = device_system(zk, b)
    pk = zkpor.ProvingKey(zk)
    try:
        pk.synth(log2, n_wires, b.n_public, len(limbs), seed)
        r = O.fr_random(205, 1)[0]; s = O.fr_random(206, 1)[0]

        def run(fail_verify=0):
            out = ctypes.create_string_buffer(8192); e = ctypes.create_string_buffer(256)
            raw = np.zeros(512, dtype=np.uint8); raw_len = ctypes.c_size_t(); tier = ctypes.c_int()
            proof = np.zeros(256, np.uint8); ch = np.zeros(32, np.uint8)
            w = np.zeros((n_wires, 4), np.uint64); h = np.zeros((D, 4), np.uint64)
            drv.prove_row_on_device.restype = ctypes.c_long
            rc = drv.prove_row_on_device(zk.h, pk.h, r1.h, sp.h, column, ctypes.c_size_t(len(column)), ctypes.c_int64(9), zkpor._p(r), zkpor._p(s), fail_verify,
                                         out, ctypes.c_size_t(8192), ctypes.byref(tier), zkpor._p(raw), ctypes.byref(raw_len), zkpor._p(proof), zkpor._p(ch),
                                         zkpor._p(w), zkpor._p(h), e, ctypes.c_size_t(256))
            return rc, out.raw[:max(rc, 0)].decode(), raw[:raw_len.value].tobytes(), tier.value, proof, bytes(ch), w, h, e.value.decode()

        rc, text, raw, tier, proof, ch, w, h, e = run()
        assert rc > 0, e
        assert tier == 50 and len(raw) == 388
        assert np.array_equal(w, SC.to_mont_limbs(b.val))                               # the device solved the program from the row's inputs
        assert int.from_bytes(ch, "big") == challenge
        assert raw == zkpor.proof_write_raw(proof, want_com[None, :], want_pok).tobytes()  # proof | 1 commitment | knowledge proof, gnark's raw form
        a_, b_, c_ = r1.eval(w)
        pad = lambda x: np.concatenate([x, np.zeros((D - x.shape[0], 4), np.uint64)])
        assert O.quotient_identity(log2, pad(a_), pad(b_), pad(c_), h, O.fr_random(4244, 1)[0])
        assert T.SynthKeyTrapdoor(seed, b.n_public, w, h[: D - 1]).check(proof, r, s)
        rows = list(csv.DictReader(io.StringIO(text)))
        assert len(rows) == 1 and base64.b64decode(rows[0]["proof_info"]) == raw and int(rows[0]["batch_number"]) == 9 and int(rows[0]["assets_count"]) == 50
        assert base64.b64decode(rows[0]["batch_commitment"]) == full["BatchCommitment"]
        assert run(1)[0] == -5 and "rejected" in run(1)[8]
        half = column[: (len(column) // 2) & ~3]                                          # a truncated column
        assert drv.prove_row_on_device(zk.h, pk.h, r1.h, sp.h, half, ctypes.c_size_t(len(half)), ctypes.c_int64(9), zkpor._p(r), zkpor._p(s), 0,
                                       ctypes.create_string_buffer(64), ctypes.c_size_t(64), ctypes.byref(ctypes.c_int()), zkpor._p(np.zeros(512, np.uint8)),
                                       ctypes.byref(ctypes.c_size_t()), zkpor._p(np.zeros(256, np.uint8)), zkpor._p(np.zeros(32, np.uint8)), None, None,
                                       ctypes.create_string_buffer(256), ctypes.c_size_t(256)) == -1     # fails in the decode stage
    finally:
        pk.close(); sp.close(); r1.close()
