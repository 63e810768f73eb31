"""-m gpu: constraint evaluation on the device: a = L.w, b = R.w, c = O.w from matrices resident in
HBM, bit-exact with the oracle's instance (oracle/algos.hpp synth_instance evaluates the same rows on the CPU), then
straight into the prove tail without a, b, c ever leaving the device."""
import ctypes

import numpy as np
import pytest

import oracle as O
import zkpor

pytestmark = pytest.mark.gpu


def _load(zk, S):
    table, mats = S.r1cs()
    r = zkpor.R1CS(zk, S.n_cons, S.n_wires, table)
    for which, (row_ptr, cid, wid) in enumerate(mats):
        r.set_matrix(which, row_ptr, cid, wid)
    return r, table, mats


@pytest.mark.parametrize("n_cons", [1, 7, 300, 5000])
def test_eval_matches_oracle(zk, n_cons):
    S = O.Synth(6, n_cons, n_public=2, seed=23 + n_cons)
    r, table, mats = _load(zk, S)
    try:
        a, b, c = r.eval(S.w)
        assert np.array_equal(a, S.a) and np.array_equal(b, S.b) and np.array_equal(c, S.c)
        # the coefficient table really exercises the three fast paths and the generic product
        one = O.fr_from_ints([1])[0]
        assert any(np.array_equal(t, one) for t in table) and table.shape[0] > 3
        # a different witness through the same matrices: c = a * b no longer holds, the evaluations still match a
        # direct restatement in Python integers
        w2 = O.fr_random(5, S.n_wires)
        a2, _, _ = r.eval(w2)
        wi = O.fr_to_ints(w2); ti = O.fr_to_ints(table)
        row_ptr, cid, wid = mats[0]
        for j in (0, n_cons // 2, n_cons - 1):
            acc = sum(ti[cid[t]] * wi[wid[t]] for t in range(int(row_ptr[j]), int(row_ptr[j + 1]))) % O.R_MOD
            assert O.fr_to_ints(a2[j:j + 1])[0] == acc
    finally:
        r.close()


def test_rows_in_evaluation_order_equal_rows_in_natural_order(zk):
    """round 6: k_r1cs_eval walks a matrix's rows by shape — term count, then the pattern of coefficient kinds, natural order inside a class — so that
    the 64 rows of a wave run the same iterations through the same branches ("r1cs_order" 1, the default); 0 is one thread per row in natural
    order.  Rows of every length from 0 to beyond the long-row limit, all four coefficient kinds mixed: the same a, b, c both ways, the zero padding up
    to the domain included, and both equal to Python integers"""
    rng = np.random.default_rng(11)
    n_w, n_c = 700, 3000
    table = O.fr_from_ints([0, 1, O.R_MOD - 1, 7, 12345678901234567890, O.R_MOD - 2])
    lens = rng.choice([0, 1, 2, 3, 3, 3, 4, 5, 8, 16, 17, 40, 79, 255, 256, 257, 600], size=n_c)
    row_ptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    nnz = int(row_ptr[-1])
    w = O.fr_random(77, n_w)
    mats = []
    for m in range(3):
        cid = rng.integers(0, 6, nnz).astype(np.uint32); wid = rng.integers(0, n_w, nnz).astype(np.uint32)
        mats.append((np.roll(lens, 17 * m), cid, wid))
    r = zkpor.R1CS(zk, n_c, n_w, table)
    D = 4096
    bufs = [zk.alloc(32 * D) for _ in range(3)]
    dw = zk.alloc(32 * n_w).upload(w)
    try:
        rps = []
        for m, (ln, cid, wid) in enumerate(mats):
            rp = np.concatenate([[0], np.cumsum(ln)]).astype(np.uint64)
            rps.append(rp)
            r.set_matrix(m, rp, cid, wid)
        got = {}
        for order in (1, 0):
            zk.set_param("r1cs_order", order)
            for b_ in bufs:
                b_.upload(np.full((D, 4), 0xABCDEF, np.uint64))
            r.eval_dev(dw.ptr, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, D)
            got[order] = [b_.download(np.uint64, (D, 4)) for b_ in bufs]
        for x, y in zip(got[0], got[1]):
            assert np.array_equal(x, y) and not x[n_c:].any()
        wi = O.fr_to_ints(w); ti = O.fr_to_ints(table)
        for m, (ln, cid, wid) in enumerate(mats):
            vals = O.fr_to_ints(got[1][m][:n_c])
            for j in list(range(0, n_c, 97)) + [int(np.argmax(ln))]:
                acc = sum(ti[cid[t]] * wi[wid[t]] for t in range(int(rps[m][j]), int(rps[m][j + 1]))) % O.R_MOD
                assert vals[j] == acc, (m, j)
    finally:
        zk.set_param("r1cs_order", 1)
        for b_ in bufs + [dw]:
            b_.free()
        r.close()


def test_special_coefficients_and_empty_rows(zk):
    """0, 1, -1 and a generic coefficient in one row; an empty linear expression evaluates to 0"""
    table = O.fr_from_ints([0, 1, O.R_MOD - 1, 12345])
    w = O.fr_from_ints([1, 10, 20, 30])
    r = zkpor.R1CS(zk, 2, 4, table)
    try:
        row_ptr = np.array([0, 4, 4], dtype=np.uint64)
        cid = np.array([0, 1, 2, 3], dtype=np.uint32); wid = np.array([1, 2, 3, 1], dtype=np.uint32)
        for which in range(3):
            r.set_matrix(which, row_ptr, cid, wid)
        a, b, c = r.eval(w)
        want = (20 - 30 + 12345 * 10) % O.R_MOD
        assert O.fr_to_ints(a) == [want, 0] and np.array_equal(a, b) and np.array_equal(a, c)
    finally:
        r.close()


def test_rows_longer_than_a_threads_share_are_summed_by_a_wave(zk):
    """rows of more than 256 terms go to one wave each (csrc/r1cs.hip k_r1cs_eval_long): lengths on both sides of the cut and of the lane
    count, every coefficient class, short rows in between — against Python integers"""
    rng = np.random.default_rng(41)
    n_wires = 5000
    table = O.fr_from_ints([0, 1, O.R_MOD - 1, 12345, 3 ** 150 % O.R_MOD, O.R_MOD - 7])
    w = O.fr_random(9, n_wires)
    lengths = [3, 256, 257, 0, 300, 1, 1024, 4143, 64, 511, 2, 320, 5000]
    r = zkpor.R1CS(zk, len(lengths), n_wires, table)
    try:
        wi = O.fr_to_ints(w); ti = O.fr_to_ints(table)
        want = []
        for which in range(3):
            ln = np.roll(np.array(lengths), which)
            row_ptr = np.concatenate([[0], np.cumsum(ln)]).astype(np.uint64)
            cid = rng.integers(0, table.shape[0], size=int(ln.sum())).astype(np.uint32)
            wid = rng.integers(0, n_wires, size=int(ln.sum())).astype(np.uint32)
            r.set_matrix(which, row_ptr, cid, wid)
            want.append([sum(ti[cid[t]] * wi[wid[t]] for t in range(int(row_ptr[j]), int(row_ptr[j + 1]))) % O.R_MOD for j in range(len(lengths))])
        got = r.eval(w)
        for which in range(3):
            assert O.fr_to_ints(got[which]) == want[which], which
    finally:
        r.close()


def test_rejects_bad_indices(zk):
    table = O.fr_from_ints([1, 2])
    r = zkpor.R1CS(zk, 1, 3, table)
    try:
        with pytest.raises(zkpor.ZkporError):                       # wire id out of range
            r.set_matrix(0, np.array([0, 1], np.uint64), np.array([0], np.uint32), np.array([3], np.uint32))
        with pytest.raises(zkpor.ZkporError):                       # coefficient id out of range
            r.set_matrix(0, np.array([0, 1], np.uint64), np.array([2], np.uint32), np.array([0], np.uint32))
        with pytest.raises(zkpor.ZkporError):                       # row_ptr does not span nnz
            r.set_matrix(0, np.array([0, 2], np.uint64), np.array([0], np.uint32), np.array([0], np.uint32))
        with pytest.raises(zkpor.ZkporError):                       # evaluating before all three matrices are loaded
            r.eval(O.fr_from_ints([1, 2, 3]))
    finally:
        r.close()


def test_witness_to_proof_without_abc_on_the_host(zk):
    """w -> (a, b, c on the device, zero padded to the domain) -> prove tail: same proof as the host-buffer path"""
    S = O.Synth(6, 700, n_public=2, seed=31)
    r, _, _ = _load(zk, S)
    pk = zkpor.ProvingKey(zk)
    D = 1 << S.log2d
    bufs = [zk.alloc(32 * D) for _ in range(3)]
    dw = zk.alloc(32 * S.n_wires).upload(S.w)
    try:
        z = np.zeros(S.n_wires, dtype=np.uint8)
        pk.set_g1(zkpor.G1_A, S.A); pk.set_g1(zkpor.G1_B, S.B1); pk.set_g2(zkpor.G2_B, S.B2)
        pk.set_g1(zkpor.G1_K, S.K[S.n_public:]); pk.set_g1(zkpor.G1_Z, S.Z)
        pk.set_consts(S.abd1[0], S.abd1[1], S.abd1[2], S.bd2[0], S.bd2[1], S.log2d, z, z, S.n_wires, S.n_public)
        r.eval_dev(dw.ptr, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, D)
        pad = bufs[0].download(np.uint64, (D, 4))
        assert np.array_equal(pad[:S.n_cons], S.a) and not pad[S.n_cons:].any()
        rr = O.fr_random(5, 1)[0]; ss = O.fr_random(6, 1)[0]
        proof = zk.prove_tail_dev(pk, dw.ptr, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, rr, ss)
        assert np.array_equal(proof, S.prove_tail(rr, ss)) and S.verify_pairing(proof)
    finally:
        pk.close(); r.close(); dw.free()
        for b in bufs:
            b.free()


def test_prove_r1cs_host_pointer_form(zk):
    """zkpor_prove_r1cs: w in host memory + matrices resident = the proof of the (w, a, b, c) host form = the oracle's; a second context
    of the same GPU proves against the same matrices; mismatches are refused"""
    S = O.Synth(6, 700, n_public=2, seed=35)
    r, _, _ = _load(zk, S)
    pk = zkpor.ProvingKey(zk)
    zk2 = zkpor.Context(0)
    try:
        z = np.zeros(S.n_wires, dtype=np.uint8)
        pk.set_g1(zkpor.G1_A, S.A); pk.set_g1(zkpor.G1_B, S.B1); pk.set_g2(zkpor.G2_B, S.B2)
        pk.set_g1(zkpor.G1_K, S.K[S.n_public:]); pk.set_g1(zkpor.G1_Z, S.Z)
        pk.set_consts(S.abd1[0], S.abd1[1], S.abd1[2], S.bd2[0], S.bd2[1], S.log2d, z, z, S.n_wires, S.n_public)
        for i, ctx in enumerate((zk, zk2, zk)):
            rr = O.fr_random(50 + i, 1)[0]; ss = O.fr_random(60 + i, 1)[0]
            want = S.prove_tail(rr, ss)
            got = ctx.prove_r1cs(pk, r, S.w, rr, ss)
            assert np.array_equal(got, want)
            assert np.array_equal(ctx.prove_tail(pk, S.w, S.a, S.b, S.c, rr, ss), want)
        assert S.verify_pairing(got)
        # a constraint system with another wire count than the key
        S2 = O.Synth(6, 650, n_public=2, seed=36)
        if S2.n_wires != S.n_wires:
            r2, _, _ = _load(zk, S2)
            try:
                with pytest.raises(zkpor.ZkporError, match="number of wires"):
                    zk.prove_r1cs(pk, r2, S2.w, rr, ss)
            finally:
                r2.close()
        # non-canonical blinding is refused as in the other forms
        bad = np.full(4, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
        with pytest.raises(zkpor.ZkporError, match="blinding"):
            zk.prove_r1cs(pk, r, S.w, bad, ss)
    finally:
        zk2.close(); pk.close(); r.close()


def test_exported_container_loads_and_evaluates(zk):
    """f1 ingestion end to end: the flat container go/export_r1cs writes (here: written by tests/r1cs_container.py from the oracle's
    instance) -> host/r1cs_file.hpp (C++: header walk on mapped bytes, zkpor_r1cs_create / set_matrix) -> a, b, c on the device"""
    import os
    import r1cs_container as RC
    from test_dispatcher_gpu import drv as _drv_fixture  # noqa: F401
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "hostlib", "libdispatch_gpu.so")
    lib = ctypes.CDLL(so)
    S = O.Synth(6, 700, n_public=2, seed=77)
    data = RC.from_synth(S, commitments=[(9, [3, 4], [])])
    buf = np.frombuffer(data, dtype=np.uint8)
    a = np.empty((S.n_cons, 4), np.uint64); b = np.empty_like(a); c = np.empty_like(a)
    err = ctypes.create_string_buffer(256)
    rc = lib.r1cs_file_eval(zkpor._p(buf), ctypes.c_size_t(buf.size), zkpor._p(np.ascontiguousarray(S.w)), zkpor._p(a), zkpor._p(b), zkpor._p(c), err, ctypes.c_size_t(256))
    assert rc == 0, err.value.decode()
    assert np.array_equal(a, S.a) and np.array_equal(b, S.b) and np.array_equal(c, S.c)
    # a wire id beyond n_wires inside the terms is caught by the device-side loader, not by the header walk
    bad = bytearray(data); bad[-4:] = (0xFFFFFFF0).to_bytes(4, "little")
    bb = np.frombuffer(bytes(bad), dtype=np.uint8)
    assert lib.r1cs_file_eval(zkpor._p(bb), ctypes.c_size_t(bb.size), zkpor._p(np.ascontiguousarray(S.w)), zkpor._p(a), zkpor._p(b), zkpor._p(c), err, ctypes.c_size_t(256)) == -3
