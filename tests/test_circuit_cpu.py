"""The circuit compiler (host/circuit/*.hpp): BatchCreateUserCircuit.Define (circuit/batch_create_user_circuit.go:98-323, circuit/utils.go:12-225)
restated on a gnark-shaped R1CS frontend and compiled to constraint matrices + solver program — CPU checks, no device:
  * a synthetic batch (host/circuit/synth_batch.hpp) is ACCEPTED by the circuit's interpreter (every assertion: Merkle paths, CEX / batch
    commitments, tier arithmetic, range checks, the log-derivative sums) and its hashes are the ORACLE's (the restatement pinned by the
    reference's user_config.json fixture) — so the in-circuit Poseidon gadget, the native hash path and the witness assignment agree;
  * the host executor (host/solver_exec.hpp) reproduces the interpreter's wire vector bit for bit, for the native Poseidon instruction and
    for gnark's own form (three constraint instructions per S-box), with the new lookup / count / Poseidon instruction kinds;
  * a tampered witness is refused; the constraint census sits where the reference's README says the real circuit does."""
import numpy as np
import pytest

import circuit as C
import oracle as O

R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def ints(a):
    return O.fr_to_ints(np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4))


@pytest.mark.parametrize("shape", [(3, 6, 3), (4, 10, 5), (6, 6, 2)])   # the last: every slot of the list a real asset (the zkpor500 shape)
def test_interpreter_accepts_a_synthetic_batch_and_the_host_executor_reproduces_it(shape):
    inp = C.synth_inputs(*shape, seed=11)
    c = C.Circuit(*shape, inputs=inp)
    try:
        assert c.n_public == 2 and c.n_public - 1 + c.n_secret == inp.shape[0]
        vals = c.values()
        assert np.array_equal(vals[1:1 + inp.shape[0]], inp)              # wire 0 = ONE, then the assignment
        w = c.solve_host(inp, C.default_commitment(), threads=3, check_rows=True)   # check_rows: every constraint holds for the solved wires
        assert np.array_equal(w, vals)
        # the census: lookups, range checks and Poseidon calls per the shape of Define
        T, A, U = shape
        assert c.census["lookup_call"] == U * (2 + 3 * T) + U            # user table + price table + 3 tier lookups per asset, + the powers table
        assert c.census["poseidon_call"] == 1 + 2 + 1 + U * (1 + 1 + 1 + 28)
        assert c.census["poseidon_perm_t3"] == 28 * U
        assert c.n_committed == len(set(c.committed().tolist())) and (np.diff(c.committed().astype(np.int64)) > 0).all()
        assert c.commitment_wire not in set(c.committed().tolist())
        ls = c.level_sizes()
        assert ls.sum() == c.n_instructions and (ls > 0).all()
    finally:
        c.close()


def test_gnarks_own_form_of_the_poseidon_gadget_gives_the_same_wires():
    """poseidon_native = False: the gadget as three constraint instructions per S-box, levels along every hash chain (what gnark's own
    solver walks) — the same matrices, the same wire values, thousands of levels instead of dozens"""
    shape = (3, 6, 2)
    inp = C.synth_inputs(*shape, seed=5)
    a = C.Circuit(*shape, inputs=inp, poseidon_native=True)
    b = C.Circuit(*shape, inputs=inp, poseidon_native=False)
    try:
        assert a.n_wires == b.n_wires and a.n_constraints == b.n_constraints and np.array_equal(a.values(), b.values())
        for m in range(3):
            for x, y in zip(a.matrix(m), b.matrix(m)):
                assert np.array_equal(x, y)
        assert b.n_levels > 20 * a.n_levels and b.n_instructions > a.n_instructions
        w = b.solve_host(inp, C.default_commitment(), threads=2)
        assert np.array_equal(w, b.values())
    finally:
        a.close(); b.close()


def test_the_batchs_hashes_are_the_oracles():
    """decode the assignment back into accounts / CEX records and let oracle/ (pinned by the reference's fixture) hash them: CEX commitment,
    account totals, leaves, Merkle proofs against the batch's root"""
    T, A, U = 4, 10, 6
    inp = C.synth_inputs(T, A, U, seed=21, first_index=9)
    v = ints(inp)
    per_cex, per_user = 114, 7 * T + 5 * A + 30
    consts = np.zeros(A, dtype=O.CEX_CONST_DTYPE); totals = np.zeros((1, A), dtype=O.CEX_TOTALS_DTYPE)
    for i in range(A):
        b = 6 + per_cex * i
        totals[0, i] = (v[b], v[b + 1], v[b + 3], v[b + 4], v[b + 5])
        consts[i]["base_price"] = v[b + 2]
        for li, name in enumerate(("loan", "margin", "portfolio_margin")):
            for k in range(12):
                bd, ratio = v[b + 6 + 36 * li + 3 * k], v[b + 6 + 36 * li + 3 * k + 1]
                consts[i][name][k]["boundary"] = (bd & ((1 << 64) - 1), bd >> 64)
                consts[i][name][k]["ratio"] = ratio
    before = O.cex_commitments(consts, totals)[0]
    assert ints(before)[0] == v[2], "BeforeCEXAssetsCommitment is not the oracle's commitment of the CEX asset list"
    accounts = np.zeros(U, dtype=O.ACCOUNT_DTYPE); assets = []
    ub = 6 + per_cex * A
    for u in range(U):
        base = ub + per_user * u
        meta = base + 7 * T
        accounts[u]["asset_off"] = len(assets)
        for p in range(A):
            e, d, lo, ma, pm = v[meta + 5 * p: meta + 5 * p + 5]
            if e | d | lo | ma | pm:
                assets.append((e, d, lo, ma, pm, p, 0))
        accounts[u]["n_assets"] = len(assets) - accounts[u]["asset_off"]
        accounts[u]["id_be"] = np.frombuffer(v[meta + 5 * A + 1].to_bytes(32, "big"), dtype=np.uint8)
    assets = np.array(assets, dtype=O.ASSET_DTYPE) if assets else np.zeros(0, dtype=O.ASSET_DTYPE)
    accounts, valid = O.account_totals(accounts, assets, consts)
    assert valid.all(), "the synthetic accounts obey the parser's rules (collateral <= equity, debt <= collateral)"
    leaves = O.account_leaves(accounts, assets, T)
    root = O.fr_from_ints([v[1]])[0]
    for u in range(U):
        base = ub + per_user * u
        meta = base + 7 * T
        key = v[meta + 5 * A]
        proof = O.fr_from_ints(v[meta + 5 * A + 2: meta + 5 * A + 30])
        assert key == 9 + u and O.merkle_verify(root, key, proof, leaves[u]), f"user {u}: the oracle's leaf does not reach the batch's root"


def test_a_tampered_witness_is_refused():
    shape = (3, 6, 2)
    inp = C.synth_inputs(*shape, seed=3)
    per_user = 7 * 3 + 5 * 6 + 30
    meta = 6 + 114 * 6 + 7 * 3
    for pos, what in ((1, "AccountTreeRoot"), (2, "BeforeCEXAssetsCommitment"), (meta, "a balance"), (6 + 2, "a price"), (meta + per_user + 30 + 1, "AccountIdHash")):
        bad = inp.copy()
        bad[pos, 0] ^= np.uint64(1)
        with pytest.raises(RuntimeError):
            C.Circuit(*shape, inputs=bad)
        with pytest.raises(RuntimeError):   # and the executor, which sees only the program, fails on an assertion / a hint too
            c = C.Circuit(*shape)
            try:
                c.solve_host(bad, C.default_commitment(), threads=2)
            finally:
                c.close()


def test_constraint_count_sits_where_the_reference_says():
    """README.md:12-14 of the reference: ~6.63 M constraints without users, 42.3 k per user of the 50-asset tier (281.2 k at 500 assets).  The
    restated gadgets are recalled, not gnark's sources, so the counts are compared, not claimed: base and per-user within 8 %"""
    one = C.Circuit(50, 500, 1); two = C.Circuit(50, 500, 2)
    try:
        per_user = two.n_constraints - one.n_constraints
        base = one.n_constraints - per_user
        assert abs(per_user - 42300) / 42300 < 0.08, per_user
        assert abs(base - 6630000) / 6630000 < 0.08, base
        # zkpor50_1380 then fits the 2^26 domain the reference targets (README.md:18-21)
        assert base + 1380 * per_user < 1 << 26 and base + 1380 * per_user > 1 << 25
    finally:
        one.close(); two.close()


def test_where_the_constraint_gap_to_the_reference_sits():
    """README.md:12-14 again, taken apart: the reference's three figures are a line in the user's asset count —
    281 200 - 42 300 over 450 slots = 530.9 constraints per user asset slot, 15 756 per user on top, 6.63 M shared.  The same three numbers
    of the restatement, from four compilations: the shared base is within 2 %, the per-user fixed part within 5 %, and the gap sits in
    the per-asset slope (-36 of 531 constraints per slot: the slot's 4 CmpNOp / 4 AssertIsLessOrEqualNOp / 29 lookup queries and its share
    of the collateral arithmetic are recalled expansions, circuit/utils.go:12-225) — three quarters of the per-user gap at 50 assets,
    nearly all of it at 500."""
    n = {}
    census = {}
    for sh in ((50, 500, 1), (50, 500, 2), (500, 500, 1), (500, 500, 2)):
        c = C.Circuit(*sh)
        n[sh] = c.n_constraints; census[sh] = dict(c.census)
        c.close()
    pu50 = n[(50, 500, 2)] - n[(50, 500, 1)]; pu500 = n[(500, 500, 2)] - n[(500, 500, 1)]
    base = n[(50, 500, 1)] - pu50
    slope = (pu500 - pu50) / 450.0; fixed = pu50 - 50 * slope
    ref_slope = (281200 - 42300) / 450.0; ref_fixed = 42300 - 50 * ref_slope
    assert abs(base / 6630000 - 1) < 0.02, base
    assert abs(fixed / ref_fixed - 1) < 0.05, fixed
    assert -0.09 < slope / ref_slope - 1 < -0.04, slope                      # the gap: 495 against 531 per slot
    for T, ref in ((50, 42300), (500, 281200)):
        gap = ref - (fixed + T * slope)
        assert gap > 0 and (ref_slope - slope) * T / gap > 0.7, (T, gap)
    # gadget calls per user asset slot (what the slope is made of): every count is linear in the slot count
    per_slot = {k: (census[(500, 500, 2)].get(k, 0) - census[(500, 500, 1)].get(k, 0) - census[(50, 500, 2)].get(k, 0) + census[(50, 500, 1)].get(k, 0)) / 450.0
                for k in ("assert", "inverse", "mul", "lookup_query", "is_zero", "cmp", "assert_le", "assert_bool", "lookup_call")}
    assert per_slot["cmp"] == 4 and per_slot["assert_le"] == 4 and abs(per_slot["lookup_query"] - 29) < 1e-9 and per_slot["lookup_call"] == 3
    assert 150 < per_slot["inverse"] < 160 and 170 < per_slot["assert"] < 190


def test_the_wire_vector_satisfies_the_statement_in_python_integers():
    """an evaluator of (L w) o (R w) = O w in Python integers (tests/r1cs_bigint.py: the matrices, the coefficient table and w only —
    no header of the package's executors) accepts the host executor's wire vector, finds the assigned inputs in their slots, and
    names the rows a flipped wire breaks (VERDICT r04 missing #4: the solver had no check outside its own package)"""
    import r1cs_bigint as RB
    shape = (4, 10, 5)
    inp = C.synth_inputs(*shape, seed=13)
    c = C.Circuit(*shape)
    try:
        w = c.solve_host(inp, C.default_commitment(), threads=3, check_rows=False)
        mats = [c.matrix(m) for m in range(3)]
        bad, _ = RB.failing_rows(c.coeff(), mats, w)
        assert bad.size == 0
        assert np.array_equal(w[1:1 + inp.shape[0]], inp) and ints(w[:1]) == [1]
        t = w.copy()
        k = c.n_wires // 2
        t[k] = O.fr_from_ints([(ints(t[k:k + 1])[0] + 1) % R])[0]
        bad, _ = RB.failing_rows(c.coeff(), mats, t)
        assert bad.size > 0
        # the oracle's evaluator (C++, its own field arithmetic: what the GPU tests apply at sizes Python integers are too slow for) agrees row for row
        assert O.r1cs_failing_rows(c.coeff(), mats, w) == (0, None)
        assert O.r1cs_failing_rows(c.coeff(), mats, t) == (int(bad.size), int(bad[0]))
        lo, hi = mats[0][0], None
        # every failing row really mentions the flipped wire in one of its three expressions
        for row in bad[:20]:
            assert any(k in mats[m][2][mats[m][0][row]:mats[m][0][row + 1]] for m in range(3))
    finally:
        c.close()


def test_assertions_are_flagged_check_in_the_container():
    """host/solver_file.hpp INSTR_CHECK: the R1C instructions without an unknown wire carry bit 8 of their kind word (an executor whose caller
    checks every row afterwards may leave them out); everything else reads the low byte — the host executor above parsed the same container"""
    c = C.Circuit(3, 6, 2)
    try:
        raw = np.frombuffer(c.solver_container(), dtype=np.uint8)
        assert bytes(raw[:8]) == b"ZKPSOLV\x02"
        n_instr, n_levels, n_names, n_call = np.frombuffer(raw[8:40].tobytes(), dtype=np.uint64)
        off = 40
        for _ in range(int(n_names)):
            l = int(np.frombuffer(raw[off:off + 4].tobytes(), dtype=np.uint32)[0]); off += 4 + l
        off += (8 - off % 8) % 8
        kinds = np.frombuffer(raw[off:off + 4 * int(n_instr)].tobytes(), dtype=np.uint32)
        flagged = (kinds & 0x100) != 0
        assert int(n_instr) == c.n_instructions and (kinds & ~np.uint32(0x1ff)).max() == 0
        assert ((kinds[flagged] & 0xff) == 0).all()                               # only constraints are checks
        n_assert = c.census["assert"] + c.census.get("assert_bool", 0)
        assert flagged.sum() >= c.census["assert"] and flagged.sum() > 0.1 * c.n_instructions, (int(flagged.sum()), n_assert)
    finally:
        c.close()


def test_the_check_flag_is_validated_and_changes_nothing_for_the_host_executor():
    """bit 8 of a kind word is legal on a constraint instruction of a version-2 container only; any other high bit, or the flag on a hint /
    lookup / Poseidon instruction, is a malformed container.  The host executor runs flagged instructions as the assertions they are
    (a tampered root is refused by it, above)"""
    import ctypes
    L = C.host_lib()
    c = C.Circuit(3, 6, 2)
    try:
        sv = bytes(c.solver_container())
        raw = np.frombuffer(sv, dtype=np.uint8)
        n_names = int(np.frombuffer(raw[24:32].tobytes(), dtype=np.uint64)[0])
        off = 40
        for _ in range(n_names):
            off += 4 + int(np.frombuffer(raw[off:off + 4].tobytes(), dtype=np.uint32)[0])
        off += (8 - off % 8) % 8
        kinds = np.frombuffer(raw[off:off + 4 * c.n_instructions].tobytes(), dtype=np.uint32)
        counts = (ctypes.c_uint64 * 4)(); err = ctypes.create_string_buffer(200)
        parse = lambda b: L.zkh_solver_parse(b, ctypes.c_size_t(len(b)), counts, err, ctypes.c_size_t(200))
        assert parse(sv) == 0
        def with_kind(i, k):
            m = bytearray(sv); m[off + 4 * i: off + 4 * i + 4] = int(k).to_bytes(4, "little"); return bytes(m)
        a_check = int(np.nonzero((kinds & 0x100) != 0)[0][0]); a_hint = int(np.nonzero(kinds == 1)[0][0]); a_pos = int(np.nonzero(kinds == 4)[0][0])
        assert parse(with_kind(a_check, 0)) == 0                                   # the flag is optional
        for i, k in ((a_hint, 0x101), (a_pos, 0x104), (a_check, 0x200), (a_check, 0x10100), (a_check, 0x105)):
            assert parse(with_kind(i, k)) != 0 and b"instruction kind" in err.value
    finally:
        c.close()


def _container_parts(raw):
    """(bytes up to the instruction table, kinds, args, level_ptr, level_instr, the rest) of a version-2 solver container"""
    raw = np.frombuffer(bytes(raw), dtype=np.uint8)
    n_instr, n_levels, n_names, _n_call = (int(x) for x in np.frombuffer(raw[8:40].tobytes(), dtype=np.uint64))
    off = 40
    for _ in range(n_names):
        off += 4 + int(np.frombuffer(raw[off:off + 4].tobytes(), dtype=np.uint32)[0])
    off += (8 - off % 8) % 8
    head = raw[:off].tobytes()
    kinds = np.frombuffer(raw[off:off + 4 * n_instr].tobytes(), dtype=np.uint32); off += 4 * n_instr
    args = np.frombuffer(raw[off:off + 4 * n_instr].tobytes(), dtype=np.uint32); off += 4 * n_instr
    off += (8 - off % 8) % 8
    lp = np.frombuffer(raw[off:off + 8 * (n_levels + 1)].tobytes(), dtype=np.uint64).astype(np.int64); off += 8 * (n_levels + 1)
    li = np.frombuffer(raw[off:off + 4 * int(lp[-1])].tobytes(), dtype=np.uint32); off += 4 * int(lp[-1])
    off += (8 - off % 8) % 8
    return head, kinds, args, lp, li, raw[off:].tobytes()


def _container_with_levels(head, kinds, args, levels, rest):
    lp = np.concatenate([[0], np.cumsum([len(l) for l in levels])]).astype("<u8")
    li = np.concatenate(levels).astype("<u4")
    body = head + kinds.astype("<u4").tobytes() + args.astype("<u4").tobytes()
    body += b"\0" * (-len(body) % 8)
    body += lp.tobytes() + li.tobytes()
    body += b"\0" * (-len(body) % 8)
    return body + rest


def test_the_join_level_of_the_long_call_is_a_promise_the_program_keeps():
    """poseidon(async = 2): the challenge sponge carries the level it must be complete in front of; nothing between its own level and that one
    reads its wires.  Checked by moving the call: at the END of the last level before its join the program still solves to the same wires
    (so an executor may run it beside everything in between); one level later — behind its first consumer — it does not"""
    shape = (3, 6, 2)
    inp = C.synth_inputs(*shape, seed=8)
    c = C.Circuit(*shape, inputs=inp)
    try:
        head, kinds, args, lp, li, rest = _container_parts(c.solver_container())
        cd = np.frombuffer(rest, dtype=np.uint32)
        calls = [i for i in np.nonzero((kinds & 0xff) == 4)[0] if (cd[args[i] + 3] >> 17) != 0]
        assert len(calls) == 1                                             # the RLC challenge
        call = int(calls[0]); flags = int(cd[args[call] + 3])
        join = (flags >> 17) - 1                                           # level index, counted from 0
        assert flags & (1 << 16) and int(cd[args[call]]) == shape[2] + 1   # ASYNC; its inputs: one id hash per user + the batch commitment
        levels = [li[lp[l]:lp[l + 1]].copy() for l in range(len(lp) - 1)]
        own = next(l for l, ids in enumerate(levels) if call in ids)
        assert own < join < len(levels) and join - own - 1 == c.census["levels_beside_the_long_call"] >= 28
        want = c.values()
        moved = [ids[ids != call] for ids in levels]
        late = list(moved); late[join - 1] = np.append(late[join - 1], np.uint32(call))
        w = c.solve_host_with(_container_with_levels(head, kinds, args, late, rest), inp, C.default_commitment(), threads=2)
        assert np.array_equal(w, want)
        too_late = list(moved); too_late[join] = np.append(too_late[join], np.uint32(call))
        with pytest.raises(RuntimeError):
            c.solve_host_with(_container_with_levels(head, kinds, args, too_late, rest), inp, C.default_commitment(), threads=2)
    finally:
        c.close()


def test_check_instructions_assign_no_wire():
    """the promise behind INSTR_CHECK: a program with every flagged instruction turned into a skipped one still assigns every wire, to the same
    values (an executor whose caller verifies a x b = c on every row loses nothing by leaving them out); un-flagged constraint instructions
    are not dispensable — skipping ONE of them leaves a wire unassigned"""
    shape = (3, 6, 2)
    inp = C.synth_inputs(*shape, seed=12)
    c = C.Circuit(*shape, inputs=inp)
    try:
        head, kinds, args, lp, li, rest = _container_parts(c.solver_container())
        levels = [li[lp[l]:lp[l + 1]] for l in range(len(lp) - 1)]
        flagged = (kinds & 0x100) != 0
        k2 = kinds.copy(); k2[flagged] = 2                                  # INSTR_SKIP
        w = c.solve_host_with(_container_with_levels(head, k2, args, levels, rest), inp, C.default_commitment(), threads=2)
        assert np.array_equal(w, c.values()) and flagged.sum() > 10000
        plain = np.nonzero(kinds == 0)[0]
        k3 = kinds.copy(); k3[plain[len(plain) // 2]] = 2
        with pytest.raises(RuntimeError):
            c.solve_host_with(_container_with_levels(head, k3, args, levels, rest), inp, C.default_commitment(), threads=2)
    finally:
        c.close()


def test_a_compiled_circuit_travels_to_other_ranks_through_shared_memory():
    """bench.py --gpus N: rank 0 compiles the tier's circuit and the others MAP its arrays from /dev/shm (circuit.py export_shared / SharedCircuit)
    instead of compiling eight times side by side — the same dimensions, census and arrays behind the same read interface, and the files may be
    unlinked as soon as every rank has mapped them"""
    import os
    cir = C.Circuit(4, 12, 3)
    tag = f"test_{os.getpid()}"
    try:
        d = C.export_shared(cir, tag)
        assert os.path.exists(os.path.join(d, "meta.json"))
        sh = C.SharedCircuit(tag)
        C.unlink_shared(tag)                                    # mapped pages stay valid after the unlink
        assert not os.path.exists(d)
        assert sh.shape == cir.shape and sh.dims == cir.dims and sh.census == cir.census
        for name in ("n_wires", "n_public", "n_secret", "n_constraints", "n_committed", "commitment_wire", "n_instructions", "n_levels"):
            assert getattr(sh, name) == getattr(cir, name)
        assert np.array_equal(sh.coeff(), cir.coeff()) and np.array_equal(sh.committed(), cir.committed())
        for m in range(3):
            for x, y in zip(sh.matrix(m), cir.matrix(m)):
                assert x.dtype == y.dtype and np.array_equal(x, y)
        for x, y in zip(sh.infinity_masks(), cir.infinity_masks()):
            assert np.array_equal(x, y)
        assert np.array_equal(sh.level_sizes(), cir.level_sizes())
        assert np.array_equal(sh.solver_container(), cir.solver_container())
        sh.close()
    finally:
        C.unlink_shared(tag)
        cir.close()
