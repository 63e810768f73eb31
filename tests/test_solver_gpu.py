"""-m gpu: the solver program executed on the device (csrc/solver.hip, include/zkpor.h zkpor_solver_*; SURVEY.md §8 f4 — r1cs.Solve inside
groth16.Prove, prover.go:269) through the C ABI: the wire vector must equal the builder's Python-integer values bit for bit (wide levels =
one thread per instruction, narrow levels = one workgroup stepping through a run), a . b = c must hold on the device's own evaluation of the
result, pre-filled wires must be taken as they are, external hints must pause the run and resume it, and every failure mode of the
reference solver must come back as an error."""
import ctypes
import os

import numpy as np
import pytest

import solver_circuit as SC
import zkpor

pytestmark = pytest.mark.gpu


def device_system(zk, b, solver=None):
    table, mats = b.tables()
    r = zkpor.R1CS(zk, len(b.rows), len(b.val), table)
    for m in range(3):
        r.set_matrix(m, *mats[m])
    return r, zkpor.Solver(r, b.solver_bytes() if solver is None else solver)


def inputs_of(b, vals=None):
    return SC.to_mont_limbs((b.val if vals is None else vals)[:b.n_public + b.n_secret])


@pytest.mark.parametrize("seed,users,chain", [(1, 1, True), (2, 7, True), (3, 40, False), (4, 700, False)])
def test_wire_vector_equals_the_builders(zk, seed, users, chain):
    b = SC.demo_circuit(seed, users, chain=chain)
    r, s = device_system(zk, b)
    try:
        d = s.dims()
        assert d["instructions"] == len(b.instr) and d["levels"] == len(b.levels()) and d["external_levels"] == 0
        w, st = s.run(inputs_of(b))
        assert np.array_equal(w, SC.to_mont_limbs(b.val))
        n_hint = sum(1 for k, _, _, _ in b.instr if k == 1)
        assert (st["constraint_instructions"], st["hint_instructions"], st["skipped"]) == (len(b.instr) - n_hint, n_hint, 0)
        widths = [len(l) for l in b.levels()]
        runs = sum(1 for i, n in enumerate(widths) if n > 512 or i == 0 or widths[i - 1] > 512)    # a launch per wide level, one per run of narrow ones
        runs += sum(1 for n in widths if n >= 1024)       # + the wave kernel for long constraints behind every level of `solver_tree_from` instructions and more
        assert st["launches"] == runs
        a, bb, c = r.eval(w)                         # the device's own a, b, c of the solved vector
        dw = [zk.alloc(a.nbytes).upload(x) for x in (a, bb)]
        out = zk.alloc(a.nbytes)
        zk._ck(zk.lib.zkpor_dev_fr_mul(zk.h, ctypes.c_void_p(out.ptr), ctypes.c_void_p(dw[0].ptr), ctypes.c_void_p(dw[1].ptr), ctypes.c_size_t(a.shape[0])))
        prod = out.download(np.uint64, a.shape)
        for x in dw + [out]:
            x.free()
        assert np.array_equal(prod, c)
        w2, _ = s.run(inputs_of(b))                  # the handle is reusable
        assert np.array_equal(w2, w)
    finally:
        s.close(); r.close()


def test_hints_at_every_operand_width(zk):
    rng = np.random.default_rng(21)
    cases = []
    for abits, bbits in [(20, 7), (64, 64), (130, 63), (200, 64), (253, 1), (253, 65), (253, 128), (250, 200), (100, 250), (0, 9), (33, 32), (64, 33)]:
        for _ in range(3):
            a = int.from_bytes(rng.bytes(32), "big") >> (256 - abits) if abits else 0
            d = (int.from_bytes(rng.bytes(32), "big") >> (256 - bbits)) | 1
            cases.append((a % SC.R, d % SC.R or 1))
    b = SC.Builder([5], [x for ab in cases for x in ab])
    base = b.n_public
    for i in range(len(cases)):
        b.integer_division(b.wire(base + 2 * i), b.wire(base + 2 * i + 1))
    for i, (bits, limb) in enumerate([(16, 16), (64, 16), (70, 7), (250, 60), (253, 64), (128, 13), (1, 1), (96, 32), (99, 33), (64, 31)]):
        v = cases[i][0] & ((1 << bits) - 1)
        b.range_check(b.wire(b.mul(b.const(v), b.const(1), "val")), bits, limb)
    for i, n in enumerate([1, 8, 31, 32, 33, 64, 65, 128, 253]):
        v = cases[3 + i][0] & ((1 << n) - 1)
        b.to_binary(b.wire(b.mul(b.const(v), b.const(1), "val")), n)
    vals = [int.from_bytes(rng.bytes(32), "big") % SC.R for _ in range(6)] + [0]
    b2 = SC.Builder([1], vals)
    for i in range(7):
        b2.is_zero(b2.wire(b2.n_public + i))
    for i in range(0, 6, 2):
        b2.inverse(b2.wire(b2.n_public + i))
        b2.div_left(b2.wire(b2.n_public + i), b2.wire(b2.n_public + i + 1))
    for bb in (b, b2):
        r, s = device_system(zk, bb)
        try:
            w, _ = s.run(inputs_of(bb))
            assert np.array_equal(w, SC.to_mont_limbs(bb.val))
        finally:
            s.close(); r.close()


def test_prefilled_wires_and_skipped_instructions(zk):
    """the S-box wires arrive from the device generators: their instructions are marked skipped, the values are taken as known"""
    b = SC.demo_circuit(7, 8)
    wires = b.wires_of_tag("sbox")
    r, s = device_system(zk, b, solver=b.solver_bytes(skip_tags=("sbox",)))
    try:
        pre = [(i, SC.to_mont_limbs([b.val[i]])[0]) for i in wires]
        w, st = s.run(inputs_of(b), prefilled=pre)
        assert np.array_equal(w, SC.to_mont_limbs(b.val)) and st["skipped"] == len(wires)
        with pytest.raises(zkpor.ZkporError):        # without the values the dependants cannot be solved: an error, not a vector
            s.run(inputs_of(b))
    finally:
        s.close(); r.close()


def test_generator_wires_are_handed_over_on_the_device(zk):
    """the device form of the hand-over: a generator's slots are scattered onto their wire ids AND marked assigned in one pass
    (zkpor_witgen_scatter_known_dev), the solver program then starts with those flags (d_known) — nothing of it touches the host"""
    b = SC.demo_circuit(17, 60, chain=False)
    wires = b.wires_of_tag("sbox")
    r, s = device_system(zk, b, solver=b.solver_bytes(skip_tags=("sbox",)))
    n_wires = len(b.val)
    n_in = b.n_public + b.n_secret
    d_w = zk.alloc(32 * n_wires); d_known = zk.alloc(n_wires)
    d_src = zk.alloc(32 * len(wires)); d_ids = zk.alloc(4 * len(wires))
    try:
        host_w = np.zeros((n_wires, 4), np.uint64)
        host_w[:n_in] = inputs_of(b)
        d_w.upload(host_w)
        d_known.upload(np.zeros(n_wires, np.uint8))
        order = np.random.default_rng(3).permutation(len(wires))                          # slots in any order
        d_src.upload(SC.to_mont_limbs([b.val[wires[i]] for i in order]))
        d_ids.upload(np.array([wires[i] for i in order], dtype=np.uint32))
        zk.witgen_scatter_known_dev(d_w.ptr, d_known.ptr, d_src.ptr, d_ids.ptr, len(wires))
        assert s.start_dev(d_w.ptr, n_in, d_known.ptr) == zkpor.NOT_PAUSED
        assert np.array_equal(d_w.download(np.uint64, (n_wires, 4)), SC.to_mont_limbs(b.val))
        known = d_known.download(np.uint8, (n_wires,))
        assert known.all()                                                                # inputs, generator wires and solved wires alike
        d_known.upload(np.zeros(n_wires, np.uint8))                                       # without the hand-over the program cannot finish
        d_w.upload(host_w)
        with pytest.raises(zkpor.ZkporError):
            s.start_dev(d_w.ptr, n_in, d_known.ptr)
    finally:
        for x in (d_w, d_known, d_src, d_ids):
            x.free()
        s.close(); r.close()


def test_external_hints_pause_and_resume(zk):
    """a hint without a native implementation (gnark's BSB22 commitment placeholder in the real circuit) stops the run in front of it: the caller
    reads the evaluated inputs, provides the outputs and resumes — two of them in one level, a third one later, wires in between depend on them"""
    rng = np.random.default_rng(5)
    vals = [int.from_bytes(rng.bytes(32), "big") % SC.R for _ in range(6)]
    b = SC.Builder([9], vals)
    base = b.n_public
    x = b.mul(b.wire(base), b.wire(base + 1))
    ext = lambda ins: ((sum(ins) * 7 + 3) % SC.R, (ins[0] * ins[0]) % SC.R)        # what the "caller" computes: two outputs
    e1_in = [b.add(b.wire(x), b.wire(base + 2, 5)), b.wire(base + 3)]
    o1 = b.hint("bsb22CommitmentComputePlaceholder", e1_in, list(ext([b.eval(e) for e in e1_in])))
    e2_in = [b.add(b.wire(x), b.wire(base + 4)), b.const(11)]                     # same level as the first one
    o2 = b.hint("bsb22CommitmentComputePlaceholder", e2_in, list(ext([b.eval(e) for e in e2_in])))
    y = b.mul(b.wire(o1[0]), b.wire(o2[1]))                                           # depends on both
    b.is_zero(b.sub(b.wire(y), b.wire(y)))
    e3_in = [b.wire(y), b.wire(o1[1])]
    o3 = b.hint("AnotherHintOfTheCaller", e3_in, list(ext([b.eval(e) for e in e3_in])))
    inv = b.inverse(b.add(b.wire(o3[0]), b.const(1)))
    e4_in = [b.wire(inv), b.const(5)]
    b.hint("AnotherHintOfTheCaller", e4_in, list(ext([b.eval(e) for e in e4_in])))      # an external hint in the LAST level of the program
    r, s = device_system(zk, b)
    d_w = zk.alloc(len(b.val) * 32)
    try:
        assert s.dims()["external_levels"] == 3
        with pytest.raises(zkpor.ZkporError):        # the host-buffer form does not serve external hints
            s.run(inputs_of(b))
        host_w = np.zeros((len(b.val), 4), np.uint64)
        n_in = b.n_public + b.n_secret
        host_w[:n_in] = inputs_of(b)
        d_w.upload(host_w)
        served = []
        paused = s.start_dev(d_w.ptr, n_in)
        while paused != zkpor.NOT_PAUSED:
            ins_vals, n_out = s.external_inputs(paused)
            d_in = zk.alloc(ins_vals.nbytes)                                             # the device form gives the same values
            s.external_inputs_dev(paused, d_in.ptr, ins_vals.shape[0])
            assert np.array_equal(d_in.download(np.uint64, ins_vals.shape), ins_vals)
            d_in.free()
            ints = SC.from_mont_limbs(ins_vals)
            assert n_out == 2
            with pytest.raises(zkpor.ZkporError):    # wrong instruction / wrong output count are refused, the pause stays
                s.external_outputs(paused + 1, SC.to_mont_limbs([1, 2]))
            with pytest.raises(zkpor.ZkporError):
                s.external_outputs(paused, SC.to_mont_limbs([1]))
            s.external_outputs(paused, SC.to_mont_limbs(list(ext(ints))))
            served.append((paused, ints))
            paused = s.resume_dev()
        assert len(served) == 4 and served[0][0] < served[1][0]
        assert [x[1] for x in served] == [[b.eval(e) for e in ins] for ins in (e1_in, e2_in, e3_in, e4_in)]
        w = d_w.download(np.uint64, (len(b.val), 4))
        assert np.array_equal(w, SC.to_mont_limbs(b.val))
        with pytest.raises(zkpor.ZkporError):        # nothing left to resume
            s.resume_dev()
    finally:
        d_w.free(); s.close(); r.close()


def test_a_level_with_more_external_hints_than_the_old_list_held(zk):
    """5 000 external hints side by side in ONE level (the list the level kernels report them through held 4 096 until round 4 and a level
    beyond it was ZKPOR_E_STATE): the list now holds every external hint of the program; all are served, in instruction
    order, and the wires behind them come out right"""
    n = 5000
    rng = np.random.default_rng(8)
    vals = [int(v) for v in rng.integers(1, 1 << 62, size=n)]
    b = SC.Builder([3], vals)
    base = b.n_public
    ext = lambda ins: ((ins[0] * 3 + 1) % SC.R,)
    outs = []
    for i in range(n):
        e_in = [b.wire(base + i)]
        outs.append(b.hint("AnotherHintOfTheCaller", e_in, list(ext([b.eval(e) for e in e_in])))[0])
    acc = b.mul(b.wire(outs[0]), b.wire(outs[-1]))
    b.is_zero(b.sub(b.wire(acc), b.wire(acc)))
    r, s = device_system(zk, b)
    d_w = zk.alloc(len(b.val) * 32)
    try:
        host_w = np.zeros((len(b.val), 4), np.uint64)
        n_in = b.n_public + b.n_secret
        host_w[:n_in] = inputs_of(b)
        d_w.upload(host_w)
        served = 0
        last = -1
        paused = s.start_dev(d_w.ptr, n_in)
        while paused != zkpor.NOT_PAUSED:
            assert paused > last
            last = paused
            ins_vals, n_out = s.external_inputs(paused)
            assert n_out == 1
            s.external_outputs(paused, SC.to_mont_limbs(list(ext(SC.from_mont_limbs(ins_vals)))))
            served += 1
            paused = s.resume_dev()
        assert served == n
        assert np.array_equal(d_w.download(np.uint64, (len(b.val), 4)), SC.to_mont_limbs(b.val))
    finally:
        d_w.free(); s.close(); r.close()


def test_failures_are_errors(zk):
    def fails(b, text, vals=None, solver=None):
        r, s = device_system(zk, b, solver=solver)
        try:
            with pytest.raises(zkpor.ZkporError) as e:
                s.run(inputs_of(b, vals))
            assert text in str(e.value), str(e.value)
        finally:
            s.close(); r.close()
    b = SC.demo_circuit(9, 4)
    n_in = b.n_public + b.n_secret
    vals = list(b.val[:n_in]); vals[b.n_public] = 1 << 70           # violates a range check: the decomposition hint refuses
    fails(b, "hint failed", vals=vals)
    vals = list(b.val[:n_in]); vals[b.n_public + 4] = 0             # a zero price: IntegerDivision refuses like big.Int.DivMod panics
    fails(b, "hint failed", vals=vals)
    lv = b.levels()
    fails(b, "solver:", solver=b.solver_bytes(levels=lv[::-1]))     # wrong level order: two unknown wires or an unsolved hint input
    short = [l[:] for l in lv]; short[-2] = short[-2][:-1]
    fails(b, "solver:", solver=b.solver_bytes(levels=short))        # an instruction missing from the levels: a wire is never assigned
    b3 = SC.Builder([1], [0])
    b3.inverse(b3.wire(b3.n_public))
    fails(b3, "division by zero")
    b4 = SC.Builder([1], [3, 4])
    b4.assert_mul(b4.wire(b4.n_public), b4.wire(b4.n_public + 1), b4.const(13))
    fails(b4, "constraint not satisfied")
    # a malformed container is refused when the solver is created
    table, mats = b.tables()
    r = zkpor.R1CS(zk, len(b.rows), len(b.val), table)
    for m in range(3):
        r.set_matrix(m, *mats[m])
    try:
        sv = bytearray(b.solver_bytes())
        with pytest.raises(zkpor.ZkporError):
            zkpor.Solver(r, sv[:len(sv) // 2])
        bad = bytearray(sv); bad[0] = ord("X")
        with pytest.raises(zkpor.ZkporError):
            zkpor.Solver(r, bad)
        b5 = SC.demo_circuit(9, 4)
        b5.instr[3] = (0, len(b5.rows) + 5, b5.instr[3][2], b5.instr[3][3])     # names a constraint outside the system
        with pytest.raises(zkpor.ZkporError):
            zkpor.Solver(r, b5.solver_bytes())
    finally:
        r.close()


def synth_solver_bytes(S, mats):
    """the solver program of the oracle's synthetic instance (oracle/algos.hpp synth_instance): constraint k is a multiplication gate that
    defines wire 1 + n_inputs + k from earlier wires — one instruction per constraint, levels from the wire dependencies"""
    import struct
    n_in = S.n_wires - S.n_cons
    level_of = np.zeros(S.n_wires, dtype=np.int64)
    levels = {}
    for k in range(S.n_cons):
        deps = [int(w) for m in (0, 1) for w in mats[m][2][int(mats[m][0][k]):int(mats[m][0][k + 1])]]
        lvl = 1 + max([int(level_of[w]) for w in deps] + [0])
        level_of[n_in + k] = lvl
        levels.setdefault(lvl, []).append(k)
    order = [levels[l] for l in sorted(levels)]
    out = bytearray(b"ZKPSOLV\x01") + struct.pack("<4Q", S.n_cons, len(order), 0, 0)
    out += np.zeros(S.n_cons, dtype="<u4").tobytes() + np.arange(S.n_cons, dtype="<u4").tobytes()
    out += b"\0" * (-len(out) % 8)
    ptr = np.cumsum([0] + [len(l) for l in order]).astype("<u8")
    out += ptr.tobytes() + np.array([i for l in order for i in l], dtype="<u4").tobytes()
    out += b"\0" * (-len(out) % 8)
    return bytes(out), len(order)


@pytest.mark.parametrize("n_inputs,n_cons", [(6, 700), (40, 2000)])
def test_prove_from_the_assigned_inputs_on_the_device(zk, n_inputs, n_cons):
    """groth16.Prove (prover.go:269) as ONE call from the assigned inputs: solver program -> w, constraint matrices -> a, b, c, prove tail, all in
    HBM (zkpor_prove_inputs) — the proof equals the oracle's proof of its own solved instance bit for bit and passes the pairing check; a second
    context of the GPU proves against the same resident program; an input that breaks nothing still proves, one of the wrong length is refused"""
    import oracle as O
    S = O.Synth(n_inputs, n_cons, n_public=2, seed=77 + n_cons)
    table, mats = S.r1cs()
    r = zkpor.R1CS(zk, S.n_cons, S.n_wires, table)
    for which, (row_ptr, cid, wid) in enumerate(mats):
        r.set_matrix(which, row_ptr, cid, wid)
    prog, n_levels = synth_solver_bytes(S, mats)
    s = zkpor.Solver(r, prog)
    pk = zkpor.ProvingKey(zk)
    zk2 = zkpor.Context(0)
    try:
        assert s.dims()["levels"] == n_levels and n_levels > 3
        n_in = S.n_wires - S.n_cons
        w, _ = s.run(S.w[:n_in])
        assert np.array_equal(w, S.w)                                   # the program reproduces the oracle's wire vector
        z = np.zeros(S.n_wires, dtype=np.uint8)
        pk.set_g1(zkpor.G1_A, S.A); pk.set_g1(zkpor.G1_B, S.B1); pk.set_g2(zkpor.G2_B, S.B2)
        pk.set_g1(zkpor.G1_K, S.K[S.n_public:]); pk.set_g1(zkpor.G1_Z, S.Z)
        pk.set_consts(S.abd1[0], S.abd1[1], S.abd1[2], S.bd2[0], S.bd2[1], S.log2d, z, z, S.n_wires, S.n_public)
        for i, ctx in enumerate((zk, zk2, zk)):
            rr = O.fr_random(150 + i, 1)[0]; ss = O.fr_random(160 + i, 1)[0]
            got = ctx.prove_inputs(pk, r, s, S.w[:n_in], rr, ss)
            assert np.array_equal(got, S.prove_tail(rr, ss))
        assert S.verify_pairing(got)
        with pytest.raises(zkpor.ZkporError):
            zk.prove_inputs(pk, r, s, S.w[:n_in - 1], rr, ss)           # one input short: wires stay unassigned
        bad = np.full(4, 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
        with pytest.raises(zkpor.ZkporError, match="blinding"):
            zk.prove_inputs(pk, r, s, S.w[:n_in], bad, ss)
    finally:
        zk2.close(); pk.close(); s.close(); r.close()


def _commitment_circuit(users, challenge):
    """U users with two 32-bit balances each: range checks (their 16-bit limbs are the COMMITTED wires), gnark's commitment placeholder hint
    over (index, limbs...) returning `challenge`, then the log-derivative shape: one inverse wire 1 / (challenge - limb) per limb and a
    running product that ties everything after the hint to the challenge.  Returns (builder, limb wire ids, hint instruction index)."""
    rng = np.random.default_rng(91)
    vals = [int(x) for x in rng.integers(1, 1 << 32, size=2 * users)]
    b = SC.Builder([7, 11], vals)
    base = b.n_public
    limbs = []
    for i in range(2 * users):
        limbs += b.range_check(b.wire(base + i), 32, 16)
    hint_at = len(b.instr)
    (ch,) = b.hint("bsb22CommitmentComputePlaceholder", [b.const(0)] + [b.wire(l) for l in limbs], [challenge])
    acc = b.mul(b.wire(ch), b.wire(ch))
    for i, l in enumerate(limbs):
        inv = b.inverse(b.sub(b.wire(ch), b.wire(l)))
        if i % 8 == 0:
            acc = b.mul(b.wire(acc), b.wire(inv))
    b.is_zero(b.sub(b.wire(acc), b.wire(ch)))
    return b, limbs, hint_at


def test_groth16_prove_with_a_commitment_entirely_on_the_device(zk):
    """What go/zkporgpu/solver.go ProveOnDevice does, step by step through the C ABI: inputs up, the solver program runs until gnark's BSB22
    placeholder, the committed wires go from the device straight into zkpor_commit_dev, the challenge is hashed on the host
    (host/bsb22_challenge.hpp — gnark's hash_to_field over the marshalled commitment), handed back, the run resumes; a, b, c are evaluated in
    HBM and the prove tail follows.  Checked: the commitment and its knowledge proof against the synthetic key's trapdoor, the challenge
    against an independent restatement, the whole wire vector against the builder's integers, a . b = c, and the proof in the exponent."""
    import ctypes
    import oracle as O
    import trapdoor as T
    from test_bsb22_challenge_cpu import fr_hash_py
    users, seed, log2 = 300, 0x5A4B504F52, 13
    b0, limbs, hint_at = _commitment_circuit(users, 0)            # pass 1: the committed values do not depend on the challenge
    committed = SC.to_mont_limbs([b0.val[l] for l in limbs])
    want_com, want_pok = T.expected_commitment(seed, committed)
    be = np.zeros(64, np.uint8)
    zk._ck(zk.lib.zkpor_g1_marshal(zkpor._p(want_com), zkpor._p(be)))
    challenge = fr_hash_py(bytes(be), b"bsb22-commitment", 1)[0]  # independent restatement of the hint's output
    b, limbs, hint_at = _commitment_circuit(users, challenge)     # pass 2: every wire value known to the builder
    n_wires, n_cons = len(b.val), len(b.rows)
    D = 1 << log2
    assert n_cons <= D and len(limbs) == 4 * users
    r, s = device_system(zk, b)
    pk = zkpor.ProvingKey(zk)
    bufs = [zk.alloc(32 * n) for n in (n_wires, D, D, D)]
    d_in = zk.alloc(32 * (1 + len(limbs)))
    vp = ctypes.c_void_p
    try:
        pk.synth(log2, n_wires, b.n_public, len(limbs), seed)
        n_in = b.n_public + b.n_secret
        host_w = np.zeros((n_wires, 4), np.uint64)
        host_w[:n_in] = inputs_of(b)
        bufs[0].upload(host_w)
        paused = s.start_dev(bufs[0].ptr, n_in)
        assert paused == hint_at
        s.external_inputs_dev(paused, d_in.ptr, 1 + len(limbs))
        got_in = d_in.download(np.uint64, (1 + len(limbs), 4))
        assert np.array_equal(got_in[1:], committed) and not got_in[0].any()      # (index 0, the committed wires in order)
        com = np.empty(8, np.uint64); pok = np.empty(8, np.uint64)
        zk._ck(zk.lib.zkpor_commit_dev(zk.h, pk.h, vp(d_in.ptr + 32), ctypes.c_size_t(len(limbs)), zkpor._p(com), zkpor._p(pok)))
        assert np.array_equal(com, want_com) and np.array_equal(pok, want_pok)
        host = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "zkmerkle-proof-of-solvency_amd", "libzkpor_host.so"))
        zk._ck(zk.lib.zkpor_g1_marshal(zkpor._p(com), zkpor._p(be)))
        out = (ctypes.c_uint8 * 32)()
        assert host.zkh_bsb22_challenge(bytes(be), None, ctypes.c_size_t(0), out) == 0
        assert int.from_bytes(bytes(out), "big") == challenge
        s.external_outputs(paused, SC.to_mont_limbs([int.from_bytes(bytes(out), "big")]))
        assert s.resume_dev() == zkpor.NOT_PAUSED
        w = bufs[0].download(np.uint64, (n_wires, 4))
        assert np.array_equal(w, SC.to_mont_limbs(b.val))
        r.eval_dev(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, D)
        abc = [x.download(np.uint64, (D, 4)) for x in bufs[1:]]
        assert np.array_equal(O.fr_mul(abc[0], abc[1]), abc[2]) and not abc[0][n_cons:].any()
        rr = O.fr_random(171, 1)[0]; ss = O.fr_random(172, 1)[0]
        proof = zk.prove_tail_dev(pk, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, rr, ss)
        h = bufs[1].download(np.uint64, (D, 4))                   # prove_tail_dev leaves h in a, in the order of the key's Z
        assert O.quotient_identity(log2, abc[0], abc[1], abc[2], h, O.fr_random(4243, 1)[0])
        td = T.SynthKeyTrapdoor(seed, b.n_public, w, h[: D - 1])
        assert td.check(proof, rr, ss) and not td.check(proof, ss, rr)
    finally:
        d_in.free()
        for x in bufs:
            x.free()
        pk.close(); s.close(); r.close()


def test_poseidon_generator_feeds_the_solver_program_on_the_device(zk):
    """f4 (c) + (d) together, with the real Poseidon parameters: the in-circuit gadget's S-box wires (Merkle paths of width-3 hashes, width-13 sponge
    blocks) come from the device generator (zkpor_witgen_poseidon_trace_dev), are scattered onto their wire ids and marked assigned
    (zkpor_witgen_scatter_known_dev), and the solver program — their instructions skipped — runs on what is left: the root / output assertions.
    The wire vector equals the builder's; a corrupted generator slot makes the program fail."""
    import oracle as O
    params = {}
    for t in (3, 13):
        rp, rc, mds = O.poseidon_params(t)
        params[t] = (rp, O.fr_to_ints(rc), O.fr_to_ints(mds))
    b, rec = SC.poseidon_circuit(params, seed=5, paths=6, depth=5, wide=3)
    r, s = device_system(zk, b, solver=b.solver_bytes(skip_tags=("sbox",)))
    n_wires, n_in = len(b.val), b.n_public + b.n_secret
    d_w = zk.alloc(32 * n_wires); d_known = zk.alloc(n_wires)
    held = []
    try:
        host_w = np.zeros((n_wires, 4), np.uint64)
        host_w[:n_in] = inputs_of(b)

        def generate(corrupt=False):
            d_w.upload(host_w)
            d_known.upload(np.zeros(n_wires, np.uint8))
            for t, perms in rec.items():
                n = len(perms)
                ns = zk.witgen_poseidon_sboxes(t)
                states = SC.to_mont_limbs([v for inputs, _ in perms for v in inputs])
                ids = np.zeros(3 * ns * n, np.uint32)
                for i, (_, wires) in enumerate(perms):
                    ids[np.arange(3 * ns) * n + i] = wires                    # slot (s * 3 + c) of permutation i -> its wire
                d_st = zk.alloc(states.nbytes).upload(states); d_tr = zk.alloc(3 * ns * n * 32); d_ids = zk.alloc(ids.nbytes).upload(ids)
                held.extend([d_st, d_tr, d_ids])
                zk.witgen_poseidon_trace_dev(t, d_st.ptr, n, d_tr.ptr)
                if corrupt and t == 3:
                    tr = d_tr.download(np.uint64, (3 * ns * n, 4)); tr[11, 0] ^= np.uint64(1); d_tr.upload(tr)
                zk.witgen_scatter_known_dev(d_w.ptr, d_known.ptr, d_tr.ptr, d_ids.ptr, 3 * ns * n)

        generate()
        assert s.start_dev(d_w.ptr, n_in, d_known.ptr) == zkpor.NOT_PAUSED
        assert np.array_equal(d_w.download(np.uint64, (n_wires, 4)), SC.to_mont_limbs(b.val))
        assert s.dims()["skipped"] == sum(len(w) for perms in rec.values() for _, w in perms)
        assert r.check_dev(d_w.ptr) == (0, None)                                   # the final check of every constraint, on the device
        # a corrupted generator slot (row 11 of the trace = slot 0 of permutation 11: x^2 of its first S-box): the skipped instruction's
        # constraint is seen by nobody but the final check
        generate(corrupt=True)
        assert s.start_dev(d_w.ptr, n_in, d_known.ptr) == zkpor.NOT_PAUSED
        n_bad, first = r.check_dev(d_w.ptr)
        bad_wire = rec[3][11][1][0]
        assert n_bad >= 1 and first == next(i for i, row in enumerate(b.rows) if row[2] == [(b.cid(1), bad_wire)])
    finally:
        for x in held + [d_w, d_known]:
            x.free()
        s.close(); r.close()
