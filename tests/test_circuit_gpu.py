"""groth16.Prove of the compiled BatchCreateUserCircuit ON THE DEVICE (src/prover/prover/prover.go:254-274): the
assigned inputs go up, the solver program runs in HBM (csrc/solver.hip: generic constraint / hint / lookup instructions, the Poseidon
instruction per thread and group-cooperative, the count hints, the CEX commitments on the side stream), pauses at gnark's BSB22 placeholder,
the committed wires go straight into zkpor_commit_dev, the challenge comes back, the run resumes, a / b / c are evaluated and the prove tail
follows.  Checked: the wire vector bit for bit against the circuit's interpreter (host/circuit/frontend.hpp witness mode) and the host
executor, every constraint on the device (zkpor_r1cs_check_dev), commitment + knowledge proof + proof against the synthetic key's trapdoor
(a key with the CIRCUIT's sparsity: zkpor_pk_synth_masked)."""
import numpy as np
import pytest

import circuit as C
import oracle as O
import trapdoor as T
import zkpor

pytestmark = pytest.mark.gpu
SEED = 0x5A4B504F52


def prove_once(zk, shape, variant, seed=7, compare=True, reps=1, defer=None):
    inp = C.synth_inputs(*shape, seed=seed)
    cir = C.Circuit(*shape)
    zk.set_param("solver_poseidon", variant)
    if defer is not None:
        zk.set_param("poseidon_defer", defer)
    log2 = max(4, int(np.ceil(np.log2(cir.n_constraints))))
    D = 1 << log2
    inf_a, inf_b = cir.infinity_masks()
    removed = np.concatenate([cir.committed(), np.array([cir.commitment_wire], dtype=np.uint32)])
    pk = zkpor.ProvingKey(zk)
    dc = None
    bufs = []
    try:
        pk.synth_masked(log2, cir.n_wires, cir.n_public, inf_a, inf_b, removed, cir.n_committed, SEED)
        dc = C.DeviceCircuit(zk, cir)
        bufs = [zk.alloc(32 * n) for n in (cir.n_wires, D, D, D, cir.n_committed + 1)]
        for rep in range(reps):
            com, pok, ch = C.solve_on_device(zk, dc, pk, bufs[0].ptr, bufs[4].ptr, inp)
            assert dc.r1cs.check_dev(bufs[0].ptr) == (0, None)
            dc.r1cs.eval_dev(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, D)
            rr = O.fr_random(171 + rep, 1)[0]; ss = O.fr_random(272 + rep, 1)[0]
            proof = zk.prove_tail_dev(pk, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, rr, ss)
        w = bufs[0].download(np.uint64, (cir.n_wires, 4))
        h = bufs[1].download(np.uint64, (D, 4))
        cv = bufs[4].download(np.uint64, (cir.n_committed + 1, 4))[1:]
        assert np.array_equal(cv, w[cir.committed()])                     # the hint's inputs are the committed wires, in basis order
        ec, ek = T.expected_commitment(SEED, cv)
        assert np.array_equal(com, ec) and np.array_equal(pok, ek)
        td = T.SynthKeyTrapdoor(SEED, cir.n_public, w, h[: D - 1], masks=(inf_a, inf_b, removed))
        assert td.check(proof, rr, ss) and not td.check(proof, ss, rr)
        assert np.array_equal(w[cir.commitment_wire], ch)
        assert np.array_equal(w[1:1 + inp.shape[0]], inp)                 # the assignment sits in its slots
        # the DEVICE's wire vector against the statement itself — (L w) o (R w) = O w from the matrices, the coefficient table and w alone, by evaluators
        # outside the package: the oracle's (oracle/capi.cpp orc_r1cs_failing_rows, its own field arithmetic; every size up to 2^24 rows) and, for the
        # small shapes, Python integers as well (tests/r1cs_bigint.py)
        mats = [cir.matrix(m) for m in range(3)]
        if cir.n_constraints <= 1 << 24:
            assert O.r1cs_failing_rows(cir.coeff(), mats, w) == (0, None)
        if cir.n_constraints <= 1 << 20:
            import r1cs_bigint as RB
            bad, _ = RB.failing_rows(cir.coeff(), mats, w)
            assert bad.size == 0, bad[:10]
        if compare:
            ref = C.Circuit(*shape, inputs=inp, commitment=ch)            # the interpreter, given the challenge the device derived
            try:
                assert np.array_equal(w, ref.values())
            finally:
                ref.close()
            assert np.array_equal(w, cir.solve_host(inp, ch, threads=8, check_rows=False))
        return dc.solver.dims()
    finally:
        for b in bufs:
            b.free()
        if dc:
            dc.close()
        pk.close(); cir.close()
        zk.set_param("solver_poseidon", 1); zk.set_param("poseidon_defer", 64)


@pytest.mark.parametrize("variant", [1, 0])
@pytest.mark.parametrize("shape", [(3, 6, 3), (20, 40, 4)])
def test_small_batches_bit_exact_with_the_interpreter_both_poseidon_kernels(zk, shape, variant):
    """(20, 40, 4) walks the ragged sponge widths 3, 5, 6, 9 and 13; (3, 6, 3) the widths 2, 3, 4, 6, 7, 13"""
    prove_once(zk, shape, variant)


@pytest.mark.parametrize("defer", [0, 65535])
@pytest.mark.parametrize("shape", [(3, 6, 3), (20, 40, 4)])
def test_parked_sbox_inputs_expand_to_the_same_wires(zk, shape, defer):
    """"poseidon_defer" (round 6): launches of up to that many 16-lane calls park every S-box input raw in the S-box's own wire slots and
    k_sbox_expand behind them recomputes x^2, x^4, x^5, converts and writes (csrc/poseidon.hip).  0: no launch does (the waves convert as they go);
    65535: every launch of these shapes does, the wide Merkle levels included; the default (64: the sponge, the CEX chains, the narrow levels) runs in every
    other test of this file.  Same wires as the interpreter each way."""
    prove_once(zk, shape, 1, defer=defer)


def test_batched_inversions_of_wide_levels_change_no_wire(zk):
    """levels from `solver_batch_from` generic instructions on run four instructions per thread, their divisions sharing ONE field inversion
    (csrc/solver.hip k_solve_level_batched); the default threshold (2^21) is only met by production-size batches — lowered here"""
    zk.set_param("solver_batch_from", 64)
    try:
        prove_once(zk, (5, 20, 6), 1)
    finally:
        zk.set_param("solver_batch_from", 1 << 21)


def test_one_inversion_per_workgroup_and_long_constraints_by_waves_change_no_wire(zk):
    """levels from `solver_tree_from` generic instructions on: the divisions of a workgroup share one inversion (product tree in LDS) and the
    constraints of more than `solver_long` terms (256: the 1 024-term sums behind the log-derivative arguments) are left to one wave each
    (k_solve_long).  A small batch's wide levels hold no such constraint and the default `solver_tree_from` (1 024) passes most of them by —
    both lowered here so that every wide level takes the path and thousands of constraints go through the wave kernel, with one and with four
    instructions per thread; `solver_long` 0 walks them in their own thread: the same wires every way"""
    try:
        for tree_from, batch_from, long_ in ((1, 1 << 21, 6), (1, 2, 6), (1, 1 << 21, 0)):     # 6: the range checks' recompositions, the lookups' rows ... are "long"
            zk.set_param("solver_tree_from", tree_from); zk.set_param("solver_batch_from", batch_from); zk.set_param("solver_long", long_)
            prove_once(zk, (5, 20, 6), 1)
    finally:
        zk.set_param("solver_tree_from", 1024); zk.set_param("solver_batch_from", 1 << 21); zk.set_param("solver_long", 256)


def test_the_long_call_beside_the_independent_levels_changes_no_wire(zk):
    """the RLC challenge's sponge carries a join level (host/circuit/frontend.hpp poseidon(async = 2), solver_file.hpp POSEIDON_JOIN_SHIFT): the
    executor starts it on a side stream at its level and joins it in front of its first consumer; the levels in between (Merkle paths, hints,
    the big count hint) do not read it.  `solver_beside` 0 runs it in place.  Same wires both ways (prove_once compares with the interpreter)"""
    cir = C.Circuit(5, 20, 6)
    try:
        assert cir.census["levels_beside_the_long_call"] >= 28            # at least the Merkle levels
    finally:
        cir.close()
    try:
        for beside, pre_join in ((1, 1), (0, 1), (1, 0), (0, 0)):      # "solver_pre_join" (round 6): the sponge's input expressions evaluated in front of it (1, the default) or inside it
            zk.set_param("solver_beside", beside); zk.set_param("solver_pre_join", pre_join)
            prove_once(zk, (5, 20, 6), 1, reps=2)
        zk.set_param("solver_pre_join", 1)
        prove_once(zk, (5, 20, 6), 0)                                   # the one-thread-per-call kernel reads the pre-evaluated inputs too
    finally:
        zk.set_param("solver_beside", 1); zk.set_param("solver_pre_join", 1)


def test_the_500_asset_tier_shape(zk):
    """T = all assets (the zkpor500 shape: every slot of the user's list is a real CEX asset): sponges of 1000 / 3000 elements"""
    prove_once(zk, (30, 30, 2), 1)


def test_configs0_real_size_8_users_of_the_50_asset_tier(zk):
    """BASELINE.json configs[0]: zkpor50 tier, 8 users — 6.86 M constraints here (the reference's README gives 6.97 M for the real circuit),
    domain 2^23, solved and proved on the device, three proofs from one loaded program"""
    dims = prove_once(zk, (50, 500, 8), 1, reps=3)
    assert dims["levels"] < 3000 and dims["external_levels"] == 1


def test_rows_written_by_the_poseidon_instructions_equal_the_evaluated_ones(zk):
    """zkpor_solver_set_abc_dev / zkpor_solver_eval_abc_dev: the Poseidon instructions write a, b, c of their own constraint rows (they hold the S-box
    input; its expression has 14 to 79 terms), the evaluation kernel does the rest — bit-identical to evaluating every row from the matrices.
    With the per-thread kernel nothing is written by the solver and the same call evaluates everything."""
    shape = (20, 40, 4)
    inp = C.synth_inputs(*shape, seed=9)
    cir = C.Circuit(*shape)
    pk = zkpor.ProvingKey(zk)
    dc = C.DeviceCircuit(zk, cir)
    log2 = int(np.ceil(np.log2(cir.n_constraints)))
    D = 1 << log2
    bufs = [zk.alloc(32 * n) for n in (cir.n_wires, cir.n_committed + 1, D, D, D, D, D, D)]
    try:
        pk.synth(log2, cir.n_wires, cir.n_public, cir.n_committed, SEED)
        for variant, defer in ((1, 64), (1, 65535), (1, 0), (0, 64)):
            zk.set_param("solver_poseidon", variant); zk.set_param("poseidon_defer", defer)
            for b in bufs[2:]:
                b.upload(np.full((D, 4), 0xDEADBEEF, np.uint64))
            dc.solver.set_abc_dev(bufs[2].ptr, bufs[3].ptr, bufs[4].ptr)
            C.solve_on_device(zk, dc, pk, bufs[0].ptr, bufs[1].ptr, inp)
            dc.solver.eval_abc_dev(bufs[0].ptr, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr, D)
            dc.r1cs.eval_dev(bufs[0].ptr, bufs[5].ptr, bufs[6].ptr, bufs[7].ptr, D)
            for x, y in zip(bufs[2:5], bufs[5:8]):
                assert np.array_equal(x.download(np.uint64, (D, 4)), y.download(np.uint64, (D, 4)))
        dc.solver.set_abc_dev(None, None, None)
        assert cir.census["poseidon_call"] > 100
    finally:
        zk.set_param("solver_poseidon", 1); zk.set_param("poseidon_defer", 64)
        for b in bufs:
            b.free()
        dc.close(); pk.close(); cir.close()


def test_prefetching_the_next_proofs_hash_chains_changes_no_wire(zk):
    """zkpor_solver_prefetch_dev: the ASYNC instructions (the two CEX commitments) of the next proof started on the side stream before its run —
    three proofs over two alternating wire vectors, each bit-identical to a run without prefetch; a prefetch for a vector the next run does
    not use is abandoned"""
    shape = (4, 30, 3)
    inputs = [C.synth_inputs(*shape, seed=s) for s in (1, 2, 3)]
    cir = C.Circuit(*shape)
    pk = zkpor.ProvingKey(zk)
    dc = C.DeviceCircuit(zk, cir)
    log2 = int(np.ceil(np.log2(cir.n_constraints)))
    n_in = cir.n_public + cir.n_secret
    wb = [zk.alloc(32 * cir.n_wires) for _ in range(3)]
    cvb = zk.alloc(32 * (cir.n_committed + 1))
    try:
        pk.synth(log2, cir.n_wires, cir.n_public, cir.n_committed, SEED)
        plain = []
        for inp in inputs:
            C.solve_on_device(zk, dc, pk, wb[2].ptr, cvb.ptr, inp)
            plain.append(wb[2].download(np.uint64, (cir.n_wires, 4)))
        assert not np.array_equal(plain[0], plain[1])
        C.stage_inputs(zk, dc, wb[0].ptr, inputs[0])
        dc.solver.prefetch_dev(wb[0].ptr, n_in)
        for i, inp in enumerate(inputs):
            cur, nxt = wb[i % 2], wb[(i + 1) % 2]
            C.solve_on_device(zk, dc, pk, cur.ptr, cvb.ptr, inp, staged=True)
            if i + 1 < len(inputs):
                C.stage_inputs(zk, dc, nxt.ptr, inputs[i + 1])
                dc.solver.prefetch_dev(nxt.ptr, n_in)
            assert dc.r1cs.check_dev(cur.ptr) == (0, None)
            assert np.array_equal(cur.download(np.uint64, (cir.n_wires, 4)), plain[i])
        # a prefetch nobody consumes: the next run is given another vector and must not be disturbed
        C.stage_inputs(zk, dc, wb[0].ptr, inputs[2])
        dc.solver.prefetch_dev(wb[0].ptr, n_in)
        C.solve_on_device(zk, dc, pk, wb[2].ptr, cvb.ptr, inputs[1])
        assert np.array_equal(wb[2].download(np.uint64, (cir.n_wires, 4)), plain[1])
    finally:
        for b in wb:
            b.free()
        cvb.free(); dc.close(); pk.close(); cir.close()


@pytest.mark.isolated
def test_two_workers_of_one_gpu_solve_and_prove_side_by_side(zk):
    """one zkpor_r1cs, two solvers on two contexts (zkpor_solver_create_on), two host threads: each proves its own batches — solver program,
    commitment, a / b / c (zkpor_r1cs_eval_on), prove tail — while the other does the same; every wire vector equals the single-worker run's"""
    import threading
    shape = (4, 30, 3)
    inputs = [C.synth_inputs(*shape, seed=s) for s in (11, 12, 13, 14)]
    cir = C.Circuit(*shape)
    log2 = int(np.ceil(np.log2(cir.n_constraints)))
    D = 1 << log2
    pk = zkpor.ProvingKey(zk)
    ctx2 = zkpor.Context(0)
    dc1 = C.DeviceCircuit(zk, cir)
    dc2 = C.DeviceCircuit(ctx2, cir, share=dc1)
    res = {}
    try:
        pk.synth(log2, cir.n_wires, cir.n_public, cir.n_committed, SEED)

        def run(ctx, dc, mine, tag):
            bufs = [ctx.alloc(32 * n) for n in (cir.n_wires, D, D, D, cir.n_committed + 1)]
            try:
                for k in mine:
                    for _rep in range(2):
                        com, pok, ch = C.solve_on_device(ctx, dc, pk, bufs[0].ptr, bufs[4].ptr, inputs[k])
                        dc.r1cs.eval_dev(bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, D, ctx=ctx)
                        rr = O.fr_random(500 + k, 1)[0]; ss = O.fr_random(600 + k, 1)[0]
                        proof = ctx.prove_tail_dev(pk, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, rr, ss)
                    res[(tag, k)] = (bufs[0].download(np.uint64, (cir.n_wires, 4)), bufs[1].download(np.uint64, (D, 4)), proof, rr, ss, com)
            except Exception as e:     # surface in the main thread
                res[("error", tag)] = e
            finally:
                for b in bufs:
                    b.free()

        run(zk, dc1, [0, 1, 2, 3], "single")
        t1 = threading.Thread(target=run, args=(zk, dc1, [0, 1], "pair")); t2 = threading.Thread(target=run, args=(ctx2, dc2, [2, 3], "pair"))
        t1.start(); t2.start(); t1.join(); t2.join()
        assert not [k for k in res if k[0] == "error"], res
        for k in range(4):
            ws, hs, proof_s, rr, ss, com_s = res[("single", k)]
            wp, hp, proof_p, _, _, com_p = res[("pair", k)]
            assert np.array_equal(ws, wp) and np.array_equal(hs, hp) and np.array_equal(proof_s, proof_p) and np.array_equal(com_s, com_p)
            assert T.SynthKeyTrapdoor(SEED, cir.n_public, wp, hp[: D - 1]).check(proof_p, rr, ss)
    finally:
        dc2.close(); dc1.close(); ctx2.close(); pk.close(); cir.close()


def test_a_witness_the_circuit_rejects_is_an_error_not_a_proof(zk):
    shape = (3, 6, 2)
    inp = C.synth_inputs(*shape, seed=3)
    cir = C.Circuit(*shape)
    pk = zkpor.ProvingKey(zk)
    dc = C.DeviceCircuit(zk, cir)
    log2 = int(np.ceil(np.log2(cir.n_constraints)))
    bufs = [zk.alloc(32 * cir.n_wires), zk.alloc(32 * (cir.n_committed + 1))]
    try:
        pk.synth(log2, cir.n_wires, cir.n_public, cir.n_committed, SEED)
        meta = 6 + 114 * 6 + 7 * 3
        for pos in (1, meta):      # the tree root (caught by the final root assertion), a balance (caught on the way)
            bad = inp.copy(); bad[pos, 0] ^= np.uint64(1)
            with pytest.raises(zkpor.ZkporError) as e:
                C.solve_on_device(zk, dc, pk, bufs[0].ptr, bufs[1].ptr, bad)
            assert "solver:" in str(e.value)
        C.solve_on_device(zk, dc, pk, bufs[0].ptr, bufs[1].ptr, inp)       # and the program is usable afterwards
        assert dc.r1cs.check_dev(bufs[0].ptr) == (0, None)
    finally:
        for b in bufs:
            b.free()
        dc.close(); pk.close(); cir.close()


def test_assertions_left_to_the_row_check_of_eval_abc(zk):
    """with zkpor_solver_set_abc_dev the run leaves its CHECK instructions (the assertions: a third of the program) out and zkpor_solver_eval_abc_dev
    verifies a x b = c on EVERY row instead: the same wires, a witness that only an assertion rejects (the tree root) passes the run and is
    refused there with the row's number; a witness a hint rejects still fails in the run; `solver_defer_checks` 0 executes them as before"""
    shape = (3, 6, 2)
    inp = C.synth_inputs(*shape, seed=3)
    cir = C.Circuit(*shape)
    pk = zkpor.ProvingKey(zk)
    dc = C.DeviceCircuit(zk, cir)
    log2 = int(np.ceil(np.log2(cir.n_constraints)))
    D = 1 << log2
    bufs = [zk.alloc(32 * n) for n in (cir.n_wires, cir.n_committed + 1, D, D, D, cir.n_wires)]
    try:
        pk.synth(log2, cir.n_wires, cir.n_public, cir.n_committed, SEED)
        C.solve_on_device(zk, dc, pk, bufs[5].ptr, bufs[1].ptr, inp)                       # the plain run: every instruction
        launches_plain = dc.solver.dims()["launches_last_run"]
        dc.solver.set_abc_dev(bufs[2].ptr, bufs[3].ptr, bufs[4].ptr)
        C.solve_on_device(zk, dc, pk, bufs[0].ptr, bufs[1].ptr, inp)
        dc.solver.eval_abc_dev(bufs[0].ptr, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr, D)
        assert np.array_equal(bufs[0].download(np.uint64, (cir.n_wires, 4)), bufs[5].download(np.uint64, (cir.n_wires, 4)))
        assert dc.solver.dims()["launches_last_run"] <= launches_plain
        bad = inp.copy(); bad[1, 0] ^= np.uint64(1)                                         # AccountTreeRoot: only the root assertions see it
        C.solve_on_device(zk, dc, pk, bufs[0].ptr, bufs[1].ptr, bad)
        with pytest.raises(zkpor.ZkporError) as e:
            dc.solver.eval_abc_dev(bufs[0].ptr, bufs[2].ptr, bufs[3].ptr, bufs[4].ptr, D)
        assert "not satisfied" in str(e.value) and "#" in str(e.value)
        failing, first = dc.r1cs.check_dev(bufs[0].ptr)
        assert failing >= shape[2] and f"#{first}" in str(e.value) and f"{failing} constraints" in str(e.value)   # one root comparison per user (+ the batch commitment)
        meta = 6 + 114 * 6 + 7 * 3
        worse = inp.copy(); worse[meta, 0] ^= np.uint64(1)                                  # a balance: a lookup / hint on the way refuses it
        with pytest.raises(zkpor.ZkporError):
            C.solve_on_device(zk, dc, pk, bufs[0].ptr, bufs[1].ptr, worse)
        zk.set_param("solver_defer_checks", 0)
        with pytest.raises(zkpor.ZkporError) as e:
            C.solve_on_device(zk, dc, pk, bufs[0].ptr, bufs[1].ptr, bad)
        assert "solver:" in str(e.value)
        dc.solver.set_abc_dev(None, None, None)
    finally:
        zk.set_param("solver_defer_checks", 1)
        for b in bufs:
            b.free()
        dc.close(); pk.close(); cir.close()
