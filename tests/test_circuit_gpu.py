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


def prove_once(zk, shape, variant, seed=7, compare=True, reps=1):
    inp = C.synth_inputs(*shape, seed=seed)
    cir = C.Circuit(*shape)
    zk.set_param("solver_poseidon", variant)
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
        zk.set_param("solver_poseidon", 1)


@pytest.mark.parametrize("variant", [1, 0])
@pytest.mark.parametrize("shape", [(3, 6, 3), (20, 40, 4)])
def test_small_batches_bit_exact_with_the_interpreter_both_poseidon_kernels(zk, shape, variant):
    """(20, 40, 4) walks the ragged sponge widths 3, 5, 6, 9 and 13; (3, 6, 3) the widths 2, 3, 4, 6, 7, 13"""
    prove_once(zk, shape, variant)


def test_the_500_asset_tier_shape(zk):
    """T = all assets (the zkpor500 shape: every slot of the user's list is a real CEX asset): sponges of 1000 / 3000 elements"""
    prove_once(zk, (30, 30, 2), 1)


def test_configs0_real_size_8_users_of_the_50_asset_tier(zk):
    """BASELINE.json configs[0]: zkpor50 tier, 8 users — 6.86 M constraints here (the reference's README gives 6.97 M for the real circuit),
    domain 2^23, solved and proved on the device, three proofs from one loaded program"""
    dims = prove_once(zk, (50, 500, 8), 1, reps=3)
    assert dims["levels"] < 3000 and dims["external_levels"] == 1


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
