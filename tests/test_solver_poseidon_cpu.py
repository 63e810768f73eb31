"""CPU: the solver executors on a circuit made of the in-circuit Poseidon gadget with the REAL parameters (oracle/poseidon.hpp, pinned by the
reference's user_config.json fixture): Merkle paths of width-3 hashes (chains of permutations) and width-13 sponge blocks.  Three ways to the same
wire vector: everything solved by the host executor; the S-box wires pre-filled from the oracle's traced permutation in the slot order of the
device generator (zkpor_witgen_poseidon_trace_dev) with their instructions skipped — on the host executor and on the host-compiled device
semantics.  The gadget's outputs are checked against the oracle's own permutation."""
import ctypes
import os

import numpy as np
import pytest

import oracle as O
import solver_circuit as SC
from test_solver_exec_cpu import solve
from test_solver_logic_cpu import run_logic


def params():
    out = {}
    for t in (3, 13):
        rp, rc, mds = O.poseidon_params(t)
        out[t] = (rp, O.fr_to_ints(rc), O.fr_to_ints(mds))
    return out


def prefilled_from_the_oracle(rec):
    """(wire id, value) for every S-box wire, taken from the oracle's traced permutation in the generator's slot order"""
    pre = []
    for t, perms in rec.items():
        states = SC.to_mont_limbs([v for inputs, _ in perms for v in inputs]).reshape(len(perms), t, 4)
        _, trace = O.poseidon_permute_trace(states, t)                     # trace[(s * 3 + c), i]
        vals = O.fr_to_ints(trace.reshape(-1, 4))
        n = len(perms)
        for i, (_, wires) in enumerate(perms):
            for slot, w in enumerate(wires):
                pre.append((w, vals[slot * n + i]))
    return pre


def test_gadget_outputs_are_the_oracles_permutation_and_every_route_gives_the_same_wires():
    b, rec = SC.poseidon_circuit(params(), seed=3, paths=2, depth=3, wide=1)
    for t, perms in rec.items():
        for _, wires in perms:
            assert len(wires) == 3 * (8 * t + params()[t][0])
    n_sbox = sum(len(w) for perms in rec.values() for _, w in perms)
    assert len(b.wires_of_tag("sbox")) == n_sbox
    rc, w, a, bb, c, stats, err = solve(b, threads=3)
    assert rc == 0, err
    assert np.array_equal(w, SC.to_mont_limbs(b.val))
    # outputs = what the oracle computes: Merkle roots by chained hashing, sponge blocks by one permutation
    pub = b.val[1:b.n_public]
    depth = 3
    for p in range(2):
        inputs, _ = rec[3][p * depth + depth - 1]
        out = O.fr_to_ints(O.poseidon_permute(SC.to_mont_limbs(inputs)))
        assert out[1] == pub[p]
    inputs, _ = rec[13][0]
    assert O.fr_to_ints(O.poseidon_permute(SC.to_mont_limbs(inputs)))[1] == pub[2]
    # S-box wires from the oracle's trace, their instructions skipped: host executor and device semantics
    pre = prefilled_from_the_oracle(rec)
    assert sorted(i for i, _ in pre) == sorted(b.wires_of_tag("sbox"))
    sv = b.solver_bytes(skip_tags=("sbox",))
    rc, w2, *_rest, stats, err = solve(b, solver=sv, prefilled=pre, threads=2)
    assert rc == 0, err
    assert np.array_equal(w2, w) and stats[2] == n_sbox
    rc, w3, info = run_logic(b, solver=sv, prefilled=pre)
    assert rc == 0, info
    assert np.array_equal(w3, w)
    # a wrong trace value is caught by the assertions that remain
    bad = list(pre); bad[7] = (bad[7][0], bad[7][1] ^ 1)
    rc, *_x, err = solve(b, solver=sv, prefilled=bad)
    assert rc != 0
