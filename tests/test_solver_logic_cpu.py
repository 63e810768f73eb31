"""CPU: the instruction semantics of the DEVICE solver executor (csrc/solver_instr.cuh — what one GPU thread does for one instruction of a level;
SURVEY.md §8 f4, r1cs.Solve inside groth16.Prove, prover.go:269), compiled for the host (tests/hostlib/solver_logic.cpp) and walked level by
level: the wire vector must equal the builder's Python-integer values and the host executor's (host/solver_exec.hpp), in either order within a
level; the hints must agree with Python integers at every operand width; failures must come back as the same error codes."""
import ctypes
import os

import numpy as np
import pytest

import solver_circuit as SC

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LOGIC = ctypes.CDLL(os.path.join(ROOT, "tests", "hostlib", "libsolverlogic.so"))
HOST = ctypes.CDLL(os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd", "libzkpor_host.so"))


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def run_logic(b, solver=None, inputs=None, prefilled=None, back_to_front=0):
    r1 = b.r1cs_bytes()
    sv = b.solver_bytes() if solver is None else solver
    inp = SC.to_mont_limbs(b.val[:b.n_public + b.n_secret]) if inputs is None else inputs
    w = np.zeros((len(b.val), 4), np.uint64)
    info = np.zeros(2, np.uint64)
    ids = np.array([i for i, _ in (prefilled or [])], dtype=np.uint32)
    vals = SC.to_mont_limbs([v for _, v in (prefilled or [])]) if prefilled else np.zeros((0, 4), np.uint64)
    rc = LOGIC.sl_run(r1, ctypes.c_size_t(len(r1)), sv, ctypes.c_size_t(len(sv)), _p(inp), ctypes.c_size_t(inp.shape[0]), _p(ids), _p(vals),
                      ctypes.c_size_t(len(ids)), ctypes.c_int(back_to_front), _p(w), _p(info))
    return rc, w, [int(x) for x in info]


def run_host(b):
    r1, sv = b.r1cs_bytes(), b.solver_bytes()
    n_in = b.n_public + b.n_secret
    inp = SC.to_mont_limbs(b.val[:n_in])
    w = np.zeros((len(b.val), 4), np.uint64)
    st = np.zeros(3, np.uint64); err = ctypes.create_string_buffer(256)
    ids = np.zeros(0, np.uint32); vals = np.zeros((0, 4), np.uint64)
    rc = HOST.zkh_solve(r1, ctypes.c_size_t(len(r1)), sv, ctypes.c_size_t(len(sv)), _p(inp), ctypes.c_size_t(n_in), _p(ids), _p(vals), ctypes.c_size_t(0),
                        ctypes.c_int(2), _p(w), None, None, None, _p(st), err, ctypes.c_size_t(256))
    assert rc == 0, err.value
    return w


@pytest.mark.parametrize("order", [0, 1], ids=["front_to_back", "back_to_front"])
@pytest.mark.parametrize("seed,users,chain", [(1, 1, True), (2, 7, True), (3, 40, False)])
def test_wire_vector_equals_the_builders_and_the_host_executors(seed, users, chain, order):
    b = SC.demo_circuit(seed, users, chain=chain)
    rc, w, info = run_logic(b, back_to_front=order)
    assert rc == 0, info
    assert np.array_equal(w, SC.to_mont_limbs(b.val))
    assert np.array_equal(w, run_host(b))


def test_hints_at_every_operand_width():
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
        for order in (0, 1):
            rc, w, info = run_logic(bb, back_to_front=order)
            assert rc == 0, info
            assert np.array_equal(w, SC.to_mont_limbs(bb.val))


def test_prefilled_wires_and_skipped_instructions():
    b = SC.demo_circuit(7, 8)
    wires = b.wires_of_tag("sbox")
    sv = b.solver_bytes(skip_tags=("sbox",))
    rc, w, info = run_logic(b, solver=sv, prefilled=[(i, b.val[i]) for i in wires])
    assert rc == 0, info
    assert np.array_equal(w, SC.to_mont_limbs(b.val))
    rc, *_ = run_logic(b, solver=sv)
    assert rc != 0                        # without the values the dependants cannot be solved: an error, not a vector


def test_failures_carry_the_host_executors_codes():
    b = SC.demo_circuit(9, 4)
    n_in = b.n_public + b.n_secret
    vals = list(b.val[:n_in]); vals[b.n_public] = 1 << 70           # violates a range check: the decomposition hint refuses
    assert run_logic(b, inputs=SC.to_mont_limbs(vals))[0] == 24
    vals = list(b.val[:n_in]); vals[b.n_public + 4] = 0             # a zero price: IntegerDivision refuses
    assert run_logic(b, inputs=SC.to_mont_limbs(vals))[0] == 24
    lv = b.levels()
    assert run_logic(b, solver=b.solver_bytes(levels=lv[::-1]))[0] in (11, 23)     # wrong level order
    short = [l[:] for l in lv]; short[-2] = short[-2][:-1]
    assert run_logic(b, solver=b.solver_bytes(levels=short))[0] != 0                # a wire never assigned
    b2 = SC.demo_circuit(9, 2)
    b2.hint_names[b2.hint_names.index("InvZero")] = "SomeHintOfAnotherCircuit"
    assert run_logic(b2)[0] == 21
    b3 = SC.Builder([1], [0])
    b3.inverse(b3.wire(b3.n_public))                                                # 1 / 0
    assert run_logic(b3)[0] == 14
    b4 = SC.Builder([1], [3, 4])
    b4.assert_mul(b4.wire(b4.n_public), b4.wire(b4.n_public + 1), b4.const(13))    # 3 * 4 = 13
    assert run_logic(b4)[0] == 12


def test_binary_gcd_inverse_equals_the_fermat_power():
    """fr_inverse (binary extended Euclid, what a GPU thread runs for a division) against Fr::inv (a^(r-2)) and Python: random residues, the
    small and the large ones, powers of two, 0 -> 0"""
    rng = np.random.default_rng(8)
    vals = [int.from_bytes(rng.bytes(32), "big") % SC.R for _ in range(400)]
    vals += [0, 1, 2, 3, SC.R - 1, SC.R - 2, (SC.R + 1) // 2, 1 << 32, (1 << 64) - 1, 1 << 253, SC.MONT, pow(SC.MONT, SC.R - 2, SC.R)]
    vals += [1 << k for k in range(0, 254, 7)] + [(SC.R - (1 << k)) % SC.R for k in range(0, 254, 11)]
    a = SC.to_mont_limbs(vals)
    got = np.zeros_like(a); fermat = np.zeros_like(a)
    LOGIC.sl_fr_inverse(_p(a), _p(got), ctypes.c_size_t(len(vals)))
    LOGIC.sl_fr_inv_fermat(_p(a), _p(fermat), ctypes.c_size_t(len(vals)))
    assert np.array_equal(got, fermat)
    assert np.array_equal(got, SC.to_mont_limbs([pow(v, SC.R - 2, SC.R) if v else 0 for v in vals]))
