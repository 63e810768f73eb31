"""TEST INFRASTRUCTURE — an R1CS evaluator in Python integers that shares nothing with the package's solver executors.

What the solver (gnark: constraint/bn254/solver.go; here csrc/solver.hip, host/solver_exec.hpp, host/circuit/frontend.hpp's interpreter)
has to deliver is a wire vector w with (L w) o (R w) = O w on every row and the assigned inputs in their slots; hints are advice, any w that
satisfies the rows is a valid witness.  This file checks exactly that from the STATEMENT — the three sparse matrices, the coefficient
table and w, all converted to Python integers through the oracle's limb conversion — with numpy object arithmetic: no header of the
package (solver_instr.cuh, frontend.hpp, fr_host.hpp) is involved, and neither is the device (VERDICT r04 missing #4).
"""
import numpy as np

R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001


def to_ints(limbs_mont):
    """Montgomery limbs (n, 4) uint64 -> Python integers in [0, R), through oracle/ (orc_fr_to_canon)"""
    import oracle as O
    return np.array(O.fr_to_ints(np.ascontiguousarray(limbs_mont, dtype=np.uint64).reshape(-1, 4)), dtype=object)


def mat_vec(row_ptr, coeff_ids, wire_ids, coeff_ints, w_ints):
    """(M w)[row] mod R for a CSR matrix whose entries are indices into the coefficient table"""
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    n = row_ptr.shape[0] - 1
    out = np.zeros(n, dtype=object)
    if len(coeff_ids) == 0:
        return out
    prod = coeff_ints[np.asarray(coeff_ids, dtype=np.int64)] * w_ints[np.asarray(wire_ids, dtype=np.int64)]
    nonempty = row_ptr[1:] > row_ptr[:-1]
    starts = row_ptr[:-1][nonempty]
    sums = np.add.reduceat(prod, starts)          # consecutive non-empty rows: each sum ends where the next begins
    out[nonempty] = sums % R
    return out


def failing_rows(coeff_table_mont, mats, w_mont):
    """rows on which (L w)(R w) != O w; mats = [(row_ptr, coeff_ids, wire_ids)] for L, R, O"""
    co = to_ints(coeff_table_mont)
    w = to_ints(w_mont)
    a, b, c = (mat_vec(rp, ci, wi, co, w) for rp, ci, wi in mats)
    bad = np.nonzero((a * b - c) % R != 0)[0]
    return bad, (a, b, c)
