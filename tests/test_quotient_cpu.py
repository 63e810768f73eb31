"""CPU: the FFT-free checker of computeH's output (oracle/quotient.hpp) — the thing bench.py's `checked` leg and the full-size
GPU tests use to verify the device's h from the inputs alone.  It must accept what the oracle's computeH (gnark's, restated in
oracle/algos.hpp:228-246) and the CPU port's produce, in both coefficient orders, and reject any single wrong coefficient, a
wrong input, a shifted vector."""
import numpy as np
import pytest

import oracle as O


@pytest.mark.parametrize("log2d,short", [(1, 0), (3, 0), (3, 1), (9, 0), (12, 7), (13, 100), (15, 0)])
def test_quotient_identity_accepts_compute_h(log2d, short):
    n = (1 << log2d) - short
    a = O.fr_random(10 + log2d, n); b = O.fr_random(20 + log2d, n); c = O.fr_mul(a, b)
    h = O.compute_h(a, b, c, log2d)
    for seed in (1, 2):
        tau = O.fr_random(900 + seed, 1)[0]
        ok, vals = O.quotient_identity(log2d, a, b, c, h, tau, want_values=True)
        assert ok and np.array_equal(vals[4], vals[5])
        assert O.quotient_identity(log2d, a, b, c, O.bit_reverse(h, log2d), tau, h_bitrev=False)
    assert O.quotient_identity(log2d, a, b, c, O.fast_compute_h(a, b, c, log2d), O.fr_random(3, 1)[0])


def test_quotient_identity_values_are_the_polynomials():
    """A(tau), B(tau), C(tau), H(tau) themselves, against direct evaluation from coefficient vectors obtained another way"""
    log2d = 6
    n = 1 << log2d
    a = O.fr_random(1, n); b = O.fr_random(2, n); c = O.fr_mul(a, b)
    h = O.compute_h(a, b, c, log2d)
    tau = O.fr_random(5, 1)[0]
    _, vals = O.quotient_identity(log2d, a, b, c, h, tau, want_values=True)
    R = O.R_MOD
    t = O.fr_to_ints(tau.reshape(1, 4))[0]

    def horner(coef_ints):
        acc = 0
        for v in reversed(coef_ints):
            acc = (acc * t + v) % R
        return acc

    for k, ev in enumerate((a, b, c)):
        # coefficients by the naive inverse DFT: coef_k = 1/n sum_i ev_i w^(-ik)
        evi = O.fr_to_ints(ev)
        w = pow(5, (R - 1) >> log2d, R)
        wi = pow(w, R - 2, R)
        ninv = pow(n, R - 2, R)
        coef = [sum(evi[i] * pow(wi, i * kk, R) for i in range(n)) * ninv % R for kk in range(n)]
        assert O.fr_to_ints(vals[k].reshape(1, 4))[0] == horner(coef)
    hn = O.fr_to_ints(O.bit_reverse(h, log2d))
    assert O.fr_to_ints(vals[3].reshape(1, 4))[0] == horner(hn)


def test_quotient_identity_rejects_wrong_vectors():
    log2d = 12
    n = 1 << log2d
    a = O.fr_random(1, n); b = O.fr_random(2, n); c = O.fr_mul(a, b)
    h = O.compute_h(a, b, c, log2d)
    tau = O.fr_random(77, 1)[0]
    assert O.quotient_identity(log2d, a, b, c, h, tau)
    for pos in (0, 1, n // 2, n - 2, n - 1):
        for limb in (0, 3):
            bad = h.copy()
            bad[pos, limb] ^= np.uint64(1)
            assert not O.quotient_identity(log2d, a, b, c, bad, tau)
    assert not O.quotient_identity(log2d, a, b, c, np.roll(h, 1, axis=0), tau)
    assert not O.quotient_identity(log2d, a, b, c, h, tau, h_bitrev=False)      # the wrong coefficient order
    c2 = c.copy(); c2[5] = a[5]
    assert not O.quotient_identity(log2d, a, b, c2, h, tau)
    assert not O.quotient_identity(log2d, b, a, c, O.compute_h(a, a, c, log2d), tau)


def test_quotient_identity_refuses_a_domain_point():
    log2d = 4
    n = 1 << log2d
    a = O.fr_random(1, n); b = O.fr_random(2, n); c = O.fr_mul(a, b)
    h = O.compute_h(a, b, c, log2d)
    w = pow(5, (O.R_MOD - 1) >> log2d, O.R_MOD)
    with pytest.raises(ValueError):
        O.quotient_identity(log2d, a, b, c, h, O.fr_from_ints([pow(w, 3, O.R_MOD)])[0])
