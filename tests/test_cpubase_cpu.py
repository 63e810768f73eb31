"""CPU: the performance-minded port behind bench.py's cpu_baseline (oracle/cpubase.hpp: no-carry Montgomery, signed-digit
XYZZ Pippenger, cache-blocked FFT) computes what the plain oracle computes — it is a baseline only if it does the same work."""
import numpy as np
import pytest

import oracle as O


@pytest.mark.parametrize("n,window", [(1, 0), (37, 0), (700, 4), (5000, 0), (5000, 13), (70000, 16)])
def test_fast_msm_equals_oracle(n, window):
    sc = O.fr_random(n, n)
    sc[0] = O.fr_from_ints([0])[0]
    if n > 3:
        sc[1] = O.fr_from_ints([0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000000])[0]   # r - 1: every signed-digit carry
        sc[2] = O.fr_from_ints([1])[0]
        sc[3] = sc[2]
    base = O.fr_random(1000 + n, min(n, 512))
    p1 = np.tile(O.g1_from_scalars(base), (n // base.shape[0] + 1, 1))[:n].copy()
    if n > 10:
        p1[5] = 0                                  # a point at infinity
        p1[7] = p1[6]                              # equal points in one bucket (doubling branch) when the scalars match
        sc[7] = sc[6]
    assert np.array_equal(O.fast_g1_msm(p1, sc, window), O.g1_msm(p1, sc))
    if n <= 5000:
        p2 = np.tile(O.g2_from_scalars(base[:64]), (n // min(64, base.shape[0]) + 1, 1))[:n].copy()
        assert np.array_equal(O.fast_g2_msm(p2, sc, window), O.g2_msm(p2, sc))


@pytest.mark.parametrize("log2,short", [(3, 0), (10, 5), (16, 1), (17, 100)])
def test_fast_compute_h_equals_oracle(log2, short):
    n = (1 << log2) - short
    a = O.fr_random(1, n); b = O.fr_random(2, n); c = O.fr_mul(a, b)
    assert np.array_equal(O.fast_compute_h(a, b, c, log2), O.compute_h(a, b, c, log2))
