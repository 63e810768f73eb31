"""-m gpu: bit-exact parity of the HIP NTT / computeH (through the C ABI) with the CPU oracle's restatement of
gnark-crypto fft.Domain.FFT / FFTInverse and gnark computeH."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(params=[1, 0], ids=["limbs29", "limbs32"])
def zkv(zk, request):
    """both arithmetic variants of the pass kernel: 9 x 29-bit lazy limbs (the default) and 8 x 32-bit limbs"""
    zk.set_param("ntt_variant", request.param)
    yield zk
    zk.set_param("ntt_variant", 1)


_CASES = {}


def _fft_case(log2n, inverse, decimation, coset):
    """input and oracle output, computed once and shared by the two kernel variants (the oracle is the slow side)"""
    key = (log2n, inverse, decimation, coset)
    if key not in _CASES:
        a = O.fr_random(1000 + log2n, 1 << log2n)
        _CASES[key] = (a, O.fft(a, log2n, inverse, decimation, coset))
    return _CASES[key]


@pytest.mark.parametrize("log2n", [1, 2, 3, 7, 8, 9, 10, 13, 17, 18])
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("decimation", [O.DIT, O.DIF])
@pytest.mark.parametrize("coset", [False, True])
def test_fft_matches_oracle(zkv, log2n, inverse, decimation, coset):
    zk = zkv
    a, ref = _fft_case(log2n, inverse, decimation, coset)
    got = zk.fft(a, log2n, inverse, decimation, coset)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("log2n", [19, 20, 22])
@pytest.mark.parametrize("inverse", [False, True])
@pytest.mark.parametrize("decimation", [O.DIT, O.DIF])
@pytest.mark.parametrize("coset", [False, True])
def test_fft_matches_oracle_three_wide_fields(zk, log2n, inverse, decimation, coset):
    """past 2^18 the plan has three fields with 9-bit upper fields (19 = 8+6+5 ... 22 = 8+7+7, 26 = 8+9+9): bit-exact with the
    oracle's radix-2 FFT for every mode gnark-crypto's Domain offers (default 29-bit kernel)"""
    a, ref = _fft_case(log2n, inverse, decimation, coset)
    assert np.array_equal(zk.fft(a, log2n, inverse, decimation, coset), ref)
    _CASES.pop((log2n, inverse, decimation, coset))      # 128 MiB per case at 2^22


def test_fft_edge_values(zkv):
    """inputs at the ends of the canonical range and all-equal vectors (worst-case limb carries in the lazy form)"""
    zk = zkv
    n = 10
    edge = [0, 1, 2, O.R_MOD - 1, O.R_MOD - 2, (1 << 253), (1 << 232) - 1, (1 << 29) - 1, 1 << 29, (O.R_MOD - 1) // 2]
    vals = (edge * ((1 << n) // len(edge) + 1))[: 1 << n]
    for a in (O.fr_from_ints(vals), O.fr_from_ints([O.R_MOD - 1] * (1 << n)), O.fr_from_ints([0] * (1 << n))):
        for inverse in (False, True):
            for dec in (O.DIT, O.DIF):
                for coset in (False, True):
                    assert np.array_equal(zk.fft(a, n, inverse, dec, coset), O.fft(a, n, inverse, dec, coset))


def test_fft_roundtrip_2_20(zk):
    # size-independent property at a larger size: iFFT_DIT(FFT_DIF(x)) == x on the coset
    n = 20
    a = O.fr_random(77, 1 << n)
    f = zk.fft(a, n, False, O.DIF, True)
    back = zk.fft(f, n, True, O.DIT, True)
    assert np.array_equal(back, a)


@pytest.mark.parametrize("log2d,ncons", [(3, 5), (8, 256), (10, 1000), (12, 4096)])   # 2^17 and up: the _large cases below
def test_compute_h_matches_oracle(zkv, log2d, ncons):
    zk = zkv
    key = ("h", log2d, ncons)
    if key not in _CASES:
        a = O.fr_random(1, ncons); b = O.fr_random(2, ncons)
        c = O.fr_mul(a, b)
        _CASES[key] = (a, b, c, O.compute_h(a, b, c, log2d))
    a, b, c, ref = _CASES[key]
    got = zk.compute_h(a, b, c, log2d)
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("log2d,short", [(19, 7), (20, 0), (22, 12345)])
def test_compute_h_matches_oracle_large(zkv, log2d, short):
    """computeH on random a, b, c past 2^18: bit-exact with the CPU port's computeH (oracle/cpubase.hpp, which
    tests/test_cpubase_cpu.py pins to the plain oracle; at 2^19 the plain oracle is run as well) and consistent with the
    FFT-free definition of the quotient (oracle/quotient.hpp)"""
    zk = zkv
    ncons = (1 << log2d) - short
    key = ("hl", log2d, ncons)
    if key not in _CASES:
        a = O.fr_random(31, ncons); b = O.fr_random(32, ncons); c = O.fr_mul(a, b)   # c = a.b on the domain: the quotient is a polynomial
        ref = O.fast_compute_h(a, b, c, log2d)
        if log2d <= 19:
            assert np.array_equal(ref, O.compute_h(a, b, c, log2d))
        assert O.quotient_identity(log2d, a, b, c, ref, O.fr_random(5, 1)[0])
        _CASES[key] = (a, b, c, ref)
    a, b, c, ref = _CASES[key]
    got = zk.compute_h(a, b, c, log2d)
    assert np.array_equal(got, ref)


def test_compute_h_fault_injection_detected(zk):
    """the quotient identity is sensitive to a single wrong twiddle: one flipped bit in one entry of the tabulated inter-pass
    twiddles (zkpor_set_param debug_ntt_fault) changes h and the identity fails; flipping it back restores h"""
    log2d = 18
    n = 1 << log2d
    a = O.fr_random(41, n); b = O.fr_random(42, n); c = O.fr_mul(a, b)
    tau = O.fr_random(6, 1)[0]
    good = zk.compute_h(a, b, c, log2d)
    assert O.quotient_identity(log2d, a, b, c, good, tau)
    zk.set_param("debug_ntt_fault", log2d)
    try:
        bad = zk.compute_h(a, b, c, log2d)
    finally:
        zk.set_param("debug_ntt_fault", log2d)
    assert not np.array_equal(bad, good)
    assert not O.quotient_identity(log2d, a, b, c, bad, tau)
    assert np.array_equal(zk.compute_h(a, b, c, log2d), good)


@pytest.mark.parametrize("log2d", [9, 17, 20])
def test_compute_h_fused_passes_equal_separate_passes(zk, log2d):
    """computeH's fused kernels ("ntt_fuse" 1, the default) — k_ntt_mid29: the two passes over the lowest field (inverse DIF last, coset
    DIT first) on one tile; k_ntt_top29: the last DIT pass of a, b and c, the pointwise quotient and the first DIF pass of h on one tile of
    the highest field — are bit-identical to the 21 separate passes + pointwise kernel, which the tests above pin to the oracle"""
    n = (1 << log2d) - 3
    a = O.fr_random(61, n); b = O.fr_random(62, n); c = O.fr_mul(a, b)
    fused = zk.compute_h(a, b, c, log2d)
    zk.set_param("ntt_fuse", 0)
    try:
        separate = zk.compute_h(a, b, c, log2d)
    finally:
        zk.set_param("ntt_fuse", 1)
    assert np.array_equal(fused, separate)
    if log2d <= 17:
        assert np.array_equal(fused, O.compute_h(a, b, c, log2d))


@pytest.mark.parametrize("log2n", [9, 13, 17, 18, 20])
def test_generated_twiddles_equal_tabulated_ones(zk, log2n):
    """"ntt_twiddles": the inter-pass twiddles w^((l k) << s0) as the product of two half-table entries (one more field product per element) instead of a
    read from the field's table (2 GiB per direction for the highest field at 2^26: 15 GB of the 86 GB a computeH moved).  Every mode of the
    transform and computeH with its fused kernels, generated everywhere (2) and where the table exceeds 16 MiB (1: from 2^20 up): bit-identical"""
    a = O.fr_random(2000 + log2n, 1 << log2n)
    n = (1 << log2n) - 5
    ha = O.fr_random(71, n); hb = O.fr_random(72, n); hc = O.fr_mul(ha, hb)
    want = {}
    for mode in (0, 2, 1):
        zk.set_param("ntt_twiddles", mode)
        try:
            for inverse in (False, True):
                for dec in (O.DIT, O.DIF):
                    for coset in (False, True):
                        got = zk.fft(a, log2n, inverse, dec, coset)
                        if mode == 0:
                            want[(inverse, dec, coset)] = got
                        else:
                            assert np.array_equal(got, want[(inverse, dec, coset)]), (mode, inverse, dec, coset)
            h = zk.compute_h(ha, hb, hc, log2n)
            if mode == 0:
                want["h"] = h
                if log2n <= 17:
                    assert np.array_equal(h, O.compute_h(ha, hb, hc, log2n))
            else:
                assert np.array_equal(h, want["h"]), mode
        finally:
            zk.set_param("ntt_twiddles", 0)


@pytest.mark.parametrize("log2d", [9, 10, 12, 17, 20])
@pytest.mark.parametrize("valid", [True, False])
def test_six_transforms_give_the_seven_transform_h_for_any_c(zk, log2d, valid):
    """"ntt_h" 1 (round 6, the default): c is only taken to its coefficients; den * c is subtracted behind h's inverse coset transform, which is linear —
    gnark's computeH takes c to the coset as well (seven transforms; "ntt_h" 0).  The same h bit for bit, also for a c that is NOT a b on the domain
    (nothing in the rearrangement uses the constraint), against the seven-transform schedule and, where the oracle finishes in seconds, against the
    oracle's literal restatement of gnark's computeH (oracle.compute_h: seven transforms)"""
    n = (1 << log2d) - (5 if log2d > 9 else 0)
    a = O.fr_random(71, n); b = O.fr_random(72, n)
    c = O.fr_mul(a, b) if valid else O.fr_random(73, n)
    six = zk.compute_h(a, b, c, log2d)
    zk.set_param("ntt_h", 0)
    try:
        seven = zk.compute_h(a, b, c, log2d)
    finally:
        zk.set_param("ntt_h", 1)
    assert np.array_equal(six, seven)
    if log2d <= 10:      # (the plain oracle takes 10 s at 2^12, 15 s at 2^17; the seven-transform schedule is pinned to it at those sizes by the tests above)
        assert np.array_equal(six, O.compute_h(a, b, c, log2d))
    # in place on the device with preserved inputs is what the prove tail runs: tests/test_groth16_gpu.py (proofs bit-exact with the oracle under both schedules)


def test_compute_h_zero_and_ragged(zk):
    # n_constraints = 0..1: zero padding path; h of the zero polynomial is zero
    z = np.zeros((1, 4), np.uint64)
    assert not zk.compute_h(z, z, z, 6).any()
    a = O.fr_random(3, 1)
    got = zk.compute_h(a, a, O.fr_mul(a, a), 6)
    assert np.array_equal(got, O.compute_h(a, a, O.fr_mul(a, a), 6))
