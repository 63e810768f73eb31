"""CPU suite: the 9 x 29-bit SIGNED lazy field representation (csrc/fe29.cuh) used inside the MSM hot loop, portable path,
against the oracle: conversions, product, fused double product, add/sub offsets, the zero filter, and the specialised
mixed addition (bounds on limb growth included)."""
import ctypes
import os

import numpy as np

import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


def _lib():
    from test_oracle_cpu import _hostlib
    return _hostlib()


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _rnd_fp(seed, n):
    return O.fp_from_ints(O.limbs_to_ints(O.fr_random(seed, n)))


EDGE = [0, 1, 2, O.P_MOD - 1, O.P_MOD - 2, (1 << 253), (1 << 232) - 1, (1 << 29) - 1, 1 << 29]


def test_roundtrip_and_product():
    L = _lib()
    n = 3000
    a = np.concatenate([_rnd_fp(1, n), O.fp_from_ints(EDGE)]); b = np.concatenate([_rnd_fp(2, n), O.fp_from_ints(EDGE[::-1])])
    o = np.empty_like(a)
    L.hm_fp29_roundtrip(_p(a), _p(o), ctypes.c_size_t(len(a)))
    assert np.array_equal(o, a)
    L.hm_fp29_mul(_p(a), _p(b), _p(o), ctypes.c_size_t(len(a)))
    assert np.array_equal(o, O.fp_mul(a, b))


def test_add_sub_mul2():
    L = _lib()
    n = 2000
    a = np.concatenate([_rnd_fp(3, n), O.fp_from_ints(EDGE)]); b = np.concatenate([_rnd_fp(4, n), O.fp_from_ints(EDGE[::-1])])
    s = np.empty_like(a); d = np.empty_like(a); ng = np.empty_like(a)
    L.hm_fp29_addsub(_p(a), _p(b), _p(s), _p(d), _p(ng), ctypes.c_size_t(len(a)))
    assert np.array_equal(s, O.fp_add(a, b)) and np.array_equal(d, O.fp_sub(a, b))
    assert np.array_equal(ng, O.fp_sub(O.fp_from_ints([0] * len(a)), a))
    c = np.concatenate([_rnd_fp(5, n), O.fp_from_ints(EDGE)]); e = np.concatenate([_rnd_fp(6, n), O.fp_from_ints(EDGE)])
    o = np.empty_like(a)
    L.hm_fp29_mul2(_p(a), _p(b), _p(c), _p(e), _p(o), ctypes.c_size_t(len(a)))
    assert np.array_equal(o, O.fp_sub(O.fp_mul(a, b), O.fp_mul(c, e)))      # y3(A,B,C,D) = A*B - C*D, one reduction


def test_zero_filter():
    L = _lib()
    a = _rnd_fp(7, 200)
    for i in range(200):
        assert L.hm_fp29_is_zero(_p(a[i:i + 1].copy()), _p(a[i:i + 1].copy())) == 1
        assert L.hm_fp29_is_zero(_p(a[i:i + 1].copy()), _p(a[(i + 1) % 200:(i + 1) % 200 + 1].copy())) == 0
    z = O.fp_from_ints([0])
    assert L.hm_fp29_is_zero(_p(z.copy()), _p(z.copy())) == 1


def test_mixed_addition_matches_oracle_and_stays_bounded():
    L = _lib()
    sc = O.fr_random(8, 400)
    pts = O.g1_from_scalars(sc)
    ones = O.fr_from_ints([1] * 400)
    out = np.empty(8, np.uint64); top = ctypes.c_uint32()
    L.hm_g1_sum29(_p(pts), None, ctypes.c_size_t(400), _p(out), ctypes.byref(top))
    assert np.array_equal(out, O.g1_msm(pts, ones, -1))
    assert top.value < (1 << 25)                        # |value| < 2^257 and tight limbs held at every step
    # signed digits: subtract every third point
    sign = np.array([1 if i % 3 == 0 else 0 for i in range(400)], dtype=np.uint8)
    pm = O.fr_from_ints([O.R_MOD - 1 if s else 1 for s in sign])
    L.hm_g1_sum29(_p(pts), _p(sign), ctypes.c_size_t(400), _p(out), ctypes.byref(top))
    assert np.array_equal(out, O.g1_msm(pts, pm, -1)) and top.value < (1 << 25)
    # doubling (same point twice), cancellation (P, -P) and restart after infinity
    neg = pts[:1].copy(); neg[:, 4:8] = O.fp_sub(O.fp_from_ints([0]), neg[:, 4:8])
    seq = np.concatenate([pts[:3], pts[:3], pts[5:6], pts[:1], neg, pts[9:12]])
    L.hm_g1_sum29(_p(seq), None, ctypes.c_size_t(len(seq)), _p(out), ctypes.byref(top))
    assert np.array_equal(out, O.g1_msm(seq, ones[:len(seq)], -1)) and top.value < (1 << 25)
    pair = np.concatenate([pts[:1], neg])
    L.hm_g1_sum29(_p(pair), None, ctypes.c_size_t(2), _p(out), ctypes.byref(top))
    assert not out.any()


def test_general_addition_and_doubling():
    """xyzz29_add / xyzz29_dbl (the partial-sum recursion and the bucket reduction): fold round-robin partial sums, the
    P == Q branch, a general doubling, and P + (-P)"""
    L = _lib()
    n = 300
    pts = O.g1_from_scalars(O.fr_random(21, n)); pts[17] = 0
    ones = O.fr_from_ints([1] * n)
    total = O.g1_msm(pts, ones, -1)
    out = np.empty(8, np.uint64); top = ctypes.c_uint32()
    for groups in (1, 2, 7, 16):
        L.hm_g1_sum29_general(_p(pts), ctypes.c_size_t(n), groups, 0, 0, _p(out), ctypes.byref(top))
        assert np.array_equal(out, total) and top.value < (1 << 25)
    L.hm_g1_sum29_general(_p(pts), ctypes.c_size_t(n), 5, 1, 0, _p(out), ctypes.byref(top))
    assert np.array_equal(out, O.g1_msm(pts, O.fr_from_ints([4] * n), -1)) and top.value < (1 << 25)
    L.hm_g1_sum29_general(_p(pts), ctypes.c_size_t(n), 5, 1, 1, _p(out), ctypes.byref(top))
    assert not out.any()
    # fewer points than accumulators: empty accumulators are skipped
    L.hm_g1_sum29_general(_p(pts[:3].copy()), ctypes.c_size_t(3), 16, 0, 0, _p(out), ctypes.byref(top))
    assert np.array_equal(out, O.g1_msm(pts[:3], ones[:3], -1))


def test_scalar_field_product_and_reductions():
    """Fr on the same limbs (the NTT's arithmetic): product, the product-free reduce32 on +/-32a for both fields, and the
    reduce32_pos -> pack32 -> from32<0> memory form of the inter-pass arrays"""
    L = _lib()
    n = 3000
    edge_r = [0, 1, 2, O.R_MOD - 1, O.R_MOD - 2, 1 << 253, (1 << 232) - 1, (1 << 29) - 1, 1 << 29]
    a = np.concatenate([O.fr_random(11, n), O.fr_from_ints(edge_r)]); b = np.concatenate([O.fr_random(12, n), O.fr_from_ints(edge_r[::-1])])
    o = np.empty_like(a)
    L.hm_fr29_mul(_p(a), _p(b), _p(o), ctypes.c_size_t(len(a)))
    assert np.array_equal(o, O.fr_mul(a, b))
    ap = np.concatenate([_rnd_fp(13, n), O.fp_from_ints(EDGE)])
    for neg in (0, 1):
        op = np.empty_like(ap); orr = np.empty_like(a)
        assert L.hm_fe29_reduce32(_p(ap), _p(a), ctypes.c_int(neg), _p(op), _p(orr), ctypes.c_size_t(len(a))) == 1
        zero_p = O.fp_from_ints([0] * len(ap)); zero_r = O.fr_from_ints([0] * len(a))
        assert np.array_equal(op, O.fp_sub(zero_p, ap) if neg else ap)
        assert np.array_equal(orr, O.fr_sub(zero_r, a) if neg else a)
        o2 = np.empty_like(a)
        assert L.hm_fr29_pack_roundtrip(_p(a), ctypes.c_int(neg), _p(o2), ctypes.c_size_t(len(a))) == 1
        two_a = O.fr_add(a, a)
        assert np.array_equal(o2, O.fr_sub(zero_r, two_a) if neg else two_a)
