"""-m gpu: bit-exact parity of the HIP MSM (through the C ABI) against the CPU oracle.
Edge cases follow SURVEY.md §7 step 3: zeros, ones, r-1, repeated points, P/-P collisions, infinity points."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[1, 0], ids=["limbs29", "limbs32"])
def _pipeline(zk, request):
    """every case runs through both accumulation pipelines: the 9 x 29-bit lazy one with raw register images end to end
    (the default) and the 8 x 32-bit one"""
    zk.set_param("msm_g1_variant", request.param)
    zk.set_param("msm_g2_variant", request.param)
    yield
    zk.set_param("msm_g1_variant", 1)
    zk.set_param("msm_g2_variant", 1)


def _g1_eq(jac, aff_expected):
    return np.array_equal(O.g1_jac_to_affine(jac)[0], aff_expected)


def _g2_eq(jac, aff_expected):
    return np.array_equal(O.g2_jac_to_affine(jac)[0], aff_expected)


@pytest.mark.parametrize("n", [1, 2, 3, 63, 64, 65, 257, 1000, 4096])
def test_g1_msm_random(zk, n):
    sc = O.fr_random(100 + n, n)
    pts = O.g1_from_scalars(O.fr_random(200 + n, n))
    assert _g1_eq(zk.msm_g1(pts, sc), O.g1_msm(pts, sc))


@pytest.mark.parametrize("window", [2, 3, 5, 8, 11, 13, 16])
@pytest.mark.parametrize("chunk", [4, 32])
def test_g1_msm_windows_chunks(zk, window, chunk):
    n = 777
    sc = O.fr_random(5, n)
    pts = O.g1_from_scalars(O.fr_random(6, n))
    zk.set_param("msm_window", window)
    zk.set_param("msm_chunk", chunk)
    try:
        assert _g1_eq(zk.msm_g1(pts, sc), O.g1_msm(pts, sc))
    finally:
        zk.set_param("msm_window", 0)
        zk.set_param("msm_chunk", 0)


@pytest.mark.parametrize("window", [4, 9, 14, 0])
def test_g1_msm_reduction_tail_both_forms(zk, window):
    """the small levels of the bucket reduction as one lane per bucket (suffix scan + tree sums, the default) and as the serial walk
    (msm_reduce_scan 0): same sums; points repeated so that equal operands (the doubling branch) meet inside the scan"""
    n = 3000
    base = O.g1_from_scalars(O.fr_random(61, 40))
    pts = np.ascontiguousarray(base[np.arange(n) % 40])
    sc = O.fr_random(62, n)
    sc[::7] = sc[3]                      # equal scalars on equal points: equal bucket contents in neighbouring lanes
    want = O.g1_msm(pts, sc)
    zk.set_param("msm_window", window)
    try:
        for form in (1, 0):
            zk.set_param("msm_reduce_scan", form)
            assert _g1_eq(zk.msm_g1(pts, sc), want)
    finally:
        zk.set_param("msm_reduce_scan", 1)
        zk.set_param("msm_window", 0)


@pytest.mark.parametrize("window", [4, 9, 14, 0])
def test_g2_msm_reduction_tail_all_forms(zk, window):
    """G2: the small reduction levels as one lane PAIR per bucket (k_reduce_scan29_g2, the default), with the round-2 setting (scan for G1
    only: msm_reduce_scan 2) and as the serial walk (0): same sums; repeated points so that equal operands meet inside the scan"""
    n = 1500
    base = O.g2_from_scalars(O.fr_random(63, 24))
    pts = np.ascontiguousarray(base[np.arange(n) % 24])
    sc = O.fr_random(64, n)
    sc[::5] = sc[2]
    want = O.g2_msm(pts, sc)
    zk.set_param("msm_window", window)
    try:
        for form in (1, 2, 0):
            zk.set_param("msm_reduce_scan", form)
            assert _g2_eq(zk.msm_g2(pts, sc), want)
    finally:
        zk.set_param("msm_reduce_scan", 1)
        zk.set_param("msm_window", 0)


@pytest.mark.parametrize("tail", [0, 4, 5, 8, 16])
def test_msm_tail_chunk_values(zk, tail):
    """the small partial-sum levels with short chunks (msm_tail_chunk; 0 = the level-1 chunk everywhere): one heavy bucket spanning hundreds
    of chunks forces several recursion levels, random scalars give the many-small-segments case; G1 and G2"""
    n = 6000
    pts1 = O.g1_from_scalars(O.fr_random(65, n))
    pts2 = O.g2_from_scalars(O.fr_random(66, 1200))
    heavy = np.repeat(O.fr_random(67, 1), n, axis=0)
    mixed = O.fr_random(68, n)
    mixed[: n // 2] = mixed[0]
    zk.set_param("msm_tail_chunk", tail)
    try:
        for chunk in (32, 8):
            zk.set_param("msm_chunk", chunk)
            for sc in (heavy, mixed):
                assert _g1_eq(zk.msm_g1(pts1, sc), O.g1_msm(pts1, sc))
            assert _g2_eq(zk.msm_g2(pts2, mixed[:1200]), O.g2_msm(pts2, mixed[:1200]))
            assert _g2_eq(zk.msm_g2(pts2, heavy[:1200]), O.g2_msm(pts2, heavy[:1200]))
    finally:
        zk.set_param("msm_tail_chunk", 8)
        zk.set_param("msm_chunk", 0)


def test_chunk_parameters_below_four_are_refused(zk):
    """a level of the partial-sum recursion turns T threads into 2 T / chunk: below 4 entries per thread it would never reach one thread"""
    import zkpor
    for name in ("msm_chunk", "msm_tail_chunk"):
        for bad in (1, 2, 3, -1):
            with pytest.raises(zkpor.ZkporError):
                zk.set_param(name, bad)
    zk.set_param("msm_chunk", 0)
    zk.set_param("msm_tail_chunk", 8)


def test_g1_msm_edge_scalars(zk):
    n = 600
    pts = O.g1_from_scalars(O.fr_random(7, n))
    vals = [0, 1, O.R_MOD - 1, 2, O.R_MOD - 2, 1 << 16, (1 << 16) - 1, 1 << 253, 0xFFFF, 3]
    sc = O.fr_from_ints([vals[i % len(vals)] for i in range(n)])
    assert _g1_eq(zk.msm_g1(pts, sc), O.g1_msm(pts, sc))
    # all zero -> infinity; all one -> plain sum (one giant bucket: exercises the partial-sum recursion)
    z = O.fr_from_ints([0] * n)
    assert not O.g1_jac_to_affine(zk.msm_g1(pts, z)).any()
    ones = O.fr_from_ints([1] * n)
    assert _g1_eq(zk.msm_g1(pts, ones), O.g1_msm(pts, ones))


def test_g1_msm_skew_heavy_bucket(zk):
    # 5000 equal scalars: a single bucket per window spanning >100 chunks -> several recursion levels
    n = 5000
    pts = O.g1_from_scalars(O.fr_random(8, n))
    sc = np.repeat(O.fr_random(9, 1), n, axis=0)
    zk.set_param("msm_chunk", 8)
    try:
        assert _g1_eq(zk.msm_g1(pts, sc), O.g1_msm(pts, sc))
    finally:
        zk.set_param("msm_chunk", 0)


def test_g1_msm_repeated_negated_infinity_points(zk):
    n = 512
    base = O.g1_from_scalars(O.fr_random(10, 8))
    pts = np.empty((n, 8), dtype=np.uint64)
    for i in range(n):
        pts[i] = base[i % 8]
    # negate every third point (y -> -y), make every fifth the point at infinity (0,0)
    neg_y = O.fp_sub(O.fp_from_ints([0] * n), pts[:, 4:8])
    for i in range(0, n, 3):
        pts[i, 4:8] = neg_y[i]
    for i in range(0, n, 5):
        pts[i] = 0
    sc = O.fr_from_ints([(i % 4) + 1 for i in range(n)])  # few distinct scalars: P+P and P+(-P) inside buckets
    assert _g1_eq(zk.msm_g1(pts, sc), O.g1_msm(pts, sc))
    sc = O.fr_random(11, n)
    assert _g1_eq(zk.msm_g1(pts, sc), O.g1_msm(pts, sc))


def test_g1_msm_trapdoor_linearity_2_16(zk):
    # size-independent property: points s_i*G => MSM == (sum s_i w_i) * G
    n = 1 << 16
    s = O.fr_random(12, n)
    pts = O.g1_from_scalars(s)
    w = O.fr_random(13, n)
    expect = O.g1_from_scalars(O.fr_dot(s, w).reshape(1, 4))[0]
    assert _g1_eq(zk.msm_g1(pts, w), expect)


def test_g1_msm_empty(zk):
    out = zk.msm_g1(np.zeros((0, 8), np.uint64), np.zeros((0, 4), np.uint64))
    assert not O.g1_jac_to_affine(out).any()


@pytest.mark.parametrize("n", [1, 5, 64, 300, 2048])
def test_g2_msm_random(zk, n):
    sc = O.fr_random(300 + n, n)
    pts = O.g2_from_scalars(O.fr_random(400 + n, n))
    assert _g2_eq(zk.msm_g2(pts, sc), O.g2_msm(pts, sc))


def test_g2_msm_edge(zk):
    n = 256
    base = O.g2_from_scalars(O.fr_random(14, 4))
    pts = np.empty((n, 16), dtype=np.uint64)
    for i in range(n):
        pts[i] = base[i % 4]
    for i in range(0, n, 7):
        pts[i] = 0
    sc = O.fr_from_ints([(i % 3) for i in range(n)])
    assert _g2_eq(zk.msm_g2(pts, sc), O.g2_msm(pts, sc))
    s = O.fr_random(15, n)
    pts = O.g2_from_scalars(s)
    w = O.fr_random(16, n)
    expect = O.g2_from_scalars(O.fr_dot(s, w).reshape(1, 4))[0]
    assert _g2_eq(zk.msm_g2(pts, w), expect)


def test_g1_msm_split_matches_unsplit(zk):
    """config 5: the MSM sharded by contiguous index ranges over 8 'ranks' (emulated on one device) and combined
    with the host-side Jacobian sum equals the unsplit result"""
    import zkpor
    n = 20000
    sc = O.fr_random(31, n)
    pts = O.g1_from_scalars(O.fr_random(32, n))
    dp = zk.alloc(64 * n).upload(pts); ds = zk.alloc(32 * n).upload(sc)
    try:
        parts = np.stack([zkpor.msm_split_g1(zk, dp.ptr, ds.ptr, n, r, 8) for r in range(8)])
        assert _g1_eq(zkpor.g1_jac_sum(parts), O.g1_msm(pts, sc))
    finally:
        dp.free(); ds.free()


def test_g1_public_known_answer(zk):
    """2 * (1, 2) = the EIP-196 doubling vector, as a one-point MSM and as G + G (two points, unit scalars)"""
    g = O.g1_from_scalars(O.fr_from_ints([1]))
    want = np.concatenate([O.fp_from_ints([0x030644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd3]),
                           O.fp_from_ints([0x15ed738c0e0a7c92e7845f96b2ae9c0a68a6a449e3538fc7ff3ebf7a5a18a2c4])]).reshape(8)
    assert _g1_eq(zk.msm_g1(g, O.fr_from_ints([2])), want)
    assert _g1_eq(zk.msm_g1(np.concatenate([g, g]), O.fr_from_ints([1, 1])), want)
