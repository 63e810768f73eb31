"""-m gpu: the digit-stream sort (csrc/sort.hip) alone, through `zkpor_msm_digits_dev`, against a Python-integer restatement of the signed-digit
decomposition.  What every multi-exponentiation takes on trust from it (gnark-crypto's partitionScalars + bucket walk, call site
src/prover/prover/prover.go:269): the multiset of (bucket key, point index | sign) entries is exactly the decomposition's, every bucket's entries
are contiguous (keys ascending), the absence flags ride in bits 30 / 31, and the entry counts the accumulation launches are sized by are right.
Order INSIDE a bucket is not part of the contract (a group law is commutative) and is not compared."""
import numpy as np
import pytest

import oracle as O
import zkpor

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True, params=[1, 0], ids=["staged", "lite"])
def _scatter_form(zk, request):
    """every case through both forms of the scatter passes: a tile's entries staged through LDS (whole runs per store) and straight to memory
    from a 4 KB workgroup ("sort_stage" 0)"""
    zk.set_param("sort_stage", request.param)
    yield
    zk.set_param("sort_stage", 1)


def _expected(ints, c, W, piece, bpw, tables, absent0=None, absent1=None):
    """the decomposition in Python integers: digit w of scalar i, signed, in (-2^(c-1), 2^(c-1)]"""
    half = 1 << (c - 1)
    out = []
    for i, s in enumerate(ints):
        carry = 0
        fl = (1 << 30 if absent0 is not None and absent0[i] else 0) | (1 << 31 if absent1 is not None and absent1[i] else 0)
        for w in range(W):
            d = (s & ((1 << c) - 1)) + carry
            s >>= c
            carry = 1 if d > half else 0
            if carry:
                d = (1 << c) - d
            if d:
                q = w // piece
                out.append((((w - q * piece) * bpw + d - 1) << 32) | (((i * tables + q) << 1) | carry | fl))
        assert carry == 0 and s == 0          # W digits cover 254 bits + the last carry
    return np.sort(np.array(out, dtype=np.uint64)) if out else np.zeros(0, np.uint64)


def _scalars(n, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "uniform":
        return [int(x) for x in O.fr_to_ints(O.fr_random(seed, n))]
    if kind == "witness":       # the mixture of a solved wire vector: half zeros, ones, small values, full-width residues
        full = O.fr_to_ints(O.fr_random(seed, n))
        sel = rng.integers(0, 100, n)
        return [0 if t < 45 else 1 if t < 55 else int(rng.integers(0, 1 << 16)) if t < 60 else int(f) for t, f in zip(sel, full)]
    if kind == "edges":
        base = [0, 1, 2, O.R_MOD - 1, O.R_MOD - 2, (1 << 253), (1 << 21), (1 << 21) + 1, (1 << 22) - 1, (1 << 22), (1 << 44) - 1, (1 << 128) + 1]
        return [base[i % len(base)] for i in range(n)]
    if kind == "ones":
        return [1] * n
    raise ValueError(kind)


def _run(zk, ints, tables, window, absent0=None, absent1=None):
    n = len(ints)
    sc = O.fr_from_ints(ints) if n else np.zeros((0, 4), np.uint64)
    buf = zk.alloc(max(32, 32 * n))
    if n:
        buf.upload(sc)
    zk.set_param("msm_window", window)
    try:
        return zk.msm_digits_dev(buf.ptr, n, tables, absent0, absent1)
    finally:
        zk.set_param("msm_window", 0)
        buf.free()


def _check(keys, vals, info, ints, tables, absent0=None, absent1=None):
    c, W, piece, bpw = info["c"], info["W"], info["piece"], info["bpw"]
    want = _expected(ints, c, W, piece, bpw, tables, absent0, absent1)
    assert info["entries"] == want.size == keys.size == vals.size
    assert bool(np.all(keys[1:] >= keys[:-1]))                                   # every bucket contiguous, buckets in key order
    assert keys.size == 0 or int(keys.max()) < piece * bpw
    got = np.sort((keys.astype(np.uint64) << np.uint64(32)) | vals.astype(np.uint64))
    assert np.array_equal(got, want)                                             # exactly the decomposition's entries, nothing lost, nothing twice
    if absent0 is not None:
        assert info["entries_group0"] == int(np.count_nonzero((vals >> 30) & 1 == 0))
    if absent1 is not None:
        assert info["entries_group1"] == int(np.count_nonzero((vals >> 31) & 1 == 0))


@pytest.mark.parametrize("kind", ["uniform", "witness", "edges", "ones"])
@pytest.mark.parametrize("n", [1, 2, 255, 256, 257, 2049, 6000])
def test_digit_stream_is_the_decomposition_grouped_by_bucket(zk, n, kind):
    ints = _scalars(n, 1000 + n, kind)
    for tables, window in ((1, 0), (4, 0), (1, 13), (4, 16)):
        keys, vals, info = _run(zk, ints, tables, window)
        _check(keys, vals, info, ints, tables)


@pytest.mark.parametrize("window,tables", [(2, 1), (3, 4), (5, 1), (8, 3), (9, 1), (10, 4), (17, 1), (18, 4), (19, 2), (20, 1), (22, 4), (23, 4), (24, 4)])
def test_every_level_count_of_the_sort(zk, window, tables):
    """window sizes that give 1, 2 and 3 sort levels, key spaces that are not a power of two (piece = 3, 5, 13 windows), 128 digits per scalar
    (several level-0 tiles per scalar block), sparse streams in a huge key space (most segments of the last level empty)"""
    ints = _scalars(1500, 77 + window, "witness") + _scalars(300, 78, "edges")
    keys, vals, info = _run(zk, ints, tables, window)
    assert info["c"] == window
    assert info["levels"] == max(1, -(-(info["piece"] * info["bpw"] - 1).bit_length() // 9))
    _check(keys, vals, info, ints, tables)


@pytest.mark.parametrize("window,tables", [(22, 4), (20, 1)])
def test_compile_time_level_0_equals_the_generic_one(zk, window, tables):
    """the production shapes have a level 0 with the window, the digit count and the table count as template parameters (digits by static shifts,
    kept in registers between the passes, zero scalars skipped before their Montgomery reduction); "sort_generic" 1 runs the runtime-window
    kernels on the same input: the same entries (both against the Python decomposition), blocks of zero scalars and a ragged tail included"""
    ints = _scalars(3000, 31, "witness") + [0] * 700 + _scalars(411, 32, "edges") + [0] * 5
    rng = np.random.default_rng(3)
    a0 = (rng.integers(0, 7, len(ints)) == 0).astype(np.uint8)
    got = {}
    for generic in (0, 1):
        zk.set_param("sort_generic", generic)
        try:
            keys, vals, info = _run(zk, ints, tables, window, a0, None)
        finally:
            zk.set_param("sort_generic", 0)
        _check(keys, vals, info, ints, tables, a0, None)
        got[generic] = np.sort((keys.astype(np.uint64) << np.uint64(32)) | vals.astype(np.uint64))
    assert np.array_equal(got[0], got[1])


def test_absence_flags_and_per_array_counts(zk):
    n = 5000
    ints = _scalars(n, 5, "witness")
    rng = np.random.default_rng(9)
    a0 = (rng.integers(0, 10, n) == 0).astype(np.uint8)
    a1 = (rng.integers(0, 4, n) == 0).astype(np.uint8)
    for absent0, absent1 in ((a0, a1), (a0, None), (None, a1)):
        keys, vals, info = _run(zk, ints, 4, 0, absent0, absent1)
        _check(keys, vals, info, ints, 4, absent0, absent1)


def test_empty_and_all_zero_inputs(zk):
    keys, vals, info = _run(zk, [], 1, 0)
    assert keys.size == 0 and info["entries"] == 0
    keys, vals, info = _run(zk, [0] * 1000, 4, 0)
    assert keys.size == 0 and info["entries"] == 0


def test_a_million_scalars_with_the_production_window(zk):
    """22-bit windows over 4 tables (the production configuration: 3 bucket windows of 2^21, 23 key bits, levels of 7 + 8 + 8 bits), 2^20 scalars
    of the witness mixture generated on the device: keys ascending, counts consistent, and the multiset checked through order-independent sums"""
    n = 1 << 20
    buf = zk.alloc(32 * n)
    zk.fill_fr(buf, n, 4242, 1)
    sc = buf.download(np.uint64, (n, 4))
    zk.set_param("msm_window", 22)
    try:
        keys, vals, info = zk.msm_digits_dev(buf.ptr, n, 4, cap=n * 12)
    finally:
        zk.set_param("msm_window", 0)
        buf.free()
    assert info["c"] == 22 and info["W"] == 12 and info["piece"] == 3 and info["levels"] == 3
    assert bool(np.all(keys[1:] >= keys[:-1])) and int(keys.max()) < 3 * (1 << 21)
    ints = O.fr_to_ints(sc[:3000])
    want = _expected([int(x) for x in ints], 22, 12, 3, 1 << 21, 4)
    pairs = (keys.astype(np.uint64) << np.uint64(32)) | vals.astype(np.uint64)
    first = np.sort(pairs[(vals >> 1) < 3000 * 4])                                # every entry of the first 3000 scalars, wherever the sort put it
    assert np.array_equal(first, want)
    # each scalar contributes at most one entry per (table, window) slot and the point index names scalar and table: (val >> 1, key // bpw) unique
    slot = (vals.astype(np.uint64) >> np.uint64(1)) * np.uint64(3) + (keys.astype(np.uint64) >> np.uint64(21))
    assert np.unique(slot).size == slot.size
    # the value of every scalar is recovered from its entries: sum over entries of +-(|d|) * 2^(c * w) = scalar (checked mod 2^64 on all 2^20)
    d = ((keys & ((1 << 21) - 1)).astype(np.int64) + 1) * np.where(vals & 1, -1, 1)
    w = ((vals >> 1) % 4).astype(np.int64) * 3 + (keys >> 21).astype(np.int64)
    idx = (vals >> 1) // 4
    lo = np.zeros(n, dtype=np.uint64)
    for ww in range(3):                                                            # digits 0..2 reach bit 66: enough for the low 64 bits
        m_ = w == ww
        np.add.at(lo, idx[m_], (d[m_].astype(np.uint64)) << np.uint64(22 * ww))
    canon = np.array([int(x) & 0xFFFFFFFFFFFFFFFF for x in O.fr_to_ints(sc[:: 997])], dtype=np.uint64)
    assert np.array_equal(lo[:: 997], canon)
