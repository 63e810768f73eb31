"""-m gpu: the device generators of the structured witness (include/zkpor.h zkpor_witgen_*) against the oracle.
  * Poseidon S-box traces (the x^2, x^4, x^5 wires of every S-box of the in-circuit gadget) for the four widths the circuit uses —
    3 (Merkle levels, circuit/utils.go:12-21), 5 and 6 (asset-id hash, account leaf: batch_create_user_circuit.go:181,270), 13 (the
    full sponge blocks of the asset / CEX commitments, circuit/utils.go:28-49) — bit-exact with the oracle's plain HADES permutation
    with tracing; the device runs the optimised partial rounds, so this also pins that its S-box inputs are the true ones.
  * 16-bit range-check limbs, table multiplicities, the inverse wires of the log-derivative argument, the slot -> wire scatter:
    against Python integers."""
import numpy as np
import pytest

import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("t,count", [(3, 1), (3, 1000), (5, 257), (6, 64), (13, 1), (13, 300)])
def test_poseidon_trace_matches_oracle(zk, t, count):
    states = O.fr_random(100 + t, count * t).reshape(count, t, 4)
    states[:, 0] = O.fr_from_ints([0])[0]                       # the capacity element of a first block
    if count > 2:
        states[1, 0] = O.fr_random(7, 1)[0]                     # a chained block: capacity carried over
        states[2] = O.fr_from_ints([0] * t).reshape(t, 4)       # the all-zero state
    ns = zk.witgen_poseidon_sboxes(t)
    assert ns == 8 * t + {3: 57, 5: 60, 6: 60, 13: 65}[t]
    ref_st, ref_tr = O.poseidon_permute_trace(states, t)
    got_st, got_tr = zk.witgen_poseidon_trace(states, t)
    assert got_tr.shape == (3 * ns, count, 4)
    assert np.array_equal(got_st, ref_st)
    assert np.array_equal(got_tr, ref_tr)
    # the traced permutation IS the hash the tree uses: state[1] of [0, l, r] = the 2-to-1 node hash
    if t == 3:
        node = O.poseidon_hash(np.stack([states[0, 1], states[0, 2]]))
        assert np.array_equal(got_st[0, 1], node)
    # x^4 = (x^2)^2 and x^5 relate as the gadget's three constraints say (spot check of slot order)
    x2, x4 = got_tr[0::3], got_tr[1::3]
    assert np.array_equal(O.fr_mul(x2.reshape(-1, 4), x2.reshape(-1, 4)), x4.reshape(-1, 4))


def test_poseidon_trace_rejects_other_widths(zk):
    import zkpor
    assert zk.witgen_poseidon_sboxes(4) == 0
    with pytest.raises(zkpor.ZkporError):
        zk.witgen_poseidon_trace(O.fr_random(1, 4), 4)


@pytest.mark.parametrize("nb_limbs,n", [(1, 10), (4, 5000), (8, 1000), (15, 100)])
def test_limbs_and_multiplicities(zk, nb_limbs, n):
    rng = np.random.default_rng(nb_limbs)
    vals = [int(rng.integers(0, 1 << 62)) * int(rng.integers(0, 1 << 62)) * int(rng.integers(0, 1 << 62)) * int(rng.integers(1, 1 << 62)) % (1 << (16 * nb_limbs)) for _ in range(n)]
    vals[0] = 0
    vals[1 % n] = (1 << (16 * nb_limbs)) - 1
    limbs, mult, bad = zk.witgen_limbs(O.fr_from_ints(vals), nb_limbs)
    assert bad == 0
    exp_mult = np.zeros(65536, np.int64)
    for l in range(nb_limbs):
        got = O.fr_to_ints(limbs[l])
        exp = [(v >> (16 * l)) & 0xffff for v in vals]
        assert got == exp
        np.add.at(exp_mult, exp, 1)
    assert np.array_equal(mult.astype(np.int64), exp_mult)
    assert int(mult.sum()) == nb_limbs * n


def test_limbs_out_of_range_counted(zk):
    vals = [5, 1 << 64, (1 << 64) - 1, O.R_MOD - 1, 1 << 200]
    _, _, bad = zk.witgen_limbs(O.fr_from_ints(vals), 4)
    assert bad == 3


@pytest.mark.parametrize("n", [1, 7, 8, 9, 4099])
def test_inverse_wires(zk, n):
    vals = O.fr_random(50 + n, n)
    ch = O.fr_random(99, 1)[0]
    if n > 3:
        vals[3] = ch                                        # a zero denominator: reported, output 0, the rest unaffected
    out, bad = zk.witgen_inverse(vals, ch)
    assert bad == (1 if n > 3 else 0)
    den = O.fr_sub(np.tile(ch, (n, 1)), vals)
    prod = O.fr_to_ints(O.fr_mul(out, den))
    for i, p in enumerate(prod):
        assert p == (0 if (n > 3 and i == 3) else 1)


def test_scatter_to_wire_ids(zk):
    n, nw = 1000, 5000
    rng = np.random.default_rng(3)
    ids = rng.permutation(nw)[:n].astype(np.uint32)
    src = O.fr_random(8, n)
    w0 = O.fr_random(9, nw)
    dw = zk.alloc(w0.nbytes).upload(w0); ds = zk.alloc(src.nbytes).upload(src); di = zk.alloc(ids.nbytes).upload(ids)
    try:
        zk.witgen_scatter_dev(dw.ptr, ds.ptr, di.ptr, n)
        w = dw.download(np.uint64, (nw, 4))
    finally:
        dw.free(); ds.free(); di.free()
    exp = w0.copy(); exp[ids] = src
    assert np.array_equal(w, exp)


@pytest.mark.parametrize("nbits,n", [(1, 5), (8, 1000), (64, 300), (128, 257), (254, 10)])
def test_bit_decompositions(zk, nbits, n):
    rng = np.random.default_rng(nbits)
    vals = [int.from_bytes(rng.bytes(32), "big") % min(1 << nbits, O.R_MOD) for _ in range(n)]
    vals[0] = 0
    vals[-1] = min((1 << nbits) - 1, O.R_MOD - 1)
    bits, bad = zk.witgen_bits(O.fr_from_ints(vals), nbits)
    assert bad == 0
    for b in range(nbits):
        assert O.fr_to_ints(bits[b]) == [(v >> b) & 1 for v in vals]
    if nbits < 254:
        _, bad = zk.witgen_bits(O.fr_from_ints([1 << nbits, 1, (1 << nbits) + 5]), nbits)
        assert bad == 2


def test_lookup_results(zk):
    table = O.fr_random(21, 2500)                       # one user's 2500-entry asset table (circuit/batch_create_user_circuit.go:154-161)
    rng = np.random.default_rng(4)
    idx = [int(x) for x in rng.integers(0, 2500, size=4000)]
    idx[0] = 0; idx[1] = 2499
    out, bad = zk.witgen_gather(table, O.fr_from_ints(idx))
    assert bad == 0 and np.array_equal(out, table[idx])
    out, bad = zk.witgen_gather(table, O.fr_from_ints([5, 2500, O.R_MOD - 1, 1 << 40]))
    assert bad == 3 and np.array_equal(out[0], table[5]) and not out[1:].any()


def test_integer_division_by_the_percentage_multiplier(zk):
    """circuit.IntegerDivision(dividend, utils.PercentageMultiplier) over 136-bit dividends: quotient and remainder as big.Int.DivMod gives them,
    and the identity the circuit asserts (circuit/utils.go:175): q * 100 + rem == dividend"""
    rng = np.random.default_rng(8)
    vals = [int.from_bytes(rng.bytes(17), "big") for _ in range(3000)] + [0, 99, 100, 101, (1 << 136) - 1, O.R_MOD - 1]
    q, rem = zk.witgen_divmod_small(O.fr_from_ints(vals), 100)
    assert O.fr_to_ints(q) == [v // 100 for v in vals] and O.fr_to_ints(rem) == [v % 100 for v in vals]
    q7, r7 = zk.witgen_divmod_small(O.fr_from_ints(vals), 4294967295)
    assert O.fr_to_ints(q7) == [v // 4294967295 for v in vals] and O.fr_to_ints(r7) == [v % 4294967295 for v in vals]
    import zkpor
    with pytest.raises(zkpor.ZkporError):
        zk.witgen_divmod_small(O.fr_from_ints([1]), 0)
