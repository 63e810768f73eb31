"""-m gpu: CEX asset-list commitments and batch commitments on the device (SURVEY.md §8 a12 / f3;
src/witness/witness/witness.go:159-198, src/utils/utils.go:26-88,779-800) bit-exact with the oracle."""
import numpy as np
import pytest

import cex_cases as C
import oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_assets,n_states", [(1, 1), (9, 3), (500, 2), (37, 130)])
def test_cex_commitments_match_oracle(zk, n_assets, n_states):
    consts = C.make_assets(n_assets, seed=10 + n_assets)
    totals = C.make_totals(n_states, n_assets, seed=20 + n_states)
    got = zk.cex_commitments(consts, totals)
    ref = O.fr_to_be(O.cex_commitments(consts, totals))
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("mode", [0, 1])
def test_cex_commitments_one_thread_and_sixteen_lanes_per_state(zk, mode):
    """the 834-permutation chain of a production-size state per thread ("poseidon_coop" 0) and across sixteen lanes (1): the same commitments"""
    zk.set_param("poseidon_coop", mode)
    try:
        for n_assets, n_states in ((500, 3), (7, 5)):
            consts = C.make_assets(n_assets, seed=40 + n_assets); totals = C.make_totals(n_states, n_assets, seed=50 + n_states)
            assert np.array_equal(zk.cex_commitments(consts, totals), O.fr_to_be(O.cex_commitments(consts, totals)))
    finally:
        zk.set_param("poseidon_coop", -1)


def test_commitment_depends_on_every_field(zk):
    consts = C.make_assets(5, seed=1)
    totals = C.make_totals(1, 5, seed=2)
    base = zk.cex_commitments(consts, totals)[0].tobytes()
    for name in O.CEX_TOTALS_DTYPE.names:
        t2 = totals.copy(); t2[0, 3][name] ^= np.uint64(1)
        assert zk.cex_commitments(consts, t2)[0].tobytes() != base
    c2 = consts.copy(); c2[4]["margin"][11]["ratio"] ^= 1
    assert zk.cex_commitments(c2, totals)[0].tobytes() != base
    c3 = consts.copy(); c3[0]["base_price"] += np.uint64(1)
    assert zk.cex_commitments(c3, totals)[0].tobytes() != base


def test_batch_commitments_match_oracle(zk):
    n = 70
    roots = O.fr_to_be(O.fr_random(1, n)); before = O.fr_to_be(O.fr_random(2, n)); after = O.fr_to_be(O.fr_random(3, n))
    mn = np.arange(n, dtype=np.uint32) * 1380; mx = mn + 1379
    mn[0] = 0                                                     # the [0x00] case of witness.go:186-193
    mx[-1] = 0xFFFFFFFF
    got = zk.batch_commitments(roots, before, after, mn, mx)
    for i in (0, 1, n // 2, n - 1):
        el = np.concatenate([O.fr_from_be(roots[i:i + 1]), O.fr_from_be(before[i:i + 1]), O.fr_from_be(after[i:i + 1]),
                             O.fr_from_ints([int(mn[i]), int(mx[i])])])
        assert got[i].tobytes() == O.fr_to_be(O.poseidon_hash(el)[None, :])[0].tobytes()


def test_production_asset_table_on_the_device(zk):
    """the reference's 483-asset table (src/utils/cex_assets_info.csv, padded to 500): commitments of a few CEX states and
    the totals of random accounts over its real tier lists, device = oracle"""
    import refdata as R
    import zkpor
    _, consts = R.load_cex_assets_500()
    totals = C.make_totals(3, 500, seed=8)
    assert np.array_equal(zk.cex_commitments(consts, totals), O.fr_to_be(O.cex_commitments(consts, totals)))
    rng = np.random.default_rng(4)
    n_acc = 2000
    acc = np.zeros(n_acc, dtype=zkpor.ACCOUNT_DTYPE)
    k = rng.integers(0, 51, size=n_acc)
    off = np.concatenate([[0], np.cumsum(k)[:-1]])
    acc["n_assets"] = k; acc["asset_off"] = off
    assets = np.zeros(int(k.sum()), dtype=zkpor.ASSET_DTYPE)
    for i in range(n_acc):
        assets["index"][off[i]:off[i] + k[i]] = np.sort(rng.choice(483, size=k[i], replace=False))
    eq = rng.integers(0, 1 << 46, size=assets.shape[0], dtype=np.uint64)
    assets["equity"] = eq; assets["debt"] = rng.integers(0, 1 << 34, size=assets.shape[0], dtype=np.uint64)
    assets["loan"] = eq // np.uint64(3); assets["margin"] = eq // np.uint64(5); assets["portfolio_margin"] = eq // np.uint64(7)
    got, valid, _ = zk.account_totals(acc, assets, consts)
    ref, ref_valid = O.account_totals(acc, assets, consts)
    assert np.array_equal(valid, ref_valid)
    for field in ("equity", "debt", "collateral"):
        assert np.array_equal(got[field], ref[field])
