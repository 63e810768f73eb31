"""-m gpu: account totals and collateral tier claims on the device (zkpor_account_totals; src/utils/utils.go:608-615,648-685,
circuit/utils.go:227-278) bit-exact with the oracle, through the reference's own 21-row tier table, and feeding the leaf hash."""
import json
import os

import numpy as np
import pytest

import cex_cases as C
import oracle as O
import zkpor

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _accounts(n_acc, n_cex, seed, max_assets=12):
    rng = np.random.default_rng(seed)
    acc = np.zeros(n_acc, dtype=zkpor.ACCOUNT_DTYPE)
    k = rng.integers(0, max_assets + 1, size=n_acc)
    off = np.concatenate([[0], np.cumsum(k)[:-1]])
    acc["n_assets"] = k; acc["asset_off"] = off
    acc["id_be"] = rng.integers(0, 256, size=(n_acc, 32), dtype=np.uint8); acc["id_be"][:, 0] &= 0x0F
    assets = np.zeros(int(k.sum()), dtype=zkpor.ASSET_DTYPE)
    for i in range(n_acc):
        sl = slice(off[i], off[i] + k[i])
        assets["index"][sl] = np.sort(rng.choice(n_cex, size=k[i], replace=False))
    eq = rng.integers(0, 1 << 44, size=assets.shape[0], dtype=np.uint64)
    assets["equity"] = eq
    assets["debt"] = rng.integers(0, 1 << 30, size=assets.shape[0], dtype=np.uint64)
    third = eq // np.uint64(3)
    assets["loan"] = third; assets["margin"] = third // np.uint64(2); assets["portfolio_margin"] = third // np.uint64(4)
    return acc, assets


def test_totals_match_oracle_and_feed_the_leaf_hash(zk):
    n_cex = 40
    consts = C.make_assets(n_cex, seed=2)
    # prices small enough that position values fall inside the tier ranges, large enough to cross several tiers
    consts["base_price"] = np.random.default_rng(3).integers(1, 1 << 56, size=n_cex, dtype=np.uint64)
    acc, assets = _accounts(3000, n_cex, seed=5)
    # a few accounts that must be flagged invalid: collateral above equity, debt above collateral
    a0 = int(acc["asset_off"][np.argmax(acc["n_assets"] > 0)])
    assets["loan"][a0] = assets["equity"][a0] + np.uint64(1)
    last = int(np.nonzero(acc["n_assets"] > 0)[0][-1]); a1 = int(acc["asset_off"][last])
    assets["debt"][a1] = np.uint64((1 << 63)); assets["loan"][a1] = 0; assets["margin"][a1] = 0; assets["portfolio_margin"][a1] = 0
    got, valid, tiers = zk.account_totals(acc, assets, consts, want_tiers=True)
    ref, ref_valid = O.account_totals(acc, assets, consts)
    for name in ("equity", "debt", "collateral"):
        assert np.array_equal(got[name], ref[name]), name
    assert np.array_equal(valid, ref_valid) and valid.sum() < valid.size and valid.sum() > valid.size // 2
    # tier claims against the oracle's per-position query
    rng = np.random.default_rng(9)
    for j in rng.choice(assets.shape[0], size=60, replace=False):
        c = consts[int(assets["index"][j])]
        for g, (group, field) in enumerate((("loan", "loan"), ("margin", "margin"), ("portfolio_margin", "portfolio_margin"))):
            tl = [(int(t["boundary"][0]) | (int(t["boundary"][1]) << 64), int(t["ratio"])) for t in c[group]]
            idx, flag, _, _ = O.tier_query(tl, int(assets[field][j]) * int(c["base_price"]))
            assert (int(tiers[j, 2 * g]), int(tiers[j, 2 * g + 1])) == (idx, flag)
    # the totals are what the leaf hash consumes
    sel = np.nonzero(valid)[0][:64]
    sub = got[sel].copy()
    sub_assets = np.concatenate([assets[int(o):int(o) + int(k)] for o, k in zip(sub["asset_off"], sub["n_assets"])]) if sub["n_assets"].sum() else assets[:0]
    sub["asset_off"] = np.concatenate([[0], np.cumsum(sub["n_assets"])[:-1]])
    leaves = zk.poseidon_leaves(sub, sub_assets, 50)
    assert np.array_equal(leaves, O.fr_to_be(O.account_leaves(sub, sub_assets, 50)))


def test_reference_tier_table_through_the_device(zk):
    """the 21-row table of circuit/get_and_check_tier_ratios_query_results_test.go:145-170 (tests/golden/): one account
    per row with a single loan position; short tier lists are extended the way PaddingTierRatios does (boundary 2^118,
    ratio 0), which leaves every in-range claim of the table unchanged"""
    d = json.load(open(os.path.join(HERE, "golden", "collateral_tier_cases.json")))
    MAX = int(d["max_tier_boundary"])
    rows = [c for c in d["cases"] if int(c["collateral"]) < (1 << 64)]
    consts = np.zeros(len(rows), dtype=O.CEX_CONST_DTYPE)
    acc = np.zeros(len(rows), dtype=zkpor.ACCOUNT_DTYPE)
    assets = np.zeros(len(rows), dtype=zkpor.ASSET_DTYPE)
    for i, c in enumerate(rows):
        consts[i]["base_price"] = d["price"]
        for group in ("loan", "margin", "portfolio_margin"):
            for t in range(12):
                b, r = (c["tiers"][t] if t < len(c["tiers"]) else (MAX, 0))
                consts[i][group][t]["boundary"] = (b & ((1 << 64) - 1), b >> 64); consts[i][group][t]["ratio"] = r
        acc[i]["n_assets"] = 1; acc[i]["asset_off"] = i
        assets[i]["index"] = i; assets[i]["loan"] = int(c["collateral"]); assets[i]["equity"] = int(c["collateral"])
    got, valid, tiers = zk.account_totals(acc, assets, consts, want_tiers=True)
    for i, c in enumerate(rows):
        n = len(c["tiers"])
        idx, flag = int(tiers[i, 0]), int(tiers[i, 1])
        # beyond the listed tiers the padded list continues with boundary 2^118: "above every listed boundary" shows up as
        # index n (flag 0) there, which is the unpadded list's (n - 1, flag 1)
        claim = (n - 1, 1) if idx >= n else (idx, flag)
        assert (claim == (c["index"], c["flag"])) == (not c["expect_fail"]), c["name"]
        if not c["expect_fail"]:
            tl = [tuple(t) for t in c["tiers"]]
            want = O.tier_query(tl, int(c["collateral"]) * d["price"])[2]
            assert int(got[i]["collateral"][0]) | (int(got[i]["collateral"][1]) << 64) == want, c["name"]


def test_reference_sample_data_on_the_device(zk):
    """the same golden data as tests/test_oracle_cpu.py::test_reference_sample_data_totals_and_validity, through
    zkpor_account_totals: the fixture account's three totals and the 90 / 10 and 80 / 20 valid / invalid counts"""
    import refdata as R
    symbols, consts = R.load_cex_assets()
    cfg = json.load(open(os.path.join(HERE, "golden", "reference_user_config.json")))
    acc, assets = R.fixture_account(cfg)
    tot, valid, _ = zk.account_totals(acc, assets, consts)
    assert (R.u128(tot[0]["equity"]), R.u128(tot[0]["debt"]), R.u128(tot[0]["collateral"])) == (cfg["TotalEquity"], cfg["TotalDebt"], cfg["TotalCollateral"])
    assert valid[0] == 1
    for name, want in (("reference_sample_users0.csv", (90, 10)), ("reference_sample_users1.csv", (80, 20))):
        acc, assets, parsed = R.load_users(os.path.join(HERE, "golden", name), symbols)
        got, valid, _ = zk.account_totals(acc, assets, consts)
        ref, ref_valid = O.account_totals(acc, assets, consts)
        good = valid.astype(bool) & parsed
        assert (int(good.sum()), int((~good).sum())) == want, name
        assert np.array_equal(valid, ref_valid)
        for field in ("equity", "debt", "collateral"):
            assert np.array_equal(got[field], ref[field])
