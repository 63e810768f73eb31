"""Readers for the reference's sample DATA files kept under tests/golden/ (src/sampledata/cex_assets_info.csv,
sample_users0.csv, sample_users1.csv — the inputs of its TestParseUserDataSet, src/utils/utils_test.go:138-177), restating
the parsing rules of src/utils/utils.go: ConvertFloatStrToUint64 (:687-701: decimal * multiplier, truncated, must be a
uint64), ParseCexAssetInfoFromFile (:436-506: price x 10^8, or 10^14 for the two-digit assets), ParseTiersRatioFromStr
(:371-418: "lo-hi:ratio" lists, boundaries x 10^16, padded to 12 tiers with boundary 2^118 / ratio 0) and
ReadUserDataFromCsvFile (:508-640: six columns per asset, balances x 10^8 or x 10^2)."""
import csv
import os
from decimal import Decimal

import numpy as np

import oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
TWO_DIGITS = {"bttc", "shib", "lunc", "xec", "win", "bidr", "spell", "hot", "doge", "pepe", "floki", "idrt", "dogs", "bonk", "1000sats",
              "neiro", "1000pepper", "not", "nft", "bome", "1mbabydoge", "hmstr"}   # utils.AssetTypeForTwoDigits (constants.go:45-90)
MAX_BOUNDARY = 1 << 118


def to_uint64(text, multiplier):
    """ConvertFloatStrToUint64; None where the reference returns an error"""
    if text == "0.0":
        return 0
    try:
        v = int(Decimal(text) * multiplier)      # decimal.BigInt() truncates toward zero
    except Exception:
        return None
    return v if 0 <= v < (1 << 64) else None


def parse_tiers(enc):
    enc = enc.strip("[]")
    tiers = []
    if enc:
        for part in enc.split(","):
            rng, ratio = part.strip().split(":")
            lo, hi = rng.split("-")
            tiers.append((to_uint64(hi.strip(), 1) * 10 ** 16, to_uint64(ratio.strip(), 1)))
    return tiers + [(MAX_BOUNDARY, 0)] * (12 - len(tiers))


def load_cex_assets(path=os.path.join(HERE, "golden", "reference_cex_assets_info.csv")):
    """(symbols, consts[CEX_CONST_DTYPE]) in file order = asset index order"""
    rows = list(csv.reader(open(path)))[1:]
    consts = np.zeros(len(rows), dtype=O.CEX_CONST_DTYPE)
    symbols = []
    for i, r in enumerate(rows):
        sym = r[0].lower(); symbols.append(sym)
        consts[i]["base_price"] = to_uint64(r[1], 10 ** 14 if sym in TWO_DIGITS else 10 ** 8)
        for group, col in (("loan", 2), ("margin", 3), ("portfolio_margin", 4)):
            for t, (b, ratio) in enumerate(parse_tiers(r[col])):
                consts[i][group][t]["boundary"] = (b & ((1 << 64) - 1), b >> 64)
                consts[i][group][t]["ratio"] = ratio
    return symbols, consts


def load_users(path, symbols):
    """(accounts[ACCOUNT_DTYPE] with asset ranges, assets[ASSET_DTYPE], parse_ok[n]) — every row of the file, in order"""
    rows = list(csv.reader(open(path)))
    n_assets = (len(rows[0]) - 3) // 6
    rows = rows[1:]
    acc = np.zeros(len(rows), dtype=O.ACCOUNT_DTYPE)
    assets = []
    ok = np.ones(len(rows), dtype=bool)
    for i, r in enumerate(rows):
        acc[i]["id_be"] = np.frombuffer(bytes.fromhex(r[1]), dtype=np.uint8)
        acc[i]["asset_off"] = len(assets)
        mine = []
        for j in range(n_assets):
            mult = 100 if symbols[j] in TWO_DIGITS else 10 ** 8
            vals = [to_uint64(r[j * 6 + c], mult) for c in (2, 3, 5, 6, 7)]
            if any(v is None for v in vals):
                ok[i] = False
                break
            eq, debt, loan, margin, pm = vals
            if eq != 0 or debt != 0:
                mine.append((eq, debt, loan, margin, pm, j, 0))
        if ok[i]:
            assets.extend(mine)
            acc[i]["n_assets"] = len(mine)
    arr = np.zeros(len(assets), dtype=O.ASSET_DTYPE)
    for k, a in enumerate(assets):
        arr[k] = a
    return acc, arr, ok


def fixture_account(cfg):
    """the account of src/verifier/config/user_config.json (tests/golden/reference_user_config.json) in packed form"""
    acc = np.zeros(1, dtype=O.ACCOUNT_DTYPE)
    acc[0]["n_assets"] = len(cfg["Assets"])
    acc[0]["id_be"] = np.frombuffer(bytes.fromhex(cfg["AccountIdHash"]), dtype=np.uint8)
    assets = np.zeros(len(cfg["Assets"]), dtype=O.ASSET_DTYPE)
    for k, a in enumerate(cfg["Assets"]):
        assets[k] = (a["Equity"], a["Debt"], a["Loan"], a["Margin"], a["PortfolioMargin"], a["Index"], 0)
    return acc, assets


def u128(pair):
    return int(pair[0]) | (int(pair[1]) << 64)


def fixture_leaf_inputs(cfg, n_assets=350):
    """The element lists behind the account leaf of tests/golden/reference_user_config.json.  The fixture was produced by an
    EARLIER revision of the reference (found by search): 350 assets, five u64 per asset in positional
    order with no index field — (equity, debt, loan, margin, portfolio margin) — packed three per element as
    a * 2^128 + b * 2^64 + c, hashed as one chained Poseidon; leaf = Poseidon(id, TotalEquity, TotalDebt, TotalCollateral,
    commitment).  The field layout is history; what it pins is the hash: 584 elements = 48 full blocks of 12 + a block of 8."""
    by_index = {a["Index"]: a for a in cfg["Assets"]}
    flat = []
    for i in range(n_assets):
        a = by_index.get(i)
        flat += [a["Equity"], a["Debt"], a["Loan"], a["Margin"], a["PortfolioMargin"]] if a else [0, 0, 0, 0, 0]
    flat += [0] * ((-len(flat)) % 3)
    elements = [flat[i] * (1 << 128) + flat[i + 1] * (1 << 64) + flat[i + 2] for i in range(0, len(flat), 3)]
    leaf_head = [int(cfg["AccountIdHash"], 16), cfg["TotalEquity"], cfg["TotalDebt"], cfg["TotalCollateral"]]
    return elements, leaf_head


def load_cex_assets_500(path=os.path.join(HERE, "golden", "reference_cex_assets_info_483.csv")):
    """the production-sized table of src/utils/cex_assets_info.csv (TestParseCexAssetInfoFromFile: 483 real assets), filled up
    to utils.AssetCounts = 500 with the reserved entries of ParseCexAssetInfoFromFile (:494-503): price 0, padding tiers"""
    symbols, consts = load_cex_assets(path)
    full = np.zeros(500, dtype=O.CEX_CONST_DTYPE)
    full[:len(symbols)] = consts
    for i in range(len(symbols), 500):
        for group in ("loan", "margin", "portfolio_margin"):
            for t in range(12):
                full[i][group][t]["boundary"] = (MAX_BOUNDARY & ((1 << 64) - 1), MAX_BOUNDARY >> 64)
    return symbols + ["reserved"] * (500 - len(symbols)), full
