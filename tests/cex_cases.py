"""Shared synthetic CEX asset tables for the commitment tests, and a Python big-integer restatement of
ConvertAssetInfoToBytes / ConvertTierRatiosToBytes (src/utils/utils.go:26-88) used to check the oracle."""
import numpy as np

import oracle as O

MAX_BOUNDARY = 1 << 118      # utils.MaxTierBoundaryValue (constants.go:29): the padding value of PaddingTierRatios


def make_assets(n_assets, seed, n_real_tiers=(0, 1, 5, 12)):
    rng = np.random.default_rng(seed)
    consts = np.zeros(n_assets, dtype=O.CEX_CONST_DTYPE)
    for a in range(n_assets):
        consts[a]["base_price"] = int(rng.integers(0, 1 << 60)) if a % 7 else 0          # reserved slots have price 0
        for g, group in enumerate(("loan", "margin", "portfolio_margin")):
            real = n_real_tiers[(a + g) % len(n_real_tiers)]
            prev = 0
            for i in range(12):
                if i < real:
                    prev = prev + (int(rng.integers(1, 1 << 50)) << 50) + int(rng.integers(0, 1 << 50))
                    b, ratio = min(prev, MAX_BOUNDARY - 1), int(rng.integers(0, 101))
                else:
                    b, ratio = MAX_BOUNDARY, 0                                          # PaddingTierRatios (utils.go:348-369)
                consts[a][group][i]["boundary"] = (b & ((1 << 64) - 1), b >> 64)
                consts[a][group][i]["ratio"] = ratio
    return consts


def make_totals(n_states, n_assets, seed):
    rng = np.random.default_rng(seed)
    t = np.zeros((n_states, n_assets), dtype=O.CEX_TOTALS_DTYPE)
    for name in O.CEX_TOTALS_DTYPE.names:
        t[name] = rng.integers(0, 1 << 63, size=(n_states, n_assets), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(n_states, n_assets), dtype=np.uint64)
    t[0, 0] = (0, 0, 0, 0, 0)
    t[-1, -1] = ((1 << 64) - 1,) * 5
    return t


def elements_bigint(consts, totals_row):
    """the 20 integers per asset exactly as the Go code builds them"""
    out = []
    for a in range(consts.shape[0]):
        c, t = consts[a], totals_row[a]
        out.append(int(t["total_equity"]) * (1 << 128) + int(t["total_debt"]) * (1 << 64) + int(c["base_price"]))
        out.append(int(t["loan_collateral"]) * (1 << 128) + int(t["margin_collateral"]) * (1 << 64) + int(t["portfolio_margin_collateral"]))
        for group in ("loan", "margin", "portfolio_margin"):
            tr = c[group]
            for i in range(0, 12, 2):
                b0 = int(tr[i]["boundary"][0]) | (int(tr[i]["boundary"][1]) << 64)
                b1 = int(tr[i + 1]["boundary"][0]) | (int(tr[i + 1]["boundary"][1]) << 64)
                out.append(int(tr[i]["ratio"]) + b0 * 256 + int(tr[i + 1]["ratio"]) * (1 << 126) + b1 * (1 << 134))
    return out
