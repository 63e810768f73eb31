"""CPU: circuit witness assignment (host/witness_assign.hpp = circuit.SetBatchCreateUserCircuitWitness + calcAndSetCollateralInfo,
circuit/batch_create_user_circuit.go:334-436, circuit/utils.go:227-278) against a line-by-line Python restatement of the Go code on
the same decoded witness: the vector frontend.NewWitness would hand the solver (public first, struct declaration order)."""
import base64
import ctypes

import numpy as np
import pytest

import gobs2 as G
from test_dispatcher_cpu import host  # noqa: F401

R = 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001
TIERS = [50, 500]


def _big(b):
    return 0 if b is None else int.from_bytes(b[1:], "big")


def _dense(op, n_cex):
    dense = [{"Index": p, "Equity": 0, "Debt": 0, "Loan": 0, "Margin": 0, "PortfolioMargin": 0} for p in range(n_cex)]
    for a in op["Assets"]:
        dense[a["Index"]] = dict(a)
    return dense


def _empty(a):
    return not (a["Debt"] or a["Equity"] or a["Margin"] or a["PortfolioMargin"] or a["Loan"])


def restate(full):
    """the Go code, in its own order and with its own variable names"""
    n_cex = len(full["BeforeCexAssets"])
    be = lambda b: int.from_bytes(b, "big") % R
    out = [be(full["BatchCommitment"]), be(full["AccountTreeRoot"]), be(full["BeforeCEXAssetsCommitment"]), be(full["AfterCEXAssetsCommitment"]),
           full["MinAccountIndex"], full["MaxAccountIndex"]]
    for c in full["BeforeCexAssets"]:
        out += [c["TotalEquity"], c["TotalDebt"], c["BasePrice"], c["LoanCollateral"], c["MarginCollateral"], c["PortfolioMarginCollateral"]]
        for name in ("LoanRatios", "MarginRatios", "PortfolioMarginRatios"):
            for t in c[name]:
                out += [_big(t["BoundaryValue"]), t["Ratio"], _big(t["PrecomputedValue"])]
    ops = [dict(op, Assets=_dense(op, n_cex)) for op in full["CreateUserOps"]]
    count = sum(1 for a in ops[0]["Assets"] if not _empty(a))
    target = next(t for t in TIERS if count <= t)
    for op in ops:
        existing = [a["Index"] for a in op["Assets"] if not _empty(a)]
        padding = target - len(existing)
        infos, cur_pad, cur_idx = [], 0, 0
        for v in existing:
            if cur_pad < padding:
                for k in range(cur_idx, v):
                    cur_pad += 1
                    infos.append([k, 0, 0, 0, 0, 0, 0])
                    if cur_pad >= padding:
                        break
            um = op["Assets"][v]; p = full["BeforeCexAssets"][v]
            ua = [v]
            for amount, name in ((um["Loan"], "LoanRatios"), (um["Margin"], "MarginRatios"), (um["PortfolioMargin"], "PortfolioMarginRatios")):
                val = amount * p["BasePrice"]
                for i, t in enumerate(p[name]):
                    if val <= _big(t["BoundaryValue"]):
                        ua += [i, 0]
                        break
                else:
                    ua += [len(p[name]) - 1, 1]
            infos.append(ua)
            cur_idx = v + 1
        while len(infos) < target:
            infos.append([cur_idx, 0, 0, 0, 0, 0, 0]); cur_idx += 1
        for ua in infos:
            out += ua
        for a in op["Assets"]:
            out += [a["Equity"], a["Debt"], a["Loan"], a["Margin"], a["PortfolioMargin"]]
        out += [op["AccountIndex"], be(op["AccountIdHash"])] + [be(x) for x in op["AccountProof"]]
    return out, target


def _assign(host, col, cap=5_000_000):
    out = np.zeros((cap, 4), dtype=np.uint64)
    counts = (ctypes.c_uint64 * 3)()
    err = ctypes.create_string_buffer(256)
    tiers = (ctypes.c_int * len(TIERS))(*TIERS)
    host.zkh_witness_assign.restype = ctypes.c_long
    n = host.zkh_witness_assign(col, ctypes.c_size_t(len(col)), tiers, len(TIERS), out.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(cap), counts, err, ctypes.c_size_t(256))
    return n, out, list(counts), err.value.decode()


def _column(host, seed, users, assets, cex):
    out = ctypes.create_string_buffer(1 << 26)
    host.zkh_witness_synth_encode.restype = ctypes.c_long
    n = host.zkh_witness_synth_encode(ctypes.c_uint64(seed), users, assets, cex, 1, 2, out, ctypes.c_size_t(1 << 26))
    assert n > 0
    return out.raw[:n]


@pytest.mark.parametrize("users,assets", [(1, 1), (5, 7), (9, 50), (3, 51)])
def test_assignment_matches_the_go_code_restated(host, users, assets):
    cex = 500
    col = _column(host, 42 + users, users, assets, cex)
    n, out, counts, err = _assign(host, col)
    assert n > 0, err
    full = G.synth_witness(42 + users, users, assets, cex)
    # boundaries of the synthetic witness are random (not ascending): the claim search still follows the same rule on both sides
    want, target = restate(full)
    assert counts == [1, len(want) - 1, target] and n == len(want)
    assert target == (50 if assets <= 50 else 500)
    assert n == 1 + 5 + 114 * cex + users * (7 * target + 5 * cex + 30)          # the secret-input count formula
    got = [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in out[:n]]
    assert got == want


def test_first_user_decides_the_tier_and_overfull_users_are_refused(host):
    """targetCounts comes from CreateUserOps[0] (:363-366): a later user with more assets than that tier cannot be assigned
    (Go panics with index out of range on Assets[index])"""
    # hand-built: user 0 holds 1 asset, user 1 holds 51 -> tier 50 from user 0, user 1 does not fit
    def asset(i, eq):
        return {"Index": i, "Equity": eq}
    def user(idx, n):
        return {"Assets": [asset(i, 5 + i) for i in range(n)], "AccountIndex": idx, "AccountIdHash": bytes([idx]) * 32, "AccountProof": [bytes([k]) * 32 for k in range(28)]}
    cexs = [{"TotalEquity": 1, "BasePrice": 3, "Index": i, "LoanRatios": [{"BoundaryValue": b"\x02\x64", "Ratio": 50, "PrecomputedValue": b"\x02\x32"}] * 12}
            for i in range(500)]
    for n1, ok in ((50, True), (51, False)):
        w = {"BatchCommitment": b"\x01" * 32, "AccountTreeRoot": b"\x02" * 32, "BeforeCexAssets": cexs, "CreateUserOps": [user(0, 1), user(1, n1)]}
        B, U, S = 5, 3, 6
        types = {
            65: {"kind": "struct", "name": "BatchCreateUserWitness", "fields": [("BatchCommitment", B), ("AccountTreeRoot", B), ("BeforeCexAssets", 66), ("CreateUserOps", 71)]},
            66: {"kind": "slice", "name": "", "elem": 67},
            67: {"kind": "struct", "name": "CexAssetInfo", "fields": [("TotalEquity", U), ("BasePrice", U), ("Index", U), ("LoanRatios", 68)]},
            68: {"kind": "array", "name": "", "elem": 69, "len": 12},
            69: {"kind": "struct", "name": "TierRatio", "fields": [("BoundaryValue", 70), ("Ratio", U), ("PrecomputedValue", 70)]},
            70: {"kind": "gobenc", "name": "Int"},
            71: {"kind": "slice", "name": "", "elem": 72},
            72: {"kind": "struct", "name": "CreateUserOperation", "fields": [("Assets", 73), ("AccountIndex", U), ("AccountIdHash", B), ("AccountProof", 75)]},
            73: {"kind": "slice", "name": "", "elem": 74},
            74: {"kind": "struct", "name": "AccountAsset", "fields": [("Index", U), ("Equity", U)]},
            75: {"kind": "array", "name": "", "elem": B, "len": 28},
        }
        col = base64.b64encode(G.s2_literal_block(G.gob_encode(types, sorted(types), 65, w)))
        n, out, counts, err = _assign(host, col)
        if ok:
            assert n == 1 + 5 + 114 * 500 + 2 * (7 * 50 + 2500 + 30) and counts[2] == 50
            # user 0: asset 0 real (Loan 0 x price 3 = 0 <= 100: index 0, flag 0), then padding indices 1..49
            base = 6 + 114 * 500
            infos = [[int(out[base + 7 * k + f][0]) for f in range(7)] for k in range(50)]
            assert infos == [[k, 0, 0, 0, 0, 0, 0] for k in range(50)]
            # the margin / portfolio-margin tier lists were never sent: nil boundaries read as 0, a zero collateral still finds tier 0
        else:
            assert n == -1 and "more assets than the batch's tier" in err
