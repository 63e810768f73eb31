"""round-4 probe: the cooperative Poseidon kernels on launches of different sizes (4 096 CEX states and fewer; tier-500 leaves at 2^10 .. 2^15
accounts), checked against the oracle on a slice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
for p in (os.path.join(ROOT, "zkmerkle-proof-of-solvency_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import oracle as O
import zkpor as Z
import cex_cases as CC

ctx = Z.Context(0)
consts = CC.make_assets(500, seed=3); totals = CC.make_totals(4096, 500, seed=4)
want = O.fr_to_be(O.cex_commitments(consts, totals[:2]))
for n_states in (256, 1024, 4096):
    ctx.cex_commitments(consts, totals[:8])
    best = 1e9
    for _ in range(3):
        ctx.phase_reset()
        got = ctx.cex_commitments(consts, totals[:n_states])
        best = min(best, ctx.phase_ms("cex_commitments")[0])
    print(f"cex states {n_states}: {best:.1f} ms  ok {np.array_equal(got[:2], want)}", flush=True)
rng = np.random.default_rng(1)
tier = 500
n_gen = 1 << 15
acc = np.zeros(n_gen, dtype=Z.ACCOUNT_DTYPE)
k = rng.integers(tier // 10, tier + 1, size=n_gen)
off = np.concatenate([[0], np.cumsum(k)[:-1]])
acc["n_assets"] = k; acc["asset_off"] = off
acc["id_be"][:, 24:] = rng.integers(0, 256, size=(n_gen, 8), dtype=np.uint8)
acc["equity"][:, 0] = rng.integers(0, 1 << 40, size=n_gen, dtype=np.uint64)
tot = int(k.sum())
assets = np.zeros(tot, dtype=Z.ASSET_DTYPE)
for name in ("equity", "debt", "loan", "margin", "portfolio_margin"):
    assets[name] = rng.integers(0, 1 << 40, size=tot, dtype=np.uint64)
start = rng.integers(0, 500 - k + 1)
assets["index"] = (np.repeat(start, k) + (np.arange(tot) - np.repeat(off, k))).astype(np.uint32)
wantl = O.fr_to_be(O.account_leaves(acc[:64].copy(), assets, tier))
for lg in (10, 12, 13, 14, 15):
    n = 1 << lg
    for coop in (1, 0):
        ctx.set_param("poseidon_coop", coop)
        ctx.poseidon_leaves(acc[:256], assets, tier)
        best = 1e9
        for _ in range(3):
            ctx.phase_reset()
            got = ctx.poseidon_leaves(acc[:n], assets, tier)
            best = min(best, ctx.phase_ms("poseidon_leaf")[0])
        print(f"tier-500 leaves 2^{lg} {'16 lanes per account' if coop else 'one thread per account'}: {best:.2f} ms  {n / best * 1e3:.0f} accounts/s  ok {np.array_equal(got[:64], wantl)}", flush=True)
