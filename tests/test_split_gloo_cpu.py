"""CPU suite: the exchange of the single-proof split (zkmerkle-proof-of-solvency_amd/split.py: scatter of h, all-gather of the
576-byte partial sums, host-side addition and proof assembly) with two ranks over gloo.  There is no GPU here, so step 3 — each
rank's five partial multi-exponentiations, zkpor_prove_sums_dev on a GPU box — is stood in for by the oracle's MSMs over the same
ranges; everything else is the product's code (split.py, zkpor_g1/g2_jac_sum, zkpor_prove_assemble are host-only)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
root = %(root)r
for p in (root, os.path.join(root, "oracle"), os.path.join(root, "zkmerkle-proof-of-solvency_amd")):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
import oracle as O, zkpor, split
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
S = O.Synth(5, 60, n_public=2, seed=37)
D = 1 << S.log2d
r = O.fr_random(5, 1)[0]; s = O.fr_random(6, 1)[0]
one = O.fp_from_ints([1])[0]
def jac1(aff): return np.concatenate([aff, one]) if aff.any() else np.zeros(12, np.uint64)
def jac2(aff): return np.concatenate([aff, one, np.zeros(4, np.uint64)]) if aff.any() else np.zeros(24, np.uint64)
w_lo, w_hi = split.wire_range(S.n_wires, rank, world)
z_lo, z_hi = split.z_range(D, rank, world)
K = S.K.copy(); K[:S.n_public] = 0                       # the prover's K leaves the public wires out
calls = []
def w_sums():                                            # stand-ins for zkpor_prove_sums_dev on this rank's shard
    w = S.w[w_lo:w_hi]
    return np.concatenate([jac1(O.g1_msm(S.A[w_lo:w_hi], w)), jac1(O.g1_msm(S.B1[w_lo:w_hi], w)), jac2(O.g2_msm(S.B2[w_lo:w_hi], w)),
                           jac1(O.g1_msm(K[w_lo:w_hi], w)), np.zeros(12, np.uint64)]).view(np.uint8)
def early_fn():
    calls.append("early"); return w_sums()
def sums_fn(h_mine, early):
    calls.append("late")
    h = h_mine.numpy().view(np.uint64).reshape(-1, 4)
    z = np.concatenate([np.zeros(60, np.uint64), jac1(O.g1_msm(S.Z[z_lo:z_hi], h[:z_hi - z_lo]))]).view(np.uint8)
    return split.merge_sums(early if early is not None else w_sums(), z)
h_full = torch.from_numpy(O.compute_h(S.a, S.b, S.c, S.log2d).view(np.uint8).reshape(-1).copy()) if rank == 0 else None
h_mine = torch.empty(32 * split.z_block(D, world), dtype=torch.uint8)
consts = (S.abd1[0], S.abd1[1], S.abd1[2], S.bd2[0], S.bd2[1])
proof = split.exchange_and_assemble(dist, rank, world, h_full, h_mine, sums_fn, consts, r, s, early_fn=early_fn)
ok = bool(np.array_equal(proof, S.prove_tail(r, s))) and S.verify_pairing(proof)
ok = ok and calls == (["late"] if rank == 0 else ["early", "late"])     # only the peers run the early phase
print(json.dumps({"rank": rank, "ok": ok}), flush=True)
dist.destroy_process_group()
'''


def test_split_exchange_two_ranks_gloo(tmp_path):
    script = tmp_path / "split_worker.py"
    script.write_text(WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29617", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        o, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-3000:]
        assert json.loads(o.strip().splitlines()[-1])["ok"]


H_WORKER = r'''
import json, os, sys
root = %(root)r
for p in (root, os.path.join(root, "tools"), os.path.join(root, "zkmerkle-proof-of-solvency_amd")):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
import ntt_model as M, split
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
n, kl, km, wlog = 7, 3, 2, 1                       # fields (0,3) (3,2) (5,2); 2 ranks
N = 1 << n; nl = n - wlog
import random
random.seed(3)
full = {k: [random.randrange(M.R) for _ in range(N)] for k in "abc"}
enc = lambda vals: torch.from_numpy(np.frombuffer(b"".join(v.to_bytes(32, "little") for v in vals), dtype=np.uint8).copy())
dec = lambda t: [int.from_bytes(t.numpy().tobytes()[32 * i:32 * i + 32], "little") for i in range(t.numel() // 32)]
ten = {k: enc(full[k][rank::world]) for k in "abc"}           # D_low slices
tmp = torch.empty_like(ten["a"])
Tf, Ti = M.Tables(n, False), M.Tables(n, True)
fields = M.plan_fields(n, kl, km)
ginv = pow(M.G, M.R - 2, M.R); ninv = pow(N, M.R - 2, M.R)
den = pow((pow(M.G, N, M.R) - 1) %% M.R, M.R - 2, M.R)
pre = lambda p: pow(M.G, M.rev(p, n), M.R) * ninv %% M.R       # g^rev(p) / N, as compute_h_dev folds it
post = lambda p: pow(ginv, M.rev(p, n), M.R) * ninv %% M.R
def run(k, T, dif, dist_, flds, first=None, last=None):        # stand-in for one ntt_shard_stage on tensor k
    v = dec(ten[k])
    order = list(reversed(fields)) if dif else list(fields)
    for idx, (lo, kb) in enumerate(order):
        if (lo, kb) not in flds: continue
        sc = first if idx == 0 else (last if idx == len(order) - 1 else None)
        M.pass_local(v, n, wlog, rank, dist_, lo, kb, T, dif, sc)
    ten[k].copy_(enc(v))
upper, lowest = fields[1:], fields[:1]
def step_fn(s):                                                # the four steps of zkpor_compute_h_shard_dev, on the model
    if s == 0:
        for k in "abc": run(k, Ti, True, "low", upper)
    elif s == 1:                                               # "ntt_h" 1: c ends at its coefficients (times den / N), in D_high, and stays there
        for k in "ab": run(k, Ti, True, "high", lowest); run(k, Tf, False, "high", lowest, first=pre)
        run("c", Ti, True, "high", lowest, last=lambda p: den * ninv %% M.R)
    elif s == 2:
        for k in "ab": run(k, Tf, False, "low", upper)
        a, b = dec(ten["a"]), dec(ten["b"])
        ten["a"].copy_(enc([x * y * den %% M.R for x, y in zip(a, b)]))
        run("a", Ti, True, "low", upper)
    else:
        run("a", Ti, True, "high", lowest, last=post)
        ten["a"].copy_(enc([(x - z) %% M.R for x, z in zip(dec(ten["a"]), dec(ten["c"]))]))
def transpose_fn(out, inp, interleave):                        # stand-in for zkpor_shard_transpose_dev
    Mch = (1 << nl) // world
    x = inp.view(world, Mch, 32) if interleave else inp.view(Mch, world, 32)
    out.copy_(x.transpose(0, 1).contiguous().view(-1))
split.compute_h_sharded(dist, world, step_fn, transpose_fn, ten["a"], ten["b"], ten["c"], tmp)
# reference: the unsharded sequence of compute_h_dev on the model
def fwd_inv(x):
    x = M.fft(x, n, True, True, False, kl, km)                                 # inverse DIF (1/N inside)
    return M.fft(x, n, False, False, True, kl, km)                             # forward DIT on the coset
A, B, C = (fwd_inv(full[k]) for k in "abc")
P = [(x * y - z) * den %% M.R for x, y, z in zip(A, B, C)]
H = M.fft(P, n, True, True, True, kl, km)
blk = N // world
ok = dec(ten["a"]) == H[rank * blk:(rank + 1) * blk]
print(json.dumps({"rank": rank, "ok": bool(ok)}), flush=True)
dist.destroy_process_group()
'''


def test_sharded_compute_h_exchange_two_ranks_gloo(tmp_path):
    """split.compute_h_sharded (the order of steps, transposes and all_to_all_single calls) over gloo with two ranks; the device
    steps are stood in for by tools/ntt_model.py's index-exact passes on a 128-point domain"""
    script = tmp_path / "h_worker.py"
    script.write_text(H_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29619", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    for p in procs:
        o, err = p.communicate(timeout=240)
        assert p.returncode == 0, err[-3000:]
        assert json.loads(o.strip().splitlines()[-1])["ok"]
