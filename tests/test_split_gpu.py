"""-m gpu: one proof split over several GPUs (BASELINE.json configs[4]) — the shard entry points of the C ABI
(zkpor_pk_keep_range, zkpor_prove_sums_dev, zkpor_prove_assemble) on one device: the `world` ranks are run one after the other,
their 576-byte partial sums added on the host, and the assembled proof must equal the unsplit proof bit for bit (and the
oracle's, and verify under the pairing).  The RCCL exchange itself runs in test_split_nccl_single_rank (world = 1 on this
box) and, over gloo with two ranks, in test_split_gloo_cpu.py."""
import ctypes
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import oracle as O
import split
import zkpor

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(zk, S):
    pk = zkpor.ProvingKey(zk)
    inf_a = np.array([not S.A[i].any() for i in range(S.n_wires)], dtype=np.uint8)
    inf_b = np.array([not S.B1[i].any() for i in range(S.n_wires)], dtype=np.uint8)
    pk.set_g1(zkpor.G1_A, S.A[inf_a == 0]); pk.set_g1(zkpor.G1_B, S.B1[inf_b == 0]); pk.set_g2(zkpor.G2_B, S.B2[inf_b == 0])
    pk.set_g1(zkpor.G1_K, S.K[S.n_public:]); pk.set_g1(zkpor.G1_Z, S.Z)
    pk.set_consts(S.abd1[0], S.abd1[1], S.abd1[2], S.bd2[0], S.bd2[1], S.log2d, inf_a, inf_b, S.n_wires, S.n_public)
    return pk


def _affine5(sums):
    """the five Jacobian sums as affine coordinates (representation-independent)"""
    u = np.ascontiguousarray(sums, dtype=np.uint8).view(np.uint64)
    g1 = [O.g1_jac_to_affine(u[a:a + 12].copy())[0].tolist() for a in (0, 12, 48, 60)]
    return g1 + [O.g2_jac_to_affine(u[24:48].copy())[0].tolist()]


@pytest.mark.parametrize("world", [1, 2, 4, 8])
def test_split_sums_reassemble_to_the_unsplit_proof(zk, world):
    S = O.Synth(6, 300, n_public=2, seed=31)
    D = 1 << S.log2d
    r = O.fr_random(5, 1)[0]; s = O.fr_random(6, 1)[0]
    whole = _load(zk, S)
    try:
        expect = zk.prove_tail(whole, S.w, S.a, S.b, S.c, r, s)
        consts = whole.consts()
    finally:
        whole.close()
    assert np.array_equal(expect, S.prove_tail(r, s))
    h = O.compute_h(S.a, S.b, S.c, S.log2d)                   # input of step 3 (the device's computeH is covered elsewhere)
    dw = zk.alloc(32 * S.n_wires).upload(S.w)
    dh = zk.alloc(32 * D).upload(h)
    parts = []
    try:
        for rank in range(world):
            pk = _load(zk, S)
            try:
                w_lo, w_hi = split.wire_range(S.n_wires, rank, world)
                z_lo, z_hi = split.z_range(D, rank, world)
                pk.keep_range(w_lo, w_hi, z_lo, z_hi)
                with pytest.raises(zkpor.ZkporError, match="the key is a shard"):
                    zk.prove_tail(pk, S.w, S.a, S.b, S.c, r, s)
                parts.append(zk.prove_sums_dev(pk, dw.ptr + 32 * w_lo, dh.ptr + 32 * z_lo))
                # the two halves a peer computes before / after h arrives add up to the same five sums
                w_part = zk.prove_sums_dev(pk, dw.ptr + 32 * w_lo, None)
                assert not w_part[544:576].any()                       # Z.h slot: the point at infinity (Jacobian Z = 0)
                if z_hi > z_lo:
                    h_part = zk.prove_sums_dev(pk, None, dh.ptr + 32 * z_lo)
                    assert not any(h_part[a:b].any() for a, b in ((64, 96), (160, 192), (320, 384), (448, 480)))
                    assert _affine5(split.merge_sums(w_part, h_part)) == _affine5(parts[-1])
            finally:
                pk.close()
    finally:
        dw.free(); dh.free()
    proof = zkpor.prove_assemble(consts, split.add_partial_sums(np.stack(parts)), r, s)
    assert np.array_equal(proof, expect) and S.verify_pairing(proof)


def test_split_at_2p18_synthetic_key(zk):
    """device-generated key (10 % / 1.6 % infinity points, witness-like scalars): 4 shards == whole"""
    log2 = 18; n = 1 << log2; world = 4
    r = O.fr_random(7, 1)[0]; s = O.fr_random(8, 1)[0]
    bufs = {k: zk.alloc(32 * n) for k in ("w", "a", "b", "c", "h")}
    try:
        zk.fill_fr(bufs["w"], n, 2, 1); zk.fill_fr(bufs["a"], n, 11, 0); zk.fill_fr(bufs["b"], n, 12, 0)
        vp = ctypes.c_void_p
        zk._ck(zk.lib.zkpor_dev_fr_mul(zk.h, vp(bufs["c"].ptr), vp(bufs["a"].ptr), vp(bufs["b"].ptr), ctypes.c_size_t(n)))
        keep = {k: bufs[k].download(np.uint64, (n, 4)) for k in ("a", "b", "c")}
        pk = zkpor.ProvingKey(zk)
        try:
            pk.synth(log2, n, 3, 0, 77)
            consts = pk.consts()
            expect = zk.prove_tail_dev(pk, bufs["w"].ptr, bufs["a"].ptr, bufs["b"].ptr, bufs["c"].ptr, r, s)
        finally:
            pk.close()
        for k in ("a", "b", "c"):
            bufs[k].upload(keep[k])
        zk.compute_h_dev(log2, bufs["a"].ptr, bufs["b"].ptr, bufs["c"].ptr)          # h left in a
        parts = []
        for rank in range(world):
            pk = zkpor.ProvingKey(zk)
            try:
                pk.synth(log2, n, 3, 0, 77)
                w_lo, w_hi = split.wire_range(n, rank, world); z_lo, z_hi = split.z_range(n, rank, world)
                pk.keep_range(w_lo, w_hi, z_lo, z_hi)
                parts.append(zk.prove_sums_dev(pk, bufs["w"].ptr + 32 * w_lo, bufs["a"].ptr + 32 * z_lo))
            finally:
                pk.close()
        proof = zkpor.prove_assemble(consts, split.add_partial_sums(np.stack(parts)), r, s)
        assert np.array_equal(proof, expect)
    finally:
        for b in bufs.values():
            b.free()


@pytest.mark.parametrize("world,from_file", [(2, False), (4, True)])
def test_shards_loaded_straight_from_the_key_file(zk, tmp_path, world, from_file):
    """every rank decompresses only its ranges of the container (zkpor_pk_load_gnark_shard): same sums as trimming a whole key,
    and the reassembled proof is the oracle's"""
    import gnark_keyfile as GK
    S = O.Synth(6, 300, n_public=2, seed=43)
    D = 1 << S.log2d
    r = O.fr_random(5, 1)[0]; s = O.fr_random(6, 1)[0]
    data, inf_a, inf_b = GK.pk_bytes_from_synth(S)
    assert inf_a.any() or inf_b.any()          # the compacted-index translation is exercised
    src = data
    if from_file:
        src = str(tmp_path / "split.pk"); open(src, "wb").write(data)
    h = O.compute_h(S.a, S.b, S.c, S.log2d)
    dw = zk.alloc(32 * S.n_wires).upload(S.w); dh = zk.alloc(32 * D).upload(h)
    parts = []; consts = None
    try:
        for rank in range(world):
            w_lo, w_hi = split.wire_range(S.n_wires, rank, world); z_lo, z_hi = split.z_range(D, rank, world)
            pk = zkpor.ProvingKey(zk); ref = _load(zk, S)
            try:
                L = pk.load_gnark_shard(src, S.n_public, w_lo, w_hi, z_lo, z_hi)
                assert L["n_wires"] == S.n_wires
                consts = pk.consts()
                got = zk.prove_sums_dev(pk, dw.ptr + 32 * w_lo, dh.ptr + 32 * z_lo)
                ref.keep_range(w_lo, w_hi, z_lo, z_hi)
                # the same five POINTS (affine): the Jacobian triple of a sum depends on the order of the additions inside a bucket, and the digit-stream sort
                # of round 6 ranks the entries of a bucket with LDS atomics — the order, and with it the representation, may change from run to run
                assert _affine5(got) == _affine5(zk.prove_sums_dev(ref, dw.ptr + 32 * w_lo, dh.ptr + 32 * z_lo))
                with pytest.raises(zkpor.ZkporError, match="the key is a shard"):
                    zk.prove_tail(pk, S.w, S.a, S.b, S.c, r, s)
                parts.append(got)
            finally:
                pk.close(); ref.close()
    finally:
        dw.free(); dh.free()
    proof = zkpor.prove_assemble(consts, split.add_partial_sums(np.stack(parts)), r, s)
    assert np.array_equal(proof, S.prove_tail(r, s)) and S.verify_pairing(proof)
    pk = zkpor.ProvingKey(zk)
    try:
        with pytest.raises(zkpor.ZkporError, match="shard range outside the key"):
            pk.load_gnark_shard(data, S.n_public, 0, S.n_wires + 1, 0, 1)
    finally:
        pk.close()


def test_an_empty_sum_leaves_no_event_behind(zk):
    """"msm_chain" (round 6): a sum's chain of short launches runs on a second stream and the call's final wait is for the LAST chain's event.  A sum
    without entries (h = 0: Z.h; w = 0: the four witness sums) launches nothing and records nothing — the wait must then be for the last sum that did
    run, not for a stale event: the other sums of such a call equal those of the full call, several times over"""
    S = O.Synth(7, 900, n_public=2, seed=57)
    D = 1 << S.log2d
    pk = _load(zk, S)
    h = O.compute_h(S.a, S.b, S.c, S.log2d)
    dw = zk.alloc(32 * S.n_wires).upload(S.w); dh = zk.alloc(32 * D).upload(h)
    zw = zk.alloc(32 * S.n_wires).upload(np.zeros((S.n_wires, 4), np.uint64)); zh = zk.alloc(32 * D).upload(np.zeros((D, 4), np.uint64))
    inf = lambda sums, lo, hi: not np.asarray(sums)[lo + (hi - lo) * 2 // 3:hi].any()      # Jacobian Z = 0
    try:
        full = _affine5(zk.prove_sums_dev(pk, dw.ptr, dh.ptr))
        for chain in (2, 2, 0, 2):              # 2: the chain stream also for a tail on the context's ordinary streams
            zk.set_param("msm_chain", chain)
            a = zk.prove_sums_dev(pk, dw.ptr, zh.ptr)                  # Z.h empty, it is the LAST sum queued
            assert inf(a, 480, 576) and _affine5(a)[:3] == full[:3] and _affine5(a)[4] == full[4]
            b = zk.prove_sums_dev(pk, zw.ptr, dh.ptr)                  # only Z.h runs
            assert all(inf(b, lo, hi) for lo, hi in ((0, 96), (96, 192), (192, 384), (384, 480))) and _affine5(b)[3] == full[3]
            assert _affine5(zk.prove_sums_dev(pk, dw.ptr, dh.ptr)) == full
    finally:
        zk.set_param("msm_chain", 2)
        for x in (dw, dh, zw, zh):
            x.free()
        pk.close()


def test_keep_range_rejects_bad_ranges(zk):
    S = O.Synth(4, 20, n_public=2, seed=4)
    pk = _load(zk, S)
    try:
        D = 1 << S.log2d
        for args in ((5, 5, 0, 1), (0, S.n_wires + 1, 0, 1), (0, 4, 3, 2), (0, 4, 0, D)):
            with pytest.raises(zkpor.ZkporError, match="shard range outside the key"):
                pk.keep_range(*args)
    finally:
        pk.close()


NCCL_WORKER = r'''
import ctypes, json, os, sys
root = %(root)r
for p in (root, os.path.join(root, "zkmerkle-proof-of-solvency_amd")):
    sys.path.insert(0, p)
import numpy as np, torch, torch.distributed as dist
import zkpor, split
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"]); local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
ctx = zkpor.Context(local, torch.cuda.current_stream().cuda_stream)
log2 = 20; n = 1 << log2
dev = lambda nb: torch.empty(nb, dtype=torch.uint8, device="cuda")
w, a, b, c = dev(32 * n), dev(32 * n), dev(32 * n), dev(32 * n)
vp = ctypes.c_void_p
fill = lambda t, seed, kind: ctx._ck(ctx.lib.zkpor_dev_fill_fr(ctx.h, vp(t.data_ptr()), ctypes.c_size_t(n), ctypes.c_uint64(seed), ctypes.c_int(kind)))
fill(w, 2, 1); fill(a, 11, 0); fill(b, 12, 0)
ctx._ck(ctx.lib.zkpor_dev_fr_mul(ctx.h, vp(c.data_ptr()), vp(a.data_ptr()), vp(b.data_ptr()), ctypes.c_size_t(n)))
r = np.array([3, 1, 4, 1], dtype=np.uint64); s = np.array([2, 7, 1, 8], dtype=np.uint64)
a1, b1, c1 = a.clone(), b.clone(), c.clone()
whole = zkpor.ProvingKey(ctx); whole.synth(log2, n, 3, 0, 5)
expect = ctx.prove_tail_dev(whole, w.data_ptr(), a1.data_ptr(), b1.data_ptr(), c1.data_ptr(), r, s)
whole.close()
pk = zkpor.ProvingKey(ctx); pk.synth(log2, n, 3, 0, 5)
sp = split.SplitProver(ctx, pk, rank, world, dist)
h_mine = dev(sp.h_block_bytes())
a2, b2, c2 = a.clone(), b.clone(), c.clone()
ok = True
for it in range(3):                       # back to back: computeH is still in flight when the exchange is entered
    if rank == 0:
        a.copy_(a2); b.copy_(b2); c.copy_(c2); torch.cuda.synchronize()
        ctx.compute_h_dev(log2, a.data_ptr(), b.data_ptr(), c.data_ptr())
    proof = sp.prove(w.data_ptr(), a if rank == 0 else None, h_mine, r, s)
    ok = ok and bool(np.array_equal(proof, expect))
if world >= 2 and (world & (world - 1)) == 0:      # computeH sharded too: every rank starts from its D_low slices of a, b, c
    sl = lambda t: t.view(n, 32)[rank::world].contiguous().view(-1)
    for it in range(2):
        la, lb, lc = sl(a2), sl(b2), sl(c2)
        tmp = torch.empty_like(la)
        proof = sp.prove_sharded_h(w.data_ptr(), la, lb, lc, tmp, r, s)
        ok = ok and bool(np.array_equal(proof, expect))
print(json.dumps({"rank": rank, "ok": ok}), flush=True)
pk.close(); ctx.close(); dist.destroy_process_group()
'''


def test_split_nccl_single_rank(tmp_path):
    """the exchange of split.py over RCCL (scatter + all-gather) with the one rank this box has; proof == unsplit proof"""
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]      # RCCL prints its library path on stdout at exit
    assert lines and json.loads(lines[-1])["ok"]


def _emulated_all_to_all(zk, bufs, tmp, n_local, wlog, to_high):
    """what dist.all_to_all_single + zkpor_shard_transpose_dev do between the ranks, with device-to-device copies on one GPU"""
    W = 1 << wlog
    chunk = 32 << (n_local - wlog)
    cp = lambda dst, src: zk._ck(zk.lib.zkpor_dev_copy(zk.h, ctypes.c_void_p(dst), ctypes.c_void_p(src), ctypes.c_size_t(chunk)))
    if not to_high:                                       # sender-side de-interleave, then the exchange
        for r in range(W):
            zk.shard_transpose_dev(tmp[r].ptr, bufs[r].ptr, n_local, wlog, False)
        for s in range(W):
            for d in range(W):
                cp(bufs[d].ptr + s * chunk, tmp[s].ptr + d * chunk)
    else:                                                 # the exchange, then the receiver-side interleave
        for s in range(W):
            for d in range(W):
                cp(tmp[d].ptr + s * chunk, bufs[s].ptr + d * chunk)
        for r in range(W):
            zk.shard_transpose_dev(bufs[r].ptr, tmp[r].ptr, n_local, wlog, True)


@pytest.mark.parametrize("six", [1, 0])
@pytest.mark.parametrize("log2,wlog", [(20, 2), (20, 3), (17, 1), (22, 3)])
def test_compute_h_sharded_equals_unsharded(zk, log2, wlog, six):
    """computeH spread over 2^wlog ranks (run one after the other here, the all-to-alls emulated by copies): the blocks of h that
    the ranks end with, concatenated, are bit for bit the h of the unsharded computeH.  six = 1 ("ntt_h" 1, the default): c stops at its
    coefficients in step 1, is not exchanged again and is subtracted by step 3; 0: gnark's seven transforms, c exchanged like a and b"""
    n = 1 << log2; W = 1 << wlog; nl = log2 - wlog
    zk.set_param("ntt_h", six)
    full = {k: zk.alloc(32 * n) for k in "abc"}
    loc = {k: [zk.alloc(32 << nl) for _ in range(W)] for k in "abc"}
    tmp = [zk.alloc(32 << nl) for _ in range(W)]
    try:
        zk.fill_fr(full["a"], n, 11, 0); zk.fill_fr(full["b"], n, 12, 0)
        zk._ck(zk.lib.zkpor_dev_fr_mul(zk.h, ctypes.c_void_p(full["c"].ptr), ctypes.c_void_p(full["a"].ptr), ctypes.c_void_p(full["b"].ptr), ctypes.c_size_t(n)))
        host = {k: full[k].download(np.uint64, (n, 4)) for k in "abc"}
        host["c"][5] = host["a"][7]                         # a*b - c not identically zero on the domain
        full["c"].upload(host["c"])
        for k in "abc":                                     # D_low: rank r holds the elements at positions p = r mod W
            for r in range(W):
                loc[k][r].upload(np.ascontiguousarray(host[k][r::W]))
        zk.compute_h_dev(log2, full["a"].ptr, full["b"].ptr, full["c"].ptr)
        expect = full["a"].download(np.uint64, (n, 4))
        assert expect.any()
        ptrs = lambda r: (loc["a"][r].ptr, loc["b"][r].ptr, loc["c"][r].ptr)
        for r in range(W):
            zk.compute_h_shard_dev(log2, wlog, r, *ptrs(r), 0)
        for k in "abc":
            _emulated_all_to_all(zk, loc[k], tmp, nl, wlog, True)
        for r in range(W):
            zk.compute_h_shard_dev(log2, wlog, r, *ptrs(r), 1)
        for k in ("ab" if six else "abc"):
            _emulated_all_to_all(zk, loc[k], tmp, nl, wlog, False)
        for r in range(W):
            zk.compute_h_shard_dev(log2, wlog, r, loc["a"][r].ptr, loc["b"][r].ptr, None if six else loc["c"][r].ptr, 2)
        _emulated_all_to_all(zk, loc["a"], tmp, nl, wlog, True)
        if six:
            with pytest.raises(zkpor.ZkporError):
                zk.compute_h_shard_dev(log2, wlog, 0, loc["a"][0].ptr, None, None, 3)      # step 3 subtracts c: it has to be there
        for r in range(W):
            zk.compute_h_shard_dev(log2, wlog, r, loc["a"][r].ptr, None, loc["c"][r].ptr if six else None, 3)
        got = np.concatenate([loc["a"][r].download(np.uint64, (1 << nl, 4)) for r in range(W)])
        assert np.array_equal(got, expect)
    finally:
        zk.set_param("ntt_h", 1)
        for b in list(full.values()) + tmp + [x for v in loc.values() for x in v]:
            b.free()


@pytest.mark.isolated
def test_the_ranks_one_after_the_other_on_one_gpu():
    """tools/split_one_gpu.py at 2^20: the run that was done once at 2^28 on one MI355X (profiles/r06_split_2p28_eight_shards_one_gpu.json; BASELINE.json
    configs[4]) — computeH sharded 8 ways with the all-to-alls as device copies, the key's eight ranges summed one after the other, the partial sums
    added on the host: the sharded h is the unsharded h bit for bit and satisfies the quotient identity, the proof is what the key's discrete logs predict"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import split_one_gpu
    res = split_one_gpu.run(20, 3, check_h="yes", log=lambda s: None)
    assert res["sharded_h_equals_unsharded_h"] is True and res["h_satisfies_the_quotient_identity"] is True
    assert res["proof_equals_the_trapdoor_prediction"] is True and res["another_blinding_is_rejected"] is True
    assert len(res["sums_ms_per_rank"]) == 8


def test_split_nccl_all_visible_gpus(tmp_path):
    """the same worker with one rank per visible GPU (2, 4 or 8): scatter / all-to-all / all-gather over RCCL between real devices,
    rank-0 computeH and sharded computeH; opt-in (ZKPOR_TEST_MULTI_GPU=1) and skipped on a one-GPU box"""
    import torch
    world = torch.cuda.device_count()
    world = 8 if world >= 8 else 4 if world >= 4 else 2 if world >= 2 else 1
    if world < 2 or os.environ.get("ZKPOR_TEST_MULTI_GPU") != "1":
        pytest.skip("needs at least two GPUs and ZKPOR_TEST_MULTI_GPU=1 (not yet run on a multi-GPU node)")
    script = tmp_path / "nccl_worker.py"
    script.write_text(NCCL_WORKER % {"root": ROOT})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29633", WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(world)]
    for p in procs:
        o, err = p.communicate(timeout=600)
        assert p.returncode == 0, err[-2000:]
        lines = [l for l in o.splitlines() if l.startswith("{")]
        assert lines and json.loads(lines[-1])["ok"]
