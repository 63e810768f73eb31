"""One Groth16 proof split over several MI355X (SURVEY.md §8e "optional single-proof split"; BASELINE.json configs[4]: a
2^28-constraint proof whose key + workspace do not fit one GPU's 288 GB next to its vectors).

Every rank keeps one contiguous range of each key array (zkpor_pk_keep_range) and the matching range of the wire vector.
Per proof:
  1. rank 0 runs computeH on the full a, b, c                               (the NTTs do not shard without an all-to-all)
  2. h is scattered in `world` equal blocks                                  RCCL scatter: (world-1)/world of 32*D bytes leave
                                                                             rank 0, one block per peer, one xGMI link each
  3. every rank: A.w, B1.w, B2.w, K.w over one sorted digit stream of its w range, Z.h over its h block  (zkpor_prove_sums_dev);
     the peers run the four w-sums BEFORE step 2, under rank 0's computeH, and only Z.h after it
  4. the 576-byte partial sums are all-gathered                              RCCL all-gather, latency-bound (4.6 KB at 8 ranks)
  5. every rank adds the partials on the host (zkpor_g1/g2_jac_sum: RCCL has no reduction over curve points) and assembles the
     proof (zkpor_prove_assemble) — all ranks end with the same 256 bytes.
With a power-of-two number of ranks computeH shards as well (SplitProver.prove_sharded_h / compute_h_sharded): every field pass of
the transform is local under one of two distributions of the index bits, so steps 1-2 become six all-to-alls of 1/world of a
vector per rank (seven with gnark's seven-transform schedule, "ntt_h" 0) and nobody waits for rank 0.
gnark has no counterpart: its MultiExp splits over CPU tasks inside one process.
torch.distributed is plumbing here (backend "nccl" = RCCL on the GPUs, "gloo" in the CPU test of the exchange logic)."""
import numpy as np

import zkpor

SUM_BYTES = 576
_G1_SLOTS = ((0, 96), (96, 192), (384, 480), (480, 576))   # A.w, B1.w, K.w, Z.h
_G2_SLOT = (192, 384)                                      # B2.w


def wire_range(n_wires, rank, world):
    return n_wires * rank // world, n_wires * (rank + 1) // world


def z_block(D, world):
    if D % world:
        raise ValueError("the domain size must be a multiple of the number of ranks")
    return D // world


def z_range(D, rank, world):
    """h is scattered in equal blocks of D/world scalars; Z has D-1 points, so the last block's final scalar has no point"""
    blk = z_block(D, world)
    return rank * blk, min((rank + 1) * blk, D - 1)


def add_partial_sums(parts):
    """parts: (world, 576) uint8 -> (576,) uint8, component-wise group addition on the host"""
    parts = np.ascontiguousarray(parts, dtype=np.uint8).reshape(-1, SUM_BYTES)
    out = np.empty(SUM_BYTES, dtype=np.uint8)
    for lo, hi in _G1_SLOTS:
        out[lo:hi] = zkpor.g1_jac_sum(np.ascontiguousarray(parts[:, lo:hi]).view(np.uint64)).view(np.uint8)
    lo, hi = _G2_SLOT
    out[lo:hi] = zkpor.g2_jac_sum(np.ascontiguousarray(parts[:, lo:hi]).view(np.uint64)).view(np.uint8)
    return out


def merge_sums(w_part, h_part):
    """w_part: 576 bytes with the four w-sums (Z.h slot = infinity); h_part: 576 bytes with only Z.h -> all five"""
    out = np.array(w_part, dtype=np.uint8, copy=True).reshape(SUM_BYTES)
    out[480:576] = np.asarray(h_part, dtype=np.uint8).reshape(SUM_BYTES)[480:576]
    return out


def exchange_and_assemble(dist, rank, world, h_full, h_mine, sums_fn, consts, r, s, device_sync=None, early_fn=None):
    """steps 2-5.  h_full: 1-D uint8 tensor of 32*D bytes on rank 0 (None elsewhere); h_mine: this rank's receive buffer of
    32*D/world bytes; sums_fn(h_mine, early) -> 576 uint8 (step 3: zkpor_prove_sums_dev on the GPUs).  early_fn() -> the four
    w-sums: the peers run it BEFORE the scatter, i.e. while rank 0 is still inside computeH (they do not need h for A, B1, B2, K);
    its result is handed to sums_fn, which then only adds Z.h.  Returns the proof."""
    import torch
    blk = h_mine.numel()
    early = early_fn() if (early_fn is not None and rank != 0) else None
    if device_sync is not None:
        device_sync()                  # computeH ran on the library's stream, which the collective's stream does not follow
    if dist is None:
        h_mine.copy_(h_full[:blk])
    else:
        chunks = [h_full[i * blk:(i + 1) * blk] for i in range(world)] if rank == 0 else None
        dist.scatter(h_mine, chunks, src=0)
    if device_sync is not None:
        device_sync()                  # ... and the library's streams do not follow the collective's stream either
    mine = np.ascontiguousarray(sums_fn(h_mine, early), dtype=np.uint8).reshape(SUM_BYTES)
    if dist is None:
        parts = mine[None, :]
    else:
        t = torch.from_numpy(mine.copy()).to(h_mine.device)
        allp = torch.empty(world * SUM_BYTES, dtype=torch.uint8, device=h_mine.device)
        dist.all_gather_into_tensor(allp, t)
        parts = allp.cpu().numpy().reshape(world, SUM_BYTES)
    return zkpor.prove_assemble(consts, add_partial_sums(parts), r, s)


def compute_h_sharded(dist, world, step_fn, transpose_fn, a, b, c, tmp, device_sync=None, six=True):
    """computeH over all ranks (zkpor_compute_h_shard_dev, csrc/ntt.hip): a, b, c are this rank's D_low slices (elements at
    positions p = rank mod world) as 1-D uint8 tensors, tmp a scratch tensor of the same size; on return `a` holds this rank's
    contiguous block of h.  step_fn(k) runs step k on (a, b, c); transpose_fn(out, inp, interleave) is zkpor_shard_transpose_dev.
    Six all-to-alls of 1/world of a vector per rank — all seven xGMI links of a GPU busy at once — replace the NTT on one GPU: with the library's
    default schedule ("ntt_h" 1, six = True) c stops at its coefficients in step 1 and is NOT exchanged again — step 3 subtracts it where it lies;
    six = False is gnark's seven-transform schedule ("ntt_h" 0: c goes to the coset as well, seven all-to-alls).  The caller keeps the two in step."""
    sync = device_sync if device_sync is not None else (lambda: None)

    def to_high(x):                      # D_low -> D_high: chunk d of the local array goes to rank d, the receiver interleaves
        sync()
        dist.all_to_all_single(tmp, x)
        sync()
        transpose_fn(x, tmp, True)

    def to_low(x):                       # D_high -> D_low: de-interleave, chunk d goes to rank d
        transpose_fn(tmp, x, False)
        sync()
        dist.all_to_all_single(x, tmp)
        sync()

    step_fn(0)
    for x in (a, b, c):
        to_high(x)
    step_fn(1)
    for x in ((a, b) if six else (a, b, c)):
        to_low(x)
    step_fn(2)
    to_high(a)
    step_fn(3)
    sync()


class SplitProver:
    """rank-local half of the split: owns the shard of the key on this GPU.  `pk` must be fully loaded (every rank loads or
    synthesises the same key); it is cut down to this rank's ranges here."""

    def __init__(self, ctx, pk, rank, world, dist, six_transforms=True):
        self.ctx, self.pk, self.rank, self.world, self.dist = ctx, pk, rank, world, dist
        self.six = bool(six_transforms)      # the sharded computeH's schedule: the library's parameter and the exchange pattern are set together
        _, self.n_wires = pk.g1_dev(zkpor.G1_A)
        _, nz = pk.g1_dev(zkpor.G1_Z)
        self.D = nz + 1
        self.consts = pk.consts()
        self.w_lo, self.w_hi = wire_range(self.n_wires, rank, world)
        self.z_lo, self.z_hi = z_range(self.D, rank, world)
        pk.keep_range(self.w_lo, self.w_hi, self.z_lo, self.z_hi)

    def prove_sharded_h(self, d_w_full, a, b, c, tmp, r, s):
        """everything sharded: a, b, c = this rank's D_low slices of the constraint evaluations (torch uint8 on the device,
        32 * D / world bytes each), overwritten; no scatter — computeH ends with every rank holding its block of h"""
        import torch
        wlog = self.world.bit_length() - 1
        if (1 << wlog) != self.world or wlog < 1:
            raise ValueError("the sharded computeH needs a power-of-two number of ranks >= 2")
        log2 = self.D.bit_length() - 1
        nl = log2 - wlog
        self.ctx.set_param("ntt_h", 1 if self.six else 0)
        compute_h_sharded(self.dist, self.world,
                          lambda k: self.ctx.compute_h_shard_dev(log2, wlog, self.rank, a.data_ptr(), b.data_ptr(), c.data_ptr(), k),
                          lambda out, inp, il: self.ctx.shard_transpose_dev(out.data_ptr(), inp.data_ptr(), nl, wlog, il),
                          a, b, c, tmp, device_sync=torch.cuda.synchronize, six=self.six)
        mine = self.ctx.prove_sums_dev(self.pk, d_w_full + 32 * self.w_lo, a.data_ptr())
        t = torch.from_numpy(np.ascontiguousarray(mine).copy()).to(a.device)
        allp = torch.empty(self.world * SUM_BYTES, dtype=torch.uint8, device=a.device)
        self.dist.all_gather_into_tensor(allp, t)
        parts = allp.cpu().numpy().reshape(self.world, SUM_BYTES)
        return zkpor.prove_assemble(self.consts, add_partial_sums(parts), r, s)

    def h_block_bytes(self):
        return 32 * z_block(self.D, self.world)

    def prove(self, d_w_full, h_full, h_mine, r, s):
        """d_w_full: device pointer of the whole wire vector on this GPU (only this rank's range is read); h_full: rank 0's h
        (torch uint8 on the device, as compute_h_dev left it), None on the other ranks; h_mine: receive buffer"""
        import torch
        d_w = d_w_full + 32 * self.w_lo

        def early():                       # peers, during rank 0's computeH: A.w, B1.w, B2.w, K.w of this shard
            return self.ctx.prove_sums_dev(self.pk, d_w, None)

        def sums(hm, w_part):
            if w_part is None:             # rank 0 (or no early phase): all five over one call
                return self.ctx.prove_sums_dev(self.pk, d_w, hm.data_ptr())
            return merge_sums(w_part, self.ctx.prove_sums_dev(self.pk, None, hm.data_ptr()))

        return exchange_and_assemble(self.dist, self.rank, self.world, h_full, h_mine, sums, self.consts, r, s,
                                     device_sync=torch.cuda.synchronize, early_fn=early if self.z_hi > self.z_lo else None)
