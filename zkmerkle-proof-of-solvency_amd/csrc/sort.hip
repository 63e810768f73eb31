// MSM steps 1 + 2, hand-written for gfx950: scalars -> signed c-bit digits -> (bucket key, point index | sign) entries GROUPED BY BUCKET.
// Replaces gnark-crypto's partitionScalars + the per-chunk bucket walk of ecc/bn254/multiexp.go (third-party, recalled; reference call site
// src/prover/prover/prover.go:269).  Rounds 1-5 used rocPRIM's onesweep radix sort behind a separate decompose kernel; this file is what the
// consumer actually needs, and nothing more:
//
//   * The level-1 accumulation only needs every bucket's entries CONTIGUOUS (a group law is commutative): no stability, no order inside a
//     bucket.  So the sort is most-significant-digit first, and an entry's rank inside a tile is one LDS atomic (`ds_add_rtn`) instead of a
//     stable match / ballot ranking; a tile's run in each child segment is reserved with ONE global atomic per (tile, child).
//   * Level 0 is FUSED with the digit decomposition: the scalars are read twice (count, scatter: 2 x 32 B per scalar) and the unsorted entry
//     stream — 96 B per scalar written, then read by a histogram pass and the first sort pass — never exists.
//   * Every level is count -> exclusive scan -> scatter over the key prefix `key >> shift`: the children of level l are the parents of level
//     l + 1, tiles never straddle a parent, so a tile touches at most 2^r <= 512 children (LDS counters), and after the scatter the cursor array
//     IS the array of segment ends the next level walks.  23 key bits (3 bucket windows of 2^21 at the production size) = levels of 7 + 8 + 8 bits.
//   * The kernels run BESIDE the VALU-bound NTT passes and bucket accumulations of the main stream ("memory-bound helpers must run
//     on a small persistent grid"): 256-thread workgroups, <= 56 VGPRs and 40 KB of LDS so that one fits next to three resident workgroups of the
//     level-1 kernel, a persistent grid that takes work in chunks from a ticket counter, entries staged through LDS so that a wave's stores cover
//     whole runs.  rocPRIM's 1024-thread onesweep workgroups waited for 16 free wave slots on ONE compute unit (profiles/r03_timeline_before.txt).
//
// Output: keys ascending (the children of every level are laid out in prefix order), values = (point index << 1 | sign) | absence flags.
#include "msm_kernels.cuh"
namespace zk {

namespace {
constexpr u32 DS_THREADS = 256, DS_MAXR = 9, DS_NB = 1u << DS_MAXR;
constexpr u32 DS_CHUNK = 1u << 16;                 // entries per ticket of a level >= 1 kernel
constexpr u32 DS_SCALARS_PER_TICKET = 2048;        // scalars per ticket of the level-0 kernels
// TILE = the entries a workgroup stages in LDS at a time ("sort_tile": 4096 / 2048 / 1024 -> 40 / 24 / 16 KB of LDS per workgroup).  The footprint
// decides what the sort displaces on a compute unit it shares with the main stream's kernels (four 36 KB NTT tiles, three 35 KB level-1 workgroups);
// a smaller tile writes shorter runs.  A level-0 tile = 256 scalars x TILE / 256 digits.

// low c bits of s, then s >>= c (static register indexing only: no scratch)
ZK_D u32 take_digit(Fr& s, int c) {
    u32 d = s.v[0] & ((1u << c) - 1u);
#pragma unroll
    for (int i = 0; i < 7; ++i) s.v[i] = (s.v[i] >> c) | (s.v[i + 1] << (32 - c));
    s.v[7] >>= c;
    return d;
}

struct DsCfg { int c, W, tables, piece; u32 bpw; };

template <u32 TILE>
struct DsShared {
    u32 sk[TILE], sv[TILE];
    u32 cnt[DS_NB], cur[DS_NB], lpre[DS_NB], gbase[DS_NB];
    u32 wsum[4], misc[8];
};

// after a count phase: reserve this tile's run in every child (one global atomic each), the tile-local exclusive prefix of the counts, and the
// placing cursors.  nb <= 512 counters, two per thread.  Leaves cnt[] zeroed for the next tile.
template <class SH>
ZK_D void ds_reserve(SH& S, u32 nb, u32* __restrict__ cursor, u32 child0, u32 n_child) {
    const u32 t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    const u32 i0 = 2u * t, i1 = 2u * t + 1u;
    const u32 a = i0 < nb ? S.cnt[i0] : 0u, b = i1 < nb ? S.cnt[i1] : 0u;
    u32 x = a + b;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const u32 y = __shfl_up(x, off);
        if ((int)lane >= off) x += y;
    }
    if (lane == 63u) S.wsum[wv] = x;
    if (a && child0 + i0 < n_child) S.gbase[i0] = atomicAdd(cursor + child0 + i0, a);
    if (b && child0 + i1 < n_child) S.gbase[i1] = atomicAdd(cursor + child0 + i1, b);
    __syncthreads();
    u32 ex = x - (a + b);
    for (u32 w = 0; w < wv; ++w) ex += S.wsum[w];
    if (i0 < nb) { S.lpre[i0] = ex; S.cur[i0] = ex; S.cnt[i0] = 0u; }
    if (i1 < nb) { S.lpre[i1] = ex + a; S.cur[i1] = ex + a; S.cnt[i1] = 0u; }
    __syncthreads();
}

// the staged tile (sk / sv hold `len` entries grouped by child, child d at [lpre[d], lpre[d] + count)) -> its reserved runs: consecutive lanes write
// consecutive addresses inside a run
template <class SH>
ZK_D void ds_write_out(const SH& S, u32 len, int shift, u32 mask, u32* __restrict__ out_k, u32* __restrict__ out_v) {
    for (u32 j = threadIdx.x; j < len; j += DS_THREADS) {
        const u32 k = S.sk[j];
        const u32 d = (k >> shift) & mask;
        const u32 dest = S.gbase[d] + (j - S.lpre[d]);
        out_k[dest] = k;
        out_v[dest] = S.sv[j];
    }
}

// ------------------------------------------------------------------------------------------------ level 0: fused with the decomposition
// digits w0 .. w0 + nd - 1 of a scalar whose already-shifted remainder is `t` and whose pending carry is `carry` (both advanced); f(key, val) per
// non-zero digit.  Digit w of the scalar = bucket window w % piece against table w / piece of point i (msm.cuh MsmCfg).
template <class Fn>
ZK_D void ds_digits(Fr& t, u32& carry, int w0, int nd, const DsCfg& cfg, u32 i, u32 flags, Fn f) {
    const u32 half = 1u << (cfg.c - 1);
    for (int w = w0; w < w0 + nd; ++w) {
        u32 d = take_digit(t, cfg.c) + carry;
        carry = d > half ? 1u : 0u;
        d = carry ? (1u << cfg.c) - d : d;
        if (d) {
            const u32 q = (u32)w / (u32)cfg.piece;
            f(((u32)w - q * (u32)cfg.piece) * cfg.bpw + (d - 1u), ((i * (u32)cfg.tables + q) << 1) | carry | flags);   // carry == 1 <=> the digit is negative
        }
    }
}

// counts: C0[key >> shift0] over every entry, counter[0] = entries, counter[1 + g] = entries whose point is present in array group g (the sizes of the
// per-array streams, msm_digits.hip k_filter_write).  ticket: one u32, zero on entry.
__global__ __launch_bounds__(DS_THREADS) void k_dsort_count0(const Fr* __restrict__ scalars, u32 n, DsCfg cfg, int shift0, u32 nb0, u32* __restrict__ C0,
                                                             u32* __restrict__ counter, u32* __restrict__ ticket, const u32* __restrict__ absent0,
                                                             const u32* __restrict__ absent1) {
    __shared__ u32 cnt[DS_NB];
    __shared__ u32 s_ticket;
    for (u32 i = threadIdx.x; i < nb0; i += DS_THREADS) cnt[i] = 0u;
    u32 tot = 0, tot0 = 0, tot1 = 0;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        const u32 base = s_ticket * DS_SCALARS_PER_TICKET;
        if (base >= n) break;
        for (u32 i = base + threadIdx.x; i < base + DS_SCALARS_PER_TICKET && i < n; i += DS_THREADS) {
            Fr t = Fr::from_mont(scalars[i]);
            u32 carry = 0, c_here = 0;
            ds_digits(t, carry, 0, cfg.W, cfg, i, 0u, [&](u32 key, u32) { atomicAdd(&cnt[key >> shift0], 1u); ++c_here; });
            tot += c_here;
            if (absent0 || absent1) {
                const u32 word = i >> 5, bit = i & 31u;
                if (!(absent0 && ((absent0[word] >> bit) & 1u))) tot0 += c_here;
                if (!(absent1 && ((absent1[word] >> bit) & 1u))) tot1 += c_here;
            }
        }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < nb0; i += DS_THREADS) if (cnt[i]) atomicAdd(C0 + i, cnt[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { tot += __shfl_down(tot, off); tot0 += __shfl_down(tot0, off); tot1 += __shfl_down(tot1, off); }
    if ((threadIdx.x & 63u) == 0u) {
        if (tot) atomicAdd(counter, tot);
        if (tot0) atomicAdd(counter + 1, tot0);
        if (tot1) atomicAdd(counter + 2, tot1);
    }
}

// scatter: C0 holds the children's start offsets on entry and their end offsets on exit
template <u32 TILE>
__global__ __launch_bounds__(DS_THREADS) void k_dsort_scatter0(const Fr* __restrict__ scalars, u32 n, DsCfg cfg, int shift0, u32 nb0, u32* __restrict__ C0,
                                                               u32* __restrict__ ticket, const u32* __restrict__ absent0, const u32* __restrict__ absent1,
                                                               u32* __restrict__ out_k, u32* __restrict__ out_v) {
    __shared__ DsShared<TILE> S;
    constexpr int DS_DW = (int)(TILE / DS_THREADS);
    for (u32 i = threadIdx.x; i < DS_NB; i += DS_THREADS) S.cnt[i] = 0u;
    const u32 mask = 0xffffffffu;      // level 0: the child IS key >> shift0
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) S.misc[0] = atomicAdd(ticket, 1u);
        __syncthreads();
        const u32 base = S.misc[0] * DS_SCALARS_PER_TICKET;
        if (base >= n) break;
        for (u32 blk = base; blk < base + DS_SCALARS_PER_TICKET && blk < n; blk += DS_THREADS) {
            const u32 i = blk + threadIdx.x;
            const bool live = i < n;
            Fr t = live ? Fr::from_mont(scalars[i]) : Fr::zero();
            u32 carry = 0, flags = 0;
            if (live && (absent0 || absent1)) {   // bits 30 / 31 of every value this scalar emits: its point is absent from group 0 / 1
                const u32 word = i >> 5, bit = i & 31u;
                if (absent0 && ((absent0[word] >> bit) & 1u)) flags |= VAL_ABSENT0;
                if (absent1 && ((absent1[word] >> bit) & 1u)) flags |= VAL_ABSENT1;
            }
            for (int w0 = 0; w0 < cfg.W; w0 += DS_DW) {
                const int nd = cfg.W - w0 < DS_DW ? cfg.W - w0 : DS_DW;
                // pass 1 over this tile's digits: count per child (on a copy of the running state)
                u32 mine = 0;
                if (live) {
                    Fr t1 = t; u32 c1 = carry;
                    ds_digits(t1, c1, w0, nd, cfg, i, flags, [&](u32 key, u32) { atomicAdd(&S.cnt[key >> shift0], 1u); ++mine; });
                }
                __syncthreads();
                // the tile's length: sum of `mine`
                u32 x = mine;
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) x += __shfl_down(x, off);
                if ((threadIdx.x & 63u) == 0u) S.misc[4 + (threadIdx.x >> 6)] = x;
                ds_reserve(S, nb0, C0, 0u, nb0);
                const u32 len = S.misc[4] + S.misc[5] + S.misc[6] + S.misc[7];
                // pass 2: the same digits again, placed
                if (live) ds_digits(t, carry, w0, nd, cfg, i, flags, [&](u32 key, u32 val) { const u32 p = atomicAdd(&S.cur[key >> shift0], 1u); S.sk[p] = key; S.sv[p] = val; });
                __syncthreads();
                ds_write_out(S, len, shift0, mask, out_k, out_v);
                __syncthreads();
            }
        }
    }
}

// ---- level 0 with the window, the digit count and the table count known at compile time (the production shapes) ----
// The generic kernels above shift the whole 256-bit scalar once per digit and recompute the digits in every pass (about 1 700 VALU instructions per
// scalar, zero or not); the main stream's kernels are VALU-bound, so every instruction here is paid for there (profiles/r06_sort_grid_tile_sweep.json).
// With C, W, M constant a digit is two static-index shifts, the W keys stay in registers between the passes, and a scalar whose words are all zero
// (half of a solved wire vector) is skipped before its Montgomery reduction.
template <int C, int W, int M>
ZK_D u32 ds_digits_static(const Fr& s, u32 (&key)[W], u32& neg) {
    constexpr int PIECE = (W + M - 1) / M;
    constexpr u32 HALF = 1u << (C - 1), MASK = (1u << C) - 1u;
    u32 carry = 0, nz = 0;
    neg = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) {
        const int bit = w * C, limb = bit >> 5, off = bit & 31;
        u32 d = limb < 8 ? s.v[limb < 8 ? limb : 7] >> off : 0u;
        if (off + C > 32 && limb + 1 < 8) d |= s.v[limb + 1 < 8 ? limb + 1 : 7] << (32 - off);
        d = (d & MASK) + carry;
        carry = d > HALF ? 1u : 0u;
        d = carry ? (1u << C) - d : d;
        const int q = w / PIECE;
        key[w] = (u32)(w - q * PIECE) * HALF + (d - 1u);
        nz |= (d ? 1u : 0u) << w;
        neg |= carry << w;
    }
    return nz;
}
ZK_D bool fr_words_zero(const Fr& s) { return (s.v[0] | s.v[1] | s.v[2] | s.v[3] | s.v[4] | s.v[5] | s.v[6] | s.v[7]) == 0u; }

template <int C, int W, int M>
__global__ __launch_bounds__(DS_THREADS) void k_dsort_count0_s(const Fr* __restrict__ scalars, u32 n, int shift0, u32 nb0, u32* __restrict__ C0,
                                                               u32* __restrict__ counter, u32* __restrict__ ticket, const u32* __restrict__ absent0,
                                                               const u32* __restrict__ absent1) {
    __shared__ u32 cnt[DS_NB];
    __shared__ u32 s_ticket;
    for (u32 i = threadIdx.x; i < nb0; i += DS_THREADS) cnt[i] = 0u;
    u32 tot = 0, tot0 = 0, tot1 = 0;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) s_ticket = atomicAdd(ticket, 1u);
        __syncthreads();
        const u32 base = s_ticket * DS_SCALARS_PER_TICKET;
        if (base >= n) break;
        for (u32 i = base + threadIdx.x; i < base + DS_SCALARS_PER_TICKET && i < n; i += DS_THREADS) {
            const Fr raw = scalars[i];
            if (fr_words_zero(raw)) continue;
            const Fr t = Fr::from_mont(raw);
            u32 key[W], neg;
            const u32 nz = ds_digits_static<C, W, M>(t, key, neg);
#pragma unroll
            for (int w = 0; w < W; ++w) if ((nz >> w) & 1u) atomicAdd(&cnt[key[w] >> shift0], 1u);
            const u32 c_here = (u32)__popc(nz);
            tot += c_here;
            if (absent0 || absent1) {
                const u32 word = i >> 5, bit = i & 31u;
                if (!(absent0 && ((absent0[word] >> bit) & 1u))) tot0 += c_here;
                if (!(absent1 && ((absent1[word] >> bit) & 1u))) tot1 += c_here;
            }
        }
    }
    __syncthreads();
    for (u32 i = threadIdx.x; i < nb0; i += DS_THREADS) if (cnt[i]) atomicAdd(C0 + i, cnt[i]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { tot += __shfl_down(tot, off); tot0 += __shfl_down(tot0, off); tot1 += __shfl_down(tot1, off); }
    if ((threadIdx.x & 63u) == 0u) {
        if (tot) atomicAdd(counter, tot);
        if (tot0) atomicAdd(counter + 1, tot0);
        if (tot1) atomicAdd(counter + 2, tot1);
    }
}

template <int C, int W, int M>
__global__ __launch_bounds__(DS_THREADS) void k_dsort_scatter0_s(const Fr* __restrict__ scalars, u32 n, int shift0, u32 nb0, u32* __restrict__ C0,
                                                                 u32* __restrict__ ticket, const u32* __restrict__ absent0, const u32* __restrict__ absent1,
                                                                 u32* __restrict__ out_k, u32* __restrict__ out_v) {
    static_assert(W * DS_THREADS <= 4096, "one tile holds a block's digits");
    __shared__ DsShared<4096> S;
    for (u32 i = threadIdx.x; i < DS_NB; i += DS_THREADS) S.cnt[i] = 0u;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) S.misc[0] = atomicAdd(ticket, 1u);
        __syncthreads();
        const u32 base = S.misc[0] * DS_SCALARS_PER_TICKET;
        if (base >= n) break;
        for (u32 blk = base; blk < base + DS_SCALARS_PER_TICKET && blk < n; blk += DS_THREADS) {
            const u32 i = blk + threadIdx.x;
            u32 key[W], neg = 0, nz = 0;
            if (i < n) {
                const Fr raw = scalars[i];
                if (!fr_words_zero(raw)) nz = ds_digits_static<C, W, M>(Fr::from_mont(raw), key, neg);
            }
#pragma unroll
            for (int w = 0; w < W; ++w) if ((nz >> w) & 1u) atomicAdd(&S.cnt[key[w] >> shift0], 1u);
            u32 x = (u32)__popc(nz);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) x += __shfl_down(x, off);
            if ((threadIdx.x & 63u) == 0u) S.misc[4 + (threadIdx.x >> 6)] = x;
            __syncthreads();
            const u32 len = S.misc[4] + S.misc[5] + S.misc[6] + S.misc[7];
            if (len == 0u) { __syncthreads(); continue; }            // a block of zero scalars (uniform: every thread read the same sums)
            ds_reserve(S, nb0, C0, 0u, nb0);
            if (nz) {
                u32 flags = 0;
                if (absent0 || absent1) {   // bits 30 / 31 of every value this scalar emits: its point is absent from group 0 / 1
                    const u32 word = i >> 5, bit = i & 31u;
                    if (absent0 && ((absent0[word] >> bit) & 1u)) flags |= VAL_ABSENT0;
                    if (absent1 && ((absent1[word] >> bit) & 1u)) flags |= VAL_ABSENT1;
                }
                constexpr int PIECE = (W + M - 1) / M;
#pragma unroll
                for (int w = 0; w < W; ++w)
                    if ((nz >> w) & 1u) {
                        const u32 p = atomicAdd(&S.cur[key[w] >> shift0], 1u);
                        S.sk[p] = key[w];
                        S.sv[p] = ((i * (u32)M + (u32)(w / PIECE)) << 1) | ((neg >> w) & 1u) | flags;
                    }
            }
            __syncthreads();
            ds_write_out(S, len, shift0, 0xffffffffu, out_k, out_v);
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------ levels >= 1
// first index s in [0, n_par) with ends[s] > pos (ends ascending, ends[n_par - 1] > pos): a 256-ary search, one probe per thread and round
template <class SH>
ZK_D u32 ds_find_seg(SH& S, const u32* __restrict__ ends, u32 n_par, u32 pos) {
    u32 lo = 0, hi = n_par;
    const u32 t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    for (;;) {
        const u32 span = hi - lo;
        const u32 stride = (span + DS_THREADS - 1u) / DS_THREADS;
        u32 idx = lo + (t + 1u) * stride - 1u;
        const bool inside = lo + t * stride < hi;
        if (idx >= hi) idx = hi - 1u;
        const bool flag = inside && ends[idx] > pos;
        const unsigned long long b = __ballot(flag);
        __syncthreads();                                   // wsum of the previous round / caller has been read
        if (lane == 0u) S.wsum[wv] = b ? wv * 64u + (u32)__ffsll((long long)b) - 1u : 0xffffffffu;
        __syncthreads();
        u32 tmin = S.wsum[0];
        for (u32 w = 1; w < 4; ++w) tmin = S.wsum[w] < tmin ? S.wsum[w] : tmin;
        if (tmin == 0xffffffffu) return n_par - 1u;        // cannot happen (pos < ends[n_par - 1]); keeps the walk inside the array
        const u32 nlo = lo + tmin * stride;
        if (stride == 1u) return nlo;
        hi = nlo + stride < hi ? nlo + stride : hi;
        lo = nlo;
    }
}

// counts of one level: C[key >> shift] over all M entries.  ends = the previous level's cursor array (= its children's end offsets), n_par of them.
__global__ __launch_bounds__(DS_THREADS) void k_dsort_count(const u32* __restrict__ keys, u32 M, int shift, int r, const u32* __restrict__ ends, u32 n_par,
                                                            u32* __restrict__ C, u32 n_child, u32* __restrict__ ticket) {
    __shared__ DsShared<64> S;      // counters only: nothing is staged
    const u32 nb = 1u << r, mask = nb - 1u;
    for (u32 i = threadIdx.x; i < DS_NB; i += DS_THREADS) S.cnt[i] = 0u;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) S.misc[0] = atomicAdd(ticket, 1u);
        __syncthreads();
        u32 pos = S.misc[0] * DS_CHUNK;
        if (pos >= M) break;
        const u32 cend = (M - pos > DS_CHUNK) ? pos + DS_CHUNK : M;
        while (pos < cend) {
            const u32 seg = ds_find_seg(S, ends, n_par, pos);
            const u32 seg_end = ends[seg];
            const u32 stop = seg_end < cend ? seg_end : cend;
#pragma unroll 4
            for (u32 j = pos + threadIdx.x; j < stop; j += DS_THREADS) atomicAdd(&S.cnt[(keys[j] >> shift) & mask], 1u);
            __syncthreads();
            for (u32 d = threadIdx.x; d < nb; d += DS_THREADS) {
                const u32 c = S.cnt[d];
                if (c) { S.cnt[d] = 0u; if ((seg << r) + d < n_child) atomicAdd(C + (seg << r) + d, c); }
            }
            pos = stop;
        }
    }
}

template <u32 TILE>
__global__ __launch_bounds__(DS_THREADS) void k_dsort_scatter(const u32* __restrict__ keys, const u32* __restrict__ vals, u32 M, int shift, int r,
                                                              const u32* __restrict__ ends, u32 n_par, u32* __restrict__ C, u32 n_child,
                                                              u32* __restrict__ ticket, u32* __restrict__ out_k, u32* __restrict__ out_v) {
    __shared__ DsShared<TILE> S;
    const u32 nb = 1u << r, mask = nb - 1u;
    for (u32 i = threadIdx.x; i < DS_NB; i += DS_THREADS) S.cnt[i] = 0u;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) S.misc[0] = atomicAdd(ticket, 1u);
        __syncthreads();
        u32 pos = S.misc[0] * DS_CHUNK;
        if (pos >= M) break;
        const u32 cend = (M - pos > DS_CHUNK) ? pos + DS_CHUNK : M;
        u32 seg = 0, seg_end = 0;
        while (pos < cend) {
            if (pos >= seg_end) { seg = ds_find_seg(S, ends, n_par, pos); seg_end = ends[seg]; }
            u32 len = seg_end - pos;
            if (len > cend - pos) len = cend - pos;
            if (len > TILE) len = TILE;
#pragma unroll 4
            for (u32 j = threadIdx.x; j < len; j += DS_THREADS) atomicAdd(&S.cnt[(keys[pos + j] >> shift) & mask], 1u);
            __syncthreads();
            ds_reserve(S, nb, C, seg << r, n_child);
#pragma unroll 4
            for (u32 j = threadIdx.x; j < len; j += DS_THREADS) {
                const u32 k = keys[pos + j];
                const u32 p = atomicAdd(&S.cur[(k >> shift) & mask], 1u);
                S.sk[p] = k;
                S.sv[p] = vals[pos + j];
            }
            __syncthreads();
            ds_write_out(S, len, shift, mask, out_k, out_v);
            __syncthreads();
            pos += len;
        }
    }
}

// ------------------------------------------------------------------------------------------------ scatter without LDS staging ("sort_stage" 0)
// What the sort costs the main stream is residence (profiles/r06_sort_v2_and_ntt_twiddles_sweep.json): a workgroup that stages 4096 entries holds
// 40 KB of LDS and displaces one of the four NTT tiles of its compute unit.  This form keeps only the counters in LDS (4 KB): after the count pass a
// child's cursor in LDS is its reserved GLOBAL position, and the placing pass stores every entry straight to memory (two 4-byte stores, neighbouring
// ranks to neighbouring addresses — merged in the L2, not in the wave).  No prefix scan, one barrier less per tile, and since nothing is staged a
// tile may be long (LITE_TILE entries: longer runs per child).
constexpr u32 LITE_TILE = 16384;
struct DsLite { u32 cnt[DS_NB], cur[DS_NB], wsum[4], misc[8]; };

// cnt[] -> reserved global positions in cur[]; cnt[] zeroed; ends with a barrier
ZK_D void ds_reserve_lite(DsLite& S, u32 nb, u32* __restrict__ cursor, u32 child0, u32 n_child) {
    for (u32 d = threadIdx.x; d < nb; d += DS_THREADS) {
        const u32 c = S.cnt[d];
        if (c) { S.cnt[d] = 0u; if (child0 + d < n_child) S.cur[d] = atomicAdd(cursor + child0 + d, c); }
    }
    __syncthreads();
}

__global__ __launch_bounds__(DS_THREADS) void k_dsort_scatter_lite(const u32* __restrict__ keys, const u32* __restrict__ vals, u32 M, int shift, int r,
                                                                   const u32* __restrict__ ends, u32 n_par, u32* __restrict__ C, u32 n_child,
                                                                   u32* __restrict__ ticket, u32* __restrict__ out_k, u32* __restrict__ out_v) {
    __shared__ DsLite S;
    const u32 nb = 1u << r, mask = nb - 1u;
    for (u32 i = threadIdx.x; i < DS_NB; i += DS_THREADS) S.cnt[i] = 0u;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) S.misc[0] = atomicAdd(ticket, 1u);
        __syncthreads();
        u32 pos = S.misc[0] * DS_CHUNK;
        if (pos >= M) break;
        const u32 cend = (M - pos > DS_CHUNK) ? pos + DS_CHUNK : M;
        u32 seg = 0, seg_end = 0;
        while (pos < cend) {
            if (pos >= seg_end) { seg = ds_find_seg(S, ends, n_par, pos); seg_end = ends[seg]; }
            u32 len = seg_end - pos;
            if (len > cend - pos) len = cend - pos;
            if (len > LITE_TILE) len = LITE_TILE;
#pragma unroll 4
            for (u32 j = threadIdx.x; j < len; j += DS_THREADS) atomicAdd(&S.cnt[(keys[pos + j] >> shift) & mask], 1u);
            __syncthreads();
            ds_reserve_lite(S, nb, C, seg << r, n_child);
#pragma unroll 4
            for (u32 j = threadIdx.x; j < len; j += DS_THREADS) {
                const u32 k = keys[pos + j];
                const u32 dest = atomicAdd(&S.cur[(k >> shift) & mask], 1u);
                out_k[dest] = k;
                out_v[dest] = vals[pos + j];
            }
            __syncthreads();
            pos += len;
        }
    }
}

template <int C, int W, int M>
__global__ __launch_bounds__(DS_THREADS) void k_dsort_scatter0_lite(const Fr* __restrict__ scalars, u32 n, int shift0, u32 nb0, u32* __restrict__ C0,
                                                                    u32* __restrict__ ticket, const u32* __restrict__ absent0, const u32* __restrict__ absent1,
                                                                    u32* __restrict__ out_k, u32* __restrict__ out_v) {
    __shared__ DsLite S;
    for (u32 i = threadIdx.x; i < DS_NB; i += DS_THREADS) S.cnt[i] = 0u;
    for (;;) {
        __syncthreads();
        if (threadIdx.x == 0) S.misc[0] = atomicAdd(ticket, 1u);
        __syncthreads();
        const u32 base = S.misc[0] * DS_SCALARS_PER_TICKET;
        if (base >= n) break;
        // a tile = TWO blocks of 256 scalars (nothing is staged: the keys of a thread's two scalars wait in registers)
        for (u32 blk = base; blk < base + DS_SCALARS_PER_TICKET && blk < n; blk += 2u * DS_THREADS) {
            u32 key[2][W], neg[2] = {0, 0}, nz[2] = {0, 0};
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32 i = blk + (u32)h * DS_THREADS + threadIdx.x;
                if (i < n && i < base + DS_SCALARS_PER_TICKET) {
                    const Fr raw = scalars[i];
                    if (!fr_words_zero(raw)) nz[h] = ds_digits_static<C, W, M>(Fr::from_mont(raw), key[h], neg[h]);
                }
#pragma unroll
                for (int w = 0; w < W; ++w) if ((nz[h] >> w) & 1u) atomicAdd(&S.cnt[key[h][w] >> shift0], 1u);
            }
            __syncthreads();
            ds_reserve_lite(S, nb0, C0, 0u, nb0);
            constexpr int PIECE = (W + M - 1) / M;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (!nz[h]) continue;
                const u32 i = blk + (u32)h * DS_THREADS + threadIdx.x;
                u32 flags = 0;
                if (absent0 || absent1) {   // bits 30 / 31 of every value this scalar emits: its point is absent from group 0 / 1
                    const u32 word = i >> 5, bit = i & 31u;
                    if (absent0 && ((absent0[word] >> bit) & 1u)) flags |= VAL_ABSENT0;
                    if (absent1 && ((absent1[word] >> bit) & 1u)) flags |= VAL_ABSENT1;
                }
#pragma unroll
                for (int w = 0; w < W; ++w)
                    if ((nz[h] >> w) & 1u) {
                        const u32 dest = atomicAdd(&S.cur[key[h][w] >> shift0], 1u);
                        out_k[dest] = key[h][w];
                        out_v[dest] = ((i * (u32)M + (u32)(w / PIECE)) << 1) | ((neg[h] >> w) & 1u) | flags;
                    }
            }
            __syncthreads();
        }
    }
}

// ------------------------------------------------------------------------------------------------ exclusive scan of a counter array, in place
constexpr u32 SCAN_BLOCK = 4096;   // 256 threads x 16
__global__ __launch_bounds__(256) void k_scan_sums(const u32* __restrict__ C, u32 n, u32* __restrict__ bs) {
    __shared__ u32 ws[4];
    const u32 base = blockIdx.x * SCAN_BLOCK;
    u32 x = 0;
    for (u32 k = 0; k < 16; ++k) { const u32 i = base + k * 256u + threadIdx.x; if (i < n) x += C[i]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_down(x, off);
    if ((threadIdx.x & 63u) == 0u) ws[threadIdx.x >> 6] = x;
    __syncthreads();
    if (threadIdx.x == 0) bs[blockIdx.x] = ws[0] + ws[1] + ws[2] + ws[3];
}
// one workgroup: exclusive scan of the block sums (a few thousand at most), in place
__global__ __launch_bounds__(256) void k_scan_top(u32* __restrict__ bs, u32 nblocks) {
    __shared__ u32 ws[4];
    __shared__ u32 carry_s;
    if (threadIdx.x == 0) carry_s = 0u;
    __syncthreads();
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    for (u32 base = 0; base < nblocks; base += 256u) {
        const u32 i = base + threadIdx.x;
        const u32 v = i < nblocks ? bs[i] : 0u;
        u32 x = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) { const u32 y = __shfl_up(x, off); if ((int)lane >= off) x += y; }
        if (lane == 63u) ws[wv] = x;
        __syncthreads();
        u32 ex = carry_s + x - v;
        for (u32 w = 0; w < wv; ++w) ex += ws[w];
        if (i < nblocks) bs[i] = ex;
        __syncthreads();
        if (threadIdx.x == 255u) carry_s = ex + v;
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_scan_apply(u32* __restrict__ C, u32 n, const u32* __restrict__ bs) {
    __shared__ u32 ws[4];
    const u32 base = blockIdx.x * SCAN_BLOCK + threadIdx.x * 16u;   // 16 consecutive counters per thread
    u32 v[16];
    u32 s = 0;
#pragma unroll
    for (u32 k = 0; k < 16; ++k) { v[k] = base + k < n ? C[base + k] : 0u; s += v[k]; }
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    u32 x = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const u32 y = __shfl_up(x, off); if ((int)lane >= off) x += y; }
    if (lane == 63u) ws[wv] = x;
    __syncthreads();
    u32 ex = bs[blockIdx.x] + x - s;
    for (u32 w = 0; w < wv; ++w) ex += ws[w];
#pragma unroll
    for (u32 k = 0; k < 16; ++k) { if (base + k < n) C[base + k] = ex; ex += v[k]; }
}

int32_t scan_in_place(zkpor_ctx* ctx, u32* C, u32 n, u32* bs) {
    const u32 nblocks = (n + SCAN_BLOCK - 1u) / SCAN_BLOCK;
    hipLaunchKernelGGL(k_scan_sums, dim3(nblocks), dim3(256), 0, ctx->stream, C, n, bs);
    hipLaunchKernelGGL(k_scan_top, dim3(1), dim3(256), 0, ctx->stream, bs, nblocks);
    hipLaunchKernelGGL(k_scan_apply, dim3(nblocks), dim3(256), 0, ctx->stream, C, n, bs);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
}  // namespace

// the levels of a key space of NB buckets: the bits are dealt as evenly as ceil(key_bits / 9) levels allow, the narrower levels first (level 0
// has the emptiest tiles: a witness vector is half zeros)
DigitSortPlan digit_sort_plan(const MsmCfg& cfg) {
    DigitSortPlan p;
    const int kb = cfg.key_bits;
    p.nlev = (kb + (int)DS_MAXR - 1) / (int)DS_MAXR;
    if (p.nlev < 1) p.nlev = 1;
    const int base = kb / p.nlev, extra = kb % p.nlev;
    int used = 0;
    size_t off = 0;
    for (int l = 0; l < p.nlev; ++l) {
        p.r[l] = base + (l >= p.nlev - extra ? 1 : 0);
        used += p.r[l];
        p.shift[l] = kb - used;
        p.n_child[l] = ((cfg.NB - 1u) >> p.shift[l]) + 1u;
        p.off_C[l] = off;
        off += align_up((size_t)p.n_child[l] * sizeof(u32), 256);
    }
    const u32 nblocks = (p.n_child[p.nlev - 1] + SCAN_BLOCK - 1u) / SCAN_BLOCK;
    p.off_bs = off;
    off += align_up((size_t)(nblocks + 1u) * sizeof(u32), 256);
    p.bytes = off;
    return p;
}

// counter: 64 zeroable bytes = [0] entries, [1], [2] per-array entries, [4 ..] the kernels' tickets.  temp: plan.bytes.  (kA, vA), (kB, vB): two buffer
// pairs of n * W entries; the grouped stream ends up in one of them.  Synchronises ctx->stream once (the host needs the entry counts to size the
// accumulation launches).
int32_t digit_sort(zkpor_ctx* ctx, const Fr* d_scalars, u32 n, const MsmCfg& cfg, const DigitSortPlan& plan, u32* kA, u32* vA, u32* kB, u32* vB,
                   u32* counter, char* temp, const u32* absent0, const u32* absent1, u32 Ms[3], u32** k_out, u32** v_out) {
    const DsCfg dc{cfg.c, cfg.W, cfg.m, cfg.piece, cfg.bpw};
    int grid = ctx->sort_grid;
    if (grid <= 0) {
        hipDeviceProp_t prop;
        ZK_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
        grid = prop.multiProcessorCount / 2;      // one 256-thread workgroup on every second compute unit: what the sort costs the main stream grows with its residence, and
        if (grid < 1) grid = 1;                   // 128 workgroups still finish both digit streams inside a proof (profiles/r06_sort_grid_tile_sweep.json: 64 starve the accumulations)
    }
    const int tile = ctx->sort_tile == 1024 || ctx->sort_tile == 2048 ? ctx->sort_tile : 4096;
    // the shapes with a compile-time level 0: 22-bit windows over 4 tables (the production key), 20-bit windows over plain arrays; "sort_generic" 1 = never
    const bool lite = ctx->sort_stage == 0;      // "sort_stage" 0: entries go straight to memory (4 KB of LDS per workgroup), 1: staged through LDS
    const int spec = (ctx->sort_generic || tile != 4096) ? 0 : (cfg.c == 22 && cfg.W == 12 && cfg.m == 4) ? 1 : (cfg.c == 20 && cfg.W == 13 && cfg.m == 1) ? 2 : 0;
    u32* C[4];
    for (int l = 0; l < plan.nlev; ++l) C[l] = (u32*)(temp + plan.off_C[l]);
    u32* bs = (u32*)(temp + plan.off_bs);
    u32* ticket = counter + 4;
    {
        PhaseScope ps(ctx, "msm_decompose");
        ZK_HIP(ctx, hipMemsetAsync(counter, 0, 64, ctx->stream));
        ZK_HIP(ctx, hipMemsetAsync(temp, 0, plan.off_bs, ctx->stream));
        const u32 tickets0 = (n + DS_SCALARS_PER_TICKET - 1u) / DS_SCALARS_PER_TICKET;
        const u32 g0 = tickets0 < (u32)grid ? tickets0 : (u32)grid;
        if (spec == 1) hipLaunchKernelGGL((k_dsort_count0_s<22, 12, 4>), dim3(g0 ? g0 : 1u), dim3(DS_THREADS), 0, ctx->stream, d_scalars, n, plan.shift[0], plan.n_child[0], C[0], counter, ticket, absent0, absent1);
        else if (spec == 2) hipLaunchKernelGGL((k_dsort_count0_s<20, 13, 1>), dim3(g0 ? g0 : 1u), dim3(DS_THREADS), 0, ctx->stream, d_scalars, n, plan.shift[0], plan.n_child[0], C[0], counter, ticket, absent0, absent1);
        else hipLaunchKernelGGL(k_dsort_count0, dim3(g0 ? g0 : 1u), dim3(DS_THREADS), 0, ctx->stream, d_scalars, n, dc, plan.shift[0], plan.n_child[0], C[0], counter, ticket,
                           absent0, absent1);
        ZK_KERNEL_CHECK(ctx);
    }
    ZK_HIP(ctx, hipMemcpyAsync(Ms, counter, 12, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const u32 M = Ms[0];
    *k_out = kB; *v_out = vB;
    if (M == 0) return ZKPOR_OK;
    PhaseScope ps(ctx, "msm_sort");
    {
        ZK_TRY(scan_in_place(ctx, C[0], plan.n_child[0], bs));
        const u32 tickets0 = (n + DS_SCALARS_PER_TICKET - 1u) / DS_SCALARS_PER_TICKET;
        const u32 g0 = tickets0 < (u32)grid ? tickets0 : (u32)grid;
        if (spec == 1 && lite) hipLaunchKernelGGL((k_dsort_scatter0_lite<22, 12, 4>), dim3(g0), dim3(DS_THREADS), 0, ctx->stream, d_scalars, n, plan.shift[0], plan.n_child[0], C[0], ticket + 1, absent0, absent1, kB, vB);
        else if (spec == 2 && lite) hipLaunchKernelGGL((k_dsort_scatter0_lite<20, 13, 1>), dim3(g0), dim3(DS_THREADS), 0, ctx->stream, d_scalars, n, plan.shift[0], plan.n_child[0], C[0], ticket + 1, absent0, absent1, kB, vB);
        else if (spec == 1) hipLaunchKernelGGL((k_dsort_scatter0_s<22, 12, 4>), dim3(g0), dim3(DS_THREADS), 0, ctx->stream, d_scalars, n, plan.shift[0], plan.n_child[0], C[0], ticket + 1, absent0, absent1, kB, vB);
        else if (spec == 2) hipLaunchKernelGGL((k_dsort_scatter0_s<20, 13, 1>), dim3(g0), dim3(DS_THREADS), 0, ctx->stream, d_scalars, n, plan.shift[0], plan.n_child[0], C[0], ticket + 1, absent0, absent1, kB, vB);
        else if (tile == 1024) hipLaunchKernelGGL(k_dsort_scatter0<1024>, dim3(g0), dim3(DS_THREADS), 0, ctx->stream, d_scalars, n, dc, plan.shift[0], plan.n_child[0], C[0], ticket + 1, absent0, absent1, kB, vB);
        else if (tile == 2048) hipLaunchKernelGGL(k_dsort_scatter0<2048>, dim3(g0), dim3(DS_THREADS), 0, ctx->stream, d_scalars, n, dc, plan.shift[0], plan.n_child[0], C[0], ticket + 1, absent0, absent1, kB, vB);
        else hipLaunchKernelGGL(k_dsort_scatter0<4096>, dim3(g0), dim3(DS_THREADS), 0, ctx->stream, d_scalars, n, dc, plan.shift[0], plan.n_child[0], C[0], ticket + 1, absent0, absent1, kB, vB);
        ZK_KERNEL_CHECK(ctx);
    }
    u32 *src_k = kB, *src_v = vB, *dst_k = kA, *dst_v = vA;
    const u32 chunks = (M + DS_CHUNK - 1u) / DS_CHUNK;
    const u32 g = chunks < (u32)grid ? chunks : (u32)grid;
    for (int l = 1; l < plan.nlev; ++l) {
        hipLaunchKernelGGL(k_dsort_count, dim3(g), dim3(DS_THREADS), 0, ctx->stream, src_k, M, plan.shift[l], plan.r[l], C[l - 1], plan.n_child[l - 1], C[l], plan.n_child[l],
                           ticket + 2 * l);
        ZK_KERNEL_CHECK(ctx);
        ZK_TRY(scan_in_place(ctx, C[l], plan.n_child[l], bs));
        if (lite) hipLaunchKernelGGL(k_dsort_scatter_lite, dim3(g), dim3(DS_THREADS), 0, ctx->stream, src_k, src_v, M, plan.shift[l], plan.r[l], C[l - 1], plan.n_child[l - 1], C[l], plan.n_child[l], ticket + 2 * l + 1, dst_k, dst_v);
        else if (tile == 1024) hipLaunchKernelGGL(k_dsort_scatter<1024>, dim3(g), dim3(DS_THREADS), 0, ctx->stream, src_k, src_v, M, plan.shift[l], plan.r[l], C[l - 1], plan.n_child[l - 1], C[l], plan.n_child[l], ticket + 2 * l + 1, dst_k, dst_v);
        else if (tile == 2048) hipLaunchKernelGGL(k_dsort_scatter<2048>, dim3(g), dim3(DS_THREADS), 0, ctx->stream, src_k, src_v, M, plan.shift[l], plan.r[l], C[l - 1], plan.n_child[l - 1], C[l], plan.n_child[l], ticket + 2 * l + 1, dst_k, dst_v);
        else hipLaunchKernelGGL(k_dsort_scatter<4096>, dim3(g), dim3(DS_THREADS), 0, ctx->stream, src_k, src_v, M, plan.shift[l], plan.r[l], C[l - 1], plan.n_child[l - 1], C[l], plan.n_child[l], ticket + 2 * l + 1, dst_k, dst_v);
        ZK_KERNEL_CHECK(ctx);
        std::swap(src_k, dst_k); std::swap(src_v, dst_v);
    }
    *k_out = src_k; *v_out = src_v;
    return ZKPOR_OK;
}

}  // namespace zk
