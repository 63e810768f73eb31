// Partial-sum recursion and bucket reduction on raw 29-bit accumulator images (fe29.cuh): the stages after the level-1
// accumulation never leave the lazy 9 x 29-bit form — buckets, partials, running sums are all XYZZ29 register images in
// memory, the products are the 206-instruction inline ones, and only the W per-window results are converted (on the
// host).  Same algorithms as k_acc_levelN / k_reduce_level (msm_kernels.cuh), generic over a policy:
//   G1: one accumulator per lane, image = 36 words;  G2: one accumulator per lane PAIR (fp2_lanepair.cuh), each lane holds
//   one Fp2 component, image = 2 x 36 words.
#pragma once
#include "msm.cuh"
#include "fe29.cuh"

namespace zk {

static constexpr u32 PART_EMPTY = 0x80000000u, PART_KEY = 0x7fffffffu;  // see k_acc_levelN29

struct Pol29G1 {
    typedef Fp29 F;
    typedef XYZZ29T<Fp29> Acc;
    static constexpr u32 LANES = 1, WORDS = RAW29_WORDS;
    ZK_D static Acc load(const u32* base, size_t idx, u32) { return raw29_load(base + idx * WORDS); }
    ZK_D static void store(u32* base, size_t idx, const Acc& a, u32) { raw29_store(base + idx * WORDS, a); }
};

// level >= 2 of the segmented sum: entries are partial sums, keys still sorted; see k_acc_levelN
template <class Pol>
__global__ __launch_bounds__(256) void k_acc_levelN29(const u32* __restrict__ keys, const u32* __restrict__ src, u32 M, int L,
                                                      u32* __restrict__ buckets, u32* __restrict__ out_keys, u32* __restrict__ out_part) {
    typedef typename Pol::Acc Acc;
    const u32 gt = blockIdx.x * 256u + threadIdx.x;
    const u32 t = gt / Pol::LANES, par = gt % Pol::LANES;
    const u32 T = (M + (u32)L - 1u) / (u32)L;
    if (t >= T) return;
    const u32 start = t * (u32)L;
    const u32 end = (start + (u32)L < M) ? start + (u32)L : M;
    // bit 31 of a key marks an entry whose partial sum is infinity and was never written (PART_EMPTY): most chunks lie
    // inside one bucket, so half of all partials are; skipping their 144-byte images halves this kernel's HBM traffic
    const u32 prev = start > 0 ? (keys[start - 1] & PART_KEY) : NOKEY;
    const u32 next = end < M ? (keys[end] & PART_KEY) : NOKEY;
    Acc acc = Acc::inf();
    u32 cur = keys[start] & PART_KEY;
    const u32 first_key = cur;
    u32 last_key = cur;
    bool first = true, head_written = false;
    for (u32 j = start; j < end; ++j) {
        const u32 kraw = keys[j];
        const u32 k = kraw & PART_KEY;
        if (k != cur) {
            const bool head = first && cur == prev;
            if (head) { Pol::store(out_part, 2 * (size_t)t, acc, par); head_written = true; }
            else if (!acc.is_inf()) Pol::store(buckets, cur, acc, par);
            first = false;
            cur = k;
            acc = Acc::inf();
        }
        last_key = k;
        if (!(kraw & PART_EMPTY)) {
            Acc p = Pol::load(src, j, par);
            xyzz29_add<typename Pol::F>(acc, p);
        }
    }
    const bool acc_head = first && cur == prev;
    const bool acc_tail = !acc_head && cur == next;
    if (!acc_head && !acc_tail && !acc.is_inf()) Pol::store(buckets, cur, acc, par);
    if (T > 1) {
        if (acc_head) Pol::store(out_part, 2 * (size_t)t, acc, par);
        if (acc_tail) Pol::store(out_part, 2 * (size_t)t + 1, acc, par);
        if (par == 0) {
            out_keys[2 * t] = first_key | ((acc_head || head_written) ? 0u : PART_EMPTY);
            out_keys[2 * t + 1] = last_key | (acc_tail ? 0u : PART_EMPTY);
        }
    }
}

// one level of the bucket reduction; see k_reduce_level for the algebra.  The additions of one step share ONE copy of
// the addition code (the step index is wave-uniform), which keeps the loop body inside the instruction cache.
template <class Pol, bool HAS_Y>
__global__ __launch_bounds__(128) void k_reduce_level29(const u32* __restrict__ Sin, const u32* __restrict__ Yin, u32 n_groups, u32 g,
                                                        int dbl, u32* __restrict__ Sout, u32* __restrict__ Yout) {
    typedef typename Pol::Acc Acc;
    typedef typename Pol::F F;
    const u32 gt = blockIdx.x * 128u + threadIdx.x;
    const u32 j = gt / Pol::LANES, par = gt % Pol::LANES;
    if (j >= n_groups) return;
    Acc run = Acc::inf(), wacc = Acc::inf(), ysum = Acc::inf();
    const size_t base = (size_t)j * g;
    for (u32 k = g; k-- > 0;) {
#pragma unroll 1
        for (int s = 0; s < (HAS_Y ? 3 : 2); ++s) {
            Acc a, b;
            if (s == 0) { a = run; b = Pol::load(Sin, base + k, par); }
            else if (s == 1) { a = wacc; b = run; }
            else { a = ysum; b = Pol::load(Yin, base + k, par); }
            xyzz29_add<F>(a, b);
            if (s == 0) run = a; else if (s == 1) wacc = a; else ysum = a;
        }
    }
    Pol::store(Sout, j, run, par);
    if (HAS_Y) {
        for (int d = 0; d < dbl; ++d) wacc = xyzz29_dbl<F>(wacc);
        xyzz29_add<F>(ysum, wacc);
        Pol::store(Yout, j, ysum, par);
    } else {
        Pol::store(Yout, j, wacc, par);
    }
}

}  // namespace zk
