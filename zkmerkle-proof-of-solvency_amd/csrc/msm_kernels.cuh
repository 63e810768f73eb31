// Device kernels of the Pippenger MSM (see msm.cuh for the algorithm).  Included only by the translation units
// that instantiate them: msm_digits.hip, msm_g1_hot.hip, msm_g1_cold.hip, msm_g2.hip.
#pragma once
#include "msm.cuh"

namespace zk {

// ------------------------------------------------------------------------------------------------ accumulate
template <class F>
struct AccOut {
    XYZZ<F>* buckets;
    u32* out_keys;
    XYZZ<F>* out_part;
};

// Level 1: affine key points gathered through the sorted (key,val) stream.
// The (key,val) words of a block's chunks are staged through LDS in sub-phases of ACC_SUB entries per thread with
// coalesced global loads: read one-by-one they cost a 64-byte fabric request per 4-byte word (measured with
// rocprofv3 FETCH_SIZE: 185 B/entry instead of ~72), which made this kernel HBM-bound.
#ifndef ZK_ACC_SUB   // experiment hook (tools/r03_accsub.sh): entries per thread and staging phase; LDS per workgroup = 2 x 256 x (ZK_ACC_SUB + 1) x 4 B
#define ZK_ACC_SUB 16
#endif
static constexpr int ACC_SUB = ZK_ACC_SUB;
static constexpr int ACC_PITCH = ACC_SUB + 1;  // odd pitch: conflict-free column reads

// cooperative load of phase `ph`: row r of the tile = the ACC_SUB entries [ (row0+r)*L + ph*ACC_SUB, ... ) of thread r
ZK_D void acc_stage(const u32* __restrict__ keys, const u32* __restrict__ vals, u32 M, int L, u32 row0, u32 rows, int ph,
                    u32* sk, u32* sv) {
    const u32 total = rows * (u32)ACC_SUB;
    for (u32 idx = threadIdx.x; idx < total; idx += blockDim.x) {
        u32 r = idx / (u32)ACC_SUB, c = idx % (u32)ACC_SUB;
        u32 col = (u32)ph * ACC_SUB + c;
        u64 g = (u64)(row0 + r) * (u32)L + col;
        if (col < (u32)L && g < M) {
            sk[r * ACC_PITCH + c] = keys[g];
            sv[r * ACC_PITCH + c] = vals[g] & VAL_MASK;   // bits 30 / 31 carry the per-array absence flags of the filter (msm_digits.hip)
        }
    }
}

template <class F>
__global__ __launch_bounds__(256) void k_acc_level1(const u32* __restrict__ keys, const u32* __restrict__ vals,
                                                    const Affine<F>* __restrict__ pts, u32 M, int L,
                                                    XYZZ<F>* __restrict__ buckets, u32* __restrict__ out_keys,
                                                    XYZZ<F>* __restrict__ out_part) {
    __shared__ u32 sk[256 * ACC_PITCH];
    __shared__ u32 sv[256 * ACC_PITCH];
    const u32 row0 = blockIdx.x * 256u;
    const u32 t = row0 + threadIdx.x;
    const u32 T = (M + (u32)L - 1u) / (u32)L;
    const bool live = t < T;
    const u32 start = live ? t * (u32)L : 0u;
    const u32 end = live ? ((start + (u32)L < M) ? start + (u32)L : M) : 0u;
    const u32 prev = (live && start > 0) ? keys[start - 1] : NOKEY;
    const u32 next = (live && end < M) ? keys[end] : NOKEY;
    XYZZ<F> acc = XYZZ<F>::inf();
    u32 cur = live ? keys[start] : NOKEY;
    const u32 first_key = cur;
    u32 last_key = cur;
    bool first = true, head_written = false, tail_written = false;
    const int nphase = (L + ACC_SUB - 1) / ACC_SUB;
    const u32 rows = (T - row0 < 256u) ? T - row0 : 256u;
    for (int ph = 0; ph < nphase; ++ph) {
        if (ph) __syncthreads();
        acc_stage(keys, vals, M, L, row0, rows, ph, sk, sv);
        __syncthreads();
        if (!live) continue;
        const u32 j0 = start + (u32)ph * ACC_SUB;
        const u32 j1 = (j0 + ACC_SUB < end) ? j0 + ACC_SUB : end;
        for (u32 j = j0; j < j1; ++j) {
            const u32 k = sk[threadIdx.x * ACC_PITCH + (j - j0)];
            const u32 v = sv[threadIdx.x * ACC_PITCH + (j - j0)];
            if (k != cur) {
                if (first && cur == prev) { out_part[2 * t] = acc; head_written = true; }
                else buckets[cur] = acc;
                first = false;
                cur = k;
                acc = XYZZ<F>::inf();
            }
            last_key = k;
            Affine<F> p = pts[v >> 1];
            if (!p.is_inf()) {
                if (v & 1u) p.y = F::neg(p.y);
                xyzz_madd<F>(acc, p.x, p.y);
            }
        }
    }
    if (!live) return;
    if (first && cur == prev) { out_part[2 * t] = acc; head_written = true; }
    else if (cur == next) { out_part[2 * t + 1] = acc; tail_written = true; }
    else buckets[cur] = acc;
    if (T > 1) {
        if (!head_written) out_part[2 * t] = XYZZ<F>::inf();
        if (!tail_written) out_part[2 * t + 1] = XYZZ<F>::inf();
        out_keys[2 * t] = first_key;
        out_keys[2 * t + 1] = last_key;
    }
}

// Level >= 2: the entries are XYZZ partial sums (keys still sorted); finished runs are ADDED into their bucket
// (a bucket is touched by exactly one thread per level, and levels are separate launches).
template <class F>
__global__ __launch_bounds__(256) void k_acc_levelN(const u32* __restrict__ keys, const XYZZ<F>* __restrict__ src,
                                                    u32 M, int L, XYZZ<F>* __restrict__ buckets,
                                                    u32* __restrict__ out_keys, XYZZ<F>* __restrict__ out_part) {
    const u32 t = blockIdx.x * 256u + threadIdx.x;
    const u32 T = (M + (u32)L - 1u) / (u32)L;
    if (t >= T) return;
    const u32 start = t * (u32)L;
    const u32 end = (start + (u32)L < M) ? start + (u32)L : M;
    const u32 prev = start > 0 ? keys[start - 1] : NOKEY;
    const u32 next = end < M ? keys[end] : NOKEY;
    XYZZ<F> acc = XYZZ<F>::inf();
    u32 cur = keys[start];
    bool first = true, head_written = false, tail_written = false;
    // a bucket receives exactly one non-infinity finished run over the whole recursion (its entries are contiguous:
    // either it finished inside one chunk at an earlier level, or all of it is still in flight), so a plain
    // store suffices; entries with an infinity payload only pad the key stream
    auto finish = [&](u32 key) {
        if (!acc.is_inf()) buckets[key] = acc;
    };
    for (u32 j = start; j < end; ++j) {
        const u32 k = keys[j];
        if (k != cur) {
            if (first && cur == prev) { out_part[2 * t] = acc; head_written = true; }
            else finish(cur);
            first = false;
            cur = k;
            acc = XYZZ<F>::inf();
        }
        XYZZ<F> p = src[j];
        xyzz_add<F>(acc, p);
    }
    if (first && cur == prev) { out_part[2 * t] = acc; head_written = true; }
    else if (cur == next) { out_part[2 * t + 1] = acc; tail_written = true; }
    else finish(cur);
    if (T > 1) {
        if (!head_written) out_part[2 * t] = XYZZ<F>::inf();
        if (!tail_written) out_part[2 * t + 1] = XYZZ<F>::inf();
        out_keys[2 * t] = keys[start];
        out_keys[2 * t + 1] = keys[end - 1];
    }
}

// ------------------------------------------------------------------------------------------------ reduce
// One level of the bucket reduction.  Group j covers Sin[j*g .. j*g+g):
//   Sout[j] = sum_k x_k                                   (plain sum: the next level's input)
//   Yout[j] = sum_k Yin[j*g+k]  +  2^dbl * sum_k (k+1) x_k  (weighted sums of all levels so far, pre-scaled)
// With g_1..g_L the group sizes and T the grand total, the window sum is  Y_L - T * (g_1 + g_1 g_2 + ...)
// (host side, msm_accumulate).  Small groups (16) keep every level wide and the dependent-add chain short.
template <class F, bool HAS_Y>
__global__ __launch_bounds__(64) void k_reduce_level(const XYZZ<F>* __restrict__ Sin, const XYZZ<F>* __restrict__ Yin,
                                                     u32 n_groups, u32 g, int dbl, XYZZ<F>* __restrict__ Sout,
                                                     XYZZ<F>* __restrict__ Yout) {
    const u32 j = blockIdx.x * 64u + threadIdx.x;
    if (j >= n_groups) return;
    XYZZ<F> run = XYZZ<F>::inf(), wacc = XYZZ<F>::inf(), ysum = XYZZ<F>::inf();
    const size_t base = (size_t)j * g;
    for (u32 k = g; k-- > 0;) {
        XYZZ<F> x = Sin[base + k];
        xyzz_add<F>(run, x);
        xyzz_add<F>(wacc, run);
        if (HAS_Y) {
            XYZZ<F> y = Yin[base + k];
            xyzz_add<F>(ysum, y);
        }
    }
    Sout[j] = run;
    if (HAS_Y) {
        for (int d = 0; d < dbl; ++d) wacc = xyzz_dbl<F>(wacc);
        xyzz_add<F>(ysum, wacc);
        Yout[j] = ysum;
    } else {
        Yout[j] = wacc;  // first level: dbl == 0 and there is no previous Y
    }
}

}  // namespace zk
