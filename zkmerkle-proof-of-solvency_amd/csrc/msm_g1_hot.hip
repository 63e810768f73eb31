// MSM step 3, G1, level 1 — the dominant kernel of the prover: field products fully inlined.
#include "msm_kernels.cuh"
#include "msm_kernels29.cuh"
// occupancy experiment hook (tools/r02_occupancy.sh): -DZK_L1_WAVES=n pins the level-1 kernels to n waves per SIMD; the default lets
// the register allocator decide (149 VGPRs -> 3 waves per SIMD)
#ifdef ZK_L1_WAVES
#define ZK_L1_OCCUPANCY __attribute__((amdgpu_waves_per_eu(ZK_L1_WAVES, ZK_L1_WAVES)))
#else
#define ZK_L1_OCCUPANCY
#endif
namespace zk {

// 2 * (+/- P) for a key point, out of line and fed from memory (see xyzz29_madd)
__device__ __noinline__ XYZZ29 dbl_point_g1(const Affine<Fp>* p, bool neg) {
    Affine<Fp> q = *p;
    return xyzz29_dbl_affine<Fp29>(Fp29::from32<5>(q.x), Fp29::cneg(Fp29::from32<5>(q.y), neg));
}

// k_acc_level1<Fp> with the accumulator and all arithmetic on 9 x 29-bit limbs (fe29.cuh).  Everything it writes is a
// raw register image (raw29_store): finished buckets go to `braw`, the <= 2 runs cut by the chunk edge to `praw` (2 per
// chunk) — the key-change path, taken by some lane of a wave in about half of all iterations, is 36 plain stores.  The
// later stages (msm_kernels29.cuh) consume the images as they are.
__global__ __launch_bounds__(256) ZK_L1_OCCUPANCY void k_acc_level1_fp29(const u32* __restrict__ keys, const u32* __restrict__ vals,
                                                         const Affine<Fp>* __restrict__ pts, u32 M, int L,
                                                         u32* __restrict__ braw, u32* __restrict__ out_keys, u32* __restrict__ praw) {
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
    XYZZ29 acc = XYZZ29::inf();
    u32 cur = live ? keys[start] : NOKEY;
    const u32 first_key = cur;
    u32 last_key = cur;
    bool first = true, head_written = false;
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
                const bool head = first && cur == prev;
                raw29_store(head ? praw + 2 * (size_t)t * RAW29_WORDS : braw + (size_t)cur * RAW29_WORDS, acc);
                head_written |= head;
                first = false;
                cur = k;
                acc = XYZZ29::inf();
            }
            last_key = k;
            Affine<Fp> p = pts[v >> 1];
            if (!p.is_inf()) {
                const Affine<Fp>* pp = pts + (v >> 1);
                const bool neg = (v & 1u) != 0;
                xyzz29_madd<Fp29>(acc, Fp29::from32<5>(p.x), Fp29::cneg(Fp29::from32<5>(p.y), neg), [=]() { return dbl_point_g1(pp, neg); });
            }
        }
    }
    if (!live) return;
    // the chunk's last run: a head (the previous chunk ends in the same bucket), a tail (the next chunk starts in it) or complete
    const bool acc_head = first && cur == prev;
    const bool acc_tail = !acc_head && cur == next;
    if (!acc_head && !acc_tail) raw29_store(braw + (size_t)cur * RAW29_WORDS, acc);
    if (T > 1) {  // a partial that does not exist is flagged in its key (PART_EMPTY) instead of being written as zeros
        if (acc_head) raw29_store(praw + 2 * (size_t)t * RAW29_WORDS, acc);
        if (acc_tail) raw29_store(praw + (2 * (size_t)t + 1) * RAW29_WORDS, acc);
        out_keys[2 * t] = first_key | ((acc_head || head_written) ? 0u : PART_EMPTY);
        out_keys[2 * t + 1] = last_key | (acc_tail ? 0u : PART_EMPTY);
    }
}

int32_t launch_level1(zkpor_ctx* ctx, const u32* keys, const u32* vals, const Affine<Fp>* pts, u32 M, int L, u32 NB,
                      XYZZ<Fp>* buckets, u32* out_keys, XYZZ<Fp>* out_part) {
    u32 T = (M + (u32)L - 1u) / (u32)L;
    ZK_HIP(ctx, hipMemsetAsync(buckets, 0, (size_t)NB * sizeof(XYZZ<Fp>), ctx->stream));
    {
        PhaseScope ps(ctx, "k_acc_level1_g1");  // the kernel alone: bench.py's roofline launch time must match rocprofv3's
        hipLaunchKernelGGL(k_acc_level1<Fp>, dim3((T + 255u) / 256u), dim3(256), 0, ctx->stream, keys, vals, pts, M, L, buckets, out_keys, out_part);
    }
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
// ---- the 29-bit pipeline (raw images end to end) ----
int32_t launch_level1_29(zkpor_ctx* ctx, const u32* keys, const u32* vals, const Affine<Fp>* pts, u32 M, int L, u32 NB,
                         u32* braw, u32* out_keys, u32* praw) {
    u32 T = (M + (u32)L - 1u) / (u32)L;
    ZK_HIP(ctx, hipMemsetAsync(braw, 0, (size_t)NB * RAW29_WORDS * 4, ctx->stream));
    {
        PhaseScope ps(ctx, "k_acc_level1_g1");
        hipLaunchKernelGGL(k_acc_level1_fp29, dim3((T + 255u) / 256u), dim3(256), 0, ctx->stream, keys, vals, pts, M, L, braw, out_keys, praw);
    }
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
template <>
int32_t launch_levelN29<Fp>(zkpor_ctx* ctx, const u32* keys, const u32* src, u32 M, int L, u32* braw, u32* out_keys, u32* out_part) {
    u32 T = (M + (u32)L - 1u) / (u32)L;
    hipLaunchKernelGGL(k_acc_levelN29<Pol29G1>, dim3((T + 255u) / 256u), dim3(256), 0, ctx->stream, keys, src, M, L, braw, out_keys, out_part);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
// The same level for SMALL inputs (the last three or four levels of every reduction: a few thousand buckets down to one per
// window), one LANE per bucket instead of one thread per group.  k_reduce_level29 walks a group's g buckets with a serial chain
// of 3 g additions — with fewer groups than the GPU has SIMDs that chain IS the kernel's duration (0.3 ms per level whatever its
// size).  Here the g lanes of a group compute the same three sums as a suffix scan and two tree sums: 3 log2(g) additions deep.
//     run  = sum_k S_k                 = T_0,  T_k = sum_{k' >= k} S_k'   (inclusive suffix scan)
//     wacc = sum_k (k + 1) S_k         = sum_k T_k                          (tree sum of the scan)
//     ysum = sum_k Y_k                                                       (tree sum)
// One loop, one copy of the addition code (the step is wave-uniform); the operand of the other lane comes through 36 lane shuffles.
ZK_D XYZZ29 shfl_down29(const XYZZ29& a, u32 d) {
    XYZZ29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.x.l[i] = (u32)__shfl_down((int)a.x.l[i], d, 64); r.y.l[i] = (u32)__shfl_down((int)a.y.l[i], d, 64);
        r.zz.l[i] = (u32)__shfl_down((int)a.zz.l[i], d, 64); r.zzz.l[i] = (u32)__shfl_down((int)a.zzz.l[i], d, 64);
    }
    return r;
}
template <bool HAS_Y>
__global__ __launch_bounds__(256) void k_reduce_scan29_g1(const u32* __restrict__ Sin, const u32* __restrict__ Yin, u32 n_groups, int gl,
                                                           int dbl, u32* __restrict__ Sout, u32* __restrict__ Yout) {
    const u32 g = 1u << gl;                          // <= 16: a group never straddles a wave
    const u32 gt = blockIdx.x * 256u + threadIdx.x;  // = j * g + k: the element's index
    const u32 j = gt >> gl, k = gt & (g - 1u);
    const bool live = j < n_groups;
    XYZZ29 T = live ? raw29_load(Sin + (size_t)gt * RAW29_WORDS) : XYZZ29::inf();
    XYZZ29 W = XYZZ29::inf(), Y = XYZZ29::inf();
    if (HAS_Y && live) Y = raw29_load(Yin + (size_t)gt * RAW29_WORDS);
    const int phases = HAS_Y ? 3 : 2;
#pragma unroll 1
    for (int it = 0; it < phases * gl; ++it) {
        const int phase = it / gl, st = it - phase * gl;
        const u32 d = 1u << st;
        if (phase == 1 && st == 0) W = T;            // the scan is complete: T_k on every lane
        XYZZ29 a = phase == 0 ? T : (phase == 1 ? W : Y);
        const XYZZ29 b = shfl_down29(a, d);          // every lane takes part in the shuffle
        const bool doit = phase == 0 ? (k + d < g) : ((k & (2u * d - 1u)) == 0u);
        if (doit) xyzz29_add<Fp29>(a, b);
        if (phase == 0) T = a; else if (phase == 1) W = a; else Y = a;
    }
    if (!live || k != 0) return;
    raw29_store(Sout + (size_t)j * RAW29_WORDS, T);
    if (HAS_Y) {
        for (int q = 0; q < dbl; ++q) W = xyzz29_dbl<Fp29>(W);
        xyzz29_add<Fp29>(Y, W);
        raw29_store(Yout + (size_t)j * RAW29_WORDS, Y);
    } else {
        raw29_store(Yout + (size_t)j * RAW29_WORDS, W);
    }
}

template <>
int32_t launch_reduce29<Fp>(zkpor_ctx* ctx, const u32* Sin, const u32* Yin, u32 n_groups, u32 g, int dbl, u32* Sout, u32* Yout) {
    // few buckets: one lane per bucket (above ~2^15 buckets the serial walk has enough threads to fill the GPU and does less work)
    if (ctx->msm_reduce_scan && g <= 16u && (g & (g - 1u)) == 0u && g >= 2u && (size_t)n_groups * g <= ((size_t)1 << 15)) {
        int gl = 0;
        while ((1u << gl) < g) ++gl;
        dim3 grid_s((unsigned)(((size_t)n_groups * g + 255u) / 256u));
        if (Yin) hipLaunchKernelGGL((k_reduce_scan29_g1<true>), grid_s, dim3(256), 0, ctx->stream, Sin, Yin, n_groups, gl, dbl, Sout, Yout);
        else hipLaunchKernelGGL((k_reduce_scan29_g1<false>), grid_s, dim3(256), 0, ctx->stream, Sin, Yin, n_groups, gl, dbl, Sout, Yout);
        ZK_KERNEL_CHECK(ctx);
        return ZKPOR_OK;
    }
    dim3 grid((n_groups + 127u) / 128u);
    if (Yin) hipLaunchKernelGGL((k_reduce_level29<Pol29G1, true>), grid, dim3(128), 0, ctx->stream, Sin, Yin, n_groups, g, dbl, Sout, Yout);
    else hipLaunchKernelGGL((k_reduce_level29<Pol29G1, false>), grid, dim3(128), 0, ctx->stream, Sin, Yin, n_groups, g, dbl, Sout, Yout);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
}  // namespace zk
