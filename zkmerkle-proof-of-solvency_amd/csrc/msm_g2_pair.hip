// MSM step 3, G2, level 1: the single most expensive launch of a proof.  One G2 accumulator per LANE PAIR
// (fp2_lanepair.cuh): even lanes carry the real, odd lanes the imaginary Fp component.
#include "msm_kernels.cuh"
#include "msm_kernels29.cuh"
#include "fp2_lanepair.cuh"

// occupancy experiment hook (tools/r02_occupancy.sh): -DZK_L1_WAVES=n pins the level-1 kernels to n waves per SIMD; the default lets
// the register allocator decide (149 VGPRs -> 3 waves per SIMD)
#ifdef ZK_L1_WAVES
#define ZK_L1_OCCUPANCY __attribute__((amdgpu_waves_per_eu(ZK_L1_WAVES, ZK_L1_WAVES)))
#else
#define ZK_L1_OCCUPANCY
#endif
namespace zk {

__device__ __forceinline__ void lp_store(XYZZ<Fp2>* dst, const XYZZ<Fp2L>& a, u32 par) {
    Fp* d = (Fp*)dst;
    d[par] = a.x.c; d[2 + par] = a.y.c; d[4 + par] = a.zz.c; d[6 + par] = a.zzz.c;
}

__global__ __launch_bounds__(256) void k_acc_level1_g2pair(const u32* __restrict__ keys, const u32* __restrict__ vals,
                                                           const Affine<Fp2>* __restrict__ pts, u32 M, int L,
                                                           XYZZ<Fp2>* __restrict__ buckets, u32* __restrict__ out_keys,
                                                           XYZZ<Fp2>* __restrict__ out_part) {
    __shared__ u32 sk[128 * ACC_PITCH];
    __shared__ u32 sv[128 * ACC_PITCH];
    const u32 row0 = blockIdx.x * 128u;          // 128 lane pairs per block
    const u32 lr = threadIdx.x >> 1, par = threadIdx.x & 1u;
    const u32 t = row0 + lr;
    const u32 T = (M + (u32)L - 1u) / (u32)L;
    const bool live = t < T;                       // both lanes of a pair agree
    const u32 start = live ? t * (u32)L : 0u;
    const u32 end = live ? ((start + (u32)L < M) ? start + (u32)L : M) : 0u;
    const u32 prev = (live && start > 0) ? keys[start - 1] : NOKEY;
    const u32 next = (live && end < M) ? keys[end] : NOKEY;
    XYZZ<Fp2L> acc = XYZZ<Fp2L>::inf();
    u32 cur = live ? keys[start] : NOKEY;
    const u32 first_key = cur;
    u32 last_key = cur;
    bool first = true, head_written = false, tail_written = false;
    const int nphase = (L + ACC_SUB - 1) / ACC_SUB;
    const u32 rows = (T - row0 < 128u) ? T - row0 : 128u;
    for (int ph = 0; ph < nphase; ++ph) {
        if (ph) __syncthreads();
        acc_stage(keys, vals, M, L, row0, rows, ph, sk, sv);
        __syncthreads();
        if (!live) continue;
        const u32 j0 = start + (u32)ph * ACC_SUB;
        const u32 j1 = (j0 + ACC_SUB < end) ? j0 + ACC_SUB : end;
        for (u32 j = j0; j < j1; ++j) {
            const u32 k = sk[lr * ACC_PITCH + (j - j0)];
            const u32 v = sv[lr * ACC_PITCH + (j - j0)];
            if (k != cur) {
                if (first && cur == prev) { lp_store(out_part + 2 * t, acc, par); head_written = true; }
                else lp_store(buckets + cur, acc, par);
                first = false;
                cur = k;
                acc = XYZZ<Fp2L>::inf();
            }
            last_key = k;
            const Fp* pp = (const Fp*)(pts + (v >> 1));
            Fp2L px = {pp[par]}, py = {pp[2 + par]};
            if (!(px.is_zero() & py.is_zero())) {
                if (v & 1u) py = Fp2L::neg(py);
                xyzz_madd<Fp2L>(acc, px, py);
            }
        }
    }
    if (!live) return;
    if (first && cur == prev) { lp_store(out_part + 2 * t, acc, par); head_written = true; }
    else if (cur == next) { lp_store(out_part + 2 * t + 1, acc, par); tail_written = true; }
    else lp_store(buckets + cur, acc, par);
    if (T > 1) {
        XYZZ<Fp2L> z = XYZZ<Fp2L>::inf();
        if (!head_written) lp_store(out_part + 2 * t, z, par);
        if (!tail_written) lp_store(out_part + 2 * t + 1, z, par);
        if (par == 0) {
            out_keys[2 * t] = first_key;
            out_keys[2 * t + 1] = last_key;
        }
    }
}

// ---- 29-bit signed lazy form of the level-1 kernel (fe29.cuh + Fp2L29) ----
// raw register image of a lane-pair accumulator: each lane parks its own component (RAW29_WORDS words; pair = 288 B)
__device__ __forceinline__ void lp_raw_store(u32* dst, const XYZZ29T<Fp2L29>& a, u32 par) {
    XYZZ29 c = {a.x.c, a.y.c, a.zz.c, a.zzz.c};
    raw29_store(dst + par * RAW29_WORDS, c);
}
__device__ __forceinline__ XYZZ29T<Fp2L29> lp_raw_load(const u32* src, u32 par) {
    XYZZ29 c = raw29_load(src + par * RAW29_WORDS);
    XYZZ29T<Fp2L29> a;
    a.x.c = c.x; a.y.c = c.y; a.zz.c = c.zz; a.zzz.c = c.zzz;
    return a;
}

// 2 * (+/- P) for a key point, out of line and fed from memory (see xyzz29_madd in fe29.cuh)
__device__ __noinline__ XYZZ29T<Fp2L29> dbl_point_g2(const Fp* pp, u32 par, bool neg) {
    Fp2L29 x = {Fp29::from32<5>(pp[par])};
    Fp2L29 y = {Fp29::cneg(Fp29::from32<5>(pp[2 + par]), neg)};
    return xyzz29_dbl_affine<Fp2L29>(x, y);
}

// the lane-pair policy of the 29-bit pipeline (msm_kernels29.cuh)
struct Pol29G2 {
    typedef Fp2L29 F;
    typedef XYZZ29T<Fp2L29> Acc;
    static constexpr u32 LANES = 2, WORDS = 2 * RAW29_WORDS;
    ZK_D static Acc load(const u32* base, size_t idx, u32 par) { return lp_raw_load(base + idx * WORDS, par); }
    ZK_D static void store(u32* base, size_t idx, const Acc& a, u32 par) { lp_raw_store(base + idx * WORDS, a, par); }
};

// see k_acc_level1_fp29 (msm_g1_hot.hip): raw images for buckets (braw) and the chunk's two partials (praw)
__global__ __launch_bounds__(256) ZK_L1_OCCUPANCY void k_acc_level1_g2pair29(const u32* __restrict__ keys, const u32* __restrict__ vals,
                                                             const Affine<Fp2>* __restrict__ pts, u32 M, int L,
                                                             u32* __restrict__ braw, u32* __restrict__ out_keys, u32* __restrict__ praw) {
    __shared__ u32 sk[128 * ACC_PITCH];
    __shared__ u32 sv[128 * ACC_PITCH];
    typedef XYZZ29T<Fp2L29> Acc;
    const u32 row0 = blockIdx.x * 128u;
    const u32 lr = threadIdx.x >> 1, par = threadIdx.x & 1u;
    const u32 t = row0 + lr;
    const u32 T = (M + (u32)L - 1u) / (u32)L;
    const bool live = t < T;
    const u32 start = live ? t * (u32)L : 0u;
    const u32 end = live ? ((start + (u32)L < M) ? start + (u32)L : M) : 0u;
    const u32 prev = (live && start > 0) ? keys[start - 1] : NOKEY;
    const u32 next = (live && end < M) ? keys[end] : NOKEY;
    Acc acc = Acc::inf();
    u32 cur = live ? keys[start] : NOKEY;
    const u32 first_key = cur;
    u32 last_key = cur;
    bool first = true, head_written = false;
    const int nphase = (L + ACC_SUB - 1) / ACC_SUB;
    const u32 rows = (T - row0 < 128u) ? T - row0 : 128u;
    for (int ph = 0; ph < nphase; ++ph) {
        if (ph) __syncthreads();
        acc_stage(keys, vals, M, L, row0, rows, ph, sk, sv);
        __syncthreads();
        if (!live) continue;
        const u32 j0 = start + (u32)ph * ACC_SUB;
        const u32 j1 = (j0 + ACC_SUB < end) ? j0 + ACC_SUB : end;
        for (u32 j = j0; j < j1; ++j) {
            const u32 k = sk[lr * ACC_PITCH + (j - j0)];
            const u32 v = sv[lr * ACC_PITCH + (j - j0)];
            if (k != cur) {
                const bool head = first && cur == prev;
                lp_raw_store(head ? praw + 2 * (size_t)t * (2 * RAW29_WORDS) : braw + (size_t)cur * (2 * RAW29_WORDS), acc, par);
                head_written |= head;
                first = false;
                cur = k;
                acc = Acc::inf();
            }
            last_key = k;
            const Fp* pp = (const Fp*)(pts + (v >> 1));
            Fp px = pp[par], py = pp[2 + par];
            u32 nz = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) nz |= px.v[i] | py.v[i];
            nz |= lp_swap(nz);
            if (nz) {
                Fp2L29 x29 = {Fp29::from32<5>(px)};
                Fp2L29 y29 = {Fp29::cneg(Fp29::from32<5>(py), (v & 1u) != 0)};
                const bool neg = (v & 1u) != 0;
                xyzz29_madd<Fp2L29>(acc, x29, y29, [=]() { return dbl_point_g2(pp, par, neg); });
            }
        }
    }
    if (!live) return;
    const bool acc_head = first && cur == prev;
    const bool acc_tail = !acc_head && cur == next;
    if (!acc_head && !acc_tail) lp_raw_store(braw + (size_t)cur * (2 * RAW29_WORDS), acc, par);
    if (T > 1) {
        if (acc_head) lp_raw_store(praw + 2 * (size_t)t * (2 * RAW29_WORDS), acc, par);
        if (acc_tail) lp_raw_store(praw + (2 * (size_t)t + 1) * (2 * RAW29_WORDS), acc, par);
        if (par == 0) {
            out_keys[2 * t] = first_key | ((acc_head || head_written) ? 0u : PART_EMPTY);
            out_keys[2 * t + 1] = last_key | (acc_tail ? 0u : PART_EMPTY);
        }
    }
}

__device__ __forceinline__ XYZZ<Fp2L> lp_load(const XYZZ<Fp2>* src, u32 par) {
    const Fp* d = (const Fp*)src;
    XYZZ<Fp2L> a;
    a.x.c = d[par]; a.y.c = d[2 + par]; a.zz.c = d[4 + par]; a.zzz.c = d[6 + par];
    return a;
}

// k_acc_levelN on lane pairs (see msm_kernels.cuh for the algorithm)
__global__ __launch_bounds__(256) void k_acc_levelN_g2pair(const u32* __restrict__ keys, const XYZZ<Fp2>* __restrict__ src,
                                                           u32 M, int L, XYZZ<Fp2>* __restrict__ buckets,
                                                           u32* __restrict__ out_keys, XYZZ<Fp2>* __restrict__ out_part) {
    const u32 gt = blockIdx.x * 256u + threadIdx.x;
    const u32 t = gt >> 1, par = gt & 1u;
    const u32 T = (M + (u32)L - 1u) / (u32)L;
    if (t >= T) return;
    const u32 start = t * (u32)L;
    const u32 end = (start + (u32)L < M) ? start + (u32)L : M;
    const u32 prev = start > 0 ? keys[start - 1] : NOKEY;
    const u32 next = end < M ? keys[end] : NOKEY;
    XYZZ<Fp2L> acc = XYZZ<Fp2L>::inf();
    u32 cur = keys[start];
    bool first = true, head_written = false, tail_written = false;
    for (u32 j = start; j < end; ++j) {
        const u32 k = keys[j];
        if (k != cur) {
            if (first && cur == prev) { lp_store(out_part + 2 * t, acc, par); head_written = true; }
            else if (!acc.is_inf()) lp_store(buckets + cur, acc, par);
            first = false;
            cur = k;
            acc = XYZZ<Fp2L>::inf();
        }
        XYZZ<Fp2L> p = lp_load(src + j, par);
        xyzz_add<Fp2L>(acc, p);
    }
    if (first && cur == prev) { lp_store(out_part + 2 * t, acc, par); head_written = true; }
    else if (cur == next) { lp_store(out_part + 2 * t + 1, acc, par); tail_written = true; }
    else if (!acc.is_inf()) lp_store(buckets + cur, acc, par);
    if (T > 1) {
        XYZZ<Fp2L> z = XYZZ<Fp2L>::inf();
        if (!head_written) lp_store(out_part + 2 * t, z, par);
        if (!tail_written) lp_store(out_part + 2 * t + 1, z, par);
        if (par == 0) {
            out_keys[2 * t] = keys[start];
            out_keys[2 * t + 1] = keys[end - 1];
        }
    }
}

// k_reduce_level on lane pairs
template <bool HAS_Y>
__global__ __launch_bounds__(128) void k_reduce_level_g2pair(const XYZZ<Fp2>* __restrict__ Sin, const XYZZ<Fp2>* __restrict__ Yin,
                                                             u32 n_groups, u32 g, int dbl, XYZZ<Fp2>* __restrict__ Sout,
                                                             XYZZ<Fp2>* __restrict__ Yout) {
    const u32 gt = blockIdx.x * 128u + threadIdx.x;
    const u32 j = gt >> 1, par = gt & 1u;
    if (j >= n_groups) return;
    XYZZ<Fp2L> run = XYZZ<Fp2L>::inf(), wacc = XYZZ<Fp2L>::inf(), ysum = XYZZ<Fp2L>::inf();
    const size_t base = (size_t)j * g;
    for (u32 k = g; k-- > 0;) {
        XYZZ<Fp2L> x = lp_load(Sin + base + k, par);
        xyzz_add<Fp2L>(run, x);
        xyzz_add<Fp2L>(wacc, run);
        if (HAS_Y) {
            XYZZ<Fp2L> y = lp_load(Yin + base + k, par);
            xyzz_add<Fp2L>(ysum, y);
        }
    }
    lp_store(Sout + j, run, par);
    if (HAS_Y) {
        for (int d = 0; d < dbl; ++d) wacc = xyzz_dbl<Fp2L>(wacc);
        xyzz_add<Fp2L>(ysum, wacc);
        lp_store(Yout + j, ysum, par);
    } else {
        lp_store(Yout + j, wacc, par);
    }
}

int32_t launch_levelN(zkpor_ctx* ctx, const u32* keys, const XYZZ<Fp2>* src, u32 M, int L, XYZZ<Fp2>* buckets,
                      u32* out_keys, XYZZ<Fp2>* out_part) {
    u32 T = (M + (u32)L - 1u) / (u32)L;
    hipLaunchKernelGGL(k_acc_levelN_g2pair, dim3((2u * T + 255u) / 256u), dim3(256), 0, ctx->stream, keys, src, M, L, buckets, out_keys, out_part);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
int32_t launch_reduce(zkpor_ctx* ctx, const XYZZ<Fp2>* Sin, const XYZZ<Fp2>* Yin, u32 n_groups, u32 g, int dbl,
                      XYZZ<Fp2>* Sout, XYZZ<Fp2>* Yout) {
    dim3 grid((2u * n_groups + 127u) / 128u);
    if (Yin) hipLaunchKernelGGL(k_reduce_level_g2pair<true>, grid, dim3(128), 0, ctx->stream, Sin, Yin, n_groups, g, dbl, Sout, Yout);
    else hipLaunchKernelGGL(k_reduce_level_g2pair<false>, grid, dim3(128), 0, ctx->stream, Sin, Yin, n_groups, g, dbl, Sout, Yout);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}

int32_t launch_level1(zkpor_ctx* ctx, const u32* keys, const u32* vals, const Affine<Fp2>* pts, u32 M, int L, u32 NB,
                      XYZZ<Fp2>* buckets, u32* out_keys, XYZZ<Fp2>* out_part) {
    u32 T = (M + (u32)L - 1u) / (u32)L;
    ZK_HIP(ctx, hipMemsetAsync(buckets, 0, (size_t)NB * sizeof(XYZZ<Fp2>), ctx->stream));
    {
        PhaseScope ps(ctx, "k_acc_level1_g2");
        hipLaunchKernelGGL(k_acc_level1_g2pair, dim3((T + 127u) / 128u), dim3(256), 0, ctx->stream, keys, vals, pts, M, L, buckets, out_keys, out_part);
    }
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
// ---- the 29-bit pipeline ----
int32_t launch_level1_29(zkpor_ctx* ctx, const u32* keys, const u32* vals, const Affine<Fp2>* pts, u32 M, int L, u32 NB,
                         u32* braw, u32* out_keys, u32* praw) {
    u32 T = (M + (u32)L - 1u) / (u32)L;
    ZK_HIP(ctx, hipMemsetAsync(braw, 0, (size_t)NB * (2 * RAW29_WORDS) * 4, ctx->stream));
    {
        PhaseScope ps(ctx, "k_acc_level1_g2");
        hipLaunchKernelGGL(k_acc_level1_g2pair29, dim3((T + 127u) / 128u), dim3(256), 0, ctx->stream, keys, vals, pts, M, L, braw, out_keys, praw);
    }
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
template <>
int32_t launch_levelN29<Fp2>(zkpor_ctx* ctx, const u32* keys, const u32* src, u32 M, int L, u32* braw, u32* out_keys, u32* out_part) {
    u32 T = (M + (u32)L - 1u) / (u32)L;
    hipLaunchKernelGGL(k_acc_levelN29<Pol29G2>, dim3((2u * T + 255u) / 256u), dim3(256), 0, ctx->stream, keys, src, M, L, braw, out_keys, out_part);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
// Lane-parallel form of a SMALL reduction level, as k_reduce_scan29_g1 (msm_g1_hot.hip) but one bucket per lane PAIR: the g <= 16 pairs of
// a group lie in one wave, the other pair's operand comes through 36 lane shuffles over 2 d lanes, the additions are the lane-pair ones.
// 3 log2(g) additions deep instead of 3 g: the last four levels of the G2 reduction were 0.5 ms each whatever their size.
__device__ __forceinline__ XYZZ29T<Fp2L29> lp_shfl_down29(const XYZZ29T<Fp2L29>& a, u32 lanes) {
    XYZZ29T<Fp2L29> r;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        r.x.c.l[i] = (u32)__shfl_down((int)a.x.c.l[i], lanes, 64); r.y.c.l[i] = (u32)__shfl_down((int)a.y.c.l[i], lanes, 64);
        r.zz.c.l[i] = (u32)__shfl_down((int)a.zz.c.l[i], lanes, 64); r.zzz.c.l[i] = (u32)__shfl_down((int)a.zzz.c.l[i], lanes, 64);
    }
    return r;
}
template <bool HAS_Y>
__global__ __launch_bounds__(256) void k_reduce_scan29_g2(const u32* __restrict__ Sin, const u32* __restrict__ Yin, u32 n_groups, int gl,
                                                           int dbl, u32* __restrict__ Sout, u32* __restrict__ Yout) {
    typedef XYZZ29T<Fp2L29> Acc;
    const u32 g = 1u << gl;                          // <= 16: a group's 2 g lanes never straddle a wave
    const u32 gt = blockIdx.x * 256u + threadIdx.x;
    const u32 e = gt >> 1, par = gt & 1u;            // e = j * g + k: the element's index
    const u32 j = e >> gl, k = e & (g - 1u);
    const bool live = j < n_groups;
    Acc T = live ? Pol29G2::load(Sin, e, par) : Acc::inf();
    Acc W = Acc::inf(), Y = Acc::inf();
    if (HAS_Y && live) Y = Pol29G2::load(Yin, e, par);
    const int phases = HAS_Y ? 3 : 2;
#pragma unroll 1
    for (int it = 0; it < phases * gl; ++it) {
        const int phase = it / gl, st = it - phase * gl;
        const u32 d = 1u << st;
        if (phase == 1 && st == 0) W = T;
        Acc a = phase == 0 ? T : (phase == 1 ? W : Y);
        const Acc b = lp_shfl_down29(a, 2u * d);
        const bool doit = phase == 0 ? (k + d < g) : ((k & (2u * d - 1u)) == 0u);   // the same for both lanes of a pair
        if (doit) xyzz29_add<Fp2L29>(a, b);
        if (phase == 0) T = a; else if (phase == 1) W = a; else Y = a;
    }
    if (!live || k != 0) return;
    Pol29G2::store(Sout, j, T, par);
    if (HAS_Y) {
        for (int q = 0; q < dbl; ++q) W = xyzz29_dbl<Fp2L29>(W);
        xyzz29_add<Fp2L29>(Y, W);
        Pol29G2::store(Yout, j, Y, par);
    } else {
        Pol29G2::store(Yout, j, W, par);
    }
}

template <>
int32_t launch_reduce29<Fp2>(zkpor_ctx* ctx, const u32* Sin, const u32* Yin, u32 n_groups, u32 g, int dbl, u32* Sout, u32* Yout) {
    if (ctx->msm_reduce_scan == 1 && g <= 16u && (g & (g - 1u)) == 0u && g >= 2u && (size_t)n_groups * g <= ((size_t)1 << 15)) {
        int gl = 0;
        while ((1u << gl) < g) ++gl;
        dim3 grid_s((unsigned)((2u * (size_t)n_groups * g + 255u) / 256u));
        if (Yin) hipLaunchKernelGGL((k_reduce_scan29_g2<true>), grid_s, dim3(256), 0, ctx->stream, Sin, Yin, n_groups, gl, dbl, Sout, Yout);
        else hipLaunchKernelGGL((k_reduce_scan29_g2<false>), grid_s, dim3(256), 0, ctx->stream, Sin, Yin, n_groups, gl, dbl, Sout, Yout);
        ZK_KERNEL_CHECK(ctx);
        return ZKPOR_OK;
    }
    dim3 grid((2u * n_groups + 127u) / 128u);
    if (Yin) hipLaunchKernelGGL((k_reduce_level29<Pol29G2, true>), grid, dim3(128), 0, ctx->stream, Sin, Yin, n_groups, g, dbl, Sout, Yout);
    else hipLaunchKernelGGL((k_reduce_level29<Pol29G2, false>), grid, dim3(128), 0, ctx->stream, Sin, Yin, n_groups, g, dbl, Sout, Yout);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
}  // namespace zk
