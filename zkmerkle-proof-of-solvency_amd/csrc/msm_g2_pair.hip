// MSM step 3, G2, level 1: the single most expensive launch of a proof.  One G2 accumulator per LANE PAIR
// (fp2_lanepair.cuh): even lanes carry the real, odd lanes the imaginary Fp component.
#include "msm_kernels.cuh"
#include "fp2_lanepair.cuh"

namespace zk {

__device__ __forceinline__ void lp_store(XYZZ<Fp2>* dst, const XYZZ<Fp2L>& a, u32 par) {
    Fp* d = (Fp*)dst;
    d[par] = a.x.c; d[2 + par] = a.y.c; d[4 + par] = a.zz.c; d[6 + par] = a.zzz.c;
}

__global__ __launch_bounds__(256) void k_acc_level1_g2pair(const u32* __restrict__ keys, const u32* __restrict__ vals,
                                                           const Affine<Fp2>* __restrict__ pts, u32 M, int L,
                                                           XYZZ<Fp2>* __restrict__ buckets, u32* __restrict__ out_keys,
                                                           XYZZ<Fp2>* __restrict__ out_part) {
    const u32 gt = blockIdx.x * 256u + threadIdx.x;
    const u32 t = gt >> 1, par = gt & 1u;
    const u32 T = (M + (u32)L - 1u) / (u32)L;
    if (t >= T) return;  // both lanes of a pair leave together
    const u32 start = t * (u32)L;
    const u32 end = (start + (u32)L < M) ? start + (u32)L : M;
    const u32 prev = start > 0 ? keys[start - 1] : NOKEY;
    const u32 next = end < M ? keys[end] : NOKEY;
    XYZZ<Fp2L> acc = XYZZ<Fp2L>::inf();
    u32 cur = keys[start];
    bool first = true, head_written = false, tail_written = false;
    for (u32 j = start; j < end; ++j) {
        const u32 k = keys[j];
        const u32 v = vals[j];
        if (k != cur) {
            if (first && cur == prev) { lp_store(out_part + 2 * t, acc, par); head_written = true; }
            else lp_store(buckets + cur, acc, par);
            first = false;
            cur = k;
            acc = XYZZ<Fp2L>::inf();
        }
        const Fp* pp = (const Fp*)(pts + (v >> 1));
        Fp2L px = {pp[par]}, py = {pp[2 + par]};
        if (!(px.is_zero() & py.is_zero())) {
            if (v & 1u) py = Fp2L::neg(py);
            xyzz_madd<Fp2L>(acc, px, py);
        }
    }
    if (first && cur == prev) { lp_store(out_part + 2 * t, acc, par); head_written = true; }
    else if (cur == next) { lp_store(out_part + 2 * t + 1, acc, par); tail_written = true; }
    else lp_store(buckets + cur, acc, par);
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

int32_t launch_level1(zkpor_ctx* ctx, const u32* keys, const u32* vals, const Affine<Fp2>* pts, u32 M, int L,
                      XYZZ<Fp2>* buckets, u32* out_keys, XYZZ<Fp2>* out_part) {
    u32 T = (M + (u32)L - 1u) / (u32)L;
    PhaseScope ps(ctx, "k_acc_level1_g2");
    hipLaunchKernelGGL(k_acc_level1_g2pair, dim3((2u * T + 255u) / 256u), dim3(256), 0, ctx->stream, keys, vals, pts, M, L, buckets, out_keys, out_part);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
}  // namespace zk
