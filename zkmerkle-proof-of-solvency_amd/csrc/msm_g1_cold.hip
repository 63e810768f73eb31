// MSM steps 3 (levels >= 2) and 4 for G1, built with out-of-line field products (-DZK_MUL_NOINLINE): measured
// FASTER than fully inlined bodies here (three inlined general additions overflow the instruction cache).
#include "msm_kernels.cuh"
namespace zk {
int32_t launch_levelN(zkpor_ctx* ctx, const u32* keys, const XYZZ<Fp>* src, u32 M, int L, XYZZ<Fp>* buckets,
                      u32* out_keys, XYZZ<Fp>* out_part) {
    u32 T = (M + (u32)L - 1u) / (u32)L;
    hipLaunchKernelGGL(k_acc_levelN<Fp>, dim3((T + 255u) / 256u), dim3(256), 0, ctx->stream, keys, src, M, L, buckets, out_keys, out_part);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
int32_t launch_reduce(zkpor_ctx* ctx, const XYZZ<Fp>* Sin, const XYZZ<Fp>* Yin, u32 n_groups, u32 g, int dbl,
                      XYZZ<Fp>* Sout, XYZZ<Fp>* Yout) {
    if (Yin) hipLaunchKernelGGL((k_reduce_level<Fp, true>), dim3((n_groups + 63u) / 64u), dim3(64), 0, ctx->stream, Sin, Yin, n_groups, g, dbl, Sout, Yout);
    else hipLaunchKernelGGL((k_reduce_level<Fp, false>), dim3((n_groups + 63u) / 64u), dim3(64), 0, ctx->stream, Sin, Yin, n_groups, g, dbl, Sout, Yout);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
}  // namespace zk
