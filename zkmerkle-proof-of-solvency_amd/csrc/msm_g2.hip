// MSM step 3 (levels >= 2) and step 4 for G2 (Fp2 coordinates in one lane), built with out-of-line Fp products
// (-DZK_MUL_NOINLINE).  The hot level-1 kernel lives in msm_g2_pair.hip.
#include "msm_kernels.cuh"
namespace zk {
int32_t launch_levelN(zkpor_ctx* ctx, const u32* keys, const XYZZ<Fp2>* src, u32 M, int L, XYZZ<Fp2>* buckets,
                      u32* out_keys, XYZZ<Fp2>* out_part) {
    u32 T = (M + (u32)L - 1u) / (u32)L;
    hipLaunchKernelGGL(k_acc_levelN<Fp2>, dim3((T + 255u) / 256u), dim3(256), 0, ctx->stream, keys, src, M, L, buckets, out_keys, out_part);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
int32_t launch_reduce(zkpor_ctx* ctx, const XYZZ<Fp2>* Sin, const XYZZ<Fp2>* Yin, u32 n_groups, u32 g, int dbl,
                      XYZZ<Fp2>* Sout, XYZZ<Fp2>* Yout) {
    if (Yin) hipLaunchKernelGGL((k_reduce_level<Fp2, true>), dim3((n_groups + 63u) / 64u), dim3(64), 0, ctx->stream, Sin, Yin, n_groups, g, dbl, Sout, Yout);
    else hipLaunchKernelGGL((k_reduce_level<Fp2, false>), dim3((n_groups + 63u) / 64u), dim3(64), 0, ctx->stream, Sin, Yin, n_groups, g, dbl, Sout, Yout);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
}  // namespace zk
