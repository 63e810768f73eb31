// MSM step 3, G1, level 1 — the dominant kernel of the prover: field products fully inlined.
#include "msm_kernels.cuh"
namespace zk {
int32_t launch_level1(zkpor_ctx* ctx, const u32* keys, const u32* vals, const Affine<Fp>* pts, u32 M, int L,
                      XYZZ<Fp>* buckets, u32* out_keys, XYZZ<Fp>* out_part) {
    u32 T = (M + (u32)L - 1u) / (u32)L;
    PhaseScope ps(ctx, "k_acc_level1_g1");
    hipLaunchKernelGGL(k_acc_level1<Fp>, dim3((T + 255u) / 256u), dim3(256), 0, ctx->stream, keys, vals, pts, M, L, buckets, out_keys, out_part);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
}  // namespace zk
