// MSM step 1: scalar -> signed-digit (bucket key, point index) stream.  See msm.cuh.
#include "msm_kernels.cuh"
namespace zk {

// ------------------------------------------------------------------------------------------------ decompose
// low c bits of s, then s >>= c (static register indexing only: no scratch)
ZK_D u32 take_digit(Fr& s, int c) {
    u32 d = s.v[0] & ((1u << c) - 1u);
#pragma unroll
    for (int i = 0; i < 7; ++i) s.v[i] = (s.v[i] >> c) | (s.v[i + 1] << (32 - c));
    s.v[7] >>= c;
    return d;
}

// One thread per scalar.  Two passes over the digits (count, then write) so nothing spills; entries of one
// 256-thread block are appended with ONE global atomic.
__global__ __launch_bounds__(256) void k_decompose(const Fr* __restrict__ scalars, u32 n, int c, int W, int tables, int piece, u32 bpw,
                                                   u32* __restrict__ out_keys, u32* __restrict__ out_vals,
                                                   u32* __restrict__ counter) {
    __shared__ u32 wave_tot[4];
    __shared__ u32 block_base;
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    Fr s = Fr::zero();
    u32 cnt = 0;
    const u32 half = 1u << (c - 1);
    if (i < n) {
        s = Fr::from_mont(scalars[i]);
        Fr t = s;
        u32 carry = 0;
        for (int w = 0; w < W; ++w) {
            u32 d = take_digit(t, c) + carry;
            carry = d > half ? 1u : 0u;
            d = carry ? (1u << c) - d : d;
            cnt += d != 0;
        }
    }
    // block-exclusive prefix of cnt
    u32 x = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        u32 y = __shfl_up(x, off);
        if ((int)lane >= off) x += y;
    }
    if (lane == 63) wave_tot[wv] = x;
    __syncthreads();
    if (threadIdx.x == 0) {
        u32 t = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
        block_base = t ? atomicAdd(counter, t) : 0u;
    }
    __syncthreads();
    u32 pos = block_base + x - cnt;
    for (u32 k = 0; k < wv; ++k) pos += wave_tot[k];
    if (i < n && cnt) {
        u32 carry = 0;
        for (int w = 0; w < W; ++w) {
            u32 d = take_digit(s, c) + carry;
            carry = d > half ? 1u : 0u;
            d = carry ? (1u << c) - d : d;
            if (d) {
                // digit w of the scalar = bucket window w % piece against table w / piece of point i (msm.cuh MsmCfg)
                const u32 q = (u32)w / (u32)piece;
                out_keys[pos] = ((u32)w - q * (u32)piece) * bpw + (d - 1u);
                out_vals[pos] = ((i * (u32)tables + q) << 1) | carry;  // carry == 1 <=> the digit is negative
                ++pos;
            }
        }
    }
}

int32_t launch_decompose(zkpor_ctx* ctx, const Fr* d_scalars, u32 n, const MsmCfg& cfg, u32* keys, u32* vals, u32* counter) {
    u32 blocks = (n + 255u) / 256u;
    hipLaunchKernelGGL(k_decompose, dim3(blocks), dim3(256), 0, ctx->stream, d_scalars, n, cfg.c, cfg.W, cfg.m, cfg.piece, cfg.bpw, keys, vals, counter);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
}  // namespace zk
