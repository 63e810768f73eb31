// MSM, beside the digit stream (sort.hip builds it): the per-array stream filter and the "debug_validate" check.  See msm.cuh.
#include "msm_kernels.cuh"
namespace zk {

// ------------------------------------------------------------------------------------------------ per-array streams
// A STABLE filter of the sorted (key, val) stream that drops the entries whose point is absent from a group of key arrays
// (group 0: pk.InfinityB for B1 / B2; group 1: the public and committed wires K leaves out).  The shared stream of w serves four
// arrays; in the level-1 kernel an entry whose point is infinity costs a full addition slot of its wave (the other lanes add, this
// one idles) and a 64-byte gather — 25 % of K's and 10 % of B's lane-time at the production shapes.  The absence flags ride in bits 30 / 31
// of the values (set by sort.hip k_dsort_scatter0), so the filter is pure streaming.  Both groups are produced in ONE
// pass pair over the stream, by a SMALL persistent grid (one or two workgroups per CU): the kernels run beside the VALU-bound
// accumulation of A and must take memory bandwidth, not wave slots (a library compaction with a full-size grid cost the proof 39 ms —
// profiles/r03_filter.txt).  Pass 1 counts what each workgroup's contiguous segment keeps, pass 2 writes: order, hence every run of
// equal keys, is preserved.
constexpr u32 FILT_PER_THREAD = 8, FILT_TILE = 256 * FILT_PER_THREAD;

__global__ __launch_bounds__(256) void k_filter_count(const u32* __restrict__ vals, u32 M, u32 seg, u32* __restrict__ seg_counts) {
    const u32 lo = blockIdx.x * seg, hi = (lo + seg < M) ? lo + seg : M;
    u32 c0 = 0, c1 = 0;
    for (u32 base = lo + threadIdx.x * FILT_PER_THREAD; base < hi; base += FILT_TILE) {
        if (base + FILT_PER_THREAD <= hi) {
            const uint4 a = *(const uint4*)(vals + base), b = *(const uint4*)(vals + base + 4);   // base is a multiple of 8: 32-byte aligned
            const u32 v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (u32 k = 0; k < 8; ++k) { c0 += (v[k] & VAL_ABSENT0) ? 0u : 1u; c1 += (v[k] & VAL_ABSENT1) ? 0u : 1u; }
        } else {
            for (u32 k = 0; base + k < hi; ++k) { const u32 v = vals[base + k]; c0 += (v & VAL_ABSENT0) ? 0u : 1u; c1 += (v & VAL_ABSENT1) ? 0u : 1u; }
        }
    }
    __shared__ u32 w0[4], w1[4];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { c0 += __shfl_down(c0, off); c1 += __shfl_down(c1, off); }
    if ((threadIdx.x & 63u) == 0) { w0[threadIdx.x >> 6] = c0; w1[threadIdx.x >> 6] = c1; }
    __syncthreads();
    if (threadIdx.x == 0) { seg_counts[2 * blockIdx.x] = w0[0] + w0[1] + w0[2] + w0[3]; seg_counts[2 * blockIdx.x + 1] = w1[0] + w1[1] + w1[2] + w1[3]; }
}

__global__ __launch_bounds__(256) void k_filter_write(const u32* __restrict__ keys, const u32* __restrict__ vals, u32 M, u32 seg,
                                                      const u32* __restrict__ seg_counts, u32* __restrict__ k0, u32* __restrict__ v0,
                                                      u32* __restrict__ k1, u32* __restrict__ v1) {
    __shared__ u32 red0[4], red1[4], tile0[4], tile1[4];
    // this segment's offsets in the two outputs: the counts of the segments before it
    u32 o0 = 0, o1 = 0;
    for (u32 b = threadIdx.x; b < blockIdx.x; b += 256u) { o0 += seg_counts[2 * b]; o1 += seg_counts[2 * b + 1]; }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) { o0 += __shfl_down(o0, off); o1 += __shfl_down(o1, off); }
    const u32 lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (lane == 0) { red0[wv] = o0; red1[wv] = o1; }
    __syncthreads();
    u32 out0 = red0[0] + red0[1] + red0[2] + red0[3], out1 = red1[0] + red1[1] + red1[2] + red1[3];
    const u32 lo = blockIdx.x * seg, hi = (lo + seg < M) ? lo + seg : M;
    for (u32 tb = lo; tb < hi; tb += FILT_TILE) {
        const u32 base = tb + threadIdx.x * FILT_PER_THREAD;
        u32 kk[FILT_PER_THREAD], vv[FILT_PER_THREAD];
        u32 f0 = 0, f1 = 0;
        if (base + FILT_PER_THREAD <= hi) {
            const uint4 ka = *(const uint4*)(keys + base), kb = *(const uint4*)(keys + base + 4);
            const uint4 va = *(const uint4*)(vals + base), vb = *(const uint4*)(vals + base + 4);
            kk[0] = ka.x; kk[1] = ka.y; kk[2] = ka.z; kk[3] = ka.w; kk[4] = kb.x; kk[5] = kb.y; kk[6] = kb.z; kk[7] = kb.w;
            vv[0] = va.x; vv[1] = va.y; vv[2] = va.z; vv[3] = va.w; vv[4] = vb.x; vv[5] = vb.y; vv[6] = vb.z; vv[7] = vb.w;
#pragma unroll
            for (u32 k = 0; k < 8; ++k) { if (!(vv[k] & VAL_ABSENT0)) f0 |= 1u << k; if (!(vv[k] & VAL_ABSENT1)) f1 |= 1u << k; }
        } else {
#pragma unroll
            for (u32 k = 0; k < FILT_PER_THREAD; ++k) {
                if (base + k < hi) {
                    kk[k] = keys[base + k]; vv[k] = vals[base + k];
                    if (!(vv[k] & VAL_ABSENT0)) f0 |= 1u << k;
                    if (!(vv[k] & VAL_ABSENT1)) f1 |= 1u << k;
                }
            }
        }
        const u32 c0 = __popc(f0), c1 = __popc(f1);
        u32 x0 = c0, x1 = c1;   // inclusive scan over the wave, then over the four waves
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            u32 y0 = __shfl_up(x0, off), y1 = __shfl_up(x1, off);
            if ((int)lane >= off) { x0 += y0; x1 += y1; }
        }
        __syncthreads();   // tile0 / tile1 of the previous tile have been read
        if (lane == 63) { tile0[wv] = x0; tile1[wv] = x1; }
        __syncthreads();
        u32 p0 = out0 + x0 - c0, p1 = out1 + x1 - c1;
        for (u32 w = 0; w < wv; ++w) { p0 += tile0[w]; p1 += tile1[w]; }
#pragma unroll
        for (u32 k = 0; k < FILT_PER_THREAD; ++k) {
            if (k0 && ((f0 >> k) & 1u)) { k0[p0] = kk[k]; v0[p0] = vv[k]; ++p0; }
            if (k1 && ((f1 >> k) & 1u)) { k1[p1] = kk[k]; v1[p1] = vv[k]; ++p1; }
        }
        out0 += tile0[0] + tile0[1] + tile0[2] + tile0[3];
        out1 += tile1[0] + tile1[1] + tile1[2] + tile1[3];
    }
}

// seg_counts: 2 * grid u32 of scratch.  A group whose output pointers are null is not written.
int32_t launch_filter(zkpor_ctx* ctx, const u32* keys, const u32* vals, u32 M, u32* seg_counts, u32 grid, u32* k0, u32* v0, u32* k1, u32* v1) {
    if (M == 0) return ZKPOR_OK;
    u32 seg = (M + grid - 1) / grid;
    seg = ((seg + FILT_TILE - 1) / FILT_TILE) * FILT_TILE;           // whole tiles per segment
    const u32 blocks = (M + seg - 1) / seg;
    hipLaunchKernelGGL(k_filter_count, dim3(blocks), dim3(256), 0, ctx->stream, vals, M, seg, seg_counts);
    hipLaunchKernelGGL(k_filter_write, dim3(blocks), dim3(256), 0, ctx->stream, keys, vals, M, seg, seg_counts, k0, v0, k1, v1);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}

// ------------------------------------------------------------------------------------------------ "debug_validate"
// what the level-1 kernels take on trust: keys ascending and below the bucket count, point indices inside the array.  bad[0] = violations,
// bad[1] = the first offending position, bad[2] / bad[3] = its key / value.
__global__ __launch_bounds__(256) void k_validate_stream(const u32* __restrict__ keys, const u32* __restrict__ vals, u32 M, u32 NB, u32 n_idx, u32* __restrict__ bad) {
    for (u32 i = blockIdx.x * 256u + threadIdx.x; i < M; i += gridDim.x * 256u) {
        const u32 k = keys[i], v = (vals[i] & VAL_MASK) >> 1;
        if (k >= NB || (n_idx && v >= n_idx) || (i && keys[i - 1] > k)) {
            atomicAdd(bad, 1u);
            if (atomicMin(bad + 1, i) > i) { bad[2] = k; bad[3] = vals[i]; }
        }
    }
}
int32_t validate_stream(zkpor_ctx* ctx, const u32* keys, const u32* vals, u32 M, u32 NB, u32 n_idx, const char* what) {
    if (!M) return ZKPOR_OK;
    if (!ctx->dbg_buf) ZK_HIP(ctx, hipMalloc((void**)&ctx->dbg_buf, 16));
    const u32 init[4] = {0u, 0xffffffffu, 0u, 0u};
    u32 got[4] = {0, 0, 0, 0};
    ZK_HIP(ctx, hipMemcpyAsync(ctx->dbg_buf, init, 16, hipMemcpyHostToDevice, ctx->stream));
    u32 blocks = (M + 255u) / 256u;
    if (blocks > 1024u) blocks = 1024u;
    hipLaunchKernelGGL(k_validate_stream, dim3(blocks), dim3(256), 0, ctx->stream, keys, vals, M, NB, n_idx, ctx->dbg_buf);
    ZK_KERNEL_CHECK(ctx);
    ZK_HIP(ctx, hipMemcpyAsync(got, ctx->dbg_buf, 16, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (got[0]) {
        char msg[256];
        snprintf(msg, sizeof msg, "msm: the sorted digit stream of a %s accumulation is corrupt: %u of %u entries (first at %u: key %u of %u buckets, value 0x%08x, %u point indices)",
                 what, got[0], M, got[1], got[2], NB, got[3], n_idx);
        ctx->err = msg;
        return ZKPOR_E_STATE;
    }
    return ZKPOR_OK;
}

}  // namespace zk
