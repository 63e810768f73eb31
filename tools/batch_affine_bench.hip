// Batch-affine bucket accumulation against the XYZZ mixed addition, measured ("give batch-affine the fe52 treatment").
//
// An affine addition with a SHARED inversion costs 6 field products (3 for Montgomery's trick, lambda, lambda^2, lambda * dx) against 10 for the
// XYZZ mixed addition the level-1 kernels use (fe29.cuh xyzz29_madd).  The inversion must be amortised: here, as proposed, every thread keeps K
// independent affine accumulators in registers, multiplies its K denominators (thread-local prefix products), the 256 threads of a workgroup
// combine their products through LDS (log-step prefix and suffix product scans), ONE thread inverts the workgroup's product by Fermat while the
// others wait, and every thread back-substitutes.  Everything is favourable to batch-affine: no gathers (points come from registers / a tiny
// table), no bucket boundaries, no doubling / infinity cases, full occupancy.  Both kernels use the production field arithmetic (Fp29).
// Result -> profiles/r03_batch_affine.txt.  Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkmerkle-proof-of-solvency_amd/csrc -I include
//        tools/batch_affine_bench.hip -o tools/bin/batch_affine_bench
#include "common.cuh"
#include "fe29.cuh"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
using namespace zk;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ Fp29 ld(const u32* p) { Fp29 r; for (int i = 0; i < 9; ++i) r.l[i] = p[i]; return r; }
__device__ __forceinline__ void st(u32* p, const Fp29& v) { for (int i = 0; i < 9; ++i) p[i] = v.l[i]; }

// a^(p-2) with the production product (254 squarings + the multiplications of the exponent's set bits)
__device__ __noinline__ Fp29 inv29(const Fp29& a) {
    Fp29 r = Fp29::one(), b = a;
    // p - 2, little-endian 32-bit words
    const u32 e[8] = {0xd87cfd45u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
    for (int i = 0; i < 254; ++i) {
        if ((e[i >> 5] >> (i & 31)) & 1u) r = Fp29::mul(r, b);
        b = Fp29::sqr(b);
    }
    return r;
}

// baseline: `rounds` x K mixed additions per thread into K XYZZ accumulators (K only to mirror the other kernel's work per round)
template <int K>
__global__ __launch_bounds__(256) void k_xyzz(const u32* __restrict__ pts, u32* __restrict__ out, int rounds) {
    XYZZ29T<Fp29> acc[K];
    Fp29 px = ld(pts + 18 * (threadIdx.x & 63)), py = ld(pts + 18 * (threadIdx.x & 63) + 9);
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] = XYZZ29T<Fp29>{px, py, Fp29::one(), Fp29::one()};
    for (int r = 0; r < rounds; ++r) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
            xyzz29_madd<Fp29>(acc[j], px, py);
            px = Fp29::reduce32(Fp29::add_l(px, acc[j].zz));   // a fresh "point" every time (not on the curve: the cost is what is measured)
        }
    }
    Fp29 s = acc[0].x;
#pragma unroll
    for (int j = 1; j < K; ++j) s = Fp29::reduce32(Fp29::add_l(s, acc[j].x));
    st(out + 9 * (blockIdx.x * 256u + threadIdx.x), s);
}

// batch-affine: K accumulators per thread, one inversion per workgroup and round
template <int K>
__global__ __launch_bounds__(256) void k_batch_affine(const u32* __restrict__ pts, u32* __restrict__ out, int rounds, u32* __restrict__ bad) {
    __shared__ u32 pre[256 * 9], suf[256 * 9], total_inv[9];
    Fp29 ax[K], ay[K];
    Fp29 px = ld(pts + 18 * (threadIdx.x & 63)), py = ld(pts + 18 * (threadIdx.x & 63) + 9);
#pragma unroll
    for (int j = 0; j < K; ++j) { ax[j] = Fp29::reduce32(Fp29::add_l(px, Fp29::one())); ay[j] = py; }
    const u32 t = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        // 1. denominators and thread-local prefix products
        Fp29 dx[K], lp[K];
        Fp29 run = Fp29::one();
#pragma unroll
        for (int j = 0; j < K; ++j) {
            dx[j] = Fp29::normed(Fp29::sub_l(px, ax[j]));
            lp[j] = run;
            run = Fp29::mul(run, dx[j]);
        }
        // 2. workgroup: exclusive prefix and suffix products of the 256 thread products (log-step scans through LDS)
        st(pre + 9 * t, run); st(suf + 9 * t, run);
        __syncthreads();
        for (u32 d = 1; d < 256; d <<= 1) {
            Fp29 a = ld(pre + 9 * t), b = ld(suf + 9 * t);
            const bool hp = t >= d, hs = t + d < 256;
            Fp29 ap = hp ? ld(pre + 9 * (t - d)) : Fp29::one(), bs = hs ? ld(suf + 9 * (t + d)) : Fp29::one();
            __syncthreads();
            if (hp) st(pre + 9 * t, Fp29::mul(a, ap));
            if (hs) st(suf + 9 * t, Fp29::mul(b, bs));
            __syncthreads();
        }
        // 3. one inversion for the workgroup
        if (t == 0) st(total_inv, inv29(ld(suf)));   // suf[0] = the product of all 256
        __syncthreads();
        // 4. this thread's inverse = total^-1 * (product of the others) = total^-1 * pre[t-1] * suf[t+1]
        Fp29 inv = ld(total_inv);
        if (t > 0) inv = Fp29::mul(inv, ld(pre + 9 * (t - 1)));
        if (t < 255) inv = Fp29::mul(inv, ld(suf + 9 * (t + 1)));
        __syncthreads();
        // 5. back-substitution and the affine additions
#pragma unroll
        for (int j = K - 1; j >= 0; --j) {
            Fp29 dinv = Fp29::mul(inv, lp[j]);
            if (r == 0 && j == 0 && !Fp29::normed(Fp29::sub_l(Fp29::mul(dinv, dx[j]), Fp29::one())).is_zero_mod_p()) atomicAdd(bad, 1u);   // 1/dx really is the inverse
            inv = Fp29::mul(inv, dx[j]);
            Fp29 lam = Fp29::mul(Fp29::normed(Fp29::sub_l(py, ay[j])), dinv);
            Fp29 x3 = Fp29::normed(Fp29::sub_l(Fp29::sub_l(Fp29::sqr(lam), ax[j]), px));
            Fp29 y3 = Fp29::normed(Fp29::sub_l(Fp29::mul(lam, Fp29::normed(Fp29::sub_l(ax[j], x3))), ay[j]));
            ax[j] = Fp29::reduce32(x3); ay[j] = Fp29::reduce32(y3);
        }
        px = Fp29::reduce32(Fp29::add_l(px, ax[0]));
    }
    Fp29 s = ax[0];
#pragma unroll
    for (int j = 1; j < K; ++j) s = Fp29::reduce32(Fp29::add_l(s, ax[j]));
    st(out + 9 * (blockIdx.x * 256u + threadIdx.x), s);
}

template <class Fn>
static float time_it(Fn fn) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    fn(); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        (void)hipEventRecord(a); fn(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d\n", prop.gcnArchName, prop.multiProcessorCount);
    const int blocks = 256 * 8;
    std::vector<u32> h(64 * 18);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (u32)((i * 2654435761u + 12345u) & 0x0fffffffu);
    u32 *pts, *out, *bad;
    CHECK(hipMalloc(&pts, h.size() * 4)); CHECK(hipMalloc(&out, (size_t)blocks * 256 * 9 * 4)); CHECK(hipMalloc(&bad, 4)); CHECK(hipMemset(bad, 0, 4));
    CHECK(hipMemcpy(pts, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int rounds = 24;
    auto report = [&](const char* name, int K, float ms) {
        double adds = (double)blocks * 256 * K * rounds;
        printf("%-44s K=%d  %8.3f ms  %.3e additions/s\n", name, K, ms, adds / (ms * 1e-3));
    };
    float ms;
    ms = time_it([&] { hipLaunchKernelGGL(k_xyzz<4>, dim3(blocks), dim3(256), 0, 0, pts, out, rounds); });
    report("XYZZ mixed addition (production formula)", 4, ms);
    ms = time_it([&] { hipLaunchKernelGGL(k_batch_affine<2>, dim3(blocks), dim3(256), 0, 0, pts, out, rounds, bad); });
    report("batch-affine, inversion per workgroup round", 2, ms);
    ms = time_it([&] { hipLaunchKernelGGL(k_batch_affine<4>, dim3(blocks), dim3(256), 0, 0, pts, out, rounds, bad); });
    report("batch-affine, inversion per workgroup round", 4, ms);
    ms = time_it([&] { hipLaunchKernelGGL(k_batch_affine<6>, dim3(blocks), dim3(256), 0, 0, pts, out, rounds, bad); });
    report("batch-affine, inversion per workgroup round", 6, ms);
    ms = time_it([&] { hipLaunchKernelGGL(k_batch_affine<8>, dim3(blocks), dim3(256), 0, 0, pts, out, rounds, bad); });
    report("batch-affine, inversion per workgroup round", 8, ms);
    CHECK(hipDeviceSynchronize());
    u32 hb = 1;
    CHECK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
    printf("inverse check (1/dx * dx == 1 for the first accumulator of every thread, first round of every launch): %u failures\n", hb);
    return 0;
}
