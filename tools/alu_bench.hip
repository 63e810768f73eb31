// Micro-benchmarks that decide the big-integer strategy on gfx950: raw VALU issue rates of the candidate
// multiply instructions and the throughput of the Montgomery product / curve addition built from them.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkmerkle-proof-of-solvency_amd/csrc tools/alu_bench.hip -o gpurun_out/alu_bench
#include "ec.cuh"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
using namespace zk;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void k_raw(u32* out, int iters, u32 seed) {
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3;
    u32 b = blockIdx.x * 11 + 5;
    u64 c0 = a0, c1 = a1, c2 = a2, c3 = a3, c4 = a0 + 9, c5 = a1 + 9, c6 = a2 + 9, c7 = a3 + 9;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a0 * 0.5, d5 = a1 * 0.5, d6 = a2 * 0.5, d7 = a3 * 0.5, db = 1.0000001;
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {  // v_mad_u64_u32, 8 independent chains
            asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                         "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a0), "v"(b) : "vcc");
        } else if (OP == 1) {  // v_mul_lo_u32
            asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                         "v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (OP == 2) {  // v_mul_hi_u32
            asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                         "v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (OP == 3) {  // v_add_co_u32 + v_addc_co_u32 pairs (4 pairs = 8 instr)
            asm volatile("v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                         "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %4, vcc\n v_add_co_u32 %2, vcc, %2, %4\n v_addc_co_u32 %3, vcc, %3, %4, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b) : "vcc");
        } else if (OP == 4) {  // v_fma_f64, 8 independent chains
            asm volatile("v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %1, %1, %8, %1\n v_fma_f64 %2, %2, %8, %2\n v_fma_f64 %3, %3, %8, %3\n"
                         "v_fma_f64 %4, %4, %8, %4\n v_fma_f64 %5, %5, %8, %5\n v_fma_f64 %6, %6, %8, %6\n v_fma_f64 %7, %7, %8, %7\n"
                         : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db));
        } else if (OP == 5) {  // v_mad_u32_u24
            asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0\n"
                         "v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (OP == 6) {  // v_mul_hi_u32_u24
            asm volatile("v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4\n"
                         "v_mul_hi_u32_u24 %0, %0, %4\n v_mul_hi_u32_u24 %1, %1, %4\n v_mul_hi_u32_u24 %2, %2, %4\n v_mul_hi_u32_u24 %3, %3, %4\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(b));
        } else if (OP == 7) {  // mad + addc pair as used by the FIPS multiplier (dependent chain on one accumulator)
            asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                         "v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_addc_co_u32 %1, vcc, 0, %1, vcc\n"
                         : "+v"(c0), "+v"(a1) : "v"(a0), "v"(b) : "vcc");
        } else if (OP == 8) {  // v_lshl_add_u64 (64-bit add), 8 chains
            asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                         "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(c0 | 1));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ (u32)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7) ^ (u32)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
}

// CIOS written in plain C (what the compiler makes of it) for comparison with the inline-asm FIPS in fe.cuh
__device__ __forceinline__ Fp mul_cios(const Fp& a, const Fp& b) {
    u32 t[10] = {0};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) { c += (u64)a.v[j] * b.v[i] + t[j]; t[j] = (u32)c; c >>= 32; }
        c += t[8]; t[8] = (u32)c; t[9] = (u32)(c >> 32);
        u32 m = t[0] * FpParams::INV;
        c = (u64)m * FpParams::mod(0) + t[0]; c >>= 32;
#pragma unroll
        for (int j = 1; j < 8; ++j) { c += (u64)m * FpParams::mod(j) + t[j]; t[j - 1] = (u32)c; c >>= 32; }
        c += t[8]; t[7] = (u32)c; t[8] = t[9] + (u32)(c >> 32);
    }
    return Fp::reduce_once(t, t[8]);
}

template <int V>
__global__ __launch_bounds__(256) void k_mul(Fp* x, int iters) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fp a = x[2 * i], b = x[2 * i + 1];
    for (int k = 0; k < iters; ++k) {
        if (V == 0) { a = Fp::mul(a, b); b = Fp::mul(b, a); }
        else if (V == 1) { a = mul_cios(a, b); b = mul_cios(b, a); }
        else { a = Fp::mul_call(a, b); b = Fp::mul_call(b, a); }
    }
    x[2 * i] = a; x[2 * i + 1] = b;
}
// two independent product chains per thread (ILP 2)
__global__ __launch_bounds__(256) void k_mul_ilp2(Fp* x, int iters) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fp a = x[2 * i], b = x[2 * i + 1], c = Fp::add(a, b), d = Fp::sub(a, b);
    for (int k = 0; k < iters; ++k) { a = Fp::mul(a, b); c = Fp::mul(c, d); b = Fp::mul(b, a); d = Fp::mul(d, c); }
    x[2 * i] = Fp::add(a, c); x[2 * i + 1] = Fp::add(b, d);
}
__global__ __launch_bounds__(256) void k_madd(const G1Affine* p, G1XYZZ* out, int iters) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    G1XYZZ acc = out[i];
    G1Affine q = p[i];
    for (int k = 0; k < iters; ++k) { xyzz_madd<Fp>(acc, q.x, q.y); q.x = Fp::add(q.x, acc.zz); }
    out[i] = acc;
}
__global__ __launch_bounds__(256) void k_addsub(Fp* x, int iters) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fp a = x[2 * i], b = x[2 * i + 1];
    for (int k = 0; k < iters; ++k) { a = Fp::add(a, b); b = Fp::sub(b, a); }
    x[2 * i] = a; x[2 * i + 1] = b;
}

template <class Fn>
float time_it(Fn fn, int reps = 3) {
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    fn();
    (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(a);
        fn();
        (void)hipEventRecord(b);
        (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device %s CUs %d clock %d kHz\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate);
    const int blocks = 256 * 8, threads = 256;
    const double nthreads = (double)blocks * threads;
    u32* out; CHECK(hipMalloc(&out, blocks * threads * 4));
    const char* names[] = {"v_mad_u64_u32 (8 chains)", "v_mul_lo_u32", "v_mul_hi_u32", "v_add_co/addc", "v_fma_f64", "v_mad_u32_u24", "v_mul_hi_u32_u24", "mad64+addc dependent", "v_lshl_add_u64"};
    const int iters = 4096;
    auto run = [&](int op) {
        switch (op) {
            case 0: hipLaunchKernelGGL(k_raw<0>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u); break;
            case 1: hipLaunchKernelGGL(k_raw<1>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u); break;
            case 2: hipLaunchKernelGGL(k_raw<2>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u); break;
            case 3: hipLaunchKernelGGL(k_raw<3>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u); break;
            case 4: hipLaunchKernelGGL(k_raw<4>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u); break;
            case 5: hipLaunchKernelGGL(k_raw<5>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u); break;
            case 6: hipLaunchKernelGGL(k_raw<6>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u); break;
            case 7: hipLaunchKernelGGL(k_raw<7>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u); break;
            case 8: hipLaunchKernelGGL(k_raw<8>, dim3(blocks), dim3(threads), 0, 0, out, iters, 1u); break;
        }
    };
    for (int op = 0; op < 9; ++op) {
        float ms = time_it([&] { run(op); });
        double instr = nthreads * iters * 8.0;
        double per_s = instr / (ms * 1e-3);
        // cycles per wave-instruction per SIMD at 2.4 GHz: (256 CU * 4 SIMD * 2.4e9) / (wave-instr/s)
        double cyc = 256.0 * 4 * 2.4e9 / (per_s / 64.0);
        printf("RAW %-28s %8.3f ms  %.3e lane-ops/s  ~%.2f cyc/wave-instr/SIMD@2.4GHz\n", names[op], ms, per_s, cyc);
    }
    // field products
    size_t n = (size_t)blocks * threads;
    std::vector<Fp> h(2 * n);
    for (size_t i = 0; i < 2 * n; ++i) for (int j = 0; j < 8; ++j) h[i].v[j] = (u32)(i * 2654435761u + j * 40503u + 12345u) & (j == 7 ? 0x1fffffffu : 0xffffffffu);
    Fp* dx; CHECK(hipMalloc(&dx, 2 * n * sizeof(Fp)));
    CHECK(hipMemcpy(dx, h.data(), 2 * n * sizeof(Fp), hipMemcpyHostToDevice));
    const int mi = 512;
    float ms;
    ms = time_it([&] { hipLaunchKernelGGL(k_mul<0>, dim3(blocks), dim3(threads), 0, 0, dx, mi); });
    printf("FP  mul FIPS asm (inline)      %8.3f ms  %.3e modmul/s\n", ms, n * 2.0 * mi / (ms * 1e-3));
    ms = time_it([&] { hipLaunchKernelGGL(k_mul<1>, dim3(blocks), dim3(threads), 0, 0, dx, mi); });
    printf("FP  mul CIOS compiler          %8.3f ms  %.3e modmul/s\n", ms, n * 2.0 * mi / (ms * 1e-3));
    ms = time_it([&] { hipLaunchKernelGGL(k_mul<2>, dim3(blocks), dim3(threads), 0, 0, dx, mi); });
    printf("FP  mul FIPS asm (call)        %8.3f ms  %.3e modmul/s\n", ms, n * 2.0 * mi / (ms * 1e-3));
    ms = time_it([&] { hipLaunchKernelGGL(k_mul_ilp2, dim3(blocks), dim3(threads), 0, 0, dx, mi); });
    printf("FP  mul FIPS asm ILP2          %8.3f ms  %.3e modmul/s\n", ms, n * 4.0 * mi / (ms * 1e-3));
    ms = time_it([&] { hipLaunchKernelGGL(k_addsub, dim3(blocks), dim3(threads), 0, 0, dx, mi * 8); });
    printf("FP  add+sub                    %8.3f ms  %.3e addsub/s\n", ms, n * 2.0 * mi * 8 / (ms * 1e-3));
    // occupancy sweep for the inline multiplier: fewer blocks
    for (int b : {256, 512, 1024, 2048}) {
        ms = time_it([&] { hipLaunchKernelGGL(k_mul<0>, dim3(b), dim3(threads), 0, 0, dx, mi); });
        printf("FP  mul FIPS blocks=%-5d       %8.3f ms  %.3e modmul/s\n", b, ms, (double)b * threads * 2.0 * mi / (ms * 1e-3));
    }
    // curve addition (a smaller grid than the product tests: the r01 run of this tail, 2048 blocks over half-initialised
    // accumulators, ended in a GPU memory fault after printing everything above; every byte both kernels touch is initialised now)
    const int mblocks = 512;
    const size_t mn = (size_t)mblocks * threads;
    G1XYZZ* dacc; G1Affine* dp;
    CHECK(hipMalloc(&dacc, mn * sizeof(G1XYZZ))); CHECK(hipMalloc(&dp, mn * sizeof(G1Affine)));
    CHECK(hipMemset(dacc, 0, mn * sizeof(G1XYZZ)));                       // every accumulator starts at infinity
    std::vector<G1Affine> gp(mn);
    for (size_t i = 0; i < mn; ++i) { gp[i].x = Fp::from_u32(1); gp[i].y = Fp::from_u32(2); }   // the generator, Montgomery form
    CHECK(hipMemcpy(dp, gp.data(), mn * sizeof(G1Affine), hipMemcpyHostToDevice));
    ms = time_it([&] { hipLaunchKernelGGL(k_madd, dim3(mblocks), dim3(threads), 0, 0, dp, dacc, 64); });
    CHECK(hipDeviceSynchronize());
    printf("G1  xyzz_madd                  %8.3f ms  %.3e madd/s\n", ms, mn * 64.0 / (ms * 1e-3));
    return 0;
}
