// fe52: a 254-bit Montgomery product on 5 x 52-bit limbs held in doubles, partial products split into exact high and low halves
// by two v_fma_f64 (after Emmart et al., "Faster Modular Exponentiation Using Double Precision Floating Point
// Arithmetic on the GPU"), column sums accumulated as 64-bit integers.  PROTOTYPE for one question: does a
// DFMA formulation beat the 9 x 29-bit v_mad_i64_i32 product of fe29.cuh (206 VALU instructions, 1.74e11 products/s measured)?
// v_fma_f64 issues at the same rate as v_mad_u64_u32 on gfx950 (profiles/r01_alu_microbench.txt), so the answer is the
// instruction count per product — counted from the ISA of this file (tools/count_isa.py) and confirmed by the measured rate.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I zkmerkle-proof-of-solvency_amd/csrc tools/fe52_bench.hip -o tools/bin/fe52_bench
#include "fe29.cuh"
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
using namespace zk;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

struct Fe52 { double l[5]; };  // value = sum l[i] 2^(52 i), every limb an integer in [0, 2^52); Montgomery radix 2^260

__device__ __forceinline__ long long bits(double x) { return __double_as_longlong(x); }
// BN254 base field modulus p in 52-bit limbs and -p^-1 mod 2^52
__device__ __constant__ double kP52[5] = {0xc16d87cfd47p0 * 1.0, 0, 0, 0, 0};  // filled by the host (see main)
__device__ __constant__ double kNinv52;

// one 52 x 52 -> 104 bit partial product as two exact halves, added into 64-bit column sums — in the DEFAULT rounding mode
// (round to nearest), so no MODE register games:
//   hi = fma(a, b, 2^104)                   = 2^104 + H 2^52,  H = round(ab / 2^52)        (ulp at 2^104 is 2^52)
//   lo = fma(a, b, 2^104 - hi) + 1.5 2^52   = 1.5 2^52 + L,    L = ab - H 2^52 in [-2^51, 2^51]   (both steps exact; the offset
//        cannot be folded into the addend: 2^104 + 1.5 2^52 is not representable)
// The bit patterns are added as integers: the mantissa of hi is H, of lo is 2^51 + L; the exponent words and the 2^51 are removed
// once per column.  q of the reduction is the SIGNED residue L of t n' (|q| <= 2^51), its products with p use 1.5 2^104 so that
// a negative product still lands in [2^104, 2^105).
#define C1 0x1p104
#define C1S 0x1.8p104
template <bool SIGNED>
__device__ __forceinline__ void pp(double a, double b, long long& col_lo, long long& col_hi) {
    double hi = __builtin_fma(a, b, SIGNED ? C1S : C1);
    double lo = __builtin_fma(a, b, (SIGNED ? C1S : C1) - hi) + 0x1.8p52;
    col_lo += bits(lo);
    col_hi += bits(hi);
}

__device__ __forceinline__ Fe52 mul52(const Fe52& a, const Fe52& b) {
    long long t[11];
    const long long bias_lo = bits(0x1.8p52), bias_hi = bits(C1), bias_his = bits(C1S);
#pragma unroll
    for (int k = 0; k < 11; ++k) t[k] = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) pp<false>(a.l[i], b.l[j], t[i + j], t[i + j + 1]);
    // number of lo terms in column k: min(k, 8 - k) + 1 for k <= 8; hi terms: the same for column k - 1 (compile-time constants)
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        int nlo = k <= 8 ? (k < 4 ? k : 8 - k < 4 ? 8 - k : 4) + 1 : 0;
        int nhi = k >= 1 ? ((k - 1) < 4 ? (k - 1) : 8 - (k - 1) < 4 ? 8 - (k - 1) : 4) + 1 : 0;
        t[k] -= nlo * bias_lo + nhi * bias_hi;
    }
    // + p R: the signed q below can leave the quotient one p short; the result stays in (0, 3p)
#pragma unroll
    for (int j = 0; j < 5; ++j) t[5 + j] += (long long)kP52[j];
    // Montgomery reduction, one limb per step: q = signed residue of (t[i] n') mod 2^52, t += q p 2^(52 i)
    const long long M52 = (1LL << 52) - 1;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        double ti = __longlong_as_double((t[i] & M52) | bits(0x1p52)) - 0x1p52;   // low 52 bits of the column as a double
        double hq = __builtin_fma(ti, kNinv52, C1);
        double q = __builtin_fma(ti, kNinv52, C1 - hq);                            // in [-2^51, 2^51], = ti n' mod 2^52
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            long long l = 0, h = 0;
            pp<true>(q, kP52[j], l, h);
            t[i + j] += l - bias_lo;
            t[i + j + 1] += h - bias_his;
        }
        t[i + 1] += t[i] >> 52;   // t[i] is now 0 mod 2^52: pass its (signed) carry on
    }
    // limbs 5..9 hold the result; normalise to 52-bit limbs and back to doubles
    Fe52 r;
#pragma unroll
    for (int k = 5; k < 10; ++k) {
        long long v = t[k];
        if (k < 9) { t[k + 1] += v >> 52; v &= M52; }
        r.l[k - 5] = __longlong_as_double(v | bits(0x1p52)) - 0x1p52;
    }
    return r;
}

__global__ __launch_bounds__(256) void k_mul52(double* x, int iters) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fe52 a, b;
#pragma unroll
    for (int k = 0; k < 5; ++k) { a.l[k] = x[10 * i + k]; b.l[k] = x[10 * i + 5 + k]; }
    for (int k = 0; k < iters; ++k) { a = mul52(a, b); b = mul52(b, a); }
#pragma unroll
    for (int k = 0; k < 5; ++k) { x[10 * i + k] = a.l[k]; x[10 * i + 5 + k] = b.l[k]; }
}

// the incumbent: the 9 x 29-bit product of fe29.cuh in the same harness
__global__ __launch_bounds__(256) void k_mul29(u32* x, int iters) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    Fp29 a, b;
#pragma unroll
    for (int k = 0; k < 9; ++k) { a.l[k] = x[18 * i + k]; b.l[k] = x[18 * i + 9 + k]; }
    for (int k = 0; k < iters; ++k) { a = Fp29::mul(a, b); b = Fp29::mul(b, a); }
#pragma unroll
    for (int k = 0; k < 9; ++k) { x[18 * i + k] = a.l[k]; x[18 * i + 9 + k] = b.l[k]; }
}

// host reference of the same Montgomery product (radix 2^260) on unsigned __int128 arithmetic, to check the prototype's values
typedef unsigned __int128 u128;
struct Big { u64 w[5]; };  // 320 bits
static bool geq(const u64* a, const u64* b, int n) { for (int i = n - 1; i >= 0; --i) { if (a[i] != b[i]) return a[i] > b[i]; } return true; }

int main() {
    setvbuf(stdout, NULL, _IONBF, 0);
    // p in 64-bit words, then in 52-bit limbs
    const u64 P[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
    auto limb52 = [&](const u64* w, int k) -> u64 {  // bits [52k, 52k+52) of a 256-bit number
        int bit = 52 * k, word = bit / 64, off = bit % 64;
        u64 v = word < 4 ? w[word] >> off : 0;
        if (off > 12 && word + 1 < 4) v |= w[word + 1] << (64 - off);
        return v & ((1ULL << 52) - 1);
    };
    double hp[5];
    u64 p52[5];
    for (int k = 0; k < 5; ++k) { p52[k] = limb52(P, k); hp[k] = (double)p52[k]; }
    // -p^-1 mod 2^52 by Newton iteration on 64-bit words
    u64 inv = 1;
    for (int i = 0; i < 6; ++i) inv *= 2 - P[0] * inv;
    u64 ninv = (0 - inv) & ((1ULL << 52) - 1);
    double hn = (double)ninv;
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(kP52), hp, sizeof hp));
    CHECK(hipMemcpyToSymbol(HIP_SYMBOL(kNinv52), &hn, sizeof hn));

    // ---- correctness of the prototype on a few values: one product per thread, compared with a host Montgomery product mod p, R = 2^260
    const int blocks = 256 * 8, threads = 256;
    const size_t n = (size_t)blocks * threads;
    std::vector<double> hx(10 * n);
    std::vector<u64> raw(10 * n);
    u64 s = 12345;
    auto next = [&] { s += 0x9e3779b97f4a7c15ULL; u64 z = s; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); };
    for (size_t i = 0; i < 10 * n; ++i) {
        u64 v = next() & ((1ULL << 52) - 1);
        if (i % 5 == 4) v &= (1ULL << 44) - 1;   // top limb: value < 2^252 < p
        raw[i] = v; hx[i] = (double)v;
    }
    double* dx;
    CHECK(hipMalloc(&dx, 10 * n * sizeof(double)));
    CHECK(hipMemcpy(dx, hx.data(), 10 * n * sizeof(double), hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_mul52, dim3(1), dim3(256), 0, 0, dx, 1);   // a = a b R^-1, b = b a' R^-1 for the first 256 threads
    CHECK(hipDeviceSynchronize());
    std::vector<double> got(10 * 256);
    CHECK(hipMemcpy(got.data(), dx, got.size() * sizeof(double), hipMemcpyDeviceToHost));
    // host: (a b R^-1) mod p through the same limb-wise reduction on exact integers
    int bad = 0;
    for (int t = 0; t < 256 && bad < 3; ++t) {
        const u64* a = &raw[10 * t]; const u64* b = &raw[10 * t + 5];
        u128 col[11] = {0};
        for (int i = 0; i < 5; ++i) for (int j = 0; j < 5; ++j) { u128 pr = (u128)a[i] * b[j]; col[i + j] += (u64)(pr & ((1ULL << 52) - 1)); col[i + j + 1] += (u64)(pr >> 52); }
        for (int i = 0; i < 5; ++i) {
            u64 q = (u64)(((u128)(u64)(col[i] & ((1ULL << 52) - 1)) * ninv) & ((1ULL << 52) - 1));
            for (int j = 0; j < 5; ++j) { u128 pr = (u128)q * p52[j]; col[i + j] += (u64)(pr & ((1ULL << 52) - 1)); col[i + j + 1] += (u64)(pr >> 52); }
            col[i + 1] += col[i] >> 52;
        }
        u64 r[5];
        for (int k = 5; k < 10; ++k) { if (k < 9) col[k + 1] += col[k] >> 52; r[k - 5] = (u64)(col[k] & ((1ULL << 52) - 1)); }
        // both sides are a b R^-1 up to a small multiple of p (the device uses signed quotient digits and an offset of p): compare mod p
        auto modp = [&](const u64* l52, u64 top_extra) {   // value = sum l52[k] 2^(52k) (+ top limb may exceed 52 bits), reduced below p
            u128 acc[5];
            for (int k = 0; k < 5; ++k) acc[k] = l52[k];
            acc[4] += (u128)top_extra << 52;
            std::vector<u64> v(5);
            for (int rep = 0; rep < 8; ++rep) {
                // compare acc with p52 limb-wise after normalising
                for (int k = 0; k < 4; ++k) { acc[k + 1] += acc[k] >> 52; acc[k] &= ((1ULL << 52) - 1); }
                bool ge = true;
                for (int k = 4; k >= 0; --k) { if ((u64)acc[k] != p52[k]) { ge = (u64)acc[k] > p52[k]; break; } }
                if (!ge) break;
                // acc -= p (borrow through 52-bit limbs)
                long long borrow = 0;
                for (int k = 0; k < 5; ++k) { long long d = (long long)(u64)acc[k] - (long long)p52[k] - borrow; borrow = d < 0; if (d < 0 && k < 4) d += 1LL << 52; acc[k] = (u64)d; }
            }
            for (int k = 0; k < 5; ++k) v[k] = (u64)acc[k];
            return v;
        };
        u64 g52[5];
        for (int k = 0; k < 5; ++k) g52[k] = (u64)got[10 * t + k];
        u64 gtop = g52[4] >> 52; g52[4] &= (1ULL << 52) - 1;
        if (modp(g52, gtop) != modp(r, 0)) { if (!bad) printf("fe52 MISMATCH thread %d: device limb0 %llx host limb0 %llx\n", t, (unsigned long long)g52[0], (unsigned long long)r[0]); ++bad; }
    }
    printf("fe52 prototype vs host integers on 256 products: %s\n", bad ? "MISMATCH" : "ok");

    // ---- rates
    auto time_it = [&](auto fn) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        fn(); (void)hipDeviceSynchronize();
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) { (void)hipEventRecord(e0); fn(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        return best;
    };
    const int mi = 512;
    CHECK(hipMemcpy(dx, hx.data(), 10 * n * sizeof(double), hipMemcpyHostToDevice));
    float ms = time_it([&] { hipLaunchKernelGGL(k_mul52, dim3(blocks), dim3(threads), 0, 0, dx, mi); });
    printf("FP  mul fe52 (5 x 52-bit, v_fma_f64 hi/lo)   %8.3f ms  %.3e modmul/s\n", ms, n * 2.0 * mi / (ms * 1e-3));
    u32* d29;
    CHECK(hipMalloc(&d29, 18 * n * 4));
    std::vector<u32> h29(18 * n);
    for (size_t i = 0; i < 18 * n; ++i) h29[i] = (u32)next() & ((i % 9 == 8) ? 0x3fffff : 0x1fffffff);
    CHECK(hipMemcpy(d29, h29.data(), 18 * n * 4, hipMemcpyHostToDevice));
    ms = time_it([&] { hipLaunchKernelGGL(k_mul29, dim3(blocks), dim3(threads), 0, 0, d29, mi); });
    printf("FP  mul fe29 (9 x 29-bit, v_mad_i64_i32)     %8.3f ms  %.3e modmul/s\n", ms, n * 2.0 * mi / (ms * 1e-3));
    return 0;
}
