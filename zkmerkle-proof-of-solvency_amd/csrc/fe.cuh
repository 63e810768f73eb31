// BN254 base field Fp and scalar field Fr for gfx950: 8 x 32-bit little-endian limbs, Montgomery form with
// R = 2^256.  Bit-for-bit the in-memory layout of gnark-crypto's fp.Element / fr.Element ([4]uint64 LE) that the
// reference hands to groth16.Prove (src/prover/prover/prover.go:269), so host buffers are consumed without
// conversion.
//
// The multiplier is a product-scanning (column-wise) Montgomery: every 32x32 partial product is ONE
// v_mad_u64_u32 into a 64-bit column accumulator plus ONE v_addc collecting the carry-out, i.e. 2 VALU issues per
// partial product and no zero-extension moves (an operand-scanning CIOS costs ~4 per product on this ISA
// because CDNA has no multiply-add with carry-in).  The same routine is host-compilable (plain C++) so the
// library's host-side finishing steps and the CPU unit tests run the identical arithmetic.
#pragma once
#include <stdint.h>
#include <stddef.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#if defined(__HIP_DEVICE_COMPILE__)
#define ZK_HD __host__ __device__ __forceinline__
#else
#define ZK_HD __host__ __device__ inline  /* host pass: let the x86 inliner decide (forcing it explodes compile time) */
#endif
#define ZK_D __device__ __forceinline__
#define ZK_HD_NOINLINE __host__ __device__ __attribute__((noinline))
#else
#define ZK_HD inline
#define ZK_D inline
#define ZK_HD_NOINLINE __attribute__((noinline))
#endif

namespace zk {

typedef uint32_t u32;
typedef uint64_t u64;

struct FpParams {
    ZK_HD static constexpr u32 mod(int i) {
        constexpr u32 m[8] = {0xd87cfd47u, 0x3c208c16u, 0x6871ca8du, 0x97816a91u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return m[i];
    }
    ZK_HD static constexpr u32 one(int i) {  // R mod p
        constexpr u32 m[8] = {0xc58f0d9du, 0xd35d438du, 0xf5c70b3du, 0x0a78eb28u, 0x7879462cu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return m[i];
    }
    ZK_HD static constexpr u32 r2(int i) {  // R^2 mod p
        constexpr u32 m[8] = {0x538afa89u, 0xf32cfc5bu, 0xd44501fbu, 0xb5e71911u, 0x0a417ff6u, 0x47ab1effu, 0xcab8351fu, 0x06d89f71u};
        return m[i];
    }
    static constexpr u32 INV = 0xe4866389u;  // -p^-1 mod 2^32
};
struct FrParams {
    ZK_HD static constexpr u32 mod(int i) {
        constexpr u32 m[8] = {0xf0000001u, 0x43e1f593u, 0x79b97091u, 0x2833e848u, 0x8181585du, 0xb85045b6u, 0xe131a029u, 0x30644e72u};
        return m[i];
    }
    ZK_HD static constexpr u32 one(int i) {
        constexpr u32 m[8] = {0x4ffffffbu, 0xac96341cu, 0x9f60cd29u, 0x36fc7695u, 0x7879462eu, 0x666ea36fu, 0x9a07df2fu, 0x0e0a77c1u};
        return m[i];
    }
    ZK_HD static constexpr u32 r2(int i) {
        constexpr u32 m[8] = {0xae216da7u, 0x1bb8e645u, 0xe35c59e3u, 0x53fe3ab1u, 0x53bb8085u, 0x8c49833du, 0x7f4e44a5u, 0x0216d0b1u};
        return m[i];
    }
    static constexpr u32 INV = 0xefffffffu;
};

// 96-bit column accumulator step: (ovf:acc) += a*b
ZK_HD void mac96(u64& acc, u32& ovf, u32 a, u32 b) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(ovf)
        : "v"(a), "v"(b)
        : "vcc");
#else
    u64 p = (u64)a * b;
    acc += p;
    ovf += acc < p;
#endif
}
// same with a compile-time-constant multiplicand (modulus limb): lets the assembler use a literal / SGPR
ZK_HD void mac96c(u64& acc, u32& ovf, u32 a, u32 c) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %0\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc"
        : "+v"(acc), "+v"(ovf)
        : "v"(a), "s"(c)
        : "vcc");
#else
    mac96(acc, ovf, a, c);
#endif
}

#if defined(__HIP_DEVICE_COMPILE__)
#include "fe_asm.inc"
#endif

template <class P>
struct Fe {
    u32 v[8];

    ZK_HD static Fe zero() {
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = 0;
        return r;
    }
    ZK_HD static Fe one() {
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = P::one(i);
        return r;
    }
    ZK_HD static Fe r2() {
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = P::r2(i);
        return r;
    }
    ZK_HD bool is_zero() const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) o |= v[i];
        return o == 0;
    }
    ZK_HD bool operator==(const Fe& b) const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) o |= v[i] ^ b.v[i];
        return o == 0;
    }
    ZK_HD bool operator!=(const Fe& b) const { return !(*this == b); }

    // r = (t >= mod) ? t - mod : t, where `hi` is an extra (257th) bit of t
    ZK_HD static Fe reduce_once(const u32* t, u32 hi) {
        u32 s[8];
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u64 d = (u64)t[i] - P::mod(i) - bw;
            s[i] = (u32)d;
            bw = (u32)(d >> 32) & 1u;
        }
        bool ge = (hi != 0) | (bw == 0);
        Fe r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = ge ? s[i] : t[i];
        return r;
    }
    ZK_HD static Fe add(const Fe& a, const Fe& b) {
        u32 t[8];
        u32 c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u64 s = (u64)a.v[i] + b.v[i] + c;
            t[i] = (u32)s;
            c = (u32)(s >> 32);
        }
        return reduce_once(t, c);  // p, r < 2^254 so c is always 0 here; kept for generality
    }
    ZK_HD static Fe dbl(const Fe& a) { return add(a, a); }
    ZK_HD static Fe sub(const Fe& a, const Fe& b) {
        u32 t[8];
        u32 bw = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u64 d = (u64)a.v[i] - b.v[i] - bw;
            t[i] = (u32)d;
            bw = (u32)(d >> 32) & 1u;
        }
        u32 mask = 0u - bw;  // add the modulus back when the subtraction borrowed
        Fe r;
        u32 c = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u64 s = (u64)t[i] + (P::mod(i) & mask) + c;
            r.v[i] = (u32)s;
            c = (u32)(s >> 32);
        }
        return r;
    }
    ZK_HD static Fe neg(const Fe& a) {
        Fe z = zero();
        return sub(z, a);  // sub(0,0) = 0
    }
    // Montgomery product a*b*R^-1 mod m, product scanning with interleaved reduction (FIPS)
    ZK_HD static Fe mul_body(const Fe& a, const Fe& b) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(ZK_NO_ASM_MUL)
        Fe r = a;
        mont_mul_asm<P>(r.v, b.v);
        return r;
#else
        return mul_portable(a, b);
#endif
    }
    // portable C++ form of the same product-scanning algorithm (host builds; reference for the asm block)
    ZK_HD static Fe mul_portable(const Fe& a, const Fe& b) {
        u32 m[8];
        u32 t[8];
        u64 acc = 0;
        u32 ovf = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
#pragma unroll
            for (int i = 0; i <= k; ++i) mac96(acc, ovf, a.v[i], b.v[k - i]);
#pragma unroll
            for (int i = 0; i < k; ++i) mac96c(acc, ovf, m[i], P::mod(k - i));
            m[k] = (u32)acc * P::INV;
            mac96c(acc, ovf, m[k], P::mod(0));
            acc = (acc >> 32) | ((u64)ovf << 32);
            ovf = 0;
        }
#pragma unroll
        for (int k = 8; k < 16; ++k) {
#pragma unroll
            for (int i = k - 7; i < 8; ++i) mac96(acc, ovf, a.v[i], b.v[k - i]);
#pragma unroll
            for (int i = k - 7; i < 8; ++i) mac96c(acc, ovf, m[i], P::mod(k - i));
            t[k - 8] = (u32)acc;
            acc = (acc >> 32) | ((u64)ovf << 32);
            ovf = 0;
        }
        return reduce_once(t, (u32)acc);
    }
    // Out-of-line copy (operands and result by value = in VGPRs under the AMDGPU calling convention).  Translation
    // units built with -DZK_MUL_NOINLINE (G2 and the cold partial-sum / reduction kernels) call this instead of
    // inlining ~330 instructions per product: code size and compile time drop by an order of magnitude there.
    ZK_HD_NOINLINE static Fe mul_call(Fe a, Fe b) { return mul_body(a, b); }
    ZK_HD static Fe mul(const Fe& a, const Fe& b) {
#if defined(ZK_MUL_NOINLINE)
        return mul_call(a, b);
#else
        return mul_body(a, b);
#endif
    }
    ZK_HD static Fe sqr(const Fe& a) { return mul(a, a); }
    ZK_HD static Fe from_mont(const Fe& a) {  // canonical value in the same limb layout
        Fe o = zero();
        o.v[0] = 1;
        return mul(a, o);
    }
    ZK_HD static Fe to_mont(const Fe& a) { return mul(a, r2()); }
    ZK_HD static Fe from_u32(u32 x) {
        Fe o = zero();
        o.v[0] = x;
        return to_mont(o);
    }
    // a^e, e given as 8 little-endian 32-bit limbs (not secret: square-and-multiply)
    ZK_HD static Fe pow(const Fe& a, const u32* e, int nlimbs) {
        Fe r = one();
        for (int i = nlimbs * 32 - 1; i >= 0; --i) {
            r = sqr(r);
            if ((e[i >> 5] >> (i & 31)) & 1u) r = mul(r, a);
        }
        return r;
    }
    ZK_HD static Fe pow_u64(const Fe& a, u64 e) {
        u32 l[2] = {(u32)e, (u32)(e >> 32)};
        return pow(a, l, 2);
    }
    ZK_HD static Fe inv(const Fe& a) {  // Fermat: a^(m-2); inv(0) = 0
        u32 e[8];
        u32 bw = 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            u64 d = (u64)P::mod(i) - bw;
            e[i] = (u32)d;
            bw = (u32)(d >> 32) & 1u;
        }
        return pow(a, e, 8);
    }
};

typedef Fe<FpParams> Fp;
typedef Fe<FrParams> Fr;

// ---------------------------------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1)
struct Fp2 {
    Fp a0, a1;
    ZK_HD static Fp2 zero() { return {Fp::zero(), Fp::zero()}; }
    ZK_HD static Fp2 one() { return {Fp::one(), Fp::zero()}; }
    ZK_HD bool is_zero() const { return a0.is_zero() & a1.is_zero(); }
    ZK_HD bool operator==(const Fp2& b) const { return (a0 == b.a0) & (a1 == b.a1); }
    ZK_HD bool operator!=(const Fp2& b) const { return !(*this == b); }
    ZK_HD static Fp2 add(const Fp2& x, const Fp2& y) { return {Fp::add(x.a0, y.a0), Fp::add(x.a1, y.a1)}; }
    ZK_HD static Fp2 sub(const Fp2& x, const Fp2& y) { return {Fp::sub(x.a0, y.a0), Fp::sub(x.a1, y.a1)}; }
    ZK_HD static Fp2 neg(const Fp2& x) { return {Fp::neg(x.a0), Fp::neg(x.a1)}; }
    ZK_HD static Fp2 dbl(const Fp2& x) { return {Fp::dbl(x.a0), Fp::dbl(x.a1)}; }
    ZK_HD static Fp2 mul(const Fp2& x, const Fp2& y) {  // Karatsuba, 3 base products
        Fp v0 = Fp::mul(x.a0, y.a0);
        Fp v1 = Fp::mul(x.a1, y.a1);
        Fp s = Fp::mul(Fp::add(x.a0, x.a1), Fp::add(y.a0, y.a1));
        return {Fp::sub(v0, v1), Fp::sub(Fp::sub(s, v0), v1)};
    }
    ZK_HD static Fp2 sqr(const Fp2& x) {  // (a0+a1)(a0-a1), 2 a0 a1
        Fp t = Fp::mul(x.a0, x.a1);
        Fp c0 = Fp::mul(Fp::add(x.a0, x.a1), Fp::sub(x.a0, x.a1));
        return {c0, Fp::dbl(t)};
    }
    ZK_HD static Fp2 inv(const Fp2& x) {
        Fp n = Fp::inv(Fp::add(Fp::sqr(x.a0), Fp::sqr(x.a1)));
        return {Fp::mul(x.a0, n), Fp::neg(Fp::mul(x.a1, n))};
    }
};

}  // namespace zk
