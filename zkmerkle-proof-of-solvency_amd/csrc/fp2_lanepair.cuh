// Fp2 arithmetic with ONE Fp component per lane (gfx950 device code only): lanes 2k / 2k+1 of a wavefront hold the
// real / imaginary part of the same Fp2 value and exchange operands with DPP quad_perm [1,0,3,2].
//
// Why: a G2 bucket accumulator in one lane needs >256 VGPRs (XYZZ<Fp2> = 64 registers + temporaries) and ran at one
// wave per SIMD with scratch spills.  Split across a lane pair the per-lane state is that of a G1 accumulator, and
// the complex product needs no Karatsuba: each lane computes its output component as ONE fused double product
//   even lane: a0*b0 + a1*(p - b1)      odd lane: a1*b0 + a0*b1
// with a single interleaved Montgomery reduction (192 multiply-add pairs instead of 2 x 128).
#pragma once
#include "ec.cuh"
#include "fe29.cuh"

namespace zk {

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ u32 lp_swap(u32 x) {
    return (u32)__builtin_amdgcn_mov_dpp((int)x, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}
#else
__host__ __device__ inline u32 lp_swap(u32 x) { return x; }  // host pass only parses this header
#endif

struct Fp2L {
    Fp c;  // this lane's component

    ZK_HD static bool odd() {
#if defined(__HIP_DEVICE_COMPILE__)
        return (threadIdx.x & 1u) != 0;
#else
        return false;
#endif
    }
    ZK_HD static Fp partner(const Fp& x) {
        Fp r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = lp_swap(x.v[i]);
        return r;
    }
    ZK_HD static Fp2L zero() { return {Fp::zero()}; }
    ZK_HD static Fp2L one() {
        Fp o = Fp::one();
        Fp z = Fp::zero();
        Fp2L r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.c.v[i] = odd() ? z.v[i] : o.v[i];
        return r;
    }
    ZK_HD bool is_zero() const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) o |= c.v[i];
        o |= lp_swap(o);
        return o == 0;
    }
    ZK_HD bool operator==(const Fp2L& b) const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 8; ++i) o |= c.v[i] ^ b.c.v[i];
        o |= lp_swap(o);
        return o == 0;
    }
    ZK_HD bool operator!=(const Fp2L& b) const { return !(*this == b); }
    ZK_HD static Fp2L add(const Fp2L& x, const Fp2L& y) { return {Fp::add(x.c, y.c)}; }
    ZK_HD static Fp2L sub(const Fp2L& x, const Fp2L& y) { return {Fp::sub(x.c, y.c)}; }
    ZK_HD static Fp2L neg(const Fp2L& x) { return {Fp::neg(x.c)}; }
    ZK_HD static Fp2L dbl(const Fp2L& x) { return {Fp::dbl(x.c)}; }

    // (a*X + b*Y) * R^-1 mod p for a,b < p and X,Y <= p: fused product scanning, one reduction, result < 1.5p
    ZK_HD static Fp mul2(const Fp& a, const Fp& X, const Fp& b, const Fp& Y) {
#if defined(__HIP_DEVICE_COMPILE__)
        Fp r = a;
        mont_mul2_asm<FpParams>(r.v, X.v, b.v, Y.v);
        return r;
#else
        return Fp::add(Fp::mul(a, X), Fp::mul(b, Y));  // host pass only parses this header
#endif
    }
    ZK_HD static Fp2L mul(const Fp2L& x, const Fp2L& y) {
        const bool od = odd();
        Fp ao = partner(x.c), bo = partner(y.c);
        // p - bo (in [1, p]); only the even lane uses it
        Fp nb;
        {
            u32 bw = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                u64 d = (u64)FpParams::mod(i) - bo.v[i] - bw;
                nb.v[i] = (u32)d;
                bw = (u32)(d >> 32) & 1u;
            }
        }
        Fp X, Y;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            X.v[i] = od ? bo.v[i] : y.c.v[i];
            Y.v[i] = od ? y.c.v[i] : nb.v[i];
        }
        return {mul2(x.c, X, ao, Y)};
    }
    // even: (a0 + a1)(a0 - a1)      odd: (a1 + a1) * a0
    ZK_HD static Fp2L sqr(const Fp2L& x) {
        const bool od = odd();
        Fp ao = partner(x.c);
        Fp s;
#pragma unroll
        for (int i = 0; i < 8; ++i) s.v[i] = od ? x.c.v[i] : ao.v[i];
        Fp U = Fp::add(x.c, s);
        Fp d = Fp::sub(x.c, ao);
        Fp V;
#pragma unroll
        for (int i = 0; i < 8; ++i) V.v[i] = od ? ao.v[i] : d.v[i];
        return {Fp::mul_body(U, V)};
    }
};

// ---- the same lane-pair layout on the 29-bit signed lazy form (fe29.cuh): interface of the generic xyzz29_madd ----
struct Fp2L29 {
    Fp29 c;  // this lane's component
    static constexpr bool kLazyDiff = false;  // the mixed addition keeps its normalising differences on this field

    ZK_HD static bool odd() { return Fp2L::odd(); }
    ZK_HD static Fp29 partner(const Fp29& x) {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = lp_swap(x.l[i]);
        return r;
    }
    ZK_HD static Fp2L29 zero() { return {Fp29::zero()}; }
    ZK_HD static Fp2L29 one() {
        Fp29 o = Fp29::one();
        Fp2L29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.c.l[i] = odd() ? 0u : o.l[i];
        return r;
    }
    ZK_HD bool all_limbs_zero() const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) o |= c.l[i];
        o |= lp_swap(o);
        return o == 0;
    }
    ZK_HD bool zero_mod_p() const {  // both components = 0 (mod p)
        u32 z = c.is_zero_mod_p() ? 1u : 0u;
        z &= lp_swap(z);
        return z != 0;
    }
    ZK_HD static Fp2L29 add_l(const Fp2L29& x, const Fp2L29& y) { return {Fp29::add_l(x.c, y.c)}; }
    ZK_HD static Fp2L29 sub_l(const Fp2L29& x, const Fp2L29& y) { return {Fp29::sub_l(x.c, y.c)}; }
    ZK_HD static Fp2L29 add_n(const Fp2L29& x, const Fp2L29& y) { return {Fp29::add_n(x.c, y.c)}; }
    ZK_HD static Fp2L29 sub_n(const Fp2L29& x, const Fp2L29& y) { return {Fp29::sub_n(x.c, y.c)}; }
    ZK_HD static Fp2L29 neg(const Fp2L29& x) { return {Fp29::neg(x.c)}; }
    ZK_HD static Fp2L29 normed(const Fp2L29& x) { return {Fp29::normed(x.c)}; }
    ZK_HD static Fp2L29 reduce32(const Fp2L29& x) { return {Fp29::reduce32(x.c)}; }
    // (a0 + a1 u)(b0 + b1 u): even lane a0*b0 + a1*(-b1), odd lane a1*b0 + a0*b1 — one fused double product each
    ZK_HD static Fp2L29 mul(const Fp2L29& x, const Fp2L29& y) {
        const bool od = odd();
        Fp29 ao = partner(x.c), bo = partner(y.c);
        Fp29 X, Y;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            X.l[i] = od ? bo.l[i] : y.c.l[i];
            Y.l[i] = od ? y.c.l[i] : 0u - bo.l[i];
        }
        return {Fp29::mul2(x.c, X, ao, Y)};
    }
    // even: (a0 + a1)(a0 - a1)      odd: (a1 + a1) * a0      (U loose, V tight)
    ZK_HD static Fp2L29 sqr(const Fp2L29& x) {
        const bool od = odd();
        Fp29 ao = partner(x.c);
        Fp29 U, V;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            U.l[i] = x.c.l[i] + (od ? x.c.l[i] : ao.l[i]);
            V.l[i] = od ? ao.l[i] : x.c.l[i] - ao.l[i];
        }
        Fp29::norm(V.l);
        return {Fp29::mul(V, U)};
    }
    // R*D - Y1*PPP: two complex products (a four-product fusion would overflow the signed 64-bit columns)
    ZK_HD static Fp2L29 y3(const Fp2L29& R, const Fp2L29& D, const Fp2L29& Y1, const Fp2L29& PPP) {
        return sub_n(mul(R, D), mul(Y1, PPP));
    }
};

}  // namespace zk
