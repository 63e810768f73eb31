// Fp on 9 x 29-bit SIGNED limbs with Montgomery radix R' = 2^261 — the in-register form of the MSM hot loops.
//
// Why a second representation: on gfx950 every integer VALU instruction issues at the same rate (measured,
// profiles/r01_alu_microbench.txt), and the 8 x 32-bit product spends half of its instructions collecting carries
// (v_mad_u64_u32 + v_addc per partial product).  With 29-bit limbs a whole column of partial products fits a 64-bit
// accumulator: no carry collection (206 instructions instead of 313; measured 1.74e11 vs 1.25e11 products/s).
// Residues are LAZY and SIGNED: a value is only defined mod p, limbs are two's-complement int32, so a difference
// needs no multiple-of-p offset, negation is a limb-wise negate, and because R' = 2^261 = 169 p a product of
// operands of magnitude up to ~7p returns with magnitude < 1.3p — no conditional subtraction anywhere in the loop.
//
// Contracts (asserted by tests/test_fe29_cpu.py through the portable path below; units of p for values):
//   tight : |limb_i| <= 2^29 + 2 for i < 8, limb 8 carries the rest (|value| < 2^258)
//   loose : |limb_i| < 2^30 (a sum or difference of two tight values)
//   mul   : one operand tight, the other loose;   |a*b|/169 + 1 bounds the magnitude of the result
//   mul2  : a*b + c*d, all four operands tight
//   norm  : one parallel signed carry sweep, any int32 limbs -> tight, limb 0 exact (= value mod 2^29)
// Memory stays in gnark's 8 x 32 Montgomery form (R = 2^256): from32<5>() re-limbs value*32 (== the R' form),
// to32_div32() multiplies by 2^256 * 2^-261, reduces to [0,p) and re-packs.
#pragma once
#include "fe.cuh"

namespace zk {

struct Fp29Params {
    ZK_HD static constexpr u32 mod29(int i) {
        constexpr u32 m[9] = {0x187cfd47u, 0x010460b6u, 0x1c72a34fu, 0x02d522d0u, 0x1585d978u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
        return m[i];
    }
    ZK_HD static constexpr u32 one29(int i) {  // 2^261 mod p
        constexpr u32 m[9] = {0x157ccc21u, 0x141c2758u, 0x185230d3u, 0x014c0419u, 0x0aa36fb9u, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
        return m[i];
    }
    static constexpr u32 INV29 = 0x04866389u;   // -p^-1 mod 2^29
    static constexpr u32 PINV29 = 0x1b799c77u;  //  p^-1 mod 2^29
};

struct Fr29Params {  // the scalar field, same shape (both moduli share their top 125 bits, so every bound carries over)
    ZK_HD static constexpr u32 mod29(int i) {
        constexpr u32 m[9] = {0x10000001u, 0x1f0fac9fu, 0x0e5c2450u, 0x07d090f3u, 0x1585d283u, 0x02db40c0u, 0x00a6e141u, 0x0e5c2634u, 0x0030644eu};
        return m[i];
    }
    ZK_HD static constexpr u32 one29(int i) {  // 2^261 mod r
        constexpr u32 m[9] = {0x0fffff57u, 0x1ea70ab4u, 0x052c068bu, 0x17504f49u, 0x0aa8075bu, 0x1d4240ceu, 0x11d54c07u, 0x052ac7a8u, 0x000dc836u};
        return m[i];
    }
    static constexpr u32 INV29 = 0x0fffffffu;   // -r^-1 mod 2^29
    static constexpr u32 PINV29 = 0x10000001u;  //  r^-1 mod 2^29
};

#if defined(__HIP_DEVICE_COMPILE__)
#include "fe29_asm.inc"
#endif

// PP = limb constants of the modulus, F32 = the 8 x 32 memory type of the same field (Fe<FpParams> / Fe<FrParams>)
template <class PP, class F32>
struct Fe29T {
    u32 l[9];  // two's-complement int32 limbs
    static constexpr u32 M29 = 0x1fffffffu;
    typedef PP P;
    typedef Fe29T Fp29;  // the member functions below were written for the base field; the name is kept local
    // differences of two (near-)tight values may feed mul / sqr / mul2 as they are: |limb| < 2^29 + 4 keeps every column of
    // 9 (18) operand products + 9 reduction products below 27 * 2^58 < 2^63, and the zero test reads only l[0] mod 2^29
    static constexpr bool kLazyDiff = true;
    typedef F32 Fp;
    typedef int32_t i32;
    typedef int64_t i64;

    ZK_HD static Fp29 zero() {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = 0;
        return r;
    }
    ZK_HD static Fp29 one() {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = P::one29(i);
        return r;
    }
    // parallel signed carry sweep
    ZK_HD static void norm(u32* x) {
        u32 c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { c[i] = (u32)((i32)x[i] >> 29); x[i] &= M29; }
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i + 1] += c[i];
    }
    // exact sequential signed carry propagation: limbs 0..7 in [0, 2^29), limb 8 signed (cold paths)
    ZK_HD static void carry_exact(u32* x) {
        for (int i = 0; i < 8; ++i) { x[i + 1] += (u32)((i32)x[i] >> 29); x[i] &= M29; }
    }
    // portable products (host builds; same column schedule as fe29_asm.inc)
    ZK_HD static Fp29 mul_portable(const Fp29& a, const Fp29& b) {
        u32 q[9];
        Fp29 r;
        i64 acc = 0;
        for (int k = 0; k < 18; ++k) {
            for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); ++i) acc += (i64)(i32)a.l[i] * (i32)b.l[k - i];
            for (int i = (k > 8 ? k - 8 : 0); i < k && i <= 8; ++i) acc += (i64)q[i] * (i64)P::mod29(k - i);
            if (k < 9) { q[k] = ((u32)acc * P::INV29) & M29; acc += (i64)q[k] * (i64)P::mod29(0); }
            else if (k < 17) r.l[k - 9] = (u32)acc & M29;
            else r.l[8] = (u32)acc;
            acc >>= 29;
        }
        return r;
    }
    ZK_HD static Fp29 mul2_portable(const Fp29& a, const Fp29& b, const Fp29& c, const Fp29& d) {
        u32 q[9];
        Fp29 r;
        i64 acc = 0;
        for (int k = 0; k < 18; ++k) {
            for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); ++i)
                acc += (i64)(i32)a.l[i] * (i32)b.l[k - i] + (i64)(i32)c.l[i] * (i32)d.l[k - i];
            for (int i = (k > 8 ? k - 8 : 0); i < k && i <= 8; ++i) acc += (i64)q[i] * (i64)P::mod29(k - i);
            if (k < 9) { q[k] = ((u32)acc * P::INV29) & M29; acc += (i64)q[k] * (i64)P::mod29(0); }
            else if (k < 17) r.l[k - 9] = (u32)acc & M29;
            else r.l[8] = (u32)acc;
            acc >>= 29;
        }
        return r;
    }
    ZK_HD static Fp29 mul(const Fp29& a, const Fp29& b) {
#if defined(__HIP_DEVICE_COMPILE__)
        Fp29 r;
        mont_mul29_asm<P>(r.l, a.l, b.l);
        return r;
#else
        return mul_portable(a, b);
#endif
    }
    // a TIGHT: the device form doubles the limbs once and takes the 36 off-diagonal products a single time (178 instructions)
    ZK_HD static Fp29 sqr(const Fp29& a) {
#if defined(__HIP_DEVICE_COMPILE__)
        Fp29 r;
        mont_sqr29_asm<P>(r.l, a.l);
        return r;
#else
        return mul_portable(a, a);
#endif
    }
    ZK_HD static Fp29 mul2(const Fp29& a, const Fp29& b, const Fp29& c, const Fp29& d) {
#if defined(__HIP_DEVICE_COMPILE__)
        Fp29 r;
        mont_mul2_29_asm<P>(r.l, a.l, b.l, c.l, d.l);
        return r;
#else
        return mul2_portable(a, b, c, d);
#endif
    }
#if defined(__HIP_DEVICE_COMPILE__)
    // products whose first factor(s) are WAVE-UNIFORM constants (table rows read through the scalar cache): the constant
    // limbs are SGPR operands of the multiply-adds — no VGPR copies.  A non-uniform `k` would silently use lane 0's value.
    __device__ __forceinline__ static Fp29 mulk(const Fp29& k, const Fp29& b) {
        Fp29 r;
        mont_mul29_k_asm<P>(r.l, k.l, b.l);
        return r;
    }
    __device__ __forceinline__ static Fp29 mul2k(const Fp29& k0, const Fp29& b, const Fp29& k1, const Fp29& d) {
        Fp29 r;
        mont_mul2_29_k_asm<P>(r.l, k0.l, b.l, k1.l, d.l);
        return r;
    }
#endif
    // limb-wise, no carry sweep: tight (+/-) tight -> loose
    ZK_HD static Fp29 add_l(const Fp29& a, const Fp29& b) {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
        return r;
    }
    ZK_HD static Fp29 sub_l(const Fp29& a, const Fp29& b) {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] - b.l[i];
        return r;
    }
    ZK_HD static Fp29 neg(const Fp29& a) {  // tight -> tight
        Fp29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = 0u - a.l[i];
        return r;
    }
    ZK_HD static Fp29 cneg(const Fp29& a, bool n) {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = n ? 0u - a.l[i] : a.l[i];
        return r;
    }
    ZK_HD static Fp29 normed(Fp29 a) { norm(a.l); return a; }
    ZK_HD static Fp29 add_n(const Fp29& a, const Fp29& b) { return normed(add_l(a, b)); }
    ZK_HD static Fp29 sub_n(const Fp29& a, const Fp29& b) { return normed(sub_l(a, b)); }

    // x == 0 (mod p) for a normalised x with |x| < 64p.  Fast filter on the exact low limb: x = k*p forces
    // k = x0 * p^-1 (mod 2^29) with |k| < 64; anything else is non-zero (a random x enters the slow path with
    // probability 2^-22).
    ZK_HD bool is_zero_mod_p() const {
        u32 k = (l[0] * P::PINV29) & M29;
        if (k >= 64u && k <= M29 - 63u) return false;
        // limbs by value: an out-of-line MEMBER would force *this into scratch in every iteration of the callers' loops
        return equals_kp9(l[0], l[1], l[2], l[3], l[4], l[5], l[6], l[7], l[8], (i32)(k << 3) >> 3);  // sign-extend the 29-bit residue
    }
    ZK_HD bool equals_kp(i32 k) const { return equals_kp9(l[0], l[1], l[2], l[3], l[4], l[5], l[6], l[7], l[8], k); }
    ZK_HD_NOINLINE static bool equals_kp9(u32 a0, u32 a1, u32 a2, u32 a3, u32 a4, u32 a5, u32 a6, u32 a7, u32 a8, i32 k) {
        u32 t[9];
        i64 carry = 0;
        for (int i = 0; i < 9; ++i) {
            i64 v = (i64)P::mod29(i) * k + carry;
            t[i] = i < 8 ? (u32)(v & M29) : (u32)v;
            carry = v >> 29;
        }
        u32 x[9] = {a0, a1, a2, a3, a4, a5, a6, a7, a8};
        carry_exact(x);
        u32 d = 0;
        for (int i = 0; i < 9; ++i) d |= x[i] ^ t[i];
        return d == 0;
    }
    ZK_HD bool all_limbs_zero() const {
        u32 o = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) o |= l[i];
        return o == 0;
    }

    // ---- conversions -------------------------------------------------------------------------------------------
    // 8x32 (value v < 2^256) -> 9x29 limbs of v * 2^SH  (SH = 5: gnark Montgomery form -> R' form; SH = 0: plain)
    template <int SH>
    ZK_HD static Fp29 from32(const Fp& a) {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int bit = 29 * i - SH;  // first bit of v that lands in limb i
            u32 v;
            if (bit < 0) {
                v = (a.v[0] << (-bit)) & M29;
            } else {
                const int w = bit >> 5, s = bit & 31;
                u64 two = a.v[w];
                if (w + 1 < 8) two |= (u64)a.v[w + 1] << 32;
                v = (u32)(two >> s);
                if (i < 8) v &= M29;
            }
            r.l[i] = v;
        }
        return r;
    }
    // from32<5>() output (or its limb-wise negation), |v| < 32p  ->  same residue in (-0.1p, 2.4p), tight limbs, no product:
    // q = floor(floor(v / 2^254) * 169/128) never exceeds floor(v/p) and falls short of it by at most 2
    ZK_HD static Fp29 reduce32(const Fp29& a) {
        const i32 q = (((i32)a.l[8] >> 22) * 169) >> 7;
        Fp29 r;
        i64 c = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            i64 t = (i64)(i32)a.l[i] - (i64)q * (i64)P::mod29(i) + c;
            r.l[i] = i < 8 ? ((u32)t & M29) : (u32)t;
            c = t >> 29;
        }
        return r;
    }
    // the same with the quotient estimate lowered by one: result in (0.9p, 3.4p), limbs 0..7 exact in [0, 2^29),
    // limb 8 >= 0 — the form pack32() stores
    ZK_HD static Fp29 reduce32_pos(const Fp29& a) {
        const i32 q = ((((i32)a.l[8] >> 22) * 169) >> 7) - 1;
        Fp29 r;
        i64 c = 0;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            i64 t = (i64)(i32)a.l[i] - (i64)q * (i64)P::mod29(i) + c;
            r.l[i] = i < 8 ? ((u32)t & M29) : (u32)t;
            c = t >> 29;
        }
        return r;
    }
    // exact non-negative limbs (reduce32_pos output, value < 2^256) -> 8 x 32 words of the same integer; from32<0>
    // is the inverse.  The NTT keeps its inter-pass arrays in this form (R' domain, value in [0, 4p)).
    ZK_HD Fp pack32() const {
        Fp r;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int bit = 32 * w;
            const int i = bit / 29, s = bit % 29;
            u64 v = (u64)l[i] >> s;
            int have = 29 - s;
            if (i + 1 < 9) { v |= (u64)l[i + 1] << have; have += 29; }
            if (have < 32 && i + 2 < 9) v |= (u64)l[i + 2] << have;
            r.v[w] = (u32)v;
        }
        return r;
    }
    // lazy value x (|x| < 16p) -> canonical 8x32 of (x * 2^256 * 2^-261) mod p = x / 32: back to gnark's form
    ZK_HD static Fp to32_div32(const Fp29& x) {
        Fp29 c256 = zero();
        c256.l[8] = 1u << 24;  // 2^256
        Fp29 y = mul(x, c256);  // in (-p, 2p)
        return y.to32_canonical();
    }
    // reduce a value in (-2p, 4p) to [0,p) and re-pack to 8x32
    ZK_HD Fp to32_canonical() const {
        u32 x[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) x[i] = l[i];
        carry_exact(x);
        for (int rep = 0; rep < 2; ++rep) {  // negative: add p
            if ((i32)x[8] < 0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) x[i] += P::mod29(i);
                carry_exact(x);
            }
        }
        for (int rep = 0; rep < 3; ++rep) {  // conditional subtractions of p
            u32 t[9];
            u32 bw = 0;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                u32 d = x[i] - P::mod29(i) - bw;
                bw = i < 8 ? (d >> 31) : ((i32)d < 0 ? 1u : 0u);
                t[i] = i < 8 ? (d & M29) : d;
            }
            if (!bw) {
#pragma unroll
                for (int i = 0; i < 9; ++i) x[i] = t[i];
            }
        }
        Fp r;
#pragma unroll
        for (int w = 0; w < 8; ++w) {
            const int bit = 32 * w;
            const int i = bit / 29, s = bit % 29;
            u64 v = (u64)x[i] >> s;
            int have = 29 - s;
            if (i + 1 < 9) { v |= (u64)x[i + 1] << have; have += 29; }
            if (have < 32 && i + 2 < 9) v |= (u64)x[i + 2] << have;
            r.v[w] = (u32)v;
        }
        return r;
    }

    // ---- the interface the generic 29-bit group law uses (real field: one lane per element) ----------------------
    ZK_HD bool zero_mod_p() const { return is_zero_mod_p(); }
    // R*D - Y1*PPP with ONE reduction (all four tight)
    ZK_HD static Fp29 y3(const Fp29& R, const Fp29& D, const Fp29& Y1, const Fp29& PPP) { return mul2(R, D, neg(Y1), PPP); }
};
typedef Fe29T<Fp29Params, Fe<FpParams>> Fp29;
typedef Fe29T<Fr29Params, Fe<FrParams>> Fr29;

// ---- XYZZ accumulator on a 29-bit field F (Fp29, or the lane-pair Fp2 of fp2_lanepair.cuh) ------------------------
template <class F>
struct XYZZ29T {
    F x, y, zz, zzz;
    ZK_HD static XYZZ29T inf() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
    ZK_HD bool is_inf() const { return zz.all_limbs_zero(); }  // ZZ of a live accumulator is a product of non-zero residues
};

// doubling of an affine point (coordinates in R' form, magnitude < 32p): cold path, out of line
template <class F>
ZK_HD_NOINLINE XYZZ29T<F> xyzz29_dbl_affine(F x2, F y2) {
    F one = F::one();
    F x1 = F::mul(one, x2), y1 = F::mul(one, y2);  // magnitude < 1.4p
    F U = F::add_n(y1, y1);
    F V = F::sqr(U);
    F W = F::mul(U, V);
    F S = F::mul(x1, V);
    F X2 = F::sqr(x1);
    F M = F::normed(F::add_l(F::add_l(X2, X2), X2));
    F X3 = F::normed(F::sub_l(F::sub_l(F::sqr(M), S), S));
    F Y3 = F::y3(M, F::sub_n(S, X3), y1, W);
    return {X3, Y3, V, W};
}

// acc += (x2, y2): madd-2008-s.  Magnitude bounds (units of p, R'/p = 169; real | complex components):
//   x2,y2 < 32 | ZZ,ZZZ < 1.1 | X1 < 4.2|4.7, Y1 < 1.1|2.3 | U2,S2 < 1.4 | P < 5.4|6.1, R < 2.3|3.7 | PP < 1.9, PPP,Q < 1.2
//   X3 = R^2 - PPP - 2Q | D = Q - X3 < 5.8 | Y3 = R*D - Y1*PPP (real: one fused reduction)
// `dbl` supplies 2*(x2, y2) in the (rare) P == Q case.  The kernels pass a callable that re-reads the point from memory
// inside an out-of-line function: handing x2, y2 to one by value made the compiler park 72 bytes in scratch in EVERY
// iteration (measured: 25 GB of extra HBM writes per launch).
template <class F, class Dbl>
ZK_HD void xyzz29_madd(XYZZ29T<F>& acc, const F& x2, const F& y2, Dbl dbl) {
    if (acc.is_inf()) {  // divergent whenever any lane of the wave starts a bucket: keep it free of products
        acc.x = F::reduce32(x2); acc.y = F::reduce32(y2); acc.zz = F::one(); acc.zzz = acc.zz;
        return;
    }
    F U2 = F::mul(acc.zz, x2);
    F S2 = F::mul(acc.zzz, y2);
    F Pd = F::kLazyDiff ? F::sub_l(U2, acc.x) : F::sub_n(U2, acc.x);
    F Rd = F::kLazyDiff ? F::sub_l(S2, acc.y) : F::sub_n(S2, acc.y);
    if (Pd.zero_mod_p()) {
        if (Rd.zero_mod_p()) acc = dbl();
        else acc = XYZZ29T<F>::inf();
        return;
    }
    F PP = F::sqr(Pd);
    F PPP = F::mul(Pd, PP);
    F Q = F::mul(acc.x, PP);
    F X3 = F::normed(F::sub_l(F::sub_l(F::sub_l(F::sqr(Rd), PPP), Q), Q));
    F D = F::kLazyDiff ? F::sub_l(Q, X3) : F::sub_n(Q, X3);
    acc.y = F::y3(Rd, D, acc.y, PPP);
    acc.x = X3;
    acc.zz = F::mul(acc.zz, PP);
    acc.zzz = F::mul(acc.zzz, PPP);
}

// general doubling, dbl-2008-s-1 (a = 0).  Bounds: X1 < 4.7, Y1 < 2.3 | U < 4.6, V < 1.2, W,S < 1.1 | M = 3 X1^2 < 3.4 |
// X3 = M^2 - 2S < 3.2 | D = S - X3 < 4.3 | Y3 = M*D - W*Y1 < 1.1
template <class F>
ZK_HD XYZZ29T<F> xyzz29_dbl(const XYZZ29T<F>& p) {
    if (p.is_inf()) return p;
    F U = F::add_n(p.y, p.y);
    F V = F::sqr(U);
    F W = F::mul(U, V);
    F S = F::mul(p.x, V);
    F X2 = F::sqr(p.x);
    F M = F::normed(F::add_l(F::add_l(X2, X2), X2));
    F X3 = F::normed(F::sub_l(F::sub_l(F::sqr(M), S), S));
    F Y3 = F::y3(M, F::sub_n(S, X3), p.y, W);
    return {X3, Y3, F::mul(V, p.zz), F::mul(W, p.zzz)};
}

// acc += p, both XYZZ: add-2008-s with the exceptional cases.  Bounds (inputs as left by madd / add / dbl: X < 4.7, Y < 2.3,
// ZZ, ZZZ < 1.1): U1,U2,S1,S2 < 1.1 | P,R < 2.2 | PP,PPP,Q < 1.1 | X3 = R^2 - PPP - 2Q < 4.2 | D = Q - X3 < 5.3 |
// Y3 = R*D - S1*PPP < 1.1 | ZZ3, ZZZ3 < 1.1
template <class F>
ZK_HD void xyzz29_add(XYZZ29T<F>& acc, const XYZZ29T<F>& p) {
    if (p.is_inf()) return;
    if (acc.is_inf()) { acc = p; return; }
    F U1 = F::mul(acc.x, p.zz), U2 = F::mul(p.x, acc.zz);
    F S1 = F::mul(acc.y, p.zzz), S2 = F::mul(p.y, acc.zzz);
    F Pd = F::sub_n(U2, U1);
    F Rd = F::sub_n(S2, S1);
    if (Pd.zero_mod_p()) {
        if (Rd.zero_mod_p()) acc = xyzz29_dbl<F>(p);
        else acc = XYZZ29T<F>::inf();
        return;
    }
    F PP = F::sqr(Pd);
    F PPP = F::mul(Pd, PP);
    F Q = F::mul(U1, PP);
    F X3 = F::normed(F::sub_l(F::sub_l(F::sub_l(F::sqr(Rd), PPP), Q), Q));
    F D = F::sub_n(Q, X3);
    acc.y = F::y3(Rd, D, S1, PPP);
    acc.x = X3;
    acc.zz = F::mul(F::mul(acc.zz, p.zz), PP);
    acc.zzz = F::mul(F::mul(acc.zzz, p.zzz), PPP);
}

template <class F>
ZK_HD void xyzz29_madd(XYZZ29T<F>& acc, const F& x2, const F& y2) {
    xyzz29_madd<F>(acc, x2, y2, [&]() { return xyzz29_dbl_affine<F>(x2, y2); });
}

typedef XYZZ29T<Fp29> XYZZ29;

// Raw accumulator image: the 4 x 9 signed limbs as they sit in registers (144 B, R' domain, lazily reduced; all-zero =
// infinity).  The level-1 kernels park finished buckets in this form because the store sits on a divergent path
// (some lane of a wave closes a bucket in ~half of all iterations): 36 plain stores instead of four domain changes.
static constexpr int RAW29_WORDS = 36;
#if defined(__HIPCC__)  // uint4 is a HIP vector type; tests/hostlib builds this header with plain g++
ZK_HD void raw29_store(u32* dst, const XYZZ29& a) {
    uint4* d = (uint4*)dst;
    u32 w[36];
#pragma unroll
    for (int i = 0; i < 9; ++i) { w[i] = a.x.l[i]; w[9 + i] = a.y.l[i]; w[18 + i] = a.zz.l[i]; w[27 + i] = a.zzz.l[i]; }
#pragma unroll
    for (int i = 0; i < 9; ++i) d[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
ZK_HD XYZZ29 raw29_load(const u32* src) {
    const uint4* d = (const uint4*)src;
    u32 w[36];
#pragma unroll
    for (int i = 0; i < 9; ++i) { uint4 q = d[i]; w[4 * i] = q.x; w[4 * i + 1] = q.y; w[4 * i + 2] = q.z; w[4 * i + 3] = q.w; }
    XYZZ29 a;
#pragma unroll
    for (int i = 0; i < 9; ++i) { a.x.l[i] = w[i]; a.y.l[i] = w[9 + i]; a.zz.l[i] = w[18 + i]; a.zzz.l[i] = w[27 + i]; }
    return a;
}
#endif

}  // namespace zk
