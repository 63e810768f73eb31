// Fp on 9 x 29-bit limbs with Montgomery radix R' = 2^261 — the in-register form of the MSM hot loops.
//
// Why a second representation: on gfx950 every integer VALU instruction issues at the same rate (measured,
// profiles/r01_alu_microbench.txt), and the 8 x 32-bit product spends half of its instructions collecting carries
// (v_mad_u64_u32 + v_addc per partial product).  With 29-bit limbs a whole column of partial products fits a 64-bit
// accumulator, so the carry collection disappears (206 instructions instead of 313; measured 1.56e11 vs 1.22e11
// products/s), additions are 9 independent adds, and because R' = 2^261 is 169 x p a product of operands as large as
// ~17p comes back below 3p without any conditional subtraction ("lazy" residues: every value is only defined mod p).
//
// Contracts (checked by tests/test_fe29_cpu.py through the portable path below):
//   tight  : limbs 0..7 <= 2^29 + 2^5, limb 8 carries the rest of the value (values stay < 2^259)
//   mul    : inputs with limbs < 2^30 ; output tight, value < a*b/2^261 + p
//   mul2   : a*b + c*d with ALL inputs tight
//   add/sub: inputs tight, output tight (one parallel carry sweep); sub adds a multiple of p (SUBC<K>) that must
//            dominate the subtrahend: K*p > b
// Memory stays in gnark's 8 x 32 Montgomery form (R = 2^256): from32_shift5() re-limbs (value * 32 == the R' form),
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
    // K*p written with every limb i < 8 raised by 2^30 (borrowed from the limb above): a - b + SUBC<K> has
    // non-negative limbs for any tight b < K*p
    ZK_HD static constexpr u32 subc4(int i) {
        constexpr u32 m[9] = {0x41f3f51cu, 0x441182d9u, 0x51ca8d3au, 0x4b548b41u, 0x561765deu, 0x4b6d0300u, 0x429b8502u, 0x597098ceu, 0x00c19137u};
        return m[i];
    }
    ZK_HD static constexpr u32 subc8(int i) {
        constexpr u32 m[9] = {0x43e7ea38u, 0x482305b4u, 0x43951a76u, 0x56a91685u, 0x4c2ecbbeu, 0x56da0603u, 0x45370a06u, 0x52e1319eu, 0x01832271u};
        return m[i];
    }
    ZK_HD static constexpr u32 subc16(int i) {
        constexpr u32 m[9] = {0x47cfd470u, 0x50460b6au, 0x472a34eeu, 0x4d522d0cu, 0x585d977fu, 0x4db40c08u, 0x4a6e140fu, 0x45c2633eu, 0x030644e5u};
        return m[i];
    }
    ZK_HD static constexpr u32 subc64(int i) {
        constexpr u32 m[9] = {0x5f3f51c0u, 0x41182daeu, 0x5ca8d3c0u, 0x5548b436u, 0x41765e03u, 0x56d03029u, 0x49b85043u, 0x57098cffu, 0x0c19139au};
        return m[i];
    }
    static constexpr u32 INV29 = 0x04866389u;   // -p^-1 mod 2^29
    static constexpr u32 PINV29 = 0x1b799c77u;  //  p^-1 mod 2^29
};

#if defined(__HIP_DEVICE_COMPILE__)
#include "fe29_asm.inc"
#endif

struct Fp29 {
    u32 l[9];
    static constexpr u32 M29 = 0x1fffffffu;
    typedef Fp29Params P;

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
    // one parallel carry sweep: limbs < 2^32 in, limbs 0..7 <= 2^29 + 7 out, limb 0 exact (= value mod 2^29)
    ZK_HD static void norm(u32* x) {
        u32 c[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { c[i] = x[i] >> 29; x[i] &= M29; }
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i + 1] += c[i];
    }
    // exact sequential carry propagation (cold paths)
    ZK_HD static void carry_exact(u32* x) {
        for (int i = 0; i < 8; ++i) { x[i + 1] += x[i] >> 29; x[i] &= M29; }
    }
    // portable product (host builds; same column schedule as fe29_asm.inc)
    ZK_HD static Fp29 mul_portable(const Fp29& a, const Fp29& b) {
        u32 q[9];
        Fp29 r;
        u64 acc = 0;
        for (int k = 0; k < 18; ++k) {
            for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); ++i) acc += (u64)a.l[i] * b.l[k - i];
            for (int i = (k > 8 ? k - 8 : 0); i < k && i <= 8; ++i) acc += (u64)q[i] * P::mod29(k - i);
            if (k < 9) { q[k] = ((u32)acc * P::INV29) & M29; acc += (u64)q[k] * P::mod29(0); }
            else if (k < 17) r.l[k - 9] = (u32)acc & M29;
            else r.l[8] = (u32)acc;
            acc >>= 29;
        }
        return r;
    }
    ZK_HD static Fp29 mul2_portable(const Fp29& a, const Fp29& b, const Fp29& c, const Fp29& d) {
        u32 q[9];
        Fp29 r;
        u64 acc = 0;
        for (int k = 0; k < 18; ++k) {
            for (int i = (k > 8 ? k - 8 : 0); i <= (k < 8 ? k : 8); ++i) acc += (u64)a.l[i] * b.l[k - i] + (u64)c.l[i] * d.l[k - i];
            for (int i = (k > 8 ? k - 8 : 0); i < k && i <= 8; ++i) acc += (u64)q[i] * P::mod29(k - i);
            if (k < 9) { q[k] = ((u32)acc * P::INV29) & M29; acc += (u64)q[k] * P::mod29(0); }
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
    ZK_HD static Fp29 sqr(const Fp29& a) { return mul(a, a); }
    ZK_HD static Fp29 mul2(const Fp29& a, const Fp29& b, const Fp29& c, const Fp29& d) {
#if defined(__HIP_DEVICE_COMPILE__)
        Fp29 r;
        mont_mul2_29_asm<P>(r.l, a.l, b.l, c.l, d.l);
        return r;
#else
        return mul2_portable(a, b, c, d);
#endif
    }
    ZK_HD static Fp29 add(const Fp29& a, const Fp29& b) {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) r.l[i] = a.l[i] + b.l[i];
        norm(r.l);
        return r;
    }
    // a - b + K*p, K in {4, 8, 16}: requires b tight and b < K*p
    template <int K>
    ZK_HD static Fp29 sub(const Fp29& a, const Fp29& b) {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            u32 c = K == 4 ? P::subc4(i) : (K == 8 ? P::subc8(i) : P::subc16(i));
            r.l[i] = a.l[i] + c - b.l[i];
        }
        norm(r.l);
        return r;
    }
    // neg ? 64p - a : a, limbwise and WITHOUT a carry sweep: limbs stay < 2^31, which a product tolerates in ONE operand
    // when the other is tight (9 * 2^31 * 2^29.x + 9 * 2^58 < 2^64).  a must be tight and < 64p.
    ZK_HD static Fp29 cneg_loose(const Fp29& a, bool neg) {
        Fp29 r;
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            u32 t = P::subc64(i) - a.l[i];
            r.l[i] = neg ? t : a.l[i];
        }
        return r;
    }
    // x == 0 (mod p) for a tight x < 2^259.  Fast filter on the exact low limb: x = k*p implies k = x0 * p^-1 mod 2^29
    // and k < 64; anything else is non-zero (probability of entering the slow path for a random x: 2^-23).
    ZK_HD bool is_zero_mod_p() const {
        u32 k = (l[0] * P::PINV29) & M29;
        if (k >= 64u) return false;
        return equals_kp(k);
    }
    ZK_HD_NOINLINE bool equals_kp(u32 k) const {
        // compare with k*p limb by limb after bringing both to canonical limbs
        u32 t[9];
        u64 carry = 0;
        for (int i = 0; i < 9; ++i) {
            u64 v = (u64)P::mod29(i) * k + carry;
            t[i] = i < 8 ? (u32)(v & M29) : (u32)v;
            carry = v >> 29;
        }
        u32 x[9];
        for (int i = 0; i < 9; ++i) x[i] = l[i];
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
    // 8x32 (value v < 2^256) -> 9x29 limbs of v * 2^sh  (sh = 5: gnark Montgomery form -> R' form; sh = 0: plain)
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
    // tight 9x29 value x (< 2^259) -> canonical 8x32 of (x * 2^256 * 2^-261) mod p = x / 32: back to gnark's form
    ZK_HD static Fp to32_div32(const Fp29& x) {
        Fp29 c256 = zero();
        c256.l[8] = 1u << 24;  // 2^256
        Fp29 y = mul(x, c256);  // < x/32 + p < 2p
        return y.to32_canonical();
    }
    // fully reduce a tight value < 4p to [0,p) and re-pack to 8x32
    ZK_HD Fp to32_canonical() const {
        u32 x[9];
#pragma unroll
        for (int i = 0; i < 9; ++i) x[i] = l[i];
        carry_exact(x);
        for (int rep = 0; rep < 3; ++rep) {  // conditional subtractions of p
            u32 t[9];
            u32 bw = 0;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                u32 d = x[i] - P::mod29(i) - bw;
                bw = i < 8 ? (d >> 31) : ((int)d < 0 ? 1u : 0u);
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
};

// ---- XYZZ accumulator in 29-bit form and the specialised mixed addition -----------------------------------------
struct XYZZ29 {
    Fp29 x, y, zz, zzz;
    ZK_HD static XYZZ29 inf() { return {Fp29::zero(), Fp29::zero(), Fp29::zero(), Fp29::zero()}; }
    ZK_HD bool is_inf() const { return zz.all_limbs_zero(); }  // ZZ of a live accumulator is a product of non-zero residues
};

// doubling of an affine point (x2, y2 in R' form, values < 32p): cold path, out of line
ZK_HD_NOINLINE XYZZ29 xyzz29_dbl_affine(Fp29 x2, Fp29 y2) {
    Fp29 one = Fp29::one();
    Fp29 x1 = Fp29::mul(x2, one), y1 = Fp29::mul(y2, one);  // bring below 2p
    Fp29 U = Fp29::add(y1, y1);
    Fp29 V = Fp29::sqr(U);
    Fp29 W = Fp29::mul(U, V);
    Fp29 S = Fp29::mul(x1, V);
    Fp29 X2 = Fp29::sqr(x1);
    Fp29 M = Fp29::add(Fp29::add(X2, X2), X2);
    Fp29 X3 = Fp29::sub<8>(Fp29::sqr(M), Fp29::add(S, S));
    Fp29 Y3 = Fp29::mul2(M, Fp29::sub<16>(S, X3), Fp29::sub<4>(Fp29::zero(), y1), W);  // one fused reduction: < 1.4p
    return {X3, Y3, V, W};
}

// acc += (x2, y2): madd-2008-s with value bounds (units of p; R'/p = 169):
//   x2,y2 < 32 | ZZ,ZZZ < 3 | X1 < 9.2, Y1 < 1.4 | U2,S2 < 1.6 | P < 17.6, R < 5.6 | PP < 2.9, PPP < 1.3, Q < 1.2
//   X3 = R^2 - (PPP + 2Q) + 8p < 9.2 | Y3 = [R*(Q - X3 + 16p) + (4p - Y1)*PPP] / R' + p < 1.4   (one fused reduction)
ZK_HD void xyzz29_madd(XYZZ29& acc, const Fp29& x2, const Fp29& y2) {
    if (acc.is_inf()) {
        Fp29 one = Fp29::one();
        acc.x = Fp29::mul(x2, one); acc.y = Fp29::mul(y2, one); acc.zz = one; acc.zzz = one;
        return;
    }
    Fp29 U2 = Fp29::mul(x2, acc.zz);
    Fp29 S2 = Fp29::mul(y2, acc.zzz);
    Fp29 Pd = Fp29::sub<16>(U2, acc.x);
    Fp29 Rd = Fp29::sub<4>(S2, acc.y);
    if (Pd.is_zero_mod_p()) {
        if (Rd.is_zero_mod_p()) acc = xyzz29_dbl_affine(x2, y2);
        else acc = XYZZ29::inf();
        return;
    }
    Fp29 PP = Fp29::sqr(Pd);
    Fp29 PPP = Fp29::mul(Pd, PP);
    Fp29 Q = Fp29::mul(acc.x, PP);
    Fp29 X3 = Fp29::sub<8>(Fp29::sqr(Rd), Fp29::add(PPP, Fp29::add(Q, Q)));
    Fp29 D = Fp29::sub<16>(Q, X3);
    Fp29 E = Fp29::sub<4>(Fp29::zero(), acc.y);
    acc.y = Fp29::mul2(Rd, D, E, PPP);
    acc.x = X3;
    acc.zz = Fp29::mul(acc.zz, PP);
    acc.zzz = Fp29::mul(acc.zzz, PPP);
}

}  // namespace zk
