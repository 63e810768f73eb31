// Host build (g++) of the product's OWN device/host arithmetic headers (csrc/fe.cuh, csrc/ec.cuh) so the CPU
// test-suite can compare them with the independent oracle without a GPU.  Test code only.
#include "fe.cuh"
#include "ec.cuh"
#include <string.h>
using namespace zk;
extern "C" {
void hm_fp_mul(const Fp* a, const Fp* b, Fp* o, size_t n) { for (size_t i = 0; i < n; ++i) o[i] = Fp::mul(a[i], b[i]); }
void hm_fp_add(const Fp* a, const Fp* b, Fp* o, size_t n) { for (size_t i = 0; i < n; ++i) o[i] = Fp::add(a[i], b[i]); }
void hm_fp_sub(const Fp* a, const Fp* b, Fp* o, size_t n) { for (size_t i = 0; i < n; ++i) o[i] = Fp::sub(a[i], b[i]); }
void hm_fp_inv(const Fp* a, Fp* o, size_t n) { for (size_t i = 0; i < n; ++i) o[i] = Fp::inv(a[i]); }
void hm_fr_mul(const Fr* a, const Fr* b, Fr* o, size_t n) { for (size_t i = 0; i < n; ++i) o[i] = Fr::mul(a[i], b[i]); }
void hm_fr_add(const Fr* a, const Fr* b, Fr* o, size_t n) { for (size_t i = 0; i < n; ++i) o[i] = Fr::add(a[i], b[i]); }
void hm_fr_sub(const Fr* a, const Fr* b, Fr* o, size_t n) { for (size_t i = 0; i < n; ++i) o[i] = Fr::sub(a[i], b[i]); }
void hm_fr_inv(const Fr* a, Fr* o, size_t n) { for (size_t i = 0; i < n; ++i) o[i] = Fr::inv(a[i]); }
void hm_fr_from_mont(const Fr* a, Fr* o, size_t n) { for (size_t i = 0; i < n; ++i) o[i] = Fr::from_mont(a[i]); }
// sum of affine points via XYZZ mixed adds -> affine
void hm_g1_sum(const G1Affine* p, size_t n, G1Affine* out) {
    G1XYZZ acc = G1XYZZ::inf();
    for (size_t i = 0; i < n; ++i) if (!p[i].is_inf()) xyzz_madd<Fp>(acc, p[i].x, p[i].y);
    *out = xyzz_to_affine<Fp>(acc);
}
void hm_g2_sum(const G2Affine* p, size_t n, G2Affine* out) {
    G2XYZZ acc = G2XYZZ::inf();
    for (size_t i = 0; i < n; ++i) if (!p[i].is_inf()) xyzz_madd<Fp2>(acc, p[i].x, p[i].y);
    *out = xyzz_to_affine<Fp2>(acc);
}
// pairwise tree: exercises xyzz_add (general add) and the Jacobian conversion
void hm_g1_sum_tree(const G1Affine* p, size_t n, G1Jac* out) {
    G1XYZZ a = G1XYZZ::inf(), b = G1XYZZ::inf();
    for (size_t i = 0; i < n; ++i) { if (p[i].is_inf()) continue; if (i & 1) xyzz_madd<Fp>(a, p[i].x, p[i].y); else xyzz_madd<Fp>(b, p[i].x, p[i].y); }
    xyzz_add<Fp>(a, b);
    *out = xyzz_to_jacobian<Fp>(a);
}
void hm_g2_sum_tree(const G2Affine* p, size_t n, G2Jac* out) {
    G2XYZZ a = G2XYZZ::inf(), b = G2XYZZ::inf();
    for (size_t i = 0; i < n; ++i) { if (p[i].is_inf()) continue; if (i & 1) xyzz_madd<Fp2>(a, p[i].x, p[i].y); else xyzz_madd<Fp2>(b, p[i].x, p[i].y); }
    xyzz_add<Fp2>(a, b);
    *out = xyzz_to_jacobian<Fp2>(a);
}
void hm_g1_mul(const G1Affine* p, const Fr* k_mont, G1Affine* out) {
    Fr c = Fr::from_mont(*k_mont);
    *out = xyzz_to_affine<Fp>(xyzz_mul_limbs<Fp>(xyzz_from_affine<Fp>(*p), c.v));
}
}

// ---- 9 x 29-bit signed lazy representation (csrc/fe29.cuh), portable path ----
#include "fe29.cuh"
extern "C" {
// out = a*b (both gnark 8x32 Montgomery form) computed THROUGH the 29-bit path: (a*32)*(b*32)/2^261 -> /32 -> canonical
void hm_fp29_mul(const Fp* a, const Fp* b, Fp* o, size_t n) {
    for (size_t i = 0; i < n; ++i) o[i] = Fp29::to32_div32(Fp29::mul(Fp29::from32<5>(a[i]), Fp29::from32<5>(b[i])));
}
void hm_fp29_roundtrip(const Fp* a, Fp* o, size_t n) { for (size_t i = 0; i < n; ++i) o[i] = Fp29::to32_div32(Fp29::mul(Fp29::one(), Fp29::from32<5>(a[i]))); }
// (a - b), (a + b), (-a) through the lazy signed forms
void hm_fp29_addsub(const Fp* a, const Fp* b, Fp* osum, Fp* odiff, Fp* oneg, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        Fp29 x = Fp29::mul(Fp29::one(), Fp29::from32<5>(a[i])), y = Fp29::mul(Fp29::one(), Fp29::from32<5>(b[i]));
        osum[i] = Fp29::to32_div32(Fp29::add_n(x, y));
        odiff[i] = Fp29::to32_div32(Fp29::sub_n(x, y));
        oneg[i] = Fp29::to32_div32(Fp29::neg(x));
    }
}
// a*b - c*d fused (the Y3 shape)
void hm_fp29_mul2(const Fp* a, const Fp* b, const Fp* c, const Fp* d, Fp* o, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        Fp29 one = Fp29::one();
        Fp29 A = Fp29::mul(one, Fp29::from32<5>(a[i])), B = Fp29::mul(one, Fp29::from32<5>(b[i]));
        Fp29 C = Fp29::mul(one, Fp29::from32<5>(c[i])), D = Fp29::mul(one, Fp29::from32<5>(d[i]));
        o[i] = Fp29::to32_div32(Fp29::y3(A, B, C, D));
    }
}
int hm_fp29_is_zero(const Fp* a, const Fp* b) {  // a - b == 0 mod p through sub_n + filter
    Fp29 x = Fp29::mul(Fp29::one(), Fp29::from32<5>(*a)), y = Fp29::mul(Fp29::one(), Fp29::from32<5>(*b));
    return Fp29::sub_n(x, y).is_zero_mod_p() ? 1 : 0;
}
// signed sum of affine points (sign[i] != 0 subtracts) with the 29-bit mixed addition; tracks limb/value bounds
void hm_g1_sum29(const G1Affine* p, const uint8_t* sign, size_t n, G1Affine* out, uint32_t* max_top_limb) {
    XYZZ29 acc = XYZZ29::inf();
    uint32_t mt = 0;
    for (size_t i = 0; i < n; ++i) {
        if (p[i].is_inf()) continue;
        xyzz29_madd<Fp29>(acc, Fp29::from32<5>(p[i].x), Fp29::cneg(Fp29::from32<5>(p[i].y), sign && sign[i]));
        const Fp29* c[4] = {&acc.x, &acc.y, &acc.zz, &acc.zzz};
        for (auto* f : c) {
            int32_t top = (int32_t)f->l[8];
            uint32_t a = (uint32_t)(top < 0 ? -top : top);
            if (a > mt) mt = a;
            for (int k = 0; k < 8; ++k) { int32_t v = (int32_t)f->l[k]; if (v < -8 || v > (1 << 29) + 8) mt = 0xffffffffu; }
        }
    }
    *max_top_limb = mt;
    G1XYZZ o;
    if (acc.is_inf()) o = G1XYZZ::inf();
    else { o.x = Fp29::to32_div32(acc.x); o.y = Fp29::to32_div32(acc.y); o.zz = Fp29::to32_div32(acc.zz); o.zzz = Fp29::to32_div32(acc.zzz); }
    *out = xyzz_to_affine<Fp>(o);
}
// ---- the scalar field on the same limbs (NTT), and the product-free reductions / packing ----
void hm_fr29_mul(const Fr* a, const Fr* b, Fr* o, size_t n) {
    for (size_t i = 0; i < n; ++i) o[i] = Fr29::to32_div32(Fr29::mul(Fr29::from32<5>(a[i]), Fr29::from32<5>(b[i])));
}
// reduce32 on +/- (a * 32): the value mod m must be unchanged, limbs tight, magnitude < 3m; returns 0 on a bound violation
int hm_fe29_reduce32(const Fp* a, const Fr* ar, int negate, Fp* o, Fr* orr, size_t n) {
    int ok = 1;
    for (size_t i = 0; i < n; ++i) {
        Fp29 x = Fp29::reduce32(Fp29::cneg(Fp29::from32<5>(a[i]), negate != 0));
        Fr29 y = Fr29::reduce32(Fr29::cneg(Fr29::from32<5>(ar[i]), negate != 0));
        for (int k = 0; k < 8; ++k) ok &= x.l[k] < (1u << 29) && y.l[k] < (1u << 29);
        ok &= (int32_t)x.l[8] > -(1 << 20) && (int32_t)x.l[8] < (3 << 22) && (int32_t)y.l[8] > -(1 << 20) && (int32_t)y.l[8] < (3 << 22);
        o[i] = Fp29::to32_div32(x); orr[i] = Fr29::to32_div32(y);
    }
    return ok;
}
// reduce32_pos -> pack32 -> from32<0>: the NTT's inter-pass memory form; input = sum of two unreduced a*32 (the largest
// magnitude the NTT feeds it, 64 m) or its negation.  out = 2a (or -2a), and the packed integer must be < 2^256.
int hm_fr29_pack_roundtrip(const Fr* a, int negate, Fr* o, size_t n) {
    int ok = 1;
    for (size_t i = 0; i < n; ++i) {
        Fr29 x = Fr29::from32<5>(a[i]);
        Fr29 s = Fr29::cneg(Fr29::add_l(x, x), negate != 0);
        Fr29 r = Fr29::reduce32_pos(s);
        ok &= (int32_t)r.l[8] >= 0 && r.l[8] < (1u << 24);
        Fr packed = r.pack32();
        Fr29 back = Fr29::from32<0>(packed);
        for (int k = 0; k < 9; ++k) ok &= back.l[k] == r.l[k];
        o[i] = Fr29::to32_div32(back);
    }
    return ok;
}
// general XYZZ + XYZZ in the 29-bit form (xyzz29_add / xyzz29_dbl): the points are first summed in `groups` round-robin
// accumulators with mixed additions, then the accumulators are folded with the general addition; `dup` folds accumulator 0
// in twice (forces the doubling branch), and finally the total is added to its own negation when `cancel` is set.
void hm_g1_sum29_general(const G1Affine* p, size_t n, int groups, int dup, int cancel, G1Affine* out, uint32_t* max_top_limb) {
    XYZZ29 acc[16];
    for (int g = 0; g < groups; ++g) acc[g] = XYZZ29::inf();
    for (size_t i = 0; i < n; ++i) {
        if (p[i].is_inf()) continue;
        xyzz29_madd<Fp29>(acc[i % groups], Fp29::from32<5>(p[i].x), Fp29::from32<5>(p[i].y));
    }
    uint32_t mt = 0;
    auto track = [&](const XYZZ29& a) {
        const Fp29* c[4] = {&a.x, &a.y, &a.zz, &a.zzz};
        for (auto* f : c) {
            int32_t top = (int32_t)f->l[8];
            uint32_t v = (uint32_t)(top < 0 ? -top : top);
            if (v > mt) mt = v;
            for (int k = 0; k < 8; ++k) { int32_t l = (int32_t)f->l[k]; if (l < -8 || l > (1 << 29) + 8) mt = 0xffffffffu; }
        }
    };
    XYZZ29 tot = XYZZ29::inf();
    for (int g = 0; g < groups; ++g) { xyzz29_add<Fp29>(tot, acc[g]); track(tot); }
    if (dup) {  // tot := 2 * tot through the P == Q branch, then a plain general doubling on top
        XYZZ29 copy = tot;
        xyzz29_add<Fp29>(tot, copy); track(tot);
        tot = xyzz29_dbl<Fp29>(tot); track(tot);
    }
    if (cancel) {
        XYZZ29 neg = tot;
        neg.y = Fp29::neg(neg.y);
        xyzz29_add<Fp29>(tot, neg);
    }
    *max_top_limb = mt;
    G1XYZZ o;
    if (tot.is_inf()) o = G1XYZZ::inf();
    else { o.x = Fp29::to32_div32(tot.x); o.y = Fp29::to32_div32(tot.y); o.zz = Fp29::to32_div32(tot.zz); o.zzz = Fp29::to32_div32(tot.zzz); }
    *out = xyzz_to_affine<Fp>(o);
}
}
