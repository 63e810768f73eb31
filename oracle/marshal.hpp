// TEST INFRASTRUCTURE (oracle): gnark-crypto's compressed point encoding for BN254, both directions.
// Third-party: github.com/bnb-chain/gnark-crypto v0.14.1-0.20240910145340-609ab3a7eb9b, ecc/bn254/marshal.go
// (G1Affine.Bytes / SetBytes, G2Affine.Bytes / SetBytes) and fp.Element.LexicographicallyLargest,
// E2.LexicographicallyLargest — absent from /root/reference; the reference reaches it through pk.WriteTo
// (src/keygen/main.go:46) and pk.UnsafeReadFrom (src/prover/prover/prover.go:343).  "parity unpinned": the reference
// holds no golden bytes for this encoding; the constants below are the published ones
//   mMask 0b11<<6, mUncompressed 0b00<<6, mCompressedInfinity 0b01<<6, mCompressedSmallest 0b10<<6,
//   mCompressedLargest 0b11<<6; coordinates big-endian; G2 = X.A1 | X.A0
// and the tests pin what can be pinned without gnark: the generator's bytes by hand, decompress(compress(P)) == P,
// and on-curve / range rejection.
#pragma once
#include "bn254.hpp"

namespace orc {

static inline bool fp_lex_largest(const Fp& y) {  // y > (p-1)/2
    u64 c[4], h[4];
    y.to_canon(c);
    for (int i = 0; i < 4; ++i) h[i] = (FpTag::MOD[i] >> 1) | (i < 3 ? FpTag::MOD[i + 1] << 63 : 0);
    return cmp256(c, h) > 0;
}
static inline bool fp2_lex_largest(const Fp2& y) { return y.a1.is_zero() ? fp_lex_largest(y.a0) : fp_lex_largest(y.a1); }

static inline void g1_compress(const G1A& p, uint8_t out[32]) {
    if (p.is_inf()) { memset(out, 0, 32); out[0] = 0x40; return; }
    p.x.to_be_bytes(out);
    out[0] |= fp_lex_largest(p.y) ? 0xC0 : 0x80;
}
static inline void g2_compress(const G2A& p, uint8_t out[64]) {
    if (p.is_inf()) { memset(out, 0, 64); out[0] = 0x40; return; }
    p.x.a1.to_be_bytes(out);
    p.x.a0.to_be_bytes(out + 32);
    out[0] |= fp2_lex_largest(p.y) ? 0xC0 : 0x80;
}

static inline bool fp_sqrt(const Fp& a, Fp* r) {  // p = 3 mod 4
    u64 e[4], t[4], one[4] = {1, 0, 0, 0};
    add256(t, FpTag::MOD, one);
    for (int i = 0; i < 4; ++i) e[i] = (t[i] >> 2) | (i < 3 ? t[i + 1] << 62 : 0);
    Fp s = Fp::pow(a, e, 4);
    *r = s;
    return Fp::sqr(s) == a;
}
static inline bool fp2_sqrt(const Fp2& a, Fp2* r) {
    Fp inv2 = Fp::inv(Fp::from_u64(2));
    if (a.a1.is_zero()) {
        Fp s;
        if (fp_sqrt(a.a0, &s)) { *r = {s, Fp::zero()}; return true; }
        if (fp_sqrt(Fp::neg(a.a0), &s)) { *r = {Fp::zero(), s}; return true; }
        return false;
    }
    Fp n;
    if (!fp_sqrt(Fp::add(Fp::sqr(a.a0), Fp::sqr(a.a1)), &n)) return false;
    Fp d = Fp::mul(Fp::add(a.a0, n), inv2), x0;
    if (!fp_sqrt(d, &x0)) {
        d = Fp::mul(Fp::sub(a.a0, n), inv2);
        if (!fp_sqrt(d, &x0)) return false;
    }
    Fp x1 = Fp::mul(Fp::mul(a.a1, inv2), Fp::inv(x0));
    Fp2 c = {x0, x1};
    if (!(Fp2::sqr(c) == a)) return false;
    *r = c;
    return true;
}
static inline bool fp_from_be_strict(const uint8_t* b, bool mask2, Fp* out) {
    uint8_t t[32];
    memcpy(t, b, 32);
    if (mask2) t[0] &= 0x3f;
    u64 c[4];
    for (int i = 0; i < 4; ++i) {
        c[i] = 0;
        for (int j = 0; j < 8; ++j) c[i] = (c[i] << 8) | t[(3 - i) * 8 + j];
    }
    if (cmp256(c, FpTag::MOD) >= 0) return false;
    *out = Fp::from_canon(c);
    return true;
}
// 0 ok; 1 not a compressed point; 2 coordinate out of range; 3 not on the curve
static inline int g1_decompress(const uint8_t in[32], G1A* out) {
    uint8_t flag = in[0] & 0xC0;
    *out = {Fp::zero(), Fp::zero()};
    if (flag == 0x40) return 0;
    if (flag == 0x00) return 1;
    Fp x, y;
    if (!fp_from_be_strict(in, true, &x)) return 2;
    if (!fp_sqrt(Fp::add(Fp::mul(Fp::sqr(x), x), Fp::from_u64(3)), &y)) return 3;
    if (fp_lex_largest(y) != (flag == 0xC0)) y = Fp::neg(y);
    *out = {x, y};
    return 0;
}
static inline int g2_decompress(const uint8_t in[64], G2A* out) {
    uint8_t flag = in[0] & 0xC0;
    *out = {Fp2::zero(), Fp2::zero()};
    if (flag == 0x40) return 0;
    if (flag == 0x00) return 1;
    Fp2 x, y;
    if (!fp_from_be_strict(in, true, &x.a1) || !fp_from_be_strict(in + 32, false, &x.a0)) return 2;
    if (!fp2_sqrt(Fp2::add(Fp2::mul(Fp2::sqr(x), x), g2_b()), &y)) return 3;
    if (fp2_lex_largest(y) != (flag == 0xC0)) y = Fp2::neg(y);
    *out = {x, y};
    return 0;
}

}  // namespace orc
