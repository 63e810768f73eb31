// BN254 scalar field on the HOST for the solver executor (host/solver_exec.hpp): gnark-crypto's in-memory form — 4 x 64-bit
// little-endian limbs, Montgomery, R = 2^256 — so that coefficient tables and wire vectors are shared with the device and with
// gnark without conversion.  CIOS product on unsigned __int128; nothing here is used by the device path.
#pragma once
#include <cstdint>
#include <cstring>

namespace zkpor_host {

struct FrH {
    uint64_t v[4];
    typedef unsigned __int128 u128;
    static constexpr uint64_t M0 = 0x43e1f593f0000001ULL, M1 = 0x2833e84879b97091ULL, M2 = 0xb85045b68181585dULL, M3 = 0x30644e72e131a029ULL;
    static constexpr uint64_t INV = 0xc2e1f593efffffffULL;   // -r^-1 mod 2^64
    static const uint64_t* mod() { static const uint64_t m[4] = {M0, M1, M2, M3}; return m; }
    static FrH zero() { return FrH{{0, 0, 0, 0}}; }
    static FrH one() { return FrH{{0xac96341c4ffffffbULL, 0x36fc76959f60cd29ULL, 0x666ea36f7879462eULL, 0x0e0a77c19a07df2fULL}}; }   // R mod r
    static FrH r2() { return FrH{{0x1bb8e645ae216da7ULL, 0x53fe3ab1e35c59e3ULL, 0x8c49833d53bb8085ULL, 0x0216d0b17f4e44a5ULL}}; }    // R^2 mod r
    bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
    bool operator==(const FrH& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2] && v[3] == o.v[3]; }
    static bool geq_mod(const uint64_t a[4]) {
        const uint64_t* m = mod();
        for (int i = 3; i >= 0; --i) { if (a[i] != m[i]) return a[i] > m[i]; }
        return true;
    }
    static void sub_mod(uint64_t a[4]) {
        const uint64_t* m = mod();
        u128 bw = 0;
        for (int i = 0; i < 4; ++i) { u128 d = (u128)a[i] - m[i] - (uint64_t)bw; a[i] = (uint64_t)d; bw = (d >> 64) & 1; }
    }
    static FrH add(const FrH& a, const FrH& b) {
        FrH r;
        u128 c = 0;
        for (int i = 0; i < 4; ++i) { c += (u128)a.v[i] + b.v[i]; r.v[i] = (uint64_t)c; c >>= 64; }
        if (c || geq_mod(r.v)) sub_mod(r.v);
        return r;
    }
    static FrH sub(const FrH& a, const FrH& b) {
        FrH r;
        u128 bw = 0;
        for (int i = 0; i < 4; ++i) { u128 d = (u128)a.v[i] - b.v[i] - (uint64_t)bw; r.v[i] = (uint64_t)d; bw = (d >> 64) & 1; }
        if (bw) { const uint64_t* m = mod(); u128 c = 0; for (int i = 0; i < 4; ++i) { c += (u128)r.v[i] + m[i]; r.v[i] = (uint64_t)c; c >>= 64; } }
        return r;
    }
    static FrH neg(const FrH& a) { return a.is_zero() ? a : sub(zero(), a); }
    static FrH mul(const FrH& a, const FrH& b) {
        const uint64_t* m = mod();
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            u128 c = 0;
            for (int j = 0; j < 4; ++j) { c += (u128)a.v[j] * b.v[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
            const uint64_t q = t[0] * INV;
            c = (u128)q * m[0] + t[0]; c >>= 64;
            for (int j = 1; j < 4; ++j) { c += (u128)q * m[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
        }
        FrH r{{t[0], t[1], t[2], t[3]}};
        if (t[4] || geq_mod(r.v)) sub_mod(r.v);
        return r;
    }
    static FrH sqr(const FrH& a) { return mul(a, a); }
    static FrH from_canon(const uint64_t c[4]) { FrH x{{c[0], c[1], c[2], c[3]}}; return mul(x, r2()); }
    static FrH from_u64(uint64_t x) { uint64_t c[4] = {x, 0, 0, 0}; return from_canon(c); }
    void to_canon(uint64_t out[4]) const { FrH o{{1, 0, 0, 0}}; FrH r = mul(*this, o); memcpy(out, r.v, 32); }
    // a^(r-2); inv(0) = 0 (what gnark's field does as well)
    static FrH inv(const FrH& a) {
        if (a.is_zero()) return a;
        const uint64_t* m = mod();
        uint64_t e[4] = {m[0] - 2, m[1], m[2], m[3]};
        FrH r = one(), b = a;
        for (int i = 0; i < 254; ++i) {
            if ((e[i >> 6] >> (i & 63)) & 1) r = mul(r, b);
            b = sqr(b);
        }
        return r;
    }
};

// unsigned 256-bit helpers on canonical values (hints work on integers, as gnark's hint functions work on *big.Int)
struct U256 {
    uint64_t w[4];
    static U256 of(const FrH& a) { U256 r; a.to_canon(r.w); return r; }
    FrH fr() const { return FrH::from_canon(w); }
    bool is_zero() const { return (w[0] | w[1] | w[2] | w[3]) == 0; }
    int bitlen() const { for (int i = 3; i >= 0; --i) if (w[i]) return 64 * i + 64 - __builtin_clzll(w[i]); return 0; }
    bool bit(int i) const { return i < 256 && ((w[i >> 6] >> (i & 63)) & 1); }
    static int cmp(const U256& a, const U256& b) { for (int i = 3; i >= 0; --i) if (a.w[i] != b.w[i]) return a.w[i] < b.w[i] ? -1 : 1; return 0; }
    static U256 sub(const U256& a, const U256& b) {
        U256 r; unsigned __int128 bw = 0;
        for (int i = 0; i < 4; ++i) { unsigned __int128 d = (unsigned __int128)a.w[i] - b.w[i] - (uint64_t)bw; r.w[i] = (uint64_t)d; bw = (d >> 64) & 1; }
        return r;
    }
    U256 shl1() const { U256 r; uint64_t c = 0; for (int i = 0; i < 4; ++i) { r.w[i] = (w[i] << 1) | c; c = w[i] >> 63; } return r; }
    // big.Int.DivMod for non-negative operands: q = a / b, rem = a mod b (b != 0); schoolbook shift-subtract
    // bits [lo, lo + n) of the value, n <= 64
    uint64_t bits(int lo, int n) const {
        if (lo >= 256 || n <= 0) return 0;
        const int wi = lo >> 6, sh = lo & 63;
        uint64_t x = w[wi] >> sh;
        if (sh && wi + 1 < 4) x |= w[wi + 1] << (64 - sh);
        return n >= 64 ? x : (x & (((uint64_t)1 << n) - 1));
    }
    static void divmod(const U256& a, const U256& b, U256* q, U256* rem) {
        if ((b.w[1] | b.w[2] | b.w[3]) == 0) {   // a one-word divisor (every division the circuit makes: prices, the base 100): word-wise long division
            const uint64_t d = b.w[0];
            U256 qq;
            unsigned __int128 r = 0;
            for (int i = 3; i >= 0; --i) { unsigned __int128 cur = (r << 64) | a.w[i]; qq.w[i] = (uint64_t)(cur / d); r = cur % d; }
            *q = qq; *rem = U256{{(uint64_t)r, 0, 0, 0}};
            return;
        }
        U256 qq{{0, 0, 0, 0}}, r{{0, 0, 0, 0}};
        for (int i = a.bitlen() - 1; i >= 0; --i) {
            r = r.shl1();
            if (a.bit(i)) r.w[0] |= 1;
            if (cmp(r, b) >= 0) { r = sub(r, b); qq.w[i >> 6] |= (uint64_t)1 << (i & 63); }
        }
        *q = qq; *rem = r;
    }
};

}  // namespace zkpor_host
