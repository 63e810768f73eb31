// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path: only tests/, __graft_entry__.smoke()
// and bench.py's cpu_baseline leg may use anything under oracle/.
//
// CPU restatement of the BN254 arithmetic the reference reaches through gnark-crypto
// (github.com/bnb-chain/gnark-crypto v0.14.1-0.20240910145340-609ab3a7eb9b, pinned at
// /root/reference/go.mod:57-60; NOT vendored under /root/reference, so the published algorithms are
// restated here): ecc/bn254/fp, ecc/bn254/fr (4x64-bit little-endian limbs, Montgomery form, R = 2^256),
// ecc/bn254/internal/fptower E2, ecc/bn254 G1Affine/G1Jac/G2Affine/G2Jac.
// Reference call sites: src/prover/prover/prover.go:269 (groth16.Prove), src/utils/utils.go:744-750.
//
// Independent of the product's 8x32-bit-limb device arithmetic (different limb width, different
// reduction schedule) so that agreement between the two is meaningful.
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>
#include <string>

namespace orc {

typedef unsigned __int128 u128;
typedef uint64_t u64;

struct U256 {
    u64 v[4];
};

static inline int cmp256(const u64* a, const u64* b) {
    for (int i = 3; i >= 0; --i) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}
static inline u64 add256(u64* r, const u64* a, const u64* b) {
    u128 c = 0;
    for (int i = 0; i < 4; ++i) {
        c += (u128)a[i] + b[i];
        r[i] = (u64)c;
        c >>= 64;
    }
    return (u64)c;
}
static inline u64 sub256(u64* r, const u64* a, const u64* b) {
    u64 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (u64)d;
        borrow = (u64)(d >> 64) & 1;
    }
    return borrow;
}

// Tag types carry the modulus; all derived constants are computed at start-up and self-checked in
// oracle_selftest() against the values listed in SURVEY.md §8(c).
struct FpTag {
    static constexpr u64 MOD[4] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL,
                                   0x30644e72e131a029ULL};
    static constexpr u64 INV = 0x87d20782e4866389ULL;  // -p^-1 mod 2^64
};
struct FrTag {
    static constexpr u64 MOD[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL,
                                   0x30644e72e131a029ULL};
    static constexpr u64 INV = 0xc2e1f593efffffffULL;  // -r^-1 mod 2^64
};

template <class T>
struct Fe {
    u64 v[4];  // Montgomery form, little-endian limbs (gnark-crypto in-memory layout)

    static const Fe& R2() {  // 2^512 mod m, by 512 modular doublings of 1
        static Fe r2 = [] {
            Fe x;
            x.v[0] = 1; x.v[1] = x.v[2] = x.v[3] = 0;
            for (int i = 0; i < 512; ++i) x = raw_add(x, x);
            return x;
        }();
        return r2;
    }
    static Fe raw_add(const Fe& a, const Fe& b) {
        Fe r;
        u64 c = add256(r.v, a.v, b.v);
        u64 t[4];
        u64 bw = sub256(t, r.v, T::MOD);
        if (c || !bw) memcpy(r.v, t, 32);
        return r;
    }
    static Fe zero() { Fe r; memset(r.v, 0, 32); return r; }
    static Fe one() { return from_u64(1); }
    static Fe from_u64(u64 x) {
        Fe r; r.v[0] = x; r.v[1] = r.v[2] = r.v[3] = 0;
        return mul(r, R2());
    }
    // canonical little-endian limbs (must be < m) -> Montgomery
    static Fe from_canon(const u64* c) {
        Fe r; memcpy(r.v, c, 32);
        return mul(r, R2());
    }
    void to_canon(u64* out) const {
        Fe o; o.v[0] = 1; o.v[1] = o.v[2] = o.v[3] = 0;
        Fe r = mul(*this, o);
        memcpy(out, r.v, 32);
    }
    // big-endian 32 bytes, reduced mod m (gnark SetBytes semantics for len<=32: value mod m)
    static Fe from_be_bytes(const uint8_t* b, size_t len) {
        // general length: Horner over bytes (slow path fine for an oracle)
        Fe acc = zero();
        Fe k256 = from_u64(256);
        for (size_t i = 0; i < len; ++i) acc = add(mul(acc, k256), from_u64(b[i]));
        return acc;
    }
    void to_be_bytes(uint8_t* out) const {
        u64 c[4];
        to_canon(c);
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 8; ++j) out[31 - (i * 8 + j)] = (uint8_t)(c[i] >> (8 * j));
    }
    static Fe add(const Fe& a, const Fe& b) { return raw_add(a, b); }
    static Fe sub(const Fe& a, const Fe& b) {
        Fe r;
        if (sub256(r.v, a.v, b.v)) add256(r.v, r.v, T::MOD);
        return r;
    }
    static Fe neg(const Fe& a) {
        if (a.is_zero()) return a;
        Fe r; sub256(r.v, T::MOD, a.v);
        return r;
    }
    static Fe dbl(const Fe& a) { return raw_add(a, a); }
    // CIOS Montgomery product, 64-bit limbs with 128-bit intermediates
    static Fe mul(const Fe& a, const Fe& b) {
        u64 t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; ++i) {
            u128 c = 0;
            for (int j = 0; j < 4; ++j) {
                c += (u128)a.v[j] * b.v[i] + t[j];
                t[j] = (u64)c;
                c >>= 64;
            }
            c += t[4];
            t[4] = (u64)c;
            t[5] = (u64)(c >> 64);
            u64 m = t[0] * T::INV;
            c = (u128)m * T::MOD[0] + t[0];
            c >>= 64;
            for (int j = 1; j < 4; ++j) {
                c += (u128)m * T::MOD[j] + t[j];
                t[j - 1] = (u64)c;
                c >>= 64;
            }
            c += t[4];
            t[3] = (u64)c;
            t[4] = t[5] + (u64)(c >> 64);
        }
        Fe r;
        memcpy(r.v, t, 32);
        u64 s[4];
        u64 bw = sub256(s, r.v, T::MOD);
        if (t[4] || !bw) memcpy(r.v, s, 32);
        return r;
    }
    static Fe sqr(const Fe& a) { return mul(a, a); }
    static Fe pow(const Fe& a, const u64* e, int nlimbs) {
        Fe r = one();
        for (int i = nlimbs * 64 - 1; i >= 0; --i) {
            r = sqr(r);
            if ((e[i / 64] >> (i % 64)) & 1) r = mul(r, a);
        }
        return r;
    }
    static Fe pow_u64(const Fe& a, u64 e) { return pow(a, &e, 1); }
    static Fe inv(const Fe& a) {  // Fermat; inv(0) = 0 (gnark-crypto Inverse convention)
        u64 e[4];
        u64 two[4] = {2, 0, 0, 0};
        sub256(e, T::MOD, two);
        return pow(a, e, 4);
    }
    bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
    bool operator==(const Fe& o) const { return memcmp(v, o.v, 32) == 0; }
    bool operator!=(const Fe& o) const { return !(*this == o); }
};

typedef Fe<FpTag> Fp;
typedef Fe<FrTag> Fr;

// ---------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1)
struct Fp2 {
    Fp a0, a1;
    static Fp2 zero() { return {Fp::zero(), Fp::zero()}; }
    static Fp2 one() { return {Fp::one(), Fp::zero()}; }
    static Fp2 add(const Fp2& x, const Fp2& y) { return {Fp::add(x.a0, y.a0), Fp::add(x.a1, y.a1)}; }
    static Fp2 sub(const Fp2& x, const Fp2& y) { return {Fp::sub(x.a0, y.a0), Fp::sub(x.a1, y.a1)}; }
    static Fp2 neg(const Fp2& x) { return {Fp::neg(x.a0), Fp::neg(x.a1)}; }
    static Fp2 dbl(const Fp2& x) { return add(x, x); }
    static Fp2 mul(const Fp2& x, const Fp2& y) {  // schoolbook, 4 products (oracle: clarity over speed)
        Fp ac = Fp::mul(x.a0, y.a0), bd = Fp::mul(x.a1, y.a1);
        Fp ad = Fp::mul(x.a0, y.a1), bc = Fp::mul(x.a1, y.a0);
        return {Fp::sub(ac, bd), Fp::add(ad, bc)};
    }
    static Fp2 sqr(const Fp2& x) { return mul(x, x); }
    static Fp2 inv(const Fp2& x) {
        Fp n = Fp::add(Fp::sqr(x.a0), Fp::sqr(x.a1));
        Fp ni = Fp::inv(n);
        return {Fp::mul(x.a0, ni), Fp::neg(Fp::mul(x.a1, ni))};
    }
    bool is_zero() const { return a0.is_zero() && a1.is_zero(); }
    bool operator==(const Fp2& o) const { return a0 == o.a0 && a1 == o.a1; }
    bool operator!=(const Fp2& o) const { return !(*this == o); }
};

// ---------------------------------------------------------------- short-Weierstrass groups, a = 0
// Generic over the coordinate field F (Fp for G1, Fp2 for G2). Affine infinity is encoded as (0,0)
// (gnark-crypto G1Affine/G2Affine convention); Jacobian infinity is Z = 0.
template <class F>
struct Aff {
    F x, y;
    bool is_inf() const { return x.is_zero() && y.is_zero(); }
    bool operator==(const Aff& o) const { return x == o.x && y == o.y; }
};
template <class F>
struct Jac {
    F x, y, z;
    static Jac inf() { return {F::one(), F::one(), F::zero()}; }
    bool is_inf() const { return z.is_zero(); }
};

template <class F>
static Jac<F> to_jac(const Aff<F>& p) {
    if (p.is_inf()) return Jac<F>::inf();
    return {p.x, p.y, F::one()};
}
template <class F>
static Aff<F> to_aff(const Jac<F>& p) {
    if (p.is_inf()) return {F::zero(), F::zero()};
    F zi = F::inv(p.z);
    F zi2 = F::sqr(zi);
    return {F::mul(p.x, zi2), F::mul(p.y, F::mul(zi2, zi))};
}
template <class F>
static Jac<F> jdbl(const Jac<F>& p) {  // dbl-2009-l
    if (p.is_inf()) return p;
    F A = F::sqr(p.x), B = F::sqr(p.y), C = F::sqr(B);
    F D = F::dbl(F::sub(F::sub(F::sqr(F::add(p.x, B)), A), C));
    F E = F::add(F::dbl(A), A);
    F Fq = F::sqr(E);
    F X3 = F::sub(Fq, F::dbl(D));
    F C8 = F::dbl(F::dbl(F::dbl(C)));
    F Y3 = F::sub(F::mul(E, F::sub(D, X3)), C8);
    F Z3 = F::dbl(F::mul(p.y, p.z));
    return {X3, Y3, Z3};
}
template <class F>
static Jac<F> jadd(const Jac<F>& p, const Jac<F>& q) {  // add-2007-bl with the exceptional cases
    if (p.is_inf()) return q;
    if (q.is_inf()) return p;
    F Z1Z1 = F::sqr(p.z), Z2Z2 = F::sqr(q.z);
    F U1 = F::mul(p.x, Z2Z2), U2 = F::mul(q.x, Z1Z1);
    F S1 = F::mul(F::mul(p.y, q.z), Z2Z2), S2 = F::mul(F::mul(q.y, p.z), Z1Z1);
    if (U1 == U2) {
        if (S1 == S2) return jdbl(p);
        return Jac<F>::inf();
    }
    F H = F::sub(U2, U1);
    F I = F::sqr(F::dbl(H));
    F J = F::mul(H, I);
    F r = F::dbl(F::sub(S2, S1));
    F V = F::mul(U1, I);
    F X3 = F::sub(F::sub(F::sqr(r), J), F::dbl(V));
    F Y3 = F::sub(F::mul(r, F::sub(V, X3)), F::dbl(F::mul(S1, J)));
    F Z3 = F::mul(F::sub(F::sub(F::sqr(F::add(p.z, q.z)), Z1Z1), Z2Z2), H);
    return {X3, Y3, Z3};
}
template <class F>
static Jac<F> jadd_aff(const Jac<F>& p, const Aff<F>& q) {
    return jadd(p, to_jac(q));
}
template <class F>
static Jac<F> jneg(const Jac<F>& p) { return {p.x, F::neg(p.y), p.z}; }
template <class F>
static Aff<F> aneg(const Aff<F>& p) { return {p.x, F::neg(p.y)}; }

// scalar given as canonical (non-Montgomery) little-endian limbs
template <class F>
static Jac<F> jmul(const Jac<F>& p, const u64* k, int nlimbs = 4) {
    Jac<F> r = Jac<F>::inf();
    for (int i = nlimbs * 64 - 1; i >= 0; --i) {
        r = jdbl(r);
        if ((k[i / 64] >> (i % 64)) & 1) r = jadd(r, p);
    }
    return r;
}
template <class F>
static Jac<F> jmul_fr(const Jac<F>& p, const Fr& k) {
    u64 c[4];
    k.to_canon(c);
    return jmul(p, c);
}

typedef Aff<Fp> G1A;
typedef Jac<Fp> G1J;
typedef Aff<Fp2> G2A;
typedef Jac<Fp2> G2J;

static inline G1A g1_gen() { return {Fp::from_u64(1), Fp::from_u64(2)}; }
static inline G2A g2_gen() {
    // standard BN254 (alt_bn128) G2 generator, as in gnark-crypto bn254.go / EIP-197
    static const u64 x0[4] = {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL};
    static const u64 x1[4] = {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL};
    static const u64 y0[4] = {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL};
    static const u64 y1[4] = {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL};
    return {{Fp::from_canon(x0), Fp::from_canon(x1)}, {Fp::from_canon(y0), Fp::from_canon(y1)}};
}
static inline bool g1_on_curve(const G1A& p) {
    if (p.is_inf()) return true;
    Fp rhs = Fp::add(Fp::mul(Fp::sqr(p.x), p.x), Fp::from_u64(3));
    return Fp::sqr(p.y) == rhs;
}
static inline Fp2 g2_b() {  // b' = 3/(9+u)
    Fp2 d = {Fp::from_u64(9), Fp::from_u64(1)};
    Fp2 three = {Fp::from_u64(3), Fp::zero()};
    return Fp2::mul(three, Fp2::inv(d));
}
static inline bool g2_on_curve(const G2A& p) {
    if (p.is_inf()) return true;
    Fp2 rhs = Fp2::add(Fp2::mul(Fp2::sqr(p.x), p.x), g2_b());
    return Fp2::sqr(p.y) == rhs;
}

// deterministic PRNG shared by oracle fixtures (SplitMix64)
struct SplitMix {
    u64 s;
    explicit SplitMix(u64 seed) : s(seed) {}
    u64 next() {
        u64 z = (s += 0x9e3779b97f4a7c15ULL);
        z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
        z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
        return z ^ (z >> 31);
    }
    Fr fr() {  // ~uniform Fr: 256 random bits reduced by Montgomery product with R2 (value*R mod r)
        Fr x;
        for (int i = 0; i < 4; ++i) x.v[i] = next();
        x.v[3] &= 0x3fffffffffffffffULL;  // < 2^254 < 2r, then conditional subtract
        u64 t[4];
        if (!sub256(t, x.v, FrTag::MOD)) memcpy(x.v, t, 32);
        return x;  // interpreted as a Montgomery-form element (uniform either way)
    }
};

}  // namespace orc
