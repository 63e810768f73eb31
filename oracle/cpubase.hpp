// ORACLE — TEST INFRASTRUCTURE ONLY: the CPU BASELINE of bench.py (`cpu_baseline`, kind "port") and nothing else.
//
// A performance-minded CPU port of what groth16.Prove does after the solver (reference call site
// src/prover/prover/prover.go:269), organised the way gnark / gnark-crypto organise it so that the number printed beside the
// GPU figure is a credible stand-in for the reference's own prover on the same host cores (gnark itself cannot be built here:
// no Go toolchain, modules un-vendored — DESIGN.md §4):
//   * field product: "no-carry" CIOS Montgomery on 4 x 64-bit limbs (gnark-crypto field/goff: the spare top bit of the BN254
//     moduli removes the extra carry word), E2 products by Karatsuba (3 base products);
//   * MultiExp (ecc/bn254/multiexp.go): scalars out of Montgomery form -> signed c-bit digits (c = 16 at these sizes),
//     buckets in extended Jacobian coordinates (X, Y, ZZ, ZZZ: g1JacExtended.addMixed, 8M + 2S), one task per (window, chunk of
//     points) spread over all cores, per-window running-sum reduction, Horner over the windows;
//   * FFT (fr/fft): radix-2 DIF / DIT without bit reversal, the upper stages parallel over the butterflies of a stage, the lower
//     stages parallel over cache-sized sub-transforms; coset scaling fused with the neighbouring pass;
//   * computeH (backend/groth16/bn254/prove.go): 3 iFFT (DIF) -> 3 coset FFT (DIT) -> (a b - c) / (g^D - 1) -> coset iFFT (DIF).
// Checked against the plain oracle (algos.hpp) in tests/test_cpubase_cpu.py; it is never the thing the GPU path is compared to.
#pragma once
#include "algos.hpp"
#include <chrono>

namespace orc {
namespace fast {

// ---------------------------------------------------------------------------------------------- field
template <class T>
struct FF {  // same memory layout as Fe<T>
    u64 v[4];
    static inline FF zero() { return FF{{0, 0, 0, 0}}; }
    inline bool is_zero() const { return (v[0] | v[1] | v[2] | v[3]) == 0; }
    inline bool operator==(const FF& o) const { return v[0] == o.v[0] && v[1] == o.v[1] && v[2] == o.v[2] && v[3] == o.v[3]; }
    static inline void cond_sub(FF& r) {
        u64 s[4];
        u64 bw = sub256(s, r.v, T::MOD);
        if (!bw) { r.v[0] = s[0]; r.v[1] = s[1]; r.v[2] = s[2]; r.v[3] = s[3]; }
    }
    static inline FF add(const FF& a, const FF& b) {  // inputs < m < 2^254: no carry out of 256 bits
        FF r;
        add256(r.v, a.v, b.v);
        cond_sub(r);
        return r;
    }
    static inline FF dbl(const FF& a) { return add(a, a); }
    static inline FF sub(const FF& a, const FF& b) {
        FF r;
        if (sub256(r.v, a.v, b.v)) add256(r.v, r.v, T::MOD);
        return r;
    }
    static inline FF neg(const FF& a) {
        if (a.is_zero()) return a;
        FF r;
        sub256(r.v, T::MOD, a.v);
        return r;
    }
    static inline FF mul(const FF& a, const FF& b) {
        u64 t0 = 0, t1 = 0, t2 = 0, t3 = 0;
#define ORC_ROUND(bi)                                                   \
    {                                                                   \
        u128 A = (u128)a.v[0] * (bi) + t0;                              \
        u64 m = (u64)A * T::INV;                                        \
        u128 C = (u128)m * T::MOD[0] + (u64)A;                          \
        A = (u128)a.v[1] * (bi) + t1 + (u64)(A >> 64);                  \
        C = (u128)m * T::MOD[1] + (u64)A + (u64)(C >> 64); t0 = (u64)C; \
        A = (u128)a.v[2] * (bi) + t2 + (u64)(A >> 64);                  \
        C = (u128)m * T::MOD[2] + (u64)A + (u64)(C >> 64); t1 = (u64)C; \
        A = (u128)a.v[3] * (bi) + t3 + (u64)(A >> 64);                  \
        C = (u128)m * T::MOD[3] + (u64)A + (u64)(C >> 64); t2 = (u64)C; \
        t3 = (u64)(C >> 64) + (u64)(A >> 64);                           \
    }
        ORC_ROUND(b.v[0]) ORC_ROUND(b.v[1]) ORC_ROUND(b.v[2]) ORC_ROUND(b.v[3])
#undef ORC_ROUND
        FF r{{t0, t1, t2, t3}};
        cond_sub(r);
        return r;
    }
    static inline FF sqr(const FF& a) { return mul(a, a); }
};
typedef FF<FpTag> FFp;
typedef FF<FrTag> FFr;

struct FFp2 {
    FFp a0, a1;
    static inline FFp2 zero() { return {FFp::zero(), FFp::zero()}; }
    inline bool is_zero() const { return a0.is_zero() && a1.is_zero(); }
    inline bool operator==(const FFp2& o) const { return a0 == o.a0 && a1 == o.a1; }
    static inline FFp2 add(const FFp2& x, const FFp2& y) { return {FFp::add(x.a0, y.a0), FFp::add(x.a1, y.a1)}; }
    static inline FFp2 sub(const FFp2& x, const FFp2& y) { return {FFp::sub(x.a0, y.a0), FFp::sub(x.a1, y.a1)}; }
    static inline FFp2 dbl(const FFp2& x) { return add(x, x); }
    static inline FFp2 neg(const FFp2& x) { return {FFp::neg(x.a0), FFp::neg(x.a1)}; }
    static inline FFp2 mul(const FFp2& x, const FFp2& y) {  // Karatsuba: u^2 = -1
        FFp ac = FFp::mul(x.a0, y.a0), bd = FFp::mul(x.a1, y.a1);
        FFp k = FFp::mul(FFp::add(x.a0, x.a1), FFp::add(y.a0, y.a1));
        return {FFp::sub(ac, bd), FFp::sub(FFp::sub(k, ac), bd)};
    }
    static inline FFp2 sqr(const FFp2& x) {  // (a + b)(a - b), 2ab
        FFp s = FFp::add(x.a0, x.a1), d = FFp::sub(x.a0, x.a1), p = FFp::mul(x.a0, x.a1);
        return {FFp::mul(s, d), FFp::dbl(p)};
    }
};

static inline FFp one_of(FFp*) { Fp o = Fp::one(); FFp r; memcpy(r.v, o.v, 32); return r; }
static inline FFp2 one_of(FFp2*) { return {one_of((FFp*)nullptr), FFp::zero()}; }

// ---------------------------------------------------------------------------------------------- curve (y^2 = x^3 + b, a = 0)
template <class F> struct AffF { F x, y; inline bool is_inf() const { return x.is_zero() && y.is_zero(); } };
template <class F>
struct Ext {  // extended Jacobian: x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2; infinity: ZZ = 0
    F X, Y, ZZ, ZZZ;
    static inline Ext inf() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
    inline bool is_inf() const { return ZZ.is_zero(); }
};
template <class F>
static inline void ext_dbl_affine(Ext<F>& r, const AffF<F>& p) {  // mdbl-2008-s-1
    F U = F::dbl(p.y), V = F::sqr(U), W = F::mul(U, V), S = F::mul(p.x, V);
    F xx = F::sqr(p.x), M = F::add(F::dbl(xx), xx);
    r.X = F::sub(F::sqr(M), F::dbl(S));
    r.Y = F::sub(F::mul(M, F::sub(S, r.X)), F::mul(W, p.y));
    r.ZZ = V; r.ZZZ = W;
}
template <class F>
static inline void ext_dbl(Ext<F>& p) {  // dbl-2008-s-1
    if (p.is_inf()) return;
    F U = F::dbl(p.Y), V = F::sqr(U), W = F::mul(U, V), S = F::mul(p.X, V);
    F xx = F::sqr(p.X), M = F::add(F::dbl(xx), xx);
    F X3 = F::sub(F::sqr(M), F::dbl(S));
    F Y3 = F::sub(F::mul(M, F::sub(S, X3)), F::mul(W, p.Y));
    p.ZZ = F::mul(V, p.ZZ); p.ZZZ = F::mul(W, p.ZZZ);
    p.X = X3; p.Y = Y3;
}
// acc += (x, +-y): g1JacExtended.addMixed / subMixed (madd-2008-s), 8M + 2S
template <class F>
static inline void ext_add_mixed(Ext<F>& a, const AffF<F>& q, bool negate) {
    if (q.is_inf()) return;
    F qy = negate ? F::neg(q.y) : q.y;
    if (a.is_inf()) { a.X = q.x; a.Y = qy; a.ZZ = one_of((F*)nullptr); a.ZZZ = a.ZZ; return; }
    F P = F::sub(F::mul(q.x, a.ZZ), a.X), R = F::sub(F::mul(qy, a.ZZZ), a.Y);
    if (P.is_zero()) {
        if (R.is_zero()) { AffF<F> t{q.x, qy}; ext_dbl_affine(a, t); }
        else a = Ext<F>::inf();
        return;
    }
    F PP = F::sqr(P), PPP = F::mul(P, PP), Q = F::mul(a.X, PP);
    F X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
    a.Y = F::sub(F::mul(R, F::sub(Q, X3)), F::mul(a.Y, PPP));
    a.X = X3;
    a.ZZ = F::mul(a.ZZ, PP); a.ZZZ = F::mul(a.ZZZ, PPP);
}
template <class F>
static inline void ext_add(Ext<F>& a, const Ext<F>& b) {  // add-2008-s, 12M + 2S
    if (b.is_inf()) return;
    if (a.is_inf()) { a = b; return; }
    F U1 = F::mul(a.X, b.ZZ), U2 = F::mul(b.X, a.ZZ), S1 = F::mul(a.Y, b.ZZZ), S2 = F::mul(b.Y, a.ZZZ);
    F P = F::sub(U2, U1), R = F::sub(S2, S1);
    if (P.is_zero()) {
        if (R.is_zero()) ext_dbl(a); else a = Ext<F>::inf();
        return;
    }
    F PP = F::sqr(P), PPP = F::mul(P, PP), Q = F::mul(U1, PP);
    F X3 = F::sub(F::sub(F::sqr(R), PPP), F::dbl(Q));
    a.Y = F::sub(F::mul(R, F::sub(Q, X3)), F::mul(S1, PPP));
    a.X = X3;
    a.ZZ = F::mul(F::mul(a.ZZ, b.ZZ), PP); a.ZZZ = F::mul(F::mul(a.ZZZ, b.ZZZ), PPP);
}

// ---------------------------------------------------------------------------------------------- MultiExp
static inline int threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

// digits[i * W + w] in [-2^(c-1), 2^(c-1)]: signed windows of the canonical scalar, carry into the next window
static inline void signed_digits(const Fr* sc, size_t n, int c, int W, std::vector<int32_t>& out) {
    out.resize(n * (size_t)W);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        u64 k[5];
        sc[i].to_canon(k);
        k[4] = 0;
        int carry = 0;
        for (int w = 0; w < W; ++w) {
            int bit0 = w * c;
            u64 d = k[bit0 / 64] >> (bit0 % 64);
            if (bit0 % 64 + c > 64) d |= k[bit0 / 64 + 1] << (64 - bit0 % 64);
            int64_t v = (int64_t)(d & (((u64)1 << c) - 1)) + carry;
            carry = 0;
            if (v > ((int64_t)1 << (c - 1))) { v -= (int64_t)1 << c; carry = 1; }
            out[i * (size_t)W + w] = (int32_t)v;
        }
    }
}

template <class F, class A>  // A = Aff<Fp> / Aff<Fp2> of the oracle (same layout as AffF<F>)
static Ext<F> multi_exp(const A* pts_in, const Fr* sc, size_t n, int c = 0) {
    static_assert(sizeof(A) == sizeof(AffF<F>), "layout");
    const AffF<F>* pts = (const AffF<F>*)pts_in;
    if (n == 0) return Ext<F>::inf();
    const int nt = threads();
    int chunks = 0;
    if (c == 0) {
        // gnark-crypto's MultiExp picks the window by minimising the additions per task and splits the points recursively until all
        // cores have work, re-picking the window for the split size.  Same idea, explicit: cost of a task = its mixed additions
        // (10 products each) + the running-sum reduction of its 2^(c-1) buckets (2 general additions, 14 products each); makespan =
        // rounds of `nt` tasks.
        double best = 1e300;
        for (int cc = 4; cc <= 16; ++cc) {
            const int Wc = (255 + cc - 1) / cc + ((255 % cc) == 0 ? 1 : 0);
            for (int ch = 1; ch <= 4096; ch *= 2) {
                if (ch > 1 && n / (size_t)ch < 64) break;
                double task = (double)n / ch * 10.0 + (double)((size_t)1 << (cc - 1)) * 28.0;
                double rounds = (double)(((size_t)Wc * ch + nt - 1) / nt);
                double cost = rounds * task + (double)Wc * ch * 14.0;   // + the serial recombination of the task results
                if (cost < best) { best = cost; c = cc; chunks = ch; }
            }
        }
    }
    const int W = (255 + c - 1) / c + ((255 % c) == 0 ? 1 : 0);  // room for the last carry
    if (chunks == 0) {
        chunks = 1;
        while (chunks * W < 2 * nt && n / (size_t)(2 * chunks) >= ((size_t)1 << (c - 1))) chunks *= 2;
    }
    std::vector<int32_t> dig;
    signed_digits(sc, n, c, W, dig);
    const size_t nb = (size_t)1 << (c - 1);
    std::vector<Ext<F>> part((size_t)W * chunks);
#pragma omp parallel for schedule(dynamic, 1)
    for (int task = 0; task < W * chunks; ++task) {
        const int w = task / chunks, ch = task % chunks;
        const size_t lo = n * (size_t)ch / chunks, hi = n * (size_t)(ch + 1) / chunks;
        std::vector<Ext<F>> bucket(nb, Ext<F>::inf());
        for (size_t i = lo; i < hi; ++i) {
            int32_t d = dig[i * (size_t)W + w];
            if (d > 0) ext_add_mixed(bucket[d - 1], pts[i], false);
            else if (d < 0) ext_add_mixed(bucket[-d - 1], pts[i], true);
        }
        Ext<F> run = Ext<F>::inf(), tot = Ext<F>::inf();
        for (size_t b = nb; b-- > 0;) {
            ext_add(run, bucket[b]);
            ext_add(tot, run);
        }
        part[task] = tot;
    }
    Ext<F> acc = Ext<F>::inf();
    for (int w = W - 1; w >= 0; --w) {
        for (int k = 0; k < c; ++k) ext_dbl(acc);
        for (int ch = 0; ch < chunks; ++ch) ext_add(acc, part[(size_t)w * chunks + ch]);
    }
    return acc;
}

// Ext over the fast field -> the oracle's affine point (one inversion, through the oracle's own field)
static inline Aff<Fp> to_oracle_affine(const Ext<FFp>& p) {
    if (p.is_inf()) return {Fp::zero(), Fp::zero()};
    Fp X, Y, ZZ, ZZZ;
    memcpy(X.v, p.X.v, 32); memcpy(Y.v, p.Y.v, 32); memcpy(ZZ.v, p.ZZ.v, 32); memcpy(ZZZ.v, p.ZZZ.v, 32);
    return {Fp::mul(X, Fp::inv(ZZ)), Fp::mul(Y, Fp::inv(ZZZ))};
}
static inline Aff<Fp2> to_oracle_affine(const Ext<FFp2>& p) {
    if (p.is_inf()) return {Fp2::zero(), Fp2::zero()};
    Fp2 X, Y, ZZ, ZZZ;
    memcpy(&X, &p.X, 64); memcpy(&Y, &p.Y, 64); memcpy(&ZZ, &p.ZZ, 64); memcpy(&ZZZ, &p.ZZZ, 64);
    return {Fp2::mul(X, Fp2::inv(ZZ)), Fp2::mul(Y, Fp2::inv(ZZZ))};
}

// ---------------------------------------------------------------------------------------------- FFT
struct FastDomain {
    int k;
    size_t n;
    std::vector<FFr> tw, twi;  // w^i and w^-i, i < n/2
    std::vector<FFr> cs, csi;  // g^i and g^-i / n ... coset powers (g = 5), natural index
    FFr n_inv, den;
    explicit FastDomain(int log2n) : k(log2n), n((size_t)1 << log2n) {
        Domain d0(1);  // only for the constants below
        (void)d0;
        Fr w = fr_root_of_unity_2_28();
        for (int i = k; i < 28; ++i) w = Fr::sqr(w);
        Fr wi = Fr::inv(w), g = Fr::from_u64(5), gi = Fr::inv(g), ninv = Fr::inv(Fr::from_u64((u64)n));
        memcpy(n_inv.v, ninv.v, 32);
        const size_t h = n / 2 ? n / 2 : 1;
        tw.resize(h); twi.resize(h); cs.resize(n); csi.resize(n);
        powers(tw, w); powers(twi, wi); powers(cs, g); powers(csi, gi);
        // den = (g^n - 1)^-1
        Fr gn = g;
        for (int i = 0; i < k; ++i) gn = Fr::sqr(gn);
        Fr dn = Fr::inv(Fr::sub(gn, Fr::one()));
        memcpy(den.v, dn.v, 32);
    }
    static void powers(std::vector<FFr>& out, const Fr& base) {  // out[i] = base^i, parallel by blocks
        const size_t n = out.size();
        const size_t B = 1 << 12;
        const size_t nblk = (n + B - 1) / B;
        std::vector<Fr> start(nblk);
        u64 e[1] = {B};
        Fr step = Fr::pow(base, e, 1);
        Fr x = Fr::one();
        for (size_t b = 0; b < nblk; ++b) { start[b] = x; x = Fr::mul(x, step); }
        FFr fb;
        memcpy(fb.v, base.v, 32);
#pragma omp parallel for schedule(static)
        for (size_t b = 0; b < nblk; ++b) {
            FFr y;
            memcpy(y.v, start[b].v, 32);
            for (size_t i = b * B; i < n && i < (b + 1) * B; ++i) { out[i] = y; y = FFr::mul(y, fb); }
        }
    }
};

static const int LOCAL_LOG = 15;  // 2^15 elements x 32 B = 1 MiB: one sub-transform stays in a core's L2

static inline void dif_stage(FFr* a, size_t n, size_t half, const FFr* tw, size_t stride) {
#pragma omp parallel for schedule(static)
    for (size_t t = 0; t < n / 2; ++t) {
        size_t blk = t / half, j = t % half;
        FFr* p = a + blk * 2 * half + j;
        FFr u = p[0], v = p[half];
        p[0] = FFr::add(u, v);
        p[half] = FFr::mul(FFr::sub(u, v), tw[j * stride]);
    }
}
static inline void dit_stage(FFr* a, size_t n, size_t half, const FFr* tw, size_t stride) {
#pragma omp parallel for schedule(static)
    for (size_t t = 0; t < n / 2; ++t) {
        size_t blk = t / half, j = t % half;
        FFr* p = a + blk * 2 * half + j;
        FFr u = p[0], v = FFr::mul(p[half], tw[j * stride]);
        p[0] = FFr::add(u, v);
        p[half] = FFr::sub(u, v);
    }
}
// natural in -> bit-reversed out
static void dif(FFr* a, int k, const std::vector<FFr>& tw) {
    const size_t n = (size_t)1 << k;
    size_t half = n / 2, stride = 1;
    for (; half >= ((size_t)1 << LOCAL_LOG); half /= 2, stride *= 2) dif_stage(a, n, half, tw.data(), stride);
    if (half == 0) return;
    const size_t blk_len = 2 * half;  // remaining stages are local to blocks of this length
#pragma omp parallel for schedule(static)
    for (size_t blk = 0; blk < n; blk += blk_len) {
        size_t st = stride;
        for (size_t h = half; h >= 1; h /= 2, st *= 2)
            for (size_t b2 = blk; b2 < blk + blk_len; b2 += 2 * h)
                for (size_t j = 0; j < h; ++j) {
                    FFr u = a[b2 + j], v = a[b2 + j + h];
                    a[b2 + j] = FFr::add(u, v);
                    a[b2 + j + h] = FFr::mul(FFr::sub(u, v), tw[j * st]);
                }
    }
}
// bit-reversed in -> natural out
static void dit(FFr* a, int k, const std::vector<FFr>& tw) {
    const size_t n = (size_t)1 << k;
    const size_t local = n < ((size_t)1 << LOCAL_LOG) ? n : ((size_t)1 << LOCAL_LOG);
#pragma omp parallel for schedule(static)
    for (size_t blk = 0; blk < n; blk += local) {
        size_t st = n / 2;
        for (size_t h = 1; h < local; h *= 2, st /= 2)
            for (size_t b2 = blk; b2 < blk + local; b2 += 2 * h)
                for (size_t j = 0; j < h; ++j) {
                    FFr u = a[b2 + j], v = FFr::mul(a[b2 + j + h], tw[j * st]);
                    a[b2 + j] = FFr::add(u, v);
                    a[b2 + j + h] = FFr::sub(u, v);
                }
    }
    size_t half = local, stride = n / 2 / local;
    for (; half < n; half *= 2, stride /= 2) dit_stage(a, n, half, tw.data(), stride);
}

// computeH of gnark's prove.go on zero-padded a, b, c of n = 2^k elements; h is left in a, bit-reversed order
static void compute_h(const FastDomain& d, FFr* a, FFr* b, FFr* c) {
    const size_t n = d.n;
    const int k = d.k;
    FFr* v[3] = {a, b, c};
    for (int t = 0; t < 3; ++t) {
        dif(v[t], k, d.twi);  // FFTInverse(DIF): evaluations -> coefficients, bit-reversed; the 1/n is folded into the coset scaling
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i) {  // position i holds coefficient bitrev(i): multiply by g^bitrev(i) / n
            size_t r = bitrev(i, k);
            v[t][i] = FFr::mul(v[t][i], FFr::mul(d.cs[r], d.n_inv));
        }
        dit(v[t], k, d.tw);  // FFT(DIT, OnCoset): coefficients (bit-reversed) -> evaluations on g<w>, natural order
    }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) a[i] = FFr::mul(FFr::sub(FFr::mul(a[i], b[i]), c[i]), d.den);
    dif(a, k, d.twi);  // FFTInverse(DIF, OnCoset)
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) {
        size_t r = bitrev(i, k);
        a[i] = FFr::mul(a[i], FFr::mul(d.csi[r], d.n_inv));
    }
}

struct TailTimes { double fft_s, msm_g1_s, msm_g2_s, commit_s; };

// one prove tail's worth of CPU work at domain 2^k: computeH + 4 G1 MultiExps + 1 G2 MultiExp of n points + 2 commitment MultiExps
// of n_commit points, on all cores.  Outputs only what keeps the compiler honest.
static TailTimes prove_tail_work(int k, const Aff<Fp>* g1, const Aff<Fp2>* g2, const Fr* w, Fr* a, Fr* b, Fr* c, size_t n_commit,
                                 Aff<Fp>* g1_out, Aff<Fp2>* g2_out) {
    const size_t n = (size_t)1 << k;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](auto t0, auto t1) { return std::chrono::duration<double>(t1 - t0).count(); };
    TailTimes tt{};
    auto t0 = now();
    {
        FastDomain d(k);
        t0 = now();  // gnark's domain (twiddles) is part of the key, precomputed at load time
        compute_h(d, (FFr*)a, (FFr*)b, (FFr*)c);
    }
    auto t1 = now();
    tt.fft_s = secs(t0, t1);
    Ext<FFp> acc = multi_exp<FFp>(g1, w, n);            // A
    ext_add(acc, multi_exp<FFp>(g1, w, n));              // B1
    ext_add(acc, multi_exp<FFp>(g1, w, n));              // K
    ext_add(acc, multi_exp<FFp>(g1, a, n - 1));          // Z . h
    auto t2 = now();
    tt.msm_g1_s = secs(t1, t2);
    Ext<FFp2> acc2 = multi_exp<FFp2>(g2, w, n);          // B2
    auto t3 = now();
    tt.msm_g2_s = secs(t2, t3);
    if (n_commit) {
        ext_add(acc, multi_exp<FFp>(g1, w, n_commit));
        ext_add(acc, multi_exp<FFp>(g1, w, n_commit));
    }
    tt.commit_s = secs(t3, now());
    *g1_out = to_oracle_affine(acc);
    *g2_out = to_oracle_affine(acc2);
    return tt;
}

}  // namespace fast
}  // namespace orc
