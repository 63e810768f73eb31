// ORACLE — TEST INFRASTRUCTURE ONLY (see bn254.hpp header).
//
// CPU restatement of the algorithms on the hot path of groth16.Prove (reference call site
// src/prover/prover/prover.go:269; algorithm in bnb-chain/gnark v0.10.1-0.20240910145009-4b5261061f04
// backend/groth16/bn254/prove.go and bnb-chain/gnark-crypto ecc/bn254/{multiexp.go,fr/fft/fft.go,
// fr/pedersen/pedersen.go}, both un-vendored — published algorithms restated).
// PARITY STATUS: MSM / NTT / proof bytes are "parity unpinned" at the bit level against gnark itself
// (the reference holds no golden vectors for them and proofs are randomized); they are pinned here
// by algebraic identities and by a trapdoor-known Groth16 setup checked in the exponent
// (groth16_check_in_exponent).  Poseidon IS pinned by reference data (see poseidon.hpp).
#pragma once
#include "bn254.hpp"
#include "pairing.hpp"
#include <cassert>
#include <cstdio>
#include <algorithm>

namespace orc {

// ------------------------------------------------------------------------------------------ MSM
// result = sum_i scalars[i] * points[i]; scalars are Montgomery-form Fr (as gnark holds them).
template <class F>
static Jac<F> msm_naive(const Aff<F>* pts, const Fr* sc, size_t n) {
    Jac<F> acc = Jac<F>::inf();
    for (size_t i = 0; i < n; ++i) {
        if (pts[i].is_inf()) continue;
        acc = jadd(acc, jmul_fr(to_jac(pts[i]), sc[i]));
    }
    return acc;
}

// Bucket method (Pippenger), unsigned c-bit windows, Jacobian buckets; restates the structure of
// gnark-crypto MultiExp (multiexp.go: window decomposition -> bucket accumulation -> per-window
// running-sum reduction -> Horner over windows) without its signed-digit / batch-affine refinements
// (those change speed, not the group element).
template <class F>
static Jac<F> msm_pippenger(const Aff<F>* pts, const Fr* sc, size_t n, int c = 0) {
    if (n == 0) return Jac<F>::inf();
    if (c == 0) {
        c = 4;
        while ((1ull << (c + 3)) < n && c < 16) ++c;
    }
    const int W = (254 + c - 1) / c;
    std::vector<U256> canon(n);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i) sc[i].to_canon(canon[i].v);
    // tasks = (window, chunk of points): every task owns a private bucket array, chunk results are added per window
    int nthreads = 1;
#ifdef _OPENMP
    nthreads = omp_get_max_threads();
#endif
    int chunks = 1;
    while (chunks * W < 2 * nthreads && (n / (size_t)(2 * chunks)) >= ((size_t)1 << c)) chunks *= 2;
    std::vector<Jac<F>> part((size_t)W * chunks);
#pragma omp parallel for schedule(dynamic, 1)
    for (int task = 0; task < W * chunks; ++task) {
        const int w = task / chunks, ch = task % chunks;
        const size_t lo = n * (size_t)ch / chunks, hi = n * (size_t)(ch + 1) / chunks;
        std::vector<Jac<F>> bucket((size_t)1 << c, Jac<F>::inf());
        const int bit0 = w * c;
        for (size_t i = lo; i < hi; ++i) {
            if (pts[i].is_inf()) continue;
            u64 d = canon[i].v[bit0 / 64] >> (bit0 % 64);
            if (bit0 % 64 + c > 64 && bit0 / 64 + 1 < 4) d |= canon[i].v[bit0 / 64 + 1] << (64 - bit0 % 64);
            d &= ((u64)1 << c) - 1;
            if (d) bucket[d] = jadd_aff(bucket[d], pts[i]);
        }
        Jac<F> run = Jac<F>::inf(), tot = Jac<F>::inf();
        for (size_t b = ((size_t)1 << c) - 1; b >= 1; --b) {
            run = jadd(run, bucket[b]);
            tot = jadd(tot, run);
        }
        part[task] = tot;
    }
    std::vector<Jac<F>> wsum(W, Jac<F>::inf());
    for (int w = 0; w < W; ++w)
        for (int ch = 0; ch < chunks; ++ch) wsum[w] = jadd(wsum[w], part[(size_t)w * chunks + ch]);
    Jac<F> acc = Jac<F>::inf();
    for (int w = W - 1; w >= 0; --w) {
        for (int k = 0; k < c; ++k) acc = jdbl(acc);
        acc = jadd(acc, wsum[w]);
    }
    return acc;
}

// fixed-base scalar multiplication table (8-bit windows) used to build trapdoor-known keys quickly
template <class F>
struct FixedBase {
    std::vector<Aff<F>> tab;  // [32][256]
    explicit FixedBase(const Aff<F>& g) : tab(32 * 256) {
        Jac<F> base = to_jac(g);
        for (int w = 0; w < 32; ++w) {
            Jac<F> acc = Jac<F>::inf();
            tab[w * 256] = {F::zero(), F::zero()};
            for (int d = 1; d < 256; ++d) {
                acc = jadd(acc, base);
                tab[w * 256 + d] = to_aff(acc);
            }
            base = jadd(acc, base);  // 256 * base
        }
    }
    Jac<F> mul(const Fr& k) const {
        u64 c[4];
        k.to_canon(c);
        Jac<F> acc = Jac<F>::inf();
        for (int w = 0; w < 32; ++w) {
            unsigned d = (unsigned)(c[w / 8] >> (8 * (w % 8))) & 0xff;
            if (d) acc = jadd_aff(acc, tab[w * 256 + d]);
        }
        return acc;
    }
    Aff<F> mul_aff(const Fr& k) const { return to_aff(mul(k)); }
};

// ------------------------------------------------------------------------------------------ NTT
// Restates gnark-crypto fr/fft: Domain (generator of the 2^k subgroup from the 2^28-th root of unity,
// coset shift = FrMultiplicativeGen = 5), difFFT / ditFFT, FFT/FFTInverse with OnCoset.
static inline Fr fr_root_of_unity_2_28() {
    // 5^((r-1)/2^28): value listed in SURVEY.md §8(c); recomputed here and checked in the self-test
    u64 e[4];
    u64 one[4] = {1, 0, 0, 0};
    sub256(e, FrTag::MOD, one);
    // e >>= 28
    for (int i = 0; i < 4; ++i) e[i] = (e[i] >> 28) | (i < 3 ? (e[i + 1] << 36) : 0);
    return Fr::pow(Fr::from_u64(5), e, 4);
}

struct Domain {
    int log2n;
    size_t n;
    Fr gen, gen_inv, n_inv, coset, coset_inv;
    std::vector<Fr> tw, tw_inv;  // w^i, i < n/2
    explicit Domain(int k) : log2n(k), n((size_t)1 << k) {
        Fr w = fr_root_of_unity_2_28();
        for (int i = k; i < 28; ++i) w = Fr::sqr(w);
        gen = w;
        gen_inv = Fr::inv(w);
        n_inv = Fr::inv(Fr::from_u64((u64)n));
        coset = Fr::from_u64(5);
        coset_inv = Fr::inv(coset);
        tw.resize(n / 2 ? n / 2 : 1);
        tw_inv.resize(tw.size());
        Fr a = Fr::one(), b = Fr::one();
        for (size_t i = 0; i < tw.size(); ++i) {
            tw[i] = a; tw_inv[i] = b;
            a = Fr::mul(a, gen); b = Fr::mul(b, gen_inv);
        }
    }
};

static inline size_t bitrev(size_t i, int k) {
    size_t r = 0;
    for (int b = 0; b < k; ++b) r |= ((i >> b) & 1) << (k - 1 - b);
    return r;
}
static inline void bit_reverse(Fr* a, int k) {
    size_t n = (size_t)1 << k;
    for (size_t i = 0; i < n; ++i) {
        size_t j = bitrev(i, k);
        if (i < j) std::swap(a[i], a[j]);
    }
}
// decimation in frequency: natural-order input -> bit-reversed output
static inline void dif_fft(Fr* a, int k, const std::vector<Fr>& tw) {
    size_t n = (size_t)1 << k;
    for (size_t half = n / 2, stride = 1; half >= 1; half /= 2, stride *= 2) {
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t blk = 0; blk < n; blk += 2 * half) {
            for (size_t j = 0; j < half; ++j) {
                Fr u = a[blk + j], v = a[blk + j + half];
                a[blk + j] = Fr::add(u, v);
                a[blk + j + half] = Fr::mul(Fr::sub(u, v), tw[j * stride]);
            }
        }
    }
}
// decimation in time: bit-reversed input -> natural-order output
static inline void dit_fft(Fr* a, int k, const std::vector<Fr>& tw) {
    size_t n = (size_t)1 << k;
    for (size_t half = 1, stride = n / 2; half < n; half *= 2, stride /= 2) {
#pragma omp parallel for schedule(static) if (n >= 4096)
        for (size_t blk = 0; blk < n; blk += 2 * half) {
            for (size_t j = 0; j < half; ++j) {
                Fr u = a[blk + j], v = Fr::mul(a[blk + j + half], tw[j * stride]);
                a[blk + j] = Fr::add(u, v);
                a[blk + j + half] = Fr::sub(u, v);
            }
        }
    }
}
enum Decimation { DIT = 0, DIF = 1 };

static inline void fft_forward(const Domain& d, Fr* a, Decimation dec, bool on_coset) {
    if (on_coset) {
        // a[i] *= g^i (natural index); for DIT the input is bit-reversed so index through bitrev
        std::vector<Fr> pw(d.n);
        Fr x = Fr::one();
        for (size_t i = 0; i < d.n; ++i) { pw[i] = x; x = Fr::mul(x, d.coset); }
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < d.n; ++i) {
            size_t idx = dec == DIT ? bitrev(i, d.log2n) : i;
            a[i] = Fr::mul(a[i], pw[idx]);
        }
    }
    if (dec == DIF) dif_fft(a, d.log2n, d.tw); else dit_fft(a, d.log2n, d.tw);
}
static inline void fft_inverse(const Domain& d, Fr* a, Decimation dec, bool on_coset) {
    if (dec == DIF) dif_fft(a, d.log2n, d.tw_inv); else dit_fft(a, d.log2n, d.tw_inv);
    if (!on_coset) {
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < d.n; ++i) a[i] = Fr::mul(a[i], d.n_inv);
        return;
    }
    std::vector<Fr> pw(d.n);
    Fr x = d.n_inv;
    for (size_t i = 0; i < d.n; ++i) { pw[i] = x; x = Fr::mul(x, d.coset_inv); }
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < d.n; ++i) {
        size_t idx = dec == DIF ? bitrev(i, d.log2n) : i;  // DIF output is bit-reversed
        a[i] = Fr::mul(a[i], pw[idx]);
    }
}

// computeH (gnark backend/groth16/bn254/prove.go): a,b,c are the constraint evaluations (length
// n_cons <= D, zero-padded to D).  Returns h in BIT-REVERSED order: the last transform is a DIF inverse
// and gnark >= 0.9 does not undo it — its setup stores pk.G1.Z bit-reversed instead (3P-recalled; the
// product exposes the order as a pk flag, see include/zkpor.h ZKPOR_Z_ORDER_*).
static inline std::vector<Fr> compute_h(const Domain& d, const Fr* a_in, const Fr* b_in, const Fr* c_in,
                                        size_t n_cons) {
    std::vector<Fr> a(d.n, Fr::zero()), b(d.n, Fr::zero()), c(d.n, Fr::zero());
    std::copy(a_in, a_in + n_cons, a.begin());
    std::copy(b_in, b_in + n_cons, b.begin());
    std::copy(c_in, c_in + n_cons, c.begin());
    fft_inverse(d, a.data(), DIF, false);
    fft_inverse(d, b.data(), DIF, false);
    fft_inverse(d, c.data(), DIF, false);
    fft_forward(d, a.data(), DIT, true);
    fft_forward(d, b.data(), DIT, true);
    fft_forward(d, c.data(), DIT, true);
    u64 e = (u64)d.n;
    Fr den = Fr::inv(Fr::sub(Fr::pow_u64(d.coset, e), Fr::one()));
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < d.n; ++i) a[i] = Fr::mul(Fr::sub(Fr::mul(a[i], b[i]), c[i]), den);
    fft_inverse(d, a.data(), DIF, true);
    return a;
}

// ------------------------------------------------------------------------------------------ Groth16
// Synthetic, trapdoor-known instance used to validate the whole prove tail without pairings.
struct R1CSRow { std::vector<std::pair<uint32_t, Fr>> a, b, c; };

struct SynthKey {
    int log2d;
    size_t n_wires, n_public, n_cons;
    // toxic waste (kept: this is a TEST key)
    Fr tau, alpha, beta, gamma, delta;
    // per-wire polynomial evaluations at tau
    std::vector<Fr> At, Bt, Ct;
    // proving key, wire-indexed (no infinity compaction here; the product handles gnark's compaction)
    std::vector<G1A> A, B1, K, Z;
    std::vector<G2A> B2;
    G1A alpha1, beta1, delta1;
    G2A beta2, delta2;
    std::vector<Fr> Zt;  // dlog of Z[j] (in the order Z is stored)
    bool z_bitrev;
};

struct SynthInstance {
    std::vector<R1CSRow> rows;
    std::vector<Fr> w;        // full wire assignment, w[0] = 1
    std::vector<Fr> a, b, c;  // constraint evaluations
};

static inline SynthInstance synth_instance(size_t n_inputs, size_t n_cons, u64 seed) {
    SplitMix rng(seed);
    SynthInstance s;
    s.w.push_back(Fr::one());
    for (size_t i = 0; i < n_inputs; ++i) s.w.push_back(rng.fr());
    auto lin = [&](std::vector<std::pair<uint32_t, Fr>>& out) {
        int terms = 1 + (int)(rng.next() % 3);
        Fr acc = Fr::zero();
        for (int t = 0; t < terms; ++t) {
            uint32_t idx = (uint32_t)(rng.next() % s.w.size());
            Fr coef = (rng.next() & 3) ? Fr::from_u64(1 + rng.next() % 7) : rng.fr();
            out.push_back({idx, coef});
            acc = Fr::add(acc, Fr::mul(coef, s.w[idx]));
        }
        return acc;
    };
    for (size_t k = 0; k < n_cons; ++k) {
        R1CSRow r;
        Fr av = lin(r.a), bv = lin(r.b);
        Fr cv = Fr::mul(av, bv);
        uint32_t nw = (uint32_t)s.w.size();
        s.w.push_back(cv);
        r.c.push_back({nw, Fr::one()});
        s.rows.push_back(r);
        s.a.push_back(av); s.b.push_back(bv); s.c.push_back(cv);
    }
    return s;
}

static inline SynthKey synth_setup(const SynthInstance& inst, size_t n_public, u64 seed, bool z_bitrev) {
    SynthKey k;
    size_t n_cons = inst.rows.size();
    int lg = 1;
    while (((size_t)1 << lg) < n_cons) ++lg;
    k.log2d = lg;
    k.n_cons = n_cons;
    k.n_wires = inst.w.size();
    k.n_public = n_public;
    k.z_bitrev = z_bitrev;
    SplitMix rng(seed);
    k.tau = rng.fr(); k.alpha = rng.fr(); k.beta = rng.fr(); k.gamma = rng.fr(); k.delta = rng.fr();
    Domain d(lg);
    // Lagrange basis at tau: L_j(tau) = (tau^D - 1)/D * w^j / (tau - w^j)
    Fr tD = Fr::pow_u64(k.tau, (u64)d.n);
    Fr zt = Fr::sub(tD, Fr::one());
    Fr pref = Fr::mul(zt, d.n_inv);
    std::vector<Fr> L(d.n);
    Fr wj = Fr::one();
    for (size_t j = 0; j < d.n; ++j) {
        L[j] = Fr::mul(Fr::mul(pref, wj), Fr::inv(Fr::sub(k.tau, wj)));
        wj = Fr::mul(wj, d.gen);
    }
    k.At.assign(k.n_wires, Fr::zero()); k.Bt = k.At; k.Ct = k.At;
    for (size_t j = 0; j < n_cons; ++j) {
        for (auto& t : inst.rows[j].a) k.At[t.first] = Fr::add(k.At[t.first], Fr::mul(t.second, L[j]));
        for (auto& t : inst.rows[j].b) k.Bt[t.first] = Fr::add(k.Bt[t.first], Fr::mul(t.second, L[j]));
        for (auto& t : inst.rows[j].c) k.Ct[t.first] = Fr::add(k.Ct[t.first], Fr::mul(t.second, L[j]));
    }
    FixedBase<Fp> g1(g1_gen());
    FixedBase<Fp2> g2(g2_gen());
    Fr dinv = Fr::inv(k.delta);
    k.A.resize(k.n_wires); k.B1.resize(k.n_wires); k.B2.resize(k.n_wires); k.K.resize(k.n_wires);
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t i = 0; i < k.n_wires; ++i) {
        k.A[i] = g1.mul_aff(k.At[i]);
        k.B1[i] = g1.mul_aff(k.Bt[i]);
        k.B2[i] = g2.mul_aff(k.Bt[i]);
        if (i < n_public) {
            k.K[i] = {Fp::zero(), Fp::zero()};  // public wires live in the vk, not in pk.K
        } else {
            Fr kv = Fr::add(Fr::add(Fr::mul(k.beta, k.At[i]), Fr::mul(k.alpha, k.Bt[i])), k.Ct[i]);
            k.K[i] = g1.mul_aff(Fr::mul(kv, dinv));
        }
    }
    k.Z.resize(d.n - 1); k.Zt.resize(d.n - 1);
    Fr zd = Fr::mul(zt, dinv);
    std::vector<Fr> znat(d.n);
    Fr x = zd;
    for (size_t j = 0; j < d.n; ++j) { znat[j] = x; x = Fr::mul(x, k.tau); }
    if (z_bitrev) bit_reverse(znat.data(), lg);  // index D-1 is a fixed point, so truncation is order-safe
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t j = 0; j < d.n - 1; ++j) { k.Zt[j] = znat[j]; k.Z[j] = g1.mul_aff(znat[j]); }
    k.alpha1 = g1.mul_aff(k.alpha); k.beta1 = g1.mul_aff(k.beta); k.delta1 = g1.mul_aff(k.delta);
    k.beta2 = g2.mul_aff(k.beta); k.delta2 = g2.mul_aff(k.delta);
    return k;
}

struct ProofPts { G1A ar, krs; G2A bs; };

// The tail of groth16.Prove after the solver (gnark prove.go: computeH, the five MultiExps, r/s
// blinding).  r, s injectable so the result is deterministic.
static inline ProofPts groth16_prove_tail(const SynthKey& k, const SynthInstance& inst, const Fr& r, const Fr& s) {
    Domain d(k.log2d);
    std::vector<Fr> h = compute_h(d, inst.a.data(), inst.b.data(), inst.c.data(), inst.a.size());
    if (!k.z_bitrev) bit_reverse(h.data(), k.log2d);
    G1J ar = msm_pippenger(k.A.data(), inst.w.data(), k.n_wires);
    ar = jadd_aff(ar, k.alpha1);
    G1J dr = jmul_fr(to_jac(k.delta1), r);
    ar = jadd(ar, dr);
    G1J bs1 = msm_pippenger(k.B1.data(), inst.w.data(), k.n_wires);
    bs1 = jadd_aff(bs1, k.beta1);
    bs1 = jadd(bs1, jmul_fr(to_jac(k.delta1), s));
    G2J bs2 = msm_pippenger(k.B2.data(), inst.w.data(), k.n_wires);
    bs2 = jadd_aff(bs2, k.beta2);
    bs2 = jadd(bs2, jmul_fr(to_jac(k.delta2), s));
    G1J krs = msm_pippenger(k.K.data(), inst.w.data(), k.n_wires);  // K[i]=inf for public wires
    krs = jadd(krs, msm_pippenger(k.Z.data(), h.data(), k.Z.size()));
    Fr kr = Fr::neg(Fr::mul(r, s));
    krs = jadd(krs, jmul_fr(to_jac(k.delta1), kr));
    krs = jadd(krs, jmul_fr(ar, s));
    krs = jadd(krs, jmul_fr(bs1, r));
    return {to_aff(ar), to_aff(krs), to_aff(bs2)};
}

// Groth16 verification equation checked on discrete logs (possible because the key's toxic waste is
// known): with a = dlog(Ar), b = dlog(Bs), the prover's Krs must equal
//   ( a*b - alpha*beta - sum_pub w_i (beta A_i + alpha B_i + C_i) ) / delta  * G1.
static inline bool groth16_check_in_exponent(const SynthKey& k, const SynthInstance& inst, const Fr& r,
                                             const Fr& s, const ProofPts& pr) {
    Fr a = Fr::add(k.alpha, Fr::mul(r, k.delta));
    Fr b = Fr::add(k.beta, Fr::mul(s, k.delta));
    Fr pub = Fr::zero();
    for (size_t i = 0; i < k.n_wires; ++i) {
        a = Fr::add(a, Fr::mul(inst.w[i], k.At[i]));
        b = Fr::add(b, Fr::mul(inst.w[i], k.Bt[i]));
        if (i < k.n_public) {
            Fr kv = Fr::add(Fr::add(Fr::mul(k.beta, k.At[i]), Fr::mul(k.alpha, k.Bt[i])), k.Ct[i]);
            pub = Fr::add(pub, Fr::mul(inst.w[i], kv));
        }
    }
    FixedBase<Fp> g1(g1_gen());
    FixedBase<Fp2> g2(g2_gen());
    if (!(g1.mul_aff(a) == pr.ar)) return false;
    if (!(g2.mul_aff(b) == pr.bs)) return false;
    Fr krs = Fr::mul(Fr::sub(Fr::sub(Fr::mul(a, b), Fr::mul(k.alpha, k.beta)), pub), Fr::inv(k.delta));
    return g1.mul_aff(krs) == pr.krs;
}

// ---- the verifier's view: a verifying key and the pairing equation ------------------------------------------------
// gnark groth16.Verify (backend/groth16/bn254/verify.go, called at prover.go:276 and src/verifier/main.go:263,284):
//   e(Ar, Bs) == e(alpha, beta) * e(sum_pub w_i K_i^vk, gamma) * e(Krs, delta),   K_i^vk = (beta A_i + alpha B_i + C_i)/gamma
// checked as a product of Miller loops equal to one.  The vk is derived from the synthetic key (the toxic waste is only
// used to FORM the vk, as a real setup does; the check itself uses nothing but the vk, the public wires and the proof).
struct SynthVK {
    G1A alpha1;
    G2A beta2, gamma2, delta2;
    std::vector<G1A> Kpub;
};
static inline SynthVK synth_vk(const SynthKey& k) {
    SynthVK vk;
    FixedBase<Fp> g1(g1_gen());
    FixedBase<Fp2> g2(g2_gen());
    vk.alpha1 = k.alpha1; vk.beta2 = k.beta2; vk.delta2 = k.delta2;
    vk.gamma2 = g2.mul_aff(k.gamma);
    Fr ginv = Fr::inv(k.gamma);
    for (size_t i = 0; i < k.n_public; ++i) {
        Fr kv = Fr::add(Fr::add(Fr::mul(k.beta, k.At[i]), Fr::mul(k.alpha, k.Bt[i])), k.Ct[i]);
        vk.Kpub.push_back(g1.mul_aff(Fr::mul(kv, ginv)));
    }
    return vk;
}
static inline bool groth16_verify_pairing(const SynthVK& vk, const Fr* public_wires, const ProofPts& pr) {
    if (!g1_on_curve(pr.ar) || !g1_on_curve(pr.krs) || !g2_on_curve(pr.bs)) return false;
    G1J acc = G1J::inf();
    for (size_t i = 0; i < vk.Kpub.size(); ++i) acc = jadd(acc, jmul_fr(to_jac(vk.Kpub[i]), public_wires[i]));
    G1A P[4] = {pr.ar, aneg(vk.alpha1), aneg(to_aff(acc)), aneg(pr.krs)};
    G2A Q[4] = {pr.bs, vk.beta2, vk.gamma2, vk.delta2};
    return pairing_product_is_one(P, Q, 4);
}
// ---- Groth16 with one BSB22 commitment (gnark backend/groth16/bn254 setup.go / prove.go / verify.go with r1cs.CommitmentInfo) ----
// Setup moves the privately committed wires out of pk.G1.K (delta-divided) into the Pedersen basis, gamma-divided like the public
// wires:  Basis_i = ((beta A_i + alpha B_i + C_i) / gamma) G1,  BasisExpSigma_i = sigma Basis_i.  The prover sends
// D = sum_committed w_i Basis_i with a proof of knowledge; Krs sums pk.G1.K over the remaining private wires only; the verifier adds
// D to the public-input sum:   e(Ar, Bs) == e(alpha, beta) e(sum_pub w_i K_i^vk + D, gamma) e(Krs, delta),  e(D, sigma G2) == e(pok, G2).
// (The commitment wire's own value — a hash of D the verifier recomputes — is the solver's and the hash-to-field's business and is
// not modelled: in this synthetic system it is an ordinary public wire.)
static inline void synth_commitment_basis(const SynthKey& k, const uint32_t* committed, size_t n, const Fr& sigma, G1A* basis, G1A* basis_sigma) {
    FixedBase<Fp> g1(g1_gen());
    Fr ginv = Fr::inv(k.gamma);
    for (size_t j = 0; j < n; ++j) {
        uint32_t i = committed[j];
        Fr kv = Fr::mul(Fr::add(Fr::add(Fr::mul(k.beta, k.At[i]), Fr::mul(k.alpha, k.Bt[i])), k.Ct[i]), ginv);
        basis[j] = g1.mul_aff(kv);
        basis_sigma[j] = g1.mul_aff(Fr::mul(kv, sigma));
    }
}
static inline bool groth16_verify_pairing_commit(const SynthVK& vk, const Fr* public_wires, const ProofPts& pr, const G1A& commitment,
                                                 const G1A& pok, const G2A& g2_sigma) {
    if (!g1_on_curve(pr.ar) || !g1_on_curve(pr.krs) || !g2_on_curve(pr.bs) || !g1_on_curve(commitment) || !g1_on_curve(pok)) return false;
    G1J acc = to_jac(commitment);
    for (size_t i = 0; i < vk.Kpub.size(); ++i) acc = jadd(acc, jmul_fr(to_jac(vk.Kpub[i]), public_wires[i]));
    G1A P[4] = {pr.ar, aneg(vk.alpha1), aneg(to_aff(acc)), aneg(pr.krs)};
    G2A Q[4] = {pr.bs, vk.beta2, vk.gamma2, vk.delta2};
    if (!pairing_product_is_one(P, Q, 4)) return false;
    G1A P2[2] = {commitment, aneg(pok)};
    G2A Q2[2] = {g2_sigma, g2_gen()};
    return pairing_product_is_one(P2, Q2, 2);
}

// Pedersen proof of knowledge (gnark-crypto fr/pedersen VerifyingKey.Verify): with BasisExpSigma_i = sigma * Basis_i,
//   e(commitment, sigma * G2) == e(pok, G2)
static inline bool pedersen_verify_pairing(const G1A& commitment, const G1A& pok, const G2A& g2_sigma) {
    G1A P[2] = {commitment, aneg(pok)};
    G2A Q[2] = {g2_sigma, g2_gen()};
    return pairing_product_is_one(P, Q, 2);
}

}  // namespace orc
