// ORACLE — TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py's untimed `checked` leg).  Never linked into the product.
//
// An FFT-free verifier of computeH's output (gnark backend/groth16/bn254/prove.go computeH; reference call site
// src/prover/prover/prover.go:269).  Groth16's quotient is DEFINED by
//
//        H(X) * (X^D - 1)  =  A(X) * B(X) - C(X),
//
// where A, B, C are the polynomials of degree < D that take the values a_i, b_i, c_i on the domain {w^i}.  The identity is
// checked at one point tau outside the domain (Schwartz-Zippel: a wrong h of degree < D passes with probability <= 2D/r < 2^-226):
//
//        A(tau) = (tau^D - 1)/D * sum_i a_i * w^i / (tau - w^i)        (barycentric Lagrange on the roots of unity)
//        H(tau) = sum_k h_k tau^k                                       (h = coefficient vector, natural or bit-reversed)
//
// Nothing here shares code with the FFTs of algos.hpp / cpubase.hpp or with the device passes: field products, one batch
// inversion per block (Montgomery's trick) and power tables only.  O(D) work, OpenMP over blocks: ~1 s at D = 2^26 on 16 threads.
#pragma once
#include <vector>

#include "algos.hpp"

namespace orc_quot {
using namespace orc;

struct Eval { Fr At, Bt, Ct, Ht, lhs, rhs; };

// a, b, c: n_cons values each (implicitly zero-padded to D = 2^log2d); h: D coefficients, coefficient k at position k
// (h_bitrev = 0) or at position bitrev(k) (h_bitrev = 1: what gnark's computeH returns and pk.G1.Z is ordered by)
static inline Eval quotient_identity(int log2d, const Fr* a, const Fr* b, const Fr* c, size_t n_cons, const Fr* h, int h_bitrev,
                                     const Fr& tau) {
    const size_t D = (size_t)1 << log2d;
    Fr w = fr_root_of_unity_2_28();
    for (int i = log2d; i < 28; ++i) w = Fr::sqr(w);
    const int lb = log2d < 12 ? log2d : 12;              // block = 2^lb consecutive positions
    const size_t B = (size_t)1 << lb, nblk = D >> lb;
    // ---- barycentric sums over the first n_cons positions
    Fr sA = Fr::zero(), sB = Fr::zero(), sC = Fr::zero();
    const size_t cblk = (n_cons + B - 1) >> lb;
#pragma omp parallel
    {
        Fr lA = Fr::zero(), lB = Fr::zero(), lC = Fr::zero();
        std::vector<Fr> x(B), d(B), pre(B);
#pragma omp for schedule(static) nowait
        for (size_t blk = 0; blk < cblk; ++blk) {
            const size_t lo = blk << lb, hi = (lo + B < n_cons) ? lo + B : n_cons, m = hi - lo;
            Fr xi = Fr::pow_u64(w, (u64)lo);
            Fr run = Fr::one();
            for (size_t j = 0; j < m; ++j) {
                x[j] = xi;
                d[j] = Fr::sub(tau, xi);
                pre[j] = run;                        // product of d[0..j)
                run = Fr::mul(run, d[j]);
                xi = Fr::mul(xi, w);
            }
            Fr inv = Fr::inv(run);                    // tau is outside the domain: no factor is zero
            for (size_t j = m; j-- > 0;) {
                Fr dinv = Fr::mul(inv, pre[j]);
                inv = Fr::mul(inv, d[j]);
                Fr wt = Fr::mul(x[j], dinv);
                lA = Fr::add(lA, Fr::mul(a[lo + j], wt));
                lB = Fr::add(lB, Fr::mul(b[lo + j], wt));
                lC = Fr::add(lC, Fr::mul(c[lo + j], wt));
            }
        }
#pragma omp critical
        { sA = Fr::add(sA, lA); sB = Fr::add(sB, lB); sC = Fr::add(sC, lC); }
    }
    Fr tD = tau;
    for (int i = 0; i < log2d; ++i) tD = Fr::sqr(tD);
    const Fr zh = Fr::sub(tD, Fr::one());                                  // tau^D - 1
    const Fr scale = Fr::mul(zh, Fr::inv(Fr::from_u64((u64)D)));
    Eval e;
    e.At = Fr::mul(sA, scale); e.Bt = Fr::mul(sB, scale); e.Ct = Fr::mul(sC, scale);
    // ---- H(tau): position p = (p_hi << lb) | p_lo holds coefficient k(p); tau^k(p) = T_lo[p_lo] * (one power per block)
    const int hb = log2d - lb;
    std::vector<Fr> tlo(B);
    for (size_t j = 0; j < B; ++j) {
        u64 k = h_bitrev ? ((u64)bitrev(j, lb) << hb) : (u64)j;
        tlo[j] = Fr::pow_u64(tau, k);
    }
    Fr sH = Fr::zero();
#pragma omp parallel
    {
        Fr lH = Fr::zero();
#pragma omp for schedule(static) nowait
        for (size_t blk = 0; blk < nblk; ++blk) {
            u64 k = h_bitrev ? (u64)bitrev(blk, hb) : ((u64)blk << lb);
            Fr base = Fr::pow_u64(tau, k);
            Fr loc = Fr::zero();
            const Fr* hp = h + (blk << lb);
            for (size_t j = 0; j < B; ++j) loc = Fr::add(loc, Fr::mul(hp[j], tlo[j]));
            lH = Fr::add(lH, Fr::mul(loc, base));
        }
#pragma omp critical
        sH = Fr::add(sH, lH);
    }
    e.Ht = sH;
    e.lhs = Fr::mul(sH, zh);
    e.rhs = Fr::sub(Fr::mul(e.At, e.Bt), e.Ct);
    return e;
}

}  // namespace orc_quot
