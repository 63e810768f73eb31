// TEST INFRASTRUCTURE (oracle): a pairing on BN254, so that proofs produced by the device can be put through the
// verification equation itself — the acceptance test the reference applies after every proof
// (groth16.Verify, src/prover/prover/prover.go:276; src/verifier/main.go:263,284) — and not only through the
// discrete-log restatement of algos.hpp (which needs the toxic waste).  gnark's verifier uses the optimal ate pairing
// (gnark-crypto v0.14 ecc/bn254/pairing.go, absent from /root/reference); an accept/reject decision is the same under
// ANY non-degenerate bilinear pairing on (G1, G2), so this file implements the simplest one, the reduced Tate
// pairing  t(P, Q) = f_{r,P}(psi(Q)) ^ ((p^12 - 1)/r):
//   * Miller loop over the bits of r with the running point in G1 (plain Fp arithmetic, affine),
//   * Q in G2 (on the D-type twist y^2 = x^3 + 3/xi, xi = 9 + u) mapped to E(Fp12) by psi(x', y') = (x' w^2, y' w^3),
//     Fp12 = Fp2[w]/(w^6 - xi); vertical lines lie in the subfield Fp6 = Fp2[w^2] and vanish under the final power,
//   * the final exponent as one plain square-and-multiply with the constant of pairing_consts.inc.
// No Frobenius constants, no tower tricks: slow (tens of ms) and easy to audit.  Checked by tests/test_oracle_cpu.py
// (bilinearity in both arguments, non-degeneracy, e(aP, Q) = e(P, aQ)).
#pragma once
#include "bn254.hpp"

namespace orc {

#include "pairing_consts.inc"

static inline Fp2 mul_xi(const Fp2& a) {  // (a0 + a1 u)(9 + u)
    Fp n0 = Fp::sub(Fp::mul(Fp::from_u64(9), a.a0), a.a1);
    Fp n1 = Fp::add(Fp::mul(Fp::from_u64(9), a.a1), a.a0);
    return {n0, n1};
}

struct Fp12 {
    Fp2 c[6];  // sum c[i] w^i
    static Fp12 one() {
        Fp12 r;
        for (int i = 0; i < 6; ++i) r.c[i] = Fp2::zero();
        r.c[0] = Fp2::one();
        return r;
    }
    static Fp12 mul(const Fp12& a, const Fp12& b) {
        Fp2 t[11];
        for (int i = 0; i < 11; ++i) t[i] = Fp2::zero();
        for (int i = 0; i < 6; ++i) {
            if (a.c[i].is_zero()) continue;
            for (int j = 0; j < 6; ++j) {
                if (b.c[j].is_zero()) continue;
                t[i + j] = Fp2::add(t[i + j], Fp2::mul(a.c[i], b.c[j]));
            }
        }
        Fp12 r;
        for (int i = 0; i < 6; ++i) r.c[i] = i + 6 < 11 ? Fp2::add(t[i], mul_xi(t[i + 6])) : t[i];
        return r;
    }
    static Fp12 pow(const Fp12& a, const u64* e, int nlimbs) {
        Fp12 r = one();
        bool started = false;
        for (int i = nlimbs * 64 - 1; i >= 0; --i) {
            if (started) r = mul(r, r);
            if ((e[i / 64] >> (i % 64)) & 1) { r = started ? mul(r, a) : a; started = true; }
        }
        return r;
    }
    bool is_one() const {
        if (!(c[0] == Fp2::one())) return false;
        for (int i = 1; i < 6; ++i) if (!c[i].is_zero()) return false;
        return true;
    }
    bool operator==(const Fp12& o) const {
        for (int i = 0; i < 6; ++i) if (!(c[i] == o.c[i])) return false;
        return true;
    }
};

// line through T with slope lam, evaluated at psi(Q):  (lam x_T - y_T) - lam x' w^2 + y' w^3
static inline Fp12 line_eval(const Fp& lam, const G1A& T, const G2A& Q) {
    Fp12 l;
    for (int i = 0; i < 6; ++i) l.c[i] = Fp2::zero();
    l.c[0] = {Fp::sub(Fp::mul(lam, T.x), T.y), Fp::zero()};
    l.c[2] = {Fp::neg(Fp::mul(lam, Q.x.a0)), Fp::neg(Fp::mul(lam, Q.x.a1))};
    l.c[3] = Q.y;
    return l;
}

// f_{r,P}(psi(Q)) without the final exponentiation; 1 if either point is infinity
static inline Fp12 miller_tate(const G1A& P, const G2A& Q) {
    Fp12 f = Fp12::one();
    if (P.is_inf() || Q.is_inf()) return f;
    const u64* r = FrTag::MOD;
    G1A T = P;
    bool t_inf = false;
    int top = 253;
    while (!((r[top / 64] >> (top % 64)) & 1)) --top;
    for (int i = top - 1; i >= 0; --i) {
        f = Fp12::mul(f, f);
        if (!t_inf) {
            // tangent at T (y_T != 0: the group has odd order)
            Fp lam = Fp::mul(Fp::mul(Fp::from_u64(3), Fp::sqr(T.x)), Fp::inv(Fp::dbl(T.y)));
            f = Fp12::mul(f, line_eval(lam, T, Q));
            Fp x3 = Fp::sub(Fp::sqr(lam), Fp::dbl(T.x));
            Fp y3 = Fp::sub(Fp::mul(lam, Fp::sub(T.x, x3)), T.y);
            T = {x3, y3};
        }
        if ((r[i / 64] >> (i % 64)) & 1) {
            if (t_inf) { T = P; t_inf = false; continue; }
            if (T.x == P.x) {
                // T = -P (only at the very last step, T + P = infinity): vertical line, killed by the final power.
                // T = P cannot happen: it would need k = 1 (mod r) for a proper prefix k of r.
                t_inf = true;
                continue;
            }
            Fp lam = Fp::mul(Fp::sub(P.y, T.y), Fp::inv(Fp::sub(P.x, T.x)));
            f = Fp12::mul(f, line_eval(lam, T, Q));
            Fp x3 = Fp::sub(Fp::sub(Fp::sqr(lam), T.x), P.x);
            Fp y3 = Fp::sub(Fp::mul(lam, Fp::sub(T.x, x3)), T.y);
            T = {x3, y3};
        }
    }
    return f;
}
static inline Fp12 final_exp(const Fp12& f) { return Fp12::pow(f, TATE_EXP, TATE_EXP_LIMBS); }
static inline Fp12 pairing(const G1A& P, const G2A& Q) { return final_exp(miller_tate(P, Q)); }
// prod_i t(P_i, Q_i) == 1, with one shared final exponentiation (what a verifier computes)
static inline bool pairing_product_is_one(const G1A* P, const G2A* Q, size_t n) {
    Fp12 f = Fp12::one();
    for (size_t i = 0; i < n; ++i) f = Fp12::mul(f, miller_tate(P[i], Q[i]));
    return final_exp(f).is_one();
}

}  // namespace orc
