// BN254 G1 (over Fp) and G2 (over Fp2) group law for the MSM kernels, generic in the coordinate field F.
// Buckets are kept in extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2): a mixed
// addition with an affine key point costs 8M + 2S and needs no inversion; infinity is ZZ = 0.
// Affine infinity is (0,0) — the convention of the gnark G1Affine/G2Affine values the reference's proving key
// holds (pk.G1.A etc., loaded at src/prover/prover/prover.go:336-349).
#pragma once
#include "fe.cuh"

namespace zk {

template <class F>
struct Affine {
    F x, y;
    ZK_HD bool is_inf() const { return x.is_zero() & y.is_zero(); }
};
template <class F>
struct XYZZ {
    F x, y, zz, zzz;
    ZK_HD static XYZZ inf() { return {F::zero(), F::zero(), F::zero(), F::zero()}; }
    ZK_HD bool is_inf() const { return zz.is_zero(); }
};
template <class F>
struct Jacobian {
    F x, y, z;
};

// doubling of an affine point into XYZZ (mdbl-2008-s-1)
// (cold path: out of line and by value so that a caller's accumulator can stay in registers)
template <class F>
ZK_HD_NOINLINE XYZZ<F> xyzz_dbl_affine(F x1, F y1) {
    F U = F::dbl(y1);
    F V = F::sqr(U);
    F W = F::mul(U, V);
    F S = F::mul(x1, V);
    F X2 = F::sqr(x1);
    F M = F::add(F::dbl(X2), X2);
    F X3 = F::sub(F::sqr(M), F::dbl(S));
    F Y3 = F::sub(F::mul(M, F::sub(S, X3)), F::mul(W, y1));
    return {X3, Y3, V, W};
}
// doubling of an XYZZ point (dbl-2008-s-1)
template <class F>
ZK_HD_NOINLINE XYZZ<F> xyzz_dbl(XYZZ<F> p) {
    if (p.is_inf()) return p;
    F U = F::dbl(p.y);
    F V = F::sqr(U);
    F W = F::mul(U, V);
    F S = F::mul(p.x, V);
    F X2 = F::sqr(p.x);
    F M = F::add(F::dbl(X2), X2);
    F X3 = F::sub(F::sqr(M), F::dbl(S));
    F Y3 = F::sub(F::mul(M, F::sub(S, X3)), F::mul(W, p.y));
    return {X3, Y3, F::mul(V, p.zz), F::mul(W, p.zzz)};
}
// acc += (x2, y2) affine, non-infinity (madd-2008-s), exceptional cases handled
template <class F>
ZK_HD void xyzz_madd(XYZZ<F>& acc, const F& x2, const F& y2) {
    if (acc.is_inf()) {
        acc.x = x2; acc.y = y2; acc.zz = F::one(); acc.zzz = F::one();
        return;
    }
    F U2 = F::mul(x2, acc.zz);
    F S2 = F::mul(y2, acc.zzz);
    F Pd = F::sub(U2, acc.x);
    F Rd = F::sub(S2, acc.y);
    if (Pd.is_zero()) {
        if (Rd.is_zero()) acc = xyzz_dbl_affine<F>(x2, y2);
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP = F::sqr(Pd);
    F PPP = F::mul(Pd, PP);
    F Q = F::mul(acc.x, PP);
    F X3 = F::sub(F::sub(F::sqr(Rd), PPP), F::dbl(Q));
    F Y3 = F::sub(F::mul(Rd, F::sub(Q, X3)), F::mul(acc.y, PPP));
    acc.x = X3; acc.y = Y3;
    acc.zz = F::mul(acc.zz, PP);
    acc.zzz = F::mul(acc.zzz, PPP);
}
// acc += q (both XYZZ) (add-2008-s), exceptional cases handled
template <class F>
ZK_HD void xyzz_add(XYZZ<F>& acc, const XYZZ<F>& q) {
    if (q.is_inf()) return;
    if (acc.is_inf()) { acc = q; return; }
    F U1 = F::mul(acc.x, q.zz);
    F U2 = F::mul(q.x, acc.zz);
    F S1 = F::mul(acc.y, q.zzz);
    F S2 = F::mul(q.y, acc.zzz);
    F Pd = F::sub(U2, U1);
    F Rd = F::sub(S2, S1);
    if (Pd.is_zero()) {
        if (Rd.is_zero()) acc = xyzz_dbl<F>(acc);
        else acc = XYZZ<F>::inf();
        return;
    }
    F PP = F::sqr(Pd);
    F PPP = F::mul(Pd, PP);
    F Q = F::mul(U1, PP);
    F X3 = F::sub(F::sub(F::sqr(Rd), PPP), F::dbl(Q));
    F Y3 = F::sub(F::mul(Rd, F::sub(Q, X3)), F::mul(S1, PPP));
    acc.x = X3; acc.y = Y3;
    acc.zz = F::mul(F::mul(acc.zz, q.zz), PP);
    acc.zzz = F::mul(F::mul(acc.zzz, q.zzz), PPP);
}
// out-of-line general addition for the (cold, code-size-heavy) partial-sum and reduction kernels
template <class F>
ZK_HD_NOINLINE XYZZ<F> xyzz_add_nl(const XYZZ<F>& a, const XYZZ<F>& b) {
    XYZZ<F> r = a;
    xyzz_add<F>(r, b);
    return r;
}
template <class F>
ZK_HD XYZZ<F> xyzz_neg(const XYZZ<F>& p) { return {p.x, F::neg(p.y), p.zz, p.zzz}; }
template <class F>
ZK_HD XYZZ<F> xyzz_from_affine(const Affine<F>& p) {
    if (p.is_inf()) return XYZZ<F>::inf();
    return {p.x, p.y, F::one(), F::one()};
}
// XYZZ -> Jacobian without inversion: Z = ZZZ, X' = X*ZZ^2, Y' = Y*ZZZ^2  (uses ZZ^3 = ZZZ^2)
template <class F>
ZK_HD Jacobian<F> xyzz_to_jacobian(const XYZZ<F>& p) {
    if (p.is_inf()) return {F::one(), F::one(), F::zero()};
    return {F::mul(p.x, F::sqr(p.zz)), F::mul(p.y, F::sqr(p.zzz)), p.zzz};
}
template <class F>
ZK_HD Affine<F> xyzz_to_affine(const XYZZ<F>& p) {
    if (p.is_inf()) return {F::zero(), F::zero()};
    return {F::mul(p.x, F::inv(p.zz)), F::mul(p.y, F::inv(p.zzz))};
}
// k * p for a small unsigned k (host-side finishing steps)
template <class F>
ZK_HD XYZZ<F> xyzz_mul_u64(const XYZZ<F>& p, u64 k) {
    XYZZ<F> r = XYZZ<F>::inf();
    for (int i = 63; i >= 0; --i) {
        r = xyzz_dbl<F>(r);
        if ((k >> i) & 1) xyzz_add<F>(r, p);
    }
    return r;
}
// k * p for a canonical (non-Montgomery) 256-bit scalar given as 8 u32 limbs
template <class F>
ZK_HD XYZZ<F> xyzz_mul_limbs(const XYZZ<F>& p, const u32* k) {
    XYZZ<F> r = XYZZ<F>::inf();
    for (int i = 255; i >= 0; --i) {
        r = xyzz_dbl<F>(r);
        if ((k[i >> 5] >> (i & 31)) & 1u) xyzz_add<F>(r, p);
    }
    return r;
}

typedef Affine<Fp> G1Affine;
typedef XYZZ<Fp> G1XYZZ;
typedef Jacobian<Fp> G1Jac;
typedef Affine<Fp2> G2Affine;
typedef XYZZ<Fp2> G2XYZZ;
typedef Jacobian<Fp2> G2Jac;

}  // namespace zk
