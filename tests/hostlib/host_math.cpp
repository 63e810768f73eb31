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
