// NTT domain tables and entry points (ntt.hip)
#pragma once
#include "common.cuh"
namespace zk {
struct NttDomain {
    int n = 0, tb = 0;
    Fr* mem = nullptr;
    Fr *tw_lo, *tw_hi, *twi_lo, *twi_hi;        // w^e, w^-e split tables
    Fr *g_lo, *g_hi, *gi_lo, *gi_hi;            // g^e, g^-e (g = 5, the coset shift)
    Fr *g_hi_ninv, *gi_hi_ninv;                 // high tables with 1/N folded in
    Fr *small_fwd, *small_inv;                  // w_512^j, w_512^-j
    Fr n_inv, den;                              // 1/N, 1/(g^N - 1)
    Fr* full_fwd[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // per field: tabulated
    Fr* full_inv[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // inter-pass twiddles
};
int32_t ntt_domain_get(zkpor_ctx* ctx, int n, NttDomain** out);
void ntt_domains_free(zkpor_ctx* ctx);
int32_t ntt_dev(zkpor_ctx* ctx, Fr* d_x, int n, bool inverse, bool dif, bool on_coset);
int32_t compute_h_dev(zkpor_ctx* ctx, int n, Fr* a, Fr* b, Fr* c);
}  // namespace zk
