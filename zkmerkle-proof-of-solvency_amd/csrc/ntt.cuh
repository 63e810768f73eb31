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
    // the same constants for the 29-bit kernel (k_ntt_pass29): Montgomery radix 2^261, packed as 8 x 32-bit words
    bool have29 = false;
    Fr* mem29 = nullptr;
    Fr *g_lo29, *g_hi29, *gi_lo29, *gi_hi29, *g_hi_ninv29, *gi_hi_ninv29;
    Fr *tw_lo29 = nullptr, *tw_hi29 = nullptr, *twi_lo29 = nullptr, *twi_hi29 = nullptr;   // w^e, w^-e half tables, 2^261 form ("ntt_twiddles" 1)
    u32 *small_fwd29, *small_inv29;             // 256 x 9 limbs each
    Fr n_inv29;
    Fr* full_fwd29[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    Fr* full_inv29[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};
int32_t ntt_domain_get(zkpor_ctx* ctx, int n, NttDomain** out);
void ntt_domains_free(zkpor_ctx* ctx);
int32_t ntt_debug_fault(zkpor_ctx* ctx, int n);
int32_t ntt_dev(zkpor_ctx* ctx, Fr* d_x, int n, bool inverse, bool dif, bool on_coset);
int32_t compute_h_dev(zkpor_ctx* ctx, int n, Fr* a, Fr* b, Fr* c, const Fr* a_in = nullptr, const Fr* b_in = nullptr, const Fr* c_in = nullptr);
}  // namespace zk
