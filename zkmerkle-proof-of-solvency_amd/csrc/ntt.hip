// Fr-domain NTT and the H-polynomial quotient for gfx950 — replaces gnark-crypto fr/fft (Domain.FFT / FFTInverse
// with DIF/DIT and OnCoset) and gnark's computeH (backend/groth16/bn254/prove.go), which groth16.Prove runs on the
// host (reference call site src/prover/prover/prover.go:269).
//
// Structure (index algebra proven against a naive DFT in tools/ntt_model.py, which this file transcribes):
//   the log2(N) index bits are split into fields of <= 9 bits (26 = 8 + 9 + 9).  One launch ("pass") performs all
//   radix-2 stages of one field on LDS-resident tiles (2^kb field elements x C adjacent columns, <= 64 KiB), so a
//   2^26 transform is 3 read+write sweeps of HBM instead of 26.  Between fields the four-step twiddle
//   w_N^(l * k * 2^s0) is applied on the fly from two L2-resident half-size tables (2 x 2^13 entries) rather than a
//   1 GiB twiddle array; coset powers g^i and the 1/N factor are folded into the load of the first / store of the
//   last pass the same way.  DIF passes run high field -> low field (natural in, bit-reversed out), DIT passes
//   low -> high (bit-reversed in, natural out), so computeH never needs a bit-reversal permutation.
#include "common.cuh"
#include "ntt.cuh"
#include "fe29.cuh"

namespace zk {

// Fr constants (Montgomery form)
__host__ __device__ inline Fr fr_const(const u32 (&l)[8]) {
    Fr r;
    for (int i = 0; i < 8; ++i) r.v[i] = l[i];
    return r;
}
static const u32 W28_M[8] = {0x80d13d9cu, 0x636e7355u, 0x2445ffd6u, 0xa22bf374u, 0x1eb203d8u, 0x56452ac0u, 0x2963f9e7u, 0x1860ef94u};
static const u32 FIVE_M[8] = {0x9fffffe6u, 0x1b0d0ef9u, 0xa32a913fu, 0xeaba68a3u, 0xd8dd0689u, 0x47d8eb76u, 0x20f5bbc3u, 0x15d00855u};

// out[e] = base^(e << shift) * mult, e < count
__global__ void k_pow_table(Fr base, Fr mult, int shift, u32 count, Fr* out) {
    u32 e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= count) return;
    u64 ex = (u64)e << shift;
    Fr r = Fr::one(), b = base;
    while (ex) {
        if (ex & 1) r = Fr::mul(r, b);
        b = Fr::sqr(b);
        ex >>= 1;
    }
    out[e] = Fr::mul(r, mult);
}

// full inter-pass twiddle table of one field: out[k << lo | l] = w^((l*k) << s0)
__global__ void k_twiddle_full(const Fr* lo_t, const Fr* hi_t, int tb, int lo, int kb, int s0, Fr* out) {
    size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= ((size_t)1 << (lo + kb))) return;
    u32 l = (u32)(idx & (((size_t)1 << lo) - 1)), k = (u32)(idx >> lo);
    u32 e = (l * k) << s0;
    out[idx] = Fr::mul(lo_t[e & ((1u << tb) - 1u)], hi_t[e >> tb]);
}

ZK_D u32 brev(u32 x, int bits) { return bits ? (__brev(x) >> (32 - bits)) : 0u; }

struct PassArgs {
    Fr* x;
    int n, lo, kb, clog;
    const Fr* small;             // w_512^j (forward or inverse), 256 entries
    const Fr* tw_lo; const Fr* tw_hi; int tb;  // w_N^e = tw_lo[e & mask] * tw_hi[e >> tb]
    const Fr* tw_full;           // optional: the field's inter-pass twiddles tabulated, [k_m << lo | l] (HBM is plentiful)
    int scale_load, scale_store;  // 0 none, 1 constant, 2 g^p, 3 g^rev(p)   (g tables may carry a folded constant)
    const Fr* g_lo; const Fr* g_hi;
    Fr konst;
};

ZK_D Fr table_pow(const Fr* lo, const Fr* hi, int tb, u32 e) {
    return Fr::mul(lo[e & ((1u << tb) - 1u)], hi[e >> tb]);
}
ZK_D Fr apply_scale(const PassArgs& A, int mode, const Fr& v, u32 p) {
    if (mode == 1) return Fr::mul(v, A.konst);
    u32 e = mode == 2 ? p : brev(p, A.n);
    return Fr::mul(v, table_pow(A.g_lo, A.g_hi, A.tb, e));
}

template <bool DIF>
__global__ __launch_bounds__(256) void k_ntt_pass(PassArgs A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    Fr* tile = (Fr*)smem_raw;
    const int kb = A.kb, lo = A.lo, clog = A.clog;
    const u32 F = 1u << kb, C = 1u << clog;
    const int s0 = A.n - lo - kb;
    // block -> (hi, l0) for lo > 0, or hi0 for lo == 0 (columns = consecutive tiles)
    u32 hi, l0;
    if (lo > 0) {
        u32 lgroups = (1u << lo) >> clog;
        hi = blockIdx.x / lgroups;
        l0 = (blockIdx.x % lgroups) << clog;
    } else {
        hi = blockIdx.x << clog;
        l0 = 0;
    }
    const u32 total = F << clog;
    // ---- load (+ DIT pre-twiddle / scale)
    for (u32 idx = threadIdx.x; idx < total; idx += 256u) {
        u32 m, c, p, li;
        if (lo > 0) { c = idx & (C - 1u); m = idx >> clog; p = (hi << (lo + kb)) + (m << lo) + l0 + c; li = m * C + c; }
        else { m = idx & (F - 1u); c = idx >> kb; p = ((hi + c) << kb) + m; li = c * F + m; }
        Fr v = A.x[p];
        if (A.scale_load) v = apply_scale(A, A.scale_load, v, p);
        if (!DIF && lo > 0) {
            const u32 km = brev(m, kb);
            if (A.tw_full) v = Fr::mul(v, A.tw_full[((size_t)km << lo) | (l0 + c)]);
            else v = Fr::mul(v, table_pow(A.tw_lo, A.tw_hi, A.tb, ((l0 + c) * km) << s0));
        }
        tile[li] = v;
    }
    __syncthreads();
    // ---- radix-2 stages inside the field
    const u32 nbf = total >> 1;
    for (int j = 0; j < kb; ++j) {
        const int hlog = DIF ? (kb - 1 - j) : j;
        const u32 half = 1u << hlog;
        const int tshift = (DIF ? j : (kb - 1 - j)) + (9 - kb);
        for (u32 t = threadIdx.x; t < nbf; t += 256u) {
            u32 q, c;
            if (lo > 0) { c = t & (C - 1u); q = t >> clog; }
            else { q = t & ((F >> 1) - 1u); c = t >> (kb - 1); }
            u32 pos = q & (half - 1u);
            u32 i0 = ((q >> hlog) << (hlog + 1)) + pos;
            u32 i1 = i0 + half;
            u32 a0 = lo > 0 ? i0 * C + c : c * F + i0;
            u32 a1 = lo > 0 ? i1 * C + c : c * F + i1;
            Fr a = tile[a0], b = tile[a1];
            Fr w = A.small[pos << tshift];
            if (DIF) {
                tile[a0] = Fr::add(a, b);
                tile[a1] = Fr::mul(Fr::sub(a, b), w);
            } else {
                b = Fr::mul(b, w);
                tile[a0] = Fr::add(a, b);
                tile[a1] = Fr::sub(a, b);
            }
        }
        __syncthreads();
    }
    // ---- store (+ DIF post-twiddle / scale)
    for (u32 idx = threadIdx.x; idx < total; idx += 256u) {
        u32 m, c, p, li;
        if (lo > 0) { c = idx & (C - 1u); m = idx >> clog; p = (hi << (lo + kb)) + (m << lo) + l0 + c; li = m * C + c; }
        else { m = idx & (F - 1u); c = idx >> kb; p = ((hi + c) << kb) + m; li = c * F + m; }
        Fr v = tile[li];
        if (DIF && lo > 0) {
            const u32 km = brev(m, kb);
            if (A.tw_full) v = Fr::mul(v, A.tw_full[((size_t)km << lo) | (l0 + c)]);
            else v = Fr::mul(v, table_pow(A.tw_lo, A.tw_hi, A.tb, ((l0 + c) * km) << s0));
        }
        if (A.scale_store) v = apply_scale(A, A.scale_store, v, p);
        A.x[p] = v;
    }
}

// ------------------------------------------------------------------------------------------------ 29-bit pass
// The same pass on 9 x 29-bit signed lazy limbs (fe29.cuh, Fr29): a butterfly is one 206-instruction product plus a
// product-free 30-instruction reduction instead of a 313-instruction product and two carry-chained add/sub with
// conditional corrections.  Between passes the array holds Montgomery-radix-2^261 residues in [0, 4r) packed into the
// same 8 x 32-bit words (reduce32_pos + pack32 at the store, from32<0> at the load), so passes stay in place; the first
// pass reads gnark's form (from32<5> = value * 32 = the 2^261 form, unreduced) and the last one writes it
// (to32_div32).  All constants come from tables already in the 2^261 form.
struct PassArgs29 {
    Fr* x;
    const Fr* src;               // where this pass LOADS from (x unless the transform's first pass reads the caller's untouched input)
    int n, lo, kb, clog;
    const u32* small29;          // w_512^j, 9 limbs each
    const Fr* tw_full;           // [k_m << lo | l], packed 2^261 form; null = generated: w^e = tw_lo[e & mask] * tw_hi[e >> tb] ("ntt_twiddles" 1)
    const Fr* tw_lo; const Fr* tw_hi;   // the two half tables of w (or 1 / w), 2^261 form: 2 x 2^(n/2) entries, L2-resident
    int scale_load, scale_store;  // 0 none, 1 constant, 2 g^p, 3 g^rev(p)
    const Fr* g_lo; const Fr* g_hi; int tb;
    Fr konst;
    int in_gnark, out_gnark;
    // sharded transforms (ntt_shard_stage): x is one rank's local array of 2^n elements of a 2^n_glob transform; the exponents of
    // the inter-pass twiddles and of the scale need the GLOBAL memory position: p_glob = ((p << p_shift) | p_or) + p_add
    int n_glob, p_shift;
    u32 p_or, p_add;
    // computeH without c's coset transform ("ntt_h" 1): the LAST pass of h subtracts sub[p] (packed 2^261 form, [0, 4r)) behind its store scale
    const Fr* sub = nullptr;
};
ZK_D Fr29 ld29(const u32* t) {
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = t[i];
    return r;
}
ZK_D void st29(u32* t, const Fr29& v) {
#pragma unroll
    for (int i = 0; i < 9; ++i) t[i] = v.l[i];
}
// the inter-pass twiddle w^((l * km) << s0) of element (km, l) of a field (l = the GLOBAL low index): from the field's table, or — "ntt_twiddles" 1,
// unsharded transforms — as the product of two half-table entries: one more field product per element instead of a 32-byte read from a 2 GiB table
ZK_D Fr29 twiddle29(const PassArgs29& A, u32 km, u32 l, int lo_eff) {
    if (A.tw_full) return Fr29::from32<0>(A.tw_full[((size_t)km << lo_eff) | l]);
    const u32 e = (l * km) << (A.n_glob - lo_eff - A.kb);
    return Fr29::mul(Fr29::from32<0>(A.tw_lo[e & ((1u << A.tb) - 1u)]), Fr29::from32<0>(A.tw_hi[e >> A.tb]));
}
ZK_D Fr29 scale29(const PassArgs29& A, int mode, const Fr29& v, u32 p) {
    if (mode == 1) return Fr29::mul(v, Fr29::from32<0>(A.konst));
    u32 e = mode == 2 ? p : brev(p, A.n_glob);
    Fr29 g = Fr29::mul(Fr29::from32<0>(A.g_lo[e & ((1u << A.tb) - 1u)]), Fr29::from32<0>(A.g_hi[e >> A.tb]));
    return Fr29::mul(v, g);
}

// all radix-2 stages of one field on an LDS tile of (2^kb x 2^clog) elements (element i at tile + 9 i); ends with a barrier
template <bool DIF>
ZK_D void ntt_stages29(u32* tile, const u32* __restrict__ small29, const int kb, const int lo, const int clog) {
    const u32 F = 1u << kb, C = 1u << clog;
    const u32 total = F << clog;
    // stages two at a time: a thread takes the four elements that differ in the two index bits of stages j and j+1,
    // does both stages in registers (4 products, as two radix-2 stages would) and touches LDS once instead of twice
    int j = 0;
    for (; j + 1 < kb; j += 2) {
        const int hA = DIF ? (kb - 1 - j) : j;          // bit of the first stage of the pair
        const int hB = DIF ? hA - 1 : hA + 1;           // bit of the second
        const int hl = DIF ? hB : hA;                   // the lower of the two
        const int tsA = (DIF ? j : (kb - 1 - j)) + (9 - kb);
        const int tsB = (DIF ? j + 1 : (kb - 2 - j)) + (9 - kb);
        const u32 ngrp = total >> 2;
        for (u32 t = threadIdx.x; t < ngrp; t += 256u) {
            u32 q, c;
            if (lo > 0) { c = t & (C - 1u); q = t >> clog; }
            else { q = t & ((F >> 2) - 1u); c = t >> (kb - 2); }
            const u32 low = q & ((1u << hl) - 1u);
            const u32 base = ((q >> hl) << (hl + 2)) + low;
            const u32 s1 = 1u << hl, s2 = 2u << hl;
            // x0..x3 = base + {0, s1, s2, s1+s2}
            u32* p0 = tile + 9u * (lo > 0 ? base * C + c : c * F + base);
            const u32 estep = 9u * (lo > 0 ? C : 1u);
            u32* p1 = p0 + estep * s1; u32* p2 = p0 + estep * s2; u32* p3 = p2 + estep * s1;
            Fr29 x0 = ld29(p0), x1 = ld29(p1), x2 = ld29(p2), x3 = ld29(p3);
            if (DIF) {
                // stage A pairs elements 2^hA apart (x0,x2),(x1,x3): twiddle position = index below bit hA
                const u32 posA0 = low, posA1 = low + s1;
                Fr29 wa0 = ld29(small29 + 9u * (posA0 << tsA)), wa1 = ld29(small29 + 9u * (posA1 << tsA));
                Fr29 a0 = Fr29::reduce32(Fr29::add_l(x0, x2)), a2 = Fr29::mul(wa0, Fr29::sub_l(x0, x2));
                Fr29 a1 = Fr29::reduce32(Fr29::add_l(x1, x3)), a3 = Fr29::mul(wa1, Fr29::sub_l(x1, x3));
                // stage B pairs elements 2^hB apart (a0,a1),(a2,a3): position = index below bit hB = low
                st29(p0, Fr29::reduce32(Fr29::add_l(a0, a1)));
                st29(p2, Fr29::reduce32(Fr29::add_l(a2, a3)));
                if (hl == 0) {  // the field's last stage: every twiddle is w^0 = 1, a product-free reduction replaces the product
                    st29(p1, Fr29::reduce32(Fr29::sub_l(a0, a1)));
                    st29(p3, Fr29::reduce32(Fr29::sub_l(a2, a3)));
                } else {
                    Fr29 wb = ld29(small29 + 9u * (low << tsB));
                    st29(p1, Fr29::mul(wb, Fr29::sub_l(a0, a1)));
                    st29(p3, Fr29::mul(wb, Fr29::sub_l(a2, a3)));
                }
            } else {
                // stage A pairs (x0,x1),(x2,x3) (distance 2^hA), position = low
                Fr29 t1 = x1, t3 = x3;
                if (hl != 0) {  // (the field's first stage has twiddle w^0 = 1 throughout: no product)
                    Fr29 wa = ld29(small29 + 9u * (low << tsA));
                    t1 = Fr29::mul(x1, wa); t3 = Fr29::mul(x3, wa);
                }
                Fr29 a0 = Fr29::reduce32(Fr29::add_l(x0, t1)), a1 = Fr29::reduce32(Fr29::sub_l(x0, t1));
                Fr29 a2 = Fr29::reduce32(Fr29::add_l(x2, t3)), a3 = Fr29::reduce32(Fr29::sub_l(x2, t3));
                // stage B pairs (a0,a2),(a1,a3) (distance 2^hB), positions low and low + 2^hA
                Fr29 wb0 = ld29(small29 + 9u * (low << tsB)), wb1 = ld29(small29 + 9u * ((low + s1) << tsB));
                Fr29 u2 = Fr29::mul(a2, wb0), u3 = Fr29::mul(a3, wb1);
                st29(p0, Fr29::reduce32(Fr29::add_l(a0, u2)));
                st29(p2, Fr29::reduce32(Fr29::sub_l(a0, u2)));
                st29(p1, Fr29::reduce32(Fr29::add_l(a1, u3)));
                st29(p3, Fr29::reduce32(Fr29::sub_l(a1, u3)));
            }
        }
        __syncthreads();
    }
    const u32 nbf = total >> 1;
    for (; j < kb; ++j) {
        const int hlog = DIF ? (kb - 1 - j) : j;
        const u32 half = 1u << hlog;
        const int tshift = (DIF ? j : (kb - 1 - j)) + (9 - kb);
        for (u32 t = threadIdx.x; t < nbf; t += 256u) {
            u32 q, c;
            if (lo > 0) { c = t & (C - 1u); q = t >> clog; }
            else { q = t & ((F >> 1) - 1u); c = t >> (kb - 1); }
            u32 pos = q & (half - 1u);
            u32 i0 = ((q >> hlog) << (hlog + 1)) + pos;
            u32 i1 = i0 + half;
            u32* p0 = tile + 9u * (lo > 0 ? i0 * C + c : c * F + i0);
            u32* p1 = tile + 9u * (lo > 0 ? i1 * C + c : c * F + i1);
            Fr29 a = ld29(p0), b = ld29(p1);
            if (DIF) {
                st29(p0, Fr29::reduce32(Fr29::add_l(a, b)));
                if (half == 1u) st29(p1, Fr29::reduce32(Fr29::sub_l(a, b)));      // last stage: twiddle 1
                else st29(p1, Fr29::mul(ld29(small29 + 9u * (pos << tshift)), Fr29::sub_l(a, b)));
            } else {
                Fr29 tb_ = half == 1u ? b : Fr29::mul(b, ld29(small29 + 9u * (pos << tshift)));
                st29(p0, Fr29::reduce32(Fr29::add_l(a, tb_)));
                st29(p1, Fr29::reduce32(Fr29::sub_l(a, tb_)));
            }
        }
        __syncthreads();
    }
}

template <bool DIF>
__global__ __launch_bounds__(256) void k_ntt_pass29(PassArgs29 A) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32* tile = (u32*)smem_raw;  // element i at tile + 9 i (odd stride: conflict-free)
    const int kb = A.kb, lo = A.lo, clog = A.clog;
    const u32 F = 1u << kb, C = 1u << clog;
    u32 hi, l0;
    if (lo > 0) {
        u32 lgroups = (1u << lo) >> clog;
        hi = blockIdx.x / lgroups;
        l0 = (blockIdx.x % lgroups) << clog;
    } else {
        hi = blockIdx.x << clog;
        l0 = 0;
    }
    const u32 total = F << clog;
    for (u32 idx = threadIdx.x; idx < total; idx += 256u) {
        u32 m, c, p, li;
        if (lo > 0) { c = idx & (C - 1u); m = idx >> clog; p = (hi << (lo + kb)) + (m << lo) + l0 + c; li = m * C + c; }
        else { m = idx & (F - 1u); c = idx >> kb; p = ((hi + c) << kb) + m; li = c * F + m; }
        const Fr raw = A.src[p];
        Fr29 v = A.in_gnark ? Fr29::from32<5>(raw) : Fr29::from32<0>(raw);
        if (A.scale_load) v = scale29(A, A.scale_load, v, ((p << A.p_shift) | A.p_or) + A.p_add);
        if (!DIF && lo > 0) v = Fr29::mul(v, twiddle29(A, brev(m, kb), ((l0 + c) << A.p_shift) | A.p_or, lo + A.p_shift));
        else if (!DIF && A.in_gnark && !A.scale_load) v = Fr29::reduce32(v);  // the product-free first stage adds two loaded values: keep |v| < 32r
        st29(tile + 9u * li, v);
    }
    __syncthreads();
    ntt_stages29<DIF>(tile, A.small29, kb, lo, clog);
    for (u32 idx = threadIdx.x; idx < total; idx += 256u) {
        u32 m, c, p, li;
        if (lo > 0) { c = idx & (C - 1u); m = idx >> clog; p = (hi << (lo + kb)) + (m << lo) + l0 + c; li = m * C + c; }
        else { m = idx & (F - 1u); c = idx >> kb; p = ((hi + c) << kb) + m; li = c * F + m; }
        Fr29 v = ld29(tile + 9u * li);
        if (DIF && lo > 0) v = Fr29::mul(v, twiddle29(A, brev(m, kb), ((l0 + c) << A.p_shift) | A.p_or, lo + A.p_shift));
        if (A.scale_store) v = scale29(A, A.scale_store, v, ((p << A.p_shift) | A.p_or) + A.p_add);
        if (A.sub) v = Fr29::sub_l(v, Fr29::from32<0>(A.sub[p]));      // a product's (-r, 2r) minus [0, 4r): inside what to32_div32 / reduce32_pos take
        A.x[p] = A.out_gnark ? Fr29::to32_div32(v) : Fr29::reduce32_pos(v).pack32();
    }
}

// computeH runs, per vector, an inverse DIF transform (fields high -> low) and then a forward coset DIT transform (fields low -> high): the
// last pass of the first and the first pass of the second work on the SAME tiles of the lowest field (lo == 0: contiguous groups of 2^kb
// elements).  This kernel does both on one tile — DIF stages with the inverse table, the coset scale g^rev(p) / N, DIT stages with the
// forward table: one load and one store instead of two of each, no round trip through gnark's form in between (21 -> 18 launches and
// 6 x 2 GiB less traffic per computeH at 2^26).  A = the DIF pass's arguments (its x / src, small29 = inverse table), B = the DIT pass's
// (scale tables, small29 = forward table, out_gnark).  Bit-identical to the two passes: only a canonicalisation in between is skipped.
__global__ __launch_bounds__(256) void k_ntt_mid29(PassArgs29 A, PassArgs29 B) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32* tile = (u32*)smem_raw;
    const int kb = A.kb, clog = A.clog;
    const u32 F = 1u << kb;
    const u32 hi = blockIdx.x << clog;
    const u32 total = F << clog;
    for (u32 idx = threadIdx.x; idx < total; idx += 256u) {
        const u32 m = idx & (F - 1u), c = idx >> kb, p = ((hi + c) << kb) + m, li = c * F + m;
        const Fr raw = A.src[p];
        st29(tile + 9u * li, A.in_gnark ? Fr29::from32<5>(raw) : Fr29::from32<0>(raw));
    }
    __syncthreads();
    ntt_stages29<true>(tile, A.small29, kb, 0, clog);
    for (u32 idx = threadIdx.x; idx < total; idx += 256u) {
        const u32 m = idx & (F - 1u), c = idx >> kb, p = ((hi + c) << kb) + m, li = c * F + m;
        st29(tile + 9u * li, scale29(B, B.scale_load, ld29(tile + 9u * li), p));
    }
    __syncthreads();
    ntt_stages29<false>(tile, B.small29, kb, 0, clog);
    for (u32 idx = threadIdx.x; idx < total; idx += 256u) {
        const u32 m = idx & (F - 1u), c = idx >> kb, p = ((hi + c) << kb) + m, li = c * F + m;
        const Fr29 v = ld29(tile + 9u * li);
        B.x[p] = B.out_gnark ? Fr29::to32_div32(v) : Fr29::reduce32_pos(v).pack32();
    }
}

// The other fusable place of computeH: the LAST pass of the three forward coset transforms (DIT, highest field), the pointwise step
// h = (a b - c) / (g^D - 1) and the FIRST pass of the inverse coset transform of h (DIF, highest field) all work on the same tiles of the
// highest field.  One kernel: for a, then b, then c — load with the DIT pre-twiddle, DIT stages — with the running product kept in
// registers (a thread reads back exactly the LDS slots it loaded, so no barrier separates the vectors), then h into the tile, DIF stages,
// store with the DIF post-twiddle.  Saves the three stores, the pointwise kernel (two 32-bit products per element become two 29-bit
// ones), one load, and three launches; with k_ntt_mid29: 21 + 1 -> 15 launches and 14 x 2 GiB less traffic per computeH at 2^26.
// T = the DIT pass's arguments for a (xb, xc: the other two vectors); I = the DIF pass's arguments for a; den29 = 32 / (g^D - 1).
// NV = 2 ("ntt_h" 1): a and b only, h' = a b / (g^D - 1) — c never goes to the coset, its coefficients are subtracted at the very end (compute_h_dev)
template <int PER, int NV>   // tile elements per thread: 2 or 4; vectors: 3 or 2
__global__ __launch_bounds__(256) void k_ntt_top29(PassArgs29 T, Fr* xb, Fr* xc, PassArgs29 I, Fr den29) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    u32* tile = (u32*)smem_raw;
    const int kb = T.kb, lo = T.lo, clog = T.clog;
    const u32 C = 1u << clog;
    const u32 lgroups = (1u << lo) >> clog;
    const u32 hi = blockIdx.x / lgroups, l0 = (blockIdx.x % lgroups) << clog;
    Fr29 acc[PER];
    Fr* const xs[3] = {T.x, xb, xc};
#pragma unroll 1
    for (int v = 0; v < NV; ++v) {
        const Fr* x = xs[v];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const u32 idx = threadIdx.x + 256u * k;
            const u32 c = idx & (C - 1u), m = idx >> clog, p = (hi << (lo + kb)) + (m << lo) + l0 + c;
            Fr29 e = Fr29::from32<0>(x[p]);
            e = Fr29::mul(e, twiddle29(T, brev(m, kb), l0 + c, lo));
            st29(tile + 9u * idx, e);
        }
        __syncthreads();
        ntt_stages29<false>(tile, T.small29, kb, lo, clog);
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const Fr29 e = ld29(tile + 9u * (threadIdx.x + 256u * k));
            if (v == 0) acc[k] = e;
            else if (v == 1) { acc[k] = Fr29::mul(acc[k], e); if (NV == 2) acc[k] = Fr29::mul(Fr29::from32<0>(den29), acc[k]); }
            else acc[k] = Fr29::mul(Fr29::from32<0>(den29), Fr29::sub_l(acc[k], e));
        }
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) st29(tile + 9u * (threadIdx.x + 256u * k), acc[k]);
    __syncthreads();
    ntt_stages29<true>(tile, I.small29, kb, lo, clog);
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const u32 idx = threadIdx.x + 256u * k;
        const u32 c = idx & (C - 1u), m = idx >> clog, p = (hi << (lo + kb)) + (m << lo) + l0 + c;
        Fr29 e = ld29(tile + 9u * idx);
        e = Fr29::mul(e, twiddle29(I, brev(m, kb), l0 + c, lo));
        I.x[p] = Fr29::reduce32_pos(e).pack32();
    }
}

// packed 2^261-form table -> 9-limb rows
__global__ void k_unpack29(const Fr* in, u32* out, u32 count) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fr29 v = Fr29::from32<0>(in[i]);
    for (int k = 0; k < 9; ++k) out[9u * i + k] = v.l[k];
}

// a = (a*b - c) * den (c may be null: a = a*b * den)
__global__ __launch_bounds__(256) void k_h_pointwise(Fr* a, const Fr* b, const Fr* c, Fr den, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Fr ab = Fr::mul(a[i], b[i]);
    a[i] = Fr::mul(c ? Fr::sub(ab, c[i]) : ab, den);
}

// ------------------------------------------------------------------------------------------------ host side
struct Field { int lo, kb; };
static int plan_fields(int n, Field* f) {  // same rule as tools/ntt_model.py plan_fields(kb_low=8, kb_max=9)
    const int kb_low = 8, kb_max = 9;
    if (n <= kb_low) { f[0] = {0, n}; return 1; }
    int rest = n - kb_low;
    int k = (rest + kb_max - 1) / kb_max;
    int base = rest / k, extra = rest % k;
    f[0] = {0, kb_low};
    int lo = kb_low;
    for (int i = 0; i < k; ++i) {
        int kb = base + (i < extra ? 1 : 0);
        f[1 + i] = {lo, kb};
        lo += kb;
    }
    return 1 + k;
}

static int32_t make_table(zkpor_ctx* ctx, const Fr& base, const Fr& mult, int shift, u32 count, Fr* out) {
    hipLaunchKernelGGL(k_pow_table, dim3((count + 255) / 256), dim3(256), 0, ctx->stream, base, mult, shift, count, out);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}

int32_t ntt_domain_get(zkpor_ctx* ctx, int n, NttDomain** out) {
    if (n < 1 || n > 28) { ctx->err = "ntt: log2 size must be in [1,28]"; return ZKPOR_E_ARG; }
    auto it = ctx->ntt_domains.find(n);
    if (it != ctx->ntt_domains.end()) { *out = (NttDomain*)it->second; return ZKPOR_OK; }
    NttDomain* d = new NttDomain();
    d->n = n;
    d->tb = (n + 1) / 2;
    const u32 nlo = 1u << d->tb, nhi = 1u << (n - d->tb);
    size_t total = (size_t)4 * (nlo + nhi) + (size_t)2 * nhi + 512;
    ZK_HIP(ctx, hipMalloc((void**)&d->mem, total * sizeof(Fr)));
    Fr* p = d->mem;
    auto take = [&](size_t cnt) { Fr* r = p; p += cnt; return r; };
    d->tw_lo = take(nlo); d->tw_hi = take(nhi); d->twi_lo = take(nlo); d->twi_hi = take(nhi);
    d->g_lo = take(nlo); d->g_hi = take(nhi); d->gi_lo = take(nlo); d->gi_hi = take(nhi);
    d->g_hi_ninv = take(nhi); d->gi_hi_ninv = take(nhi);
    d->small_fwd = take(256); d->small_inv = take(256);
    Fr w28 = fr_const(W28_M);
    Fr w = w28;
    for (int i = n; i < 28; ++i) w = Fr::sqr(w);
    Fr wi = Fr::inv(w);
    Fr w512 = w28;
    for (int i = 9; i < 28; ++i) w512 = Fr::sqr(w512);
    Fr w512i = Fr::inv(w512);
    Fr g = fr_const(FIVE_M), gi = Fr::inv(g);
    Fr one = Fr::one();
    // 2^n in Montgomery form, then its inverse
    Fr two = Fr::add(one, one), pw = one;
    for (int i = 0; i < n; ++i) pw = Fr::mul(pw, two);
    d->n_inv = Fr::inv(pw);
    d->den = Fr::inv(Fr::sub(Fr::pow_u64(g, (u64)1 << n), one));
    ZK_TRY(make_table(ctx, w, one, 0, nlo, d->tw_lo));
    ZK_TRY(make_table(ctx, w, one, d->tb, nhi, d->tw_hi));
    ZK_TRY(make_table(ctx, wi, one, 0, nlo, d->twi_lo));
    ZK_TRY(make_table(ctx, wi, one, d->tb, nhi, d->twi_hi));
    ZK_TRY(make_table(ctx, g, one, 0, nlo, d->g_lo));
    ZK_TRY(make_table(ctx, g, one, d->tb, nhi, d->g_hi));
    ZK_TRY(make_table(ctx, gi, one, 0, nlo, d->gi_lo));
    ZK_TRY(make_table(ctx, gi, one, d->tb, nhi, d->gi_hi));
    ZK_TRY(make_table(ctx, g, d->n_inv, d->tb, nhi, d->g_hi_ninv));
    ZK_TRY(make_table(ctx, gi, d->n_inv, d->tb, nhi, d->gi_hi_ninv));
    ZK_TRY(make_table(ctx, w512, one, 0, 256, d->small_fwd));
    ZK_TRY(make_table(ctx, w512i, one, 0, 256, d->small_inv));
    // tabulated inter-pass twiddles per field (forward and inverse): 2 x 32 B x 2^(lo+kb) bytes each; skipped when the
    // device is short of memory (the kernels then fall back to the two-table product)
    {
        Field f[8];
        int nf = plan_fields(n, f);
        for (int i = 0; i < nf && i < 8; ++i) {
            d->full_fwd[i] = d->full_inv[i] = nullptr;
            if (f[i].lo == 0) continue;
            size_t cnt = (size_t)1 << (f[i].lo + f[i].kb);
            int s0 = n - f[i].lo - f[i].kb;
            Fr *a = nullptr, *b = nullptr;
            if (hipMalloc((void**)&a, cnt * sizeof(Fr)) != hipSuccess) { (void)hipGetLastError(); continue; }
            if (hipMalloc((void**)&b, cnt * sizeof(Fr)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(a); continue; }
            unsigned blocks = (unsigned)((cnt + 255) / 256);
            hipLaunchKernelGGL(k_twiddle_full, dim3(blocks), dim3(256), 0, ctx->stream, d->tw_lo, d->tw_hi, d->tb, f[i].lo, f[i].kb, s0, a);
            hipLaunchKernelGGL(k_twiddle_full, dim3(blocks), dim3(256), 0, ctx->stream, d->twi_lo, d->twi_hi, d->tb, f[i].lo, f[i].kb, s0, b);
            ZK_KERNEL_CHECK(ctx);
            d->full_fwd[i] = a; d->full_inv[i] = b;
        }
    }
    // ---- constants of the 29-bit kernel: everything multiplied by 32 (Montgomery radix 2^256 -> 2^261)
    {
        Fr c32 = one;
        for (int i = 0; i < 5; ++i) c32 = Fr::add(c32, c32);
        size_t total29 = (size_t)4 * nlo + (size_t)6 * nhi + 512 + 2 * 288;  // 288 Fr = 256 x 9 words
        d->have29 = false;
        if (hipMalloc((void**)&d->mem29, total29 * sizeof(Fr)) == hipSuccess) {
            Fr* q = d->mem29;
            auto take29 = [&](size_t cnt) { Fr* r = q; q += cnt; return r; };
            d->g_lo29 = take29(nlo); d->gi_lo29 = take29(nlo);
            d->g_hi29 = take29(nhi); d->gi_hi29 = take29(nhi); d->g_hi_ninv29 = take29(nhi); d->gi_hi_ninv29 = take29(nhi);
            Fr* tw_hi32 = take29(nhi); Fr* twi_hi32 = take29(nhi);
            d->tw_hi29 = tw_hi32; d->twi_hi29 = twi_hi32;
            d->tw_lo29 = take29(nlo); d->twi_lo29 = take29(nlo);
            Fr* sm_f = take29(256); Fr* sm_i = take29(256);
            d->small_fwd29 = (u32*)take29(288); d->small_inv29 = (u32*)take29(288);
            ZK_TRY(make_table(ctx, g, c32, 0, nlo, d->g_lo29));
            ZK_TRY(make_table(ctx, gi, c32, 0, nlo, d->gi_lo29));
            ZK_TRY(make_table(ctx, g, c32, d->tb, nhi, d->g_hi29));
            ZK_TRY(make_table(ctx, gi, c32, d->tb, nhi, d->gi_hi29));
            ZK_TRY(make_table(ctx, g, Fr::mul(d->n_inv, c32), d->tb, nhi, d->g_hi_ninv29));
            ZK_TRY(make_table(ctx, gi, Fr::mul(d->n_inv, c32), d->tb, nhi, d->gi_hi_ninv29));
            ZK_TRY(make_table(ctx, w, c32, d->tb, nhi, tw_hi32));
            ZK_TRY(make_table(ctx, wi, c32, d->tb, nhi, twi_hi32));
            ZK_TRY(make_table(ctx, w, c32, 0, nlo, d->tw_lo29));
            ZK_TRY(make_table(ctx, wi, c32, 0, nlo, d->twi_lo29));
            ZK_TRY(make_table(ctx, w512, c32, 0, 256, sm_f));
            ZK_TRY(make_table(ctx, w512i, c32, 0, 256, sm_i));
            hipLaunchKernelGGL(k_unpack29, dim3(1), dim3(256), 0, ctx->stream, sm_f, d->small_fwd29, 256u);
            hipLaunchKernelGGL(k_unpack29, dim3(1), dim3(256), 0, ctx->stream, sm_i, d->small_inv29, 256u);
            ZK_KERNEL_CHECK(ctx);
            d->n_inv29 = Fr::mul(d->n_inv, c32);
            bool ok = true;
            Field f[8];
            int nf = plan_fields(n, f);
            for (int i = 0; i < nf && i < 8 && ok; ++i) {
                if (f[i].lo == 0) continue;
                size_t cnt = (size_t)1 << (f[i].lo + f[i].kb);
                int s0 = n - f[i].lo - f[i].kb;
                Fr *a = nullptr, *b = nullptr;
                if (hipMalloc((void**)&a, cnt * sizeof(Fr)) != hipSuccess) { (void)hipGetLastError(); ok = false; break; }
                if (hipMalloc((void**)&b, cnt * sizeof(Fr)) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(a); ok = false; break; }
                unsigned blocks = (unsigned)((cnt + 255) / 256);
                hipLaunchKernelGGL(k_twiddle_full, dim3(blocks), dim3(256), 0, ctx->stream, d->tw_lo, tw_hi32, d->tb, f[i].lo, f[i].kb, s0, a);
                hipLaunchKernelGGL(k_twiddle_full, dim3(blocks), dim3(256), 0, ctx->stream, d->twi_lo, twi_hi32, d->tb, f[i].lo, f[i].kb, s0, b);
                ZK_KERNEL_CHECK(ctx);
                d->full_fwd29[i] = a; d->full_inv29[i] = b;
            }
            d->have29 = ok;
        } else {
            (void)hipGetLastError();
        }
    }
    ctx->ntt_domains[n] = d;
    *out = d;
    return ZKPOR_OK;
}

// Fault injection for the tests of the h check (tests/test_fullsize_gpu.py): flips one bit of one tabulated inter-pass twiddle of
// the highest field of the 2^n domain (inverse table of the 29-bit kernel when present, else of the 32-bit one).  Calling it a
// second time restores the table.  Never called by the product path (zkpor_set_param "debug_ntt_fault").
__global__ void k_flip_bit(u32* word) { *word ^= 4u; }
int32_t ntt_debug_fault(zkpor_ctx* ctx, int n) {
    NttDomain* d;
    ZK_TRY(ntt_domain_get(ctx, n, &d));
    Field f[8];
    const int nf = plan_fields(n, f);
    if (nf < 2) { ctx->err = "debug_ntt_fault: the domain has a single field (no inter-pass twiddles)"; return ZKPOR_E_ARG; }
    Fr* t = (ctx->ntt_variant == 1 && d->have29) ? d->full_inv29[nf - 1] : d->full_inv[nf - 1];
    if (!t) { ctx->err = "debug_ntt_fault: no tabulated twiddles for this domain"; return ZKPOR_E_STATE; }
    const size_t cnt = (size_t)1 << (f[nf - 1].lo + f[nf - 1].kb);
    hipLaunchKernelGGL(k_flip_bit, dim3(1), dim3(1), 0, ctx->stream, (u32*)(t + (cnt / 3)) + 1);
    ZK_KERNEL_CHECK(ctx);
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKPOR_OK;
}

void ntt_domains_free(zkpor_ctx* ctx) {
    for (auto& kv : ctx->ntt_domains) {
        NttDomain* d = (NttDomain*)kv.second;
        if (d->mem) (void)hipFree(d->mem);
        if (d->mem29) (void)hipFree(d->mem29);
        for (int i = 0; i < 8; ++i) { if (d->full_fwd29[i]) (void)hipFree(d->full_fwd29[i]); if (d->full_inv29[i]) (void)hipFree(d->full_inv29[i]); }
        for (int i = 0; i < 8; ++i) { if (d->full_fwd[i]) (void)hipFree(d->full_fwd[i]); if (d->full_inv[i]) (void)hipFree(d->full_inv[i]); }
        delete d;
    }
    ctx->ntt_domains.clear();
}

// scale codes: 0 none, 1 constant, 2 g^p, 3 g^rev(p); g tables chosen by the caller
struct ScaleSpec { int mode = 0; const Fr* g_lo = nullptr; const Fr* g_hi = nullptr; Fr konst; };

static int32_t run_passes29(zkpor_ctx* ctx, NttDomain* d, Fr* x, bool inverse, bool dif, const ScaleSpec& first_load,
                            const ScaleSpec& last_store, const Fr* src, int step_lo = 0, int step_hi = 99, PassArgs29* only_args = nullptr, const Fr* last_sub = nullptr);
// src (optional): the transform reads its input from there and leaves it untouched; x receives every pass's output
static int32_t run_passes(zkpor_ctx* ctx, NttDomain* d, Fr* x, bool inverse, bool dif, const ScaleSpec& first_load,
                          const ScaleSpec& last_store, const Fr* src = nullptr) {
    Field f[8];
    int nf = plan_fields(d->n, f);
    if (ctx->ntt_variant == 1 && d->have29 && !(first_load.mode && last_store.mode && nf == 1)) return run_passes29(ctx, d, x, inverse, dif, first_load, last_store, src);
    if (src && src != x) ZK_HIP(ctx, hipMemcpyAsync(x, src, sizeof(Fr) << d->n, hipMemcpyDeviceToDevice, ctx->stream));   // the 32-bit kernel works in place only
    for (int step = 0; step < nf; ++step) {
        const Field& fl = dif ? f[nf - 1 - step] : f[step];
        PassArgs A;
        A.x = x; A.n = d->n; A.lo = fl.lo; A.kb = fl.kb;
        int cmax = fl.kb >= 9 ? 2 : 3;  // tile <= 64 KiB
        int avail = fl.lo > 0 ? fl.lo : d->n - fl.kb;
        A.clog = avail < cmax ? avail : cmax;
        A.small = inverse ? d->small_inv : d->small_fwd;
        A.tw_lo = inverse ? d->twi_lo : d->tw_lo;
        A.tw_hi = inverse ? d->twi_hi : d->tw_hi;
        {
            int fi = dif ? nf - 1 - step : step;
            A.tw_full = inverse ? d->full_inv[fi] : d->full_fwd[fi];
        }
        A.tb = d->tb;
        A.scale_load = 0; A.scale_store = 0; A.g_lo = nullptr; A.g_hi = nullptr; A.konst = Fr::one();
        if (step == 0 && first_load.mode) { A.scale_load = first_load.mode; A.g_lo = first_load.g_lo; A.g_hi = first_load.g_hi; A.konst = first_load.konst; }
        if (step == nf - 1 && last_store.mode) {
            if (A.scale_load && (A.g_lo != last_store.g_lo && last_store.mode != 1 && A.scale_load != 1)) { ctx->err = "ntt: conflicting scale tables"; return ZKPOR_E_ARG; }
            A.scale_store = last_store.mode;
            if (last_store.mode != 1) { A.g_lo = last_store.g_lo; A.g_hi = last_store.g_hi; } else A.konst = last_store.konst;
        }
        u32 blocks = (u32)(((size_t)1 << d->n) >> (fl.kb + A.clog));
        size_t smem = ((size_t)sizeof(Fr) << fl.kb) << A.clog;
        if (dif) hipLaunchKernelGGL(k_ntt_pass<true>, dim3(blocks), dim3(256), smem, ctx->stream, A);
        else hipLaunchKernelGGL(k_ntt_pass<false>, dim3(blocks), dim3(256), smem, ctx->stream, A);
        ZK_KERNEL_CHECK(ctx);
    }
    return ZKPOR_OK;
}

// which 2^261-form table stands in for a 2^256-form one
static const Fr* table29(const NttDomain* d, const Fr* t) {
    if (t == d->g_lo) return d->g_lo29;
    if (t == d->gi_lo) return d->gi_lo29;
    if (t == d->g_hi) return d->g_hi29;
    if (t == d->gi_hi) return d->gi_hi29;
    if (t == d->g_hi_ninv) return d->g_hi_ninv29;
    if (t == d->gi_hi_ninv) return d->gi_hi_ninv29;
    return nullptr;
}
static int32_t launch_pass29(zkpor_ctx* ctx, const PassArgs29& A, bool dif) {
    u32 blocks = (u32)(((size_t)1 << A.n) >> (A.kb + A.clog));
    size_t smem = ((size_t)36 << A.kb) << A.clog;
    if (smem > 64 * 1024) {  // beyond the default dynamic-LDS limit (gfx950 has 160 KiB per CU); per device, so no caching here
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_pass29<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_pass29<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    if (dif) hipLaunchKernelGGL(k_ntt_pass29<true>, dim3(blocks), dim3(256), smem, ctx->stream, A);
    else hipLaunchKernelGGL(k_ntt_pass29<false>, dim3(blocks), dim3(256), smem, ctx->stream, A);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}
static int32_t run_passes29(zkpor_ctx* ctx, NttDomain* d, Fr* x, bool inverse, bool dif, const ScaleSpec& first_load,
                            const ScaleSpec& last_store, const Fr* src, int step_lo, int step_hi, PassArgs29* only_args, const Fr* last_sub) {
    Field f[8];
    int nf = plan_fields(d->n, f);
    Fr c32 = Fr::one();
    for (int i = 0; i < 5; ++i) c32 = Fr::add(c32, c32);
    if (step_hi > nf) step_hi = nf;
    for (int step = step_lo; step < step_hi; ++step) {
        const int fi = dif ? nf - 1 - step : step;
        const Field& fl = f[fi];
        PassArgs29 A;
        A.x = x; A.src = (step == 0 && src) ? src : x; A.n = d->n; A.lo = fl.lo; A.kb = fl.kb;
        int cmax = ctx->ntt_tile_log - fl.kb;  // default: tile of <= 1024 elements x 36 B, four blocks (16 waves) per CU
        if (cmax < 0) cmax = 0;
        int avail = fl.lo > 0 ? fl.lo : d->n - fl.kb;
        A.clog = avail < cmax ? avail : cmax;
        A.small29 = inverse ? d->small_inv29 : d->small_fwd29;
        A.tw_full = inverse ? d->full_inv29[fi] : d->full_fwd29[fi];
        A.tw_lo = inverse ? d->twi_lo29 : d->tw_lo29; A.tw_hi = inverse ? d->twi_hi29 : d->tw_hi29;
        // "ntt_twiddles" 1: the fields whose table does not fit the L2 (the highest field of a 2^26 domain: 2 GiB per direction) generate their twiddles
        // (2: every field, whatever its size — the tests' way to reach the generated form at small sizes)
        if ((ctx->ntt_twiddles == 1 && ((size_t)32 << (fl.lo + fl.kb)) > ((size_t)16 << 20)) || ctx->ntt_twiddles == 2) A.tw_full = nullptr;
        A.tb = d->tb;
        A.scale_load = 0; A.scale_store = 0; A.g_lo = nullptr; A.g_hi = nullptr; A.konst = c32;
        A.in_gnark = step == 0; A.out_gnark = step == nf - 1;
        A.n_glob = d->n; A.p_shift = 0; A.p_or = 0; A.p_add = 0;
        if (step == 0 && first_load.mode) {
            A.scale_load = first_load.mode;
            if (first_load.mode == 1) A.konst = Fr::mul(first_load.konst, c32);
            else { A.g_lo = table29(d, first_load.g_lo); A.g_hi = table29(d, first_load.g_hi); }
        }
        if (step == nf - 1 && last_store.mode) {
            if (A.scale_load) { ctx->err = "ntt: scale on both ends of one pass"; return ZKPOR_E_ARG; }
            A.scale_store = last_store.mode;
            if (last_store.mode == 1) A.konst = Fr::mul(last_store.konst, c32);
            else { A.g_lo = table29(d, last_store.g_lo); A.g_hi = table29(d, last_store.g_hi); }
        }
        if ((A.scale_load > 1 || A.scale_store > 1) && (!A.g_lo || !A.g_hi)) { ctx->err = "ntt: no 2^261-form table for this scale"; return ZKPOR_E_ARG; }
        if (step == nf - 1) A.sub = last_sub;
        if (only_args) { *only_args = A; return ZKPOR_OK; }   // the caller launches a fused kernel with these arguments
        ZK_TRY(launch_pass29(ctx, A, dif));
    }
    return ZKPOR_OK;
}

// inverse DIF transform followed by forward coset DIT transform of one vector, as computeH needs them, with the two passes over the lowest
// field fused (k_ntt_mid29).  Falls back to the two plain transforms when the fusion does not apply.
static bool ntt_fusable(zkpor_ctx* ctx, NttDomain* d) {
    Field f[8];
    const int nf = plan_fields(d->n, f);
    if (!(ctx->ntt_variant == 1 && d->have29 && ctx->ntt_fuse && nf >= 2)) return false;
    const int top = nf - 1;
    int cmax = ctx->ntt_tile_log - f[top].kb;
    if (cmax < 0) cmax = 0;
    const int clog = f[top].lo < cmax ? f[top].lo : cmax;
    const int tl = f[top].kb + clog;             // log2 of the top field's tile: the fused top kernel keeps 2 or 4 elements per thread
    return tl == 9 || tl == 10;
}
// skip_top: leave out the DIT pass of the highest field (the caller runs k_ntt_top29 over all three vectors instead)
static int32_t run_inverse_then_coset_forward(zkpor_ctx* ctx, NttDomain* d, Fr* x, const ScaleSpec& pre, const Fr* src, bool skip_top) {
    Field f[8];
    const int nf = plan_fields(d->n, f);
    ScaleSpec none;
    if (!ntt_fusable(ctx, d)) {
        ZK_TRY(run_passes(ctx, d, x, true, true, none, none, src));
        return run_passes(ctx, d, x, false, false, pre, none);
    }
    PassArgs29 A, B;
    ZK_TRY(run_passes29(ctx, d, x, true, true, none, none, src, 0, nf - 1));                   // DIF: every field but the lowest
    ZK_TRY(run_passes29(ctx, d, x, true, true, none, none, src, nf - 1, nf, &A));             // the lowest field's DIF pass: arguments only
    ZK_TRY(run_passes29(ctx, d, x, false, false, pre, none, nullptr, 0, 1, &B));              // the lowest field's DIT pass: arguments only
    A.in_gnark = 0; A.out_gnark = 0;                                                            // nf >= 2: the DIF pass is not the first, the DIT pass not the last
    if (A.kb != B.kb || A.clog != B.clog || A.lo != 0 || B.lo != 0) { ctx->err = "ntt: fused pass geometry mismatch"; return ZKPOR_E_STATE; }
    {
        const u32 blocks = (u32)(((size_t)1 << d->n) >> (A.kb + A.clog));
        const size_t smem = ((size_t)36 << A.kb) << A.clog;
        if (smem > 64 * 1024) ZK_HIP(ctx, hipFuncSetAttribute((const void*)k_ntt_mid29, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(k_ntt_mid29, dim3(blocks), dim3(256), smem, ctx->stream, A, B);
        ZK_KERNEL_CHECK(ctx);
    }
    return run_passes29(ctx, d, x, false, false, pre, none, nullptr, 1, skip_top ? nf - 1 : nf);   // DIT: the fields above the lowest
}

// ---- one transform spread over W = 2^wlog GPUs (index algebra modelled in tools/ntt_model.py fft_sharded) ----
// x is this rank's local array of 2^(n - wlog) elements.  Every field pass is local under one of two distributions of the
// memory position p: D_low (rank = low wlog bits, local index p >> wlog) for the fields above the lowest, D_high (rank = top
// wlog bits, local index p mod 2^(n-wlog)) for the lowest field.  stage 0 runs the passes before the exchange, stage 1 the ones
// after it (DIF: upper fields under D_low, then the lowest under D_high; DIT: the reverse).  The passes are the unsharded kernels
// on the local array; only the exponents of the inter-pass twiddles and of the scale use the global position.
// last_packed: the transform's last pass leaves the packed 2^261 form instead of gnark's; last_sub: it subtracts last_sub[p] (that form) behind its scale
static int32_t ntt_shard_stage(zkpor_ctx* ctx, NttDomain* d, Fr* x, int wlog, int rank, bool inverse, bool dif,
                               const ScaleSpec& first_load, const ScaleSpec& last_store, int stage, bool last_packed = false, const Fr* last_sub = nullptr) {
    Field f[8];
    const int nf = plan_fields(d->n, f);
    if (!d->have29 || ctx->ntt_variant != 1) { ctx->err = "ntt: the sharded transform needs the 29-bit kernels"; return ZKPOR_E_STATE; }
    if (nf < 2 || wlog < 1 || wlog >= f[0].kb || wlog > f[nf - 1].kb || rank < 0 || rank >= (1 << wlog)) {
        ctx->err = "ntt: this size cannot be split over that many ranks"; return ZKPOR_E_ARG;
    }
    const int nl = d->n - wlog;
    Fr c32 = Fr::one();
    for (int i = 0; i < 5; ++i) c32 = Fr::add(c32, c32);
    for (int step = 0; step < nf; ++step) {
        const int fi = dif ? nf - 1 - step : step;
        const bool high = fi == 0;                       // the lowest field runs under D_high
        const int st = dif ? (high ? 1 : 0) : (high ? 0 : 1);
        if (st != stage) continue;
        const Field& fl = f[fi];
        PassArgs29 A;
        A.x = x; A.src = x; A.n = nl; A.kb = fl.kb;
        A.lo = high ? 0 : fl.lo - wlog;
        A.n_glob = d->n;
        A.p_shift = high ? 0 : wlog; A.p_or = high ? 0u : (u32)rank; A.p_add = high ? ((u32)rank << nl) : 0u;
        int cmax = ctx->ntt_tile_log - fl.kb;
        if (cmax < 0) cmax = 0;
        int avail = A.lo > 0 ? A.lo : nl - fl.kb;
        A.clog = avail < cmax ? avail : cmax;
        A.small29 = inverse ? d->small_inv29 : d->small_fwd29;
        A.tw_full = inverse ? d->full_inv29[fi] : d->full_fwd29[fi];
        A.tw_lo = inverse ? d->twi_lo29 : d->tw_lo29; A.tw_hi = inverse ? d->twi_hi29 : d->tw_hi29;   // (a sharded transform always reads its tables)
        A.tb = d->tb;
        A.scale_load = 0; A.scale_store = 0; A.g_lo = nullptr; A.g_hi = nullptr; A.konst = c32;
        A.in_gnark = step == 0; A.out_gnark = step == nf - 1;
        if (step == 0 && first_load.mode) {
            A.scale_load = first_load.mode;
            if (first_load.mode == 1) A.konst = Fr::mul(first_load.konst, c32);
            else { A.g_lo = table29(d, first_load.g_lo); A.g_hi = table29(d, first_load.g_hi); }
        }
        if (step == nf - 1 && last_store.mode) {
            A.scale_store = last_store.mode;
            if (last_store.mode == 1) A.konst = Fr::mul(last_store.konst, c32);
            else { A.g_lo = table29(d, last_store.g_lo); A.g_hi = table29(d, last_store.g_hi); }
        }
        if ((A.scale_load > 1 || A.scale_store > 1) && (!A.g_lo || !A.g_hi)) { ctx->err = "ntt: no 2^261-form table for this scale"; return ZKPOR_E_ARG; }
        if (step == nf - 1) { if (last_packed) A.out_gnark = 0; A.sub = last_sub; }
        ZK_TRY(launch_pass29(ctx, A, dif));
    }
    return ZKPOR_OK;
}

// the local transposes either side of the all-to-all: [W][M] <-> [M][W] (M = 2^(n - 2 wlog) elements per chunk)
__global__ void k_shard_transpose(const Fr* __restrict__ in, Fr* __restrict__ out, u32 mlog, u32 wlog, int interleave) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ((size_t)1 << (mlog + wlog))) return;
    const u32 W1 = (1u << wlog) - 1u;
    if (interleave) { size_t i = t >> wlog; u32 r = (u32)t & W1; out[t] = in[((size_t)r << mlog) + i]; }       // out[i*W + r] = in[r*M + i]
    else { size_t i = t & (((size_t)1 << mlog) - 1); u32 r = (u32)(t >> mlog); out[t] = in[(i << wlog) + r]; }  // out[r*M + i] = in[i*W + r]
}

// computeH of one proof over 2^wlog ranks, in four steps with an all-to-all after steps 0, 1 and 2 (split.py drives them):
//   step 0  a, b, c (D_low):  inverse DIF, upper fields                                   -> exchange a, b, c to D_high
//   step 1  a, b, c (D_high): inverse DIF lowest field; forward coset DIT lowest field    -> exchange a, b, c to D_low
//   step 2  a, b, c (D_low):  forward DIT upper fields; a = (a b - c) den; inverse coset DIF upper fields on a -> exchange a
//   step 3  a (D_high):       inverse DIF lowest field with the g^-rev(p)/N scale: this rank's block of h, in the order of pk->Z
// "ntt_h" 1 (the default, compute_h_dev): c stops at its coefficients — step 1 ends c's inverse transform with the constant den / N and leaves it, in
// D_high, where h's last pass (step 3, D_high as well, the same local positions) subtracts it.  c is NOT exchanged after step 1 (six all-to-alls of a
// vector instead of seven), step 2 does not touch it, step 3 takes it.
int32_t compute_h_shard_step(zkpor_ctx* ctx, int n, int wlog, int rank, Fr* a, Fr* b, Fr* c, int step) {
    NttDomain* d;
    ZK_TRY(ntt_domain_get(ctx, n, &d));
    ScaleSpec none, pre, post;
    pre.mode = 3; pre.g_lo = d->g_lo; pre.g_hi = d->g_hi_ninv;
    post.mode = 3; post.g_lo = d->gi_lo; post.g_hi = d->gi_hi_ninv;
    Fr* v[3] = {a, b, c};
    const size_t NL = (size_t)1 << (n - wlog);
    const bool six = ctx->ntt_h == 1;
    if (step == 0) {
        PhaseScope ps(ctx, "ntt");
        for (int i = 0; i < 3; ++i) ZK_TRY(ntt_shard_stage(ctx, d, v[i], wlog, rank, true, true, none, none, 0));
    } else if (step == 1) {
        PhaseScope ps(ctx, "ntt");
        for (int i = 0; i < (six ? 2 : 3); ++i) {
            ZK_TRY(ntt_shard_stage(ctx, d, v[i], wlog, rank, true, true, none, none, 1));
            ZK_TRY(ntt_shard_stage(ctx, d, v[i], wlog, rank, false, false, pre, none, 0));
        }
        if (six) {
            ScaleSpec kc; kc.mode = 1; kc.konst = Fr::mul(d->den, d->n_inv);
            ZK_TRY(ntt_shard_stage(ctx, d, c, wlog, rank, true, true, none, kc, 1, true));
        }
    } else if (step == 2) {
        {
            PhaseScope ps(ctx, "ntt");
            for (int i = 0; i < (six ? 2 : 3); ++i) ZK_TRY(ntt_shard_stage(ctx, d, v[i], wlog, rank, false, false, pre, none, 1));
        }
        {
            PhaseScope ps(ctx, "pointwise");
            hipLaunchKernelGGL(k_h_pointwise, dim3((unsigned)((NL + 255) / 256)), dim3(256), 0, ctx->stream, a, b, six ? nullptr : c, d->den, NL);
            ZK_KERNEL_CHECK(ctx);
        }
        PhaseScope ps(ctx, "ntt");
        ZK_TRY(ntt_shard_stage(ctx, d, a, wlog, rank, true, true, none, post, 0));
    } else if (step == 3) {
        PhaseScope ps(ctx, "ntt");
        if (six && !c) { ctx->err = "computeH shard: step 3 takes c as step 1 left it (\"ntt_h\" 1: it is not exchanged after step 1)"; return ZKPOR_E_ARG; }
        ZK_TRY(ntt_shard_stage(ctx, d, a, wlog, rank, true, true, none, post, 1, false, six ? c : nullptr));
    } else {
        ctx->err = "computeH shard: step must be 0..3"; return ZKPOR_E_ARG;
    }
    return ZKPOR_OK;
}

int32_t shard_transpose(zkpor_ctx* ctx, Fr* out, const Fr* in, int n_local, int wlog, bool interleave) {
    if (n_local < wlog) { ctx->err = "shard transpose: array smaller than the rank count"; return ZKPOR_E_ARG; }
    size_t total = (size_t)1 << n_local;
    hipLaunchKernelGGL(k_shard_transpose, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, in, out, (u32)(n_local - wlog), (u32)wlog, interleave ? 1 : 0);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}

// gnark-crypto semantics: Domain.FFT / FFTInverse(a, decimation, OnCoset?) in place on the device
int32_t ntt_dev(zkpor_ctx* ctx, Fr* d_x, int n, bool inverse, bool dif, bool on_coset) {
    NttDomain* d;
    ZK_TRY(ntt_domain_get(ctx, n, &d));
    PhaseScope ps(ctx, "ntt");
    ScaleSpec load, store;
    if (!inverse) {
        if (on_coset) { load.mode = dif ? 2 : 3; load.g_lo = d->g_lo; load.g_hi = d->g_hi; }
    } else {
        if (on_coset) { store.mode = dif ? 3 : 2; store.g_lo = d->gi_lo; store.g_hi = d->gi_hi_ninv; }
        else { store.mode = 1; store.konst = d->n_inv; }
    }
    return run_passes(ctx, d, d_x, inverse, dif, load, store);
}

// computeH in place: a,b,c hold 2^n evaluations (zero padded); on return a = h in BIT-REVERSED order.
// The 1/N of the three inverse transforms is folded into the coset pre-scale of the forward ones.
int32_t compute_h_dev(zkpor_ctx* ctx, int n, Fr* a, Fr* b, Fr* c, const Fr* a_in, const Fr* b_in, const Fr* c_in) {
    NttDomain* d;
    ZK_TRY(ntt_domain_get(ctx, n, &d));
    ScaleSpec none, pre, post;
    pre.mode = 3; pre.g_lo = d->g_lo; pre.g_hi = d->g_hi_ninv;      // g^rev(p) / N at the DIT load
    post.mode = 3; post.g_lo = d->gi_lo; post.g_hi = d->gi_hi_ninv;  // g^-rev(p) / N at the last DIF store
    {
        PhaseScope ps(ctx, "ntt");
        Fr* v[3] = {a, b, c};
        const Fr* in[3] = {a_in, b_in, c_in};     // inputs the caller wants preserved: the first pass reads them, a / b / c are the work buffers
        const bool fuse = ntt_fusable(ctx, d);
        // "ntt_h" 1 (round 6): SIX transforms instead of gnark's seven.  The inverse coset transform is linear, so
        //   h = icFFT(den (a_c b_c - c_c)) = icFFT(den a_c b_c) - den c,      c = the COEFFICIENTS of c, which its inverse transform has already produced:
        // c never goes to the coset; its DIF inverse transform ends with the constant den / N at the store, stays in the packed inter-pass form and in
        // the bit-reversed order h's own last DIF pass stores in, and that pass subtracts it behind its post-twiddle.  The same h for EVERY a, b, c (no
        // use is made of a b = c on the domain), bit for bit: tests/test_ntt_gpu.py runs both schedules against the oracle.
        const bool skip_c = fuse && ctx->ntt_h == 1;
        for (int i = 0; i < (skip_c ? 2 : 3); ++i) ZK_TRY(run_inverse_then_coset_forward(ctx, d, v[i], pre, in[i], fuse));
        if (skip_c) {
            Field f[8];
            const int nf = plan_fields(n, f);
            ScaleSpec kc; kc.mode = 1; kc.konst = Fr::mul(d->den, d->n_inv);
            PassArgs29 L;
            ZK_TRY(run_passes29(ctx, d, c, true, true, none, none, c_in, 0, nf - 1));
            ZK_TRY(run_passes29(ctx, d, c, true, true, none, kc, c_in, nf - 1, nf, &L));
            L.out_gnark = 0;                     // stays in the packed 2^261 form: read back once, by h's last pass
            ZK_TRY(launch_pass29(ctx, L, true));
        }
        if (fuse) {
            Field f[8];
            const int nf = plan_fields(n, f);
            PassArgs29 T, I;
            ZK_TRY(run_passes29(ctx, d, a, false, false, pre, none, nullptr, nf - 1, nf, &T));   // DIT, highest field: arguments only
            ZK_TRY(run_passes29(ctx, d, a, true, true, none, post, nullptr, 0, 1, &I));          // DIF of the inverse coset transform, highest field
            if (T.kb != I.kb || T.clog != I.clog || T.lo != I.lo || T.lo == 0) { ctx->err = "ntt: fused top pass geometry mismatch"; return ZKPOR_E_STATE; }
            Fr c32 = Fr::one();
            for (int i = 0; i < 5; ++i) c32 = Fr::add(c32, c32);
            const Fr den29 = Fr::mul(d->den, c32);
            const int tl = T.kb + T.clog;
            const u32 blocks = (u32)(((size_t)1 << n) >> tl);
            const size_t smem = (size_t)36 << tl;
            if (skip_c) {
                if (tl == 10) hipLaunchKernelGGL((k_ntt_top29<4, 2>), dim3(blocks), dim3(256), smem, ctx->stream, T, b, c, I, den29);
                else hipLaunchKernelGGL((k_ntt_top29<2, 2>), dim3(blocks), dim3(256), smem, ctx->stream, T, b, c, I, den29);
            } else if (tl == 10) hipLaunchKernelGGL((k_ntt_top29<4, 3>), dim3(blocks), dim3(256), smem, ctx->stream, T, b, c, I, den29);
            else hipLaunchKernelGGL((k_ntt_top29<2, 3>), dim3(blocks), dim3(256), smem, ctx->stream, T, b, c, I, den29);
            ZK_KERNEL_CHECK(ctx);
            ZK_TRY(run_passes29(ctx, d, a, true, true, none, post, nullptr, 1, nf, nullptr, skip_c ? c : nullptr));   // the remaining DIF passes of h (the last one: minus den c)
            return ZKPOR_OK;
        }
    }
    {
        PhaseScope ps(ctx, "pointwise");
        size_t N = (size_t)1 << n;
        hipLaunchKernelGGL(k_h_pointwise, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, a, b, c, d->den, N);
        ZK_KERNEL_CHECK(ctx);
    }
    {
        PhaseScope ps(ctx, "ntt");
        ZK_TRY(run_passes(ctx, d, a, true, true, none, post));
    }
    return ZKPOR_OK;
}

}  // namespace zk

using namespace zk;

extern "C" {

int32_t zkpor_fft(zkpor_ctx* ctx, uint64_t* a, int log2n, int inverse, int decimation, int on_coset) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !a || log2n < 1 || log2n > 28) return ZKPOR_E_ARG;
    size_t bytes = ((size_t)32) << log2n;
    Fr* d = nullptr;
    ZK_HIP(ctx, hipMalloc((void**)&d, bytes));
    int32_t rc = ZKPOR_OK;
    if (zk::h2d_sync(ctx, d, a, bytes) != ZKPOR_OK) { ctx->err = "H2D failed"; rc = ZKPOR_E_HIP; }
    if (rc == ZKPOR_OK) rc = ntt_dev(ctx, d, log2n, inverse != 0, decimation == 1, on_coset != 0);
    if (rc == ZKPOR_OK && hipMemcpyAsync(a, d, bytes, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "D2H failed"; rc = ZKPOR_E_HIP; }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return rc;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_fft_dev(zkpor_ctx* ctx, void* d_a, int log2n, int inverse, int decimation, int on_coset) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !d_a || log2n < 1 || log2n > 28) return ZKPOR_E_ARG;
    return ntt_dev(ctx, (Fr*)d_a, log2n, inverse != 0, decimation == 1, on_coset != 0);
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_compute_h_shard_dev(zkpor_ctx* ctx, int log2_domain, int world_log2, int rank, void* d_a, void* d_b, void* d_c, int step) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !d_a || log2_domain < 1 || log2_domain > 28) return ZKPOR_E_ARG;
    if (step < 3 && (!d_b || (!d_c && !(step == 2 && ctx->ntt_h == 1)))) return ZKPOR_E_ARG;
    return compute_h_shard_step(ctx, log2_domain, world_log2, rank, (Fr*)d_a, (Fr*)d_b, (Fr*)d_c, step);
} ZK_ABI_CATCH_IN(ctx)
int32_t zkpor_shard_transpose_dev(zkpor_ctx* ctx, void* d_out, const void* d_in, int log2_local, int world_log2, int interleave) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !d_out || !d_in || d_out == d_in || world_log2 < 1 || log2_local < 2 * world_log2) return ZKPOR_E_ARG;
    return shard_transpose(ctx, (Fr*)d_out, (const Fr*)d_in, log2_local, world_log2, interleave != 0);
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_compute_h_dev(zkpor_ctx* ctx, int log2_domain, void* d_a, void* d_b, void* d_c) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !d_a || !d_b || !d_c) return ZKPOR_E_ARG;
    return compute_h_dev(ctx, log2_domain, (Fr*)d_a, (Fr*)d_b, (Fr*)d_c);
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_compute_h(zkpor_ctx* ctx, int log2_domain, const uint64_t* a, const uint64_t* b, const uint64_t* c,
                        size_t n_constraints, uint64_t* h_out) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !a || !b || !c || !h_out || log2_domain < 1 || log2_domain > 28) return ZKPOR_E_ARG;
    size_t N = (size_t)1 << log2_domain;
    if (n_constraints > N) return ZKPOR_E_ARG;
    Fr* d = nullptr;
    ZK_HIP(ctx, hipMalloc((void**)&d, 3 * N * sizeof(Fr)));
    int32_t rc = ZKPOR_OK;
    const uint64_t* src[3] = {a, b, c};
    for (int i = 0; i < 3 && rc == ZKPOR_OK; ++i) {
        if (hipMemsetAsync(d + i * N, 0, N * sizeof(Fr), ctx->stream) != hipSuccess ||
            zk::h2d_sync(ctx, d + i * N, src[i], n_constraints * sizeof(Fr)) != ZKPOR_OK) {
            ctx->err = "H2D failed"; rc = ZKPOR_E_HIP;
        }
    }
    if (rc == ZKPOR_OK) rc = compute_h_dev(ctx, log2_domain, d, d + N, d + 2 * N);
    if (rc == ZKPOR_OK && hipMemcpyAsync(h_out, d, N * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "D2H failed"; rc = ZKPOR_E_HIP; }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return rc;
} ZK_ABI_CATCH_IN(ctx)

}  // extern "C"
