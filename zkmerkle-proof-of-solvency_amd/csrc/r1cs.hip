// Constraint evaluation on the device: a = L.w, b = R.w, c = O.w for an R1CS held in HBM — SURVEY.md §8 row f1, the step
// that feeds computeH.  In the reference this is part of gnark's solver (constraint/bn254 solver, reached through
// groth16.Prove at src/prover/prover/prover.go:269): after the wire vector w is known, every constraint's three linear
// expressions are evaluated on the CPU and the three 2^26 x 32 B vectors would have to cross PCIe per proof (6.4 GB).
// With the matrices resident (about 12 GB per tier) only w crosses.
// Layout follows gnark's compiled constraint system: a linear expression is a list of terms (coefficient id, wire id),
// coefficients live in a small shared table (most terms use 1 or -1); three CSR matrices share the table.
#include "common.cuh"
#include <memory>
#include "r1cs.cuh"

namespace zk {

struct R1csDev {
    const Fr* coeff; const uint8_t* kind;
    const uint64_t* row_ptr[3]; const u32* cid[3]; const u32* wid[3];
    const u32* perm[3]; u32 n_perm[3];     // the rows of up to R1CS_LONG_ROW terms in evaluation order (r1cs.cuh); null = natural order
};

// one thread per (matrix, row), the rows taken in the matrix's evaluation order (perm: rows of equal length and coefficient pattern side by side, so
// that the 64 rows of a wave run the same number of iterations through the same branches); the threads behind them write the zero padding computeH
// expects in rows n_constraints .. domain - 1.
// skip (may be NULL): one bit per row, set = somebody else has written a, b, c of that row already (the solver's Poseidon instruction)
__global__ __launch_bounds__(256) void k_r1cs_eval(R1csDev M, const Fr* __restrict__ w, size_t n_constraints, size_t domain,
                                                   Fr* __restrict__ a, Fr* __restrict__ b, Fr* __restrict__ c, const u32* __restrict__ skip) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    const int m = blockIdx.y;
    const u32* perm = M.perm[m];
    const size_t n_rows = perm ? (size_t)M.n_perm[m] : n_constraints;
    Fr* out = m == 0 ? a : (m == 1 ? b : c);
    if (i >= n_rows) {                                       // padding rows
        const size_t row = n_constraints + (i - n_rows);
        if (row < domain) out[row] = Fr::zero();
        return;
    }
    const size_t row = perm ? (size_t)perm[i] : i;
    if (skip && ((skip[row >> 5] >> (row & 31)) & 1u)) return;
    Fr acc = Fr::zero();
    const uint64_t t0 = M.row_ptr[m][row], t1 = M.row_ptr[m][row + 1];
    if (t1 - t0 > (uint64_t)R1CS_LONG_ROW) return;           // k_r1cs_eval_long's (natural order only: the evaluation order leaves them out)
    for (uint64_t t = t0; t < t1; ++t) {
        const u32 ci = M.cid[m][t];
        const uint8_t kind = M.kind[ci];
        if (kind == 3) continue;
        const Fr x = w[M.wid[m][t]];
        if (kind == 1) acc = Fr::add(acc, x);
        else if (kind == 2) acc = Fr::sub(acc, x);
        else acc = Fr::add(acc, Fr::mul(M.coeff[ci], x));
    }
    out[row] = acc;
}

// the rows of more than R1CS_LONG_ROW terms of one matrix, one wave each: lane l takes terms l, l + 64, ... and the 64 partial sums meet
// through shuffles (field addition is exact: the order does not matter)
__global__ __launch_bounds__(256) void k_r1cs_eval_long(R1csDev M, int m, const u32* __restrict__ rows, size_t n_rows, const Fr* __restrict__ w,
                                                        Fr* __restrict__ out, const u32* __restrict__ skip) {
    const size_t i = (size_t)blockIdx.x * 4u + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (i >= n_rows) return;
    const u32 row = rows[i];
    if (skip && ((skip[row >> 5] >> (row & 31)) & 1u)) return;
    const uint64_t t0 = M.row_ptr[m][row], t1 = M.row_ptr[m][row + 1];
    Fr acc = Fr::zero();
    for (uint64_t t = t0 + (uint64_t)lane; t < t1; t += 64u) {
        const u32 ci = M.cid[m][t];
        const uint8_t kind = M.kind[ci];
        if (kind == 3) continue;
        const Fr x = w[M.wid[m][t]];
        if (kind == 1) acc = Fr::add(acc, x);
        else if (kind == 2) acc = Fr::sub(acc, x);
        else acc = Fr::add(acc, Fr::mul(M.coeff[ci], x));
    }
    for (int off = 32; off >= 1; off >>= 1) {
        Fr o;
        for (int k = 0; k < 8; ++k) o.v[k] = (u32)__shfl_down((int)acc.v[k], off, 64);
        acc = Fr::add(acc, o);
    }
    if (lane == 0) out[row] = acc;
}

// the final check of a solved wire vector: L.w * R.w = O.w on every row; out[0] = rows that fail, out[1] = the lowest failing row
__global__ __launch_bounds__(256) void k_r1cs_check(R1csDev M, const Fr* __restrict__ w, size_t n_constraints, unsigned long long* out) {
    const size_t row = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (row >= n_constraints) return;
    Fr v[3];
    for (int m = 0; m < 3; ++m) {
        Fr acc = Fr::zero();
        const uint64_t t0 = M.row_ptr[m][row], t1 = M.row_ptr[m][row + 1];
        for (uint64_t t = t0; t < t1; ++t) {
            const u32 ci = M.cid[m][t];
            const uint8_t kind = M.kind[ci];
            if (kind == 3) continue;
            const Fr x = w[M.wid[m][t]];
            if (kind == 1) acc = Fr::add(acc, x);
            else if (kind == 2) acc = Fr::sub(acc, x);
            else acc = Fr::add(acc, Fr::mul(M.coeff[ci], x));
        }
        v[m] = acc;
    }
    if (Fr::mul(v[0], v[1]) != v[2]) { atomicAdd(&out[0], 1ull); atomicMin(&out[1], (unsigned long long)row); }
}

static void r1cs_free(zkpor_r1cs* r) {
    if (r->coeff) (void)hipFree(r->coeff);
    if (r->coeff_kind) (void)hipFree(r->coeff_kind);
    for (int m = 0; m < 3; ++m) {
        if (r->row_ptr[m]) (void)hipFree(r->row_ptr[m]);
        if (r->cid[m]) (void)hipFree(r->cid[m]);
        if (r->wid[m]) (void)hipFree(r->wid[m]);
        if (r->long_rows[m]) (void)hipFree(r->long_rows[m]);
        if (r->perm[m]) (void)hipFree(r->perm[m]);
    }
    delete r;
}

}  // namespace zk

using namespace zk;
namespace zk {
int32_t r1cs_eval_on(zkpor_ctx* ctx, zkpor_r1cs* r, const void* d_w, void* d_a, void* d_b, void* d_c, size_t domain_size, const u32* d_skip) {
    if (!ctx || !r || !d_w || !d_a || !d_b || !d_c) return ZKPOR_E_ARG;
    if (ctx->device != r->ctx->device) { ctx->err = "r1cs: the matrices live on another GPU than the context"; return ZKPOR_E_ARG; }
    if (domain_size < r->n_constraints) { ctx->err = "r1cs: domain smaller than the constraint count"; return ZKPOR_E_ARG; }
    for (int m = 0; m < 3; ++m) if (!r->row_ptr[m]) { ctx->err = "r1cs: matrix " + std::to_string(m) + " not loaded"; return ZKPOR_E_STATE; }
    if (domain_size == 0) return ZKPOR_OK;
    R1csDev M;
    M.coeff = r->coeff; M.kind = r->coeff_kind;
    const bool sorted = ctx->r1cs_order != 0;
    size_t threads = 0;
    for (int m = 0; m < 3; ++m) {
        M.row_ptr[m] = r->row_ptr[m]; M.cid[m] = r->cid[m]; M.wid[m] = r->wid[m];
        M.perm[m] = sorted ? r->perm[m] : nullptr; M.n_perm[m] = (u32)r->n_perm[m];
        const size_t t = (M.perm[m] ? r->n_perm[m] : r->n_constraints) + (domain_size - r->n_constraints);
        if (t > threads) threads = t;
    }
    PhaseScope ps(ctx, "r1cs_eval");
    hipLaunchKernelGGL(k_r1cs_eval, dim3((unsigned)((threads + 255) / 256), 3), dim3(256), 0, ctx->stream, M, (const Fr*)d_w,
                       r->n_constraints, domain_size, (Fr*)d_a, (Fr*)d_b, (Fr*)d_c, d_skip);
    ZK_KERNEL_CHECK(ctx);
    Fr* outs[3] = {(Fr*)d_a, (Fr*)d_b, (Fr*)d_c};
    for (int m = 0; m < 3; ++m) {
        if (!r->n_long[m]) continue;
        hipLaunchKernelGGL(k_r1cs_eval_long, dim3((unsigned)((r->n_long[m] + 3) / 4)), dim3(256), 0, ctx->stream, M, m, (const u32*)r->long_rows[m], r->n_long[m],
                           (const Fr*)d_w, outs[m], d_skip);
        ZK_KERNEL_CHECK(ctx);
    }
    return ZKPOR_OK;
}
void r1cs_dims(const zkpor_r1cs* r, size_t* n_constraints, size_t* n_wires, int* device) {
    *n_constraints = r->n_constraints; *n_wires = r->n_wires; *device = r->ctx->device;
}
}  // namespace zk

extern "C" {

int32_t zkpor_r1cs_create(zkpor_ctx* ctx, size_t n_constraints, size_t n_wires, const uint64_t* coeff_table, size_t n_coeff,
                          zkpor_r1cs** out) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !out || !coeff_table || n_coeff == 0 || n_wires == 0 || n_wires > 0xffffffffull) return ZKPOR_E_ARG;
    std::unique_ptr<zkpor_r1cs, void (*)(zkpor_r1cs*)> hold(new zkpor_r1cs(), r1cs_free);   // freed on every error return and on an exception stopped at the ABI
    zkpor_r1cs* r = hold.get();
    r->ctx = ctx; r->n_constraints = n_constraints; r->n_wires = n_wires; r->n_coeff = n_coeff;
    std::vector<uint8_t> kind(n_coeff, 0);
    const Fr* tab = (const Fr*)coeff_table;
    r->h_coeff.assign(tab, tab + n_coeff);
    const Fr one = Fr::one(), mone = Fr::neg(one);
    for (size_t i = 0; i < n_coeff; ++i) {
        if (tab[i].is_zero()) kind[i] = 3;
        else if (memcmp(&tab[i], &one, sizeof(Fr)) == 0) kind[i] = 1;
        else if (memcmp(&tab[i], &mone, sizeof(Fr)) == 0) kind[i] = 2;
    }
    if (hipMalloc((void**)&r->coeff, n_coeff * sizeof(Fr)) != hipSuccess || hipMalloc((void**)&r->coeff_kind, n_coeff) != hipSuccess) {
        (void)hipGetLastError(); ctx->err = "r1cs: out of device memory"; return ZKPOR_E_OOM;
    }
    if (zk::h2d_sync(ctx, r->coeff, tab, n_coeff * sizeof(Fr)) != ZKPOR_OK ||
        zk::h2d_sync(ctx, r->coeff_kind, kind.data(), n_coeff) != ZKPOR_OK) { ctx->err = "r1cs: H2D failed"; return ZKPOR_E_HIP; }
    *out = hold.release();
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
void zkpor_r1cs_destroy(zkpor_r1cs* r) try {
    ZK_ENTER(r ? r->ctx->device : -1);
    if (!r) return;
    (void)hipStreamSynchronize(r->ctx->stream);
    r1cs_free(r);
} catch (...) { zk::abi_exception("exception in zkpor_r1cs_destroy"); }
int32_t zkpor_r1cs_set_matrix(zkpor_r1cs* r, int which, const uint64_t* row_ptr, const uint32_t* coeff_ids, const uint32_t* wire_ids,
                              size_t nnz) try {
    ZK_ENTER(r ? r->ctx->device : -1);
    if (!r || which < 0 || which > 2 || !row_ptr || (nnz && (!coeff_ids || !wire_ids))) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = r->ctx;
    // validate on the host: the kernel indexes with these
    if (row_ptr[0] != 0 || row_ptr[r->n_constraints] != nnz) { ctx->err = "r1cs: row_ptr does not span the terms"; return ZKPOR_E_ARG; }
    for (size_t i = 0; i < r->n_constraints; ++i) if (row_ptr[i + 1] < row_ptr[i]) { ctx->err = "r1cs: row_ptr is not monotone"; return ZKPOR_E_ARG; }
    for (size_t t = 0; t < nnz; ++t)
        if (coeff_ids[t] >= r->n_coeff || wire_ids[t] >= r->n_wires) { ctx->err = "r1cs: term " + std::to_string(t) + " indexes outside the tables"; return ZKPOR_E_ARG; }
    if (r->row_ptr[which]) { (void)hipFree(r->row_ptr[which]); r->row_ptr[which] = nullptr; }
    if (r->cid[which]) { (void)hipFree(r->cid[which]); r->cid[which] = nullptr; }
    if (r->wid[which]) { (void)hipFree(r->wid[which]); r->wid[which] = nullptr; }
    if (r->long_rows[which]) { (void)hipFree(r->long_rows[which]); r->long_rows[which] = nullptr; }
    if (r->perm[which]) { (void)hipFree(r->perm[which]); r->perm[which] = nullptr; }
    r->n_long[which] = 0; r->n_perm[which] = 0;
    std::vector<uint32_t> longs;
    for (size_t i = 0; i < r->n_constraints; ++i) if (row_ptr[i + 1] - row_ptr[i] > (uint64_t)R1CS_LONG_ROW) longs.push_back((uint32_t)i);
    if (r->n_constraints > 0xffffffffull) { ctx->err = "r1cs: more than 2^32 constraints"; return ZKPOR_E_ARG; }
    if (hipMalloc((void**)&r->row_ptr[which], (r->n_constraints + 1) * 8) != hipSuccess || hipMalloc((void**)&r->cid[which], (nnz ? nnz : 1) * 4) != hipSuccess ||
        hipMalloc((void**)&r->wid[which], (nnz ? nnz : 1) * 4) != hipSuccess) { (void)hipGetLastError(); ctx->err = "r1cs: out of device memory"; return ZKPOR_E_OOM; }
    ZK_TRY(zk::h2d_sync(ctx, r->row_ptr[which], row_ptr, (r->n_constraints + 1) * 8));
    if (nnz) {
        ZK_TRY(zk::h2d_sync(ctx, r->cid[which], coeff_ids, nnz * 4));
        ZK_TRY(zk::h2d_sync(ctx, r->wid[which], wire_ids, nnz * 4));
    }
    if (!longs.empty()) {
        if (hipMalloc((void**)&r->long_rows[which], longs.size() * 4) != hipSuccess) { (void)hipGetLastError(); ctx->err = "r1cs: out of device memory"; return ZKPOR_E_OOM; }
        ZK_TRY(zk::h2d_sync(ctx, r->long_rows[which], longs.data(), longs.size() * 4));
        r->n_long[which] = longs.size();
    }
    {   // the evaluation order of the rows of up to R1CS_LONG_ROW terms: a stable counting sort by (term count, pattern of coefficient kinds).  The pattern of
        // a short row is its kinds themselves (2 bits per term, hashed to a byte), of a longer one the number of its generic coefficients: rows a gadget
        // emits for every user / asset / round meet in one class, in natural order.  One pass over the terms on the host, once per loaded matrix.
        const size_t n = r->n_constraints;
        std::vector<uint8_t> kind(r->n_coeff);
        const Fr one = Fr::one(), mone = Fr::neg(one);
        for (size_t i = 0; i < r->n_coeff; ++i) kind[i] = r->h_coeff[i].is_zero() ? 3 : (memcmp(&r->h_coeff[i], &one, sizeof(Fr)) == 0 ? 1 : (memcmp(&r->h_coeff[i], &mone, sizeof(Fr)) == 0 ? 2 : 0));
        std::vector<uint32_t> key(n);
        constexpr uint32_t CLASSES = (R1CS_LONG_ROW + 2u) << 8;
        std::vector<uint32_t> start(CLASSES + 1, 0);
        size_t kept = 0;
        for (size_t i = 0; i < n; ++i) {
            const uint64_t t0 = row_ptr[i], len = row_ptr[i + 1] - t0;
            if (len > (uint64_t)R1CS_LONG_ROW) { key[i] = 0xffffffffu; continue; }
            uint32_t sig = 0;
            if (len <= 16) { for (uint64_t t = 0; t < len; ++t) sig = sig * 4u + kind[coeff_ids[t0 + t]]; sig = (sig * 2654435761u) >> 24; }
            else { uint32_t g = 0; for (uint64_t t = 0; t < len; ++t) g += kind[coeff_ids[t0 + t]] == 0; sig = g > 255u ? 255u : g; }
            key[i] = ((uint32_t)len << 8) | sig;
            ++start[key[i] + 1];
            ++kept;
        }
        for (uint32_t k = 0; k < CLASSES; ++k) start[k + 1] += start[k];
        std::vector<uint32_t> perm(kept ? kept : 1);
        for (size_t i = 0; i < n; ++i) if (key[i] != 0xffffffffu) perm[start[key[i]]++] = (uint32_t)i;
        if (hipMalloc((void**)&r->perm[which], (kept ? kept : 1) * 4) != hipSuccess) { (void)hipGetLastError(); ctx->err = "r1cs: out of device memory"; return ZKPOR_E_OOM; }
        ZK_TRY(zk::h2d_sync(ctx, r->perm[which], perm.data(), (kept ? kept : 1) * 4));
        r->n_perm[which] = kept;
    }
    r->nnz[which] = nnz;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((r ? r->ctx : nullptr))
int32_t zkpor_r1cs_eval_dev(zkpor_r1cs* r, const void* d_w, void* d_a, void* d_b, void* d_c, size_t domain_size) try {
    ZK_ENTER(r ? r->ctx->device : -1);
    if (!r) return ZKPOR_E_ARG;
    return zk::r1cs_eval_on(r->ctx, r, d_w, d_a, d_b, d_c, domain_size);
} ZK_ABI_CATCH_IN((r ? r->ctx : nullptr))
/* the same queued on ANOTHER context of the GPU (a second worker's stream and timers; the matrices are only read) */
int32_t zkpor_r1cs_eval_on(zkpor_ctx* ctx, zkpor_r1cs* r, const void* d_w, void* d_a, void* d_b, void* d_c, size_t domain_size) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !r) return ZKPOR_E_ARG;
    return zk::r1cs_eval_on(ctx, r, d_w, d_a, d_b, d_c, domain_size);
} ZK_ABI_CATCH_IN(ctx)
/* every constraint against a wire vector on the device: counts[0] = rows with L.w * R.w != O.w, counts[1] = the lowest such row */
int32_t zkpor_r1cs_check_dev(zkpor_r1cs* r, const void* d_w, uint64_t counts[2]) try {
    ZK_ENTER(r ? r->ctx->device : -1);
    if (!r || !d_w || !counts) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = r->ctx;
    for (int m = 0; m < 3; ++m) if (!r->row_ptr[m]) { ctx->err = "r1cs: matrix " + std::to_string(m) + " not loaded"; return ZKPOR_E_STATE; }
    counts[0] = 0; counts[1] = ~0ull;
    if (r->n_constraints == 0) return ZKPOR_OK;
    unsigned long long* d_out = nullptr;
    ZK_HIP(ctx, hipMalloc((void**)&d_out, 16));
    int32_t rc = ZKPOR_OK;
    R1csDev M;
    M.coeff = r->coeff; M.kind = r->coeff_kind;
    for (int m = 0; m < 3; ++m) { M.row_ptr[m] = r->row_ptr[m]; M.cid[m] = r->cid[m]; M.wid[m] = r->wid[m]; M.perm[m] = nullptr; M.n_perm[m] = 0; }
    if (hipMemcpyAsync(d_out, counts, 16, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "r1cs: H2D failed"; rc = ZKPOR_E_HIP; }
    if (rc == ZKPOR_OK) {
        hipLaunchKernelGGL(k_r1cs_check, dim3((unsigned)((r->n_constraints + 255) / 256)), dim3(256), 0, ctx->stream, M, (const Fr*)d_w, r->n_constraints, d_out);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(counts, d_out, 16, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "r1cs: check failed to launch"; rc = ZKPOR_E_HIP; }
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_out);
    return rc;
} ZK_ABI_CATCH_IN((r ? r->ctx : nullptr))
/* host-buffer form for tests and small circuits: w in, a/b/c (n_constraints each) out */
int32_t zkpor_r1cs_eval(zkpor_r1cs* r, const uint64_t* w, uint64_t* a, uint64_t* b, uint64_t* c) try {
    ZK_ENTER(r ? r->ctx->device : -1);
    if (!r || !w || !a || !b || !c) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = r->ctx;
    const size_t n = r->n_constraints;
    Fr* d = nullptr;
    ZK_HIP(ctx, hipMalloc((void**)&d, (r->n_wires + 3 * (n ? n : 1)) * sizeof(Fr)));
    int32_t rc = ZKPOR_OK;
    if (zk::h2d_sync(ctx, d, w, r->n_wires * sizeof(Fr)) != ZKPOR_OK) { ctx->err = "r1cs: H2D failed"; rc = ZKPOR_E_HIP; }
    Fr* da = d + r->n_wires;
    if (rc == ZKPOR_OK) rc = zkpor_r1cs_eval_dev(r, d, da, da + n, da + 2 * n, n);
    uint64_t* outs[3] = {a, b, c};
    for (int m = 0; m < 3 && rc == ZKPOR_OK && n; ++m)
        if (hipMemcpyAsync(outs[m], da + m * n, n * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "r1cs: D2H failed"; rc = ZKPOR_E_HIP; }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d);
    return rc;
} ZK_ABI_CATCH_IN((r ? r->ctx : nullptr))

}  // extern "C"
