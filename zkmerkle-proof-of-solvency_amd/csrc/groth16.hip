// HBM-resident Groth16 proving key and the prove tail (everything groth16.Prove does after the R1CS solver):
// computeH, the A / B1 / B2 / K / Z multi-exponentiations, r/s blinding, and the Pedersen commitment MSMs.
// Reference: src/prover/prover/prover.go:269 (groth16.Prove), :285-367 (LoadSnarkParamsOnce), :201 (WriteRawTo);
// algorithm restated from bnb-chain/gnark backend/groth16/bn254/prove.go.
#include "common.cuh"
#include "msm.cuh"
#include "ntt.cuh"
#include <errno.h>
#include <functional>
#include <new>
#include <sys/random.h>
#include <type_traits>

using namespace zk;

namespace zk { int32_t decompress_to_device(zkpor_ctx* ctx, bool g2, const uint8_t* host_in, size_t n, void* d_out); }  // decompress.hip

struct zkpor_pk {
    zkpor_ctx* ctx = nullptr;
    // as uploaded (gnark's compacted arrays)
    void* g1_raw[ZKPOR_G1_NUM] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    size_t g1_raw_n[ZKPOR_G1_NUM] = {0, 0, 0, 0, 0, 0};
    void* g2_raw = nullptr;
    size_t g2_raw_n = 0;
    // wire-indexed / final arrays
    G1Affine *A = nullptr, *B1 = nullptr, *K = nullptr, *Z = nullptr, *CB = nullptr, *CBS = nullptr;
    G2Affine* B2 = nullptr;
    size_t n_wires = 0, n_public = 0, nZ = 0, nC = 0;
    int log2_domain = 0;
    G1Affine alpha, beta, delta;
    G2Affine beta2, delta2;
    bool ready = false;
    bool shard = false;  // holds only a contiguous range of every array (zkpor_pk_keep_range): sums only, no whole proof
    int tab_shift_w = 0, tab_shift_z = 0, tab_shift_c = 0;  // bits between consecutive tables of the w-, h- and commitment-indexed arrays
    // per-array digit streams (msm_digits.hip k_filter_write): bit w set = wire w has no point in B1 / B2 (pk.InfinityB) resp. in K (the
    // public and committed wires); built from the wire-indexed arrays when the key is finalised; nullptr = nothing worth filtering
    u32 *absentB = nullptr, *absentK = nullptr;
    double fracB = 0.0, fracK = 0.0;
    int tab_m = 1;       // fixed-base tables per point (context parameter "msm_tables" at load time): every array then holds
                         // n * tab_m points, entry i * tab_m + q = 2^(q * piece * c) P_i (msm.cuh MsmCfg)
};

zkpor_ctx* zk_pk_ctx(zkpor_pk* pk) { return pk->ctx; }  // for keyfile.hip

namespace {

// dst[i] = map[i] == 0xffffffff ? infinity : src[map[i]]
template <class P>
__global__ void k_expand(const P* __restrict__ src, const u32* __restrict__ map, P* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 m = map[i];
    P out;
    if (m == 0xffffffffu) memset(&out, 0, sizeof(P)); else out = src[m];
    dst[i] = out;
}

template <class P>
int32_t expand(zkpor_ctx* ctx, const void* raw, size_t raw_n, const std::vector<u32>& map, P** out) {
    size_t n = map.size();
    u32* dmap = nullptr;
    P* dst = nullptr;
    ZK_HIP(ctx, hipMalloc((void**)&dmap, n * 4 + 4));
    if (hipMalloc((void**)&dst, n * sizeof(P) + 16) != hipSuccess) { (void)hipFree(dmap); ctx->err = "pk: out of device memory"; return ZKPOR_E_OOM; }
    ZK_HIP(ctx, hipMemcpyAsync(dmap, map.data(), n * 4, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_expand<P>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const P*)raw, dmap, dst, n);
    ZK_KERNEL_CHECK(ctx);
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(dmap);
    (void)raw_n;
    *out = dst;
    return ZKPOR_OK;
}

// ---- synthetic key generation (TEST/BENCH utility) -------------------------------------------------------
__host__ __device__ inline u64 smix(u64 x) {
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}
static const int SYNTH_RUN = 32;
// scalar of point i of array `arr`: k(run) + j*q with 64-bit k, q — the trapdoor the parity tests recompute
__host__ __device__ inline u64 synth_k(u64 seed, int arr, u64 run) {
    u64 x = seed ^ ((u64)(arr + 1) * 0xa0761d6478bd642fULL) ^ (run * 0xe7037ed1a0b428dbULL);
    return smix(x) | 1ULL;
}
static const u64 SYNTH_Q = 0x9e3779b97f4a7c15ULL;

// infinity pattern of the synthetic wire-indexed arrays: 0 none, else i is infinity when hash(i) % mod == 0
__host__ __device__ inline bool synth_is_inf(u64 i, u32 mod) {
    if (!mod) return false;
    u64 x = i * 0xd6e8feb86659fd93ULL;
    x ^= x >> 32;
    return (x % mod) == 0;
}

template <class F>
__global__ __launch_bounds__(64) void k_synth_points(Affine<F> gen, Affine<F> qpt, u64 seed, int arr, size_t n,
                                                     u32 inf_mod, size_t inf_below, Affine<F>* out) {
    size_t run = (size_t)blockIdx.x * 64 + threadIdx.x;
    size_t base = run * SYNTH_RUN;
    if (base >= n) return;
    u64 k = synth_k(seed, arr, run);
    XYZZ<F> p = XYZZ<F>::inf();
    XYZZ<F> g = xyzz_from_affine<F>(gen);
    for (int b = 63; b >= 0; --b) {
        p = xyzz_dbl<F>(p);
        if ((k >> b) & 1) p = xyzz_add_nl<F>(p, g);
    }
    // run of SYNTH_RUN points p, p+Q, p+2Q, ... ; batch inversion of ZZ*ZZZ over the run
    XYZZ<F> pts[SYNTH_RUN];
    F pref[SYNTH_RUN];
    F accp = F::one();
    XYZZ<F> q = xyzz_from_affine<F>(qpt);
    for (int j = 0; j < SYNTH_RUN; ++j) {
        pts[j] = p;
        pref[j] = accp;
        accp = F::mul(accp, F::mul(p.zz, p.zzz));
        p = xyzz_add_nl<F>(p, q);
    }
    F inv = F::inv(accp);
    for (int j = SYNTH_RUN - 1; j >= 0; --j) {
        F zi = F::mul(inv, pref[j]);                       // 1/(zz*zzz) of point j
        inv = F::mul(inv, F::mul(pts[j].zz, pts[j].zzz));
        size_t i = base + j;
        if (i < n) {
            Affine<F> a;
            if (i < inf_below || synth_is_inf(i, inf_mod)) { a.x = F::zero(); a.y = F::zero(); }
            else { a.x = F::mul(pts[j].x, F::mul(zi, pts[j].zzz)); a.y = F::mul(pts[j].y, F::mul(zi, pts[j].zz)); }
            out[i] = a;
        }
    }
}

// zero (= infinity) every point whose mask byte is set
template <class P>
__global__ __launch_bounds__(256) void k_mask_points(P* __restrict__ pts, const uint8_t* __restrict__ mask, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n || !mask[i]) return;
    P z;
    memset(&z, 0, sizeof z);
    pts[i] = z;
}

G1Affine g1_generator() {
    G1Affine g;
    g.x = Fp::from_u32(1);
    g.y = Fp::from_u32(2);
    return g;
}
G2Affine g2_generator() {
    static const u32 x0[8] = {0xd992f6edu, 0x46debd5cu, 0xf75edaddu, 0x674322d4u, 0x5e5c4479u, 0x426a0066u, 0x121f1e76u, 0x1800deefu};
    static const u32 x1[8] = {0xaef312c2u, 0x97e485b7u, 0x35a9e712u, 0xf1aa4933u, 0x31fb5d25u, 0x7260bfb7u, 0x920d483au, 0x198e9393u};
    static const u32 y0[8] = {0x66fa7daau, 0x4ce6cc01u, 0x0c43d37bu, 0xe3d1e769u, 0x8dcb408fu, 0x4aab7180u, 0xdb8c6debu, 0x12c85ea5u};
    static const u32 y1[8] = {0xd122975bu, 0x55acdadcu, 0x70b38ef3u, 0xbc4b3133u, 0x690c3395u, 0xec9e99adu, 0x585ff075u, 0x090689d0u};
    auto mk = [](const u32* l) { Fp r; for (int i = 0; i < 8; ++i) r.v[i] = l[i]; return Fp::to_mont(r); };
    G2Affine g;
    g.x = {mk(x0), mk(x1)};
    g.y = {mk(y0), mk(y1)};
    return g;
}

template <class F>
int32_t synth_array(zkpor_ctx* ctx, const Affine<F>& gen, u64 seed, int arr, size_t n, u32 inf_mod, size_t inf_below,
                    Affine<F>** out) {
    Affine<F>* d = nullptr;
    ZK_HIP(ctx, hipMalloc((void**)&d, (n ? n : 1) * sizeof(Affine<F>) + 16));
    if (n) {
        // Q = SYNTH_Q * gen (host)
        XYZZ<F> q = xyzz_mul_u64<F>(xyzz_from_affine<F>(gen), SYNTH_Q);
        Affine<F> qa = xyzz_to_affine<F>(q);
        size_t runs = (n + SYNTH_RUN - 1) / SYNTH_RUN;
        hipLaunchKernelGGL(k_synth_points<F>, dim3((unsigned)((runs + 63) / 64)), dim3(64), 0, ctx->stream, gen, qa, seed,
                           arr, n, inf_mod, inf_below, d);
        ZK_KERNEL_CHECK(ctx);
    }
    *out = d;
    return ZKPOR_OK;
}

// ---- fixed-base tables (msm.cuh MsmCfg): out[i * m + q] = 2^(q * shift_bits) src[i], affine -----------------------------------
template <class F>
__global__ __launch_bounds__(64) void k_build_tables(const Affine<F>* __restrict__ src, size_t n, int m, int shift_bits, Affine<F>* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    Affine<F> p = src[i];
    out[i * (size_t)m] = p;
    XYZZ<F> acc = xyzz_from_affine<F>(p);
    for (int q = 1; q < m; ++q) {
        for (int b = 0; b < shift_bits; ++b) acc = xyzz_dbl<F>(acc);
        Affine<F> a;
        if (acc.is_inf()) { a.x = F::zero(); a.y = F::zero(); }
        else {  // one inversion: ZZ^3 = ZZZ^2  =>  1/ZZ = (ZZ / ZZZ)^2
            F iz = F::inv(acc.zzz);
            F t = F::mul(acc.zz, iz);
            a.x = F::mul(acc.x, F::sqr(t));
            a.y = F::mul(acc.y, iz);
        }
        out[i * (size_t)m + q] = a;
    }
}
// replace a wire-indexed array by its table form (m > 1); the plain array is freed
template <class F>
int32_t make_tables(zkpor_ctx* ctx, Affine<F>** arr, size_t n, int m, int shift_bits) {
    if (m <= 1 || !*arr || n == 0) return ZKPOR_OK;
    Affine<F>* out = nullptr;
    if (hipMalloc((void**)&out, n * (size_t)m * sizeof(Affine<F>) + 16) != hipSuccess) { (void)hipGetLastError(); ctx->err = "pk: out of device memory for the fixed-base tables"; return ZKPOR_E_OOM; }
    hipLaunchKernelGGL(k_build_tables<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (const Affine<F>*)*arr, n, m, shift_bits, out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(out); ctx->err = std::string("pk: table construction: ") + hipGetErrorString(e); return ZKPOR_E_HIP; }
    (void)hipFree(*arr);
    *arr = out;
    return ZKPOR_OK;
}

// bit i of bits = point i of the (plain, wire-indexed) array is infinity; *count = how many
template <class P>
__global__ void k_absent_bits(const P* __restrict__ pts, size_t n, u32* __restrict__ bits, u32* __restrict__ count) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    const bool inf = i < n && pts[i].is_inf();
    const u64 b = __ballot(inf);
    const u32 lane = threadIdx.x & 63u;
    if (lane == 0 && i < n) { bits[2 * (i >> 6)] = (u32)b; bits[2 * (i >> 6) + 1] = (u32)(b >> 32); if (b) atomicAdd(count, (u32)__popcll(b)); }   // a wave wholly past n writes nothing
}
void pk_free_masks(zkpor_pk* pk) {
    if (pk->absentB) (void)hipFree(pk->absentB);
    if (pk->absentK) (void)hipFree(pk->absentK);
    pk->absentB = pk->absentK = nullptr;
    pk->fracB = pk->fracK = 0.0;
}
// the masks of the per-array digit streams, from the final wire-indexed arrays themselves (so every load path — set_consts, the
// .pk container, the synthetic key — gets them the same way).  A group is filtered only when >= 3 % of its wires are absent.
int32_t pk_build_masks(zkpor_pk* pk) {
    zkpor_ctx* ctx = pk->ctx;
    pk_free_masks(pk);
    if (pk->shard || pk->n_wires == 0 || !pk->B1 || !pk->K) return ZKPOR_OK;
    const size_t n = pk->n_wires, words = 2 * ((n + 63) / 64) + 2;
    u32 *bB = nullptr, *bK = nullptr, *cnt = nullptr;
    if (hipMalloc((void**)&bB, words * 4) != hipSuccess || hipMalloc((void**)&bK, words * 4) != hipSuccess || hipMalloc((void**)&cnt, 8) != hipSuccess) {
        (void)hipGetLastError();
        if (bB) (void)hipFree(bB); if (bK) (void)hipFree(bK); if (cnt) (void)hipFree(cnt);
        return ZKPOR_OK;   // no memory for the masks: prove with the shared stream
    }
    ZK_HIP(ctx, hipMemsetAsync(cnt, 0, 8, ctx->stream));
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_absent_bits<G1Affine>, dim3(blocks), dim3(256), 0, ctx->stream, (const G1Affine*)pk->B1, n, bB, cnt);
    hipLaunchKernelGGL(k_absent_bits<G1Affine>, dim3(blocks), dim3(256), 0, ctx->stream, (const G1Affine*)pk->K, n, bK, cnt + 1);
    u32 h[2] = {0, 0};
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(h, cnt, 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(cnt);
    if (e != hipSuccess) { (void)hipFree(bB); (void)hipFree(bK); ctx->err = std::string("pk: mask construction: ") + hipGetErrorString(e); return ZKPOR_E_HIP; }
    pk->fracB = (double)h[0] / (double)n; pk->fracK = (double)h[1] / (double)n;
    if (pk->fracB >= 0.03) pk->absentB = bB; else (void)hipFree(bB);
    if (pk->fracK >= 0.03) pk->absentK = bK; else (void)hipFree(bK);
    return ZKPOR_OK;
}

// after the wire-indexed arrays are final: turn them into tables if the context asks for it.  The spacing of an array's tables is
// the spacing its multi-exponentiation will use (piece * c of msm_cfg for the array's length); prove_sums / commit check it.
int32_t pk_apply_tables(zkpor_pk* pk) {
    zkpor_ctx* ctx = pk->ctx;
    pk->tab_m = 1;
    ZK_TRY(pk_build_masks(pk));
    const int m = ctx->msm_tables;
    if (m <= 1) return ZKPOR_OK;
    MsmCfg cw = msm_cfg(ctx, pk->n_wires, m), cz = msm_cfg(ctx, pk->nZ ? pk->nZ : 1, m), cc = msm_cfg(ctx, pk->nC ? pk->nC : 1, m);
    pk->tab_shift_w = cw.piece * cw.c; pk->tab_shift_z = cz.piece * cz.c; pk->tab_shift_c = cc.piece * cc.c;
    pk->ready = false;  // a failure half-way leaves arrays of different shapes
    ZK_TRY(make_tables<Fp>(ctx, &pk->A, pk->n_wires, m, pk->tab_shift_w));
    ZK_TRY(make_tables<Fp>(ctx, &pk->B1, pk->n_wires, m, pk->tab_shift_w));
    ZK_TRY(make_tables<Fp>(ctx, &pk->K, pk->n_wires, m, pk->tab_shift_w));
    ZK_TRY(make_tables<Fp2>(ctx, &pk->B2, pk->n_wires, m, pk->tab_shift_w));
    ZK_TRY(make_tables<Fp>(ctx, &pk->Z, pk->nZ, m, pk->tab_shift_z));
    ZK_TRY(make_tables<Fp>(ctx, &pk->CB, pk->nC, m, pk->tab_shift_c));
    ZK_TRY(make_tables<Fp>(ctx, &pk->CBS, pk->nC, m, pk->tab_shift_c));
    pk->tab_m = m;
    pk->ready = true;
    return ZKPOR_OK;
}
int32_t check_tables(zkpor_ctx* ctx, zkpor_pk* pk, const MsmCfg& cfg, int shift) {
    if (pk->tab_m > 1 && cfg.piece * cfg.c != shift) { ctx->err = "msm: the window in force differs from the one the key's fixed-base tables were built for (msm_window changed after the key was loaded)"; return ZKPOR_E_STATE; }
    return ZKPOR_OK;
}

void pk_free_arrays(zkpor_pk* pk) {
    for (int i = 0; i < ZKPOR_G1_NUM; ++i) if (pk->g1_raw[i]) { (void)hipFree(pk->g1_raw[i]); pk->g1_raw[i] = nullptr; }
    if (pk->g2_raw) { (void)hipFree(pk->g2_raw); pk->g2_raw = nullptr; }
    void* arrs[] = {pk->A, pk->B1, pk->K, pk->Z, pk->CB, pk->CBS, pk->B2};
    for (void* p : arrs) if (p) (void)hipFree(p);
    pk->A = pk->B1 = pk->K = pk->Z = pk->CB = pk->CBS = nullptr;
    pk->B2 = nullptr;
    pk_free_masks(pk);
    pk->ready = false;
}

inline G1XYZZ g1x(const G1Affine& a) { return xyzz_from_affine<Fp>(a); }

}  // namespace

extern "C" {

int32_t zkpor_pk_create(zkpor_ctx* ctx, zkpor_pk** out) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !out) return ZKPOR_E_ARG;
    zkpor_pk* pk = new (std::nothrow) zkpor_pk();
    if (!pk) return ZKPOR_E_OOM;
    pk->ctx = ctx;
    *out = pk;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
void zkpor_pk_destroy(zkpor_pk* pk) try {
    ZK_ENTER(pk ? pk->ctx->device : -1);
    if (!pk) return;
    (void)hipStreamSynchronize(pk->ctx->stream);
    pk_free_arrays(pk);
    delete pk;
} catch (...) { zk::abi_exception("exception in zkpor_pk_destroy"); }

int32_t zkpor_pk_set_g1(zkpor_pk* pk, int which, const void* pts, size_t n) try {
    ZK_ENTER(pk ? pk->ctx->device : -1);
    if (!pk || which < 0 || which >= ZKPOR_G1_NUM || (n && !pts)) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = pk->ctx;
    if (pk->g1_raw[which]) { ZK_HIP(ctx, hipFree(pk->g1_raw[which])); pk->g1_raw[which] = nullptr; }
    ZK_HIP(ctx, hipMalloc(&pk->g1_raw[which], (n ? n : 1) * 64));
    ZK_TRY(zk::h2d_sync(ctx, pk->g1_raw[which], pts, n * 64));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pk->g1_raw_n[which] = n;
    pk->ready = false;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((pk ? pk->ctx : nullptr))
int32_t zkpor_pk_set_g2(zkpor_pk* pk, int which, const void* pts, size_t n) try {
    ZK_ENTER(pk ? pk->ctx->device : -1);
    if (!pk || which != ZKPOR_G2_B || (n && !pts)) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = pk->ctx;
    if (pk->g2_raw) { ZK_HIP(ctx, hipFree(pk->g2_raw)); pk->g2_raw = nullptr; }
    ZK_HIP(ctx, hipMalloc(&pk->g2_raw, (n ? n : 1) * 128));
    ZK_TRY(zk::h2d_sync(ctx, pk->g2_raw, pts, n * 128));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    pk->g2_raw_n = n;
    pk->ready = false;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((pk ? pk->ctx : nullptr))

// compressed input (what pk.WriteTo put on disk, src/keygen/main.go:46): decompressed on the device, decompress.hip
int32_t zkpor_pk_set_g1_compressed(zkpor_pk* pk, int which, const uint8_t* compressed32, size_t n) try {
    ZK_ENTER(pk ? pk->ctx->device : -1);
    if (!pk || which < 0 || which >= ZKPOR_G1_NUM || (n && !compressed32)) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = pk->ctx;
    if (pk->g1_raw[which]) { ZK_HIP(ctx, hipFree(pk->g1_raw[which])); pk->g1_raw[which] = nullptr; }
    pk->g1_raw_n[which] = 0;
    pk->ready = false;
    ZK_HIP(ctx, hipMalloc(&pk->g1_raw[which], (n ? n : 1) * 64));
    ZK_TRY(zk::decompress_to_device(ctx, false, compressed32, n, pk->g1_raw[which]));
    pk->g1_raw_n[which] = n;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((pk ? pk->ctx : nullptr))
int32_t zkpor_pk_set_g2_compressed(zkpor_pk* pk, int which, const uint8_t* compressed64, size_t n) try {
    ZK_ENTER(pk ? pk->ctx->device : -1);
    if (!pk || which != ZKPOR_G2_B || (n && !compressed64)) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = pk->ctx;
    if (pk->g2_raw) { ZK_HIP(ctx, hipFree(pk->g2_raw)); pk->g2_raw = nullptr; }
    pk->g2_raw_n = 0;
    pk->ready = false;
    ZK_HIP(ctx, hipMalloc(&pk->g2_raw, (n ? n : 1) * 128));
    ZK_TRY(zk::decompress_to_device(ctx, true, compressed64, n, pk->g2_raw));
    pk->g2_raw_n = n;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((pk ? pk->ctx : nullptr))

}  // extern "C"

// Re-lay the uploaded (compacted) arrays out wire-indexed and mark the key ready.  `removed[i]` != 0: wire i has no K point.
// For a shard (one rank's contiguous range of a split key, keyfile.hip) the arrays and masks cover that range only and Z holds
// z_n points already in the prover's (bit-reversed) order.
int32_t zk_pk_finalize(zkpor_pk* pk, const void* alpha, const void* beta, const void* delta, const void* beta2, const void* delta2,
                       int log2_domain, const uint8_t* inf_a, const uint8_t* inf_b, size_t n_wires, const uint8_t* removed,
                       size_t n_public, int z_order, bool shard, size_t z_n) {
    zkpor_ctx* ctx = pk->ctx;
    memcpy(&pk->alpha, alpha, 64); memcpy(&pk->beta, beta, 64); memcpy(&pk->delta, delta, 64);
    memcpy(&pk->beta2, beta2, 128); memcpy(&pk->delta2, delta2, 128);
    pk->log2_domain = log2_domain;
    pk->n_wires = n_wires; pk->n_public = n_public;
    // index maps wire -> position in gnark's compacted arrays
    std::vector<u32> mapA(n_wires), mapB(n_wires), mapK(n_wires);
    u32 ra = 0, rb = 0, rk = 0;
    for (size_t i = 0; i < n_wires; ++i) {
        mapA[i] = (inf_a && inf_a[i]) ? 0xffffffffu : ra++;
        mapB[i] = (inf_b && inf_b[i]) ? 0xffffffffu : rb++;
        mapK[i] = removed[i] ? 0xffffffffu : rk++;
    }
    if (ra != pk->g1_raw_n[ZKPOR_G1_A] || rb != pk->g1_raw_n[ZKPOR_G1_B] || rb != pk->g2_raw_n || rk != pk->g1_raw_n[ZKPOR_G1_K]) {
        ctx->err = "pk: array lengths do not match the infinity / committed masks";
        return ZKPOR_E_STATE;
    }
    size_t D = (size_t)1 << log2_domain;
    if (!shard) z_n = D - 1;
    if (pk->g1_raw_n[ZKPOR_G1_Z] != z_n) { ctx->err = shard ? "pk: Z does not hold the shard's range" : "pk: Z must hold 2^log2_domain - 1 points"; return ZKPOR_E_STATE; }
    if (shard && z_order != ZKPOR_Z_ORDER_BITREV) { ctx->err = "pk: a shard's Z must already be in the prover's order"; return ZKPOR_E_ARG; }
    if (pk->g1_raw_n[ZKPOR_G1_COMMIT_BASIS] != pk->g1_raw_n[ZKPOR_G1_COMMIT_BASIS_SIGMA]) { ctx->err = "pk: commitment bases differ in length"; return ZKPOR_E_STATE; }
    void* olds[] = {pk->A, pk->B1, pk->K, pk->Z, pk->CB, pk->CBS, pk->B2};
    for (void* p : olds) if (p) (void)hipFree(p);
    pk->A = pk->B1 = pk->K = pk->Z = pk->CB = pk->CBS = nullptr; pk->B2 = nullptr;
    ZK_TRY(expand<G1Affine>(ctx, pk->g1_raw[ZKPOR_G1_A], ra, mapA, &pk->A));
    ZK_TRY(expand<G1Affine>(ctx, pk->g1_raw[ZKPOR_G1_B], rb, mapB, &pk->B1));
    ZK_TRY(expand<G2Affine>(ctx, pk->g2_raw, rb, mapB, &pk->B2));
    ZK_TRY(expand<G1Affine>(ctx, pk->g1_raw[ZKPOR_G1_K], rk, mapK, &pk->K));
    // Z: the prover produces h bit-reversed; bring a natural-order Z into that order once
    pk->nZ = z_n;
    {
        std::vector<u32> mapZ(z_n);
        for (size_t j = 0; j < z_n; ++j) {
            if (z_order == ZKPOR_Z_ORDER_NATURAL) {
                u32 r = 0;
                for (int b = 0; b < log2_domain; ++b) r |= (u32)((j >> b) & 1) << (log2_domain - 1 - b);
                mapZ[j] = r;  // r == D-1 only for j == D-1, which is outside the array
            } else mapZ[j] = (u32)j;
        }
        ZK_TRY(expand<G1Affine>(ctx, pk->g1_raw[ZKPOR_G1_Z], z_n, mapZ, &pk->Z));
    }
    pk->nC = pk->g1_raw_n[ZKPOR_G1_COMMIT_BASIS];
    // commitment bases are used as uploaded: hand the raw buffers over
    pk->CB = (G1Affine*)pk->g1_raw[ZKPOR_G1_COMMIT_BASIS]; pk->g1_raw[ZKPOR_G1_COMMIT_BASIS] = nullptr;
    pk->CBS = (G1Affine*)pk->g1_raw[ZKPOR_G1_COMMIT_BASIS_SIGMA]; pk->g1_raw[ZKPOR_G1_COMMIT_BASIS_SIGMA] = nullptr;
    // the compacted uploads are no longer needed
    for (int i = 0; i < ZKPOR_G1_NUM; ++i) if (pk->g1_raw[i]) { (void)hipFree(pk->g1_raw[i]); pk->g1_raw[i] = nullptr; pk->g1_raw_n[i] = 0; }
    if (pk->g2_raw) { (void)hipFree(pk->g2_raw); pk->g2_raw = nullptr; pk->g2_raw_n = 0; }
    pk->ready = true;
    pk->shard = shard;
    pk->tab_m = 1;
    if (!shard) ZK_TRY(pk_apply_tables(pk));
    return ZKPOR_OK;
}

extern "C" {

int32_t zkpor_pk_set_consts(zkpor_pk* pk, const void* alpha, const void* beta, const void* delta, const void* beta2,
                            const void* delta2, int log2_domain, const uint8_t* inf_a, const uint8_t* inf_b,
                            size_t n_wires, size_t n_public, const uint32_t* committed_idx, size_t n_committed,
                            int z_order) try {
    ZK_ENTER(pk ? pk->ctx->device : -1);
    if (!pk || !alpha || !beta || !delta || !beta2 || !delta2 || log2_domain < 1 || log2_domain > 28) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = pk->ctx;
    if (n_wires == 0 || n_wires >= 0xffffffffull || n_public > n_wires) { ctx->err = "pk: bad wire counts"; return ZKPOR_E_ARG; }
    std::vector<uint8_t> removed(n_wires, 0);
    for (size_t i = 0; i < n_public; ++i) removed[i] = 1;
    for (size_t j = 0; j < n_committed; ++j) {
        if (committed_idx[j] >= n_wires) { ctx->err = "pk: committed index out of range"; return ZKPOR_E_ARG; }
        removed[committed_idx[j]] = 1;
    }
    return zk_pk_finalize(pk, alpha, beta, delta, beta2, delta2, log2_domain, inf_a, inf_b, n_wires, removed.data(), n_public, z_order,
                          false, 0);
} ZK_ABI_CATCH_IN((pk ? pk->ctx : nullptr))

int32_t zkpor_pk_synth(zkpor_pk* pk, int log2_domain, size_t n_wires, size_t n_public, size_t n_committed, uint64_t seed) try {
    ZK_ENTER(pk ? pk->ctx->device : -1);
    if (!pk || log2_domain < 1 || log2_domain > 28 || n_wires == 0 || n_public > n_wires) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = pk->ctx;
    pk_free_arrays(pk);
    size_t D = (size_t)1 << log2_domain;
    G1Affine g1 = g1_generator();
    G2Affine g2 = g2_generator();
    pk->log2_domain = log2_domain; pk->n_wires = n_wires; pk->n_public = n_public; pk->nZ = D - 1; pk->nC = n_committed;
    ZK_TRY(synth_array<Fp>(ctx, g1, seed, ZKPOR_G1_A, n_wires, 64, 0, &pk->A));          // ~1.6% infinity
    ZK_TRY(synth_array<Fp>(ctx, g1, seed, ZKPOR_G1_B, n_wires, 10, 0, &pk->B1));         // ~10% infinity
    ZK_TRY(synth_array<Fp2>(ctx, g2, seed, ZKPOR_G1_B, n_wires, 10, 0, &pk->B2));        // same scalars as B1
    ZK_TRY(synth_array<Fp>(ctx, g1, seed, ZKPOR_G1_K, n_wires, 4, n_public, &pk->K));    // public + ~25% "committed"
    ZK_TRY(synth_array<Fp>(ctx, g1, seed, ZKPOR_G1_Z, D - 1, 0, 0, &pk->Z));
    ZK_TRY(synth_array<Fp>(ctx, g1, seed, ZKPOR_G1_COMMIT_BASIS, n_committed, 0, 0, &pk->CB));
    ZK_TRY(synth_array<Fp>(ctx, g1, seed, ZKPOR_G1_COMMIT_BASIS_SIGMA, n_committed, 0, 0, &pk->CBS));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    // alpha, beta, delta: fixed small multiples of the generators (host)
    pk->alpha = xyzz_to_affine<Fp>(xyzz_mul_u64<Fp>(g1x(g1), synth_k(seed, 100, 0)));
    pk->beta = xyzz_to_affine<Fp>(xyzz_mul_u64<Fp>(g1x(g1), synth_k(seed, 101, 0)));
    pk->delta = xyzz_to_affine<Fp>(xyzz_mul_u64<Fp>(g1x(g1), synth_k(seed, 102, 0)));
    pk->beta2 = xyzz_to_affine<Fp2>(xyzz_mul_u64<Fp2>(xyzz_from_affine<Fp2>(g2), synth_k(seed, 101, 0)));
    pk->delta2 = xyzz_to_affine<Fp2>(xyzz_mul_u64<Fp2>(xyzz_from_affine<Fp2>(g2), synth_k(seed, 102, 0)));
    pk->ready = true;
    pk->shard = false;
    return pk_apply_tables(pk);
} ZK_ABI_CATCH_IN((pk ? pk->ctx : nullptr))

/* the synthetic key with a CIRCUIT's sparsity instead of the seeded one: A / B1 / B2 are infinity exactly where inf_a / inf_b say
 * (gnark: a wire that appears in no L / R row), K exactly at the public wires and at removed_idx (the committed wires + the commitment
 * wire); the Pedersen bases hold n_basis points.  Same point generator, so oracle/trapdoor.py predicts every sum once it is given the masks. */
int32_t zkpor_pk_synth_masked(zkpor_pk* pk, int log2_domain, size_t n_wires, size_t n_public, const uint8_t* inf_a, const uint8_t* inf_b,
                              const uint32_t* removed_idx, size_t n_removed, size_t n_basis, uint64_t seed) try {
    ZK_ENTER(pk ? pk->ctx->device : -1);
    if (!pk || log2_domain < 1 || log2_domain > 28 || n_wires == 0 || n_public > n_wires || (n_removed && !removed_idx)) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = pk->ctx;
    std::vector<uint8_t> mk(n_wires, 0);
    for (size_t j = 0; j < n_removed; ++j) { if (removed_idx[j] >= n_wires) { ctx->err = "pk: removed wire out of range"; return ZKPOR_E_ARG; } mk[removed_idx[j]] = 1; }
    pk_free_arrays(pk);
    const size_t D = (size_t)1 << log2_domain;
    G1Affine g1 = g1_generator();
    G2Affine g2 = g2_generator();
    pk->log2_domain = log2_domain; pk->n_wires = n_wires; pk->n_public = n_public; pk->nZ = D - 1; pk->nC = n_basis;
    ZK_TRY(synth_array<Fp>(ctx, g1, seed, ZKPOR_G1_A, n_wires, 0, 0, &pk->A));
    ZK_TRY(synth_array<Fp>(ctx, g1, seed, ZKPOR_G1_B, n_wires, 0, 0, &pk->B1));
    ZK_TRY(synth_array<Fp2>(ctx, g2, seed, ZKPOR_G1_B, n_wires, 0, 0, &pk->B2));
    ZK_TRY(synth_array<Fp>(ctx, g1, seed, ZKPOR_G1_K, n_wires, 0, n_public, &pk->K));
    ZK_TRY(synth_array<Fp>(ctx, g1, seed, ZKPOR_G1_Z, D - 1, 0, 0, &pk->Z));
    ZK_TRY(synth_array<Fp>(ctx, g1, seed, ZKPOR_G1_COMMIT_BASIS, n_basis, 0, 0, &pk->CB));
    ZK_TRY(synth_array<Fp>(ctx, g1, seed, ZKPOR_G1_COMMIT_BASIS_SIGMA, n_basis, 0, 0, &pk->CBS));
    uint8_t* d_mask = nullptr;
    ZK_HIP(ctx, hipMalloc((void**)&d_mask, n_wires));
    const unsigned grid = (unsigned)((n_wires + 255) / 256);
    int32_t rc = ZKPOR_OK;
    auto apply = [&](const uint8_t* h, int which) {
        if (!h || rc != ZKPOR_OK) return;
        if (zk::h2d_sync(ctx, d_mask, h, n_wires) != ZKPOR_OK) { ctx->err = "pk: H2D failed"; rc = ZKPOR_E_HIP; return; }
        if (which == 0) hipLaunchKernelGGL(k_mask_points<G1Affine>, dim3(grid), dim3(256), 0, ctx->stream, pk->A, d_mask, n_wires);
        if (which == 1) { hipLaunchKernelGGL(k_mask_points<G1Affine>, dim3(grid), dim3(256), 0, ctx->stream, pk->B1, d_mask, n_wires);
                          hipLaunchKernelGGL(k_mask_points<G2Affine>, dim3(grid), dim3(256), 0, ctx->stream, pk->B2, d_mask, n_wires); }
        if (which == 2) hipLaunchKernelGGL(k_mask_points<G1Affine>, dim3(grid), dim3(256), 0, ctx->stream, pk->K, d_mask, n_wires);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "pk: mask kernel failed"; rc = ZKPOR_E_HIP; }   // the host mask is reused
    };
    apply(inf_a, 0); apply(inf_b, 1); apply(mk.data(), 2);
    (void)hipFree(d_mask);
    if (rc != ZKPOR_OK) return rc;
    pk->alpha = xyzz_to_affine<Fp>(xyzz_mul_u64<Fp>(g1x(g1), synth_k(seed, 100, 0)));
    pk->beta = xyzz_to_affine<Fp>(xyzz_mul_u64<Fp>(g1x(g1), synth_k(seed, 101, 0)));
    pk->delta = xyzz_to_affine<Fp>(xyzz_mul_u64<Fp>(g1x(g1), synth_k(seed, 102, 0)));
    pk->beta2 = xyzz_to_affine<Fp2>(xyzz_mul_u64<Fp2>(xyzz_from_affine<Fp2>(g2), synth_k(seed, 101, 0)));
    pk->delta2 = xyzz_to_affine<Fp2>(xyzz_mul_u64<Fp2>(xyzz_from_affine<Fp2>(g2), synth_k(seed, 102, 0)));
    pk->ready = true;
    pk->shard = false;
    return pk_apply_tables(pk);
} ZK_ABI_CATCH_IN((pk ? pk->ctx : nullptr))

int32_t zkpor_pk_dims(zkpor_pk* pk, uint64_t dims[6]) try {
    if (!pk || !dims) return ZKPOR_E_ARG;
    if (!pk->ready) return ZKPOR_E_STATE;
    dims[0] = pk->n_wires; dims[1] = pk->n_public; dims[2] = pk->nC; dims[3] = pk->nZ;
    dims[4] = (uint64_t)pk->log2_domain; dims[5] = (uint64_t)pk->tab_m;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((pk ? pk->ctx : nullptr))

int32_t zkpor_pk_g1_dev(zkpor_pk* pk, int which, void** dev_ptr, size_t* n) try {
    ZK_ENTER(pk ? pk->ctx->device : -1);
    if (!pk || !dev_ptr || !n) return ZKPOR_E_ARG;
    if (!pk->ready) return ZKPOR_E_STATE;
    if (pk->tab_m > 1) { pk->ctx->err = "pk: the arrays are interleaved fixed-base tables (msm_tables > 1), not plain point arrays"; return ZKPOR_E_STATE; }
    switch (which) {
        case ZKPOR_G1_A: *dev_ptr = pk->A; *n = pk->n_wires; break;
        case ZKPOR_G1_B: *dev_ptr = pk->B1; *n = pk->n_wires; break;
        case ZKPOR_G1_K: *dev_ptr = pk->K; *n = pk->n_wires; break;
        case ZKPOR_G1_Z: *dev_ptr = pk->Z; *n = pk->nZ; break;
        case ZKPOR_G1_COMMIT_BASIS: *dev_ptr = pk->CB; *n = pk->nC; break;
        case ZKPOR_G1_COMMIT_BASIS_SIGMA: *dev_ptr = pk->CBS; *n = pk->nC; break;
        default: return ZKPOR_E_ARG;
    }
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((pk ? pk->ctx : nullptr))
int32_t zkpor_pk_g2_dev(zkpor_pk* pk, int which, void** dev_ptr, size_t* n) try {
    ZK_ENTER(pk ? pk->ctx->device : -1);
    if (!pk || !dev_ptr || !n || which != ZKPOR_G2_B) return ZKPOR_E_ARG;
    if (!pk->ready) return ZKPOR_E_STATE;
    if (pk->tab_m > 1) { pk->ctx->err = "pk: the arrays are interleaved fixed-base tables (msm_tables > 1), not plain point arrays"; return ZKPOR_E_STATE; }
    *dev_ptr = pk->B2; *n = pk->n_wires;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((pk ? pk->ctx : nullptr))

// ------------------------------------------------------------------------------------------------ prove tail
}  // extern "C"
namespace {

struct ProveSums { G1XYZZ A, B1, K, Z; G2XYZZ B2; };  // the five multi-exponentiations of groth16.Prove

// the caller's vectors when they are still in HOST memory (zkpor_prove_tail): prove_sums then carries them across PCIe itself,
// in the order the GPU needs them, under its own kernels
struct HostInputs { const void *w, *a, *b, *c; size_t n_constraints; };
// device-resident a, b, c the caller wants back untouched (zkpor_prove_tail_dev_keep): computeH's first pass reads them
struct KeptInputs { const void *a, *b, *c; };

// Queue everything in groth16.Prove between the solver and the blinding: h = computeH(a, b, c) when d_b is given (else d_a
// already holds the h scalars matching pk->Z), then A.w, B1.w, B2.w, K.w over one sorted digit stream of w and Z.h.
// Works on a whole key and on a shard (pk->n_wires / pk->nZ are then the shard's lengths and d_w / d_a its scalar ranges).
//
// Two orders of the same work.  Inputs resident in HBM: computeH first on the main stream with decompose + sort of w hidden
// under it on the auxiliary stream, then A, B1, K, B2 (sort of h hidden under them), then Z.  Inputs in host memory (`host`):
// w crosses PCIe first, its digit stream is built, A, B1 and K start — and a, b, c (3/4 of the bytes) cross on the copy stream
// while those three accumulations run; then computeH, B2 (sort of h hidden under it) and Z.
int32_t prove_sums(zkpor_ctx* ctx, zkpor_pk* pk, const void* d_w, void* d_a, void* d_b, void* d_c, ProveSums* out, bool do_w = true,
                   bool do_h = true, const std::function<void()>* while_gpu_runs = nullptr, const HostInputs* host = nullptr,
                   GpuTurn* turn = nullptr, const KeptInputs* kept = nullptr) {
    const int n = pk->log2_domain;
    const size_t nZ = do_h ? pk->nZ : 0;
    // Two HIP streams: the ALU-bound work (NTTs, bucket accumulations) on the context's stream, the HBM-bound digit
    // streams (decompose + radix sort) on an auxiliary one.  No host synchronisation on the main stream until all five sums are queued.
    if (!ctx->aux_stream) {
        // the digit streams feed the accumulations that wait for them: "aux_priority" 1 puts the auxiliary stream on the highest
        // stream priority so that its HBM-bound kernels are not starved by the VALU-bound ones of the main stream
        int lo = 0, hi = 0;
        if (ctx->aux_priority && hipDeviceGetStreamPriorityRange(&lo, &hi) == hipSuccess && hi != lo)
            ZK_HIP(ctx, hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, hi));
        else
            ZK_HIP(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    }
    const hipStream_t caller_s = ctx->stream;
    hipStream_t main_s = ctx->stream, aux_s = ctx->aux_stream;
    // the mask is for a tail that runs BESIDE another worker's device solve (the *_dev split: zkpor_solver_start_dev ... zkpor_prove_tail_dev).  A
    // host-pointer call holds the device turn from before its own solve to its last kernel — no solve ever overlaps its tail — so masking it would only
    // take compute units away from VALU-bound kernels (up to 12 % per tail on the dispatcher's host-solver path)
    const bool masked = (ctx->tail_reserve_cus > 0 || ctx->tail_streams) && !host && !turn;
    if (masked) {
        // "tail_reserve_cus": everything the tail queues goes to two streams whose CU mask leaves some compute units free.  A mask bit i is
        // compute unit i / 8 of XCD i % 8 on this part (the driver deals the bits round-robin over the XCDs), so clearing the first R bits
        // frees R / 8 units on each of the eight XCDs.
        // Only the streams this setting USES are created (an idle hardware queue is not free: profiles/r06_tail_mode_sweep.json): the main stream of this
        // reserve value; for the digit streams either the masked one of the pair ("tail_aux_masked" 1) or the unmasked one (0, the default; with a
        // reserve of 0 — "tail_streams" — the two are the same thing and the unmasked one serves).  Created once, kept until the context goes.
        const bool want_masked_aux = ctx->tail_aux_masked && ctx->tail_reserve_cus > 0;
        if (!ctx->tail_stream) {
            if (ctx->tail_sets.size() >= zkpor_ctx::TAIL_SETS_MAX) { ctx->err = "prove: no masked stream left for this tail_reserve_cus"; return ZKPOR_E_STATE; }
            hipStream_t st = nullptr;
            ZK_TRY(stream_create_own_queue(ctx, &st, ctx->tail_reserve_cus));
            ctx->tail_sets.push_back({ctx->tail_reserve_cus, st, nullptr, nullptr});
            ctx->tail_stream = st; ctx->tail_aux = nullptr; ctx->tail_chain = nullptr;
        }
        if (ctx->msm_chain && !ctx->tail_chain) {
            hipStream_t st = nullptr;
            ZK_TRY(stream_create_own_queue(ctx, &st, ctx->tail_reserve_cus));
            for (auto& ts : ctx->tail_sets) if (ts.reserve == ctx->tail_reserve_cus) ts.chain = st;
            ctx->tail_chain = st;
        }
        if (want_masked_aux && !ctx->tail_aux) {
            hipStream_t st = nullptr;
            ZK_TRY(stream_create_own_queue(ctx, &st, ctx->tail_reserve_cus));
            for (auto& ts : ctx->tail_sets) if (ts.reserve == ctx->tail_reserve_cus) ts.aux = st;
            ctx->tail_aux = st;
        }
        if (!want_masked_aux && !ctx->tail_aux_free) ZK_TRY(stream_create_own_queue(ctx, &ctx->tail_aux_free, 0));
        // the digit streams (decompose, sort, filter) are HBM-bound helpers that starve beside the VALU-bound kernels of the main stream
        // (profiles/r03_timeline_*.txt); "tail_aux_masked" 0 lets them use the reserved compute units as well — on a stream with its own hardware queue
        main_s = ctx->tail_stream; aux_s = want_masked_aux ? ctx->tail_aux : ctx->tail_aux_free;
    }
    // "msm_chain": the sums' partial-sum levels / reductions / copies on a stream of their own (msm.cuh MsmChain), two workspace regions taking turns
    // Only for a tail on streams with hardware queues of their own (`masked`: two workers per GPU, the headline shape) unless "msm_chain" is 2: on ORDINARY streams —
    // which the runtime multiplexes onto four hardware queues in order — the chain's short launches land in a queue behind the digit stream's long sort
    // kernels, or in front of the next level-1 kernel: measured 281 -> 359 ms per prove tail (gpurun_out/r06x, one worker), against 318 -> 298 ms where every
    // stream has its queue.  2 = also then (tests run both paths).
    hipStream_t chain_s = nullptr;
    if (ctx->msm_chain) {
        if (masked) chain_s = ctx->tail_chain;
        else if (ctx->msm_chain == 2) {
            if (!ctx->chain_stream) ZK_TRY(stream_create_own_queue(ctx, &ctx->chain_stream, 0));    // its own hardware queue (full CU mask)
            chain_s = ctx->chain_stream;
        }
    }
    hipEvent_t e_start = ev_get(ctx), e_h = ev_get(ctx), e_w = ev_get(ctx), e_hs = ev_get(ctx), e_up = ev_get(ctx), e_wB = ev_get(ctx), e_wK = ev_get(ctx);
    MsmChain chains[5];
    for (auto& ch : chains) { ch.stream = chain_s; ch.ev_level1 = ev_get(ctx); ch.ev_done = ev_get(ctx); }
    struct EvGuard { zkpor_ctx* c; hipEvent_t e[7]; hipStream_t m; MsmChain* ch; ~EvGuard() { c->stream = m; for (auto x : e) c->event_pool.push_back(x); for (int i = 0; i < 5; ++i) { c->event_pool.push_back(ch[i].ev_level1); c->event_pool.push_back(ch[i].ev_done); } } } guard{ctx, {e_start, e_h, e_w, e_hs, e_up, e_wB, e_wK}, caller_s, chains};
    // several workers of a GPU with a reserved-CU tail: ONE prove tail at a time (two would only time-slice each other on the same compute units), so
    // that the other worker's SOLVE is what runs beside it; the waiting worker sleeps here, its solver's prefetched chains keep running
    // Round 6: when ANOTHER worker's tail holds the device, this proof's digit stream of w (decompose + sort + filters: bandwidth and LDS, no field
    // arithmetic) is built BEFORE the turn is waited for — beside the other tail's accumulations, where LDS is free — instead of beside this proof's
    // own NTT passes, whose four 36 KB tiles per compute unit it would displace ("tail_digits_early", default 1; a free device keeps the usual order).
    GpuTurn own_turn;
    bool early_digits = false;
    if (masked) {
        if (ctx->tail_digits_early && do_w && !host && !own_turn.try_acquire(ctx)) early_digits = true;
        else { own_turn.acquire(ctx); turn = &own_turn; }       // (a no-op when try_acquire got it)
    }
    if (main_s != caller_s) {   // whatever the caller queued on the context's stream (the solver, a / b / c, uploads) comes first
        ZK_HIP(ctx, hipEventRecord(e_start, caller_s));
        ZK_HIP(ctx, hipStreamWaitEvent(main_s, e_start, 0));
        ctx->stream = main_s;
    }
    // per-array digit streams: B1 / B2 and K get the shared stream of w minus the entries of their absent points
    StreamFilter filt;
    if (ctx->msm_filter && do_w) { filt.absent[0] = pk->absentB; filt.absent[1] = pk->absentK; }
    const int n_filters = (filt.absent[0] ? 1 : 0) + (filt.absent[1] ? 1 : 0);
    MsmCfg cfgw = msm_cfg(ctx, pk->n_wires, pk->tab_m);
    MsmCfg cfgh = msm_cfg(ctx, nZ ? nZ : 1, pk->tab_m);
    ZK_TRY(check_tables(ctx, pk, cfgw, pk->tab_shift_w));
    if (nZ) ZK_TRY(check_tables(ctx, pk, cfgh, pk->tab_shift_z));
    size_t sortw = 0, sorth = 0;
    size_t need_dw = digits_ws_bytes(ctx, pk->n_wires, cfgw, &sortw, n_filters);
    size_t need_dh = digits_ws_bytes(ctx, nZ ? nZ : 1, cfgh, &sorth);
    size_t need_aw = accumulate_ws_bytes<Fp2>(cfgw, pk->n_wires * (size_t)cfgw.W);
    size_t need_ah = accumulate_ws_bytes<Fp>(cfgh, (nZ ? nZ : 1) * (size_t)cfgh.W);
    // one accumulation region the sums reuse in stream order — or, with a chain stream, two that take turns: region 0 (here, behind the digit streams) for
    // A, K, Z, region 1 (ctx->ws2, below) for B1 and B2.  A sum's level-1 kernel waits for the chain of the region's previous user (which has had a whole
    // level-1 kernel's time to finish)
    size_t region0 = need_aw > need_ah ? need_aw : need_ah;
    if (chain_s) {
        const size_t need_aw1 = accumulate_ws_bytes<Fp>(cfgw, pk->n_wires * (size_t)cfgw.W);
        region0 = need_aw1 > need_ah ? need_aw1 : need_ah;
    }
    ZK_TRY(ws_reserve(ctx, need_dw + need_dh + region0));
    ZK_TRY(ensure_pinned(ctx, 16 * MSM_SLOT_BYTES));
    char* pin = (char*)ctx->pinned;
    const size_t D = (size_t)1 << n;
    {   // ZKPOR_DEBUG_ADDR=1: the address ranges this call's kernels may touch, on stderr — a GPU memory fault report names an address, this names the buffer
        static const bool dbg = [] { const char* e = getenv("ZKPOR_DEBUG_ADDR"); return e && e[0] == '1'; }();
        if (dbg) {
            size_t fr = 0, tot = 0;
            (void)hipMemGetInfo(&fr, &tot);
            fprintf(stderr, "[zkpor addr] prove_sums log2=%d host=%d nw=%zu nZ=%zu ws=%p+%zu (dw %zu dh %zu) stage=%p+%zu w=%p a=%p b=%p c=%p A=%p B1=%p K=%p Z=%p B2=%p mB=%p mK=%p pin=%p free=%zuMB\n",
                    n, host ? 1 : 0, pk->n_wires, nZ, (void*)ctx->ws, ctx->ws_cap, need_dw, need_dh, (void*)ctx->stage, ctx->stage_cap, d_w, d_a, d_b, d_c, (void*)pk->A, (void*)pk->B1, (void*)pk->K,
                    (void*)pk->Z, (void*)pk->B2, (void*)pk->absentB, (void*)pk->absentK, (void*)pin, fr >> 20);
        }
    }
    DigitStream dsw, dsh, dswB, dswK;
    auto digits_w = [&](hipEvent_t after) -> int32_t {   // the digit stream of the witness on the auxiliary stream (serves A, B1, B2, K)
        ctx->stream = aux_s;
        // `after` orders the digit kernels (they write the shared workspace, they read w) behind whatever is already queued on the main / the caller's stream —
        // also on the host path, where e_up (the copy stream) alone would not; free when that stream is idle
        ZK_HIP(ctx, hipStreamWaitEvent(aux_s, after, 0));
        if (host) ZK_HIP(ctx, hipStreamWaitEvent(aux_s, e_up, 0));
        // e_w: the sorted shared stream (A starts on it); e_wB: + the B filter; e_wK: + the K filter (the filters hide under A)
        ZK_TRY(msm_digits(ctx, (const Fr*)d_w, pk->n_wires, cfgw, sortw, &dsw, n_filters ? &filt : nullptr, &dswB, &dswK, e_w, e_wB));
        ZK_HIP(ctx, hipEventRecord(e_wK, aux_s));
        ctx->stats["msm_entries_w"] = dsw.M; ctx->stats["msm_entries_w_B"] = dswB.M; ctx->stats["msm_entries_w_K"] = dswK.M;   // bucket additions of A / B1, B2 / K
        ctx->stream = main_s;
        return ZKPOR_OK;
    };
    size_t off_dh = need_dw;
    if (early_digits) {
        if (main_s == caller_s) ZK_HIP(ctx, hipEventRecord(e_start, caller_s));     // (else recorded above: everything the caller queued, the solver's w included)
        ZK_TRY(digits_w(e_start));
        off_dh = ctx->ws_off;
        own_turn.acquire(ctx); turn = &own_turn;
    }
    ZK_HIP(ctx, hipEventRecord(e_start, main_s));
    if (host) {
        // w first: everything the witness sums need
        ZK_TRY(host_upload(ctx, const_cast<void*>(d_w), host->w, pk->n_wires * sizeof(Fr)));
        ZK_HIP(ctx, hipEventRecord(e_up, ctx->copy_stream));
    } else if (d_b) {
        // 1. h = computeH(a,b,c) on the main stream, left in d_a (bit-reversed = the order of pk->Z)
        if (kept) ZK_TRY(compute_h_dev(ctx, n, (Fr*)d_a, (Fr*)d_b, (Fr*)d_c, (const Fr*)kept->a, (const Fr*)kept->b, (const Fr*)kept->c));
        else ZK_TRY(compute_h_dev(ctx, n, (Fr*)d_a, (Fr*)d_b, (Fr*)d_c));
    }
    if (!host) ZK_HIP(ctx, hipEventRecord(e_h, main_s));
    // 2. digit stream of the witness on the auxiliary stream (unless it was built before the turn)
    const size_t mark = need_dw + need_dh;
    MsmPending pA, pB1, pK, pB2, pZ;
    hipEvent_t region_busy[2] = {nullptr, nullptr};     // ev_done of the chain that used the region last
    int taken_region[5] = {0, 0, 0, 0, 0};
    hipEvent_t last_chain = nullptr, last_level1 = nullptr;
    // sum `which` (0 A, 1 B1, 2 K, 3 B2, 4 Z) takes its region: the level-1 kernel (it clears the buckets) behind the region's previous chain.
    // Region 1 is an arena of its own, sized when B1 is queued — by then the host knows the entry count of the B stream — for the larger of its two users
    // (B2's G2 images) plus an eighth: 6.4 GB for zkpor50_1380 instead of the 13.2 GB a worst-case stream (every digit of every scalar non-zero) would need,
    // so that the two regions together are no larger than the one region of round 5.  It grows (behind a wait for the chain stream) when a witness needs more;
    // if it cannot, the sum runs unchained in region 0.
    struct ArenaSwap {   // ws_alloc serves from ctx->ws: a sum of region 1 sees ctx->ws2 as its arena for the duration of its queueing
        zkpor_ctx* c; char* ws; size_t cap; bool on = false;
        explicit ArenaSwap(zkpor_ctx* c_) : c(c_), ws(c_->ws), cap(c_->ws_cap) {}
        void to_ws2() { c->ws = c->ws2; c->ws_cap = c->ws2_cap; c->ws_off = 0; on = true; }
        ~ArenaSwap() { if (on) { c->ws = ws; c->ws_cap = cap; } }
    };
    auto region_of = [&](int which) { return (chain_s && (which == 1 || which == 3)) ? 1 : 0; };
    auto take_region = [&](int which, ArenaSwap& arena) -> const MsmChain* {
        int r = region_of(which);
        if (r == 1) {
            size_t need = accumulate_ws_bytes<Fp2>(cfgw, dswB.M) + (1u << 20);
            if (need > ctx->ws2_cap) {
                need += need / 8;
                (void)hipStreamSynchronize(chain_s);      // region 1's previous user
                if (ctx->ws2) { (void)hipFree(ctx->ws2); ctx->ws2 = nullptr; ctx->ws2_cap = 0; }
                if (hipMalloc((void**)&ctx->ws2, need) == hipSuccess) ctx->ws2_cap = need;
                else { (void)hipGetLastError(); ctx->ws2 = nullptr; r = 0; }
            }
        }
        if (r == 1) arena.to_ws2(); else ctx->ws_off = mark;
        if (!chain_s) return nullptr;
        if (region_busy[0] && r == 0) (void)hipStreamWaitEvent(main_s, region_busy[0], 0);
        if (region_busy[1] && r == 1) (void)hipStreamWaitEvent(main_s, region_busy[1], 0);
        taken_region[which] = r;
        return (r == 1 || region_of(which) == 0) ? &chains[which] : nullptr;     // a region-1 sum that fell back to region 0 runs unchained
    };
    // ... and leaves its events behind — only if it recorded them (an empty sum launches nothing)
    auto queued = [&](int which, const MsmPending& p) {
        if (!p.chained) return;
        region_busy[taken_region[which]] = chains[which].ev_done;
        last_chain = chains[which].ev_done; last_level1 = chains[which].ev_level1;
    };
    auto queue_b2 = [&]() -> int32_t {
        ArenaSwap arena(ctx);
        const MsmChain* ch = take_region(3, arena);
        ZK_TRY(msm_accumulate_launch<Fp2>(ctx, dswB, pk->B2, pin + 6 * MSM_SLOT_BYTES, pin + 8 * MSM_SLOT_BYTES, &pB2, ch));
        queued(3, pB2);
        return ZKPOR_OK;
    };
    if (do_w) {
        if (!early_digits) {
            ZK_TRY(digits_w(e_start));
            off_dh = ctx->ws_off;
        }
        // 3. queue the witness accumulations (they reuse one workspace region in stream order)
        // "msm_filter" 2: A waits for the filter as well, so that the filter runs on an otherwise idle GPU (with a full-size grid) instead of
        // beside A's VALU-bound kernel, where it is starved and starves (profiles/r03_filter.txt)
        ZK_HIP(ctx, hipStreamWaitEvent(main_s, (n_filters && ctx->msm_filter == 2) ? e_wK : e_w, 0));
        {
            ArenaSwap arena(ctx);
            const MsmChain* ch = take_region(0, arena);
            ZK_TRY(msm_accumulate_launch<Fp>(ctx, dsw, pk->A, pin + 0 * MSM_SLOT_BYTES, pin + 1 * MSM_SLOT_BYTES, &pA, ch));
            queued(0, pA);
        }
        ZK_HIP(ctx, hipStreamWaitEvent(main_s, e_wB, 0));
        {
            ArenaSwap arena(ctx);
            const MsmChain* ch = take_region(1, arena);
            ZK_TRY(msm_accumulate_launch<Fp>(ctx, dswB, pk->B1, pin + 2 * MSM_SLOT_BYTES, pin + 3 * MSM_SLOT_BYTES, &pB1, ch));
            queued(1, pB1);
        }
        ZK_HIP(ctx, hipStreamWaitEvent(main_s, e_wK, 0));
        {
            ArenaSwap arena(ctx);
            const MsmChain* ch = take_region(2, arena);
            ZK_TRY(msm_accumulate_launch<Fp>(ctx, dswK, pk->K, pin + 4 * MSM_SLOT_BYTES, pin + 5 * MSM_SLOT_BYTES, &pK, ch));
            queued(2, pK);
        }
        if (!host) ZK_TRY(queue_b2());
    }
    if (host) {
        // a, b, c cross PCIe while A, B1, K run; the rows past n_constraints are the zero padding computeH expects
        const void* src[3] = {host->a, host->b, host->c};
        void* dst[3] = {d_a, d_b, d_c};
        for (int i = 0; i < 3; ++i) {
            if (host->n_constraints < D)
                ZK_HIP(ctx, hipMemsetAsync((Fr*)dst[i] + host->n_constraints, 0, (D - host->n_constraints) * sizeof(Fr), ctx->copy_stream));
            ZK_TRY(host_upload(ctx, dst[i], src[i], host->n_constraints * sizeof(Fr)));
        }
        ZK_HIP(ctx, hipEventRecord(e_up, ctx->copy_stream));
        ZK_HIP(ctx, hipStreamWaitEvent(main_s, e_up, 0));
        ZK_TRY(compute_h_dev(ctx, n, (Fr*)d_a, (Fr*)d_b, (Fr*)d_c));
        ZK_HIP(ctx, hipEventRecord(e_h, main_s));
        if (do_w) ZK_TRY(queue_b2());
    }
    if (nZ) {
        // 4. digit stream of h on the auxiliary stream, overlapping the accumulations queued after computeH
        ctx->stream = aux_s;
        ctx->ws_off = off_dh;
        ZK_HIP(ctx, hipStreamWaitEvent(aux_s, e_h, 0));
        ZK_TRY(msm_digits(ctx, (const Fr*)d_a, nZ, cfgh, sorth, &dsh));
        ctx->stats["msm_entries_h"] = dsh.M;
        ZK_HIP(ctx, hipEventRecord(e_hs, aux_s));
        ctx->stream = main_s;
        // 5. Z . h
        ZK_HIP(ctx, hipStreamWaitEvent(main_s, e_hs, 0));
        ArenaSwap arena(ctx);
        const MsmChain* ch = take_region(4, arena);
        ZK_TRY(msm_accumulate_launch<Fp>(ctx, dsh, pk->Z, pin + 10 * MSM_SLOT_BYTES, pin + 11 * MSM_SLOT_BYTES, &pZ, ch));
        queued(4, pZ);
    }
    if (last_chain) ZK_HIP(ctx, hipStreamWaitEvent(main_s, last_chain, 0));   // the chain stream runs in order: its last chain is behind every other
    if (while_gpu_runs) (*while_gpu_runs)();  // host work that needs no device result: everything is queued, nothing is waited for yet
    if (turn && last_level1) {
        // the last sum's level-1 kernel is the last launch of this call that fills the device: behind it only a chain of short dependent launches is
        // left, and the next caller's first kernels (its NTT passes) may as well run beside that
        ZK_HIP(ctx, hipEventSynchronize(last_level1));
        turn->release();
    }
    ZK_HIP(ctx, hipStreamSynchronize(main_s));
    if (turn) turn->release();   // the device is free for the next caller while this one finishes on the host
    HostPhase hp(ctx, "host_assembly");
    if (do_w) {
        msm_accumulate_finish<Fp>(pA, &out->A);
        msm_accumulate_finish<Fp>(pB1, &out->B1);
        msm_accumulate_finish<Fp>(pK, &out->K);
        msm_accumulate_finish<Fp2>(pB2, &out->B2);
    } else {
        out->A = out->B1 = out->K = G1XYZZ::inf();
        out->B2 = G2XYZZ::inf();
    }
    if (nZ) msm_accumulate_finish<Fp>(pZ, &out->Z); else out->Z = G1XYZZ::inf();
    return ZKPOR_OK;
}

// blinding and assembly on the host: Ar = alpha + A.w + r delta, Bs = beta + B.w + s delta (G1 and G2),
// Krs = K.w + Z.h + s Ar + r Bs1 - rs delta.  Four of the six scalar multiplications involve only the key and (r, s): they are
// done by blind_prepare() while the GPU is still busy with the sums (the host thread would otherwise just wait for the stream).
struct Blind { Fr rc, sc; G1XYZZ dr, ds, dkrs; G2XYZZ d2s; };
Blind blind_prepare(const G1Affine& delta, const G2Affine& delta2, const uint64_t r[4], const uint64_t s[4]) {
    Blind b;
    Fr rm, sm;
    memcpy(&rm, r, 32); memcpy(&sm, s, 32);
    b.rc = Fr::from_mont(rm); b.sc = Fr::from_mont(sm);
    Fr krc = Fr::from_mont(Fr::neg(Fr::mul(rm, sm)));
    G1XYZZ d1 = g1x(delta);
    b.dr = xyzz_mul_limbs<Fp>(d1, b.rc.v);
    b.ds = xyzz_mul_limbs<Fp>(d1, b.sc.v);
    b.dkrs = xyzz_mul_limbs<Fp>(d1, krc.v);
    b.d2s = xyzz_mul_limbs<Fp2>(xyzz_from_affine<Fp2>(delta2), b.sc.v);
    return b;
}
void assemble(const G1Affine& alpha, const G1Affine& beta, const G2Affine& beta2, const ProveSums& m, const Blind& b,
              uint8_t proof_out[256]) {
    G1XYZZ ar = m.A;
    xyzz_add<Fp>(ar, g1x(alpha));
    xyzz_add<Fp>(ar, b.dr);
    G1XYZZ bs1 = m.B1;
    xyzz_add<Fp>(bs1, g1x(beta));
    xyzz_add<Fp>(bs1, b.ds);
    G2XYZZ bs2 = m.B2;
    xyzz_add<Fp2>(bs2, xyzz_from_affine<Fp2>(beta2));
    xyzz_add<Fp2>(bs2, b.d2s);
    G1XYZZ krs = m.K;
    xyzz_add<Fp>(krs, m.Z);
    xyzz_add<Fp>(krs, b.dkrs);
    xyzz_add<Fp>(krs, xyzz_mul_limbs<Fp>(ar, b.sc.v));
    xyzz_add<Fp>(krs, xyzz_mul_limbs<Fp>(bs1, b.rc.v));
    G1Affine ara = xyzz_to_affine<Fp>(ar), krsa = xyzz_to_affine<Fp>(krs);
    G2Affine bsa = xyzz_to_affine<Fp2>(bs2);
    memcpy(proof_out, &ara, 64); memcpy(proof_out + 64, &bsa, 128); memcpy(proof_out + 192, &krsa, 64);
}

// r, s must be canonical residues (limbs, as an integer, below the modulus): anything else is not an fr.Element gnark could hold
bool fr_canonical(const uint64_t x[4]) {
    for (int i = 3; i >= 0; --i) {
        u64 m = ((u64)FrParams::mod(2 * i + 1) << 32) | FrParams::mod(2 * i);
        if (x[i] != m) return x[i] < m;
    }
    return false;
}
int32_t check_blinding(zkpor_ctx* ctx, const uint64_t r[4], const uint64_t s[4]) {
    if (!fr_canonical(r) || !fr_canonical(s)) { if (ctx) ctx->err = "prove: blinding scalar not below the modulus"; return ZKPOR_E_ARG; }
    return ZKPOR_OK;
}
int32_t check_same_gpu(zkpor_ctx* ctx, zkpor_pk* pk) {
    if (pk->ctx->device != ctx->device) { ctx->err = "prove: the key lives on another GPU than the context"; return ZKPOR_E_ARG; }
    return ZKPOR_OK;
}

template <class F>
XYZZ<F> jac_in(const uint8_t* p) {
    Jacobian<F> j;
    memcpy(&j, p, sizeof(j));
    if (j.z.is_zero()) return XYZZ<F>::inf();
    F zz = F::sqr(j.z);
    return XYZZ<F>{j.x, j.y, zz, F::mul(zz, j.z)};
}
template <class F>
void jac_out(const XYZZ<F>& p, uint8_t* out) {
    Jacobian<F> j = xyzz_to_jacobian<F>(p);
    memcpy(out, &j, sizeof(j));
}

}  // namespace
extern "C" {

int32_t zkpor_prove_tail_dev(zkpor_ctx* ctx, zkpor_pk* pk, const void* d_w, void* d_a, void* d_b, void* d_c,
                             const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[256]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !pk || !d_w || !d_a || !d_b || !d_c || !r || !s || !proof_out) return ZKPOR_E_ARG;
    if (!pk->ready) { ctx->err = "prove: key not loaded"; return ZKPOR_E_STATE; }
    if (pk->shard) { ctx->err = "prove: the key is a shard (zkpor_pk_keep_range): use zkpor_prove_sums_dev + zkpor_prove_assemble"; return ZKPOR_E_STATE; }
    ZK_TRY(check_same_gpu(ctx, pk));
    ZK_TRY(check_blinding(ctx, r, s));
    ProveSums m;
    Blind bl;
    const std::function<void()> prep = [&] { bl = blind_prepare(pk->delta, pk->delta2, r, s); };
    ZK_TRY(prove_sums(ctx, pk, d_w, d_a, d_b, d_c, &m, true, true, &prep));
    HostPhase hp(ctx, "host_assembly");
    assemble(pk->alpha, pk->beta, pk->beta2, m, bl, proof_out);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_prove_tail_dev_keep(zkpor_ctx* ctx, zkpor_pk* pk, const void* d_w, const void* d_a, const void* d_b, const void* d_c, void* d_wa,
                                  void* d_wb, void* d_wc, const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[256]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !pk || !d_w || !d_a || !d_b || !d_c || !d_wa || !d_wb || !d_wc || !r || !s || !proof_out) return ZKPOR_E_ARG;
    if (d_wa == d_a || d_wb == d_b || d_wc == d_c) { ctx->err = "prove: the work buffers must differ from the inputs (use zkpor_prove_tail_dev to work in place)"; return ZKPOR_E_ARG; }
    if (!pk->ready) { ctx->err = "prove: key not loaded"; return ZKPOR_E_STATE; }
    if (pk->shard) { ctx->err = "prove: the key is a shard (zkpor_pk_keep_range): use zkpor_prove_sums_dev + zkpor_prove_assemble"; return ZKPOR_E_STATE; }
    ZK_TRY(check_same_gpu(ctx, pk));
    ZK_TRY(check_blinding(ctx, r, s));
    ProveSums m;
    Blind bl;
    const std::function<void()> prep = [&] { bl = blind_prepare(pk->delta, pk->delta2, r, s); };
    const KeptInputs kept{d_a, d_b, d_c};
    ZK_TRY(prove_sums(ctx, pk, d_w, d_wa, d_wb, d_wc, &m, true, true, &prep, nullptr, nullptr, &kept));
    HostPhase hp(ctx, "host_assembly");
    assemble(pk->alpha, pk->beta, pk->beta2, m, bl, proof_out);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

// ---- single-proof split (BASELINE.json configs[4]): every GPU holds a contiguous range of each key array
int32_t zkpor_pk_keep_range(zkpor_pk* pk, size_t wire_lo, size_t wire_hi, size_t z_lo, size_t z_hi) try {
    ZK_ENTER(pk ? pk->ctx->device : -1);
    if (!pk) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = pk->ctx;
    if (!pk->ready) { ctx->err = "pk: key not loaded"; return ZKPOR_E_STATE; }
    if (pk->tab_m > 1) { ctx->err = "pk: a key with fixed-base tables cannot be cut into a shard (load it with msm_tables = 1)"; return ZKPOR_E_STATE; }
    if (wire_lo >= wire_hi || wire_hi > pk->n_wires || z_lo > z_hi || z_hi > pk->nZ) { ctx->err = "pk: shard range outside the key"; return ZKPOR_E_ARG; }
    auto cut = [&](auto** arr, size_t lo, size_t hi) -> int32_t {
        using P = std::remove_pointer_t<std::remove_pointer_t<decltype(arr)>>;
        P* fresh = nullptr;
        size_t n = hi - lo;
        if (hipMalloc((void**)&fresh, (n ? n : 1) * sizeof(P)) != hipSuccess) { ctx->err = "pk: shard allocation failed"; return ZKPOR_E_OOM; }
        if (n && hipMemcpyAsync(fresh, *arr + lo, n * sizeof(P), hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { (void)hipFree(fresh); ctx->err = "pk: shard copy failed"; return ZKPOR_E_HIP; }
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { (void)hipFree(fresh); ctx->err = "pk: shard copy failed"; return ZKPOR_E_HIP; }
        (void)hipFree(*arr);
        *arr = fresh;
        return ZKPOR_OK;
    };
    pk->ready = false;  // a failure half-way leaves arrays of different lengths: the key must be reloaded
    ZK_TRY(cut(&pk->A, wire_lo, wire_hi));
    ZK_TRY(cut(&pk->B1, wire_lo, wire_hi));
    ZK_TRY(cut(&pk->K, wire_lo, wire_hi));
    ZK_TRY(cut(&pk->B2, wire_lo, wire_hi));
    ZK_TRY(cut(&pk->Z, z_lo, z_hi));
    pk->n_wires = wire_hi - wire_lo;
    pk->nZ = z_hi - z_lo;
    pk->shard = true;
    pk_free_masks(pk);   // a shard's arrays were cut: it proves with the shared stream
    pk->ready = true;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((pk ? pk->ctx : nullptr))

int32_t zkpor_prove_sums_dev(zkpor_ctx* ctx, zkpor_pk* pk, const void* d_w, const void* d_h, uint8_t sums_out[576]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !pk || (!d_w && !d_h) || !sums_out) return ZKPOR_E_ARG;
    if (!pk->ready) { ctx->err = "prove: key not loaded"; return ZKPOR_E_STATE; }
    ProveSums m;
    ZK_TRY(prove_sums(ctx, pk, d_w, const_cast<void*>(d_h), nullptr, nullptr, &m, d_w != nullptr, d_h != nullptr));
    jac_out<Fp>(m.A, sums_out); jac_out<Fp>(m.B1, sums_out + 96); jac_out<Fp2>(m.B2, sums_out + 192);
    jac_out<Fp>(m.K, sums_out + 384); jac_out<Fp>(m.Z, sums_out + 480);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_prove_assemble(const void* alpha, const void* beta, const void* delta, const void* beta2, const void* delta2,
                             const uint8_t sums[576], const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[256]) try {
    if (!alpha || !beta || !delta || !beta2 || !delta2 || !sums || !r || !s || !proof_out) return ZKPOR_E_ARG;
    ZK_TRY(check_blinding(nullptr, r, s));
    G1Affine a1, b1, d1; G2Affine b2, d2;
    memcpy(&a1, alpha, 64); memcpy(&b1, beta, 64); memcpy(&d1, delta, 64); memcpy(&b2, beta2, 128); memcpy(&d2, delta2, 128);
    ProveSums m;
    m.A = jac_in<Fp>(sums); m.B1 = jac_in<Fp>(sums + 96); m.B2 = jac_in<Fp2>(sums + 192);
    m.K = jac_in<Fp>(sums + 384); m.Z = jac_in<Fp>(sums + 480);
    assemble(a1, b1, b2, m, blind_prepare(d1, d2, r, s), proof_out);
    return ZKPOR_OK;
} ZK_ABI_CATCH

int32_t zkpor_pk_consts(zkpor_pk* pk, void* alpha, void* beta, void* delta, void* beta2, void* delta2) try {
    ZK_ENTER(pk ? pk->ctx->device : -1);
    if (!pk || !alpha || !beta || !delta || !beta2 || !delta2) return ZKPOR_E_ARG;
    if (!pk->ready) { pk->ctx->err = "pk: key not loaded"; return ZKPOR_E_STATE; }
    memcpy(alpha, &pk->alpha, 64); memcpy(beta, &pk->beta, 64); memcpy(delta, &pk->delta, 64);
    memcpy(beta2, &pk->beta2, 128); memcpy(delta2, &pk->delta2, 128);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((pk ? pk->ctx : nullptr))

int32_t zkpor_prove_tail(zkpor_ctx* ctx, zkpor_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b,
                         const uint64_t* c, size_t n_constraints, const uint64_t r[4], const uint64_t s[4],
                         uint8_t proof_out[256]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !pk || !w || !a || !b || !c || !r || !s || !proof_out) return ZKPOR_E_ARG;
    if (!pk->ready) { ctx->err = "prove: key not loaded"; return ZKPOR_E_STATE; }
    if (pk->shard) { ctx->err = "prove: the key is a shard (zkpor_pk_keep_range): use zkpor_prove_sums_dev + zkpor_prove_assemble"; return ZKPOR_E_STATE; }
    size_t D = (size_t)1 << pk->log2_domain;
    if (n_constraints > D) { ctx->err = "prove: more constraints than the domain"; return ZKPOR_E_ARG; }
    ZK_TRY(check_same_gpu(ctx, pk));
    ZK_TRY(check_blinding(ctx, r, s));
    // persistent staging in HBM (no allocation per proof); the copies are queued by prove_sums in the order the GPU needs them
    ZK_TRY(stage_reserve(ctx, (3 * D + pk->n_wires) * sizeof(Fr)));
    Fr* d = (Fr*)ctx->stage;
    HostInputs host{w, a, b, c, n_constraints};
    ProveSums m;
    Blind bl;
    const std::function<void()> prep = [&] { bl = blind_prepare(pk->delta, pk->delta2, r, s); };
    int32_t rc = ZKPOR_OK;
    GpuTurn turn;
    if (ctx->host_order == 0 && turn.try_acquire(ctx)) {
        // nobody else is on the device: w first, a, b, c underneath the witness sums (the shortest single proof)
        rc = prove_sums(ctx, pk, d + 3 * D, d, d + D, d + 2 * D, &m, true, true, &prep, &host, &turn);
    } else {
        // another caller's proof is running (or "host_order" 1): everything crosses PCIe underneath it, then this proof takes its
        // turn with all inputs resident, in the resident order (computeH first, the digit stream of w hidden under it)
        const void* src[4] = {w, a, b, c};
        Fr* dst[4] = {d + 3 * D, d, d + D, d + 2 * D};
        for (int i = 0; i < 4 && rc == ZKPOR_OK; ++i) {
            const size_t n = i ? n_constraints : pk->n_wires;
            if (i && n < D && hipMemsetAsync(dst[i] + n, 0, (D - n) * sizeof(Fr), ctx->copy_stream) != hipSuccess) { ctx->err = "prove: memset failed"; rc = ZKPOR_E_HIP; break; }
            rc = host_upload(ctx, dst[i], src[i], n * sizeof(Fr));
        }
        hipEvent_t e_up = ev_get(ctx);
        if (rc == ZKPOR_OK && hipEventRecord(e_up, ctx->copy_stream) != hipSuccess) { ctx->err = "prove: event on the copy stream failed"; rc = ZKPOR_E_HIP; }
        if (rc == ZKPOR_OK) {
            turn.acquire(ctx);
            if (hipStreamWaitEvent(ctx->stream, e_up, 0) != hipSuccess) { ctx->err = "prove: event on the copy stream failed"; rc = ZKPOR_E_HIP; }
        }
        ctx->event_pool.push_back(e_up);
        if (rc == ZKPOR_OK) rc = prove_sums(ctx, pk, d + 3 * D, d, d + D, d + 2 * D, &m, true, true, &prep, nullptr, &turn);
    }
    if (rc != ZKPOR_OK) {  // nothing may still read the caller's memory or the staging area when the call returns
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
        zk::drain_tail_streams(ctx);
        return rc;
    }
    HostPhase hp(ctx, "host_assembly");
    assemble(pk->alpha, pk->beta, pk->beta2, m, bl, proof_out);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

// Host-pointer form with the constraint matrices resident (zkpor_r1cs_*): only w crosses PCIe (n_wires x 32 B instead of
// n_wires + 3 n_constraints); a, b, c are evaluated in the staging area, then the resident order of prove_sums runs
// (computeH first, decompose + sort of w hidden under it).
int32_t zkpor_prove_r1cs(zkpor_ctx* ctx, zkpor_pk* pk, zkpor_r1cs* r1cs, const uint64_t* w, const uint64_t r[4], const uint64_t s[4],
                         uint8_t proof_out[256]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !pk || !r1cs || !w || !r || !s || !proof_out) return ZKPOR_E_ARG;
    if (!pk->ready) { ctx->err = "prove: key not loaded"; return ZKPOR_E_STATE; }
    if (pk->shard) { ctx->err = "prove: the key is a shard (zkpor_pk_keep_range): use zkpor_prove_sums_dev + zkpor_prove_assemble"; return ZKPOR_E_STATE; }
    const size_t D = (size_t)1 << pk->log2_domain;
    size_t nc = 0, nw = 0;
    int dev = -1;
    r1cs_dims(r1cs, &nc, &nw, &dev);
    if (dev != ctx->device) { ctx->err = "prove: the constraint matrices live on another GPU than the context"; return ZKPOR_E_ARG; }
    if (nw != pk->n_wires) { ctx->err = "prove: the constraint system and the key disagree on the number of wires"; return ZKPOR_E_ARG; }
    if (nc > D) { ctx->err = "prove: more constraints than the domain"; return ZKPOR_E_ARG; }
    ZK_TRY(check_same_gpu(ctx, pk));
    ZK_TRY(check_blinding(ctx, r, s));
    ZK_TRY(stage_reserve(ctx, (3 * D + pk->n_wires) * sizeof(Fr)));
    Fr* d = (Fr*)ctx->stage;
    Fr* d_w = d + 3 * D;
    auto drain = [&] {  // nothing may still read the caller's memory or the staging area when the call returns
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
        zk::drain_tail_streams(ctx);
    };
    GpuTurn turn;
    int32_t rc = host_upload(ctx, d_w, w, pk->n_wires * sizeof(Fr));
    hipEvent_t e_up = ev_get(ctx);
    if (rc == ZKPOR_OK && hipEventRecord(e_up, ctx->copy_stream) != hipSuccess) { ctx->err = "prove: event on the copy stream failed"; rc = ZKPOR_E_HIP; }
    if (rc == ZKPOR_OK) {
        turn.acquire(ctx);   // w crossed underneath whoever was on the device
        if (hipStreamWaitEvent(ctx->stream, e_up, 0) != hipSuccess) { ctx->err = "prove: event on the copy stream failed"; rc = ZKPOR_E_HIP; }
    }
    ctx->event_pool.push_back(e_up);
    if (rc == ZKPOR_OK) rc = r1cs_eval_on(ctx, r1cs, d_w, d, d + D, d + 2 * D, D);
    ProveSums m;
    Blind bl;
    const std::function<void()> prep = [&] { bl = blind_prepare(pk->delta, pk->delta2, r, s); };
    if (rc == ZKPOR_OK) rc = prove_sums(ctx, pk, d_w, d, d + D, d + 2 * D, &m, true, true, &prep, nullptr, &turn);
    if (rc != ZKPOR_OK) { drain(); return rc; }
    HostPhase hp(ctx, "host_assembly");
    assemble(pk->alpha, pk->beta, pk->beta2, m, bl, proof_out);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

// groth16.Prove from the ASSIGNED INPUTS, everything after them on the device (SURVEY §8 f4 + f1 + a6): the inputs (1 + nPublic + nSecret
// elements, gnark's order) cross PCIe, the solver program fills the wire vector in HBM (csrc/solver.hip), a, b, c are evaluated from it
// (csrc/r1cs.hip) and the prove tail runs in its resident order.  For circuits whose solver program holds no external hint; a circuit with a
// BSB22 commitment drives the same steps itself (zkpor_solver_start_dev / _external_* / _resume_dev, zkpor_commit_dev, zkpor_r1cs_eval_dev,
// zkpor_prove_tail_dev — INTEGRATION.md §1c).
int32_t zkpor_prove_inputs(zkpor_ctx* ctx, zkpor_pk* pk, zkpor_r1cs* r1cs, zkpor_solver* solver, const uint64_t* inputs, size_t n_inputs,
                           const uint64_t r[4], const uint64_t s[4], uint8_t proof_out[256]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !pk || !r1cs || !solver || !inputs || !r || !s || !proof_out) return ZKPOR_E_ARG;
    if (!pk->ready) { ctx->err = "prove: key not loaded"; return ZKPOR_E_STATE; }
    if (pk->shard) { ctx->err = "prove: the key is a shard (zkpor_pk_keep_range): use zkpor_prove_sums_dev + zkpor_prove_assemble"; return ZKPOR_E_STATE; }
    if (solver_r1cs(solver) != r1cs) { ctx->err = "prove: the solver program was created on another constraint system"; return ZKPOR_E_ARG; }
    const size_t D = (size_t)1 << pk->log2_domain;
    size_t nc = 0, nw = 0;
    int dev = -1;
    r1cs_dims(r1cs, &nc, &nw, &dev);
    if (dev != ctx->device) { ctx->err = "prove: the constraint matrices live on another GPU than the context"; return ZKPOR_E_ARG; }
    if (nw != pk->n_wires) { ctx->err = "prove: the constraint system and the key disagree on the number of wires"; return ZKPOR_E_ARG; }
    if (nc > D) { ctx->err = "prove: more constraints than the domain"; return ZKPOR_E_ARG; }
    if (n_inputs == 0 || n_inputs > nw) { ctx->err = "prove: the assignment must hold 1 + nPublic + nSecret elements"; return ZKPOR_E_ARG; }
    ZK_TRY(check_same_gpu(ctx, pk));
    ZK_TRY(check_blinding(ctx, r, s));
    ZK_TRY(stage_reserve(ctx, (3 * D + pk->n_wires) * sizeof(Fr)));
    Fr* d = (Fr*)ctx->stage;
    Fr* d_w = d + 3 * D;
    auto drain = [&] {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamSynchronize(ctx->stream);
        if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
        zk::drain_tail_streams(ctx);
    };
    GpuTurn turn;
    int32_t rc = host_upload(ctx, d_w, inputs, n_inputs * sizeof(Fr));
    if (rc == ZKPOR_OK && hipStreamSynchronize(ctx->copy_stream) != hipSuccess) { ctx->err = "prove: the upload of the inputs failed"; rc = ZKPOR_E_HIP; }
    if (rc == ZKPOR_OK) {
        turn.acquire(ctx);
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "prove: stream"; rc = ZKPOR_E_HIP; }   // the solver runs on its own context's stream
    }
    if (rc == ZKPOR_OK) {
        uint32_t paused = 0xffffffffu;
        rc = zkpor_solver_start_dev(solver, d_w, n_inputs, nullptr, &paused);
        if (rc != ZKPOR_OK) ctx->err = solver_ctx(solver)->err;
        else if (paused != 0xffffffffu) { ctx->err = "prove: the solver program holds an external hint (instruction " + std::to_string(paused) + "): drive the steps yourself"; rc = ZKPOR_E_STATE; }
    }
    if (rc == ZKPOR_OK) rc = r1cs_eval_on(ctx, r1cs, d_w, d, d + D, d + 2 * D, D);
    ProveSums m;
    Blind bl;
    const std::function<void()> prep = [&] { bl = blind_prepare(pk->delta, pk->delta2, r, s); };
    if (rc == ZKPOR_OK) rc = prove_sums(ctx, pk, d_w, d, d + D, d + 2 * D, &m, true, true, &prep, nullptr, &turn);
    if (rc != ZKPOR_OK) { drain(); return rc; }
    HostPhase hp(ctx, "host_assembly");
    assemble(pk->alpha, pk->beta, pk->beta2, m, bl, proof_out);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

// uniform Fr from the operating system's CSPRNG: 32 bytes from getrandom(2), top two bits cleared, rejected unless below the
// modulus (acceptance ~ 0.76) — the construction of gnark-crypto's fr.Element.SetRandom.  The canonical limbs are used as the
// Montgomery representation directly: x -> x R^-1 is a bijection of Fr, so the residue is uniform either way.
static int32_t fr_random_os(zkpor_ctx* ctx, uint64_t out[4]) {
    for (int tries = 0; tries < 256; ++tries) {
        size_t got = 0;
        while (got < 32) {
            ssize_t k = getrandom((char*)out + got, 32 - got, 0);
            if (k < 0) { if (errno == EINTR) continue; ctx->err = "prove: getrandom failed"; return ZKPOR_E_STATE; }
            got += (size_t)k;
        }
        out[3] &= 0x3fffffffffffffffULL;
        if (fr_canonical(out)) return ZKPOR_OK;
    }
    ctx->err = "prove: getrandom returned no canonical value";
    return ZKPOR_E_STATE;
}
int32_t zkpor_prove_tail_rand(zkpor_ctx* ctx, zkpor_pk* pk, const uint64_t* w, const uint64_t* a, const uint64_t* b,
                              const uint64_t* c, size_t n_constraints, uint64_t r_out[4], uint64_t s_out[4],
                              uint8_t proof_out[256]) try {
    if (!ctx) return ZKPOR_E_ARG;
    uint64_t r[4], s[4];
    ZK_TRY(fr_random_os(ctx, r));
    ZK_TRY(fr_random_os(ctx, s));
    int32_t rc = zkpor_prove_tail(ctx, pk, w, a, b, c, n_constraints, r, s, proof_out);
    if (r_out) memcpy(r_out, r, 32);
    if (s_out) memcpy(s_out, s, 32);
    return rc;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_commit_dev(zkpor_ctx* ctx, zkpor_pk* pk, const void* d_values, size_t n, uint8_t out_commit[64], uint8_t out_pok[64]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !pk || (n && !d_values) || !out_commit || !out_pok) return ZKPOR_E_ARG;
    if (!pk->ready) { ctx->err = "commit: key not loaded"; return ZKPOR_E_STATE; }
    if (n != pk->nC) { ctx->err = "commit: value count differs from the commitment basis"; return ZKPOR_E_ARG; }
    G1XYZZ c1 = G1XYZZ::inf(), c2 = G1XYZZ::inf();
    if (n) {
        MsmCfg cfg = msm_cfg(ctx, n, pk->tab_m);
        ZK_TRY(check_tables(ctx, pk, cfg, pk->tab_shift_c));
        size_t st = 0;
        size_t need = digits_ws_bytes(ctx, n, cfg, &st) + accumulate_ws_bytes<Fp>(cfg, n * (size_t)cfg.W);
        DigitStream ds;
        ZK_TRY(ws_reserve(ctx, need));
        ZK_TRY(msm_digits(ctx, (const Fr*)d_values, n, cfg, st, &ds));
        size_t mark = ctx->ws_off;
        ZK_TRY(ensure_pinned(ctx, 16 * MSM_SLOT_BYTES));
        char* pin = (char*)ctx->pinned;
        MsmPending p1, p2;
        ZK_TRY(msm_accumulate_launch<Fp>(ctx, ds, pk->CB, pin + 12 * MSM_SLOT_BYTES, pin + 13 * MSM_SLOT_BYTES, &p1));
        ctx->ws_off = mark;
        ZK_TRY(msm_accumulate_launch<Fp>(ctx, ds, pk->CBS, pin + 14 * MSM_SLOT_BYTES, pin + 15 * MSM_SLOT_BYTES, &p2));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        msm_accumulate_finish<Fp>(p1, &c1);
        msm_accumulate_finish<Fp>(p2, &c2);
    }
    G1Affine a1 = xyzz_to_affine<Fp>(c1), a2 = xyzz_to_affine<Fp>(c2);
    memcpy(out_commit, &a1, 64); memcpy(out_pok, &a2, 64);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
int32_t zkpor_commit(zkpor_ctx* ctx, zkpor_pk* pk, const uint64_t* values, size_t n, uint8_t out_commit[64], uint8_t out_pok[64]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !pk || (n && !values) || !out_commit || !out_pok) return ZKPOR_E_ARG;
    if (!pk->ready) { ctx->err = "commit: key not loaded"; return ZKPOR_E_STATE; }
    // the committed values go behind the prove tail's vectors in the staging area, so a commit between two proofs (the BSB22
    // hint runs inside the solver) never forces the area to be re-laid out
    size_t D = (size_t)1 << pk->log2_domain;
    size_t off = (3 * D + pk->n_wires) * sizeof(Fr);
    ZK_TRY(stage_reserve(ctx, off + (n ? n : 1) * sizeof(Fr)));
    Fr* d = (Fr*)(ctx->stage + off);
    if (n) {
        ZK_TRY(host_upload(ctx, d, values, n * sizeof(Fr)));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    }
    // no turn on the device for these two short sums (common.cuh GpuTurn): they run next to whatever proof is on the GPU — waiting
    // for it would keep this caller from moving its proof's vectors across PCIe in the meantime
    return zkpor_commit_dev(ctx, pk, d, n, out_commit, out_pok);
} ZK_ABI_CATCH_IN(ctx)

static void fp_be(const Fp& x, uint8_t* out) {
    Fp c = Fp::from_mont(x);
    for (int i = 0; i < 8; ++i)
        for (int j = 0; j < 4; ++j) out[31 - (i * 4 + j)] = (uint8_t)(c.v[i] >> (8 * j));
}
// gnark-crypto RawBytes: an uncompressed point is X | Y big-endian with the two top bits of byte 0 clear (mUncompressed = 0b00);
// the point at infinity is the flag mUncompressedInfinity = 0b01 << 6 = 0x40 in byte 0 and zeros after it (marshal.go) — NOT 64 zero
// bytes, which a strict decoder would read as the off-curve point (0, 0)
static void g1_raw(const Fp* xy, uint8_t* out) {
    if (xy[0].is_zero() && xy[1].is_zero()) { memset(out, 0, 64); out[0] = 0x40; return; }
    fp_be(xy[0], out); fp_be(xy[1], out + 32);
}
static void g2_raw(const Fp* p, uint8_t* out) {  // p = X.A0, X.A1, Y.A0, Y.A1 -> X.A1 | X.A0 | Y.A1 | Y.A0
    if (p[0].is_zero() && p[1].is_zero() && p[2].is_zero() && p[3].is_zero()) { memset(out, 0, 128); out[0] = 0x40; return; }
    fp_be(p[1], out); fp_be(p[0], out + 32); fp_be(p[3], out + 64); fp_be(p[2], out + 96);
}
// G1Affine.Marshal() of one point handed out by this library (Montgomery limbs): X | Y big-endian, the identity as 0x40 | zeros.
// What gnark hashes into the BSB22 challenge (constraint.SerializeCommitment).  Host arithmetic only.
int32_t zkpor_g1_marshal(const uint8_t affine[64], uint8_t out[64]) try {
    if (!affine || !out) return ZKPOR_E_ARG;
    Fp xy[2];
    memcpy(xy, affine, 64);
    g1_raw(xy, out);
    return ZKPOR_OK;
} ZK_ABI_CATCH
int32_t zkpor_proof_write_raw(const uint8_t proof[256], const uint8_t* commitments, uint32_t n_commitments,
                              const uint8_t pok[64], uint8_t* out, size_t out_cap, size_t* out_len) try {
    if (!proof || !out || !out_len || (n_commitments && (!commitments || !pok))) return ZKPOR_E_ARG;
    size_t need = 256 + 4 + (size_t)n_commitments * 64 + 64;
    if (out_cap < need) return ZKPOR_E_ARG;
    const Fp* f = (const Fp*)proof;
    g1_raw(f, out);                                                   // Ar
    g2_raw(f + 2, out + 64);                                          // Bs
    g1_raw(f + 6, out + 192);                                         // Krs
    out[256] = (uint8_t)(n_commitments >> 24); out[257] = (uint8_t)(n_commitments >> 16);
    out[258] = (uint8_t)(n_commitments >> 8); out[259] = (uint8_t)n_commitments;
    size_t off = 260;
    for (uint32_t i = 0; i < n_commitments; ++i) {
        g1_raw((const Fp*)(commitments + 64 * (size_t)i), out + off);
        off += 64;
    }
    if (pok) g1_raw((const Fp*)pok, out + off);
    else { memset(out + off, 0, 64); out[off] = 0x40; }               // no commitment: the knowledge proof is the identity
    off += 64;
    *out_len = off;
    return ZKPOR_OK;
} ZK_ABI_CATCH

}  // extern "C"
