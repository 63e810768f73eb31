// Pippenger multi-scalar multiplication for gfx950 — replaces gnark-crypto's G1Jac.MultiExp / G2Jac.MultiExp that
// groth16.Prove runs on the host (reference call site src/prover/prover/prover.go:269).
//
// MI355X-first structure ("sort once, accumulate many"):
//   1. sort.hip      scalars (Montgomery Fr) -> signed c-bit digits -> (key = window|bucket, val = point index|sign)
//                    pairs; zero digits are dropped (witness vectors are full of them).
//   2. sort.hip: the pairs grouped by key, most significant digit first (own kernels; step 1 is fused into its first level): every bucket
//                    becomes a contiguous run.
//   3. k_acc_level1  perfectly load-balanced segmented sum: each thread owns exactly L consecutive sorted
//                    entries (not a bucket), gathers the 64/128-byte affine points and accumulates runs in XYZZ
//                    registers.  Runs that lie inside the chunk are finished buckets and go straight to HBM;
//                    the (at most two) runs cut by the chunk boundary go to a partials array.
//      k_acc_levelN  the same algorithm applied recursively to the partials (XYZZ+XYZZ) until one chunk
//                    remains.  Any bucket skew (e.g. 25% of witness scalars being 1) costs log_L extra
//                    tiny launches instead of serialising a thread.
//   4. k_reduce_groups  running-sum bucket reduction in three group levels (sum and weighted sum per group).
//   5. host: per-window recombination + Horner over windows (a few hundred group operations).
// Steps 1-2 depend only on the scalars: groth16's A, B1, B2 and K multi-exponentiations all use the witness, so
// the prover runs 1-2 once and 3-5 four times against wire-indexed key arrays (DigitStream below).
#pragma once
#include "common.cuh"
#include "fe29.cuh"
#include <type_traits>

namespace zk {

static constexpr u32 NOKEY = 0xffffffffu;

struct MsmCfg {
    int c;     // window bits
    int W;     // signed digits per scalar = ceil(255 / c)
    int m;     // fixed-base tables per point (1 = none): table q holds 2^(q * piece * c) P, so digit t of a scalar is an entry of
    int piece; //   bucket window t % piece against point i * m + t / piece; piece = ceil(W / m) bucket windows remain
    u32 bpw;   // buckets per window = 2^(c-1)
    u32 NB;    // total buckets = bpw * piece
    int L;     // entries per accumulation thread
    int key_bits;
    u32 g1, n1, g2, n2;  // reduce group sizes: g1*n1 = bpw, g2*n2 = n1
};

inline int log2_ceil(size_t n) {
    int k = 0;
    while (((size_t)1 << k) < n) ++k;
    return k;
}
// Fixed-base tables (the key arrays never change): with m tables per point the W digits of a scalar share ceil(W / m) bucket
// windows, so the window can grow — 22 bits and 12 digits instead of 20 bits and 13 at 2^26 — at the SAME number of buckets (the
// bucket reduction does not grow), for m times the key memory: HBM capacity bought back as 8 % fewer bucket additions.
inline MsmCfg msm_cfg(zkpor_ctx* ctx, size_t n, int tables = 1) {
    MsmCfg m;
    int c = ctx->msm_window;
    if (c <= 0) {
        c = log2_ceil(n) - (tables > 1 ? 4 : 6);
        if (c < 4) c = 4;
        if (c > (tables > 1 ? 22 : 20)) c = tables > 1 ? 22 : 20;
    }
    if (c < 2) c = 2;
    if (c > 24) c = 24;   // "msm_window" by hand up to 24 (11 digits, 2^23 buckets per window: measured against the reduction's cost in round 5); automatic stays <= 22
    m.c = c;
    m.W = (255 + c - 1) / c;
    m.m = tables < 1 ? 1 : tables;
    m.piece = (m.W + m.m - 1) / m.m;
    m.bpw = 1u << (c - 1);
    m.NB = m.bpw * (u32)m.piece;
    m.L = ctx->msm_chunk >= 4 ? ctx->msm_chunk : (n >= ((size_t)1 << 22) ? 64 : 32);
    m.key_bits = log2_ceil(m.NB);
    if (m.key_bits < 1) m.key_bits = 1;
    m.g1 = m.bpw < 128 ? m.bpw : 128;
    m.n1 = m.bpw / m.g1;
    m.g2 = m.n1 < 64 ? m.n1 : 64;
    m.n2 = m.n1 / m.g2;
    return m;
}

// ------------------------------------------------------------------------------------------------ launchers
// defined next to the kernel instantiations (one translation unit per field / inlining policy)
// sort.hip: the digit stream of a scalar vector, grouped by bucket (keys ascending).  The plan = the levels of the most-significant-digit-first
// sort for a key space and the layout of its counter arrays in the scratch area.
struct DigitSortPlan {
    int nlev = 1;
    int r[4] = {0, 0, 0, 0}, shift[4] = {0, 0, 0, 0};   // level l splits on (key >> shift[l]) & (2^r[l] - 1)
    u32 n_child[4] = {0, 0, 0, 0};                      // key prefixes at level l: ((NB - 1) >> shift[l]) + 1
    size_t off_C[4] = {0, 0, 0, 0}, off_bs = 0, bytes = 0;
};
DigitSortPlan digit_sort_plan(const MsmCfg& cfg);
int32_t digit_sort(zkpor_ctx* ctx, const Fr* d_scalars, u32 n, const MsmCfg& cfg, const DigitSortPlan& plan, u32* kA, u32* vA, u32* kB, u32* vB, u32* counter,
                   char* temp, const u32* absent0, const u32* absent1, u32 Ms[3], u32** k_out, u32** v_out);
int32_t launch_filter(zkpor_ctx* ctx, const u32* keys, const u32* vals, u32 M, u32* seg_counts, u32 grid, u32* k0, u32* v0, u32* k1, u32* v1);
// "debug_validate": checks a sorted stream on ctx->stream and WAITS for the verdict (msm_digits.hip)
int32_t validate_stream(zkpor_ctx* ctx, const u32* keys, const u32* vals, u32 M, u32 NB, u32 n_idx, const char* what);
constexpr u32 FILTER_MAX_GRID = 2048;
// a digit-stream value = (point index << 1 | sign) in bits 0..29; bit 30 / 31 = the point is absent from array group 0 / 1
constexpr u32 VAL_MASK = 0x3fffffffu, VAL_ABSENT0 = 1u << 30, VAL_ABSENT1 = 1u << 31;
int32_t launch_level1(zkpor_ctx* ctx, const u32* keys, const u32* vals, const Affine<Fp>* pts, u32 M, int L, u32 NB,
                      XYZZ<Fp>* buckets, u32* out_keys, XYZZ<Fp>* out_part);
int32_t launch_level1(zkpor_ctx* ctx, const u32* keys, const u32* vals, const Affine<Fp2>* pts, u32 M, int L, u32 NB,
                      XYZZ<Fp2>* buckets, u32* out_keys, XYZZ<Fp2>* out_part);
// the 29-bit pipeline: buckets, partials and running sums are raw XYZZ29 register images (36 words per Fp component);
// level 1 clears the buckets itself
int32_t launch_level1_29(zkpor_ctx* ctx, const u32* keys, const u32* vals, const Affine<Fp>* pts, u32 M, int L, u32 NB,
                         u32* braw, u32* out_keys, u32* praw);
int32_t launch_level1_29(zkpor_ctx* ctx, const u32* keys, const u32* vals, const Affine<Fp2>* pts, u32 M, int L, u32 NB,
                         u32* braw, u32* out_keys, u32* praw);
template <class F>
int32_t launch_levelN29(zkpor_ctx* ctx, const u32* keys, const u32* src, u32 M, int L, u32* braw, u32* out_keys, u32* out_part);
template <class F>
int32_t launch_reduce29(zkpor_ctx* ctx, const u32* Sin, const u32* Yin, u32 n_groups, u32 g, int dbl, u32* Sout, u32* Yout);
template <> int32_t launch_levelN29<Fp>(zkpor_ctx*, const u32*, const u32*, u32, int, u32*, u32*, u32*);
template <> int32_t launch_levelN29<Fp2>(zkpor_ctx*, const u32*, const u32*, u32, int, u32*, u32*, u32*);
template <> int32_t launch_reduce29<Fp>(zkpor_ctx*, const u32*, const u32*, u32, u32, int, u32*, u32*);
template <> int32_t launch_reduce29<Fp2>(zkpor_ctx*, const u32*, const u32*, u32, u32, int, u32*, u32*);
template <class F>
constexpr size_t raw_words() { return sizeof(XYZZ<F>) / 128 * 36; }  // 36 (G1) or 72 (G2) u32 per accumulator image
int32_t launch_levelN(zkpor_ctx* ctx, const u32* keys, const XYZZ<Fp>* src, u32 M, int L, XYZZ<Fp>* buckets,
                      u32* out_keys, XYZZ<Fp>* out_part);
int32_t launch_levelN(zkpor_ctx* ctx, const u32* keys, const XYZZ<Fp2>* src, u32 M, int L, XYZZ<Fp2>* buckets,
                      u32* out_keys, XYZZ<Fp2>* out_part);
int32_t launch_reduce(zkpor_ctx* ctx, const XYZZ<Fp>* Sin, const XYZZ<Fp>* Yin, u32 n_groups, u32 g, int dbl,
                      XYZZ<Fp>* Sout, XYZZ<Fp>* Yout);
int32_t launch_reduce(zkpor_ctx* ctx, const XYZZ<Fp2>* Sin, const XYZZ<Fp2>* Yin, u32 n_groups, u32 g, int dbl,
                      XYZZ<Fp2>* Sout, XYZZ<Fp2>* Yout);

// ------------------------------------------------------------------------------------------------ host driver
struct DigitStream {  // sorted digits of one scalar vector, reusable across point arrays
    MsmCfg cfg;
    u32* keys = nullptr;
    u32* vals = nullptr;
    u32 M = 0;
    u32 n_idx = 0;   // point indices the values may hold: scalars x tables (checked by "debug_validate")
};

// per-array streams (msm_digits.hip k_filter_write): up to two groups of arrays whose absent points are dropped from the shared stream
struct StreamFilter {
    const u32* absent[2] = {nullptr, nullptr};   // device bitmaps over the scalars' indices; nullptr = group not filtered
};

inline size_t digits_ws_bytes(zkpor_ctx* ctx, size_t n, const MsmCfg& cfg, size_t* sort_temp, int n_filters = 0) {
    size_t cap = n * (size_t)cfg.W;
    WsPlan p;
    p.add<u32>(cap); p.add<u32>(cap); p.add<u32>(cap); p.add<u32>(cap); p.add<u32>(64);
    (void)ctx;
    size_t tb = digit_sort_plan(cfg).bytes;
    if (n_filters > 0 && tb < 2 * FILTER_MAX_GRID * sizeof(u32)) tb = 2 * FILTER_MAX_GRID * sizeof(u32);   // the filter's segment counts reuse the sort's scratch
    *sort_temp = tb;
    p.add<char>(tb + 256);
    // the first filtered stream lands in the sort's spare buffer pair, every further one needs its own
    for (int f = 1; f < n_filters; ++f) { p.add<u32>(cap); p.add<u32>(cap); }
    return p.total;
}

// decompose + sort.  Workspace must already be reserved; allocates from it.
inline int32_t msm_digits(zkpor_ctx* ctx, const Fr* d_scalars, size_t n, const MsmCfg& cfg, size_t sort_temp,
                          DigitStream* out, const StreamFilter* filt = nullptr, DigitStream* out_f0 = nullptr, DigitStream* out_f1 = nullptr,
                          hipEvent_t ev_sorted = nullptr, hipEvent_t ev_f0 = nullptr) {
    size_t cap = n * (size_t)cfg.W;
    if (cap >= 0xfffffff0ull || n * (size_t)cfg.m >= (1ull << 31)) { ctx->err = "msm: too many digit entries for 32-bit indexing"; return ZKPOR_E_ARG; }
    u32* k0 = ws_alloc<u32>(ctx, cap); u32* k1 = ws_alloc<u32>(ctx, cap);
    u32* v0 = ws_alloc<u32>(ctx, cap); u32* v1 = ws_alloc<u32>(ctx, cap);
    u32* counter = ws_alloc<u32>(ctx, 64);
    char* temp = ws_alloc<char>(ctx, sort_temp + 256);
    if (!k0 || !k1 || !v0 || !v1 || !counter || !temp) { ctx->err = "msm: workspace too small"; return ZKPOR_E_OOM; }
    out->cfg = cfg;
    out->n_idx = (u32)(n * (size_t)cfg.m);
    const bool flags_fit = n * (size_t)cfg.m < (1ull << 29);   // the absence flags live in bits 30 / 31 of a value
    u32 Ms[3] = {0, 0, 0};
    // decompose + group by bucket (sort.hip; phases "msm_decompose" = the counting pass over the scalars, "msm_sort" = everything after it)
    ZK_TRY(digit_sort(ctx, d_scalars, (u32)n, cfg, digit_sort_plan(cfg), k0, v0, k1, v1, counter, temp, (filt && flags_fit) ? filt->absent[0] : nullptr,
                      (filt && flags_fit) ? filt->absent[1] : nullptr, Ms, &out->keys, &out->vals));
    const u32 M = Ms[0];
    out->M = M;
    if (ev_sorted) ZK_HIP(ctx, hipEventRecord(ev_sorted, ctx->stream));
    // per-array streams: ONE stable filter pass pair over the sorted stream produces both groups; their sizes were counted by the decomposition
    if (out_f0) *out_f0 = *out;                            // not filtered: the shared stream itself
    if (out_f1) *out_f1 = *out;
    const bool g0 = flags_fit && filt && filt->absent[0] && out_f0 && M, g1 = flags_fit && filt && filt->absent[1] && out_f1 && M;
    if (g0 || g1) {
        u32 *fk[2] = {nullptr, nullptr}, *fv[2] = {nullptr, nullptr};
        bool spare_used = false;
        for (int f = 0; f < 2; ++f) {
            if (!(f == 0 ? g0 : g1)) continue;
            if (!spare_used) { fk[f] = (out->keys == k0) ? k1 : k0; fv[f] = (out->vals == v0) ? v1 : v0; spare_used = true; }   // the sort's spare pair
            else { fk[f] = ws_alloc<u32>(ctx, cap); fv[f] = ws_alloc<u32>(ctx, cap); if (!fk[f] || !fv[f]) { ctx->err = "msm: workspace too small"; return ZKPOR_E_OOM; } }
        }
        PhaseScope ps(ctx, "msm_filter");
        int grid = ctx->msm_filter_grid > 0 ? ctx->msm_filter_grid : 256;
        if (grid > (int)FILTER_MAX_GRID) grid = FILTER_MAX_GRID;
        ZK_TRY(launch_filter(ctx, out->keys, out->vals, M, (u32*)temp, (u32)grid, fk[0], fv[0], fk[1], fv[1]));
        if (g0) { out_f0->keys = fk[0]; out_f0->vals = fv[0]; out_f0->M = Ms[1]; }
        if (g1) { out_f1->keys = fk[1]; out_f1->vals = fv[1]; out_f1->M = Ms[2]; }
    }
    if (ev_f0) ZK_HIP(ctx, hipEventRecord(ev_f0, ctx->stream));
    return ZKPOR_OK;
}

// threads of the second partial-sum level for T1 level-1 threads (its 2 T2 outputs size the second partials buffer): chunks of L, or the
// short chunks of a small level (msm_tail_chunk >= 4, levels below 2^21 entries); monotonic in T1, so a bound for every smaller stream
inline size_t level2_threads(size_t T1, int L) {
    size_t by_l = (2 * T1 + L - 1) / L;
    size_t small = 2 * T1 < ((size_t)1 << 21) ? 2 * T1 : ((size_t)1 << 21);
    size_t by_tail = (small + 3) / 4;
    return by_l > by_tail ? by_l : by_tail;
}

template <class F>
inline size_t accumulate_ws_bytes(const MsmCfg& cfg, size_t max_entries) {
    WsPlan p;
    size_t T1 = (max_entries + cfg.L - 1) / cfg.L;
    size_t T2 = level2_threads(T1, cfg.L);
    const size_t img = raw_words<F>() * 4;  // the raw 29-bit image is the larger of the two element formats
    p.add<char>(cfg.NB * img);
    p.add<char>((2 * T1 + 2) * img); p.add<u32>(2 * T1 + 2);
    p.add<char>((2 * T2 + 2) * img); p.add<u32>(2 * T2 + 2);
    size_t half = (size_t)cfg.NB / 2 + 1;
    p.add<char>(half * img); p.add<char>(half * img);                  // S, Y of odd levels
    p.add<char>((half / 2 + 1) * img); p.add<char>((half / 2 + 1) * img);  // S, Y of even levels
    return p.total;
}

struct MsmPending {  // a multi-exponentiation whose kernels are queued; its per-window finals land in pinned host memory
    MsmCfg cfg;
    u64 Kmul = 0;
    bool empty = true;
    bool raw29 = false;  // finals are raw 29-bit images (converted on the host) rather than XYZZ<F>
    bool chained = false;  // the part after level 1 went to the chain stream: its events WERE recorded (an empty sum records nothing)
    void* hS = nullptr;  // W finals, pinned
    void* hY = nullptr;
};

// Round 6 ("msm_chain"): everything of a multi-exponentiation AFTER its level-1 kernel — the partial-sum levels, the bucket reduction, the copies of the
// finals: ~40 short dependent launches, 5-6 ms during which the device is mostly idle — may run on a second stream, beside the NEXT sum's level-1
// kernel on ctx->stream.  ev_level1 is recorded on ctx->stream behind the level-1 kernel (the chain waits for it), ev_done on the chain stream behind the
// last copy: whoever reuses this sum's workspace region, and the final host wait, must wait for ev_done.
struct MsmChain { hipStream_t stream = nullptr; hipEvent_t ev_level1 = nullptr, ev_done = nullptr; };

// queue bucket accumulation + reduction for one point array on ctx->stream; NO host synchronisation: the W per-window
// (S, Y) pairs are copied to the pinned slots in stream order, so the next accumulation may reuse the workspace.
template <class F>
inline int32_t msm_accumulate_launch(zkpor_ctx* ctx, const DigitStream& ds, const Affine<F>* d_pts, void* pinned_S,
                                     void* pinned_Y, MsmPending* out, const MsmChain* chain = nullptr) {
    const MsmCfg& cfg = ds.cfg;
    out->cfg = cfg; out->hS = pinned_S; out->hY = pinned_Y; out->Kmul = 0; out->chained = false;
    out->empty = ds.M == 0;
    if (ds.M == 0) return ZKPOR_OK;
    const u32 M = ds.M;
    const int L = cfg.L;
    size_t T1 = ((size_t)M + L - 1) / L;
    size_t T2 = level2_threads(T1, L);
    const bool raw = std::is_same<F, Fp>::value ? ctx->g1_variant == 1 : ctx->g2_variant == 1;
    const size_t img = raw_words<F>() * 4;
    char* buckets = ws_alloc<char>(ctx, cfg.NB * img);
    char* pa = ws_alloc<char>(ctx, (2 * T1 + 2) * img); u32* ka = ws_alloc<u32>(ctx, 2 * T1 + 2);
    char* pb = ws_alloc<char>(ctx, (2 * T2 + 2) * img); u32* kb = ws_alloc<u32>(ctx, 2 * T2 + 2);
    size_t half = (size_t)cfg.NB / 2 + 1;
    char* Sa = ws_alloc<char>(ctx, half * img); char* Ya = ws_alloc<char>(ctx, half * img);
    char* Sb = ws_alloc<char>(ctx, (half / 2 + 1) * img); char* Yb = ws_alloc<char>(ctx, (half / 2 + 1) * img);
    if (!buckets || !pa || !ka || !pb || !kb || !Sa || !Ya || !Sb || !Yb) {
        ctx->err = "msm: workspace too small"; return ZKPOR_E_OOM;
    }
    out->raw29 = raw;
    if (ctx->debug_validate) ZK_TRY(validate_stream(ctx, ds.keys, ds.vals, M, (u32)cfg.NB, ds.n_idx, std::is_same<F, Fp>::value ? "G1" : "G2"));
    {
        PhaseScope ps(ctx, "msm_accumulate");
        if (raw) ZK_TRY(launch_level1_29(ctx, ds.keys, ds.vals, d_pts, M, L, (u32)cfg.NB, (u32*)buckets, ka, (u32*)pa));
        else ZK_TRY(launch_level1(ctx, ds.keys, ds.vals, d_pts, M, L, (u32)cfg.NB, (XYZZ<F>*)buckets, ka, (XYZZ<F>*)pa));
    }
    // the rest on the chain stream (when there is one); the context's stream is restored on every way out
    struct StreamBack { zkpor_ctx* c; hipStream_t s; ~StreamBack() { c->stream = s; } } stream_back{ctx, ctx->stream};
    const bool chained = chain && chain->stream && chain->stream != ctx->stream;
    if (chained) {
        ZK_HIP(ctx, hipEventRecord(chain->ev_level1, ctx->stream));
        ZK_HIP(ctx, hipStreamWaitEvent(chain->stream, chain->ev_level1, 0));
        ctx->stream = chain->stream;
    }
    {
        PhaseScope ps(ctx, "msm_accumulate");
        size_t T = T1;
        char* src = pa; u32* srck = ka; char* dst = pb; u32* dstk = kb;
        while (T > 1) {
            u32 Mn = (u32)(2 * T);
            // a small level is as long as one thread's serial chain of additions: short chunks there (chunks of >= 4: every level turns T threads into at most T / 2 + 1, down to one)
            const int Ln = (ctx->msm_tail_chunk && Mn < (1u << 21) && ctx->msm_tail_chunk < L) ? ctx->msm_tail_chunk : L;
            size_t Tn = ((size_t)Mn + Ln - 1) / Ln;
            if (raw) ZK_TRY(launch_levelN29<F>(ctx, srck, (const u32*)src, Mn, Ln, (u32*)buckets, dstk, (u32*)dst));
            else ZK_TRY(launch_levelN(ctx, srck, (const XYZZ<F>*)src, Mn, Ln, (XYZZ<F>*)buckets, dstk, (XYZZ<F>*)dst));
            std::swap(src, dst); std::swap(srck, dstk);
            T = Tn;
        }
    }
    const char* fin_S = nullptr;
    const char* fin_Y = nullptr;
    u64 Kmul = 0;  // g_1 + g_1 g_2 + ... + g_1..g_{L-1}
    {
        PhaseScope ps(ctx, "msm_reduce");
        int rem = cfg.c - 1, done_bits = 0, level = 0;
        u32 count = cfg.NB;  // elements at the current level (all windows)
        const char* Sin = buckets;
        const char* Yin = nullptr;
        while (rem > 0) {
            int gl = rem < 4 ? rem : 4;
            u32 ng = count >> gl;
            char* So = (level & 1) ? Sb : Sa;
            char* Yo = (level & 1) ? Yb : Ya;
            if (raw) ZK_TRY(launch_reduce29<F>(ctx, (const u32*)Sin, (const u32*)Yin, ng, 1u << gl, done_bits, (u32*)So, (u32*)Yo));
            else ZK_TRY(launch_reduce(ctx, (const XYZZ<F>*)Sin, (const XYZZ<F>*)Yin, ng, 1u << gl, done_bits, (XYZZ<F>*)So, (XYZZ<F>*)Yo));
            done_bits += gl; rem -= gl; count = ng; Sin = So; Yin = Yo; ++level;
            if (rem > 0) Kmul += (u64)1 << done_bits;
        }
        fin_S = Sin; fin_Y = Yin;
    }
    const size_t Wn = (size_t)cfg.piece;
    const size_t fin_bytes = Wn * (raw ? img : sizeof(XYZZ<F>));
    out->Kmul = Kmul;
    ZK_HIP(ctx, hipMemcpyAsync(pinned_S, fin_S, fin_bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipMemcpyAsync(pinned_Y, fin_Y, fin_bytes, hipMemcpyDeviceToHost, ctx->stream));
    if (chained) { ZK_HIP(ctx, hipEventRecord(chain->ev_done, chain->stream)); out->chained = true; }
    return ZKPOR_OK;
}

// raw 29-bit image -> XYZZ<F> on the host (portable path of fe29.cuh)
inline Fp raw29_fp_host(const u32* w) {
    Fp29 v;
    for (int i = 0; i < 9; ++i) v.l[i] = w[i];
    return Fp29::to32_div32(v);
}
inline XYZZ<Fp> raw29_to_xyzz_host(const u32* img, const Fp*) {
    bool inf = true;
    for (int i = 0; i < 9; ++i) inf &= img[18 + i] == 0;
    if (inf) return XYZZ<Fp>::inf();
    return {raw29_fp_host(img), raw29_fp_host(img + 9), raw29_fp_host(img + 18), raw29_fp_host(img + 27)};
}
inline XYZZ<Fp2> raw29_to_xyzz_host(const u32* img, const Fp2*) {  // [component 0: x y zz zzz][component 1: x y zz zzz]
    bool inf = true;
    for (int i = 0; i < 9; ++i) inf &= img[18 + i] == 0 && img[36 + 18 + i] == 0;
    if (inf) return XYZZ<Fp2>::inf();
    XYZZ<Fp2> r;
    r.x = {raw29_fp_host(img), raw29_fp_host(img + 36)};
    r.y = {raw29_fp_host(img + 9), raw29_fp_host(img + 36 + 9)};
    r.zz = {raw29_fp_host(img + 18), raw29_fp_host(img + 36 + 18)};
    r.zzz = {raw29_fp_host(img + 27), raw29_fp_host(img + 36 + 27)};
    return r;
}

// after the stream has been synchronised: window sum = Y_L - Kmul * T, then Horner with 2^c over the windows (host)
template <class F>
inline void msm_accumulate_finish(const MsmPending& p, XYZZ<F>* result) {
    XYZZ<F> acc = XYZZ<F>::inf();
    if (!p.empty) {
        const XYZZ<F>* hS = (const XYZZ<F>*)p.hS;
        const XYZZ<F>* hY = (const XYZZ<F>*)p.hY;
        const size_t rw = raw_words<F>();
        for (int w = p.cfg.piece - 1; w >= 0; --w) {
            for (int k = 0; k < p.cfg.c; ++k) acc = xyzz_dbl<F>(acc);
            XYZZ<F> win = p.raw29 ? raw29_to_xyzz_host((const u32*)p.hY + (size_t)w * rw, (const F*)nullptr) : hY[w];
            XYZZ<F> sw = p.raw29 ? raw29_to_xyzz_host((const u32*)p.hS + (size_t)w * rw, (const F*)nullptr) : hS[w];
            if (p.Kmul) xyzz_add<F>(win, xyzz_neg<F>(xyzz_mul_u64<F>(sw, p.Kmul)));
            xyzz_add<F>(acc, win);
        }
    }
    *result = acc;
}

static constexpr size_t MSM_SLOT_BYTES = 128 * 288;  // one pinned (S or Y) slot: up to 128 windows (c = 2) of G2 raw images (288 B)

int32_t ensure_pinned(zkpor_ctx* ctx, size_t bytes);

// synchronous form: queue, wait, finish
template <class F>
inline int32_t msm_accumulate(zkpor_ctx* ctx, const DigitStream& ds, const Affine<F>* d_pts, XYZZ<F>* result) {
    ZK_TRY(ensure_pinned(ctx, 16 * MSM_SLOT_BYTES));
    MsmPending p;
    ZK_TRY(msm_accumulate_launch<F>(ctx, ds, d_pts, ctx->pinned, (char*)ctx->pinned + MSM_SLOT_BYTES, &p));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    msm_accumulate_finish<F>(p, result);
    return ZKPOR_OK;
}

// whole MSM for device-resident points and scalars
template <class F>
inline int32_t msm_dev(zkpor_ctx* ctx, const Affine<F>* d_pts, const Fr* d_scalars, size_t n, XYZZ<F>* result) {
    if (n == 0) { *result = XYZZ<F>::inf(); return ZKPOR_OK; }
    // the digit stream is indexed with 32 bits: above 2^27 points (2^28-constraint keys) run slices and add the results
    const size_t SLICE = (size_t)1 << 27;
    if (n > SLICE) {
        XYZZ<F> acc = XYZZ<F>::inf();
        for (size_t off = 0; off < n; off += SLICE) {
            XYZZ<F> part;
            size_t m = n - off < SLICE ? n - off : SLICE;
            ZK_TRY(msm_dev<F>(ctx, d_pts + off, d_scalars + off, m, &part));
            xyzz_add<F>(acc, part);
        }
        *result = acc;
        return ZKPOR_OK;
    }
    MsmCfg cfg = msm_cfg(ctx, n);
    size_t sort_temp = 0;
    size_t need = digits_ws_bytes(ctx, n, cfg, &sort_temp) + accumulate_ws_bytes<F>(cfg, n * (size_t)cfg.W);
    ZK_TRY(ws_reserve(ctx, need));
    DigitStream ds;
    ZK_TRY(msm_digits(ctx, d_scalars, n, cfg, sort_temp, &ds));
    return msm_accumulate<F>(ctx, ds, d_pts, result);
}

}  // namespace zk
