// Context, workspace arena, error plumbing and phase timers shared by the gfx950 kernels' host drivers.
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <chrono>
#include <vector>
#include <map>
#include <stdio.h>
#include <new>
#include <exception>
#include <string.h>
#include <cstring>
#include "../../include/zkpor.h"
#include "fe.cuh"
#include "ec.cuh"

struct PhaseTimer {
    double ms = 0;
    uint64_t calls = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct zkpor_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool stream_pooled = false;      // "stream_own_queue": the own stream has a hardware queue of its own and goes back to the process-wide pool, not to hipStreamDestroy
    hipStream_t aux_stream = nullptr;  // digit streams (decompose + sort) of the prove tail run here, beside the ALU-bound kernels
    std::string err;
    // bump-allocated workspace, regrown on demand
    char* ws = nullptr;
    size_t ws_cap = 0, ws_off = 0;
    // tuning
    int msm_window = 0;  // 0 = auto
    int msm_tables = 1;  // fixed-base tables per key point, applied to keys loaded AFTER the parameter is set (msm.cuh MsmCfg)
    int msm_chunk = 0;               // sorted entries per level-1 thread: 0 = 64 from 2^22 scalars up (fewer partial sums: -5 ms per 2^26 proof), else 32
    int g1_variant = 1;  // level-1 G1 accumulation arithmetic: 0 = 8 x 32-bit limbs, 1 = 9 x 29-bit limbs (fe29.cuh)
    int g2_variant = 1;  // same for the G2 lane-pair kernel
    int ntt_tile_log = 10;  // log2 of the elements per LDS tile of k_ntt_pass29 (36 B each)
    int ntt_variant = 1; // NTT pass arithmetic: 0 = 8 x 32-bit limbs, 1 = 9 x 29-bit lazy limbs (k_ntt_pass29)
    int pos_out = 1, pos_carry = 0;
    // Poseidon parameter tables on device (built lazily)
    void* pos_tables = nullptr;
    // NTT domains by log2 size (ntt.cuh NttDomain*)
    std::map<int, void*> ntt_domains;
    // timers
    std::map<std::string, PhaseTimer> phases;
    std::map<std::string, uint64_t> stats;   // counters of the last call, by name (zkpor_stat): digit-stream sizes of the last prove tail
    std::vector<hipEvent_t> event_pool;
    // small pinned host staging buffer
    void* pinned = nullptr;
    size_t pinned_cap = 0;
    // host-pointer boundary (zkpor_prove_tail, zkpor_commit): persistent device staging for the caller's vectors, a copy
    // stream, and pinned bounce buffers that carry pageable host memory across PCIe (api_core.hip host_upload)
    char* stage = nullptr;
    size_t stage_cap = 0;
    hipStream_t copy_stream = nullptr;
    void* bounce = nullptr;          // zk::Bounce*
    int copy_threads = 4;            // n > 0 = n host threads fill pinned bounce buffers (30 GB/s; the default since round 5); 0 = the HIP runtime page-locks pageable ranges on the fly (56 GB/s measured; see host_upload)
    int copy_chunk_mb = 32;          // size of one of the four pinned bounce buffers
    int host_order = 0;              // zkpor_prove_tail from host memory: 0 = w first, a/b/c under the witness sums; 1 = everything first, then the resident order
    int msm_reduce_scan = 1;         // small bucket-reduction levels: one lane (G2: lane pair) per bucket, scan + tree sums; 2 = G1 only (round 2), 0 = serial walk
    int msm_tail_chunk = 8;          // entries per thread of the SMALL partial-sum levels (< 2^21 entries): their duration is the serial chain, not the work; 0 = msm_chunk
    int msm_chain = 2;               // the prove tail's sums: everything after a sum's level-1 kernel (partial-sum levels, bucket reduction, copies) on a second stream with a hardware queue of its own, beside the next sum's level-1 kernel (msm.cuh MsmChain) — 2 (the default): every tail, 1: only tails on "tail_streams" / "tail_reserve_cus" streams, 0 = one stream
    int msm_filter = 1;              // per-array digit streams: drop the entries of absent points before B1 / B2 and K (msm_digits.hip)
    int msm_filter_grid = 0;         // workgroups of the filter kernels (0 = 256: one per CU — bandwidth, not wave slots)
    int ntt_twiddles = 0;            // inter-pass twiddles of the fields whose table exceeds the L2 (2 GiB per direction at 2^26): 0 = read from the table, 1 = generated from two half tables (one more product per element, 15 GB less traffic per computeH)
    int ntt_fuse = 1;                // computeH: the two passes over the lowest field (inverse DIF last, coset DIT first) in one kernel (ntt.hip k_ntt_mid29)
    int ntt_h = 1;                   // computeH: 1 = six transforms (c's coefficients are subtracted behind h's inverse coset transform instead of c going to the coset: the transform is linear), 0 = gnark's seven; the same h bit for bit
    int sort_grid = 0;               // workgroups of the digit-stream sort's persistent kernels (sort.hip): 0 = one per two compute units (128 on an MI355X)
    int sort_stage = 1;              // the sort's scatter passes: 1 = a tile's entries staged through LDS (whole runs per store), 0 = straight to memory (4 KB of LDS per workgroup)
    int sort_generic = 0;            // 1: the runtime-window level 0 of the sort even for the shapes that have a compile-time one (tests compare the two)
    int sort_tile = 0;               // entries a sort workgroup stages in LDS at a time: 0 = 4096 (40 KB of LDS), 2048 (24 KB), 1024 (16 KB)
    int aux_priority = 0;            // 1: the auxiliary (digit-stream) HIP stream is created with the highest stream priority
    int r1cs_order = 1;              // a, b, c = L.w, R.w, O.w: 1 = the rows in the matrix's evaluation order (equal length and coefficient pattern side by side: r1cs.cuh), 0 = natural order
    int solver_defer_checks = 1;     // with zkpor_solver_set_abc_dev: the run leaves its CHECK instructions (assertions) out and zkpor_solver_eval_abc_dev verifies a x b = c on EVERY row; 0 = the run executes them
    int64_t solver_tree_from = 1024; // levels from this many generic instructions on: the divisions of a workgroup share one inversion, long constraints go to k_solve_long
    int solver_beside = 1;           // Poseidon calls with a join level run on a side stream beside the levels up to it; 0 = in place, as ordinary calls
    int solver_pre_join = 1;         // the inputs of a Poseidon call with a join level (the challenge sponge: 1 392 expressions) are evaluated side by side in front of the serial kernel (solver.hip k_hint_inputs), as the ASYNC calls' always were; 0 = inside it, one lane at a time
    int solver_long = 256;           // wide levels hand constraints of more terms than this to a wave each (k_solve_long); 0 = every instruction in its own thread
    int solver_chain = 1;            // runs of one-instruction levels in k_solve_chain (decoded side by side, values handed on in registers); 0 = the narrow kernel
    int64_t solver_batch_from = 1 << 21;  // levels from this many generic instructions on run 4 per thread with ONE inversion per thread (solver.hip)
    int poseidon_coop = -1;          // account leaves / CEX commitments 16 lanes per hash chain: -1 = when a launch has fewer than 65 536 chains, 0 = never, 1 = always
    int solver_poseidon = 1;         // the solver program's Poseidon instruction: 1 = sixteen lanes per call (latency), 0 = one thread per call
    int64_t poseidon_defer = 64;     // Poseidon launches of up to this many calls (16 lanes each) park their S-box inputs raw and leave conversions, wires and rows to a wide kernel behind them (poseidon.hip k_sbox_expand); 0 = never
    int gpu_token = 1;               // host-pointer calls of several contexts on one GPU take turns on the device (api_core.hip GpuTurn)
    int tail_reserve_cus = 0;        // > 0: the prove tail's kernels (NTT passes, digit streams, accumulations) run on streams whose CU mask leaves this many
                                     // compute units free (evenly over the XCDs) — for the narrow, dependent launches of ANOTHER worker's solver program, which
                                     // otherwise queue behind full-size MSM grids (solve(i+1) beside tail(i): host/prover_host.hpp workers, bench.py end_to_end)
    int tail_digits_early = 1;       // a tail that has to wait for the device turn builds its digit stream of w first, beside the other worker's tail (groth16.hip prove_sums)
    int tail_streams = 0;            // 1: the prove tail runs on its own pair of streams (hardware queues of their own) and takes the device turn even WITHOUT a reserve
                                     // (tail_reserve_cus 0: every compute unit) — for workers whose solver runs on a high-priority stream ("stream_priority")
    int stream_priority = 0;         // 1: the context's own stream was re-created with the highest stream priority (the solver's narrow dependent launches then get the
                                     // compute units that free up before another worker's full-size tail grids do, without a CU reserve)
    int tail_aux_masked = 0;         // 1: the digit streams of a masked tail keep to the tail's CU mask; 0: they may use the reserved units too
    // The masked streams, created on first use and NEVER destroyed before the context itself: one (main, aux) pair per value "tail_reserve_cus" has
    // had (at most TAIL_SETS_MAX values per context), tail_aux_free (every CU, own hardware queue) once.  Round 5 destroyed and re-created them when the
    // parameter changed and the second generation crashed inside the HIP runtime: events of the context (the phase timers' pending pairs, the pool)
    // still name the stream they were last recorded on.  tail_stream / tail_aux are the pair of the current value (null until a tail has run with it).
    struct TailSet { int reserve; hipStream_t main, aux, chain; };
    static constexpr size_t TAIL_SETS_MAX = 4;
    std::vector<TailSet> tail_sets;
    hipStream_t tail_stream = nullptr, tail_aux = nullptr, tail_aux_free = nullptr;
    hipStream_t tail_chain = nullptr;           // "msm_chain": the chain stream of the current reserve value's set (own hardware queue, the tail's mask)
    hipStream_t chain_stream = nullptr;         // ... and of a tail that runs on the context's own stream (an ordinary stream)
    char* ws2 = nullptr; size_t ws2_cap = 0;    // "msm_chain": the second accumulation region (B1, B2), sized by the sums' ACTUAL entry counts, grow-only (groth16.hip prove_sums)
    std::vector<hipStream_t> retired_streams;   // streams a handle of this context replaced (a solver's first side streams): destroyed with the context
    std::vector<hipStream_t> retired_own_queue; // ... those of them that have a hardware queue of their own: back to the process-wide pool with the context
    int debug_validate = 0;          // 1: every sorted digit stream is checked (keys ascending and below the bucket count, point indices inside the array) on the
                                     // accumulating stream before its level-1 kernel reads it; a violation is ZKPOR_E_STATE instead of a GPU memory fault
    uint32_t* dbg_buf = nullptr;     // 4 words of device memory for that check
};

#define ZK_HIP(ctx, call)                                                                             \
    do {                                                                                              \
        hipError_t e__ = (call);                                                                      \
        if (e__ != hipSuccess) {                                                                      \
            (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                          \
            return e__ == hipErrorOutOfMemory ? ZKPOR_E_OOM : ZKPOR_E_HIP;                            \
        }                                                                                             \
    } while (0)
#define ZK_TRY(expr)                  \
    do {                              \
        int32_t rc__ = (expr);        \
        if (rc__ != ZKPOR_OK) return rc__; \
    } while (0)
#define ZK_KERNEL_CHECK(ctx) ZK_HIP(ctx, hipGetLastError())

// The exception firewall of the C ABI (include/zkpor.h: "never throw"): every `int32_t zkpor_*` entry point is a function-try-block
// that ends in ZK_ABI_CATCH (tools/abi_firewall.py keeps it so).  std::vector / std::string / std::function / std::thread inside the
// library can throw; a cgo caller cannot unwind — the reference's prover logs an error return and moves on
// (src/prover/prover/prover.go:269-272), it must never be killed by its backend.  The text of the exception is kept per host thread
// and shown by the next zkpor_last_error of that thread.
// ZK_ABI_CATCH_IN(ctx-expression): the same, for entry points that have a context in scope (the parameters of a function are visible in the
// handlers of its function-try-block) — the text goes straight into THAT context's error string, inside a nested try: under cgo a goroutine may
// move to another OS thread between the failing call and zkpor_last_error, and a per-thread text would be lost, or shown later for another context
// (ADVICE r05).  Entry points without a handle keep the per-thread buffer, which every entry (ZK_ENTER) clears.
namespace zk { void abi_exception(const char* what) noexcept; void abi_exception_in(zkpor_ctx* ctx, const char* what) noexcept; void abi_exception_clear() noexcept; void abort_trace_install(); }
#define ZK_ABI_CATCH                                                                                      \
    catch (const std::bad_alloc&) { zk::abi_exception("std::bad_alloc"); return ZKPOR_E_OOM; }           \
    catch (const std::exception& e__) { zk::abi_exception(e__.what()); return ZKPOR_E_HIP; }             \
    catch (...) { zk::abi_exception("unknown C++ exception"); return ZKPOR_E_HIP; }
#define ZK_ABI_CATCH_IN(ctx_expr)                                                                                   \
    catch (const std::bad_alloc&) { zk::abi_exception_in((ctx_expr), "std::bad_alloc"); return ZKPOR_E_OOM; }       \
    catch (const std::exception& e__) { zk::abi_exception_in((ctx_expr), e__.what()); return ZKPOR_E_HIP; }         \
    catch (...) { zk::abi_exception_in((ctx_expr), "unknown C++ exception"); return ZKPOR_E_HIP; }

namespace zk {

// HIP's current device is a per-host-thread setting and new threads start on device 0: every entry point that takes a
// context / key / tree / R1CS handle binds the calling thread to the handle's GPU for the duration of the call and restores
// what was there before, so a handle may be used from any thread (one caller at a time) — allocations, events and launches
// always land on the GPU the handle was created for.
struct DevGuard {
    int prev = -1;
    explicit DevGuard(int dev) {
        abi_exception_clear();   // a text left by an earlier call of this thread that nobody asked for is not this call's error
        if (dev < 0) return;
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) cur = -1;
        if (cur != dev) { (void)hipSetDevice(dev); prev = cur; }
    }
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
    DevGuard(const DevGuard&) = delete;
    DevGuard& operator=(const DevGuard&) = delete;
};
#define ZK_ENTER(dev_expr) zk::DevGuard zk_dev_guard__(dev_expr)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// make sure the workspace can hold `bytes`; invalidates previous ws_alloc results
inline int32_t ws_reserve(zkpor_ctx* ctx, size_t bytes) {
    bytes = align_up(bytes, 1 << 20);
    ctx->ws_off = 0;
    if (bytes <= ctx->ws_cap) return ZKPOR_OK;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->ws) { ZK_HIP(ctx, hipFree(ctx->ws)); ctx->ws = nullptr; ctx->ws_cap = 0; }
    ZK_HIP(ctx, hipMalloc((void**)&ctx->ws, bytes));
    ctx->ws_cap = bytes;
    return ZKPOR_OK;
}
template <class T>
inline T* ws_alloc(zkpor_ctx* ctx, size_t count) {
    size_t bytes = align_up(count * sizeof(T), 256);
    if (ctx->ws_off + bytes > ctx->ws_cap) return nullptr;
    T* p = (T*)(ctx->ws + ctx->ws_off);
    ctx->ws_off += bytes;
    return p;
}
struct WsPlan {  // accumulate sizes first, then reserve once
    size_t total = 0;
    template <class T>
    void add(size_t count) { total += align_up(count * sizeof(T), 256); }
};

inline hipEvent_t ev_get(zkpor_ctx* ctx) {
    if (!ctx->event_pool.empty()) {
        hipEvent_t e = ctx->event_pool.back();
        ctx->event_pool.pop_back();
        return e;
    }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}
struct PhaseScope {  // RAII: GPU time of everything enqueued on the stream while alive
    zkpor_ctx* ctx;
    PhaseTimer* t;
    hipEvent_t a, b;
    PhaseScope(zkpor_ctx* c, const char* name) : ctx(c), t(&c->phases[name]) {
        a = ev_get(c); b = ev_get(c);
        (void)hipEventRecord(a, c->stream);
    }
    ~PhaseScope() {
        (void)hipEventRecord(b, ctx->stream);
        t->pending.push_back({a, b});
        t->calls++;
    }
};
struct HostPhase {  // RAII: wall-clock time of host code, reported through the same phase table
    PhaseTimer* t;
    std::chrono::steady_clock::time_point t0;
    HostPhase(zkpor_ctx* c, const char* name) : t(&c->phases[name]), t0(std::chrono::steady_clock::now()) {}
    ~HostPhase() { t->ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); t->calls++; }
};
inline void phase_resolve(zkpor_ctx* ctx, PhaseTimer& t) {
    for (auto& pr : t.pending) {
        (void)hipEventSynchronize(pr.second);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, pr.first, pr.second);
        t.ms += ms;
        ctx->event_pool.push_back(pr.first);
        ctx->event_pool.push_back(pr.second);
    }
    t.pending.clear();
}

// api_core.hip: grow-only device staging area of the context (contents undefined after a growth)
int32_t stage_reserve(zkpor_ctx* ctx, size_t bytes);
// queue a host -> device copy on ctx->copy_stream.  Page-locked sources (hipHostMalloc / zkpor_host_register) are handed to
// the DMA engine directly; pageable ones (a Go / numpy heap slice) go through pinned bounce buffers filled by a few host
// threads, chunk k+1 being copied by the CPU while chunk k crosses PCIe.  Returns when the last chunk is QUEUED: the source
// range is no longer needed afterwards unless it was page-locked (then: until the copy stream has drained).
int32_t host_upload(zkpor_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, bool allow_runtime_pin = true, hipStream_t on = nullptr);
// host memory of unknown kind -> device, complete on return; never lets the runtime page-lock the caller's range (api_core.hip)
int32_t h2d_sync(zkpor_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
// A HIP stream with its OWN hardware queue.  The runtime multiplexes ordinary streams onto a pool of four hardware queues, in order: a kernel
// of one stream waits behind whatever another stream of the same queue launched before it — a 0.2 s hash chain on a solver's side stream in front
// of a decompose on the digit stream was a 135 ms hole in every proof (profiles/r05_timeline_hole.txt).  A stream created with a CU mask gets a
// queue of its own; `reserve` compute units are left out of the mask (0 = every CU: only the queue is wanted).  Such streams synchronise with the
// legacy NULL stream, which the per-proof paths therefore never use.
int32_t stream_create_own_queue(zkpor_ctx* ctx, hipStream_t* out, int reserve_cus);
// ... and back: such a stream is never destroyed, it waits in a process-wide pool for the next caller with the same (device, reserve) — api_core.hip
void stream_release_own_queue(int device, hipStream_t st, int reserve_cus);
// after a failed call: nothing the prove tail queued on its own streams (all of them: the unmasked digit stream too) may still touch the stage or the workspace
inline void drain_tail_streams(zkpor_ctx* ctx) {
    for (hipStream_t st : {ctx->tail_stream, ctx->tail_aux, ctx->tail_aux_free, ctx->tail_chain, ctx->chain_stream}) if (st) (void)hipStreamSynchronize(st);
}
void bounce_free(zkpor_ctx* ctx);
// One caller at a time runs the GPU part of a host-pointer call on a device (the others keep moving their vectors across PCIe
// meanwhile).  Without it two callers drift into lockstep: their kernels share the GPU, finish together, and then both copy at
// the same time with the GPU idle (measured: profiles/r02_boundary_turns.txt).  A turn is held from "my vectors are across" to
// "my last kernel is done"; RAII, released on every path.
struct GpuTurn {
    int dev = -1;
    bool held = false;
    GpuTurn() = default;
    ~GpuTurn() { release(); }
    GpuTurn(const GpuTurn&) = delete;
    GpuTurn& operator=(const GpuTurn&) = delete;
    bool try_acquire(zkpor_ctx* ctx);   // false = somebody else is on the device (or already held)
    void acquire(zkpor_ctx* ctx);       // waits for the turn; no-op when the context's "gpu_token" parameter is 0
    void release();
};
// r1cs.hip: a, b, c = L.w, R.w, O.w queued on `ctx`'s stream (any context of the GPU the matrices live on: they are only read)
int32_t r1cs_eval_on(zkpor_ctx* ctx, zkpor_r1cs* r, const void* d_w, void* d_a, void* d_b, void* d_c, size_t domain_size, const u32* d_skip = nullptr);
void r1cs_dims(const zkpor_r1cs* r, size_t* n_constraints, size_t* n_wires, int* device);

// solver.hip: the context and the constraint system a solver program was created on
zkpor_ctx* solver_ctx(zkpor_solver* s);
zkpor_r1cs* solver_r1cs(zkpor_solver* s);

}  // namespace zk
