// C ABI (include/zkpor.h): context, device memory helpers, tuning, phase timers, generic MSM entry points.
#include "common.cuh"
#include "msm.cuh"
#include "ntt.cuh"
#include <condition_variable>
#include <memory>
#include <mutex>
#include <execinfo.h>
#include <fcntl.h>
#include <signal.h>
#include <sys/prctl.h>
#include <unistd.h>
#include <stdexcept>
#include <thread>

using namespace zk;

namespace zk {
static thread_local char g_abi_exception[200] = {0};
void abi_exception(const char* what) noexcept { snprintf(g_abi_exception, sizeof g_abi_exception, "%s", what ? what : "?"); }
void abi_exception_clear() noexcept { g_abi_exception[0] = 0; }
void abi_exception_in(zkpor_ctx* ctx, const char* what) noexcept {
    if (ctx) {
        try { ctx->err = std::string("C++ exception stopped at the ABI: ") + (what ? what : "?"); return; } catch (...) {}   // no memory for the text either: the thread's buffer
    }
    abi_exception(what);
}
void pos_tables_free(zkpor_ctx* ctx);
int32_t ensure_pinned(zkpor_ctx* ctx, size_t bytes) {
    if (bytes <= ctx->pinned_cap) return ZKPOR_OK;
    if (ctx->pinned) { ZK_HIP(ctx, hipHostFree(ctx->pinned)); ctx->pinned = nullptr; ctx->pinned_cap = 0; }
    ZK_HIP(ctx, hipHostMalloc(&ctx->pinned, bytes, hipHostMallocDefault));
    ctx->pinned_cap = bytes;
    return ZKPOR_OK;
}

int32_t stage_reserve(zkpor_ctx* ctx, size_t bytes) {
    bytes = align_up(bytes, 1 << 20);
    if (bytes <= ctx->stage_cap) return ZKPOR_OK;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->copy_stream) ZK_HIP(ctx, hipStreamSynchronize(ctx->copy_stream));
    if (ctx->stage) { ZK_HIP(ctx, hipFree(ctx->stage)); ctx->stage = nullptr; ctx->stage_cap = 0; }
    ZK_HIP(ctx, hipMalloc((void**)&ctx->stage, bytes));
    ctx->stage_cap = bytes;
    return ZKPOR_OK;
}

// Pinned bounce buffers + the host threads that fill them.  One per context, created on the first pageable upload.
struct Bounce {
    static constexpr int SLOTS = 4;
    size_t CHUNK = (size_t)32 << 20;   // context parameter "copy_chunk_mb"
    char* buf[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    bool used[SLOTS] = {false, false, false, false};
    int next = 0;
    // worker threads: each copies slice t of the current chunk
    std::vector<std::thread> th;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    uint64_t gen = 0;
    int pending = 0;
    bool quit = false;
    char* dst = nullptr;
    const char* src = nullptr;
    size_t len = 0;
    int parts = 1;

    static void slice(size_t len, int parts, int t, size_t* off, size_t* n) {
        size_t per = (len / parts + 63) & ~(size_t)63;
        size_t o = per * t;
        if (o > len) o = len;
        *off = o;
        *n = (t == parts - 1) ? len - o : (o + per > len ? len - o : per);
    }
    void worker(int t) {
        uint64_t seen = 0;
        for (;;) {
            std::unique_lock<std::mutex> lk(mu);
            cv_go.wait(lk, [&] { return quit || gen != seen; });
            if (quit) return;
            seen = gen;
            char* d = dst; const char* s = src; size_t l = len; int p = parts;
            lk.unlock();
            size_t off, n;
            slice(l, p, t, &off, &n);
            if (n) memcpy(d + off, s + off, n);
            lk.lock();
            if (--pending == 0) cv_done.notify_one();
        }
    }
    void start(int n_threads) {
        parts = n_threads < 1 ? 1 : n_threads;
        for (int t = 1; t < parts; ++t) th.emplace_back([this, t] { worker(t); });
    }
    void copy(char* d, const char* s, size_t l) {  // all threads (the caller is thread 0)
        if (parts > 1) {
            std::unique_lock<std::mutex> lk(mu);
            dst = d; src = s; len = l; pending = parts - 1; ++gen;
            cv_go.notify_all();
        }
        size_t off, n;
        slice(l, parts, 0, &off, &n);
        if (n) memcpy(d + off, s + off, n);
        if (parts > 1) {
            std::unique_lock<std::mutex> lk(mu);
            cv_done.wait(lk, [&] { return pending == 0; });
        }
    }
    ~Bounce() {
        {
            std::unique_lock<std::mutex> lk(mu);
            quit = true;
            cv_go.notify_all();
        }
        for (auto& t : th) t.join();
        for (int i = 0; i < SLOTS; ++i) {
            if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (buf[i]) (void)hipHostFree(buf[i]);
        }
    }
};

void bounce_free(zkpor_ctx* ctx) {
    if (ctx->bounce && ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);  // no DMA may still read the buffers
    if (ctx->bounce && ctx->stream) (void)hipStreamSynchronize(ctx->stream);             // (h2d_sync copies on the context's own stream)
    delete (Bounce*)ctx->bounce;
    ctx->bounce = nullptr;
}

// ---- ZKPOR_ABORT_TRACE=1: the native stack of whichever thread raises SIGABRT, on stderr, before the process dies ----
// The HIP / HSA runtimes abort() from their own threads on a device exception, and a Python caller's faulthandler can only say "some thread
// that is not mine" (GPUTEST_r04: a silent rc 134 inside zkpor_prove_tail).  Async-signal-safe enough for a process that is dying anyway:
// backtrace() is pre-loaded at install time, the symbols go straight to fd 2.  The previous handler (faulthandler's) runs afterwards.
namespace {
struct sigaction g_prev_abrt, g_prev_segv, g_prev_bus;
bool g_abort_trace_on = false;
int g_abort_fd = 2;     // ZKPOR_ABORT_TRACE=1: stderr; any other value: a file of that name (a test runner that captures fd 2 loses it with the process)
void abort_trace(int sig) {
    const int fd = g_abort_fd;
    static const char head_a[] = "\n[zkpor] SIGABRT raised on thread ", head_s[] = "\n[zkpor] SIGSEGV raised on thread ", head_b[] = "\n[zkpor] SIGBUS raised on thread ";
    if (sig == SIGSEGV) (void)!write(fd, head_s, sizeof head_s - 1);
    else if (sig == SIGBUS) (void)!write(fd, head_b, sizeof head_b - 1);
    else (void)!write(fd, head_a, sizeof head_a - 1);
    char name[32] = {0};
    (void)prctl(PR_GET_NAME, name, 0, 0, 0);
    (void)!write(fd, name, strlen(name));
    static const char mid[] = " - native stack:\n";
    (void)!write(fd, mid, sizeof mid - 1);
    void* fr[96];
    const int n = backtrace(fr, 96);
    backtrace_symbols_fd(fr, n, fd);
    (void)sigaction(sig, sig == SIGSEGV ? &g_prev_segv : sig == SIGBUS ? &g_prev_bus : &g_prev_abrt, nullptr);     // hand over to whoever was there before (Python's faulthandler, else the default action)
    (void)raise(sig);
}
}  // namespace
void abort_trace_install() {
    static std::once_flag once;
    std::call_once(once, [] {
        const char* e = getenv("ZKPOR_ABORT_TRACE");
        if (!e || !*e || strcmp(e, "0") == 0) return;
        if (strcmp(e, "1") != 0) {
            const int fd = open(e, O_WRONLY | O_CREAT | O_APPEND, 0644);
            if (fd >= 0) g_abort_fd = fd;
        }
        void* fr[4];
        (void)backtrace(fr, 4);                      // loads the unwinder now, not inside the handler
        struct sigaction sa;
        memset(&sa, 0, sizeof sa);
        sa.sa_handler = abort_trace;
        sigemptyset(&sa.sa_mask);
        sa.sa_flags = SA_NODEFER;
        g_abort_trace_on = sigaction(SIGABRT, &sa, &g_prev_abrt) == 0;
        // round 6: a memory fault of a HOST thread inside the library or the runtime as well (the masked-stream crash of round 5 was one: "Segmentation
        // fault" inside zkpor_prove_tail_dev and nothing else).  Installed BEFORE Python's faulthandler only if the library is loaded first; either way the
        // previous handler runs after the stack is out.
        sa.sa_flags = SA_NODEFER | SA_ONSTACK;
        (void)sigaction(SIGSEGV, &sa, &g_prev_segv);
        (void)sigaction(SIGBUS, &sa, &g_prev_bus);
    });
}

// ---- turns on the device (common.cuh GpuTurn): one flag per GPU, shared by every context of the process ----
namespace {
struct TurnState { std::mutex mu; std::condition_variable cv; bool busy = false; };
TurnState& turn_state(int dev) {
    static TurnState st[64];
    return st[dev < 0 ? 0 : dev % 64];
}
}  // namespace
bool GpuTurn::try_acquire(zkpor_ctx* ctx) {
    if (held) return false;
    if (!ctx->gpu_token) { return true; }   // arbitration off: everybody always has the turn
    TurnState& t = turn_state(ctx->device);
    std::lock_guard<std::mutex> lk(t.mu);
    if (t.busy) return false;
    t.busy = true; held = true; dev = ctx->device;
    return true;
}
void GpuTurn::acquire(zkpor_ctx* ctx) {
    if (held || !ctx->gpu_token) return;
    TurnState& t = turn_state(ctx->device);
    std::unique_lock<std::mutex> lk(t.mu);
    t.cv.wait(lk, [&] { return !t.busy; });
    t.busy = true; held = true; dev = ctx->device;
}
void GpuTurn::release() {
    if (!held) return;
    TurnState& t = turn_state(dev);
    { std::lock_guard<std::mutex> lk(t.mu); t.busy = false; }
    t.cv.notify_one();
    held = false;
}

// CU-masked streams are never handed back to the runtime: a context (or a solver) that goes away parks them in a process-wide pool and the next one that
// asks for the same (device, reserve) takes them from there.  Round 5's crash came with the SECOND generation of masked streams of a process; round 6
// still met it once when a worker context was destroyed and created again (a host SIGSEGV inside zkpor_prove_tail_dev, gpurun_out/r06d) although no
// stream of a LIVE context was destroyed any more.  hipStreamDestroy of a masked stream is the one call both runs had in common; the pool removes it, and
// bounds the hardware queues a process ever creates by the most it uses at one time.
namespace {
struct PooledStream { int device, reserve; hipStream_t st; };
std::mutex g_stream_pool_mu;
std::vector<PooledStream> g_stream_pool;
}  // namespace
// at process exit the parked streams go back to the runtime BEFORE its own exit handlers run (atexit: last registered, first run) — left alive they
// took rocprofv3's teardown down with them (a SIGSEGV in __cxa_finalize after every output was written: gpurun_out/r06p)
void stream_pool_drain_at_exit() {
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    for (auto& p : g_stream_pool) (void)hipStreamDestroy(p.st);
    g_stream_pool.clear();
}
void stream_release_own_queue(int device, hipStream_t st, int reserve_cus) {
    if (!st) return;
    (void)hipStreamSynchronize(st);
    static std::once_flag once;
    std::call_once(once, [] { (void)atexit(stream_pool_drain_at_exit); });
    std::lock_guard<std::mutex> lk(g_stream_pool_mu);
    // an idle hardware queue is not free: a process that had collected ~25 of them (a sweep over five reserve values, two workers each) ran the SAME
    // region 1.7x slower than at its start (profiles/r06_tail_mode_sweep.json) — the pool is for re-use by the next context, not a museum
    if (g_stream_pool.size() >= 12) { (void)hipStreamDestroy(st); return; }
    try { g_stream_pool.push_back({device, reserve_cus, st}); } catch (...) { /* out of host memory: the stream is leaked, not destroyed */ }
}
int32_t stream_create_own_queue(zkpor_ctx* ctx, hipStream_t* out, int reserve_cus) {
    {
        std::lock_guard<std::mutex> lk(g_stream_pool_mu);
        for (size_t i = 0; i < g_stream_pool.size(); ++i)
            if (g_stream_pool[i].device == ctx->device && g_stream_pool[i].reserve == reserve_cus) {
                *out = g_stream_pool[i].st;
                g_stream_pool.erase(g_stream_pool.begin() + (long)i);
                return ZKPOR_OK;
            }
    }
    hipDeviceProp_t prop;
    ZK_HIP(ctx, hipGetDeviceProperties(&prop, ctx->device));
    const int cus = prop.multiProcessorCount;
    if (reserve_cus < 0 || reserve_cus >= cus) { ctx->err = "stream: the CU mask would leave no compute unit"; return ZKPOR_E_ARG; }
    // mask bit i is compute unit i / 8 of XCD i % 8 on this part (the driver deals the bits round-robin over the XCDs): clearing the first R bits
    // frees R / 8 units on each of the eight XCDs
    std::vector<uint32_t> mask((size_t)(cus + 31) / 32, 0u);
    for (int i = reserve_cus; i < cus; ++i) mask[(size_t)i / 32] |= 1u << (i % 32);
    ZK_HIP(ctx, hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data()));
    return ZKPOR_OK;
}

// `on`: the stream the copies are queued on — the context's copy stream (created on first use) for the host-pointer prove tail, whose uploads run
// beside kernels; the context's own stream for h2d_sync (a sixth stream would share a hardware queue with one of the others)
int32_t host_upload(zkpor_ctx* ctx, void* d_dst, const void* h_src, size_t bytes, bool allow_runtime_pin, hipStream_t on) {
    if (!bytes) return ZKPOR_OK;
    if (!on) {
        if (!ctx->copy_stream) ZK_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
        on = ctx->copy_stream;
    }
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, h_src) == hipSuccess && at.type == hipMemoryTypeHost) {
        ZK_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, on));
        return ZKPOR_OK;
    }
    (void)hipGetLastError();  // an unregistered pointer is the expected case, not an error
    if (ctx->copy_threads == 0 && allow_runtime_pin) {
        // "copy_threads" 0 (opt-in since round 5): hand the pageable range to the HIP runtime, which page-locks it on the fly and lets the DMA
        // engine read the caller's pages directly (no CPU copy at all; the call may block until the range has been read).  Fastest (56 GB/s),
        // but the GPU then reads pages the kernel may still migrate — a transparent-huge-page collapse of a MADV_HUGEPAGE range (numpy marks
        // every array of >= 4 MiB so) under the copy is the one explanation round 5 found for GPUTEST_r04's abort.
        ZK_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, on));
        return ZKPOR_OK;
    }
    Bounce* b = (Bounce*)ctx->bounce;
    if (!b) {
        // built completely in a local object and published only once it works: a pinned allocation that fails half way must not
        // leave a half-initialised Bounce (null buffers, no workers) for the next upload to copy into
        std::unique_ptr<Bounce> nb(new (std::nothrow) Bounce());
        if (!nb) { ctx->err = "host_upload: out of host memory"; return ZKPOR_E_OOM; }
        nb->CHUNK = (size_t)ctx->copy_chunk_mb << 20;
        for (int i = 0; i < Bounce::SLOTS; ++i) {
            ZK_HIP(ctx, hipHostMalloc((void**)&nb->buf[i], nb->CHUNK, hipHostMallocDefault));   // ~Bounce frees what exists on the way out
            ZK_HIP(ctx, hipEventCreateWithFlags(&nb->ev[i], hipEventDisableTiming));
        }
        nb->start(ctx->copy_threads > 0 ? ctx->copy_threads : 1);
        b = nb.release();
        ctx->bounce = b;
    }
    const char* src = (const char*)h_src;
    char* dst = (char*)d_dst;
    for (size_t off = 0; off < bytes; off += b->CHUNK) {
        size_t n = bytes - off < b->CHUNK ? bytes - off : b->CHUNK;
        int s = b->next;
        b->next = (s + 1) % Bounce::SLOTS;
        if (b->used[s]) ZK_HIP(ctx, hipEventSynchronize(b->ev[s]));  // the DMA that last read this slot is done
        b->copy(b->buf[s], src + off, n);
        ZK_HIP(ctx, hipMemcpyAsync(dst + off, b->buf[s], n, hipMemcpyHostToDevice, on));
        ZK_HIP(ctx, hipEventRecord(b->ev[s], on));
        b->used[s] = true;
    }
    return ZKPOR_OK;
}
// Host memory of unknown kind -> device, complete when the call returns.  Small ranges go through the runtime's own staging buffers (a CPU copy
// inside hipMemcpy); large pageable ones through the context's pinned bounce buffers — the library never lets the DMA engine read pages it
// does not own (see host_upload).  dst must not be in use by work queued on other streams than ctx->stream.
int32_t h2d_sync(zkpor_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
    if (!bytes) return ZKPOR_OK;
    if (bytes <= ((size_t)256 << 10)) {      // below every pinning threshold of the runtime: staged by the runtime itself
        ZK_HIP(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return ZKPOR_OK;
    }
    ZK_TRY(host_upload(ctx, d_dst, h_src, bytes, false, ctx->stream));     // in the context's own stream: behind whatever it queued before
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKPOR_OK;
}
}  // namespace zk

// ---- seeded Fr fill (bench/test inputs generated in HBM) ----
__device__ __forceinline__ u64 splitmix(u64& s) {
    u64 z = (s += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
__global__ void k_fill_fr(Fr* out, size_t n, u64 seed, int kind) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 s = seed ^ (i * 0xd1342543de82ef95ULL + 0x2545f4914f6cdd1dULL);
    u64 sel = splitmix(s) % 100;
    Fr x = Fr::zero();
    int bits = 254;
    if (kind == 1) {          // zkpor50_1380 estimate: 25% {0,1}, 20% < 2^16, 5% < 2^64, 50% uniform
        if (sel < 25) bits = 1;
        else if (sel < 45) bits = 16;
        else if (sel < 50) bits = 64;
    } else if (kind == 2) {   // zkpor500_200 estimate (Appendix B: Poseidon share falls, limb / boolean share rises): 35 / 30 / 5 / 30
        if (sel < 35) bits = 1;
        else if (sel < 65) bits = 16;
        else if (sel < 70) bits = 64;
    }
    u64 r0 = splitmix(s), r1 = splitmix(s), r2 = splitmix(s), r3 = splitmix(s);
    if (bits == 1) { r0 &= 1; r1 = r2 = r3 = 0; }
    else if (bits == 16) { r0 &= 0xffff; r1 = r2 = r3 = 0; }
    else if (bits == 64) { r1 = r2 = r3 = 0; }
    else { r3 &= 0x0fffffffffffffffULL; }  // < 2^252 < r: canonical
    x.v[0] = (u32)r0; x.v[1] = (u32)(r0 >> 32); x.v[2] = (u32)r1; x.v[3] = (u32)(r1 >> 32);
    x.v[4] = (u32)r2; x.v[5] = (u32)(r2 >> 32); x.v[6] = (u32)r3; x.v[7] = (u32)(r3 >> 32);
    out[i] = Fr::to_mont(x);
}

// host-side group additions for combining partial results (the single-proof MSM split: every rank sums its slice
// of the key, the 8 partial Jacobian points are all-gathered and added here — RCCL has no reduction over a curve)
template <class F>
static void jac_sum(const Jacobian<F>* parts, size_t count, Jacobian<F>* out) {
    XYZZ<F> acc = XYZZ<F>::inf();
    for (size_t i = 0; i < count; ++i) {
        const Jacobian<F>& j = parts[i];
        if (j.z.is_zero()) continue;
        // Jacobian (X,Y,Z) -> XYZZ (X, Y, Z^2, Z^3)
        XYZZ<F> p = {j.x, j.y, F::sqr(j.z), F::mul(F::sqr(j.z), j.z)};
        xyzz_add<F>(acc, p);
    }
    *out = xyzz_to_jacobian<F>(acc);
}

template <class F>
static int32_t msm_host(zkpor_ctx* ctx, const void* pts, const uint64_t* scalars, size_t n, XYZZ<F>* r) {
    if (n == 0) { *r = XYZZ<F>::inf(); return ZKPOR_OK; }
    void *dp = nullptr, *dsc = nullptr;
    ZK_HIP(ctx, hipMalloc(&dp, n * sizeof(Affine<F>)));
    hipError_t e = hipMalloc(&dsc, n * 32);
    if (e != hipSuccess) { (void)hipFree(dp); ctx->err = "hipMalloc scalars"; return ZKPOR_E_OOM; }
    int32_t rc = ZKPOR_OK;
    if (h2d_sync(ctx, dp, pts, n * sizeof(Affine<F>)) != ZKPOR_OK || h2d_sync(ctx, dsc, scalars, n * 32) != ZKPOR_OK) {
        ctx->err = "H2D copy failed"; rc = ZKPOR_E_HIP;
    }
    if (rc == ZKPOR_OK) rc = msm_dev<F>(ctx, (const Affine<F>*)dp, (const Fr*)dsc, n, r);
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(dp); (void)hipFree(dsc);
    return rc;
}

__global__ void k_fr_mul(Fr* out, const Fr* a, const Fr* b, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = Fr::mul(a[i], b[i]);
}

extern "C" {

int32_t zkpor_dev_fr_mul(zkpor_ctx* ctx, void* d_out, const void* d_a, const void* d_b, size_t n) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || (n && (!d_out || !d_a || !d_b))) return ZKPOR_E_ARG;
    if (n == 0) return ZKPOR_OK;
    hipLaunchKernelGGL(k_fr_mul, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (Fr*)d_out, (const Fr*)d_a, (const Fr*)d_b, n);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
int32_t zkpor_dev_copy(zkpor_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || (bytes && (!d_dst || !d_src))) return ZKPOR_E_ARG;
    ZK_HIP(ctx, hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_init(int device, void* stream, zkpor_ctx** out) try {
    if (!out) return ZKPOR_E_ARG;
    *out = nullptr;
    zk::abort_trace_install();
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device < 0 || device >= count) return ZKPOR_E_NODEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return ZKPOR_E_NODEVICE;
    // kernels are compiled for gfx950 only: refuse anything else rather than fail at first launch
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ZKPOR_E_NODEVICE;
    if (hipSetDevice(device) != hipSuccess) return ZKPOR_E_NODEVICE;
    zkpor_ctx* ctx = new (std::nothrow) zkpor_ctx();
    if (!ctx) return ZKPOR_E_OOM;
    ctx->device = device;
    if (stream) {
        ctx->stream = (hipStream_t)stream;
    } else {
        // Round 6: the context's own stream has a hardware queue of its own ("stream_own_queue" 1, the default).  The runtime deals ORDINARY streams onto four
        // hardware queues in creation order; two contexts of one GPU whose streams land in one queue wait behind each other's launches — the same
        // two-worker region ran 327 or 308 ms per proof depending on what the process had created before (profiles/r06_stream_own_queue_ab.json)
        if (zk::stream_create_own_queue(ctx, &ctx->stream, 0) == ZKPOR_OK) ctx->stream_pooled = true;
        else {
            (void)hipGetLastError();
            if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { delete ctx; return ZKPOR_E_HIP; }
        }
        ctx->own_stream = true;
    }
    *out = ctx;
    return ZKPOR_OK;
} ZK_ABI_CATCH

void zkpor_destroy(zkpor_ctx* ctx) try {
    if (!ctx) return;
    ZK_ENTER(ctx->device);
    // every stream drained first, then the events (they name the stream they were last recorded on), then the streams
    (void)hipStreamSynchronize(ctx->stream);
    for (hipStream_t st : {ctx->aux_stream, ctx->copy_stream, ctx->tail_aux_free, ctx->chain_stream}) if (st) (void)hipStreamSynchronize(st);
    for (auto& ts : ctx->tail_sets) { (void)hipStreamSynchronize(ts.main); if (ts.aux) (void)hipStreamSynchronize(ts.aux); if (ts.chain) (void)hipStreamSynchronize(ts.chain); }
    for (hipStream_t st : ctx->retired_streams) (void)hipStreamSynchronize(st);
    for (hipStream_t st : ctx->retired_own_queue) (void)hipStreamSynchronize(st);
    for (auto& kv : ctx->phases)
        for (auto& pr : kv.second.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto e : ctx->event_pool) (void)hipEventDestroy(e);
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->ws2) (void)hipFree(ctx->ws2);
    if (ctx->dbg_buf) (void)hipFree(ctx->dbg_buf);
    pos_tables_free(ctx);
    ntt_domains_free(ctx);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    bounce_free(ctx);  // drains the copy stream first: no DMA may still read the pinned buffers
    if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); ctx->copy_stream = nullptr; }
    if (ctx->stage) (void)hipFree(ctx->stage);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    zk::stream_release_own_queue(ctx->device, ctx->chain_stream, 0);
    for (auto& ts : ctx->tail_sets) { zk::stream_release_own_queue(ctx->device, ts.main, ts.reserve); zk::stream_release_own_queue(ctx->device, ts.aux, ts.reserve); zk::stream_release_own_queue(ctx->device, ts.chain, ts.reserve); }
    zk::stream_release_own_queue(ctx->device, ctx->tail_aux_free, 0);
    for (hipStream_t st : ctx->retired_streams) (void)hipStreamDestroy(st);
    for (hipStream_t st : ctx->retired_own_queue) zk::stream_release_own_queue(ctx->device, st, 0);
    if (ctx->own_stream) { if (ctx->stream_pooled) zk::stream_release_own_queue(ctx->device, ctx->stream, 0); else (void)hipStreamDestroy(ctx->stream); }
    delete ctx;
} catch (...) { zk::abi_exception("exception in zkpor_destroy"); }

const char* zkpor_last_error(zkpor_ctx* ctx) {
    if (!ctx) return "null context";
    if (zk::g_abi_exception[0]) {   // an exception stopped at the ABI on this thread since the last call: it is the error
        try { ctx->err = std::string("C++ exception stopped at the ABI: ") + zk::g_abi_exception; } catch (...) { return zk::g_abi_exception; }
        zk::g_abi_exception[0] = 0;
    }
    return ctx->err.c_str();
}

int32_t zkpor_sync(zkpor_ctx* ctx) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx) return ZKPOR_E_ARG;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_trim(zkpor_ctx* ctx) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx) return ZKPOR_E_ARG;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (hipStream_t st : {ctx->aux_stream, ctx->copy_stream, ctx->tail_aux_free, ctx->chain_stream}) if (st) ZK_HIP(ctx, hipStreamSynchronize(st));
    for (auto& ts : ctx->tail_sets) for (hipStream_t st : {ts.main, ts.aux, ts.chain}) if (st) ZK_HIP(ctx, hipStreamSynchronize(st));
    if (ctx->ws) { (void)hipFree(ctx->ws); ctx->ws = nullptr; ctx->ws_cap = 0; ctx->ws_off = 0; }
    if (ctx->ws2) { (void)hipFree(ctx->ws2); ctx->ws2 = nullptr; ctx->ws2_cap = 0; }
    if (ctx->stage) { (void)hipFree(ctx->stage); ctx->stage = nullptr; ctx->stage_cap = 0; }
    ntt_domains_free(ctx);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

uint32_t zkpor_abi_version(void) { return ZKPOR_ABI_VERSION; }

int32_t zkpor_set_param(zkpor_ctx* ctx, const char* name, int64_t value) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !name) return ZKPOR_E_ARG;
    std::string n(name);
    if (n == "msm_window") ctx->msm_window = (int)value;
    else if (n == "msm_chunk") { if (value != 0 && (value < 4 || value > 4096)) { ctx->err = "msm_chunk must be 0 (automatic) or 4..4096 (a level of the partial-sum recursion turns T threads into 2 T / chunk)"; return ZKPOR_E_ARG; } ctx->msm_chunk = (int)value; }
    else if (n == "msm_tables") { if (value < 1 || value > 8) { ctx->err = "msm_tables must be in [1,8]"; return ZKPOR_E_ARG; } ctx->msm_tables = (int)value; }
    else if (n == "msm_g1_variant") ctx->g1_variant = (int)value;
    else if (n == "msm_g2_variant") ctx->g2_variant = (int)value;
    else if (n == "ntt_variant") ctx->ntt_variant = (int)value;
    else if (n == "ntt_tile_log") { if (value < 9 || value > 12) { ctx->err = "ntt_tile_log must be in [9,12]"; return ZKPOR_E_ARG; } ctx->ntt_tile_log = (int)value; }
    else if (n == "msm_filter") { if (value < 0 || value > 2) { ctx->err = "msm_filter must be 0, 1 or 2"; return ZKPOR_E_ARG; } ctx->msm_filter = (int)value; }
    else if (n == "msm_filter_grid") { if (value < 0 || value > 2048) { ctx->err = "msm_filter_grid must be in [0,2048]"; return ZKPOR_E_ARG; } ctx->msm_filter_grid = (int)value; }
    else if (n == "ntt_twiddles") { if (value < 0 || value > 2) { ctx->err = "ntt_twiddles must be 0 (tables), 1 (generated where the table exceeds 16 MiB) or 2 (generated everywhere)"; return ZKPOR_E_ARG; } ctx->ntt_twiddles = (int)value; }
    else if (n == "ntt_h") { if (value < 0 || value > 1) { ctx->err = "ntt_h must be 0 (seven transforms) or 1 (six)"; return ZKPOR_E_ARG; } ctx->ntt_h = (int)value; }
    else if (n == "ntt_fuse") { if (value < 0 || value > 1) { ctx->err = "ntt_fuse must be 0 or 1"; return ZKPOR_E_ARG; } ctx->ntt_fuse = (int)value; }
    else if (n == "sort_grid") { if (value < 0 || value > 8192) { ctx->err = "sort_grid must be in [0,8192] (0 = two workgroups per compute unit)"; return ZKPOR_E_ARG; } ctx->sort_grid = (int)value; }
    else if (n == "sort_stage") { if (value < 0 || value > 1) { ctx->err = "sort_stage must be 0 or 1"; return ZKPOR_E_ARG; } ctx->sort_stage = (int)value; }
    else if (n == "sort_generic") { if (value < 0 || value > 1) { ctx->err = "sort_generic must be 0 or 1"; return ZKPOR_E_ARG; } ctx->sort_generic = (int)value; }
    else if (n == "sort_tile") { if (value != 0 && value != 1024 && value != 2048 && value != 4096) { ctx->err = "sort_tile must be 0 (4096), 1024, 2048 or 4096"; return ZKPOR_E_ARG; } ctx->sort_tile = (int)value; }
    else if (n == "sort_block") { if (value != 0 && value != 256 && value != 512) { ctx->err = "sort_block must be 0, 256 or 512"; return ZKPOR_E_ARG; } }   // rounds 3-5: the workgroup size of rocPRIM's onesweep; accepted and ignored since the sort is sort.hip's
    else if (n == "msm_chain") { if (value < 0 || value > 2) { ctx->err = "msm_chain must be 0, 1 (tails on streams with their own hardware queues) or 2 (every tail)"; return ZKPOR_E_ARG; } ctx->msm_chain = (int)value; }
    else if (n == "msm_reduce_scan") { if (value < 0 || value > 2) { ctx->err = "msm_reduce_scan must be 0, 1 or 2"; return ZKPOR_E_ARG; } ctx->msm_reduce_scan = (int)value; }
    else if (n == "msm_tail_chunk") { if (value != 0 && (value < 4 || value > 64)) { ctx->err = "msm_tail_chunk must be 0 or 4..64"; return ZKPOR_E_ARG; } ctx->msm_tail_chunk = (int)value; }
    else if (n == "aux_priority") {
        if (value < 0 || value > 1) { ctx->err = "aux_priority must be 0 or 1"; return ZKPOR_E_ARG; }
        if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); ctx->aux_stream = nullptr; }
        ctx->aux_priority = (int)value;
    }
    else if (n == "poseidon_coop") { if (value < -1 || value > 1) { ctx->err = "poseidon_coop must be -1 (by size), 0 or 1"; return ZKPOR_E_ARG; } ctx->poseidon_coop = (int)value; }
    else if (n == "r1cs_order") { if (value < 0 || value > 1) { ctx->err = "r1cs_order must be 0 (natural) or 1 (by row shape)"; return ZKPOR_E_ARG; } ctx->r1cs_order = (int)value; }
    else if (n == "solver_defer_checks") { if (value < 0 || value > 1) { ctx->err = "solver_defer_checks must be 0 or 1"; return ZKPOR_E_ARG; } ctx->solver_defer_checks = (int)value; }
    else if (n == "solver_tree_from") { if (value < 1) { ctx->err = "solver_tree_from must be positive"; return ZKPOR_E_ARG; } ctx->solver_tree_from = value; }
    else if (n == "solver_pre_join") { if (value < 0 || value > 1) { ctx->err = "solver_pre_join must be 0 or 1"; return ZKPOR_E_ARG; } ctx->solver_pre_join = (int)value; }
    else if (n == "solver_beside") { if (value < 0 || value > 1) { ctx->err = "solver_beside must be 0 or 1"; return ZKPOR_E_ARG; } ctx->solver_beside = (int)value; }
    else if (n == "solver_long") { if (value < 0 || value > (1 << 30)) { ctx->err = "solver_long must be 0 (off) or a term count"; return ZKPOR_E_ARG; } ctx->solver_long = (int)value; }
    else if (n == "solver_chain") { if (value < 0 || value > 1) { ctx->err = "solver_chain must be 0 or 1"; return ZKPOR_E_ARG; } ctx->solver_chain = (int)value; }
    else if (n == "solver_batch_from") { if (value < 1) { ctx->err = "solver_batch_from must be positive"; return ZKPOR_E_ARG; } ctx->solver_batch_from = value; }
    else if (n == "poseidon_defer") { if (value < 0 || value > 65535) { ctx->err = "poseidon_defer must be 0 ... 65535 (calls per launch)"; return ZKPOR_E_ARG; } ctx->poseidon_defer = value; }
    else if (n == "solver_poseidon") { if (value < 0 || value > 1) { ctx->err = "solver_poseidon must be 0 (one thread per call) or 1 (sixteen lanes per call)"; return ZKPOR_E_ARG; } ctx->solver_poseidon = (int)value; }
    else if (n == "gpu_token") { if (value < 0 || value > 1) { ctx->err = "gpu_token must be 0 or 1"; return ZKPOR_E_ARG; } ctx->gpu_token = (int)value; }
    else if (n == "host_order") { if (value < 0 || value > 1) { ctx->err = "host_order must be 0 or 1"; return ZKPOR_E_ARG; } ctx->host_order = (int)value; }
    else if (n == "copy_chunk_mb") { if (value < 1 || value > 1024) { ctx->err = "copy_chunk_mb must be in [1,1024]"; return ZKPOR_E_ARG; } if (ctx->bounce) bounce_free(ctx); ctx->copy_chunk_mb = (int)value; }
    else if (n == "copy_threads") { if (value < 0 || value > 64) { ctx->err = "copy_threads must be in [0,64]"; return ZKPOR_E_ARG; } if (ctx->bounce) bounce_free(ctx); ctx->copy_threads = (int)value; }
    else if (n == "debug_ntt_fault") {   // test-only fault injection (ntt.hip): flips a bit of a twiddle table — every later proof of the context is wrong
        const char* t = getenv("ZKPOR_TESTING");
        if (!t || strcmp(t, "1") != 0) { ctx->err = "debug_ntt_fault is a test hook: set ZKPOR_TESTING=1 in the environment to enable it"; return ZKPOR_E_ARG; }
        return ntt_debug_fault(ctx, (int)value);
    }
    else if (n == "tail_reserve_cus") {
        if (value < 0 || value > 128 || value % 8) { ctx->err = "tail_reserve_cus must be 0 or a multiple of 8 up to 128"; return ZKPOR_E_ARG; }
        // no stream is destroyed here (common.cuh tail_sets): a value the context has had before gets its old pair back, a new one a new pair at the next
        // prove tail, and a context that has used up its pairs keeps its setting and says so
        bool known = value == 0 && !ctx->tail_streams;
        for (auto& ts : ctx->tail_sets) known |= ts.reserve == (int)value;
        if (!known && ctx->tail_sets.size() >= zkpor_ctx::TAIL_SETS_MAX) {
            ctx->err = "tail_reserve_cus: this context has already created masked streams for " + std::to_string(zkpor_ctx::TAIL_SETS_MAX) + " different values (they live as long as the context); use one of those or another context";
            return ZKPOR_E_STATE;
        }
        ctx->tail_reserve_cus = (int)value;
        ctx->tail_stream = ctx->tail_aux = ctx->tail_chain = nullptr;
        for (auto& ts : ctx->tail_sets) if (ts.reserve == (int)value) { ctx->tail_stream = ts.main; ctx->tail_aux = ts.aux; ctx->tail_chain = ts.chain; }
    }
    else if (n == "tail_digits_early") { if (value < 0 || value > 1) { ctx->err = "tail_digits_early must be 0 or 1"; return ZKPOR_E_ARG; } ctx->tail_digits_early = (int)value; }
    else if (n == "tail_streams") {
        if (value < 0 || value > 1) { ctx->err = "tail_streams must be 0 or 1"; return ZKPOR_E_ARG; }
        ctx->tail_streams = (int)value;
    }
    else if (n == "stream_priority") {
        if (value < 0 || value > 1) { ctx->err = "stream_priority must be 0 or 1"; return ZKPOR_E_ARG; }
        if (!ctx->own_stream) { ctx->err = "stream_priority: the context runs on the caller's stream (zkpor_init): create that stream with the priority you want"; return ZKPOR_E_STATE; }
        if ((int)value != ctx->stream_priority) {
            int lo = 0, hi = 0;
            ZK_HIP(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
            hipStream_t fresh = nullptr;
            if (value == 1 && hi != lo) ZK_HIP(ctx, hipStreamCreateWithPriority(&fresh, hipStreamNonBlocking, hi));
            else ZK_HIP(ctx, hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
            ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            if (ctx->stream_pooled) ctx->retired_own_queue.push_back(ctx->stream); else ctx->retired_streams.push_back(ctx->stream);   // not destroyed while events of the context may name it (common.cuh)
            ctx->stream = fresh;
            ctx->stream_pooled = false;
            ctx->stream_priority = (int)value;
        }
    }
    else if (n == "stream_own_queue") {
        // the context's OWN stream (solver levels, a / b / c, the commitment) on a hardware queue of its own — the default since round 6 (zkpor_init); 0 = an ordinary stream
        if (value < 0 || value > 1) { ctx->err = "stream_own_queue must be 0 or 1"; return ZKPOR_E_ARG; }
        if (!ctx->own_stream) { ctx->err = "stream_own_queue: the context runs on the caller's stream (zkpor_init): create that stream with hipExtStreamCreateWithCUMask"; return ZKPOR_E_STATE; }
        if (value == 1 && !ctx->stream_pooled) {
            hipStream_t fresh = nullptr;
            ZK_TRY(stream_create_own_queue(ctx, &fresh, 0));
            ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->retired_streams.push_back(ctx->stream);      // not destroyed while events of the context may name it (common.cuh)
            ctx->stream = fresh;
            ctx->stream_pooled = true;
        } else if (value == 0 && ctx->stream_pooled) {        // back to an ordinary stream (experiments; the round-5 behaviour)
            hipStream_t fresh = nullptr;
            ZK_HIP(ctx, hipStreamCreateWithFlags(&fresh, hipStreamNonBlocking));
            ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
            ctx->retired_own_queue.push_back(ctx->stream);    // back to the pool with the context, never to hipStreamDestroy
            ctx->stream = fresh;
            ctx->stream_pooled = false;
        }
    }
    else if (n == "tail_aux_masked") { if (value < 0 || value > 1) { ctx->err = "tail_aux_masked must be 0 or 1"; return ZKPOR_E_ARG; } ctx->tail_aux_masked = (int)value; }
    else if (n == "debug_validate") { if (value < 0 || value > 1) { ctx->err = "debug_validate must be 0 or 1"; return ZKPOR_E_ARG; } ctx->debug_validate = (int)value; }
    else if (n == "debug_throw") {   // test-only: an exception raised INSIDE an entry point must come back as an error code (the ABI firewall)
        const char* t = getenv("ZKPOR_TESTING");
        if (!t || strcmp(t, "1") != 0) { ctx->err = "debug_throw is a test hook: set ZKPOR_TESTING=1 in the environment to enable it"; return ZKPOR_E_ARG; }
        if (value == 1) throw std::runtime_error("debug_throw 1");
        if (value == 2) throw std::bad_alloc();
        if (value == 3) throw 42;
    }
    else if (n == "poseidon_out_idx") ctx->pos_out = (int)value;
    else if (n == "poseidon_carry_idx") ctx->pos_carry = (int)value;
    else { ctx->err = "unknown parameter " + n; return ZKPOR_E_ARG; }
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

double zkpor_phase_ms(zkpor_ctx* ctx, const char* name, uint64_t* calls) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !name) return -1.0;
    auto it = ctx->phases.find(name);
    if (it == ctx->phases.end()) { if (calls) *calls = 0; return 0.0; }
    phase_resolve(ctx, it->second);
    if (calls) *calls = it->second.calls;
    return it->second.ms;
} catch (...) { zk::abi_exception("exception in zkpor_phase_ms"); return -1.0; }
int32_t zkpor_stat(zkpor_ctx* ctx, const char* name, uint64_t* value) try {
    if (!ctx || !name || !value) return ZKPOR_E_ARG;
    auto it = ctx->stats.find(name);
    *value = it == ctx->stats.end() ? 0 : it->second;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
void zkpor_phase_reset(zkpor_ctx* ctx) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx) return;
    for (auto& kv : ctx->phases) { phase_resolve(ctx, kv.second); kv.second.ms = 0; kv.second.calls = 0; }
} catch (...) { zk::abi_exception("exception in zkpor_phase_reset"); }

int32_t zkpor_dev_alloc(zkpor_ctx* ctx, size_t bytes, void** out) try {
    if (!ctx || !out) return ZKPOR_E_ARG;
    ZK_ENTER(ctx->device);
    ZK_HIP(ctx, hipMalloc(out, bytes ? bytes : 1));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
int32_t zkpor_dev_free(zkpor_ctx* ctx, void* p) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx) return ZKPOR_E_ARG;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ZK_HIP(ctx, hipFree(p));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
int32_t zkpor_dev_upload(zkpor_ctx* ctx, void* dst, const void* src, size_t bytes) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx) return ZKPOR_E_ARG;
    ZK_TRY(zk::h2d_sync(ctx, dst, src, bytes));      // complete on return: the host buffer is not retained
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
// asynchronous upload: returns once the copy is queued; the host buffer must stay valid (and should be pinned, see
// zkpor_host_register) until zkpor_sync.  Lets the next proof's witness cross PCIe under the current proof's kernels.
int32_t zkpor_dev_upload_async(zkpor_ctx* ctx, void* dst, const void* src, size_t bytes) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx) return ZKPOR_E_ARG;
    ZK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
// page-lock a caller-owned host range (a Go slice's backing array) so uploads from it run at PCIe rate and truly
// asynchronously; the caller unregisters it before freeing the memory
int32_t zkpor_host_register(zkpor_ctx* ctx, void* ptr, size_t bytes) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !ptr || !bytes) return ZKPOR_E_ARG;
    ZK_HIP(ctx, hipHostRegister(ptr, bytes, hipHostRegisterDefault));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
int32_t zkpor_host_unregister(zkpor_ctx* ctx, void* ptr) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !ptr) return ZKPOR_E_ARG;
    ZK_HIP(ctx, hipHostUnregister(ptr));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
int32_t zkpor_dev_download(zkpor_ctx* ctx, void* dst, const void* src, size_t bytes) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx) return ZKPOR_E_ARG;
    ZK_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
int32_t zkpor_dev_fill_fr(zkpor_ctx* ctx, void* d_out, size_t n, uint64_t seed, int kind) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || (!d_out && n)) return ZKPOR_E_ARG;
    if (n == 0) return ZKPOR_OK;
    hipLaunchKernelGGL(k_fill_fr, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (Fr*)d_out, n, seed, kind);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

// ---- generic MSM ----
static void store_jac_g1(const G1XYZZ& r, uint8_t* out) {
    G1Jac j = xyzz_to_jacobian<Fp>(r);
    memcpy(out, &j, 96);
}
static void store_jac_g2(const G2XYZZ& r, uint8_t* out) {
    G2Jac j = xyzz_to_jacobian<Fp2>(r);
    memcpy(out, &j, 192);
}

int32_t zkpor_g1_jac_sum(const uint8_t* parts96, size_t count, uint8_t out_jac[96]) try {
    if ((!parts96 && count) || !out_jac) return ZKPOR_E_ARG;
    G1Jac o;
    jac_sum<Fp>((const G1Jac*)parts96, count, &o);
    memcpy(out_jac, &o, 96);
    return ZKPOR_OK;
} ZK_ABI_CATCH
int32_t zkpor_g2_jac_sum(const uint8_t* parts192, size_t count, uint8_t out_jac[192]) try {
    if ((!parts192 && count) || !out_jac) return ZKPOR_E_ARG;
    G2Jac o;
    jac_sum<Fp2>((const G2Jac*)parts192, count, &o);
    memcpy(out_jac, &o, 192);
    return ZKPOR_OK;
} ZK_ABI_CATCH

int32_t zkpor_msm_g1_dev(zkpor_ctx* ctx, const void* d_points, const void* d_scalars, size_t n, uint8_t out_jac[96]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !out_jac || (n && (!d_points || !d_scalars))) return ZKPOR_E_ARG;
    G1XYZZ r;
    ZK_TRY(msm_dev<Fp>(ctx, (const G1Affine*)d_points, (const Fr*)d_scalars, n, &r));
    store_jac_g1(r, out_jac);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
int32_t zkpor_msm_g2_dev(zkpor_ctx* ctx, const void* d_points, const void* d_scalars, size_t n, uint8_t out_jac[192]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !out_jac || (n && (!d_points || !d_scalars))) return ZKPOR_E_ARG;
    G2XYZZ r;
    ZK_TRY(msm_dev<Fp2>(ctx, (const G2Affine*)d_points, (const Fr*)d_scalars, n, &r));
    store_jac_g2(r, out_jac);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_msm_digits_dev(zkpor_ctx* ctx, const void* d_scalars, size_t n, int tables, const uint8_t* absent0, const uint8_t* absent1, uint32_t* keys_out,
                             uint32_t* vals_out, size_t cap, uint64_t info[8]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !info || (n && !d_scalars) || (cap && (!keys_out || !vals_out))) return ZKPOR_E_ARG;
    if (n >= ((size_t)1 << 27) || tables < 1 || tables > 8) { ctx->err = "msm_digits: at most 2^27 scalars, 1..8 tables"; return ZKPOR_E_ARG; }
    const MsmCfg cfg = msm_cfg(ctx, n ? n : 1, tables);
    size_t sort_temp = 0;
    const size_t nwords = (n + 31) / 32;
    WsPlan extra; extra.add<u32>(nwords + 1); extra.add<u32>(nwords + 1);
    ZK_TRY(ws_reserve(ctx, digits_ws_bytes(ctx, n ? n : 1, cfg, &sort_temp) + extra.total));
    // the absence bitmaps (one BYTE per scalar from the caller, as zkpor_pk_set_consts takes its infinity masks) packed to the device's bit form
    u32* d_abs[2] = {nullptr, nullptr};
    const uint8_t* src[2] = {absent0, absent1};
    for (int g = 0; g < 2; ++g) {
        if (!src[g]) continue;
        std::vector<u32> bits(nwords + 1, 0u);
        for (size_t i = 0; i < n; ++i) if (src[g][i]) bits[i >> 5] |= 1u << (i & 31);
        d_abs[g] = ws_alloc<u32>(ctx, nwords + 1);
        ZK_TRY(h2d_sync(ctx, d_abs[g], bits.data(), (nwords + 1) * sizeof(u32)));
    }
    DigitStream ds;
    StreamFilter filt; filt.absent[0] = d_abs[0]; filt.absent[1] = d_abs[1];
    info[0] = info[1] = info[2] = 0;
    if (n) {
        // the shared stream only (no filter outputs): the sizes of the per-array streams are counted by the decomposition all the same
        size_t cap_n = n * (size_t)cfg.W;
        u32 *k0 = ws_alloc<u32>(ctx, cap_n), *k1 = ws_alloc<u32>(ctx, cap_n), *v0 = ws_alloc<u32>(ctx, cap_n), *v1 = ws_alloc<u32>(ctx, cap_n);
        u32* counter = ws_alloc<u32>(ctx, 64);
        char* temp = ws_alloc<char>(ctx, sort_temp + 256);
        if (!k0 || !k1 || !v0 || !v1 || !counter || !temp) { ctx->err = "msm_digits: workspace too small"; return ZKPOR_E_OOM; }
        u32 Ms[3] = {0, 0, 0};
        ZK_TRY(digit_sort(ctx, (const Fr*)d_scalars, (u32)n, cfg, digit_sort_plan(cfg), k0, v0, k1, v1, counter, temp, d_abs[0], d_abs[1], Ms, &ds.keys, &ds.vals));
        info[0] = Ms[0]; info[1] = Ms[1]; info[2] = Ms[2];
        if (Ms[0] > cap) { ctx->err = "msm_digits: the stream has " + std::to_string(Ms[0]) + " entries, the output buffers hold " + std::to_string(cap); return ZKPOR_E_ARG; }
        if (Ms[0]) {
            ZK_HIP(ctx, hipMemcpyAsync(keys_out, ds.keys, (size_t)Ms[0] * 4, hipMemcpyDeviceToHost, ctx->stream));
            ZK_HIP(ctx, hipMemcpyAsync(vals_out, ds.vals, (size_t)Ms[0] * 4, hipMemcpyDeviceToHost, ctx->stream));
        }
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    info[3] = (uint64_t)cfg.c; info[4] = (uint64_t)cfg.W; info[5] = (uint64_t)cfg.piece; info[6] = cfg.bpw; info[7] = (uint64_t)digit_sort_plan(cfg).nlev;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_msm_g1(zkpor_ctx* ctx, const void* points_affine, const uint64_t* scalars, size_t n, uint8_t out_jac[96]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !out_jac || (n && (!points_affine || !scalars))) return ZKPOR_E_ARG;
    G1XYZZ r;
    ZK_TRY(msm_host<Fp>(ctx, points_affine, scalars, n, &r));
    store_jac_g1(r, out_jac);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
int32_t zkpor_msm_g2(zkpor_ctx* ctx, const void* points_affine, const uint64_t* scalars, size_t n, uint8_t out_jac[192]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !out_jac || (n && (!points_affine || !scalars))) return ZKPOR_E_ARG;
    G2XYZZ r;
    ZK_TRY(msm_host<Fp2>(ctx, points_affine, scalars, n, &r));
    store_jac_g2(r, out_jac);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

}  // extern "C"
