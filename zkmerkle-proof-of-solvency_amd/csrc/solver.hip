// The solver program of a compiled circuit executed on the device — SURVEY.md §8 row f4, the generic half: what the structured generators
// (zkpor_witgen_*) do not produce is solved here, in HBM, instead of on the host.
//
// What it replaces: r1cs.Solve inside groth16.Prove (src/prover/prover/prover.go:269; gnark constraint/bn254/solver.go, 3P).  gnark walks
// `Levels [][]int`, sets of mutually independent instructions, with one goroutine per chunk of a level; here a level is ONE launch with one
// GPU thread per instruction (csrc/solver_instr.cuh holds what a thread does — the same header is unit-tested on the CPU).  The levels of
// BatchCreateUserCircuit are wide where the users sit side by side (hundreds to thousands of instructions) and narrow along the hash chains;
// runs of consecutive narrow levels (<= 512 instructions each) are executed by ONE workgroup that steps through them with a workgroup
// barrier per level, so a run of k narrow levels costs one launch instead of k.
//
// Hints: the native ones run on the device (IntegerDivision, NBits, InvZero, DecomposeHint).  Any other hint (gnark's BSB22 commitment
// placeholder: the challenge wire depends on a multi-exponentiation over the committed wires and a hash-to-field) is EXTERNAL: the run
// pauses in front of it, the caller reads the hint's inputs (zkpor_solver_external_inputs), computes the outputs by whatever means
// (zkpor_commit_dev + the challenge of host/bsb22_challenge.hpp), hands them in (zkpor_solver_external_outputs) and resumes.
#include <algorithm>
#include "common.cuh"
#include "r1cs.cuh"
#include "solver_instr.cuh"
#include "../host/solver_file.hpp"

struct zkpor_solver {
    zkpor_ctx* ctx = nullptr;
    zkpor_r1cs* r1cs = nullptr;                 // borrowed: must outlive the solver
    zkpor_host::SolverView view;                // points into `container`
    std::vector<uint8_t> container;
    std::vector<uint8_t> hint_kind;             // per hint name id
    std::vector<uint8_t> level_external;        // per level: holds at least one external hint
    uint32_t *d_kind = nullptr, *d_arg = nullptr, *d_level_instr = nullptr, *d_calldata = nullptr;
    uint64_t* d_level_ptr = nullptr;
    uint8_t *d_hint_kind = nullptr, *d_known = nullptr;
    uint32_t* d_err = nullptr;                  // [0] first error code, [1] its instruction, [2] wires never assigned, [3] externals met in the level just run
    uint32_t* d_ext = nullptr;                  // external hint instructions of the level just run (capacity EXT_CAP)
    uint64_t n_r1c = 0, n_hint = 0, n_skip = 0;
    // run state (pause / resume)
    bool running = false;
    uint64_t next_level = 0;
    std::vector<uint32_t> pending;              // external instructions of the level just run, still to be served
    uint64_t launches = 0;
    void* d_w = nullptr;
    uint8_t* known = nullptr;                   // the flags of this run: d_known or the caller's
};

namespace zk {
static constexpr u32 NARROW = 512, EXT_CAP = 4096;

ZK_D void solver_step(const SolverProg& P, u32 ins, Fr* w, uint8_t* known, u32* err, u32* ext) {
    // an external hint is not executed: it is reported, its outputs stay unknown until the caller provides them
    if (P.kind[ins] == SI_HINT) {
        const u32 arg = P.arg[ins];
        if ((u64)arg + 3 <= P.n_calldata) {
            const u32 name = P.calldata[arg];
            if (name < P.n_hint_names && P.hint_kind[name] == HK_NONE) {
                const u32 slot = atomicAdd(&err[3], 1u);
                if (slot < EXT_CAP) ext[slot] = ins;
                return;
            }
        }
    }
    const int rc = solve_instr(P, ins, w, known);
    if (rc != SE_OK && atomicCAS(&err[0], 0u, (u32)rc) == 0u) err[1] = ins;
}

// a wide level: one thread per instruction
__global__ __launch_bounds__(256) void k_solve_level(SolverProg P, const u32* __restrict__ level_instr, u64 lo, u32 n, Fr* w, uint8_t* known,
                                                     u32* err, u32* ext) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n || err[0]) return;
    solver_step(P, level_instr[lo + i], w, known, err, ext);
}

// a run of narrow levels [l0, l1): one workgroup, a barrier per level (the writes of a level are visible to the workgroup after it)
__global__ __launch_bounds__(NARROW) void k_solve_narrow(SolverProg P, const u32* __restrict__ level_instr, const u64* __restrict__ level_ptr, u64 l0,
                                                         u64 l1, Fr* w, uint8_t* known, u32* err, u32* ext) {
    for (u64 l = l0; l < l1; ++l) {
        const u64 lo = level_ptr[l];
        const u32 n = (u32)(level_ptr[l + 1] - lo);
        if (threadIdx.x < n && !err[0]) solver_step(P, level_instr[lo + threadIdx.x], w, known, err, ext);
        __threadfence_block();
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_count_unknown(const uint8_t* __restrict__ known, size_t n, u32* err) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    const bool miss = i < n && !known[i];
    const u64 b = __ballot(miss);
    if ((threadIdx.x & 63u) == 0 && b) atomicAdd(&err[2], (u32)__popcll(b));
}

// the input expressions of a hint, evaluated for the caller: out[i] = sum coeff * w over input i (one thread per input; the BSB22
// placeholder's inputs are the committed wires, thousands to millions of one-term expressions)
__global__ __launch_bounds__(256) void k_hint_inputs(SolverProg P, u32 ins, const u64* __restrict__ offs, u32 n_in, const Fr* __restrict__ w,
                                                     const uint8_t* __restrict__ known, Fr* out, u32* err) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_in) return;
    const u32* cd = P.calldata + P.arg[ins];
    u64 p = offs[i];
    const u32 nterms = cd[p++];
    Fr acc = Fr::zero();
    for (u32 k = 0; k < nterms; ++k) {
        const u32 ci = cd[p++], wi = cd[p++];
        if (!known[wi]) { if (atomicCAS(&err[0], 0u, (u32)SE_INPUT_UNSOLVED) == 0u) err[1] = ins; return; }
        si_add_term(acc, P.ckind[ci], P.coeff, ci, w[wi]);
    }
    out[i] = acc;
}
__global__ void k_hint_outputs(SolverProg P, u32 ins, const Fr* __restrict__ vals, Fr* w, uint8_t* known) {
    const u32* cd = P.calldata + P.arg[ins];
    const u32 n_out = cd[2];
    for (u32 i = threadIdx.x; i < n_out; i += blockDim.x) { w[cd[3 + i]] = vals[i]; known[cd[3 + i]] = 1; }
}

static SolverProg prog_of(const zkpor_solver* s) {
    SolverProg P;
    const zkpor_r1cs* r = s->r1cs;
    P.coeff = r->coeff; P.ckind = r->coeff_kind;
    for (int m = 0; m < 3; ++m) { P.row_ptr[m] = r->row_ptr[m]; P.cid[m] = r->cid[m]; P.wid[m] = r->wid[m]; }
    P.n_constraints = (u32)r->n_constraints; P.n_wires = (u32)r->n_wires; P.n_coeff = (u32)r->n_coeff;
    P.kind = s->d_kind; P.arg = s->d_arg; P.calldata = s->d_calldata; P.n_calldata = s->view.n_calldata;
    P.hint_kind = s->d_hint_kind; P.n_hint_names = (u32)s->hint_kind.size();
    return P;
}
static void solver_free(zkpor_solver* s) {
    void* ptrs[] = {s->d_kind, s->d_arg, s->d_level_instr, s->d_calldata, s->d_level_ptr, s->d_hint_kind, s->d_known, s->d_err, s->d_ext};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    delete s;
}
static const char* solver_error_text(u32 code) {
    switch (code) {
    case SE_ROW_RANGE: return "constraint index out of range";
    case SE_TWO_UNKNOWN: return "more than one unknown wire (wrong level order)";
    case SE_NOT_SATISFIED: return "constraint not satisfied";
    case SE_ZERO_COEFF: return "unknown wire with a zero coefficient";
    case SE_DIV_ZERO: return "division by zero";
    case SE_CALLDATA: return "call data out of range";
    case SE_NO_HINT: return "no native implementation for this hint";
    case SE_ID_RANGE: return "wire or coefficient id out of range";
    case SE_INPUT_UNSOLVED: return "hint input not solved yet";
    case SE_HINT_FAILED: return "hint failed";
    default: return "error";
    }
}

// queue levels from s->next_level on until the program ends or a level with external hints has run; then look at the flags
static int32_t solver_advance(zkpor_solver* s, uint32_t* paused_instr) {
    zkpor_ctx* ctx = s->ctx;
    const auto& v = s->view;
    const SolverProg P = prog_of(s);
    Fr* w = (Fr*)s->d_w;
    *paused_instr = 0xffffffffu;
    if (!s->pending.empty()) { *paused_instr = s->pending.front(); return ZKPOR_OK; }
    u32 h[4] = {0, 0, 0, 0};
    while (s->next_level < v.n_levels) {
        PhaseScope ps(ctx, "solver_levels");
        bool stop = false;
        while (s->next_level < v.n_levels && !stop) {
            const u64 l = s->next_level;
            const u64 lo = v.level_ptr[l], n = v.level_ptr[l + 1] - lo;
            if (n > NARROW) {
                hipLaunchKernelGGL(k_solve_level, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, P, s->d_level_instr, lo, (u32)n, w, s->known, s->d_err, s->d_ext);
                s->next_level = l + 1;
                stop = s->level_external[l];
            } else {
                u64 l1 = l;                      // the run of narrow levels starting here, ended by (and including) a level with external hints
                while (l1 < v.n_levels && v.level_ptr[l1 + 1] - v.level_ptr[l1] <= NARROW) { ++l1; if (s->level_external[l1 - 1]) { stop = true; break; } }
                hipLaunchKernelGGL(k_solve_narrow, dim3(1), dim3(NARROW), 0, ctx->stream, P, s->d_level_instr, s->d_level_ptr, l, l1, w, s->known, s->d_err, s->d_ext);
                s->next_level = l1;
            }
            ++s->launches;
        }
        ZK_KERNEL_CHECK(ctx);
        if (stop) break;
    }
    const bool finished = s->next_level >= v.n_levels;
    if (finished) {
        // counted afresh every time the end is reached: a run that paused in its LAST level has been here before, with the hint's outputs still open
        ZK_HIP(ctx, hipMemsetAsync(s->d_err + 2, 0, sizeof(u32), ctx->stream));
        hipLaunchKernelGGL(k_count_unknown, dim3((unsigned)((s->r1cs->n_wires + 255) / 256)), dim3(256), 0, ctx->stream, s->known, s->r1cs->n_wires, s->d_err);
        ZK_KERNEL_CHECK(ctx);
    }
    ZK_HIP(ctx, hipMemcpyAsync(h, s->d_err, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h[0]) {
        s->running = false;
        ctx->err = std::string("solver: ") + solver_error_text(h[0]) + " at instruction " + std::to_string(h[1]);
        return ZKPOR_E_STATE;
    }
    if (h[3]) {   // external hints met in the last level: serve them one by one (a level is small next to what a commitment costs)
        if (h[3] > EXT_CAP) { s->running = false; ctx->err = "solver: more external hints in one level than the executor records"; return ZKPOR_E_STATE; }
        s->pending.resize(h[3]);
        ZK_HIP(ctx, hipMemcpy(s->pending.data(), s->d_ext, h[3] * sizeof(u32), hipMemcpyDeviceToHost));
        std::sort(s->pending.begin(), s->pending.end());
        ZK_HIP(ctx, hipMemsetAsync(s->d_err + 3, 0, sizeof(u32), ctx->stream));
        *paused_instr = s->pending.front();
        return ZKPOR_OK;
    }
    if (finished) {
        s->running = false;
        if (h[2]) { ctx->err = "solver: " + std::to_string(h[2]) + " wires were never assigned"; return ZKPOR_E_STATE; }
    }
    return ZKPOR_OK;
}
zkpor_ctx* solver_ctx(zkpor_solver* s) { return s->ctx; }
zkpor_r1cs* solver_r1cs(zkpor_solver* s) { return s->r1cs; }
}  // namespace zk

using namespace zk;
extern "C" {

int32_t zkpor_solver_create(zkpor_r1cs* r1cs, const uint8_t* container, size_t len, zkpor_solver** out) {
    ZK_ENTER(r1cs ? r1cs->ctx->device : -1);
    if (!r1cs || !container || !out) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = r1cs->ctx;
    for (int m = 0; m < 3; ++m) if (!r1cs->row_ptr[m]) { ctx->err = "solver: the constraint matrices are not loaded"; return ZKPOR_E_STATE; }
    zkpor_solver* s = new zkpor_solver();
    s->ctx = ctx; s->r1cs = r1cs;
    s->container.assign(container, container + len);
    std::string why;
    if (zkpor_host::ParseSolverFile(s->container.data(), len, &s->view, &why) != 0) { ctx->err = why; delete s; return ZKPOR_E_ARG; }
    const auto& v = s->view;
    // validate what the kernels index with: constraint ids, call-data offsets and shapes, wire / coefficient ids inside the call data
    s->hint_kind.resize(v.hint_names.size());
    for (size_t i = 0; i < v.hint_names.size(); ++i) s->hint_kind[i] = hint_kind_of_name(v.hint_names[i].c_str());
    std::vector<uint8_t> instr_external(v.n_instructions, 0);
    for (uint64_t i = 0; i < v.n_instructions; ++i) {
        const uint32_t kind = v.kind[i], arg = v.arg[i];
        if (kind == SI_R1C) { if (arg >= r1cs->n_constraints) { ctx->err = "solver: instruction " + std::to_string(i) + " names a constraint outside the system"; delete s; return ZKPOR_E_ARG; } ++s->n_r1c; }
        else if (kind == SI_HINT) {
            bool ok = (uint64_t)arg + 3 <= v.n_calldata;
            uint64_t p = 0;
            if (ok) {
                const uint32_t* cd = v.calldata + arg;
                ok = cd[0] < v.hint_names.size() && (uint64_t)arg + 3 + cd[2] <= v.n_calldata;
                p = 3 + (uint64_t)(ok ? cd[2] : 0);
                for (uint32_t k = 0; ok && k < cd[2]; ++k) ok = cd[3 + k] < r1cs->n_wires;
                for (uint32_t k = 0; ok && k < cd[1]; ++k) {
                    ok = arg + p < v.n_calldata;
                    if (!ok) break;
                    const uint32_t nt = cd[p++];
                    ok = arg + p + 2ull * nt <= v.n_calldata;
                    for (uint32_t t = 0; ok && t < nt; ++t) { ok = cd[p] < r1cs->n_coeff && cd[p + 1] < r1cs->n_wires; p += 2; }
                }
                if (ok && s->hint_kind[cd[0]] == HK_NONE) instr_external[i] = 1;
            }
            if (!ok) { ctx->err = "solver: the call data of instruction " + std::to_string(i) + " is malformed"; delete s; return ZKPOR_E_ARG; }
            ++s->n_hint;
        } else ++s->n_skip;
    }
    s->level_external.assign(v.n_levels, 0);
    for (uint64_t l = 0; l < v.n_levels; ++l)
        for (uint64_t k = v.level_ptr[l]; k < v.level_ptr[l + 1]; ++k) if (instr_external[v.level_instr[k]]) s->level_external[l] = 1;
    const uint64_t n_li = v.level_ptr[v.n_levels];
    auto up = [&](void** d, const void* h, size_t bytes) {
        if (hipMalloc(d, bytes ? bytes : 4) != hipSuccess) { (void)hipGetLastError(); return false; }
        return bytes == 0 || hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    bool ok = up((void**)&s->d_kind, v.kind, v.n_instructions * 4) && up((void**)&s->d_arg, v.arg, v.n_instructions * 4) &&
              up((void**)&s->d_level_instr, v.level_instr, n_li * 4) && up((void**)&s->d_calldata, v.calldata, v.n_calldata * 4) &&
              up((void**)&s->d_level_ptr, v.level_ptr, (v.n_levels + 1) * 8) && up((void**)&s->d_hint_kind, s->hint_kind.data(), s->hint_kind.size()) &&
              hipMalloc((void**)&s->d_known, r1cs->n_wires) == hipSuccess && hipMalloc((void**)&s->d_err, 16) == hipSuccess &&
              hipMalloc((void**)&s->d_ext, EXT_CAP * sizeof(u32)) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); solver_free(s); ctx->err = "solver: out of device memory"; return ZKPOR_E_OOM; }
    *out = s;
    return ZKPOR_OK;
}

void zkpor_solver_destroy(zkpor_solver* s) {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s) return;
    (void)hipStreamSynchronize(s->ctx->stream);
    solver_free(s);
}

int32_t zkpor_solver_dims(const zkpor_solver* s, uint64_t dims[7]) {
    if (!s || !dims) return ZKPOR_E_ARG;
    dims[0] = s->view.n_instructions; dims[1] = s->view.n_levels; dims[2] = s->n_r1c; dims[3] = s->n_hint; dims[4] = s->n_skip;
    uint64_t ext = 0;
    for (uint8_t e : s->level_external) ext += e;
    dims[5] = ext;
    dims[6] = s->launches;
    return ZKPOR_OK;
}

int32_t zkpor_solver_start_dev(zkpor_solver* s, void* d_w, size_t n_inputs, uint8_t* d_known_or_null, uint32_t* paused_instr) {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s || !d_w || !paused_instr) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = s->ctx;
    if (n_inputs == 0 || n_inputs > s->r1cs->n_wires) { ctx->err = "solver: the assignment must hold 1 + nPublic + nSecret elements"; return ZKPOR_E_ARG; }
    s->d_w = d_w;
    s->known = d_known_or_null ? d_known_or_null : s->d_known;
    if (!d_known_or_null) ZK_HIP(ctx, hipMemsetAsync(s->d_known, 0, s->r1cs->n_wires, ctx->stream));
    ZK_HIP(ctx, hipMemsetAsync(s->known, 1, n_inputs, ctx->stream));
    ZK_HIP(ctx, hipMemsetAsync(s->d_err, 0, 16, ctx->stream));
    s->running = true; s->next_level = 0; s->pending.clear(); s->launches = 0;
    return solver_advance(s, paused_instr);
}

int32_t zkpor_solver_resume_dev(zkpor_solver* s, uint32_t* paused_instr) {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s || !paused_instr) return ZKPOR_E_ARG;
    if (!s->running) { s->ctx->err = "solver: no run to resume"; return ZKPOR_E_STATE; }
    return solver_advance(s, paused_instr);
}

// evaluates the inputs of the external hint the run is paused at into d_out (device, n_in elements); synchronous
static int32_t hint_inputs_to(zkpor_solver* s, uint32_t instr, Fr* d_out) {
    zkpor_ctx* ctx = s->ctx;
    const uint32_t* cd = s->view.calldata + s->view.arg[instr];
    const uint32_t n_in = cd[1];
    if (n_in == 0) return ZKPOR_OK;
    std::vector<uint64_t> offs(n_in);
    uint64_t p = 3 + (uint64_t)cd[2];
    for (uint32_t i = 0; i < n_in; ++i) { offs[i] = p; p += 1 + 2ull * cd[p]; }   // shapes were validated by zkpor_solver_create
    uint64_t* d_offs = nullptr;
    ZK_HIP(ctx, hipMalloc((void**)&d_offs, n_in * sizeof(uint64_t)));
    int32_t rc = ZKPOR_OK;
    u32 h[2] = {0, 0};
    if (hipMemcpyAsync(d_offs, offs.data(), n_in * sizeof(uint64_t), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "solver: H2D failed"; rc = ZKPOR_E_HIP; }
    if (rc == ZKPOR_OK) {
        hipLaunchKernelGGL(k_hint_inputs, dim3((n_in + 255u) / 256u), dim3(256), 0, ctx->stream, prog_of(s), instr, d_offs, n_in, (const Fr*)s->d_w, s->known, d_out, s->d_err);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(h, s->d_err, sizeof h, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "solver: launch failed"; rc = ZKPOR_E_HIP; }
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(d_offs);
    if (rc == ZKPOR_OK && h[0]) { s->running = false; ctx->err = std::string("solver: ") + solver_error_text(h[0]) + " at instruction " + std::to_string(h[1]); rc = ZKPOR_E_STATE; }
    return rc;
}

int32_t zkpor_solver_external_inputs(zkpor_solver* s, uint32_t instr, uint64_t* in_values, size_t capacity, size_t* n_in, size_t* n_out) {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s || !n_in || !n_out) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = s->ctx;
    if (!s->running || s->pending.empty() || s->pending.front() != instr) { ctx->err = "solver: not paused at this instruction"; return ZKPOR_E_STATE; }
    const uint32_t* cd = s->view.calldata + s->view.arg[instr];
    *n_in = cd[1]; *n_out = cd[2];
    if (!in_values) return ZKPOR_OK;              // sizes only
    if (capacity < cd[1]) { ctx->err = "solver: the buffer holds fewer elements than the hint has inputs"; return ZKPOR_E_ARG; }
    if (cd[1] == 0) return ZKPOR_OK;
    Fr* d_tmp = nullptr;                          // not the staging area: the caller's d_w may live there (zkpor_prove_inputs)
    ZK_HIP(ctx, hipMalloc((void**)&d_tmp, (size_t)cd[1] * sizeof(Fr)));
    int32_t rc = hint_inputs_to(s, instr, d_tmp);
    if (rc == ZKPOR_OK && hipMemcpy(in_values, d_tmp, (size_t)cd[1] * sizeof(Fr), hipMemcpyDeviceToHost) != hipSuccess) { ctx->err = "solver: D2H failed"; rc = ZKPOR_E_HIP; }
    (void)hipFree(d_tmp);
    return rc;
}

/* the same into device memory (d_out: capacity elements): the committed wires of a BSB22 commitment go straight to zkpor_commit_dev */
int32_t zkpor_solver_external_inputs_dev(zkpor_solver* s, uint32_t instr, void* d_out, size_t capacity) {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s || !d_out) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = s->ctx;
    if (!s->running || s->pending.empty() || s->pending.front() != instr) { ctx->err = "solver: not paused at this instruction"; return ZKPOR_E_STATE; }
    const uint32_t* cd = s->view.calldata + s->view.arg[instr];
    if (capacity < cd[1]) { ctx->err = "solver: the buffer holds fewer elements than the hint has inputs"; return ZKPOR_E_ARG; }
    return hint_inputs_to(s, instr, (Fr*)d_out);
}

int32_t zkpor_solver_external_outputs(zkpor_solver* s, uint32_t instr, const uint64_t* out_values, size_t n_out) {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s || (!out_values && n_out)) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = s->ctx;
    if (!s->running || s->pending.empty() || s->pending.front() != instr) { ctx->err = "solver: not paused at this instruction"; return ZKPOR_E_STATE; }
    const uint32_t* cd = s->view.calldata + s->view.arg[instr];
    if (n_out != cd[2]) { ctx->err = "solver: the hint has " + std::to_string(cd[2]) + " outputs"; return ZKPOR_E_ARG; }
    if (n_out) {
        Fr* d_tmp = nullptr;
        ZK_HIP(ctx, hipMalloc((void**)&d_tmp, n_out * sizeof(Fr)));
        int32_t rc = ZKPOR_OK;
        if (hipMemcpy(d_tmp, out_values, n_out * sizeof(Fr), hipMemcpyHostToDevice) != hipSuccess) { ctx->err = "solver: H2D failed"; rc = ZKPOR_E_HIP; }
        if (rc == ZKPOR_OK) {
            hipLaunchKernelGGL(k_hint_outputs, dim3(1), dim3(256), 0, ctx->stream, prog_of(s), instr, (const Fr*)d_tmp, (Fr*)s->d_w, s->known);
            if (hipGetLastError() != hipSuccess) { ctx->err = "solver: launch failed"; rc = ZKPOR_E_HIP; }
        }
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(d_tmp);
        if (rc != ZKPOR_OK) return rc;
    }
    s->pending.erase(s->pending.begin());
    return ZKPOR_OK;
}

/* host-buffer form: inputs in, full wire vector out; pre-filled wires (the device generators' in a real run) given as (id, value) pairs;
 * external hints are NOT served here (the call fails with ZKPOR_E_STATE when it meets one) */
int32_t zkpor_solver_run(zkpor_solver* s, const uint64_t* inputs, size_t n_inputs, const uint32_t* pre_ids, const uint64_t* pre_vals, size_t n_pre,
                         uint64_t* w_out, uint64_t stats[4]) {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s || !inputs || !w_out || (n_pre && (!pre_ids || !pre_vals))) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = s->ctx;
    const size_t nw = s->r1cs->n_wires;
    if (n_inputs == 0 || n_inputs > nw) { ctx->err = "solver: the assignment must hold 1 + nPublic + nSecret elements"; return ZKPOR_E_ARG; }
    std::vector<uint64_t> hw(nw * 4, 0);
    std::vector<uint8_t> hk(nw, 0);
    memcpy(hw.data(), inputs, n_inputs * 32);
    for (size_t i = 0; i < n_pre; ++i) {
        if (pre_ids[i] >= nw) { ctx->err = "solver: prefilled wire out of range"; return ZKPOR_E_ARG; }
        memcpy(&hw[4 * (size_t)pre_ids[i]], pre_vals + 4 * i, 32); hk[pre_ids[i]] = 1;
    }
    Fr* d_w = nullptr; uint8_t* d_k = nullptr;
    ZK_HIP(ctx, hipMalloc((void**)&d_w, nw * sizeof(Fr)));
    if (hipMalloc((void**)&d_k, nw) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(d_w); ctx->err = "solver: out of device memory"; return ZKPOR_E_OOM; }
    int32_t rc = ZKPOR_OK;
    uint32_t paused = 0xffffffffu;
    if (hipMemcpy(d_w, hw.data(), nw * sizeof(Fr), hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(d_k, hk.data(), nw, hipMemcpyHostToDevice) != hipSuccess) { ctx->err = "solver: H2D failed"; rc = ZKPOR_E_HIP; }
    if (rc == ZKPOR_OK) rc = zkpor_solver_start_dev(s, d_w, n_inputs, d_k, &paused);
    if (rc == ZKPOR_OK && paused != 0xffffffffu) { s->running = false; ctx->err = "solver: external hint at instruction " + std::to_string(paused) + " (serve it through zkpor_solver_start_dev / _external_* / _resume_dev)"; rc = ZKPOR_E_STATE; }
    if (rc == ZKPOR_OK && hipMemcpy(w_out, d_w, nw * sizeof(Fr), hipMemcpyDeviceToHost) != hipSuccess) { ctx->err = "solver: D2H failed"; rc = ZKPOR_E_HIP; }
    if (stats) { stats[0] = s->n_r1c; stats[1] = s->n_hint; stats[2] = s->n_skip; stats[3] = s->launches; }
    (void)hipFree(d_w); (void)hipFree(d_k);
    return rc;
}

}  // extern "C"
