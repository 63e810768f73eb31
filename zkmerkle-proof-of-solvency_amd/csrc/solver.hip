// The solver program of a compiled circuit executed on the device — SURVEY.md §8 row f4: r1cs.Solve without the host.
//
// What it replaces: r1cs.Solve inside groth16.Prove (src/prover/prover/prover.go:269; gnark constraint/bn254/solver.go, 3P).  gnark walks
// `Levels [][]int`, sets of mutually independent instructions, with one goroutine per chunk of a level.  Here the program sits in HBM next to
// the constraint matrices and a level runs as a handful of launches, one per CLASS of instruction:
//   generic    one GPU thread per instruction — solve one constraint for its single unknown wire, a native hint (circuit.IntegerDivision,
//              NBits, InvZero, DecomposeHint) or a table lookup (gnark BlueprintLookupHint): csrc/solver_instr.cuh, the header the CPU suite
//              also compiles.  Runs of consecutive narrow levels (<= 512 generic instructions, nothing else) are stepped through by ONE
//              workgroup with a barrier per level, so k narrow levels cost one launch instead of k;
//   poseidon   a whole poseidon.Poseidon(...) call per thread (csrc/poseidon.hip k_gadget_poseidon).  A call flagged ASYNC — the two
//              10 000-element CEX commitments: 834 chained permutations whose digest only feeds an assertion — runs on a side stream
//              beside all other levels and is joined in front of the last level;
//   count      gnark's logderivarg countHint (one hint over EVERY query of a table — 10^7 inputs for the range checker): three kernels,
//              one thread per table row / query / output, atomics into a histogram.
// The BSB22 commitment placeholder (and any other hint without native semantics) is EXTERNAL: the run pauses in front of it, the caller
// reads its inputs (zkpor_solver_external_inputs[_dev] — the committed wires go straight to zkpor_commit_dev), provides the output
// (the challenge of host/bsb22_challenge.hpp) and resumes.
#include <algorithm>
#include <set>
#include "common.cuh"
#include <memory>
#include "r1cs.cuh"
#include "solver_instr.cuh"
#include "../host/solver_file.hpp"

namespace zk {
int32_t gadget_poseidon_launch(zkpor_ctx* ctx, hipStream_t stream, const SolverProg& P, const u32* d_instr, u32 n, Fr* w, uint8_t* known, u32* d_err,
                               const Fr* d_pre, const u32* d_pre_off, Fr* d_a, Fr* d_b, Fr* d_c);   // poseidon.hip
}

namespace {
struct LevelPlan {          // one level of the reordered instruction list: [generic | poseidon | poseidon async | count]
    uint64_t lo = 0;
    uint32_t n_posj = 0;                         // ASYNC Poseidon calls with a join level (solver_file.hpp POSEIDON_JOIN_SHIFT): listed behind the ASYNC ones
    uint32_t n_posj_seen = 0;
    uint64_t join_level = 0;                     // ... the lowest of theirs: the level the side stream is joined in front of
    uint32_t n_chk = 0;                          // CHECK instructions (assertions), listed right behind the generic ones: left out when the caller checks every row itself
    uint32_t n_gen = 0, n_pos = 0, n_posa = 0, n_cnt = 0, cnt_first = 0;   // the level's count hints are counts[cnt_first .. + n_cnt)
    uint32_t posa_first = 0;                                               // its ASYNC calls are asyncs[posa_first .. + n_posa), in list order
    uint32_t posj_first = 0;                                               // its calls with a join level are joins[posj_first .. + n_posj)
    uint64_t cnt_rows = 0, cnt_queries = 0;                                // table rows / queries of all its count hints
    bool external = false;
};
struct BigHint {            // a hint whose inputs are addressed through a persisted offset table (count hints, external hints)
    uint32_t ins = 0, n_in = 0, n_out = 0, nb_table = 0, nb_col = 0;
    uint64_t offs_base = 0; // first entry in d_offs: word offset of input i's expression from the instruction's call data
    uint64_t nb_q = 0;
};
}  // namespace

struct zkpor_solver {
    zkpor_ctx* ctx = nullptr;
    zkpor_r1cs* r1cs = nullptr;                 // borrowed: must outlive the solver
    zkpor_host::SolverView view;                // points into `container`
    std::vector<uint8_t> container;
    std::vector<uint8_t> hint_kind;             // per hint name id
    std::vector<LevelPlan> plan;
    std::vector<BigHint> counts;                // in level order (the order the levels meet them)
    std::vector<BigHint> asyncs;                // the ASYNC Poseidon calls, in level order; nb_q = first element of the call's inputs in d_pre
    uint32_t* d_pre_off = nullptr;              // per ASYNC call (same order): that first element
    std::vector<BigHint> joins;                 // the Poseidon calls with a join level (the challenge sponge), in level order: their inputs are pre-evaluated the same way ("solver_pre_join")
    uint32_t* d_prej_off = nullptr;             // per such call: the first element of its inputs in d_pre
    zk::Fr* d_pre = nullptr;                    // their inputs, evaluated all at once in front of the (serial) sponge
    std::map<uint32_t, BigHint> externals;      // by instruction
    uint32_t *d_kind = nullptr, *d_arg = nullptr, *d_level_instr = nullptr, *d_calldata = nullptr, *d_gen_cnt = nullptr, *d_offs = nullptr;
    uint64_t* d_gen_lo = nullptr;
    uint8_t *d_hint_kind = nullptr, *d_known = nullptr;
    uint32_t* d_err = nullptr;                  // [0] first error code, [1] its instruction, [2] wires never assigned, [3] externals met in the level just run
    uint32_t* d_ext = nullptr;                  // external hint instructions of the level just run (capacity ext_cap)
    uint32_t ext_cap = 0;                       // = max(EXT_CAP, external hints of the whole program): a level can never report more than fit
    uint32_t* d_cnt = nullptr;                  // histograms of the count hints of the level being served (the largest level's table rows)
    void* d_cmeta = nullptr;                    // zk::CountDev per count hint, level order
    zk::Fr* d_tmp = nullptr;                    // scratch for external hint values (grow-only)
    size_t tmp_cap = 0;
    hipStream_t side = nullptr;                 // ASYNC instructions
    uint32_t* d_rows = nullptr;                 // one bit per constraint: a, b, c of the row are written by a (non-ASYNC) Poseidon instruction when abc is set
    uint64_t rows_covered = 0;
    zk::Fr *abc_a = nullptr, *abc_b = nullptr, *abc_c = nullptr;   // zkpor_solver_set_abc_dev
    bool abc_written = false;                   // the run that just finished wrote those rows (cooperative kernel, abc set)
    bool checks_left = false;                   // the run (being) made leaves the CHECK instructions to zkpor_solver_eval_abc_dev (abc set, solver_defer_checks)
    uint64_t n_check = 0;
    uint32_t* d_long = nullptr;                 // [0] count, [1..] the long constraints of the level being run (k_solve_long)
    uint32_t* d_gen_cnt_all = nullptr;          // generic + CHECK instructions per level (d_gen_cnt: generic only)
    uint32_t* d_perr = nullptr;                 // error words of a prefetch (its kernels run beside another run's)
    uint8_t* d_ones = nullptr;                  // n_wires bytes of 1: the `known` flags a prefetch reads (it only reads inputs)
    void* prefetched_w = nullptr;               // the wire vector whose ASYNC instructions are already running / done on the side stream
    bool skip_async = false;                    // this run consumes a prefetch
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipStream_t side2 = nullptr;                // calls with a join level (they must not queue behind the ASYNC chains of `side`)
    hipEvent_t ev_fork2 = nullptr, ev_join2 = nullptr;
    bool side2_busy = false;
    uint64_t join2_level = 0;
    uint64_t n_r1c = 0, n_hint = 0, n_skip = 0, n_lookup = 0, n_poseidon = 0;
    uint64_t async_max_wire = 0;                // the highest wire an ASYNC call's input expressions read: a prefetch is legal only if it is an input wire
    bool side_own_queue = false;                // the side streams were re-created with hardware queues of their own (several workers per GPU)
    bool run_ok = false;                        // the last run reached its end without an error: its d_w (and the rows it wrote) may be proved over
    // run state (pause / resume)
    bool running = false, side_busy = false;
    uint64_t next_level = 0;
    std::vector<uint32_t> pending;              // external instructions of the level just run, still to be served
    uint64_t launches = 0;
    void* d_w = nullptr;
    uint8_t* known = nullptr;                   // the flags of this run: d_known or the caller's
};

namespace zk {
static constexpr u32 NARROW = 512, EXT_CAP = 4096;
static constexpr int BATCH_K = 4;                 // instructions per thread of the batched level kernel
static constexpr u64 CHAIN_FROM = 16;             // runs of this many one-instruction levels go to k_solve_chain ("solver_chain" 0: never)
static constexpr u32 LONG_CAP = 1u << 20;         // constraints per level a wave each takes over (k_solve_long): beyond it they stay in their thread

ZK_D void solver_step(const SolverProg& P, u32 ins, Fr* w, uint8_t* known, u32* err, u32* ext) {
    // an external hint is not executed: it is reported, its outputs stay unknown until the caller provides them
    if (P.kind[ins] == SI_HINT) {
        const u32 name = P.calldata[P.arg[ins]];
        if (P.hint_kind[name] == HK_NONE) {
            const u32 slot = atomicAdd(&err[3], 1u);
            if (slot < P.ext_cap) ext[slot] = ins;
            return;
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
// a level whose divisions share inversions: KB instructions per thread (strided, so that neighbouring lanes still read neighbouring list
// entries), and the 256 threads of a workgroup share ONE field inversion — Montgomery's trick twice over: a thread multiplies its own
// denominators up, the workgroup multiplies the threads' products up a binary tree in LDS, one lane inverts the root (a ~40 k-instruction
// binary Euclid), and the inverse comes down the tree (inverse of a child = inverse of the parent x the sibling) and down the thread's list:
// 3 products per quotient + 3 per thread.  The log-derivative argument behind every range check and lookup is one inverse wire per query
// (2 x 10^7 per zkpor50_1380 proof, all in one level: one inversion per thread was 41 ms of the proof, per workgroup of 1 024 it is ~5).
// A workgroup without a division leaves before the tree.
template <int KB>
__global__ __launch_bounds__(256) void k_solve_level_batched(SolverProg P, const u32* __restrict__ level_instr, u64 lo, u32 n, u32 stride, Fr* w, uint8_t* known,
                                                             u32* err, u32* ext, u32* long_cnt, u32* long_list, u32 long_from) {
    __shared__ Fr tree[512];                      // node i: the product of its leaves; children 2 i and 2 i + 1, leaves 256 ..
    const u32 tid = threadIdx.x, t = blockIdx.x * 256u + tid;
    SiPending pd[KB];
    int np = 0;
    if (t < stride && !err[0]) {
#pragma unroll 1
        for (int k = 0; k < KB; ++k) {
            const u32 i = t + (u32)k * stride;
            if (i >= n) break;
            const u32 ins = level_instr[lo + i];
            if (P.kind[ins] == SI_HINT && P.hint_kind[P.calldata[P.arg[ins]]] == HK_NONE) {
                const u32 slot = atomicAdd(&err[3], 1u);
                if (slot < P.ext_cap) ext[slot] = ins;
                continue;
            }
            if (long_list && P.kind[ins] == SI_R1C && P.arg[ins] < P.n_constraints) {   // a long constraint: one thread would walk its terms alone
                const u32 row = P.arg[ins];
                const u64 nt = (P.row_ptr[0][row + 1] - P.row_ptr[0][row]) + (P.row_ptr[1][row + 1] - P.row_ptr[1][row]) + (P.row_ptr[2][row + 1] - P.row_ptr[2][row]);
                if (nt > (u64)long_from) {
                    const u32 slot = atomicAdd(long_cnt, 1u);
                    if (slot < LONG_CAP) { long_list[slot] = ins; continue; }
                }
            }
            const int rc = solve_instr(P, ins, w, known, &pd[np]);
            if (rc == SE_DEFERRED) ++np;
            else if (rc != SE_OK && atomicCAS(&err[0], 0u, (u32)rc) == 0u) err[1] = ins;
        }
    }
    if (!__syncthreads_or(np)) return;
    Fr pre[KB];                                   // pre[j] = den_0 .. den_j
    if (np) {
        pre[0] = pd[0].den;
        for (int j = 1; j < np; ++j) pre[j] = Fr::mul(pre[j - 1], pd[j].den);
    }
    tree[256u + tid] = np ? pre[np - 1] : Fr::one();
    __syncthreads();
    for (u32 sz = 128u; sz >= 1u; sz >>= 1) {
        if (tid < sz) tree[sz + tid] = Fr::mul(tree[2u * (sz + tid)], tree[2u * (sz + tid) + 1u]);
        __syncthreads();
    }
    if (tid == 0) tree[1] = fr_inverse(tree[1]);
    __syncthreads();
    for (u32 sz = 1u; sz <= 128u; sz <<= 1) {
        if (tid < sz) {
            const u32 nd = sz + tid;
            const Fr iv = tree[nd], c0 = tree[2u * nd], c1 = tree[2u * nd + 1u];
            tree[2u * nd] = Fr::mul(iv, c1);
            tree[2u * nd + 1u] = Fr::mul(iv, c0);
        }
        __syncthreads();
    }
    if (np == 0) return;
    Fr inv = tree[256u + tid];
    for (int j = np - 1; j >= 0; --j) {
        const Fr dinv = j ? Fr::mul(inv, pre[j - 1]) : inv;
        inv = Fr::mul(inv, pd[j].den);
        w[pd[j].wire] = Fr::mul(pd[j].num, dinv);
        known[pd[j].wire] = 1;
    }
}

// the constraints k_solve_level_batched left aside (more than `solver_long` terms, 256 by default: the sums behind the log-derivative arguments are 1 024 terms
// each, 18 000 of them in two levels — 9.6 ms as one thread's walk each): a wave per constraint, lane l takes terms l, l + 64, ..., the known
// parts of L, R, O and what the lanes saw of the open wire meet through shuffles, lane 0 finishes as solve_instr would
__global__ __launch_bounds__(256) void k_solve_long(SolverProg P, const u32* __restrict__ long_cnt, const u32* __restrict__ long_list, Fr* w, uint8_t* known, u32* err) {
    const u32 n = min(long_cnt[0], LONG_CAP);
    const u32 lane = threadIdx.x & 63u, nwaves = gridDim.x * 4u;
    for (u32 i = blockIdx.x * 4u + (threadIdx.x >> 6); i < n; i += nwaves) {
        const u32 ins = long_list[i], row = P.arg[ins];
        Fr v[3], uc = Fr::zero();
        int which = -1;
        u32 x = 0, two = 0;
        for (int m = 0; m < 3; ++m) {
            Fr acc = Fr::zero();
            for (u64 t = P.row_ptr[m][row] + lane; t < P.row_ptr[m][row + 1]; t += 64u) {
                const u32 wi = P.wid[m][t], ci = P.cid[m][t];
                if (known[wi]) si_add_term(acc, P.ckind[ci], P.coeff, ci, w[wi]);
                else {
                    if (which >= 0 && (which != m || x != wi)) two = 1u;
                    uc = which < 0 ? P.coeff[ci] : Fr::add(uc, P.coeff[ci]);
                    which = m; x = wi;
                }
            }
            for (int off = 32; off >= 1; off >>= 1) {
                Fr o;
                for (int k = 0; k < 8; ++k) o.v[k] = (u32)__shfl_down((int)acc.v[k], off, 64);
                acc = Fr::add(acc, o);
            }
            v[m] = acc;
        }
        for (int off = 32; off >= 1; off >>= 1) {     // the open wire as the lanes saw it: one wire, its coefficients summed
            const int ow = __shfl_down(which, off, 64);
            const u32 ox = (u32)__shfl_down((int)x, off, 64), ot = (u32)__shfl_down((int)two, off, 64);
            Fr ouc;
            for (int k = 0; k < 8; ++k) ouc.v[k] = (u32)__shfl_down((int)uc.v[k], off, 64);
            if (lane + (u32)off < 64u) {
                two |= ot;
                if (ow >= 0) {
                    if (which >= 0 && (which != ow || x != ox)) two = 1u;
                    uc = which < 0 ? ouc : Fr::add(uc, ouc);
                    which = ow; x = ox;
                }
            }
        }
        if (lane == 0) {
            const int rc = two ? (int)SE_TWO_UNKNOWN : si_r1c_finish(v, which, x, uc, w, known, nullptr);
            if (rc != SE_OK && atomicCAS(&err[0], 0u, (u32)rc) == 0u) err[1] = ins;
        }
    }
}

// a run of narrow levels [l0, l1): one workgroup, a barrier per level (the writes of a level are visible to the workgroup after it)
__global__ __launch_bounds__(NARROW) void k_solve_narrow(SolverProg P, const u32* __restrict__ level_instr, const u64* __restrict__ gen_lo,
                                                         const u32* __restrict__ gen_cnt, u64 l0, u64 l1, Fr* w, uint8_t* known, u32* err, u32* ext) {
    for (u64 l = l0; l < l1; ++l) {
        const u64 lo = gen_lo[l];
        const u32 n = gen_cnt[l];
        if (threadIdx.x < n && !err[0]) solver_step(P, level_instr[lo + threadIdx.x], w, known, err, ext);
        __threadfence_block();
        __syncthreads();
    }
}

// A CHAIN: a run of levels of one instruction each (the 2 499 successive powers of the RLC challenge, batch_create_user_circuit.go:286-289).
// In k_solve_narrow a level of the chain is a barrier plus six dependent trips to memory — instruction id, kind, row, terms, coefficient,
// the operand the previous level stored — 5.9 us each, 14.5 ms of an otherwise idle GPU per proof.  Here one workgroup takes 256 levels at a
// time: every thread DECODES one level's instruction (row, terms, coefficients, and the values of the operands that are known by then) side
// by side; then the levels run in order, one lane after the other inside a wave, one wave after the other, and the value a level produces
// travels to the next one through a register broadcast instead of through memory.  What is left per level is the field arithmetic.
// A level that does not fit the short form (not a constraint, more than two terms in an expression, an operand produced earlier in the
// same 256 by a level other than its predecessor) runs through the generic solver_step on what is in memory, and so does the rest of its 256.
struct ChOpen { u32 wid, m, ck; Fr co; };                            // a term whose wire was not assigned when the round was decoded
ZK_D u32 chain_slot(u32 x) { return (x * 2654435761u) >> 22; }      // 1 024 slots
ZK_D u32 lane_value(u32 v, u32 k) { return (u32)__builtin_amdgcn_readlane((int)v, (int)__builtin_amdgcn_readfirstlane((int)k)); }
__global__ __launch_bounds__(256) void k_solve_chain(SolverProg P, const u32* __restrict__ level_instr, const u64* __restrict__ gen_lo, u64 l0, u64 l1, Fr* w,
                                                     uint8_t* known, u32* err, u32* ext) {
    __shared__ u32 produced[1024];                // the wires this round's levels have assigned so far (open addressing)
    __shared__ u32 carry_w, carry_dirty;
    __shared__ Fr carry_v;
    const u32 tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    if (tid == 0) { carry_w = 0xffffffffu; carry_dirty = 0u; carry_v = Fr::zero(); }
    for (u64 base = l0; base < l1; base += 256u) {
        for (u32 i = tid; i < 1024u; i += 256u) produced[i] = 0xffffffffu;
        if (tid == 0) carry_dirty = 0u;           // memory is complete behind the barrier at the end of a round
        // ---- decode, all levels of the round side by side: the known part of L, R, O summed, the (at most two) open terms kept
        const u64 l = base + tid;
        const bool have = l < l1 && !err[0];
        u32 ins = 0, n_open = 0;
        bool simple = false;
        Fr a0 = Fr::zero(), a1 = Fr::zero(), a2 = Fr::zero();
        ChOpen o0, o1;
        o0.wid = o1.wid = 0xffffffffu; o0.m = o1.m = 0u; o0.ck = o1.ck = 3u; o0.co = o1.co = Fr::zero();
        if (have) {
            ins = level_instr[gen_lo[l]];
            if (P.kind[ins] == SI_R1C && P.arg[ins] < P.n_constraints) {
                const u32 row = P.arg[ins];
                simple = true;
                for (u32 m = 0; m < 3u && simple; ++m) {
                    const u64 t0 = P.row_ptr[m][row], t1 = P.row_ptr[m][row + 1];
                    if (t1 - t0 > 16u) { simple = false; break; }
                    Fr acc = Fr::zero();
                    for (u64 t = t0; t < t1; ++t) {
                        const u32 wi = P.wid[m][t], ci = P.cid[m][t];
                        if (known[wi]) si_add_term(acc, P.ckind[ci], P.coeff, ci, w[wi]);
                        else if (n_open == 2u) { simple = false; break; }
                        else {
                            ChOpen& o = n_open ? o1 : o0;
                            o.wid = wi; o.m = m; o.ck = P.ckind[ci]; o.co = P.coeff[ci];
                            ++n_open;
                        }
                    }
                    if (m == 0u) a0 = acc; else if (m == 1u) a1 = acc; else a2 = acc;
                }
            }
        }
        __syncthreads();
        // ---- the levels in order
        for (u32 wv = 0; wv < 4u; ++wv) {
            if (wave == wv) {
                u32 lastw = carry_w, dirty = carry_dirty;
                Fr lastv = carry_v;
                for (u32 k = 0; k < 64u; ++k) {
                    u32 outx = 0xffffffffu, nd = dirty;
                    Fr outv = Fr::zero();
                    if (lane == k && have) {
                        bool fast = simple && !dirty;
                        Fr v[3] = {a0, a1, a2}, uc = Fr::zero();
                        int which = -1;
                        u32 x = 0;
                        if (fast) {
#pragma unroll
                            for (u32 j = 0; j < 2u; ++j) {
                                const ChOpen& o = j ? o1 : o0;
                                if (j >= n_open) continue;
                                if (o.wid == lastw) {                            // the predecessor's wire: its value is in the registers
                                    Fr d = Fr::zero();
                                    si_add_term(d, (uint8_t)o.ck, &o.co, 0u, lastv);
                                    if (o.m == 0u) v[0] = Fr::add(v[0], d); else if (o.m == 1u) v[1] = Fr::add(v[1], d); else v[2] = Fr::add(v[2], d);
                                } else if (which >= 0 && (which != (int)o.m || x != o.wid)) fast = false;   // a second open wire: an earlier level of this round's, or an error — memory decides
                                else { uc = which < 0 ? o.co : Fr::add(uc, o.co); which = (int)o.m; x = o.wid; }
                            }
                        }
                        if (fast && which >= 0) {                       // the open wire must not be one this round has assigned already
                            for (u32 q = chain_slot(x);; q = (q + 1u) & 1023u) {
                                const u32 e = produced[q];
                                if (e == x) { fast = false; break; }
                                if (e == 0xffffffffu) break;
                            }
                        }
                        if (fast) {
                            const int rc = si_r1c_finish(v, which, x, uc, w, known, nullptr, &outv);
                            if (rc != SE_OK) { if (atomicCAS(&err[0], 0u, (u32)rc) == 0u) err[1] = ins; }
                            else if (which >= 0) {
                                outx = x;
                                u32 q = chain_slot(x);
                                while (produced[q] != 0xffffffffu) q = (q + 1u) & 1023u;
                                produced[q] = x;
                            }
                        } else {
                            __threadfence_block();
                            solver_step(P, ins, w, known, err, ext);
                            nd = 1u;
                        }
                    }
                    lastw = lane_value(outx, k);
                    dirty = lane_value(nd, k);
#pragma unroll
                    for (int i = 0; i < 8; ++i) lastv.v[i] = lane_value(outv.v[i], k);
                }
                if (lane == 0) { carry_w = lastw; carry_dirty = dirty; carry_v = lastv; }
            }
            __threadfence_block();
            __syncthreads();
        }
    }
}

// a x b = c on every row (zkpor_solver_eval_abc_dev when the run left the CHECK instructions out): out[0] = rows that fail, out[1] = the lowest one
// a Poseidon instruction that names its constraint rows (firstRow) writes a, b, c of those rows itself and r1cs_eval skips them: the rows must
// really be the call's S-box products — row firstRow + k is `... = wire firstOut + k`: an O expression of ONE term, coefficient one, on exactly
// that wire.  calls: (firstOut, firstRow, rows, instruction) quadruples.  bad[0] = violations, bad[1] = the lowest offending instruction.
__global__ __launch_bounds__(256) void k_check_poseidon_rows(const u64* __restrict__ o_ptr, const u32* __restrict__ o_cid, const u32* __restrict__ o_wid,
                                                             const uint8_t* __restrict__ ckind, const u32* __restrict__ calls, u32 n_calls, u32* __restrict__ bad) {
    for (u32 c = blockIdx.x; c < n_calls; c += gridDim.x) {
        const u32 first = calls[4 * c], row0 = calls[4 * c + 1], rows = calls[4 * c + 2], ins = calls[4 * c + 3];
        for (u32 k = threadIdx.x; k < rows; k += 256u) {
            const u64 lo = o_ptr[row0 + k], hi = o_ptr[row0 + k + 1];
            if (hi - lo != 1 || o_wid[lo] != first + k || ckind[o_cid[lo]] != 1) { atomicAdd(bad, 1u); atomicMin(bad + 1, ins); }
        }
    }
}
__global__ __launch_bounds__(256) void k_rows_check(const Fr* __restrict__ a, const Fr* __restrict__ b, const Fr* __restrict__ c, size_t n, unsigned long long* out) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    if (Fr::mul(a[i], b[i]) != c[i]) { atomicAdd(&out[0], 1ull); atomicMin(&out[1], (unsigned long long)i); }
}

__global__ __launch_bounds__(256) void k_count_unknown(const uint8_t* __restrict__ known, size_t n, u32* err) {
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    const bool miss = i < n && !known[i];
    const u64 b = __ballot(miss);
    if ((threadIdx.x & 63u) == 0 && b) atomicAdd(&err[2], (u32)__popcll(b));
}

ZK_D void report(u32* err, u32 code, u32 ins) { if (atomicCAS(&err[0], 0u, code) == 0u) err[1] = ins; }

// ---- gnark logderivarg countHint: inputs = nbTable, nbCols, the table's rows, the queries' rows -> per row how many queries equal it.
// The tables of this circuit carry their own index in column 0 (the range checker: the constants 0 .. 2^w - 1; a lookup table: rows
// (i, entry_i)), so a query finds its row by that column; the remaining columns are compared.  A table whose column 0 is not 0..n-1, a
// query outside the table or one that differs from its row fails the hint — as gnark's does ("query element not in table").
struct CountDev {            // one count hint of a level, as the kernels see it
    u32 ins, nb_table, nb_col, cnt_base;   // cnt_base: its histogram inside the level's
    u64 offs_base, nb_q, row_prefix, q_prefix;   // table rows / queries of the level's hints in front of this one
};
// the hint a flat row / query index of the level belongs to (m hints, prefixes ascending)
ZK_D u32 count_find(const CountDev* __restrict__ cm, u32 m, u64 i, bool queries) {
    u32 lo = 0, hi = m;
    while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if ((queries ? cm[mid].q_prefix : cm[mid].row_prefix) <= i) lo = mid; else hi = mid; }
    return lo;
}
// all count hints of a level in three launches: one thread per table row, per query, per output
__global__ __launch_bounds__(256) void k_count_table(SolverProg P, const CountDev* __restrict__ cm, u32 m, u64 total_rows, const u32* __restrict__ offs_pool,
                                                     const Fr* w, const uint8_t* known, u32* err) {
    const u64 g = (u64)blockIdx.x * 256u + threadIdx.x;
    if (g >= total_rows) return;
    const CountDev c = cm[count_find(cm, m, g, false)];
    const u32 i = (u32)(g - c.row_prefix);
    const u32* cd = P.calldata + P.arg[c.ins];
    const u32* offs = offs_pool + c.offs_base;
    u64 p = offs[2 + (u64)i * c.nb_col];
    Fr v;
    const int rc = si_eval_le(P, cd, p, w, known, &v);
    if (rc) { report(err, (u32)rc, c.ins); return; }
    u32 idx;
    if (!si_index(v, c.nb_table, &idx) || idx != i) report(err, SE_COUNT_TABLE, c.ins);
}
__global__ __launch_bounds__(256) void k_count_queries(SolverProg P, const CountDev* __restrict__ cm, u32 m, u64 total_q, const u32* __restrict__ offs_pool,
                                                       const Fr* w, const uint8_t* known, u32* cnt, u32* err) {
    const u64 g = (u64)blockIdx.x * 256u + threadIdx.x;
    bool pending = false;
    u32 key = 0;                                  // index of this query's counter in the level's histogram
    if (g < total_q) {
        const CountDev c = cm[count_find(cm, m, g, true)];
        const u64 q = g - c.q_prefix;
        const u32* cd = P.calldata + P.arg[c.ins];
        const u32* offs = offs_pool + c.offs_base;
        const u32* qo = offs + 2 + ((u64)c.nb_table + q) * c.nb_col;
        u64 p = qo[0];
        Fr v;
        int rc = si_eval_le(P, cd, p, w, known, &v);
        u32 idx = 0;
        if (!rc && !si_index(v, c.nb_table, &idx)) rc = SE_COUNT_QUERY;
        if (!rc) {
            const u32* to = offs + 2 + (u64)idx * c.nb_col;
            for (u32 k = 1; k < c.nb_col && !rc; ++k) {
                Fr a, b;
                u64 pa = qo[k], pb = to[k];
                rc = si_eval_le(P, cd, pa, w, known, &a);
                if (!rc) rc = si_eval_le(P, cd, pb, w, known, &b);
                if (!rc && a != b) rc = SE_COUNT_QUERY;
            }
        }
        if (rc) report(err, (u32)rc, c.ins);
        else { pending = true; key = c.cnt_base + idx; }
    }
    // range-check limbs are mostly 0 (balances far below 2^64, 128-bit checks of small values): millions of increments of ONE counter.  The
    // lanes of a wave that hit the same counter go through one atomic: a few rounds of leader election catch the frequent values, the rest
    // increment on their own (a wave of 64 distinct limbs would need 64 rounds).
    const u32 lane = threadIdx.x & 63u;
    for (int round = 0; round < 3; ++round) {
        const u64 act = __ballot(pending);
        if (!act) break;
        const int leader = __ffsll((long long)act) - 1;
        const u32 lkey = (u32)__shfl((int)key, leader, 64);
        const u64 same = __ballot(pending && key == lkey);
        if ((int)lane == leader) atomicAdd(&cnt[lkey], (u32)__popcll(same));
        if (key == lkey) pending = false;
    }
    if (pending) atomicAdd(&cnt[key], 1u);
}
__global__ __launch_bounds__(256) void k_count_out(SolverProg P, const CountDev* __restrict__ cm, u32 m, u64 total_rows, const u32* __restrict__ cnt, Fr* w, uint8_t* known) {
    const u64 g = (u64)blockIdx.x * 256u + threadIdx.x;
    if (g >= total_rows) return;
    const CountDev c = cm[count_find(cm, m, g, false)];
    const u32 i = (u32)(g - c.row_prefix);
    const u32 out = P.calldata[P.arg[c.ins] + 3 + i];
    Fr v = Fr::zero();
    v.v[0] = cnt[c.cnt_base + i];
    w[out] = Fr::to_mont(v);
    known[out] = 1;
}

// the input expressions of a hint, evaluated for the caller: out[i] = value of input i (one thread per input; the BSB22 placeholder's
// inputs are the committed wires, thousands to millions of one-term expressions)
__global__ __launch_bounds__(256) void k_hint_inputs(SolverProg P, u32 ins, const u32* __restrict__ offs, u32 n_in, const Fr* __restrict__ w,
                                                     const uint8_t* __restrict__ known, Fr* out, u32* err) {
    const u32 i = blockIdx.x * 256u + threadIdx.x;
    if (i >= n_in) return;
    const u32* cd = P.calldata + P.arg[ins];
    u64 p = offs[i];
    Fr v;
    const int rc = si_eval_le(P, cd, p, w, known, &v);
    if (rc) { report(err, (u32)rc, ins); return; }
    out[i] = v;
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
    P.ext_cap = s->ext_cap;
    return P;
}
static void solver_free(zkpor_solver* s) {
    void* ptrs[] = {s->d_rows, s->d_perr, s->d_ones, s->d_cmeta, s->d_pre_off, s->d_prej_off, s->d_pre, s->d_kind, s->d_arg, s->d_level_instr, s->d_calldata, s->d_gen_cnt, s->d_gen_cnt_all, s->d_long, s->d_offs, s->d_gen_lo, s->d_hint_kind, s->d_known, s->d_err, s->d_ext,
                    s->d_cnt, s->d_tmp};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    // side streams with hardware queues of their own go back to the pool (common.cuh stream_release_own_queue), ordinary ones to the runtime
    for (hipStream_t st : {s->side, s->side2}) {
        if (!st) continue;
        if (s->side_own_queue) stream_release_own_queue(s->ctx ? s->ctx->device : 0, st, 0);
        else { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
    }
    if (s->ev_fork2) (void)hipEventDestroy(s->ev_fork2);
    if (s->ev_join2) (void)hipEventDestroy(s->ev_join2);
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    if (s->ev_join) (void)hipEventDestroy(s->ev_join);
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
    case SE_LOOKUP_RANGE: return "lookup query too large";
    case SE_COUNT_TABLE: return "count hint: the table's first column is not 0 .. n-1 (no native semantics for such a table)";
    case SE_COUNT_QUERY: return "count hint: query element not in table";
    default: return "error";
    }
}
static int32_t tmp_reserve(zkpor_solver* s, size_t elems) {
    if (elems <= s->tmp_cap) return ZKPOR_OK;
    zkpor_ctx* ctx = s->ctx;
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (s->d_tmp) { (void)hipFree(s->d_tmp); s->d_tmp = nullptr; s->tmp_cap = 0; }
    ZK_HIP(ctx, hipMalloc((void**)&s->d_tmp, elems * sizeof(Fr)));
    s->tmp_cap = elems;
    return ZKPOR_OK;
}

static int32_t join_side2(zkpor_solver* s) {
    if (!s->side2_busy) return ZKPOR_OK;
    ZK_HIP(s->ctx, hipEventRecord(s->ev_join2, s->side2));
    ZK_HIP(s->ctx, hipStreamWaitEvent(s->ctx->stream, s->ev_join2, 0));
    s->side2_busy = false;
    return ZKPOR_OK;
}
static int32_t join_side(zkpor_solver* s) {
    if (!s->side_busy) return ZKPOR_OK;
    ZK_HIP(s->ctx, hipEventRecord(s->ev_join, s->side));
    ZK_HIP(s->ctx, hipStreamWaitEvent(s->ctx->stream, s->ev_join, 0));
    s->side_busy = false;
    return ZKPOR_OK;
}

// queue levels from s->next_level on until the program ends or a level with external hints has run; then look at the flags
static int32_t solver_advance(zkpor_solver* s, uint32_t* paused_instr) {
    zkpor_ctx* ctx = s->ctx;
    const SolverProg P = prog_of(s);
    Fr* w = (Fr*)s->d_w;
    const u64 n_levels = s->plan.size();
    *paused_instr = 0xffffffffu;
    if (!s->pending.empty()) { *paused_instr = s->pending.front(); return ZKPOR_OK; }
    u32 h[4] = {0, 0, 0, 0};
    {
        PhaseScope ps(ctx, "solver_levels");
        bool stop = false;
        while (s->next_level < n_levels && !stop) {
            const u64 l = s->next_level;
            const LevelPlan& L = s->plan[l];
            if (l + 1 == n_levels) ZK_TRY(join_side(s));          // ASYNC outputs are read by the last level only (the container's promise)
            if (s->side2_busy && l >= s->join2_level) ZK_TRY(join_side2(s));   // ... those of a call with a join level from that level on
            const u64 stop_at = s->side2_busy ? s->join2_level : n_levels;      // a run of levels does not reach across the join
            // the generic instructions this run executes: the CHECK ones (assertions) only when nobody checks the rows afterwards
            auto G = [&](const LevelPlan& M) { return M.n_gen + (s->checks_left ? 0u : M.n_chk); };
            const u32* gen_cnt = s->checks_left ? s->d_gen_cnt : s->d_gen_cnt_all;
            const u32 ng = G(L);
            const bool only_narrow = L.n_pos == 0 && L.n_posa == 0 && L.n_posj == 0 && L.n_cnt == 0 && ng <= NARROW;
            if (only_narrow) {
                // a chain — CHAIN_FROM or more levels of ONE generic instruction each, none of them external or the last — has its own kernel
                auto one = [&](u64 q) { const LevelPlan& M = s->plan[q]; return q + 1 < n_levels && q < stop_at && !M.n_pos && !M.n_posa && !M.n_posj && !M.n_cnt && G(M) == 1 && M.n_gen == 1 && !M.external; };
                auto chain_end = [&](u64 q) { while (q < n_levels && one(q)) ++q; return q; };
                if (ctx->solver_chain && one(l)) {
                    const u64 e = chain_end(l);
                    if (e - l >= CHAIN_FROM) {
                        hipLaunchKernelGGL(k_solve_chain, dim3(1), dim3(256), 0, ctx->stream, P, s->d_level_instr, s->d_gen_lo, l, e, w, s->known, s->d_err, s->d_ext);
                        ++s->launches;
                        s->next_level = e;
                        continue;
                    }
                }
                u64 l1 = l;                       // the run of narrow levels starting here, ended by (and including) a level with external hints
                while (l1 < n_levels) {
                    const LevelPlan& M = s->plan[l1];
                    if (M.n_pos || M.n_posa || M.n_posj || M.n_cnt || G(M) > NARROW || l1 >= stop_at) break;
                    if (l1 + 1 == n_levels && l1 != l && s->side_busy) break;   // the last level starts its own launch, behind the join
                    if (ctx->solver_chain && l1 != l && one(l1) && chain_end(l1) - l1 >= CHAIN_FROM) break;   // a chain starts here
                    ++l1;
                    if (M.external) { stop = true; break; }
                }
                hipLaunchKernelGGL(k_solve_narrow, dim3(1), dim3(NARROW), 0, ctx->stream, P, s->d_level_instr, s->d_gen_lo, gen_cnt, l, l1, w, s->known, s->d_err, s->d_ext);
                ++s->launches;
                s->next_level = l1;
                continue;
            }
            if (L.n_posa && !(s->skip_async && l == 0)) {   // fork: everything queued so far is visible to the side stream
                ZK_HIP(ctx, hipEventRecord(s->ev_fork, ctx->stream));
                ZK_HIP(ctx, hipStreamWaitEvent(s->side, s->ev_fork, 0));
                for (u32 k = 0; k < L.n_posa; ++k) {   // the inputs of a long call (10 000 expressions, some a thousand terms long) side by side first
                    const BigHint& a = s->asyncs[L.posa_first + k];
                    hipLaunchKernelGGL(k_hint_inputs, dim3((a.n_in + 255u) / 256u), dim3(256), 0, s->side, P, a.ins, s->d_offs + a.offs_base, a.n_in, (const Fr*)w, s->known,
                                       s->d_pre + a.nb_q, s->d_err);
                }
                ZK_TRY(gadget_poseidon_launch(ctx, s->side, P, s->d_level_instr + L.lo + L.n_gen + L.n_chk + L.n_pos, L.n_posa, w, s->known, s->d_err, s->d_pre, s->d_pre_off + L.posa_first, nullptr, nullptr, nullptr));
                s->side_busy = true;
                ++s->launches;
            }
            if (L.n_posj) {   // calls other levels run beside (the RLC challenge's 116-permutation sponge): on their own side stream, joined in front of their first consumer
                const u32* lst = s->d_level_instr + L.lo + L.n_gen + L.n_chk + L.n_pos + L.n_posa;
                const bool beside = ctx->solver_beside && !s->side2_busy;
                const hipStream_t js = beside ? s->side2 : ctx->stream;
                if (beside) {
                    ZK_HIP(ctx, hipEventRecord(s->ev_fork2, ctx->stream));
                    ZK_HIP(ctx, hipStreamWaitEvent(s->side2, s->ev_fork2, 0));
                }
                // "solver_pre_join" (round 6): the sponge's 1 392 input expressions (14 to 79 terms each) side by side first — inside the serial kernel the one lane
                // that holds input i evaluates it while the other lanes wait: 12 expressions in a row per permutation, 0.14 of the 0.36 ms a permutation took
                const bool pre = ctx->solver_pre_join != 0;
                if (pre) for (u32 k = 0; k < L.n_posj; ++k) {
                    const BigHint& a = s->joins[L.posj_first + k];
                    hipLaunchKernelGGL(k_hint_inputs, dim3((a.n_in + 255u) / 256u), dim3(256), 0, js, P, a.ins, s->d_offs + a.offs_base, a.n_in, (const Fr*)w, s->known, s->d_pre + a.nb_q, s->d_err);
                }
                ZK_TRY(gadget_poseidon_launch(ctx, js, P, lst, L.n_posj, w, s->known, s->d_err, pre ? s->d_pre : nullptr, pre ? s->d_prej_off + L.posj_first : nullptr, nullptr, nullptr, nullptr));
                if (beside) { s->side2_busy = true; s->join2_level = L.join_level; }
                ++s->launches;
            }
            if ((int64_t)ng >= ctx->solver_tree_from) {
                u32* lc = ctx->solver_long ? s->d_long : nullptr;     // long constraints go to a wave each, behind the level's launch
                if (lc) ZK_HIP(ctx, hipMemsetAsync(lc, 0, 4, ctx->stream));
                if ((int64_t)ng >= ctx->solver_batch_from) {          // enough instructions to fill the chip several per thread: divisions share an inversion
                    const u32 stride = (ng + BATCH_K - 1) / BATCH_K;
                    hipLaunchKernelGGL(k_solve_level_batched<BATCH_K>, dim3((stride + 255u) / 256u), dim3(256), 0, ctx->stream, P, s->d_level_instr, L.lo, ng, stride, w, s->known, s->d_err, s->d_ext, lc, lc ? lc + 1 : nullptr, (u32)ctx->solver_long);
                } else {                                              // one instruction per thread, one inversion per workgroup
                    hipLaunchKernelGGL(k_solve_level_batched<1>, dim3((ng + 255u) / 256u), dim3(256), 0, ctx->stream, P, s->d_level_instr, L.lo, ng, ng, w, s->known, s->d_err, s->d_ext, lc, lc ? lc + 1 : nullptr, (u32)ctx->solver_long);
                }
                ++s->launches;
                if (lc) { hipLaunchKernelGGL(k_solve_long, dim3(512), dim3(256), 0, ctx->stream, P, (const u32*)lc, (const u32*)(lc + 1), w, s->known, s->d_err); ++s->launches; }
            } else if (ng) {
                hipLaunchKernelGGL(k_solve_level, dim3((ng + 255u) / 256u), dim3(256), 0, ctx->stream, P, s->d_level_instr, L.lo, ng, w, s->known, s->d_err, s->d_ext);
                ++s->launches;
            }
            if (L.n_pos) {
                const bool rows = s->abc_written;   // decided when the run started
                ZK_TRY(gadget_poseidon_launch(ctx, ctx->stream, P, s->d_level_instr + L.lo + L.n_gen + L.n_chk, L.n_pos, w, s->known, s->d_err, nullptr, nullptr,
                                              rows ? s->abc_a : nullptr, s->abc_b, s->abc_c));
                ++s->launches;
            }
            if (L.n_cnt) {                        // every count hint of the level together: rows, queries, outputs
                const CountDev* cm = (const CountDev*)s->d_cmeta + L.cnt_first;
                ZK_HIP(ctx, hipMemsetAsync(s->d_cnt, 0, (size_t)L.cnt_rows * sizeof(u32), ctx->stream));
                hipLaunchKernelGGL(k_count_table, dim3((unsigned)((L.cnt_rows + 255) / 256)), dim3(256), 0, ctx->stream, P, cm, L.n_cnt, L.cnt_rows, s->d_offs, w, s->known, s->d_err);
                if (L.cnt_queries) hipLaunchKernelGGL(k_count_queries, dim3((unsigned)((L.cnt_queries + 255) / 256)), dim3(256), 0, ctx->stream, P, cm, L.n_cnt, L.cnt_queries, s->d_offs, w, s->known, s->d_cnt, s->d_err);
                hipLaunchKernelGGL(k_count_out, dim3((unsigned)((L.cnt_rows + 255) / 256)), dim3(256), 0, ctx->stream, P, cm, L.n_cnt, L.cnt_rows, s->d_cnt, w, s->known);
                s->launches += 3;
            }
            s->next_level = l + 1;
            stop = L.external;
        }
        ZK_KERNEL_CHECK(ctx);
    }
    const bool finished = s->next_level >= n_levels;
    if (finished) {
        ZK_TRY(join_side(s));
        ZK_TRY(join_side2(s));
        // counted afresh every time the end is reached: a run that paused in its LAST level has been here before, with the hint's outputs still open
        ZK_HIP(ctx, hipMemsetAsync(s->d_err + 2, 0, sizeof(u32), ctx->stream));
        hipLaunchKernelGGL(k_count_unknown, dim3((unsigned)((s->r1cs->n_wires + 255) / 256)), dim3(256), 0, ctx->stream, s->known, s->r1cs->n_wires, s->d_err);
        ZK_KERNEL_CHECK(ctx);
    }
    ZK_HIP(ctx, hipMemcpyAsync(h, s->d_err, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    u32 hp[2] = {0, 0};
    if (finished && s->skip_async) ZK_HIP(ctx, hipMemcpyAsync(hp, s->d_perr, sizeof hp, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (!h[0] && hp[0]) { h[0] = hp[0]; h[1] = hp[1]; }
    if (h[0]) {
        s->running = false;
        if (s->side_busy) { (void)hipStreamSynchronize(s->side); s->side_busy = false; }
        if (s->side2_busy) { (void)hipStreamSynchronize(s->side2); s->side2_busy = false; }
        ctx->err = std::string("solver: ") + solver_error_text(h[0]) + " at instruction " + std::to_string(h[1]);
        return ZKPOR_E_STATE;
    }
    if (h[3]) {   // external hints met in the last level: serve them one by one (a level is small next to what a commitment costs)
        if (h[3] > s->ext_cap) { s->running = false; ctx->err = "solver: a level reported more external hints than the program holds"; return ZKPOR_E_STATE; }   // cannot happen: the list holds every external hint of the program
        s->pending.resize(h[3]);
        ZK_HIP(ctx, hipMemcpyAsync(s->pending.data(), s->d_ext, h[3] * sizeof(u32), hipMemcpyDeviceToHost, ctx->stream));   // never the NULL stream: it waits for every blocking stream of the device
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
        std::sort(s->pending.begin(), s->pending.end());
        ZK_HIP(ctx, hipMemsetAsync(s->d_err + 3, 0, sizeof(u32), ctx->stream));
        *paused_instr = s->pending.front();
        return ZKPOR_OK;
    }
    if (finished) {
        s->running = false;
        if (h[2]) {
            ctx->err = "solver: " + std::to_string(h[2]) + " wires were never assigned";
            // a constraint that SOLVES a wire but carries the CHECK flag was left out with the assertions: the exporter's flags are wrong, not the witness
            if (s->checks_left) ctx->err += " (this run left the " + std::to_string(s->n_check) + " CHECK-flagged instructions to the row check: if the program's flags are not trustworthy — a new exporter — set the context parameter solver_defer_checks to 0)";
            return ZKPOR_E_STATE;
        }
        s->run_ok = true;      // this d_w (and the rows the run wrote) may be proved over: zkpor_solver_eval_abc_dev
    }
    return ZKPOR_OK;
}
// With several workers per GPU ("tail_reserve_cus") the side streams — 0.2 s hash chains — get hardware queues of their own, once: in the runtime's pool of
// four they would sit in front of some other stream's kernels (common.cuh stream_create_own_queue)
int32_t solver_side_queues(zkpor_solver* s) {
    zkpor_ctx* ctx = s->ctx;
    if ((ctx->tail_reserve_cus <= 0 && !ctx->tail_streams) || s->side_own_queue) return ZKPOR_OK;
    if (s->running || s->side_busy || s->side2_busy) return ZKPOR_OK;      // not under a run or a chain in flight: next time
    hipStream_t a = nullptr, b = nullptr;
    ZK_TRY(stream_create_own_queue(ctx, &a, 0));
    if (stream_create_own_queue(ctx, &b, 0) != ZKPOR_OK) { stream_release_own_queue(ctx->device, a, 0); return ZKPOR_E_HIP; }
    // the first pair is parked, not destroyed: the solver's events were recorded on it (the context destroys what it has retired)
    if (s->side) { (void)hipStreamSynchronize(s->side); ctx->retired_streams.push_back(s->side); }
    if (s->side2) { (void)hipStreamSynchronize(s->side2); ctx->retired_streams.push_back(s->side2); }
    s->side = a; s->side2 = b;
    s->side_own_queue = true;
    return ZKPOR_OK;
}
zkpor_ctx* solver_ctx(zkpor_solver* s) { return s->ctx; }
zkpor_r1cs* solver_r1cs(zkpor_solver* s) { return s->r1cs; }
}  // namespace zk

using namespace zk;
extern "C" {

int32_t zkpor_solver_create(zkpor_r1cs* r1cs, const uint8_t* container, size_t len, zkpor_solver** out) try {
    return zkpor_solver_create_on(r1cs ? r1cs->ctx : nullptr, r1cs, container, len, out);
} ZK_ABI_CATCH_IN((r1cs ? r1cs->ctx : nullptr))

/* the same program bound to ANOTHER context of the GPU the matrices live on: its launches, phase timers and error text belong to `ctx`, the
 * matrices are only read — one solver per worker context, all over one zkpor_r1cs (two workers of a GPU solve side by side) */
int32_t zkpor_solver_create_on(zkpor_ctx* ctx, zkpor_r1cs* r1cs, const uint8_t* container, size_t len, zkpor_solver** out) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !r1cs || !container || !out) return ZKPOR_E_ARG;
    if (ctx->device != r1cs->ctx->device) { ctx->err = "solver: the constraint matrices live on another GPU than the context"; return ZKPOR_E_ARG; }
    for (int m = 0; m < 3; ++m) if (!r1cs->row_ptr[m]) { ctx->err = "solver: the constraint matrices are not loaded"; return ZKPOR_E_STATE; }
    // held until the program is fully loaded: an exception on the way (std::bad_alloc from a vector that scales with the program is the realistic
    // one; it stops at the ABI firewall) or an error return frees the object and whatever it owns on the device
    std::unique_ptr<zkpor_solver, void (*)(zkpor_solver*)> hold(new zkpor_solver(), solver_free);
    zkpor_solver* s = hold.get();
    s->ctx = ctx; s->r1cs = r1cs;
    s->container.assign(container, container + len);
    std::string why;
    if (zkpor_host::ParseSolverFile(s->container.data(), len, &s->view, &why) != 0) { ctx->err = why; return ZKPOR_E_ARG; }
    const auto& v = s->view;
    auto bad = [&](const std::string& m) { ctx->err = "solver: " + m; return ZKPOR_E_ARG; };
    if (v.n_calldata >= (1ull << 32)) return bad("call data beyond 2^32 words");
    const uint64_t nw = r1cs->n_wires, ncoef = r1cs->n_coeff;
    // validate what the kernels index with — constraint ids, call-data offsets and shapes, wire / coefficient ids inside the call data —
    // and sort every instruction into its class
    s->hint_kind.resize(v.hint_names.size());
    for (size_t i = 0; i < v.hint_names.size(); ++i) s->hint_kind[i] = hint_kind_of_name(v.hint_names[i].c_str());
    enum : uint8_t { CL_GEN = 0, CL_CHK = 1, CL_POS = 2, CL_POSA = 3, CL_POSJ = 4, CL_CNT = 5 };
    std::vector<uint8_t> cls(v.n_instructions, CL_GEN), external(v.n_instructions, 0);
    std::vector<uint32_t> kinds(v.n_instructions);
    std::set<std::pair<uint32_t, uint32_t>> tables_ok;   // (block, nbEntries) already validated: a table's entries are checked once, not per lookup
    std::vector<uint32_t> offs;                          // the offset pool of count / external hints
    std::map<uint32_t, BigHint> big;                     // by instruction
    uint64_t max_table = 1, pre_total = 0;
    std::vector<uint32_t> row_bits((r1cs->n_constraints + 31) / 32 + 1, 0);
    std::vector<uint32_t> pos_rows;                      // (first output wire, first row, rows, instruction) of every call that writes its own rows
    auto const_u32 = [&](const uint32_t* cd, uint64_t p, uint32_t* out_v) {   // a constant expression's value (nbTable, nbCols)
        if (cd[p] == 0) { *out_v = 0; return true; }
        if (cd[p] != 1 || cd[p + 2] != 0) return false;
        const Fr c = Fr::from_mont(r1cs->h_coeff[cd[p + 1]]);
        for (int k = 1; k < 8; ++k) if (c.v[k]) return false;
        *out_v = c.v[0];
        return true;
    };
    for (uint64_t i = 0; i < v.n_instructions; ++i) {
        const uint32_t kind = zkpor_host::InstrKind(v, i), arg = v.arg[i];
        kinds[i] = kind;
        if (kind == SI_R1C) {
            if (arg >= r1cs->n_constraints) return bad("instruction " + std::to_string(i) + " names a constraint outside the system");
            ++s->n_r1c;
            if (zkpor_host::InstrIsCheck(v, i)) { cls[i] = CL_CHK; ++s->n_check; }
        }
        else if (kind == SI_LOOKUP) {
            if ((uint64_t)arg + 4 > v.n_calldata) return bad("the call data of lookup " + std::to_string(i) + " is malformed");
            const uint32_t* cd = v.calldata + arg;
            const bool known_table = tables_ok.count({cd[0], cd[1]}) != 0;
            if (known_table) {   // only the queries
                uint64_t p = (uint64_t)arg + 4;
                bool ok = (uint64_t)cd[3] + cd[2] <= nw;
                for (uint32_t q = 0; ok && q < cd[2]; ++q) {
                    ok = p < v.n_calldata;
                    if (!ok) break;
                    const uint64_t nt = v.calldata[p++];
                    ok = p + 2 * nt <= v.n_calldata;
                    for (uint64_t t = 0; ok && t < nt; ++t) ok = v.calldata[p + 2 * t] < ncoef && v.calldata[p + 1 + 2 * t] < nw;
                    p += 2 * nt;
                }
                if (!ok) return bad("the call data of lookup " + std::to_string(i) + " is malformed");
            } else {
                if (!zkpor_host::CheckLookupShape(v, arg, nw, ncoef)) return bad("the call data of lookup " + std::to_string(i) + " is malformed");
                tables_ok.insert({cd[0], cd[1]});
            }
            ++s->n_lookup;
        } else if (kind == SI_POSEIDON) {
            if (!zkpor_host::CheckPoseidonShape(v, arg, nw, ncoef)) return bad("the call data of Poseidon instruction " + std::to_string(i) + " is malformed");
            {
                const uint32_t fl = v.calldata[arg + 3];
                cls[i] = !(fl & zkpor_host::POSEIDON_ASYNC) ? CL_POS : ((fl >> zkpor_host::POSEIDON_JOIN_SHIFT) ? CL_POSJ : CL_POSA);
            }
            {
                const uint64_t row0 = v.calldata[arg + 4], nrows = v.calldata[arg + 2];
                if (row0 != 0xffffffffull) {
                    if (row0 + nrows > r1cs->n_constraints) return bad("Poseidon instruction " + std::to_string(i) + " names rows outside the system");
                    if (cls[i] == CL_POS) {
                        for (uint64_t rr = row0; rr < row0 + nrows; ++rr) {
                            if (row_bits[rr >> 5] & (1u << (rr & 31))) return bad("Poseidon instructions " + std::to_string(i) + " and an earlier one both claim constraint row " + std::to_string(rr));
                            row_bits[rr >> 5] |= 1u << (rr & 31);
                        }
                        s->rows_covered += nrows;
                        pos_rows.push_back(v.calldata[arg + 1]); pos_rows.push_back((uint32_t)row0); pos_rows.push_back((uint32_t)nrows); pos_rows.push_back((uint32_t)i);
                    }
                }
            }
            if (cls[i] == CL_POSA || cls[i] == CL_POSJ) {     // long serial calls: their inputs are evaluated side by side in front of them (k_hint_inputs)
                BigHint b;
                b.ins = (uint32_t)i; b.n_in = v.calldata[arg]; b.offs_base = offs.size(); b.nb_q = pre_total;
                uint64_t p = zkpor_host::POSEIDON_HDR;
                for (uint32_t k = 0; k < b.n_in; ++k) {
                    offs.push_back((uint32_t)p);
                    const uint64_t nt = v.calldata[arg + p];
                    if (cls[i] == CL_POSA) for (uint64_t t = 0; t < nt; ++t) s->async_max_wire = std::max<uint64_t>(s->async_max_wire, v.calldata[arg + p + 2 + 2 * t]);   // a prefetch reads these from the bare assignment
                    p += 1 + 2 * nt;
                }
                pre_total += b.n_in;
                big[(uint32_t)i] = b;
            }
            ++s->n_poseidon;
        } else if (kind == SI_HINT) {
            bool ok = (uint64_t)arg + 3 <= v.n_calldata;
            if (ok) {
                const uint32_t* cd = v.calldata + arg;
                ok = cd[0] < v.hint_names.size() && (uint64_t)arg + 3 + cd[2] <= v.n_calldata;
                uint64_t p = 3 + (uint64_t)(ok ? cd[2] : 0);
                for (uint32_t k = 0; ok && k < cd[2]; ++k) ok = cd[3 + k] < nw;
                const uint8_t hk = ok ? s->hint_kind[cd[0]] : 0;
                const bool keep_offs = ok && (hk == HK_NONE || hk == HK_COUNT);
                BigHint b;
                if (keep_offs) { b.ins = (uint32_t)i; b.n_in = cd[1]; b.n_out = cd[2]; b.offs_base = offs.size(); offs.reserve(offs.size() + cd[1]); }
                for (uint32_t k = 0; ok && k < cd[1]; ++k) {
                    ok = arg + p < v.n_calldata;
                    if (!ok) break;
                    if (keep_offs) offs.push_back((uint32_t)p);
                    const uint32_t nt = cd[p++];
                    ok = arg + p + 2ull * nt <= v.n_calldata;
                    for (uint32_t t = 0; ok && t < nt; ++t) { ok = cd[p] < ncoef && cd[p + 1] < nw; p += 2; }
                }
                if (ok && hk == HK_NONE) { external[i] = 1; big[(uint32_t)i] = b; }
                if (ok && hk == HK_COUNT) {
                    ok = cd[1] >= 2 && const_u32(cd, offs[b.offs_base], &b.nb_table) && const_u32(cd, offs[b.offs_base + 1], &b.nb_col) && b.nb_col >= 1 &&
                         b.nb_table >= 1 && b.nb_table == cd[2] && (uint64_t)cd[1] >= 2 + (uint64_t)b.nb_table * b.nb_col &&
                         ((uint64_t)cd[1] - 2 - (uint64_t)b.nb_table * b.nb_col) % b.nb_col == 0;
                    if (ok) { b.nb_q = ((uint64_t)cd[1] - 2 - (uint64_t)b.nb_table * b.nb_col) / b.nb_col; big[(uint32_t)i] = b; cls[i] = CL_CNT; max_table = std::max<uint64_t>(max_table, b.nb_table); }
                }
            }
            if (!ok) return bad("the call data of instruction " + std::to_string(i) + " is malformed");
            ++s->n_hint;
        } else ++s->n_skip;
    }
    // the level lists, every level reordered by class; count hints collected in level order
    const uint64_t n_li = v.level_ptr[v.n_levels];
    std::vector<uint32_t> li(n_li), gen_cnt(v.n_levels), gen_cnt_all(v.n_levels);
    std::vector<uint64_t> gen_lo(v.n_levels);
    std::vector<uint32_t> pre_off_host, prej_off_host;
    std::vector<CountDev> cmeta;
    s->plan.resize(v.n_levels);
    for (uint64_t l = 0; l < v.n_levels; ++l) {
        LevelPlan& L = s->plan[l];
        L.lo = v.level_ptr[l];
        uint64_t o = L.lo;
        for (uint8_t c = CL_GEN; c <= CL_CNT; ++c) {
            uint32_t n = 0;
            for (uint64_t k = v.level_ptr[l]; k < v.level_ptr[l + 1]; ++k) {
                const uint32_t ins = v.level_instr[k];
                if (cls[ins] != c) continue;
                li[o++] = ins; ++n;
                if (external[ins]) L.external = true;
                if (c == CL_CNT) s->counts.push_back(big[ins]);
                if (c == CL_POSA) { s->asyncs.push_back(big[ins]); pre_off_host.push_back((uint32_t)big[ins].nb_q); }
                if (c == CL_POSJ) {
                    s->joins.push_back(big[ins]); prej_off_host.push_back((uint32_t)big[ins].nb_q);
                    const uint64_t j = (uint64_t)(v.calldata[v.arg[ins] + 3] >> zkpor_host::POSEIDON_JOIN_SHIFT) - 1;
                    if (j <= l || j >= v.n_levels) return bad("Poseidon instruction " + std::to_string(ins) + " names a join level outside (its own level, the last level]");
                    L.join_level = L.n_posj_seen++ ? std::min(L.join_level, j) : j;
                }
            }
            if (c == CL_GEN) L.n_gen = n; else if (c == CL_CHK) L.n_chk = n; else if (c == CL_POS) L.n_pos = n; else if (c == CL_POSA) { L.n_posa = n; L.posa_first = (uint32_t)(s->asyncs.size() - n); } else if (c == CL_POSJ) { L.n_posj = n; L.posj_first = (uint32_t)(s->joins.size() - n); } else { L.n_cnt = n; L.cnt_first = (uint32_t)(s->counts.size() - n); }
        }
        for (uint32_t k = 0; k < L.n_cnt; ++k) {
            const BigHint& c = s->counts[L.cnt_first + k];
            CountDev d;
            d.ins = c.ins; d.nb_table = c.nb_table; d.nb_col = c.nb_col; d.cnt_base = (uint32_t)L.cnt_rows;
            d.offs_base = c.offs_base; d.nb_q = c.nb_q; d.row_prefix = L.cnt_rows; d.q_prefix = L.cnt_queries;
            cmeta.push_back(d);
            L.cnt_rows += c.nb_table; L.cnt_queries += c.nb_q;
        }
        if (L.cnt_rows >= (1ull << 32)) return bad("count hints of one level cover more than 2^32 table rows");
        max_table = std::max<uint64_t>(max_table, L.cnt_rows);
        gen_lo[l] = L.lo; gen_cnt[l] = L.n_gen; gen_cnt_all[l] = L.n_gen + L.n_chk;
        if (L.n_posa && l + 1 >= v.n_levels) return bad("an ASYNC instruction in the last level");
        if (L.n_posj && L.external) return bad("a Poseidon call with a join level in a level with external hints");
    }
    for (auto& kv : big) if (external[kv.first]) s->externals[kv.first] = kv.second;
    auto up = [&](void** d, const void* h, size_t bytes) {
        if (hipMalloc(d, bytes ? bytes : 4) != hipSuccess) { (void)hipGetLastError(); return false; }
        return bytes == 0 || zk::h2d_sync(ctx, *d, h, bytes) == ZKPOR_OK;
    };
    bool ok = up((void**)&s->d_kind, kinds.data(), v.n_instructions * 4) && up((void**)&s->d_arg, v.arg, v.n_instructions * 4) &&
              up((void**)&s->d_level_instr, li.data(), n_li * 4) && up((void**)&s->d_calldata, v.calldata, v.n_calldata * 4) &&
              up((void**)&s->d_gen_lo, gen_lo.data(), v.n_levels * 8) && up((void**)&s->d_gen_cnt, gen_cnt.data(), v.n_levels * 4) && up((void**)&s->d_gen_cnt_all, gen_cnt_all.data(), v.n_levels * 4) &&
              up((void**)&s->d_rows, row_bits.data(), row_bits.size() * 4) && up((void**)&s->d_offs, offs.data(), offs.size() * 4) && up(&s->d_cmeta, cmeta.data(), cmeta.size() * sizeof(CountDev)) && up((void**)&s->d_pre_off, pre_off_host.data(), pre_off_host.size() * 4) && up((void**)&s->d_prej_off, prej_off_host.data(), prej_off_host.size() * 4) &&
              hipMalloc((void**)&s->d_pre, (pre_total ? pre_total : 1) * sizeof(Fr)) == hipSuccess && up((void**)&s->d_hint_kind, s->hint_kind.data(), s->hint_kind.size()) &&
              hipMalloc((void**)&s->d_known, nw) == hipSuccess && hipMalloc((void**)&s->d_err, 32) == hipSuccess && hipMalloc((void**)&s->d_long, (size_t)(LONG_CAP + 1u) * 4) == hipSuccess &&
              hipMalloc((void**)&s->d_ext, (size_t)(s->ext_cap = (u32)std::max<size_t>(EXT_CAP, s->externals.size() + 1)) * sizeof(u32)) == hipSuccess && hipMalloc((void**)&s->d_cnt, max_table * sizeof(u32)) == hipSuccess &&
              hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking) == hipSuccess && hipEventCreateWithFlags(&s->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming) == hipSuccess && hipStreamCreateWithFlags(&s->side2, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_fork2, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&s->ev_join2, hipEventDisableTiming) == hipSuccess;
    if (!ok) { (void)hipGetLastError(); ctx->err = "solver: out of device memory"; return ZKPOR_E_OOM; }
    if (!pos_rows.empty()) {   // the rows the Poseidon instructions will write instead of r1cs_eval: checked against the O matrix, once, here
        u32* d_calls = nullptr;
        u32 h[2] = {0u, 0xffffffffu};
        hipError_t e = hipMalloc((void**)&d_calls, pos_rows.size() * 4);
        if (e == hipSuccess) e = hipMemcpyAsync(d_calls, pos_rows.data(), pos_rows.size() * 4, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(s->d_err + 4, h, 8, hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            const u32 n_calls = (u32)(pos_rows.size() / 4);
            hipLaunchKernelGGL(k_check_poseidon_rows, dim3(n_calls < 4096u ? n_calls : 4096u), dim3(256), 0, ctx->stream, (const u64*)r1cs->row_ptr[2], (const u32*)r1cs->cid[2],
                               (const u32*)r1cs->wid[2], (const uint8_t*)r1cs->coeff_kind, (const u32*)d_calls, n_calls, s->d_err + 4);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(h, s->d_err + 4, 8, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (d_calls) (void)hipFree(d_calls);
        if (e != hipSuccess) { ctx->err = std::string("solver: checking the Poseidon rows: ") + hipGetErrorString(e); return ZKPOR_E_HIP; }
        if (h[0]) {
            ctx->err = "solver: " + std::to_string(h[0]) + " of the constraint rows Poseidon instructions claim (firstRow) are not `... = output wire` of that call; the first such instruction is " + std::to_string(h[1]);
            return ZKPOR_E_ARG;
        }
    }
    *out = hold.release();
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

void zkpor_solver_destroy(zkpor_solver* s) try {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s) return;
    (void)hipStreamSynchronize(s->ctx->stream);
    if (s->side) (void)hipStreamSynchronize(s->side);
    if (s->side2) (void)hipStreamSynchronize(s->side2);
    solver_free(s);
} catch (...) { zk::abi_exception("exception in zkpor_solver_destroy"); }

int32_t zkpor_solver_dims(const zkpor_solver* s, uint64_t dims[7]) try {
    if (!s || !dims) return ZKPOR_E_ARG;
    dims[0] = s->view.n_instructions; dims[1] = s->view.n_levels; dims[2] = s->n_r1c; dims[3] = s->n_hint + s->n_lookup + s->n_poseidon; dims[4] = s->n_skip;
    uint64_t ext = 0;
    for (const LevelPlan& L : s->plan) ext += L.external;
    dims[5] = ext;
    dims[6] = s->launches;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((s ? zk::solver_ctx(const_cast<zkpor_solver*>(s)) : nullptr))

int32_t zkpor_solver_start_dev(zkpor_solver* s, void* d_w, size_t n_inputs, uint8_t* d_known_or_null, uint32_t* paused_instr) try {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s || !d_w || !paused_instr) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = s->ctx;
    if (n_inputs == 0 || n_inputs > s->r1cs->n_wires) { ctx->err = "solver: the assignment must hold 1 + nPublic + nSecret elements"; return ZKPOR_E_ARG; }
    ZK_TRY(solver_side_queues(s));
    s->skip_async = s->prefetched_w != nullptr && s->prefetched_w == d_w;
    if (s->side_busy && !s->skip_async) { (void)hipStreamSynchronize(s->side); s->side_busy = false; }   // an abandoned run, or a prefetch for another vector
    if (s->side2_busy) { (void)hipStreamSynchronize(s->side2); s->side2_busy = false; }
    s->prefetched_w = nullptr;
    s->d_w = d_w;
    s->known = d_known_or_null ? d_known_or_null : s->d_known;
    if (!d_known_or_null) ZK_HIP(ctx, hipMemsetAsync(s->d_known, 0, s->r1cs->n_wires, ctx->stream));
    ZK_HIP(ctx, hipMemsetAsync(s->known, 1, n_inputs, ctx->stream));
    ZK_HIP(ctx, hipMemsetAsync(s->d_err, 0, 16, ctx->stream));
    if (s->skip_async)     // the prefetched instructions' wires are (being) assigned on the side stream: flagged here, joined in front of the last level
        for (const BigHint& a : s->asyncs) { const uint32_t* cd = s->view.calldata + s->view.arg[a.ins]; ZK_HIP(ctx, hipMemsetAsync(s->known + cd[1], 1, cd[2], ctx->stream)); }
    s->abc_written = s->abc_a != nullptr && ctx->solver_poseidon == 1 && s->rows_covered > 0;
    s->checks_left = s->abc_a != nullptr && ctx->solver_defer_checks != 0 && s->n_check > 0;   // zkpor_solver_eval_abc_dev verifies a x b = c on every row instead
    s->running = true; s->run_ok = false; s->next_level = 0; s->pending.clear(); s->launches = 0;
    return solver_advance(s, paused_instr);
} ZK_ABI_CATCH_IN((s ? zk::solver_ctx(s) : nullptr))

/* a, b, c (domain-size buffers of the prove tail) for the NEXT runs: the Poseidon instructions then write the rows of their own constraints
 * while they have the S-box inputs in registers (two thirds of all terms of the real circuit's matrices sit in those rows);
 * zkpor_solver_eval_abc_dev evaluates the rest.  NULL pointers switch it off.  ASYNC / prefetched instructions never write rows. */
int32_t zkpor_solver_set_abc_dev(zkpor_solver* s, void* d_a, void* d_b, void* d_c) try {
    if (!s) return ZKPOR_E_ARG;
    if ((d_a || d_b || d_c) && !(d_a && d_b && d_c)) { s->ctx->err = "solver: a, b, c are given together or not at all"; return ZKPOR_E_ARG; }
    s->abc_a = (Fr*)d_a; s->abc_b = (Fr*)d_b; s->abc_c = (Fr*)d_c;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((s ? zk::solver_ctx(s) : nullptr))
/* a, b, c = L.w, R.w, O.w for every row the finished run has not written already (all rows when zkpor_solver_set_abc_dev was not used), zero
 * padding up to domain_size; into the buffers given to zkpor_solver_set_abc_dev, or d_a / d_b / d_c when it was not used */
int32_t zkpor_solver_eval_abc_dev(zkpor_solver* s, const void* d_w, void* d_a, void* d_b, void* d_c, size_t domain_size) try {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s || !d_w || !d_a || !d_b || !d_c) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = s->ctx;
    if ((s->abc_written || s->checks_left) && (d_a != s->abc_a || d_b != s->abc_b || d_c != s->abc_c)) { ctx->err = "solver: the run was made for other a, b, c buffers than these"; return ZKPOR_E_ARG; }
    // rows the run wrote are skipped below and assertions it left out are verified here: both only mean something for the vector THAT run solved, to its end
    if ((s->abc_written || s->checks_left) && (!s->run_ok || d_w != s->d_w)) {
        ctx->err = !s->run_ok ? "solver: no completed run to take a, b, c from (the last run failed, is paused, or none was made)" : "solver: d_w is not the wire vector the last run solved";
        return ZKPOR_E_STATE;
    }
    ZK_TRY(zk::r1cs_eval_on(ctx, s->r1cs, d_w, d_a, d_b, d_c, domain_size, s->abc_written ? s->d_rows : nullptr));
    if (!s->checks_left) return ZKPOR_OK;
    // the run left its CHECK instructions out: every row is verified here instead, a x b = c (the Poseidon rows the solver wrote included)
    const size_t n = s->r1cs->n_constraints;
    if (n == 0) return ZKPOR_OK;
    unsigned long long h[2] = {0ull, ~0ull};
    unsigned long long* d_out = (unsigned long long*)(s->d_err + 4);
    {
        PhaseScope ps(ctx, "rows_check");
        ZK_HIP(ctx, hipMemcpyAsync(d_out, h, 16, hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_rows_check, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const Fr*)d_a, (const Fr*)d_b, (const Fr*)d_c, n, d_out);
        ZK_KERNEL_CHECK(ctx);
    }
    ZK_HIP(ctx, hipMemcpyAsync(h, d_out, 16, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h[0]) { ctx->err = "solver: " + std::to_string(h[0]) + " constraints are not satisfied, the first one is #" + std::to_string(h[1]); return ZKPOR_E_STATE; }
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((s ? zk::solver_ctx(s) : nullptr))

/* the ASYNC instructions of the NEXT proof (the two CEX commitments: 834 chained permutations each, ~0.2 s of one wave) started on the side
 * stream while the current proof still runs its prove tail: d_w_next holds the next assignment (wire 0 = ONE, then the inputs) and must be the
 * vector the next zkpor_solver_start_dev is given — that run then skips them and joins the side stream in front of its last level.  They must
 * sit in the first level (they read inputs only).  One prefetch at a time. */
int32_t zkpor_solver_prefetch_dev(zkpor_solver* s, void* d_w_next, size_t n_inputs) try {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s || !d_w_next) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = s->ctx;
    if (n_inputs == 0 || n_inputs > s->r1cs->n_wires) { ctx->err = "solver: the assignment must hold 1 + nPublic + nSecret elements"; return ZKPOR_E_ARG; }
    if (s->asyncs.empty()) return ZKPOR_OK;
    ZK_TRY(solver_side_queues(s));
    if (s->plan.empty() || s->plan[0].n_posa != s->asyncs.size()) { ctx->err = "solver: an ASYNC instruction outside the first level cannot be prefetched"; return ZKPOR_E_STATE; }
    if (s->side_busy) {
        if (s->running) { ctx->err = "solver: the side stream still runs a chain of the run in progress (one prefetch at a time, after the current run's last level)"; return ZKPOR_E_STATE; }
        // an earlier prefetch nobody consumed (the caller dropped that proof): it is abandoned — wait for its chain, then start this one
        ZK_HIP(ctx, hipStreamSynchronize(s->side));
        s->side_busy = false; s->prefetched_w = nullptr;
    }
    // a prefetch evaluates the calls' inputs from the bare assignment (every wire taken as known): legal only when they read input wires
    if (s->async_max_wire >= n_inputs) { ctx->err = "solver: an ASYNC instruction reads wire " + std::to_string(s->async_max_wire) + ", not an input (the assignment holds " + std::to_string(n_inputs) + " elements): it cannot be prefetched"; return ZKPOR_E_STATE; }
    const size_t nw = s->r1cs->n_wires;
    if (!s->d_ones) {
        ZK_HIP(ctx, hipMalloc((void**)&s->d_ones, nw));
        ZK_HIP(ctx, hipMemsetAsync(s->d_ones, 1, nw, ctx->stream));      // ordered in front of the fork event below
        ZK_HIP(ctx, hipMalloc((void**)&s->d_perr, 16));
    }
    const SolverProg P = prog_of(s);
    const LevelPlan& L = s->plan[0];
    ZK_HIP(ctx, hipEventRecord(s->ev_fork, ctx->stream));          // the inputs were put there through this context
    ZK_HIP(ctx, hipStreamWaitEvent(s->side, s->ev_fork, 0));
    ZK_HIP(ctx, hipMemsetAsync(s->d_perr, 0, 16, s->side));
    for (u32 k = 0; k < L.n_posa; ++k) {
        const BigHint& a = s->asyncs[L.posa_first + k];
        hipLaunchKernelGGL(k_hint_inputs, dim3((a.n_in + 255u) / 256u), dim3(256), 0, s->side, P, a.ins, s->d_offs + a.offs_base, a.n_in, (const Fr*)d_w_next, s->d_ones,
                           s->d_pre + a.nb_q, s->d_perr);
    }
    ZK_TRY(gadget_poseidon_launch(ctx, s->side, P, s->d_level_instr + L.lo + L.n_gen + L.n_chk + L.n_pos, L.n_posa, (Fr*)d_w_next, s->d_ones, s->d_perr, s->d_pre, s->d_pre_off + L.posa_first, nullptr, nullptr, nullptr));
    s->side_busy = true;
    s->prefetched_w = d_w_next;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((s ? zk::solver_ctx(s) : nullptr))

int32_t zkpor_solver_resume_dev(zkpor_solver* s, uint32_t* paused_instr) try {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s || !paused_instr) return ZKPOR_E_ARG;
    if (!s->running) { s->ctx->err = "solver: no run to resume"; return ZKPOR_E_STATE; }
    return solver_advance(s, paused_instr);
} ZK_ABI_CATCH_IN((s ? zk::solver_ctx(s) : nullptr))

// evaluates the inputs of the external hint the run is paused at into d_out (device, n_in elements); synchronous
static int32_t hint_inputs_to(zkpor_solver* s, uint32_t instr, Fr* d_out) {
    zkpor_ctx* ctx = s->ctx;
    const BigHint& b = s->externals.at(instr);
    if (b.n_in == 0) return ZKPOR_OK;
    u32 h[2] = {0, 0};
    hipLaunchKernelGGL(k_hint_inputs, dim3((b.n_in + 255u) / 256u), dim3(256), 0, ctx->stream, prog_of(s), instr, s->d_offs + b.offs_base, b.n_in, (const Fr*)s->d_w, s->known, d_out, s->d_err);
    ZK_KERNEL_CHECK(ctx);
    ZK_HIP(ctx, hipMemcpyAsync(h, s->d_err, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (h[0]) { s->running = false; ctx->err = std::string("solver: ") + solver_error_text(h[0]) + " at instruction " + std::to_string(h[1]); return ZKPOR_E_STATE; }
    return ZKPOR_OK;
}

int32_t zkpor_solver_external_inputs(zkpor_solver* s, uint32_t instr, uint64_t* in_values, size_t capacity, size_t* n_in, size_t* n_out) try {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s || !n_in || !n_out) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = s->ctx;
    if (!s->running || s->pending.empty() || s->pending.front() != instr) { ctx->err = "solver: not paused at this instruction"; return ZKPOR_E_STATE; }
    const uint32_t* cd = s->view.calldata + s->view.arg[instr];
    *n_in = cd[1]; *n_out = cd[2];
    if (!in_values) return ZKPOR_OK;              // sizes only
    if (capacity < cd[1]) { ctx->err = "solver: the buffer holds fewer elements than the hint has inputs"; return ZKPOR_E_ARG; }
    if (cd[1] == 0) return ZKPOR_OK;
    ZK_TRY(tmp_reserve(s, cd[1]));                // not the staging area: the caller's d_w may live there (zkpor_prove_inputs)
    ZK_TRY(hint_inputs_to(s, instr, s->d_tmp));
    ZK_HIP(ctx, hipMemcpyAsync(in_values, s->d_tmp, (size_t)cd[1] * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((s ? zk::solver_ctx(s) : nullptr))

/* the same into device memory (d_out: capacity elements): the committed wires of a BSB22 commitment go straight to zkpor_commit_dev */
int32_t zkpor_solver_external_inputs_dev(zkpor_solver* s, uint32_t instr, void* d_out, size_t capacity) try {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s || !d_out) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = s->ctx;
    if (!s->running || s->pending.empty() || s->pending.front() != instr) { ctx->err = "solver: not paused at this instruction"; return ZKPOR_E_STATE; }
    const uint32_t* cd = s->view.calldata + s->view.arg[instr];
    if (capacity < cd[1]) { ctx->err = "solver: the buffer holds fewer elements than the hint has inputs"; return ZKPOR_E_ARG; }
    return hint_inputs_to(s, instr, (Fr*)d_out);
} ZK_ABI_CATCH_IN((s ? zk::solver_ctx(s) : nullptr))

int32_t zkpor_solver_external_outputs(zkpor_solver* s, uint32_t instr, const uint64_t* out_values, size_t n_out) try {
    ZK_ENTER(s ? s->ctx->device : -1);
    if (!s || (!out_values && n_out)) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = s->ctx;
    if (!s->running || s->pending.empty() || s->pending.front() != instr) { ctx->err = "solver: not paused at this instruction"; return ZKPOR_E_STATE; }
    const uint32_t* cd = s->view.calldata + s->view.arg[instr];
    if (n_out != cd[2]) { ctx->err = "solver: the hint has " + std::to_string(cd[2]) + " outputs"; return ZKPOR_E_ARG; }
    if (n_out) {
        ZK_TRY(tmp_reserve(s, n_out));
        ZK_HIP(ctx, hipMemcpyAsync(s->d_tmp, out_values, n_out * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_hint_outputs, dim3(1), dim3(256), 0, ctx->stream, prog_of(s), instr, (const Fr*)s->d_tmp, (Fr*)s->d_w, s->known);
        ZK_KERNEL_CHECK(ctx);
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));   // out_values is the caller's memory
    }
    s->pending.erase(s->pending.begin());
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((s ? zk::solver_ctx(s) : nullptr))

/* host-buffer form: inputs in, full wire vector out; pre-filled wires (the device generators' in a real run) given as (id, value) pairs;
 * external hints are NOT served here (the call fails with ZKPOR_E_STATE when it meets one) */
int32_t zkpor_solver_run(zkpor_solver* s, const uint64_t* inputs, size_t n_inputs, const uint32_t* pre_ids, const uint64_t* pre_vals, size_t n_pre,
                         uint64_t* w_out, uint64_t stats[4]) try {
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
    struct Tmp { void* p = nullptr; ~Tmp() { if (p) (void)hipFree(p); } } tw, tk;   // freed on every path
    ZK_HIP(ctx, hipMalloc(&tw.p, nw * sizeof(Fr)));
    ZK_HIP(ctx, hipMalloc(&tk.p, nw));
    ZK_TRY(zk::h2d_sync(ctx, tw.p, hw.data(), nw * sizeof(Fr)));
    ZK_TRY(zk::h2d_sync(ctx, tk.p, hk.data(), nw));
    uint32_t paused = 0xffffffffu;
    int32_t rc = zkpor_solver_start_dev(s, tw.p, n_inputs, (uint8_t*)tk.p, &paused);
    if (rc == ZKPOR_OK && paused != 0xffffffffu) { s->running = false; ctx->err = "solver: external hint at instruction " + std::to_string(paused) + " (serve it through zkpor_solver_start_dev / _external_* / _resume_dev)"; rc = ZKPOR_E_STATE; }
    if (s->side_busy) { (void)hipStreamSynchronize(s->side); s->side_busy = false; }
    if (rc == ZKPOR_OK) ZK_HIP(ctx, hipMemcpy(w_out, tw.p, nw * sizeof(Fr), hipMemcpyDeviceToHost));
    if (stats) { stats[0] = s->n_r1c; stats[1] = s->n_hint + s->n_lookup + s->n_poseidon; stats[2] = s->n_skip; stats[3] = s->launches; }
    return rc;
} ZK_ABI_CATCH_IN((s ? zk::solver_ctx(s) : nullptr))

}  // extern "C"
