// Levelized executor of a compiled R1CS's solver program — the host side of SURVEY.md §8 f4.
//
// What it replaces: r1cs.Solve inside groth16.Prove (src/prover/prover/prover.go:269; gnark constraint/bn254/solver.go, 3P): gnark's
// solver walks `Levels [][]int` — sets of mutually independent instructions — and, per instruction, either solves ONE constraint for its
// single unknown wire or calls a hint (circuit.IntegerDivision registered at prover.go:68, the std hints behind ToBinary / IsZero / range
// checks / lookups, and the BSB22 commitment placeholder).  go/export_solver (source; needs a box with Go) flattens that program — levels,
// instruction kinds, hint names, hint inputs as linear expressions, output wires — into the container read here; the matrices come from
// the r1cs container (host/r1cs_file.hpp).  The executor runs the levels on all host threads, leaves out the wires the DEVICE generators
// produce (zkpor_witgen_*: their instructions are marked `skip`, their wires arrive pre-filled) and returns the full wire vector
// w and the evaluations a, b, c that zkpor_prove_tail consumes.
//
// Container "ZKPSOLV\x01": host/solver_file.hpp.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <functional>
#include <algorithm>
#include <map>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "fr_host.hpp"
#include "poseidon_host.hpp"
#include "r1cs_file.hpp"
#include "solver_file.hpp"

namespace zkpor_host {

// A hint works on field elements; integer-valued ones convert through U256 (gnark hands *big.Int in [0, r)).  Return non-zero to fail.
typedef std::function<int(const std::vector<FrH>& in, std::vector<FrH>& out)> HintFn;

struct HintRegistry {
    std::map<std::string, HintFn> by_name;
    std::set<std::string> inverse_names;   // hints that are exactly out[0] = 1 / in[0] (0 for 0): the executor batches their inversions per level
    // the hints whose semantics are fixed by their source text and that BatchCreateUserCircuit reaches:
    static HintRegistry Standard() {
        HintRegistry r;
        // circuit.IntegerDivision (circuit/utils.go:103-110): out[0], out[1] = DivMod(in[0], in[1])
        r.by_name["IntegerDivision"] = [](const std::vector<FrH>& in, std::vector<FrH>& out) {
            if (in.size() != 2 || out.size() != 2) return 1;
            U256 a = U256::of(in[0]), b = U256::of(in[1]);
            if (b.is_zero()) return 2;                    // big.Int.DivMod panics on a zero divisor
            U256 q, m;
            U256::divmod(a, b, &q, &m);
            out[0] = q.fr(); out[1] = m.fr();
            return 0;
        };
        // gnark std/math/bits NBits (behind api.ToBinary): out[i] = bit i of in[0]
        r.by_name["NBits"] = [](const std::vector<FrH>& in, std::vector<FrH>& out) {
            if (in.size() != 1) return 1;
            U256 a = U256::of(in[0]);
            for (size_t i = 0; i < out.size(); ++i) out[i] = a.bit((int)i) ? FrH::one() : FrH::zero();
            return 0;
        };
        // gnark solver.InvZeroHint (behind api.IsZero / api.Inverse): 1 / in[0], or 0
        r.by_name["InvZero"] = [](const std::vector<FrH>& in, std::vector<FrH>& out) {
            if (in.size() != 1 || out.size() != 1) return 1;
            out[0] = FrH::inv(in[0]);
            return 0;
        };
        // gnark std/rangecheck DecomposeHint: in = (varSize, limbSize, value) -> out = limbs of limbSize bits, little-endian
        r.by_name["DecomposeHint"] = [](const std::vector<FrH>& in, std::vector<FrH>& out) {
            if (in.size() != 3) return 1;
            U256 vs = U256::of(in[0]), ls = U256::of(in[1]), v = U256::of(in[2]);
            if (ls.w[0] == 0 || ls.w[0] > 64 || (ls.w[1] | ls.w[2] | ls.w[3]) || vs.w[0] > 256) return 2;
            const int limb = (int)ls.w[0];
            if ((int)out.size() * limb < v.bitlen()) return 3;   // the value does not fit the requested limbs: the range check must fail
            for (size_t i = 0; i < out.size(); ++i) out[i] = FrH::from_u64(v.bits((int)i * limb, limb));
            return 0;
        };
        // gnark std/internal/logderivarg countHint (behind every range check and lookup table): in = (nbTable, nbCols, the table's rows,
        // the queries' rows) -> per table row how many queries equal it; a query that matches no row fails the hint
        r.by_name["countHint"] = [](const std::vector<FrH>& in, std::vector<FrH>& out) {
            if (in.size() < 2) return 1;
            const U256 nt = U256::of(in[0]), nc = U256::of(in[1]);
            if ((nt.w[1] | nt.w[2] | nt.w[3] | nc.w[1] | nc.w[2] | nc.w[3]) || nc.w[0] == 0 || nt.w[0] != out.size()) return 2;
            const size_t nb_table = nt.w[0], nb_col = nc.w[0];
            if (in.size() < 2 + nb_table * nb_col || (in.size() - 2 - nb_table * nb_col) % nb_col) return 3;
            const size_t nb_q = (in.size() - 2 - nb_table * nb_col) / nb_col;
            std::map<std::vector<uint64_t>, size_t> row_of;
            std::vector<uint64_t> key(4 * nb_col);
            auto key_at = [&](size_t base) { for (size_t c = 0; c < nb_col; ++c) memcpy(&key[4 * c], in[base + c].v, 32); };
            for (size_t i = 0; i < nb_table; ++i) { key_at(2 + i * nb_col); row_of.emplace(key, i); }   // the first of equal rows counts (gnark: map by row bytes)
            std::vector<uint64_t> cnt(nb_table, 0);
            for (size_t q = 0; q < nb_q; ++q) {
                key_at(2 + (nb_table + q) * nb_col);
                auto it = row_of.find(key);
                if (it == row_of.end()) return 4;                                                      // "query element not in table"
                ++cnt[it->second];
            }
            for (size_t i = 0; i < nb_table; ++i) out[i] = FrH::from_u64(cnt[i]);
            return 0;
        };
        // gnark registers its hints under their Go function names; the exporter keeps the last path element
        r.by_name["nBits"] = r.by_name["NBits"];
        r.by_name["InvZeroHint"] = r.by_name["InvZero"];
        r.inverse_names = {"InvZero", "InvZeroHint"};
        return r;
    }
};

struct SolveResult {
    std::vector<uint64_t> w, a, b, c;   // n_wires / n_constraints x 4 limbs (Montgomery), the form zkpor_prove_tail takes
    uint64_t solved_constraints = 0, hint_calls = 0, skipped = 0;
    double ms_setup = 0, ms_levels = 0, ms_rows = 0;   // wall time of the three parts of the call (thread 0's view)
    int threads_used = 0;
};

namespace detail {
inline FrH coeff(const R1csFileView& r, uint32_t id) { FrH c; memcpy(c.v, r.coeff + 4 * (size_t)id, 32); return c; }

// coefficient classes: almost every coefficient gnark emits is 1 or -1 (the rest are the powers of two of bit / limb recompositions and a few
// circuit constants) — a term with one of those costs an addition instead of a Montgomery product
enum : uint8_t { C_GENERAL = 0, C_ONE = 1, C_MINUS_ONE = 2, C_ZERO = 3 };
inline std::vector<uint8_t> classify_coefficients(const R1csFileView& r) {
    std::vector<uint8_t> cls(r.n_coeff, C_GENERAL);
    const FrH one = FrH::one(), mone = FrH::neg(FrH::one());
    for (uint64_t i = 0; i < r.n_coeff; ++i) {
        const FrH c = coeff(r, (uint32_t)i);
        cls[i] = c == one ? C_ONE : (c == mone ? C_MINUS_ONE : (c.is_zero() ? C_ZERO : C_GENERAL));
    }
    return cls;
}
inline void add_term(FrH& acc, uint8_t cls, const R1csFileView& r, uint32_t cid, const FrH& x) {
    if (cls == C_ONE) acc = FrH::add(acc, x);
    else if (cls == C_MINUS_ONE) acc = FrH::sub(acc, x);
    else if (cls == C_GENERAL) acc = FrH::add(acc, FrH::mul(coeff(r, cid), x));
}
// value of one side with at most one unknown wire: returns sum of known terms; *unk / *unk_coeff describe the unknown term (if any);
// a second unknown wire sets *two
inline FrH side(const R1csFileView& r, const uint8_t* cls, int which, size_t row, const FrH* w, const uint8_t* known, int64_t* unk, FrH* unk_coeff, bool* two) {
    FrH acc = FrH::zero();
    for (uint64_t k = r.row_ptr[which][row]; k < r.row_ptr[which][row + 1]; ++k) {
        const uint32_t wid = r.wire_ids[which][k], cid = r.coeff_ids[which][k];
        if (known[wid]) add_term(acc, cls[cid], r, cid, w[wid]);
        else {
            const FrH c = coeff(r, cid);
            if (*unk < 0 || *unk == (int64_t)wid) { *unk_coeff = (*unk < 0) ? c : FrH::add(*unk_coeff, c); *unk = wid; }
            else *two = true;
        }
    }
    return acc;
}

// the threads of one Solve call meet here after every level: a short spin (levels are microseconds to milliseconds apart), then a sleep on a
// condition variable — the box may hold fewer free cores than threads, and a spinning waiter would then keep the last worker off its core
struct LevelBarrier {
    const int n;
    std::atomic<int> arrived{0};
    std::atomic<uint64_t> generation{0};
    std::mutex m;
    std::condition_variable cv;
    explicit LevelBarrier(int n_) : n(n_) {}
    void wait() {
        if (n == 1) return;
        const uint64_t g = generation.load(std::memory_order_acquire);
        if (arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == n) {
            arrived.store(0, std::memory_order_relaxed);
            { std::lock_guard<std::mutex> lk(m); generation.fetch_add(1, std::memory_order_release); }
            cv.notify_all();
            return;
        }
        for (int spin = 0; spin < 4000; ++spin)
            if (generation.load(std::memory_order_acquire) != g) return;
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return generation.load(std::memory_order_acquire) != g; });
    }
};

// a wire whose value is a quotient: every thread collects the quotients of its share of a level and inverts all their denominators with ONE
// field inversion (Montgomery's trick: 3 products per denominator instead of ~380) — the instructions of a level are independent of each other,
// so nothing in the level can be waiting for these wires
struct Quotient { uint32_t wire; FrH num, den; };
inline void resolve_quotients(std::vector<Quotient>& q, std::vector<FrH>& scratch, FrH* w, uint8_t* known) {
    const size_t n = q.size();
    if (!n) return;
    scratch.resize(n);
    FrH run = FrH::one();
    for (size_t i = 0; i < n; ++i) { scratch[i] = run; run = FrH::mul(run, q[i].den); }   // scratch[i] = den_0 .. den_{i-1}
    FrH inv = FrH::inv(run);
    for (size_t i = n; i-- > 0;) {
        const FrH di = FrH::mul(inv, scratch[i]);                                          // 1 / den_i
        inv = FrH::mul(inv, q[i].den);
        w[q[i].wire] = FrH::mul(q[i].num, di);
        known[q[i].wire] = 1;
    }
    q.clear();
}
}  // namespace detail

// inputs: the assigned part of the wire vector in gnark's order (wire 0 = ONE, then public, then secret): n_public + n_secret elements.
// prefilled: optional (wire id, value) pairs produced elsewhere (the device generators), taken as known from the start.
// want_abc false: only w is produced (a, b, c are then evaluated on the device from w: zkpor_prove_r1cs) and the row check is skipped.
// Returns 0, or a non-zero code with `err` set; on success every wire is assigned and (want_abc) every constraint holds (checked).
//
// Execution: `threads` workers live for the whole call and meet at a barrier after every level; each takes a contiguous share of the level's
// instructions, evaluates them, assigns what needs no division at once and resolves its divisions (and the InvZero hints) with one inversion.
inline int SolveLevelized(const R1csFileView& r, const SolverView& s, const uint64_t* inputs, size_t n_inputs, const HintRegistry& hints,
                          const std::vector<std::pair<uint32_t, FrH>>& prefilled, int threads, SolveResult* out, std::string* err,
                          bool want_abc = true) {
    auto fail = [&](int code, const std::string& m) { if (err) *err = "solver: " + m; return code; };
    if (n_inputs != r.n_public + r.n_secret) return fail(1, "the assignment must hold nPublic + nSecret elements");
    const auto t_begin = std::chrono::steady_clock::now();
    auto ms_since = [](std::chrono::steady_clock::time_point a) { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a).count(); };
    const size_t nw = r.n_wires, nc = r.n_constraints;
    out->w.assign(nw * 4, 0);
    FrH* w = reinterpret_cast<FrH*>(out->w.data());
    std::vector<uint8_t> known_v(nw, 0);
    uint8_t* known = known_v.data();
    memcpy(w, inputs, n_inputs * 32);
    memset(known, 1, n_inputs);
    for (auto& pv : prefilled) { if (pv.first >= nw) return fail(1, "prefilled wire out of range"); w[pv.first] = pv.second; known[pv.first] = 1; }
    const std::vector<uint8_t> cls_v = detail::classify_coefficients(r);
    const uint8_t* cls = cls_v.data();
    std::vector<HintFn> fn(s.hint_names.size());
    std::vector<uint8_t> is_inverse(s.hint_names.size(), 0);
    for (size_t i = 0; i < fn.size(); ++i) {
        auto it = hints.by_name.find(s.hint_names[i]);
        if (it != hints.by_name.end()) fn[i] = it->second;   // a missing one only matters if an instruction calls it
        is_inverse[i] = fn[i] && hints.inverse_names.count(s.hint_names[i]) ? 1 : 0;
    }
    {   // the gadget instructions' call data is checked once, up front (a table's entries once per table, not once per lookup)
        std::set<std::pair<uint32_t, uint32_t>> tables_ok;
        for (uint64_t i = 0; i < s.n_instructions; ++i) {
            const uint32_t kind = InstrKind(s, i), arg = s.arg[i];
            if (kind == INSTR_POSEIDON && !CheckPoseidonShape(s, arg, nw, r.n_coeff)) return fail(20, "the call data of Poseidon instruction " + std::to_string(i) + " is malformed");
            if (kind == INSTR_LOOKUP) {
                if ((uint64_t)arg + 4 > s.n_calldata) return fail(20, "the call data of lookup " + std::to_string(i) + " is malformed");
                const std::pair<uint32_t, uint32_t> key{s.calldata[arg], s.calldata[arg + 1]};
                if (tables_ok.count(key)) {
                    // only the queries: walk them
                    uint64_t p = (uint64_t)arg + 4;
                    bool ok = (uint64_t)s.calldata[arg + 3] + s.calldata[arg + 2] <= nw;
                    for (uint32_t q = 0; ok && q < s.calldata[arg + 2]; ++q) {
                        ok = p < s.n_calldata;
                        if (!ok) break;
                        const uint64_t nt = s.calldata[p++];
                        ok = p + 2 * nt <= s.n_calldata;
                        for (uint64_t t = 0; ok && t < nt; ++t) ok = s.calldata[p + 2 * t] < r.n_coeff && s.calldata[p + 1 + 2 * t] < nw;
                        p += 2 * nt;
                    }
                    if (!ok) return fail(20, "the call data of lookup " + std::to_string(i) + " is malformed");
                } else {
                    if (!CheckLookupShape(s, arg, nw, r.n_coeff)) return fail(20, "the call data of lookup " + std::to_string(i) + " is malformed");
                    tables_ok.insert(key);
                }
            }
        }
    }
    if (threads < 1) threads = 1;
    uint64_t widest = 0;
    for (uint64_t l = 0; l < s.n_levels; ++l) widest = std::max<uint64_t>(widest, s.level_ptr[l + 1] - s.level_ptr[l]);
    // a thread per 256 instructions of the widest level (or per 1024 rows of the final evaluation) at most
    const uint64_t useful = std::max<uint64_t>((widest + 255) / 256, want_abc ? (nc + 1023) / 1024 : 1);
    const int nt = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)threads, useful));
    std::atomic<int> bad{0};
    std::string bad_msg;
    std::atomic<uint64_t> n_r1c{0}, n_hint{0}, n_skip{0};
    std::atomic<int64_t> first_bad_row{-1};
    if (want_abc) { out->a.resize(nc * 4); out->b.resize(nc * 4); out->c.resize(nc * 4); }

    struct Scratch { std::vector<detail::Quotient> q; std::vector<FrH> prod, in, o; uint64_t cnt[3] = {0, 0, 0}; };
    auto run_instr = [&](uint32_t ins, Scratch& sc) -> int {
        const uint32_t kind = InstrKind(s, ins), arg = s.arg[ins];
        if (kind == INSTR_SKIP) { ++sc.cnt[2]; return 0; }
        auto eval_le = [&](const uint32_t* cd, uint64_t& p, FrH* out) -> int {   // nTerms, (coeffId, wireId)...; shapes were checked up front
            const uint32_t nterms = cd[p++];
            FrH acc = FrH::zero();
            for (uint32_t k = 0; k < nterms; ++k) {
                const uint32_t cid = cd[p++], wid = cd[p++];
                if (!known[wid]) return 23;
                detail::add_term(acc, cls[cid], r, cid, w[wid]);
            }
            *out = acc;
            return 0;
        };
        if (kind == INSTR_LOOKUP) {              // outputs = entry[index] (gnark BlueprintLookupHint.Solve); shapes were checked before the run
            const uint32_t* cd = s.calldata + arg;
            const uint32_t* tb = s.calldata + cd[0];
            uint64_t p = 4;
            for (uint32_t q = 0; q < cd[2]; ++q) {
                FrH ix;
                if (int rc = eval_le(cd, p, &ix)) return rc;
                const U256 i = U256::of(ix);
                if ((i.w[1] | i.w[2] | i.w[3]) || i.w[0] >= cd[1]) return 24;   // "lookup query too large"
                uint64_t pe = tb[1 + i.w[0]];
                FrH v;
                if (int rc = eval_le(tb, pe, &v)) return rc;
                w[cd[3] + q] = v; known[cd[3] + q] = 1;
            }
            ++sc.cnt[1];
            return 0;
        }
        if (kind == INSTR_POSEIDON) {            // the whole sponge: every S-box's three product wires
            const uint32_t* cd = s.calldata + arg;
            sc.in.resize(cd[0]); sc.o.resize(cd[2]);
            uint64_t p = POSEIDON_HDR;
            for (uint32_t i = 0; i < cd[0]; ++i) if (int rc = eval_le(cd, p, &sc.in[i])) return rc;
            PosSponge(sc.in.data(), cd[0], sc.o.data(), (int)(cd[3] & 0xff), (int)((cd[3] >> 8) & 0xff));
            for (uint32_t i = 0; i < cd[2]; ++i) { w[cd[1] + i] = sc.o[i]; known[cd[1] + i] = 1; }
            ++sc.cnt[1];
            return 0;
        }
        if (kind == INSTR_R1C) {
            if (arg >= nc) return 10;
            int64_t unk[3] = {-1, -1, -1};
            FrH uc[3];
            bool two = false;
            FrH v[3];
            for (int m = 0; m < 3; ++m) v[m] = detail::side(r, cls, m, arg, w, known, &unk[m], &uc[m], &two);
            const int n_unk = (unk[0] >= 0) + (unk[1] >= 0) + (unk[2] >= 0);
            if (two || n_unk > 1) return 11;                     // not solvable at this level: the export's levels are wrong
            ++sc.cnt[0];
            if (n_unk == 0) return FrH::mul(v[0], v[1]) == v[2] ? 0 : 12;   // an assertion
            const int which = unk[0] >= 0 ? 0 : (unk[1] >= 0 ? 1 : 2);
            const FrH& c = uc[which];
            if (c.is_zero()) return 13;
            const bool c_one = c == FrH::one(), c_mone = !c_one && FrH::neg(c) == FrH::one();
            const uint32_t x = (uint32_t)unk[which];
            if (which == 2) {
                // O_known + c x = L R
                FrH num = FrH::sub(FrH::mul(v[0], v[1]), v[2]);
                if (c_one) { w[x] = num; known[x] = 1; }
                else if (c_mone) { w[x] = FrH::neg(num); known[x] = 1; }
                else sc.q.push_back({x, num, c});
                return 0;
            }
            // (L_known + c x) R = O  =>  x = (O - L_known R) / (c R)
            const FrH& other = v[1 - which];
            if (other.is_zero()) {                               // gnark solveR1C: nothing to divide by — the constraint must already hold
                if (!v[2].is_zero()) return 14;                  // (L_known + c x) * 0 = O needs O = 0 (else: "division by zero") ...
                w[x] = FrH::zero(); known[x] = 1;                // ... and then the wire is left at 0: api.DivUnchecked(0, 0) = 0
                return 0;
            }
            FrH num = FrH::sub(v[2], FrH::mul(v[which], other));
            if (c_one) sc.q.push_back({x, num, other});
            else if (c_mone) sc.q.push_back({x, FrH::neg(num), other});
            else sc.q.push_back({x, num, FrH::mul(c, other)});
            return 0;
        }
        // hint
        const uint32_t* cd = s.calldata + arg;
        if (arg + 3 > s.n_calldata) return 20;
        const uint32_t name = cd[0], n_in = cd[1], n_out = cd[2];
        if (name >= fn.size() || !fn[name]) return 21;
        // the shape words come from a file: they are checked against the call data's length before anything is sized or read by them
        const uint64_t room = s.n_calldata - arg - 3;
        if (n_out > room || n_in > room - n_out) return 20;     // every input takes at least its term count
        size_t p = 3 + (size_t)n_out;
        sc.in.resize(n_in); sc.o.resize(n_out);
        for (uint32_t i = 0; i < n_in; ++i) {
            if (arg + p >= s.n_calldata) return 20;
            const uint32_t nterms = cd[p++];
            if (2ull * nterms > s.n_calldata - arg - p) return 20;
            FrH acc = FrH::zero();
            for (uint32_t k = 0; k < nterms; ++k) {
                const uint32_t cid = cd[p++], wid = cd[p++];
                if (wid >= nw || cid >= r.n_coeff) return 22;
                if (!known[wid]) return 23;
                detail::add_term(acc, cls[cid], r, cid, w[wid]);
            }
            sc.in[i] = acc;
        }
        ++sc.cnt[1];
        for (uint32_t i = 0; i < n_out; ++i) if (cd[3 + i] >= nw) return 22;
        if (is_inverse[name] && n_in == 1 && n_out == 1 && !sc.in[0].is_zero()) {   // 1 / x joins the level's shared inversion
            sc.q.push_back({cd[3], FrH::one(), sc.in[0]});
            return 0;
        }
        if (fn[name](sc.in, sc.o) != 0) return 24;
        for (uint32_t i = 0; i < n_out; ++i) { const uint32_t wid = cd[3 + i]; w[wid] = sc.o[i]; known[wid] = 1; }
        return 0;
    };
    auto eval_rows = [&](size_t a0, size_t a1) {
        for (size_t row = a0; row < a1; ++row) {
            FrH v[3];
            for (int m = 0; m < 3; ++m) {
                FrH acc = FrH::zero();
                for (uint64_t k = r.row_ptr[m][row]; k < r.row_ptr[m][row + 1]; ++k) {
                    const uint32_t cid = r.coeff_ids[m][k];
                    detail::add_term(acc, cls[cid], r, cid, w[r.wire_ids[m][k]]);
                }
                v[m] = acc;
            }
            memcpy(&out->a[4 * row], v[0].v, 32); memcpy(&out->b[4 * row], v[1].v, 32); memcpy(&out->c[4 * row], v[2].v, 32);
            if (!(FrH::mul(v[0], v[1]) == v[2])) { int64_t e = -1; first_bad_row.compare_exchange_strong(e, (int64_t)row); }
        }
    };
    detail::LevelBarrier barrier(nt);
    std::atomic<int64_t> unassigned{-1};
    out->threads_used = nt;
    out->ms_setup = ms_since(t_begin);
    const auto t_levels = std::chrono::steady_clock::now();
    std::chrono::steady_clock::time_point t_rows = t_levels;
    auto worker = [&](int t) {
        Scratch sc;
        for (uint64_t l = 0; l < s.n_levels; ++l) {
            if (!bad.load(std::memory_order_relaxed)) {
                const uint64_t lo = s.level_ptr[l], n = s.level_ptr[l + 1] - lo;
                const int share = (int)std::min<uint64_t>((uint64_t)nt, (n + 255) / 256);   // small levels are not worth a thread each
                if (t < share) {
                    const uint64_t a0 = lo + n * t / share, a1 = lo + n * (t + 1) / share;
                    for (uint64_t i = a0; i < a1; ++i) {
                        const int rc = run_instr(s.level_instr[i], sc);
                        if (rc) {
                            int exp = 0;
                            if (bad.compare_exchange_strong(exp, rc)) bad_msg = "instruction " + std::to_string(s.level_instr[i]) + " of level " + std::to_string(l);
                            break;
                        }
                    }
                    detail::resolve_quotients(sc.q, sc.prod, w, known);
                }
            }
            barrier.wait();
        }
        n_r1c += sc.cnt[0]; n_hint += sc.cnt[1]; n_skip += sc.cnt[2];
        if (bad.load()) return;
        for (size_t i = nw * (size_t)t / nt; i < nw * (size_t)(t + 1) / nt; ++i)
            if (!known[i]) { int64_t e = -1; unassigned.compare_exchange_strong(e, (int64_t)i); break; }
        barrier.wait();
        if (t == 0) { out->ms_levels = ms_since(t_levels); t_rows = std::chrono::steady_clock::now(); }
        if (unassigned.load() >= 0 || !want_abc) return;
        eval_rows(nc * (size_t)t / nt, nc * (size_t)(t + 1) / nt);   // a, b, c and the final check of every constraint
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(worker, t);
        worker(0);
        for (auto& t : th) t.join();
    }
    out->ms_rows = ms_since(t_rows);
    if (bad) {
        static const std::map<int, const char*> why = {{10, "constraint index out of range"}, {11, "more than one unknown wire (wrong level order)"},
            {12, "constraint not satisfied"}, {13, "unknown wire with a zero coefficient"}, {14, "division by zero"}, {20, "call data out of range"},
            {21, "no native implementation for this hint"}, {22, "wire or coefficient id out of range"}, {23, "hint input not solved yet"}, {24, "hint failed"}};
        auto it = why.find(bad.load());
        return fail(bad.load(), std::string(it != why.end() ? it->second : "error") + " at " + bad_msg);
    }
    if (unassigned.load() >= 0) return fail(30, "wire " + std::to_string(unassigned.load()) + " was never assigned");
    if (first_bad_row >= 0) return fail(31, "constraint " + std::to_string(first_bad_row.load()) + " does not hold for the solved wires");
    out->solved_constraints = n_r1c; out->hint_calls = n_hint; out->skipped = n_skip;
    return 0;
}

}  // namespace zkpor_host
