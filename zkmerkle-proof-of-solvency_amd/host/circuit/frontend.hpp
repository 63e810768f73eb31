// A small R1CS frontend in the shape of gnark's (frontend.API over the r1cs builder, 3P: bnb-chain/gnark v0.10.1-0.20240910145009-4b5261061f04,
// go.mod:57) — just enough of it to restate BatchCreateUserCircuit.Define (circuit/batch_create_user_circuit.go:98-323, circuit/utils.go:12-225;
// host/circuit/batch_create_user.hpp) in an image without Go, and to emit what the GPU backend consumes from a compiled system:
//   * the constraint matrices L, R, O in CSR form over a shared coefficient table (zkpor_r1cs_*, the layout of go/export_r1cs),
//   * the SOLVER PROGRAM (host/solver_file.hpp: instructions, levels, hint call data — the layout of go/export_solver),
//   * gnark's commitment info (the wires a BSB22 commitment covers, the commitment wire) and which wires appear in L / R (pk.InfinityA / B).
// It is the stand-in for `frontend.Compile(ecc.BN254, r1cs.NewBuilder, circuit)` (src/keygen/main.go:30) that this repo can run: the gadget
// expansions follow gnark's std library as recalled (3P, marked below) — constraint COUNTS and wire ORDER of the real compiler are not
// claimed, the structure (which hints, which lookups, which dependency chains, how wide the levels) is what the device executor is measured on.
//
// Linear expressions are lists of (wire, coefficient id); constants are terms on wire 0 (ONE), as in gnark.  A variable×variable
// product, an assertion, a hint or a gadget call appends one instruction whose LEVEL is 1 + the highest level among the wires it reads
// (gnark's Levels [][]int are exactly that ASAP layering).  In WITNESS MODE the builder also carries the value of every wire — an
// interpreter of the circuit, independent of both executors, that the tests compare their wire vectors with.
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../fr_host.hpp"
#include "../poseidon_host.hpp"
#include "../solver_file.hpp"

namespace zkpor_circuit {
using zkpor_host::FrH;
using zkpor_host::U256;
typedef uint32_t u32;
typedef uint64_t u64;

enum : u32 { CID_ZERO = 0, CID_ONE = 1, CID_MINUS_ONE = 2 };
enum : u32 { K_R1C = 0, K_HINT = 1, K_SKIP = 2, K_LOOKUP = 3, K_POSEIDON = 4 };
static const u32 NO_WIRE = 0xffffffffu;

struct Term { u32 wire, cid; };

// small vector of terms sorted by wire id, unique wires, no zero coefficients
struct LE {
    static const u32 INL = 4;
    Term inl[INL];
    u32 n = 0;
    std::vector<Term>* big = nullptr;
    LE() {}
    LE(const LE& o) : n(o.n) { if (o.big) big = new std::vector<Term>(*o.big); else memcpy(inl, o.inl, sizeof(Term) * n); }
    LE(LE&& o) noexcept : n(o.n), big(o.big) { if (!big) memcpy(inl, o.inl, sizeof(Term) * n); o.big = nullptr; o.n = 0; }
    LE& operator=(const LE& o) { if (this != &o) { LE t(o); swap(t); } return *this; }
    LE& operator=(LE&& o) noexcept { if (this != &o) { delete big; n = o.n; big = o.big; if (!big) memcpy(inl, o.inl, sizeof(Term) * n); o.big = nullptr; o.n = 0; } return *this; }
    ~LE() { delete big; }
    void swap(LE& o) { std::swap(n, o.n); std::swap(big, o.big); Term t[INL]; memcpy(t, inl, sizeof t); memcpy(inl, o.inl, sizeof t); memcpy(o.inl, t, sizeof t); }
    u32 size() const { return n; }
    const Term* data() const { return big ? big->data() : inl; }
    Term* data() { return big ? big->data() : inl; }
    const Term& operator[](u32 i) const { return data()[i]; }
    void clear() { delete big; big = nullptr; n = 0; }
    void push(Term t) {   // caller keeps the order
        if (!big && n < INL) { inl[n++] = t; return; }
        if (!big) { big = new std::vector<Term>(inl, inl + n); big->reserve(2 * INL); }
        big->push_back(t); ++n;
    }
    bool single_wire(u32* w) const { if (n == 1 && (*this)[0].cid == CID_ONE && (*this)[0].wire != 0) { *w = (*this)[0].wire; return true; } return false; }
};

struct FrKeyHash { size_t operator()(const FrH& a) const { return (size_t)(a.v[0] * 0x9e3779b97f4a7c15ULL ^ a.v[1] ^ (a.v[2] << 1) ^ (a.v[3] >> 3)); } };

// symbolic template of ONE Poseidon permutation of width t as the gadget lays it down: per S-box the input expression over
// {ONE, the entry lanes (first round only), the x^5 wires of earlier S-boxes}; plus the t output lanes
struct PermTemplate {
    int t = 0, n_sbox = 0;
    struct Sym { u32 sym, cid; };     // sym 0 = ONE; 1..t = entry lane sym-1; t+1+q = x^5 of S-box q
    std::vector<u32> in_ptr;          // n_sbox + 1
    std::vector<Sym> in_terms;
    std::vector<u32> out_ptr;         // t + 1
    std::vector<Sym> out_terms;
    std::vector<u32> rc0_cid;         // t: round constant ids of the first round (added to the entry expressions)
};

struct LookupTable {     // logderivlookup.Table (gnark std/lookup/logderivlookup, 3P)
    std::vector<LE> entries;
    std::vector<LE> q_ind, q_val;      // results table: (index expression, result wire)
    u64 block_off = 0;                 // offset of its entry block in the call data, written at the first Lookup after the last Insert
    size_t block_entries = 0;
    bool immutable = false;
    u32 entries_level = 0;
};

struct Compiled {   // what a compile leaves behind (moved out of the builder)
    u64 n_wires = 0, n_public = 0, n_secret = 0, n_constraints = 0;
    std::vector<FrH> coeff;
    std::vector<u64> row_ptr[3];
    std::vector<u32> cid[3], wid[3];
    std::vector<u32> kind, arg;
    std::vector<u32> check;                     // instructions that only verify their constraint (solver_file.hpp INSTR_CHECK)
    std::vector<u64> level_ptr;
    std::vector<u32> level_instr;
    std::vector<u32> calldata;
    std::vector<std::string> hint_names;
    std::vector<u32> committed;        // gnark CommitmentInfo.PrivateCommitted: sorted wire ids (the hint's inputs, the Pedersen basis order)
    u32 commitment_wire = NO_WIRE;     // CommitmentIndex
    std::vector<uint8_t> in_l, in_r;   // per wire: appears in some L / R row (a wire that does not has an infinite A / B point in the key)
    std::vector<FrH> values;           // witness mode: every wire
    std::map<std::string, u64> census; // gadget counts
};

class Builder {
public:
    // n_public counts the ONE wire (gnark's convention); inputs (witness mode only): the n_public - 1 + n_secret assigned values
    Builder(u64 n_public, u64 n_secret, const FrH* inputs_or_null, bool poseidon_native = true)
        : n_public_(n_public), n_secret_(n_secret), witness_(inputs_or_null != nullptr), poseidon_native_(poseidon_native) {
        intern(FrH::zero()); intern(FrH::one()); intern(FrH::neg(FrH::one()));
        for (auto& x : pow2_cid_) x = NO_WIRE;
        n_wires_ = n_public + n_secret;
        wire_level_.assign(n_wires_, 0);
        producer_.assign(n_wires_, NO_WIRE);
        is_bool_.assign(n_wires_, 0);
        if (witness_) {
            val_.resize(n_wires_);
            val_[0] = FrH::one();
            for (u64 i = 1; i < n_wires_; ++i) val_[i] = inputs_or_null[i - 1];
        }
        for (int m = 0; m < 3; ++m) row_ptr_[m].push_back(0);
    }
    bool witness_mode() const { return witness_; }
    u64 n_wires() const { return n_wires_; }
    u64 n_constraints() const { return row_ptr_[0].size() - 1; }
    u64 n_instructions() const { return kind_.size(); }
    // ---- coefficients
    u32 intern(const FrH& c) {
        auto it = coeff_id_.find(c);
        if (it != coeff_id_.end()) return it->second;
        const u32 id = (u32)coeff_.size();
        coeff_.push_back(c);
        coeff_id_.emplace(c, id);
        return id;
    }
    const FrH& coeff(u32 id) const { return coeff_[id]; }
    u32 cid_mul(u32 a, u32 b) {
        if (a == CID_ONE) return b;
        if (b == CID_ONE) return a;
        if (a == CID_ZERO || b == CID_ZERO) return CID_ZERO;
        if (a == CID_MINUS_ONE && b == CID_MINUS_ONE) return CID_ONE;
        return intern(FrH::mul(coeff_[a], coeff_[b]));
    }
    u32 cid_add(u32 a, u32 b) {
        if (a == CID_ZERO) return b;
        if (b == CID_ZERO) return a;
        return intern(FrH::add(coeff_[a], coeff_[b]));
    }
    u32 cid_neg(u32 a) {
        if (a == CID_ONE) return CID_MINUS_ONE;
        if (a == CID_MINUS_ONE) return CID_ONE;
        if (a == CID_ZERO) return a;
        return intern(FrH::neg(coeff_[a]));
    }
    static FrH fr_u64(u64 x) { return FrH::from_u64(x); }
    static FrH fr_pow2(int k) { uint64_t c[4] = {0, 0, 0, 0}; c[k >> 6] = (u64)1 << (k & 63); return FrH::from_canon(c); }
    u32 cid_pow2(int k) { if (pow2_cid_[k] == NO_WIRE) pow2_cid_[k] = intern(fr_pow2(k)); return pow2_cid_[k]; }
    LE scale_pow2(const LE& a, int k) { return scale_cid(a, cid_pow2(k)); }
    LE scale_cid(const LE& a, u32 kc) {
        if (kc == CID_ONE) return a;
        LE r;
        if (kc == CID_ZERO) return r;
        for (u32 i = 0; i < a.size(); ++i) r.push({a[i].wire, cid_mul(a[i].cid, kc)});
        return r;
    }

    // ---- linear expressions
    LE wire(u32 w) const { LE e; e.push({w, CID_ONE}); return e; }
    LE constant(const FrH& c) { LE e; if (!c.is_zero()) e.push({0, intern(c)}); return e; }
    LE constant(u64 c) { return LE_const_cid(cid_small(c)); }
    LE LE_const_cid(u32 cid) const { LE e; if (cid != CID_ZERO) e.push({0, cid}); return e; }
    LE input(u64 i) const { return wire((u32)i); }   // wire id of an input = its position in the assignment (ONE first)
    bool is_const(const LE& e, FrH* v = nullptr) const {
        if (e.size() == 0) { if (v) *v = FrH::zero(); return true; }
        if (e.size() == 1 && e[0].wire == 0) { if (v) *v = coeff_[e[0].cid]; return true; }
        return false;
    }
    LE add(const LE& a, const LE& b) { return merge(a, b, false); }
    LE sub(const LE& a, const LE& b) { return merge(a, b, true); }
    LE add(const LE& a, const LE& b, const LE& c) { return add(add(a, b), c); }
    LE add(const LE& a, u64 k) { return add(a, constant(k)); }
    LE neg(const LE& a) { LE r; for (u32 i = 0; i < a.size(); ++i) r.push({a[i].wire, cid_neg(a[i].cid)}); return r; }
    LE scale(const LE& a, const FrH& k) {
        LE r;
        if (k.is_zero()) return r;
        const u32 kc = intern(k);
        if (kc == CID_ONE) return a;
        for (u32 i = 0; i < a.size(); ++i) r.push({a[i].wire, cid_mul(a[i].cid, kc)});
        return r;
    }
    LE scale(const LE& a, u64 k) { return scale(a, fr_u64(k)); }
    // acc += x in place (the running sums of Define: afterCexAssets, the lookup argument's two sides)
    void add_assign(LE& acc, const LE& x) {
        if (x.size() == 0) return;
        if (acc.size() && x.size() == 1 && x[0].wire > acc[acc.size() - 1].wire) { acc.push(x[0]); return; }   // appending a newer wire: the common case
        acc = merge(acc, x, false);
    }
    FrH eval(const LE& e) const {
        FrH acc = FrH::zero();
        for (u32 i = 0; i < e.size(); ++i) {
            const u32 c = e[i].cid;
            const FrH& x = val_[e[i].wire];
            if (c == CID_ONE) acc = FrH::add(acc, x);
            else if (c == CID_MINUS_ONE) acc = FrH::sub(acc, x);
            else acc = FrH::add(acc, FrH::mul(coeff_[c], x));
        }
        return acc;
    }
    const FrH& value(u32 w) const { return val_[w]; }

    // ---- the API (gnark frontend.API, r1cs builder semantics; 3P)
    LE mul(const LE& a, const LE& b) {
        FrH k;
        if (is_const(a, &k)) return scale(b, k);
        if (is_const(b, &k)) return scale(a, k);
        const u32 x = new_wire();
        if (witness_) val_[x] = FrH::mul(eval(a), eval(b));
        emit_r1c(a, b, wire(x), x);
        ++cnt_[C_mul];
        return wire(x);
    }
    LE mul(const LE& a, u64 k) { return scale(a, k); }
    void assert_r1c(const LE& l, const LE& r, const LE& o, const char* what = "assert") {
        if (witness_ && !(FrH::mul(eval(l), eval(r)) == eval(o))) throw std::runtime_error(std::string("circuit: assertion fails on the given inputs: ") + what);
        emit_r1c(l, r, o, NO_WIRE);
        ++cnt_[C_assert];
    }
    void assert_eq(const LE& a, const LE& b, const char* what = "assert_eq") {
        FrH ka, kb;
        if (is_const(a, &ka) && is_const(b, &kb)) { if (!(ka == kb)) throw std::runtime_error(std::string("circuit: constant assertion fails: ") + what); return; }
        // gnark AssertIsEqual: (a) * 1 = b, with the non-constant side on the left
        if (is_const(a)) assert_r1c(b, constant(1), a, what); else assert_r1c(a, constant(1), b, what);
    }
    void assert_bool(const LE& a) {
        u32 w;
        if (a.single_wire(&w) && is_bool_[w]) return;           // gnark marks variables it already constrained (MarkBoolean)
        FrH k;
        if (is_const(a, &k)) { if (!(k.is_zero() || k == FrH::one())) throw std::runtime_error("circuit: constant is not boolean"); return; }
        assert_r1c(a, sub(constant(1), a), LE(), "boolean");
        if (a.single_wire(&w)) is_bool_[w] = 1;
        ++cnt_[C_assert_bool];
    }
    LE select(const LE& b, const LE& x, const LE& y) {        // b ? x : y  =  y + b (x - y)
        assert_bool(b);
        FrH k;
        if (is_const(b, &k)) return k.is_zero() ? y : x;
        return add(y, mul(b, sub(x, y)));
    }
    // hint call: n_out fresh wires (consecutive); out_values (witness mode) computed by the caller's functor
    std::vector<LE> hint(const char* name, const std::vector<LE>& inputs, u32 n_out, const std::vector<FrH>* out_values) {
        const u32 name_id = hint_name(name);
        u32 lvl = 0;
        for (auto& e : inputs) lvl = std::max(lvl, level_of(e));
        ++lvl;
        const u32 first = new_wires(n_out, lvl);
        if (witness_) for (u32 i = 0; i < n_out; ++i) val_[first + i] = (*out_values)[i];
        const u64 off = calldata_.size();
        calldata_.push_back(name_id); calldata_.push_back((u32)inputs.size()); calldata_.push_back(n_out);
        for (u32 i = 0; i < n_out; ++i) calldata_.push_back(first + i);
        for (auto& e : inputs) push_le(e);
        for (u32 i = 0; i < n_out; ++i) producer_[first + i] = (u32)kind_.size();
        push_instr(K_HINT, off, lvl);
        std::vector<LE> out;
        for (u32 i = 0; i < n_out; ++i) out.push_back(wire(first + i));
        return out;
    }
    // the same, streamed (hints with millions of inputs: the count hint of the range checker, the commitment placeholder)
    struct HintOpen { u64 off; u32 lvl, n_out, n_in_left; };
    HintOpen hint_open(const char* name, u32 n_in, u32 n_out) {
        HintOpen h{calldata_.size(), 0, n_out, n_in};
        calldata_.push_back(hint_name(name)); calldata_.push_back(n_in); calldata_.push_back(n_out);
        calldata_.resize(calldata_.size() + n_out);
        return h;
    }
    void hint_input(HintOpen& h, const LE& e) { h.lvl = std::max(h.lvl, level_of(e)); push_le(e); --h.n_in_left; }
    u32 hint_close(HintOpen& h, const FrH* out_values) {   // returns the first output wire
        if (h.n_in_left) throw std::runtime_error("circuit: hint input count");
        const u32 lvl = h.lvl + 1, first = new_wires(h.n_out, lvl);
        for (u32 i = 0; i < h.n_out; ++i) calldata_[h.off + 3 + i] = first + i;
        if (witness_) for (u32 i = 0; i < h.n_out; ++i) val_[first + i] = out_values[i];
        for (u32 i = 0; i < h.n_out; ++i) producer_[first + i] = (u32)kind_.size();
        push_instr(K_HINT, h.off, lvl);
        return first;
    }
    u32 cid_small(u64 k) {   // coefficient id of a small integer (table indexes, widths, counts)
        if (k >= small_cid_.size()) { if (k >= (1u << 20)) return intern(fr_u64(k)); small_cid_.resize(std::max<size_t>(k + 1, small_cid_.size() * 2 + 64), NO_WIRE); }
        if (small_cid_[k] == NO_WIRE) small_cid_[k] = intern(fr_u64(k));
        return small_cid_[k];
    }
    // capacity for the matrices and the program (an estimate from the circuit's shape saves the doubling copies of multi-GB arrays)
    void reserve(u64 constraints, u64 terms, u64 wires, u64 calldata) {
        for (int m = 0; m < 3; ++m) row_ptr_[m].reserve(constraints + 1);
        cid_[0].reserve(terms * 2 / 5); wid_[0].reserve(terms * 2 / 5); cid_[1].reserve(terms / 2); wid_[1].reserve(terms / 2); cid_[2].reserve(terms / 10); wid_[2].reserve(terms / 10);
        kind_.reserve(constraints); arg_.reserve(constraints); level_.reserve(constraints);
        wire_level_.reserve(wires); producer_.reserve(wires); is_bool_.reserve(wires); calldata_.reserve(calldata);
    }
    LE is_zero(const LE& a) {   // gnark r1cs IsZero: x = InvZero(a); m = 1 - a x; a m = 0
        FrH k;
        if (is_const(a, &k)) return constant(k.is_zero() ? 1 : 0);
        std::vector<FrH> ov;
        if (witness_) ov.push_back(FrH::inv(eval(a)));
        const LE x = hint("InvZeroHint", {a}, 1, &ov)[0];
        const u32 m = new_wire();
        if (witness_) val_[m] = FrH::sub(FrH::one(), FrH::mul(eval(a), eval(x)));
        emit_r1c(neg(a), x, sub(wire(m), constant(1)), m);
        assert_r1c(a, wire(m), LE(), "is_zero");
        is_bool_[m] = 1;
        ++cnt_[C_is_zero];
        return wire(m);
    }
    std::vector<LE> to_binary(const LE& a, int n) {   // std/math/bits ToBinary: NBits hint, booleanity per bit, recomposition
        std::vector<FrH> ov;
        if (witness_) { const U256 v = U256::of(eval(a)); for (int i = 0; i < n; ++i) ov.push_back(v.bit(i) ? FrH::one() : FrH::zero()); }
        std::vector<LE> bits = hint("nBits", {a}, (u32)n, &ov);
        LE sum;
        for (int i = 0; i < n; ++i) {
            u32 w = 0; bits[i].single_wire(&w);
            assert_r1c(bits[i], sub(constant(1), bits[i]), LE(), "bit");
            is_bool_[w] = 1;
            add_assign(sum, scale_pow2(bits[i], i));
        }
        assert_eq(sum, a, "recompose");
        cnt_[C_to_binary_bits] += (u64)n;
        return bits;
    }
    LE inverse(const LE& a) {          // res * a = 1, solved by division
        const u32 x = new_wire();
        if (witness_) val_[x] = FrH::inv(eval(a));
        emit_r1c(wire(x), a, constant(1), x);
        ++cnt_[C_inverse];
        return wire(x);
    }
    LE div_unchecked(const LE& num, const LE& den) {   // res * den = num; 0 / 0 = 0 (gnark DivUnchecked)
        const u32 x = new_wire();
        if (witness_) val_[x] = FrH::mul(eval(num), FrH::inv(eval(den)));
        emit_r1c(wire(x), den, num, x);
        ++cnt_[C_div_unchecked];
        return wire(x);
    }
    // bnb-fork comparison helpers (API additions used at circuit/batch_create_user_circuit.go:167,224,268, circuit/utils.go:85-90,115,174;
    // their source is not in the image — restated from their contract): both operands are known to be below 2^n
    void assert_le_nop(const LE& a, const LE& b, int n) { to_binary(sub(b, a), n); ++cnt_[C_assert_le]; }   // b - a fits n bits
    LE cmp_nop(const LE& a, const LE& b, int n) {       // -1, 0, 1
        const std::vector<LE> bits = to_binary(add(sub(a, b), LE_const_cid(cid_pow2(n))), n + 1);   // 2^n + a - b: bit n = (a >= b)
        const LE eq = is_zero(sub(a, b));
        ++cnt_[C_cmp];
        return sub(sub(scale(bits[n], 2), constant(1)), eq);                                  // 2 ge - 1 - eq
    }
    // an expression as ONE wire (lookup results etc. are wires already; anything else costs one constraint)
    LE to_wire(const LE& e) {
        u32 w;
        if (e.single_wire(&w)) return e;
        const u32 x = new_wire();
        if (witness_) val_[x] = eval(e);
        emit_r1c(e, constant(1), wire(x), x);
        return wire(x);
    }

    // ---- poseidon.Poseidon(api, inputs...) (bnb fork std/hash/poseidon, 3P): sponge over blocks of 12; every S-box spends three
    // product wires (x^2, x^4, x^5), round constants and the MDS layer are linear and fold into the next S-box's input expression
    // async: the call is a long serial chain whose result only feeds assertions (the CEX commitments): the executors may run it beside
    // everything else; assertions that read its wires are scheduled in the last level
    // async 1: the digest only feeds the last level (the CEX commitments); async 2: a long call other work can run BESIDE — the instructions that
    // do not depend on its digest keep their as-soon-as-possible levels, everything downstream of it moves behind them, and the call carries
    // the level it must be complete in front of (solver_file.hpp POSEIDON_JOIN_SHIFT): the RLC challenge's 116-permutation sponge
    LE poseidon(const std::vector<LE>& inputs, int async = 0, int out_idx = 1, int carry_idx = 0);

    // ---- logderivlookup (gnark std/lookup/logderivlookup, 3P)
    int new_table() { tables_.emplace_back(); return (int)tables_.size() - 1; }
    void table_insert(int t, const LE& v) {
        if (tables_[t].immutable) throw std::runtime_error("circuit: Insert after Lookup");
        tables_[t].entries.push_back(v);
    }
    std::vector<LE> table_lookup(int t, const std::vector<LE>& inds);
    // ---- rangecheck (gnark std/rangecheck, commit-based checker, 3P): collected here, decomposed in finalize()
    void range_check(const LE& v, int bits) { FrH k; if (is_const(v, &k)) return; rc_vals_.push_back(v); rc_bits_.push_back(bits); ++cnt_[C_range_check]; }

    // ---- the deferred part of a gnark compile (api.Compiler().Defer callbacks, in registration order): the range checker's
    // decomposition + its lookup argument, every table's argument, all behind ONE commitment (std/multicommit: one Commit call)
    // witness mode: the value the commitment wire takes (the BSB22 challenge is computed outside the circuit: a multi-exponentiation
    // over the committed wires + a hash to the field, host/bsb22_challenge.hpp; any value satisfies the circuit)
    void set_commitment_value(const FrH& v) { commitment_value_ = v; }
    // level assignment of finish(): true (default) = as late as possible, false = gnark's as-soon-as-possible levels
    void set_alap(bool on) { alap_ = on; }
    void set_beside(bool on) { beside_ = on; }      // async 2 calls (poseidon()): off = ordinary calls at their level
    bool beside() const { return beside_ && alap_ && poseidon_native_; }
    void finalize_commitments();
    Compiled finish();

    // assertions whose inputs come from a long hash chain are scheduled in the last level (they produce nothing, any later level is valid)
    void defer_assertions_reading(u32 wire_lo, u32 wire_hi) { slow_lo_.push_back(wire_lo); slow_hi_.push_back(wire_hi); }

private:
    enum { C_mul, C_assert, C_assert_bool, C_is_zero, C_to_binary_bits, C_inverse, C_div_unchecked, C_assert_le, C_cmp, C_range_check, C_poseidon_call, C_lookup_call, C_lookup_query, C_table_entry, C_rc_limb, C_NUM };
    static const char* cnt_name(int i) { static const char* n[] = {"mul", "assert", "assert_bool", "is_zero", "to_binary_bits", "inverse", "div_unchecked", "assert_le", "cmp", "range_check", "poseidon_call", "lookup_call", "lookup_query", "table_entry", "range_check_limb"}; return n[i]; }
    u64 cnt_[C_NUM] = {};
    u64 perm_count_[zkpor_host::kPosMaxT + 1] = {};
    u64 n_public_, n_secret_, n_wires_ = 0;
    bool witness_, poseidon_native_;
    std::vector<FrH> coeff_;
    std::unordered_map<FrH, u32, FrKeyHash> coeff_id_;
    std::vector<FrH> val_;
    std::vector<u32> wire_level_;
    std::vector<u32> producer_;          // per wire: the instruction that assigns it (NO_WIRE: an input)
    std::vector<uint8_t> is_bool_;
    std::vector<u64> row_ptr_[3];
    std::vector<u32> cid_[3], wid_[3];
    std::vector<u32> kind_, arg_, level_;
    std::vector<u32> calldata_;
    std::vector<std::string> hint_names_;
    std::vector<LookupTable> tables_;
    std::vector<LE> rc_vals_;
    std::vector<int> rc_bits_;
    std::vector<u32> slow_lo_, slow_hi_;
    std::vector<u32> deferred_asserts_;
    std::vector<u32> check_;                    // the R1C instructions without an unknown wire, in program order
    std::vector<std::pair<u32, u32>> async_instrs_;   // (instruction, its as-soon-as-possible level)
    std::vector<u32> beside_instrs_;                  // async 2 calls
    PermTemplate tmpl_[zkpor_host::kPosMaxT + 1];
    std::vector<u32> committed_;
    u32 commitment_wire_ = NO_WIRE;
    u32 max_level_ = 0;
    bool alap_ = true, beside_ = true;
    u32 pow2_cid_[256];
    std::vector<u32> small_cid_;
    int rc_width_ = 0;
    FrH commitment_value_ = FrH::from_u64(0x5eedc0de);
    static const u32 kSumChunk = 1024;

    u32 hint_name(const char* n) {
        for (size_t i = 0; i < hint_names_.size(); ++i) if (hint_names_[i] == n) return (u32)i;
        hint_names_.push_back(n);
        return (u32)hint_names_.size() - 1;
    }
    u32 new_wire() { return new_wires(1, 0); }
    u32 new_wires(u32 n, u32 lvl) {
        const u32 first = (u32)n_wires_;
        if (n_wires_ + n >= 0xfffffff0ull) throw std::runtime_error("circuit: too many wires");
        n_wires_ += n;
        wire_level_.resize(n_wires_, lvl);
        producer_.resize(n_wires_, NO_WIRE);
        is_bool_.resize(n_wires_, 0);
        if (witness_) val_.resize(n_wires_);
        return first;
    }
    u32 level_of(const LE& e) const { u32 l = 0; for (u32 i = 0; i < e.size(); ++i) l = std::max(l, wire_level_[e[i].wire]); return l; }
    void push_le(const LE& e) { calldata_.push_back(e.size()); for (u32 i = 0; i < e.size(); ++i) { calldata_.push_back(e[i].cid); calldata_.push_back(e[i].wire); } }
    void push_instr(u32 kind, u64 arg, u32 lvl) {
        if (arg >= 0xffffffffull) throw std::runtime_error("circuit: call data beyond 2^32 words");
        kind_.push_back(kind); arg_.push_back((u32)arg); level_.push_back(lvl);
        max_level_ = std::max(max_level_, lvl);
    }
    void push_row(int m, const LE& e) {
        const u32 n = e.size();
        const Term* t = e.data();
        const size_t o = cid_[m].size();
        cid_[m].resize(o + n); wid_[m].resize(o + n);
        u32* c = cid_[m].data() + o; u32* w = wid_[m].data() + o;
        for (u32 i = 0; i < n; ++i) { c[i] = t[i].cid; w[i] = t[i].wire; }
        row_ptr_[m].push_back(o + n);
    }
    void push_row1(int m, u32 wire_id) { cid_[m].push_back(CID_ONE); wid_[m].push_back(wire_id); row_ptr_[m].push_back(cid_[m].size()); }
    // a row straight from a permutation template: symbol 0 = ONE, symbol q > t = the x^5 wire of S-box q - t - 1 of the permutation at wbase
    void push_row_sym(int m, const PermTemplate::Sym* b, const PermTemplate::Sym* e, u32 wbase, u32 t) {
        const size_t n = (size_t)(e - b), o = cid_[m].size();
        cid_[m].resize(o + n); wid_[m].resize(o + n);
        u32* c = cid_[m].data() + o; u32* w = wid_[m].data() + o;
        for (size_t i = 0; i < n; ++i) { c[i] = b[i].cid; w[i] = b[i].sym == 0 ? 0u : wbase + 3 * (b[i].sym - 1 - t) + 2; }
        row_ptr_[m].push_back(o + n);
    }
    bool reads_slow(const LE& e) const {
        for (u32 i = 0; i < e.size(); ++i) for (size_t k = 0; k < slow_lo_.size(); ++k) if (e[i].wire >= slow_lo_[k] && e[i].wire < slow_hi_[k]) return true;
        return false;
    }
    // one constraint + its instruction; out = the wire it solves for (NO_WIRE: an assertion)
    void emit_r1c(const LE& l, const LE& r, const LE& o, u32 out) {
        u32 lvl = 0;
        const LE* sides[3] = {&l, &r, &o};
        for (auto* e : sides) for (u32 i = 0; i < e->size(); ++i) if ((*e)[i].wire != out) lvl = std::max(lvl, wire_level_[(*e)[i].wire]);
        ++lvl;
        if (out != NO_WIRE) { wire_level_[out] = lvl; producer_[out] = (u32)kind_.size(); }
        push_row(0, l); push_row(1, r); push_row(2, o);
        const u64 row = row_ptr_[0].size() - 2;
        push_instr(K_R1C, row, lvl);
        if (out == NO_WIRE) check_.push_back((u32)kind_.size() - 1);          // an assertion: flagged CHECK in the container
        if (out == NO_WIRE && !slow_lo_.empty() && (reads_slow(l) || reads_slow(r) || reads_slow(o))) deferred_asserts_.push_back((u32)kind_.size() - 1);
    }
    LE merge(const LE& a, const LE& b, bool negate_b) {
        LE r;
        u32 i = 0, j = 0;
        while (i < a.size() || j < b.size()) {
            if (j >= b.size() || (i < a.size() && a[i].wire < b[j].wire)) { r.push(a[i++]); continue; }
            if (i >= a.size() || b[j].wire < a[i].wire) { r.push({b[j].wire, negate_b ? cid_neg(b[j].cid) : b[j].cid}); ++j; continue; }
            const u32 c = cid_add(a[i].cid, negate_b ? cid_neg(b[j].cid) : b[j].cid);
            if (c != CID_ZERO) r.push({a[i].wire, c});
            ++i; ++j;
        }
        return r;
    }
    const PermTemplate& perm_template(int t);
};

// ---------------------------------------------------------------------------------------------------------------- Poseidon gadget
inline const PermTemplate& Builder::perm_template(int t) {
    PermTemplate& T = tmpl_[t];
    if (T.t) return T;
    const zkpor_host::PosParams& P = zkpor_host::PosParamsOf(t);
    const int rounds = zkpor_host::kPosRF + P.rp;
    T.t = t; T.n_sbox = zkpor_host::PosSboxes(t);
    // a lane = dense coefficient vector over the symbols seen so far: [ONE, entry 0..t-1, sbox outputs...]
    const size_t nsym = 1 + (size_t)t + (size_t)T.n_sbox;
    std::vector<std::vector<FrH>> lane((size_t)t, std::vector<FrH>(nsym, FrH::zero()));
    for (int i = 0; i < t; ++i) lane[i][1 + i] = FrH::one();
    size_t used = 1 + (size_t)t;     // symbols that can be non-zero so far
    int s = 0;
    T.in_ptr.push_back(0);
    for (int r = 0; r < rounds; ++r) {
        const bool full = r < zkpor_host::kPosRF / 2 || r >= zkpor_host::kPosRF / 2 + P.rp;
        for (int i = 0; i < t; ++i) lane[i][0] = FrH::add(lane[i][0], P.rc[(size_t)r * t + i]);
        if (r == 0) for (int i = 0; i < t; ++i) T.rc0_cid.push_back(intern(P.rc[i]));
        const int nsb = full ? t : 1;
        for (int i = 0; i < nsb; ++i) {
            if (r > 0) for (size_t k = 0; k < used; ++k) if (!lane[i][k].is_zero()) T.in_terms.push_back({(u32)k, intern(lane[i][k])});
            T.in_ptr.push_back((u32)T.in_terms.size());   // round 0: empty (the entry expression + rc0 is substituted at the call)
            std::fill(lane[i].begin(), lane[i].end(), FrH::zero());
            lane[i][1 + t + s] = FrH::one();
            ++s;
        }
        used = 1 + (size_t)t + (size_t)s;
        // mix
        std::vector<std::vector<FrH>> nl((size_t)t, std::vector<FrH>(nsym, FrH::zero()));
        for (int i = 0; i < t; ++i)
            for (int j = 0; j < t; ++j) {
                const FrH& m = P.mds[(size_t)i * t + j];
                for (size_t k = 0; k < used; ++k) if (!lane[j][k].is_zero()) nl[i][k] = FrH::add(nl[i][k], FrH::mul(m, lane[j][k]));
            }
        lane.swap(nl);
    }
    T.out_ptr.push_back(0);
    for (int i = 0; i < t; ++i) {
        for (size_t k = 0; k < used; ++k) if (!lane[i][k].is_zero()) T.out_terms.push_back({(u32)k, intern(lane[i][k])});
        T.out_ptr.push_back((u32)T.out_terms.size());
    }
    return T;
}

inline LE Builder::poseidon(const std::vector<LE>& inputs, int async, int out_idx, int carry_idx) {
    if (inputs.empty()) throw std::runtime_error("circuit: Poseidon of nothing");
    const size_t n = inputs.size();
    const size_t total_sbox = zkpor_host::PosSpongeSboxes(n);
    u32 lvl = 0;
    for (auto& e : inputs) lvl = std::max(lvl, level_of(e));
    ++lvl;
    const u32 base = new_wires((u32)(3 * total_sbox), lvl);
    if (witness_) {
        std::vector<FrH> in(n), tr(3 * total_sbox);
        for (size_t i = 0; i < n; ++i) in[i] = eval(inputs[i]);
        zkpor_host::PosSponge(in.data(), n, tr.data(), out_idx, carry_idx);
        for (size_t i = 0; i < tr.size(); ++i) val_[base + i] = tr[i];
    }
    if (poseidon_native_) {
        // ONE instruction for the whole gadget call (what a gadget-aware export makes of the 3 * sboxes constraint instructions):
        // call data = nIn, first output wire, number of output wires, out_idx | carry_idx << 8, the input expressions
        const u64 off = calldata_.size();
        calldata_.push_back((u32)n); calldata_.push_back(base); calldata_.push_back((u32)(3 * total_sbox));
        calldata_.push_back((u32)out_idx | ((u32)carry_idx << 8) | ((async ? 1u : 0u) << 16));      // async 2: the join level is added in finish()
        calldata_.push_back((u32)n_constraints());      // its constraints follow as consecutive rows, three per S-box
        for (auto& e : inputs) push_le(e);
        for (size_t i = 0; i < 3 * total_sbox; ++i) producer_[base + i] = (u32)kind_.size();
        if (async == 1) {
            // finish() puts an async-1 call back at its as-soon-as-possible level while its producers may have moved late, and a prefetch
            // (zkpor_solver_prefetch_dev) evaluates its inputs from the bare assignment: both are only sound when the inputs ARE the assignment
            for (auto& e : inputs)
                for (u32 t = 0; t < e.size(); ++t)
                    if (e[t].wire >= n_public_ + n_secret_) throw std::runtime_error("circuit: an async Poseidon call may read input wires only (it read internal wire " + std::to_string(e[t].wire) + ")");
            async_instrs_.push_back({(u32)kind_.size(), lvl});
        }
        if (async == 2) beside_instrs_.push_back((u32)kind_.size());
        push_instr(K_POSEIDON, off, lvl);
    }
    // the constraints, permutation after permutation
    LE cap, out;   // carry expression (empty = 0)
    size_t done = 0;
    u32 wbase = base;
    while (done < n) {
        const int k = (int)std::min<size_t>(12, n - done), t = k + 1;
        const PermTemplate& T = perm_template(t);
        auto sym_wire = [&](u32 sym) -> u32 { return sym == 0 ? 0u : wbase + 3 * (sym - 1 - (u32)t) + 2; };   // sym > t only (entries never appear after round 0)
        auto stamp = [&](const PermTemplate::Sym* b, const PermTemplate::Sym* e) { LE r; for (auto* p = b; p < e; ++p) r.push({sym_wire(p->sym), p->cid}); return r; };
        for (int s = 0; s < T.n_sbox; ++s) {
            const u32 x2 = wbase + 3 * (u32)s, x4 = x2 + 1, x5 = x2 + 2;
            if (poseidon_native_ && s >= t) {   // rows only (the native instruction produces the wires), straight from the template
                const PermTemplate::Sym* tb = &T.in_terms[T.in_ptr[s]], *te = &T.in_terms[T.in_ptr[s + 1]];
                push_row_sym(0, tb, te, wbase, (u32)t); push_row_sym(1, tb, te, wbase, (u32)t); push_row1(2, x2);
                push_row1(0, x2); push_row1(1, x2); push_row1(2, x4);
                push_row1(0, x4); push_row_sym(1, tb, te, wbase, (u32)t); push_row1(2, x5);
                continue;
            }
            LE in;
            if (s < t) { in = (s == 0) ? cap : inputs[done + (size_t)s - 1]; in = add(in, LE_const_cid(T.rc0_cid[s])); }
            else in = stamp(&T.in_terms[T.in_ptr[s]], &T.in_terms[T.in_ptr[s + 1]]);   // template order: ONE first, then ascending wires
            if (poseidon_native_) {
                push_row(0, in); push_row(1, in); push_row1(2, x2);
                push_row1(0, x2); push_row1(1, x2); push_row1(2, x4);
                push_row1(0, x4); push_row(1, in); push_row1(2, x5);
            } else {                  // gnark's own form: three constraint instructions per S-box, levels along the chain
                emit_r1c(in, in, wire(x2), x2);
                emit_r1c(wire(x2), wire(x2), wire(x4), x4);
                emit_r1c(wire(x4), in, wire(x5), x5);
            }
        }
        cap = stamp(&T.out_terms[T.out_ptr[carry_idx]], &T.out_terms[T.out_ptr[carry_idx + 1]]);
        out = stamp(&T.out_terms[T.out_ptr[out_idx]], &T.out_terms[T.out_ptr[out_idx + 1]]);
        wbase += 3 * (u32)T.n_sbox;
        done += (size_t)k;
        ++perm_count_[t];
    }
    ++cnt_[C_poseidon_call];
    if (async == 1 && poseidon_native_) defer_assertions_reading(base, base + (u32)(3 * total_sbox));
    return out;
}


// ---------------------------------------------------------------------------------------------------------------- lookups
// Lookup(inds...) = ONE instruction with len(inds) consecutive output wires (gnark BlueprintLookupHint, constraint/blueprint_logderivlookup.go,
// 3P: the entries are stored once per table, an instruction names how many of them it may see and carries its index expressions).
// Call data of kind 3: blockOff, nbEntries, nQ, firstOut, then the nQ index expressions.  Entry block at blockOff: nEntries,
// entryOff[nEntries] (word offsets from blockOff), then the entry expressions.
inline std::vector<LE> Builder::table_lookup(int ti, const std::vector<LE>& inds) {
    LookupTable& T = tables_[ti];
    if (inds.empty()) return {};
    if (T.entries.empty()) throw std::runtime_error("circuit: Lookup in an empty table");
    if (!T.immutable) {
        T.immutable = true;
        T.block_off = calldata_.size();
        T.block_entries = T.entries.size();
        const size_t ne = T.entries.size();
        calldata_.push_back((u32)ne);
        const size_t off0 = calldata_.size();
        calldata_.resize(off0 + ne);
        for (size_t i = 0; i < ne; ++i) { calldata_[off0 + i] = (u32)(calldata_.size() - T.block_off); push_le(T.entries[i]); }
        cnt_[C_table_entry] += ne;
        for (auto& e : T.entries) T.entries_level = std::max(T.entries_level, level_of(e));
    }
    u32 lvl = T.entries_level;                                       // the blueprint reads every entry
    for (auto& e : inds) lvl = std::max(lvl, level_of(e));
    ++lvl;
    const u32 nq = (u32)inds.size();
    const u32 first = new_wires(nq, lvl);
    if (witness_) {
        for (u32 i = 0; i < nq; ++i) {
            const U256 ix = U256::of(eval(inds[i]));
            if ((ix.w[1] | ix.w[2] | ix.w[3]) || ix.w[0] >= T.entries.size()) throw std::runtime_error("circuit: lookup query too large");
            val_[first + i] = eval(T.entries[ix.w[0]]);
        }
    }
    const u64 off = calldata_.size();
    calldata_.push_back((u32)T.block_off); calldata_.push_back((u32)T.block_entries); calldata_.push_back(nq); calldata_.push_back(first);
    if (T.block_off >= 0xffffffffull) throw std::runtime_error("circuit: call data beyond 2^32 words");
    for (auto& e : inds) push_le(e);
    for (u32 i = 0; i < nq; ++i) producer_[first + i] = (u32)kind_.size();
    push_instr(K_LOOKUP, off, lvl);
    std::vector<LE> out;
    for (u32 i = 0; i < nq; ++i) { out.push_back(wire(first + i)); T.q_ind.push_back(inds[i]); T.q_val.push_back(wire(first + i)); }
    ++cnt_[C_lookup_call];
    cnt_[C_lookup_query] += nq;
    return out;
}

// ---------------------------------------------------------------------------------------------------------------- deferred commitments
// gnark runs the callbacks registered with api.Compiler().Defer after Define, in registration order: the range checker's commit
// (std/rangecheck/rangecheck_commit.go), every table's commit (std/lookup/logderivlookup), each ending in logderivarg.Build
// (std/internal/logderivarg): a count hint for the multiplicities, then — behind std/multicommit's ONE api.Commit over everything
// collected — sum_i m_i / (c - t_i) = sum_j 1 / (c - q_j).  All 3P, restated from the v0.10 sources as recalled.
inline void Builder::finalize_commitments() {
    // one argument = one table (rows of nb_col expressions) + its queries, both flat, row-major
    struct Arg { u32 nb_col = 1; std::vector<LE> table, queries; u32 first_exp = 0, nb_table = 0; };
    std::vector<Arg> args;
    std::vector<u32> commit_wires;
    auto commit_le = [&](const LE& e) { for (u32 i = 0; i < e.size(); ++i) if (e[i].wire >= n_public_) commit_wires.push_back(e[i].wire); };
    auto build_a = [&](Arg&& a, bool const_table) {
        const u32 nb_col = a.nb_col, nb_table = (u32)(a.table.size() / nb_col);
        const size_t nb_q = a.queries.size() / nb_col;
        // countHint(nbTable, nbCols, table rows..., query rows...) -> nbTable multiplicities
        std::vector<FrH> ov;
        if (witness_) {
            std::vector<u64> cnt(nb_table, 0);
            for (size_t q = 0; q < nb_q; ++q) {
                const U256 ix = U256::of(eval(a.queries[q * nb_col]));
                if ((ix.w[1] | ix.w[2] | ix.w[3]) || ix.w[0] >= nb_table) throw std::runtime_error("circuit: query element not in table");
                for (u32 c = 0; c < nb_col; ++c) if (!(eval(a.queries[q * nb_col + c]) == eval(a.table[ix.w[0] * nb_col + c]))) throw std::runtime_error("circuit: query row not in table");
                ++cnt[ix.w[0]];
            }
            for (u32 i = 0; i < nb_table; ++i) ov.push_back(FrH::from_u64(cnt[i]));
        }
        HintOpen h = hint_open("countHint", (u32)(2 + a.table.size() + a.queries.size()), nb_table);
        hint_input(h, LE_const_cid(cid_small(nb_table))); hint_input(h, LE_const_cid(cid_small(nb_col)));
        for (auto& e : a.table) hint_input(h, e);
        for (auto& e : a.queries) hint_input(h, e);
        a.first_exp = hint_close(h, witness_ ? ov.data() : nullptr);
        a.nb_table = nb_table;
        if (!const_table) for (auto& e : a.table) commit_le(e);
        for (auto& e : a.queries) commit_le(e);
        for (u32 i = 0; i < nb_table; ++i) commit_wires.push_back(a.first_exp + i);
        args.push_back(std::move(a));
    };
    // 1. the range checker: optimal limb width (2..17 bits, the count of rangecheck_commit.go nbR1CSConstraints), decomposition, table 0..2^w-1
    if (!rc_vals_.empty()) {
        int best = 0; u64 best_cost = ~0ull;
        for (int w = 2; w < 18; ++w) {
            u64 limbs = 0;
            for (int b : rc_bits_) limbs += (u64)((b + w - 1) / w);
            const u64 cost = ((u64)1 << w) + limbs + rc_vals_.size() + 1;
            if (cost < best_cost) { best_cost = cost; best = w; }
        }
        rc_width_ = best;
        Arg a;
        {
            u64 nq = 0;
            for (int b : rc_bits_) nq += (u64)((b + best - 1) / best) + (b % best ? 1 : 0);
            a.queries.reserve(nq);
        }
        const LE c_width = LE_const_cid(cid_small((u64)best));
        std::vector<FrH> ov;
        for (size_t i = 0; i < rc_vals_.size(); ++i) {
            const int bits = rc_bits_[i], nl = (bits + best - 1) / best;
            if (witness_) {
                ov.clear();
                const U256 v = U256::of(eval(rc_vals_[i]));
                if (v.bitlen() > nl * best) throw std::runtime_error("circuit: range check fails on the given inputs");
                for (int l = 0; l < nl; ++l) ov.push_back(FrH::from_u64(v.bits(l * best, best)));
            }
            HintOpen h = hint_open("DecomposeHint", 3, (u32)nl);
            hint_input(h, LE_const_cid(cid_small((u64)bits))); hint_input(h, c_width); hint_input(h, rc_vals_[i]);
            const u32 first = hint_close(h, witness_ ? ov.data() : nullptr);
            LE composed;
            for (int l = 0; l < nl; ++l) composed.push({first + (u32)l, cid_pow2(l * best)});
            assert_eq(composed, rc_vals_[i], "limbs");
            for (int l = 0; l < nl; ++l) a.queries.push_back(wire(first + (u32)l));
            if (bits % best) a.queries.push_back(scale_pow2(wire(first + (u32)nl - 1), nl * best - bits));   // the top limb shifted up must fit the width too
            cnt_[C_rc_limb] += (u64)nl;
            rc_vals_[i].clear();
        }
        rc_vals_.clear(); rc_vals_.shrink_to_fit();
        a.table.reserve((size_t)1 << best);
        for (u64 i = 0; i < ((u64)1 << best); ++i) a.table.push_back(LE_const_cid(cid_small(i)));
        build_a(std::move(a), true);
    }
    // 2. every lookup table, in creation order: rows (i, entry_i), queries (index, result)
    for (auto& T : tables_) {
        if (T.entries.empty()) continue;
        Arg a;
        a.nb_col = 2;
        a.table.reserve(2 * T.entries.size()); a.queries.reserve(2 * T.q_ind.size());
        for (size_t i = 0; i < T.entries.size(); ++i) { a.table.push_back(LE_const_cid(cid_small((u64)i))); a.table.push_back(std::move(T.entries[i])); }
        for (size_t i = 0; i < T.q_ind.size(); ++i) { a.queries.push_back(std::move(T.q_ind[i])); a.queries.push_back(std::move(T.q_val[i])); }
        T.entries.clear(); T.q_ind.clear(); T.q_val.clear();
        T.entries.shrink_to_fit(); T.q_ind.shrink_to_fit(); T.q_val.shrink_to_fit();
        build_a(std::move(a), false);
    }
    if (args.empty()) return;
    // 3. api.Commit (frontend/cs/r1cs Commit, 3P): the committed set = the WIRES of the expressions, sorted, unique, without ONE and
    // without public wires; the placeholder hint gets them as inputs and returns the commitment wire
    {
        std::sort(commit_wires.begin(), commit_wires.end());
        commit_wires.erase(std::unique(commit_wires.begin(), commit_wires.end()), commit_wires.end());
        committed_ = std::move(commit_wires);
        // inputs: the commitment's index, then the committed wires (no public / commitment wire is committed in this circuit) — the layout
        // host/prove_on_device.hpp and go/zkporgpu/solver.go serve
        HintOpen h = hint_open("bsb22CommitmentComputePlaceholder", (u32)committed_.size() + 1, 1);
        hint_input(h, LE());
        for (u32 w : committed_) hint_input(h, wire(w));
        commitment_wire_ = hint_close(h, &commitment_value_);
    }
    const LE cmt = wire(commitment_wire_);
    // 4. the arguments (every callback of std/multicommit receives the same commitment)
    for (auto& a : args) {
        const u32 nb_col = a.nb_col;
        std::vector<LE> coef{constant(1)};
        for (u32 i = 1; i < nb_col; ++i) coef.push_back(mul(coef[i - 1], cmt));
        const LE challenge = nb_col == 1 ? cmt : mul(coef[nb_col - 1], cmt);
        auto combine = [&](const LE* row) { LE r; for (u32 i = 0; i < nb_col; ++i) r = add(r, mul(coef[i], row[i])); return r; };
        // gnark adds term by term into ONE expression; a sum of 10^7 terms is one constraint row nobody can evaluate in parallel, so the
        // sums are built as trees here: every kSumChunk terms become one wire, the chunk wires are summed the same way (depth log_1024 n)
        struct SumTree {
            Builder& b; std::vector<LE> lv;
            void add(const LE& x, size_t depth = 0) {
                if (lv.size() <= depth) lv.resize(depth + 1);
                b.add_assign(lv[depth], x);
                if (lv[depth].size() >= kSumChunk) { const LE wv = b.to_wire(lv[depth]); lv[depth].clear(); add(wv, depth + 1); }
            }
            LE total() { LE r; for (auto& e : lv) r = b.add(r, e); return r; }
        };
        SumTree lp{*this, {}}, rp{*this, {}};
        for (u32 i = 0; i < a.nb_table; ++i) lp.add(div_unchecked(wire(a.first_exp + i), sub(challenge, combine(&a.table[(size_t)i * nb_col]))));
        const size_t nb_q = a.queries.size() / nb_col;
        for (size_t i = 0; i < nb_q; ++i) rp.add(inverse(sub(challenge, combine(&a.queries[i * nb_col]))));
        assert_eq(lp.total(), rp.total(), "log-derivative sums");
        a = Arg();
    }
}

inline Compiled Builder::finish() {
    Compiled c;
    c.n_wires = n_wires_; c.n_public = n_public_; c.n_secret = n_secret_; c.n_constraints = n_constraints();
    // ALAP: the levels recorded so far are gnark's (as soon as possible).  Nothing obliges an executor to them: any order that respects the
    // dependencies is a valid run, and AS LATE AS POSSIBLE is the better one for a machine that pays per level — the 2 500 sequential
    // products of the challenge powers (batch_create_user_circuit.go:286-289) each feed one product per user (:304-308), which ASAP spreads
    // over the 2 500 levels of the chain (1 + U instructions each: 2 500 launches), ALAP collects in one level behind it (the chain itself
    // becomes a run of one-instruction levels: one launch).  One reverse sweep: an instruction sits one level below its earliest consumer.
    if (alap_) {
        const std::vector<u32> asap(level_);          // as recorded: gnark's levels
        std::vector<u32> alap(kind_.size(), max_level_);
        auto need = [&](u32 wire_id, u32 me, u32 my_level) {
            const u32 p = producer_[wire_id];
            if (p != NO_WIRE && p != me && alap[p] >= my_level) alap[p] = my_level - 1;
        };
        auto need_le = [&](const u32* cd, u64& p, u32 me, u32 my_level) { const u32 nt = cd[p++]; for (u32 k = 0; k < nt; ++k) { need(cd[p + 1], me, my_level); p += 2; } };
        // a lookup reads EVERY entry of its table (the blueprint's Solve): the entries' producers sit below the table's earliest lookup — applied once
        // per table, at its first lookup in program order (the last the reverse sweep meets), not once per lookup (10^5 lookups x 2 x 10^4 entries)
        std::unordered_map<u32, std::pair<u32, u32>> table_first_min;   // block -> (first lookup instruction, lowest level among its lookups)
        for (size_t i = 0; i < kind_.size(); ++i)
            if (kind_[i] == K_LOOKUP) { const u32 blk = calldata_[arg_[i]]; if (!table_first_min.count(blk)) table_first_min[blk] = {(u32)i, max_level_ + 1}; }
        for (size_t ii = kind_.size(); ii-- > 0;) {
            const u32 i = (u32)ii, lv = alap[i];
            const u32* cd = calldata_.data() + arg_[i];
            if (kind_[i] == K_R1C) {
                for (int m = 0; m < 3; ++m) for (u64 t = row_ptr_[m][arg_[i]]; t < row_ptr_[m][arg_[i] + 1]; ++t) need(wid_[m][t], i, lv);
            } else if (kind_[i] == K_HINT) {
                u64 p = 3 + (u64)cd[2];
                for (u32 k = 0; k < cd[1]; ++k) need_le(cd, p, i, lv);
            } else if (kind_[i] == K_LOOKUP) {
                auto& fm = table_first_min[cd[0]];
                fm.second = std::min(fm.second, lv);
                if (fm.first == i) {
                    const u32* tb = calldata_.data() + cd[0];
                    for (u32 e = 0; e < tb[0]; ++e) { u64 p = tb[1 + e]; need_le(tb, p, NO_WIRE, fm.second); }
                }
                u64 p = 4;
                for (u32 q = 0; q < cd[2]; ++q) need_le(cd, p, i, lv);
            } else if (kind_[i] == K_POSEIDON) {
                u64 p = zkpor_host::POSEIDON_HDR;
                for (u32 k = 0; k < cd[0]; ++k) need_le(cd, p, i, lv);
            }
        }
        for (size_t i = 0; i < kind_.size(); ++i) level_[i] = alap[i];
        for (auto& a : async_instrs_) level_[a.first] = a.second;     // a long chain that runs beside everything else starts as EARLY as it can
        if (!beside_instrs_.empty() && poseidon_native_) {
            // async 2.  X = everything downstream of those calls (one forward pass: program order is a topological order).  The rest keeps
            // its EARLIEST level, X its LATEST, moved back far enough that the first consumer of a call stands behind all of the rest that
            // comes after the call: an executor starts the call on a side stream at its level and joins it in front of that consumer — the
            // independent levels in between (Merkle paths, range-check hints, the big count hint) run beside it.  Valid: an X instruction's
            // producers outside X sit at or before their latest level, which is below its own; nothing outside X reads X.
            std::vector<uint8_t> in_x(kind_.size(), 0);
            for (u32 b : beside_instrs_) in_x[b] = 1;
            auto from_x = [&](u32 wire_id) { const u32 p = producer_[wire_id]; return p != NO_WIRE && in_x[p]; };
            auto le_from_x = [&](const u32* cd, u64& p) { const u32 nt = cd[p++]; bool r = false; for (u32 k = 0; k < nt; ++k) { r = r || from_x(cd[p + 1]); p += 2; } return r; };
            std::unordered_map<u32, bool> table_x;                     // block -> some entry comes from X (decided at the table's first lookup: the entries are final by then)
            for (size_t ii = 0; ii < kind_.size(); ++ii) {
                const u32 i = (u32)ii;
                if (in_x[i]) continue;
                const u32* cd = calldata_.data() + arg_[i];
                bool x = false;
                if (kind_[i] == K_R1C) {
                    for (int m = 0; m < 3 && !x; ++m) for (u64 t = row_ptr_[m][arg_[i]]; t < row_ptr_[m][arg_[i] + 1] && !x; ++t) { const u32 wd = wid_[m][t]; x = producer_[wd] != i && from_x(wd); }
                } else if (kind_[i] == K_HINT) {
                    u64 p = 3 + (u64)cd[2];
                    for (u32 k = 0; k < cd[1] && !x; ++k) x = le_from_x(cd, p);
                } else if (kind_[i] == K_LOOKUP) {
                    auto it = table_x.find(cd[0]);
                    if (it == table_x.end()) {
                        const u32* tb = calldata_.data() + cd[0];
                        bool tx = false;
                        for (u32 e = 0; e < tb[0] && !tx; ++e) { u64 p = tb[1 + e]; tx = le_from_x(tb, p); }
                        it = table_x.emplace(cd[0], tx).first;
                    }
                    x = it->second;
                    u64 p = 4;
                    for (u32 q = 0; q < cd[2] && !x; ++q) x = le_from_x(cd, p);
                } else if (kind_[i] == K_POSEIDON) {
                    u64 p = zkpor_host::POSEIDON_HDR;
                    for (u32 k = 0; k < cd[0] && !x; ++k) x = le_from_x(cd, p);
                }
                in_x[i] = x ? 1 : 0;
            }
            std::vector<uint8_t> is_call(kind_.size(), 0), is_async1(kind_.size(), 0);
            for (u32 b : beside_instrs_) is_call[b] = 1;
            for (auto& a : async_instrs_) is_async1[a.first] = 1;
            u32 call_level = 0xffffffffu, rest_max = 0, x_min = 0xffffffffu;
            for (u32 b : beside_instrs_) call_level = std::min(call_level, asap[b]);
            for (size_t i = 0; i < kind_.size(); ++i) {
                if (is_call[i] || is_async1[i]) continue;
                if (in_x[i]) x_min = std::min(x_min, alap[i]);
                else if (asap[i] >= call_level) rest_max = std::max(rest_max, asap[i]);
            }
            const u32 shift = (x_min != 0xffffffffu && rest_max >= x_min) ? rest_max - x_min + 1 : 0;
            u32 join = 0xffffffffu, top = 0;
            for (size_t i = 0; i < kind_.size(); ++i) {
                if (is_async1[i]) { top = std::max(top, level_[i]); continue; }
                if (is_call[i]) level_[i] = asap[i];
                else if (in_x[i]) { level_[i] = alap[i] + shift; join = std::min(join, level_[i]); }
                else level_[i] = asap[i];
                top = std::max(top, level_[i]);
            }
            max_level_ = top;
            // the level a call must be complete in front of, as the executor counts levels (from 0), + 1 (0 = none); calls whose level does not
            // fit the field stay ordinary calls at their level
            for (u32 b : beside_instrs_) {
                u32& flags = calldata_[arg_[b] + 3];
                if (join != 0xffffffffu && join > level_[b] && join < (1u << 15)) flags |= (1u << 16) | (join << zkpor_host::POSEIDON_JOIN_SHIFT);   // level numbers start at 1 here: join = (join - 1) + 1
            }
            c.census["levels_beside_the_long_call"] = (join != 0xffffffffu && call_level != 0xffffffffu && join > call_level) ? join - call_level - 1 : 0;
        }
        producer_.clear(); producer_.shrink_to_fit();
    }
    // levels: counting sort of the instructions by level; deferred assertions go behind everything else
    const u32 last = max_level_ + 1;
    for (u32 i : deferred_asserts_) level_[i] = last;
    const u32 nl = (deferred_asserts_.empty() ? max_level_ : last);
    std::vector<u64> count(nl + 2, 0);
    for (u32 l : level_) ++count[l];
    // level 0 holds nothing (inputs); levels 1..nl
    c.level_ptr.assign(nl + 1, 0);
    for (u32 l = 1; l <= nl; ++l) c.level_ptr[l] = c.level_ptr[l - 1] + count[l];
    c.level_instr.resize(kind_.size());
    std::vector<u64> pos(c.level_ptr.begin(), c.level_ptr.end());
    for (size_t i = 0; i < kind_.size(); ++i) c.level_instr[pos[level_[i] - 1]++] = (u32)i;
    c.in_l.assign(n_wires_, 0); c.in_r.assign(n_wires_, 0);
    for (u32 w : wid_[0]) c.in_l[w] = 1;
    for (u32 w : wid_[1]) c.in_r[w] = 1;
    c.coeff = std::move(coeff_);
    for (int m = 0; m < 3; ++m) { c.row_ptr[m] = std::move(row_ptr_[m]); c.cid[m] = std::move(cid_[m]); c.wid[m] = std::move(wid_[m]); }
    c.kind = std::move(kind_); c.arg = std::move(arg_); c.check = std::move(check_); c.calldata = std::move(calldata_); c.hint_names = std::move(hint_names_);
    c.committed = std::move(committed_); c.commitment_wire = commitment_wire_;
    c.values = std::move(val_);
    for (int i = 0; i < C_NUM; ++i) c.census[cnt_name(i)] = cnt_[i];
    for (int t = 2; t <= zkpor_host::kPosMaxT; ++t) if (perm_count_[t]) c.census["poseidon_perm_t" + std::to_string(t)] = perm_count_[t];
    c.census["levels"] = c.level_ptr.size() - 1;
    c.census["rangecheck_limb_bits"] = (u64)rc_width_;
    return c;
}

// the solver container (host/solver_file.hpp, version 2: instruction kinds 3 and 4) of a compiled circuit
inline std::vector<uint8_t> SolverContainer(const Compiled& c) {
    std::vector<uint8_t> out;
    auto put = [&](const void* p, size_t n) { const uint8_t* b = (const uint8_t*)p; out.insert(out.end(), b, b + n); };
    auto pad = [&]() { while (out.size() % 8) out.push_back(0); };
    put("ZKPSOLV\x02", 8);
    const u64 h[4] = {c.kind.size(), c.level_ptr.size() - 1, c.hint_names.size(), c.calldata.size()};
    put(h, sizeof h);
    for (auto& s : c.hint_names) { const u32 l = (u32)s.size(); put(&l, 4); put(s.data(), l); }
    pad();
    {   // the kind words, assertions flagged CHECK
        std::vector<u32> kw(c.kind);
        for (u32 i : c.check) kw[i] |= zkpor_host::INSTR_CHECK;
        put(kw.data(), kw.size() * 4);
    }
    put(c.arg.data(), c.arg.size() * 4);
    pad();
    put(c.level_ptr.data(), c.level_ptr.size() * 8);
    put(c.level_instr.data(), c.level_instr.size() * 4);
    pad();
    put(c.calldata.data(), c.calldata.size() * 4);
    return out;
}

}  // namespace zkpor_circuit
