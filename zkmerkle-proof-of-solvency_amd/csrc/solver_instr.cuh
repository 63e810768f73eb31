// One instruction of a compiled circuit's solver program — the semantics shared by the device executor (csrc/solver.hip: one GPU thread per
// instruction of a level) and its CPU unit tests (tests/hostlib/solver_logic.cpp compiles this header with g++ and walks the levels serially,
// so the logic is checked in the CPU suite and only the launch plumbing is left to the -m gpu tests).
//
// What it replaces: gnark's solver inside groth16.Prove (src/prover/prover/prover.go:269; gnark constraint/bn254/solver.go solveR1C +
// the hint calls, 3P): per instruction either ONE constraint L.w * R.w = O.w is solved for its single unknown wire, or a hint computes its
// output wires from linear expressions of solved wires.  Hints with native semantics here: circuit.IntegerDivision (circuit/utils.go:103-110,
// registered at prover.go:68) and the std hints BatchCreateUserCircuit reaches (NBits behind api.ToBinary, InvZero behind api.IsZero,
// DecomposeHint behind the range checks).  Same error codes as host/solver_exec.hpp.
#pragma once
#include "fe.cuh"

namespace zk {

enum : uint8_t { HK_NONE = 0, HK_INTDIV = 1, HK_NBITS = 2, HK_INVZERO = 3, HK_DECOMPOSE = 4, HK_COUNT = 5 };
// 3, 4: the gadget instructions of a version-2 container (host/solver_file.hpp): a table lookup (gnark BlueprintLookupHint), a whole
// poseidon.Poseidon(...) call.  HK_COUNT (gnark logderivarg countHint: millions of inputs) and kind 4 have their own kernels
// (csrc/solver.hip, csrc/poseidon.hip); solve_instr refuses them.
enum : u32 { SI_R1C = 0, SI_HINT = 1, SI_SKIP = 2, SI_LOOKUP = 3, SI_POSEIDON = 4 };
enum : u32 { SI_POSEIDON_HDR = 5 };   // nIn, firstOut, nOut, flags, firstRow
enum : int {
    SE_OK = 0, SE_ROW_RANGE = 10, SE_TWO_UNKNOWN = 11, SE_NOT_SATISFIED = 12, SE_ZERO_COEFF = 13, SE_DIV_ZERO = 14, SE_CALLDATA = 20,
    SE_NO_HINT = 21, SE_ID_RANGE = 22, SE_INPUT_UNSOLVED = 23, SE_HINT_FAILED = 24, SE_LOOKUP_RANGE = 25, SE_COUNT_TABLE = 26, SE_COUNT_QUERY = 27,
    SE_DEFERRED = -1     // not an error: the instruction's wire is num / den, left to the caller (which inverts several denominators at once)
};
// a quotient an instruction leaves open (solve_instr with `pend`): w[wire] = num / den, den != 0
struct SiPending { u32 wire; Fr num, den; };

// the program and its constraint system, as pointers the executing side can read (device memory for the kernels, host memory for the CPU tests)
struct SolverProg {
    const Fr* coeff; const uint8_t* ckind;                       // coefficient table, class per coefficient: 0 general, 1 = one, 2 = minus one, 3 = zero
    const u64* row_ptr[3]; const u32* cid[3]; const u32* wid[3];  // L, R, O in CSR form (csrc/r1cs.hip)
    u32 n_constraints, n_wires, n_coeff;
    const u32* kind; const u32* arg;                              // per instruction
    const u32* calldata; u64 n_calldata;                          // hint call data (layout: host/solver_file.hpp)
    const uint8_t* hint_kind; u32 n_hint_names;                   // HK_* per hint name id
    u32 ext_cap = 0;                                              // device executor: entries of its external-hint list (every external hint of the program fits)
};

ZK_HD void si_add_term(Fr& acc, uint8_t k, const Fr* coeff, u32 cid, const Fr& x) {
    if (k == 1) acc = Fr::add(acc, x);
    else if (k == 2) acc = Fr::sub(acc, x);
    else if (k == 0) acc = Fr::add(acc, Fr::mul(coeff[cid], x));
}

// 1 / a for a Montgomery residue (0 for 0) by the binary extended Euclidean algorithm: ~760 shift / add / subtract steps on 8 limbs, about a
// third of the instructions of the Fermat power Fr::inv (254 squarings + 127 products).  The executor's levels are latency-bound — a level lasts
// as long as its slowest instruction, and that is a division (IsZero, inverse wires) — so the single-thread cost of an inversion is what
// a deep program's run time is made of.  Invariants: u x1^-1 = v x2^-1 = a (mod r) up to the tracked factors, u and v odd after the halvings.
ZK_HD Fr fr_inverse(const Fr& a) {
    if (a.is_zero()) return a;
    u32 u[8], v[8], x1[8], x2[8];
    for (int i = 0; i < 8; ++i) { u[i] = a.v[i]; v[i] = FrParams::mod(i); x1[i] = 0; x2[i] = 0; }
    x1[0] = 1;
    auto is_one = [](const u32* t) { u32 o = t[0] ^ 1u; for (int i = 1; i < 8; ++i) o |= t[i]; return o == 0; };
    auto halve = [](u32* t, u32* x) {            // t even: t /= 2, x = x / 2 mod r
        for (int i = 0; i < 7; ++i) t[i] = (t[i] >> 1) | (t[i + 1] << 31);
        t[7] >>= 1;
        u32 top = 0;
        if (x[0] & 1u) {                          // x + r < 2^255: no carry out of the 8 limbs
            u64 c = 0;
            for (int i = 0; i < 8; ++i) { c += (u64)x[i] + FrParams::mod(i); x[i] = (u32)c; c >>= 32; }
            top = (u32)c;
        }
        for (int i = 0; i < 7; ++i) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
        x[7] = (x[7] >> 1) | (top << 31);
    };
    auto sub_into = [](u32* t, const u32* s) {   // t -= s, t >= s
        u32 bw = 0;
        for (int i = 0; i < 8; ++i) { const u64 d = (u64)t[i] - s[i] - bw; t[i] = (u32)d; bw = (u32)(d >> 32) & 1u; }
    };
    auto sub_mod = [](u32* x, const u32* y) {    // x = x - y mod r, both below r
        u32 bw = 0;
        for (int i = 0; i < 8; ++i) { const u64 d = (u64)x[i] - y[i] - bw; x[i] = (u32)d; bw = (u32)(d >> 32) & 1u; }
        if (bw) { u64 c = 0; for (int i = 0; i < 8; ++i) { c += (u64)x[i] + FrParams::mod(i); x[i] = (u32)c; c >>= 32; } }
    };
    while (!is_one(u) && !is_one(v)) {
        while (!(u[0] & 1u)) halve(u, x1);
        while (!(v[0] & 1u)) halve(v, x2);
        bool ge = true;
        for (int i = 7; i >= 0; --i) if (u[i] != v[i]) { ge = u[i] > v[i]; break; }
        if (ge) { sub_into(u, v); sub_mod(x1, x2); } else { sub_into(v, u); sub_mod(x2, x1); }
    }
    Fr t;                                         // = (a R)^-1 as an integer = a^-1 R^-1
    const u32* res = is_one(u) ? x1 : x2;
    for (int i = 0; i < 8; ++i) t.v[i] = res[i];
    return Fr::mul(Fr::mul(t, Fr::r2()), Fr::r2());   // two Montgomery products by R^2: a^-1 R^-1 -> a^-1 -> a^-1 R
}

// canonical 256-bit integers on 8 x 32-bit limbs (what gnark hands a hint as *big.Int)
struct U256L {
    u32 l[8];
    ZK_HD static U256L of(const Fr& a) { Fr c = Fr::from_mont(a); U256L r; for (int i = 0; i < 8; ++i) r.l[i] = c.v[i]; return r; }
    ZK_HD Fr fr() const { Fr c; for (int i = 0; i < 8; ++i) c.v[i] = l[i]; return Fr::to_mont(c); }
    ZK_HD bool is_zero() const { u32 o = 0; for (int i = 0; i < 8; ++i) o |= l[i]; return o == 0; }
    ZK_HD bool bit(int i) const { return i < 256 && ((l[i >> 5] >> (i & 31)) & 1u); }
    ZK_HD int bitlen() const {
        for (int i = 7; i >= 0; --i) if (l[i]) { int b = 32; while (!((l[i] >> (b - 1)) & 1u)) --b; return 32 * i + b; }
        return 0;
    }
    // bits [lo, lo + n), n <= 32
    ZK_HD u32 bits(int lo, int n) const {
        if (lo >= 256 || n <= 0) return 0;
        const int wi = lo >> 5, sh = lo & 31;
        u32 x = l[wi] >> sh;
        if (sh && wi + 1 < 8) x |= l[wi + 1] << (32 - sh);
        return n >= 32 ? x : (x & ((1u << n) - 1u));
    }
    // big.Int.DivMod for non-negative operands, d != 0: shift-subtract over the dividend's bits (a hint runs once per user and division, not per wire)
    ZK_HD static void divmod(const U256L& a, const U256L& d, U256L* q, U256L* rem) {
        U256L qq, r;
        for (int i = 0; i < 8; ++i) { qq.l[i] = 0; r.l[i] = 0; }
        for (int i = a.bitlen() - 1; i >= 0; --i) {
            u32 carry = a.bit(i) ? 1u : 0u;
            for (int k = 0; k < 8; ++k) { const u32 nc = r.l[k] >> 31; r.l[k] = (r.l[k] << 1) | carry; carry = nc; }   // r < d <= 2^254: no bit is lost
            bool ge = true;
            for (int k = 7; k >= 0; --k) if (r.l[k] != d.l[k]) { ge = r.l[k] > d.l[k]; break; }
            if (ge) {
                u32 bw = 0;
                for (int k = 0; k < 8; ++k) { const u64 t = (u64)r.l[k] - d.l[k] - bw; r.l[k] = (u32)t; bw = (u32)(t >> 32) & 1u; }
                qq.l[i >> 5] |= 1u << (i & 31);
            }
        }
        *q = qq; *rem = r;
    }
};

// a linear expression of the call data at word p (nTerms, (coeffId, wireId)...): its value over the solved wires; p moves behind it.
// Ids were validated when the program was loaded.
ZK_HD int si_eval_le(const SolverProg& P, const u32* cd, u64& p, const Fr* w, const uint8_t* known, Fr* out) {
    const u32 nterms = cd[p++];
    Fr acc = Fr::zero();
    for (u32 k = 0; k < nterms; ++k) {
        const u32 ci = cd[p++], wi = cd[p++];
        if (!known[wi]) return SE_INPUT_UNSOLVED;
        si_add_term(acc, P.ckind[ci], P.coeff, ci, w[wi]);
    }
    *out = acc;
    return SE_OK;
}
// a field element as an index below `bound` (gnark: Uint64() of the canonical value)
ZK_HD bool si_index(const Fr& x, u32 bound, u32* out) {
    const Fr c = Fr::from_mont(x);
    u32 hi = 0;
    for (int i = 1; i < 8; ++i) hi |= c.v[i];
    if (hi || c.v[0] >= bound) return false;
    *out = c.v[0];
    return true;
}

// the end of solving ONE constraint (gnark solveR1C): v[m] = the known part of L, R, O, `which` = the expression the unknown wire x sits in
// (-1: none, the row is an assertion), uc = the sum of its coefficients there
ZK_HD int si_r1c_finish(const Fr* v, int which, u32 x, const Fr& uc, Fr* w, uint8_t* known, SiPending* pend, Fr* got = nullptr) {
    if (which < 0) return Fr::mul(v[0], v[1]) == v[2] ? SE_OK : SE_NOT_SATISFIED;   // an assertion
    if (uc.is_zero()) return SE_ZERO_COEFF;
    Fr val;
    if (which == 2) {
        val = Fr::sub(Fr::mul(v[0], v[1]), v[2]);                     // O_known + c x = L R
        if (uc == Fr::one()) {}
        else if (Fr::neg(uc) == Fr::one()) val = Fr::neg(val);
        else if (pend) { pend->wire = x; pend->num = val; pend->den = uc; return SE_DEFERRED; }
        else val = Fr::mul(val, fr_inverse(uc));
    } else {
        const Fr& other = v[1 - which];
        if (other.is_zero()) {                                        // gnark solveR1C: nothing to divide by — the constraint must hold as it is,
            if (!v[2].is_zero()) return SE_DIV_ZERO;                  // (L_known + c x) * 0 = O needs O = 0 (else: "division by zero") ...
            w[x] = Fr::zero();                                        // ... and the wire stays 0: api.DivUnchecked(0, 0) = 0 (std logderivarg relies on it)
            known[x] = 1;
            if (got) *got = Fr::zero();
            return SE_OK;
        }
        const Fr num = Fr::sub(v[2], Fr::mul(v[which], other));       // (L_known + c x) R = O  =>  x = (O - L_known R) / (c R)
        Fr den = other;
        if (uc == Fr::one()) {}
        else if (Fr::neg(uc) == Fr::one()) den = Fr::neg(den);
        else den = Fr::mul(den, uc);
        if (pend) { pend->wire = x; pend->num = num; pend->den = den; return SE_DEFERRED; }
        val = Fr::mul(num, fr_inverse(den));
    }
    w[x] = val;
    known[x] = 1;
    if (got) *got = val;                                              // the caller hands the value on without reading it back
    return SE_OK;
}

// Executes instruction `ins`: assigns its output wire(s) in w and marks them known.  The instructions of one level are independent: no
// instruction reads a wire another instruction of the same level assigns, so a level may run in any order or all at once.
ZK_HD int solve_instr(const SolverProg& P, u32 ins, Fr* w, uint8_t* known, SiPending* pend = nullptr) {
    const u32 kind = P.kind[ins], arg = P.arg[ins];
    if (kind == SI_SKIP) return SE_OK;
    if (kind == SI_LOOKUP) {   // outputs = entry[index]: call data blockOff, nbEntries, nQ, firstOut, the index expressions (shapes validated at load)
        const u32* cd = P.calldata + arg;
        const u32* tb = P.calldata + cd[0];
        const u32 nb = cd[1], nq = cd[2], first = cd[3];
        u64 p = 4;
        for (u32 q = 0; q < nq; ++q) {
            Fr ix, v;
            int rc = si_eval_le(P, cd, p, w, known, &ix);
            if (rc) return rc;
            u32 i;
            if (!si_index(ix, nb, &i)) return SE_LOOKUP_RANGE;            // gnark: "lookup query too large"
            u64 pe = tb[1 + i];
            rc = si_eval_le(P, tb, pe, w, known, &v);
            if (rc) return rc;
            w[first + q] = v;
            known[first + q] = 1;
        }
        return SE_OK;
    }
    if (kind != SI_R1C && kind != SI_HINT) return SE_NO_HINT;             // kind 4 runs in its own kernel
    if (kind == SI_R1C) {
        if (arg >= P.n_constraints) return SE_ROW_RANGE;
        Fr v[3], uc = Fr::zero();
        int which = -1;
        u32 x = 0;
        for (int m = 0; m < 3; ++m) {
            Fr acc = Fr::zero();
            for (u64 t = P.row_ptr[m][arg]; t < P.row_ptr[m][arg + 1]; ++t) {
                const u32 wi = P.wid[m][t], ci = P.cid[m][t];
                if (known[wi]) si_add_term(acc, P.ckind[ci], P.coeff, ci, w[wi]);
                else {
                    if (which >= 0 && (which != m || x != wi)) return SE_TWO_UNKNOWN;   // not solvable at this level: the export's levels are wrong
                    uc = which < 0 ? P.coeff[ci] : Fr::add(uc, P.coeff[ci]);
                    which = m; x = wi;
                }
            }
            v[m] = acc;
        }
        return si_r1c_finish(v, which, x, uc, w, known, pend);
    }
    // hint: nameId, nIn, nOut, out wire ids[nOut], then per input: nTerms, (coeffId, wireId)[nTerms]
    if ((u64)arg + 3 > P.n_calldata) return SE_CALLDATA;
    const u32* cd = P.calldata + arg;
    const u32 name = cd[0], n_in = cd[1], n_out = cd[2];
    if (name >= P.n_hint_names || P.hint_kind[name] == HK_NONE) return SE_NO_HINT;
    if ((u64)arg + 3 + n_out > P.n_calldata) return SE_CALLDATA;
    for (u32 i = 0; i < n_out; ++i) if (cd[3 + i] >= P.n_wires) return SE_ID_RANGE;
    Fr in[3];
    if (n_in > 3) return SE_HINT_FAILED;                                   // no native hint takes more
    u64 p = 3 + (u64)n_out;
    for (u32 i = 0; i < n_in; ++i) {
        if (arg + p >= P.n_calldata) return SE_CALLDATA;
        const u32 nterms = cd[p++];
        if (arg + p + 2ull * nterms > P.n_calldata) return SE_CALLDATA;
        Fr acc = Fr::zero();
        for (u32 k = 0; k < nterms; ++k) {
            const u32 ci = cd[p++], wi = cd[p++];
            if (wi >= P.n_wires || ci >= P.n_coeff) return SE_ID_RANGE;
            if (!known[wi]) return SE_INPUT_UNSOLVED;
            si_add_term(acc, P.ckind[ci], P.coeff, ci, w[wi]);
        }
        in[i] = acc;
    }
    const u32* outw = cd + 3;
    switch (P.hint_kind[name]) {
    case HK_INTDIV: {                                                      // out[0], out[1] = DivMod(in[0], in[1])
        if (n_in != 2 || n_out != 2) return SE_HINT_FAILED;
        const U256L a = U256L::of(in[0]), d = U256L::of(in[1]);
        if (d.is_zero()) return SE_HINT_FAILED;                            // big.Int.DivMod panics on a zero divisor
        U256L q, rem;
        U256L::divmod(a, d, &q, &rem);
        w[outw[0]] = q.fr(); w[outw[1]] = rem.fr();
        break;
    }
    case HK_NBITS: {                                                       // out[i] = bit i of in[0]
        if (n_in != 1) return SE_HINT_FAILED;
        const U256L a = U256L::of(in[0]);
        for (u32 i = 0; i < n_out; ++i) w[outw[i]] = a.bit((int)i) ? Fr::one() : Fr::zero();
        break;
    }
    case HK_INVZERO: {                                                     // 1 / in[0], or 0
        if (n_in != 1 || n_out != 1) return SE_HINT_FAILED;
        if (pend && !in[0].is_zero()) { pend->wire = outw[0]; pend->num = Fr::one(); pend->den = in[0]; return SE_DEFERRED; }
        w[outw[0]] = fr_inverse(in[0]);
        break;
    }
    case HK_DECOMPOSE: {                                                   // in = (varSize, limbSize, value) -> limbs of limbSize bits, little-endian
        if (n_in != 3) return SE_HINT_FAILED;
        const U256L vs = U256L::of(in[0]), ls = U256L::of(in[1]), val = U256L::of(in[2]);
        u32 hi = 0;
        for (int i = 1; i < 8; ++i) hi |= ls.l[i] | vs.l[i];
        if (hi || ls.l[0] == 0 || ls.l[0] > 64 || vs.l[0] > 256) return SE_HINT_FAILED;
        const int limb = (int)ls.l[0];
        if ((u64)n_out * (u64)limb < (u64)val.bitlen()) return SE_HINT_FAILED;   // the value does not fit the requested limbs: the range check must fail
        for (u32 i = 0; i < n_out; ++i) {
            const int lo = (int)i * limb;
            U256L x;
            for (int k = 0; k < 8; ++k) x.l[k] = 0;
            x.l[0] = val.bits(lo, limb < 32 ? limb : 32);
            if (limb > 32) x.l[1] = val.bits(lo + 32, limb - 32);
            w[outw[i]] = x.fr();
        }
        break;
    }
    default: return SE_NO_HINT;
    }
    for (u32 i = 0; i < n_out; ++i) known[outw[i]] = 1;
    return SE_OK;
}

// host side (both executors' setup): the native hint a name stands for; gnark registers hints under their Go function names and the
// exporter keeps the last path element
inline uint8_t hint_kind_of_name(const char* n) {
    auto eq = [&](const char* x) { const char* a = n; while (*a && *a == *x) { ++a; ++x; } return *a == 0 && *x == 0; };
    if (eq("IntegerDivision")) return HK_INTDIV;
    if (eq("NBits") || eq("nBits")) return HK_NBITS;
    if (eq("InvZero") || eq("InvZeroHint")) return HK_INVZERO;
    if (eq("DecomposeHint")) return HK_DECOMPOSE;
    if (eq("countHint")) return HK_COUNT;
    return HK_NONE;
}

}  // namespace zk
