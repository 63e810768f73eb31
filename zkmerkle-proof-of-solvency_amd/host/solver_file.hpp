// The solver-program container written by go/export_solver (source) and read by the host executor (host/solver_exec.hpp) and by the device
// executor (csrc/solver.hip): the flattened form of gnark's compiled solver program — Levels [][]int, instruction kinds, hint names, hint
// inputs as linear expressions, output wires (gnark constraint/bn254/solver.go and constraint/core.go, 3P; reached through groth16.Prove at
// src/prover/prover/prover.go:269).  The matrices come from the r1cs container (host/r1cs_file.hpp).
//
// Container "ZKPSOLV\x01" (little-endian):
//   u64 nInstructions, nLevels, nHintNames, nCallData
//   hint names: per name u32 length + bytes, then padding to 8
//   u32 kind[nInstructions]            0 = solve constraint `arg`; 1 = hint with call data at `arg`; 2 = same as 0/1 but skipped (pre-filled);
//                                      version 2 ("ZKPSOLV\x02") adds the two gadget instructions a gadget-aware export emits:
//                                      3 = table lookup (gnark BlueprintLookupHint), 4 = a whole poseidon.Poseidon(...) call
//   u32 arg[nInstructions]
//   u64 levelPtr[nLevels + 1]; u32 levelInstr[levelPtr[nLevels]]; pad to 8
//   u32 callData[nCallData]: per hint  nameId, nIn, nOut, out wire ids[nOut], then per input: nTerms, (coeffId, wireId)[nTerms]
//     kind 3 (lookup):   blockOff, nbEntries, nQ, firstOutWire, then nQ index expressions; outputs = wires firstOutWire .. +nQ-1 =
//                        entry[index].  Entry block at blockOff (shared by all lookups of one table, gnark stores it once per blueprint):
//                        nEntries, entryOff[nEntries] (word offsets from blockOff), the entry expressions; nbEntries <= nEntries
//     kind 4 (poseidon): nIn, firstOutWire, nOutWires, flags (bits 0-7 digest lane, 8-15 carry lane, bit 16 = ASYNC: the outputs are read
//                        by the last level only), firstRow (the call's constraints are rows firstRow .. firstRow + nOutWires - 1, three per
//                        S-box: in * in = x^2, x^2 * x^2 = x^4, x^4 * in = x^5 — an executor that has the S-box input in hand writes a, b, c of
//                        these rows itself; 0xffffffff = not consecutive / unknown), then nIn input expressions; outputs = the three product wires (x^2, x^4, x^5) of
//                        every S-box of the sponge over the inputs, permutation after permutation in round order
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace zkpor_host {

struct SolverView {
    uint64_t n_instructions = 0, n_levels = 0, n_calldata = 0;
    int version = 1;
    std::vector<std::string> hint_names;
    const uint32_t* kind = nullptr;
    const uint32_t* arg = nullptr;
    const uint64_t* level_ptr = nullptr;
    const uint32_t* level_instr = nullptr;
    const uint32_t* calldata = nullptr;
};
enum { INSTR_R1C = 0, INSTR_HINT = 1, INSTR_SKIP = 2, INSTR_LOOKUP = 3, INSTR_POSEIDON = 4 };
// bit 8 of a version-2 R1C instruction's kind word: CHECK — the constraint has no unknown wire by the time its level runs (an assertion: gnark's
// solver only verifies it).  An executor whose caller evaluates a, b, c of EVERY row afterwards and checks a x b = c there may leave these
// instructions out (csrc/solver.hip with zkpor_solver_set_abc_dev); every other executor treats the word as its low byte says.
enum : uint32_t { INSTR_CHECK = 1u << 8 };
// flags word of a kind-4 instruction: digest lane | carry lane << 8 | ASYNC << 16 | (join level + 1) << 17.  ASYNC alone: the digest only feeds the
// LAST level.  ASYNC with a join level J (counted from 0 in the container's level order): the call may run beside levels [its own, J) — nothing in
// them reads its wires — and must be complete in front of level J.
enum : uint32_t { POSEIDON_ASYNC = 1u << 16, POSEIDON_JOIN_SHIFT = 17, POSEIDON_HDR = 5 };   // header words of a kind-4 instruction's call data

// shape checks of the two gadget instructions against the call data (both executors call this before they trust an offset).
// A version-1 stream's kind 3 is the old "skipped" alias: callers map it to INSTR_SKIP.
inline bool CheckLookupShape(const SolverView& v, uint64_t arg, uint64_t n_wires, uint64_t n_coeff) {
    if (arg + 4 > v.n_calldata) return false;
    const uint32_t* cd = v.calldata + arg;
    const uint64_t block = cd[0], nb = cd[1], nq = cd[2], first = cd[3];
    if (block + 1 > v.n_calldata) return false;
    const uint32_t* tb = v.calldata + block;
    const uint64_t ne = tb[0];
    if (nb == 0 || nb > ne || block + 1 + ne > v.n_calldata || first + nq > n_wires) return false;
    for (uint64_t i = 0; i < nb; ++i) {
        const uint64_t o = block + tb[1 + i];
        if (o >= v.n_calldata) return false;
        const uint64_t nt = v.calldata[o];
        if (o + 1 + 2 * nt > v.n_calldata) return false;
        for (uint64_t t = 0; t < nt; ++t) if (v.calldata[o + 1 + 2 * t] >= n_coeff || v.calldata[o + 2 + 2 * t] >= n_wires) return false;
    }
    uint64_t p = arg + 4;
    for (uint64_t i = 0; i < nq; ++i) {
        if (p >= v.n_calldata) return false;
        const uint64_t nt = v.calldata[p++];
        if (p + 2 * nt > v.n_calldata) return false;
        for (uint64_t t = 0; t < nt; ++t) if (v.calldata[p + 2 * t] >= n_coeff || v.calldata[p + 1 + 2 * t] >= n_wires) return false;
        p += 2 * nt;
    }
    return true;
}
inline uint64_t PoseidonSboxCount(uint64_t n_in) {   // S-boxes of the sponge over n_in inputs (blocks of 12 + one ragged block)
    static const int rp[] = {56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65};
    const uint64_t full = n_in / 12, rem = n_in % 12;
    return full * (8 * 13 + 65) + (rem ? 8 * (rem + 1) + rp[rem + 1 - 2] : 0);
}
inline bool CheckPoseidonShape(const SolverView& v, uint64_t arg, uint64_t n_wires, uint64_t n_coeff) {
    if (arg + POSEIDON_HDR > v.n_calldata) return false;
    const uint32_t* cd = v.calldata + arg;
    const uint64_t n_in = cd[0], first = cd[1], n_out = cd[2];
    const uint32_t flags = cd[3];
    if (n_in == 0 || n_out != 3 * PoseidonSboxCount(n_in) || first + n_out > n_wires) return false;
    const uint32_t last_t = (uint32_t)(n_in % 12 ? n_in % 12 + 1 : 13);
    if ((flags & 0xff) >= last_t || (n_in > 12 && ((flags >> 8) & 0xff) >= 13u)) return false;   // digest lane of the last block, carry lane of the full ones
    uint64_t p = arg + POSEIDON_HDR;
    for (uint64_t i = 0; i < n_in; ++i) {
        if (p >= v.n_calldata) return false;
        const uint64_t nt = v.calldata[p++];
        if (p + 2 * nt > v.n_calldata) return false;
        for (uint64_t t = 0; t < nt; ++t) if (v.calldata[p + 2 * t] >= n_coeff || v.calldata[p + 1 + 2 * t] >= n_wires) return false;
        p += 2 * nt;
    }
    return true;
}

// the kind an executor acts on: a version-1 stream used 3 as a second spelling of "skipped"
inline uint32_t InstrKind(const SolverView& v, uint64_t i) { const uint32_t k = v.version == 1 ? v.kind[i] : (v.kind[i] & 0xffu); return (v.version == 1 && k == 3) ? (uint32_t)INSTR_SKIP : k; }
inline bool InstrIsCheck(const SolverView& v, uint64_t i) { return v.version >= 2 && (v.kind[i] & INSTR_CHECK) != 0; }

inline int ParseSolverFile(const uint8_t* data, size_t len, SolverView* out, std::string* err) {
    auto fail = [&](const char* m) { if (err) *err = std::string("solver file: ") + m; return 1; };
    size_t off = 0;
    auto need = [&](uint64_t n) { return n <= len && off <= len - n; };
    if (!data || !need(8) || memcmp(data, "ZKPSOLV", 7) != 0 || (data[7] != 1 && data[7] != 2)) return fail("bad magic");
    const uint32_t max_kind = data[7] == 1 ? 3 : 4;
    off = 8;
    uint64_t h[4];
    if (!need(sizeof h)) return fail("truncated header");
    memcpy(h, data + off, sizeof h); off += sizeof h;
    SolverView v;
    v.n_instructions = h[0]; v.n_levels = h[1]; v.n_calldata = h[3];
    const uint64_t n_names = h[2];
    if (v.n_instructions >= (1ull << 32) || v.n_levels > v.n_instructions + 1 || n_names > 4096 || v.n_calldata >= (1ull << 40)) return fail("bad counts");
    for (uint64_t i = 0; i < n_names; ++i) {
        uint32_t l;
        if (!need(4)) return fail("truncated names");
        memcpy(&l, data + off, 4); off += 4;
        if (l > 256 || !need(l)) return fail("bad name");
        v.hint_names.emplace_back((const char*)data + off, l); off += l;
    }
    off += (8 - off % 8) % 8;
    auto take32 = [&](uint64_t n, const uint32_t** p) { if (!need(n * 4)) return false; *p = (const uint32_t*)(data + off); off += n * 4; return true; };
    if (!take32(v.n_instructions, &v.kind) || !take32(v.n_instructions, &v.arg)) return fail("truncated instruction table");
    off += (8 - off % 8) % 8;
    if (!need((v.n_levels + 1) * 8)) return fail("truncated levels");
    v.level_ptr = (const uint64_t*)(data + off); off += (v.n_levels + 1) * 8;
    if (v.level_ptr[0] != 0) return fail("levels must start at 0");
    for (uint64_t l = 0; l < v.n_levels; ++l) if (v.level_ptr[l + 1] < v.level_ptr[l]) return fail("levels not monotone");
    const uint64_t n_li = v.level_ptr[v.n_levels];
    if (n_li > v.n_instructions) return fail("more level entries than instructions");
    if (!take32(n_li, &v.level_instr)) return fail("truncated level entries");
    off += (8 - off % 8) % 8;
    if (!take32(v.n_calldata, &v.calldata)) return fail("truncated call data");
    for (uint64_t i = 0; i < n_li; ++i) if (v.level_instr[i] >= v.n_instructions) return fail("level entry out of range");
    for (uint64_t i = 0; i < v.n_instructions; ++i) {
        const uint32_t k = v.kind[i];
        if (data[7] == 1 ? k > max_kind : ((k & 0xffu) > max_kind || (k & ~(0xffu | INSTR_CHECK)) != 0 || ((k & INSTR_CHECK) && (k & 0xffu) != INSTR_R1C))) return fail("unknown instruction kind");
    }
    v.version = data[7];
    *out = std::move(v);
    return 0;
}

}  // namespace zkpor_host
