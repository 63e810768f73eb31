// The solver-program container written by go/export_solver (source) and read by the host executor (host/solver_exec.hpp) and by the device
// executor (csrc/solver.hip): the flattened form of gnark's compiled solver program — Levels [][]int, instruction kinds, hint names, hint
// inputs as linear expressions, output wires (gnark constraint/bn254/solver.go and constraint/core.go, 3P; reached through groth16.Prove at
// src/prover/prover/prover.go:269).  The matrices come from the r1cs container (host/r1cs_file.hpp).
//
// Container "ZKPSOLV\x01" (little-endian):
//   u64 nInstructions, nLevels, nHintNames, nCallData
//   hint names: per name u32 length + bytes, then padding to 8
//   u32 kind[nInstructions]            0 = solve constraint `arg`; 1 = hint with call data at `arg`; 2 = same as 0/1 but skipped (pre-filled)
//   u32 arg[nInstructions]
//   u64 levelPtr[nLevels + 1]; u32 levelInstr[levelPtr[nLevels]]; pad to 8
//   u32 callData[nCallData]: per hint  nameId, nIn, nOut, out wire ids[nOut], then per input: nTerms, (coeffId, wireId)[nTerms]
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>

namespace zkpor_host {

struct SolverView {
    uint64_t n_instructions = 0, n_levels = 0, n_calldata = 0;
    std::vector<std::string> hint_names;
    const uint32_t* kind = nullptr;
    const uint32_t* arg = nullptr;
    const uint64_t* level_ptr = nullptr;
    const uint32_t* level_instr = nullptr;
    const uint32_t* calldata = nullptr;
};
enum { INSTR_R1C = 0, INSTR_HINT = 1, INSTR_SKIP = 2 };

inline int ParseSolverFile(const uint8_t* data, size_t len, SolverView* out, std::string* err) {
    auto fail = [&](const char* m) { if (err) *err = std::string("solver file: ") + m; return 1; };
    size_t off = 0;
    auto need = [&](uint64_t n) { return n <= len && off <= len - n; };
    if (!data || !need(8) || memcmp(data, "ZKPSOLV\x01", 8) != 0) return fail("bad magic");
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
    for (uint64_t i = 0; i < v.n_instructions; ++i) if (v.kind[i] > 3) return fail("unknown instruction kind");
    *out = std::move(v);
    return 0;
}

}  // namespace zkpor_host
