// The constraint system resident in HBM (csrc/r1cs.hip owns it; csrc/solver.hip reads it): gnark's compiled R1CS — three CSR matrices over one
// shared coefficient table (a term is (coefficient id, wire id)).
#pragma once
#include "common.cuh"

// a row of more terms than this is left to one WAVE (64 lanes stride over its terms) instead of one thread: zkpor50_1380 has 21 000 such rows
// (the sums behind the range checker's and the lookups' log-derivative arguments, the CEX commitment's 3 454-term inputs), 4.7 % of the
// terms — and they were 18 of k_r1cs_eval's 45 ms, each a serial walk of one thread (profiles/r04_r1cs_row_classes.txt)
#define R1CS_LONG_ROW 256
struct zkpor_r1cs {
    zkpor_ctx* ctx = nullptr;
    size_t n_constraints = 0, n_wires = 0, n_coeff = 0;
    zk::Fr* coeff = nullptr;       // Montgomery
    uint8_t* coeff_kind = nullptr;  // 0 generic, 1 = one, 2 = minus one, 3 = zero
    uint64_t* row_ptr[3] = {nullptr, nullptr, nullptr};
    uint32_t* cid[3] = {nullptr, nullptr, nullptr};
    uint32_t* wid[3] = {nullptr, nullptr, nullptr};
    size_t nnz[3] = {0, 0, 0};
    uint32_t* long_rows[3] = {nullptr, nullptr, nullptr};   // rows of more than R1CS_LONG_ROW terms: summed by a wave each (k_r1cs_eval_long)
    size_t n_long[3] = {0, 0, 0};
    // round 6: the other rows in the order k_r1cs_eval walks them — by term count, then by the pattern of their coefficient kinds (rows of one gadget
    // side by side), natural order inside a class.  One thread per row in NATURAL order made every wave last as long as its longest row and run
    // every branch its 64 rows took: 9.3 G wave-instructions per zkpor50_1380 proof for 0.5 G of lane work (profiles/r06_pmc_valu.json)
    uint32_t* perm[3] = {nullptr, nullptr, nullptr};
    size_t n_perm[3] = {0, 0, 0};
    std::vector<zk::Fr> h_coeff;    // host copy of the table (the solver reads constant hint inputs — table sizes — when a program is loaded)
};

