// The constraint system resident in HBM (csrc/r1cs.hip owns it; csrc/solver.hip reads it): gnark's compiled R1CS — three CSR matrices over one
// shared coefficient table (a term is (coefficient id, wire id)).
#pragma once
#include "common.cuh"

struct zkpor_r1cs {
    zkpor_ctx* ctx = nullptr;
    size_t n_constraints = 0, n_wires = 0, n_coeff = 0;
    zk::Fr* coeff = nullptr;       // Montgomery
    uint8_t* coeff_kind = nullptr;  // 0 generic, 1 = one, 2 = minus one, 3 = zero
    uint64_t* row_ptr[3] = {nullptr, nullptr, nullptr};
    uint32_t* cid[3] = {nullptr, nullptr, nullptr};
    uint32_t* wid[3] = {nullptr, nullptr, nullptr};
    size_t nnz[3] = {0, 0, 0};
    std::vector<zk::Fr> h_coeff;    // host copy of the table (the solver reads constant hint inputs — table sizes — when a program is loaded)
};

