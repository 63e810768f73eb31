// Prover.GenerateAndVerifyProof + the row it produces (src/prover/prover/prover.go:161-236, :250-283), end to end over the C ABI:
//
//   BatchWitness.WitnessData --DecodeBatchWitness--> utils.BatchCreateUserWitness            (utils.go:704-742; witness_codec.hpp)
//     --SetBatchCreateUserCircuitWitness--> the assigned circuit / NewWitness vector           (circuit/...:334-436; witness_assign.hpp)
//     --SOLVER (gnark's r1cs.Solve + hints: NOT part of this repo, a callback)--> w, a, b, c    (prover.go:269, inside groth16.Prove)
//          the BSB22 hint inside it calls back for the Pedersen commitment --zkpor_commit--> (commitment, knowledge proof)
//          and gets the hint's output, the challenge hash_to_field(commitment) (bsb22_challenge.hpp), to carry on solving
//     --zkpor_prove_tail--> Ar, Bs, Krs --zkpor_proof_write_raw--> proof.WriteRawTo bytes      (prover.go:201)
//     --MakeProofRow--> the `proof` table row (prover.go:227-236; proof_row.hpp)
//
// groth16.Verify (prover.go:276) stays gnark's; a caller that has a verifier passes it as `verify`.
// This is the function a worker of host/prover_host.hpp's Dispatcher / Pipeline runs per batch (ProveFn).  C++ because the build
// image has no Go; names follow the reference.
#pragma once
#include <functional>
#include <string>
#include <vector>
#include "../../include/zkpor.h"
#include "bsb22_challenge.hpp"
#include "proof_row.hpp"
#include "witness_assign.hpp"
#include "witness_codec.hpp"

namespace zkpor_host {

struct SolvedWitness {                    // what r1cs.Solve leaves behind (gnark R1CSSolution: W, A, B, C), Montgomery limbs
    std::vector<uint64_t> w, a, b, c;     // 4 limbs per element; a, b, c have n_constraints elements
    size_t n_constraints = 0;
    bool has_commitment = false;
    uint8_t commitment[64] = {0}, pok[64] = {0};  // filled by the commit callback during the solve
};
// the Pedersen commitment the BSB22 hint asks for in the middle of the solve: values = the private committed wires (Montgomery Fr).
// challenge = the value the hint returns to the solver (32 B big-endian, canonical): hash_to_field(commitment) — this circuit commits
// to no public wire; a circuit that does calls Bsb22Challenge(commitment, public values) itself.
typedef std::function<int(const uint64_t* values, size_t n, uint8_t commitment[64], uint8_t pok[64], uint8_t challenge[32])> CommitFn;
// the solver: assigned inputs -> full solution; calls `commit` when the circuit has a commitment.  0 = ok
typedef std::function<int(const AssignedWitness& in, const CommitFn& commit, SolvedWitness* out)> SolveFn;
typedef std::function<int(const std::string& raw_proof, const BatchCreateUserWitnessW& w)> VerifyFn;  // groth16.Verify stand-in; may be empty

enum ProveBatchErr { PB_OK = 0, PB_DECODE = 1, PB_ASSIGN = 2, PB_SOLVE = 3, PB_PROVE = 4, PB_VERIFY = 5 };

// one batch: returns PB_OK and fills `row` (+ `tier`), or says which stage failed (`err` holds the reason)
inline int GenerateAndVerifyProof(zkpor_ctx* ctx, zkpor_pk* pk, const std::string& witness_data, int64_t batch_number,
                                  const std::vector<int>& asset_counts_tiers, const SolveFn& solve, const uint64_t r[4], const uint64_t s[4],
                                  const VerifyFn& verify, ProofRow* row, int* tier, std::string* err) {
    BatchCreateUserWitnessW w;
    try { w = DecodeBatchWitness(witness_data, true); }
    catch (const std::exception& e) { if (err) *err = std::string("decode: ") + e.what(); return PB_DECODE; }
    AssignedWitness in;
    std::string why;
    if (!SetBatchCreateUserCircuitWitness(w, asset_counts_tiers, &in, &why)) { if (err) *err = "assign: " + why; return PB_ASSIGN; }
    SolvedWitness sol;
    CommitFn commit = [&](const uint64_t* values, size_t n, uint8_t c[64], uint8_t k[64], uint8_t challenge[32]) -> int {
        int32_t rc = zkpor_commit(ctx, pk, values, n, c, k);
        if (rc == ZKPOR_OK) {
            sol.has_commitment = true; memcpy(sol.commitment, c, 64); memcpy(sol.pok, k, 64);
            if (challenge) {   // gnark hashes commitment.Marshal(): big-endian canonical bytes, not the limbs
                uint8_t be[64];
                zkpor_g1_marshal(c, be);
                memcpy(challenge, Bsb22Challenge(be).data(), 32);
            }
        }
        return rc;
    };
    if (solve(in, commit, &sol) != 0) { if (err) *err = "solve: the solver reported an error"; return PB_SOLVE; }
    if (sol.a.size() != 4 * sol.n_constraints || sol.b.size() != sol.a.size() || sol.c.size() != sol.a.size()) { if (err) *err = "solve: a, b, c sizes"; return PB_SOLVE; }
    // zkpor_prove_tail reads n_wires x 32 bytes of w and n_constraints x 32 of a, b, c: a solver that returns another circuit's
    // (e.g. the other tier's) vectors must end in PB_SOLVE, not in an out-of-bounds read
    uint64_t dims[6];
    if (zkpor_pk_dims(pk, dims) != ZKPOR_OK) { if (err) *err = "prove: the key is not loaded"; return PB_PROVE; }
    if (sol.w.size() != 4 * dims[0]) {
        if (err) *err = "solve: w has " + std::to_string(sol.w.size() / 4) + " wires, the key has " + std::to_string(dims[0]);
        return PB_SOLVE;
    }
    if (sol.n_constraints > ((uint64_t)1 << dims[4])) { if (err) *err = "solve: more constraints than the key's domain"; return PB_SOLVE; }
    uint8_t proof[256];
    int32_t rc = zkpor_prove_tail(ctx, pk, sol.w.data(), sol.a.data(), sol.b.data(), sol.c.data(), sol.n_constraints, r, s, proof);
    if (rc != ZKPOR_OK) { if (err) *err = std::string("prove: ") + zkpor_last_error(ctx); return PB_PROVE; }
    uint8_t raw[256 + 4 + 64 + 64];
    size_t raw_len = 0;
    rc = zkpor_proof_write_raw(proof, sol.has_commitment ? sol.commitment : nullptr, sol.has_commitment ? 1u : 0u, sol.has_commitment ? sol.pok : nullptr,
                               raw, sizeof raw, &raw_len);
    if (rc != ZKPOR_OK) { if (err) *err = "prove: raw encoding"; return PB_PROVE; }
    std::string raw_s((const char*)raw, raw_len);
    if (verify && verify(raw_s, w) != 0) { if (err) *err = "verify: the proof was rejected"; return PB_VERIFY; }   // prover.go:276-279
    *row = MakeProofRow(raw_s, w.BeforeCEXAssetsCommitment, w.AfterCEXAssetsCommitment, w.AccountTreeRoot, w.BatchCommitment,
                        w.MinAccountIndex, w.MaxAccountIndex, in.tier, batch_number);
    if (tier) *tier = in.tier;
    return PB_OK;
}

}  // namespace zkpor_host
