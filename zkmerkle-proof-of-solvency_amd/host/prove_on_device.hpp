// groth16.Prove (src/prover/prover/prover.go:269) from the ASSIGNED INPUTS with nothing but the inputs crossing PCIe — the C++ twin of
// go/zkporgpu/solver.go ProveOnDevice, and GenerateAndVerifyProof (prover.go:250-283; host/prove_batch.hpp) built on it:
//
//   inputs (1 | public | secret, Montgomery) --upload--> d_w
//     --zkpor_solver_start_dev--> the exported solver program fills the wire vector in HBM (csrc/solver.hip) ...
//         ... and pauses at gnark's BSB22 commitment placeholder: inputs = (commitment index, hashed public wires..., committed wires...)
//         --zkpor_solver_external_inputs_dev--> the committed wires, still on the device --zkpor_commit_dev--> (commitment, knowledge proof)
//         challenge = hash_to_field(commitment.Marshal() | hashed wires)   (bsb22_challenge.hpp: gnark's prove.go hashing)
//         --zkpor_solver_external_outputs--> the hint's output wire;  --zkpor_solver_resume_dev--> the rest of the program
//     --zkpor_solver_eval_abc_dev--> a, b, c (+ a x b = c on every row)   --zkpor_prove_tail_dev--> Ar, Bs, Krs
//
// The solver callback of prove_batch.hpp (gnark's solver, host memory) is not needed on this path.  C++ because the build image has no Go.
#pragma once
#include <cstring>
#include <functional>
#include <string>
#include <vector>
#include "../../include/zkpor.h"
#include "bsb22_challenge.hpp"
#include "fr_host.hpp"
#include "proof_row.hpp"
#include "witness_assign.hpp"
#include "witness_codec.hpp"

namespace zkpor_host {

// device memory of one proof in flight: w (n_wires) and a, b, c (domain each); reused from proof to proof by a worker
struct DeviceProofBuffers {
    zkpor_ctx* ctx = nullptr;
    void* w = nullptr;
    void* abc[3] = {nullptr, nullptr, nullptr};
    size_t n_wires = 0, domain = 0;
    int Reserve(zkpor_ctx* c, size_t wires, size_t dom) {
        if (c == ctx && wires == n_wires && dom == domain && w) return 0;
        Release();
        ctx = c; n_wires = wires; domain = dom;
        if (zkpor_dev_alloc(c, wires * 32, &w) != ZKPOR_OK) return 1;
        for (auto& p : abc) if (zkpor_dev_alloc(c, dom * 32, &p) != ZKPOR_OK) return 1;
        return 0;
    }
    void Release() {
        if (ctx) { if (w) zkpor_dev_free(ctx, w); for (auto& p : abc) if (p) zkpor_dev_free(ctx, p); }
        w = nullptr; abc[0] = abc[1] = abc[2] = nullptr; n_wires = domain = 0;
    }
    ~DeviceProofBuffers() { Release(); }
    DeviceProofBuffers() = default;
    DeviceProofBuffers(const DeviceProofBuffers&) = delete;
    DeviceProofBuffers& operator=(const DeviceProofBuffers&) = delete;
};

struct DeviceProof {
    uint8_t proof[256] = {0};                       // Ar | Bs | Krs, Montgomery limbs (zkpor_prove_tail's form)
    bool has_commitment = false;
    uint8_t commitment[64] = {0}, pok[64] = {0};
    uint8_t challenge[32] = {0};                    // the hint's output, big-endian canonical (what gnark's solver would have been handed)
};

// inputs: 1 + nPublic + nSecret Montgomery elements (wire 0 = ONE).  n_hashed: how many of the commitment hint's inputs after the index
// are hashed next to the commitment instead of being committed (gnark CommitmentInfo.PublicAndCommitmentCommitted; 0 for BatchCreateUserCircuit).
// 0 = ok; else `err` says which step failed.
inline int ProveOnDevice(zkpor_ctx* ctx, zkpor_pk* pk, zkpor_r1cs* r1cs, zkpor_solver* solver, DeviceProofBuffers* bufs, const uint64_t* inputs,
                         size_t n_inputs, size_t n_hashed, const uint64_t r[4], const uint64_t s[4], DeviceProof* out, std::string* err) {
    auto fail = [&](const char* step) { if (err) *err = std::string(step) + ": " + zkpor_last_error(ctx); return 1; };
    uint64_t dims[6];
    if (zkpor_pk_dims(pk, dims) != ZKPOR_OK) { if (err) *err = "prove: the key is not loaded"; return 1; }
    const size_t n_wires = dims[0], domain = (size_t)1 << dims[4];
    if (n_inputs == 0 || n_inputs > n_wires) { if (err) *err = "prove: the assignment does not fit the key's wire count"; return 1; }
    if (bufs->Reserve(ctx, n_wires, domain) != 0) return fail("device buffers");
    if (zkpor_dev_upload(ctx, bufs->w, inputs, n_inputs * 32) != ZKPOR_OK) return fail("upload");
    uint32_t paused = 0xffffffffu;
    // a, b, c named before the run: the Poseidon instructions write their own rows, the assertions are left to eval_abc's a x b = c pass (include/zkpor.h)
    if (zkpor_solver_set_abc_dev(solver, bufs->abc[0], bufs->abc[1], bufs->abc[2]) != ZKPOR_OK) return fail("a, b, c buffers");
    if (zkpor_solver_start_dev(solver, bufs->w, n_inputs, nullptr, &paused) != ZKPOR_OK) return fail("solve");
    while (paused != 0xffffffffu) {
        size_t n_in = 0, n_out = 0;
        if (zkpor_solver_external_inputs(solver, paused, nullptr, 0, &n_in, &n_out) != ZKPOR_OK) return fail("solve");
        if (out->has_commitment || n_out != 1 || n_in < 1 + n_hashed) { if (err) *err = "solve: the program stops at a hint this prover does not serve (instruction " + std::to_string(paused) + ")"; return 1; }
        void* d_in = nullptr;
        if (zkpor_dev_alloc(ctx, n_in * 32, &d_in) != ZKPOR_OK) return fail("device buffers");
        const size_t n_committed = n_in - 1 - n_hashed;
        int32_t rc = zkpor_solver_external_inputs_dev(solver, paused, d_in, n_in);
        if (rc == ZKPOR_OK) rc = zkpor_commit_dev(ctx, pk, (const char*)d_in + 32 * (1 + n_hashed), n_committed, out->commitment, out->pok);
        std::vector<uint64_t> hashed(4 * n_hashed);
        if (rc == ZKPOR_OK && n_hashed) rc = zkpor_dev_download(ctx, hashed.data(), (const char*)d_in + 32, n_hashed * 32);
        zkpor_dev_free(ctx, d_in);
        if (rc != ZKPOR_OK) return fail("commit");
        out->has_commitment = true;
        uint8_t be[64];
        zkpor_g1_marshal(out->commitment, be);
        std::vector<std::string> hashed_be;
        for (size_t i = 0; i < n_hashed; ++i) {      // Montgomery limbs -> canonical, big-endian (constraint.SerializeCommitment)
            FrH v; memcpy(v.v, &hashed[4 * i], 32);
            uint64_t c[4]; v.to_canon(c);
            std::string b32(32, '\0');
            for (int k = 0; k < 32; ++k) b32[31 - k] = (char)(c[k / 8] >> (8 * (k % 8)));
            hashed_be.push_back(b32);
        }
        const std::string ch = Bsb22Challenge(be, hashed_be);
        memcpy(out->challenge, ch.data(), 32);
        uint64_t canon[4] = {0, 0, 0, 0};
        for (int k = 0; k < 32; ++k) canon[k / 8] |= (uint64_t)(uint8_t)ch[31 - k] << (8 * (k % 8));
        const FrH chm = FrH::from_canon(canon);
        if (zkpor_solver_external_outputs(solver, paused, chm.v, 1) != ZKPOR_OK) return fail("solve");
        if (zkpor_solver_resume_dev(solver, &paused) != ZKPOR_OK) return fail("solve");
    }
    if (zkpor_solver_eval_abc_dev(solver, bufs->w, bufs->abc[0], bufs->abc[1], bufs->abc[2], domain) != ZKPOR_OK) return fail("constraint evaluation");   // and every row's a x b = c
    if (zkpor_prove_tail_dev(ctx, pk, bufs->w, bufs->abc[0], bufs->abc[1], bufs->abc[2], r, s, out->proof) != ZKPOR_OK) return fail("prove");
    return 0;
}

typedef std::function<int(const std::string& raw_proof, const BatchCreateUserWitnessW& w)> VerifyOnDeviceFn;  // groth16.Verify stand-in; may be empty
enum ProveOnDeviceErr { POD_OK = 0, POD_DECODE = 1, POD_ASSIGN = 2, POD_PROVE = 4, POD_VERIFY = 5 };

// Prover.GenerateAndVerifyProof (prover.go:250-283) on this path: witness-table row in, proof-table row out; the solver is the resident program
inline int GenerateAndVerifyProofOnDevice(zkpor_ctx* ctx, zkpor_pk* pk, zkpor_r1cs* r1cs, zkpor_solver* solver, DeviceProofBuffers* bufs,
                                          const std::string& witness_data, int64_t batch_number, const std::vector<int>& asset_counts_tiers,
                                          const uint64_t r[4], const uint64_t s[4], const VerifyOnDeviceFn& verify, ProofRow* row, int* tier,
                                          DeviceProof* proof_out, std::string* err) {
    BatchCreateUserWitnessW w;
    try { w = DecodeBatchWitness(witness_data, true); }
    catch (const std::exception& e) { if (err) *err = std::string("decode: ") + e.what(); return POD_DECODE; }
    AssignedWitness in;
    std::string why;
    if (!SetBatchCreateUserCircuitWitness(w, asset_counts_tiers, &in, &why)) { if (err) *err = "assign: " + why; return POD_ASSIGN; }
    std::vector<uint64_t> inputs(4 * (1 + in.values.size()));       // wire 0 = ONE, then public, then secret: gnark's wire order
    const FrH one = FrH::one();
    memcpy(inputs.data(), one.v, 32);
    for (size_t i = 0; i < in.values.size(); ++i) { const FrH m = FrH::from_canon(in.values[i].data()); memcpy(&inputs[4 * (1 + i)], m.v, 32); }
    DeviceProof p;
    if (ProveOnDevice(ctx, pk, r1cs, solver, bufs, inputs.data(), 1 + in.values.size(), 0, r, s, &p, &why) != 0) { if (err) *err = why; return POD_PROVE; }
    uint8_t raw[256 + 4 + 64 + 64];
    size_t raw_len = 0;
    if (zkpor_proof_write_raw(p.proof, p.has_commitment ? p.commitment : nullptr, p.has_commitment ? 1u : 0u, p.has_commitment ? p.pok : nullptr,
                              raw, sizeof raw, &raw_len) != ZKPOR_OK) { if (err) *err = "prove: raw encoding"; return POD_PROVE; }
    std::string raw_s((const char*)raw, raw_len);
    if (verify && verify(raw_s, w) != 0) { if (err) *err = "verify: the proof was rejected"; return POD_VERIFY; }   // prover.go:276-279
    *row = MakeProofRow(raw_s, w.BeforeCEXAssetsCommitment, w.AfterCEXAssetsCommitment, w.AccountTreeRoot, w.BatchCommitment,
                        w.MinAccountIndex, w.MaxAccountIndex, in.tier, batch_number);
    if (tier) *tier = in.tier;
    if (proof_out) *proof_out = p;
    return POD_OK;
}

}  // namespace zkpor_host
