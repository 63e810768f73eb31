// TEST DRIVER (-m gpu): the reference's prover service loop (host/prover_host.hpp: Dispatcher -> Prover::Run, mirroring
// src/prover/prover/prover.go:139-247) driving REAL proofs through the C ABI with several contexts — one worker thread per
// context, contexts mapped round-robin onto the visible GPUs (on a one-GPU box: several contexts on device 0, which is also
// how two proofs are kept in flight per GPU).  Plays the role of the Go caller; loaded by tests/test_dispatcher_gpu.py.
#include "../../zkmerkle-proof-of-solvency_amd/host/prover_host.hpp"
#include "../../zkmerkle-proof-of-solvency_amd/host/r1cs_file.hpp"
#include "../../zkmerkle-proof-of-solvency_amd/host/prove_batch.hpp"
#include "../../zkmerkle-proof-of-solvency_amd/host/prove_on_device.hpp"
#include "../../include/zkpor.h"
#include <atomic>
#include <cstring>

using namespace zkpor_host;

extern "C" {

// WitnessData of batch h = w | a | b | c (n_wires + 3 n_constraints Fr, Montgomery limbs) — what the solver would hand over.
// blinding: r = r_s[8 h .. 8 h + 4), s = r_s[8 h + 4 .. 8 h + 8).  proofs_out: n_batches x 256 B, indexed by height.
// worker_of (n_batches ints): which worker proved each height.  Returns 0, or a negative code (last error text in err).
int dispatch_gpu_run(int n_workers, int n_devices, int64_t n_batches, int log2_domain, size_t n_wires, size_t n_constraints,
                     uint64_t key_seed, const uint64_t* vectors, const uint64_t* r_s, uint8_t* proofs_out, int* worker_of, int* made_out,
                     char* err, size_t err_len) {
    std::vector<zkpor_ctx*> ctx(n_workers, nullptr);
    std::vector<zkpor_pk*> pk(n_devices, nullptr);
    auto fail = [&](const char* what, zkpor_ctx* c) {
        snprintf(err, err_len, "%s: %s", what, c ? zkpor_last_error(c) : "");
        return -1;
    };
    // contexts are created HERE, on the caller's thread, and used from the worker threads: the library binds every call to the
    // handle's device itself (device affinity)
    for (int g = 0; g < n_workers; ++g)
        if (zkpor_init(g % n_devices, nullptr, &ctx[g]) != ZKPOR_OK) return fail("zkpor_init", nullptr);
    for (int d = 0; d < n_devices; ++d) {  // one key per GPU, shared by that GPU's contexts
        if (zkpor_pk_create(ctx[d], &pk[d]) != ZKPOR_OK) return fail("pk_create", ctx[d]);
        if (zkpor_pk_synth(pk[d], log2_domain, n_wires, 3, 0, key_seed) != ZKPOR_OK) return fail("pk_synth", ctx[d]);
    }
    const size_t per = (n_wires + 3 * n_constraints) * 4;
    std::atomic<int> bad{0};
    Dispatcher disp(n_workers, [&](int g, const BatchWitness& bw, std::string* raw, int* assets) -> int {
        const uint64_t* v = (const uint64_t*)bw.WitnessData.data();
        const uint64_t* w = v; const uint64_t* a = w + 4 * n_wires; const uint64_t* b = a + 4 * n_constraints; const uint64_t* c = b + 4 * n_constraints;
        uint8_t proof[256];
        int32_t rc = zkpor_prove_tail(ctx[g], pk[g % n_devices], w, a, b, c, n_constraints, r_s + 8 * bw.Height, r_s + 8 * bw.Height + 4, proof);
        if (rc != ZKPOR_OK) { bad++; return 1; }
        raw->assign((const char*)proof, 256);
        *assets = 50;
        worker_of[bw.Height] = g;
        return 0;
    });
    for (int64_t h = 0; h < n_batches; ++h) {
        BatchWitness bw;
        bw.Height = h;
        bw.WitnessData.assign((const char*)(vectors + (size_t)h * per), per * 8);
        disp.witnessModel.CreateBatchWitness(bw);
        disp.queue.LPush(h);
    }
    std::vector<int> made = disp.Run(false);
    int rc = 0;
    for (int g = 0; g < n_workers; ++g) { made_out[g] = made[g]; if (made[g] < 0) rc = -2; }
    if (bad.load()) { rc = -3; snprintf(err, err_len, "prove failed: %s", zkpor_last_error(ctx[0])); }
    if (rc == 0 && (disp.proofModel.Count() != (size_t)n_batches || disp.witnessModel.CountByStatus(StatusFinished) != (size_t)n_batches)) {
        rc = -4; snprintf(err, err_len, "rows: %zu proofs, %zu finished of %ld", disp.proofModel.Count(), disp.witnessModel.CountByStatus(StatusFinished), (long)n_batches);
    }
    for (int64_t h = 0; h < n_batches && rc == 0; ++h) {
        Proof p;
        if (disp.proofModel.GetProofByBatchNumber(h, &p) != Ok || p.ProofInfo.size() != 256) { rc = -5; break; }
        memcpy(proofs_out + 256 * h, p.ProofInfo.data(), 256);
    }
    for (auto* k : pk) if (k) zkpor_pk_destroy(k);
    for (auto* c : ctx) if (c) zkpor_destroy(c);
    return rc;
}

// f1 ingestion: the exported constraint system (go/export_r1cs container) -> HBM -> a, b, c = L.w, R.w, O.w
int r1cs_file_eval(const uint8_t* data, size_t len, const uint64_t* w, uint64_t* a, uint64_t* b, uint64_t* c, char* err, size_t err_len) {
    R1csFileView v;
    std::string why;
    if (ParseR1csFile(data, len, &v, &why) != 0) { snprintf(err, err_len, "%s", why.c_str()); return -1; }
    zkpor_ctx* ctx = nullptr;
    if (zkpor_init(0, nullptr, &ctx) != ZKPOR_OK) { snprintf(err, err_len, "zkpor_init"); return -2; }
    zkpor_r1cs* r = nullptr;
    int rc = 0;
    if (LoadR1cs(ctx, v, &r) != ZKPOR_OK) { snprintf(err, err_len, "load: %s", zkpor_last_error(ctx)); rc = -3; }
    else if (zkpor_r1cs_eval(r, w, a, b, c) != ZKPOR_OK) { snprintf(err, err_len, "eval: %s", zkpor_last_error(ctx)); rc = -4; }
    if (r) zkpor_r1cs_destroy(r);
    zkpor_destroy(ctx);
    return rc;
}

// a1 end to end: a witness-table row -> decode -> assign -> solver (stub: the test's own solution of its own circuit, after checking
// what it was handed) -> commitment -> prove tail -> raw proof -> proof-table row as a CSV line.  The key is loaded by the caller
// (Python) into ctx/pk handles passed in as pointers.
static uint8_t g_last_challenge[32];
// what the commit callback handed the (stub) solver as the BSB22 hint's output in the last prove_batch_row call
void prove_batch_last_challenge(uint8_t out[32]) { memcpy(out, g_last_challenge, 32); }
long prove_batch_row(void* ctx_h, void* pk_h, const char* column, size_t column_len, int64_t batch, const uint64_t* w, const uint64_t* a,
                     const uint64_t* b, const uint64_t* c, size_t n_wires, size_t n_constraints, const uint64_t* committed, size_t n_committed,
                     const uint64_t* r, const uint64_t* s, uint64_t expect_inputs, int fail_stage, char* out, size_t cap, int* tier,
                     uint8_t* raw_out, size_t* raw_len, char* err, size_t err_len) {
    zkpor_ctx* ctx = (zkpor_ctx*)ctx_h;
    zkpor_pk* pk = (zkpor_pk*)pk_h;
    std::string raw_seen;
    SolveFn solve = [&](const AssignedWitness& in, const CommitFn& commit, SolvedWitness* sol) -> int {
        if (in.values.size() != expect_inputs || in.n_public != 1) return 2;   // the assigned vector really reached the solver
        if (fail_stage == PB_SOLVE) return 1;
        sol->w.assign(w, w + 4 * n_wires); sol->a.assign(a, a + 4 * n_constraints);
        sol->b.assign(b, b + 4 * n_constraints); sol->c.assign(c, c + 4 * n_constraints);
        sol->n_constraints = n_constraints;
        if (n_committed) { uint8_t cm[64], k[64]; if (commit(committed, n_committed, cm, k, g_last_challenge) != 0) return 3; }
        return 0;
    };
    VerifyFn verify = [&](const std::string& raw, const BatchCreateUserWitnessW&) -> int { raw_seen = raw; return fail_stage == PB_VERIFY ? 1 : 0; };
    ProofRow row;
    std::string why;
    int rc = GenerateAndVerifyProof(ctx, pk, std::string(column, column_len), batch, {50, 500}, solve, r, s, verify, &row, tier, &why);
    if (rc != PB_OK) { snprintf(err, err_len, "%s", why.c_str()); return -rc; }
    if (raw_seen.size() > 512) return -100;
    memcpy(raw_out, raw_seen.data(), raw_seen.size());
    *raw_len = raw_seen.size();
    std::string line = std::string(ProofCsvHeader()) + ProofCsvLine(row);
    if (line.size() > cap) return -101;
    memcpy(out, line.data(), line.size());
    return (long)line.size();
}
// a1 end to end WITHOUT a host solver (host/prove_on_device.hpp): a witness-table row -> decode -> assign -> the inputs cross PCIe -> the solver
// program, the BSB22 commitment, a / b / c and the prove tail on the device -> raw proof -> proof-table row as a CSV line.  For the test the
// solved wire vector and h are copied back as well (w_out: n_wires x 4, h_out: domain x 4; NULL = not wanted).
long prove_row_on_device(void* ctx_h, void* pk_h, void* r1cs_h, void* solver_h, const char* column, size_t column_len, int64_t batch, const uint64_t* r,
                         const uint64_t* s, int fail_verify, char* out, size_t cap, int* tier, uint8_t* raw_out, size_t* raw_len, uint8_t* proof256,
                         uint8_t* challenge32, uint64_t* w_out, uint64_t* h_out, char* err, size_t err_len) {
    zkpor_ctx* ctx = (zkpor_ctx*)ctx_h;
    DeviceProofBuffers bufs;
    std::string raw_seen;
    VerifyOnDeviceFn verify = [&](const std::string& raw, const BatchCreateUserWitnessW&) -> int { raw_seen = raw; return fail_verify ? 1 : 0; };
    ProofRow row;
    DeviceProof p;
    std::string why;
    int rc = GenerateAndVerifyProofOnDevice(ctx, (zkpor_pk*)pk_h, (zkpor_r1cs*)r1cs_h, (zkpor_solver*)solver_h, &bufs, std::string(column, column_len), batch,
                                            {50, 500}, r, s, verify, &row, tier, &p, &why);
    if (rc != POD_OK) { snprintf(err, err_len, "%s", why.c_str()); return -rc; }
    if (raw_seen.size() > 512) return -100;
    memcpy(raw_out, raw_seen.data(), raw_seen.size());
    *raw_len = raw_seen.size();
    memcpy(proof256, p.proof, 256);
    memcpy(challenge32, p.challenge, 32);
    if (w_out && zkpor_dev_download(ctx, w_out, bufs.w, bufs.n_wires * 32) != ZKPOR_OK) return -102;
    if (h_out && zkpor_dev_download(ctx, h_out, bufs.abc[0], bufs.domain * 32) != ZKPOR_OK) return -102;   // prove_tail_dev leaves h in a
    std::string line = std::string(ProofCsvHeader()) + ProofCsvLine(row);
    if (line.size() > cap) return -101;
    memcpy(out, line.data(), line.size());
    return (long)line.size();
}
}
