// C entry points over prover_host.hpp so the CPU test-suite (ctypes) can drive the dispatcher the way
// src/prover/prover/prover_test.go:TestMockProver drives the reference (many fake provers, no SNARK).
#include "prover_host.hpp"
#include "proof_row.hpp"
#include "r1cs_file.hpp"
#include <atomic>
#include <cstring>
using namespace zkpor_host;

extern "C" {
typedef int (*zkh_prove_cb)(int gpu, int64_t height, const char* witness, size_t witness_len, char* proof_out, size_t cap, size_t* proof_len, int* assets);

struct zkh_dispatcher {
    Dispatcher* d;
    std::atomic<long> calls{0};
};

zkh_dispatcher* zkh_create(int n_gpus, zkh_prove_cb cb, int brpop_timeout_ms) {
    auto* h = new zkh_dispatcher();
    h->d = new Dispatcher(n_gpus, [cb, h](int gpu, const BatchWitness& w, std::string* raw, int* assets) {
        h->calls++;
        char buf[512];
        size_t len = 0;
        int rc = cb(gpu, w.Height, w.WitnessData.data(), w.WitnessData.size(), buf, sizeof buf, &len, assets);
        if (rc == 0) raw->assign(buf, len);
        return rc;
    }, std::chrono::milliseconds(brpop_timeout_ms));
    return h;
}
void zkh_destroy(zkh_dispatcher* h) { delete h->d; delete h; }
void zkh_add_witness(zkh_dispatcher* h, int64_t height, const char* data, size_t len, int status) {
    BatchWitness w; w.Height = height; w.WitnessData.assign(data, len); w.Status = status;
    h->d->witnessModel.CreateBatchWitness(w);
}
void zkh_push_task(zkh_dispatcher* h, int64_t height) { h->d->queue.LPush(height); }
// made_out: n_gpus entries
void zkh_run(zkh_dispatcher* h, int rerun, int* made_out, int n) {
    std::vector<int> m = h->d->Run(rerun != 0);
    for (int i = 0; i < n && i < (int)m.size(); ++i) made_out[i] = m[i];
}
long zkh_count_status(zkh_dispatcher* h, int status) { return (long)h->d->witnessModel.CountByStatus(status); }
long zkh_count_proofs(zkh_dispatcher* h) { return (long)h->d->proofModel.Count(); }
long zkh_prove_calls(zkh_dispatcher* h) { return h->calls.load(); }
int zkh_get_proof(zkh_dispatcher* h, int64_t batch, char* out, size_t cap, size_t* len) {
    Proof p;
    if (h->d->proofModel.GetProofByBatchNumber(batch, &p) != Ok) return 1;
    size_t n = p.ProofInfo.size() < cap ? p.ProofInfo.size() : cap;
    memcpy(out, p.ProofInfo.data(), n);
    *len = n;
    return 0;
}
int zkh_insert_proof(zkh_dispatcher* h, int64_t batch) { Proof p; p.BatchNumber = batch; return h->d->proofModel.CreateProof(p); }
void zkh_shard_range(int64_t n, int rank, int world, int64_t* lo, int64_t* hi) { shard_range(n, rank, world, lo, hi); }
// one proof-table row as a CSV line (header first when with_header != 0); returns the length, or -1 if `cap` is too small
long zkh_proof_csv(const char* raw, size_t raw_len, const char* before32, const char* after32, const char* root32, const char* commit,
                   size_t commit_len, uint32_t min_idx, uint32_t max_idx, int assets, int64_t batch, int with_header, char* out, size_t cap) {
    ProofRow r = MakeProofRow(std::string(raw, raw_len), std::string(before32, 32), std::string(after32, 32), std::string(root32, 32),
                              std::string(commit, commit_len), min_idx, max_idx, assets, batch);
    std::string line = (with_header ? std::string(ProofCsvHeader()) : std::string()) + ProofCsvLine(r);
    if (line.size() > cap) return -1;
    memcpy(out, line.data(), line.size());
    return (long)line.size();
}
// header walk of the flat constraint-system container (host/r1cs_file.hpp; no device): counts = n_constraints, n_wires, n_public,
// n_secret, n_coeff, nnzL, nnzR, nnzO, n_commitments, number of wires K leaves out; committed_out (may be NULL) receives those wires
int zkh_r1cs_parse(const uint8_t* data, size_t len, uint64_t counts[10], uint32_t* committed_out, size_t committed_cap, char* err, size_t err_len) {
    R1csFileView v;
    std::string why;
    if (ParseR1csFile(data, len, &v, &why) != 0) {
        if (err && err_len) { size_t n = why.size() < err_len - 1 ? why.size() : err_len - 1; memcpy(err, why.data(), n); err[n] = 0; }
        return 1;
    }
    std::vector<uint32_t> cw = v.CommittedWires();
    uint64_t c[10] = {v.n_constraints, v.n_wires, v.n_public, v.n_secret, v.n_coeff, v.nnz[0], v.nnz[1], v.nnz[2], v.commitments.size(), cw.size()};
    memcpy(counts, c, sizeof c);
    if (committed_out) for (size_t i = 0; i < cw.size() && i < committed_cap; ++i) committed_out[i] = cw[i];
    return 0;
}
// the solver || GPU pipeline with sleeping stages (CPU test of the queueing logic): stats = wall_s, proofs, solver_busy_s,
// solver_blocked_s, gpu_busy_s, gpu_starved_s, max_queued; fail_at >= 0 makes the solve of that height fail
int zkh_pipeline_sim(int n_solvers, int n_gpu_workers, size_t depth, int64_t n_batches, int solve_ms, int prove_ms, int64_t fail_at,
                     double stats[7], int64_t* proved_heights) {
    Pipeline<int64_t> p(n_solvers, n_gpu_workers, depth);
    std::mutex mu;
    size_t k = 0;
    PipelineStats st;
    int rc = p.Run(n_batches,
        [&](int64_t h, int64_t* out) { std::this_thread::sleep_for(std::chrono::milliseconds(solve_ms)); if (h == fail_at) return 7; *out = h * 3 + 1; return 0; },
        [&](int, int64_t h, int64_t& in) {
            std::this_thread::sleep_for(std::chrono::milliseconds(prove_ms));
            if (in != h * 3 + 1) return 9;
            std::lock_guard<std::mutex> g(mu);
            proved_heights[k++] = h;
            return 0;
        }, &st);
    double o[7] = {st.wall_s, (double)st.proofs, st.solver_busy_s, st.solver_blocked_s, st.gpu_busy_s, st.gpu_starved_s, (double)st.max_queued};
    memcpy(stats, o, sizeof o);
    return rc;
}
}
