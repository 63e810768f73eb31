// C entry points over prover_host.hpp so the CPU test-suite (ctypes) can drive the dispatcher the way
// src/prover/prover/prover_test.go:TestMockProver drives the reference (many fake provers, no SNARK).
#include "prover_host.hpp"
#include "proof_row.hpp"
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
}
