// Host-side mirror of the reference's prover service for ONE process driving several GPUs
// (src/prover/prover/prover.go: NewProver :45, FetchBatchWitness :86, FetchBatchWitnessForRerun :107, Run :139).
// The reference fans work out with N OS processes + a Redis list (BRPOP, prover.go:72-84) + MySQL row status;
// here the list is an in-process queue and the tables are interfaces, so one process can feed 8 MI355X contexts
// (one worker thread per GPU, each calling the C ABI of include/zkpor.h through `ProveFn`).
// Semantics kept: exactly-once hand-out, status CAS Published -> Received -> Finished
// (witness_model.go:129-152), duplicate-proof guard (prover.go:208-225), rerun scan Received-then-Published
// (:107-137), exit when the queue is empty (:155-159).  Names follow the reference.
// The Go toolchain is absent from the build image, so this is C++ (the reference is compiled code).
#pragma once
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace zkpor_host {

enum { StatusPublished = 0, StatusReceived = 1, StatusFinished = 2 };  // witness_model.go:12-16
enum Err { Ok = 0, DbErrNotFound = 1, DbErrDuplicate = 2, QueueNil = 3, ProveFailed = 4 };

struct BatchWitness {  // witness_model.go:43-48
    int64_t Height = 0;
    std::string WitnessData;  // base64(s2(gob(BatchCreateUserWitness))): written by witness_host.hpp MakeWitnessRows, read by
                              // witness_codec.hpp DecodeBatchWitness (= utils.DecodeBatchWitness, src/utils/utils.go:704-742)
    int Status = StatusPublished;
};
struct Proof {  // proof_model.go:29-39
    std::string ProofInfo;  // base64(raw proof) in the reference; raw bytes here
    int64_t BatchNumber = 0;
    int AssetsCount = 0;
};

class WitnessModel {  // in-memory stand-in for the gorm table; every method is one "transaction"
public:
    void CreateBatchWitness(const BatchWitness& w) { std::lock_guard<std::mutex> g(mu_); rows_[w.Height] = w; }
    // compare-and-set of the status of the row at `height` (GetAndUpdateBatchesWitnessByHeight, witness_model.go:129)
    Err GetAndUpdateBatchesWitnessByHeight(int64_t height, int before, int after, std::vector<BatchWitness>* out) {
        std::lock_guard<std::mutex> g(mu_);
        auto it = rows_.find(height);
        if (it == rows_.end() || it->second.Status != before) return DbErrNotFound;
        it->second.Status = after;
        out->assign(1, it->second);
        return Ok;
    }
    Err GetLatestBatchWitnessByStatus(int status, BatchWitness* out) {
        std::lock_guard<std::mutex> g(mu_);
        for (auto it = rows_.rbegin(); it != rows_.rend(); ++it)
            if (it->second.Status == status) { *out = it->second; return Ok; }
        return DbErrNotFound;
    }
    // Rerun with several workers in ONE process: GetLatestBatchWitnessByStatus does not change the row (the reference's rerun is
    // one process per invocation, prover.go:107-137), so N workers would all fetch — and prove — the same height.  A worker
    // therefore takes the latest row of `status` that no other worker of this process is holding; the claim is in-memory only
    // (a crash loses it, which is what rerun is for) and is dropped by ReleaseClaim once the row is Finished.
    Err ClaimLatestBatchWitnessByStatus(int status, BatchWitness* out) {
        std::lock_guard<std::mutex> g(mu_);
        for (auto it = rows_.rbegin(); it != rows_.rend(); ++it)
            if (it->second.Status == status && !claimed_.count(it->first)) { claimed_[it->first] = true; *out = it->second; return Ok; }
        return DbErrNotFound;
    }
    void ReleaseClaim(int64_t height) { std::lock_guard<std::mutex> g(mu_); claimed_.erase(height); }
    Err UpdateBatchWitnessStatus(const BatchWitness& w, int status) {
        std::lock_guard<std::mutex> g(mu_);
        auto it = rows_.find(w.Height);
        if (it == rows_.end()) return DbErrNotFound;
        it->second.Status = status;
        return Ok;
    }
    size_t CountByStatus(int status) {
        std::lock_guard<std::mutex> g(mu_);
        size_t n = 0;
        for (auto& kv : rows_) n += kv.second.Status == status;
        return n;
    }
private:
    std::mutex mu_;
    std::map<int64_t, BatchWitness> rows_;
    std::map<int64_t, bool> claimed_;
};

class ProofModel {  // BatchNumber is unique (proof_model.go:33)
public:
    Err CreateProof(const Proof& p) {
        std::lock_guard<std::mutex> g(mu_);
        if (!rows_.emplace(p.BatchNumber, p).second) return DbErrDuplicate;
        return Ok;
    }
    Err GetProofByBatchNumber(int64_t n, Proof* out) {
        std::lock_guard<std::mutex> g(mu_);
        auto it = rows_.find(n);
        if (it == rows_.end()) return DbErrNotFound;
        if (out) *out = it->second;
        return Ok;
    }
    size_t Count() { std::lock_guard<std::mutex> g(mu_); return rows_.size(); }
private:
    std::mutex mu_;
    std::map<int64_t, Proof> rows_;
};

class TaskQueue {  // the Redis list por_batch_task_queue_{suffix}: LPUSH by dbtool, BRPOP by provers
public:
    void LPush(int64_t height) {
        { std::lock_guard<std::mutex> g(mu_); q_.push_front(height); }
        cv_.notify_one();
    }
    Err BRPop(std::chrono::milliseconds timeout, int64_t* height) {  // redis.Nil after the timeout
        std::unique_lock<std::mutex> g(mu_);
        if (!cv_.wait_for(g, timeout, [&] { return !q_.empty(); })) return QueueNil;
        *height = q_.back();
        q_.pop_back();
        return Ok;
    }
private:
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<int64_t> q_;
};

// proves one batch on one GPU context: fills the raw proof bytes and the tier; non-zero = error
typedef std::function<int(int gpu, const BatchWitness&, std::string* proof_raw, int* assets_count)> ProveFn;

class Prover {
public:
    Prover(WitnessModel* wm, ProofModel* pm, TaskQueue* q, int gpu, ProveFn fn, std::chrono::milliseconds brpop_timeout)
        : witnessModel(wm), proofModel(pm), queue(q), gpu_(gpu), prove_(std::move(fn)), timeout_(brpop_timeout) {}

    Err FetchBatchWitness(std::vector<BatchWitness>* out) {  // prover.go:86-105
        int64_t h;
        Err e = queue->BRPop(timeout_, &h);
        if (e != Ok) return e;
        return witnessModel->GetAndUpdateBatchesWitnessByHeight(h, StatusPublished, StatusReceived, out);
    }
    Err FetchBatchWitnessForRerun(std::vector<BatchWitness>* out) {  // prover.go:107-137
        BatchWitness w;
        Err e = witnessModel->ClaimLatestBatchWitnessByStatus(StatusReceived, &w);
        if (e == DbErrNotFound) e = witnessModel->ClaimLatestBatchWitnessByStatus(StatusPublished, &w);
        if (e != Ok) return e;
        out->assign(1, w);
        return Ok;
    }
    // prover.go:139-247.  Returns the number of proofs this worker created, or -1 after a prove failure.
    int Run(bool rerun) {
        int made = 0;
        for (;;) {
            std::vector<BatchWitness> batch;
            Err e = rerun ? FetchBatchWitnessForRerun(&batch) : FetchBatchWitness(&batch);
            if (e == QueueNil) return made;                      // "There is no task left in task queue" (:155-159)
            if (e == DbErrNotFound) return made;                 // queue mode: "no published status witness in db, so quit" (:150-154);
                                                                 // rerun: "no received status witness in db, so quit" (:167-171)
            for (auto& bw : batch) {
                std::string raw;
                int assets = 0;
                if (prove_(gpu_, bw, &raw, &assets) != 0) { if (rerun) witnessModel->ReleaseClaim(bw.Height); return -1; }
                if (proofModel->GetProofByBatchNumber(bw.Height, nullptr) == Ok) {  // duplicate-proof guard (:208-225)
                    witnessModel->UpdateBatchWitnessStatus(bw, StatusFinished);
                    if (rerun) witnessModel->ReleaseClaim(bw.Height);
                    continue;
                }
                Proof row;
                row.ProofInfo = raw; row.BatchNumber = bw.Height; row.AssetsCount = assets;
                Err ce = proofModel->CreateProof(row);
                // the unique index on BatchNumber decided a race between the guard above and this insert: the height IS proved
                // (by whoever won), which is all the status column records — not a failure of this worker
                if (ce != Ok && ce != DbErrDuplicate) { if (rerun) witnessModel->ReleaseClaim(bw.Height); return -1; }
                witnessModel->UpdateBatchWitnessStatus(bw, StatusFinished);
                if (rerun) witnessModel->ReleaseClaim(bw.Height);
                if (ce == Ok) ++made;
            }
        }
    }
    WitnessModel* witnessModel;
    ProofModel* proofModel;
    TaskQueue* queue;
private:
    int gpu_;
    ProveFn prove_;
    std::chrono::milliseconds timeout_;
};

// one worker thread (= one Prover, one zkpor_ctx) per GPU
class Dispatcher {
public:
    Dispatcher(int n_gpus, ProveFn fn, std::chrono::milliseconds brpop_timeout = std::chrono::milliseconds(50))
        : n_(n_gpus), fn_(std::move(fn)), timeout_(brpop_timeout) {}
    WitnessModel witnessModel;
    ProofModel proofModel;
    TaskQueue queue;
    // returns per-worker proof counts (or -1)
    std::vector<int> Run(bool rerun) {
        std::vector<int> made(n_, 0);
        std::vector<std::thread> th;
        for (int g = 0; g < n_; ++g)
            th.emplace_back([&, g] {
                Prover p(&witnessModel, &proofModel, &queue, g, fn_, timeout_);
                made[g] = p.Run(rerun);
            });
        for (auto& t : th) t.join();
        return made;
    }
private:
    int n_;
    ProveFn fn_;
    std::chrono::milliseconds timeout_;
};

// ---- solver || GPU pipeline (SURVEY.md §7 hard part 2) -------------------------------------------------------------------
// groth16.Prove is two very different halves: r1cs.Solve + hints on host cores (seconds per proof, a6.1 — stays gnark's Go code),
// then the prove tail on the GPU (~0.4 s).  Run back to back per proof the GPU idles > 90 % of the time; the service loop therefore
// runs them as two stages: `n_solvers` host threads solve batch i+1, i+2, ... while the GPU workers prove batch i, joined by a
// BOUNDED queue (a solved witness is ~8.6 GB of vectors: the bound is what keeps host memory finite when the GPU is the slow
// side).  Order of completion is free, every batch is handed over exactly once.
struct PipelineStats {
    double wall_s = 0;
    size_t proofs = 0;
    double solver_busy_s = 0;   // summed over solver threads: time inside solve()
    double solver_blocked_s = 0;  // ... waiting for room in the queue (GPU is the bottleneck)
    double gpu_busy_s = 0;      // summed over GPU workers: time inside prove()
    double gpu_starved_s = 0;   // ... waiting for a solved witness (solver is the bottleneck)
    size_t max_queued = 0;
};

template <class Solved>
class Pipeline {
public:
    typedef std::function<int(int64_t height, Solved* out)> SolveFn;            // host: witness row -> w, a, b, c
    typedef std::function<int(int gpu_worker, int64_t height, Solved& in)> ProveFn;  // device: prove tail + row insert
    Pipeline(int n_solvers, int n_gpu_workers, size_t queue_depth) : ns_(n_solvers), ng_(n_gpu_workers), depth_(queue_depth ? queue_depth : 1) {}

    // heights [0, n_batches); returns 0, or the first non-zero code of a stage (the pipeline then drains and stops)
    int Run(int64_t n_batches, SolveFn solve, ProveFn prove, PipelineStats* st) {
        typedef std::chrono::steady_clock clk;
        auto secs = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double>(b - a).count(); };
        std::mutex mu;
        std::condition_variable cv_room, cv_item;
        std::deque<std::pair<int64_t, Solved>> q;
        int64_t next = 0;
        int solvers_left = ns_, rc = 0;
        PipelineStats s;
        auto t0 = clk::now();
        std::vector<std::thread> th;
        for (int i = 0; i < ns_; ++i)
            th.emplace_back([&] {
                double busy = 0, blocked = 0;
                for (;;) {
                    int64_t h;
                    {
                        std::lock_guard<std::mutex> g(mu);
                        if (rc != 0 || next >= n_batches) break;
                        h = next++;
                    }
                    Solved item;
                    auto a = clk::now();
                    int e = solve(h, &item);
                    auto b = clk::now();
                    busy += secs(a, b);
                    std::unique_lock<std::mutex> g(mu);
                    if (e != 0) { if (rc == 0) rc = e; break; }
                    cv_room.wait(g, [&] { return q.size() < depth_ || rc != 0; });
                    blocked += secs(b, clk::now());
                    if (rc != 0) break;
                    q.emplace_back(h, std::move(item));
                    if (q.size() > s.max_queued) s.max_queued = q.size();
                    cv_item.notify_one();
                }
                std::lock_guard<std::mutex> g(mu);
                s.solver_busy_s += busy; s.solver_blocked_s += blocked;
                if (--solvers_left == 0) cv_item.notify_all();
                cv_room.notify_all();
            });
        for (int w = 0; w < ng_; ++w)
            th.emplace_back([&, w] {
                double busy = 0, starved = 0;
                size_t done = 0;
                for (;;) {
                    std::unique_lock<std::mutex> g(mu);
                    auto a = clk::now();
                    cv_item.wait(g, [&] { return !q.empty() || solvers_left == 0 || rc != 0; });
                    starved += secs(a, clk::now());
                    if (q.empty()) break;  // solvers are gone (or failed) and nothing is queued
                    auto item = std::move(q.front());
                    q.pop_front();
                    cv_room.notify_one();
                    g.unlock();
                    auto b = clk::now();
                    int e = prove(w, item.first, item.second);
                    busy += secs(b, clk::now());
                    if (e != 0) { std::lock_guard<std::mutex> g2(mu); if (rc == 0) rc = e; cv_room.notify_all(); cv_item.notify_all(); break; }
                    ++done;
                }
                std::lock_guard<std::mutex> g(mu);
                s.gpu_busy_s += busy; s.gpu_starved_s += starved; s.proofs += done;
            });
        for (auto& t : th) t.join();
        s.wall_s = secs(t0, clk::now());
        if (st) *st = s;
        return rc;
    }
private:
    int ns_, ng_;
    size_t depth_;
};

// static contiguous sharding used by the one-process-per-GPU launcher (bench.py / torchrun): rank r of `world`
// proves heights [lo, hi) — every height exactly once, sizes differ by at most one
inline void shard_range(int64_t n_batches, int rank, int world, int64_t* lo, int64_t* hi) {
    int64_t base = n_batches / world, extra = n_batches % world;
    *lo = rank * base + (rank < extra ? rank : extra);
    *hi = *lo + base + (rank < extra ? 1 : 0);
}

}  // namespace zkpor_host
