// C entry points over prover_host.hpp so the CPU test-suite (ctypes) can drive the dispatcher the way
// src/prover/prover/prover_test.go:TestMockProver drives the reference (many fake provers, no SNARK).
#include "prover_host.hpp"
#include "proof_row.hpp"
#include "r1cs_file.hpp"
#include "witness_codec.hpp"
#include "witness_assign.hpp"
#include "bsb22_challenge.hpp"
#include "solver_exec.hpp"
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
// ---- witness row codec (host/witness_codec.hpp), driven by tests/test_witness_codec_cpu.py ----
static uint64_t tmix(uint64_t& s) { uint64_t z = (s += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); }
static Bytes tbytes(uint64_t& s, size_t n) { Bytes b; for (size_t i = 0; i < n; ++i) b.push_back((char)(uint8_t)tmix(s)); return b; }
// a deterministic witness the Python test regenerates from the same seed (tests/gobs2.py synth_witness)
static BatchCreateUserWitnessW synth_witness(uint64_t seed, int users, int assets_per_user, int cex_assets) {
    uint64_t s = seed;
    BatchCreateUserWitnessW w;
    w.BatchCommitment = tbytes(s, 32); w.AccountTreeRoot = tbytes(s, 32);
    w.BeforeCEXAssetsCommitment = tbytes(s, 32); w.AfterCEXAssetsCommitment = tbytes(s, 32);
    w.MinAccountIndex = (uint32_t)tmix(s); w.MaxAccountIndex = (uint32_t)tmix(s);
    for (int i = 0; i < cex_assets; ++i) {
        CexAssetInfoW c;
        c.TotalEquity = tmix(s); c.TotalDebt = tmix(s) >> 20; c.BasePrice = tmix(s) >> 40;
        c.Symbol = (i % 3 == 0) ? std::string() : ("sym" + std::to_string(i));
        c.Index = (uint32_t)i;
        c.LoanCollateral = tmix(s) >> 8; c.MarginCollateral = (i % 2) ? tmix(s) : 0; c.PortfolioMarginCollateral = tmix(s) >> 1;
        std::array<TierRatioW, kTierCount>* lists[3] = {&c.LoanRatios, &c.MarginRatios, &c.PortfolioMarginRatios};
        for (int l = 0; l < 3; ++l)
            for (int t = 0; t < kTierCount; ++t) {
                TierRatioW& tr = (*lists[l])[t];
                if (l == 2 && i % 4 == 0) continue;   // an all-zero array: omitted on the wire
                tr.BoundaryValue = BigIntW::from_u128(((unsigned __int128)(tmix(s) >> 10) << 64) | tmix(s));
                tr.Ratio = (uint8_t)(tmix(s) % 101);
                tr.PrecomputedValue = BigIntW::from_u128(t == 0 ? 0 : (unsigned __int128)tmix(s) * (unsigned)(t));  // zero but present for t = 0
            }
        w.BeforeCexAssets.push_back(c);
    }
    for (int u = 0; u < users; ++u) {
        CreateUserOperationW op;
        for (int a = 0; a < assets_per_user; ++a) {
            AccountAssetW x;
            x.Index = (uint16_t)((a * 7 + u) % kAssetCounts);
            x.Equity = tmix(s) >> 24; x.Debt = (a % 2) ? tmix(s) >> 30 : 0; x.Loan = tmix(s) >> 50; x.Margin = 0; x.PortfolioMargin = tmix(s);
            op.Assets.push_back(x);
        }
        op.AccountIndex = (uint32_t)(1000 + u);
        op.AccountIdHash = tbytes(s, 32);
        for (int k = 0; k < kAccountTreeDepth; ++k) op.AccountProof[k] = tbytes(s, 32);
        w.CreateUserOps.push_back(op);
    }
    return w;
}
static long put_out(const std::string& r, char* out, size_t cap) {
    if (r.size() > cap) return -(long)r.size();
    memcpy(out, r.data(), r.size());
    return (long)r.size();
}
// stage: 0 = gob bytes, 1 = s2 block of them, 2 = the base64 column.  Returns the length (negative: needed capacity), -1 on error
long zkh_witness_synth_encode(uint64_t seed, int users, int assets_per_user, int cex_assets, int s2_level, int stage, char* out, size_t cap) {
    try {
        BatchCreateUserWitnessW w = synth_witness(seed, users, assets_per_user, cex_assets);
        Bytes g = witness_gob::Encode(w);
        if (stage == 0) return put_out(g, out, cap);
        Bytes z = s2::Encode(g, s2_level);
        if (stage == 1) return put_out(z, out, cap);
        return put_out(base64_std(z), out, cap);
    } catch (const std::exception&) { return -1; }
}
// utils.DecodeBatchWitness on `in` (any valid stream), then serializeWorker again: what a decode -> encode round trip preserves
long zkh_witness_reencode(const char* in, size_t in_len, int expand_assets, int s2_level, char* out, size_t cap, char* err, size_t err_len) {
    try {
        BatchCreateUserWitnessW w = DecodeBatchWitness(std::string(in, in_len), expand_assets != 0);
        return put_out(EncodeBatchWitness(w, s2_level), out, cap);
    } catch (const std::exception& e) {
        if (err && err_len) { snprintf(err, err_len, "%s", e.what()); }
        return -1;
    }
}
long zkh_s2(const char* in, size_t in_len, int decode, int level, char* out, size_t cap, char* err, size_t err_len) {
    try {
        return put_out(decode ? s2::Decode(std::string(in, in_len)) : s2::Encode(std::string(in, in_len), level), out, cap);
    } catch (const std::exception& e) {
        if (err && err_len) { snprintf(err, err_len, "%s", e.what()); }
        return -1;
    }
}
// the worked example of the encoding/gob package documentation: type Point struct{ X, Y int }; Point{22, 33} (type id 65)
long zkh_gob_point_example(char* out, size_t cap) {
    gob::WireType t; t.kind = gob::WireType::Struct; t.id = 65; t.name = "Point"; t.fields = {{"X", gob::tInt}, {"Y", gob::tInt}};
    Bytes o;
    gob::put_message(o, gob::type_definition(t));
    Bytes v;
    gob::put_int(v, 65);
    gob::put_uint(v, 1); gob::put_int(v, 22); gob::put_uint(v, 1); gob::put_int(v, 33); gob::put_uint(v, 0);
    gob::put_message(o, v);
    return put_out(o, out, cap);
}
// decode a WitnessData column and assign the circuit witness (host/witness_assign.hpp): out = n x 32 bytes (4 x u64 LE canonical limbs),
// counts = n_public, n_secret, tier.  Returns the number of values (negative: needed capacity in values), -1 on error
long zkh_witness_assign(const char* column, size_t column_len, const int* tiers, int n_tiers, uint64_t* out, size_t cap_values, uint64_t counts[3],
                        char* err, size_t err_len) {
    try {
        BatchCreateUserWitnessW w = DecodeBatchWitness(std::string(column, column_len), true);
        AssignedWitness a;
        std::string why;
        if (!SetBatchCreateUserCircuitWitness(w, std::vector<int>(tiers, tiers + n_tiers), &a, &why)) { if (err && err_len) snprintf(err, err_len, "%s", why.c_str()); return -1; }
        counts[0] = a.n_public; counts[1] = a.n_secret; counts[2] = (uint64_t)a.tier;
        if (a.values.size() > cap_values) return -(long)a.values.size();
        memcpy(out, a.values.data(), a.values.size() * 32);
        return (long)a.values.size();
    } catch (const std::exception& e) {
        if (err && err_len) snprintf(err, err_len, "%s", e.what());
        return -1;
    }
}
// ---- bsb22_challenge.hpp ----
void zkh_sha256(const uint8_t* msg, size_t len, uint8_t out[32]) { Sha256 h; h.Write(msg, len); h.Sum(out); }
// 0 = ok, 1 = refused (RFC 9380 limits)
int zkh_expand_msg_xmd(const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, uint8_t* out, size_t out_len) {
    try {
        std::string u = ExpandMsgXmd(std::string((const char*)msg, msg_len), std::string((const char*)dst, dst_len), out_len);
        memcpy(out, u.data(), out_len);
        return 0;
    } catch (const std::exception&) { return 1; }
}
// fr.Hash: count elements of 32 bytes big-endian each
int zkh_fr_hash(const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, size_t count, uint8_t* out) {
    try {
        auto v = FrHash(std::string((const char*)msg, msg_len), std::string((const char*)dst, dst_len), count);
        for (size_t i = 0; i < count; ++i) memcpy(out + 32 * i, v[i].data(), 32);
        return 0;
    } catch (const std::exception&) { return 1; }
}
// the BSB22 hint's output for one commitment: n_public values of 32 bytes big-endian follow the 64-byte commitment in the hash input
int zkh_bsb22_challenge(const uint8_t commitment[64], const uint8_t* public_be32, size_t n_public, uint8_t out[32]) {
    try {
        std::vector<std::string> pub;
        for (size_t i = 0; i < n_public; ++i) pub.emplace_back((const char*)public_be32 + 32 * i, 32);
        std::string c = Bsb22Challenge(commitment, pub);
        memcpy(out, c.data(), 32);
        return 0;
    } catch (const std::exception&) { return 1; }
}

// ---- the levelized solver executor (host/solver_exec.hpp): no device involved ----
// field helpers for the tests: out = a * b, out = a^-1 (Montgomery limbs), canonical <-> Montgomery
void zkh_fr_mul(const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n) {
    for (size_t i = 0; i < n; ++i) { FrH x, y; memcpy(x.v, a + 4 * i, 32); memcpy(y.v, b + 4 * i, 32); FrH r = FrH::mul(x, y); memcpy(out + 4 * i, r.v, 32); }
}
void zkh_fr_inv(const uint64_t* a, uint64_t* out, size_t n) {
    for (size_t i = 0; i < n; ++i) { FrH x; memcpy(x.v, a + 4 * i, 32); FrH r = FrH::inv(x); memcpy(out + 4 * i, r.v, 32); }
}
void zkh_fr_from_canon(const uint64_t* c, uint64_t* out, size_t n) { for (size_t i = 0; i < n; ++i) { FrH r = FrH::from_canon(c + 4 * i); memcpy(out + 4 * i, r.v, 32); } }
void zkh_fr_to_canon(const uint64_t* a, uint64_t* out, size_t n) { for (size_t i = 0; i < n; ++i) { FrH x; memcpy(x.v, a + 4 * i, 32); x.to_canon(out + 4 * i); } }
// header walk of the solver container: counts = nInstructions, nLevels, nHintNames, nCallData
int zkh_solver_parse(const uint8_t* data, size_t len, uint64_t counts[4], char* err, size_t err_len) {
    SolverView v;
    std::string why;
    if (ParseSolverFile(data, len, &v, &why) != 0) {
        if (err && err_len) { size_t n = why.size() < err_len - 1 ? why.size() : err_len - 1; memcpy(err, why.data(), n); err[n] = 0; }
        return 1;
    }
    counts[0] = v.n_instructions; counts[1] = v.n_levels; counts[2] = v.hint_names.size(); counts[3] = v.n_calldata;
    return 0;
}
// solve: inputs = nPublic + nSecret elements; pre_ids / pre_vals = wires filled elsewhere (the device generators); w_out nWires x 4,
// a/b/c_out nConstraints x 4 (all three NULL: w only, no row check); stats = solved constraints, hint calls, skipped instructions.  0 = ok, else the executor's code + err text
int zkh_solve(const uint8_t* r1cs, size_t r1cs_len, const uint8_t* solv, size_t solv_len, const uint64_t* inputs, size_t n_inputs,
              const uint32_t* pre_ids, const uint64_t* pre_vals, size_t n_pre, int threads, uint64_t* w_out, uint64_t* a_out, uint64_t* b_out,
              uint64_t* c_out, uint64_t stats[3], char* err, size_t err_len) {
    auto put = [&](const std::string& why) { if (err && err_len) { size_t n = why.size() < err_len - 1 ? why.size() : err_len - 1; memcpy(err, why.data(), n); err[n] = 0; } };
    R1csFileView rv;
    SolverView sv;
    std::string why;
    if (ParseR1csFile(r1cs, r1cs_len, &rv, &why) != 0) { put(why); return 1; }
    if (ParseSolverFile(solv, solv_len, &sv, &why) != 0) { put(why); return 1; }
    std::vector<std::pair<uint32_t, FrH>> pre(n_pre);
    for (size_t i = 0; i < n_pre; ++i) { pre[i].first = pre_ids[i]; memcpy(pre[i].second.v, pre_vals + 4 * i, 32); }
    SolveResult res;
    const bool want_abc = a_out && b_out && c_out;   // all three or none: without them only w is produced (a, b, c on the device: zkpor_prove_r1cs)
    int rc = SolveLevelized(rv, sv, inputs, n_inputs, HintRegistry::Standard(), pre, threads, &res, &why, want_abc);
    if (rc != 0) { put(why); return rc; }
    memcpy(w_out, res.w.data(), res.w.size() * 8);
    if (want_abc) { memcpy(a_out, res.a.data(), res.a.size() * 8); memcpy(b_out, res.b.data(), res.b.size() * 8); memcpy(c_out, res.c.data(), res.c.size() * 8); }
    stats[0] = res.solved_constraints; stats[1] = res.hint_calls; stats[2] = res.skipped;
    return 0;
}
}

// ---- the circuit compiler (host/circuit/*.hpp): BatchCreateUserCircuit.Define restated, compiled to matrices + solver program ----
#include "circuit/synth_batch.hpp"
extern "C" {
struct zkc_circuit {
    zkpor_circuit::Compiled c;
    zkpor_circuit::CircuitShape shape;
    std::vector<uint8_t> solver_container;
    std::string census_json;
};
static void zkc_put_err(char* err, size_t err_len, const std::string& why) { if (err && err_len) snprintf(err, err_len, "%s", why.c_str()); }
// inputs_mont (may be NULL = compile only): n_public - 1 + n_secret Montgomery elements — then the compile also INTERPRETS the circuit
// (every wire's value, every assertion checked; an assertion that fails is an error).  commitment_mont (may be NULL): the value of the
// BSB22 commitment wire in that interpretation.  poseidon_native 0: gnark's own form (three constraint instructions per S-box).
zkc_circuit* zkc_compile_batch_create_user(uint32_t user_assets, uint32_t all_assets, uint32_t users, const uint64_t* inputs_mont,
                                           const uint64_t* commitment_mont, int poseidon_native, char* err, size_t err_len) {
    try {
        auto* z = new zkc_circuit();
        z->shape.userAssetCounts = user_assets; z->shape.allAssetCounts = all_assets; z->shape.batchCounts = users;
        if (user_assets == 0 || users == 0 || user_assets > all_assets) { delete z; zkc_put_err(err, err_len, "bad shape"); return nullptr; }
        zkpor_circuit::Builder b(z->shape.n_public(), z->shape.n_secret(), (const FrH*)inputs_mont, poseidon_native != 0);
        if (commitment_mont) { FrH v; memcpy(v.v, commitment_mont, 32); b.set_commitment_value(v); }
        if (const char* e = getenv("ZKPOR_CIRCUIT_BESIDE")) b.set_beside(atoi(e) != 0);   // experiments: 0 = the challenge sponge as an ordinary call
        zkpor_circuit::DefineBatchCreateUser(b, z->shape);
        z->c = b.finish();
        z->solver_container = zkpor_circuit::SolverContainer(z->c);
        std::string j = "{";
        for (auto& kv : z->c.census) { if (j.size() > 1) j += ", "; j += "\"" + kv.first + "\": " + std::to_string(kv.second); }
        z->census_json = j + "}";
        return z;
    } catch (const std::exception& e) { zkc_put_err(err, err_len, e.what()); return nullptr; }
}
void zkc_free(zkc_circuit* z) { delete z; }
// dims: n_wires, n_public, n_secret, n_constraints, n_coeff, nnzL, nnzR, nnzO, n_instructions, n_levels, n_calldata, n_committed, commitment wire,
// wires without an L term, wires without an R term, solver container bytes
void zkc_dims(const zkc_circuit* z, uint64_t dims[16]) {
    const auto& c = z->c;
    uint64_t no_l = 0, no_r = 0;
    for (uint8_t f : c.in_l) no_l += !f;
    for (uint8_t f : c.in_r) no_r += !f;
    const uint64_t d[16] = {c.n_wires, c.n_public, c.n_secret, c.n_constraints, c.coeff.size(), c.cid[0].size(), c.cid[1].size(), c.cid[2].size(),
                            c.kind.size(), c.level_ptr.size() - 1, c.calldata.size(), c.committed.size(), c.commitment_wire, no_l, no_r, z->solver_container.size()};
    memcpy(dims, d, sizeof d);
}
const uint64_t* zkc_coeff(const zkc_circuit* z) { return (const uint64_t*)z->c.coeff.data(); }
const uint64_t* zkc_row_ptr(const zkc_circuit* z, int m) { return z->c.row_ptr[m].data(); }
const uint32_t* zkc_cid(const zkc_circuit* z, int m) { return z->c.cid[m].data(); }
const uint32_t* zkc_wid(const zkc_circuit* z, int m) { return z->c.wid[m].data(); }
const uint32_t* zkc_committed(const zkc_circuit* z) { return z->c.committed.data(); }
const uint8_t* zkc_in_l(const zkc_circuit* z) { return z->c.in_l.data(); }
const uint8_t* zkc_in_r(const zkc_circuit* z) { return z->c.in_r.data(); }
const uint64_t* zkc_values(const zkc_circuit* z) { return z->c.values.empty() ? nullptr : (const uint64_t*)z->c.values.data(); }
const uint8_t* zkc_solver_container(const zkc_circuit* z) { return z->solver_container.data(); }
const uint64_t* zkc_level_ptr(const zkc_circuit* z) { return z->c.level_ptr.data(); }
const char* zkc_census(const zkc_circuit* z) { return z->census_json.c_str(); }
// a valid synthetic batch of that shape (host/circuit/synth_batch.hpp): the assignment in Montgomery form (out: n_public - 1 + n_secret
// elements); returns the element count, or -1
long zkc_synth_inputs(uint32_t user_assets, uint32_t all_assets, uint32_t users, uint64_t seed, uint32_t first_index, uint64_t* out, size_t cap, char* err, size_t err_len) {
    try {
        zkpor_circuit::CircuitShape S; S.userAssetCounts = user_assets; S.allAssetCounts = all_assets; S.batchCounts = users;
        const auto w = zkpor_circuit::SynthBatchWitness(S, seed, first_index);
        const std::vector<FrH> a = zkpor_circuit::AssignMont(w, user_assets);
        if (a.size() != S.n_public() - 1 + S.n_secret()) { zkc_put_err(err, err_len, "assignment length differs from the circuit's input count"); return -1; }
        if (a.size() > cap) { zkc_put_err(err, err_len, "buffer too small"); return -1; }
        memcpy(out, a.data(), a.size() * 32);
        return (long)a.size();
    } catch (const std::exception& e) { zkc_put_err(err, err_len, e.what()); return -1; }
}
// the compiled program on the host executor (host/solver_exec.hpp) — commitment_mont: what the BSB22 placeholder returns.
// w_out: n_wires x 4.  0 = ok.
int zkc_solve_host_with(const zkc_circuit* z, const uint8_t* container, size_t container_len, const uint64_t* inputs_mont, const uint64_t* commitment_mont,
                        int threads, uint64_t* w_out, int check_rows, char* err, size_t err_len);
int zkc_solve_host(const zkc_circuit* z, const uint64_t* inputs_mont, const uint64_t* commitment_mont, int threads, uint64_t* w_out, int check_rows,
                   char* err, size_t err_len) {
    return zkc_solve_host_with(z, z->solver_container.data(), z->solver_container.size(), inputs_mont, commitment_mont, threads, w_out, check_rows, err, err_len);
}
// the same over ANOTHER program for the circuit's matrices (tests: a container with its levels rearranged must still solve — or must not)
int zkc_solve_host_with(const zkc_circuit* z, const uint8_t* container, size_t container_len, const uint64_t* inputs_mont, const uint64_t* commitment_mont,
                        int threads, uint64_t* w_out, int check_rows, char* err, size_t err_len) {
    const auto& c = z->c;
    R1csFileView rv;
    rv.n_constraints = c.n_constraints; rv.n_wires = c.n_wires; rv.n_public = c.n_public; rv.n_secret = c.n_secret; rv.n_coeff = c.coeff.size();
    rv.coeff = (const uint64_t*)c.coeff.data();
    for (int m = 0; m < 3; ++m) { rv.nnz[m] = c.cid[m].size(); rv.row_ptr[m] = c.row_ptr[m].data(); rv.coeff_ids[m] = c.cid[m].data(); rv.wire_ids[m] = c.wid[m].data(); }
    SolverView sv;
    std::string why;
    if (ParseSolverFile(container, container_len, &sv, &why) != 0) { zkc_put_err(err, err_len, why); return 1; }
    HintRegistry reg = HintRegistry::Standard();
    FrH cm; memcpy(cm.v, commitment_mont, 32);
    reg.by_name["bsb22CommitmentComputePlaceholder"] = [cm](const std::vector<FrH>&, std::vector<FrH>& out) { if (out.size() != 1) return 1; out[0] = cm; return 0; };
    std::vector<uint64_t> in(4 * (c.n_public + c.n_secret));
    const FrH one = FrH::one();
    memcpy(in.data(), one.v, 32);
    memcpy(in.data() + 4, inputs_mont, 32 * (c.n_public + c.n_secret - 1));
    SolveResult res;
    int rc = SolveLevelized(rv, sv, in.data(), c.n_public + c.n_secret, reg, {}, threads, &res, &why, check_rows != 0);
    if (rc != 0) { zkc_put_err(err, err_len, why); return rc; }
    memcpy(w_out, res.w.data(), res.w.size() * 8);
    return 0;
}
}
