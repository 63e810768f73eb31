// Reader of the flat constraint-system container written by go/export_r1cs: the compiled R1CS of
// groth16.NewCS().ReadFrom (src/prover/prover/prover.go:317-324) as a coefficient table + three CSR matrices + gnark's commitment
// info + wire counts, little-endian, 8-byte aligned sections so the file can be mapped and handed to zkpor_r1cs_* without copies.
// gnark's own .r1cs is CBOR + its instruction encoding (third-party, version-bound); the exporter runs once per tier on a box with
// Go, this reader has no dependency.  Layout: go/export_r1cs/main.go header comment.
#pragma once
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#include "../../include/zkpor.h"

namespace zkpor_host {

struct R1csCommitment {
    uint64_t commitment_index = 0;
    std::vector<uint32_t> private_committed, public_and_commitment_committed;
};

struct R1csFileView {  // pointers into the caller's (mapped) bytes
    uint64_t n_constraints = 0, n_wires = 0, n_public = 0, n_secret = 0, n_coeff = 0;
    uint64_t nnz[3] = {0, 0, 0};
    std::vector<R1csCommitment> commitments;
    const uint64_t* coeff = nullptr;        // n_coeff x 4
    const uint64_t* row_ptr[3] = {nullptr, nullptr, nullptr};
    const uint32_t* coeff_ids[3] = {nullptr, nullptr, nullptr};
    const uint32_t* wire_ids[3] = {nullptr, nullptr, nullptr};
    // the wires pk.G1.K leaves out besides the public ones: what zkpor_pk_set_consts / zkpor_pk_load_gnark take as committed_idx
    std::vector<uint32_t> CommittedWires() const {
        std::vector<uint32_t> out;
        for (auto& c : commitments) {
            out.insert(out.end(), c.private_committed.begin(), c.private_committed.end());
            out.push_back((uint32_t)c.commitment_index);
        }
        return out;
    }
};

// 0 = ok; every count is checked against the length of the stream before a pointer is formed
inline int ParseR1csFile(const uint8_t* data, size_t len, R1csFileView* out, std::string* err) {
    auto fail = [&](const char* m) { if (err) *err = std::string("r1cs file: ") + m; return 1; };
    size_t off = 0;
    auto need = [&](uint64_t n) { return n <= len && off <= len - n; };
    if (!data || !need(8) || memcmp(data, "ZKPR1CS\x01", 8) != 0) return fail("bad magic");
    off = 8;
    uint64_t h[9];
    if (!need(sizeof h)) return fail("truncated header");
    memcpy(h, data + off, sizeof h);
    off += sizeof h;
    R1csFileView v;
    v.n_constraints = h[0]; v.n_wires = h[1]; v.n_public = h[2]; v.n_secret = h[3]; v.n_coeff = h[4];
    v.nnz[0] = h[5]; v.nnz[1] = h[6]; v.nnz[2] = h[7];
    const uint64_t n_com = h[8];
    if (v.n_wires == 0 || v.n_wires >= 0xffffffffull || v.n_public + v.n_secret > v.n_wires || v.n_public == 0) return fail("bad wire counts");
    if (v.n_coeff == 0 || v.n_coeff >= 0xffffffffull || v.n_constraints >= (1ull << 32)) return fail("bad counts");
    if (n_com > 64) return fail("too many commitments");
    for (uint64_t i = 0; i < n_com; ++i) {
        uint64_t c[3];
        if (!need(sizeof c)) return fail("truncated commitment info");
        memcpy(c, data + off, sizeof c);
        off += sizeof c;
        if (c[0] >= v.n_wires || c[1] > v.n_wires || c[2] > v.n_wires || !need(4 * (c[1] + c[2]))) return fail("bad commitment info");
        R1csCommitment rc;
        rc.commitment_index = c[0];
        rc.private_committed.resize(c[1]);
        rc.public_and_commitment_committed.resize(c[2]);
        memcpy(rc.private_committed.data(), data + off, 4 * c[1]);
        off += 4 * c[1];
        memcpy(rc.public_and_commitment_committed.data(), data + off, 4 * c[2]);
        off += 4 * c[2];
        for (uint32_t w : rc.private_committed) if (w >= v.n_wires) return fail("committed wire out of range");
        v.commitments.push_back(std::move(rc));
    }
    off = (off + 7) & ~(size_t)7;
    if (!need(32 * v.n_coeff)) return fail("truncated coefficient table");
    v.coeff = (const uint64_t*)(data + off);
    off += 32 * v.n_coeff;
    for (int m = 0; m < 3; ++m) {
        if (!need(8 * (v.n_constraints + 1))) return fail("truncated row pointers");
        v.row_ptr[m] = (const uint64_t*)(data + off);
        off += 8 * (v.n_constraints + 1);
        if (v.row_ptr[m][0] != 0 || v.row_ptr[m][v.n_constraints] != v.nnz[m]) return fail("row pointers do not span the terms");
        if (v.nnz[m] > (len - off) / 8) return fail("truncated terms");
        v.coeff_ids[m] = (const uint32_t*)(data + off);
        off += 4 * v.nnz[m];
        v.wire_ids[m] = (const uint32_t*)(data + off);
        off += 4 * v.nnz[m];
        off = (off + 7) & ~(size_t)7;
        if (off > len) return fail("truncated terms");
    }
    if (off != len) return fail("trailing bytes");
    *out = std::move(v);
    return 0;
}

// the mapped system -> HBM (zkpor_r1cs_set_matrix validates every coefficient / wire id and the monotonicity of the row pointers)
inline int32_t LoadR1cs(zkpor_ctx* ctx, const R1csFileView& v, zkpor_r1cs** out) {
    zkpor_r1cs* r = nullptr;
    int32_t rc = zkpor_r1cs_create(ctx, v.n_constraints, v.n_wires, v.coeff, v.n_coeff, &r);
    if (rc != ZKPOR_OK) return rc;
    for (int m = 0; m < 3; ++m) {
        rc = zkpor_r1cs_set_matrix(r, m, v.row_ptr[m], v.coeff_ids[m], v.wire_ids[m], v.nnz[m]);
        if (rc != ZKPOR_OK) { zkpor_r1cs_destroy(r); return rc; }
    }
    *out = r;
    return ZKPOR_OK;
}

}  // namespace zkpor_host
