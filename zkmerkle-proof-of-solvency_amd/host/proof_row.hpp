// The `proof` table row and the proof CSV line exactly as the reference writes them, so that what this backend emits is read
// by the UNMODIFIED src/verifier (SURVEY.md §8b "file/wire formats that must not change"):
//   row     src/prover/prover/prover.go:180-236, proof_model.go:29-39
//             ProofInfo               = base64.StdEncoding(proof.WriteRawTo bytes)           (zkpor_proof_write_raw, 388 B)
//             CexAssetListCommitments = json.Marshal([][]byte{before, after})  -> ["<b64>","<b64>"]
//             AccountTreeRoots        = json.Marshal([][]byte{root})           -> ["<b64>"]
//             BatchCommitment         = base64.StdEncoding(batch commitment)
//   CSV     src/dbtool/main.go:260-289 (gocsv.MarshalFile = encoding/csv), read back at src/verifier/main.go:127-141:
//             batch_number,proof_info,cex_asset_list_commitments,account_tree_roots,batch_commitment,min_account_index,
//             max_account_index,assets_count ; a field holding a quote or a comma is quoted with "" escapes, lines end in \n.
// Host-only, no device.  Go is absent from the build image, hence C++ (the reference is compiled code).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace zkpor_host {

inline std::string base64_std(const uint8_t* p, size_t n) {  // RFC 4648 §4 with padding = Go base64.StdEncoding
    static const char T[] = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    std::string o;
    o.reserve((n + 2) / 3 * 4);
    size_t i = 0;
    for (; i + 3 <= n; i += 3) {
        uint32_t v = (uint32_t)p[i] << 16 | (uint32_t)p[i + 1] << 8 | p[i + 2];
        o += T[v >> 18]; o += T[(v >> 12) & 63]; o += T[(v >> 6) & 63]; o += T[v & 63];
    }
    if (n - i == 1) {
        uint32_t v = (uint32_t)p[i] << 16;
        o += T[v >> 18]; o += T[(v >> 12) & 63]; o += "==";
    } else if (n - i == 2) {
        uint32_t v = (uint32_t)p[i] << 16 | (uint32_t)p[i + 1] << 8;
        o += T[v >> 18]; o += T[(v >> 12) & 63]; o += T[(v >> 6) & 63]; o += '=';
    }
    return o;
}
inline std::string base64_std(const std::string& s) { return base64_std((const uint8_t*)s.data(), s.size()); }

// encoding/json of a [][]byte: every element as a base64 string (none of base64's characters is escaped by Go)
inline std::string json_bytes_array(const std::vector<std::string>& items) {
    std::string o = "[";
    for (size_t i = 0; i < items.size(); ++i) {
        if (i) o += ',';
        o += '"'; o += base64_std(items[i]); o += '"';
    }
    return o + "]";
}

struct ProofRow {  // proof_model.go:29-39 (gorm.Model columns aside)
    std::string ProofInfo, CexAssetListCommitments, AccountTreeRoots, BatchCommitment;
    uint32_t MinAccountIndex = 0, MaxAccountIndex = 0;
    int AssetsCount = 0;
    int64_t BatchNumber = 0;
};

// prover.go:180-236: raw = proof.WriteRawTo bytes; the four 32-byte values come from the decoded witness
inline ProofRow MakeProofRow(const std::string& raw_proof, const std::string& before_cex_commitment, const std::string& after_cex_commitment,
                             const std::string& account_tree_root, const std::string& batch_commitment, uint32_t min_account_index,
                             uint32_t max_account_index, int assets_count, int64_t batch_number) {
    ProofRow r;
    r.ProofInfo = base64_std(raw_proof);
    r.CexAssetListCommitments = json_bytes_array({before_cex_commitment, after_cex_commitment});
    r.AccountTreeRoots = json_bytes_array({account_tree_root});
    r.BatchCommitment = base64_std(batch_commitment);
    r.MinAccountIndex = min_account_index; r.MaxAccountIndex = max_account_index;
    r.AssetsCount = assets_count; r.BatchNumber = batch_number;
    return r;
}

inline std::string csv_field(const std::string& f) {  // encoding/csv Writer.fieldNeedsQuotes + quoting
    bool q = f.empty() ? false : (f[0] == ' ');
    for (char c : f) if (c == ',' || c == '"' || c == '\r' || c == '\n') q = true;
    if (!q) return f;
    std::string o = "\"";
    for (char c : f) { if (c == '"') o += '"'; o += c; }
    return o + "\"";
}
inline const char* ProofCsvHeader() {
    return "batch_number,proof_info,cex_asset_list_commitments,account_tree_roots,batch_commitment,min_account_index,max_account_index,assets_count\n";
}
inline std::string ProofCsvLine(const ProofRow& r) {
    return std::to_string(r.BatchNumber) + "," + csv_field(r.ProofInfo) + "," + csv_field(r.CexAssetListCommitments) + "," +
           csv_field(r.AccountTreeRoots) + "," + csv_field(r.BatchCommitment) + "," + std::to_string(r.MinAccountIndex) + "," +
           std::to_string(r.MaxAccountIndex) + "," + std::to_string(r.AssetsCount) + "\n";
}

}  // namespace zkpor_host
