// Host-side mirror of the reference's account tree API over the device-resident tree of libzkpor
// (src/utils/merkletree/merkletree.go: NewFixedDepthMerkleTree :137, Set :179, Build :192, Root :282, Get :287,
// GetProof :297, VerifyProof :334; src/utils/account_tree.go: NewAccountTree :14, VerifyMerkleProof :25).
// Same names, argument meaning and error behaviour, so src/witness/main.go:130-199 (buildAccountTree) reads the same
// against this class: parallel leaf hashing becomes one zkpor_poseidon_leaves call, Set/Build/GetProof keep their
// shape.  The Go toolchain is absent from the build image, hence C++ (the reference is compiled code); the cgo
// binding of the same entry points is in INTEGRATION.md.  No CPU fallback: every hash is computed by the library.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/zkpor.h"

namespace zkpor_host {

typedef std::array<uint8_t, 32> Hash32;
static const int AccountTreeDepth = 28;  // src/utils/constants.go:18

class FixedDepthMerkleTree {
public:
    // panics of the reference constructor (depth > 32, depth <= 0, capacity > 2^depth) surface as std::invalid_argument
    FixedDepthMerkleTree(zkpor_ctx* ctx, int depth, const Hash32& nilLeafHash, uint64_t capacity) : ctx_(ctx), depth_(depth) {
        int32_t rc = zkpor_tree_create(ctx, depth, nilLeafHash.data(), capacity, &t_);
        if (rc != ZKPOR_OK) throw std::invalid_argument(std::string("NewFixedDepthMerkleTree: ") + zkpor_last_error(ctx));
    }
    ~FixedDepthMerkleTree() { zkpor_tree_destroy(t_); }
    FixedDepthMerkleTree(const FixedDepthMerkleTree&) = delete;
    FixedDepthMerkleTree& operator=(const FixedDepthMerkleTree&) = delete;

    // Set stores one leaf; returns false with Error() set for a key >= capacity (the reference returns an error)
    bool Set(uint32_t key, const Hash32& value) { return SetMany(&key, value.data(), 1); }
    // the batched form the device wants: n keys, n x 32 bytes
    bool SetMany(const uint32_t* keys, const uint8_t* values32, size_t n) { return ok(zkpor_tree_set(t_, keys, values32, n)); }
    void Build() { if (!ok(zkpor_tree_build(t_))) throw std::runtime_error("Build: " + err_); }
    Hash32 Root() const { Hash32 r; zkpor_tree_root(t_, r.data()); return r; }
    Hash32 Get(uint32_t key) { Hash32 r; if (!ok(zkpor_tree_get(t_, &key, 1, r.data()))) throw std::runtime_error("Get: " + err_); return r; }
    // GetProof: depth siblings, leaf level first; false with Error() set when key >= 2^depth
    bool GetProof(uint32_t key, std::vector<Hash32>* proof) {
        proof->resize(depth_);
        return ok(zkpor_tree_get_proofs(t_, &key, 1, (*proof)[0].data()));
    }
    // U proofs in one launch (witness.go:323 fetches one per user of the batch)
    bool GetProofs(const std::vector<uint32_t>& keys, std::vector<Hash32>* proofs) {
        proofs->resize(keys.size() * (size_t)depth_);
        if (keys.empty()) return true;
        return ok(zkpor_tree_get_proofs(t_, keys.data(), keys.size(), (*proofs)[0].data()));
    }
    const std::string& Error() const { return err_; }
    zkpor_tree* handle() { return t_; }

private:
    bool ok(int32_t rc) { if (rc == ZKPOR_OK) return true; err_ = zkpor_last_error(ctx_); return false; }
    zkpor_ctx* ctx_;
    zkpor_tree* t_ = nullptr;
    int depth_;
    std::string err_;
};

// merkletree.VerifyProof (:334-355)
inline bool VerifyProof(zkpor_ctx* ctx, const Hash32& root, uint32_t key, const std::vector<Hash32>& proof, const Hash32& leaf, int depth) {
    if ((int)proof.size() != depth || (depth < 32 && (key >> depth) != 0)) return false;
    uint8_t okb = 0;
    if (zkpor_merkle_verify_proofs(ctx, root.data(), &key, proof[0].data(), leaf.data(), 1, depth, &okb) != ZKPOR_OK) return false;
    return okb != 0;
}
// utils.NewAccountTree (account_tree.go:14-23) / utils.VerifyMerkleProof (:25-29)
inline FixedDepthMerkleTree* NewAccountTree(zkpor_ctx* ctx, const Hash32& NilAccountHash, uint64_t capacity) {
    return new FixedDepthMerkleTree(ctx, AccountTreeDepth, NilAccountHash, capacity);
}
inline bool VerifyMerkleProof(zkpor_ctx* ctx, const Hash32& root, uint32_t accountIndex, const std::vector<Hash32>& proof, const Hash32& node) {
    return VerifyProof(ctx, root, accountIndex, proof, node, AccountTreeDepth);
}

}  // namespace zkpor_host
