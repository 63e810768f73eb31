// Drives zkpor_host::FixedDepthMerkleTree (host/merkle_tree.hpp) the way src/witness/main.go:130-199 drives the
// reference tree: NewAccountTree -> Set per account -> Build -> Root / GetProof / VerifyMerkleProof.
// usage: tree_driver <nil_leaf_hex> <capacity> <key>...     leaf of key k = Fr(k+1) (merkletree_test.go:32-37)
// prints: root <hex> / proof <key> <hex x 28> per key / verify <0|1>; exit code 0 iff every API call behaved
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../zkmerkle-proof-of-solvency_amd/host/merkle_tree.hpp"
using namespace zkpor_host;

static Hash32 from_hex(const char* s) {
    Hash32 h{};
    for (int i = 0; i < 32; ++i) { unsigned v = 0; sscanf(s + 2 * i, "%2x", &v); h[i] = (uint8_t)v; }
    return h;
}
static void print_hex(const Hash32& h) { for (uint8_t b : h) printf("%02x", b); }
static Hash32 leaf_value(uint64_t k) {
    Hash32 h{};
    uint64_t v = k + 1;
    for (int i = 0; i < 8; ++i) h[31 - i] = (uint8_t)(v >> (8 * i));
    return h;
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    zkpor_ctx* ctx = nullptr;
    if (zkpor_init(0, nullptr, &ctx) != ZKPOR_OK) { fprintf(stderr, "no device\n"); return 3; }
    Hash32 nil = from_hex(argv[1]);
    uint64_t capacity = strtoull(argv[2], nullptr, 10);
    int bad = 0;
    {
        FixedDepthMerkleTree* tree = NewAccountTree(ctx, nil, capacity);
        std::vector<uint32_t> keys;
        for (int i = 3; i < argc; ++i) keys.push_back((uint32_t)strtoul(argv[i], nullptr, 10));
        for (uint32_t k : keys) if (!tree->Set(k, leaf_value(k))) { fprintf(stderr, "Set(%u): %s\n", k, tree->Error().c_str()); bad = 1; }
        if (tree->Set((uint32_t)capacity, leaf_value(0))) { fprintf(stderr, "Set beyond capacity succeeded\n"); bad = 1; }
        tree->Build();
        Hash32 root = tree->Root();
        printf("root "); print_hex(root); printf("\n");
        int all_ok = 1;
        for (uint32_t k : keys) {
            std::vector<Hash32> proof;
            if (!tree->GetProof(k, &proof)) { bad = 1; continue; }
            printf("proof %u", k);
            for (auto& p : proof) { printf(" "); print_hex(p); }
            printf("\n");
            all_ok &= VerifyMerkleProof(ctx, root, k, proof, tree->Get(k)) ? 1 : 0;
            all_ok &= VerifyMerkleProof(ctx, root, k ^ 1u, proof, tree->Get(k)) ? 0 : 1;  // wrong index must fail
        }
        printf("verify %d\n", all_ok);
        try { FixedDepthMerkleTree t2(ctx, 4, nil, 17); bad = 1; } catch (const std::invalid_argument&) {}
        delete tree;
    }
    zkpor_destroy(ctx);
    return bad;
}
