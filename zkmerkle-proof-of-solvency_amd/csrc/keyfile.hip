// gnark Groth16 key containers: the byte stream `pk.WriteTo` puts on disk (src/keygen/main.go:46) and
// `pk.UnsafeReadFrom` reads back (src/prover/prover/prover.go:343).  The container is walked on the host (a few hundred
// bytes of headers and length prefixes; the point arrays are only located, not touched), then the arrays are handed —
// still compressed, straight out of the mapped file — to the device decompressor (decompress.hip) through the same
// setters a cgo caller would use.  Nothing here is a CPU fallback: without a device the loaders fail like every other call.
//
// Layout (gnark v0.10 backend/groth16/bn254/marshal.go writeTo + gnark-crypto v0.14 fft.Domain.WriteTo and
// pedersen.ProvingKey.WriteTo, as pinned by go.mod:57-60; third-party, restated from the published sources — the
// reference holds no key file to pin it against):
//   domain   : Cardinality u64 | CardinalityInv | Generator | GeneratorInv | FrMultiplicativeGen | FrMultiplicativeGenInv
//              (5 x 32 B big-endian Fr) | withPrecompute (1 B, written by gnark-crypto >= v0.12; older streams lack it)
//   G1       : Alpha | Beta | Delta (32 B compressed each)
//   slices   : A | B | Z | K, each u32 big-endian length + 32 B compressed points
//   G2       : Beta | Delta (64 B compressed each), then slice B (u32 length + 64 B points)
//   wires    : nbWires u64 | NbInfinityA u64 | NbInfinityB u64 | InfinityA (nbWires x 1 B) | InfinityB (nbWires x 1 B)
//   Pedersen : u32 number of commitment keys, then per key: Basis | BasisExpSigma (u32 length + 32 B points each)
// All integers big-endian.  The walk is self-checking: every count must be consistent with the others and the stream must
// end exactly at the last byte, which is also how the two domain-header variants are told apart.
#include "common.cuh"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

zkpor_ctx* zk_pk_ctx(zkpor_pk* pk);  // groth16.hip
int32_t zk_pk_finalize(zkpor_pk* pk, const void* alpha, const void* beta, const void* delta, const void* beta2, const void* delta2,
                       int log2_domain, const uint8_t* inf_a, const uint8_t* inf_b, size_t n_wires, const uint8_t* removed,
                       size_t n_public, int z_order, bool shard, size_t z_n);  // groth16.hip

namespace {

struct Cursor {
    const uint8_t* p;
    size_t len, off = 0;
    bool ok = true;
    bool need(size_t n) { if (!ok || n > len - off) { ok = false; return false; } return true; }
    uint64_t be(int bytes) {
        if (!need((size_t)bytes)) return 0;
        uint64_t v = 0;
        for (int i = 0; i < bytes; ++i) v = (v << 8) | p[off + i];
        off += bytes;
        return v;
    }
    // a length-prefixed point slice: returns the count, stores the offset of the first point
    uint64_t slice(size_t elem, uint64_t* first) {
        uint64_t n = be(4);
        *first = off;
        if (!ok || n > (len - off) / elem) { ok = false; return 0; }
        off += n * elem;
        return n;
    }
    void skip(size_t n) { if (need(n)) off += n; }
};

bool walk(const uint8_t* data, size_t len, uint32_t domain_bytes, zkpor_pk_layout_t* L, std::string* why) {
    memset(L, 0, sizeof(*L));
    Cursor c{data, len};
    L->domain_cardinality = c.be(8);
    L->domain_header_bytes = domain_bytes;
    c.skip(domain_bytes - 8);
    if (!c.ok) { *why = "truncated inside the domain header"; return false; }
    if (L->domain_cardinality == 0 || (L->domain_cardinality & (L->domain_cardinality - 1)) || L->domain_cardinality > (1ull << 28)) {
        *why = "domain cardinality is not a power of two <= 2^28"; return false;
    }
    if (domain_bytes == 169 && data[168] > 1) { *why = "withPrecompute byte is not 0/1"; return false; }
    L->off_alpha = c.off; c.skip(3 * 32);
    L->n_a = c.slice(32, &L->off_a);
    L->n_b1 = c.slice(32, &L->off_b1);
    L->n_z = c.slice(32, &L->off_z);
    L->n_k = c.slice(32, &L->off_k);
    L->off_beta2 = c.off; c.skip(2 * 64);
    L->n_b2 = c.slice(64, &L->off_b2);
    L->n_wires = c.be(8);
    L->n_inf_a = c.be(8);
    L->n_inf_b = c.be(8);
    if (!c.ok) { *why = "truncated inside the point arrays"; return false; }
    if (L->n_wires > len) { *why = "wire count larger than the stream"; return false; }
    L->off_inf_a = c.off; c.skip(L->n_wires);
    L->off_inf_b = c.off; c.skip(L->n_wires);
    L->n_commitment_keys = (uint32_t)c.be(4);
    if (!c.ok) { *why = "truncated inside the infinity masks"; return false; }
    for (uint32_t k = 0; k < L->n_commitment_keys; ++k) {
        uint64_t o1, o2;
        uint64_t n1 = c.slice(32, &o1);
        uint64_t n2 = c.slice(32, &o2);
        if (!c.ok) { *why = "truncated inside a commitment key"; return false; }
        if (n1 != n2) { *why = "commitment key: Basis and BasisExpSigma differ in length"; return false; }
        if (k == 0) { L->n_basis = n1; L->off_basis = o1; L->n_basis_sigma = n2; L->off_basis_sigma = o2; }
    }
    L->bytes_total = c.off;
    if (c.off != len) { *why = "stream does not end after the last commitment key"; return false; }
    // cross-checks between the counts
    if (L->n_a + L->n_inf_a != L->n_wires) { *why = "len(A) + NbInfinityA != nbWires"; return false; }
    if (L->n_b1 + L->n_inf_b != L->n_wires) { *why = "len(B) + NbInfinityB != nbWires"; return false; }
    if (L->n_b2 != L->n_b1) { *why = "G1.B and G2.B differ in length"; return false; }
    if (L->n_z != L->domain_cardinality && L->n_z + 1 != L->domain_cardinality) { *why = "len(Z) is neither the domain size nor one less"; return false; }
    if (L->n_k > L->n_wires) { *why = "len(K) > nbWires"; return false; }
    uint64_t ia = 0, ib = 0;
    for (uint64_t i = 0; i < L->n_wires; ++i) {
        uint8_t a = data[L->off_inf_a + i], b = data[L->off_inf_b + i];
        if (a > 1 || b > 1) { *why = "infinity mask byte is not 0/1"; return false; }
        ia += a; ib += b;
    }
    if (ia != L->n_inf_a || ib != L->n_inf_b) { *why = "infinity masks do not add up to NbInfinityA/B"; return false; }
    return true;
}

void put_err(char* err, size_t err_len, const std::string& s) {
    if (err && err_len) { snprintf(err, err_len, "%s", s.c_str()); }
}

int32_t layout(const uint8_t* data, size_t len, zkpor_pk_layout_t* out, std::string* why) {
    if (!data || !out) { *why = "null argument"; return ZKPOR_E_ARG; }
    std::string w169, w168;
    if (walk(data, len, 169, out, &w169)) return ZKPOR_OK;
    zkpor_pk_layout_t alt;
    if (walk(data, len, 168, &alt, &w168)) { *out = alt; return ZKPOR_OK; }
    *why = "not a gnark bn254 Groth16 proving key (pk.WriteTo): " + w169 + " [with the withPrecompute byte]; " + w168 + " [without]";
    return ZKPOR_E_ARG;
}

}  // namespace

extern "C" {

int32_t zkpor_pk_gnark_layout(const uint8_t* data, size_t len, zkpor_pk_layout_t* out, char* err, size_t err_len) try {
    std::string why;
    int32_t rc = layout(data, len, out, &why);
    if (rc != ZKPOR_OK) put_err(err, err_len, why);
    return rc;
} ZK_ABI_CATCH

int32_t zkpor_pk_load_gnark_mem(zkpor_pk* pk, const uint8_t* data, size_t len, size_t n_public, const uint32_t* committed_idx,
                                size_t n_committed, int z_order, zkpor_pk_layout_t* info) try {
    ZK_ENTER(pk ? zk_pk_ctx(pk)->device : -1);
    if (!pk) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = zk_pk_ctx(pk);
    if (z_order != ZKPOR_Z_ORDER_BITREV && z_order != ZKPOR_Z_ORDER_NATURAL) { ctx->err = "pk file: unknown z_order"; return ZKPOR_E_ARG; }
    zkpor_pk_layout_t L;
    std::string why;
    int32_t rc = layout(data, len, &L, &why);
    if (rc != ZKPOR_OK) { ctx->err = why; return rc; }
    if (info) *info = L;
    if (L.n_wires == 0 || L.n_wires >= 0xffffffffull) { ctx->err = "pk file: wire count out of range"; return ZKPOR_E_ARG; }
    // K leaves out the public wires, the committed wires and the commitment wires (Appendix A.1): the caller names them
    if (n_public > L.n_wires || L.n_k + n_public + n_committed != L.n_wires) {
        ctx->err = "pk file: len(K) = " + std::to_string(L.n_k) + " does not equal nbWires - n_public - n_committed = " +
                   std::to_string(L.n_wires) + " - " + std::to_string(n_public) + " - " + std::to_string(n_committed);
        return ZKPOR_E_ARG;
    }
    if (L.n_commitment_keys > 1) { ctx->err = "pk file: more than one commitment key (the reference circuit has one)"; return ZKPOR_E_ARG; }
    int log2d = 0;
    while ((1ull << log2d) < L.domain_cardinality) ++log2d;

    uint8_t g1c[3 * 64], g2c[2 * 128];
    rc = zkpor_g1_decompress(ctx, data + L.off_alpha, 3, g1c);
    if (rc != ZKPOR_OK) { ctx->err = "pk file: G1 alpha/beta/delta: " + ctx->err; return rc; }
    rc = zkpor_g2_decompress(ctx, data + L.off_beta2, 2, g2c);
    if (rc != ZKPOR_OK) { ctx->err = "pk file: G2 beta/delta: " + ctx->err; return rc; }

    struct { int which; uint64_t off, n; const char* name; } g1s[] = {
        {ZKPOR_G1_A, L.off_a, L.n_a, "G1.A"}, {ZKPOR_G1_B, L.off_b1, L.n_b1, "G1.B"}, {ZKPOR_G1_K, L.off_k, L.n_k, "G1.K"},
        // the prover uses the first Cardinality-1 points of Z; dropping the last one is order-independent for a stream
        // of Cardinality points in bit-reversed order (index 2^k - 1 is its own reversal)
        {ZKPOR_G1_Z, L.off_z, L.domain_cardinality - 1, "G1.Z"},
        {ZKPOR_G1_COMMIT_BASIS, L.off_basis, L.n_basis, "CommitmentKeys[0].Basis"},
        {ZKPOR_G1_COMMIT_BASIS_SIGMA, L.off_basis_sigma, L.n_basis_sigma, "CommitmentKeys[0].BasisExpSigma"}};
    for (auto& a : g1s) {
        rc = zkpor_pk_set_g1_compressed(pk, a.which, a.n ? data + a.off : data, a.n);
        if (rc != ZKPOR_OK) { ctx->err = std::string("pk file: ") + a.name + ": " + ctx->err; return rc; }
    }
    rc = zkpor_pk_set_g2_compressed(pk, ZKPOR_G2_B, L.n_b2 ? data + L.off_b2 : data, L.n_b2);
    if (rc != ZKPOR_OK) { ctx->err = "pk file: G2.B: " + ctx->err; return rc; }
    return zkpor_pk_set_consts(pk, g1c, g1c + 64, g1c + 128, g2c, g2c + 128, log2d, data + L.off_inf_a, data + L.off_inf_b,
                               (size_t)L.n_wires, n_public, committed_idx, n_committed, z_order);
} ZK_ABI_CATCH_IN((pk ? zk_pk_ctx(pk) : nullptr))

// One rank's share of a split key: wires [wire_lo, wire_hi) and Z points [z_lo, z_hi) only.  A, B and K are
// stored compacted, so the wire range is translated into ranges of the compacted arrays by counting the mask bytes in front of it;
// only those sub-ranges are uploaded and decompressed — a GPU never holds more than its share of a 2^28 key.
int32_t zkpor_pk_load_gnark_shard_mem(zkpor_pk* pk, const uint8_t* data, size_t len, size_t n_public, const uint32_t* committed_idx,
                                      size_t n_committed, size_t wire_lo, size_t wire_hi, size_t z_lo, size_t z_hi, int z_order,
                                      zkpor_pk_layout_t* info) try {
    ZK_ENTER(pk ? zk_pk_ctx(pk)->device : -1);
    if (pk && z_order != ZKPOR_Z_ORDER_BITREV) {
        // [z_lo, z_hi) is a range of the PROVER's order of h; in a natural-order file those points are scattered over the whole
        // Z section.  Load the whole key with ZKPOR_Z_ORDER_NATURAL and cut it with zkpor_pk_keep_range instead.
        zk_pk_ctx(pk)->err = "pk file: a shard can only be cut from a Z stored in the prover's (bit-reversed) order";
        return ZKPOR_E_ARG;
    }
    if (!pk) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = zk_pk_ctx(pk);
    zkpor_pk_layout_t L;
    std::string why;
    int32_t rc = layout(data, len, &L, &why);
    if (rc != ZKPOR_OK) { ctx->err = why; return rc; }
    if (info) *info = L;
    if (L.n_wires == 0 || L.n_wires >= 0xffffffffull) { ctx->err = "pk file: wire count out of range"; return ZKPOR_E_ARG; }
    if (n_public > L.n_wires || L.n_k + n_public + n_committed != L.n_wires) {
        ctx->err = "pk file: len(K) does not equal nbWires - n_public - n_committed"; return ZKPOR_E_ARG;
    }
    if (L.n_commitment_keys > 1) { ctx->err = "pk file: more than one commitment key (the reference circuit has one)"; return ZKPOR_E_ARG; }
    if (wire_lo >= wire_hi || wire_hi > L.n_wires || z_lo > z_hi || z_hi > L.domain_cardinality - 1) {
        ctx->err = "pk file: shard range outside the key"; return ZKPOR_E_ARG;
    }
    int log2d = 0;
    while ((1ull << log2d) < L.domain_cardinality) ++log2d;
    std::vector<uint8_t> removed(L.n_wires, 0);
    for (size_t i = 0; i < n_public; ++i) removed[i] = 1;
    for (size_t j = 0; j < n_committed; ++j) {
        if (committed_idx[j] >= L.n_wires) { ctx->err = "pk file: committed index out of range"; return ZKPOR_E_ARG; }
        removed[committed_idx[j]] = 1;
    }
    const uint8_t* ia = data + L.off_inf_a;
    const uint8_t* ib = data + L.off_inf_b;
    uint64_t a_lo = 0, b_lo = 0, k_lo = 0, a_n = 0, b_n = 0, k_n = 0, pub = 0;
    for (size_t i = 0; i < wire_hi; ++i) {
        const bool in = i >= wire_lo;
        (in ? a_n : a_lo) += !ia[i];
        (in ? b_n : b_lo) += !ib[i];
        (in ? k_n : k_lo) += !removed[i];
        if (in && i < n_public) ++pub;
    }
    uint8_t g1c[3 * 64], g2c[2 * 128];
    rc = zkpor_g1_decompress(ctx, data + L.off_alpha, 3, g1c);
    if (rc != ZKPOR_OK) { ctx->err = "pk file: G1 alpha/beta/delta: " + ctx->err; return rc; }
    rc = zkpor_g2_decompress(ctx, data + L.off_beta2, 2, g2c);
    if (rc != ZKPOR_OK) { ctx->err = "pk file: G2 beta/delta: " + ctx->err; return rc; }
    struct { int which; uint64_t off, n; const char* name; } g1s[] = {
        {ZKPOR_G1_A, L.off_a + 32 * a_lo, a_n, "G1.A"}, {ZKPOR_G1_B, L.off_b1 + 32 * b_lo, b_n, "G1.B"},
        {ZKPOR_G1_K, L.off_k + 32 * k_lo, k_n, "G1.K"}, {ZKPOR_G1_Z, L.off_z + 32 * z_lo, z_hi - z_lo, "G1.Z"},
        {ZKPOR_G1_COMMIT_BASIS, L.off_basis, L.n_basis, "CommitmentKeys[0].Basis"},
        {ZKPOR_G1_COMMIT_BASIS_SIGMA, L.off_basis_sigma, L.n_basis_sigma, "CommitmentKeys[0].BasisExpSigma"}};
    for (auto& a : g1s) {
        rc = zkpor_pk_set_g1_compressed(pk, a.which, a.n ? data + a.off : data, a.n);
        if (rc != ZKPOR_OK) { ctx->err = std::string("pk file: ") + a.name + ": " + ctx->err; return rc; }
    }
    rc = zkpor_pk_set_g2_compressed(pk, ZKPOR_G2_B, b_n ? data + L.off_b2 + 64 * b_lo : data, b_n);
    if (rc != ZKPOR_OK) { ctx->err = "pk file: G2.B: " + ctx->err; return rc; }
    return zk_pk_finalize(pk, g1c, g1c + 64, g1c + 128, g2c, g2c + 128, log2d, ia + wire_lo, ib + wire_lo, wire_hi - wire_lo,
                          removed.data() + wire_lo, (size_t)pub, ZKPOR_Z_ORDER_BITREV, true, z_hi - z_lo);
} ZK_ABI_CATCH_IN((pk ? zk_pk_ctx(pk) : nullptr))

static int32_t with_mapped_file(zkpor_ctx* ctx, const char* path, const std::function<int32_t(const uint8_t*, size_t)>& fn) {
    int fd = open(path, O_RDONLY);
    if (fd < 0) { ctx->err = std::string("pk file: cannot open ") + path; return ZKPOR_E_ARG; }
    struct stat st;
    if (fstat(fd, &st) != 0 || st.st_size <= 0) { close(fd); ctx->err = std::string("pk file: cannot stat ") + path; return ZKPOR_E_ARG; }
    void* m = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { ctx->err = std::string("pk file: cannot map ") + path; return ZKPOR_E_ARG; }
    (void)madvise(m, (size_t)st.st_size, MADV_SEQUENTIAL);
    int32_t rc = fn((const uint8_t*)m, (size_t)st.st_size);
    munmap(m, (size_t)st.st_size);
    return rc;
}

int32_t zkpor_pk_load_gnark_shard(zkpor_pk* pk, const char* path, size_t n_public, const uint32_t* committed_idx, size_t n_committed,
                                  size_t wire_lo, size_t wire_hi, size_t z_lo, size_t z_hi, int z_order, zkpor_pk_layout_t* info) try {
    ZK_ENTER(pk ? zk_pk_ctx(pk)->device : -1);
    if (!pk || !path) return ZKPOR_E_ARG;
    return with_mapped_file(zk_pk_ctx(pk), path, [&](const uint8_t* d, size_t n) {
        return zkpor_pk_load_gnark_shard_mem(pk, d, n, n_public, committed_idx, n_committed, wire_lo, wire_hi, z_lo, z_hi, z_order, info);
    });
} ZK_ABI_CATCH_IN((pk ? zk_pk_ctx(pk) : nullptr))

int32_t zkpor_pk_load_gnark(zkpor_pk* pk, const char* path, size_t n_public, const uint32_t* committed_idx, size_t n_committed,
                            int z_order, zkpor_pk_layout_t* info) try {
    ZK_ENTER(pk ? zk_pk_ctx(pk)->device : -1);
    if (!pk || !path) return ZKPOR_E_ARG;
    return with_mapped_file(zk_pk_ctx(pk), path, [&](const uint8_t* d, size_t n) {
        return zkpor_pk_load_gnark_mem(pk, d, n, n_public, committed_idx, n_committed, z_order, info);
    });
} ZK_ABI_CATCH_IN((pk ? zk_pk_ctx(pk) : nullptr))

}  // extern "C"
