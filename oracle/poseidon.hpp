// ORACLE — TEST INFRASTRUCTURE ONLY (see bn254.hpp header).
//
// Poseidon over BN254 Fr as used by the reference through the bnb-chain gnark-crypto fork's
// ecc/bn254/fr/poseidon package (un-vendored; call sites: src/utils/constants.go:126,
// src/utils/account_tree.go:19,27, src/utils/utils.go:188-221,744-750,765,780,
// src/witness/witness/witness.go:139,193, src/verifier/main.go:71,79,93,245).
//
// What is pinned and by what:
//  * permutation parameters (x^5 S-box, R_F = 8, R_P(t), round constants, Cauchy MDS): the published
//    Grain-LFSR procedure (Poseidon reference generate_parameters_grain.sage, field=1 sbox=0 n=254) —
//    restated in grain_params() below; reproduces the circomlib/iden3 known answers
//    (tests/golden/poseidon_iden3_kats.json) for widths 2,3,5,6,7,15.
//  * hash wrapper for width 3 (two inputs): the reference's own data fixture
//    src/verifier/config/user_config.json — for every level k >= 15 its Merkle proof satisfies
//    Proof[k+1] == permute([0, Proof[k], Proof[k]])[1]  (12 exact 254-bit matches; element [0] does NOT
//    match).  So: capacity element state[0] = 0, inputs in state[1..], digest = state[1].
//  * chaining over blocks of 12, ragged last block, the 5-input leaf hash: the SAME fixture end to end — its account (350
//    assets x five u64, packed three per element = 584 elements: 48 full blocks + a block of 8; then
//    Poseidon(id, TotalEquity, TotalDebt, TotalCollateral, commitment)) hashes to a leaf whose 28 siblings reach the
//    fixture's Root with digest = state[1] and the capacity element state[0] carried between blocks; the three other
//    (digest, carry) conventions do not (tests/test_oracle_cpu.py test_reference_fixture_end_to_end_leaf_pins_the_sponge).
//    The fixture's FIELD layout is that of an earlier revision of the reference (tests/refdata.py); the hash is today's.
//    The conventions stay run-time parameters (PoseidonConv), default (1, 0) = what the data pins.
#pragma once
#include "bn254.hpp"
#include <map>
#include <mutex>

namespace orc {

static const int POSEIDON_RF = 8;
static inline int poseidon_rp(int t) {
    static const int tab[] = {56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65, 70, 60, 64, 68};
    return tab[t - 2];
}

struct PoseidonParams {
    int t, rp;
    std::vector<Fr> rc;   // (RF+RP)*t
    std::vector<Fr> mds;  // t*t row-major, out[i] = sum_j mds[i*t+j]*in[j]
};

struct Grain80 {
    uint8_t s[80];
    int head = 0;
    Grain80(int t, int rf, int rp) {
        int k = 0;
        auto put = [&](unsigned v, int w) { for (int i = w - 1; i >= 0; --i) s[k++] = (v >> i) & 1; };
        put(1, 2); put(0, 4); put(254, 12); put((unsigned)t, 12); put((unsigned)rf, 10); put((unsigned)rp, 10);
        for (int i = 0; i < 30; ++i) s[k++] = 1;
        for (int i = 0; i < 160; ++i) step();
    }
    int at(int i) const { return s[(head + i) % 80]; }
    int step() {
        int nb = at(62) ^ at(51) ^ at(38) ^ at(23) ^ at(13) ^ at(0);
        s[head] = (uint8_t)nb;
        head = (head + 1) % 80;
        return nb;
    }
    int bit() {
        for (;;) {
            int b1 = step(), b2 = step();
            if (b1) return b2;
        }
    }
    // 254 bits, MSB first, as canonical little-endian limbs
    void bits254(u64* out) {
        out[0] = out[1] = out[2] = out[3] = 0;
        for (int i = 253; i >= 0; --i)
            if (bit()) out[i / 64] |= (u64)1 << (i % 64);
    }
};

static inline PoseidonParams grain_params(int t) {
    PoseidonParams p;
    p.t = t; p.rp = poseidon_rp(t);
    Grain80 g(t, POSEIDON_RF, p.rp);
    int n = (POSEIDON_RF + p.rp) * t;
    p.rc.resize(n);
    for (int i = 0; i < n; ++i) {
        u64 v[4];
        do { g.bits254(v); } while (cmp256(v, FrTag::MOD) >= 0);  // rejection sampling
        p.rc[i] = Fr::from_canon(v);
    }
    std::vector<Fr> xy(2 * t);
    for (;;) {
        for (int i = 0; i < 2 * t; ++i) {
            u64 v[4];
            g.bits254(v);  // no rejection here: reduced mod r (F(x) in the reference script)
            while (cmp256(v, FrTag::MOD) >= 0) sub256(v, v, FrTag::MOD);
            xy[i] = Fr::from_canon(v);
        }
        bool ok = true;
        for (int i = 0; i < 2 * t && ok; ++i)
            for (int j = i + 1; j < 2 * t; ++j)
                if (xy[i] == xy[j]) { ok = false; break; }
        for (int i = 0; i < t && ok; ++i)
            for (int j = 0; j < t; ++j)
                if (Fr::add(xy[i], xy[t + j]).is_zero()) { ok = false; break; }
        if (ok) break;
    }
    p.mds.resize(t * t);
    for (int i = 0; i < t; ++i)
        for (int j = 0; j < t; ++j) p.mds[i * t + j] = Fr::inv(Fr::add(xy[i], xy[t + j]));
    return p;
}

static inline const PoseidonParams& poseidon_params(int t) {
    static std::map<int, PoseidonParams> cache;
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(t);
    if (it == cache.end()) it = cache.emplace(t, grain_params(t)).first;
    return it->second;
}

static inline Fr pow5(const Fr& x) {
    Fr x2 = Fr::sqr(x);
    return Fr::mul(Fr::sqr(x2), x);
}

// plain (un-optimised) HADES permutation: ARK -> S-box (full / state[0] only) -> MDS
static inline void poseidon_permute(Fr* st, int t) {
    const PoseidonParams& p = poseidon_params(t);
    std::vector<Fr> tmp(t);
    int k = 0;
    for (int r = 0; r < POSEIDON_RF + p.rp; ++r) {
        for (int i = 0; i < t; ++i) st[i] = Fr::add(st[i], p.rc[k++]);
        if (r < POSEIDON_RF / 2 || r >= POSEIDON_RF / 2 + p.rp)
            for (int i = 0; i < t; ++i) st[i] = pow5(st[i]);
        else
            st[0] = pow5(st[0]);
        for (int i = 0; i < t; ++i) {
            Fr acc = Fr::zero();
            for (int j = 0; j < t; ++j) acc = Fr::add(acc, Fr::mul(p.mds[i * t + j], st[j]));
            tmp[i] = acc;
        }
        for (int i = 0; i < t; ++i) st[i] = tmp[i];
    }
}

// the same permutation, recording what the in-circuit gadget allocates: per S-box the wires x^2, x^4, x^5 (three multiplications;
// round constants and the MDS layer are linear and cost no wire).  trace[(s * 3 + c)]: S-box s in round order, lanes 0..t-1 within a
// full round.  Checker of zkpor_witgen_poseidon_trace_dev (the device runs the OPTIMISED partial rounds; the S-box inputs coincide).
static inline void poseidon_permute_trace(Fr* st, int t, Fr* trace) {
    const PoseidonParams& p = poseidon_params(t);
    std::vector<Fr> tmp(t);
    int k = 0;
    size_t s = 0;
    auto sbox = [&](const Fr& x) {
        Fr x2 = Fr::sqr(x), x4 = Fr::sqr(x2), x5 = Fr::mul(x4, x);
        trace[3 * s] = x2; trace[3 * s + 1] = x4; trace[3 * s + 2] = x5;
        ++s;
        return x5;
    };
    for (int r = 0; r < POSEIDON_RF + p.rp; ++r) {
        for (int i = 0; i < t; ++i) st[i] = Fr::add(st[i], p.rc[k++]);
        if (r < POSEIDON_RF / 2 || r >= POSEIDON_RF / 2 + p.rp)
            for (int i = 0; i < t; ++i) st[i] = sbox(st[i]);
        else
            st[0] = sbox(st[0]);
        for (int i = 0; i < t; ++i) {
            Fr acc = Fr::zero();
            for (int j = 0; j < t; ++j) acc = Fr::add(acc, Fr::mul(p.mds[i * t + j], st[j]));
            tmp[i] = acc;
        }
        for (int i = 0; i < t; ++i) st[i] = tmp[i];
    }
}

struct PoseidonConv { int out_idx, carry_idx; };
static inline PoseidonConv& poseidon_conv() {
    static PoseidonConv c = {1, 0};
    return c;
}

// poseidon.Poseidon(input...) of the bnb fork: blocks of maxLength = 12, width = block+1
static inline Fr poseidon_hash(const Fr* in, size_t n) {
    assert(n >= 1);
    const PoseidonConv cv = poseidon_conv();
    Fr st[13];
    Fr cap = Fr::zero();
    size_t i = 0;
    Fr out = Fr::zero();
    while (i < n) {
        size_t blk = std::min<size_t>(12, n - i);
        st[0] = cap;
        for (size_t j = 0; j < blk; ++j) st[1 + j] = in[i + j];
        poseidon_permute(st, (int)blk + 1);
        cap = st[cv.carry_idx];
        out = st[cv.out_idx];
        i += blk;
    }
    return out;
}

// --------------------------------------------------------------------------------- account leaves
// src/utils/types.go:25-32 AccountAsset
struct AccountAsset { uint16_t index; u64 equity, debt, loan, margin, portfolio_margin; };

// src/utils/utils.go:147-186 PaddingAccountAssets: fill to the tier with the lowest unused indices
static inline std::vector<u64> padding_account_assets(const AccountAsset* assets, size_t n, int tier) {
    std::vector<u64> out((size_t)tier * 6, 0);
    int padding = tier - (int)n, cur_pad = 0, cur_idx = 0, index = 0;
    for (size_t i = 0; i < n; ++i) {
        if (cur_pad < padding) {
            for (int j = cur_idx; j < (int)assets[i].index; ++j) {
                cur_pad++;
                out[index * 6] = (u64)j;
                index++;
                if (cur_pad >= padding) break;
            }
        }
        out[index * 6 + 0] = assets[i].index;
        out[index * 6 + 1] = assets[i].equity;
        out[index * 6 + 2] = assets[i].debt;
        out[index * 6 + 3] = assets[i].loan;
        out[index * 6 + 4] = assets[i].margin;
        out[index * 6 + 5] = assets[i].portfolio_margin;
        index++;
        cur_idx = assets[i].index + 1;
    }
    for (int i = index; i < tier; ++i) { out[i * 6] = (u64)cur_idx; cur_idx++; }
    return out;
}
// a*2^128 + b*2^64 + c as an Fr (src/utils/utils.go:188-221, constants.go:30-31)
static inline Fr pack3(u64 a, u64 b, u64 c) {
    u64 v[4] = {c, b, a, 0};
    return Fr::from_canon(v);
}
// src/utils/utils.go:188-221 ComputeUserAssetsCommitment
static inline Fr user_assets_commitment(const AccountAsset* assets, size_t n, int tier) {
    std::vector<u64> flat = padding_account_assets(assets, n, tier);
    size_t ne = ((size_t)tier * 6 + 2) / 3;
    std::vector<Fr> el(ne);
    for (size_t i = 0; i < ne; ++i) {
        u64 a = 3 * i < flat.size() ? flat[3 * i] : 0;
        u64 b = 3 * i + 1 < flat.size() ? flat[3 * i + 1] : 0;
        u64 c = 3 * i + 2 < flat.size() ? flat[3 * i + 2] : 0;
        el[i] = pack3(a, b, c);
    }
    return poseidon_hash(el.data(), ne);
}
// src/utils/utils.go:744-750 AccountInfoToHash; totals are < 2^128 big-ints passed as canonical limbs
static inline Fr account_leaf_hash(const Fr& id, const Fr& equity, const Fr& debt, const Fr& collateral,
                                   const AccountAsset* assets, size_t n, int tier) {
    Fr in[5] = {id, equity, debt, collateral, user_assets_commitment(assets, n, tier)};
    return poseidon_hash(in, 5);
}

// --------------------------------------------------------------------------------- CEX asset list
// src/utils/utils.go: ConvertTierRatiosToBytes (:26-51), ConvertAssetInfoToBytes (:53-88), ComputeCexAssetsCommitment
// (:779-800) and the two per-batch commitments of Witness.Run (src/witness/witness/witness.go:159-183): every byte string
// is a big-endian big integer that hasher.Write turns into an Fr (mod r); the integers are sums of shifted fields.
struct TierRatio { u64 boundary[2]; uint8_t ratio; };  // BoundaryValue as a little-endian 128-bit integer
struct CexAssetConst { u64 base_price; TierRatio loan[12], margin[12], pm[12]; };
struct CexTotals { u64 total_equity, total_debt, loan_collateral, margin_collateral, portfolio_margin_collateral; };

static inline void be_add_shifted(uint8_t be[40], const u64 v[2], int shift) {  // be (320-bit big-endian) += v << shift
    unsigned carry = 0;
    for (int bit_byte = 0; bit_byte < 17 || carry; ++bit_byte) {
        // byte `bit_byte` of (v << (shift % 8)), placed at byte offset shift / 8
        unsigned __int128 vv = ((unsigned __int128)v[1] << 64) | v[0];
        unsigned byte = 0;
        if (bit_byte < 17) {
            int sh = 8 * bit_byte - (shift % 8);
            unsigned __int128 part = sh >= 128 ? 0 : (sh >= 0 ? (vv >> sh) : (vv << (-sh)));
            byte = (unsigned)(part & 0xff);
        }
        int pos = 39 - (shift / 8 + bit_byte);
        if (pos < 0) break;
        unsigned t = be[pos] + byte + carry;
        be[pos] = (uint8_t)t;
        carry = t >> 8;
    }
}
static inline Fr tier_pair_element(const TierRatio& lo, const TierRatio& hi) {
    uint8_t be[40] = {0};
    u64 r0[2] = {lo.ratio, 0}, r1[2] = {hi.ratio, 0};
    be_add_shifted(be, r0, 0);
    be_add_shifted(be, lo.boundary, 8);     // * Uint8MaxValueBigInt  (256)
    be_add_shifted(be, r1, 126);            // * Uint126MaxValueBigInt
    be_add_shifted(be, hi.boundary, 134);   // * Uint134MaxValueBigInt
    return Fr::from_be_bytes(be, 40);
}
static inline Fr pack3_u64(u64 a, u64 b, u64 c) {  // a * 2^128 + b * 2^64 + c
    u64 limbs[4] = {c, b, a, 0};
    return Fr::from_canon(limbs);
}
static inline Fr cex_assets_commitment(const CexAssetConst* consts, const CexTotals* totals, size_t n_assets) {
    std::vector<Fr> el;
    el.reserve(n_assets * 20);
    for (size_t a = 0; a < n_assets; ++a) {
        el.push_back(pack3_u64(totals[a].total_equity, totals[a].total_debt, consts[a].base_price));
        el.push_back(pack3_u64(totals[a].loan_collateral, totals[a].margin_collateral, totals[a].portfolio_margin_collateral));
        const TierRatio* groups[3] = {consts[a].loan, consts[a].margin, consts[a].pm};
        for (auto* g : groups)
            for (int i = 0; i < 12; i += 2) el.push_back(tier_pair_element(g[i], g[i + 1]));
    }
    return poseidon_hash(el.data(), el.size());
}

// --------------------------------------------------------------------------------- collateral valuation
// src/utils/utils.go: CalculatePrecomputedValue (:420-432), CalculateAssetValueViaTiersRatio (:663-685),
// CalculateAssetValueForCollateral (:648-661), the account totals of ParseUserDataSet (:608-615), and the tier index /
// flag choice of circuit/utils.go calcAndSetCollateralInfo (:227-278).  Integers are big.Int there; every value on this
// path is below 2^128 for inputs the parser accepts (u64 balance x u64 price, boundaries <= 2^118), so unsigned __int128
// restates it exactly; `overflow` reports the cases where it would not.
typedef unsigned __int128 u128;
static inline u128 tier_boundary(const TierRatio& t) { return ((u128)t.boundary[1] << 64) | t.boundary[0]; }
static inline void tier_precomputed(const TierRatio* tiers, int n, u128* pre) {
    u128 acc = 0, prev = 0;
    for (int i = 0; i < n; ++i) {
        u128 b = tier_boundary(tiers[i]);
        acc += (b - prev) * tiers[i].ratio / 100;
        pre[i] = acc;
        prev = b;
    }
}
// calcAndSetCollateralInfo: the first tier whose boundary is >= value, flag 0; above every boundary: last tier, flag 1
static inline void tier_index_flag(u128 value, const TierRatio* tiers, int n, int* index, int* flag) {
    for (int i = 0; i < n; ++i)
        if (value <= tier_boundary(tiers[i])) { *index = i; *flag = 0; return; }
    *index = n - 1; *flag = 1;
}
static inline u128 asset_value_via_tiers(u128 value, const TierRatio* tiers, int n) {
    if (n == 0) return 0;
    u128 pre[64];
    tier_precomputed(tiers, n, pre);
    for (int i = 0; i < n; ++i)
        if (value <= tier_boundary(tiers[i])) {
            u128 v = i ? value - tier_boundary(tiers[i - 1]) : value;
            u128 res = v * tiers[i].ratio / 100;
            return i ? res + pre[i - 1] : res;
        }
    return pre[n - 1];
}
struct AccountTotals { u128 equity, debt, collateral; bool valid; };
// totals of one account over its asset list; valid = the checks of ParseUserDataSet (:599-606, :620): every asset's
// loan + margin + portfolio margin <= equity (no u64 overflow), total collateral >= total debt
static inline AccountTotals account_totals(const AccountAsset* assets, size_t n, const CexAssetConst* cex) {
    AccountTotals t = {0, 0, 0, true};
    for (size_t i = 0; i < n; ++i) {
        const AccountAsset& a = assets[i];
        const CexAssetConst& c = cex[a.index];
        u64 s1 = a.loan + a.margin;
        u64 s2 = s1 + a.portfolio_margin;
        if (s1 < a.loan || s2 < s1 || s2 > a.equity) t.valid = false;
        t.equity += (u128)a.equity * c.base_price;
        t.debt += (u128)a.debt * c.base_price;
        t.collateral += asset_value_via_tiers((u128)a.loan * c.base_price, c.loan, 12) +
                        asset_value_via_tiers((u128)a.margin * c.base_price, c.margin, 12) +
                        asset_value_via_tiers((u128)a.portfolio_margin * c.base_price, c.pm, 12);
    }
    if (t.collateral < t.debt) t.valid = false;
    return t;
}

// --------------------------------------------------------------------------------- Merkle tree
// src/utils/merkletree/merkletree.go: nilHashes (:159-170), Build (:192-279), GetProof (:297-308),
// VerifyProof (:334-355).  levels[l] holds ceil(n/2^l) computed nodes; anything to the right is
// nil[l].  All n leaves are "dirty" (src/witness/main.go:130-199 sets every account).
struct MerkleTree {
    int depth;
    std::vector<Fr> nil;                  // nil[0..depth]
    std::vector<std::vector<Fr>> levels;  // levels[0] = leaves
    Fr root;
    const Fr& node(int level, size_t pos) const {
        return pos < levels[level].size() ? levels[level][pos] : nil[level];
    }
    std::vector<Fr> proof(uint32_t key) const {
        std::vector<Fr> p(depth);
        size_t pos = key;
        for (int l = 0; l < depth; ++l) { p[l] = node(l, pos ^ 1); pos >>= 1; }
        return p;
    }
};
static inline Fr hash2(const Fr& l, const Fr& r) {
    Fr in[2] = {l, r};
    return poseidon_hash(in, 2);
}
static inline MerkleTree merkle_build(const Fr* leaves, size_t n, int depth, const Fr& nil_leaf) {
    MerkleTree t;
    t.depth = depth;
    t.nil.resize(depth + 1);
    t.nil[0] = nil_leaf;
    for (int l = 1; l <= depth; ++l) t.nil[l] = hash2(t.nil[l - 1], t.nil[l - 1]);
    t.levels.resize(depth + 1);
    t.levels[0].assign(leaves, leaves + n);
    for (int l = 1; l <= depth; ++l) {
        size_t m = (t.levels[l - 1].size() + 1) / 2;
        t.levels[l].resize(m);
#pragma omp parallel for schedule(static) if (m >= 64)
        for (size_t i = 0; i < m; ++i) t.levels[l][i] = hash2(t.node(l - 1, 2 * i), t.node(l - 1, 2 * i + 1));
    }
    t.root = t.node(depth, 0);
    return t;
}
// The same tree with arbitrary (sparse) keys set — merkletree.go Set (:179-187), Build over the dirty positions only
// (:192-279), getNodeAt falling back to nilHashes for clean positions (:315-331).  Maps instead of flat buffers +
// bitsets: the oracle only has to agree on values.
struct SparseMerkleTree {
    int depth;
    std::vector<Fr> nil;
    std::vector<std::map<uint64_t, Fr>> levels;  // levels[0] = set leaves
    Fr root;
    SparseMerkleTree(int d, const Fr& nil_leaf) : depth(d), nil(d + 1), levels(d + 1) {
        nil[0] = nil_leaf;
        for (int l = 1; l <= d; ++l) nil[l] = hash2(nil[l - 1], nil[l - 1]);
        root = nil[d];
    }
    void set(uint32_t key, const Fr& v) { levels[0][key] = v; }
    const Fr& node(int level, uint64_t pos) const {
        auto it = levels[level].find(pos);
        return it == levels[level].end() ? nil[level] : it->second;
    }
    void build() {
        for (int l = 1; l <= depth; ++l) {
            levels[l].clear();
            for (auto& kv : levels[l - 1]) {
                uint64_t p = kv.first >> 1;
                if (levels[l].count(p)) continue;
                levels[l][p] = hash2(node(l - 1, 2 * p), node(l - 1, 2 * p + 1));
            }
        }
        root = node(depth, 0);
    }
    std::vector<Fr> proof(uint32_t key) const {
        std::vector<Fr> p(depth);
        uint64_t pos = key;
        for (int l = 0; l < depth; ++l) { p[l] = node(l, pos ^ 1); pos >>= 1; }
        return p;
    }
};
static inline bool merkle_verify(const Fr& root, uint32_t key, const std::vector<Fr>& proof, const Fr& leaf) {
    Fr node = leaf;
    for (size_t i = 0; i < proof.size(); ++i)
        node = (key >> i) & 1 ? hash2(proof[i], node) : hash2(node, proof[i]);
    return node == root;
}

}  // namespace orc
