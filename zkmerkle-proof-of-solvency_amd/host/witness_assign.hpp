// Circuit witness assignment (SURVEY.md §8 a3 + a5): from a decoded utils.BatchCreateUserWitness to the vector
// frontend.NewWitness hands the solver (src/prover/prover/prover.go:257-260).
//   circuit.SetBatchCreateUserCircuitWitness   circuit/batch_create_user_circuit.go:334-436
//   calcAndSetCollateralInfo                   circuit/utils.go:227-278
//   utils.GetNonEmptyAssetsCountOfUser / IsAssetEmpty   src/utils/utils.go:111-133
// The vector is the circuit struct walked in declaration order (circuit/types.go:14-62, batch_create_user_circuit.go:11-20) with the
// public variable (BatchCommitment, tag `gnark:",public"`) first and the secret ones after it — the order gnark's schema walk
// gives frontend.NewWitness; element counts: 1 public + 5 + 114 per CEX asset + (7 T + 5 AssetCounts + 30) per user
// (4,031,406 values for zkpor50_1380, 1,263,006 for zkpor500_200: SURVEY.md §8).  Values are canonical integers mod r as 4 x u64
// little-endian limbs (what fr.Element.SetInterface makes of uint64 / []byte / *big.Int inputs before the Montgomery step).
// Host-only.  C++ because the build image has no Go; names follow the reference.
#pragma once
#include <array>
#include <cstdint>
#include <string>
#include <vector>
#include "witness_codec.hpp"

namespace zkpor_host {

typedef std::array<uint64_t, 4> FrCanon;

struct AssignedWitness {
    std::vector<FrCanon> values;  // public first, then secret
    size_t n_public = 0, n_secret = 0;
    int tier = 0;                 // T: assets per user in this batch's circuit (the key the prover loads: prover.go:254-256)
};

namespace assign_detail {
static const uint64_t kR[4] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
inline bool geq_r(const FrCanon& v) {
    for (int i = 3; i >= 0; --i) if (v[i] != kR[i]) return v[i] > kR[i];
    return true;
}
inline void sub_r(FrCanon& v) {
    unsigned __int128 borrow = 0;
    for (int i = 0; i < 4; ++i) {
        unsigned __int128 d = (unsigned __int128)v[i] - kR[i] - (uint64_t)borrow;
        v[i] = (uint64_t)d;
        borrow = (d >> 64) & 1;
    }
}
inline FrCanon from_u64(uint64_t x) { return FrCanon{x, 0, 0, 0}; }
// big-endian bytes (a hash, or a big.Int magnitude) -> value mod r; more than 32 bytes is refused (nothing in the witness is that long)
inline bool from_be(const Bytes& b, FrCanon* out) {
    if (b.size() > 32) return false;
    FrCanon v{0, 0, 0, 0};
    for (size_t i = 0; i < b.size(); ++i) {
        size_t bit = 8 * (b.size() - 1 - i);
        v[bit / 64] |= (uint64_t)(uint8_t)b[i] << (bit % 64);
    }
    while (geq_r(v)) sub_r(v);  // a 256-bit value is below 6 r
    *out = v;
    return true;
}
inline unsigned __int128 u128_of(const BigIntW& b, bool* ok) {
    unsigned __int128 v = 0;
    if (b.neg || b.mag.size() > 16) { *ok = false; return 0; }
    for (char c : b.mag) v = (v << 8) | (uint8_t)c;
    return v;
}
}  // namespace assign_detail

inline bool IsAssetEmpty(const AccountAssetW& a) { return !(a.Debt | a.Equity | a.Margin | a.PortfolioMargin | a.Loan); }  // utils.go:111-116
// utils.go:118-133: the smallest tier that holds the user's non-empty assets; 0 if none does
inline int GetNonEmptyAssetsCountOfUser(const std::vector<AccountAssetW>& assets, const std::vector<int>& assetCountsTiers) {
    int count = 0;
    for (auto& a : assets) count += !IsAssetEmpty(a);
    for (int t : assetCountsTiers) if (count <= t) return t;
    return 0;
}

// `w` as utils.DecodeBatchWitness returns it: every user's asset list DENSE (AssetCounts entries, entry p has Index p).
// assetCountsTiers ascending (utils.AssetCountsTiers, e.g. {50, 500}).  Returns false with a reason where the reference would
// panic (index out of range) or produce a witness the circuit cannot be solved with.
inline bool SetBatchCreateUserCircuitWitness(const BatchCreateUserWitnessW& w, const std::vector<int>& assetCountsTiers, AssignedWitness* out,
                                             std::string* err) {
    using namespace assign_detail;
    auto fail = [&](const char* m) { if (err) *err = m; return false; };
    if (w.CreateUserOps.empty()) return fail("no CreateUserOps (the reference indexes CreateUserOps[0])");
    const size_t nCex = w.BeforeCexAssets.size();
    for (auto& op : w.CreateUserOps) {
        if (op.Assets.size() != nCex) return fail("a user's asset list is not dense over the CEX assets (decode with expand_assets)");
        for (size_t p = 0; p < op.Assets.size(); ++p) if (op.Assets[p].Index != p) return fail("dense asset list out of order");
    }
    const int T = GetNonEmptyAssetsCountOfUser(w.CreateUserOps[0].Assets, assetCountsTiers);  // decided by the first user (:363-366)
    if (T <= 0) return fail("the first user's assets fit no tier");
    AssignedWitness a;
    a.tier = T;
    a.n_public = 1;
    a.values.reserve(1 + 5 + 114 * nCex + w.CreateUserOps.size() * (7 * (size_t)T + 5 * nCex + 30));
    auto push_be = [&](const Bytes& b) { FrCanon v; if (!from_be(b, &v)) return false; a.values.push_back(v); return true; };
    auto push_big = [&](const BigIntW& b) { if (!b.present) { a.values.push_back(from_u64(0)); return true; } if (b.neg) return false; return push_be(b.mag); };
    if (!push_be(w.BatchCommitment) || !push_be(w.AccountTreeRoot) || !push_be(w.BeforeCEXAssetsCommitment) || !push_be(w.AfterCEXAssetsCommitment))
        return fail("a commitment is longer than 32 bytes");
    a.values.push_back(from_u64(w.MinAccountIndex));
    a.values.push_back(from_u64(w.MaxAccountIndex));
    for (auto& c : w.BeforeCexAssets) {
        for (uint64_t v : {c.TotalEquity, c.TotalDebt, c.BasePrice, c.LoanCollateral, c.MarginCollateral, c.PortfolioMarginCollateral}) a.values.push_back(from_u64(v));
        for (auto* list : {&c.LoanRatios, &c.MarginRatios, &c.PortfolioMarginRatios})
            for (auto& t : *list) {
                if (!push_big(t.BoundaryValue)) return fail("tier boundary is not a small non-negative integer");
                a.values.push_back(from_u64(t.Ratio));
                if (!push_big(t.PrecomputedValue)) return fail("tier precomputed value is not a small non-negative integer");
            }
    }
    // calcAndSetCollateralInfo: first tier whose boundary is >= collateral x price, else (last tier, flag 1)
    auto claim = [&](const std::array<TierRatioW, kTierCount>& tiers, uint64_t amount, uint64_t price, uint64_t* index, uint64_t* flag) {
        unsigned __int128 v = (unsigned __int128)amount * price;
        for (int i = 0; i < kTierCount; ++i) {
            bool ok = true;
            unsigned __int128 b = u128_of(tiers[i].BoundaryValue, &ok);
            if (!ok) return false;
            if (v <= b) { *index = (uint64_t)i; *flag = 0; return true; }
        }
        *index = kTierCount - 1; *flag = 1;
        return true;
    };
    for (auto& op : w.CreateUserOps) {
        std::vector<int> existingKeys;
        for (auto& u : op.Assets) if (!IsAssetEmpty(u)) existingKeys.push_back((int)u.Index);
        const int paddingCounts = T - (int)existingKeys.size();
        if (paddingCounts < 0) return fail("a user holds more assets than the batch's tier");  // Go: index out of range on Assets[index]
        std::vector<std::array<uint64_t, 7>> infos((size_t)T);
        int currentPaddingCounts = 0, currentAssetIndex = 0, index = 0;
        for (int v : existingKeys) {
            if (currentPaddingCounts < paddingCounts) {
                for (int k = currentAssetIndex; k < v; ++k) {
                    currentPaddingCounts += 1;
                    infos[index++] = {(uint64_t)k, 0, 0, 0, 0, 0, 0};
                    if (currentPaddingCounts >= paddingCounts) break;
                }
            }
            const AccountAssetW& um = op.Assets[v];
            const CexAssetInfoW& p = w.BeforeCexAssets[v];
            std::array<uint64_t, 7> ua{(uint64_t)v, 0, 0, 0, 0, 0, 0};
            if (!claim(p.LoanRatios, um.Loan, p.BasePrice, &ua[1], &ua[2]) || !claim(p.MarginRatios, um.Margin, p.BasePrice, &ua[3], &ua[4]) ||
                !claim(p.PortfolioMarginRatios, um.PortfolioMargin, p.BasePrice, &ua[5], &ua[6]))
                return fail("tier boundary does not fit 128 bits");
            if (index >= T) return fail("asset padding overflows the tier");
            infos[index++] = ua;
            currentAssetIndex = v + 1;
        }
        for (int k = index; k < T; ++k) { infos[k] = {(uint64_t)currentAssetIndex, 0, 0, 0, 0, 0, 0}; currentAssetIndex += 1; }
        for (auto& ua : infos) for (uint64_t x : ua) a.values.push_back(from_u64(x));
        for (auto& u : op.Assets)   // AssetsForUpdateCex[j] = the j-th entry of the dense list
            for (uint64_t x : {u.Equity, u.Debt, u.Loan, u.Margin, u.PortfolioMargin}) a.values.push_back(from_u64(x));
        a.values.push_back(from_u64(op.AccountIndex));
        if (!push_be(op.AccountIdHash)) return fail("AccountIdHash is longer than 32 bytes");
        for (auto& pr : op.AccountProof) if (!push_be(pr)) return fail("a proof element is longer than 32 bytes");
    }
    a.n_secret = a.values.size() - a.n_public;
    *out = std::move(a);
    return true;
}

}  // namespace zkpor_host
