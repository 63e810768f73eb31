// A VALID synthetic witness batch for BatchCreateUserCircuit of any shape (T assets per user, nCex CEX assets, U users), made on the host:
// what the reference's witness service would hand the prover for such a batch (utils.BatchCreateUserWitness, src/utils/types.go:50-60;
// produced by Witness.Run src/witness/witness/witness.go:138-206 from the account tree and the CEX asset list).  Test / bench input for the
// compiled circuit (host/circuit/batch_create_user.hpp): random prices, tier tables (utils.CalculatePrecomputedValue + PaddingTierRatios,
// src/utils/utils.go:348-369,420-432), users with up to T non-empty assets whose collateral obeys the parser's rules (utils.go:590-630),
// the depth-28 Poseidon account tree over their leaves (AccountInfoToHash utils.go:744-750, NilAccountHash constants.go:125-127), the two
// CEX commitments and the batch commitment.  Everything is hashed with host/poseidon_host.hpp; the tests cross-check leaf, tree and
// commitments against oracle/ (the restatement pinned by the reference's fixture), so that "the circuit accepts this batch" also says the
// in-circuit gadgets agree with the native hash path.
#pragma once
#include <map>
#include "batch_create_user.hpp"
#include "../witness_assign.hpp"

namespace zkpor_circuit {
using zkpor_host::BatchCreateUserWitnessW;
using zkpor_host::Bytes;

namespace synth_detail {
typedef unsigned __int128 u128;
struct Rng { u64 s; u64 next() { u64 z = (s += 0x9e3779b97f4a7c15ULL); z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); } };
inline FrH fr_u128(u128 v) { uint64_t c[4] = {(u64)v, (u64)(v >> 64), 0, 0}; return FrH::from_canon(c); }
inline Bytes be32(const FrH& a) {
    uint64_t c[4]; a.to_canon(c);
    Bytes b(32, '\0');
    for (int i = 0; i < 32; ++i) b[31 - i] = (char)(uint8_t)(c[i >> 3] >> (8 * (i & 7)));
    return b;
}
inline FrH fr_be(const Bytes& b) { zkpor_host::FrCanon v; zkpor_host::assign_detail::from_be(b, &v); return FrH::from_canon(v.data()); }
inline u128 boundary_u128(const zkpor_host::BigIntW& b) { bool ok = true; return zkpor_host::assign_detail::u128_of(b, &ok); }
// utils.CalculateAssetValueViaTiersRatio (src/utils/utils.go:663-685)
inline u128 tier_value(u128 cv, const std::array<zkpor_host::TierRatioW, zkpor_host::kTierCount>& t) {
    for (int i = 0; i < zkpor_host::kTierCount; ++i) {
        const u128 b = boundary_u128(t[i].BoundaryValue);
        if (cv <= b) {
            u128 v = cv;
            if (i) v -= boundary_u128(t[i - 1].BoundaryValue);
            u128 res = v * t[i].Ratio / 100;
            if (i) res += boundary_u128(t[i - 1].PrecomputedValue);
            return res;
        }
    }
    return boundary_u128(t[zkpor_host::kTierCount - 1].PrecomputedValue);
}
inline FrH hash(const std::vector<FrH>& in) { return zkpor_host::PosSponge(in.data(), in.size()); }
// the 20 elements per CEX asset of the commitment (circuit fillCexAssetCommitment circuit/utils.go:72-81 == utils.ConvertAssetInfoToBytes utils.go:53-88)
inline void cex_elements(const zkpor_host::CexAssetInfoW& a, std::vector<FrH>* out) {
    const FrH p64 = Builder::fr_pow2(64), p128 = Builder::fr_pow2(128), p8 = Builder::fr_pow2(8), p126 = Builder::fr_pow2(126), p134 = Builder::fr_pow2(134);
    auto f = [](u64 x) { return FrH::from_u64(x); };
    out->push_back(FrH::add(FrH::add(FrH::mul(f(a.TotalEquity), p128), FrH::mul(f(a.TotalDebt), p64)), f(a.BasePrice)));
    out->push_back(FrH::add(FrH::add(FrH::mul(f(a.LoanCollateral), p128), FrH::mul(f(a.MarginCollateral), p64)), f(a.PortfolioMarginCollateral)));
    for (auto* l : {&a.LoanRatios, &a.MarginRatios, &a.PortfolioMarginRatios})
        for (int i = 0; i < zkpor_host::kTierCount; i += 2) {
            const FrH b0 = fr_u128(boundary_u128((*l)[i].BoundaryValue)), b1 = fr_u128(boundary_u128((*l)[i + 1].BoundaryValue));
            const FrH v = FrH::add(f((*l)[i].Ratio), FrH::mul(b0, p8));
            const FrH v1 = FrH::add(FrH::mul(f((*l)[i + 1].Ratio), p126), FrH::mul(b1, p134));
            out->push_back(FrH::add(v, v1));
        }
}
}  // namespace synth_detail

// users get account indexes first_index .. first_index + U - 1; extra_leaves more accounts (random leaf hashes) sit behind them so that
// the proofs carry non-nil siblings on several levels
inline BatchCreateUserWitnessW SynthBatchWitness(const CircuitShape& S, u64 seed, uint32_t first_index = 0, uint32_t extra_leaves = 5) {
    using namespace synth_detail;
    using namespace zkpor_host;
    Rng g{seed};
    const u32 T = S.userAssetCounts, nCex = S.allAssetCounts, U = S.batchCounts;
    if (T == 0 || T > nCex || U == 0 || nCex > 0xffff) throw std::invalid_argument("synth: bad shape");
    BatchCreateUserWitnessW w;
    w.BeforeCexAssets.resize(nCex);
    for (u32 i = 0; i < nCex; ++i) {
        CexAssetInfoW& c = w.BeforeCexAssets[i];
        c.Index = i; c.Symbol = "a" + std::to_string(i);
        c.BasePrice = 1 + g.next() % (1u << 24);
        c.TotalEquity = g.next() % ((u64)1 << 48); c.TotalDebt = g.next() % ((u64)1 << 40);
        c.LoanCollateral = g.next() % ((u64)1 << 44); c.MarginCollateral = g.next() % ((u64)1 << 44); c.PortfolioMarginCollateral = g.next() % ((u64)1 << 44);
        for (auto* l : {&c.LoanRatios, &c.MarginRatios, &c.PortfolioMarginRatios}) {
            const int real = 1 + (int)(g.next() % 11);
            u128 b = 0, pre = 0;
            for (int k = 0; k < kTierCount; ++k) {
                if (k < real) {
                    const u128 prev = b;
                    b += 1 + (u128)(g.next() % ((u64)1 << 50)) * (k == 0 ? 1 : (1 + g.next() % 4096));
                    (*l)[k].Ratio = (uint8_t)(g.next() % 101);
                    pre += (b - prev) * (*l)[k].Ratio / 100;      // utils.CalculatePrecomputedValue
                    (*l)[k].BoundaryValue = BigIntW::from_u128(b);
                } else {                                          // utils.PaddingTierRatios
                    (*l)[k].BoundaryValue = BigIntW::from_u128((u128)1 << 118);
                    (*l)[k].Ratio = 0;
                }
                (*l)[k].PrecomputedValue = BigIntW::from_u128(pre);
            }
        }
    }
    w.CreateUserOps.resize(U);
    std::vector<u128> totEq(U), totDebt(U), totCol(U);
    for (u32 u = 0; u < U; ++u) {
        CreateUserOperationW& op = w.CreateUserOps[u];
        op.AccountIndex = first_index + u;
        Bytes id(32, '\0');
        for (int k = 1; k < 32; ++k) id[k] = (char)(uint8_t)g.next();
        op.AccountIdHash = id;
        op.Assets.resize(nCex);
        for (u32 p = 0; p < nCex; ++p) op.Assets[p].Index = (uint16_t)p;
        const u32 k_u = (u32)(g.next() % (T + 1));               // 0 .. T non-empty assets (an all-empty user is a padding account)
        std::map<u32, bool> pick;
        while (pick.size() < k_u) pick[(u32)(g.next() % nCex)] = true;
        u128 eq = 0, col = 0;
        u32 debt_slot = nCex;
        for (auto& kv : pick) {
            AccountAssetW& a = op.Assets[kv.first];
            const CexAssetInfoW& c = w.BeforeCexAssets[kv.first];
            a.Equity = 1 + g.next() % ((u64)1 << 36);
            const u64 third = a.Equity / 3;
            a.Loan = (g.next() & 1) ? g.next() % (third + 1) : 0;
            a.Margin = (g.next() & 1) ? g.next() % (third + 1) : 0;
            a.PortfolioMargin = (g.next() & 3) == 0 ? g.next() % (third + 1) : 0;
            eq += (u128)a.Equity * c.BasePrice;
            col += tier_value((u128)a.Loan * c.BasePrice, c.LoanRatios) + tier_value((u128)a.Margin * c.BasePrice, c.MarginRatios) +
                   tier_value((u128)a.PortfolioMargin * c.BasePrice, c.PortfolioMarginRatios);
            debt_slot = kv.first;
        }
        u128 debt = 0;
        if (debt_slot < nCex && col > 0 && (g.next() & 1)) {     // a debt the collateral covers
            const u64 price = w.BeforeCexAssets[debt_slot].BasePrice;
            const u128 max_units = col / price;
            const u64 d = (u64)(max_units > ((u64)1 << 30) ? g.next() % ((u64)1 << 30) : (max_units ? g.next() % (u64)(max_units + 1) : 0));
            op.Assets[debt_slot].Debt = d;
            debt = (u128)d * price;
        }
        totEq[u] = eq; totDebt[u] = debt; totCol[u] = col;
    }
    // the padded asset lists the circuit hashes: run the assignment once on the unfinished witness and read the indexes back
    for (auto* b : {&w.BatchCommitment, &w.AccountTreeRoot, &w.BeforeCEXAssetsCommitment, &w.AfterCEXAssetsCommitment}) *b = Bytes(32, '\0');
    for (auto& op : w.CreateUserOps) for (auto& p : op.AccountProof) p = Bytes(32, '\0');
    AssignedWitness pre;
    std::string err;
    if (!SetBatchCreateUserCircuitWitness(w, {(int)T}, &pre, &err)) throw std::runtime_error("synth: " + err);
    // leaves
    const FrH p64 = Builder::fr_pow2(64), p128 = Builder::fr_pow2(128);
    std::vector<FrH> leaves(U);
    for (u32 u = 0; u < U; ++u) {
        const CreateUserOperationW& op = w.CreateUserOps[u];
        std::vector<FrH> packed;
        for (u32 j = 0; j < T; ++j) {
            const u64 idx = pre.values[S.user_base(u) - 1 + 7 * (u64)j][0];   // the assignment has no ONE wire: position = wire id - 1
            const AccountAssetW& a = op.Assets[idx];
            packed.push_back(FrH::add(FrH::add(FrH::mul(FrH::from_u64(idx), p128), FrH::mul(FrH::from_u64(a.Equity), p64)), FrH::from_u64(a.Debt)));
            packed.push_back(FrH::add(FrH::add(FrH::mul(FrH::from_u64(a.Loan), p128), FrH::mul(FrH::from_u64(a.Margin), p64)), FrH::from_u64(a.PortfolioMargin)));
        }
        const FrH commitment = hash(packed);
        leaves[u] = hash({fr_be(op.AccountIdHash), fr_u128(totEq[u]), fr_u128(totDebt[u]), fr_u128(totCol[u]), commitment});
    }
    // the sparse depth-28 tree: level 0 = leaves at first_index .., nil elsewhere
    const int depth = CircuitShape::AccountTreeDepth;
    std::vector<FrH> nil(depth + 1);
    nil[0] = hash(std::vector<FrH>(5, FrH::zero()));
    for (int l = 1; l <= depth; ++l) nil[l] = hash({nil[l - 1], nil[l - 1]});
    std::vector<std::map<u64, FrH>> lvl(depth + 1);
    for (u32 u = 0; u < U; ++u) lvl[0][(u64)first_index + u] = leaves[u];
    for (u32 e = 0; e < extra_leaves; ++e) { uint64_t c[4] = {g.next(), g.next(), g.next(), g.next() >> 4}; lvl[0][(u64)first_index + U + e] = FrH::from_canon(c); }
    for (int l = 0; l < depth; ++l) {
        auto at = [&](u64 pos) { auto it = lvl[l].find(pos); return it == lvl[l].end() ? nil[l] : it->second; };
        for (auto& kv : lvl[l]) { const u64 p = kv.first >> 1; if (!lvl[l + 1].count(p)) lvl[l + 1][p] = hash({at(2 * p), at(2 * p + 1)}); }
    }
    const FrH root = lvl[depth].begin()->second;
    w.AccountTreeRoot = be32(root);
    for (u32 u = 0; u < U; ++u) {
        u64 pos = (u64)first_index + u;
        for (int l = 0; l < depth; ++l) {
            auto it = lvl[l].find(pos ^ 1);
            w.CreateUserOps[u].AccountProof[l] = be32(it == lvl[l].end() ? nil[l] : it->second);
            pos >>= 1;
        }
    }
    // CEX commitments before / after, batch commitment (witness.go:159-198)
    std::vector<FrH> el;
    for (auto& c : w.BeforeCexAssets) cex_elements(c, &el);
    const FrH before = hash(el);
    std::vector<CexAssetInfoW> after = w.BeforeCexAssets;
    for (auto& op : w.CreateUserOps)
        for (u32 p = 0; p < nCex; ++p) {
            const AccountAssetW& a = op.Assets[p];
            after[p].TotalEquity += a.Equity; after[p].TotalDebt += a.Debt; after[p].LoanCollateral += a.Loan;
            after[p].MarginCollateral += a.Margin; after[p].PortfolioMarginCollateral += a.PortfolioMargin;
        }
    el.clear();
    for (auto& c : after) cex_elements(c, &el);
    const FrH after_c = hash(el);
    w.BeforeCEXAssetsCommitment = be32(before); w.AfterCEXAssetsCommitment = be32(after_c);
    w.MinAccountIndex = first_index; w.MaxAccountIndex = first_index + U - 1;
    w.BatchCommitment = be32(hash({root, before, after_c, FrH::from_u64(w.MinAccountIndex), FrH::from_u64(w.MaxAccountIndex)}));
    return w;
}

// the assignment of a witness as the solver takes it: Montgomery limbs, public first, without the ONE wire
inline std::vector<FrH> AssignMont(const BatchCreateUserWitnessW& w, u32 T) {
    zkpor_host::AssignedWitness a;
    std::string err;
    if (!zkpor_host::SetBatchCreateUserCircuitWitness(w, {(int)T}, &a, &err)) throw std::runtime_error("assign: " + err);
    std::vector<FrH> out(a.values.size());
    for (size_t i = 0; i < out.size(); ++i) out[i] = FrH::from_canon(a.values[i].data());
    return out;
}

}  // namespace zkpor_circuit
