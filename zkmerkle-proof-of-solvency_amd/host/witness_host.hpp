// Host-side mirror of the reference's witness batch loop (src/witness/witness/witness.go: Witness.Run :138-206,
// fillCreateUserOp :319-340; utils.SafeAdd src/utils/utils.go:318-324) over the C ABI of include/zkpor.h.
//
// The reference walks the batches serially because the CEX totals are running sums: per batch it hashes the whole CEX
// asset list twice (before / after, 2 x 834 Poseidon permutations on one core), fetches one Merkle proof per user and
// hashes the batch commitment.  Only the running sums are inherently serial — and they are 64-bit additions.  Here
//   1. the sums are accumulated on the host exactly as fillCreateUserOp does (same order, same overflow panic),
//      recording the CEX totals at every batch boundary;
//   2. ONE zkpor_cex_commitments call hashes all boundary states (After of batch i is Before of batch i+1, so
//      nBatches + 1 states instead of 2 nBatches hashes), ONE zkpor_tree_get_proofs call fetches every user's proof,
//      ONE zkpor_batch_commitments call hashes all batch commitments.
//   3. every batch becomes a row of the witness table in the reference's own encoding — WitnessData =
//      base64(s2(gob(utils.BatchCreateUserWitness))) (serializeWorker :215-232; host/witness_codec.hpp) — so the UNMODIFIED prover
//      service (utils.DecodeBatchWitness, src/utils/utils.go:704-742) reads what this loop writes.
// Field names follow utils.BatchCreateUserWitness (src/utils/types.go:50-60).  C++ because the build image has no Go.
#pragma once
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/zkpor.h"
#include "witness_codec.hpp"

namespace zkpor_host {

typedef std::array<uint8_t, 32> WHash32;

struct CreateUserOperation {             // types.go:43-48 (Assets stay with the caller: the op refers to them)
    uint32_t AccountIndex = 0;
    WHash32 AccountIdHash{};
    const zkpor_asset_t* Assets = nullptr;  // sorted by Index, as utils.AccountInfo holds them
    size_t nAssets = 0;
};
struct BatchCreateUserWitness {          // types.go:50-60, the parts computed by Witness.Run
    WHash32 BatchCommitment{}, AccountTreeRoot{}, BeforeCEXAssetsCommitment{}, AfterCEXAssetsCommitment{};
    uint32_t MinAccountIndex = 0, MaxAccountIndex = 0;
    std::vector<zkpor_cex_totals_t> BeforeCexAssets;   // the running totals entering the batch (prices / tiers are constant)
    std::vector<WHash32> AccountProofs;                // userOpsPerBatch x depth, leaf-level sibling first
    size_t firstOp = 0;                                // index of the batch's first CreateUserOperation in the op list
};

inline uint64_t SafeAdd(uint64_t a, uint64_t b) {      // utils.go:318-324
    uint64_t c = a + b;
    if (c < a) throw std::overflow_error("overflow for balance");
    return c;
}

class Witness {
public:
    Witness(zkpor_ctx* ctx, zkpor_tree* accountTree, int treeDepth, std::vector<zkpor_cex_asset_const_t> cexAssetConsts,
            std::vector<zkpor_cex_totals_t> cexTotals)
        : ctx_(ctx), tree_(accountTree), depth_(treeDepth), consts_(std::move(cexAssetConsts)), totals_(std::move(cexTotals)) {
        if (consts_.size() != totals_.size() || consts_.empty()) throw std::invalid_argument("cex asset table / totals size mismatch");
    }
    // all batches of one tier: ops.size() must be a multiple of userOpsPerBatch (the reference pads the account list first,
    // utils.PaddingAccounts).  Throws std::overflow_error where the reference panics, std::runtime_error on a library error.
    std::vector<BatchCreateUserWitness> Run(const std::vector<CreateUserOperation>& ops, size_t userOpsPerBatch) {
        if (userOpsPerBatch == 0 || ops.size() % userOpsPerBatch) throw std::invalid_argument("ops is not a whole number of batches");
        const size_t nBatches = ops.size() / userOpsPerBatch, nA = consts_.size();
        std::vector<BatchCreateUserWitness> out(nBatches);
        if (nBatches == 0) return out;
        // 1. running totals (fillCreateUserOp :330-336), boundary states recorded
        std::vector<zkpor_cex_totals_t> states((nBatches + 1) * nA);
        std::copy(totals_.begin(), totals_.end(), states.begin());
        std::vector<uint32_t> keys(ops.size());
        for (size_t b = 0; b < nBatches; ++b) {
            out[b].BeforeCexAssets = totals_;
            out[b].firstOp = b * userOpsPerBatch;
            for (size_t j = b * userOpsPerBatch; j < (b + 1) * userOpsPerBatch; ++j) {
                const CreateUserOperation& op = ops[j];
                keys[j] = op.AccountIndex;
                for (size_t p = 0; p < op.nAssets; ++p) {
                    const zkpor_asset_t& a = op.Assets[p];
                    if (a.index >= nA) throw std::out_of_range("asset index outside the CEX asset table");
                    zkpor_cex_totals_t& t = totals_[a.index];
                    t.total_equity = SafeAdd(t.total_equity, a.equity);
                    t.total_debt = SafeAdd(t.total_debt, a.debt);
                    t.loan_collateral = SafeAdd(t.loan_collateral, a.loan);
                    t.margin_collateral = SafeAdd(t.margin_collateral, a.margin);
                    t.portfolio_margin_collateral = SafeAdd(t.portfolio_margin_collateral, a.portfolio_margin);
                }
            }
            std::copy(totals_.begin(), totals_.end(), states.begin() + (b + 1) * nA);
            out[b].MinAccountIndex = ops[b * userOpsPerBatch].AccountIndex;                 // witness.go:175-176
            out[b].MaxAccountIndex = ops[(b + 1) * userOpsPerBatch - 1].AccountIndex;
        }
        // 2. every boundary state's commitment in one launch
        std::vector<WHash32> com(nBatches + 1);
        ck(zkpor_cex_commitments(ctx_, consts_.data(), nA, states.data(), nBatches + 1, com[0].data()), "zkpor_cex_commitments");
        // 3. every user's Merkle proof in one launch (fillCreateUserOp :323), the root, every batch commitment
        std::vector<WHash32> proofs(ops.size() * (size_t)depth_);
        ck(zkpor_tree_get_proofs(tree_, keys.data(), keys.size(), proofs[0].data()), "zkpor_tree_get_proofs");
        WHash32 root;
        ck(zkpor_tree_root(tree_, root.data()), "zkpor_tree_root");
        std::vector<WHash32> roots(nBatches, root), before(nBatches), after(nBatches), bc(nBatches);
        std::vector<uint32_t> mn(nBatches), mx(nBatches);
        for (size_t b = 0; b < nBatches; ++b) { before[b] = com[b]; after[b] = com[b + 1]; mn[b] = out[b].MinAccountIndex; mx[b] = out[b].MaxAccountIndex; }
        ck(zkpor_batch_commitments(ctx_, roots[0].data(), before[0].data(), after[0].data(), mn.data(), mx.data(), nBatches, bc[0].data()),
           "zkpor_batch_commitments");
        for (size_t b = 0; b < nBatches; ++b) {
            out[b].AccountTreeRoot = root;
            out[b].BeforeCEXAssetsCommitment = before[b];
            out[b].AfterCEXAssetsCommitment = after[b];
            out[b].BatchCommitment = bc[b];
            out[b].AccountProofs.assign(proofs.begin() + b * userOpsPerBatch * depth_, proofs.begin() + (b + 1) * userOpsPerBatch * depth_);
        }
        return out;
    }
    const std::vector<zkpor_cex_totals_t>& CexTotals() const { return totals_; }

private:
    void ck(int32_t rc, const char* what) {
        if (rc != ZKPOR_OK) throw std::runtime_error(std::string(what) + ": " + zkpor_last_error(ctx_));
    }
    zkpor_ctx* ctx_;
    zkpor_tree* tree_;
    int depth_;
    std::vector<zkpor_cex_asset_const_t> consts_;
    std::vector<zkpor_cex_totals_t> totals_;
};

// ---- the witness table row (witness_model.go:43-48) ------------------------------------------------------------------
struct WitnessRow { int64_t Height = 0; std::string WitnessData; int Status = 0; };  // Status 0 = StatusPublished

// [TierCount]TierRatio of one list as the reference holds it: boundary and ratio from the (already padded) constant table,
// PrecomputedValue by CalculatePrecomputedValue (src/utils/utils.go:420-432): running sum of (boundary_i - boundary_{i-1}) * ratio_i / 100
inline std::array<TierRatioW, kTierCount> TierRatiosToWire(const zkpor_tier_ratio_t* t) {
    std::array<TierRatioW, kTierCount> out;
    unsigned __int128 prev = 0, pre = 0;
    for (int i = 0; i < kTierCount; ++i) {
        unsigned __int128 b = ((unsigned __int128)t[i].boundary[1] << 64) | t[i].boundary[0];
        pre += (b - prev) * t[i].ratio / 100;
        prev = b;
        out[i].BoundaryValue = BigIntW::from_u128(b);
        out[i].Ratio = t[i].ratio;
        out[i].PrecomputedValue = BigIntW::from_u128(pre);
    }
    return out;
}

// one batch of Witness::Run as utils.BatchCreateUserWitness: hashes as 32-byte strings, the CEX asset list with the totals that
// entered the batch, every operation with its stored (sparse) asset list and its 28 siblings.  symbols may be NULL.
inline BatchCreateUserWitnessW ToWire(const BatchCreateUserWitness& b, const std::vector<CreateUserOperation>& ops, size_t userOpsPerBatch,
                                      int depth, const std::vector<zkpor_cex_asset_const_t>& consts, const std::vector<std::string>* symbols = nullptr) {
    if (depth != kAccountTreeDepth) throw std::invalid_argument("the witness row holds AccountTreeDepth = 28 siblings per user");
    auto h = [](const WHash32& x) { return Bytes((const char*)x.data(), 32); };
    BatchCreateUserWitnessW w;
    w.BatchCommitment = h(b.BatchCommitment); w.AccountTreeRoot = h(b.AccountTreeRoot);
    w.BeforeCEXAssetsCommitment = h(b.BeforeCEXAssetsCommitment); w.AfterCEXAssetsCommitment = h(b.AfterCEXAssetsCommitment);
    w.MinAccountIndex = b.MinAccountIndex; w.MaxAccountIndex = b.MaxAccountIndex;
    w.BeforeCexAssets.resize(consts.size());
    for (size_t i = 0; i < consts.size(); ++i) {
        CexAssetInfoW& c = w.BeforeCexAssets[i];
        const zkpor_cex_totals_t& t = b.BeforeCexAssets[i];
        c.TotalEquity = t.total_equity; c.TotalDebt = t.total_debt; c.BasePrice = consts[i].base_price;
        if (symbols && i < symbols->size()) c.Symbol = (*symbols)[i];
        c.Index = (uint32_t)i;
        c.LoanCollateral = t.loan_collateral; c.MarginCollateral = t.margin_collateral; c.PortfolioMarginCollateral = t.portfolio_margin_collateral;
        c.LoanRatios = TierRatiosToWire(consts[i].loan); c.MarginRatios = TierRatiosToWire(consts[i].margin);
        c.PortfolioMarginRatios = TierRatiosToWire(consts[i].portfolio_margin);
    }
    w.CreateUserOps.resize(userOpsPerBatch);
    for (size_t j = 0; j < userOpsPerBatch; ++j) {
        const CreateUserOperation& op = ops[b.firstOp + j];
        CreateUserOperationW& o = w.CreateUserOps[j];
        o.Assets.resize(op.nAssets);
        for (size_t p = 0; p < op.nAssets; ++p) {
            const zkpor_asset_t& a = op.Assets[p];
            o.Assets[p] = AccountAssetW{(uint16_t)a.index, a.equity, a.debt, a.loan, a.margin, a.portfolio_margin};
        }
        o.AccountIndex = op.AccountIndex;
        o.AccountIdHash = h(op.AccountIdHash);
        for (int k = 0; k < kAccountTreeDepth; ++k) o.AccountProof[k] = h(b.AccountProofs[j * (size_t)depth + k]);
    }
    return w;
}

// the rows Witness.Run hands to WriteBatchWitnessToDB (witness.go:199-203 / :215-232): height = batch number, Status = Published
inline std::vector<WitnessRow> MakeWitnessRows(const std::vector<BatchCreateUserWitness>& batches, const std::vector<CreateUserOperation>& ops,
                                               size_t userOpsPerBatch, int depth, const std::vector<zkpor_cex_asset_const_t>& consts,
                                               int64_t firstHeight = 0, const std::vector<std::string>* symbols = nullptr) {
    std::vector<WitnessRow> rows(batches.size());
    for (size_t i = 0; i < batches.size(); ++i) {
        rows[i].Height = firstHeight + (int64_t)i;
        rows[i].WitnessData = EncodeBatchWitness(ToWire(batches[i], ops, userOpsPerBatch, depth, consts, symbols));
        rows[i].Status = 0;
    }
    return rows;
}

}  // namespace zkpor_host
