// BatchCreateUserCircuit.Define restated against host/circuit/frontend.hpp — the circuit the reference compiles at key generation
// (src/keygen/main.go:30: frontend.Compile over circuit.NewBatchCreateUserCircuit(T, 500, U)) and solves inside every groth16.Prove
// (src/prover/prover/prover.go:269).  Statement by statement after
//   circuit/batch_create_user_circuit.go:98-323   Define
//   circuit/utils.go:12-225                        verifyMerkleProof, accountIdToMerkleHelper, computeUserAssetsCommitment, fillCexAssetCommitment,
//                                                  convertTierRatiosToVariables, generateRapidArithmeticForCollateral, getAndCheckTierRatiosQueryResults,
//                                                  checkAndGetIntegerDivisionRes, construct*TierRatiosLookupTable
// with the reference's variable names.  The input wires are the assignment vector of host/witness_assign.hpp (public first, declaration
// order: circuit/types.go:14-62, batch_create_user_circuit.go:11-20), so a witness row decoded and assigned by this repo's host code is
// what the compiled program is solved with.  Why it exists: the image has no Go, so gnark's compiled system of the real circuit cannot be
// exported here; this gives the device executor a program with the real circuit's gadgets, dependency chains and widths at the real size.
#pragma once
#include "frontend.hpp"

namespace zkpor_circuit {

struct CircuitShape {
    u32 userAssetCounts = 50;   // T
    u32 allAssetCounts = 500;   // utils.AssetCounts
    u32 batchCounts = 1380;     // users per batch
    static const u32 TierCount = 12, AccountTreeDepth = 28;
    u64 per_cex() const { return 6 + 3 * 3 * (u64)TierCount; }                                  // 114
    u64 per_user() const { return 7 * (u64)userAssetCounts + 5 * (u64)allAssetCounts + 2 + AccountTreeDepth; }
    u64 n_public() const { return 2; }                                                         // ONE, BatchCommitment
    u64 n_secret() const { return 5 + per_cex() * allAssetCounts + per_user() * batchCounts; }
    // wire ids of the inputs
    u64 cex_base(u32 i) const { return 7 + per_cex() * i; }
    u64 user_base(u32 u) const { return 7 + per_cex() * allAssetCounts + per_user() * u; }
};

namespace bcu {
struct TierRatio { LE BoundaryValue, Ratio, PrecomputedValue; };
struct CexAssetInfo {
    LE TotalEquity, TotalDebt, BasePrice, LoanCollateral, MarginCollateral, PortfolioMarginCollateral;
    std::vector<TierRatio> LoanRatios, MarginRatios, PortfolioMarginRatios;
};
struct UserAssetInfo { LE AssetIndex, LoanCollateralIndex, LoanCollateralFlag, MarginCollateralIndex, MarginCollateralFlag, PortfolioMarginCollateralIndex, PortfolioMarginCollateralFlag; };
struct UserAssetMeta { LE Equity, Debt, LoanCollateral, MarginCollateral, PortfolioMarginCollateral; };

struct Ctx {
    Builder& api;
    FrH Uint64MaxValueFr, Uint64MaxValueFrSquare, Uint8MaxValueFr, Uint126MaxValueFr, Uint134MaxValueFr, MaxTierBoundaryValueFr, PercentageMultiplierFr;
    explicit Ctx(Builder& b) : api(b) {
        Uint64MaxValueFr = Builder::fr_pow2(64); Uint64MaxValueFrSquare = Builder::fr_pow2(128); Uint8MaxValueFr = Builder::fr_pow2(8);
        Uint126MaxValueFr = Builder::fr_pow2(126); Uint134MaxValueFr = Builder::fr_pow2(134); MaxTierBoundaryValueFr = Builder::fr_pow2(118);
        PercentageMultiplierFr = FrH::from_u64(100);
    }
};

// circuit/utils.go:166-177
inline LE checkAndGetIntegerDivisionRes(Ctx& c, const LE& dividend) {
    Builder& api = c.api;
    std::vector<FrH> ov;
    if (api.witness_mode()) {
        U256 q, rem;
        U256::divmod(U256::of(api.eval(dividend)), U256{{100, 0, 0, 0}}, &q, &rem);
        ov = {q.fr(), rem.fr()};
    }
    std::vector<LE> quotientRes = api.hint("IntegerDivision", {dividend, api.constant(c.PercentageMultiplierFr)}, 2, &ov);
    api.range_check(quotientRes[0], 128);
    api.range_check(quotientRes[1], 8);
    // remainder must satisfy 0 <= r < PercentageMultiplier
    api.assert_eq(api.cmp_nop(quotientRes[1], api.constant(c.PercentageMultiplierFr), 8), api.constant(FrH::neg(FrH::one())), "remainder < 100");
    api.assert_eq(api.add(api.scale(quotientRes[0], c.PercentageMultiplierFr), quotientRes[1]), dividend, "q * 100 + r = dividend");
    return quotientRes[0];
}

// circuit/utils.go:83-101
inline void generateRapidArithmeticForCollateral(Ctx& c, std::vector<TierRatio>& tierRatios) {
    Builder& api = c.api;
    tierRatios[0].PrecomputedValue = checkAndGetIntegerDivisionRes(c, api.mul(tierRatios[0].BoundaryValue, tierRatios[0].Ratio));
    api.assert_le_nop(tierRatios[0].Ratio, api.constant(c.PercentageMultiplierFr), 8);
    api.assert_le_nop(tierRatios[0].BoundaryValue, api.constant(c.MaxTierBoundaryValueFr), 128);
    for (size_t i = 1; i < tierRatios.size(); ++i) {
        api.assert_le_nop(tierRatios[i - 1].BoundaryValue, tierRatios[i].BoundaryValue, 128);
        api.assert_le_nop(tierRatios[i].Ratio, api.constant(c.PercentageMultiplierFr), 8);
        api.assert_le_nop(tierRatios[i].BoundaryValue, api.constant(c.MaxTierBoundaryValueFr), 128);
        const LE diffBoundary = api.sub(tierRatios[i].BoundaryValue, tierRatios[i - 1].BoundaryValue);
        const LE current = checkAndGetIntegerDivisionRes(c, api.mul(diffBoundary, tierRatios[i].Ratio));
        tierRatios[i].PrecomputedValue = api.add(tierRatios[i - 1].PrecomputedValue, current);
    }
    for (size_t i = 0; i < tierRatios.size(); ++i) {
        api.range_check(tierRatios[i].PrecomputedValue, 128);
        api.range_check(tierRatios[i].Ratio, 8);
        api.range_check(tierRatios[i].BoundaryValue, 128);
    }
}

// circuit/utils.go:63-81
inline void convertTierRatiosToVariables(Ctx& c, const std::vector<TierRatio>& ratios, LE* res) {
    Builder& api = c.api;
    for (size_t i = 0; i < ratios.size(); i += 2) {
        const LE v = api.add(ratios[i].Ratio, api.scale(ratios[i].BoundaryValue, c.Uint8MaxValueFr));
        const LE v1 = api.add(api.scale(ratios[i + 1].Ratio, c.Uint126MaxValueFr), api.scale(ratios[i + 1].BoundaryValue, c.Uint134MaxValueFr));
        res[i / 2] = api.add(v, v1);
    }
}
inline int getVariableCountOfCexAsset(const CexAssetInfo& a) { return 2 + (int)(a.LoanRatios.size() / 2 + a.MarginRatios.size() / 2 + a.PortfolioMarginRatios.size() / 2); }
inline void fillCexAssetCommitment(Ctx& c, const CexAssetInfo& asset, int currentIndex, std::vector<LE>& commitments) {
    Builder& api = c.api;
    const int counts = getVariableCountOfCexAsset(asset);
    commitments[(size_t)currentIndex * counts] = api.add(api.scale(asset.TotalEquity, c.Uint64MaxValueFrSquare), api.scale(asset.TotalDebt, c.Uint64MaxValueFr), asset.BasePrice);
    commitments[(size_t)currentIndex * counts + 1] = api.add(api.scale(asset.LoanCollateral, c.Uint64MaxValueFrSquare), api.scale(asset.MarginCollateral, c.Uint64MaxValueFr), asset.PortfolioMarginCollateral);
    convertTierRatiosToVariables(c, asset.LoanRatios, &commitments[(size_t)currentIndex * counts + 2]);
    convertTierRatiosToVariables(c, asset.MarginRatios, &commitments[(size_t)currentIndex * counts + 2 + asset.LoanRatios.size() / 2]);
    convertTierRatiosToVariables(c, asset.PortfolioMarginRatios, &commitments[(size_t)currentIndex * counts + 2 + asset.LoanRatios.size() / 2 + asset.MarginRatios.size() / 2]);
}

// circuit/utils.go:179-225 (the three constructors differ in the list they read)
inline int constructTierRatiosLookupTable(Ctx& c, const std::vector<CexAssetInfo>& cexAssetInfo, int which) {
    Builder& api = c.api;
    const int t = api.new_table();
    for (auto& a : cexAssetInfo) {
        for (int k = 0; k < 3; ++k) api.table_insert(t, api.constant(0));   // dummy tier ratio
        const std::vector<TierRatio>& l = which == 0 ? a.LoanRatios : (which == 1 ? a.MarginRatios : a.PortfolioMarginRatios);
        for (auto& r : l) { api.table_insert(t, r.BoundaryValue); api.table_insert(t, r.Ratio); api.table_insert(t, r.PrecomputedValue); }
    }
    return t;
}

// circuit/utils.go:112-164
inline LE getAndCheckTierRatiosQueryResults(Ctx& c, int tierRatiosTable, const LE& assetIndex, const LE& userCollateral, LE collateralIndex,
                                            const LE& collateralFlag, const LE& assetPrice, u32 collateralTierRatiosLen, u32 maxCollateralTierIndex) {
    Builder& api = c.api;
    // Constrain collateralIndex to [0, maxCollateralTierIndex] to prevent cross-asset lookup table access.
    api.assert_le_nop(collateralIndex, api.constant(maxCollateralTierIndex), 4);
    api.assert_bool(collateralFlag);
    // when collateralFlag == 1, collateralIndex must point to the last tier.
    api.assert_eq(api.mul(collateralFlag, api.sub(collateralIndex, api.constant(maxCollateralTierIndex))), api.constant(0), "flag => last tier");
    const int numOfTierRatioFields = 3;
    std::vector<LE> queries(6);
    const LE gap = api.mul(assetIndex, collateralTierRatiosLen);
    const LE collateralValue = api.mul(userCollateral, assetPrice);
    // When cv == 0, collateralIndex must be 0 (the dummy tier slot).
    api.assert_eq(api.mul(api.is_zero(collateralValue), collateralIndex), api.constant(0), "cv == 0 => index 0");
    for (int i = 0; i < 2; ++i) {
        const LE startPosition = api.mul(collateralIndex, 3);
        queries[i * numOfTierRatioFields + 0] = api.add(startPosition, gap);
        queries[i * numOfTierRatioFields + 1] = api.add(startPosition, api.add(gap, 1));
        queries[i * numOfTierRatioFields + 2] = api.add(startPosition, api.add(gap, 2));
        collateralIndex = api.add(collateralIndex, 1);
    }
    const std::vector<LE> results = api.table_lookup(tierRatiosTable, queries);
    // Lower bound: when cv != 0, cv must be strictly greater than results[0] (the lower tier boundary).
    const LE lowerDiff = api.sub(collateralValue, api.add(results[0], 1));
    api.range_check(api.select(api.is_zero(collateralValue), api.constant(0), lowerDiff), 128);
    // Upper bound (merged check for both flag values)
    const LE leqDiff = api.sub(results[3], collateralValue);
    const LE gtDiff = api.sub(collateralValue, api.add(results[3], 1));
    api.range_check(api.select(collateralFlag, gtDiff, leqDiff), 128);
    // when flag=1, collateralValue must still be <= MaxTierBoundaryValue.
    const LE maxBoundaryDiff = api.sub(api.constant(c.MaxTierBoundaryValueFr), collateralValue);
    api.range_check(api.select(collateralFlag, maxBoundaryDiff, api.constant(0)), 128);
    // diffValue = (collateralValue - lower boundary value) * ratio
    const LE diffValue = api.mul(api.sub(collateralValue, results[0]), results[4]);
    const LE quotient = checkAndGetIntegerDivisionRes(c, diffValue);
    return api.select(api.is_zero(collateralFlag), api.add(results[2], quotient), results[5]);
}

// circuit/utils.go:12-21
inline void verifyMerkleProof(Ctx& c, const LE& merkleRoot, LE node, const std::vector<LE>& proofSet, const std::vector<LE>& helper) {
    Builder& api = c.api;
    for (size_t i = 0; i < proofSet.size(); ++i) {
        api.assert_bool(helper[i]);
        const LE d1 = api.select(helper[i], proofSet[i], node);
        const LE d2 = api.select(helper[i], node, proofSet[i]);
        node = api.poseidon({d1, d2});
    }
    api.assert_eq(merkleRoot, node, "merkle root");
}

// circuit/utils.go:28-49
inline LE computeUserAssetsCommitment(Ctx& c, const std::vector<LE>& flattenAssets) {
    Builder& api = c.api;
    const size_t nEles = (flattenAssets.size() + 2) / 3, quotientEles = flattenAssets.size() / 3, remainderEles = flattenAssets.size() % 3;
    std::vector<LE> tmpUserAssets(nEles);
    for (size_t i = 0; i < quotientEles; ++i)
        tmpUserAssets[i] = api.add(api.scale(flattenAssets[3 * i], c.Uint64MaxValueFrSquare), api.scale(flattenAssets[3 * i + 1], c.Uint64MaxValueFr), flattenAssets[3 * i + 2]);
    LE lastEle;
    for (size_t i = 0; i < remainderEles; ++i) lastEle = api.add(api.scale(lastEle, c.Uint64MaxValueFr), flattenAssets[3 * quotientEles + i]);
    for (size_t i = remainderEles; i < 3; ++i) lastEle = api.scale(lastEle, c.Uint64MaxValueFr);
    if (remainderEles > 0) tmpUserAssets[quotientEles] = lastEle;
    return api.poseidon(tmpUserAssets);
}
}  // namespace bcu

// circuit/batch_create_user_circuit.go:98-323.  Ends with the deferred commitments (what gnark's compile appends after Define returns).
inline void DefineBatchCreateUser(Builder& api, const CircuitShape& S) {
    using namespace bcu;
    Ctx c(api);
    const u32 T = S.userAssetCounts, nCex = S.allAssetCounts, U = S.batchCounts, TC = CircuitShape::TierCount;
    {   // measured on small batches: ~13.1 k constraints per CEX asset; per user ~560 + 526 per asset slot + 26 per CEX asset
        const u64 cons = 13200ull * nCex + (u64)U * (600 + 530ull * T + 27ull * nCex);
        api.reserve(cons + cons / 16, cons * 10, cons + cons / 8, cons * 2);
    }
    // the circuit struct over the input wires
    const LE BatchCommitment = api.input(1), AccountTreeRoot = api.input(2), BeforeCEXAssetsCommitment = api.input(3), AfterCEXAssetsCommitment = api.input(4),
             MinAccountIndex = api.input(5), MaxAccountIndex = api.input(6);
    std::vector<CexAssetInfo> BeforeCexAssets(nCex);
    for (u32 i = 0; i < nCex; ++i) {
        u64 w = S.cex_base(i);
        CexAssetInfo& a = BeforeCexAssets[i];
        a.TotalEquity = api.input(w++); a.TotalDebt = api.input(w++); a.BasePrice = api.input(w++);
        a.LoanCollateral = api.input(w++); a.MarginCollateral = api.input(w++); a.PortfolioMarginCollateral = api.input(w++);
        for (auto* l : {&a.LoanRatios, &a.MarginRatios, &a.PortfolioMarginRatios}) {
            l->resize(TC);
            for (u32 j = 0; j < TC; ++j) { (*l)[j].BoundaryValue = api.input(w++); (*l)[j].Ratio = api.input(w++); (*l)[j].PrecomputedValue = api.input(w++); }
        }
    }
    struct Op { std::vector<UserAssetInfo> Assets; u64 meta_base; LE AccountIndex, AccountIdHash; std::vector<LE> AccountProof; };
    auto user_op = [&](u32 u) {
        Op op;
        u64 w = S.user_base(u);
        op.Assets.resize(T);
        for (u32 j = 0; j < T; ++j) {
            UserAssetInfo& a = op.Assets[j];
            a.AssetIndex = api.input(w++); a.LoanCollateralIndex = api.input(w++); a.LoanCollateralFlag = api.input(w++);
            a.MarginCollateralIndex = api.input(w++); a.MarginCollateralFlag = api.input(w++);
            a.PortfolioMarginCollateralIndex = api.input(w++); a.PortfolioMarginCollateralFlag = api.input(w++);
        }
        op.meta_base = w; w += 5 * (u64)nCex;
        op.AccountIndex = api.input(w++); op.AccountIdHash = api.input(w++);
        for (u32 j = 0; j < CircuitShape::AccountTreeDepth; ++j) op.AccountProof.push_back(api.input(w++));
        return op;
    };
    auto meta = [&](const Op& op, u32 j, int field) { return api.input(op.meta_base + 5 * (u64)j + field); };   // AssetsForUpdateCex[j].{Equity, Debt, Loan, Margin, PortfolioMargin}

    // verify MinAccountIndex and MaxAccountIndex match the first and last op
    { const Op first = user_op(0), last = user_op(U - 1);
      api.assert_eq(MinAccountIndex, first.AccountIndex, "min index");
      api.assert_eq(MaxAccountIndex, last.AccountIndex, "max index"); }
    // verify whether BatchCommitment is computed correctly
    const LE actualBatchCommitment = api.poseidon({AccountTreeRoot, BeforeCEXAssetsCommitment, AfterCEXAssetsCommitment, MinAccountIndex, MaxAccountIndex});
    api.assert_eq(BatchCommitment, actualBatchCommitment, "batch commitment");
    const int countOfCexAsset = getVariableCountOfCexAsset(BeforeCexAssets[0]);
    std::vector<LE> cexAssets((size_t)nCex * countOfCexAsset);
    std::vector<CexAssetInfo> afterCexAssets(nCex);

    // verify whether beforeCexAssetsCommitment is computed correctly
    const int assetPriceTable = api.new_table();
    for (u32 i = 0; i < nCex; ++i) {
        CexAssetInfo& a = BeforeCexAssets[i];
        api.range_check(a.TotalEquity, 64); api.range_check(a.TotalDebt, 64); api.range_check(a.BasePrice, 64);
        api.range_check(a.LoanCollateral, 64); api.range_check(a.MarginCollateral, 64); api.range_check(a.PortfolioMarginCollateral, 64);
        fillCexAssetCommitment(c, a, (int)i, cexAssets);
        generateRapidArithmeticForCollateral(c, a.LoanRatios);
        generateRapidArithmeticForCollateral(c, a.MarginRatios);
        generateRapidArithmeticForCollateral(c, a.PortfolioMarginRatios);
        afterCexAssets[i] = a;
        api.table_insert(assetPriceTable, a.BasePrice);
    }
    const LE actualCexAssetsCommitment = api.poseidon(cexAssets, /*async=*/true);
    api.assert_eq(BeforeCEXAssetsCommitment, actualCexAssetsCommitment, "before CEX commitment");
    cexAssets.clear(); cexAssets.shrink_to_fit();

    const int loanTierRatiosTable = constructTierRatiosLookupTable(c, BeforeCexAssets, 0);
    const int marginTierRatiosTable = constructTierRatiosLookupTable(c, BeforeCexAssets, 1);
    const int portfolioMarginTierRatiosTable = constructTierRatiosLookupTable(c, BeforeCexAssets, 2);
    std::vector<LE> userAssetIdHashes(U + 1);
    std::vector<std::vector<LE>> userAssetsResults(U), userAssetsQueries(U);

    LE prevAccountIndex;
    for (u32 i = 0; i < U; ++i) {
        const Op op = user_op(i);
        // verify AccountIndex increments by 1 across the batch
        if (i > 0) api.assert_eq(op.AccountIndex, api.add(prevAccountIndex, 1), "consecutive account index");
        prevAccountIndex = op.AccountIndex;
        const std::vector<LE> accountIndexHelper = api.to_binary(op.AccountIndex, CircuitShape::AccountTreeDepth);   // accountIdToMerkleHelper
        LE totalUserEquity, totalUserDebt, totalUserCollateralRealValue;
        const std::vector<UserAssetInfo>& userAssets = op.Assets;
        // construct lookup table for user assets
        const int userAssetsLookupTable = api.new_table();
        for (u32 j = 0; j < nCex; ++j) for (int f = 0; f < 5; ++f) api.table_insert(userAssetsLookupTable, meta(op, j, f));
        // all the user assetIndexes are unique: increasing
        for (u32 j = 0; j + 1 < T; ++j) {
            api.range_check(userAssets[j].AssetIndex, 16);
            const LE cr = api.cmp_nop(userAssets[j + 1].AssetIndex, userAssets[j].AssetIndex, 16);
            api.assert_eq(cr, api.constant(1), "asset indexes increase");
        }
        api.range_check(userAssets[T - 1].AssetIndex, 16);
        // one Variable can store 15 assetIds
        std::vector<LE> assetIdsToVariables((T + 14) / 15);
        for (size_t j = 0; j < assetIdsToVariables.size(); ++j) {
            LE v;
            for (size_t p = j * 15; p < (j + 1) * 15 && p < T; ++p) v = api.add(v, api.scale(userAssets[p].AssetIndex, Builder::fr_pow2(16 * (int)(p % 15))));
            assetIdsToVariables[j] = v;
        }
        userAssetIdHashes[i] = api.poseidon(assetIdsToVariables);
        // construct query to get user assets
        userAssetsQueries[i].resize((size_t)T * 5);
        std::vector<LE> assetPriceQueries(T);
        const int numOfAssetsFields = 6;
        for (u32 j = 0; j < T; ++j) {
            const LE p = api.mul(userAssets[j].AssetIndex, 5);
            for (int k = 0; k < 5; ++k) userAssetsQueries[i][(size_t)j * 5 + k] = api.add(p, (u64)k);
            assetPriceQueries[j] = userAssets[j].AssetIndex;
        }
        userAssetsResults[i] = api.table_lookup(userAssetsLookupTable, userAssetsQueries[i]);
        const std::vector<LE> assetPriceResponses = api.table_lookup(assetPriceTable, assetPriceQueries);
        std::vector<LE> flattenAssetFieldsForHash((size_t)T * numOfAssetsFields);
        for (u32 j = 0; j < T; ++j) {
            const LE& userEquity = userAssetsResults[i][(size_t)j * 5];
            const LE& userDebt = userAssetsResults[i][(size_t)j * 5 + 1];
            const LE& userLoanCollateral = userAssetsResults[i][(size_t)j * 5 + 2];
            const LE& userMarginCollateral = userAssetsResults[i][(size_t)j * 5 + 3];
            const LE& userPortfolioMarginCollateral = userAssetsResults[i][(size_t)j * 5 + 4];
            api.range_check(userEquity, 64); api.range_check(userDebt, 64); api.range_check(userLoanCollateral, 64);
            api.range_check(userMarginCollateral, 64); api.range_check(userPortfolioMarginCollateral, 64);
            LE* f = &flattenAssetFieldsForHash[(size_t)j * numOfAssetsFields];
            f[0] = userAssets[j].AssetIndex; f[1] = userEquity; f[2] = userDebt; f[3] = userLoanCollateral; f[4] = userMarginCollateral; f[5] = userPortfolioMarginCollateral;
            const LE assetTotalCollateral = api.add(userLoanCollateral, userMarginCollateral, userPortfolioMarginCollateral);
            api.range_check(assetTotalCollateral, 64);
            api.assert_le_nop(assetTotalCollateral, userEquity, 64);
            const u32 flattenTierRatiosLength = 3 * (TC + 1);
            const LE loanRealValue = getAndCheckTierRatiosQueryResults(c, loanTierRatiosTable, userAssets[j].AssetIndex, userLoanCollateral,
                userAssets[j].LoanCollateralIndex, userAssets[j].LoanCollateralFlag, assetPriceResponses[j], flattenTierRatiosLength, TC - 1);
            const LE marginRealValue = getAndCheckTierRatiosQueryResults(c, marginTierRatiosTable, userAssets[j].AssetIndex, userMarginCollateral,
                userAssets[j].MarginCollateralIndex, userAssets[j].MarginCollateralFlag, assetPriceResponses[j], flattenTierRatiosLength, TC - 1);
            const LE portfolioMarginRealValue = getAndCheckTierRatiosQueryResults(c, portfolioMarginTierRatiosTable, userAssets[j].AssetIndex, userPortfolioMarginCollateral,
                userAssets[j].PortfolioMarginCollateralIndex, userAssets[j].PortfolioMarginCollateralFlag, assetPriceResponses[j], flattenTierRatiosLength, TC - 1);
            totalUserCollateralRealValue = api.add(api.add(totalUserCollateralRealValue, loanRealValue), api.add(marginRealValue, portfolioMarginRealValue));
            totalUserEquity = api.add(totalUserEquity, api.mul(userEquity, assetPriceResponses[j]));
            totalUserDebt = api.add(totalUserDebt, api.mul(userDebt, assetPriceResponses[j]));
        }
        for (u32 j = 0; j < nCex; ++j) {
            api.add_assign(afterCexAssets[j].TotalEquity, meta(op, j, 0));
            api.add_assign(afterCexAssets[j].TotalDebt, meta(op, j, 1));
            api.add_assign(afterCexAssets[j].LoanCollateral, meta(op, j, 2));
            api.add_assign(afterCexAssets[j].MarginCollateral, meta(op, j, 3));
            api.add_assign(afterCexAssets[j].PortfolioMarginCollateral, meta(op, j, 4));
        }
        // make sure user's total Debt is less or equal than total collateral
        api.range_check(totalUserDebt, 128);
        api.range_check(totalUserCollateralRealValue, 128);
        api.assert_le_nop(totalUserDebt, totalUserCollateralRealValue, 128);
        const LE userAssetsCommitment = computeUserAssetsCommitment(c, flattenAssetFieldsForHash);
        const LE accountHash = api.poseidon({op.AccountIdHash, totalUserEquity, totalUserDebt, totalUserCollateralRealValue, userAssetsCommitment});
        // verify the account hash against the final merkle tree root
        verifyMerkleProof(c, AccountTreeRoot, accountHash, op.AccountProof, accountIndexHelper);
    }

    // make sure user assets contains all non-zero assets of AssetsForUpdateCex: random linear combination
    userAssetIdHashes[U] = BatchCommitment;
    const LE randomChallenge = api.poseidon(userAssetIdHashes, /*async=*/api.beside() ? 2 : 0);     // 116 permutations in a row: everything that does not need the challenge runs beside it
    std::vector<LE> powersOfRandomChallenge(5 * (size_t)nCex);
    powersOfRandomChallenge[0] = randomChallenge;
    const int powersOfRandomChallengeLookupTable = api.new_table();
    api.table_insert(powersOfRandomChallengeLookupTable, randomChallenge);
    for (size_t i = 1; i < powersOfRandomChallenge.size(); ++i) {
        powersOfRandomChallenge[i] = api.mul(powersOfRandomChallenge[i - 1], randomChallenge);
        api.table_insert(powersOfRandomChallengeLookupTable, powersOfRandomChallenge[i]);
    }
    for (u32 i = 0; i < U; ++i) {
        const Op op = user_op(i);
        const std::vector<LE> powersOfRCResults = api.table_lookup(powersOfRandomChallengeLookupTable, userAssetsQueries[i]);
        LE sumA, sumB;
        for (size_t j = 0; j < powersOfRCResults.size(); ++j) api.add_assign(sumA, api.mul(powersOfRCResults[j], userAssetsResults[i][j]));
        for (u32 j = 0; j < nCex; ++j)
            for (int f = 0; f < 5; ++f) api.add_assign(sumB, api.mul(meta(op, j, f), powersOfRandomChallenge[5 * (size_t)j + f]));
        api.assert_eq(sumA, sumB, "random linear combination");
        userAssetsQueries[i].clear(); userAssetsResults[i].clear();
    }
    std::vector<LE> tempAfterCexAssets((size_t)nCex * countOfCexAsset);
    for (u32 j = 0; j < nCex; ++j) {
        api.range_check(afterCexAssets[j].TotalEquity, 64); api.range_check(afterCexAssets[j].TotalDebt, 64);
        api.range_check(afterCexAssets[j].LoanCollateral, 64); api.range_check(afterCexAssets[j].MarginCollateral, 64);
        api.range_check(afterCexAssets[j].PortfolioMarginCollateral, 64);
        fillCexAssetCommitment(c, afterCexAssets[j], (int)j, tempAfterCexAssets);
    }
    // verify AfterCEXAssetsCommitment is computed correctly
    const LE actualAfterCEXAssetsCommitment = api.poseidon(tempAfterCexAssets, /*async=*/true);
    api.assert_eq(actualAfterCEXAssetsCommitment, AfterCEXAssetsCommitment, "after CEX commitment");
    tempAfterCexAssets.clear();
    afterCexAssets.clear();
    api.finalize_commitments();
}

}  // namespace zkpor_circuit
