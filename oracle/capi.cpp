// ORACLE — TEST INFRASTRUCTURE ONLY.  C entry points so tests/ and bench.py (cpu_baseline leg) can drive
// the CPU restatement through ctypes.  Layouts are the product's (gnark in-memory): Fr/Fp = 4 x u64
// little-endian limbs in Montgomery form; G1 affine = X,Y (64 B); G2 affine = X.A0,X.A1,Y.A0,Y.A1 (128 B).
#include "algos.hpp"
#include "poseidon.hpp"
#include "marshal.hpp"
#include "cpubase.hpp"
#include "quotient.hpp"
#include <memory>

using namespace orc;

static_assert(sizeof(Fr) == 32 && sizeof(G1A) == 64 && sizeof(G2A) == 128, "layout");

extern "C" {

int orc_selftest() {
    // constants of SURVEY.md §8(c)
    {
        u64 one[4] = {1, 0, 0, 0};
        (void)one;
        if (FpTag::MOD[0] * FpTag::INV != (u64)-1) return 1;
        if (FrTag::MOD[0] * FrTag::INV != (u64)-1) return 2;
    }
    Fr w = fr_root_of_unity_2_28();
    {
        // 19103219067921713944291392827692070036145651957329286315305642004821462161904
        static const u64 exp[4] = {0x9bd61b6e725b19f0ULL, 0x402d111e41112ed4ULL, 0x00e0a7eb8ef62abcULL,
                                   0x2a3c09f0a58a7e85ULL};
        u64 c[4];
        w.to_canon(c);
        if (memcmp(c, exp, 32)) return 3;
        Fr x = w;
        for (int i = 0; i < 27; ++i) x = Fr::sqr(x);
        if (x == Fr::one()) return 4;
        if (!(Fr::sqr(x) == Fr::one())) return 5;
    }
    if (!g1_on_curve(g1_gen())) return 6;
    if (!g2_on_curve(g2_gen())) return 7;
    if (!jmul(to_jac(g1_gen()), FrTag::MOD).is_inf()) return 8;
    if (!jmul(to_jac(g2_gen()), FrTag::MOD).is_inf()) return 9;
    {
        SplitMix rng(7);
        Fr a = rng.fr(), b = rng.fr();
        if (!(Fr::mul(a, Fr::inv(a)) == Fr::one())) return 10;
        Fp2 z = {Fp::from_u64(3), Fp::from_u64(5)};
        if (!(Fp2::mul(z, Fp2::inv(z)) == Fp2::one())) return 11;
        // (a+b)G == aG + bG on both groups
        G1J l = jmul_fr(to_jac(g1_gen()), Fr::add(a, b));
        G1J r = jadd(jmul_fr(to_jac(g1_gen()), a), jmul_fr(to_jac(g1_gen()), b));
        if (!(to_aff(l) == to_aff(r))) return 12;
        G2J l2 = jmul_fr(to_jac(g2_gen()), Fr::add(a, b));
        G2J r2 = jadd(jmul_fr(to_jac(g2_gen()), a), jmul_fr(to_jac(g2_gen()), b));
        if (!(to_aff(l2) == to_aff(r2))) return 13;
    }
    return 0;
}

// ---- field helpers (batch) ----
void orc_fp_mul(const Fp* a, const Fp* b, Fp* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = Fp::mul(a[i], b[i]); }
void orc_fp_add(const Fp* a, const Fp* b, Fp* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = Fp::add(a[i], b[i]); }
void orc_fp_sub(const Fp* a, const Fp* b, Fp* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = Fp::sub(a[i], b[i]); }
void orc_fp_inv(const Fp* a, Fp* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = Fp::inv(a[i]); }
void orc_fr_mul(const Fr* a, const Fr* b, Fr* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = Fr::mul(a[i], b[i]); }
void orc_fr_add(const Fr* a, const Fr* b, Fr* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = Fr::add(a[i], b[i]); }
void orc_fr_sub(const Fr* a, const Fr* b, Fr* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = Fr::sub(a[i], b[i]); }
void orc_fr_inv(const Fr* a, Fr* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = Fr::inv(a[i]); }
void orc_fr_from_canon(const u64* c, Fr* out, size_t n) {
#pragma omp parallel for schedule(static) if (n > 4096)
    for (size_t i = 0; i < n; ++i) out[i] = Fr::from_canon(c + 4 * i);
}
void orc_fr_to_canon(const Fr* a, u64* out, size_t n) { for (size_t i = 0; i < n; ++i) a[i].to_canon(out + 4 * i); }
void orc_fp_from_canon(const u64* c, Fp* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = Fp::from_canon(c + 4 * i); }
void orc_fp_to_canon(const Fp* a, u64* out, size_t n) { for (size_t i = 0; i < n; ++i) a[i].to_canon(out + 4 * i); }
// seeded uniform Fr (Montgomery form)
void orc_fr_random(u64 seed, Fr* out, size_t n) { SplitMix r(seed); for (size_t i = 0; i < n; ++i) out[i] = r.fr(); }
// sum_i a[i]*b[i]
void orc_fr_dot(const Fr* a, const Fr* b, size_t n, Fr* out) {
    int nt = 1;
#ifdef _OPENMP
    nt = omp_get_max_threads();
#endif
    (void)nt;
    Fr acc = Fr::zero();
#pragma omp parallel
    {
        Fr loc = Fr::zero();
#pragma omp for schedule(static) nowait
        for (size_t i = 0; i < n; ++i) loc = Fr::add(loc, Fr::mul(a[i], b[i]));
#pragma omp critical
        acc = Fr::add(acc, loc);
    }
    *out = acc;
}

// ---- group helpers ----
// points[i] = scalars[i] * G (fixed-base tables), affine
void orc_g1_from_scalars(const Fr* sc, size_t n, G1A* out) {
    static FixedBase<Fp> fb(g1_gen());
#pragma omp parallel for schedule(dynamic, 64)
    for (size_t i = 0; i < n; ++i) out[i] = fb.mul_aff(sc[i]);
}
void orc_g2_from_scalars(const Fr* sc, size_t n, G2A* out) {
    static FixedBase<Fp2> fb(g2_gen());
#pragma omp parallel for schedule(dynamic, 64)
    for (size_t i = 0; i < n; ++i) out[i] = fb.mul_aff(sc[i]);
}
int orc_g1_on_curve(const G1A* p, size_t n) { for (size_t i = 0; i < n; ++i) if (!g1_on_curve(p[i])) return 0; return 1; }
int orc_g2_on_curve(const G2A* p, size_t n) { for (size_t i = 0; i < n; ++i) if (!g2_on_curve(p[i])) return 0; return 1; }
void orc_g1_jac_to_affine(const G1J* p, G1A* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = to_aff(p[i]); }
void orc_g2_jac_to_affine(const G2J* p, G2A* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = to_aff(p[i]); }
// XYZZ (X,Y,ZZ,ZZZ) -> affine: x = X/ZZ, y = Y/ZZZ; ZZ == 0 is infinity
void orc_g1_xyzz_to_affine(const Fp* p, G1A* out, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        const Fp* q = p + 4 * i;
        if (q[2].is_zero()) { out[i] = {Fp::zero(), Fp::zero()}; continue; }
        out[i] = {Fp::mul(q[0], Fp::inv(q[2])), Fp::mul(q[1], Fp::inv(q[3]))};
    }
}
void orc_g1_add_affine(const G1A* a, const G1A* b, G1A* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = to_aff(jadd(to_jac(a[i]), to_jac(b[i])));
}
void orc_g2_add_affine(const G2A* a, const G2A* b, G2A* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = to_aff(jadd(to_jac(a[i]), to_jac(b[i])));
}
void orc_g1_scalar_mul(const G1A* p, const Fr* k, G1A* out, size_t n) {
    for (size_t i = 0; i < n; ++i) out[i] = to_aff(jmul_fr(to_jac(p[i]), k[i]));
}

void orc_g1_msm(const G1A* pts, const Fr* sc, size_t n, int window, G1A* out) {
    *out = to_aff(window < 0 ? msm_naive(pts, sc, n) : msm_pippenger(pts, sc, n, window));
}
void orc_g2_msm(const G2A* pts, const Fr* sc, size_t n, int window, G2A* out) {
    *out = to_aff(window < 0 ? msm_naive(pts, sc, n) : msm_pippenger(pts, sc, n, window));
}

// ---- NTT ----
// decimation: 0 = DIT (bit-reversed in, natural out), 1 = DIF (natural in, bit-reversed out)
void orc_fft(Fr* a, int log2n, int inverse, int decimation, int on_coset) {
    Domain d(log2n);
    if (inverse) fft_inverse(d, a, (Decimation)decimation, on_coset != 0);
    else fft_forward(d, a, (Decimation)decimation, on_coset != 0);
}
void orc_bit_reverse(Fr* a, int log2n) { bit_reverse(a, log2n); }
// naive O(n^2) DFT in natural order: out[k] = sum_j a[j] * (shift*w^k)^j   (independent check of the FFTs)
void orc_dft_naive(const Fr* a, int log2n, int on_coset, Fr* out) {
    Domain d(log2n);
    for (size_t k = 0; k < d.n; ++k) {
        Fr x = Fr::pow_u64(d.gen, (u64)k);
        if (on_coset) x = Fr::mul(x, d.coset);
        Fr acc = Fr::zero(), p = Fr::one();
        for (size_t j = 0; j < d.n; ++j) { acc = Fr::add(acc, Fr::mul(a[j], p)); p = Fr::mul(p, x); }
        out[k] = acc;
    }
}
void orc_compute_h(const Fr* a, const Fr* b, const Fr* c, size_t n_cons, int log2d, Fr* out) {
    Domain d(log2d);
    std::vector<Fr> h = compute_h(d, a, b, c, n_cons);
    memcpy(out, h.data(), h.size() * sizeof(Fr));
}

// ---- Poseidon / Merkle ----
void orc_poseidon_set_convention(int out_idx, int carry_idx) { poseidon_conv() = {out_idx, carry_idx}; }
int orc_poseidon_rp(int t) { return poseidon_rp(t); }
void orc_poseidon_params(int t, Fr* rc_out, Fr* mds_out) {
    const PoseidonParams& p = poseidon_params(t);
    memcpy(rc_out, p.rc.data(), p.rc.size() * sizeof(Fr));
    memcpy(mds_out, p.mds.data(), p.mds.size() * sizeof(Fr));
}
void orc_poseidon_permute(Fr* state, int t) { poseidon_permute(state, t); }
void orc_poseidon_hash(const Fr* in, size_t n, Fr* out) { *out = poseidon_hash(in, n); }
// count permutations of width t: states count x t (in place), trace in the DEVICE's slot order [(s * 3 + c) * count + i]
void orc_poseidon_permute_trace(Fr* states, int t, size_t count, Fr* trace) {
    const size_t ns = (size_t)POSEIDON_RF * t + poseidon_rp(t);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < count; ++i) {
        std::vector<Fr> tr(3 * ns);
        poseidon_permute_trace(states + i * t, t, tr.data());
        for (size_t k = 0; k < 3 * ns; ++k) trace[k * count + i] = tr[k];
    }
}
// batched 2->1 hashing: out[i] = H(in[2i], in[2i+1])
void orc_poseidon_hash2_batch(const Fr* in, size_t n_pairs, Fr* out) {
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n_pairs; ++i) out[i] = hash2(in[2 * i], in[2 * i + 1]);
}
// accounts in the product's packed layout (include/zkpor.h zkpor_account_t): see there
struct PackedAccountHdr {
    uint8_t id_be[32];
    u64 equity[2], debt[2], collateral[2];  // little-endian 128-bit
    uint32_t n_assets, asset_off;           // into the asset array
};
struct PackedAsset { u64 equity, debt, loan, margin, portfolio_margin; uint32_t index, pad; };
static_assert(sizeof(PackedAccountHdr) == 88 && sizeof(PackedAsset) == 48, "layout");

void orc_account_leaves(const PackedAccountHdr* acc, const PackedAsset* assets, size_t n, int tier, Fr* out) {
#pragma omp parallel for schedule(dynamic, 16)
    for (size_t i = 0; i < n; ++i) {
        std::vector<AccountAsset> aa(acc[i].n_assets);
        for (uint32_t j = 0; j < acc[i].n_assets; ++j) {
            const PackedAsset& p = assets[acc[i].asset_off + j];
            aa[j] = {(uint16_t)p.index, p.equity, p.debt, p.loan, p.margin, p.portfolio_margin};
        }
        Fr id = Fr::from_be_bytes(acc[i].id_be, 32);
        u64 e[4] = {acc[i].equity[0], acc[i].equity[1], 0, 0};
        u64 d[4] = {acc[i].debt[0], acc[i].debt[1], 0, 0};
        u64 c[4] = {acc[i].collateral[0], acc[i].collateral[1], 0, 0};
        out[i] = account_leaf_hash(id, Fr::from_canon(e), Fr::from_canon(d), Fr::from_canon(c), aa.data(),
                                   aa.size(), tier);
    }
}
// CEX commitments in the product's packed layout (include/zkpor.h zkpor_cex_asset_const_t / zkpor_cex_totals_t)
struct PackedTier { u64 boundary[2]; uint8_t ratio; uint8_t pad[7]; };
struct PackedCexConst { u64 base_price; PackedTier loan[12], margin[12], pm[12]; };
static_assert(sizeof(PackedCexConst) == 872 && sizeof(CexTotals) == 40, "layout");
void orc_cex_commitments(const PackedCexConst* consts, size_t n_assets, const CexTotals* totals, size_t n_states, Fr* out) {
    std::vector<CexAssetConst> c(n_assets);
    for (size_t a = 0; a < n_assets; ++a) {
        c[a].base_price = consts[a].base_price;
        for (int i = 0; i < 12; ++i) {
            c[a].loan[i] = {{consts[a].loan[i].boundary[0], consts[a].loan[i].boundary[1]}, consts[a].loan[i].ratio};
            c[a].margin[i] = {{consts[a].margin[i].boundary[0], consts[a].margin[i].boundary[1]}, consts[a].margin[i].ratio};
            c[a].pm[i] = {{consts[a].pm[i].boundary[0], consts[a].pm[i].boundary[1]}, consts[a].pm[i].ratio};
        }
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (size_t s = 0; s < n_states; ++s) out[s] = cex_assets_commitment(c.data(), totals + s * n_assets, n_assets);
}
// collateral valuation.  tiers: n x (boundary lo, boundary hi, ratio) as 3 u64 each; value / results as 2 u64 (little-endian 128 bit)
void orc_tier_query(const u64* tiers3, int n, const u64 value[2], int* index, int* flag, u64 out_value[2], u64* precomputed2) {
    std::vector<TierRatio> t(n);
    for (int i = 0; i < n; ++i) t[i] = {{tiers3[3 * i], tiers3[3 * i + 1]}, (uint8_t)tiers3[3 * i + 2]};
    u128 v = ((u128)value[1] << 64) | value[0];
    tier_index_flag(v, t.data(), n, index, flag);
    u128 r = asset_value_via_tiers(v, t.data(), n);
    out_value[0] = (u64)r; out_value[1] = (u64)(r >> 64);
    if (precomputed2) {
        u128 pre[64];
        tier_precomputed(t.data(), n, pre);
        for (int i = 0; i < n; ++i) { precomputed2[2 * i] = (u64)pre[i]; precomputed2[2 * i + 1] = (u64)(pre[i] >> 64); }
    }
}
// fills equity / debt / collateral of the packed account headers, valid[i] = 1 / 0
void orc_account_totals(PackedAccountHdr* acc, const PackedAsset* assets, size_t n, const PackedCexConst* consts, size_t n_cex, uint8_t* valid) {
    std::vector<CexAssetConst> c(n_cex);
    for (size_t a = 0; a < n_cex; ++a) {
        c[a].base_price = consts[a].base_price;
        for (int i = 0; i < 12; ++i) {
            c[a].loan[i] = {{consts[a].loan[i].boundary[0], consts[a].loan[i].boundary[1]}, consts[a].loan[i].ratio};
            c[a].margin[i] = {{consts[a].margin[i].boundary[0], consts[a].margin[i].boundary[1]}, consts[a].margin[i].ratio};
            c[a].pm[i] = {{consts[a].pm[i].boundary[0], consts[a].pm[i].boundary[1]}, consts[a].pm[i].ratio};
        }
    }
#pragma omp parallel for schedule(dynamic, 64)
    for (size_t i = 0; i < n; ++i) {
        std::vector<AccountAsset> aa(acc[i].n_assets);
        for (uint32_t j = 0; j < acc[i].n_assets; ++j) {
            const PackedAsset& p = assets[acc[i].asset_off + j];
            aa[j] = {(uint16_t)p.index, p.equity, p.debt, p.loan, p.margin, p.portfolio_margin};
        }
        AccountTotals t = account_totals(aa.data(), aa.size(), c.data());
        acc[i].equity[0] = (u64)t.equity; acc[i].equity[1] = (u64)(t.equity >> 64);
        acc[i].debt[0] = (u64)t.debt; acc[i].debt[1] = (u64)(t.debt >> 64);
        acc[i].collateral[0] = (u64)t.collateral; acc[i].collateral[1] = (u64)(t.collateral >> 64);
        if (valid) valid[i] = t.valid ? 1 : 0;
    }
}
// levels_out (optional): concatenation of levels 1..depth, level l holding ceil(n/2^l) nodes
void orc_merkle_build(const Fr* leaves, size_t n, int depth, const Fr* nil_leaf, Fr* levels_out, Fr* nil_out,
                      Fr* root_out) {
    MerkleTree t = merkle_build(leaves, n, depth, *nil_leaf);
    if (levels_out) {
        size_t off = 0;
        for (int l = 1; l <= depth; ++l) {
            memcpy(levels_out + off, t.levels[l].data(), t.levels[l].size() * sizeof(Fr));
            off += t.levels[l].size();
        }
    }
    if (nil_out) memcpy(nil_out, t.nil.data(), (depth + 1) * sizeof(Fr));
    *root_out = t.root;
}
// sparse tree: set n (key, leaf) pairs, Build, then proofs (nq x depth) for query keys, root and optional verification
void orc_sparse_tree(const uint32_t* keys, const Fr* leaves, size_t n, int depth, const Fr* nil_leaf,
                     const uint32_t* qkeys, size_t nq, Fr* proofs_out, Fr* root_out) {
    SparseMerkleTree t(depth, *nil_leaf);
    for (size_t i = 0; i < n; ++i) t.set(keys[i], leaves[i]);
    t.build();
    for (size_t i = 0; i < nq; ++i) {
        std::vector<Fr> p = t.proof(qkeys[i]);
        memcpy(proofs_out + i * depth, p.data(), depth * sizeof(Fr));
    }
    *root_out = t.root;
}
int orc_merkle_verify(const Fr* root, uint32_t key, const Fr* proof, int depth, const Fr* leaf) {
    std::vector<Fr> p(proof, proof + depth);
    return merkle_verify(*root, key, p, *leaf) ? 1 : 0;
}
void orc_fr_to_be(const Fr* a, uint8_t* out, size_t n) { for (size_t i = 0; i < n; ++i) a[i].to_be_bytes(out + 32 * i); }
void orc_fr_from_be(const uint8_t* in, Fr* out, size_t n) { for (size_t i = 0; i < n; ++i) out[i] = Fr::from_be_bytes(in + 32 * i, 32); }

// ---- synthetic Groth16 ----
struct SynthHandle { SynthInstance inst; SynthKey key; };

void* orc_synth_create(size_t n_inputs, size_t n_cons, size_t n_public, u64 seed, int z_bitrev) {
    auto* h = new SynthHandle;
    h->inst = synth_instance(n_inputs, n_cons, seed);
    h->key = synth_setup(h->inst, n_public, seed ^ 0x5A4B504F52ULL, z_bitrev != 0);
    return h;
}
void orc_synth_destroy(void* p) { delete (SynthHandle*)p; }
// dims: [log2d, n_wires, n_public, n_cons, nZ]
void orc_synth_dims(void* p, u64* dims) {
    auto* h = (SynthHandle*)p;
    dims[0] = h->key.log2d; dims[1] = h->key.n_wires; dims[2] = h->key.n_public; dims[3] = h->key.n_cons;
    dims[4] = h->key.Z.size();
}
void orc_synth_export(void* p, G1A* A, G1A* B1, G2A* B2, G1A* K, G1A* Z, G1A* abd1 /*alpha,beta,delta*/,
                      G2A* bd2 /*beta,delta*/, Fr* w, Fr* a, Fr* b, Fr* c) {
    auto* h = (SynthHandle*)p;
    const SynthKey& k = h->key;
    memcpy(A, k.A.data(), k.A.size() * 64); memcpy(B1, k.B1.data(), k.B1.size() * 64);
    memcpy(B2, k.B2.data(), k.B2.size() * 128); memcpy(K, k.K.data(), k.K.size() * 64);
    memcpy(Z, k.Z.data(), k.Z.size() * 64);
    abd1[0] = k.alpha1; abd1[1] = k.beta1; abd1[2] = k.delta1;
    bd2[0] = k.beta2; bd2[1] = k.delta2;
    memcpy(w, h->inst.w.data(), h->inst.w.size() * 32);
    memcpy(a, h->inst.a.data(), h->inst.a.size() * 32);
    memcpy(b, h->inst.b.data(), h->inst.b.size() * 32);
    memcpy(c, h->inst.c.data(), h->inst.c.size() * 32);
}
// out: Ar (G1 affine 64 B) | Bs (G2 affine 128 B) | Krs (G1 affine 64 B), Montgomery limbs
void orc_synth_prove_tail(void* p, const Fr* r, const Fr* s, uint8_t* out256) {
    auto* h = (SynthHandle*)p;
    ProofPts pr = groth16_prove_tail(h->key, h->inst, *r, *s);
    memcpy(out256, &pr.ar, 64); memcpy(out256 + 64, &pr.bs, 128); memcpy(out256 + 192, &pr.krs, 64);
}
int orc_synth_check(void* p, const Fr* r, const Fr* s, const uint8_t* proof256) {
    auto* h = (SynthHandle*)p;
    ProofPts pr;
    memcpy(&pr.ar, proof256, 64); memcpy(&pr.bs, proof256 + 64, 128); memcpy(&pr.krs, proof256 + 192, 64);
    return groth16_check_in_exponent(h->key, h->inst, *r, *s, pr) ? 1 : 0;
}
// the verifier's equation with a real pairing (algos.hpp groth16_verify_pairing); public wires = w[0..n_public)
int orc_synth_verify_pairing(void* p, const uint8_t* proof256) {
    auto* h = (SynthHandle*)p;
    ProofPts pr;
    memcpy(&pr.ar, proof256, 64); memcpy(&pr.bs, proof256 + 64, 128); memcpy(&pr.krs, proof256 + 192, 64);
    SynthVK vk = synth_vk(h->key);
    return groth16_verify_pairing(vk, h->inst.w.data(), pr) ? 1 : 0;
}
// BSB22: the Pedersen key of the committed wires for this synthetic setup, and the commitment-extended verification equation
void orc_synth_commitment_basis(void* p, const uint32_t* committed, size_t n, const Fr* sigma, G1A* basis, G1A* basis_sigma) {
    auto* h = (SynthHandle*)p;
    synth_commitment_basis(h->key, committed, n, *sigma, basis, basis_sigma);
}
int orc_synth_verify_pairing_commit(void* p, const uint8_t* proof256, const G1A* commitment, const G1A* pok, const G2A* g2_sigma) {
    auto* h = (SynthHandle*)p;
    ProofPts pr;
    memcpy(&pr.ar, proof256, 64); memcpy(&pr.bs, proof256 + 64, 128); memcpy(&pr.krs, proof256 + 192, 64);
    SynthVK vk = synth_vk(h->key);
    return groth16_verify_pairing_commit(vk, h->inst.w.data(), pr, *commitment, *pok, *g2_sigma) ? 1 : 0;
}
// out: 6 Fp2 coefficients (12 Fp, Montgomery) of the reduced Tate pairing t(P, Q)
void orc_pairing(const G1A* P, const G2A* Q, Fp* out12) {
    Fp12 f = pairing(*P, *Q);
    memcpy(out12, &f, sizeof(Fp12));
}
void orc_fp12_mul(const Fp* a, const Fp* b, Fp* out) {
    Fp12 x, y; memcpy(&x, a, sizeof(Fp12)); memcpy(&y, b, sizeof(Fp12));
    Fp12 r = Fp12::mul(x, y); memcpy(out, &r, sizeof(Fp12));
}
void orc_fp12_pow_fr(const Fp* a, const Fr* e, Fp* out) {
    Fp12 x; memcpy(&x, a, sizeof(Fp12));
    u64 c[4]; e->to_canon(c);
    Fp12 r = Fp12::pow(x, c, 4); memcpy(out, &r, sizeof(Fp12));
}
// Pedersen: basis_i = b_i * G1, basis_sigma_i = sigma * basis_i, g2_sigma = sigma * G2 for the caller's scalars
void orc_g2_mul_gen(const Fr* k, G2A* out) { FixedBase<Fp2> g2(g2_gen()); *out = g2.mul_aff(*k); }
int orc_pedersen_verify_pairing(const G1A* commitment, const G1A* pok, const G2A* g2_sigma) {
    return pedersen_verify_pairing(*commitment, *pok, *g2_sigma) ? 1 : 0;
}
// the synthetic instance's R1CS as term lists: which 0 = a (L), 1 = b (R), 2 = c (O).  Pass NULL outputs to get nnz.
size_t orc_synth_r1cs(void* p, int which, u64* row_ptr, uint32_t* wire_ids, Fr* coeffs) {
    auto* h = (SynthHandle*)p;
    size_t nnz = 0;
    for (size_t j = 0; j < h->inst.rows.size(); ++j) {
        const R1CSRow& r = h->inst.rows[j];
        const auto& terms = which == 0 ? r.a : (which == 1 ? r.b : r.c);
        if (row_ptr) row_ptr[j] = nnz;
        for (auto& t : terms) {
            if (wire_ids) wire_ids[nnz] = t.first;
            if (coeffs) coeffs[nnz] = t.second;
            ++nnz;
        }
    }
    if (row_ptr) row_ptr[h->inst.rows.size()] = nnz;
    return nnz;
}
// gnark-crypto compressed point encoding (marshal.hpp)
void orc_g1_compress(const G1A* p, size_t n, uint8_t* out) { for (size_t i = 0; i < n; ++i) g1_compress(p[i], out + 32 * i); }
void orc_g2_compress(const G2A* p, size_t n, uint8_t* out) { for (size_t i = 0; i < n; ++i) g2_compress(p[i], out + 64 * i); }
int orc_g1_decompress(const uint8_t* in, size_t n, G1A* out) {
    for (size_t i = 0; i < n; ++i) { int rc = g1_decompress(in + 32 * i, &out[i]); if (rc) return rc; }
    return 0;
}
int orc_g2_decompress(const uint8_t* in, size_t n, G2A* out) {
    for (size_t i = 0; i < n; ++i) { int rc = g2_decompress(in + 64 * i, &out[i]); if (rc) return rc; }
    return 0;
}
// gnark raw proof encoding of the three points (proof.WriteRawTo, prover.go:201): big-endian
// Ar.X|Ar.Y | Bs.X.A1|Bs.X.A0|Bs.Y.A1|Bs.Y.A0 | Krs.X|Krs.Y   (256 B; commitments follow separately)
void orc_proof_raw(const uint8_t* proof256, uint8_t* out256) {
    // gnark-crypto marshal.go RawBytes: infinity = flag 0b01 in the two top bits of byte 0 (0x40), rest zero
    const Fp* f = (const Fp*)proof256;
    auto zero = [&](int lo, int hi) { for (int i = lo; i < hi; ++i) if (!f[i].is_zero()) return false; return true; };
    memset(out256, 0, 256);
    if (zero(0, 2)) out256[0] = 0x40; else { f[0].to_be_bytes(out256); f[1].to_be_bytes(out256 + 32); }
    if (zero(2, 6)) out256[64] = 0x40;
    else {
        f[3].to_be_bytes(out256 + 64); f[2].to_be_bytes(out256 + 96);
        f[5].to_be_bytes(out256 + 128); f[4].to_be_bytes(out256 + 160);
    }
    if (zero(6, 8)) out256[192] = 0x40; else { f[6].to_be_bytes(out256 + 192); f[7].to_be_bytes(out256 + 224); }
}


// ---- the CPU baseline of bench.py (cpubase.hpp): performance-minded port, checked against the plain oracle in tests ----
void orc_fast_g1_msm(const G1A* pts, const Fr* sc, size_t n, int window, G1A* out) {
    *out = fast::to_oracle_affine(fast::multi_exp<fast::FFp>(pts, sc, n, window));
}
void orc_fast_g2_msm(const G2A* pts, const Fr* sc, size_t n, int window, G2A* out) {
    *out = fast::to_oracle_affine(fast::multi_exp<fast::FFp2>(pts, sc, n, window));
}
// a, b, c: 2^log2d elements each (zero padded); h is left in a (bit-reversed order)
void orc_fast_compute_h(Fr* a, Fr* b, Fr* c, int log2d) {
    fast::FastDomain d(log2d);
    fast::compute_h(d, (fast::FFr*)a, (fast::FFr*)b, (fast::FFr*)c);
}
// times: fft, 4 x G1, G2, 2 x commitment (seconds); outputs keep the work observable
void orc_fast_prove_tail_work(int log2d, const G1A* g1, const G2A* g2, const Fr* w, Fr* a, Fr* b, Fr* c, size_t n_commit, double* times,
                              G1A* g1_out, G2A* g2_out) {
    fast::TailTimes t = fast::prove_tail_work(log2d, g1, g2, w, a, b, c, n_commit, g1_out, g2_out);
    times[0] = t.fft_s; times[1] = t.msm_g1_s; times[2] = t.msm_g2_s; times[3] = t.commit_s;
}
int orc_threads() { return fast::threads(); }

// <s, x> over Fr for the first n points of one synthetic key array (trapdoor.py synth_dot in one fused, OpenMP pass — the numpy form
// stays as its cross-check in tests/test_trapdoor_cpu.py): s_i = k(i / 32) + (i % 32) q with the 64-bit mixer of zkpor_pk_synth,
// zero where the synthetic point is infinity (hash test mod inf_mod, or i < inf_below)
void orc_synth_dot(u64 seed, int arr, const Fr* x, size_t n, u64 inf_mod, size_t inf_below, Fr* out) {
    auto smix = [](u64 v) { v += 0x9e3779b97f4a7c15ULL; v = (v ^ (v >> 30)) * 0xbf58476d1ce4e5b9ULL; v = (v ^ (v >> 27)) * 0x94d049bb133111ebULL; return v ^ (v >> 31); };
    const u64 q = 0x9e3779b97f4a7c15ULL;
    Fr acc = Fr::zero();
#pragma omp parallel
    {
        Fr loc = Fr::zero();
#pragma omp for schedule(static) nowait
        for (size_t i = 0; i < n; ++i) {
            if (i < inf_below) continue;
            if (inf_mod) { u64 h = (u64)i * 0xd6e8feb86659fd93ULL; h ^= h >> 32; if (h % inf_mod == 0) continue; }
            const u64 run = i / 32, j = i % 32;
            const u64 k = smix(seed ^ ((u64)(arr + 1) * 0xa0761d6478bd642fULL) ^ (run * 0xe7037ed1a0b428dbULL)) | 1;
            const unsigned __int128 sv = (unsigned __int128)k + (unsigned __int128)j * q;
            u64 c[4] = {(u64)sv, (u64)(sv >> 64), 0, 0};
            loc = Fr::add(loc, Fr::mul(Fr::from_canon(c), x[i]));
        }
#pragma omp critical
        acc = Fr::add(acc, loc);
    }
    *out = acc;
}

// ---- FFT-free check of computeH's output (quotient.hpp): out6 = A(tau), B(tau), C(tau), H(tau), H(tau)(tau^D-1), A(tau)B(tau)-C(tau);
// returns 1 when the last two agree, 0 when they differ, -1 when tau lies in the domain
int orc_quotient_identity(int log2d, const Fr* a, const Fr* b, const Fr* c, size_t n_cons, const Fr* h, int h_bitrev, const Fr* tau, Fr* out6) {
    Fr t = *tau;
    Fr tD = t;
    for (int i = 0; i < log2d; ++i) tD = Fr::sqr(tD);
    if (Fr::sub(tD, Fr::one()).is_zero()) return -1;
    orc_quot::Eval e = orc_quot::quotient_identity(log2d, a, b, c, n_cons, h, h_bitrev, t);
    if (out6) { out6[0] = e.At; out6[1] = e.Bt; out6[2] = e.Ct; out6[3] = e.Ht; out6[4] = e.lhs; out6[5] = e.rhs; }
    return Fr::sub(e.lhs, e.rhs).is_zero() ? 1 : 0;
}

// ---- R1CS satisfaction from the statement alone (TEST INFRASTRUCTURE; gnark's contract for r1cs.Solve — constraint/bn254/solver.go, reached through
// groth16.Prove at src/prover/prover/prover.go:269 — is a wire vector with (L w) o (R w) = O w on every row): three CSR matrices whose entries index a
// coefficient table, Montgomery limbs, this file's own field arithmetic (bn254.hpp) — nothing of the package's executors.  Returns the number of failing rows,
// *first_bad = the lowest one (or n_constraints).  An index outside its table counts the row as failing.
uint64_t orc_r1cs_failing_rows(const Fr* coeff, uint64_t n_coeff, const uint64_t* const row_ptr[3], const uint32_t* const cid[3], const uint32_t* const wid[3],
                               uint64_t n_constraints, const Fr* w, uint64_t n_wires, uint64_t* first_bad) {
    uint64_t bad = 0, lowest = n_constraints;
#pragma omp parallel for schedule(static) reduction(+ : bad) reduction(min : lowest)
    for (uint64_t r = 0; r < n_constraints; ++r) {
        Fr v[3];
        bool ok = true;
        for (int m = 0; m < 3; ++m) {
            Fr acc = Fr::zero();
            for (uint64_t p = row_ptr[m][r]; p < row_ptr[m][r + 1]; ++p) {
                if (cid[m][p] >= n_coeff || wid[m][p] >= n_wires) { ok = false; break; }
                acc = Fr::add(acc, Fr::mul(coeff[cid[m][p]], w[wid[m][p]]));
            }
            v[m] = acc;
        }
        if (!ok || !Fr::sub(Fr::mul(v[0], v[1]), v[2]).is_zero()) { ++bad; if (r < lowest) lowest = r; }
    }
    if (first_bad) *first_bad = lowest;
    return bad;
}

}  // extern "C"
