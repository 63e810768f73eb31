// Poseidon account-tree hashing for gfx950 — replaces the CPU loops of the reference's witness service:
//   leaf hashes   utils.AccountInfoToHash + ComputeUserAssetsCommitment + PaddingAccountAssets
//                 (src/utils/utils.go:744-750, :188-221, :147-186; driven by src/witness/main.go:130-199)
//   tree build    merkletree.FixedDepthMerkleTree.Build / nilHashes / Root
//                 (src/utils/merkletree/merkletree.go:192-279, :159-170)
//   hash function poseidon.Poseidon / PoseidonBytes / NewPoseidon of the bnb-chain gnark-crypto fork
//                 (un-vendored; x^5 HADES permutation, R_F = 8, R_P(t), Grain-LFSR parameters, inputs chained in
//                 blocks of 12 through the capacity element — SURVEY.md Appendix A.6).
// The permutation parameters are generated at start-up by the published Grain procedure (poseidon_params below,
// an independent restatement of the one in oracle/poseidon.hpp; tests compare the two and the iden3 KATs).
#include "common.cuh"
#include "fe29.cuh"
#include "solver_instr.cuh"
#include <vector>

namespace zk {

static const int POS_RF = 8;
static const int POS_MAX_T = 13;
static inline int pos_rp(int t) {
    static const int tab[] = {56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65};
    return tab[t - 2];
}

// device blob, per width t in [2,13]: rc[(8+rp)*t] | mds[t*t] | prc[rp*t] | sparse[rp*(2t-1)] | post[(t-1)*(t-1)]
// prc/sparse/post are the optimised partial rounds (see optimise_partial_rounds)
struct PosTables {
    Fr* dev = nullptr;
    u32* dev29 = nullptr;  // the same blob as 9 x 29-bit limbs in the 2^261 Montgomery form (fe29.cuh), same element offsets
    u32 rc_off[POS_MAX_T + 1];
    u32 mds_off[POS_MAX_T + 1];
    u32 prc_off[POS_MAX_T + 1];
    u32 sp_off[POS_MAX_T + 1];
    u32 post_off[POS_MAX_T + 1];
    u32* devc = nullptr;   // the 29-bit tables once more, laid out for the 16-lanes-per-call kernels (k_tables_coop below)
    u32 c_off[POS_MAX_T + 1];
    std::vector<Fr> host;
};

// ---- Grain LFSR (Poseidon reference parameter generation: field = 1, sbox = 0 (x^alpha), n = 254) ----
struct GrainLfsr {
    uint8_t s[80];
    int head = 0;
    GrainLfsr(int t, int rf, int rp) {
        int k = 0;
        auto put = [&](unsigned v, int w) { for (int i = w - 1; i >= 0; --i) s[k++] = (uint8_t)((v >> i) & 1u); };
        put(1, 2); put(0, 4); put(254, 12); put((unsigned)t, 12); put((unsigned)rf, 10); put((unsigned)rp, 10);
        while (k < 80) s[k++] = 1;
        for (int i = 0; i < 160; ++i) clock();
    }
    int tap(int i) const { return s[(head + i) % 80]; }
    int clock() {
        int nb = tap(62) ^ tap(51) ^ tap(38) ^ tap(23) ^ tap(13) ^ tap(0);
        s[head] = (uint8_t)nb;
        head = (head + 1) % 80;
        return nb;
    }
    int next_bit() {  // self-shrinking: keep the second bit of a pair when the first is 1
        for (;;) {
            int a = clock(), b = clock();
            if (a) return b;
        }
    }
    Fr next_254(bool* below_modulus) {  // canonical limbs, MSB first
        Fr x = Fr::zero();
        for (int i = 253; i >= 0; --i)
            if (next_bit()) x.v[i >> 5] |= 1u << (i & 31);
        bool lt = false;
        for (int i = 7; i >= 0; --i) {
            if (x.v[i] != FrParams::mod(i)) { lt = x.v[i] < FrParams::mod(i); break; }
        }
        *below_modulus = lt;
        return x;
    }
};

static void poseidon_params(int t, std::vector<Fr>& rc, std::vector<Fr>& mds) {
    const int rp = pos_rp(t);
    GrainLfsr g(t, POS_RF, rp);
    rc.resize((size_t)(POS_RF + rp) * t);
    for (auto& c : rc) {
        bool ok;
        Fr v;
        do { v = g.next_254(&ok); } while (!ok);  // rejection sampling
        c = Fr::to_mont(v);
    }
    std::vector<Fr> xy(2 * t);
    for (;;) {
        for (auto& e : xy) {
            bool ok;
            Fr v = g.next_254(&ok);
            if (!ok) {  // reduce mod r (v < 2^254 < 2r)
                Fr m; for (int i = 0; i < 8; ++i) m.v[i] = FrParams::mod(i);
                u32 bw = 0;
                for (int i = 0; i < 8; ++i) { u64 d = (u64)v.v[i] - m.v[i] - bw; v.v[i] = (u32)d; bw = (u32)(d >> 32) & 1u; }
            }
            e = Fr::to_mont(v);
        }
        bool good = true;
        for (int i = 0; i < 2 * t && good; ++i)
            for (int j = i + 1; j < 2 * t; ++j) if (xy[i] == xy[j]) { good = false; break; }
        for (int i = 0; i < t && good; ++i)
            for (int j = 0; j < t; ++j) if (Fr::add(xy[i], xy[t + j]).is_zero()) { good = false; break; }
        if (good) break;
    }
    mds.resize((size_t)t * t);
    for (int i = 0; i < t; ++i)
        for (int j = 0; j < t; ++j) mds[(size_t)i * t + j] = Fr::inv(Fr::add(xy[i], xy[t + j]));
}

// Gauss-Jordan inverse of an n x n matrix over Fr (n <= 12); host, start-up only
static bool fr_mat_inv(std::vector<Fr>& a, int n, std::vector<Fr>& inv) {
    inv.assign((size_t)n * n, Fr::zero());
    for (int i = 0; i < n; ++i) inv[(size_t)i * n + i] = Fr::one();
    for (int c = 0; c < n; ++c) {
        int piv = -1;
        for (int r = c; r < n; ++r) if (!a[(size_t)r * n + c].is_zero()) { piv = r; break; }
        if (piv < 0) return false;
        if (piv != c) for (int k = 0; k < n; ++k) { std::swap(a[(size_t)c * n + k], a[(size_t)piv * n + k]); std::swap(inv[(size_t)c * n + k], inv[(size_t)piv * n + k]); }
        Fr d = Fr::inv(a[(size_t)c * n + c]);
        for (int k = 0; k < n; ++k) { a[(size_t)c * n + k] = Fr::mul(a[(size_t)c * n + k], d); inv[(size_t)c * n + k] = Fr::mul(inv[(size_t)c * n + k], d); }
        for (int r = 0; r < n; ++r) {
            if (r == c) continue;
            Fr f = a[(size_t)r * n + c];
            if (f.is_zero()) continue;
            for (int k = 0; k < n; ++k) {
                a[(size_t)r * n + k] = Fr::sub(a[(size_t)r * n + k], Fr::mul(f, a[(size_t)c * n + k]));
                inv[(size_t)r * n + k] = Fr::sub(inv[(size_t)r * n + k], Fr::mul(f, inv[(size_t)c * n + k]));
            }
        }
    }
    return true;
}

// Optimised partial rounds.  The S-box of a partial round touches state[0] only, so the round matrix
//   N_i = [[m00, v],[w, Mhat_i]] = diag(1, Mhat_i) * [[m00, v],[Mhat_i^-1 w, I]]
// can leave its block-diagonal factor to the NEXT round (it commutes with that round's S-box layer): N_(i+1) = M * diag(1, Mhat_i),
// round constants k_i = diag(1, Mhat_(i-1))^-1 c_i, and one dense (t-1)x(t-1) block Mhat_last after the last partial round.
// A partial round then costs 3 + (2t-1) products instead of 3 + t^2 (t = 3: 8 vs 12; t = 13: 28 vs 172).
// Derivation checked against the plain permutation in tools/poseidon_grain.py (permute_optimised) and on the device by the
// parity tests against the oracle's plain permutation.
static bool optimise_partial_rounds(int t, const std::vector<Fr>& rc, const std::vector<Fr>& mds, std::vector<Fr>& prc,
                                    std::vector<Fr>& sparse, std::vector<Fr>& post) {
    const int rp = pos_rp(t), m = t - 1;
    std::vector<Fr> N = mds, prev_inv, mhat((size_t)m * m), inv;
    prc.assign((size_t)rp * t, Fr::zero());
    sparse.assign((size_t)rp * (2 * t - 1), Fr::zero());
    for (int i = 0; i < rp; ++i) {
        for (int r = 0; r < m; ++r) for (int c = 0; c < m; ++c) mhat[(size_t)r * m + c] = N[(size_t)(r + 1) * t + (c + 1)];
        std::vector<Fr> tmp = mhat;
        if (m > 0 && !fr_mat_inv(tmp, m, inv)) return false;
        Fr* sp = &sparse[(size_t)i * (2 * t - 1)];
        sp[0] = N[0];
        for (int c = 0; c < m; ++c) sp[1 + c] = N[1 + c];                       // v
        for (int r = 0; r < m; ++r) {                                           // what = Mhat^-1 w
            Fr acc = Fr::zero();
            for (int c = 0; c < m; ++c) acc = Fr::add(acc, Fr::mul(inv[(size_t)r * m + c], N[(size_t)(c + 1) * t]));
            sp[t + r] = acc;
        }
        const Fr* c_i = &rc[(size_t)(POS_RF / 2 + i) * t];
        Fr* k = &prc[(size_t)i * t];
        k[0] = c_i[0];
        for (int r = 0; r < m; ++r) {
            if (i == 0) k[1 + r] = c_i[1 + r];
            else {
                Fr acc = Fr::zero();
                for (int c = 0; c < m; ++c) acc = Fr::add(acc, Fr::mul(prev_inv[(size_t)r * m + c], c_i[1 + c]));
                k[1 + r] = acc;
            }
        }
        prev_inv = inv;
        // N <- M * diag(1, Mhat)
        std::vector<Fr> Nn((size_t)t * t);
        for (int r = 0; r < t; ++r) {
            Nn[(size_t)r * t] = mds[(size_t)r * t];
            for (int c = 1; c < t; ++c) {
                Fr acc = Fr::zero();
                for (int x = 1; x < t; ++x) acc = Fr::add(acc, Fr::mul(mds[(size_t)r * t + x], mhat[(size_t)(x - 1) * m + (c - 1)]));
                Nn[(size_t)r * t + c] = acc;
            }
        }
        N.swap(Nn);
    }
    post = mhat;
    return true;
}

// element i of the 2^256-form blob -> 9 limbs of the 2^261 form
__global__ void k_tables_to_limbs29(const Fr* in, u32* out, u32 count) {
    u32 i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    Fr c32 = Fr::one();
    for (int k = 0; k < 5; ++k) c32 = Fr::add(c32, c32);
    Fr29 v = Fr29::from32<0>(Fr::mul(in[i], c32));
    for (int k = 0; k < 9; ++k) out[9u * i + k] = v.l[k];
}

// The cooperative kernels (namespace coop) hold one state element per lane and every lane needs ITS constant of the round: read from the
// element-major blob that is one 36-byte run per lane — 18 cache lines per load instruction, and four waves of a CU behind one vector L1
// (measured: a launch of one wave per SIMD ran 1.6x slower per wave than one wave per CU).  Here the same constants lane-interleaved: a
// ROW is 9 limbs x 16 lanes (limb i of lane j at word 16 i + j), so a load instruction of a group reads 64 consecutive bytes, and the four
// groups of a wave read the same 64.  Rows of width t, from c_off[t]:
//   0..7                  full-round constants (the four rounds before, the four after the partial ones); lane j: rc[r t + j]
//   8 + 3 i + {0, 1, 2}   partial round i: its constants k_i; lane 0: m00, lane j: v_j; lane 0: m00, lane j: what_j
//   8 + 3 rp + c          column c of the round matrix; lane j: M[j][c]
//   8 + 3 rp + t + c      column c of the dense block left over by the optimised rounds; lane j >= 1: post[j - 1][c]
// Lanes j >= t hold lane 0's entries (they run along on a valid element).
ZK_HD constexpr u32 coop_rows(int t, int rp) { return 8u + 3u * (u32)rp + 2u * (u32)t - 1u; }
struct CoopSrc { u32 rc_off, mds_off, prc_off, sp_off, post_off, c_off; int t, rp; };
__global__ void k_tables_coop(const u32* __restrict__ tab29, CoopSrc S, u32* __restrict__ out) {
    const int t = S.t, rp = S.rp;
    const u32 words = coop_rows(t, rp) * 144u;
    for (u32 idx = blockIdx.x * blockDim.x + threadIdx.x; idx < words; idx += gridDim.x * blockDim.x) {
        const u32 row = idx / 144u, limb = (idx % 144u) / 16u, lane = idx % 16u;
        const u32 jj = lane < (u32)t ? lane : 0u;
        u32 src;
        if (row < 8u) { const u32 r = row < 4u ? row : (u32)(POS_RF / 2 + rp) + (row - 4u); src = S.rc_off + r * (u32)t + jj; }
        else if (row < 8u + 3u * (u32)rp) {
            const u32 i = (row - 8u) / 3u, kind = (row - 8u) % 3u, sp = S.sp_off + i * (2u * (u32)t - 1u);
            src = kind == 0u ? S.prc_off + i * (u32)t + jj : (kind == 1u ? sp + jj : (jj == 0u ? sp : sp + (u32)t + jj - 1u));
        } else if (row < 8u + 3u * (u32)rp + (u32)t) src = S.mds_off + jj * (u32)t + (row - 8u - 3u * (u32)rp);
        else src = S.post_off + (jj ? jj - 1u : 0u) * (u32)(t - 1) + (row - 8u - 3u * (u32)rp - (u32)t);
        out[S.c_off + idx] = tab29[9u * src + limb];
    }
}

static int32_t pos_tables_get(zkpor_ctx* ctx, PosTables** out) {
    if (ctx->pos_tables) { *out = (PosTables*)ctx->pos_tables; return ZKPOR_OK; }
    PosTables* T = new PosTables();
    for (int t = 2; t <= POS_MAX_T; ++t) {
        std::vector<Fr> rc, mds;
        poseidon_params(t, rc, mds);
        T->rc_off[t] = (u32)T->host.size();
        T->host.insert(T->host.end(), rc.begin(), rc.end());
        T->mds_off[t] = (u32)T->host.size();
        T->host.insert(T->host.end(), mds.begin(), mds.end());
        std::vector<Fr> prc, sparse, post;
        if (!optimise_partial_rounds(t, rc, mds, prc, sparse, post)) { delete T; ctx->err = "poseidon: singular partial-round block"; return ZKPOR_E_STATE; }
        T->prc_off[t] = (u32)T->host.size();
        T->host.insert(T->host.end(), prc.begin(), prc.end());
        T->sp_off[t] = (u32)T->host.size();
        T->host.insert(T->host.end(), sparse.begin(), sparse.end());
        T->post_off[t] = (u32)T->host.size();
        T->host.insert(T->host.end(), post.begin(), post.end());
    }
    ZK_HIP(ctx, hipMalloc((void**)&T->dev, T->host.size() * sizeof(Fr)));
    ZK_HIP(ctx, hipMemcpyAsync(T->dev, T->host.data(), T->host.size() * sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
    ZK_HIP(ctx, hipMalloc((void**)&T->dev29, T->host.size() * 36));
    hipLaunchKernelGGL(k_tables_to_limbs29, dim3((unsigned)((T->host.size() + 255) / 256)), dim3(256), 0, ctx->stream, T->dev, T->dev29, (u32)T->host.size());
    ZK_KERNEL_CHECK(ctx);
    u32 cw = 0;
    for (int t = 2; t <= POS_MAX_T; ++t) { T->c_off[t] = cw; cw += coop_rows(t, pos_rp(t)) * 144u; }
    ZK_HIP(ctx, hipMalloc((void**)&T->devc, (size_t)cw * 4));
    for (int t = 2; t <= POS_MAX_T; ++t) {
        const CoopSrc S{T->rc_off[t], T->mds_off[t], T->prc_off[t], T->sp_off[t], T->post_off[t], T->c_off[t], t, pos_rp(t)};
        hipLaunchKernelGGL(k_tables_coop, dim3(32), dim3(256), 0, ctx->stream, (const u32*)T->dev29, S, T->devc);
        ZK_KERNEL_CHECK(ctx);
    }
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->pos_tables = T;
    *out = T;
    return ZKPOR_OK;
}
void pos_tables_free(zkpor_ctx* ctx) {
    if (!ctx->pos_tables) return;
    PosTables* T = (PosTables*)ctx->pos_tables;
    if (T->dev) (void)hipFree(T->dev);
    if (T->dev29) (void)hipFree(T->dev29);
    if (T->devc) (void)hipFree(T->devc);
    delete T;
    ctx->pos_tables = nullptr;
}

ZK_HD Fr pow5(const Fr& x) {
    Fr x2 = Fr::sqr(x);
    return Fr::mul(Fr::sqr(x2), x);
}

struct PermTab {  // pointers into the table blob for one width
    const Fr* rc; const Fr* m; const Fr* prc; const Fr* sp; const Fr* post;
};

// width-3 permutation with the state in registers (the Merkle-node hash: ~N of these per tree build)
ZK_HD void permute3(Fr& s0, Fr& s1, Fr& s2, const PermTab& T) {
    const int rp = 57;
    const Fr* m = T.m;
    for (int half = 0; half < 2; ++half) {
        for (int rr = 0; rr < POS_RF / 2; ++rr) {
            const Fr* c = T.rc + 3 * (half ? POS_RF / 2 + rp + rr : rr);
            s0 = pow5(Fr::add(s0, c[0])); s1 = pow5(Fr::add(s1, c[1])); s2 = pow5(Fr::add(s2, c[2]));
            Fr t0 = Fr::add(Fr::add(Fr::mul(m[0], s0), Fr::mul(m[1], s1)), Fr::mul(m[2], s2));
            Fr t1 = Fr::add(Fr::add(Fr::mul(m[3], s0), Fr::mul(m[4], s1)), Fr::mul(m[5], s2));
            Fr t2 = Fr::add(Fr::add(Fr::mul(m[6], s0), Fr::mul(m[7], s1)), Fr::mul(m[8], s2));
            s0 = t0; s1 = t1; s2 = t2;
        }
        if (half) break;
        for (int i = 0; i < rp; ++i) {  // sparse partial rounds: 3 + 5 products
            const Fr* k = T.prc + 3 * i;
            const Fr* sp = T.sp + 5 * i;
            Fr x0 = pow5(Fr::add(s0, k[0]));
            s1 = Fr::add(s1, k[1]); s2 = Fr::add(s2, k[2]);
            s0 = Fr::add(Fr::add(Fr::mul(sp[0], x0), Fr::mul(sp[1], s1)), Fr::mul(sp[2], s2));
            s1 = Fr::add(s1, Fr::mul(sp[3], x0));
            s2 = Fr::add(s2, Fr::mul(sp[4], x0));
        }
        Fr u1 = Fr::add(Fr::mul(T.post[0], s1), Fr::mul(T.post[1], s2));
        Fr u2 = Fr::add(Fr::mul(T.post[2], s1), Fr::mul(T.post[3], s2));
        s1 = u1; s2 = u2;
    }
}
// generic width (state in memory): used for leaf hashing and the generic hash entry point
ZK_HD void permute_generic(Fr* st, int t, const PermTab& T, int rp) {
    Fr tmp[POS_MAX_T];
    const int m1 = t - 1;
    for (int half = 0; half < 2; ++half) {
        for (int rr = 0; rr < POS_RF / 2; ++rr) {
            const Fr* c = T.rc + (size_t)t * (half ? POS_RF / 2 + rp + rr : rr);
            for (int i = 0; i < t; ++i) st[i] = pow5(Fr::add(st[i], c[i]));
            for (int i = 0; i < t; ++i) {
                Fr acc = Fr::mul(T.m[i * t], st[0]);
                for (int j = 1; j < t; ++j) acc = Fr::add(acc, Fr::mul(T.m[i * t + j], st[j]));
                tmp[i] = acc;
            }
            for (int i = 0; i < t; ++i) st[i] = tmp[i];
        }
        if (half) break;
        for (int i = 0; i < rp; ++i) {
            const Fr* k = T.prc + (size_t)i * t;
            const Fr* sp = T.sp + (size_t)i * (2 * t - 1);
            Fr x0 = pow5(Fr::add(st[0], k[0]));
            Fr acc = Fr::mul(sp[0], x0);
            for (int j = 1; j < t; ++j) {
                Fr sj = Fr::add(st[j], k[j]);
                acc = Fr::add(acc, Fr::mul(sp[j], sj));
                st[j] = Fr::add(sj, Fr::mul(sp[t + j - 1], x0));
            }
            st[0] = acc;
        }
        for (int r = 0; r < m1; ++r) {
            Fr acc = Fr::mul(T.post[r * m1], st[1]);
            for (int c = 1; c < m1; ++c) acc = Fr::add(acc, Fr::mul(T.post[r * m1 + c], st[1 + c]));
            tmp[r] = acc;
        }
        for (int r = 0; r < m1; ++r) st[1 + r] = tmp[r];
    }
}

// ---- the wide permutations on 9 x 29-bit lazy limbs, state in registers ---------------------------------------------------
// The sponge's full blocks use width 13 (12 inputs + capacity): 8 of them per tier-50 account, 834 per CEX commitment.
// With the width a compile-time constant the state lives in registers (13 x 9 limbs) instead of scratch, a product is the
// 206-instruction lazy one, a sum of products pairs up into fused double products (one reduction per pair), and an
// addition is 9 independent adds + a 30-instruction product-free reduction — about 1.8x fewer instructions than the
// 32-bit generic path, which stays for the ragged last block and the small widths.
#if defined(__HIP_DEVICE_COMPILE__)
// a table row; the tables are read-only and the same for every lane, so the loads go through the constant address space
// (scalar cache, SGPR results) — the callers pass wave-uniform pointers
ZK_D Fr29 k29(const u32* base, int idx) {
    typedef const u32 __attribute__((address_space(4))) cu32;
    cu32* p = (cu32*)(uintptr_t)(base + 9 * idx);
    Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = p[i];
    return r;
}
template <int N>
ZK_D Fr29 dot29(const u32* consts, const Fr29* v) {  // sum_j consts[j] * v[j]; all operands tight
    Fr29 acc = Fr29::mul2k(k29(consts, 0), v[0], k29(consts, 1), v[1]);
#pragma unroll
    for (int p = 1; p < N / 2; ++p)
        acc = Fr29::reduce32(Fr29::add_l(acc, Fr29::mul2k(k29(consts, 2 * p), v[2 * p], k29(consts, 2 * p + 1), v[2 * p + 1])));
    if (N & 1) acc = Fr29::reduce32(Fr29::add_l(acc, Fr29::mulk(k29(consts, N - 1), v[N - 1])));
    return acc;
}
ZK_D Fr29 pow5_29(const Fr29& x) {
    Fr29 x2 = Fr29::sqr(x);
    return Fr29::mul(Fr29::sqr(x2), x);
}
// S-box trace of the structured witness generator: the in-circuit Poseidon gadget (circuit/utils.go:12-21,
// 28-49 -> std/hash/poseidon) spends three multiplication wires per S-box — x^2, x^4, x^5 — and nothing else (round constants and
// the MDS layer are linear).  The sink stores them as trace[(s * 3 + c) * count + perm]: S-box s in round order (full rounds: lanes
// 0..T-1), component c, consecutive permutations adjacent (coalesced stores); a wire map scatters the slots to gnark's wire ids.
struct TraceSink {
    Fr* base;
    size_t count, perm;
    u32 s;
};
ZK_D Fr29 pow5_29_tr(const Fr29& x, TraceSink& ts) {
    Fr29 x2 = Fr29::sqr(x);
    Fr29 x4 = Fr29::sqr(x2);
    Fr29 x5 = Fr29::mul(x4, x);
    Fr* o = ts.base + (size_t)ts.s * 3u * ts.count + ts.perm;
    o[0] = Fr29::to32_div32(x2);
    o[ts.count] = Fr29::to32_div32(x4);
    o[2 * ts.count] = Fr29::to32_div32(x5);
    ++ts.s;
    return x5;
}
template <bool TR>
ZK_D Fr29 sbox29(const Fr29& x, TraceSink* ts) {
    if (TR) return pow5_29_tr(x, *ts);
    return pow5_29(x);
}
// st: tight limbs, any magnitude below ~60 r on entry; all tables as 9-limb rows at the same element offsets as PermTab
template <int T, bool TR = false>
ZK_D void permute29(Fr29 (&st)[T], const u32* rc, const u32* m, const u32* prc, const u32* sp, const u32* post, int rp, TraceSink* ts = nullptr) {
    for (int half = 0; half < 2; ++half) {
#pragma unroll 1
        for (int rr = 0; rr < POS_RF / 2; ++rr) {
            const u32* c = rc + 9 * T * (half ? POS_RF / 2 + rp + rr : rr);
            // the full rounds (8 of 73) run from private arrays with ROLLED loops: one copy of the S-box and of the dot
            // product instead of 13 (the unrolled form was 90 KB of code, beyond the 64 KB instruction cache)
            if (T <= 4) {  // narrow states (the 2-to-1 node hash): everything unrolled, everything in registers
                Fr29 tmp[T];
#pragma unroll
                for (int i = 0; i < T; ++i) st[i] = sbox29<TR>(Fr29::reduce32(Fr29::add_l(st[i], k29(c, i))), ts);
#pragma unroll
                for (int i = 0; i < T; ++i) tmp[i] = dot29<T>(m + 9 * i * T, st);
#pragma unroll
                for (int i = 0; i < T; ++i) st[i] = tmp[i];
                continue;
            }
            Fr29 sb[T], tmp[T];
#pragma unroll
            for (int i = 0; i < T; ++i) sb[i] = st[i];
#pragma unroll 1
            for (int i = 0; i < T; ++i) sb[i] = sbox29<TR>(Fr29::reduce32(Fr29::add_l(sb[i], k29(c, i))), ts);
#pragma unroll 1
            for (int i = 0; i < T; ++i) tmp[i] = dot29<T>(m + 9 * i * T, sb);
#pragma unroll
            for (int i = 0; i < T; ++i) st[i] = tmp[i];
        }
        if (half) break;
#pragma unroll 1
        for (int i = 0; i < rp; ++i) {
            const u32* k = prc + 9 * T * i;
            const u32* s = sp + 9 * (2 * T - 1) * i;
            st[0] = sbox29<TR>(Fr29::reduce32(Fr29::add_l(st[0], k29(k, 0))), ts);  // x0
#pragma unroll
            for (int j = 1; j < T; ++j) st[j] = Fr29::reduce32(Fr29::add_l(st[j], k29(k, j)));
            Fr29 acc = dot29<T>(s, st);
#pragma unroll
            for (int j = 1; j < T; ++j) st[j] = Fr29::add_l(st[j], Fr29::mulk(k29(s, T + j - 1), st[0]));  // loose: reduced at the next round's constant add
            st[0] = acc;
        }
        {
            Fr29 tmp[T - 1];
#pragma unroll
            for (int r = 1; r < T; ++r) st[r] = Fr29::reduce32(st[r]);
            if (T <= 4) {
#pragma unroll
                for (int r = 0; r < T - 1; ++r) tmp[r] = dot29<T - 1>(post + 9 * r * (T - 1), st + 1);
            } else {
#pragma unroll 1
                for (int r = 0; r < T - 1; ++r) tmp[r] = dot29<T - 1>(post + 9 * r * (T - 1), st + 1);
            }
#pragma unroll
            for (int r = 0; r < T - 1; ++r) st[1 + r] = tmp[r];
        }
    }
}
#endif

struct PosDev {  // everything a kernel needs to hash
    const Fr* tab;
    const u32* tab29;
    u32 rc_off[POS_MAX_T + 1];
    u32 mds_off[POS_MAX_T + 1];
    u32 prc_off[POS_MAX_T + 1];
    u32 sp_off[POS_MAX_T + 1];
    u32 post_off[POS_MAX_T + 1];
    const u32* tabc;                 // lane-interleaved rows for the 16-lanes-per-call kernels (k_tables_coop)
    u32 c_off[POS_MAX_T + 1];
    ZK_HD PermTab tabs(int t) const { return {tab + rc_off[t], tab + mds_off[t], tab + prc_off[t], tab + sp_off[t], tab + post_off[t]}; }
    int rp[POS_MAX_T + 1];
    int out_idx, carry_idx;
};

#if defined(__HIP_DEVICE_COMPILE__)
// one sponge block of compile-time width out of line: its register allocation (13 x 9 limbs for the full block) stays
// separate from the callers' (account parsing, the 32-bit generic path for the remaining widths).  Instantiated for the
// full block (13) and for the two ragged widths the reference's inputs produce: 5 (100, 1000 or 10 000 elements leave 4) and
// 6 (the 5-element account leaf).
template <int t>
__device__ __noinline__ void sponge_block29(const Fr* st, const PosDev& P, Fr* cap, Fr* out) {
    Fr29 s29[t];
#pragma unroll
    for (int i = 0; i < t; ++i) s29[i] = Fr29::from32<5>(st[i]);
    // the tables are the same for every lane: make the pointers provably uniform so the constants come in through the
    // scalar cache (s_load) instead of one vector load per lane and nine VGPRs per constant
    auto uni = [](const u32* p) {
        u64 v = (u64)(uintptr_t)p;
        u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
        return (const u32*)(uintptr_t)(((u64)hi << 32) | lo);
    };
    const u32* b = P.tab29;
    const int rp = __builtin_amdgcn_readfirstlane(P.rp[t]);
    permute29<t>(s29, uni(b + 9 * (size_t)P.rc_off[t]), uni(b + 9 * (size_t)P.mds_off[t]), uni(b + 9 * (size_t)P.prc_off[t]),
                 uni(b + 9 * (size_t)P.sp_off[t]), uni(b + 9 * (size_t)P.post_off[t]), rp);
    *cap = Fr29::to32_div32(P.carry_idx ? s29[1] : s29[0]);
    *out = Fr29::to32_div32(P.out_idx ? s29[1] : s29[0]);
}
#endif

#if defined(__HIP_DEVICE_COMPILE__)
// the 2-to-1 node hash of the account tree on the 29-bit path: H(l, r) = permute([0, l, r])[out_idx]
ZK_D Fr hash2_29(const Fr& l, const Fr& r, const PosDev& P) {
    Fr29 st[3] = {Fr29::zero(), Fr29::from32<5>(l), Fr29::from32<5>(r)};
    const u32* b = P.tab29;
    permute29<3>(st, b + 9 * (size_t)P.rc_off[3], b + 9 * (size_t)P.mds_off[3], b + 9 * (size_t)P.prc_off[3],
                 b + 9 * (size_t)P.sp_off[3], b + 9 * (size_t)P.post_off[3], 57);
    return Fr29::to32_div32(P.out_idx ? st[1] : st[0]);
}
#else
__device__ Fr hash2_29(const Fr& l, const Fr& r, const PosDev& P);  // host pass: kernels only need the declaration
#endif

// streaming sponge over blocks of 12 (poseidon.Poseidon of the bnb fork)
struct Sponge {
    Fr st[POS_MAX_T];
    Fr cap, out;
    int fill;
    ZK_HD void init() { cap = Fr::zero(); out = Fr::zero(); fill = 0; }
    ZK_HD void flush(const PosDev& P) {
        if (!fill) return;
        int t = fill + 1;
        st[0] = cap;
#if defined(__HIP_DEVICE_COMPILE__)
        if (P.tab29 && (t == POS_MAX_T || t == 5 || t == 6)) {  // registers + 29-bit limbs
            if (t == POS_MAX_T) sponge_block29<POS_MAX_T>(st, P, &cap, &out);
            else if (t == 5) sponge_block29<5>(st, P, &cap, &out);
            else sponge_block29<6>(st, P, &cap, &out);
            fill = 0;
            return;
        }
#endif
        permute_generic(st, t, P.tabs(t), P.rp[t]);
        cap = st[P.carry_idx];
        out = st[P.out_idx];
        fill = 0;
    }
    ZK_HD void push(const PosDev& P, const Fr& x) {
        st[1 + fill] = x;
        if (++fill == 12) flush(P);
    }
    ZK_HD Fr finish(const PosDev& P) { flush(P); return out; }
};

// out[i] = H(in[2i], in[2i+1]); for an odd count the missing right sibling of the last pair is `nil`
__global__ __launch_bounds__(256) void k_hash2_level(const Fr* __restrict__ in, u32 n_in, Fr nil, Fr* __restrict__ out,
                                                     PosDev P) {
    u32 i = blockIdx.x * 256u + threadIdx.x;
    u32 n_out = (n_in + 1u) >> 1;
    if (i >= n_out) return;
    Fr s1 = in[2 * i];
    Fr s2 = (2 * i + 1 < n_in) ? in[2 * i + 1] : nil;
    out[i] = hash2_29(s1, s2, P);
}

// `count` independent hashes of `len` inputs each
__global__ __launch_bounds__(64, 2) void k_hash_many(const Fr* __restrict__ in, u32 len, u32 count, Fr* __restrict__ out, PosDev P) {
    u32 i = blockIdx.x * 64u + threadIdx.x;
    if (i >= count) return;
    Sponge sp;
    sp.init();
    const Fr* x = in + (size_t)i * len;
    for (u32 k = 0; k < len; ++k) sp.push(P, x[k]);
    out[i] = sp.finish(P);
}

// ---- structured witness generation: wire families of BatchCreateUserCircuit that are data-parallel ----------
#if defined(__HIP_DEVICE_COMPILE__)
template <int T>
__device__ __noinline__ void trace_block29(Fr* st, const PosDev& P, TraceSink* ts) {
    Fr29 s29[T];
#pragma unroll
    for (int i = 0; i < T; ++i) s29[i] = Fr29::from32<5>(st[i]);
    auto uni = [](const u32* p) {
        u64 v = (u64)(uintptr_t)p;
        u32 lo = __builtin_amdgcn_readfirstlane((u32)v), hi = __builtin_amdgcn_readfirstlane((u32)(v >> 32));
        return (const u32*)(uintptr_t)(((u64)hi << 32) | lo);
    };
    const u32* b = P.tab29;
    const int rp = __builtin_amdgcn_readfirstlane(P.rp[T]);
    permute29<T, true>(s29, uni(b + 9 * (size_t)P.rc_off[T]), uni(b + 9 * (size_t)P.mds_off[T]), uni(b + 9 * (size_t)P.prc_off[T]),
                       uni(b + 9 * (size_t)P.sp_off[T]), uni(b + 9 * (size_t)P.post_off[T]), rp, ts);
#pragma unroll
    for (int i = 0; i < T; ++i) st[i] = Fr29::to32_div32(s29[i]);
}
#endif
// one permutation per thread: states[perm * T + i] in, final state out (in place), the S-box wires to trace
template <int T>
__global__ __launch_bounds__(64, 2) void k_poseidon_trace(Fr* __restrict__ states, size_t count, Fr* __restrict__ trace, PosDev P) {
    size_t i = (size_t)blockIdx.x * 64u + threadIdx.x;
    if (i >= count) return;
#if defined(__HIP_DEVICE_COMPILE__)
    Fr st[T];
#pragma unroll
    for (int k = 0; k < T; ++k) st[k] = states[i * T + k];
    TraceSink ts{trace, count, i, 0u};
    trace_block29<T>(st, P, &ts);
#pragma unroll
    for (int k = 0; k < T; ++k) states[i * T + k] = st[k];
#endif
}

// ---- the solver program's Poseidon instruction (host/solver_file.hpp kind 4; csrc/solver.hip launches it) -----------------------------
// One poseidon.Poseidon(api, inputs...) call of the circuit (circuit/utils.go:17,47; circuit/batch_create_user_circuit.go:104,129,181,270,281,
// 320) = ONE instruction: the thread evaluates the call's input expressions over the solved wires, runs the sponge and writes the three
// product wires of every S-box (x^2, x^4, x^5, permutation after permutation in round order) straight to their wire ids — what gnark's
// solver reaches by solving 3 x 169 constraints per full block one after the other.  Widths 3 / 5 / 6 / 13 run on the 29-bit register path,
// the other ragged widths (a sponge over n inputs ends in a block of width n mod 12 + 1) on the plain permutation below.
#if defined(__HIP_DEVICE_COMPILE__)
ZK_D void permute_plain_trace(Fr* st, int t, const PermTab& T, int rp, TraceSink& ts) {
    Fr tmp[POS_MAX_T];
    const int rounds = POS_RF + rp;
    for (int r = 0; r < rounds; ++r) {
        const Fr* c = T.rc + (size_t)t * r;
        const bool full = r < POS_RF / 2 || r >= POS_RF / 2 + rp;
        for (int i = 0; i < t; ++i) st[i] = Fr::add(st[i], c[i]);
        for (int i = 0; i < (full ? t : 1); ++i) {
            const Fr x2 = Fr::sqr(st[i]), x4 = Fr::sqr(x2), x5 = Fr::mul(x4, st[i]);
            Fr* o = ts.base + (size_t)ts.s * 3u * ts.count + ts.perm;
            o[0] = x2; o[ts.count] = x4; o[2 * ts.count] = x5;
            ++ts.s;
            st[i] = x5;
        }
        for (int i = 0; i < t; ++i) {
            Fr acc = Fr::mul(T.m[i * t], st[0]);
            for (int j = 1; j < t; ++j) acc = Fr::add(acc, Fr::mul(T.m[i * t + j], st[j]));
            tmp[i] = acc;
        }
        for (int i = 0; i < t; ++i) st[i] = tmp[i];
    }
}
#endif
// ---- the same instruction, GROUP-COOPERATIVE (one state spread over lanes): 16 lanes own one call, lane j holds state
// element j.  Why: a call is a serial chain — 834 permutations for a CEX commitment, 9 for a user's asset commitment, 1 per Merkle level —
// and a batch has at most a few thousand of them side by side, so the duration of the level is the LATENCY of one call, not throughput.
// Per permutation one lane runs ~420 dependent products (t^2 + 3t per full round, 3 + 2t - 1 per sparse partial round); spread over the lanes
// a full round is 3 + t products deep (S-boxes side by side, one matrix row per lane), a partial round 4 (the other lanes' share of the
// dot product rides in the S-box's first product slot, the cross-lane sum is four DPP row shifts).  The three product wires of every S-box
// go out through an LDS stash, sixteen conversions to the 8 x 32-bit memory form at a time, one per lane — alone they would double a
// partial round.  Same tables, same optimised partial rounds, bit-identical wires to k_gadget_poseidon (tests/test_circuit_gpu.py).
#if defined(__HIP_DEVICE_COMPILE__)
namespace coop {
constexpr int G = 16, XS = 152;                    // lanes per call; words per group in the exchange areas (16 x 9 + padding: four groups on distinct banks)
ZK_D Fr29 ld29(const u32* p) { Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = p[i];
    return r; }
template <int N> ZK_D u32 row_shl(u32 v) { return (u32)__builtin_amdgcn_update_dpp(0, (int)v, 0x100 + N, 0xf, 0xf, true); }   // lane i <- lane i + N of its row of 16, else 0
template <int N> ZK_D Fr29 fold(const Fr29& v) { Fr29 t;               // no carry sweep: the caller counts the bits
#pragma unroll
    for (int i = 0; i < 9; ++i) t.l[i] = v.l[i] + row_shl<N>(v.l[i]);
    return t; }
// lane 0 of the row: the sum of its 16 lanes.  The lanes hold products as they leave Fr29::mul (limbs 0..7 in [0, 2^29), limb 8 small and
// signed) or zero: four of them stay below 2^31, so two sweeps do for the four folds — a signed one after the first two, an UNSIGNED one
// at the end (four swept values reach 2^31 + 8).  Result: limbs 0..7 <= 2^29 + 3, fit for an add_l + reduce32, not for a product.
ZK_D Fr29 row_sum(Fr29 v) {
    v = Fr29::normed(fold<4>(fold<8>(v)));
    v = fold<1>(fold<2>(v));
    u32 c[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { c[i] = v.l[i] >> 29; v.l[i] &= Fr29::M29; }
#pragma unroll
    for (int i = 0; i < 8; ++i) v.l[i + 1] += c[i];
    return v; }
ZK_D Fr29 from_lane(const Fr29& v, int src_lane) { Fr29 r;                                      // every lane: the value lane src_lane holds
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = (u32)__builtin_amdgcn_ds_bpermute(src_lane << 2, (int)v.l[i]);
    return r; }
ZK_D void wave_sync() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
ZK_D void put(u32* row, const Fr29& v) {
#pragma unroll
    for (int i = 0; i < 9; ++i) row[i] = v.l[i]; }
// lane j's entry of a row of the lane-interleaved tables (k_tables_coop): limb i at word 16 i + j
ZK_D Fr29 ldc(const u32* row, int j) { Fr29 r;
#pragma unroll
    for (int i = 0; i < 9; ++i) r.l[i] = row[16 * i + j];
    return r; }
// sum_c k[c] * x[c], c < n: constants from column rows c of the tables (this lane's entry), operands from the group's exchange area (LDS)
ZK_D Fr29 dot(const u32* kcols, int j, const u32* xs, int n) {
    Fr29 acc = Fr29::zero();
    int c = 0;
    for (; c + 1 < n; c += 2) acc = Fr29::reduce32(Fr29::add_l(acc, Fr29::mul2(ldc(kcols + 144 * c, j), ld29(xs + 9 * c), ldc(kcols + 144 * (c + 1), j), ld29(xs + 9 * (c + 1)))));
    if (c < n) acc = Fr29::reduce32(Fr29::add_l(acc, Fr29::mul(ldc(kcols + 144 * c, j), ld29(xs + 9 * c))));
    return acc;
}
}  // namespace coop
#endif
#if defined(__HIP_DEVICE_COMPILE__)
namespace coop {
// where the S-box wires of a traced call go; a != nullptr: the call's constraint rows (three per S-box from `row`: u u = x^2, x^2 x^2 = x^4,
// x^4 u = x^5) get their a, b, c written here as well — the S-box input u is in a register, evaluating its 14- to 79-term expression again
// for the quotient is two thirds of k_r1cs_eval's terms
struct Tr { Fr* w; u32 first, sbase, n_stash, stash_first; u32* ts; Fr* a; Fr* b; Fr* c; u32 row; };
ZK_D u32 max4(u32 v, int lane) {                                        // the maximum over the wave's four groups
    v = max(v, (u32)__builtin_amdgcn_ds_bpermute((lane ^ 16) << 2, (int)v));
    return max(v, (u32)__builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, (int)v));
}
// stashed partial-round values -> their wire ids (and rows), one conversion per lane.  Per S-box 3 values (x^2, x^4, x^5), or 4 (u first) with rows.
ZK_D void flush(Tr& tr, int j) {
    wave_sync();
    if (tr.a) {
        if ((u32)j < 4u * tr.n_stash) {
            const Fr v = Fr29::to32_div32(ld29(tr.ts + 9 * j));
            const u32 sb = tr.stash_first + ((u32)j >> 2), comp = (u32)j & 3u;
            const size_t r = (size_t)tr.row + 3u * sb;
            if (comp == 0) { tr.a[r] = v; tr.b[r] = v; tr.b[r + 2] = v; }
            else {
                tr.w[tr.first + 3u * sb + comp - 1u] = v;
                if (comp == 1) { tr.c[r] = v; tr.a[r + 1] = v; tr.b[r + 1] = v; }
                else if (comp == 2) { tr.c[r + 1] = v; tr.a[r + 2] = v; }
                else tr.c[r + 2] = v;
            }
        }
    } else if ((u32)j < 3u * tr.n_stash) tr.w[tr.first + 3u * tr.stash_first + (u32)j] = Fr29::to32_div32(ld29(tr.ts + 9 * j));
    wave_sync();
    tr.n_stash = 0;
}
// the S-box input u parked in its raw 9 x 29-bit form in the first 36 bytes of the S-box's own three wire slots (TR == 2): k_sbox_expand recomputes
// x^2, x^4, x^5 from it, converts and writes wires and rows with every lane busy — the serial wave keeps no conversion (a product and a
// canonicalisation each) and no stash on its critical path
ZK_D void park(Fr* slot, const Fr29& u) {
    u32* o = (u32*)slot;
    *(uint4*)o = make_uint4(u.l[0], u.l[1], u.l[2], u.l[3]);
    *(uint4*)(o + 4) = make_uint4(u.l[4], u.l[5], u.l[6], u.l[7]);
    o[8] = u.l[8];
}
// ONE permutation of width t on the group's lanes (lane j < t holds element j in st; act = this group has a block — idle groups of the wave
// run along on a valid dummy width so that the wave stays converged for the DPP / LDS steps of the live ones).  TR: the S-box wires go out —
// 1: converted and stored here, 2: the S-box inputs parked raw for k_sbox_expand.
template <int TR>
ZK_D void permute(Fr29& st, bool act, int t, int lane, u32* xs, const PosDev& D, Tr& tr) {
    const int j = lane & 15, base_lane = lane & ~15;
    const bool mine = act && j < t;           // this lane holds a state element
    const int tt = act ? t : 2;
    const int rp = D.rp[tt];
    const u32* rows = D.tabc + D.c_off[tt];   // lane-interleaved rows of this width (k_tables_coop): lanes j >= tt read lane 0's entries
    const u32* prow = rows + 144 * 8;         // three rows per partial round
    const u32* mm = prow + 144 * 3 * rp;      // columns of the round matrix
    const u32* post = mm + 144 * tt;          // columns of the dense block
    for (int half = 0; half < 2; ++half) {
        for (int rr = 0; rr < POS_RF / 2; ++rr) {          // full round: S-boxes side by side, one matrix row per lane
            const Fr29 u = Fr29::reduce32(Fr29::add_l(st, ldc(rows + 144 * (half * (POS_RF / 2) + rr), j)));
            const Fr29 x2 = Fr29::sqr(u), x4 = Fr29::sqr(x2), x5 = Fr29::mul(x4, u);
            if (TR == 2) {
                if (mine) park(tr.w + tr.first + 3u * (tr.sbase + (u32)j), u);
                if (act) tr.sbase += (u32)t;
            } else if (TR) {
                if (mine) {
                    Fr* o = tr.w + tr.first + 3u * (tr.sbase + (u32)j);
                    const Fr X2 = Fr29::to32_div32(x2), X4 = Fr29::to32_div32(x4), X5 = Fr29::to32_div32(x5);
                    o[0] = X2; o[1] = X4; o[2] = X5;
                    if (tr.a) {
                        const Fr U = Fr29::to32_div32(u);
                        const size_t r0 = (size_t)tr.row + 3u * (tr.sbase + (u32)j);
                        tr.a[r0] = U; tr.b[r0] = U; tr.c[r0] = X2;
                        tr.a[r0 + 1] = X2; tr.b[r0 + 1] = X2; tr.c[r0 + 1] = X4;
                        tr.a[r0 + 2] = X4; tr.b[r0 + 2] = U; tr.c[r0 + 2] = X5;
                    }
                }
                if (act) tr.sbase += (u32)t;
            }
            put(xs + 9 * j, x5);
            wave_sync();
            st = dot(mm, j, xs, tt);
            wave_sync();
        }
        if (half) break;
        for (int i = 0; i < rp; ++i) {                       // sparse partial round: 4 products deep
            const u32* pr = prow + 144 * 3 * i;
            const Fr29 s_j = Fr29::reduce32(Fr29::add_l(st, ldc(pr, j)));                    // lane 0: u = st0 + k0
            const Fr29 ka = ldc(pr + 144, j);                                                // lane 0: m00 (unused here), lane j: v_j
            const Fr29 p1 = Fr29::mul(j == 0 ? s_j : ka, s_j);                               // lane 0: x^2; lane j: v_j * s_j
            const Fr29 x4 = Fr29::sqr(p1);
            const Fr29 x5 = Fr29::mul(x4, s_j);                                              // lane 0 only is meaningful
            if (TR == 2) {
                if (j == 0 && act) park(tr.w + tr.first + 3u * tr.sbase, s_j);
                if (act) ++tr.sbase;
            } else if (TR && j == 0 && act) {
                if (tr.a) { u32* q = tr.ts + 9 * (4 * tr.n_stash); put(q, s_j); put(q + 9, p1); put(q + 18, x4); put(q + 27, x5); }
                else { u32* q = tr.ts + 9 * (3 * tr.n_stash); put(q, p1); put(q + 9, x4); put(q + 18, x5); }
            }
            const Fr29 X = from_lane(x5, base_lane);
            const Fr29 kb = ldc(pr + 288, j);                                                // lane 0: m00; lane j: what_j
            const Fr29 e = Fr29::mul(kb, X);
            Fr29 dterm = p1;
            if (j == 0 || j >= tt) dterm = Fr29::zero();
            const Fr29 Dsum = row_sum(dterm);                                                // lane 0: sum_j v_j s_j
            st = j == 0 ? Fr29::reduce32(Fr29::add_l(e, Dsum)) : Fr29::add_l(s_j, e);          // loose on lanes j: reduced at the next constant add
            if (TR == 1) {
                if (tr.n_stash == 0) tr.stash_first = tr.sbase;
                if (act) { ++tr.n_stash; ++tr.sbase; }
                if (max4(tr.n_stash, lane) == (tr.a ? 4u : 5u)) flush(tr, j);   // the wave's groups stash in lockstep only when they run the same width: flush on the fullest
            }
        }
        {   // the dense block left over by the optimised rounds: st[1..t-1] = post * st[1..t-1]
            if (TR == 1) { if (max4(tr.n_stash, lane)) flush(tr, j); }
            const Fr29 sj = Fr29::reduce32(st);
            put(xs + 9 * j, sj);
            wave_sync();
            const Fr29 nv = dot(post, j, xs + 9, tt - 1);
            if (j) st = nv;
            wave_sync();
        }
    }
}
}  // namespace coop
#endif
// one wave per workgroup, four calls per wave.  TR 1: wires (and rows) converted and written by the wave itself; 2: S-box inputs parked, k_sbox_expand behind it
template <int TR>
__global__ __launch_bounds__(64) void k_gadget_poseidon_coop(SolverProg P, const u32* __restrict__ instr, u32 n, Fr* w, uint8_t* known, u32* err, PosDev D,
                                                            const Fr* __restrict__ pre, const u32* __restrict__ pre_off, Fr* ra, Fr* rb, Fr* rc, int urgent) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace coop;
    // a launch of a handful of calls is somebody's critical path (the challenge sponge beside the Merkle levels, the CEX chains under the prove
    // tail): its waves take the issue slots of their SIMD first — measured without it: 40 ms alone, 51 ms beside other levels
    if (urgent) __builtin_amdgcn_s_setprio(3);
    __shared__ u32 xch[4 * XS], stash[4 * XS];
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, base_lane = lane & ~15;
    const u32 call = blockIdx.x * 4u + g;
    const bool live = call < n && !err[0];
    u32* xs = xch + g * XS;
    const u32 ins = live ? instr[call] : 0u;
    const u32* cd = P.calldata + (live ? P.arg[ins] : 0u);
    const u32 n_in = live ? cd[0] : 0u, n_out = live ? cd[2] : 0u, carry_lane = live ? ((cd[3] >> 8) & 0xffu) : 0u;
    const u32 pre_at = (live && pre_off) ? pre_off[call] : 0xffffffffu;   // first element of this call's inputs in `pre`, or none
    u64 p = SI_POSEIDON_HDR;                     // every lane walks the call data (the expressions have no index)
    Fr29 st = Fr29::zero();                      // lane 0: the capacity element carried from block to block
    u32 done = 0;                                // inputs absorbed
    const bool rows = ra != nullptr && live && cd[4] != 0xffffffffu;   // all groups of the wave agree when ra is null; a call without known rows stashes 3 per S-box
    Tr tr{w, live ? cd[1] : 0u, 0u, 0u, 0u, stash + g * XS, rows ? ra : nullptr, rb, rc, live ? cd[4] : 0u};
    // lanes of dead groups run along with n_in = 0 — the loop bound is the longest call of the wave's four groups
    const u32 max_in = max4(n_in, lane);
    for (u32 blk = 0; blk * 12u < max_in; ++blk) {
        const bool act = done < n_in;             // this group still has a block to absorb
        const u32 k = act ? (n_in - done < 12u ? n_in - done : 12u) : 0u;
        // absorb: lane 1 + i takes input done + i — from the pre-evaluated inputs of a long call, else the lane evaluates its own expression
        // (every lane walks the term counts: the expressions carry no index)
        int bad = 0;
        if (pre_at != 0xffffffffu) {
            if (j >= 1 && (u32)j <= k) st = Fr29::from32<5>(pre[pre_at + done + (u32)j - 1u]);
        } else {
            for (u32 i = 0; i < k; ++i) {
                if ((u32)j == i + 1u) {
                    Fr v;
                    u64 q = p;
                    bad = si_eval_le(P, cd, q, w, known, &v);
                    st = Fr29::from32<5>(v);
                }
                p += 1u + 2u * (u64)cd[p];
            }
        }
        if (bad) { if (atomicCAS(&err[0], 0u, (u32)bad) == 0u) err[1] = ins; }
        if (!act) { if (j) st = Fr29::zero(); }
        permute<TR>(st, act, (int)k + 1, lane, xs, D, tr);
        // the next block's capacity element sits on lane 0
        const Fr29 cap = from_lane(st, base_lane + (int)carry_lane);
        if (j == 0) st = cap;
        if (act) done += k;
    }
    if (live) for (u32 q = (u32)j; q < n_out; q += (u32)G) known[tr.first + q] = 1;
#endif
}

// behind k_gadget_poseidon_coop<2>: a thread per S-box of a call (blockIdx.y) — the parked input u back from the S-box's own wire slots, x^2, x^4, x^5
// from it, the three wires in memory form and, with ra, the S-box's three rows (u u = x^2, x^2 x^2 = x^4, x^4 u = x^5).  The same field elements the
// serial wave went on with, so the same canonical words as the wave's own conversions (tests/test_circuit_gpu.py runs both).
__global__ __launch_bounds__(128) void k_sbox_expand(SolverProg P, const u32* __restrict__ instr, Fr* w, const u32* __restrict__ err, PosDev D, Fr* ra, Fr* rb, Fr* rc) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (err[0]) return;                       // the wave parked nothing (or the solve is lost anyway)
    const u32* cd = P.calldata + P.arg[instr[blockIdx.y]];
    const u32 n_in = cd[0], first = cd[1], row = cd[4];
    const u32 full = n_in / 12u, rem = n_in % 12u;
    const u32 n_sbox = full * (u32)(POS_RF * 13 + D.rp[13]) + (rem ? (u32)(POS_RF * (int)(rem + 1u) + D.rp[rem + 1u]) : 0u);
    const bool rows = ra != nullptr && row != 0xffffffffu;
    for (u32 sb = blockIdx.x * 128u + threadIdx.x; sb < n_sbox; sb += gridDim.x * 128u) {
        Fr* o = w + first + 3u * sb;
        const uint4 lo = *(const uint4*)o, hi = *((const uint4*)o + 1);
        Fr29 u;
        u.l[0] = lo.x; u.l[1] = lo.y; u.l[2] = lo.z; u.l[3] = lo.w; u.l[4] = hi.x; u.l[5] = hi.y; u.l[6] = hi.z; u.l[7] = hi.w; u.l[8] = ((const u32*)o)[8];
        const Fr29 x2 = Fr29::sqr(u), x4 = Fr29::sqr(x2), x5 = Fr29::mul(x4, u);
        const Fr X2 = Fr29::to32_div32(x2), X4 = Fr29::to32_div32(x4), X5 = Fr29::to32_div32(x5);
        o[0] = X2; o[1] = X4; o[2] = X5;
        if (rows) {
            const Fr U = Fr29::to32_div32(u);
            const size_t r0 = (size_t)row + 3u * sb;
            ra[r0] = U; rb[r0] = U; rc[r0] = X2;
            ra[r0 + 1] = X2; rb[r0 + 1] = X2; rc[r0 + 1] = X4;
            ra[r0 + 2] = X4; rb[r0 + 2] = U; rc[r0 + 2] = X5;
        }
    }
#endif
}

__global__ __launch_bounds__(64, 2) void k_gadget_poseidon(SolverProg P, const u32* __restrict__ instr, u32 n, Fr* w, uint8_t* known, u32* err, PosDev D,
                                                          const Fr* __restrict__ pre, const u32* __restrict__ pre_off) {
    const u32 i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n || err[0]) return;
#if defined(__HIP_DEVICE_COMPILE__)
    const u32 ins = instr[i];
    const u32* cd = P.calldata + P.arg[ins];
    const u32 n_in = cd[0], first = cd[1], n_out = cd[2], out_carry = (cd[3] >> 8) & 0xffu;
    u64 p = SI_POSEIDON_HDR;
    Fr st[POS_MAX_T];
    Fr cap = Fr::zero();
    TraceSink ts{w + first, 1, 0, 0u};
    u32 done = 0;
    while (done < n_in) {
        const u32 k = n_in - done < 12u ? n_in - done : 12u;
        const int t = (int)k + 1;
        st[0] = cap;
        const u32 pre_at = pre_off ? pre_off[i] : 0xffffffffu;
        for (u32 j = 0; j < k; ++j) {
            if (pre_at != 0xffffffffu) { st[1 + j] = pre[pre_at + done + j]; continue; }
            const int rc = si_eval_le(P, cd, p, w, known, &st[1 + j]);
            if (rc) { if (atomicCAS(&err[0], 0u, (u32)rc) == 0u) err[1] = ins; return; }
        }
        if (t == 13) trace_block29<13>(st, D, &ts);
        else if (t == 3) trace_block29<3>(st, D, &ts);
        else if (t == 5) trace_block29<5>(st, D, &ts);
        else if (t == 6) trace_block29<6>(st, D, &ts);
        else permute_plain_trace(st, t, D.tabs(t), D.rp[t], ts);
        cap = st[out_carry];
        done += k;
    }
    for (u32 j = 0; j < n_out; ++j) known[first + j] = 1;
#endif
}

// range-check limbs (gnark std/rangecheck with a commitment: every checked value is cut into 16-bit limbs, each limb is a committed
// wire and one query of the 2^16-entry table of the log-derivative argument): limbs[l * n + i] = limb l of value i as a Montgomery Fr,
// multiplicity[limb] += 1 (the m_i wires of the argument).  Values at or above 2^(16 nb_limbs) are counted in *bad.
__global__ __launch_bounds__(256) void k_witgen_limbs(const Fr* __restrict__ values, size_t n, int nb_limbs, Fr* __restrict__ limbs, u32* __restrict__ mult, u32* __restrict__ bad) {
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    Fr c = Fr::from_mont(values[i]);
    bool over = false;
    for (int l = 0; l < 16; ++l) {
        u32 d = (c.v[l >> 1] >> ((l & 1) * 16)) & 0xffffu;
        if (l < nb_limbs) {
            Fr x = Fr::zero();
            x.v[0] = d;
            limbs[(size_t)l * n + i] = Fr::to_mont(x);
            atomicAdd(mult + d, 1u);
        } else if (d) over = true;
    }
    if (over) atomicAdd(bad, 1u);
}
// log-derivative inverse wires: out[i] = 1 / (challenge - v[i]); chunks of 8 per thread share one inversion (Montgomery's trick).
// A zero denominator (a value equal to the challenge: probability 2^-254 for an honest challenge) gives 0 and is counted.
__global__ __launch_bounds__(128) void k_witgen_inverse(const Fr* __restrict__ values, size_t n, Fr challenge, Fr* __restrict__ out, u32* __restrict__ bad) {
    const size_t t = (size_t)blockIdx.x * 128u + threadIdx.x;
    const size_t lo = t * 8u;
    if (lo >= n) return;
    const int m = (int)((n - lo) < 8 ? (n - lo) : 8);
    Fr d[8], pre[8];
    Fr run = Fr::one();
    u32 zeros = 0;
    for (int j = 0; j < m; ++j) {
        d[j] = Fr::sub(challenge, values[lo + j]);
        if (d[j].is_zero()) { d[j] = Fr::one(); zeros |= 1u << j; }
        pre[j] = run;
        run = Fr::mul(run, d[j]);
    }
    Fr inv = Fr::inv(run);
    for (int j = m - 1; j >= 0; --j) {
        Fr r = Fr::mul(inv, pre[j]);
        inv = Fr::mul(inv, d[j]);
        out[lo + j] = ((zeros >> j) & 1u) ? Fr::zero() : r;
    }
    if (zeros) atomicAdd(bad, (u32)__popc(zeros));
}
// bit decompositions (api.ToBinary / the comparison gadgets: std/math/bits NBits hint): bits[b * n + i] = bit b of value i as a Montgomery Fr
// (one or zero); a value at or above 2^nbits is counted in *bad (its recomposition constraint could not hold)
__global__ __launch_bounds__(256) void k_witgen_bits(const Fr* __restrict__ values, size_t n, int nbits, Fr* __restrict__ bits, u32* __restrict__ bad) {
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const Fr c = Fr::from_mont(values[i]);
    const Fr one = Fr::one(), zero = Fr::zero();
    bool over = false;
    for (int b = 0; b < 256; ++b) {
        const bool set = (c.v[b >> 5] >> (b & 31)) & 1u;
        if (b < nbits) bits[(size_t)b * n + i] = set ? one : zero;
        else over |= set;
    }
    if (over) atomicAdd(bad, 1u);
}
// lookup results (logderivlookup.Table.Lookup: circuit/utils.go:137, circuit/batch_create_user_circuit.go:184-195,292): out[i] = table[index_i],
// the index a field element (the query wire); an index outside the table is counted in *bad and yields zero
__global__ __launch_bounds__(256) void k_witgen_gather(const Fr* __restrict__ table, u32 table_len, const Fr* __restrict__ indices, size_t n,
                                                       Fr* __restrict__ out, u32* __restrict__ bad) {
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const Fr c = Fr::from_mont(indices[i]);
    const bool ok = !(c.v[1] | c.v[2] | c.v[3] | c.v[4] | c.v[5] | c.v[6] | c.v[7]) && c.v[0] < table_len;
    out[i] = ok ? table[c.v[0]] : Fr::zero();
    if (!ok) atomicAdd(bad, 1u);
}
// circuit.IntegerDivision (circuit/utils.go:103-110) as the circuit calls it (checkAndGetIntegerDivisionRes, :166-177): the divisor is a small
// constant (utils.PercentageMultiplier = 100): q[i], rem[i] = DivMod(value_i, divisor) by long division over the eight 32-bit words
__global__ __launch_bounds__(256) void k_witgen_divmod_small(const Fr* __restrict__ values, size_t n, u32 divisor, Fr* __restrict__ q, Fr* __restrict__ rem) {
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const Fr c = Fr::from_mont(values[i]);
    Fr quo = Fr::zero(), r = Fr::zero();
    u64 carry = 0;
#pragma unroll
    for (int k = 7; k >= 0; --k) {
        const u64 cur = (carry << 32) | c.v[k];
        quo.v[k] = (u32)(cur / divisor);
        carry = cur % divisor;
    }
    r.v[0] = (u32)carry;
    q[i] = Fr::to_mont(quo);
    rem[i] = Fr::to_mont(r);
}
// w[wire_ids[i]] = src[i]: the slots of a generator land on the wire ids of the compiled circuit (the map is part of the solver export)
__global__ __launch_bounds__(256) void k_witgen_scatter(Fr* __restrict__ w, const Fr* __restrict__ src, const u32* __restrict__ wire_ids, size_t n) {
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) w[wire_ids[i]] = src[i];
}

// the same, and the wires are marked assigned for the solver program that runs next (zkpor_solver_start_dev's d_known)
__global__ __launch_bounds__(256) void k_witgen_scatter_known(Fr* __restrict__ w, uint8_t* __restrict__ known, const Fr* __restrict__ src,
                                                              const u32* __restrict__ wire_ids, size_t n) {
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n) { const u32 id = wire_ids[i]; w[id] = src[i]; known[id] = 1; }
}

struct AccountHdr {
    uint8_t id_be[32];
    u64 equity[2], debt[2], collateral[2];
    u32 n_assets, asset_off;
};
struct AssetRec { u64 equity, debt, loan, margin, portfolio_margin; u32 index, pad; };
static_assert(sizeof(AccountHdr) == 88 && sizeof(AssetRec) == 48, "zkpor_account_t / zkpor_asset_t layout");

ZK_HD Fr pack3(u64 a, u64 b, u64 c) {  // a*2^128 + b*2^64 + c  (< 2^192 < r)
    Fr x;
    x.v[0] = (u32)c; x.v[1] = (u32)(c >> 32); x.v[2] = (u32)b; x.v[3] = (u32)(b >> 32);
    x.v[4] = (u32)a; x.v[5] = (u32)(a >> 32); x.v[6] = 0; x.v[7] = 0;
    return Fr::to_mont(x);
}
ZK_HD Fr from_u128(const u64* w) {
    Fr x;
    x.v[0] = (u32)w[0]; x.v[1] = (u32)(w[0] >> 32); x.v[2] = (u32)w[1]; x.v[3] = (u32)(w[1] >> 32);
    x.v[4] = x.v[5] = x.v[6] = x.v[7] = 0;
    return Fr::to_mont(x);
}
ZK_HD Fr from_be32(const uint8_t* b) {  // 32 big-endian bytes, reduced mod r
    Fr x;
    for (int i = 0; i < 8; ++i) {
        const uint8_t* p = b + 28 - 4 * i;
        x.v[i] = ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3];
    }
    for (int k = 0; k < 6; ++k) {  // 2^256 / r < 6
        u32 t[8];
        u32 bw = 0;
        for (int i = 0; i < 8; ++i) { u64 d = (u64)x.v[i] - FrParams::mod(i) - bw; t[i] = (u32)d; bw = (u32)(d >> 32) & 1u; }
        if (bw) break;
        for (int i = 0; i < 8; ++i) x.v[i] = t[i];
    }
    return Fr::to_mont(x);
}

// one thread per account: streams the tier-padded asset list (PaddingAccountAssets, utils.go:147-186) through the
// sponge as 2 field elements per slot, then the 5-input leaf hash
__global__ __launch_bounds__(64, 2) void k_account_leaves(const AccountHdr* __restrict__ acc, const AssetRec* __restrict__ assets,
                                                       u32 n, int tier, Fr* __restrict__ out, PosDev P) {
    u32 i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    const AccountHdr a = acc[i];
    const AssetRec* as = assets + a.asset_off;
    Sponge sp;
    sp.init();
    int padding = tier - (int)a.n_assets, cur_pad = 0;
    u32 cur_idx = 0, ai = 0;
    for (int slot = 0; slot < tier; ++slot) {
        bool real;
        if (ai < a.n_assets) real = !(cur_pad < padding && cur_idx < as[ai].index);
        else real = false;
        if (real) {
            const AssetRec r = as[ai];
            sp.push(P, pack3(r.index, r.equity, r.debt));
            sp.push(P, pack3(r.loan, r.margin, r.portfolio_margin));
            cur_idx = r.index + 1;
            ++ai;
        } else {
            sp.push(P, pack3(cur_idx, 0, 0));
            sp.push(P, Fr::zero());
            ++cur_idx;
            ++cur_pad;
        }
    }
    Fr commitment = sp.finish(P);
    sp.init();
    sp.push(P, from_be32(a.id_be));
    sp.push(P, from_u128(a.equity));
    sp.push(P, from_u128(a.debt));
    sp.push(P, from_u128(a.collateral));
    sp.push(P, commitment);
    out[i] = sp.finish(P);
}

// The same hash, one account per 16 LANES (coop::permute): a tier-500 account is a serial chain of 84 + 1 permutations, and below ~10^5 accounts
// a launch lasts as long as ONE chain — 16 k accounts took 0.28 s with one thread each (58 k accounts/s, 256 waves on 1 024 SIMDs).  Every lane
// walks the padding rule to the slot of its own element (integer work), the permutations run across the lanes.
__global__ __launch_bounds__(64) void k_account_leaves_coop(const AccountHdr* __restrict__ acc, const AssetRec* __restrict__ assets,
                                                            u32 n, int tier, Fr* __restrict__ out, PosDev D) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace coop;
    __shared__ u32 xch[4 * XS];
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, base_lane = lane & ~15;
    const u32 i = blockIdx.x * 4u + g;
    const bool live = i < n;
    u32* xs = xch + g * XS;
    AccountHdr a;
    if (live) a = acc[i]; else { a.n_assets = 0; a.asset_off = 0; }
    const AssetRec* as = assets + a.asset_off;
    const u32 n_in = live ? 2u * (u32)tier : 0u;
    // this lane's cursor over the padded slots (PaddingAccountAssets, utils.go:147-186): state after `pos` consumed slots
    int padding = tier - (int)a.n_assets, cur_pad = 0;
    u32 cur_idx = 0, ai = 0, pos = 0;
    bool slot_real = false; u32 slot_ai = 0, slot_pad_idx = 0;
    auto consume_to = [&](u32 slot) {             // consume slots pos .. slot: afterwards slot_* describe `slot`
        while (pos <= slot) {
            bool real = ai < a.n_assets ? !(cur_pad < padding && cur_idx < as[ai].index) : false;
            slot_real = real;
            if (real) { slot_ai = ai; cur_idx = as[ai].index + 1; ++ai; }
            else { slot_pad_idx = cur_idx; ++cur_idx; ++cur_pad; }
            ++pos;
        }
    };
    Fr29 st = Fr29::zero(), last = Fr29::zero();
    u32 done = 0;
    Tr tr{nullptr, 0u, 0u, 0u, 0u, nullptr};
    const u32 max_in = max4(n_in, lane);
    for (u32 blk = 0; blk * 12u < max_in; ++blk) {
        const bool act = done < n_in;
        const u32 k = act ? (n_in - done < 12u ? n_in - done : 12u) : 0u;
        if (act && j >= 1 && (u32)j <= k) {
            const u32 e = done + (u32)j - 1u;
            consume_to(e >> 1);
            Fr v;
            if (slot_real) { const AssetRec r = as[slot_ai]; v = (e & 1u) ? pack3(r.loan, r.margin, r.portfolio_margin) : pack3(r.index, r.equity, r.debt); }
            else v = (e & 1u) ? Fr::zero() : pack3(slot_pad_idx, 0, 0);
            st = Fr29::from32<5>(v);
        }
        if (!act) { if (j) st = Fr29::zero(); }
        permute<false>(st, act, (int)k + 1, lane, xs, D, tr);
        const Fr29 o = from_lane(st, base_lane + D.out_idx);
        if (act) last = o;
        const Fr29 cap = from_lane(st, base_lane + D.carry_idx);
        if (j == 0) st = cap;
        if (act) done += k;
    }
    // the leaf: Poseidon(id, equity, debt, collateral, commitment) — a fresh sponge, one block of width 6
    st = Fr29::zero();
    if (live) {
        if (j == 1) st = Fr29::from32<5>(from_be32(a.id_be));
        else if (j == 2) st = Fr29::from32<5>(from_u128(a.equity));
        else if (j == 3) st = Fr29::from32<5>(from_u128(a.debt));
        else if (j == 4) st = Fr29::from32<5>(from_u128(a.collateral));
        else if (j == 5) st = last;
    }
    permute<false>(st, live, 6, lane, xs, D, tr);
    const Fr29 leaf = from_lane(st, base_lane + D.out_idx);
    if (live && j == 0) out[i] = Fr29::to32_div32(leaf);
#endif
}
// below this many hash chains a launch is latency-bound and the chains run 16 lanes each ("poseidon_coop": -1 = this rule, 0 = never, 1 = always)
static bool use_coop(const zkpor_ctx* ctx, size_t n) { return ctx->poseidon_coop < 0 ? n < 65536 : ctx->poseidon_coop != 0; }
static void launch_account_leaves(zkpor_ctx* ctx, const AccountHdr* dacc, const AssetRec* das, u32 n, int tier, Fr* dout, const PosDev& P) {
    if (use_coop(ctx, n)) hipLaunchKernelGGL(k_account_leaves_coop, dim3((n + 3u) / 4u), dim3(64), 0, ctx->stream, dacc, das, n, tier, dout, P);
    else hipLaunchKernelGGL(k_account_leaves, dim3((n + 63u) / 64u), dim3(64), 0, ctx->stream, dacc, das, n, tier, dout, P);
}

// Montgomery Fr <-> 32-byte big-endian
__global__ void k_fr_to_be(const Fr* __restrict__ in, uint8_t* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr c = Fr::from_mont(in[i]);
    u32* o = (u32*)(out + 32 * i);
    for (int j = 0; j < 8; ++j) o[j] = __builtin_bswap32(c.v[7 - j]);
}
__global__ void k_fr_from_be(const uint8_t* __restrict__ in, Fr* __restrict__ out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    out[i] = from_be32(in + 32 * i);
}

static int32_t pos_dev(zkpor_ctx* ctx, PosDev* P) {
    PosTables* T;
    ZK_TRY(pos_tables_get(ctx, &T));
    P->tab = T->dev;
    P->tab29 = T->dev29;
    P->tabc = T->devc;
    for (int t = 2; t <= POS_MAX_T; ++t) P->c_off[t] = T->c_off[t];
    for (int t = 2; t <= POS_MAX_T; ++t) {
        P->rc_off[t] = T->rc_off[t]; P->mds_off[t] = T->mds_off[t]; P->prc_off[t] = T->prc_off[t];
        P->sp_off[t] = T->sp_off[t]; P->post_off[t] = T->post_off[t]; P->rp[t] = pos_rp(t);
    }
    if (ctx->pos_out < 0 || ctx->pos_out > 1 || ctx->pos_carry < 0 || ctx->pos_carry > 1) { ctx->err = "poseidon: convention indices must be 0 or 1"; return ZKPOR_E_ARG; }
    P->out_idx = ctx->pos_out; P->carry_idx = ctx->pos_carry;
    return ZKPOR_OK;
}

// csrc/solver.hip: n Poseidon instructions (ids in d_instr) of one level, one thread each, on `stream`
// d_pre / d_pre_off (may be NULL): inputs evaluated beforehand, pre_off[i] = first element of call i's inputs (0xffffffff: evaluate in the kernel).
// d_a / d_b / d_c (may be NULL; cooperative kernel only): the calls' constraint rows are written too.
int32_t gadget_poseidon_launch(zkpor_ctx* ctx, hipStream_t stream, const SolverProg& P, const u32* d_instr, u32 n, Fr* w, uint8_t* known, u32* d_err,
                               const Fr* d_pre, const u32* d_pre_off, Fr* d_a, Fr* d_b, Fr* d_c) {
    if (n == 0) return ZKPOR_OK;
    PosDev D;
    ZK_TRY(pos_dev(ctx, &D));
    if (ctx->solver_poseidon == 0) hipLaunchKernelGGL(k_gadget_poseidon, dim3((n + 63u) / 64u), dim3(64), 0, stream, P, d_instr, n, w, known, d_err, D, d_pre, d_pre_off);
    else if ((int64_t)n <= ctx->poseidon_defer) {
        // a launch this narrow is somebody's critical path (the challenge sponge: one call of 116 permutations; the CEX chains: two of 834; a Merkle level):
        // the waves park their S-box inputs and a wide kernel behind them does the conversions ("poseidon_defer": calls per launch up to which)
        hipLaunchKernelGGL(k_gadget_poseidon_coop<2>, dim3((n + 3u) / 4u), dim3(64), 0, stream, P, d_instr, n, w, known, d_err, D, d_pre, d_pre_off, d_a, d_b, d_c, n <= 64u ? 1 : 0);
        hipLaunchKernelGGL(k_sbox_expand, dim3(n <= 8u ? 256u : n <= 512u ? 16u : 1u, n), dim3(128), 0, stream, P, d_instr, w, (const u32*)d_err, D, d_a, d_b, d_c);
    } else hipLaunchKernelGGL(k_gadget_poseidon_coop<1>, dim3((n + 3u) / 4u), dim3(64), 0, stream, P, d_instr, n, w, known, d_err, D, d_pre, d_pre_off, d_a, d_b, d_c, n <= 64u ? 1 : 0);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// CEX asset-list commitments and batch commitments: what Witness.Run computes per batch
// (src/witness/witness/witness.go:159-198) and utils.ComputeCexAssetsCommitment (src/utils/utils.go:779-800).
// The commitment of a CEX state is one chained Poseidon over 20 elements per asset (ConvertAssetInfoToBytes,
// utils.go:53-88): two packed totals and 3 x 6 packed tier-ratio pairs (ConvertTierRatiosToBytes, utils.go:26-51).
// The reference computes it twice per batch on one core (running totals); once the running totals of all batches are
// known (a prefix sum over the user operations) every batch's commitments are independent, so they are hashed here one
// state per thread.  Tier ratios and prices do not change between batches: they are packed once.
struct TierRatioRec { u64 boundary[2]; uint8_t ratio; uint8_t pad[7]; };                      // zkpor_tier_ratio_t
struct CexAssetConst { u64 base_price; TierRatioRec loan[12], margin[12], portfolio_margin[12]; };  // zkpor_cex_asset_const_t
struct CexTotals { u64 total_equity, total_debt, loan_collateral, margin_collateral, portfolio_margin_collateral; };
static_assert(sizeof(TierRatioRec) == 24 && sizeof(CexAssetConst) == 872 && sizeof(CexTotals) == 40, "zkpor_cex_* layout");

// ratio_i + boundary_i * 2^8 + ratio_{i+1} * 2^126 + boundary_{i+1} * 2^134 as a 256-bit integer (utils.go:33-47: plain
// big-integer sums; a boundary of exactly 2^118, the padding value, carries into the next field exactly as there)
ZK_D void add_shifted(u32* acc, const u64* v128, int shift) {  // acc += v128 << shift, 8 x 32-bit limbs
    u32 w[4] = {(u32)v128[0], (u32)(v128[0] >> 32), (u32)v128[1], (u32)(v128[1] >> 32)};
    u64 carry = 0;
    const int ws = shift >> 5, bs = shift & 31;
    u32 prev = 0;
    for (int i = 0; i + ws < 8; ++i) {
        u32 cur = i < 4 ? w[i] : 0u;
        u32 part = bs ? ((cur << bs) | (prev >> (32 - bs))) : cur;
        prev = cur;
        u64 t = (u64)acc[i + ws] + part + carry;
        acc[i + ws] = (u32)t;
        carry = t >> 32;
        if (i >= 5 && !carry && !prev) break;
    }
}
__global__ void k_cex_tier_elems(const CexAssetConst* __restrict__ consts, u32 n_assets, Fr* __restrict__ out /* n_assets x 18 */) {
    u32 g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_assets * 18u) return;
    const u32 a = g / 18u, e = g % 18u;
    const TierRatioRec* tr = e < 6 ? consts[a].loan : (e < 12 ? consts[a].margin : consts[a].portfolio_margin);
    const TierRatioRec& lo = tr[2 * (e % 6)];
    const TierRatioRec& hi = tr[2 * (e % 6) + 1];
    Fr x = Fr::zero();
    u64 r0[2] = {lo.ratio, 0}, r1[2] = {hi.ratio, 0};
    add_shifted(x.v, r0, 0);
    add_shifted(x.v, lo.boundary, 8);
    add_shifted(x.v, r1, 126);
    add_shifted(x.v, hi.boundary, 134);
    // the sum is < 2^253 < 2r: one conditional subtraction makes it canonical (hasher.Write reduces mod r)
    u32 t[8]; u32 bw = 0;
    for (int i = 0; i < 8; ++i) { u64 d = (u64)x.v[i] - FrParams::mod(i) - bw; t[i] = (u32)d; bw = (u32)(d >> 32) & 1u; }
    if (!bw) for (int i = 0; i < 8; ++i) x.v[i] = t[i];
    out[g] = Fr::to_mont(x);
}
// one CEX state per thread: 20 elements per asset through the streaming sponge
__global__ __launch_bounds__(64, 2) void k_cex_commitments(const CexAssetConst* __restrict__ consts, const Fr* __restrict__ tier_elems,
                                                        u32 n_assets, const CexTotals* __restrict__ totals, u32 n_states,
                                                        Fr* __restrict__ out, PosDev P) {
    u32 i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n_states) return;
    Sponge sp;
    sp.init();
    const CexTotals* t = totals + (size_t)i * n_assets;
    for (u32 a = 0; a < n_assets; ++a) {
        const CexTotals c = t[a];
        sp.push(P, pack3(c.total_equity, c.total_debt, consts[a].base_price));
        sp.push(P, pack3(c.loan_collateral, c.margin_collateral, c.portfolio_margin_collateral));
        for (int e = 0; e < 18; ++e) sp.push(P, tier_elems[a * 18u + e]);
    }
    out[i] = sp.finish(P);
}
// one CEX state per 16 lanes: the commitment is ONE chain of 834 permutations (500 assets x 20 elements), and a run has few states — two per
// batch, a few thousand for a whole 10^8-account run — so the launch lasts as long as one chain: 1.28 s per thread, ~0.2 s across the lanes
__global__ __launch_bounds__(64) void k_cex_commitments_coop(const CexAssetConst* __restrict__ consts, const Fr* __restrict__ tier_elems,
                                                             u32 n_assets, const CexTotals* __restrict__ totals, u32 n_states,
                                                             Fr* __restrict__ out, PosDev D) {
#if defined(__HIP_DEVICE_COMPILE__)
    using namespace coop;
    __shared__ u32 xch[4 * XS];
    const int lane = threadIdx.x & 63, g = lane >> 4, j = lane & 15, base_lane = lane & ~15;
    const u32 i = blockIdx.x * 4u + g;
    const bool live = i < n_states;
    u32* xs = xch + g * XS;
    const CexTotals* t = totals + (size_t)(live ? i : 0u) * n_assets;
    const u32 n_in = live ? n_assets * 20u : 0u;
    Fr29 st = Fr29::zero(), last = Fr29::zero();
    u32 done = 0;
    Tr tr{nullptr, 0u, 0u, 0u, 0u, nullptr};
    const u32 max_in = max4(n_in, lane);
    for (u32 blk = 0; blk * 12u < max_in; ++blk) {
        const bool act = done < n_in;
        const u32 k = act ? (n_in - done < 12u ? n_in - done : 12u) : 0u;
        if (act && j >= 1 && (u32)j <= k) {
            const u32 e = done + (u32)j - 1u, a = e / 20u, f = e % 20u;
            Fr v;
            if (f == 0) { const CexTotals c = t[a]; v = pack3(c.total_equity, c.total_debt, consts[a].base_price); }
            else if (f == 1) { const CexTotals c = t[a]; v = pack3(c.loan_collateral, c.margin_collateral, c.portfolio_margin_collateral); }
            else v = tier_elems[a * 18u + f - 2u];
            st = Fr29::from32<5>(v);
        }
        if (!act) { if (j) st = Fr29::zero(); }
        permute<false>(st, act, (int)k + 1, lane, xs, D, tr);
        const Fr29 o = from_lane(st, base_lane + D.out_idx);
        if (act) last = o;
        const Fr29 cap = from_lane(st, base_lane + D.carry_idx);
        if (j == 0) st = cap;
        if (act) done += k;
    }
    if (live && j == 0) out[i] = Fr29::to32_div32(last);
#endif
}
// BatchCommitment = PoseidonBytes(root, before, after, min index, max index) (witness.go:185-198); an index of 0 is the
// byte string [0x00] there, i.e. the element 0 either way
__global__ __launch_bounds__(64) void k_batch_commitments(const uint8_t* __restrict__ roots, const uint8_t* __restrict__ before,
                                                          const uint8_t* __restrict__ after, const u32* __restrict__ min_idx,
                                                          const u32* __restrict__ max_idx, u32 n, Fr* __restrict__ out, PosDev P) {
    u32 i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    Sponge sp;
    sp.init();
    sp.push(P, from_be32(roots + 32 * (size_t)i));
    sp.push(P, from_be32(before + 32 * (size_t)i));
    sp.push(P, from_be32(after + 32 * (size_t)i));
    sp.push(P, pack3(0, 0, min_idx[i]));
    sp.push(P, pack3(0, 0, max_idx[i]));
    out[i] = sp.finish(P);
}

// ------------------------------------------------------------------------------------------------------------------
// Account totals and collateral tiers: TotalEquity, TotalDebt, TotalCollateral of every account —
// the three big integers that enter its leaf hash (src/utils/utils.go:608-615, CalculateAssetValueForCollateral :648-661,
// CalculateAssetValueViaTiersRatio :663-685, CalculatePrecomputedValue :420-432) — and, per asset, the tier index / flag the
// circuit witness carries (circuit/utils.go calcAndSetCollateralInfo :227-278).  One thread per account; integers are
// big.Int in the reference and fit unsigned 128-bit here (u64 balance x u64 price, boundaries <= 2^118).
typedef unsigned __int128 u128;
ZK_D u128 tr_boundary(const TierRatioRec& t) { return ((u128)t.boundary[1] << 64) | t.boundary[0]; }
// value of a collateral position through one tier list, and the (index, flag) claim for the circuit
ZK_D u128 tier_value(u128 value, const TierRatioRec* tiers, uint8_t* idx_out, uint8_t* flag_out) {
    u128 pre = 0, prev_b = 0;
    for (int i = 0; i < 12; ++i) {
        const u128 b = tr_boundary(tiers[i]);
        if (value <= b) {
            *idx_out = (uint8_t)i; *flag_out = 0;
            return pre + (value - prev_b) * tiers[i].ratio / 100;
        }
        pre += (b - prev_b) * tiers[i].ratio / 100;
        prev_b = b;
    }
    *idx_out = 11; *flag_out = 1;
    return pre;
}
__global__ __launch_bounds__(128) void k_account_totals(AccountHdr* __restrict__ acc, const AssetRec* __restrict__ assets, u32 n,
                                                        const CexAssetConst* __restrict__ cex, u32 n_cex, uint8_t* __restrict__ tier_info,
                                                        uint8_t* __restrict__ valid) {
    u32 i = blockIdx.x * 128u + threadIdx.x;
    if (i >= n) return;
    const u32 na = acc[i].n_assets, off = acc[i].asset_off;
    u128 eq = 0, debt = 0, col = 0;
    bool ok = true;
    for (u32 j = 0; j < na; ++j) {
        const AssetRec a = assets[off + j];
        if (a.index >= n_cex) { ok = false; continue; }
        const CexAssetConst& c = cex[a.index];
        const u64 s1 = a.loan + a.margin, s2 = s1 + a.portfolio_margin;
        if (s1 < a.loan || s2 < s1 || s2 > a.equity) ok = false;                 // ParseUserDataSet :599-606
        const u128 price = c.base_price;
        const u128 e = (u128)a.equity * price, d = (u128)a.debt * price;
        if (eq + e < eq || debt + d < debt) ok = false;
        eq += e; debt += d;
        uint8_t ti[6];
        const u128 v = tier_value((u128)a.loan * price, c.loan, &ti[0], &ti[1]) + tier_value((u128)a.margin * price, c.margin, &ti[2], &ti[3]) +
                       tier_value((u128)a.portfolio_margin * price, c.portfolio_margin, &ti[4], &ti[5]);
        if (col + v < col) ok = false;
        col += v;
        if (tier_info) for (int k = 0; k < 6; ++k) tier_info[6 * (size_t)(off + j) + k] = ti[k];
    }
    if (col < debt) ok = false;                                                  // :620
    acc[i].equity[0] = (u64)eq; acc[i].equity[1] = (u64)(eq >> 64);
    acc[i].debt[0] = (u64)debt; acc[i].debt[1] = (u64)(debt >> 64);
    acc[i].collateral[0] = (u64)col; acc[i].collateral[1] = (u64)(col >> 64);
    if (valid) valid[i] = ok ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------------------------
// FixedDepthMerkleTree on the device (reference src/utils/merkletree/merkletree.go:27-52): leaves and every internal
// level live in HBM in Montgomery form next to one "dirty" bitset per level; a clean position reads as nilHashes[level]
// exactly as getNodeAt (:315-331) does.  Set (:179-187) only stores leaves and marks them; Build (:192-279) recomputes
// every node that has a set leaf beneath it (the reference never clears its dirty bits either, so repeated
// Set -> Build cycles give the same tree in both).
struct TreeDev {
    Fr* nodes[33];     // nodes[0] = leaves
    u32* dirty[33];    // bitset, one bit per position of the level
    u64 cnt[33];       // positions backed by memory at each level: ceil(capacity / 2^l), up to 2^32
    const Fr* nil;     // nil[0..depth]
    int depth;
};
ZK_D bool tree_bit(const u32* bs, u64 i) { return (bs[i >> 5] >> (i & 31)) & 1u; }
ZK_D Fr tree_node(const TreeDev& T, int level, u64 pos, u64 count) {
    return (pos < count && tree_bit(T.dirty[level], pos)) ? T.nodes[level][pos] : T.nil[level];
}

// nil[l] = H(nil[l-1], nil[l-1]), l = 1..depth (merkletree.go:159-170); one thread: 32 permutations at most
__global__ void k_tree_nil_chain(Fr* nil, int depth, PosDev P) {
    if (threadIdx.x || blockIdx.x) return;
    for (int l = 1; l <= depth; ++l) {
        Fr s0 = Fr::zero(), s1 = nil[l - 1], s2 = nil[l - 1];
        permute3(s0, s1, s2, P.tabs(3));
        nil[l] = P.out_idx == 0 ? s0 : s1;
    }
}
// Set for a batch of (key, 32-byte big-endian value) pairs; later entries win for duplicate keys only by race, as in the
// reference (concurrent Set of one key is undefined there too)
__global__ void k_tree_set(TreeDev T, const u32* __restrict__ keys, const uint8_t* __restrict__ vals, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u32 k = keys[i];
    T.nodes[0][k] = from_be32(vals + 32 * i);
    atomicOr(&T.dirty[0][k >> 5], 1u << (k & 31));
}
// Set of a contiguous key range from device-resident Montgomery leaves
__global__ void k_tree_set_range(TreeDev T, u64 first, const Fr* __restrict__ vals, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    u64 k = first + i;
    T.nodes[0][k] = vals[i];
    atomicOr(&T.dirty[0][k >> 5], 1u << (k & 31));
}
// one level of Build: node p of `level` is recomputed iff child 2p or 2p+1 is dirty; its dirty bit is (re)written by a
// wave-wide ballot, so the bitset needs no clearing pass.  Block = 256 consecutive positions.
__global__ __launch_bounds__(256) void k_tree_level(TreeDev T, int level, u64 count, u64 child_count, PosDev P) {
    const u64 p = (u64)blockIdx.x * 256u + threadIdx.x;
    bool d = false;
    Fr l, r;
    if (p < count) {
        const bool dl = 2 * p < child_count && tree_bit(T.dirty[level - 1], 2 * p);
        const bool dr = 2 * p + 1 < child_count && tree_bit(T.dirty[level - 1], 2 * p + 1);
        d = dl | dr;
        if (d) {
            l = dl ? T.nodes[level - 1][2 * p] : T.nil[level - 1];
            r = dr ? T.nodes[level - 1][2 * p + 1] : T.nil[level - 1];
        }
    }
    const u64 mask = __ballot(d);
    const u64 wave_first = p & ~(u64)63;
    if ((threadIdx.x & 63u) == 0 && wave_first < count) {
        T.dirty[level][wave_first >> 5] = (u32)mask;
        if (wave_first + 32 < count) T.dirty[level][(wave_first >> 5) + 1] = (u32)(mask >> 32);
    }
    if (!d) return;
    T.nodes[level][p] = hash2_29(l, r, P);
}
// GetProof (:297-308) for n keys: out[(i * depth + level) * 32 ...] = sibling at `level`, big-endian
__global__ void k_tree_proofs(TreeDev T, const u32* __restrict__ keys, size_t n, uint8_t* __restrict__ out) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n * (size_t)T.depth) return;
    const int level = (int)(g % T.depth);
    const u64 pos = ((u64)keys[g / T.depth] >> level) ^ 1u;
    const u64 count = T.cnt[level];
    Fr c = Fr::from_mont(tree_node(T, level, pos, count));
    u32* o = (u32*)(out + 32 * g);
    for (int j = 0; j < 8; ++j) o[j] = __builtin_bswap32(c.v[7 - j]);
}
// Get (:287-294) for n keys
__global__ void k_tree_get(TreeDev T, const u32* __restrict__ keys, size_t n, uint8_t* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const u64 count = T.cnt[0];
    Fr c = Fr::from_mont(tree_node(T, 0, keys[i], count));
    u32* o = (u32*)(out + 32 * i);
    for (int j = 0; j < 8; ++j) o[j] = __builtin_bswap32(c.v[7 - j]);
}
// VerifyProof (:334-355) for n independent (key, leaf, proof) triples against one root
__global__ __launch_bounds__(64) void k_verify_proofs(const uint8_t* __restrict__ root, const u32* __restrict__ keys,
                                                      const uint8_t* __restrict__ proofs, const uint8_t* __restrict__ leaves,
                                                      size_t n, int depth, uint8_t* __restrict__ ok, PosDev P) {
    size_t i = (size_t)blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    Fr node = from_be32(leaves + 32 * i);
    const u32 key = keys[i];
    for (int l = 0; l < depth; ++l) {
        Fr sib = from_be32(proofs + 32 * (i * depth + l));
        Fr s0 = Fr::zero(), s1, s2;
        if ((key >> l) & 1u) { s1 = sib; s2 = node; } else { s1 = node; s2 = sib; }
        permute3(s0, s1, s2, P.tabs(3));
        node = P.out_idx == 0 ? s0 : s1;
    }
    Fr want = from_be32(root);
    bool eq = true;
    for (int j = 0; j < 8; ++j) eq &= node.v[j] == want.v[j];
    ok[i] = eq && (depth >= 32 || (key >> depth) == 0) ? 1 : 0;
}


}  // namespace zk

// the handle behind zkpor_tree_* (include/zkpor.h)
struct zkpor_tree {
    zkpor_ctx* ctx = nullptr;
    int depth = 0;
    uint64_t capacity = 0;
    uint64_t count[33] = {0};
    zk::TreeDev dev;
    zk::Fr* nil_dev = nullptr;
    zk::Fr nil_host[33];
    zk::Fr root;
    std::vector<void*> allocs;
};

namespace zk {

static void tree_free(zkpor_tree* t) {
    for (void* p : t->allocs) (void)hipFree(p);
    delete t;
}
template <class T>
static int32_t tree_alloc(zkpor_tree* t, size_t count, T** out, bool zero) {
    zkpor_ctx* ctx = t->ctx;
    void* p = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); ctx->err = "tree: out of device memory"; return ZKPOR_E_OOM; }
    t->allocs.push_back(p);
    if (zero) ZK_HIP(ctx, hipMemsetAsync(p, 0, bytes, ctx->stream));
    *out = (T*)p;
    return ZKPOR_OK;
}
// staging helper: copy host bytes to a temporary device buffer
struct DevTmp {
    void* p = nullptr;
    ~DevTmp() { if (p) (void)hipFree(p); }
    int32_t put(zkpor_ctx* ctx, const void* src, size_t bytes) {
        ZK_HIP(ctx, hipMalloc(&p, bytes ? bytes : 1));
        if (bytes) ZK_TRY(zk::h2d_sync(ctx, p, src, bytes));
        return ZKPOR_OK;
    }
    int32_t make(zkpor_ctx* ctx, size_t bytes) { ZK_HIP(ctx, hipMalloc(&p, bytes ? bytes : 1)); return ZKPOR_OK; }
};
static void fr_to_be_host(const Fr& m, uint8_t out[32]) {
    Fr c = Fr::from_mont(m);
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) out[31 - (i * 4 + j)] = (uint8_t)(c.v[i] >> (8 * j));
}
// tree over d_leaves[0..n) (Montgomery); d_levels (optional) receives levels 1..depth concatenated
static int32_t merkle_build_core(zkpor_ctx* ctx, const Fr* d_leaves, size_t n, int depth, const Fr& nil_leaf,
                                 Fr* d_levels, Fr* root) {
    if (depth < 1 || depth > 32 || n > ((size_t)1 << depth)) { ctx->err = "merkle: bad depth / leaf count"; return ZKPOR_E_ARG; }
    PosDev P;
    ZK_TRY(pos_dev(ctx, &P));
    // nil-subtree chain (merkletree.go:159-170), hashed on the device like everything else
    std::vector<Fr> nil(depth + 1);
    {
        DevTmp dn;
        ZK_TRY(dn.make(ctx, 33 * sizeof(Fr)));
        ZK_HIP(ctx, hipMemcpyAsync(dn.p, &nil_leaf, sizeof(Fr), hipMemcpyHostToDevice, ctx->stream));
        hipLaunchKernelGGL(k_tree_nil_chain, dim3(1), dim3(64), 0, ctx->stream, (Fr*)dn.p, depth, P);
        ZK_KERNEL_CHECK(ctx);
        ZK_HIP(ctx, hipMemcpyAsync(nil.data(), dn.p, (depth + 1) * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (n == 0) { *root = nil[depth]; return ZKPOR_OK; }
    PhaseScope ps(ctx, "poseidon_tree");
    // ping-pong buffers when the caller does not want the levels
    Fr* scratch = nullptr;
    size_t half = (n + 1) / 2;
    if (!d_levels) {
        ZK_TRY(ws_reserve(ctx, (half + (half + 1) / 2 + 64) * sizeof(Fr)));
        scratch = ws_alloc<Fr>(ctx, half + (half + 1) / 2 + 32);
        if (!scratch) { ctx->err = "merkle: workspace"; return ZKPOR_E_OOM; }
    }
    const Fr* src = d_leaves;
    size_t m = n;
    Fr* lvl_ptr = d_levels;
    Fr* pp[2] = {scratch, scratch ? scratch + half : nullptr};
    int pi = 0;
    const Fr* last = nullptr;
    for (int l = 1; l <= depth; ++l) {
        size_t mo = (m + 1) / 2;
        Fr* dst = d_levels ? lvl_ptr : pp[pi];
        hipLaunchKernelGGL(k_hash2_level, dim3((unsigned)((mo + 255) / 256)), dim3(256), 0, ctx->stream, src, (u32)m, nil[l - 1], dst, P);
        ZK_KERNEL_CHECK(ctx);
        src = dst; m = mo; last = dst;
        if (d_levels) lvl_ptr += mo; else pi ^= 1;
    }
    ZK_HIP(ctx, hipMemcpyAsync(root, last, sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKPOR_OK;
}


static size_t levels_total(size_t n, int depth) {
    size_t tot = 0, m = n;
    for (int l = 1; l <= depth; ++l) { m = (m + 1) / 2; tot += m; }
    return tot;
}
}  // namespace zk

using namespace zk;

extern "C" {

int32_t zkpor_poseidon_hash(zkpor_ctx* ctx, const uint64_t* inputs, size_t len, size_t count, uint64_t* out) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !inputs || !out || len < 1 || count == 0) return ZKPOR_E_ARG;
    PosDev P;
    ZK_TRY(pos_dev(ctx, &P));
    Fr *din = nullptr, *dout = nullptr;
    ZK_HIP(ctx, hipMalloc((void**)&din, len * count * sizeof(Fr)));
    if (hipMalloc((void**)&dout, count * sizeof(Fr)) != hipSuccess) { (void)hipFree(din); ctx->err = "out of device memory"; return ZKPOR_E_OOM; }
    int32_t rc = ZKPOR_OK;
    if (zk::h2d_sync(ctx, din, inputs, len * count * sizeof(Fr)) != ZKPOR_OK) { ctx->err = "H2D failed"; rc = ZKPOR_E_HIP; }
    if (rc == ZKPOR_OK) {
        if (len == 2) hipLaunchKernelGGL(k_hash2_level, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, ctx->stream, din, (u32)(2 * count), Fr::zero(), dout, P);
        else hipLaunchKernelGGL(k_hash_many, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, ctx->stream, din, (u32)len, (u32)count, dout, P);
        if (hipGetLastError() != hipSuccess) { ctx->err = "poseidon launch failed"; rc = ZKPOR_E_HIP; }
    }
    if (rc == ZKPOR_OK && hipMemcpyAsync(out, dout, count * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "D2H failed"; rc = ZKPOR_E_HIP; }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(din); (void)hipFree(dout);
    return rc;
} ZK_ABI_CATCH_IN(ctx)

size_t zkpor_witgen_poseidon_sboxes(int t) {
    if (t != 3 && t != 5 && t != 6 && t != 13) return 0;
    return (size_t)POS_RF * t + (size_t)pos_rp(t);
}

int32_t zkpor_witgen_poseidon_trace_dev(zkpor_ctx* ctx, int t, void* d_states, size_t count, void* d_trace) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !d_states || !d_trace || count == 0) return ZKPOR_E_ARG;
    if (t != 3 && t != 5 && t != 6 && t != 13) { ctx->err = "witgen: the circuit's Poseidon widths are 3, 5, 6 and 13"; return ZKPOR_E_ARG; }
    PosDev P;
    ZK_TRY(pos_dev(ctx, &P));
    if (!P.tab29) { ctx->err = "witgen: the 29-bit Poseidon tables are not available"; return ZKPOR_E_STATE; }
    PhaseScope ps(ctx, "witgen_poseidon");
    dim3 grid((unsigned)((count + 63) / 64)), block(64);
    if (t == 3) hipLaunchKernelGGL(k_poseidon_trace<3>, grid, block, 0, ctx->stream, (Fr*)d_states, count, (Fr*)d_trace, P);
    else if (t == 5) hipLaunchKernelGGL(k_poseidon_trace<5>, grid, block, 0, ctx->stream, (Fr*)d_states, count, (Fr*)d_trace, P);
    else if (t == 6) hipLaunchKernelGGL(k_poseidon_trace<6>, grid, block, 0, ctx->stream, (Fr*)d_states, count, (Fr*)d_trace, P);
    else hipLaunchKernelGGL(k_poseidon_trace<13>, grid, block, 0, ctx->stream, (Fr*)d_states, count, (Fr*)d_trace, P);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_witgen_limbs_dev(zkpor_ctx* ctx, const void* d_values, size_t n, int nb_limbs, void* d_limbs, void* d_multiplicity, void* d_bad) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !d_values || !d_limbs || !d_multiplicity || !d_bad || n == 0 || nb_limbs < 1 || nb_limbs > 15) return ZKPOR_E_ARG;
    PhaseScope ps(ctx, "witgen_limbs");
    hipLaunchKernelGGL(k_witgen_limbs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const Fr*)d_values, n, nb_limbs, (Fr*)d_limbs,
                       (u32*)d_multiplicity, (u32*)d_bad);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_witgen_inverse_dev(zkpor_ctx* ctx, const void* d_values, size_t n, const uint64_t challenge[4], void* d_out, void* d_bad) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !d_values || !d_out || !d_bad || !challenge || n == 0) return ZKPOR_E_ARG;
    Fr c;
    memcpy(&c, challenge, sizeof c);
    PhaseScope ps(ctx, "witgen_inverse");
    const size_t threads = (n + 7) / 8;
    hipLaunchKernelGGL(k_witgen_inverse, dim3((unsigned)((threads + 127) / 128)), dim3(128), 0, ctx->stream, (const Fr*)d_values, n, c, (Fr*)d_out, (u32*)d_bad);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_witgen_bits_dev(zkpor_ctx* ctx, const void* d_values, size_t n, int nbits, void* d_bits, void* d_bad) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !d_values || !d_bits || !d_bad || n == 0 || nbits < 1 || nbits > 254) return ZKPOR_E_ARG;
    PhaseScope ps(ctx, "witgen_bits");
    hipLaunchKernelGGL(k_witgen_bits, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const Fr*)d_values, n, nbits, (Fr*)d_bits, (u32*)d_bad);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_witgen_gather_dev(zkpor_ctx* ctx, const void* d_table, size_t table_len, const void* d_indices, size_t n, void* d_out, void* d_bad) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !d_table || !d_indices || !d_out || !d_bad || n == 0 || table_len == 0 || table_len > 0xffffffffull) return ZKPOR_E_ARG;
    PhaseScope ps(ctx, "witgen_gather");
    hipLaunchKernelGGL(k_witgen_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const Fr*)d_table, (u32)table_len, (const Fr*)d_indices, n,
                       (Fr*)d_out, (u32*)d_bad);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_witgen_divmod_small_dev(zkpor_ctx* ctx, const void* d_values, size_t n, uint32_t divisor, void* d_quotient, void* d_remainder) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !d_values || !d_quotient || !d_remainder || n == 0) return ZKPOR_E_ARG;
    if (divisor == 0) { ctx->err = "witgen: division by zero (big.Int.DivMod panics)"; return ZKPOR_E_ARG; }
    PhaseScope ps(ctx, "witgen_divmod");
    hipLaunchKernelGGL(k_witgen_divmod_small, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const Fr*)d_values, n, divisor, (Fr*)d_quotient, (Fr*)d_remainder);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_witgen_scatter_dev(zkpor_ctx* ctx, void* d_w, const void* d_src, const uint32_t* d_wire_ids, size_t n) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !d_w || !d_src || !d_wire_ids) return ZKPOR_E_ARG;
    if (n == 0) return ZKPOR_OK;
    PhaseScope ps(ctx, "witgen_scatter");
    hipLaunchKernelGGL(k_witgen_scatter, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (Fr*)d_w, (const Fr*)d_src, d_wire_ids, n);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_witgen_scatter_known_dev(zkpor_ctx* ctx, void* d_w, uint8_t* d_known, const void* d_src, const uint32_t* d_wire_ids, size_t n) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || ((!d_w || !d_known || !d_src || !d_wire_ids) && n)) return ZKPOR_E_ARG;
    if (n == 0) return ZKPOR_OK;
    PhaseScope ps(ctx, "witgen_scatter");
    hipLaunchKernelGGL(k_witgen_scatter_known, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (Fr*)d_w, d_known, (const Fr*)d_src, d_wire_ids, n);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_poseidon_leaves(zkpor_ctx* ctx, const zkpor_account_t* accounts, const zkpor_asset_t* assets,
                              size_t n_assets_total, size_t n, int tier, uint8_t* out32) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !accounts || !out32 || n == 0 || tier < 1 || (n_assets_total && !assets)) return ZKPOR_E_ARG;
    for (size_t i = 0; i < n; ++i) {
        if (accounts[i].n_assets > (uint32_t)tier || (size_t)accounts[i].asset_off + accounts[i].n_assets > n_assets_total) {
            ctx->err = "poseidon_leaves: account exceeds the tier or the asset array"; return ZKPOR_E_ARG;
        }
    }
    PosDev P;
    ZK_TRY(pos_dev(ctx, &P));
    AccountHdr* dacc = nullptr; AssetRec* das = nullptr; Fr* dout = nullptr; uint8_t* dbe = nullptr;
    int32_t rc = ZKPOR_OK;
    if (hipMalloc((void**)&dacc, n * sizeof(AccountHdr)) != hipSuccess || hipMalloc((void**)&das, (n_assets_total + 1) * sizeof(AssetRec)) != hipSuccess ||
        hipMalloc((void**)&dout, n * sizeof(Fr)) != hipSuccess || hipMalloc((void**)&dbe, n * 32) != hipSuccess) { ctx->err = "out of device memory"; rc = ZKPOR_E_OOM; }
    if (rc == ZKPOR_OK && (zk::h2d_sync(ctx, dacc, accounts, n * sizeof(AccountHdr)) != ZKPOR_OK ||
                           (n_assets_total && zk::h2d_sync(ctx, das, assets, n_assets_total * sizeof(AssetRec)) != ZKPOR_OK))) { ctx->err = "H2D failed"; rc = ZKPOR_E_HIP; }
    if (rc == ZKPOR_OK) {
        PhaseScope ps(ctx, "poseidon_leaf");
        launch_account_leaves(ctx, dacc, das, (u32)n, tier, dout, P);
        hipLaunchKernelGGL(k_fr_to_be, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dout, dbe, n);
        if (hipGetLastError() != hipSuccess) { ctx->err = "poseidon launch failed"; rc = ZKPOR_E_HIP; }
    }
    if (rc == ZKPOR_OK && hipMemcpyAsync(out32, dbe, n * 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "D2H failed"; rc = ZKPOR_E_HIP; }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(dacc); (void)hipFree(das); (void)hipFree(dout); (void)hipFree(dbe);
    return rc;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_merkle_build_dev(zkpor_ctx* ctx, const void* d_leaves_mont, size_t n, int depth,
                               const uint64_t nil_leaf_mont[4], uint64_t root_mont[4]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || (n && !d_leaves_mont) || !nil_leaf_mont || !root_mont) return ZKPOR_E_ARG;
    Fr nil, root;
    memcpy(&nil, nil_leaf_mont, 32);
    ZK_TRY(merkle_build_core(ctx, (const Fr*)d_leaves_mont, n, depth, nil, nullptr, &root));
    memcpy(root_mont, &root, 32);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

int32_t zkpor_merkle_build(zkpor_ctx* ctx, const uint8_t* leaves32_be, size_t n, int depth, const uint8_t nil_leaf[32],
                           uint8_t* levels_out, uint8_t root_out[32]) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || (n && !leaves32_be) || !nil_leaf || !root_out) return ZKPOR_E_ARG;
    if (depth < 1 || depth > 32 || n > ((size_t)1 << depth)) { ctx->err = "merkle: bad depth / leaf count"; return ZKPOR_E_ARG; }
    Fr nil = from_be32(nil_leaf);
    size_t tot = levels_out ? levels_total(n, depth) : 0;
    uint8_t* dbe = nullptr; Fr* dleaves = nullptr; Fr* dlev = nullptr;
    int32_t rc = ZKPOR_OK;
    size_t be_cnt = n > tot ? n : tot;
    if (hipMalloc((void**)&dbe, (be_cnt + 1) * 32) != hipSuccess || hipMalloc((void**)&dleaves, (n + 1) * sizeof(Fr)) != hipSuccess ||
        (tot && hipMalloc((void**)&dlev, tot * sizeof(Fr)) != hipSuccess)) { ctx->err = "out of device memory"; rc = ZKPOR_E_OOM; }
    if (rc == ZKPOR_OK && n) {
        if (zk::h2d_sync(ctx, dbe, leaves32_be, n * 32) != ZKPOR_OK) { ctx->err = "H2D failed"; rc = ZKPOR_E_HIP; }
        else hipLaunchKernelGGL(k_fr_from_be, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dbe, dleaves, n);
    }
    Fr root;
    if (rc == ZKPOR_OK) rc = merkle_build_core(ctx, dleaves, n, depth, nil, dlev, &root);
    if (rc == ZKPOR_OK) {
        Fr c = Fr::from_mont(root);
        for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) root_out[31 - (i * 4 + j)] = (uint8_t)(c.v[i] >> (8 * j));
        if (tot) {
            hipLaunchKernelGGL(k_fr_to_be, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, dlev, dbe, tot);
            if (hipMemcpyAsync(levels_out, dbe, tot * 32, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { ctx->err = "D2H failed"; rc = ZKPOR_E_HIP; }
        }
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(dbe); (void)hipFree(dleaves); if (dlev) (void)hipFree(dlev);
    return rc;
} ZK_ABI_CATCH_IN(ctx)

// ---- FixedDepthMerkleTree -------------------------------------------------------------------------------------------
int32_t zkpor_tree_create(zkpor_ctx* ctx, int depth, const uint8_t nil_leaf[32], uint64_t capacity, zkpor_tree** out) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !nil_leaf || !out) return ZKPOR_E_ARG;
    // NewFixedDepthMerkleTree panics on these (merkletree.go:138-146); here they are argument errors
    if (depth <= 0 || depth > 32) { ctx->err = "tree: depth must be in [1,32]"; return ZKPOR_E_ARG; }
    if (capacity > ((uint64_t)1 << depth)) { ctx->err = "tree: capacity exceeds maximum for given depth"; return ZKPOR_E_ARG; }
    PosDev P;
    ZK_TRY(pos_dev(ctx, &P));
    zkpor_tree* t = new zkpor_tree();
    t->ctx = ctx; t->depth = depth; t->capacity = capacity;
    uint64_t m = capacity;
    int32_t rc = ZKPOR_OK;
    for (int l = 0; l <= depth && rc == ZKPOR_OK; ++l) {
        t->count[l] = m;
        t->dev.cnt[l] = m;
        rc = tree_alloc<Fr>(t, m, &t->dev.nodes[l], false);
        if (rc == ZKPOR_OK) rc = tree_alloc<u32>(t, (m + 31) / 32 + 2, &t->dev.dirty[l], true);
        m = (m + 1) / 2;
    }
    if (rc == ZKPOR_OK) rc = tree_alloc<Fr>(t, 33, &t->nil_dev, true);
    if (rc != ZKPOR_OK) { tree_free(t); return rc; }
    t->dev.nil = t->nil_dev; t->dev.depth = depth;
    Fr nil0 = from_be32(nil_leaf);
    if (hipMemcpyAsync(t->nil_dev, &nil0, sizeof(Fr), hipMemcpyHostToDevice, ctx->stream) != hipSuccess) { ctx->err = "tree: H2D failed"; tree_free(t); return ZKPOR_E_HIP; }
    hipLaunchKernelGGL(k_tree_nil_chain, dim3(1), dim3(64), 0, ctx->stream, t->nil_dev, depth, P);
    if (hipGetLastError() != hipSuccess || hipMemcpyAsync(t->nil_host, t->nil_dev, 33 * sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "tree: nil-hash chain failed"; tree_free(t); return ZKPOR_E_HIP; }
    t->root = t->nil_host[depth];  // root of the empty tree (:172)
    *out = t;
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
void zkpor_tree_destroy(zkpor_tree* t) try {
    ZK_ENTER(t ? t->ctx->device : -1);
    if (!t) return;
    (void)hipStreamSynchronize(t->ctx->stream);
    tree_free(t);
} catch (...) { zk::abi_exception("exception in zkpor_tree_destroy"); }
int32_t zkpor_tree_nil_hash(zkpor_tree* t, int level, uint8_t out[32]) try {
    ZK_ENTER(t ? t->ctx->device : -1);
    if (!t || !out || level < 0 || level > t->depth) return ZKPOR_E_ARG;
    fr_to_be_host(t->nil_host[level], out);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((t ? t->ctx : nullptr))
int32_t zkpor_tree_set(zkpor_tree* t, const uint32_t* keys, const uint8_t* values32_be, size_t n) try {
    ZK_ENTER(t ? t->ctx->device : -1);
    if (!t || (n && (!keys || !values32_be))) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = t->ctx;
    for (size_t i = 0; i < n; ++i)
        if ((uint64_t)keys[i] >= t->capacity) {  // Set's error return (:180-182); nothing is stored
            ctx->err = "tree: key " + std::to_string(keys[i]) + " out of range for capacity " + std::to_string(t->capacity);
            return ZKPOR_E_ARG;
        }
    if (n == 0) return ZKPOR_OK;
    DevTmp dk, dv;
    ZK_TRY(dk.put(ctx, keys, n * 4));
    ZK_TRY(dv.put(ctx, values32_be, n * 32));
    hipLaunchKernelGGL(k_tree_set, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, t->dev, (const u32*)dk.p, (const uint8_t*)dv.p, n);
    ZK_KERNEL_CHECK(ctx);
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((t ? t->ctx : nullptr))
int32_t zkpor_tree_set_range_dev(zkpor_tree* t, uint64_t first_key, const void* d_leaves_mont, size_t n) try {
    ZK_ENTER(t ? t->ctx->device : -1);
    if (!t || (n && !d_leaves_mont)) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = t->ctx;
    if (first_key + n > t->capacity) { ctx->err = "tree: key range exceeds capacity"; return ZKPOR_E_ARG; }
    if (n == 0) return ZKPOR_OK;
    hipLaunchKernelGGL(k_tree_set_range, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, t->dev, (u64)first_key, (const Fr*)d_leaves_mont, n);
    ZK_KERNEL_CHECK(ctx);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((t ? t->ctx : nullptr))
int32_t zkpor_tree_build(zkpor_tree* t) try {
    ZK_ENTER(t ? t->ctx->device : -1);
    if (!t) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = t->ctx;
    PosDev P;
    ZK_TRY(pos_dev(ctx, &P));
    {
        PhaseScope ps(ctx, "poseidon_tree");
        for (int l = 1; l <= t->depth; ++l) {
            uint64_t cnt = t->count[l];
            if (cnt == 0) break;
            hipLaunchKernelGGL(k_tree_level, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, ctx->stream, t->dev, l, (u64)cnt, (u64)t->count[l - 1], P);
            ZK_KERNEL_CHECK(ctx);
        }
    }
    // root: the top node if anything beneath it was ever set, else the nil hash (:268-274)
    Fr top; u32 bit = 0;
    if (t->count[t->depth]) {
        ZK_HIP(ctx, hipMemcpyAsync(&top, t->dev.nodes[t->depth], sizeof(Fr), hipMemcpyDeviceToHost, ctx->stream));
        ZK_HIP(ctx, hipMemcpyAsync(&bit, t->dev.dirty[t->depth], 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    t->root = (bit & 1u) ? top : t->nil_host[t->depth];
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((t ? t->ctx : nullptr))
int32_t zkpor_tree_root(zkpor_tree* t, uint8_t out[32]) try {
    ZK_ENTER(t ? t->ctx->device : -1);
    if (!t || !out) return ZKPOR_E_ARG;
    fr_to_be_host(t->root, out);
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((t ? t->ctx : nullptr))
static int32_t tree_query(zkpor_tree* t, const uint32_t* keys, size_t n, uint8_t* out, bool proofs) {
    zkpor_ctx* ctx = t->ctx;
    if (n == 0) return ZKPOR_OK;
    const size_t per = proofs ? (size_t)t->depth * 32 : 32;
    DevTmp dk, dout;
    ZK_TRY(dk.put(ctx, keys, n * 4));
    ZK_TRY(dout.make(ctx, n * per));
    size_t threads = proofs ? n * (size_t)t->depth : n;
    if (proofs) hipLaunchKernelGGL(k_tree_proofs, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, t->dev, (const u32*)dk.p, n, (uint8_t*)dout.p);
    else hipLaunchKernelGGL(k_tree_get, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, ctx->stream, t->dev, (const u32*)dk.p, n, (uint8_t*)dout.p);
    ZK_KERNEL_CHECK(ctx);
    ZK_HIP(ctx, hipMemcpyAsync(out, dout.p, n * per, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKPOR_OK;
}
int32_t zkpor_tree_get(zkpor_tree* t, const uint32_t* keys, size_t n, uint8_t* out32) try {
    ZK_ENTER(t ? t->ctx->device : -1);
    if (!t || (n && (!keys || !out32))) return ZKPOR_E_ARG;
    return tree_query(t, keys, n, out32, false);  // keys >= capacity read as nil (:288-290)
} ZK_ABI_CATCH_IN((t ? t->ctx : nullptr))
int32_t zkpor_tree_get_proofs(zkpor_tree* t, const uint32_t* keys, size_t n, uint8_t* out) try {
    ZK_ENTER(t ? t->ctx->device : -1);
    if (!t || (n && (!keys || !out))) return ZKPOR_E_ARG;
    for (size_t i = 0; i < n; ++i)
        if (t->depth < 32 && ((uint64_t)keys[i] >> t->depth) != 0) {  // GetProof's error return (:298-300)
            t->ctx->err = "tree: key " + std::to_string(keys[i]) + " out of range for tree depth " + std::to_string(t->depth);
            return ZKPOR_E_ARG;
        }
    return tree_query(t, keys, n, out, true);
} ZK_ABI_CATCH_IN((t ? t->ctx : nullptr))
int32_t zkpor_merkle_verify_proofs(zkpor_ctx* ctx, const uint8_t root[32], const uint32_t* keys, const uint8_t* proofs,
                                   const uint8_t* leaves32_be, size_t n, int depth, uint8_t* ok_out) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !root || (n && (!keys || !proofs || !leaves32_be || !ok_out))) return ZKPOR_E_ARG;
    if (depth < 1 || depth > 32) { ctx->err = "merkle: bad depth"; return ZKPOR_E_ARG; }
    if (n == 0) return ZKPOR_OK;
    PosDev P;
    ZK_TRY(pos_dev(ctx, &P));
    DevTmp dr, dk, dp, dl, dok;
    ZK_TRY(dr.put(ctx, root, 32)); ZK_TRY(dk.put(ctx, keys, n * 4));
    ZK_TRY(dp.put(ctx, proofs, n * (size_t)depth * 32)); ZK_TRY(dl.put(ctx, leaves32_be, n * 32));
    ZK_TRY(dok.make(ctx, n));
    hipLaunchKernelGGL(k_verify_proofs, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (const uint8_t*)dr.p, (const u32*)dk.p,
                       (const uint8_t*)dp.p, (const uint8_t*)dl.p, n, depth, (uint8_t*)dok.p, P);
    ZK_KERNEL_CHECK(ctx);
    ZK_HIP(ctx, hipMemcpyAsync(ok_out, dok.p, n, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

// ---- CEX asset-list commitments / batch commitments -------------------------------------------------------------------
int32_t zkpor_cex_commitments(zkpor_ctx* ctx, const zkpor_cex_asset_const_t* assets, size_t n_assets, const zkpor_cex_totals_t* totals,
                              size_t n_states, uint8_t* out32) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !assets || !totals || !out32 || n_assets == 0 || n_assets > 0xffffu) return ZKPOR_E_ARG;
    if (n_states == 0) return ZKPOR_OK;
    if (n_states > 0xffffffffull) return ZKPOR_E_ARG;
    PosDev P;
    ZK_TRY(pos_dev(ctx, &P));
    DevTmp dc, dt, de, dout, dbe;
    ZK_TRY(dc.put(ctx, assets, n_assets * sizeof(CexAssetConst)));
    ZK_TRY(dt.put(ctx, totals, n_states * n_assets * sizeof(CexTotals)));
    ZK_TRY(de.make(ctx, n_assets * 18 * sizeof(Fr)));
    ZK_TRY(dout.make(ctx, n_states * sizeof(Fr)));
    ZK_TRY(dbe.make(ctx, n_states * 32));
    {
        PhaseScope ps(ctx, "cex_commitments");
        hipLaunchKernelGGL(k_cex_tier_elems, dim3((unsigned)((n_assets * 18 + 127) / 128)), dim3(128), 0, ctx->stream, (const CexAssetConst*)dc.p,
                           (u32)n_assets, (Fr*)de.p);
        ZK_KERNEL_CHECK(ctx);
        if (use_coop(ctx, n_states))
            hipLaunchKernelGGL(k_cex_commitments_coop, dim3((unsigned)((n_states + 3) / 4)), dim3(64), 0, ctx->stream, (const CexAssetConst*)dc.p,
                               (const Fr*)de.p, (u32)n_assets, (const CexTotals*)dt.p, (u32)n_states, (Fr*)dout.p, P);
        else
            hipLaunchKernelGGL(k_cex_commitments, dim3((unsigned)((n_states + 63) / 64)), dim3(64), 0, ctx->stream, (const CexAssetConst*)dc.p,
                               (const Fr*)de.p, (u32)n_assets, (const CexTotals*)dt.p, (u32)n_states, (Fr*)dout.p, P);
        ZK_KERNEL_CHECK(ctx);
    }
    hipLaunchKernelGGL(k_fr_to_be, dim3((unsigned)((n_states + 255) / 256)), dim3(256), 0, ctx->stream, (const Fr*)dout.p, (uint8_t*)dbe.p, n_states);
    ZK_KERNEL_CHECK(ctx);
    ZK_HIP(ctx, hipMemcpyAsync(out32, dbe.p, n_states * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)
int32_t zkpor_batch_commitments(zkpor_ctx* ctx, const uint8_t* roots32, const uint8_t* before32, const uint8_t* after32,
                                const uint32_t* min_index, const uint32_t* max_index, size_t n, uint8_t* out32) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || (n && (!roots32 || !before32 || !after32 || !min_index || !max_index || !out32))) return ZKPOR_E_ARG;
    if (n == 0) return ZKPOR_OK;
    PosDev P;
    ZK_TRY(pos_dev(ctx, &P));
    DevTmp dr, db, da, dmin, dmax, dout, dbe;
    ZK_TRY(dr.put(ctx, roots32, n * 32)); ZK_TRY(db.put(ctx, before32, n * 32)); ZK_TRY(da.put(ctx, after32, n * 32));
    ZK_TRY(dmin.put(ctx, min_index, n * 4)); ZK_TRY(dmax.put(ctx, max_index, n * 4));
    ZK_TRY(dout.make(ctx, n * sizeof(Fr))); ZK_TRY(dbe.make(ctx, n * 32));
    hipLaunchKernelGGL(k_batch_commitments, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, ctx->stream, (const uint8_t*)dr.p, (const uint8_t*)db.p,
                       (const uint8_t*)da.p, (const u32*)dmin.p, (const u32*)dmax.p, (u32)n, (Fr*)dout.p, P);
    ZK_KERNEL_CHECK(ctx);
    hipLaunchKernelGGL(k_fr_to_be, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (const Fr*)dout.p, (uint8_t*)dbe.p, n);
    ZK_KERNEL_CHECK(ctx);
    ZK_HIP(ctx, hipMemcpyAsync(out32, dbe.p, n * 32, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

// ---- account totals / collateral tiers ----------------------------------------------------------------------------------
int32_t zkpor_account_totals(zkpor_ctx* ctx, zkpor_account_t* accounts, const zkpor_asset_t* assets, size_t n_assets_total, size_t n,
                             const zkpor_cex_asset_const_t* cex, size_t n_cex, uint8_t* tier_info_out, uint8_t* valid_out) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || !accounts || !cex || n_cex == 0 || (n_assets_total && !assets)) return ZKPOR_E_ARG;
    if (n == 0) return ZKPOR_OK;
    if (n > 0xffffffffull || n_cex > 0xffffffffull) return ZKPOR_E_ARG;
    for (size_t i = 0; i < n; ++i)
        if ((size_t)accounts[i].asset_off + accounts[i].n_assets > n_assets_total) { ctx->err = "totals: asset range outside the asset array"; return ZKPOR_E_ARG; }
    DevTmp da, ds, dc, dt, dv;
    ZK_TRY(da.put(ctx, accounts, n * sizeof(AccountHdr)));
    ZK_TRY(ds.put(ctx, assets, n_assets_total * sizeof(AssetRec)));
    ZK_TRY(dc.put(ctx, cex, n_cex * sizeof(CexAssetConst)));
    if (tier_info_out) ZK_TRY(dt.make(ctx, n_assets_total * 6));
    if (valid_out) ZK_TRY(dv.make(ctx, n));
    {
        PhaseScope ps(ctx, "account_totals");
        hipLaunchKernelGGL(k_account_totals, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream, (AccountHdr*)da.p, (const AssetRec*)ds.p, (u32)n,
                           (const CexAssetConst*)dc.p, (u32)n_cex, (uint8_t*)dt.p, (uint8_t*)dv.p);
        ZK_KERNEL_CHECK(ctx);
    }
    ZK_HIP(ctx, hipMemcpyAsync(accounts, da.p, n * sizeof(AccountHdr), hipMemcpyDeviceToHost, ctx->stream));
    if (tier_info_out && n_assets_total) ZK_HIP(ctx, hipMemcpyAsync(tier_info_out, dt.p, n_assets_total * 6, hipMemcpyDeviceToHost, ctx->stream));
    if (valid_out) ZK_HIP(ctx, hipMemcpyAsync(valid_out, dv.p, n, hipMemcpyDeviceToHost, ctx->stream));
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN(ctx)

// ---- accounts straight into the tree -------------------------------------------------------------------------------------
// buildAccountTree (src/witness/main.go:130-199) for one chunk of accounts, without the leaves ever leaving the device:
// totals (optional) -> leaf hashes -> Set at keys first_key .. first_key + n - 1.  The caller streams a 10^8-account data
// set through this in chunks and calls zkpor_tree_build once.
int32_t zkpor_tree_set_accounts(zkpor_tree* t, uint64_t first_key, zkpor_account_t* accounts, const zkpor_asset_t* assets,
                                size_t n_assets_total, size_t n, int tier, const zkpor_cex_asset_const_t* cex_or_null, size_t n_cex,
                                uint8_t* valid_out_or_null) try {
    ZK_ENTER(t ? t->ctx->device : -1);
    if (!t || !accounts || n == 0 || tier < 1 || (n_assets_total && !assets)) return ZKPOR_E_ARG;
    zkpor_ctx* ctx = t->ctx;
    if (first_key + n > t->capacity) { ctx->err = "tree: key range exceeds capacity"; return ZKPOR_E_ARG; }
    if (n > 0xffffffffull) return ZKPOR_E_ARG;
    for (size_t i = 0; i < n; ++i)
        if (accounts[i].n_assets > (uint32_t)tier || (size_t)accounts[i].asset_off + accounts[i].n_assets > n_assets_total) {
            ctx->err = "tree: account exceeds the tier or the asset array"; return ZKPOR_E_ARG;
        }
    PosDev P;
    ZK_TRY(pos_dev(ctx, &P));
    DevTmp da, ds, dc, dv, dl;
    ZK_TRY(da.put(ctx, accounts, n * sizeof(AccountHdr)));
    ZK_TRY(ds.put(ctx, assets, n_assets_total * sizeof(AssetRec)));
    ZK_TRY(dl.make(ctx, n * sizeof(Fr)));
    if (cex_or_null) {
        if (n_cex == 0 || n_cex > 0xffffffffull) return ZKPOR_E_ARG;
        ZK_TRY(dc.put(ctx, cex_or_null, n_cex * sizeof(CexAssetConst)));
        if (valid_out_or_null) ZK_TRY(dv.make(ctx, n));
        PhaseScope ps(ctx, "account_totals");
        hipLaunchKernelGGL(k_account_totals, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, ctx->stream, (AccountHdr*)da.p, (const AssetRec*)ds.p, (u32)n,
                           (const CexAssetConst*)dc.p, (u32)n_cex, (uint8_t*)nullptr, (uint8_t*)dv.p);
        ZK_KERNEL_CHECK(ctx);
    }
    {
        PhaseScope ps(ctx, "poseidon_leaf");
        launch_account_leaves(ctx, (const AccountHdr*)da.p, (const AssetRec*)ds.p, (u32)n, tier, (Fr*)dl.p, P);
        ZK_KERNEL_CHECK(ctx);
    }
    hipLaunchKernelGGL(k_tree_set_range, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, t->dev, (u64)first_key, (const Fr*)dl.p, n);
    ZK_KERNEL_CHECK(ctx);
    if (cex_or_null) {
        ZK_HIP(ctx, hipMemcpyAsync(accounts, da.p, n * sizeof(AccountHdr), hipMemcpyDeviceToHost, ctx->stream));  // the totals, for the caller's records
        if (valid_out_or_null) ZK_HIP(ctx, hipMemcpyAsync(valid_out_or_null, dv.p, n, hipMemcpyDeviceToHost, ctx->stream));
    }
    ZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return ZKPOR_OK;
} ZK_ABI_CATCH_IN((t ? t->ctx : nullptr))

}  // extern "C"
