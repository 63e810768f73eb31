// Poseidon on the HOST over FrH: the parameters (Grain LFSR, the published procedure; third statement in this repo next to
// csrc/poseidon.hip and oracle/poseidon.hpp — the three are compared in tests) and the PLAIN HADES permutation with an S-box trace.
// Users: the circuit compiler (host/circuit/*.hpp: the coefficient vectors of the in-circuit gadget's linear layers) and the host
// executor's native Poseidon instruction (host/solver_exec.hpp).  Nothing here is on the device path.
//
// What it restates: poseidon.Poseidon of the bnb-chain gnark-crypto fork (call sites src/utils/constants.go:126, account_tree.go:19,27,
// utils.go:748,765,780) and its in-circuit twin std/hash/poseidon (circuit/utils.go:17,47; circuit/batch_create_user_circuit.go:104,129,
// 181,270,281,320): x^5, R_F = 8, R_P(t), inputs absorbed in blocks of 12 behind a capacity element, digest = state[out_idx] and the
// carry into the next block = state[carry_idx] — (1, 0), pinned by the reference's user_config.json fixture.
#pragma once
#include <cstdint>
#include <vector>
#include "fr_host.hpp"

namespace zkpor_host {

static const int kPosRF = 8;
static const int kPosMaxT = 13;
inline int PosRP(int t) { static const int tab[] = {56, 57, 56, 60, 60, 63, 64, 63, 60, 66, 60, 65}; return tab[t - 2]; }
inline int PosSboxes(int t) { return kPosRF * t + PosRP(t); }

struct PosParams {       // one width
    int t = 0, rp = 0;
    std::vector<FrH> rc;  // (RF + RP) * t
    std::vector<FrH> mds; // t * t, row-major
};

namespace pos_detail {
struct Grain {
    uint8_t s[80];
    int head = 0;
    Grain(int t, int rf, int rp) {
        int k = 0;
        auto put = [&](unsigned v, int w) { for (int i = w - 1; i >= 0; --i) s[k++] = (uint8_t)((v >> i) & 1u); };
        put(1, 2); put(0, 4); put(254, 12); put((unsigned)t, 12); put((unsigned)rf, 10); put((unsigned)rp, 10);
        while (k < 80) s[k++] = 1;
        for (int i = 0; i < 160; ++i) clock();
    }
    int tap(int i) const { return s[(head + i) % 80]; }
    int clock() { const int nb = tap(62) ^ tap(51) ^ tap(38) ^ tap(23) ^ tap(13) ^ tap(0); s[head] = (uint8_t)nb; head = (head + 1) % 80; return nb; }
    int bit() { for (;;) { const int a = clock(), b = clock(); if (a) return b; } }
    // 254 bits, most significant first; *lt = below the modulus
    void next(uint64_t out[4], bool* lt) {
        out[0] = out[1] = out[2] = out[3] = 0;
        for (int i = 253; i >= 0; --i) if (bit()) out[i >> 6] |= (uint64_t)1 << (i & 63);
        *lt = !FrH::geq_mod(out);
    }
};
}  // namespace pos_detail

inline PosParams MakePosParams(int t) {
    PosParams P;
    P.t = t; P.rp = PosRP(t);
    pos_detail::Grain g(t, kPosRF, P.rp);
    P.rc.resize((size_t)(kPosRF + P.rp) * t);
    for (auto& c : P.rc) {
        uint64_t v[4]; bool ok;
        do { g.next(v, &ok); } while (!ok);          // rejection sampling
        c = FrH::from_canon(v);
    }
    std::vector<FrH> xy(2 * (size_t)t);
    for (;;) {
        for (auto& e : xy) {
            uint64_t v[4]; bool ok;
            g.next(v, &ok);
            if (!ok) FrH::sub_mod(v);                 // v < 2^254 < 2 r
            e = FrH::from_canon(v);
        }
        bool good = true;
        for (int i = 0; i < 2 * t && good; ++i) for (int j = i + 1; j < 2 * t; ++j) if (xy[i] == xy[j]) { good = false; break; }
        for (int i = 0; i < t && good; ++i) for (int j = 0; j < t; ++j) if (FrH::add(xy[i], xy[t + j]).is_zero()) { good = false; break; }
        if (good) break;
    }
    P.mds.resize((size_t)t * t);
    for (int i = 0; i < t; ++i) for (int j = 0; j < t; ++j) P.mds[(size_t)i * t + j] = FrH::inv(FrH::add(xy[i], xy[t + j]));
    return P;
}
inline const PosParams& PosParamsOf(int t) {
    struct All { PosParams p[kPosMaxT + 1]; All() { for (int w = 2; w <= kPosMaxT; ++w) p[w] = MakePosParams(w); } };
    static const All all;   // every width at the first use (thread-safe initialisation)
    return all.p[t];
}

inline FrH PosPow5(const FrH& x, FrH* x2_out = nullptr, FrH* x4_out = nullptr) {
    const FrH x2 = FrH::sqr(x), x4 = FrH::sqr(x2);
    if (x2_out) *x2_out = x2;
    if (x4_out) *x4_out = x4;
    return FrH::mul(x4, x);
}

// the plain permutation; trace (may be NULL): 3 * PosSboxes(t) elements, (x^2, x^4, x^5) per S-box in round order (full rounds: lanes
// 0..t-1, partial rounds: lane 0) — the three multiplication wires the in-circuit gadget spends per S-box
inline void PosPermute(FrH* st, int t, FrH* trace = nullptr) {
    const PosParams& P = PosParamsOf(t);
    FrH tmp[kPosMaxT];
    size_t s = 0;
    const int rounds = kPosRF + P.rp;
    for (int r = 0; r < rounds; ++r) {
        const FrH* c = &P.rc[(size_t)r * t];
        for (int i = 0; i < t; ++i) st[i] = FrH::add(st[i], c[i]);
        const bool full = r < kPosRF / 2 || r >= kPosRF / 2 + P.rp;
        for (int i = 0; i < (full ? t : 1); ++i) {
            FrH x2, x4;
            st[i] = PosPow5(st[i], &x2, &x4);
            if (trace) { trace[3 * s] = x2; trace[3 * s + 1] = x4; trace[3 * s + 2] = st[i]; }
            ++s;
        }
        for (int i = 0; i < t; ++i) {
            FrH acc = FrH::zero();
            for (int j = 0; j < t; ++j) acc = FrH::add(acc, FrH::mul(P.mds[(size_t)i * t + j], st[j]));
            tmp[i] = acc;
        }
        for (int i = 0; i < t; ++i) st[i] = tmp[i];
    }
}

// number of permutations / S-box wires of the sponge over n inputs (n >= 1): blocks of 12, the last one ragged
inline size_t PosSpongePerms(size_t n) { return (n + 11) / 12; }
inline size_t PosSpongeSboxes(size_t n) {
    const size_t full = n / 12, rem = n % 12;
    return full * (size_t)PosSboxes(13) + (rem ? (size_t)PosSboxes((int)rem + 1) : 0);
}
// poseidon.Poseidon(inputs...): the digest; trace (may be NULL) receives 3 * PosSpongeSboxes(n) elements, permutation after permutation
inline FrH PosSponge(const FrH* in, size_t n, FrH* trace = nullptr, int out_idx = 1, int carry_idx = 0) {
    FrH cap = FrH::zero(), out = FrH::zero();
    FrH st[kPosMaxT];
    size_t done = 0;
    while (done < n) {
        const int k = (int)((n - done) < 12 ? (n - done) : 12), t = k + 1;
        st[0] = cap;
        for (int i = 0; i < k; ++i) st[1 + i] = in[done + i];
        PosPermute(st, t, trace);
        if (trace) trace += 3 * (size_t)PosSboxes(t);
        cap = st[carry_idx]; out = st[out_idx];
        done += (size_t)k;
    }
    return out;
}

}  // namespace zkpor_host
