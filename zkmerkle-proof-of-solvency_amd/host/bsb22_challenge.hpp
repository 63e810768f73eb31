// The value the BSB22 commitment hint hands back to the solver: the Fr challenge
//
//     hash_to_field( commitment.Marshal() || public committed values, 32 B big-endian each ;  DST "bsb22-commitment" )
//
// In gnark v0.10 (bnb fork pinned by the reference's go.mod:57-60; NOT under /root/reference, restated from its published
// algorithm) groth16.Prove overrides the commitment hint: it commits to the private committed wires (here: zkpor_commit), then
// writes constraint.SerializeCommitment(commitment.Marshal(), hashed, 32) into backend.ProverConfig.HashToFieldFn
// (default hash_to_field.New([]byte(constraint.CommitmentDst))) and sets the hint's output to the first fr.Bytes of Sum().
// hash_to_field.Sum = fr.Hash(msg, dst, 1)[0].Bytes(): 48 = 16 + fr.Bytes pseudo-random bytes from RFC 9380 expand_message_xmd
// over SHA-256, read big-endian, reduced mod r.  Call site in the reference: the single groth16.Prove of
// src/prover/prover/prover.go:269; the Go shim (go/zkporgpu/prove.go) uses gnark-crypto's own function — this file is the
// same arithmetic for the C++ host path (prove_batch.hpp's CommitFn callers), pinned by the RFC's known answers
// (tests/test_bsb22_challenge_cpu.py).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace zkpor_host {

// ------------------------------------------------------------------------------------------------ SHA-256 (FIPS 180-4)
class Sha256 {
  public:
    Sha256() { Reset(); }
    void Reset() {
        static const uint32_t iv[8] = {0x6a09e667u, 0xbb67ae85u, 0x3c6ef372u, 0xa54ff53au, 0x510e527fu, 0x9b05688cu, 0x1f83d9abu, 0x5be0cd19u};
        memcpy(h_, iv, sizeof iv);
        len_ = 0; fill_ = 0;
    }
    void Write(const uint8_t* p, size_t n) {
        len_ += n;
        while (n) {
            size_t k = 64 - fill_ < n ? 64 - fill_ : n;
            memcpy(buf_ + fill_, p, k);
            fill_ += k; p += k; n -= k;
            if (fill_ == 64) { Block(buf_); fill_ = 0; }
        }
    }
    void Write(const std::string& s) { Write((const uint8_t*)s.data(), s.size()); }
    void Sum(uint8_t out[32]) {  // finalises a copy: the object can keep absorbing
        Sha256 c = *this;
        const uint64_t bits = c.len_ * 8;
        uint8_t pad[72] = {0x80};
        size_t padlen = (c.fill_ < 56 ? 56 : 120) - c.fill_;
        for (int i = 0; i < 8; ++i) pad[padlen + i] = (uint8_t)(bits >> (56 - 8 * i));
        c.Write(pad, padlen + 8);
        for (int i = 0; i < 8; ++i) { out[4 * i] = c.h_[i] >> 24; out[4 * i + 1] = c.h_[i] >> 16; out[4 * i + 2] = c.h_[i] >> 8; out[4 * i + 3] = c.h_[i]; }
    }

  private:
    static uint32_t rotr(uint32_t x, int r) { return (x >> r) | (x << (32 - r)); }
    void Block(const uint8_t* p) {
        static const uint32_t K[64] = {
            0x428a2f98u, 0x71374491u, 0xb5c0fbcfu, 0xe9b5dba5u, 0x3956c25bu, 0x59f111f1u, 0x923f82a4u, 0xab1c5ed5u, 0xd807aa98u, 0x12835b01u, 0x243185beu,
            0x550c7dc3u, 0x72be5d74u, 0x80deb1feu, 0x9bdc06a7u, 0xc19bf174u, 0xe49b69c1u, 0xefbe4786u, 0x0fc19dc6u, 0x240ca1ccu, 0x2de92c6fu, 0x4a7484aau,
            0x5cb0a9dcu, 0x76f988dau, 0x983e5152u, 0xa831c66du, 0xb00327c8u, 0xbf597fc7u, 0xc6e00bf3u, 0xd5a79147u, 0x06ca6351u, 0x14292967u, 0x27b70a85u,
            0x2e1b2138u, 0x4d2c6dfcu, 0x53380d13u, 0x650a7354u, 0x766a0abbu, 0x81c2c92eu, 0x92722c85u, 0xa2bfe8a1u, 0xa81a664bu, 0xc24b8b70u, 0xc76c51a3u,
            0xd192e819u, 0xd6990624u, 0xf40e3585u, 0x106aa070u, 0x19a4c116u, 0x1e376c08u, 0x2748774cu, 0x34b0bcb5u, 0x391c0cb3u, 0x4ed8aa4au, 0x5b9cca4fu,
            0x682e6ff3u, 0x748f82eeu, 0x78a5636fu, 0x84c87814u, 0x8cc70208u, 0x90befffau, 0xa4506cebu, 0xbef9a3f7u, 0xc67178f2u};
        uint32_t w[64];
        for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)p[4 * i] << 24) | ((uint32_t)p[4 * i + 1] << 16) | ((uint32_t)p[4 * i + 2] << 8) | p[4 * i + 3];
        for (int i = 16; i < 64; ++i) {
            uint32_t s0 = rotr(w[i - 15], 7) ^ rotr(w[i - 15], 18) ^ (w[i - 15] >> 3);
            uint32_t s1 = rotr(w[i - 2], 17) ^ rotr(w[i - 2], 19) ^ (w[i - 2] >> 10);
            w[i] = w[i - 16] + s0 + w[i - 7] + s1;
        }
        uint32_t a = h_[0], b = h_[1], c = h_[2], d = h_[3], e = h_[4], f = h_[5], g = h_[6], h = h_[7];
        for (int i = 0; i < 64; ++i) {
            uint32_t S1 = rotr(e, 6) ^ rotr(e, 11) ^ rotr(e, 25);
            uint32_t t1 = h + S1 + ((e & f) ^ (~e & g)) + K[i] + w[i];
            uint32_t S0 = rotr(a, 2) ^ rotr(a, 13) ^ rotr(a, 22);
            uint32_t t2 = S0 + ((a & b) ^ (a & c) ^ (b & c));
            h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
        }
        h_[0] += a; h_[1] += b; h_[2] += c; h_[3] += d; h_[4] += e; h_[5] += f; h_[6] += g; h_[7] += h;
    }
    uint32_t h_[8];
    uint64_t len_;
    uint8_t buf_[64];
    size_t fill_;
};

// ------------------------------------------------------------------------------------------------ RFC 9380 §5.3.1 expand_message_xmd, H = SHA-256
inline std::string ExpandMsgXmd(const std::string& msg, const std::string& dst, size_t len_in_bytes) {
    const size_t ell = (len_in_bytes + 31) / 32;
    if (ell > 255 || len_in_bytes > 65535) throw std::invalid_argument("expand_message_xmd: too many bytes requested");
    if (dst.size() > 255) throw std::invalid_argument("expand_message_xmd: DST longer than 255 bytes");
    std::string dst_prime = dst;
    dst_prime.push_back((char)dst.size());
    Sha256 h;
    const uint8_t zpad[64] = {0};
    h.Write(zpad, 64);
    h.Write(msg);
    const uint8_t lib[3] = {(uint8_t)(len_in_bytes >> 8), (uint8_t)len_in_bytes, 0};
    h.Write(lib, 3);
    h.Write(dst_prime);
    uint8_t b0[32], bi[32];
    h.Sum(b0);
    h.Reset();
    h.Write(b0, 32);
    const uint8_t one = 1;
    h.Write(&one, 1);
    h.Write(dst_prime);
    h.Sum(bi);
    std::string out((const char*)bi, 32);
    for (size_t i = 2; i <= ell; ++i) {
        uint8_t x[32];
        for (int k = 0; k < 32; ++k) x[k] = b0[k] ^ bi[k];
        h.Reset();
        h.Write(x, 32);
        const uint8_t ib = (uint8_t)i;
        h.Write(&ib, 1);
        h.Write(dst_prime);
        h.Sum(bi);
        out.append((const char*)bi, 32);
    }
    out.resize(len_in_bytes);
    return out;
}

// ------------------------------------------------------------------------------------------------ fr.Hash(msg, dst, count) of gnark-crypto bn254
// count elements, each the big-endian integer of 48 pseudo-random bytes reduced mod r; returned as 32-byte big-endian canonical values
inline void FrReduceBE48(const uint8_t in[48], uint8_t out[32]) {
    // r, little-endian 64-bit limbs
    static const uint64_t R[4] = {0x43e1f593f0000001ull, 0x2833e84879b97091ull, 0xb85045b68181585dull, 0x30644e72e131a029ull};
    uint64_t acc[5] = {0, 0, 0, 0, 0};  // value < 2r at the top of every step, 5th limb for the shift
    // Horner over the bits, most significant first: acc = 2 acc + bit, conditional subtraction of r.  384 steps; host-side, once per proof
    for (int i = 0; i < 48 * 8; ++i) {
        const int bit = (in[i / 8] >> (7 - i % 8)) & 1;
        uint64_t carry = (uint64_t)bit;
        for (int k = 0; k < 5; ++k) { uint64_t v = acc[k]; acc[k] = (v << 1) | carry; carry = v >> 63; }
        // acc >= r ?
        bool ge = acc[4] != 0;
        if (!ge) {
            ge = true;
            for (int k = 3; k >= 0; --k) { if (acc[k] != R[k]) { ge = acc[k] > R[k]; break; } }
        }
        if (ge) {
            unsigned __int128 borrow = 0;
            for (int k = 0; k < 4; ++k) {
                unsigned __int128 d = (unsigned __int128)acc[k] - R[k] - (uint64_t)borrow;
                acc[k] = (uint64_t)d;
                borrow = (d >> 64) & 1;
            }
            acc[4] -= (uint64_t)borrow;
        }
    }
    for (int k = 0; k < 4; ++k) for (int b = 0; b < 8; ++b) out[31 - (8 * k + b)] = (uint8_t)(acc[k] >> (8 * b));
}
inline std::vector<std::string> FrHash(const std::string& msg, const std::string& dst, size_t count) {
    const size_t L = 48;  // 16 + fr.Bytes
    std::string u = ExpandMsgXmd(msg, dst, count * L);
    std::vector<std::string> out;
    for (size_t i = 0; i < count; ++i) {
        uint8_t e[32];
        FrReduceBE48((const uint8_t*)u.data() + i * L, e);
        out.emplace_back((const char*)e, 32);
    }
    return out;
}

inline const char* CommitmentDst() { return "bsb22-commitment"; }  // gnark constraint.CommitmentDst

// constraint.SerializeCommitment(privateCommitment, publicCommitted, 32) then HashToFieldFn.Write/Sum: the challenge as 32 B big-endian.
// commitment_raw = proof.Commitments[i].Marshal() = X || Y big-endian canonical (zkpor_g1_marshal of what zkpor_commit returns); public_committed = the public
// wires among the committed ones ("hashed"), canonical, 32 B big-endian each (none in BatchCreateUserCircuit: its single public
// input is not range-checked).
inline std::string Bsb22Challenge(const uint8_t commitment_raw[64], const std::vector<std::string>& public_committed_be32 = {},
                                  const std::string& dst = CommitmentDst()) {
    std::string msg((const char*)commitment_raw, 64);
    for (const std::string& v : public_committed_be32) {
        if (v.size() != 32) throw std::invalid_argument("Bsb22Challenge: public committed values are 32 bytes big-endian");
        msg += v;
    }
    return FrHash(msg, dst, 1)[0];
}

}  // namespace zkpor_host
