// Batch decompression of gnark-crypto's compressed BN254 points on the device — SURVEY.md §8 row f2.
// The reference writes its proving key compressed (src/keygen/main.go:46, pk.WriteTo) and spends minutes of CPU time
// per tier turning it back into affine points at start-up (pk.UnsafeReadFrom, src/prover/prover/prover.go:336-349:
// one modular square root per point, ~3x10^8 points).  Here a whole array is decompressed by one launch (one thread
// per point: y = sqrt(x^3 + b) by exponentiation, sign chosen by the flag bits), straight into the layout the MSM
// kernels consume.
//
// Wire format (gnark-crypto v0.14 ecc/bn254/marshal.go — third-party, absent from /root/reference; restated from its
// published constants): big-endian coordinates, the two most significant bits of the FIRST byte carry
//   00 uncompressed | 01 infinity | 10 compressed, y is the lexicographically smallest root | 11 ... the largest,
// G1 = 32 bytes (X), G2 = 64 bytes (X.A1 | X.A0).  "Lexicographically largest": y > (p-1)/2 as a canonical integer;
// for Fp2 the A1 component decides unless it is zero (E2.LexicographicallyLargest).
#include "common.cuh"

namespace zk {

struct DecompConsts {
    u32 sqrt_exp[8];  // (p+1)/4
    u32 half_p[8];    // (p-1)/2
    Fp three;         // curve b (G1), Montgomery
    Fp2 b2;           // twist b' = 3/(9+u)
    Fp inv2;
};

ZK_D bool geq_limbs(const u32* a, const u32* b) {  // a >= b, little-endian 8 limbs
    for (int i = 7; i >= 0; --i) {
        if (a[i] > b[i]) return true;
        if (a[i] < b[i]) return false;
    }
    return true;
}
// 32 big-endian bytes -> Montgomery Fp; `mask2` drops the two flag bits; false if the value is not < p
ZK_D bool fp_from_be(const uint8_t* b, bool mask2, Fp* out) {
    Fp x;
    for (int i = 0; i < 8; ++i) {
        const uint8_t* p = b + 28 - 4 * i;
        x.v[i] = ((u32)p[0] << 24) | ((u32)p[1] << 16) | ((u32)p[2] << 8) | (u32)p[3];
    }
    if (mask2) x.v[7] &= 0x3fffffffu;
    u32 m[8];
    for (int i = 0; i < 8; ++i) m[i] = FpParams::mod(i);
    if (geq_limbs(x.v, m)) return false;
    *out = Fp::to_mont(x);
    return true;
}
ZK_D bool fp_lex_largest(const Fp& y, const DecompConsts& K) {  // y > (p-1)/2
    Fp c = Fp::from_mont(y);
    return geq_limbs(c.v, K.half_p) && !(geq_limbs(K.half_p, c.v));
}
ZK_D bool fp_eq(const Fp& a, const Fp& b) {
    bool e = true;
    for (int i = 0; i < 8; ++i) e &= a.v[i] == b.v[i];
    return e;
}
// p = 3 (mod 4): the candidate root is a^((p+1)/4); returns whether a is a square
ZK_D bool fp_sqrt(const Fp& a, const DecompConsts& K, Fp* r) {
    Fp s = Fp::pow(a, K.sqrt_exp, 8);
    *r = s;
    return fp_eq(Fp::sqr(s), a);
}
ZK_D bool fp2_sqrt(const Fp2& a, const DecompConsts& K, Fp2* r) {
    if (a.a1.is_zero()) {
        Fp s;
        if (fp_sqrt(a.a0, K, &s)) { *r = {s, Fp::zero()}; return true; }
        if (fp_sqrt(Fp::neg(a.a0), K, &s)) { *r = {Fp::zero(), s}; return true; }  // (s u)^2 = -s^2
        return false;
    }
    Fp n;
    if (!fp_sqrt(Fp::add(Fp::sqr(a.a0), Fp::sqr(a.a1)), K, &n)) return false;  // the norm of a square is a square
    Fp d = Fp::mul(Fp::add(a.a0, n), K.inv2);
    Fp x0;
    if (!fp_sqrt(d, K, &x0)) {
        d = Fp::sub(d, n);  // (a0 - n)/2
        if (!fp_sqrt(d, K, &x0)) return false;
    }
    Fp x1 = Fp::mul(Fp::mul(a.a1, K.inv2), Fp::inv(x0));
    Fp2 c = {x0, x1};
    Fp2 sq = Fp2::sqr(c);
    if (!fp_eq(sq.a0, a.a0) || !fp_eq(sq.a1, a.a1)) return false;
    *r = c;
    return true;
}

// err[0] = 1 + index of the first offending element (atomicMin-free: any offender wins), kind in err[1]
ZK_D void flag_error(u32* err, size_t i, u32 kind) {
    if (atomicCAS(&err[0], 0u, (u32)(i + 1)) == 0u) err[1] = kind;
}
enum { DERR_FLAG = 1, DERR_RANGE = 2, DERR_NOT_ON_CURVE = 3 };

__global__ __launch_bounds__(256) void k_decompress_g1(const uint8_t* __restrict__ in, size_t n, G1Affine* __restrict__ out,
                                                       DecompConsts K, u32* err) {
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint8_t* b = in + 32 * i;
    const u32 flag = b[0] & 0xC0u;
    G1Affine p;
    p.x = Fp::zero(); p.y = Fp::zero();
    if (flag == 0x40u) { out[i] = p; return; }                      // infinity
    if (flag == 0x00u) { flag_error(err, i, DERR_FLAG); out[i] = p; return; }  // an uncompressed point is 64 bytes: not this stream
    Fp x;
    if (!fp_from_be(b, true, &x)) { flag_error(err, i, DERR_RANGE); out[i] = p; return; }
    Fp y2 = Fp::add(Fp::mul(Fp::sqr(x), x), K.three);
    Fp y;
    if (!fp_sqrt(y2, K, &y)) { flag_error(err, i, DERR_NOT_ON_CURVE); out[i] = p; return; }
    const bool want_largest = flag == 0xC0u;
    if (fp_lex_largest(y, K) != want_largest) y = Fp::neg(y);
    p.x = x; p.y = y;
    out[i] = p;
}

__global__ __launch_bounds__(256) void k_decompress_g2(const uint8_t* __restrict__ in, size_t n, G2Affine* __restrict__ out,
                                                       DecompConsts K, u32* err) {
    size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= n) return;
    const uint8_t* b = in + 64 * i;
    const u32 flag = b[0] & 0xC0u;
    G2Affine p;
    p.x = {Fp::zero(), Fp::zero()}; p.y = p.x;
    if (flag == 0x40u) { out[i] = p; return; }
    if (flag == 0x00u) { flag_error(err, i, DERR_FLAG); out[i] = p; return; }
    Fp2 x;
    if (!fp_from_be(b, true, &x.a1) || !fp_from_be(b + 32, false, &x.a0)) { flag_error(err, i, DERR_RANGE); out[i] = p; return; }
    Fp2 y2 = Fp2::add(Fp2::mul(Fp2::sqr(x), x), K.b2);
    Fp2 y;
    if (!fp2_sqrt(y2, K, &y)) { flag_error(err, i, DERR_NOT_ON_CURVE); out[i] = p; return; }
    const bool largest = y.a1.is_zero() ? fp_lex_largest(y.a0, K) : fp_lex_largest(y.a1, K);
    if (largest != (flag == 0xC0u)) y = Fp2::neg(y);
    p.x = x; p.y = y;
    out[i] = p;
}

static DecompConsts decomp_consts() {
    DecompConsts K;
    u32 p[8];
    for (int i = 0; i < 8; ++i) p[i] = FpParams::mod(i);
    // (p+1)/4 and (p-1)/2: p is odd and p = 3 (mod 4), so p+1 does not carry out of 256 bits
    u32 t[8]; u64 c = 1;
    for (int i = 0; i < 8; ++i) { c += p[i]; t[i] = (u32)c; c >>= 32; }
    for (int i = 0; i < 8; ++i) K.sqrt_exp[i] = (t[i] >> 2) | (i < 7 ? t[i + 1] << 30 : 0u);
    for (int i = 0; i < 8; ++i) K.half_p[i] = (p[i] >> 1) | (i < 7 ? p[i + 1] << 31 : 0u);  // (p-1)/2 = p >> 1
    Fp one = Fp::one();
    Fp two = Fp::add(one, one);
    K.three = Fp::add(two, one);
    K.inv2 = Fp::inv(two);
    Fp nine = Fp::add(Fp::add(K.three, K.three), K.three);
    Fp2 xi = {nine, one};
    Fp2 inv = Fp2::inv(xi);
    K.b2 = {Fp::mul(K.three, inv.a0), Fp::mul(K.three, inv.a1)};
    return K;
}

// d_out: device array of n affine points (64 B / 128 B each); host_in: n x 32 / 64 bytes on the host
int32_t decompress_to_device(zkpor_ctx* ctx, bool g2, const uint8_t* host_in, size_t n, void* d_out) {
    if (n == 0) return ZKPOR_OK;
    const size_t per = g2 ? 64 : 32;
    uint8_t* din = nullptr;
    u32* derr = nullptr;
    ZK_HIP(ctx, hipMalloc((void**)&din, n * per));
    if (hipMalloc((void**)&derr, 8) != hipSuccess) { (void)hipFree(din); ctx->err = "decompress: out of device memory"; return ZKPOR_E_OOM; }
    int32_t rc = ZKPOR_OK;
    u32 herr[2] = {0, 0};
    DecompConsts K = decomp_consts();
    if (zk::h2d_sync(ctx, din, host_in, n * per) != ZKPOR_OK ||
        hipMemsetAsync(derr, 0, 8, ctx->stream) != hipSuccess) { ctx->err = "decompress: H2D failed"; rc = ZKPOR_E_HIP; }
    if (rc == ZKPOR_OK) {
        PhaseScope ps(ctx, "decompress");
        unsigned blocks = (unsigned)((n + 255) / 256);
        if (g2) hipLaunchKernelGGL(k_decompress_g2, dim3(blocks), dim3(256), 0, ctx->stream, din, n, (G2Affine*)d_out, K, derr);
        else hipLaunchKernelGGL(k_decompress_g1, dim3(blocks), dim3(256), 0, ctx->stream, din, n, (G1Affine*)d_out, K, derr);
        if (hipGetLastError() != hipSuccess) { ctx->err = "decompress: launch failed"; rc = ZKPOR_E_HIP; }
    }
    if (rc == ZKPOR_OK && (hipMemcpyAsync(herr, derr, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                           hipStreamSynchronize(ctx->stream) != hipSuccess)) { ctx->err = "decompress: D2H failed"; rc = ZKPOR_E_HIP; }
    (void)hipFree(din); (void)hipFree(derr);
    if (rc == ZKPOR_OK && herr[0]) {
        static const char* why[] = {"", "is not a compressed point (flag bits 00)", "has a coordinate >= p", "is not on the curve"};
        ctx->err = std::string("decompress: element ") + std::to_string(herr[0] - 1) + " " + why[herr[1] < 4 ? herr[1] : 0];
        return ZKPOR_E_ARG;
    }
    return rc;
}

}  // namespace zk

using namespace zk;
extern "C" {

int32_t zkpor_g1_decompress(zkpor_ctx* ctx, const uint8_t* compressed32, size_t n, void* out_affine) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || (n && (!compressed32 || !out_affine))) return ZKPOR_E_ARG;
    if (n == 0) return ZKPOR_OK;
    void* d = nullptr;
    ZK_HIP(ctx, hipMalloc(&d, n * 64));
    int32_t rc = decompress_to_device(ctx, false, compressed32, n, d);
    if (rc == ZKPOR_OK && hipMemcpy(out_affine, d, n * 64, hipMemcpyDeviceToHost) != hipSuccess) { ctx->err = "decompress: D2H failed"; rc = ZKPOR_E_HIP; }
    (void)hipFree(d);
    return rc;
} ZK_ABI_CATCH_IN(ctx)
int32_t zkpor_g2_decompress(zkpor_ctx* ctx, const uint8_t* compressed64, size_t n, void* out_affine) try {
    ZK_ENTER(ctx ? ctx->device : -1);
    if (!ctx || (n && (!compressed64 || !out_affine))) return ZKPOR_E_ARG;
    if (n == 0) return ZKPOR_OK;
    void* d = nullptr;
    ZK_HIP(ctx, hipMalloc(&d, n * 128));
    int32_t rc = decompress_to_device(ctx, true, compressed64, n, d);
    if (rc == ZKPOR_OK && hipMemcpy(out_affine, d, n * 128, hipMemcpyDeviceToHost) != hipSuccess) { ctx->err = "decompress: D2H failed"; rc = ZKPOR_E_HIP; }
    (void)hipFree(d);
    return rc;
} ZK_ABI_CATCH_IN(ctx)

}  // extern "C"
