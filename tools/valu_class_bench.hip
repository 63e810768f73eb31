// Issue cost of every VALU instruction CLASS the hot kernels use, in SHADER CYCLES per wave-instruction per SIMD (VERDICT r02 item 3:
// the round-2 "VALU issue ceiling" assumed 4 cycles for every instruction after measuring multiply-class ones only, while
// MI355X_MICROARCH.md states 2 cycles per wave-instruction).  Method: exactly W waves per SIMD on every SIMD of the chip, each wave
// runs a loop of 32 instructions of ONE class over 8 independent register chains (no dependent-issue stalls from 2 waves up);
// every wave reads the shader clock (s_memtime) before and after, cycles per instruction = mean elapsed / (W x instructions).  The
// figure is independent of the clock the part actually runs at.  A wall-clock figure (HIP events) is printed beside it.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/valu_class_bench.hip -o gpurun_out/valu_class_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef uint32_t u32;
typedef uint64_t u64;

// 8 instructions over 8 chains; REP4 makes the 32-instruction loop body
#define REP4(x) x x x x

enum Op { ADD_U32, SUB_U32, AND_B32, XOR_B32, LSHLREV_B32, ASHRREV_I32, MOV_B32, CNDMASK_B32, ADD3_U32, LSHL_ADD_U32, AND_OR_B32, BFE_I32, ALIGNBIT_B32,
          ASHRREV_I64, LSHLREV_B64, LSHL_ADD_U64, ADD_CO_ADDC, MUL_LO_U32, MUL_HI_U32, MAD_U64_U32, MAD_I64_I32, MAD_U32_U24, FMA_F64, FMA_F32, PK_FMA_F32,
          MOV_DPP, MIX_MAD_ADD, MIX_MAD_ASHR, CNDMASK_SGPR, PAT_MMAA, PAT_M_AAAA, PAT_ADD_AND, PAT_DEP_ADD, PAT_M_A_DEP, PAT_AAMM_X2, PAT_MAD_MOV, PAT_ASHR64_AND, N_OPS };
static const char* NAMES[N_OPS] = {"v_add_u32", "v_sub_u32", "v_and_b32", "v_xor_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_cndmask_b32", "v_add3_u32",
    "v_lshl_add_u32", "v_and_or_b32", "v_bfe_i32", "v_alignbit_b32", "v_ashrrev_i64", "v_lshlrev_b64", "v_lshl_add_u64", "v_add_co_u32+v_addc_co_u32",
    "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u64_u32", "v_mad_i64_i32", "v_mad_u32_u24", "v_fma_f64", "v_fma_f32", "v_pk_fma_f32", "v_mov_b32_dpp quad_perm",
    "MIX 1:1 v_mad_i64_i32 / v_add_u32", "MIX 1:1 v_mad_i64_i32 / v_ashrrev_i64",
    "v_cndmask_b32 (mask in an SGPR pair)", "PAT mad mad add add", "PAT mad add add add add", "PAT add and add and (indep.)", "PAT add->add dependent chain (1 reg)",
    "PAT mad, add (add depends on mad lo)", "PAT add add mad mad add add mad mad", "PAT mad mov mad mov", "PAT ashr64 and ashr64 and"};

#define ONE8_32(INS) \
    asm volatile(INS " %0, %0, %8\n" INS " %1, %1, %8\n" INS " %2, %2, %8\n" INS " %3, %3, %8\n" INS " %4, %4, %8\n" INS " %5, %5, %8\n" INS " %6, %6, %8\n" INS " %7, %7, %8\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
#define ONE8_32_REV(INS) /* shift-style: op dst, amount, src */ \
    asm volatile(INS " %0, %8, %0\n" INS " %1, %8, %1\n" INS " %2, %8, %2\n" INS " %3, %8, %3\n" INS " %4, %8, %4\n" INS " %5, %8, %5\n" INS " %6, %8, %6\n" INS " %7, %8, %7\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh));
#define ONE8_3OP(INS) \
    asm volatile(INS " %0, %0, %8, %9\n" INS " %1, %1, %8, %9\n" INS " %2, %2, %8, %9\n" INS " %3, %3, %8, %9\n" INS " %4, %4, %8, %9\n" INS " %5, %5, %8, %9\n" INS " %6, %6, %8, %9\n" INS " %7, %7, %8, %9\n" \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(sh));
#define ONE8_64_REV(INS) \
    asm volatile(INS " %0, %8, %0\n" INS " %1, %8, %1\n" INS " %2, %8, %2\n" INS " %3, %8, %3\n" INS " %4, %8, %4\n" INS " %5, %8, %5\n" INS " %6, %8, %6\n" INS " %7, %8, %7\n" \
                 : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(sh));

template <int OP>
__global__ __launch_bounds__(64) void k_class(u64* cyc, u32* sink, int iters, u32 seed) {
    u32 a0 = threadIdx.x + seed, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7;
    u32 b = blockIdx.x * 11 + 5, sh = (seed & 3) + 1;
    u64 smask = __builtin_amdgcn_readfirstlane(b) * 0x9e3779b97f4a7c15ull;
    u64 c0 = a0, c1 = a1, c2 = a2, c3 = a3, c4 = a4, c5 = a5, c6 = a6, c7 = a7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3, d4 = a4, d5 = a5, d6 = a6, d7 = a7, db = 1.0000001;
    float f0 = a0, f1 = a1, f2 = a2, f3 = a3, f4 = a4, f5 = a5, f6 = a6, f7 = a7, fb = 1.0001f;
    typedef float float2_ __attribute__((ext_vector_type(2)));
    float2_ p0 = {f0, f1}, p1 = {f1, f2}, p2 = {f2, f3}, p3 = {f3, f4}, p4 = {f4, f5}, p5 = {f5, f6}, p6 = {f6, f7}, p7 = {f7, f0}, pb = {fb, fb};
    __syncthreads();
    u64 t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (OP == ADD_U32) { REP4(ONE8_32("v_add_u32")) }
        else if (OP == SUB_U32) { REP4(ONE8_32("v_sub_u32")) }
        else if (OP == AND_B32) { REP4(ONE8_32("v_and_b32")) }
        else if (OP == XOR_B32) { REP4(ONE8_32("v_xor_b32")) }
        else if (OP == LSHLREV_B32) { REP4(ONE8_32_REV("v_lshlrev_b32")) }
        else if (OP == ASHRREV_I32) { REP4(ONE8_32_REV("v_ashrrev_i32")) }
        else if (OP == MOV_B32) {
            REP4(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        }
        else if (OP == CNDMASK_B32) {
            REP4(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                              "v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
        }
        else if (OP == ADD3_U32) { REP4(ONE8_3OP("v_add3_u32")) }
        else if (OP == LSHL_ADD_U32) { REP4(ONE8_3OP("v_lshl_add_u32")) }
        else if (OP == AND_OR_B32) { REP4(ONE8_3OP("v_and_or_b32")) }
        else if (OP == BFE_I32) { REP4(ONE8_3OP("v_bfe_i32")) }
        else if (OP == ALIGNBIT_B32) { REP4(ONE8_3OP("v_alignbit_b32")) }
        else if (OP == ASHRREV_I64) { REP4(ONE8_64_REV("v_ashrrev_i64")) }
        else if (OP == LSHLREV_B64) { REP4(ONE8_64_REV("v_lshlrev_b64")) }
        else if (OP == LSHL_ADD_U64) {
            REP4(asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                              "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(c0 | 1));)
        }
        else if (OP == ADD_CO_ADDC) {
            REP4(asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n"
                              "v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "vcc");)
        }
        else if (OP == MUL_LO_U32) { REP4(ONE8_32("v_mul_lo_u32")) }
        else if (OP == MUL_HI_U32) { REP4(ONE8_32("v_mul_hi_u32")) }
        else if (OP == MAD_U64_U32) {
            REP4(asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                              "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a0), "v"(b) : "vcc");)
        }
        else if (OP == MAD_I64_I32) {
            REP4(asm volatile("v_mad_i64_i32 %0, vcc, %8, %9, %0\n v_mad_i64_i32 %1, vcc, %8, %9, %1\n v_mad_i64_i32 %2, vcc, %8, %9, %2\n v_mad_i64_i32 %3, vcc, %8, %9, %3\n"
                              "v_mad_i64_i32 %4, vcc, %8, %9, %4\n v_mad_i64_i32 %5, vcc, %8, %9, %5\n v_mad_i64_i32 %6, vcc, %8, %9, %6\n v_mad_i64_i32 %7, vcc, %8, %9, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a0), "v"(b) : "vcc");)
        }
        else if (OP == MAD_U32_U24) { REP4(ONE8_3OP("v_mad_u32_u24")) }
        else if (OP == FMA_F64) {
            REP4(asm volatile("v_fma_f64 %0, %0, %8, %0\n v_fma_f64 %1, %1, %8, %1\n v_fma_f64 %2, %2, %8, %2\n v_fma_f64 %3, %3, %8, %3\n"
                              "v_fma_f64 %4, %4, %8, %4\n v_fma_f64 %5, %5, %8, %5\n v_fma_f64 %6, %6, %8, %6\n v_fma_f64 %7, %7, %8, %7\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(db));)
        }
        else if (OP == FMA_F32) {
            REP4(asm volatile("v_fma_f32 %0, %0, %8, %0\n v_fma_f32 %1, %1, %8, %1\n v_fma_f32 %2, %2, %8, %2\n v_fma_f32 %3, %3, %8, %3\n"
                              "v_fma_f32 %4, %4, %8, %4\n v_fma_f32 %5, %5, %8, %5\n v_fma_f32 %6, %6, %8, %6\n v_fma_f32 %7, %7, %8, %7\n"
                              : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(fb));)
        }
        else if (OP == PK_FMA_F32) {
            REP4(asm volatile("v_pk_fma_f32 %0, %0, %8, %0\n v_pk_fma_f32 %1, %1, %8, %1\n v_pk_fma_f32 %2, %2, %8, %2\n v_pk_fma_f32 %3, %3, %8, %3\n"
                              "v_pk_fma_f32 %4, %4, %8, %4\n v_pk_fma_f32 %5, %5, %8, %5\n v_pk_fma_f32 %6, %6, %8, %6\n v_pk_fma_f32 %7, %7, %8, %7\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pb));)
        }
        else if (OP == MOV_DPP) {
            REP4(asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        }
        else if (OP == MIX_MAD_ADD) {  // do a 64-bit multiply-add and a plain add co-issue / overlap?
            REP4(asm volatile("v_mad_i64_i32 %0, vcc, %8, %9, %0\n v_add_u32 %4, %4, %9\n v_mad_i64_i32 %1, vcc, %8, %9, %1\n v_add_u32 %5, %5, %9\n"
                              "v_mad_i64_i32 %2, vcc, %8, %9, %2\n v_add_u32 %6, %6, %9\n v_mad_i64_i32 %3, vcc, %8, %9, %3\n v_add_u32 %7, %7, %9\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0), "v"(b) : "vcc");)
        }
        else if (OP == MIX_MAD_ASHR) {  // the column step of the 29-bit product: mads, then one 64-bit arithmetic shift
            REP4(asm volatile("v_mad_i64_i32 %0, vcc, %8, %9, %0\n v_ashrrev_i64 %4, 29, %4\n v_mad_i64_i32 %1, vcc, %8, %9, %1\n v_ashrrev_i64 %5, 29, %5\n"
                              "v_mad_i64_i32 %2, vcc, %8, %9, %2\n v_ashrrev_i64 %6, 29, %6\n v_mad_i64_i32 %3, vcc, %8, %9, %3\n v_ashrrev_i64 %7, 29, %7\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a0), "v"(b) : "vcc");)
        }

        else if (OP == CNDMASK_SGPR) {
            REP4(asm volatile("v_cndmask_b32 %0, %0, %8, %9\n v_cndmask_b32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, %9\n v_cndmask_b32 %3, %3, %8, %9\n"
                              "v_cndmask_b32 %4, %4, %8, %9\n v_cndmask_b32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, %9\n v_cndmask_b32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "s"(smask));)
        }
        else if (OP == PAT_MMAA) {
            REP4(asm volatile("v_mad_i64_i32 %0, vcc, %8, %9, %0\n v_mad_i64_i32 %1, vcc, %8, %9, %1\n v_add_u32 %4, %4, %9\n v_add_u32 %5, %5, %9\n"
                              "v_mad_i64_i32 %2, vcc, %8, %9, %2\n v_mad_i64_i32 %3, vcc, %8, %9, %3\n v_add_u32 %6, %6, %9\n v_add_u32 %7, %7, %9\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0), "v"(b) : "vcc");)
        }
        else if (OP == PAT_M_AAAA) {  // 10 instructions x 3 + 2 = 32: approximated by 8 = mad + 4 adds + mad + 2 adds
            REP4(asm volatile("v_mad_i64_i32 %0, vcc, %8, %9, %0\n v_add_u32 %4, %4, %9\n v_add_u32 %5, %5, %9\n v_add_u32 %6, %6, %9\n v_add_u32 %7, %7, %9\n"
                              "v_add_u32 %4, %4, %9\n v_add_u32 %5, %5, %9\n v_add_u32 %6, %6, %9\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0), "v"(b) : "vcc");)
        }
        else if (OP == PAT_ADD_AND) {
            REP4(asm volatile("v_add_u32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_and_b32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        }
        else if (OP == PAT_DEP_ADD) {
            REP4(asm volatile("v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_add_u32 %0, %0, %1\n"
                              : "+v"(a0) : "v"(b));)
        }
        else if (OP == PAT_M_A_DEP) {  // the add consumes the low half of the product just made (register pair aliasing via asm operands is not
                                      // expressible; use the 32-bit result of v_mul_lo instead)
            REP4(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_add_u32 %4, %4, %0\n v_mul_lo_u32 %1, %1, %8\n v_add_u32 %5, %5, %1\n"
                              "v_mul_lo_u32 %2, %2, %8\n v_add_u32 %6, %6, %2\n v_mul_lo_u32 %3, %3, %8\n v_add_u32 %7, %7, %3\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        }
        else if (OP == PAT_AAMM_X2) {
            REP4(asm volatile("v_add_u32 %4, %4, %9\n v_and_b32 %5, %5, %9\n v_mad_i64_i32 %0, vcc, %8, %9, %0\n v_mad_i64_i32 %1, vcc, %8, %9, %1\n"
                              "v_add_u32 %6, %6, %9\n v_and_b32 %7, %7, %9\n v_mad_i64_i32 %2, vcc, %8, %9, %2\n v_mad_i64_i32 %3, vcc, %8, %9, %3\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0), "v"(b) : "vcc");)
        }
        else if (OP == PAT_MAD_MOV) {
            REP4(asm volatile("v_mad_i64_i32 %0, vcc, %8, %9, %0\n v_mov_b32 %4, %5\n v_mad_i64_i32 %1, vcc, %8, %9, %1\n v_mov_b32 %5, %6\n"
                              "v_mad_i64_i32 %2, vcc, %8, %9, %2\n v_mov_b32 %6, %7\n v_mad_i64_i32 %3, vcc, %8, %9, %3\n v_mov_b32 %7, %4\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(a0), "v"(b) : "vcc");)
        }
        else if (OP == PAT_ASHR64_AND) {
            REP4(asm volatile("v_ashrrev_i64 %0, 29, %0\n v_and_b32 %4, %4, %8\n v_ashrrev_i64 %1, 29, %1\n v_and_b32 %5, %5, %8\n"
                              "v_ashrrev_i64 %2, 29, %2\n v_and_b32 %6, %6, %8\n v_ashrrev_i64 %3, 29, %3\n v_and_b32 %7, %7, %8\n"
                              : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));)
        }
    }
    u64 t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 64 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ (u32)(c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7) ^
        (u32)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7) ^ (u32)(f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7) ^ (u32)(p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y);
}

template <int OP>
static void launch(int blocks, u64* cyc, u32* sink, int iters) { hipLaunchKernelGGL(k_class<OP>, dim3(blocks), dim3(64), 0, 0, cyc, sink, iters, 1u); }
typedef void (*LaunchFn)(int, u64*, u32*, int);
template <int... I> struct Seq {};
template <int N, int... I> struct Gen : Gen<N - 1, N - 1, I...> {};
template <int... I> struct Gen<0, I...> { typedef Seq<I...> type; };
template <int... I> static void fill(LaunchFn* t, Seq<I...>) { LaunchFn f[] = {launch<I>...}; for (int i = 0; i < (int)sizeof...(I); ++i) t[i] = f[i]; }

int main(int argc, char** argv) {
    setvbuf(stdout, NULL, _IONBF, 0);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, simds = cus * 4;
    printf("device %s CUs %d nominal clock %d kHz; one workgroup = one wave; W waves per SIMD\n", prop.gcnArchName, cus, prop.clockRate);
    LaunchFn fn[N_OPS];
    fill(fn, Gen<N_OPS>::type());
    const int iters = 2048, per_iter = 32;
    const int max_blocks = simds * 8;
    u64* cyc; u32* sink;
    CHECK(hipMalloc(&cyc, max_blocks * 8)); CHECK(hipMalloc(&sink, max_blocks * 64 * 4));
    std::vector<u64> h(max_blocks);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-40s %7s %7s %7s %7s %7s   %s\n", "class", "W=1", "W=2", "W=3", "W=4", "W=8", "wall-clock cycles @nominal: W=3, W=8");
    for (int op = 0; op < N_OPS; ++op) {
        double res[5]; double wall8 = 0, wall3 = 0;
        int wi = 0;
        for (int W : {1, 2, 3, 4, 8}) {
            const int blocks = simds * W;
            fn[op](blocks, cyc, sink, 16);   // warm-up
            CHECK(hipDeviceSynchronize());
            double best = 1e30, bestwall = 1e30;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipEventRecord(e0));
                fn[op](blocks, cyc, sink, iters);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                CHECK(hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost));
                double sum = 0;
                for (int i = 0; i < blocks; ++i) sum += (double)h[i];
                double c = sum / blocks / ((double)W * iters * per_iter);
                if (c < best) best = c;
                double wc = ms * 1e-3 * (double)prop.clockRate * 1e3 / ((double)W * iters * per_iter);
                if (wc < bestwall) bestwall = wc;
            }
            res[wi++] = best;
            if (W == 8) wall8 = bestwall;
            if (W == 3) wall3 = bestwall;
        }
        printf("%-40s %7.2f %7.2f %7.2f %7.2f %7.2f   %.2f  %.2f\n", NAMES[op], res[0], res[1], res[2], res[3], res[4], wall3, wall8);
    }
    printf("(cycles of s_memtime per wave-instruction per SIMD; the W=8 column is the steady-state issue cost; a class at ~4 issues at 16 lanes per\n"
           " cycle, ~2 would be 32 lanes per cycle, ~8 is half rate, ~16 quarter rate.  If s_memtime does not tick at the shader clock on this part the\n"
           " last column, wall time x nominal clock, is the one to read.)\n");
    return 0;
}
