// tools/microbench.hip -- box calibration for the NTT kernel design.
// Measures (i) issue cost of the integer instructions a Harvey/Shoup butterfly is made of,
// (ii) streaming copy bandwidth for the access shapes the NTT kernels use.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/microbench tools/microbench.hip
#include <hip/hip_runtime.h>
#include "../agilex-ntt_amd/csrc/modarith.hpp"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include <string>
#include <atomic>
#include <chrono>
#include <fstream>
#include <thread>
#include <dirent.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

enum Op { ADD32, MAD64, MAD64_SGPR, MULLO, MULHI, MAD24, MULHI24, LSHLADD64, ADDC_PAIR, CMP_CND64, FMA64, DPP_MOV, PERMSWAP, BPERMUTE, SUB_PAIR, MUL24_E32, ADD32_E64, ADD3, CNDMASK_E32, CMP64_ONLY, CMP32_ONLY, XOR_E32, MOV_E32, LSHL_ADD_U32, NOPS, MULHI_MIX, MULLO_MIX };
static const char* op_name[] = {"v_add_u32", "v_mad_u64_u32", "v_mad_u64_u32(sgpr src)", "v_mul_lo_u32", "v_mul_hi_u32", "v_mad_u32_u24", "v_mul_hi_u32_u24",
    "v_lshl_add_u64", "v_add_co+v_addc_co (pair)", "v_cmp_ge_u64+2cndmask (triple)", "v_fma_f64", "v_mov_b32 dpp quad_perm", "v_permlane32_swap", "ds_bpermute_b32", "v_sub_co+v_subb_co (pair)", "v_mul_u32_u24_e32 (VOP2)", "v_add_u32_e64 (VOP3)", "v_add3_u32", "v_cndmask_b32_e32 (vcc)", "v_cmp_ge_u64 only", "v_cmp_lt_i32 only", "v_xor_b32_e32", "v_mov_b32_e32", "v_lshl_add_u32", "nops", "v_mul_hi_u32 + v_xor_b32 (operands stay random)", "v_mul_lo_u32 + v_xor_b32 (operands stay random)"};

// 8 independent chains per asm block, 2 blocks per loop iteration = 16 "units" per iteration
template <int OP>
__global__ void __launch_bounds__(256) alu_kernel(uint64_t* out, int iters, uint32_t sval) {
    uint32_t a = threadIdx.x * 2654435761u + 12345u, b = (threadIdx.x ^ 0x5a5a5a5au) | 1u;
    uint64_t c0 = a, c1 = b, c2 = a + 1, c3 = b + 2, c4 = a + 3, c5 = b + 4, c6 = a + 5, c7 = b + 6;
    double d0 = a, d1 = b, d2 = 1.5, d3 = 2.5, d4 = 3.5, d5 = 4.5, d6 = 5.5, d7 = 6.5, da = 1.0000001, db = 1e-9;
    uint32_t x0 = a, x1 = b, x2 = a ^ b, x3 = a + b, x4 = a - b, x5 = a * 3, x6 = b * 5, x7 = a * 7;
    uint32_t s = __builtin_amdgcn_readfirstlane(sval);
    uint64_t t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            if constexpr (OP == ADD32) {
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
            } else if constexpr (OP == MAD64) {
                asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                             "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(a), "v"(b) : "vcc");
            } else if constexpr (OP == MAD64_SGPR) {
                asm volatile("v_mad_u64_u32 %0, vcc, %8, %9, %0\n v_mad_u64_u32 %1, vcc, %8, %9, %1\n v_mad_u64_u32 %2, vcc, %8, %9, %2\n v_mad_u64_u32 %3, vcc, %8, %9, %3\n"
                             "v_mad_u64_u32 %4, vcc, %8, %9, %4\n v_mad_u64_u32 %5, vcc, %8, %9, %5\n v_mad_u64_u32 %6, vcc, %8, %9, %6\n v_mad_u64_u32 %7, vcc, %8, %9, %7\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "s"(s), "v"(b) : "vcc");
            } else if constexpr (OP == MULLO) {
                asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                             "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
            } else if constexpr (OP == MULHI) {
                asm volatile("v_mul_hi_u32 %0, %0, %8\n v_mul_hi_u32 %1, %1, %8\n v_mul_hi_u32 %2, %2, %8\n v_mul_hi_u32 %3, %3, %8\n"
                             "v_mul_hi_u32 %4, %4, %8\n v_mul_hi_u32 %5, %5, %8\n v_mul_hi_u32 %6, %6, %8\n v_mul_hi_u32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
            } else if constexpr (OP == MULHI_MIX || OP == MULLO_MIX) {
                // x <- mul(x, b) ^ a: the xor with a per-lane random word keeps the multiplier's operands full-width and changing (a bare
                // x <- mul_hi(x, b) chain decays to zero within a few steps and measures a multiplier that toggles nothing)
#define AGX_MIX(OPC) asm volatile(OPC " %0, %0, %8\n v_xor_b32 %0, %0, %9\n " OPC " %1, %1, %8\n v_xor_b32 %1, %1, %9\n " OPC " %2, %2, %8\n v_xor_b32 %2, %2, %9\n " OPC " %3, %3, %8\n v_xor_b32 %3, %3, %9\n" \
                             OPC " %4, %4, %8\n v_xor_b32 %4, %4, %9\n " OPC " %5, %5, %8\n v_xor_b32 %5, %5, %9\n " OPC " %6, %6, %8\n v_xor_b32 %6, %6, %9\n " OPC " %7, %7, %8\n v_xor_b32 %7, %7, %9\n" \
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a))
                if constexpr (OP == MULHI_MIX) AGX_MIX("v_mul_hi_u32");
                else AGX_MIX("v_mul_lo_u32");
#undef AGX_MIX
            } else if constexpr (OP == MAD24) {
                asm volatile("v_mad_u32_u24 %0, %0, %8, %9\n v_mad_u32_u24 %1, %1, %8, %9\n v_mad_u32_u24 %2, %2, %8, %9\n v_mad_u32_u24 %3, %3, %8, %9\n"
                             "v_mad_u32_u24 %4, %4, %8, %9\n v_mad_u32_u24 %5, %5, %8, %9\n v_mad_u32_u24 %6, %6, %8, %9\n v_mad_u32_u24 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));
            } else if constexpr (OP == MULHI24) {
                asm volatile("v_mul_hi_u32_u24 %0, %0, %8\n v_mul_hi_u32_u24 %1, %1, %8\n v_mul_hi_u32_u24 %2, %2, %8\n v_mul_hi_u32_u24 %3, %3, %8\n"
                             "v_mul_hi_u32_u24 %4, %4, %8\n v_mul_hi_u32_u24 %5, %5, %8\n v_mul_hi_u32_u24 %6, %6, %8\n v_mul_hi_u32_u24 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
            } else if constexpr (OP == LSHLADD64) {
                asm volatile("v_lshl_add_u64 %0, %0, 0, %8\n v_lshl_add_u64 %1, %1, 0, %8\n v_lshl_add_u64 %2, %2, 0, %8\n v_lshl_add_u64 %3, %3, 0, %8\n"
                             "v_lshl_add_u64 %4, %4, 0, %8\n v_lshl_add_u64 %5, %5, 0, %8\n v_lshl_add_u64 %6, %6, 0, %8\n v_lshl_add_u64 %7, %7, 0, %8\n"
                             : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7) : "v"(c0 | 1));
            } else if constexpr (OP == ADDC_PAIR) {
                asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n"
                             "v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "vcc");
            } else if constexpr (OP == SUB_PAIR) {
                asm volatile("v_sub_co_u32 %0, vcc, %0, %8\n v_subb_co_u32 %1, vcc, %1, %8, vcc\n v_sub_co_u32 %2, vcc, %2, %8\n v_subb_co_u32 %3, vcc, %3, %8, vcc\n"
                             "v_sub_co_u32 %4, vcc, %4, %8\n v_subb_co_u32 %5, vcc, %5, %8, vcc\n v_sub_co_u32 %6, vcc, %6, %8\n v_subb_co_u32 %7, vcc, %7, %8, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "vcc");
            } else if constexpr (OP == CMP_CND64) {
                // 4 x (64-bit compare + two selects) per block: count as 4 triples = 12 instr; reported per instruction
                asm volatile("v_cmp_ge_u64 vcc, %[c0], %[c1]\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n"
                             "v_cmp_ge_u64 vcc, %[c2], %[c3]\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n"
                             "v_cmp_ge_u64 vcc, %[c4], %[c5]\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n"
                             "v_cmp_ge_u64 vcc, %[c6], %[c7]\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7)
                             : "v"(b), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3), [c4] "v"(c4), [c5] "v"(c5), [c6] "v"(c6), [c7] "v"(c7) : "vcc");
            } else if constexpr (OP == FMA64) {
                asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                             "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                             : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(da), "v"(db));
            } else if constexpr (OP == DPP_MOV) {
                asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if constexpr (OP == PERMSWAP) {
                asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                             "v_permlane32_swap_b32 %1, %2\n v_permlane32_swap_b32 %3, %4\n v_permlane32_swap_b32 %5, %6\n v_permlane32_swap_b32 %7, %0\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if constexpr (OP == BPERMUTE) {
                uint32_t addr = ((threadIdx.x ^ 1) & 63) << 2;
                asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n"
                             "ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(addr));
            } else if constexpr (OP == MUL24_E32) {
                asm volatile("v_mul_u32_u24_e32 %0, %0, %8\n v_mul_u32_u24_e32 %1, %1, %8\n v_mul_u32_u24_e32 %2, %2, %8\n v_mul_u32_u24_e32 %3, %3, %8\n"
                             "v_mul_u32_u24_e32 %4, %4, %8\n v_mul_u32_u24_e32 %5, %5, %8\n v_mul_u32_u24_e32 %6, %6, %8\n v_mul_u32_u24_e32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
            } else if constexpr (OP == ADD32_E64) {
                asm volatile("v_add_u32_e64 %0, %0, %8\n v_add_u32_e64 %1, %1, %8\n v_add_u32_e64 %2, %2, %8\n v_add_u32_e64 %3, %3, %8\n"
                             "v_add_u32_e64 %4, %4, %8\n v_add_u32_e64 %5, %5, %8\n v_add_u32_e64 %6, %6, %8\n v_add_u32_e64 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
            } else if constexpr (OP == ADD3) {
                asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n"
                             "v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b), "v"(a));
            } else if constexpr (OP == CNDMASK_E32) {
                asm volatile("v_cndmask_b32_e32 %0, %0, %8, vcc\n v_cndmask_b32_e32 %1, %1, %8, vcc\n v_cndmask_b32_e32 %2, %2, %8, vcc\n v_cndmask_b32_e32 %3, %3, %8, vcc\n"
                             "v_cndmask_b32_e32 %4, %4, %8, vcc\n v_cndmask_b32_e32 %5, %5, %8, vcc\n v_cndmask_b32_e32 %6, %6, %8, vcc\n v_cndmask_b32_e32 %7, %7, %8, vcc\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b) : "vcc");
            } else if constexpr (OP == CMP64_ONLY) {
                asm volatile("v_cmp_ge_u64 vcc, %0, %1\n v_cmp_ge_u64 vcc, %2, %3\n v_cmp_ge_u64 vcc, %4, %5\n v_cmp_ge_u64 vcc, %6, %7\n"
                             "v_cmp_ge_u64 vcc, %1, %2\n v_cmp_ge_u64 vcc, %3, %4\n v_cmp_ge_u64 vcc, %5, %6\n v_cmp_ge_u64 vcc, %7, %0\n"
                             :: "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(c4), "v"(c5), "v"(c6), "v"(c7) : "vcc");
            } else if constexpr (OP == CMP32_ONLY) {
                asm volatile("v_cmp_lt_i32 vcc, %0, %1\n v_cmp_lt_i32 vcc, %2, %3\n v_cmp_lt_i32 vcc, %4, %5\n v_cmp_lt_i32 vcc, %6, %7\n"
                             "v_cmp_lt_i32 vcc, %1, %2\n v_cmp_lt_i32 vcc, %3, %4\n v_cmp_lt_i32 vcc, %5, %6\n v_cmp_lt_i32 vcc, %7, %0\n"
                             :: "v"(x0), "v"(x1), "v"(x2), "v"(x3), "v"(x4), "v"(x5), "v"(x6), "v"(x7) : "vcc");
            } else if constexpr (OP == XOR_E32) {
                asm volatile("v_xor_b32_e32 %0, %0, %8\n v_xor_b32_e32 %1, %1, %8\n v_xor_b32_e32 %2, %2, %8\n v_xor_b32_e32 %3, %3, %8\n"
                             "v_xor_b32_e32 %4, %4, %8\n v_xor_b32_e32 %5, %5, %8\n v_xor_b32_e32 %6, %6, %8\n v_xor_b32_e32 %7, %7, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
            } else if constexpr (OP == MOV_E32) {
                asm volatile("v_mov_b32_e32 %0, %1\n v_mov_b32_e32 %1, %2\n v_mov_b32_e32 %2, %3\n v_mov_b32_e32 %3, %4\n"
                             "v_mov_b32_e32 %4, %5\n v_mov_b32_e32 %5, %6\n v_mov_b32_e32 %6, %7\n v_mov_b32_e32 %7, %0\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7));
            } else if constexpr (OP == LSHL_ADD_U32) {
                asm volatile("v_lshl_add_u32 %0, %0, 1, %8\n v_lshl_add_u32 %1, %1, 1, %8\n v_lshl_add_u32 %2, %2, 1, %8\n v_lshl_add_u32 %3, %3, 1, %8\n"
                             "v_lshl_add_u32 %4, %4, 1, %8\n v_lshl_add_u32 %5, %5, 1, %8\n v_lshl_add_u32 %6, %6, 1, %8\n v_lshl_add_u32 %7, %7, 1, %8\n"
                             : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(b));
            }
        }
    }
    uint64_t t1 = __builtin_amdgcn_s_memtime();
    uint64_t sink = c0 ^ c1 ^ c2 ^ c3 ^ c4 ^ c5 ^ c6 ^ c7 ^ x0 ^ x1 ^ x2 ^ x3 ^ x4 ^ x5 ^ x6 ^ x7 ^ (uint64_t)(d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7);
    size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if ((threadIdx.x & 63) == 0) out[wave] = t1 - t0;
    if (sink == 0x1234567890abcdefull) out[wave] = sink;  // keep values live
}

template <int OP>
static void run_alu(uint64_t* d_out, int waves_per_simd) {
    const int iters = 4096;
    const int blocks = 256 * waves_per_simd;
    std::vector<uint64_t> h(blocks * 4);
    alu_kernel<OP><<<blocks, 256>>>(d_out, 64, 7u);  // warm
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    alu_kernel<OP><<<blocks, 256>>>(d_out, iters, 7u);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    CK(hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    double med = (double)h[h.size() / 2];
    double instr_per_wave = (double)iters * 16.0 * (OP == CMP_CND64 ? 12.0 / 16.0 : 1.0);
    // s_memtime ticks at a constant 100 MHz-derived rate on some parts; report both tick-based and wall-based figures
    double wall_cyc_at_2p4 = ms * 1e-3 * 2.4e9;
    printf("  %-34s waves/SIMD=%d  ticks/instr/wave=%7.3f  ticks/instr/SIMD=%7.3f  wall: %8.3f ms -> %6.3f clk(2.4GHz)/instr/SIMD\n",
           op_name[OP], waves_per_simd, med / instr_per_wave, med / instr_per_wave / waves_per_simd, ms,
           wall_cyc_at_2p4 / instr_per_wave / waves_per_simd);
}

// ---------------------------------------------------------------- bandwidth
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) copy16(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n16) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) out[i] = in[i];
}
__global__ void __launch_bounds__(256) copy8(const u32x2* __restrict__ in, u32x2* __restrict__ out, size_t n8) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += stride) out[i] = in[i];
}
// one block per 32 KiB slab (the n=4096 frame shape): every thread loads 8 x 16 B up front then stores
__global__ void __launch_bounds__(256) copy_slab(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t slabs) {
    for (size_t s = blockIdx.x; s < slabs; s += gridDim.x) {
        const u32x4* p = in + s * 2048; u32x4* o = out + s * 2048;
        u32x4 r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = p[threadIdx.x + 256 * k];
#pragma unroll
        for (int k = 0; k < 8; ++k) o[threadIdx.x + 256 * k] = r[k];
    }
}
// the same frame-shaped traffic in the NTT kernel's own access width and cache policy: 512 threads, 8 x 8-byte loads and stores
// per lane (element tid + 512 r), NT = non-temporal loads and stores, INPLACE = results overwrite the frame
template <bool NT>
__global__ void __launch_bounds__(512) copy_frame8(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, size_t frames) {
    for (size_t s = blockIdx.x; s < frames; s += gridDim.x) {
        const uint64_t* p = in + s * 4096; uint64_t* o = out + s * 4096;
        uint64_t r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = NT ? __builtin_nontemporal_load(p + threadIdx.x + 512 * k) : p[threadIdx.x + 512 * k];
#pragma unroll
        for (int k = 0; k < 8; ++k) { if (NT) __builtin_nontemporal_store(r[k], o + threadIdx.x + 512 * k); else o[threadIdx.x + 512 * k] = r[k]; }
    }
}
// reads coalesced, stores 16 B per lane at a lane stride of LS bytes (thread-contiguous runs): the
// store shape of a register-blocked last pass that keeps 2^R consecutive coefficients per lane
template <int LS>
__global__ void __launch_bounds__(256) copy_lane_strided_store(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t slabs) {
    constexpr int PER = LS / 16;             // 16-B chunks per lane run
    constexpr int LANES = 2048 / PER;        // lanes needed per 32 KiB slab
    for (size_t s = blockIdx.x; s < slabs; s += gridDim.x) {
        const u32x4* p = in + s * 2048; u32x4* o = out + s * 2048;
        if (LANES >= 256) {
            u32x4 r[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = p[threadIdx.x + 256 * k];
#pragma unroll
            for (int k = 0; k < 8; ++k) o[threadIdx.x * 8 + k] = r[k];
        } else if ((int)threadIdx.x < LANES) {
            u32x4 r[PER];
#pragma unroll
            for (int k = 0; k < PER; ++k) r[k] = p[threadIdx.x + LANES * k];
#pragma unroll
            for (int k = 0; k < PER; ++k) o[threadIdx.x * PER + k] = r[k];
        }
    }
}

template <typename F>
static void time_bw(const char* name, size_t bytes_moved, F launch) {
    launch(); CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 10;
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  %-44s %8.1f GB/s (read+write)\n", name, (double)bytes_moved * reps / (ms * 1e-3) / 1e9);
}


// ---------------------------------------------------------------- butterfly ALU roofline
// The product's own butterfly forms (csrc/modarith.hpp) on register-resident coefficients, no
// memory traffic inside the loop: what the VALU alone can sustain, i.e. the ceiling the NTT kernels
// are measured against.  One radix-8 pass (12 butterflies on 8 coefficients) per iteration.
template <int FORM, bool SGPR_TW>   // FORM 0 exact, 1 fast, 2 16q-lazy (every 12th..: 5 of 12 butterflies subtract), 3 = 2 with 32-bit cross products
__global__ void __launch_bounds__(512, 8) bf_kernel(uint64_t* out, const uint64_t* tw, uint64_t q, int iters) {
    using namespace agx;
    bf_consts k;
    k.q = q;
    k.nq = 0 - q;
    k.m = FORM == 0 ? (q << 1) : (q << 2);
    k.nm = opaque_sgpr64(0 - k.m);
    k.one_a = opaque_one<0>();
    k.one_b = opaque_one<1>();
    final_consts fc;
    fc.q2 = q << 1; fc.nq2 = opaque_sgpr64(0 - fc.q2); fc.q1 = q; fc.nq1 = opaque_sgpr64(0 - q);
    fc.q8 = q << 3; fc.nq8 = opaque_sgpr64(0 - fc.q8);
    uint64_t x[8];
    for (int r = 0; r < 8; ++r) x[r] = (tw[(threadIdx.x * 8 + r) & 1023] >> 3);
    // per-lane variant: 3 distinct twiddle pairs per thread stay in VGPRs (12 registers), as many as
    // the kernels keep live at a time; wave-uniform variant: 7 pairs in SGPRs
    constexpr int NTW = SGPR_TW ? 7 : 3;
    uint64_t w[NTW], wp[NTW];
    for (int j = 0; j < NTW; ++j) {
        const size_t idx = SGPR_TW ? (size_t)(j * 2) : (size_t)((threadIdx.x & 63) * 16 + j * 2);
        w[j] = tw[idx] % q;
        wp[j] = tw[idx + 1];
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int stage = 0; stage < 3; ++stage) {
            const int rb = 2 - stage;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int r0 = ((b >> rb) << (rb + 1)) | (b & ((1 << rb) - 1)), r1 = r0 | (1 << rb);
                const int j = ((1 << stage) - 1 + (r0 >> (rb + 1))) % NTW;
                if constexpr (FORM == 0) ct_butterfly_exact(x[r0], x[r1], w[j], wp[j], k);
                else if constexpr (FORM == 1) ct_butterfly_fast<true>(x[r0], x[r1], w[j], wp[j], k);
                else if constexpr (FORM == 3) {     // 16q-lazy with the cross products as 32-bit multiplies (energy A/B)
                    if (stage == 1 || (stage == 2 && b == 0)) ct_butterfly_lazy16<true, true, false, 1>(x[r0], x[r1], w[j], wp[j], k, fc);
                    else ct_butterfly_lazy16<true, false, false, 1>(x[r0], x[r1], w[j], wp[j], k, fc);
                }
                else if (stage == 1 || (stage == 2 && b == 0)) ct_butterfly_lazy16<true, true>(x[r0], x[r1], w[j], wp[j], k, fc);   // 5 of 12
                else ct_butterfly_lazy16<true, false>(x[r0], x[r1], w[j], wp[j], k, fc);
            }
        }
        if constexpr (FORM >= 2) {   // keep the values inside the form's range between iterations (not counted as butterfly work)
#pragma unroll
            for (int r = 0; r < 8; ++r) x[r] &= 0x0fffffffffffffffull;
        }
    }
    uint64_t acc = 0;
    for (int r = 0; r < 8; ++r) acc ^= x[r];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int FORM, bool SGPR_TW>
static void run_bf(const char* name, uint64_t* d_out, const uint64_t* d_tw, uint64_t q) {
    const int iters = 2000, blocks = 256 * 4;   // 4 workgroups of 512 threads per CU = 8 waves/SIMD
    bf_kernel<FORM, SGPR_TW><<<blocks, 512>>>(d_out, d_tw, q, 10);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    // burst = the first 5 launches after idle; sustained = 150 launches back to back (the power-managed
    // clock comes down within a few tens of milliseconds of continuous integer-multiply work)
    for (int phase = 0; phase < 2; ++phase) {
        const int reps = phase == 0 ? 5 : 150;
        CK(hipEventRecord(e0));
        for (int rep = 0; rep < reps; ++rep) bf_kernel<FORM, SGPR_TW><<<blocks, 512>>>(d_out, d_tw, q, iters);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double bfs = (double)reps * blocks * 512.0 * iters * 12.0;           // lane-butterflies
        const double per_s = bfs / (ms * 1e-3);
        printf("  %-44s %-9s %7.2f G butterflies/s  -> n=4096 NTT ceiling %6.1f M NTT/s (%4.1f %% of the 8 TB/s roofline)\n", name,
               phase == 0 ? "burst" : "sustained", per_s / 1e9, per_s / 24576.0 / 1e6, per_s / 24576.0 * 65536.0 / 8e12 * 100.0);
    }
}

// ---- workgroup dispatch: the NTT launch shape (16,384 workgroups x 512 threads, 34 KiB LDS) with waves
// that only wait `ticks` of s_memtime.  Ideal duration = rounds x wait; the excess is what the dispatcher
// needs to refill a slot after a workgroup retires.
extern __shared__ unsigned char mb_dyn_lds[];
// MODE 0: sleep; 1: multiply-add busy loop; 2: sleep, then the frame's stores; 3: the frame's loads, sleep, stores
template <int MODE>
__global__ void __launch_bounds__(512, 8) wait_kernel(uint64_t ticks, uint64_t* slab, uint32_t* sink) {
    const uint64_t t0 = __builtin_readcyclecounter();
    uint64_t* mine = slab + ((size_t)(blockIdx.x & 16383) << 12);
    uint64_t v[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    if constexpr (MODE == 3) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = mine[threadIdx.x + 512 * r];
    }
    if (ticks) {
        if constexpr (MODE == 1) {
            uint64_t a = threadIdx.x, b = t0 | 1;
            while (__builtin_readcyclecounter() - t0 < ticks) {
#pragma unroll
                for (int k = 0; k < 32; ++k) a = (uint64_t)(uint32_t)a * (uint32_t)b + a;
            }
            v[0] += a;
        } else {
            while (__builtin_readcyclecounter() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
        }
    }
    if constexpr (MODE >= 2) {
#pragma unroll
        for (int r = 0; r < 8; ++r) mine[threadIdx.x + 512 * r] = v[r] + 1;
    }
    if (sink && threadIdx.x == 1023) *sink = mb_dyn_lds[0] + (uint32_t)v[0];
}

template <int MODE>
static void run_dispatch_mode(const char* name, uint64_t* slab) {
    const int lds = 34816, wgs = 16384;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&wait_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf(" %s\n", name);
    double base = 0;
    for (uint64_t ticks : {0ull, 12500ull, 25000ull, 50000ull}) {
        for (int i = 0; i < 30; ++i) wait_kernel<MODE><<<wgs, 512, lds>>>(ticks, slab, nullptr);
        CK(hipEventRecord(e0));
        const int reps = 30;
        for (int i = 0; i < reps; ++i) wait_kernel<MODE><<<wgs, 512, lds>>>(ticks, slab, nullptr);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps;
        if (ticks == 0) base = us;
        printf("  wave lifetime %6llu ticks: %8.1f us per launch = %6.3f us per round of 1024 workgroups\n", (unsigned long long)ticks, us, us / 16.0);
    }
    (void)base;
}

static void run_dispatch() {
    printf("workgroup dispatch, 16384 workgroups x 512 threads, 34816 B LDS (4 per CU, 16 rounds on 256 CUs);\n"
           "the slope between rows is the tick length, the intercept what a slot needs to turn around\n");
    uint64_t* slab; CK(hipMalloc(&slab, (size_t)16384 * 32768)); CK(hipMemset(slab, 0, (size_t)16384 * 32768));
    run_dispatch_mode<0>("sleeping waves", slab);
    run_dispatch_mode<1>("waves in a multiply-add loop", slab);
    run_dispatch_mode<2>("sleeping waves that store their 32 KiB frame before they end", slab);
    run_dispatch_mode<3>("waves that load their frame, sleep, store it", slab);
    CK(hipFree(slab));
}

// ---------------------------------------------------------------- power / clock probe
// What the board draws and what clock it holds while ONE kind of work runs for ~1.2 s: the pure instruction loops, the
// butterfly loop (no memory traffic), the frame-shaped copy (no arithmetic).  Socket power and shader clock come from
// amdgpu's sysfs files of this device, sampled every 20 ms by a host thread (the same files bench.py reads).
struct SysfsProbe {
    std::string power_file, sclk_file;
    double cap_w = 0;
    explicit SysfsProbe(int dev) {
        char bdf[64] = {0};
        if (hipDeviceGetPCIBusId(bdf, sizeof(bdf), dev) != hipSuccess) return;
        std::string base = std::string("/sys/bus/pci/devices/") + bdf;
        for (char& c : base) c = (char)tolower(c);
        sclk_file = base + "/pp_dpm_sclk";
        std::string hw = base + "/hwmon";
        if (DIR* d = opendir(hw.c_str())) {
            while (dirent* e = readdir(d)) {
                if (strncmp(e->d_name, "hwmon", 5)) continue;
                for (const char* name : {"power1_average", "power1_input"}) {
                    std::string f = hw + "/" + e->d_name + "/" + name;
                    if (power_file.empty() && std::ifstream(f).good()) power_file = f;
                }
                std::ifstream cap(hw + "/" + e->d_name + "/power1_cap");
                double v;
                if (cap >> v) cap_w = v / 1e6;
            }
            closedir(d);
        }
    }
    void read(double& w, int& mhz) const {
        w = 0; mhz = 0;
        std::ifstream p(power_file);
        double v;
        if (p >> v) w = v / 1e6;
        std::ifstream s(sclk_file);
        std::string line;
        while (std::getline(s, line))
            if (line.find('*') != std::string::npos) { size_t c = line.find(':'); mhz = atoi(line.c_str() + c + 1); }
    }
};

template <typename F>
static void power_phase(const SysfsProbe& probe, const char* name, F launch, double units_per_launch, const char* unit) {
    launch(); CK(hipDeviceSynchronize());
    std::atomic<bool> stop{false};
    std::vector<double> watts; std::vector<int> mhz;
    std::thread sampler([&] {
        std::this_thread::sleep_for(std::chrono::milliseconds(300));     // skip the ramp out of idle
        while (!stop.load()) { double w; int m; probe.read(w, m); if (w > 0) watts.push_back(w); if (m > 0) mhz.push_back(m); std::this_thread::sleep_for(std::chrono::milliseconds(20)); }
    });
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const auto t0 = std::chrono::steady_clock::now();
    int launches = 0;
    CK(hipEventRecord(e0));
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 1.2) {
        for (int i = 0; i < 8; ++i) launch();
        launches += 8;
        CK(hipStreamSynchronize(0));
    }
    CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
    stop = true; sampler.join();
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    std::sort(watts.begin(), watts.end()); std::sort(mhz.begin(), mhz.end());
    printf("  %-52s %8.1f W median (%4zu samples)  sclk %4d MHz   %9.3f %s\n", name, watts.empty() ? 0.0 : watts[watts.size() / 2], watts.size(),
           mhz.empty() ? 0 : mhz[mhz.size() / 2], units_per_launch * launches / (ms * 1e-3) / 1e9, unit);
}

static void run_power() {
    SysfsProbe probe(0);
    printf("power / clock by kind of work (sysfs %s; cap %.0f W; 1.2 s per line, samples after the first 0.3 s)\n", probe.power_file.c_str(), probe.cap_w);
    uint64_t* d_out; CK(hipMalloc(&d_out, 256 * 8 * 4 * 8 * 2));
    const int blocks = 256 * 8, iters = 20000;    // 8 waves/SIMD of 256-thread workgroups
    const double lane_ops = (double)blocks * 256 * iters * 16;
    power_phase(probe, "v_add_u32 loop, 8 waves/SIMD", [&] { alu_kernel<ADD32><<<blocks, 256>>>(d_out, iters, 12345u); }, lane_ops, "G lane-ops/s");
    power_phase(probe, "v_mad_u64_u32 loop, 8 waves/SIMD", [&] { alu_kernel<MAD64><<<blocks, 256>>>(d_out, iters, 12345u); }, lane_ops, "G lane-ops/s");
    power_phase(probe, "v_xor_b32 loop, 8 waves/SIMD", [&] { alu_kernel<XOR_E32><<<blocks, 256>>>(d_out, iters, 12345u); }, lane_ops, "G lane-ops/s");
    power_phase(probe, "v_mul_hi_u32 + v_xor_b32 (random operands), 8 waves/SIMD", [&] { alu_kernel<MULHI_MIX><<<blocks, 256>>>(d_out, iters, 12345u); }, lane_ops, "G pairs/s");
    power_phase(probe, "v_mul_lo_u32 + v_xor_b32 (random operands), 8 waves/SIMD", [&] { alu_kernel<MULLO_MIX><<<blocks, 256>>>(d_out, iters, 12345u); }, lane_ops, "G pairs/s");
    power_phase(probe, "v_mul_hi_u32 x <- hi(x b) chain (DECAYS TO ZERO operands: not a multiplier figure)", [&] { alu_kernel<MULHI><<<blocks, 256>>>(d_out, iters, 12345u); }, lane_ops, "G lane-ops/s");
    power_phase(probe, "v_lshl_add_u64 loop, 8 waves/SIMD", [&] { alu_kernel<LSHLADD64><<<blocks, 256>>>(d_out, iters, 12345u); }, lane_ops, "G lane-ops/s");
    power_phase(probe, "v_mad_u64_u32 loop, 2 waves/SIMD", [&] { alu_kernel<MAD64><<<256 * 2, 256>>>(d_out, iters, 12345u); }, lane_ops / 4, "G lane-ops/s");
    CK(hipFree(d_out));
    {
        const uint64_t q = 1152921504606830593ull;
        std::vector<uint64_t> h(1024);
        uint64_t st = 42;
        for (auto& v : h) { st = st * 6364136223846793005ull + 1442695040888963407ull; v = st; }
        uint64_t *d_tw, *d_o;
        CK(hipMalloc(&d_tw, h.size() * 8)); CK(hipMalloc(&d_o, 256 * 4 * 512 * 8));
        CK(hipMemcpy(d_tw, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        const int bl = 256 * 4, it = 2000;
        power_phase(probe, "exact butterflies on registers (random data), 8 waves/SIMD", [&] { bf_kernel<0, false><<<bl, 512>>>(d_o, d_tw, q, it); }, (double)bl * 512 * it * 12, "G butterflies/s");
        power_phase(probe, "fast (8q-lazy) butterflies, same", [&] { bf_kernel<1, false><<<bl, 512>>>(d_o, d_tw, q, it); }, (double)bl * 512 * it * 12, "G butterflies/s");
        power_phase(probe, "16q-lazy butterflies on registers (random data), 8 waves/SIMD", [&] { bf_kernel<2, false><<<bl, 512>>>(d_o, d_tw, q, it); }, (double)bl * 512 * it * 12, "G butterflies/s");
        power_phase(probe, "16q-lazy, wave-uniform (SGPR) twiddles", [&] { bf_kernel<2, true><<<bl, 512>>>(d_o, d_tw, q, it); }, (double)bl * 512 * it * 12, "G butterflies/s");
        power_phase(probe, "16q-lazy, cross products as 32-bit v_mul_lo_u32", [&] { bf_kernel<3, false><<<bl, 512>>>(d_o, d_tw, q, it); }, (double)bl * 512 * it * 12, "G butterflies/s");
        std::fill(h.begin(), h.end(), 0ull);
        CK(hipMemcpy(d_tw, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        power_phase(probe, "the same on all-zero data and twiddles", [&] { bf_kernel<2, false><<<bl, 512>>>(d_o, d_tw, q, it); }, (double)bl * 512 * it * 12, "G butterflies/s");
        CK(hipFree(d_tw)); CK(hipFree(d_o));
    }
    {
        const size_t bytes = 2ull << 30;
        void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
        power_phase(probe, "copy, one workgroup per 32 KiB frame (no arithmetic)", [&] { copy_slab<<<(unsigned)(bytes / 32768), 256>>>((const u32x4*)a, (u32x4*)b, bytes / 32768); }, 2.0 * bytes, "GB/s read+write");
        const unsigned frames = (unsigned)(bytes / 32768);
        power_phase(probe, "copy in the NTT's shape: 512 threads, 8-byte accesses, plain", [&] { copy_frame8<false><<<frames, 512>>>((const uint64_t*)a, (uint64_t*)b, frames); }, 2.0 * bytes, "GB/s read+write");
        power_phase(probe, "the same with non-temporal loads and stores", [&] { copy_frame8<true><<<frames, 512>>>((const uint64_t*)a, (uint64_t*)b, frames); }, 2.0 * bytes, "GB/s read+write");
        power_phase(probe, "the same, non-temporal, in place", [&] { copy_frame8<true><<<frames, 512>>>((const uint64_t*)a, (uint64_t*)a, frames); }, 2.0 * bytes, "GB/s read+write");
        CK(hipFree(a)); CK(hipFree(b));
    }
}

int main(int argc, char** argv) {
    if (argc > 1 && std::string(argv[1]) == "pwr") { run_power(); return 0; }
    if (argc > 1 && std::string(argv[1]) == "disp") { run_dispatch(); return 0; }
    bool do_alu = true, do_bw = true, do_bf = true;
    if (argc > 1 && std::string(argv[1]) == "alu") do_bw = do_bf = false;
    if (argc > 1 && std::string(argv[1]) == "bw") do_alu = do_bf = false;
    if (argc > 1 && std::string(argv[1]) == "bf") do_alu = do_bw = false;
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz  L2=%d  LDS/block=%zu\n", prop.name, prop.multiProcessorCount, prop.clockRate, prop.l2CacheSize, prop.sharedMemPerBlock);
    if (do_alu) {
        uint64_t* d_out; CK(hipMalloc(&d_out, 256 * 8 * 4 * 8 * 2));
        printf("ALU issue cost (16 independent chains per loop body; lower = faster)\n");
        for (int w : {1, 2, 4, 8}) {
            run_alu<ADD32>(d_out, w); run_alu<MAD64>(d_out, w); run_alu<MAD64_SGPR>(d_out, w); run_alu<MULLO>(d_out, w); run_alu<MULHI>(d_out, w);
            run_alu<MAD24>(d_out, w); run_alu<MULHI24>(d_out, w); run_alu<LSHLADD64>(d_out, w); run_alu<ADDC_PAIR>(d_out, w); run_alu<SUB_PAIR>(d_out, w);
            run_alu<CMP_CND64>(d_out, w); run_alu<FMA64>(d_out, w); run_alu<DPP_MOV>(d_out, w); run_alu<PERMSWAP>(d_out, w); run_alu<BPERMUTE>(d_out, w);
            run_alu<MUL24_E32>(d_out, w); run_alu<ADD32_E64>(d_out, w); run_alu<ADD3>(d_out, w); run_alu<CNDMASK_E32>(d_out, w); run_alu<CMP64_ONLY>(d_out, w); run_alu<CMP32_ONLY>(d_out, w); run_alu<XOR_E32>(d_out, w); run_alu<MOV_E32>(d_out, w); run_alu<LSHL_ADD_U32>(d_out, w);
            printf("\n");
        }
        CK(hipFree(d_out));
    }
    if (do_bf) {
        const uint64_t q = 1152921504606830593ull;   // largest 60-bit prime = 1 mod 8192
        std::vector<uint64_t> h(1024);
        uint64_t st = 42;
        for (auto& v : h) { st = st * 6364136223846793005ull + 1442695040888963407ull; v = st; }
        uint64_t *d_tw, *d_o;
        CK(hipMalloc(&d_tw, h.size() * 8)); CK(hipMalloc(&d_o, 256 * 4 * 512 * 8));
        CK(hipMemcpy(d_tw, h.data(), h.size() * 8, hipMemcpyHostToDevice));
        printf("butterfly ALU roofline: register-resident radix-8 passes, 8 waves/SIMD, no memory traffic in the loop\n");
        run_bf<0, false>("exact form, per-lane twiddles", d_o, d_tw, q);
        run_bf<1, false>("fast form (q<=2^61), per-lane twiddles", d_o, d_tw, q);
        run_bf<2, false>("16q-lazy form (q<=2^60), per-lane twiddles", d_o, d_tw, q);
        run_bf<2, true>("16q-lazy form, wave-uniform twiddles", d_o, d_tw, q);
        CK(hipFree(d_tw)); CK(hipFree(d_o));
    }
    if (do_bw) {
        const size_t bytes = 2ull << 30;  // 2 GiB in + 2 GiB out: far beyond the 256 MiB Infinity Cache
        void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
        CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
        printf("streaming copy, 2 GiB -> 2 GiB\n");
        for (int g : {2048, 4096, 8192})
            time_bw(("copy16 grid-stride grid=" + std::to_string(g)).c_str(), 2 * bytes, [&] { copy16<<<g, 256>>>((const u32x4*)a, (u32x4*)b, bytes / 16); });
        time_bw("copy8 grid-stride grid=4096", 2 * bytes, [&] { copy8<<<4096, 256>>>((const u32x2*)a, (u32x2*)b, bytes / 8); });
        size_t slabs = bytes / 32768;
        time_bw("copy_slab (1 block per 32 KiB slab)", 2 * bytes, [&] { copy_slab<<<(unsigned)slabs, 256>>>((const u32x4*)a, (u32x4*)b, slabs); });
        time_bw("copy_slab grid=2048 looped", 2 * bytes, [&] { copy_slab<<<2048, 256>>>((const u32x4*)a, (u32x4*)b, slabs); });
        time_bw("store lane-stride 128 B (16 coeff/lane)", 2 * bytes, [&] { copy_lane_strided_store<128><<<(unsigned)slabs, 256>>>((const u32x4*)a, (u32x4*)b, slabs); });
        time_bw("store lane-stride 256 B (32 coeff/lane)", 2 * bytes, [&] { copy_lane_strided_store<256><<<(unsigned)slabs, 256>>>((const u32x4*)a, (u32x4*)b, slabs); });
        time_bw("store lane-stride 512 B (64 coeff/lane)", 2 * bytes, [&] { copy_lane_strided_store<512><<<(unsigned)slabs, 256>>>((const u32x4*)a, (u32x4*)b, slabs); });
        CK(hipFree(a)); CK(hipFree(b));
    }
    return 0;
}
