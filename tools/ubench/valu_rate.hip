// issue cost of vector instructions, one timed wave (the oldest of its SIMD), gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void k(uint32_t* out, uint64_t* cyc, uint32_t seed) {
    uint32_t a0 = threadIdx.x ^ seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const uint32_t c = seed | 1u;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < 64; ++it) {
        if (OP == 0) { REP8(asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 1) { REP8(asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 2) { REP8(asm volatile("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n v_lshrrev_b32 %4, 3, %4\n v_lshrrev_b32 %5, 3, %5\n v_lshrrev_b32 %6, 3, %6\n v_lshrrev_b32 %7, 3, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 3) { REP8(asm volatile("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 4) { REP8(asm volatile("v_max3_f32 %0, %0, %8, %8\n v_max3_f32 %1, %1, %8, %8\n v_max3_f32 %2, %2, %8, %8\n v_max3_f32 %3, %3, %8, %8\n v_max3_f32 %4, %4, %8, %8\n v_max3_f32 %5, %5, %8, %8\n v_max3_f32 %6, %6, %8, %8\n v_max3_f32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 5) { REP8(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 6) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 7) { REP8(asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cmp_gt_f32 vcc, %1, %8\n v_cmp_gt_f32 vcc, %2, %8\n v_cmp_gt_f32 vcc, %3, %8\n v_cmp_gt_f32 vcc, %4, %8\n v_cmp_gt_f32 vcc, %5, %8\n v_cmp_gt_f32 vcc, %6, %8\n v_cmp_gt_f32 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 8) { REP8(asm volatile("v_cmp_ge_u32_e64 s[20:21], %0, %8\n v_cmp_ge_u32_e64 s[20:21], %1, %8\n v_cmp_ge_u32_e64 s[20:21], %2, %8\n v_cmp_ge_u32_e64 s[20:21], %3, %8\n v_cmp_ge_u32_e64 s[20:21], %4, %8\n v_cmp_ge_u32_e64 s[20:21], %5, %8\n v_cmp_ge_u32_e64 s[20:21], %6, %8\n v_cmp_ge_u32_e64 s[20:21], %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 9) { REP8(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %1, %1, %8\n v_cvt_pk_bf16_f32 %2, %2, %8\n v_cvt_pk_bf16_f32 %3, %3, %8\n v_cvt_pk_bf16_f32 %4, %4, %8\n v_cvt_pk_bf16_f32 %5, %5, %8\n v_cvt_pk_bf16_f32 %6, %6, %8\n v_cvt_pk_bf16_f32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 10) { REP8(asm volatile("v_accvgpr_write_b32 a0, %0\n v_accvgpr_write_b32 a1, %1\n v_accvgpr_write_b32 a2, %2\n v_accvgpr_write_b32 a3, %3\n v_accvgpr_write_b32 a4, %4\n v_accvgpr_write_b32 a5, %5\n v_accvgpr_write_b32 a6, %6\n v_accvgpr_write_b32 a7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 11) { REP8(asm volatile("v_accvgpr_read_b32 %0, a0\n v_accvgpr_read_b32 %1, a1\n v_accvgpr_read_b32 %2, a2\n v_accvgpr_read_b32 %3, a3\n v_accvgpr_read_b32 %4, a4\n v_accvgpr_read_b32 %5, a5\n v_accvgpr_read_b32 %6, a6\n v_accvgpr_read_b32 %7, a7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 12) { REP8(asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 13) { REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 14) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 15) { REP8(asm volatile("v_bfe_u32 %0, %0, 16, 16\n v_bfe_u32 %1, %1, 16, 16\n v_bfe_u32 %2, %2, 16, 16\n v_bfe_u32 %3, %3, 16, 16\n v_bfe_u32 %4, %4, 16, 16\n v_bfe_u32 %5, %5, 16, 16\n v_bfe_u32 %6, %6, 16, 16\n v_bfe_u32 %7, %7, 16, 16" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 16) { REP8(asm volatile("v_perm_b32 %0, %0, %8, %8\n v_perm_b32 %1, %1, %8, %8\n v_perm_b32 %2, %2, %8, %8\n v_perm_b32 %3, %3, %8, %8\n v_perm_b32 %4, %4, %8, %8\n v_perm_b32 %5, %5, %8, %8\n v_perm_b32 %6, %6, %8, %8\n v_perm_b32 %7, %7, %8, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 17) { REP8(asm volatile("v_alignbit_b32 %0, %0, %0, 16\n v_alignbit_b32 %1, %1, %1, 16\n v_alignbit_b32 %2, %2, %2, 16\n v_alignbit_b32 %3, %3, %3, 16\n v_alignbit_b32 %4, %4, %4, 16\n v_alignbit_b32 %5, %5, %5, 16\n v_alignbit_b32 %6, %6, %6, 16\n v_alignbit_b32 %7, %7, %7, 16" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 18) { REP8(asm volatile("v_cmp_ge_u32_sdwa vcc, %0, %8 src0_sel:WORD_1 src1_sel:DWORD\n v_cmp_ge_u32_sdwa vcc, %1, %8 src0_sel:WORD_1 src1_sel:DWORD\n v_cmp_ge_u32_sdwa vcc, %2, %8 src0_sel:WORD_1 src1_sel:DWORD\n v_cmp_ge_u32_sdwa vcc, %3, %8 src0_sel:WORD_1 src1_sel:DWORD\n v_cmp_ge_u32_sdwa vcc, %4, %8 src0_sel:WORD_1 src1_sel:DWORD\n v_cmp_ge_u32_sdwa vcc, %5, %8 src0_sel:WORD_1 src1_sel:DWORD\n v_cmp_ge_u32_sdwa vcc, %6, %8 src0_sel:WORD_1 src1_sel:DWORD\n v_cmp_ge_u32_sdwa vcc, %7, %8 src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 19) { REP8(asm volatile("v_cmp_lt_u16 vcc, %0, %8\n v_cmp_lt_u16 vcc, %1, %8\n v_cmp_lt_u16 vcc, %2, %8\n v_cmp_lt_u16 vcc, %3, %8\n v_cmp_lt_u16 vcc, %4, %8\n v_cmp_lt_u16 vcc, %5, %8\n v_cmp_lt_u16 vcc, %6, %8\n v_cmp_lt_u16 vcc, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7");) }
        if (OP == 20) { REP8(asm volatile("v_cmp_gt_u32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_gt_u32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_gt_u32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_gt_u32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_gt_u32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %8, vcc\n v_cmp_gt_u32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_gt_u32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %8, vcc\n v_cmp_gt_u32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21");) }
        if (OP == 21) { REP8(asm volatile("v_cmp_gt_u32_e64 s[20:21], %0, %8\n v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cmp_gt_u32_e64 s[20:21], %1, %8\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cmp_gt_u32_e64 s[20:21], %2, %8\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cmp_gt_u32_e64 s[20:21], %3, %8\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cmp_gt_u32_e64 s[20:21], %4, %8\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cmp_gt_u32_e64 s[20:21], %5, %8\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cmp_gt_u32_e64 s[20:21], %6, %8\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cmp_gt_u32_e64 s[20:21], %7, %8\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21");) }
        if (OP == 22) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21");) }
        if (OP == 23) { REP8(asm volatile("s_mov_b64 vcc, s[20:21]\n v_cndmask_b32 %0, %0, %8, vcc\n s_mov_b64 vcc, s[20:21]\n v_cndmask_b32 %1, %1, %8, vcc\n s_mov_b64 vcc, s[20:21]\n v_cndmask_b32 %2, %2, %8, vcc\n s_mov_b64 vcc, s[20:21]\n v_cndmask_b32 %3, %3, %8, vcc\n s_mov_b64 vcc, s[20:21]\n v_cndmask_b32 %4, %4, %8, vcc\n s_mov_b64 vcc, s[20:21]\n v_cndmask_b32 %5, %5, %8, vcc\n s_mov_b64 vcc, s[20:21]\n v_cndmask_b32 %6, %6, %8, vcc\n s_mov_b64 vcc, s[20:21]\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc", "s20", "s21");) }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int OP> void run(const char* name, uint32_t* out, uint64_t* cyc) {
    hipLaunchKernelGGL(k<OP>, dim3(1), dim3(256), 0, 0, out, cyc, 12345u);
    (void)hipDeviceSynchronize();
    uint64_t c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-22s %.2f ticks per instruction\n", name, (double)c / 4096.0);
}
int main() {
    uint32_t* out; uint64_t* cyc;
    (void)hipMalloc(&out, 4096 * 4); (void)hipMalloc(&cyc, 8);
    run<0>("v_add_u32", out, cyc);
    run<1>("v_xor_b32", out, cyc);
    run<2>("v_lshrrev_b32", out, cyc);
    run<3>("v_fma_f32", out, cyc);
    run<4>("v_max3_f32", out, cyc);
    run<5>("v_cndmask_e32_vcc", out, cyc);
    run<6>("v_cndmask_e64_sgpr", out, cyc);
    run<7>("v_cmp_gt_f32_vcc", out, cyc);
    run<8>("v_cmp_ge_u32_e64", out, cyc);
    run<9>("v_cvt_pk_bf16_f32", out, cyc);
    run<10>("v_accvgpr_write", out, cyc);
    run<11>("v_accvgpr_read", out, cyc);
    run<12>("v_mov_dpp", out, cyc);
    run<13>("v_exp_f32", out, cyc);
    run<14>("v_mul_lo_u32", out, cyc);
    run<15>("v_bfe_u32", out, cyc);
    run<16>("v_perm_b32", out, cyc);
    run<17>("v_alignbit_b32", out, cyc);
    run<18>("v_cmp_ge_u32_sdwa", out, cyc);
    run<19>("v_cmp_lt_u16_vcc", out, cyc);
    run<20>("cmp_vcc+cndmask_vcc (pair)", out, cyc);
    run<21>("cmp_e64+cndmask_e64 (pair)", out, cyc);
    run<22>("cndmask_e64_vcc", out, cyc);
    run<23>("cndmask_e32 after s_mov vcc", out, cyc);
    return 0;
}
