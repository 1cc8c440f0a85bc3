// valu_rates.hip -- VALU issue-rate microbenchmark for gfx950 (measurement tool, not product code).
// Each kernel runs REPS x 64 independent-chain instructions per wave; prints wave-instructions/cycle/SIMD
// at several occupancies.  Build: hipcc --offload-arch=gfx950 -O2 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#define REPS 32768

#define CHAIN8(OP)                                                             \
    asm volatile(OP " %0, %0, %8\n\t" OP " %1, %1, %8\n\t" OP " %2, %2, %8\n\t" \
                 OP " %3, %3, %8\n\t" OP " %4, %4, %8\n\t" OP " %5, %5, %8\n\t" \
                 OP " %6, %6, %8\n\t" OP " %7, %7, %8"                          \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));

#define UN8(OP)                                                                \
    asm volatile(OP " %0, %0\n\t" OP " %1, %1\n\t" OP " %2, %2\n\t" OP " %3, %3\n\t" \
                 OP " %4, %4\n\t" OP " %5, %5\n\t" OP " %6, %6\n\t" OP " %7, %7"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));

template <int K>
__global__ void k_scalar(float *out, float c)
{
    float c2 = c * 0.5f; unsigned long long msk = 0x5555555555555555ull + (unsigned long long)(c > 2.0f);
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < REPS; i++) {
        if (K == 0) { CHAIN8("v_mul_f32") }
        if (K == 1) { CHAIN8("v_add_f32") }
        if (K == 2) { asm volatile("v_fma_f32 %0, %0, %8, %8\n\tv_fma_f32 %1, %1, %8, %8\n\tv_fma_f32 %2, %2, %8, %8\n\tv_fma_f32 %3, %3, %8, %8\n\t"
                                   "v_fma_f32 %4, %4, %8, %8\n\tv_fma_f32 %5, %5, %8, %8\n\tv_fma_f32 %6, %6, %8, %8\n\tv_fma_f32 %7, %7, %8, %8"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
        if (K == 3) { UN8("v_sqrt_f32") }
        if (K == 4) { UN8("v_rcp_f32") }
        if (K == 5) { UN8("v_rsq_f32") }
        if (K == 6) { UN8("v_log_f32") }
        if (K == 7) { CHAIN8("v_max_f32") }
        if (K == 8) { asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n\tv_cndmask_b32 %1, %1, %8, vcc\n\tv_cndmask_b32 %2, %2, %8, vcc\n\tv_cndmask_b32 %3, %3, %8, vcc\n\t"
                                   "v_cndmask_b32 %4, %4, %8, vcc\n\tv_cndmask_b32 %5, %5, %8, vcc\n\tv_cndmask_b32 %6, %6, %8, vcc\n\tv_cndmask_b32 %7, %7, %8, vcc"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc"); }
        if (K == 10) { asm volatile("v_cndmask_b32_e64 %0, %0, %8, %9\n\tv_cndmask_b32_e64 %1, %1, %8, %9\n\tv_cndmask_b32_e64 %2, %2, %8, %9\n\tv_cndmask_b32_e64 %3, %3, %8, %9\n\t"
                                   "v_cndmask_b32_e64 %4, %4, %8, %9\n\tv_cndmask_b32_e64 %5, %5, %8, %9\n\tv_cndmask_b32_e64 %6, %6, %8, %9\n\tv_cndmask_b32_e64 %7, %7, %8, %9"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(msk)); }
        if (K == 11) { asm volatile("v_cmp_lt_f32 vcc, %0, %8\n\tv_cmp_lt_f32 vcc, %1, %8\n\tv_cmp_lt_f32 vcc, %2, %8\n\tv_cmp_lt_f32 vcc, %3, %8\n\t"
                                   "v_cmp_lt_f32 vcc, %4, %8\n\tv_cmp_lt_f32 vcc, %5, %8\n\tv_cmp_lt_f32 vcc, %6, %8\n\tv_cmp_lt_f32 vcc, %7, %8"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc"); }
        if (K == 12) { asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                                   "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2)); }
        if (K == 13) { asm volatile("v_fmac_f32 %0, %8, %9\n\tv_fmac_f32 %1, %8, %9\n\tv_fmac_f32 %2, %8, %9\n\tv_fmac_f32 %3, %8, %9\n\t"
                                   "v_fmac_f32 %4, %8, %9\n\tv_fmac_f32 %5, %8, %9\n\tv_fmac_f32 %6, %8, %9\n\tv_fmac_f32 %7, %8, %9"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "v"(c2)); }
        if (K == 14) { CHAIN8("v_add_u32") }
        if (K == 15) { asm volatile("v_mul_f32 %0, %1, %2\n\tv_mul_f32 %1, %2, %3\n\tv_mul_f32 %2, %3, %4\n\tv_mul_f32 %3, %4, %5\n\t"
                                   "v_mul_f32 %4, %5, %6\n\tv_mul_f32 %5, %6, %7\n\tv_mul_f32 %6, %7, %0\n\tv_mul_f32 %7, %0, %1"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        if (K == 16) { asm volatile("v_cmp_lt_f32 vcc, %0, %4\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_cmp_lt_f32 vcc, %1, %4\n\tv_cndmask_b32 %1, %1, %4, vcc\n\t"
                                   "v_cmp_lt_f32 vcc, %2, %4\n\tv_cndmask_b32 %2, %2, %4, vcc\n\tv_cmp_lt_f32 vcc, %3, %4\n\tv_cndmask_b32 %3, %3, %4, vcc"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc"); }
        if (K == 17) { asm volatile("v_cmp_lt_f32 s[20:21], %0, %4\n\tv_cndmask_b32_e64 %0, %0, %4, s[20:21]\n\tv_cmp_lt_f32 s[22:23], %1, %4\n\tv_cndmask_b32_e64 %1, %1, %4, s[22:23]\n\t"
                                   "v_cmp_lt_f32 s[24:25], %2, %4\n\tv_cndmask_b32_e64 %2, %2, %4, s[24:25]\n\tv_cmp_lt_f32 s[26:27], %3, %4\n\tv_cndmask_b32_e64 %3, %3, %4, s[26:27]"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "s20","s21","s22","s23","s24","s25","s26","s27"); }
        if (K == 18) { asm volatile("v_cmp_lt_f32 vcc, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_cndmask_b32 %0, %0, %4, vcc\n\tv_mul_f32 %3, %3, %4\n\tv_mul_f32 %1, %1, %4\n\t"
                                   "v_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "vcc"); }
        if (K == 19) { asm volatile("v_min_f32 %0, %0, %4\n\tv_min_f32 %1, %1, %4\n\tv_min_f32 %2, %2, %4\n\tv_min_f32 %3, %3, %4\n\t"
                                   "v_min_f32 %0, %0, %4\n\tv_min_f32 %1, %1, %4\n\tv_min_f32 %2, %2, %4\n\tv_min_f32 %3, %3, %4"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c)); }
        if (K == 20) { asm volatile("v_sub_f32 %0, %0, %4\n\tv_sub_f32 %1, %1, %4\n\tv_sub_f32 %2, %2, %4\n\tv_sub_f32 %3, %3, %4\n\t"
                                   "v_mul_f32 %0, %0, %4\n\tv_mul_f32 %1, %1, %4\n\tv_mul_f32 %2, %2, %4\n\tv_mul_f32 %3, %3, %4"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c)); }
        if (K == 21) { // float / integer interleaved: do they share one issue port?
            asm volatile("v_mul_f32 %0, %0, %8\n\tv_add_u32 %4, %4, %8\n\tv_mul_f32 %1, %1, %8\n\tv_add_u32 %5, %5, %8\n\t"
                         "v_mul_f32 %2, %2, %8\n\tv_add_u32 %6, %6, %8\n\tv_mul_f32 %3, %3, %8\n\tv_add_u32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
        if (K == 22) { // float / transcendental interleaved 7:1
            asm volatile("v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_rsq_f32 %7, %7\n\t"
                         "v_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
        if (K == 23) { // float / compare+select interleaved
            asm volatile("v_mul_f32 %0, %0, %8\n\tv_cmp_lt_f32 vcc, %4, %8\n\tv_mul_f32 %1, %1, %8\n\tv_cndmask_b32 %5, %5, %8, vcc\n\t"
                         "v_mul_f32 %2, %2, %8\n\tv_cmp_lt_f32 vcc, %6, %8\n\tv_mul_f32 %3, %3, %8\n\tv_cndmask_b32 %7, %7, %8, vcc"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c) : "vcc"); }
        if (K == 24) { // VOP2 fma with a literal multiplier: D = S0 * K + S1
            asm volatile("v_fmamk_f32 %0, %0, 0xc1000000, %8\n\tv_fmamk_f32 %1, %1, 0xc1000000, %8\n\tv_fmamk_f32 %2, %2, 0xc1000000, %8\n\tv_fmamk_f32 %3, %3, 0xc1000000, %8\n\t"
                         "v_fmamk_f32 %4, %4, 0xc1000000, %8\n\tv_fmamk_f32 %5, %5, 0xc1000000, %8\n\tv_fmamk_f32 %6, %6, 0xc1000000, %8\n\tv_fmamk_f32 %7, %7, 0xc1000000, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
        if (K == 25) { // VOP3 fma with an inline constant and two VGPRs
            asm volatile("v_fma_f32 %0, %0, 2.0, %8\n\tv_fma_f32 %1, %1, 2.0, %8\n\tv_fma_f32 %2, %2, 2.0, %8\n\tv_fma_f32 %3, %3, 2.0, %8\n\t"
                         "v_fma_f32 %4, %4, 2.0, %8\n\tv_fma_f32 %5, %5, 2.0, %8\n\tv_fma_f32 %6, %6, 2.0, %8\n\tv_fma_f32 %7, %7, 2.0, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
        if (K == 26) { // the iteration's mix: 6 mul/add : 1 three-source fma with distinct registers
            asm volatile("v_mul_f32 %0, %0, %8\n\tv_add_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_fma_f32 %3, %4, %5, %3\n\t"
                         "v_mul_f32 %4, %4, %8\n\tv_add_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
        if (K == 27) { // 15 mul : 1 rsq
            asm volatile("v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_rsq_f32 %7, %7\n\t"
                         "v_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\t"
                         "v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\t"
                         "v_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %0, %0, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
        if (K == 28) { // 31 mul : 1 rsq
            asm volatile("v_rsq_f32 %7, %7\n\t"
                         "v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\t"
                         "v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %0, %0, %8\n\t"
                         "v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %0, %0, %8\n\t"
                         "v_mul_f32 %0, %0, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %0, %0, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
        if (K == 9) { // dependent chain of v_mul (latency)
            asm volatile("v_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\t"
                         "v_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1\n\tv_mul_f32 %0, %0, %1" : "+v"(a0) : "v"(c)); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// Round 4: what does an SDWA instruction cost?  The compiler turns "(uint16_t)bits == 0xffff" into v_cmp_eq_u32_sdwa ... src0_sel:WORD_0
// (the guard of the power-8 iteration) and half->float unpacking into v_cvt_f32_f16_sdwa ... src0_sel:WORD_1.  K = 0 the SDWA compare into
// an SGPR pair, 1 the same compare as plain VOP3, 2 v_and_b32 + v_cmp (what -amdgpu-sdwa-peephole=0 emits), 3 v_cmp_eq_u16, 4/5/6 =
// 0/1/3 mixed 1:7 with v_mul_f32, 7 v_cvt_f32_f16_sdwa WORD_1, 8 v_lshrrev_b32 16 + v_cvt_f32_f16, 9/10 = 7/8 mixed 1:7 (2:6) with v_mul.
template <int K>
__global__ void k_sdwa(float *out, float c)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned k16 = 0xffffu + (unsigned)(c > 2.0f);
#define CLOB "s20","s21","s22","s23","s24","s25","s26","s27"
#define MUL7 "v_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8"
#define ARGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(k16) : CLOB, "vcc"
    for (int i = 0; i < REPS; i++) {
        if (K == 0) asm volatile("v_cmp_eq_u32_sdwa s[20:21], %0, %9 src0_sel:WORD_0 src1_sel:DWORD\n\tv_cmp_eq_u32_sdwa s[22:23], %1, %9 src0_sel:WORD_0 src1_sel:DWORD\n\t"
                                 "v_cmp_eq_u32_sdwa s[24:25], %2, %9 src0_sel:WORD_0 src1_sel:DWORD\n\tv_cmp_eq_u32_sdwa s[26:27], %3, %9 src0_sel:WORD_0 src1_sel:DWORD\n\t"
                                 "v_cmp_eq_u32_sdwa s[20:21], %4, %9 src0_sel:WORD_0 src1_sel:DWORD\n\tv_cmp_eq_u32_sdwa s[22:23], %5, %9 src0_sel:WORD_0 src1_sel:DWORD\n\t"
                                 "v_cmp_eq_u32_sdwa s[24:25], %6, %9 src0_sel:WORD_0 src1_sel:DWORD\n\tv_cmp_eq_u32_sdwa s[26:27], %7, %9 src0_sel:WORD_0 src1_sel:DWORD" ARGS);
        if (K == 1) asm volatile("v_cmp_eq_u32_e64 s[20:21], %0, %9\n\tv_cmp_eq_u32_e64 s[22:23], %1, %9\n\tv_cmp_eq_u32_e64 s[24:25], %2, %9\n\tv_cmp_eq_u32_e64 s[26:27], %3, %9\n\t"
                                 "v_cmp_eq_u32_e64 s[20:21], %4, %9\n\tv_cmp_eq_u32_e64 s[22:23], %5, %9\n\tv_cmp_eq_u32_e64 s[24:25], %6, %9\n\tv_cmp_eq_u32_e64 s[26:27], %7, %9" ARGS);
        if (K == 2) asm volatile("v_and_b32 %4, %9, %0\n\tv_cmp_eq_u32_e64 s[20:21], %4, %9\n\tv_and_b32 %5, %9, %1\n\tv_cmp_eq_u32_e64 s[22:23], %5, %9\n\t"
                                 "v_and_b32 %6, %9, %2\n\tv_cmp_eq_u32_e64 s[24:25], %6, %9\n\tv_and_b32 %7, %9, %3\n\tv_cmp_eq_u32_e64 s[26:27], %7, %9" ARGS);
        if (K == 3) asm volatile("v_cmp_eq_u16_e64 s[20:21], %0, %9\n\tv_cmp_eq_u16_e64 s[22:23], %1, %9\n\tv_cmp_eq_u16_e64 s[24:25], %2, %9\n\tv_cmp_eq_u16_e64 s[26:27], %3, %9\n\t"
                                 "v_cmp_eq_u16_e64 s[20:21], %4, %9\n\tv_cmp_eq_u16_e64 s[22:23], %5, %9\n\tv_cmp_eq_u16_e64 s[24:25], %6, %9\n\tv_cmp_eq_u16_e64 s[26:27], %7, %9" ARGS);
        if (K == 4) asm volatile("v_cmp_eq_u32_sdwa s[20:21], %0, %9 src0_sel:WORD_0 src1_sel:DWORD\n\t" MUL7 ARGS);
        if (K == 5) asm volatile("v_cmp_eq_u32_e64 s[20:21], %0, %9\n\t" MUL7 ARGS);
        if (K == 6) asm volatile("v_cmp_eq_u16_e64 s[20:21], %0, %9\n\t" MUL7 ARGS);
        if (K == 7) asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\tv_cvt_f32_f16_sdwa %1, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                                 "v_cvt_f32_f16_sdwa %2, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\tv_cvt_f32_f16_sdwa %3, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                                 "v_cvt_f32_f16_sdwa %4, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\tv_cvt_f32_f16_sdwa %5, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
                                 "v_cvt_f32_f16_sdwa %6, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\tv_cvt_f32_f16_sdwa %7, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" ARGS);
        if (K == 8) asm volatile("v_lshrrev_b32 %0, 16, %0\n\tv_cvt_f32_f16 %0, %0\n\tv_lshrrev_b32 %1, 16, %1\n\tv_cvt_f32_f16 %1, %1\n\t"
                                 "v_lshrrev_b32 %2, 16, %2\n\tv_cvt_f32_f16 %2, %2\n\tv_lshrrev_b32 %3, 16, %3\n\tv_cvt_f32_f16 %3, %3" ARGS);
        if (K == 9) asm volatile("v_cvt_f32_f16_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t" MUL7 ARGS);
        if (K == 10) asm volatile("v_lshrrev_b32 %0, 16, %0\n\tv_cvt_f32_f16 %0, %0\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8" ARGS);
    }
#undef ARGS
#undef MUL7
#undef CLOB
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

// Round 4: instruction FORMS of the power-8 iteration, each as 1 instruction among 7 v_mul_f32 (an expensive form only shows in a mix:
// the SDWA compare runs at the plain rate back to back).  8 instructions per group; a full-rate form gives the v_mul figure.
template <int K>
__global__ void k_mix(float *out, float c)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float ks = c * 3.0f; const unsigned long long msk = 0x5555555555555555ull + (unsigned long long)(c > 2.0f);
#define MUL7 "v_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8"
#define ARGS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c), "s"(ks), "s"(msk) : "s20", "s21", "vcc", "scc"
#define MIX(X) asm volatile(X "\n\t" MUL7 ARGS)
    for (int i = 0; i < REPS; i++) {
        if (K == 0) MIX("v_mul_f32 %0, %0, %8");
        if (K == 1) MIX("v_min3_f32 %0, %0, |%1|, |%2|");
        if (K == 2) MIX("v_fma_f32 %0, -%0, %1, 1.0");
        if (K == 3) MIX("v_mul_f32 %0, 0x41e00000, %0");
        if (K == 4) MIX("v_fmac_f32 %0, 0xc1800000, %1");
        if (K == 5) MIX("v_cmp_gt_f32_e64 s[20:21], %9, %0");
        if (K == 6) MIX("v_cmp_lt_f32_e32 vcc, %9, %0");
        if (K == 7) MIX("v_mov_b32 %0, %1");
        if (K == 8) MIX("v_cndmask_b32_e64 %0, %0, %1, %10");
        if (K == 9) MIX("v_fma_f32 %0, %0, %9, 1.0");
        if (K == 10) MIX("v_mul_f32 %0, %9, %0");
        if (K == 11) MIX("v_mul_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf");
        if (K == 12) MIX("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf");
        if (K == 13) MIX("v_readlane_b32 s20, %0, 3");
        if (K == 14) MIX("v_readfirstlane_b32 s20, %0");
        if (K == 15) MIX("s_or_b64 s[20:21], %10, %10");
        if (K == 16) MIX("s_nop 0");
        if (K == 17) MIX("v_cvt_f32_u32 %0, %0");
        if (K == 18) MIX("v_ldexp_f32 %0, %0, %1");
        if (K == 19) MIX("v_fmamk_f32 %0, %0, 0xc1000000, %1");
        if (K == 20) MIX("v_and_b32 %0, 0xffff, %0");
        if (K == 21) MIX("v_bfe_u32 %0, %0, 8, 8");
        if (K == 22) MIX("v_mul_f32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD");
        if (K == 23) MIX("v_mul_legacy_f32 %0, %0, %1");
        if (K == 24) MIX("v_med3_f32 %0, %0, %1, %2");
        if (K == 25) MIX("v_add_co_u32 %0, vcc, %0, %1");
        if (K == 26) MIX("v_mul_lo_u32 %0, %0, %1");
        if (K == 27) MIX("v_mad_u32_u24 %0, %0, %1, %2");
        if (K == 28) MIX("v_lshlrev_b32 %0, 3, %0");
        if (K == 29) MIX("v_exp_f32 %0, %0");
        if (K == 30) MIX("v_cvt_f16_f32 %0, %0");
        if (K == 31) MIX("v_cvt_pkrtz_f16_f32 %0, %0, %1");
        if (K == 32) MIX("v_perm_b32 %0, %0, %1, %2");
        if (K == 33) MIX("v_floor_f32 %0, %0");
        if (K == 34) MIX("v_fract_f32 %0, %0");
        if (K == 35) MIX("v_cvt_i32_f32 %0, %0");
        if (K == 40) asm volatile("v_cmp_gt_f32_e64 s[20:21], %9, %0\n\tv_cmp_lt_f32_e32 vcc, %9, %1\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8" ARGS);
        if (K == 41) asm volatile("v_cmp_gt_f32_e64 s[20:21], %9, %0\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_cmp_lt_f32_e32 vcc, %9, %1\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8" ARGS);
        if (K == 42) asm volatile("v_min3_f32 %0, %0, |%2|, |%3|\n\tv_min3_f32 %0, %0, %4, %5\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8" ARGS);
        if (K == 43) asm volatile("v_min3_f32 %0, %0, |%2|, |%3|\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_min3_f32 %1, %1, %4, %5\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8" ARGS);
        if (K == 44) asm volatile("v_cmp_lt_f32_e32 vcc, %9, %1\n\tv_min3_f32 %0, %0, |%2|, |%3|\n\tv_min3_f32 %0, %0, %4, %5\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %7, %7, %8" ARGS);
        if (K == 45) asm volatile("v_cmp_lt_f32_e32 vcc, %9, %1\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_min3_f32 %0, %0, |%2|, |%3|\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_min3_f32 %1, %1, %4, %5\n\tv_mul_f32 %7, %7, %8" ARGS);
        if (K == 46) asm volatile("v_fma_f32 %0, -%2, %2, %3\n\tv_fmac_f32 %1, %0, %4\n\tv_cmp_gt_f32_e64 s[20:21], %9, %5\n\tv_cmp_lt_f32_e32 vcc, %9, %6\n\tv_fma_f32 %0, -%1, %7, 1.0\n\tv_fmac_f32 %7, %0, %7\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %5, %5, %8" ARGS);
        if (K == 47) asm volatile("v_fma_f32 %0, -%2, %2, %3\n\tv_mul_f32 %6, %6, %8\n\tv_cmp_gt_f32_e64 s[20:21], %9, %5\n\tv_fmac_f32 %1, %0, %4\n\tv_cmp_lt_f32_e32 vcc, %9, %6\n\tv_mul_f32 %5, %5, %8\n\tv_fma_f32 %0, -%1, %7, 1.0\n\tv_fmac_f32 %7, %0, %7" ARGS);
        if (K == 50) asm volatile("v_rsq_f32 %0, %0\n\tv_rsq_f32 %7, %7\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8" ARGS);
        if (K == 51) asm volatile("v_rsq_f32 %0, %0\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_rsq_f32 %7, %7\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8" ARGS);
        if (K == 52) asm volatile("v_rsq_f32 %0, %0\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_rsq_f32 %7, %7\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8" ARGS);
        if (K == 53) asm volatile("v_rsq_f32 %0, %0\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_rsq_f32 %7, %7\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8\n\tv_mul_f32 %1, %1, %8\n\tv_mul_f32 %2, %2, %8\n\tv_mul_f32 %3, %3, %8\n\tv_mul_f32 %4, %4, %8\n\tv_mul_f32 %5, %5, %8\n\tv_mul_f32 %6, %6, %8" ARGS);
    }
#undef MIX
#undef ARGS
#undef MUL7
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

typedef float float2_t __attribute__((ext_vector_type(2)));

template <int K>
__global__ void k_packed(float *out, float cc)
{
    float2_t a0 = { (float)threadIdx.x, 1 }, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float2_t c = { cc, cc };
    for (int i = 0; i < REPS; i++) {
        if (K == 0) { CHAIN8("v_pk_mul_f32") }
        if (K == 1) { CHAIN8("v_pk_add_f32") }
        if (K == 2) { asm volatile("v_pk_fma_f32 %0, %0, %8, %8\n\tv_pk_fma_f32 %1, %1, %8, %8\n\tv_pk_fma_f32 %2, %2, %8, %8\n\tv_pk_fma_f32 %3, %3, %8, %8\n\t"
                                   "v_pk_fma_f32 %4, %4, %8, %8\n\tv_pk_fma_f32 %5, %5, %8, %8\n\tv_pk_fma_f32 %6, %6, %8, %8\n\tv_pk_fma_f32 %7, %7, %8, %8"
                                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c)); }
    }
    float2_t s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}

// shader clock under a full VALU load: s_memtime ticks (shader cycles) per s_memrealtime tick (100 MHz), lane 0 of every workgroup
__global__ void k_clock(float *out, float c, unsigned long long *stamps)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < REPS; i++) { CHAIN8("v_mul_f32") }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { stamps[blockIdx.x * 2] = t1 - t0; stamps[blockIdx.x * 2 + 1] = r1 - r0; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <typename F>
static void run(const char *name, F launch, int ops_per_instr, double instr_scale = 1.0)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    float *out;
    hipMalloc(&out, (size_t)cus * 32 * 64 * 4 * 2);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-26s", name);
    for (int wps : { 1, 2, 4, 8 }) {               // waves per SIMD
        const int blocks = cus * wps, threads = 256; // 4 waves per block -> one per SIMD
        launch(blocks, threads, out);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        launch(blocks, threads, out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        double winstr = (double)blocks * 4 * REPS * 8 * instr_scale;
        double per_simd_per_s = winstr / (cus * 4.0) / (ms * 1e-3);
        printf("  wps%d: %6.3f ms %6.2f Ginstr/s/SIMD (%5.1f Tlaneops/s)", wps, ms, per_simd_per_s / 1e9,
               winstr * 64 * ops_per_instr / (ms * 1e-3) / 1e12);
    }
    printf("\n");
    hipFree(out);
}


// Round 4: does a wave64 instruction skip a 32-lane pass whose EXEC half is zero?  (wave64 issues on the SIMD-32 in two passes.)  The
// loop runs under a per-lane condition taken from a run-time mask, so the compiler narrows EXEC with s_and_saveexec; the instruction
// count per wave is the same for every mask.
__global__ void k_execmask(float *out, float c, unsigned long long mask)
{
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    if ((mask >> (threadIdx.x & 63u)) & 1ull) {
        for (int i = 0; i < REPS; i++) { CHAIN8("v_mul_f32") }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
__global__ void k_execmask_trans(float *out, float c, unsigned long long mask)
{
    float a0 = threadIdx.x + 1.0f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    if ((mask >> (threadIdx.x & 63u)) & 1ull) {
        for (int i = 0; i < REPS; i++) { UN8("v_rsq_f32") }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + c;
}
static void exec_halves()
{
    const struct { const char *name; unsigned long long m; } masks[] = {
        { "all 64 lanes", ~0ull }, { "lanes 0-31", 0xffffffffull }, { "lanes 32-63", 0xffffffff00000000ull },
        { "lanes 0-15", 0xffffull }, { "lane 0", 1ull }, { "even lanes", 0x5555555555555555ull },
        { "lanes 0-15 + 32-47", 0x0000ffff0000ffffull }, { "lanes 16-31", 0xffff0000ull }, { "lane 0 + lane 63", 0x8000000000000001ull } };
    for (int trans = 0; trans < 2; trans++)
        for (const auto &mk : masks) {
            const unsigned long long m = mk.m;
            char name[64];
            snprintf(name, sizeof name, "%s %s", trans ? "rsq" : "mul", mk.name);
            if (trans) run(name, [m](int b, int t, float *o) { hipLaunchKernelGGL(k_execmask_trans, dim3(b), dim3(t), 0, 0, o, 1.0001f, m); }, 1);
            else       run(name, [m](int b, int t, float *o) { hipLaunchKernelGGL(k_execmask, dim3(b), dim3(t), 0, 0, o, 1.0001f, m); }, 1);
        }
}

// ~100 ms of v_mul on every SIMD: the shader clock of a load that starts from idle ramps for ~20 ms (tools/clock_probe_check.py), and
// the first rows of a listing would be taken below the others
static void warm()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    float *out;
    (void)hipMalloc(&out, (size_t)prop.multiProcessorCount * 4 * 256 * 4);
    for (int rep = 0; rep < 120; rep++) hipLaunchKernelGGL(k_scalar<0>, dim3(prop.multiProcessorCount * 4), dim3(256), 0, 0, out, 1.0001f);
    (void)hipDeviceSynchronize();
    (void)hipFree(out);
}
static void sdwa_cost()
{
    warm();
#define W(K, NAME) run(NAME, [](int b, int t, float *o) { hipLaunchKernelGGL(k_sdwa<K>, dim3(b), dim3(t), 0, 0, o, 1.0001f); }, 1)
    W(0, "v_cmp_eq_u32_sdwa WORD_0"); W(1, "v_cmp_eq_u32_e64"); W(2, "v_and + v_cmp_eq_u32"); W(3, "v_cmp_eq_u16_e64");
    W(4, "7 mul : 1 cmp_sdwa"); W(5, "7 mul : 1 cmp_e64"); W(6, "7 mul : 1 cmp_u16");
    W(7, "v_cvt_f32_f16_sdwa WORD_1"); W(8, "v_lshrrev + v_cvt_f32_f16"); W(9, "7 mul : 1 cvt_sdwa"); W(10, "6 mul : lshr + cvt");
#undef W
}

static void form_costs()
{
    warm();
#define W(K, NAME) run("7 mul : 1 " NAME, [](int b, int t, float *o) { hipLaunchKernelGGL(k_mix<K>, dim3(b), dim3(t), 0, 0, o, 1.0001f); }, 1)
    W(0, "v_mul (reference)"); W(1, "v_min3 |a||b|"); W(2, "v_fma -a,b,1.0"); W(3, "v_mul literal"); W(4, "v_fmac literal"); W(5, "v_cmp e64 ->sgpr");
    W(6, "v_cmp e32 ->vcc"); W(7, "v_mov"); W(8, "v_cndmask sgpr"); W(9, "v_fma a,sgpr,1.0"); W(10, "v_mul sgpr"); W(11, "v_mul dpp quad");
    W(12, "v_mov dpp row_shr"); W(13, "v_readlane"); W(14, "v_readfirstlane"); W(15, "s_or_b64"); W(16, "s_nop 0"); W(17, "v_cvt_f32_u32");
    W(18, "v_ldexp_f32"); W(19, "v_fmamk literal"); W(20, "v_and literal"); W(21, "v_bfe_u32"); W(22, "v_mul_f32_sdwa DWORD"); W(23, "v_mul_legacy");
    W(24, "v_med3_f32"); W(25, "v_add_co_u32"); W(26, "v_mul_lo_u32"); W(27, "v_mad_u32_u24"); W(28, "v_lshlrev_b32"); W(29, "v_exp_f32");
    W(40, "[2 cmp adjacent]"); W(41, "[2 cmp apart]"); W(42, "[2 min3 adjacent dep]"); W(43, "[2 min3 apart indep]"); W(44, "[cmp,min3,min3 adjacent]"); W(45, "[cmp,min3,min3 apart]"); W(46, "[fma,fmac,cmp,cmp,fma,fmac,2mul]"); W(47, "[same, interleaved]");
#define W4(K, NAME) run("30 mul : 2 rsq " NAME, [](int b, int t, float *o) { hipLaunchKernelGGL(k_mix<K>, dim3(b), dim3(t), 0, 0, o, 1.0001f); }, 1, 4.0)
    W4(50, "[adjacent]"); W4(51, "[16 apart]"); W4(52, "[2 between]"); W4(53, "[4 between]");
#undef W4
    W(30, "v_cvt_f16_f32"); W(31, "v_cvt_pkrtz"); W(32, "v_perm_b32"); W(33, "v_floor_f32"); W(34, "v_fract_f32"); W(35, "v_cvt_i32_f32");
#undef W
}

static void clock_under_load()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    for (int wps : { 1, 2, 4, 8 }) {
        const int blocks = cus * wps;
        float *out; unsigned long long *st;
        (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&st, (size_t)blocks * 16);
        for (int rep = 0; rep < 3; rep++) hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, st);
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h(blocks * 2);
        (void)hipMemcpy(h.data(), st, (size_t)blocks * 16, hipMemcpyDeviceToHost);
        double cyc = 0, real = 0;
        for (int b = 0; b < blocks; b++) { cyc += (double)h[b * 2]; real += (double)h[b * 2 + 1]; }
        const double mhz = cyc / real * 100.0;
        // instructions per wave = REPS * 8; four waves per block on four SIMDs, wps blocks per CU
        printf("clock under v_mul load, %d waves/SIMD: s_memtime/s_memrealtime -> %.0f MHz; %.2f shader cycles per wave-instruction per SIMD\n",
               wps, mhz, cyc / blocks / ((double)REPS * 8 * wps));
        (void)hipFree(out); (void)hipFree(st);
    }
}

// the same v_mul stream after ~50 ms of continuous load: the clock of a load that starts from idle is ~2.0-2.1 GHz and ramps to 2.4 GHz
// over ~20 ms (tools/clock_probe_check.py), so the one-launch figures above are taken at the LOW clock
static void sustained()
{
    hipDeviceProp_t prop;
    (void)hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    for (int wps : { 2, 4, 8 }) {
        const int blocks = cus * wps;
        float *out; unsigned long long *st;
        (void)hipMalloc(&out, (size_t)blocks * 256 * 4); (void)hipMalloc(&st, (size_t)blocks * 16);
        const int warm = 120 / wps;
        for (int rep = 0; rep < warm; rep++) hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, st);
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipEventRecord(e0);
        const int n = 10;
        for (int rep = 0; rep < n; rep++) hipLaunchKernelGGL(k_clock, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, st);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(blocks * 2);
        (void)hipMemcpy(h.data(), st, (size_t)blocks * 16, hipMemcpyDeviceToHost);
        double cyc = 0, real = 0;
        for (int b = 0; b < blocks; b++) { cyc += (double)h[b * 2]; real += (double)h[b * 2 + 1]; }
        const double winstr = (double)blocks * 4 * REPS * 8 * n;
        printf("sustained v_mul, %d waves/SIMD: %.2f G instructions/s/SIMD at %.0f MHz (s_memtime / s_memrealtime) = one per %.2f cycles\n",
               wps, winstr / (cus * 4.0) / (ms * 1e-3) / 1e9, cyc / real * 100.0, (cyc / real * 1e8) / (winstr / (cus * 4.0) / (ms * 1e-3)));
        (void)hipFree(out); (void)hipFree(st);
    }
}

int main(int argc, char **argv)
{
    if (argc > 1 && !strcmp(argv[1], "exec")) { exec_halves(); return 0; }
    if (argc > 1 && !strcmp(argv[1], "sdwa")) { sdwa_cost(); return 0; }
    if (argc > 1 && !strcmp(argv[1], "forms")) { form_costs(); return 0; }
    clock_under_load();
    sustained();
#define S(K, NAME) run(NAME, [](int b, int t, float *o) { hipLaunchKernelGGL(k_scalar<K>, dim3(b), dim3(t), 0, 0, o, 1.0001f); }, 1)
#define P(K, NAME) run(NAME, [](int b, int t, float *o) { hipLaunchKernelGGL(k_packed<K>, dim3(b), dim3(t), 0, 0, o, 1.0001f); }, 2)
    S(0, "v_mul_f32"); S(1, "v_add_f32"); S(2, "v_fma_f32"); P(0, "v_pk_mul_f32"); P(1, "v_pk_add_f32"); P(2, "v_pk_fma_f32");
    S(3, "v_sqrt_f32"); S(4, "v_rcp_f32"); S(5, "v_rsq_f32"); S(6, "v_log_f32"); S(7, "v_max_f32"); S(8, "v_cndmask vcc"); S(10, "v_cndmask e64"); S(11, "v_cmp_lt_f32"); S(12, "v_fma 3src"); S(13, "v_fmac_f32"); S(14, "v_add_u32"); S(15, "v_mul 2vsrc"); S(9, "dep v_mul"); S(16, "cmp+cnd vcc"); S(17, "cmp+cnd sgpr"); S(18, "cmp,2mul,cnd,4mul"); S(19, "v_min_f32"); S(20, "sub/mul mix"); S(21, "mul/add_u32 1:1"); S(22, "7 mul : 1 rsq"); S(23, "mul/cmp/mul/cnd"); S(24, "v_fmamk literal"); S(25, "v_fma inline k"); S(26, "7 mul/add : 1 fma3");
    run("15 mul : 1 rsq", [](int b, int t, float *o) { hipLaunchKernelGGL(k_scalar<27>, dim3(b), dim3(t), 0, 0, o, 1.0001f); }, 1, 2.0);
    run("31 mul : 1 rsq", [](int b, int t, float *o) { hipLaunchKernelGGL(k_scalar<28>, dim3(b), dim3(t), 0, 0, o, 1.0001f); }, 1, 4.0);
    return 0;
}
