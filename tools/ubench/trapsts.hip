// trapsts.hip -- do the sticky IEEE exception bits of TRAPSTS record an underflow of a plain v_mul_f32 / v_fma_f32 without traps enabled?
// (Would make a guard for "no intermediate product underflowed" free of vector instructions.)  Measurement tool.
#include <hip/hip_runtime.h>
#include <stdio.h>
// s_getreg_b32 simm16 = (size-1) << 11 | offset << 6 | id ; HW_REG_TRAPSTS = 3, HW_REG_MODE = 1
#define GETREG(id, off, size) __builtin_amdgcn_s_getreg((((size) - 1) << 11) | ((off) << 6) | (id))
__global__ void k(const float *in, unsigned *out)
{
    const float a = in[0], b = in[1], c = in[2], d = in[3], e = in[4], f = in[5];
    out[0] = GETREG(3, 0, 32);                  // TRAPSTS at start
    out[1] = GETREG(1, 0, 32);                  // MODE
    __builtin_amdgcn_s_setreg((9 - 1) << 11 | 0 << 6 | 3, 0);   // clear EXCP[8:0]
    out[2] = GETREG(3, 0, 9);
    float r1 = a * b;                           // normal * normal -> normal, inexact
    asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    out[3] = GETREG(3, 0, 9);
    float r2 = c * d;                           // tiny and inexact (subnormal result, bits lost)
    asm volatile("s_nop 7\n\ts_nop 7" :: "v"(r2) : "memory");
    out[4] = GETREG(3, 0, 9);
    __builtin_amdgcn_s_setreg((9 - 1) << 11 | 0 << 6 | 3, 0);
    float r3 = e * f;                           // subnormal result, exact
    asm volatile("s_nop 7\n\ts_nop 7" :: "v"(r3) : "memory");
    out[5] = GETREG(3, 0, 9);
    out[6] = __float_as_uint(r1); out[7] = __float_as_uint(r2); out[8] = __float_as_uint(r3);
}
int main()
{
    float h[6] = { 1.2345678f, 3.1415927f, 1.2345678e-25f, 3.1415927e-16f, 0x1p-100f, 0x1p-40f };
    float *d; unsigned *o; unsigned ho[9];
    (void)hipMalloc(&d, sizeof h); (void)hipMalloc(&o, sizeof ho);
    (void)hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o);
    (void)hipMemcpy(ho, o, sizeof ho, hipMemcpyDeviceToHost);
    printf("TRAPSTS at start 0x%08x  MODE 0x%08x (bits 0-7 round/denorm, 9 IEEE, 12-20 EXCP_EN)\n", ho[0], ho[1]);
    printf("EXCP after clear 0x%03x; after normal inexact mul 0x%03x; after tiny inexact mul 0x%03x; after exact subnormal mul 0x%03x\n", ho[2], ho[3], ho[4], ho[5]);
    printf("(bit 0 invalid, 1 input denormal, 2 div0, 3 overflow, 4 underflow, 5 inexact)  results %08x %08x %08x\n", ho[6], ho[7], ho[8]);
    return 0;
}
