// omod_probe.hip -- does gfx950 honour VOP3 output modifiers (mul:2, div:2) in a compute kernel (MODE.IEEE = 1, f32 denormals on)?
// And after clearing MODE.IEEE with s_setreg?   hipcc --offload-arch=gfx950 -O2 omod_probe.hip -o omod_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
__global__ void k(float *out, float x, int clear_ieee)
{
    if (clear_ieee) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 9, 1), 0");
    float a, b, c, d; unsigned mode;
    asm volatile("v_mul_f32_e64 %0, %1, %1 mul:2" : "=v"(a) : "v"(x));
    asm volatile("v_rsq_f32_e64 %0, %1 div:2" : "=v"(b) : "v"(x));
    asm volatile("v_fma_f32 %0, %1, %1, %1 mul:2" : "=v"(c) : "v"(x));
    asm volatile("v_rsq_f32_e32 %0, %1" : "=v"(d) : "v"(x));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_MODE)" : "=s"(mode));
    out[0] = a; out[1] = b; out[2] = c; out[3] = d; out[4] = __uint_as_float(mode);
}
int main()
{
    float *d; (void)hipMalloc(&d, 64);
    for (int ci = 0; ci < 2; ci++) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 3.0f, ci);
        float h[5]; (void)hipMemcpy(h, d, 20, hipMemcpyDeviceToHost);
        unsigned mode; memcpy(&mode, &h[4], 4);
        printf("clear_ieee=%d MODE=0x%08x (IEEE bit %u, fp_denorm %u): 3*3 mul:2 = %g (18 if honoured), rsq(3) div:2 = %.9g (rsq = %.9g), fma(3,3,3) mul:2 = %g (24)\n",
               ci, mode, (mode >> 9) & 1u, (mode >> 4) & 15u, h[0], h[1], h[3], h[2]);
    }
    return 0;
}
