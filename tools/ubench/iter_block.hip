// iter_block.hip -- cost of ONE Mandelbulb iteration pass in isolation (all 64 lanes busy, no divergence).
// Measurement tool.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off [-fno-slp-vectorize] -I../../ray-marching-distance-fields_amd/csrc
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "rmdf_device.hpp"
using namespace rmdf;

#ifndef VARIANT
#define VARIANT 0
#endif

__device__ __forceinline__ v3 triplex_pow8_sq(float x, float y, float z, float x2, float y2, float z2)
{
    const float x4 = x2 * x2, y4 = y2 * y2, z4 = z2 * z2;
    const float k3 = y2 + x2;
#if VARIANT == 1
    const float k2 = 1.0f / sqrtf(k3 * k3 * k3 * k3 * k3 * k3 * k3);
#else
    const float k2 = rsqrt_ieee(k3 * k3 * k3 * k3 * k3 * k3 * k3);
#endif
    const float k1 = y4 + z4 + x4 - 6.0f * z2 * x2 - 6.0f * y2 * z2 + 2.0f * x2 * y2;
    const float k4 = y2 - z2 + x2;
    return mk3(-8.0f * z * k4 * (y4 * y4 - 28.0f * y4 * y2 * x2 + 70.0f * y4 * x4 - 28.0f * y2 * x2 * x4 + x4 * x4) * k1 * k2,
               64.0f * y * z * x * (y2 - x2) * k4 * (y4 - 6.0f * y2 * x2 + x4) * k1 * k2,
               -16.0f * z2 * k3 * k4 * k4 + k1 * k1);
}

__global__ __launch_bounds__(256) void k_iter(float *out, int passes, unsigned long long *cyc)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    const float posx = 0.3f + 1e-4f * (tid & 1023), posy = -0.45f + 1e-5f * (tid >> 3), posz = 0.2f;
    float wx = posx, wy = posy, wz = posz, dr = 1.0f;
    float x2 = wx * wx, y2 = wy * wy, z2 = wz * wz, r = sqrtf((x2 + y2) + z2);
    unsigned iters = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < passes; i++) {
        const v3 nw = triplex_pow8_sq(wx, wy, wz, x2, y2, z2);
        wx = nw.x + posx; wy = nw.y + posy; wz = nw.z + posz;
        const float r2 = r * r, r4 = r2 * r2, r7 = (r4 * r2) * r;
        dr = r7 * 8.0f * dr + 1.0f;
        iters++;
        x2 = wx * wx; y2 = wy * wy; z2 = wz * wz;
#if VARIANT == 1
        r = sqrtf((x2 + y2) + z2);
#else
        r = sqrt_rn((x2 + y2) + z2);
#endif
        // keep every lane inside the set's neighbourhood: restart the orbit when it escapes (select, no branch)
        const bool esc = !(r <= 4.0f);
        wx = esc ? posx : wx; wy = esc ? posy : wy; wz = esc ? posz : wz;
        x2 = esc ? posx * posx : x2; y2 = esc ? posy * posy : y2; z2 = esc ? posz * posz : z2;
        r = esc ? 0.5f : r; dr = esc ? 1.0f : dr;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[tid] = wx + wy + wz + dr + r + (float)iters;
    if ((threadIdx.x & 63) == 0) cyc[tid >> 6] = t1 - t0;
}

int main(int argc, char **argv)
{
    const int passes = 20000;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount;
    for (int wps : { 1, 2, 4 }) {
        const int blocks = cus * wps;
        float *out; unsigned long long *cyc;
        hipMalloc(&out, (size_t)blocks * 256 * 4); hipMalloc(&cyc, (size_t)blocks * 4 * 8);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_iter, dim3(blocks), dim3(256), 0, 0, out, 1000, cyc);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_iter, dim3(blocks), dim3(256), 0, 0, out, passes, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
        printf("variant %d wps %d: %.3f ms, %.1f ns/pass/wave, s_memtime cycles/pass %.1f, SIMD-ns per wave-pass %.1f\n",
               VARIANT, wps, ms, ms * 1e6 / passes, (double)c0 / passes, ms * 1e6 / passes / wps);
        hipFree(out); hipFree(cyc);
    }
    return 0;
}
