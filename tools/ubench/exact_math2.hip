// exact_math2.hip -- round-2 candidates for shorter correctly rounded sequences, checked against the compiler's IEEE
// expansions for all 2^32 float inputs (divisions: 2^32 hashed operand pairs per range).  Measurement tool, not product code.
//   sqrt_m    v_sqrt + Markstein correction with h = 0.5 * v_rcp(s0)
//   sqrt_q    v_sqrt + Markstein correction with h = 0.5 * v_rsq(x)
//   rsqrt_m   RN(1 / RN(sqrt x)): sqrt_m, then ONE Newton step on y0 = v_rcp(s0) (the reciprocal of the UNcorrected root)
//   rsqrt_m2  same with two Newton steps
//   div_m     a / b by Markstein on rcp_core(b)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ float sqrt_m(float x)
{
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rcpf(s0);
    const float d = __builtin_fmaf(-s0, s0, x);
    return __builtin_fmaf(d, h, s0);
}
__device__ __forceinline__ float sqrt_q(float x)
{
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float h = 0.5f * __builtin_amdgcn_rsqf(x);
    const float d = __builtin_fmaf(-s0, s0, x);
    return __builtin_fmaf(d, h, s0);
}
__device__ __forceinline__ float rsqrt_m(float x)
{
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float y0 = __builtin_amdgcn_rcpf(s0);
    const float d = __builtin_fmaf(-s0, s0, x);
    const float s = __builtin_fmaf(d, 0.5f * y0, s0);
    const float e = __builtin_fmaf(-s, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float rsqrt_m2(float x)
{
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float y0 = __builtin_amdgcn_rcpf(s0);
    const float d = __builtin_fmaf(-s0, s0, x);
    const float s = __builtin_fmaf(d, 0.5f * y0, s0);
    const float e = __builtin_fmaf(-s, y0, 1.0f);
    const float y1 = __builtin_fmaf(e, y0, y0);
    const float r = __builtin_fmaf(-s, y1, 1.0f);
    return __builtin_fmaf(r, y1, y1);
}
// rsqrt with the root's neighbour test replaced by Markstein, reciprocal by rcp of the corrected root + 1 Newton (2 trans + ...)
__device__ __forceinline__ float rsqrt_m3(float x)
{
    const float s0 = __builtin_amdgcn_sqrtf(x);
    const float y0 = __builtin_amdgcn_rcpf(s0);
    const float d = __builtin_fmaf(-s0, s0, x);
    const float s = __builtin_fmaf(d, 0.5f * y0, s0);
    // y0 approximates 1/s0; 1/s = y0 * (s0/s): first-order fix before the Newton step
    const float e0 = __builtin_fmaf(-s, y0, 1.0f);
    const float y1 = __builtin_fmaf(e0, y0, y0);
    const float e1 = __builtin_fmaf(-s, y1, 1.0f);
    return __builtin_fmaf(e1, y1, y1);
}
// root by sqrt_q; reciprocal of the CORRECTED root from y0 = v_rsq(x) ~ 1/s: one / two Newton steps, or one cubic step
__device__ __forceinline__ float rsqrt_q1(float x)
{
    const float s0 = __builtin_amdgcn_sqrtf(x), y0 = __builtin_amdgcn_rsqf(x);
    const float s = __builtin_fmaf(__builtin_fmaf(-s0, s0, x), 0.5f * y0, s0);
    const float e = __builtin_fmaf(-s, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float rsqrt_q2(float x)
{
    const float s0 = __builtin_amdgcn_sqrtf(x), y0 = __builtin_amdgcn_rsqf(x);
    const float s = __builtin_fmaf(__builtin_fmaf(-s0, s0, x), 0.5f * y0, s0);
    const float e = __builtin_fmaf(-s, y0, 1.0f);
    const float y1 = __builtin_fmaf(e, y0, y0);
    const float r = __builtin_fmaf(-s, y1, 1.0f);
    return __builtin_fmaf(r, y1, y1);
}
__device__ __forceinline__ float rsqrt_q3(float x)
{
    const float s0 = __builtin_amdgcn_sqrtf(x), y0 = __builtin_amdgcn_rsqf(x);
    const float s = __builtin_fmaf(__builtin_fmaf(-s0, s0, x), 0.5f * y0, s0);
    const float e = __builtin_fmaf(-s, y0, 1.0f);
    const float e2 = __builtin_fmaf(e, e, e);
    return __builtin_fmaf(e2, y0, y0);
}
__device__ __forceinline__ float div_m(float a, float b)
{
    const float y0 = __builtin_amdgcn_rcpf(b);
    const float e = __builtin_fmaf(-b, y0, 1.0f);
    const float y = __builtin_fmaf(e, y0, y0);
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r0, y, q0);
}

__device__ __forceinline__ bool same(float a, float b) { return (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b); }

__global__ void k_check(unsigned long long *counts, uint32_t *first_bad, float lo, float hi)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long c[8] = {};
    for (uint64_t i = tid; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        if (!((x >= lo) && (x <= hi))) continue;
        const float ref_s = sqrtf(x), ref_rs = 1.0f / sqrtf(x);
        const float cand[8] = { sqrt_m(x), sqrt_q(x), rsqrt_m(x), rsqrt_m2(x), rsqrt_m3(x), rsqrt_q1(x), rsqrt_q2(x), rsqrt_q3(x) };
        const float ref[8] = { ref_s, ref_s, ref_rs, ref_rs, ref_rs, ref_rs, ref_rs, ref_rs };
#pragma unroll
        for (int k = 0; k < 8; k++)
            if (!same(cand[k], ref[k])) { c[k]++; atomicMin(&first_bad[k], (uint32_t)i); }
    }
    for (int k = 0; k < 8; k++) atomicAdd(&counts[k], c[k]);
}

// a / b: b sweeps ALL bit patterns with |b| in [blo, bhi]; a = hashed, scaled into |a| in [alo, ahi] (or exactly 0 every 1024th)
__global__ void k_check_div(unsigned long long *counts, uint32_t *first_bad, float blo, float bhi, int a_emin, int a_emax)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0, n = 0;
    for (uint64_t i = tid; i < (1ull << 32); i += stride) {
        const float b = __uint_as_float((uint32_t)i);
        if (!((fabsf(b) >= blo) && (fabsf(b) <= bhi))) continue;
        for (int rep = 0; rep < 4; rep++) {
            uint32_t h = ((uint32_t)i + 0x9e3779b9u * (rep + 1)) * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const uint32_t e = (uint32_t)(a_emin + 127) + (h >> 24) % (uint32_t)(a_emax - a_emin + 1);
            float a = __uint_as_float((h & 0x807fffffu) | (e << 23));
            if ((h & 0x3ff000u) == 0) a = 0.0f;
            n++;
            if (!same(div_m(a, b), a / b)) { c++; atomicMin(&first_bad[0], (uint32_t)i); }
        }
    }
    atomicAdd(&counts[0], c); atomicAdd(&counts[1], n);
}

int main()
{
    unsigned long long *counts; uint32_t *first_bad;
    (void)hipMalloc(&counts, 8 * 8); (void)hipMalloc(&first_bad, 8 * 4);
    const char *names[8] = { "sqrt_m  (Markstein, h = rcp(s0)/2)", "sqrt_q  (Markstein, h = rsq(x)/2)", "rsqrt_m (1 Newton on rcp(s0))",
                             "rsqrt_m2 (2 Newton)", "rsqrt_m3", "rsqrt_q1 (rsq, 1 Newton)", "rsqrt_q2 (rsq, 2 Newton)", "rsqrt_q3 (rsq, cubic step)" };
    const float ranges[3][2] = { { 0x1p-100f, 0x1p100f }, { 0x1p-120f, 0x1p120f }, { 0x1p-125f, 0x1p126f } };
    for (int r = 0; r < 3; r++) {
        (void)hipMemset(counts, 0, 64); (void)hipMemset(first_bad, 0xff, 32);
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, counts, first_bad, ranges[r][0], ranges[r][1]);
        unsigned long long h[8]; uint32_t fb[8];
        (void)hipMemcpy(h, counts, 64, hipMemcpyDeviceToHost); (void)hipMemcpy(fb, first_bad, 32, hipMemcpyDeviceToHost);
        printf("range [%g, %g]\n", ranges[r][0], ranges[r][1]);
        for (int k = 0; k < 8; k++) printf("  %-40s mismatches %llu (first bad bits 0x%08x)\n", names[k], h[k], fb[k]);
    }
    const struct { float blo, bhi; int emin, emax; } dv[3] = { { 1.0f, 0x1p100f, -60, 20 }, { 0x1p-40f, 0x1p40f, -40, 40 }, { 1.0f, 0x1p126f, -100, 30 } };
    for (int r = 0; r < 3; r++) {
        (void)hipMemset(counts, 0, 64); (void)hipMemset(first_bad, 0xff, 32);
        hipLaunchKernelGGL(k_check_div, dim3(4096), dim3(256), 0, 0, counts, first_bad, dv[r].blo, dv[r].bhi, dv[r].emin, dv[r].emax);
        unsigned long long h[8]; uint32_t fb[8];
        (void)hipMemcpy(h, counts, 64, hipMemcpyDeviceToHost); (void)hipMemcpy(fb, first_bad, 32, hipMemcpyDeviceToHost);
        printf("div_m: |b| in [%g, %g], |a| in 2^[%d, %d] or 0: mismatches %llu of %llu (first bad b bits 0x%08x)\n", dv[r].blo, dv[r].bhi, dv[r].emin, dv[r].emax, h[0], h[1], fb[0]);
    }
    return 0;
}
