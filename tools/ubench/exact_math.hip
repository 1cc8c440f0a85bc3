// exact_math.hip -- exhaustive check (all 2^32 float bit patterns) that candidate short instruction
// sequences reproduce the compiler's correctly rounded sqrtf(x), 1.0f/x and 1.0f/sqrtf(x) on gfx950.
// Measurement / verification tool, not product code.  Build with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ float sqrt_a(float x)     // v_sqrt + one-ulp neighbour test
{
    float s = __builtin_amdgcn_sqrtf(x);
    float s_dn = __int_as_float(__float_as_int(s) - 1), s_up = __int_as_float(__float_as_int(s) + 1);
    float r_dn = __builtin_fmaf(-s_dn, s, x), r_up = __builtin_fmaf(-s_up, s, x);
    s = (r_dn <= 0.0f) ? s_dn : s;
    s = (r_up > 0.0f) ? s_up : s;
    return s;
}
__device__ __forceinline__ float sqrt_b(float x, float &h_out)   // rsq + Goldschmidt-style refinement
{
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y, h = 0.5f * y;
    float e = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, e, g);
    h = __builtin_fmaf(h, e, h);
    float d = __builtin_fmaf(-g, g, x);
    g = __builtin_fmaf(d, h, g);
    h_out = h;
    return g;
}
__device__ __forceinline__ float rcp_a(float b)       // v_rcp + 2 Newton steps on the reciprocal
{
    float y0 = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, y0, 1.0f);
    float y1 = __builtin_fmaf(e, y0, y0);
    float r = __builtin_fmaf(-b, y1, 1.0f);
    return __builtin_fmaf(r, y1, y1);
}
__device__ __forceinline__ float rcp_b(float b)       // v_rcp + 1 Newton step
{
    float y0 = __builtin_amdgcn_rcpf(b);
    float e = __builtin_fmaf(-b, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float rsqrt_c(float x)     // RN(1/RN(sqrt x)) without v_rcp: reuse h ~ 0.5/sqrt(x)
{
    float h;
    float g = sqrt_b(x, h);
    float y0 = h + h;
    float e = __builtin_fmaf(-g, y0, 1.0f);
    float y1 = __builtin_fmaf(e, y0, y0);
    float r = __builtin_fmaf(-g, y1, 1.0f);
    return __builtin_fmaf(r, y1, y1);
}
__device__ __forceinline__ float rsqrt_d(float x)     // same, one refinement less
{
    float h;
    float g = sqrt_b(x, h);
    float y0 = h + h;
    float e = __builtin_fmaf(-g, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
__device__ __forceinline__ float rsqrt_e(float x) { return rcp_a(sqrt_a(x)); }

// counts[k][0] = mismatches inside [lo, hi], counts[k][1] = mismatches anywhere (incl. NaN payload diffs ignored)
__global__ void k_check(unsigned long long *counts, uint32_t *first_bad, float lo, float hi)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long c[8][2] = {};
    for (uint64_t i = tid; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        const bool in = (x >= lo) && (x <= hi);
        const float ref_s = sqrtf(x), ref_r = 1.0f / x, ref_rs = 1.0f / sqrtf(x);
        float h;
        const float cand[8] = { sqrt_a(x), sqrt_b(x, h), rcp_a(x), rcp_b(x), rsqrt_c(x), rsqrt_d(x), rsqrt_e(x), 0.0f };
        const float ref[8] = { ref_s, ref_s, ref_r, ref_r, ref_rs, ref_rs, ref_rs, 0.0f };
#pragma unroll
        for (int k = 0; k < 7; k++) {
            const bool same = (__float_as_uint(cand[k]) == __float_as_uint(ref[k])) || (cand[k] != cand[k] && ref[k] != ref[k]);
            if (!same) {
                c[k][1]++;
                if (in) { c[k][0]++; atomicMin(&first_bad[k], (uint32_t)i); }
            }
        }
        // rcp also for negative arguments with |x| in range
        const bool in_neg = (-x >= lo) && (-x <= hi);
        if (in_neg) {
            if (__float_as_uint(cand[2]) != __float_as_uint(ref[2])) { c[2][0]++; atomicMin(&first_bad[2], (uint32_t)i); }
            if (__float_as_uint(cand[3]) != __float_as_uint(ref[3])) { c[3][0]++; atomicMin(&first_bad[3], (uint32_t)i); }
        }
    }
    for (int k = 0; k < 7; k++) { atomicAdd(&counts[k * 2], c[k][0]); atomicAdd(&counts[k * 2 + 1], c[k][1]); }
}

int main()
{
    unsigned long long *counts; uint32_t *first_bad;
    hipMalloc(&counts, 16 * 8); hipMalloc(&first_bad, 8 * 4);
    const char *names[7] = { "sqrt_a (v_sqrt + neighbour test)", "sqrt_b (rsq refinement)", "rcp_a (2 Newton)", "rcp_b (1 Newton)",
                             "rsqrt_c (rsq, 2 refinements)", "rsqrt_d (rsq, 1 refinement)", "rsqrt_e (rcp_a(sqrt_a))" };
    const float ranges[3][2] = { { 0x1p-100f, 0x1p100f }, { 0x1p-120f, 0x1p120f }, { 1.17549435e-38f, 3.4e38f } };
    for (int r = 0; r < 3; r++) {
        hipMemset(counts, 0, 16 * 8); hipMemset(first_bad, 0xff, 8 * 4);
        hipLaunchKernelGGL(k_check, dim3(4096), dim3(256), 0, 0, counts, first_bad, ranges[r][0], ranges[r][1]);
        unsigned long long h[16]; uint32_t fb[8];
        hipMemcpy(h, counts, 16 * 8, hipMemcpyDeviceToHost); hipMemcpy(fb, first_bad, 32, hipMemcpyDeviceToHost);
        printf("range [%g, %g]\n", ranges[r][0], ranges[r][1]);
        for (int k = 0; k < 7; k++) printf("  %-36s mismatches in range %llu (first bad bits 0x%08x), anywhere %llu\n", names[k], h[2 * k], fb[k], h[2 * k + 1]);
    }
    return 0;
}
