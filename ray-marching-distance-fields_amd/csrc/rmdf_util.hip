// rmdf_util.hip -- small gfx950 kernels around the render kernel: exhaustive self-tests of the short exact
// arithmetic sequences, frame clear, shard assembly after the gather, super-sampling resolve.
#include "rmdf_internal.hpp"

namespace rmdf {

// ------------------------------------------------------------------------------------
// Self-test of the short correctly-rounded sequences of rmdf_device.hpp against hipcc's own IEEE expansions,
// over ALL 2^32 float bit patterns: sqrt_rn vs sqrtf, rcp_rn vs 1.0f/x, log_pinned (whose internal quotient uses
// div_known_range) vs the same algorithm with the compiler's division; counts[5..7]: the Mandelbulb loop's bailout test on
// the squared radius and its two in-loop roots vs the written forms.  counts[k] = number of differing inputs
// (NaN results compare equal to NaN).  ~1 s on an MI355X.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ float log_ref_division(float x)
{
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    int32_t ix = __float_as_int(x);
    int32_t k = 0;
    if (ix < 0x00800000) {
        if ((ix & 0x7fffffff) == 0) return -__builtin_inff();
        if (ix < 0) return __builtin_nanf("");
        k = -25; x = x * 33554432.0f; ix = __float_as_int(x);
    }
    if (ix >= 0x7f800000) return x + x;
    k += (ix >> 23) - 127;
    ix &= 0x007fffff;
    const int32_t i = (ix + 0x4afb20) & 0x800000;
    x = __int_as_float(ix | (i ^ 0x3f800000));
    k += (i >> 23);
    const float f = x - 1.0f;
    const float s = f / (2.0f + f);
    const float dk = (float)k;
    const float z = s * s, w = z * z;
    const float t1 = w * (Lg2 + w * Lg4), t2 = z * (Lg1 + w * Lg3);
    const float R = t2 + t1;
    const float hfsq = (0.5f * f) * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

// div_by_table against the compiler's division: every numerator bit pattern x the 96 Cornell divisors
__global__ void k_selftest_cornell_div(unsigned long long *counts, const float *__restrict__ tab)
{
    const int tri = blockIdx.y / 3, which = blockIdx.y % 3;
    const float *t = tab + tri * CORNELL_STRIDE;
    const float len = which == 0 ? t[15] : which == 1 ? t[17] : t[22];
    const float rlen = which == 0 ? t[23] : which == 1 ? t[24] : t[25];
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (uint64_t i = tid; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        const float a = div_by_table(x, len, rlen), b = x / len;
        c += !((__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b));
    }
    if (c) atomicAdd(&counts[4], c);
}

__global__ void k_selftest_exact_math(unsigned long long *counts)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0, c5 = 0, c6 = 0, c7 = 0;
    for (uint64_t i = tid; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        // the Mandelbulb loop's forms (de_mandelbulb8): the bailout test on the squared radius, and the two roots of an
        // iteration that did not escape (d = x with k3 = x covers every d and every k3 the loop can hold)
        c5 += (sqrtf(x) > 4.0f) != (x > RMDF_MB8_D4);
        {
            // the estimate's final division: every bit pattern as the numerator against a hashed dr >= 1 (exponents 0..75 densely,
            // the guard's edge and beyond now and then), and every bit pattern as dr against a hashed numerator
            uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
            const float dr1 = __uint_as_float((h & 0x007fffffu) | ((127u + (h >> 23) % ((h >> 31) ? 128u : 76u)) << 23));
            const float a2 = __uint_as_float((h & 0x807fffffu) | ((127u - 40u + (h >> 24) % 48u) << 23));
            const float q1 = div_by_dr(x, dr1), e1 = x / dr1, q2 = div_by_dr(a2, x), e2 = a2 / x;
            c5 += !((__float_as_uint(q1) == __float_as_uint(e1)) || (q1 != q1 && e1 != e1));
            c5 += !((__float_as_uint(q2) == __float_as_uint(e2)) || (q2 != q2 && e2 != e2));
        }
        if (!(x > RMDF_MB8_D4)) {
            const float q = x * x * x * x * x * x * x;
            float r, k2;
            mb8_roots(x, x, q, r, k2);
            const float rr = sqrtf(x), kk = 1.0f / sqrtf(q);
            c6 += !((__float_as_uint(r) == __float_as_uint(rr)) || (r != r && rr != rr));
            c7 += !((__float_as_uint(k2) == __float_as_uint(kk)) || (k2 != k2 && kk != kk));
        }
        const float a0 = sqrt_rn(x), b0 = sqrtf(x);
        const float a1 = rcp_rn(x), b1 = 1.0f / x;
        const float a2 = log_pinned(x), b2 = log_ref_division(x);
        const float a3 = rsqrt_ieee(x), b3 = 1.0f / sqrtf(x);
        c0 += !((__float_as_uint(a0) == __float_as_uint(b0)) || (a0 != a0 && b0 != b0));
        c1 += !((__float_as_uint(a1) == __float_as_uint(b1)) || (a1 != a1 && b1 != b1));
        c2 += !((__float_as_uint(a2) == __float_as_uint(b2)) || (a2 != a2 && b2 != b2));
        c3 += !((__float_as_uint(a3) == __float_as_uint(b3)) || (a3 != a3 && b3 != b3));
    }
    atomicAdd(&counts[0], c0); atomicAdd(&counts[1], c1); atomicAdd(&counts[2], c2); atomicAdd(&counts[3], c3);
    atomicAdd(&counts[5], c5); atomicAdd(&counts[6], c6); atomicAdd(&counts[7], c7);
}

// The folded Mandelbulb passes (rmdf_device.hpp: mb8_iterate_t<true> behind its running-minimum guard, with the written passes as
// the fall-back) against the written loop, on 2^28 hashed points: uniform in the bounding cube, and with one, two or three
// coordinates scaled down to 2^-20 .. 2^-150 -- where partial products DO underflow and the guard has to send the estimate back
// through the written form.  counts[8] = estimates that differ in distance bits or iteration count (must be 0), counts[9] = estimates
// that took the fall-back (must not be 0: the test has to reach it).
__global__ void k_selftest_mb8_folds(unsigned long long *counts)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0, fell = 0;
    for (uint64_t i = tid; i < (1ull << 28); i += stride) {
        uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        uint32_t g = (h ^ 0x9e3779b9u) * 3266489917u; g ^= g >> 16;
        uint32_t f = (g + 0x7f4a7c15u) * 668265263u; f ^= f >> 15;
        float c[3] = { ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f) * 1.3f, ((float)(g >> 8) * (1.0f / 8388608.0f) - 1.0f) * 1.3f,
                       ((float)(f >> 8) * (1.0f / 8388608.0f) - 1.0f) * 1.3f };
        const unsigned mode = (unsigned)(i & 7u);                       // 0..3: plain; 4..7: some coordinates tiny
        if (mode >= 4u) {
            const int e = -20 - (int)((h & 0xffu) % 131u);              // 2^-20 .. 2^-150
            const float sc = __builtin_ldexpf(1.0f, e);
            if (mode == 4u || mode == 7u) c[h % 3u] *= sc;
            if (mode == 5u) { c[0] *= sc; c[1] *= sc; }
            if (mode == 6u) { c[1] *= sc; c[2] *= sc; }
            if (mode == 7u) c[(h + 1u) % 3u] *= __builtin_ldexpf(1.0f, -(int)(g & 63u));
        }
        const v3 pos = mk3(c[0], c[1], c[2]);
        unsigned ia = 0u, ib = 0u;
        unsigned nw = 0u;                                               // did the folded estimate fall back to the written form?
        const float a = de_mandelbulb8(pos, ia, RMDF_MB8_FOLD_MIN, &nw), b = de_mandelbulb8_written(pos, ib);
        bad += !(((__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b)) && ia == ib);
        fell += nw ? 1u : 0u;
    }
    if (bad) atomicAdd(&counts[8], bad);
    if (fell) atomicAdd(&counts[9], fell);
}

// Self-test of the straight-line pinned functions (rmdf_device.hpp) against their branchy fdlibm-style forms, over ALL 2^32
// float bit patterns: exp, acos, atan, sin, cos; atan2 and pow over 2^32 pseudo-random operand pairs (every bit pattern of y
// paired with a hashed x).  counts[0..6] = differing inputs (NaN == NaN).
__global__ void k_selftest_pinned_math(unsigned long long *counts)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long c[7] = { 0, 0, 0, 0, 0, 0, 0 };
    auto differ = [](float a, float b) { return !((__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b)); };
    for (uint64_t i = tid; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
        c[0] += differ(exp_pinned(x), exp_full(x));
        c[1] += differ(acos_pinned(x), acos_full(x));
        c[2] += differ(atan_pinned(x), atan_full(x));
        float s0, c0, s1, c1;
        sincos_pinned(x, s0, c0);
        sincos_full(x, s1, c1);
        c[3] += differ(s0, s1);
        c[4] += differ(c0, c1);
        uint32_t h = (uint32_t)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        // every fourth pair uses a "shader-like" second operand (finite, moderate) so that the main path is exercised densely
        const float y = (i & 3) ? __uint_as_float(h) : (float)((int)(h >> 8) - 8388608) * (1.0f / 1048576.0f);
        // another fourth: both operands forced into [2^-40, 2^40), the range in which atan2_pinned divides the short way (every
        // mantissa and sign of the first operand against a hashed second)
        const bool both_in = (i & 3) == 1;
        const float xa = both_in ? __uint_as_float(((uint32_t)i & 0x807fffffu) | ((87u + (((uint32_t)i >> 23) & 0xffu) % 80u) << 23)) : x;
        const float ya = both_in ? __uint_as_float((h & 0x807fffffu) | ((87u + ((h >> 23) & 0xffu) % 80u) << 23)) : y;
        c[5] += differ(atan2_pinned(xa, ya), atan2_full(xa, ya));
        const float pw = 2.0f + (float)(h & 1023u) * (4.5f / 1023.0f);         // the animated power range 2 .. 6.5
        const float pr = !(x > 0.0f) ? 0.0f : exp_full(pw * log_ref_division(x));
        c[6] += differ(pow_pinned(x, pw), pr);
    }
    for (int k = 0; k < 7; k++) if (c[k]) atomicAdd(&counts[k], c[k]);
}

hipError_t launch_selftest_pinned_math(unsigned long long *d_counts, hipStream_t stream)
{
    hipLaunchKernelGGL(k_selftest_pinned_math, dim3(8192), dim3(256), 0, stream, d_counts);
    return hipGetLastError();
}

// Self-test of the short quotients of the shading tail (round 3) against the compiler's division.
//   counts[0]: div_known_range itself, 2^33 operand pairs inside its stated range: every bit pattern as the numerator (exponent
//              folded into 2^-40 .. 2^40, and zero) over a hashed divisor, and every bit pattern as the divisor under a hashed numerator;
//   counts[1]: the ambient-occlusion term clamp(1 - d / e) for EVERY d (infinities and NaN included) and each of the six tap offsets e;
//   counts[2]: fresnel_conductor (eta 0.4, k 0.8) for every cosi in [-2, 2], +-inf and every NaN;
//   counts[3]: cube_texture on 2^30 hashed (direction, horizontal neighbour, vertical neighbour) triples of what the shader can hand
//              it -- normalize()'s results: unit vectors (neighbours near and far, components down to 2^-149 and zero, a neighbour whose
//              component along this lane's major axis is tiny or of the other sign) and the inf / NaN vectors a vanishing gradient
//              normalises to; the whole result (three floats of the fetched texel) must agree.
__device__ __forceinline__ uint32_t hash32(uint32_t h) { h *= 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16; return h; }
__device__ __forceinline__ float fold_exponent(uint32_t bits) { return __uint_as_float((bits & 0x807fffffu) | ((87u + ((bits >> 23) & 0xffu) % 80u) << 23)); }
__device__ __forceinline__ bool same_bits(float a, float b) { return (__float_as_uint(a) == __float_as_uint(b)) || (a != a && b != b); }

__device__ v3 selftest_vector(uint32_t h, uint32_t mode, v3 near)
{
    const uint32_t h1 = hash32(h ^ 0x68bc21ebu), h2 = hash32(h ^ 0x02e5be93u), h3 = hash32(h + 0x7f4a7c15u);
    v3 a = mk3((float)(h1 >> 8) * (1.0f / 8388608.0f) - 1.0f, (float)(h2 >> 8) * (1.0f / 8388608.0f) - 1.0f, (float)(h3 >> 8) * (1.0f / 8388608.0f) - 1.0f);
    switch (mode & 7u) {
    case 0: case 1: return normalize3(a);                                                    // any unit vector
    case 2: case 3: {                                                                        // a close neighbour (2^-3 .. 2^-18 away)
        const float sc = __builtin_ldexpf(1.0f, -3 - (int)(h1 & 15u));
        return normalize3(mk3(near.x + a.x * sc, near.y + a.y * sc, near.z + a.z * sc));
    }
    case 4: {                                                                                // unit vector with one or two tiny components
        const float sc = __builtin_ldexpf(1.0f, -(int)(h1 % 150u));
        float c[3] = { a.x, a.y, a.z };
        c[h2 % 3u] *= sc;
        if (h3 & 1u) c[(h2 + 1u) % 3u] *= __builtin_ldexpf(1.0f, -(int)(h3 % 150u));
        if (h3 & 2u) c[h2 % 3u] = (h3 & 4u) ? 0.0f : -0.0f;
        return normalize3(mk3(c[0], c[1], c[2]));
    }
    case 5: {                                                                                // `near` with ONE component tiny, zero or flipped, normalised
        float c[3] = { near.x, near.y, near.z };
        const float tiny = __builtin_ldexpf((h3 & 8u) ? -1.0f : 1.0f, -(int)(h1 % 150u));
        c[h2 % 3u] = (h3 & 3u) == 0u ? 0.0f : ((h3 & 3u) == 1u ? -c[h2 % 3u] : tiny);
        return normalize3(mk3(c[0], c[1], c[2]));                                            // (0, 0, 2^-100) -> components inf / NaN, as in the shader
    }
    case 6: {                                                                                // what normalize() makes of a vanishing gradient: inf / NaN components
        const float inf = __builtin_inff(), nan = __builtin_nanf("");
        return mk3((h1 & 1u) ? nan : ((h1 & 2u) ? inf : -inf), (h2 & 1u) ? nan : ((h2 & 2u) ? inf : -inf), (h3 & 1u) ? nan : ((h3 & 2u) ? inf : -inf));
    }
    default:                                                                                 // a NaN component now and then, else a unit vector
        if ((h1 & 63u) == 0u) return mk3(__builtin_nanf(""), a.y, a.z);
        return normalize3(a);
    }
}

__global__ void k_selftest_shading_math(unsigned long long *counts, CubeDev cube)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (uint64_t i = tid; i < (1ull << 32); i += stride) {
        const uint32_t bits = (uint32_t)i, h = hash32(bits);
        const float x = __uint_as_float(bits);
        {
            // (a numerator of -0 comes back as +0: every call site adds a non-zero constant to the quotient or squares it)
            const float a = (bits & 0x7fffffffu) == 0u ? 0.0f : fold_exponent(bits), b = fold_exponent(h);
            c0 += !same_bits(div_known_range(a, b), a / b);
            const float a2 = (h & 0xffu) == 0u ? 0.0f : fold_exponent(hash32(h)), b2 = fold_exponent(bits);
            c0 += !same_bits(div_known_range(a2, b2), a2 / b2);
            c0 += !same_bits(div_with_rcp(a, b, rcp_core(b)), a / b);
        }
        {
            const float e[6] = { 0.016f, 0.081f, 0.1f, 0.2f, 0.4f, 0.5f };
            const float y[6] = { 1.0f / 0.016f, 1.0f / 0.081f, 1.0f / 0.1f, 1.0f / 0.2f, 1.0f / 0.4f, 1.0f / 0.5f };
#pragma unroll
            for (int k = 0; k < 6; k++)
                c1 += !same_bits(ao_term<true>(x, e[k], y[k]), ao_term<false>(x, e[k], y[k]));
        }
        if (!(fabsf(x) > 2.0f) || fabsf(x) == __builtin_inff()) c2 += !same_bits(fresnel_conductor<true>(x, 0.4f, 0.8f), fresnel_conductor<false>(x, 0.4f, 0.8f));
        if ((bits >> 30) == 0u) {
            const v3 r = selftest_vector(h, h >> 29, normalize3(mk3((float)(h & 3u), (float)((h >> 2) & 3u), 1.0f)));
            const uint32_t g = hash32(h ^ 0x5bd1e995u);
            const v3 rh = selftest_vector(g, g >> 29, r), rv = selftest_vector(~g, g >> 26, r);
            const bool vh = (g & 3u) != 0u, vv = (g & 12u) != 0u;
            const v3 ta = cube_texture<true>(cube, r, vh, rh, vv, rv), tb = cube_texture<false>(cube, r, vh, rh, vv, rv);
            c3 += !(same_bits(ta.x, tb.x) && same_bits(ta.y, tb.y) && same_bits(ta.z, tb.z));
        }
    }
    if (c0) atomicAdd(&counts[0], c0);
    if (c1) atomicAdd(&counts[1], c1);
    if (c2) atomicAdd(&counts[2], c2);
    if (c3) atomicAdd(&counts[3], c3);
}

// texels of the self-test's cube map: every texel a different colour, so that two forms that pick different texels (or different
// bilinear weights) cannot agree by accident
__global__ void k_selftest_fill_cube(uint2 *texels, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = hash32((uint32_t)i + 12345u);
    const __half2 rg = __floats2half2_rn((float)(h & 1023u) * (1.0f / 64.0f), (float)((h >> 10) & 1023u) * (1.0f / 64.0f));
    const __half2 b0 = __floats2half2_rn((float)((h >> 20) & 1023u) * (1.0f / 64.0f), 0.0f);
    texels[i] = make_uint2(*reinterpret_cast<const uint32_t *>(&rg), *reinterpret_cast<const uint32_t *>(&b0));
}

// generate_ray's three quotients over EVERY frame fill_params accepts (1 <= w, h <= RMDF_MAX_FRAME_SIDE): counts[4] +=
//  * blockIdx.y = 0: (px + 0.5) / w for every side w and every pixel centre 0 .. w of it (one helper pixel past the edge included);
//    the same table serves (py + 0.5) / h;
//  * blockIdx.y = 1: for every (w, h) the reciprocal of aspect = w / h (a correctly rounded reciprocal makes the Markstein quotient
//    the IEEE one for EVERY numerator that does not underflow), and the quotient ndc.y * fov / aspect itself for the first, the
//    middle and the last row of that frame.
// The short forms against the compiler's division, as everywhere in this file.
__global__ void k_selftest_frame_quotients(unsigned long long *counts, float fov_xs)
{
    const uint64_t tid = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (uint64_t)gridDim.x * blockDim.x;
    const uint64_t S = RMDF_MAX_FRAME_SIDE;
    unsigned long long c = 0;
    if (blockIdx.y == 0) {
        for (uint64_t i = tid; i < S * (S + 1); i += stride) {
            const uint32_t w = (uint32_t)(i / (S + 1)) + 1u, px = (uint32_t)(i % (S + 1));
            if (px > w) continue;
            const float a = (float)px + 0.5f, b = (float)w;
            c += !same_bits(div_known_range(a, b), a / b);
        }
    } else {
        for (uint64_t i = tid; i < S * S; i += stride) {
            const uint32_t w = (uint32_t)(i / S) + 1u, h = (uint32_t)(i % S) + 1u;
            const float wf = (float)w, hf = (float)h, aspect = wf / hf;
            c += !same_bits(rcp_core(aspect), 1.0f / aspect);
            const uint32_t rows[3] = { 0u, h / 2u, h - 1u };
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const float ndcy = div_known_range((float)rows[k] + 0.5f, hf) * 2.0f - 1.0f, n = ndcy * fov_xs;
                c += !same_bits(div_known_range(n, aspect), n / aspect);
            }
        }
    }
    if (c) atomicAdd(&counts[4], c);
}

hipError_t launch_selftest_shading_math(unsigned long long *d_counts, void *d_texels, int face_w, float fov_xs, hipStream_t stream)
{
    const int n = 6 * (face_w + 2) * (face_w + 2);
    hipLaunchKernelGGL(k_selftest_fill_cube, dim3((n + 255) / 256), dim3(256), 0, stream, (uint2 *)d_texels, n);
    CubeDev cube;
    cube.texels = (const uint2 *)d_texels;
    cube.W = face_w;
    hipLaunchKernelGGL(k_selftest_shading_math, dim3(8192), dim3(256), 0, stream, d_counts, cube);
    hipLaunchKernelGGL(k_selftest_frame_quotients, dim3(4096, 2), dim3(256), 0, stream, d_counts, fov_xs);
    return hipGetLastError();
}

hipError_t launch_selftest_exact_math(unsigned long long *d_counts, const float *d_cornell_tab, hipStream_t stream)
{
    hipLaunchKernelGGL(k_selftest_exact_math, dim3(8192), dim3(256), 0, stream, d_counts);
    hipLaunchKernelGGL(k_selftest_cornell_div, dim3(256, 96), dim3(256), 0, stream, d_counts, d_cornell_tab);
    hipLaunchKernelGGL(k_selftest_mb8_folds, dim3(4096), dim3(256), 0, stream, d_counts);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------
// Shader-clock probe: one wave stamps s_memtime (shader cycles) and s_memrealtime (100 MHz) around a spin of `ticks` real-time
// ticks.  Launched beside running frames it reports the clock the shader engines sustain under THAT load (rmdf_probe_shader_clock).
// ------------------------------------------------------------------------------------
__global__ void k_clock_probe(unsigned long long *out, unsigned long long ticks, int busy)
{
    float a = (float)threadIdx.x, b = 1.0001f;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime(), c0 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = r0;
    while (r1 - r0 < ticks) {
        if (busy) { for (int i = 0; i < 256; i++) a = a * b + 0.5f; }      // a wave that keeps issuing vector instructions
        else      __builtin_amdgcn_s_sleep(8);
        r1 = __builtin_amdgcn_s_memrealtime();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = r1 - r0; }
    if (a == 12345.678f) out[2] = 1ull;                                    // keeps the arithmetic alive
}

hipError_t launch_clock_probe(unsigned long long *d_out, unsigned long long ticks, int busy, hipStream_t stream)
{
    hipLaunchKernelGGL(k_clock_probe, dim3(1), dim3(64), 0, stream, d_out, ticks, busy);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// small utility kernels
// ------------------------------------------------------------------------------------
__global__ void k_fill_u32(uint32_t *dst, uint32_t value, size_t n)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = value;
}

hipError_t launch_fill_u32(uint32_t *dst, uint32_t value, size_t n, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_fill_u32, dim3(blocks), dim3(256), 0, stream, dst, value, n);
    return hipGetLastError();
}

// After the gather: shard r holds the tiles shard_tiles_of_rank(r, n) in slots 0,1,2,...; every rank's shard has
// ceil(64/n) slots.  where.v[tile idx] = rank << 8 | slot.  One thread per frame pixel (coalesced writes).
__global__ void k_assemble_shards(const uint32_t *__restrict__ gathered, uint32_t *__restrict__ frame,
                                  int w, int h, int nranks, const ShardWhere where)
{
    const int tw = w / 8, th = h / 8;
    const int slots = (64 + nranks - 1) / nranks;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t n = (size_t)w * h, stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) {
        int px = (int)(i % w), py = (int)(i / w);
        int tx = px / tw, ty = py / th;
        const int rs = where.v[tx + ty * 8];
        const int rank = rs >> 8, slot = rs & 255;
        size_t src = ((size_t)rank * slots + slot) * (size_t)tw * th + (size_t)(px - tx * tw) + (size_t)(py - ty * th) * tw;
        frame[i] = gathered[src];
    }
}

// same, four pixels per thread (16-byte loads and stores, one tile lookup per thread): tiles whose width is a multiple of 4.
// HBM-bound: 8 B per pixel (4 read + 4 written).
__global__ void k_assemble_shards_x4(const uint4 *__restrict__ gathered, uint4 *__restrict__ frame,
                                     int w, int h, int nranks, const ShardWhere where)
{
    const int tw4 = w / 32, th = h / 8, w4 = w / 4;            // tile width and frame width in uint4 units
    const int slots = (64 + nranks - 1) / nranks;
    const int x4 = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y;
    if (x4 >= w4) return;
    const int tx = x4 / tw4, ty = py / th;
    const int rs = where.v[tx + ty * 8];
    const int rank = rs >> 8, slot = rs & 255;
    const size_t src = ((size_t)rank * slots + slot) * (size_t)tw4 * th + (size_t)(x4 - tx * tw4) + (size_t)(py - ty * th) * tw4;
    frame[(size_t)py * w4 + x4] = gathered[src];
}

hipError_t launch_assemble_shards(const uint32_t *d_gathered, uint32_t *d_frame, int w, int h, int nranks, const ShardWhere &where,
                                  hipStream_t stream)
{
    if (w % 32 == 0 && ((uintptr_t)d_gathered % 16) == 0 && ((uintptr_t)d_frame % 16) == 0) {
        hipLaunchKernelGGL(k_assemble_shards_x4, dim3((w / 4 + 255) / 256, h), dim3(256), 0, stream,
                           (const uint4 *)d_gathered, (uint4 *)d_frame, w, h, nranks, where);
        return hipGetLastError();
    }
    size_t n = (size_t)w * h;
    int blocks = (int)((n + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_assemble_shards, dim3(blocks), dim3(256), 0, stream, d_gathered, d_frame, w, h, nranks, where);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Super-sampling resolve: one glGenerateMipmap level of the RGBA8 frame (FrameBuffer.hs:153-154,187-195):
// 2x2 box per channel, (a+b+c+d+2)>>2.  One lane per destination pixel; each lane reads two 8-byte
// pairs (coalesced 512 B per wave-row) and writes 4 bytes.
// ------------------------------------------------------------------------------------
__global__ void k_resolve_box2(const uint2 *__restrict__ src, int dw, int dh, uint32_t *__restrict__ dst)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw || y >= dh) return;
    const uint2 top = src[(size_t)(2 * y) * dw + x], bot = src[(size_t)(2 * y + 1) * dw + x];
    // per-channel sums without carries between channels: split even / odd bytes
    const uint32_t lo = (top.x & 0x00ff00ffu) + (top.y & 0x00ff00ffu) + (bot.x & 0x00ff00ffu) + (bot.y & 0x00ff00ffu) + 0x00020002u;
    const uint32_t hi = ((top.x >> 8) & 0x00ff00ffu) + ((top.y >> 8) & 0x00ff00ffu) + ((bot.x >> 8) & 0x00ff00ffu) + ((bot.y >> 8) & 0x00ff00ffu) + 0x00020002u;
    dst[(size_t)y * dw + x] = ((lo >> 2) & 0x00ff00ffu) | (((hi >> 2) & 0x00ff00ffu) << 8);
}

hipError_t launch_resolve_box2(const uint32_t *d_src, int sw, int sh, uint32_t *d_dst, hipStream_t stream)
{
    const int dw = sw / 2, dh = sh / 2;
    if (dw <= 0 || dh <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_resolve_box2, dim3((dw + 255) / 256, dh), dim3(256), 0, stream, (const uint2 *)d_src, dw, dh, d_dst);
    return hipGetLastError();
}

}  // namespace rmdf
