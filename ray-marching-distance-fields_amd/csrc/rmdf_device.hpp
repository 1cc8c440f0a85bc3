// rmdf_device.hpp -- per-ray device arithmetic for the gfx950 sphere tracer.
//
// Semantics restated from the reference's fragment shader (fragment.shd, cited
// per function).  float32 throughout, one IEEE rounding per written operation:
// this file MUST be compiled with -ffp-contract=off and without fast-math, so
// that step counts and escape-iteration counts are reproducible bit for bit.
// GLSL built-ins whose precision GLSL leaves open are pinned in DESIGN.md
// ("spec pins"): inversesqrt = 1/sqrt (both correctly rounded), pow(r,7) =
// multiply chain, log/exp = fixed-order fdlibm-style float algorithms.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace rmdf {

// The constants the reference's shader states, by name (value, fragment.shd line).  The kernels and the host code use these names;
// rmdf_get_shader_constants hands the table out, and tests/test_reference_pins.py compares it with the values a script extracted from
// the reference itself (tests/golden/reference_pins.json) -- a typo here cannot hide behind "oracle == kernel".
#define RMDF_SHADER_CONSTANTS(X)                                                                                                  \
    X(mb_bailout, 4.0f)              /* :121 */  X(mb_iterations, 25.0f)         /* :122 */                                        \
    X(march_max_steps_default, 128.0f) /* :634 */ X(march_min_dist, 0.001f)      /* :635 */                                        \
    X(bsphere_r_power8, 1.15f)       /* :643 */  X(bsphere_r_general, 1.5f)      /* :645 */  X(bsphere_r_other, 1.0f)  /* :648 */  \
    X(ao_w0, 0.5f) X(ao_d0, 0.016f)  /* :548-549 */ X(ao_w1, 0.25f) X(ao_d1, 0.081f) /* :552-553 */                                \
    X(ao_bias, 0.29f)                /* :558 */  X(ao_gain, 3.5f)                /* :559 */                                        \
    X(cornell_ao_w0, 0.1f) X(cornell_ao_d0, 0.1f) X(cornell_ao_w1, 0.2f) X(cornell_ao_d1, 0.2f)             /* :571-576 */        \
    X(cornell_ao_w2, 0.125f) X(cornell_ao_d2, 0.4f) X(cornell_ao_w3, 0.0625f) X(cornell_ao_d3, 0.5f)        /* :579-584 */        \
    X(normal_eps, 0.00001f)          /* :466 */  X(isec_step_back, 0.00001f)     /* :751 */                                        \
    X(fresnel_eta, 0.4f) X(fresnel_k, 0.8f)      /* :799 */  X(diff_weight, 0.5f) /* :801 */                                       \
    X(diff_r, 1.0f) X(diff_g, 0.8f) X(diff_b, 0.8f) /* :802 */ X(spec_r, 0.8f) X(spec_g, 0.8f) X(spec_b, 1.0f) /* :803 */          \
    X(spec_weight_one_minus, 1.0f)   /* :804 */  X(phong_lobe_n, 8.0f)           /* :808 */  X(refl_weight, 0.1f)      /* :809 */  \
    X(exposure, 3.0f)                /* :810 */  X(phong_lobe_plus, 2.0f) X(phong_lobe_div, 2.0f)             /* :723 */           \
    X(camera_distance, 2.414213562373095f) /* :897 */ X(camera_cornell_radius, 0.4f) /* :888 */ X(camera_cornell_z, -2.0f) /* :889 */ \
    X(hfov_deg_a, 45.0f) X(hfov_deg_b, 1.5f)     /* :910 */  X(gamma, 2.2f)      /* :959 */
namespace shk {
#define RMDF_X(name, value) constexpr float name = value;
RMDF_SHADER_CONSTANTS(RMDF_X)
#undef RMDF_X
constexpr int mb_iterations_i = (int)mb_iterations;
}  // namespace shk

// Three wave operations under names of their own, because at these places their MEANING does not depend on which other lanes take part:
// RMDF_LANES_HERE(x) is "does some lane that is here with me need the rare path" in front of a per-lane fix-up (`if (any) { if (mine) ... }`)
// or the loop condition of per-lane work lists, RMDF_READLANE_HERE / RMDF_READFIRSTLANE_HERE pick an evaluation ORDER (a hint, a mask
// that only has to be a superset), never a result.  On the device they are the plain instructions (pure text: same ISA).  The CPU tier
// compiles this header one lane at a time (tests/device_on_host.cpp) and the render kernel under a SIMT emulator where __ballot is a true
// 64-lane collective (tests/kernel_on_host.cpp): there these three act on the calling lane alone, which is what makes divergent callers legal.
#ifdef RMDF_HOST_EMULATION
#define RMDF_LANES_HERE(x) ((x) ? 1ull : 0ull)
#define RMDF_READLANE_HERE(v, l) (v)
#define RMDF_READFIRSTLANE_HERE(v) (v)
#ifndef RMDF_EMU_PASS                 // (tests/koh_shim counts iteration passes per lane and per wave: the lane utilisation of a schedule, tools/emulated_schedule.py)
#define RMDF_EMU_PASS()
#define RMDF_EMU_SEGMENT_END(i0)
#define RMDF_EMU_COST(n)
#endif
#else
#define RMDF_EMU_PASS()
#define RMDF_EMU_SEGMENT_END(i0)
#define RMDF_EMU_COST(n)
#define RMDF_LANES_HERE(x) __ballot(x)
#define RMDF_READLANE_HERE(v, l) __builtin_amdgcn_readlane((v), (l))
#define RMDF_READFIRSTLANE_HERE(v) __builtin_amdgcn_readfirstlane(v)
#endif

struct v3 { float x, y, z; };

__device__ __forceinline__ v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }

// GLSL min/max/clamp: min(x,y) = y < x ? y : x; max(x,y) = x < y ? y : x
__device__ __forceinline__ float gmin(float x, float y) { return (y < x) ? y : x; }
__device__ __forceinline__ float gmax(float x, float y) { return (x < y) ? y : x; }
__device__ __forceinline__ float gclamp(float x, float lo, float hi) { return gmin(gmax(x, lo), hi); }

__device__ __forceinline__ float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// Correctly rounded sqrt / reciprocal in a handful of instructions.  hipcc's own expansion of sqrtf(x)
// and 1.0f/x is correctly rounded for every input but spends ~17 / ~14 VALU slots on scaling and
// special-case fix-ups.  For 2^-100 <= |x| <= 2^100 the short "core" sequences below return the SAME
// bits (checked for all 2^32 inputs on gfx950: tools/ubench/exact_math.hip and the gpu test
// test_exact_math_exhaustive); anything outside that range, zero, inf and NaN take the compiler's path.
__device__ __forceinline__ bool in_core_range(float x)
{
    // 2^-100 <= |x| <= 2^100 as one unsigned compare on the exponent field (NaN, inf, 0, subnormals fail)
    return ((__float_as_uint(x) & 0x7fffffffu) - 0x0d800000u) <= (0x71800000u - 0x0d800000u);
}
// y = v_rsq_f32(x) (<= 1 ulp), s0 = RN(x*y), one correction s0 + (x - s0^2) * (y/2): the correctly rounded root for EVERY
// x in [2^-100, 2^100] with ONE transcendental instruction (round 2 used v_sqrt_f32 + a neighbour test: 9 instructions; in
// this kernel's mix a transcendental costs ~4.6 simple issue slots, so sequences that bought fewer instructions with a second
// transcendental did not pay -- NOTEBOOK.md A.1).  y is handed out: it also seeds the reciprocal of the root (rsqrt_ieee).
__device__ __forceinline__ float sqrt_core_y(float x, float &y)       // x in [2^-100, 2^100]
{
    y = __builtin_amdgcn_rsqf(x);
    const float s0 = x * y, h = 0.5f * y;
    return __builtin_fmaf(__builtin_fmaf(-s0, s0, x), h, s0);
}
__device__ __forceinline__ float sqrt_core(float x) { float y; return sqrt_core_y(x, y); }
// RN(1 / s) for s = RN(sqrt x) from the same y: one Newton step.  Correctly rounded for every x in [2^-100, 2^100] except
// where the root's mantissa is all ones (s = 2^k - ulp: the classical exception of Newton reciprocals -- 200 inputs); the
// callers route lanes whose root has its low 16 mantissa bits set (a superset, one 16-bit compare, 2^-16 of all roots)
// through the compiler's expansion together with the out-of-range lanes.
__device__ __forceinline__ float rcp_of_root(float s, float y)
{
    const float e = __builtin_fmaf(-s, y, 1.0f);
    return __builtin_fmaf(e, y, y);
}
__device__ __forceinline__ bool root_needs_slow_rcp(float s) { return (uint16_t)__float_as_uint(s) == (uint16_t)0xffffu; }
// The same test for the whole wave, as the mask of (active) lanes that need it.  Written as the 16-bit compare itself because the compiler's
// form of the line above is v_cmp_eq_u32_sdwa ... src0_sel:WORD_0, and on gfx950 an SDWA instruction among ordinary VALU work costs
// about one and a half issue slots more than a plain one (tools/ubench/valu_rates sdwa -> profiles/r04_sdwa_cost.txt: 7 v_mul + 1
// compare run at 0.87 G instructions/s/SIMD with the SDWA compare, 0.93 with v_cmp_eq_u16).  Used by the guards of the power-8 iteration,
// the hottest loop of the library (-2 % on the headline frame); rsqrt_ieee keeps the compiler's form, which measured 1 % better on the Cornell box.
__device__ __forceinline__ unsigned long long root_needs_slow_rcp_lanes(float s)
{
#ifdef RMDF_HOST_EMULATION       // tests/device_on_host.cpp: this header compiled for the CPU, one lane at a time (no gfx950 assembly there)
    return root_needs_slow_rcp(s) ? 1ull : 0ull;
#else
    unsigned long long lanes;
    asm("v_cmp_eq_u16_e64 %0, %1, %2" : "=s"(lanes) : "v"(__float_as_uint(s)), "s"(0xffffu));
    return lanes;
#endif
}
__device__ __forceinline__ float rcp_core(float x)         // |x| in [2^-100, 2^100]
{
    const float y0 = __builtin_amdgcn_rcpf(x);                            // <= 1 ulp
    const float e = __builtin_fmaf(-x, y0, 1.0f);
    return __builtin_fmaf(e, y0, y0);
}
// The core sequence runs unconditionally; only when some lane of the wave is out of range (rare: the
// branch is wave-uniform) are those lanes recomputed with the compiler's expansion.
__device__ __forceinline__ float sqrt_rn(float x)
{
    float s = sqrt_core(x);
    // one unsigned compare on the raw bits: negative numbers have the sign bit set and fall out of range too
    const bool bad = (__float_as_uint(x) - 0x0d800000u) > (0x71800000u - 0x0d800000u);
    if (__builtin_expect(RMDF_LANES_HERE(bad) != 0ull, 0)) { if (bad) s = sqrtf(x); }
    return s;
}
__device__ __forceinline__ float rcp_rn(float x)
{
    float y = rcp_core(x);
    const bool bad = !in_core_range(x);
    if (__builtin_expect(RMDF_LANES_HERE(bad) != 0ull, 0)) { if (bad) y = 1.0f / x; }
    return y;
}
// a / b for operands whose range is known at the call site: |b| in [2^-40, 2^40], a = 0 or |a| in
// [2^-40, 2^40] (no range test).  y = RN(1/b); q0 = RN(a*y); q1 = RN(q0 + (a - b*q0)*y) is the correctly
// rounded quotient (Markstein); checked against the compiler's division in test_exact_math_exhaustive.
__device__ __forceinline__ float div_known_range(float a, float b)
{
    const float y = rcp_core(b);
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r0, y, q0);
}

// the same quotient with y = RN(1/b) supplied by the caller (several numerators over one divisor; a constant divisor)
__device__ __forceinline__ float div_with_rcp(float a, float b, float y)
{
    const float q0 = a * y;
    const float r0 = __builtin_fmaf(-b, q0, a);
    return __builtin_fmaf(r0, y, q0);
}
// One term of the distance-field ambient occlusion, clamp(1 - d / e, 0, 1) (fragment.shd:542-591), e a compile-time tap offset and
// y = RN(1 / e) folded by the compiler.  d is a distance estimate: usually within a few units, but +inf and NaN do occur (a sample
// point next to the Mandelbulb's pole overflows its triplex power), so an infinite first quotient is returned as it is (the
// remainder step would turn it into NaN, where the division gives inf and the term clamps to 0).  |d| below 2^-102 leaves a quotient
// the "1 -" absorbs in either form.  rmdf_selftest_shading_math: every d bit pattern x every tap offset.
template <bool FAST>
__device__ __forceinline__ float ao_term(float d, float e, float y)
{
    if (!FAST) return gclamp(1.0f - d / e, 0.0f, 1.0f);
    const float q0 = d * y;
    const float r0 = __builtin_fmaf(-e, q0, d);
    float q = __builtin_fmaf(r0, y, q0);
    q = (fabsf(q0) == __builtin_inff()) ? q0 : q;
    return gclamp(1.0f - q, 0.0f, 1.0f);
}

__device__ __forceinline__ float length3(v3 a) { return sqrt_rn(dot3(a, a)); }
// inversesqrt := 1/sqrt, two roundings.  If x is in the core range so is sqrt(x) (2^-50 .. 2^50): one range test.
__device__ __forceinline__ float rsqrt_ieee(float x)
{
    float y0;
    const float s = sqrt_core_y(x, y0);
    float y = rcp_of_root(s, y0);
    const bool bad = (int)((__float_as_uint(x) - 0x0d800000u) > (0x71800000u - 0x0d800000u)) | (int)root_needs_slow_rcp(s);
    if (__builtin_expect(RMDF_LANES_HERE(bad) != 0ull, 0)) { if (bad) y = 1.0f / sqrtf(x); }
    return y;
}
__device__ __forceinline__ v3 normalize3(v3 a)
{
    float s = rsqrt_ieee(dot3(a, a));
    return mk3(a.x * s, a.y * s, a.z * s);
}
__device__ __forceinline__ v3 sub3(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ v3 add3(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ v3 reflect3(v3 i, v3 n)
{
    float k = 2.0f * dot3(n, i);
    return mk3(i.x - k * n.x, i.y - k * n.y, i.z - k * n.z);
}

// log(x), fixed operation order; < 1 ulp.  log_core: x positive and normal, k0 = exponent offset.
__device__ __forceinline__ float log_core(float x, int32_t k0)
{
    const float ln2_hi = 6.9313812256e-01f, ln2_lo = 9.0580006145e-06f;
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    int32_t ix = __float_as_int(x);
    int32_t k = k0 + ((ix >> 23) - 127);
    ix &= 0x007fffff;
    int32_t i = (ix + 0x4afb20) & 0x800000;
    x = __int_as_float(ix | (i ^ 0x3f800000));
    k += (i >> 23);
    float f = x - 1.0f;
    float s = div_known_range(f, 2.0f + f);   // f in [-0.293, 0.415], 2+f in [1.70, 2.42]: the IEEE quotient
    float dk = (float)k;
    float z = s * s;
    float w = z * z;
    float t1 = w * (Lg2 + w * Lg4);
    float t2 = z * (Lg1 + w * Lg3);
    float R = t2 + t1;
    float hfsq = (0.5f * f) * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}
// zero, negative, subnormal, inf, NaN
// (inlined: a call inside k_render made every value that lives across it occupy a callee-saved register)
__device__ __forceinline__ float log_special(float x)
{
    int32_t ix = __float_as_int(x);
    if ((ix & 0x7fffffff) == 0) return -__builtin_inff();
    if (ix < 0) return __builtin_nanf("");
    if (ix >= 0x7f800000) return x + x;
    return log_core(x * 33554432.0f, -25);    // subnormal: scale by 2^25
}
// The core runs unconditionally; lanes outside the positive normal range are recomputed on a wave-uniform branch.
__device__ __forceinline__ float log_pinned(float x)
{
    const bool special = (__float_as_uint(x) - 0x00800000u) >= (0x7f800000u - 0x00800000u);
    float r = log_core(x, 0);
    if (__builtin_expect(RMDF_LANES_HERE(special) != 0ull, 0)) { if (special) r = log_special(x); }
    return r;
}

// Round 3: the divisions inside the straight-line pinned functions (exp, acos, atan) have operands of known range, so they take the
// 5-instruction Markstein quotient (div_known_range) instead of the compiler's ~10-instruction IEEE expansion; the branchy *_full
// forms keep the compiler's division, and rmdf_selftest_pinned_math compares the two for ALL 2^32 inputs of each function (zero
// mismatches is the proof that the quotients agree wherever these functions evaluate them).  RMDF_AB_IEEE_DIV restores the old form.
#ifdef RMDF_AB_IEEE_DIV
#define RMDF_FAST_DIV(a, b) ((a) / (b))
#define RMDF_SHADE_FAST false
#else
#define RMDF_FAST_DIV(a, b) div_known_range((a), (b))
#define RMDF_SHADE_FAST true
#endif
// exp(x), fixed operation order; x < -87 -> 0, x > 88.5 -> inf.  exp_core: -87 <= x <= 88.5.
template <bool FAST_DIV>
__device__ __forceinline__ float exp_core_t(float x)
{
    const float ln2_hi = 6.9314575195e-01f, ln2_lo = 1.4286067653e-06f, invln2 = 1.4426950216e+00f;
    const float P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    float kf = x * invln2 + ((x < 0.0f) ? -0.5f : 0.5f);
    int32_t k = (int32_t)kf;
    float t = (float)k;
    float hi = x - t * ln2_hi;
    float lo = t * ln2_lo;
    float r = hi - lo;
    float tt = r * r;
    float c = r - tt * (P1 + tt * P2);
    const float num = r * c, den = 2.0f - c;                              // den in [1.6, 2.4], |num| <= 0.13 or 0
    float y = 1.0f - ((lo - (FAST_DIV ? RMDF_FAST_DIV(num, den) : num / den)) - hi);
    // y * 2^k as the oracle writes it: two multiplications by 2^(k/2) and 2^(k - k/2).  For -87 <= x <= 88.5 the result is a NORMAL number
    // (exp(-87) = 1.6e-38 > FLT_MIN, exp(88.5) < FLT_MAX) and y in [0.7, 1.5], so neither product rounds and one v_ldexp_f32 returns the
    // same bits (round 5: nine instructions fewer per exp; the test scene evaluates ten per estimate).  The straight-line form takes the
    // ldexp, the branchy reference keeps the two products, and rmdf_selftest_pinned_math compares the two for ALL 2^32 inputs.
    if (FAST_DIV) return __builtin_ldexpf(y, k);
    int32_t k1 = k / 2, k2 = k - k1;
    y = y * __int_as_float((k1 + 127) << 23);
    y = y * __int_as_float((k2 + 127) << 23);
    return y;
}
__device__ __forceinline__ float exp_core(float x) { return exp_core_t<true>(x); }
// the branchy form (reference of the device self-test; the compiler's division)
__device__ __noinline__ float exp_full(float x)
{
    if (x != x) return x;
    if (x > 88.5f) return __builtin_inff();
    if (x < -87.0f) return 0.0f;
    return exp_core_t<false>(x);
}
// Straight-line core for every lane; NaN / overflow / underflow lanes are patched on a wave-uniform branch.
__device__ __forceinline__ float exp_pinned(float x)
{
    const bool special = !((x >= -87.0f) && (x <= 88.5f));
    float y = exp_core(special ? 0.0f : x);
    if (__builtin_expect(RMDF_LANES_HERE(special) != 0ull, 0)) {
        if (special) y = (x != x) ? x : ((x > 88.5f) ? __builtin_inff() : 0.0f);
    }
    return y;
}

// pow(x,y) = exp(y*log(x)); x <= 0 or NaN -> 0 (GLSL leaves it undefined)
__device__ __forceinline__ float pow_pinned(float x, float y)
{
    const bool bad = !(x > 0.0f);
    const float r = exp_pinned(y * log_pinned(bad ? 1.0f : x));
    return bad ? 0.0f : r;
}

// ---- distance estimators ---------------------------------------------------------

// fragment.shd:74-99
__device__ __forceinline__ v3 triplex_pow8(v3 w)
{
    float x = w.x; float x2 = x * x; float x4 = x2 * x2;
    float y = w.y; float y2 = y * y; float y4 = y2 * y2;
    float z = w.z; float z2 = z * z; float z4 = z2 * z2;

    float k3 = y2 + x2;
    float k2 = rsqrt_ieee(k3 * k3 * k3 * k3 * k3 * k3 * k3);
    float k1 = y4 + z4 + x4 - 6.0f * z2 * x2 - 6.0f * y2 * z2 + 2.0f * x2 * y2;
    float k4 = y2 - z2 + x2;

    return mk3(-8.0f * z * k4 * (y4 * y4 - 28.0f * y4 * y2 * x2 + 70.0f * y4 * x4 - 28.0f * y2 * x2 * x4 + x4 * x4) * k1 * k2,
               64.0f * y * z * x * (y2 - x2) * k4 * (y4 - 6.0f * y2 * x2 + x4) * k1 * k2,
               -16.0f * z2 * k3 * k4 * k4 + k1 * k1);
}

// a / dr at the end of a Mandelbulb estimate (fragment.shd:157): dr >= 1 by construction (dr = 8 r^7 dr + 1).  With
// y = RN(1/dr) (rcp_core: exact for every input in its range), q0 = RN(a*y) and q = RN(q0 + (a - dr*q0)*y) is the correctly
// rounded quotient (Markstein) as long as nothing underflows: guarded by dr <= 2^60 and |q| >= 2^-100 (then |a| >= 2^-100 and the
// exact remainder a - dr*q0 is a normal number); everything else -- NaN, inf, a = 0, tiny quotients -- takes the compiler's
// division on a wave-uniform branch.  8 instructions instead of the 14 of hipcc's expansion; sampled against it over 2^32
// operand pairs by rmdf_selftest_exact_math.
__device__ __forceinline__ float div_by_dr(float a, float dr)
{
    const float y = rcp_core(dr);
    const float q0 = a * y;
    float q = __builtin_fmaf(__builtin_fmaf(-dr, q0, a), y, q0);
    const bool bad = (int)!(dr <= 0x1p60f) | (int)!(fabsf(q) >= 0x1p-100f);
    if (__builtin_expect(RMDF_LANES_HERE(bad) != 0ull, 0)) { if (bad) q = a / dr; }
    return q;
}

// fragment.shd:101-158 (POWER8); iters counts the iterations that ran triplex_pow8.
// The loop is the shader's, statement for statement and rounding for rounding; what differs is how three of its operations are
// evaluated, each with the bits of the written one (rmdf_selftest_exact_math checks all three against the written forms for all
// 2^32 inputs on the device):
//  * `r > bailout` (bailout = 4) is decided on d = dot(w,w): RN(sqrt d) > 4  <=>  d > 16 + 2^-19 (the correctly rounded root is
//    monotone; the root of 16 + 2^-19 lies below the midpoint of 4 and its successor, that of 16 + 2^-18 above) -- the root
//    leaves the branch's dependency chain, and a lane that escapes takes its root once, after the loop;
//  * inside the loop d lies in [k3, 16 + 2^-19] and k3^7 in [2^-98, 2^29] once k3 >= 2^-14 is known, so both roots run the bare
//    core sequences and ONE guard per iteration (k3 < 2^-14, or the root of k3^7 ends in sixteen one-bits: rcp_of_root) sends
//    lanes through the compiler's expansions instead of a range test per root;
//  * triplex_pow8 is inlined so that its x^2 + y^2 is shared with the dot product (the same two products, one addition);
//  * the five scalings by a power of two ride on FMAs at the end of their chains, behind an underflow guard (mb8_iterate_t).
#define RMDF_MB8_D4   16.000001907348633f      /* 16 + 2^-19 = 0x41800001 */
static_assert(shk::mb_bailout == 4.0f && shk::mb_iterations_i == 25, "RMDF_MB8_D4 (the bailout test on the squared radius) is derived for bailout = 4");
#define RMDF_MB8_K3MIN 0x1p-14f
// The two roots of one Mandelbulb iteration that did not escape: r = RN(sqrt d), k2 = RN(1 / RN(sqrt q)) with q = k3^7,
// k3 <= d <= 16 + 2^-19.  One guard for both (see de_mandelbulb8).
__device__ __forceinline__ void mb8_roots(float d, float k3, float q, float &r, float &k2)
{
    r = sqrt_core(d);
    float y0;
    const float sq = sqrt_core_y(q, y0);
    k2 = rcp_of_root(sq, y0);
    const bool small = k3 < RMDF_MB8_K3MIN;
    if (__builtin_expect((RMDF_LANES_HERE(small) | root_needs_slow_rcp_lanes(sq)) != 0ull, 0)) {
        if ((int)small | (int)root_needs_slow_rcp(sq)) { r = sqrtf(d); k2 = 1.0f / sqrtf(q); }
    }
}
// Iterations i0 .. i1-1 of the loop (i1 <= 25) on the state (w, dr, r, d); a lane whose squared radius d exceeds the bailout
// leaves early.  Afterwards the estimate is complete iff d > RMDF_MB8_D4 (escaped) or i1 == 25; otherwise the same function
// resumes it from i1 -- on any lane: the distance-AO estimates of k_render are finished that way (rmdf_render.hip).
//
// FOLD = false is the shader's arithmetic operation for operation.  FOLD = true pulls the five scalings by a power of two to the
// end of their chains, where an FMA takes them for nothing:
//     -8 z k4 P8 k1 k2 + pos.x  ->  fma(-8, A, pos.x),  A = ((((z k4) P8) k1) k2)          64 y z x (y2-x2) k4 P4 k1 k2 + pos.y  ->  fma(64, B, pos.y)
//     -16 z2 k3 k4 k4 + k1 k1   ->  fma(-16, C, k1 k1), C = ((z2 k3) k4) k4                  ... + 2 x2 y2  ->  fma(2, t, ...), t = x2 y2
//     r7 8 dr + 1               ->  fma(8, r7 dr, 1)
// Scaling by a power of two commutes with rounding unless a product underflows (or overflows: excluded by k3 >= 2^-14, d <= 16), so
// the folded pass has the bits of the written one whenever every partial product of the four chains is a normal number.  That is
// implied by |A|, |B|, C, t >= 2^-40: the factors still to come are bounded (|k2| <= 2^49, |k1| <= 2^13, |P8| <= k3^4 <= 2^16,
// |P4| <= 2^9, |k4| <= 2^5, |y2 - x2| <= 2^4, |x|, |z| <= 2^2.01), which puts the smallest partial product of the longest chain
// at >= 2^-122.  `m` accumulates the minimum of those four over the passes (two instructions per pass); the CALLER compares it
// with RMDF_MB8_FOLD_MIN once per call and runs the call again with FOLD = false from the same state if any lane fell below
// (it does happen: a few thousand estimates of the headline frame -- points within 2^-7 of the bulb's axis, where k3 < 2^-14 forces
// it through m = 0 because r^7 dr may underflow too, and rays that cross a coordinate plane within ~2^-20).
#define RMDF_MB8_FOLD_MIN 0x1p-40f
template <bool FOLD>
__device__ __forceinline__ void mb8_iterate_t(v3 &w, const v3 pos, float &dr, float &r, float &d, int i0, int i1, unsigned &iters, float &m)
{
    for (int i = i0; i < i1; i++) {
        // r = length(w); if (r > bailout) break;                                    fragment.shd:137-139
        const float x = w.x, y = w.y, z = w.z;
        const float x2 = x * x, y2 = y * y, z2 = z * z;
        const float k3 = y2 + x2;                       // = x*x + y*y of the dot product (addition commutes bit for bit)
        d = k3 + z2;
        if (d > RMDF_MB8_D4) break;
        // w = triplex_pow8(w)                                                       fragment.shd:74-99
        const float x4 = x2 * x2, y4 = y2 * y2, z4 = z2 * z2;
        const float q = k3 * k3 * k3 * k3 * k3 * k3 * k3;
        const float k4 = y2 - z2 + x2;
        const float p8 = y4 * y4 - 28.0f * y4 * y2 * x2 + 70.0f * y4 * x4 - 28.0f * y2 * x2 * x4 + x4 * x4;
        const float p4 = y4 - 6.0f * y2 * x2 + x4;
        if (!FOLD) {
            const float k1 = y4 + z4 + x4 - 6.0f * z2 * x2 - 6.0f * y2 * z2 + 2.0f * x2 * y2;
            const float wx_ = -8.0f * z * k4 * p8 * k1;
            const float wy_ = 64.0f * y * z * x * (y2 - x2) * k4 * p4 * k1;
            const float wz = -16.0f * z2 * k3 * k4 * k4 + k1 * k1;
            float k2;
            mb8_roots(d, k3, q, r, k2);
            // dr = pow(r, power - 1) * power * dr + 1                               fragment.shd:148
            const float r2 = r * r, r4 = r2 * r2, r7 = (r4 * r2) * r;
            dr = r7 * 8.0f * dr + 1.0f;
            w = add3(mk3(wx_ * k2, wy_ * k2, wz), pos);
        } else {
            const float t = x2 * y2;
            const float k1 = __builtin_fmaf(2.0f, t, y4 + z4 + x4 - 6.0f * z2 * x2 - 6.0f * y2 * z2);
            const float a_ = z * k4 * p8 * k1;
            const float b_ = y * z * x * (y2 - x2) * k4 * p4 * k1;
            const float c = z2 * k3 * k4 * k4;
            const float wz = __builtin_fmaf(-16.0f, c, k1 * k1);
            // the two roots, as in mb8_roots; a lane with k3 < 2^-14 also gives up the folds (m = 0)
            r = sqrt_core(d);
            float y0;
            const float sq = sqrt_core_y(q, y0);
            float k2 = rcp_of_root(sq, y0);
            const bool small = k3 < RMDF_MB8_K3MIN;
            if (__builtin_expect((RMDF_LANES_HERE(small) | root_needs_slow_rcp_lanes(sq)) != 0ull, 0)) {
                if ((int)small | (int)root_needs_slow_rcp(sq)) { r = sqrtf(d); k2 = 1.0f / sqrtf(q); if (small) m = 0.0f; }
            }
            const float a = a_ * k2, b = b_ * k2;
            m = __builtin_fminf(__builtin_fminf(m, __builtin_fabsf(a)), __builtin_fabsf(b));     // v_min3_f32 with |.| modifiers
            m = __builtin_fminf(__builtin_fminf(m, c), t);                                       // c, t >= 0
            const float r2 = r * r, r4 = r2 * r2, r7 = (r4 * r2) * r;
            dr = __builtin_fmaf(8.0f, r7 * dr, 1.0f);
            w = mk3(__builtin_fmaf(-8.0f, a, pos.x), __builtin_fmaf(64.0f, b, pos.y), wz + pos.z);
        }
        iters++;
        RMDF_EMU_PASS();
    }
    RMDF_EMU_SEGMENT_END(i0);
}
// does the folded call have to be run again in written form (any lane of the wave decides for itself; the branch is wave-uniform)
__device__ __forceinline__ bool mb8_fold_failed(float m, float fold_min = RMDF_MB8_FOLD_MIN) { return !(m >= fold_min); }
// fragment.shd:157 -- the lanes that left through the break take their root now (d keeps its last value)
__device__ __forceinline__ float mb8_finish(float dr, float r, float d)
{
    if (d > RMDF_MB8_D4) r = sqrt_rn(d);
    return div_by_dr(0.5f * log_pinned(r) * r, dr);
}
// the written loop (the fall-back of the folded one; de_mandelbulb8_written = the reference of the device self-test)
__device__ __forceinline__ float de_mandelbulb8_written_inl(v3 pos, unsigned &iters)
{
    pos = mk3(pos.z, pos.x, pos.y);
    v3 w = pos;
    float dr = 1.0f, r = 0.0f, d = 0.0f, m = 1.0f;
    mb8_iterate_t<false>(w, pos, dr, r, d, 0, shk::mb_iterations_i, iters, m);
    return mb8_finish(dr, r, d);
}
__device__ __noinline__ float de_mandelbulb8_written(v3 pos, unsigned &iters) { return de_mandelbulb8_written_inl(pos, iters); }
// The fall-back is inlined (round 2 called it: every value of k_render that lives across the call site then had to sit in a
// callee-saved register, and two VGPRs went to scratch).  It starts from a LAUNDERED copy of the position: otherwise the compiler
// shares the first pass's squares and sums with the folded loop, keeps them alive across it, and pays five register copies per
// estimate in the hot path for a block that runs a few hundred times per frame (measured: 2 %).
__device__ __forceinline__ float de_mandelbulb8(v3 pos, unsigned &iters, float fold_min = RMDF_MB8_FOLD_MIN, unsigned *n_redone = nullptr)
{
    const v3 p = mk3(pos.z, pos.x, pos.y);
    v3 w = p;
    float dr = 1.0f, r = 0.0f, d = 0.0f, m = 1.0f;
    unsigned n = 0u;
    mb8_iterate_t<true>(w, p, dr, r, d, 0, shk::mb_iterations_i, n, m);
    float dist = mb8_finish(dr, r, d);
    const bool redo = mb8_fold_failed(m, fold_min);
    if (__builtin_expect(RMDF_LANES_HERE(redo) != 0ull, 0)) {
        if (redo) {
            v3 q = pos;
#ifndef RMDF_HOST_EMULATION
            asm volatile("" : "+v"(q.x), "+v"(q.y), "+v"(q.z));
#endif
            n = 0u;
            dist = de_mandelbulb8_written_inl(q, n);
            if (n_redone) ++*n_redone;
        }
    }
    iters += n;
    return dist;
}

// ---- pinned sin / cos / acos / atan / mod (FSMBGeneralShader, FSDETestShader) ------------------------
// fdlibm-style float algorithms, fixed operation order, identical to the oracle's restatement
__device__ __forceinline__ void rem_pio2_pinned(float x, float &r, int &q)
{
    const float invpio2 = 0.6366197466850281f;
    const float c1 = 1.5703125f, c2 = 4.837512969970703e-4f, c3 = 7.549533620476723e-8f, c4 = 2.5633440682570896e-12f;
    const float kf = rintf(x * invpio2);
    q = (int)kf;
    float t = x - kf * c1;
    t = t - kf * c2;
    t = t - kf * c3;
    t = t - kf * c4;
    r = t;
}
__device__ __forceinline__ float ksin_pinned(float x)
{
    const float S1 = -1.6666667163e-01f, S2 = 8.3333337680e-03f, S3 = -1.9841270114e-04f, S4 = 2.7557314297e-06f,
                S5 = -2.5050759689e-08f, S6 = 1.5896910177e-10f;
    const float z = x * x;
    const float v = z * x;
    const float r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x + v * (S1 + z * r);
}
__device__ __forceinline__ float kcos_pinned(float x)
{
    const float C1 = 4.1666667908e-02f, C2 = -1.3888889225e-03f, C3 = 2.4801587642e-05f, C4 = -2.7557314297e-07f,
                C5 = 2.0875723372e-09f, C6 = -1.1359647598e-11f;
    const float z = x * x;
    const float r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    return 1.0f - (0.5f * z - z * r);
}
// the branchy form (reference of the device self-test)
__device__ __noinline__ void sincos_full(float x, float &s, float &c)
{
    if (!(fabsf(x) <= 3.4e38f)) { s = x - x; c = x - x; return; }
    float r; int q;
    rem_pio2_pinned(x, r, q);
    const float ks = ksin_pinned(r), kc = kcos_pinned(r);
    switch (q & 3) {
    case 0:  s = ks;  c = kc;  break;
    case 1:  s = kc;  c = -ks; break;
    case 2:  s = -ks; c = -kc; break;
    default: s = -kc; c = ks;  break;
    }
}
// sin and cos of the same argument share the reduction (the shader always needs both).  Branch-free: the quadrant
// logic is two selects and two sign flips, inf / NaN are overridden at the end.
__device__ __forceinline__ void sincos_pinned(float x, float &s, float &c)
{
    const bool bad = !(fabsf(x) <= 3.4e38f);
    float r; int q;
    rem_pio2_pinned(bad ? 0.0f : x, r, q);
    const float ks = ksin_pinned(r), kc = kcos_pinned(r);
    const bool swap = (q & 1) != 0;
    const float s0 = swap ? kc : ks, c0 = swap ? ks : kc;
    s = (q & 2) ? -s0 : s0;                 // q mod 4 = 0: ( ks,  kc)  1: ( kc, -ks)  2: (-ks, -kc)  3: (-kc,  ks)
    c = ((q + 1) & 2) ? -c0 : c0;
    if (bad) { s = x - x; c = x - x; }
}
// (the branchy forms are inlined where the straight-line functions fall back on them -- a CALL inside k_render costs a private segment
// and pins every value that lives across it to a callee-saved register -- and exist as functions of their own for the device self-test)
__device__ __forceinline__ float acos_full_inl(float x)
{
    const float pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f, pi = 3.1415925026e+00f;
    const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
                pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f;
    const float qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const float ax = fabsf(x);
    if (!(ax <= 1.0f)) return (x - x) / (x - x);
    if (ax == 1.0f) return (x > 0.0f) ? 0.0f : pi + 2.0f * pio2_lo;
    if (ax < 0.5f) {
        if (ax <= 1.4901161e-8f) return pio2_hi + pio2_lo;
        const float z = x * x;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (x < 0.0f) {
        const float z = (1.0f + x) * 0.5f;
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float s = sqrt_rn(z);
        const float r = p / q;
        const float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    } else {
        const float z = (1.0f - x) * 0.5f;
        const float s = sqrt_rn(z);
        const float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);
        const float c = (z - df * df) / (s + df);
        const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        const float r = p / q;
        const float w = r * s + c;
        return 2.0f * (df + w);
    }
}
__device__ __noinline__ float acos_full(float x) { return acos_full_inl(x); }
// Branch-free main path for 2^-26 < |x| < 1 (one p/q, one sqrt, one division for the upper-range correction, results
// selected); |x| >= 1, tiny |x| and NaN go through acos_full on a wave-uniform branch.  Lanes of one wave span the whole
// [-1, 1] in the general-power Mandelbulb, so the three-way branch of acos_full used to run all its arms.
__device__ __forceinline__ float acos_pinned(float x)
{
    const float pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f, pi = 3.1415925026e+00f;
    const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
                pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f;
    const float qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    const float ax = fabsf(x);
    const bool special = !((ax < 1.0f) && (ax > 1.4901161e-8f));
    const bool small = ax < 0.5f;
    // (1 + x) * 0.5 for x < 0 and (1 - x) * 0.5 for x > 0 are the same operation on |x|
    const float z = small ? x * x : (1.0f - ax) * 0.5f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = RMDF_FAST_DIV(p, q);                     // z in [0, 0.5]: q in [0.3, 1], p in [0, 0.1]
    const float s = sqrt_rn(z);
    const float r_small = pio2_hi - (x - (pio2_lo - x * r));
    const float wn = r * s - pio2_lo;
    const float r_neg = pi - 2.0f * (s + wn);
    const float df = __uint_as_float(__float_as_uint(s) & 0xfffff000u);
    const float c = RMDF_FAST_DIV(z - df * df, s + df);
    const float wp = r * s + c;
    const float r_pos = 2.0f * (df + wp);
    float res = small ? r_small : ((x < 0.0f) ? r_neg : r_pos);
    if (__builtin_expect(RMDF_LANES_HERE(special) != 0ull, 0)) { if (special) res = acos_full_inl(x); }
    return res;
}
__device__ __forceinline__ float atan_full_inl(float x)
{
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    if (x != x) return x;
    const float ax = fabsf(x);
    const bool neg = (__float_as_uint(x) >> 31) != 0u;
    float hi = 0.0f, lo = 0.0f, t;
    bool small = false;
    if (ax >= 67108864.0f) {
        const float z = 1.5707962513e+00f + 7.5497894159e-08f;
        return neg ? -z : z;
    }
    if (ax < 0.4375f) {
        if (ax < 2.44140625e-4f) return x;
        small = true; t = x;
    } else if (ax < 1.1875f) {
        if (ax < 0.6875f) { hi = 4.6364760399e-01f; lo = 5.0121582440e-09f; t = (2.0f * ax - 1.0f) / (2.0f + ax); }
        else              { hi = 7.8539812565e-01f; lo = 3.7748947079e-08f; t = (ax - 1.0f) / (ax + 1.0f); }
    } else {
        if (ax < 2.4375f) { hi = 9.8279368877e-01f; lo = 3.4473217170e-08f; t = (ax - 1.5f) / (1.0f + 1.5f * ax); }
        else              { hi = 1.5707962513e+00f; lo = 7.5497894159e-08f; t = -1.0f / ax; }
    }
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    if (small) return t - t * (s1 + s2);
    const float zz = hi - ((t * (s1 + s2) - lo) - t);
    return neg ? -zz : zz;
}
__device__ __noinline__ float atan_full(float x) { return atan_full_inl(x); }
// Branch-free main path for 2^-12 <= |x| < 2^26: the argument reduction picks numerator and denominator by selects
// and divides once (the four reduced ranges of atan_full each carried their own division); everything else (tiny, huge,
// NaN) goes through atan_full on a wave-uniform branch.
__device__ __forceinline__ float atan_pinned(float x)
{
    const float aT0 = 3.3333334327e-01f, aT1 = -2.0000000298e-01f, aT2 = 1.4285714924e-01f, aT3 = -1.1111110449e-01f,
                aT4 = 9.0908870101e-02f, aT5 = -7.6918758452e-02f, aT6 = 6.6610731184e-02f, aT7 = -5.8335702866e-02f,
                aT8 = 4.9768779427e-02f, aT9 = -3.6531571299e-02f, aT10 = 1.6285819933e-02f;
    const float ax = fabsf(x);
    const bool special = !((ax >= 2.44140625e-4f) && (ax < 67108864.0f));
    const bool neg = (__float_as_uint(x) >> 31) != 0u;
    const bool small = ax < 0.4375f;
    const bool r1 = ax < 0.6875f, r2 = ax < 1.1875f, r3 = ax < 2.4375f;
    // ranges: [0.4375, 0.6875) (2ax - 1)/(2 + ax); [0.6875, 1.1875) (ax - 1)/(ax + 1); [1.1875, 2.4375) (ax - 1.5)/(1 + 1.5 ax);
    // above: -1/ax.  small: t = x, no division (the quotient below is computed and dropped)
    const float num = r1 ? (2.0f * ax - 1.0f) : (r2 ? (ax - 1.0f) : (r3 ? (ax - 1.5f) : -1.0f));
    const float den = r1 ? (2.0f + ax) : (r2 ? (ax + 1.0f) : (r3 ? (1.0f + 1.5f * ax) : ax));
    const float hi = r1 ? 4.6364760399e-01f : (r2 ? 7.8539812565e-01f : (r3 ? 9.8279368877e-01f : 1.5707962513e+00f));
    const float lo = r1 ? 5.0121582440e-09f : (r2 ? 3.7748947079e-08f : (r3 ? 3.4473217170e-08f : 7.5497894159e-08f));
    const float quo = RMDF_FAST_DIV(num, den);               // |num| <= 1, den in [1.4, 2^26]
    const float t = small ? x : quo;
    const float z = t * t;
    const float w = z * z;
    const float s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
    const float s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
    const float r_small = t - t * (s1 + s2);
    const float zz = hi - ((t * (s1 + s2) - lo) - t);
    float res = small ? r_small : (neg ? -zz : zz);
    if (__builtin_expect(RMDF_LANES_HERE(special) != 0ull, 0)) { if (special) res = atan_full_inl(x); }
    return res;
}
// GLSL atan(y, x), every special case
__device__ __forceinline__ float atan2_full_inl(float y, float x)
{
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f, pio2 = 1.5707963705e+00f;
    const float inf = __builtin_inff();
    if (x != x || y != y) return x + y;
    const int m = (int)((__float_as_uint(y) >> 31) | ((__float_as_uint(x) >> 30) & 2u));
    const float ax = fabsf(x), ay = fabsf(y);
    if (ay == 0.0f) return (m < 2) ? y : ((m == 2) ? pi : -pi);
    if (ax == 0.0f) return (m & 1) ? -pio2 : pio2;
    if (ax == inf) {
        if (ay == inf) return (m == 0) ? 0.25f * pi : (m == 1) ? -0.25f * pi : (m == 2) ? 0.75f * pi : -0.75f * pi;
        return (m == 0) ? 0.0f : (m == 1) ? -0.0f : (m == 2) ? pi : -pi;
    }
    if (ay == inf) return (m & 1) ? -pio2 : pio2;
    const float z = atan_pinned(ay / ax);
    return (m == 0) ? z : (m == 1) ? -z : (m == 2) ? pi - (z - pi_lo) : (z - pi_lo) - pi;
}

__device__ __noinline__ float atan2_full(float y, float x) { return atan2_full_inl(y, x); }

// GLSL atan(y, x): finite non-zero operands take the straight-line path; zeros, infinities and NaN go through atan2_full.
// (Its division has operands of any magnitude; widening the range test to [2^-40, 2^40) so that it could take the short quotient
// too was measured: 1.915 -> 1.922 ms for the general-power frame, not adopted.)
__device__ __forceinline__ float atan2_pinned(float y, float x)
{
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
    const float ax = fabsf(x), ay = fabsf(y);
    const bool special = !((ax > 0.0f) && (ax < __builtin_inff()) && (ay > 0.0f) && (ay < __builtin_inff()));
    const int m = (int)((__float_as_uint(y) >> 31) | ((__float_as_uint(x) >> 30) & 2u));
    const float z = atan_pinned(special ? 1.0f : ay / ax);
    float res = (m == 0) ? z : (m == 1) ? -z : (m == 2) ? pi - (z - pi_lo) : (z - pi_lo) - pi;
    if (__builtin_expect(RMDF_LANES_HERE(special) != 0ull, 0)) { if (special) res = atan2_full_inl(y, x); }
    return res;
}

// pow(x, y) = exp(y * log(x)) with the logarithm supplied: lx = log_pinned(x > 0 ? x : 1) (pow_pinned's own first step).  The general-power
// iteration raises the same r to `power` and to `power - 1` (fragment.shd:64,148): one logarithm serves both -- the same operations on
// the same operands, evaluated once.
__device__ __forceinline__ float pow_with_log(float x, float lx, float y)
{
    const float r = exp_pinned(y * lx);
    return !(x > 0.0f) ? 0.0f : r;
}

// fragment.shd:42-72; r = length(w) and lr = log_pinned(r > 0 ? r : 1) come from the caller (de_mandelbulb_general has both)
__device__ __forceinline__ v3 triplex_pow_general(v3 w, float power, float r, float lr)
{
    float theta = acos_pinned(w.z / r);
    float phi = atan2_pinned(w.y, w.x);
    const float zr = pow_with_log(r, lr, power);
    theta = theta * power;
    phi = phi * power;
    float st, ct, sp, cp;
    sincos_pinned(theta, st, ct);
    sincos_pinned(phi, sp, cp);
    return mk3(zr * (st * cp), zr * (st * sp), zr * ct);
}

// fragment.shd:101-158 without POWER8; power = fragment.shd:116-119, uniform per frame
__device__ __forceinline__ float de_mandelbulb_general(v3 pos, float power, unsigned &iters)
{
    pos = mk3(pos.z, pos.x, pos.y);
    v3 w = pos;
    float dr = 1.0f;
    float r = 0.0f;
    for (int i = 0; i < 25; i++) {
        r = length3(w);
        if (r > 4.0f) break;
        const float lr = log_pinned(!(r > 0.0f) ? 1.0f : r);
        w = triplex_pow_general(w, power, r, lr);
        w = add3(w, pos);
        dr = pow_with_log(r, lr, power - 1.0f) * power * dr + 1.0f;
        iters++;
    }
    return div_by_dr(0.5f * log_pinned(r) * r, dr);
}

// fragment.shd:21-33, 413-418, 447-456
__device__ __forceinline__ float length2_(float x, float y) { return sqrt_rn(x * x + y * y); }
__device__ __forceinline__ float de_torus(v3 p, float size, float r) { return length2_(length2_(p.x, p.y) - size, p.z) - r; }
__device__ __forceinline__ float de_rounded_box(v3 p, v3 b, float r)
{
    return length3(mk3(gmax(fabsf(p.x) - b.x, 0.0f), gmax(fabsf(p.y) - b.y, 0.0f), gmax(fabsf(p.z) - b.z, 0.0f))) - r;
}
__device__ __forceinline__ float smin_exp(float a, float b, float k)
{
    const float res = exp_pinned(-k * a) + exp_pinned(-k * b);
    return -log_pinned(res) / k;
}
__device__ __forceinline__ float de_test_scene(v3 pos)
{
    const float d_sphere = length3(pos) - 0.4f;
    const float d_torus = smin_exp(smin_exp(de_torus(pos, 0.85f, 0.1f), de_torus(mk3(pos.z, pos.x, pos.y), 0.85f, 0.1f), 64.0f),
                                   de_torus(mk3(pos.y, pos.z, pos.x), 0.85f, 0.1f), 64.0f);
    const float d_box = smin_exp(smin_exp(de_rounded_box(pos, mk3(0.8f, 0.06f, 0.06f), 0.03f),
                                          de_rounded_box(pos, mk3(0.06f, 0.8f, 0.06f), 0.03f), 64.0f),
                                 de_rounded_box(pos, mk3(0.06f, 0.06f, 0.8f), 0.03f), 64.0f);
    return smin_exp(d_box, gmin(d_sphere, d_torus), 64.0f);
}

// fragment.shd:312-321
__device__ __forceinline__ float line_seg_min_dist_sq(v3 a, v3 b, v3 p)
{
    v3 ab = sub3(b, a);
    float len_sq = dot3(ab, ab);
    float t = dot3(sub3(p, a), ab) / len_sq;
    t = gclamp(t, 0.0f, 1.0f);
    v3 proj = mk3(a.x + t * ab.x, a.y + t * ab.y, a.z + t * ab.z);
    v3 d = sub3(p, proj);
    return dot3(d, d);
}

// fragment.shd:348-372 (compute_barycentric 323-346 inlined)
__device__ __forceinline__ float de_triangle(v3 pos, v3 v0, v3 v1, v3 v2)
{
    v3 e0 = sub3(v2, v0);
    v3 e1 = sub3(v1, v0);
    v3 e2 = sub3(pos, v0);
    float dot00 = dot3(e0, e0);
    float dot01 = dot3(e0, e1);
    float dot02 = dot3(e0, e2);
    float dot11 = dot3(e1, e1);
    float dot12 = dot3(e1, e2);
    float inv_denom = 1.0f / (dot00 * dot11 - dot01 * dot01);
    float u = (dot11 * dot02 - dot01 * dot12) * inv_denom;
    float v = (dot00 * dot12 - dot01 * dot02) * inv_denom;
    if ((u >= 0.0f) && (v >= 0.0f) && (u + v < 1.0f)) {
        float k = 1.0f - (u + v);
        v3 pp = mk3(v2.x * u + v1.x * v + v0.x * k,
                    v2.y * u + v1.y * v + v0.y * k,
                    v2.z * u + v1.z * v + v0.z * k);
        return length3(sub3(pos, pp));
    }
    return sqrtf(gmin(line_seg_min_dist_sq(v0, v1, pos),
                      gmin(line_seg_min_dist_sq(v0, v2, pos), line_seg_min_dist_sq(v1, v2, pos))));
}

// fragment.shd:374-411; tri = 96 vertices (wave-uniform -> scalar loads)
__device__ __forceinline__ float de_cornell_box(v3 pos, const float *__restrict__ tri)
{
    float dist = 999.0f;
    for (int i = 0; i < 32; i++) {
        const float *t = tri + i * 9;
        dist = gmin(dist, de_triangle(pos, mk3(t[0], t[1], t[2]), mk3(t[3], t[4], t[5]), mk3(t[6], t[7], t[8])));
    }
    return dist;
}

// ---- the same distance with everything that depends only on the triangle taken from a table -----------------
// Per triangle (CORNELL_STRIDE floats, computed on the host with the operation order of de_triangle, so every
// entry has the bits the kernel would have computed): v0, v1, v2, e0 = v2-v0, e1 = v1-v0, dot00, dot01, dot11,
// inv_denom, e12 = v2-v1, len12 = dot(e12,e12), and the correctly rounded reciprocals of the three squared edge
// lengths.  Bit-identical to de_cornell_box because
//   * x / len is replaced by q1 = fma(fma(-len, q0, x), y, q0), q0 = x*y, y = RN(1/len): the correctly rounded
//     quotient for these 96 divisors (checked for every numerator bit pattern by rmdf_selftest_exact_math); numerators
//     outside 2^-60..2^60 (zeros included) take the compiler's division;
//   * min over sqrt(x_i) = sqrt(min over x_i): correctly rounded sqrt is monotone, and 999 = sqrt(998001) exactly.
// t[28..43] of a row: the pruning planes of the per-lane estimate (below).  Behind the 32 rows: a compact table of the wave-uniform
// estimate's bounds, 32 x 8 floats (plane, bounding sphere).
#define CORNELL_STRIDE 44
#define CORNELL_BOUNDS 28              /* t[28..43]: the triangle's plane and its three edge planes, four floats each */
#define CORNELL_TAB_FLOATS (32 * CORNELL_STRIDE + 32 * 8)
__device__ __forceinline__ float div_by_table(float x, float len, float rlen)
{
    const float q0 = x * rlen;
    const float r0 = __builtin_fmaf(-len, q0, x);
    float q = __builtin_fmaf(r0, rlen, q0);
    const unsigned ax = __float_as_uint(x) & 0x7fffffffu;
    const bool bad = (ax - 0x21800000u) > (0x5d800000u - 0x21800000u);   // |x| outside 2^-60 .. 2^60 (zeros too: sign of -0/len)
    if (__builtin_expect(RMDF_LANES_HERE(bad) != 0ull, 0)) { if (bad) q = x / len; }
    return q;
}
__device__ __forceinline__ float seg_dist_sq_table(v3 a, v3 ab, float len, float rlen, v3 pa, v3 p)
{
    // line_seg_min_dist_sq (fragment.shd:312-321) with ab, len_sq from the table and pa = p - a
    float t = div_by_table(dot3(pa, ab), len, rlen);
    t = gclamp(t, 0.0f, 1.0f);
    const v3 proj = mk3(a.x + t * ab.x, a.y + t * ab.y, a.z + t * ab.z);
    const v3 d = sub3(p, proj);
    return dot3(d, d);
}
// Pruning (prune != 0).  min() over the 32 triangles is exact and order-independent, so a triangle whose distance
// provably cannot undercut the running minimum may be skipped without changing a bit.  Lower bounds of the
// point-triangle distance come from more table entries per triangle (computed in double on the host): the distance to the
// triangle's plane and, in the wave-uniform estimate, |p - c| - R for a bounding sphere; in the per-lane estimate, since round 4,
// the distances beyond the three planes through the triangle's edges (a point over triangle A of a wall is at least its
// in-plane distance to the diagonal away from the wall's other triangle -- the plane and sphere tests always let that one
// through; the two combine as orthogonal components: 0.50 -> 0.12 full evaluations per estimate beyond the first, 0.32 ->
// 0.003 for the rays grazing the ceiling that were the launch's critical path).  The triangle is skipped when a bound exceeds dmax = 1.001 * sqrt(running minimum) + 1e-5 -- a
// margin four orders of magnitude above the float32 rounding of the bounds and of the reference's formulas (1e-7).
// The test is 13 instructions against ~130 for the distance; a wave skips a triangle when all its lanes do (rays of
// an 8x8 packet are close together): of 32 triangles ~5 survive on average.
//
// Order.  The sooner the running minimum is tight, the more gets skipped, and min() does not care about the order: the
// triangle that was nearest in this lane's previous estimate (`hint`, taken from the wave's first active lane so that the
// table reads stay scalar) is evaluated first, then the rest in table order.
typedef const float __attribute__((address_space(4))) cfloat;     // constant address space: wave-uniform reads become s_loads

// squared distance to triangle row t (de_triangle, fragment.shd:348-372, constants from the table); PTR = the constant-address-space
// pointer of the wave-uniform path (rows by scalar loads) or a plain pointer into an LDS copy (one row per lane: cornell_group_dist2)
template <typename PTR>
__device__ __forceinline__ float cornell_tri_dist2(v3 pos, PTR t)
{
    RMDF_EMU_COST(130);                 // (emulator only: ~130 vector instructions per point-triangle distance, tools/emulated_schedule.py)
    const v3 v0 = mk3(t[0], t[1], t[2]), v1 = mk3(t[3], t[4], t[5]), v2 = mk3(t[6], t[7], t[8]);
    const v3 e0 = mk3(t[9], t[10], t[11]), e1 = mk3(t[12], t[13], t[14]);
    const float dot00 = t[15], dot01 = t[16], dot11 = t[17], inv_denom = t[18];
    const v3 e12 = mk3(t[19], t[20], t[21]);
    const float len12 = t[22], r00 = t[23], r11 = t[24], r12 = t[25];
    const v3 e2 = sub3(pos, v0);
    const float dot02 = dot3(e0, e2), dot12 = dot3(e1, e2);
    const float u = (dot11 * dot02 - dot01 * dot12) * inv_denom;
    const float v = (dot00 * dot12 - dot01 * dot02) * inv_denom;
    if ((u >= 0.0f) && (v >= 0.0f) && (u + v < 1.0f)) {
        const float k = 1.0f - (u + v);
        const v3 pp = mk3(v2.x * u + v1.x * v + v0.x * k, v2.y * u + v1.y * v + v0.y * k, v2.z * u + v1.z * v + v0.z * k);
        const v3 d = sub3(pos, pp);
        return dot3(d, d);
    }
    const float s01 = seg_dist_sq_table(v0, e1, dot11, r11, e2, pos);                 // segment v0 v1: ab = e1
    const float s02 = seg_dist_sq_table(v0, e0, dot00, r00, e2, pos);                 // segment v0 v2: ab = e0
    const float s12 = seg_dist_sq_table(v1, e12, len12, r12, sub3(pos, v1), pos);     // segment v1 v2
    return gmin(s01, gmin(s02, s12));
}

// Candidate grid (round 2).  A host-built grid of CORNELL_GRID_N^3 cells over [-CORNELL_GRID_H, CORNELL_GRID_H]^3 holds, per cell, the
// bit mask of the triangles that can be the nearest one for SOME point of the cell: the point-triangle distance is 1-Lipschitz, so
// triangle t can win only if d(centre, t) <= min_s d(centre, s) + 2 * (half diagonal + margin) (rmdf_api.cpp: cornell_grid, double
// arithmetic).  The kernel keeps the grid in LDS; a wave ORs the masks of its lanes' cells (rays of a packet are close together: one or
// two cells) and the bound tests and row loads below run only for the triangles of that mask -- min() does not care which provably
// losing triangles are left out.  Points outside the grid (and the NO_PRUNE loop) take every triangle.
#define CORNELL_GRID_N 16
#define CORNELL_GRID_H 1.1f
// Round 3: the grid k_render uses has CORNELL_FINE_N^3 cells (1 MB: it stays in global memory, L2-resident; built from the
// CORNELL_GRID_N^3 one by halving cells, rmdf_api.cpp).  The per-lane estimate (de_cornell_box_lanes) reads one dword of it per
// estimate, needed only after the hinted triangle has been evaluated, so its latency hides -- and the finer the cells the fewer
// candidates (one frame at a time: 16^3 in LDS 0.229 ms, 24^3 0.198, 32^3 0.184, 48^3 0.172, 64^3 0.164, 96^3 0.164, 128^3 0.162).
#define CORNELL_FINE_N 64
template <int N>
__device__ __forceinline__ unsigned cornell_cell_mask_n(v3 p, const unsigned *grid)
{
    const float s = (float)N / (2.0f * CORNELL_GRID_H);
    const int ix = (int)floorf((p.x + CORNELL_GRID_H) * s), iy = (int)floorf((p.y + CORNELL_GRID_H) * s), iz = (int)floorf((p.z + CORNELL_GRID_H) * s);
    if ((unsigned)ix >= (unsigned)N || (unsigned)iy >= (unsigned)N || (unsigned)iz >= (unsigned)N) return 0xffffffffu;
    return grid[(iz * N + iy) * N + ix];
}
__device__ __forceinline__ unsigned cornell_cell_mask(v3 p, const unsigned *grid) { return cornell_cell_mask_n<CORNELL_GRID_N>(p, grid); }
// The same cell with one FMA and one floor-and-convert per axis (the per-lane estimate's lookup in the fine grid).  Which cell a point on
// a cell border falls into is not part of any contract: the grid's margin covers the rounding of the index (rmdf_api.cpp: cornell_grid).
template <int N>
__device__ __forceinline__ unsigned cornell_cell_mask_fast(v3 p, const unsigned *grid)
{
    const float s = (float)N / (2.0f * CORNELL_GRID_H), hs = CORNELL_GRID_H * s;
    int ix, iy, iz;
#ifdef RMDF_HOST_EMULATION       // (v_cvt_flr_i32_f32 = floor, then a saturating conversion; NaN -> 0)
    auto flr = [](float f) -> int { const float g = floorf(f); return !(g == g) ? 0 : (g >= 2147483648.0f ? 2147483647 : (g <= -2147483648.0f ? -2147483647 - 1 : (int)g)); };
    ix = flr(__builtin_fmaf(p.x, s, hs)); iy = flr(__builtin_fmaf(p.y, s, hs)); iz = flr(__builtin_fmaf(p.z, s, hs));
#else
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(ix) : "v"(__builtin_fmaf(p.x, s, hs)));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iy) : "v"(__builtin_fmaf(p.y, s, hs)));
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(iz) : "v"(__builtin_fmaf(p.z, s, hs)));
#endif
    if ((unsigned)ix >= (unsigned)N || (unsigned)iy >= (unsigned)N || (unsigned)iz >= (unsigned)N) return 0xffffffffu;
    return grid[(iz * N + iy) * N + ix];
}
// OR of `m` over the active lanes of the wave (exec-safe: reads only lanes that are active, one per distinct missing bit set)
__device__ __forceinline__ unsigned wave_or_active(unsigned m)
{
    unsigned acc = 0u;
    for (;;) {
        const unsigned long long need = RMDF_LANES_HERE((m & ~acc) != 0u);
        if (need == 0ull) break;
        acc |= (unsigned)RMDF_READLANE_HERE((int)m, (int)__builtin_ctzll(need));
    }
    return acc;
}

// The per-lane form of the pruned estimate (round 3).  Every lane works from ITS OWN cell mask and hint, on rows of an LDS copy of the
// table: the triangle that was nearest last time first, then one pass of bound tests over the lane's remaining candidates (the
// k-th candidate of every lane in pass k, whichever triangle that is), then the survivors, again one per lane and pass.  No scalar
// loads, no wave-uniform branches but the loop conditions: the wave-uniform loop below spends 7.6 cycles per instruction when a wave
// has the SIMD to itself (chains of scalar load -> wait -> test -> branch), straight-line vector code about 2.  min() is exact
// and order-independent and the bounds only drop provable losers: same bits.
// The candidates come from the fine grid in global memory (`fine`, CORNELL_FINE_N^3 masks); the hinted triangle is evaluated whether or
// not it is a candidate of the cell -- the minimum over a superset of the candidates is the same minimum -- so the mask is not
// needed before ~140 instructions have run, and its latency hides.
// keep (A/B only, -DRMDF_AB_SHARED_BOUNDS; not yet run on hardware): if non-null, the bound tests use a margin wider by 2e-5 and the lane's
// survivors are handed out -- de_cornell_box_lanes_kept then serves any point within 1e-5 of `pos` (the normal's other three sample points)
// without a pass of bound tests of its own.  Why that is the same minimum: the point-triangle distance is 1-Lipschitz, so for a point p'
// with |p' - pos| <= e every triangle has d'(t) >= d(t) - e and the hinted one d'(g) <= d(g) + e; a triangle whose lower bound exceeds
// d(g) + 2e (+ the usual margin) therefore cannot undercut d'(g).  The cell's mask covers p' as well: the grid's masks hold for every point
// within 1e-4 of the cell (rmdf_api.cpp: cornell_grid).
__device__ __forceinline__ float de_cornell_box_lanes(v3 pos, const float *rows, const unsigned *fine, int &hint, unsigned *keep = nullptr)
{
    const unsigned m = cornell_cell_mask_fast<CORNELL_FINE_N>(pos, fine);
    int g = hint & 31;
    float best = cornell_tri_dist2(pos, rows + g * CORNELL_STRIDE);
    // dmax^2 for dmax = 1.001 sqrt(best) + 1e-5 (the margin of de_cornell_box_table) without the root:
    // (1.001 r + 1e-5)^2 = 1.002001 b + 2.002e-5 r + 1e-10 <= 1.0031 b + 1.02e-7  (2 r <= b / 0.01 + 0.01)
    // with the margin widened to 3e-5 (keep): (1.001 r + 3e-5)^2 = 1.002001 b + 6.006e-5 r + 9e-10 <= 1.0031 b + 8.3e-7  (6.006e-5 r <= 0.0011 b + 8.2e-7)
    const float dmax2 = __builtin_fmaf(best, 1.0031f, keep ? 8.3e-7f : 1.02e-7f);
    unsigned my = m & ~(1u << g), surv = 0u;
    while (RMDF_LANES_HERE(my != 0u) != 0ull) {
        if (my != 0u) {
            const int i = (int)__builtin_ctz(my);
            my &= my - 1u;
            // lower bounds of the distance to triangle i: to its plane, and beyond each of its three edge planes (rmdf_api.cpp:
            // cornell_table).  Not part of the parity contract (they only drop provable losers), so the dot products are FMAs.
            const float4 *b = (const float4 *)(rows + i * CORNELL_STRIDE + CORNELL_BOUNDS);
            const float4 pl = b[0], ea = b[1], eb = b[2], ec = b[3];
            // The plane distance and the in-plane distance are orthogonal components of the true distance, and the in-plane distance
            // to the triangle is at least the largest of the three edge-plane distances: d^2 >= pd^2 + max(0, sa, sb, sc)^2.
            RMDF_EMU_COST(13);          // (emulator only: the bound test)
            const float pd = __builtin_fmaf(pl.z, pos.z, __builtin_fmaf(pl.y, pos.y, pl.x * pos.x)) - pl.w;
            const float sa = __builtin_fmaf(ea.z, pos.z, __builtin_fmaf(ea.y, pos.y, ea.x * pos.x)) - ea.w;
            const float sb = __builtin_fmaf(eb.z, pos.z, __builtin_fmaf(eb.y, pos.y, eb.x * pos.x)) - eb.w;
            const float sc = __builtin_fmaf(ec.z, pos.z, __builtin_fmaf(ec.y, pos.y, ec.x * pos.x)) - ec.w;
            const float sm = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(sa, sb), sc), 0.0f);
            const float bound2 = __builtin_fmaf(sm, sm, pd * pd);
            if (!(bound2 > dmax2)) surv |= 1u << i;
        }
    }
    if (keep) *keep = surv | (1u << g);
    while (RMDF_LANES_HERE(surv != 0u) != 0ull) {
        if (surv != 0u) {
            const int i = (int)__builtin_ctz(surv);
            surv &= surv - 1u;
            const float x = cornell_tri_dist2(pos, rows + i * CORNELL_STRIDE);
            if (x < best) { best = x; g = i; }
        }
    }
    hint = g;
    return sqrt_rn(best);
}
// the estimate of a point within 1e-5 of the one `kept` was made for: the kept triangles measured, nothing else (see above)
__device__ __forceinline__ float de_cornell_box_lanes_kept(v3 pos, const float *rows, unsigned kept)
{
    float best = 998001.0f;
    while (RMDF_LANES_HERE(kept != 0u) != 0ull) {
        if (kept != 0u) {
            const int i = (int)__builtin_ctz(kept);
            kept &= kept - 1u;
            const float x = cornell_tri_dist2(pos, rows + i * CORNELL_STRIDE);
            best = (x < best) ? x : best;
        }
    }
    return sqrt_rn(best);
}

// ---- the cross-lane form of the estimate (round 5): EIGHT lanes per ray ------------------------------------------------------------
// The Cornell launch lasts as long as its slowest wave, and that wave spends most of its life marching a handful of grazing rays that run
// all 128 steps: one step of the per-lane estimate above is a serial chain of ~430 instructions (the hinted triangle, a pass of bound
// tests per candidate, the survivors) that a wave with one or two live lanes issues at one instruction per ~5 cycles -- 0.9 to 1.4 us a
// step.  Once a wave is down to eight live rays (k_render), each ray moves into a GROUP of eight lanes that all hold its state, and the
// candidates of its cell are measured side by side, one triangle per lane: no bounds, no second pass, the minimum by three DPP steps.
// min() is exact and order-independent and every triangle's distance is computed by the same cornell_tri_dist2 on the same operands,
// so the estimate has the bits of the per-lane form (and of the reference's 32-triangle loop).
// Candidates: the cell's mask arrives from global memory (L2) -- a latency the per-lane form hides behind its hinted triangle; here the
// lanes measure the PREVIOUS step's candidates while it is in flight (a superset of the cell's candidates gives the same minimum: every
// value is a true triangle distance and the nearest triangle is among the cell's), and then whatever the new mask adds, usually nothing.

// index of the set bit of rank k in m (k < popcount(m)), by halving
__device__ __forceinline__ int nth_set_bit32(unsigned m, int k)
{
    int idx = 0;
    unsigned c = (unsigned)__builtin_popcount(m & 0xffffu);
    if (k >= (int)c) { k -= (int)c; m >>= 16; idx += 16; }
    c = (unsigned)__builtin_popcount(m & 0xffu);
    if (k >= (int)c) { k -= (int)c; m >>= 8; idx += 8; }
    c = (unsigned)__builtin_popcount(m & 0xfu);
    if (k >= (int)c) { k -= (int)c; m >>= 4; idx += 4; }
    c = (unsigned)__builtin_popcount(m & 0x3u);
    if (k >= (int)c) { k -= (int)c; m >>= 2; idx += 2; }
    if (k >= (int)(m & 1u)) idx += 1;
    return idx;
}
// min over the eight lanes of a group (lanes 8g .. 8g+7), result in all of them; `x < v ? x : v` as everywhere in this estimate
__device__ __forceinline__ float group8_min(float v)
{
    float o;
    o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
    v = (o < v) ? o : v;
    o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
    v = (o < v) ? o : v;
    o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xf, 0xf, false));    // row_half_mirror: lane i <-> 7 - i
    v = (o < v) ? o : v;
    return v;
}
__device__ __forceinline__ float group4_min(float v)       // min over the four lanes of a quad, result in all of them
{
    float o;
    o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false));     // quad_perm [1,0,3,2]
    v = (o < v) ? o : v;
    o = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false));     // quad_perm [2,3,0,1]
    v = (o < v) ? o : v;
    return v;
}
// One estimate for the ray of this lane's group; EVERY lane of the wave calls it (live = the group has a ray; the others ignore the result).  sub = lane & 7.  prev: the candidates measured speculatively (in/out: this step's mask for the next step).
template <int G = 8>        // lanes per ray: 8 (the product), or 4 (A/B: -DRMDF_AB_XL_G=4 -- sixteen rays per wave, two DPP steps; not yet run on hardware)
__device__ __forceinline__ float de_cornell_box_group8(bool live, v3 pos, const float *rows, const unsigned *fine, int sub, unsigned &prev)
{
    static_assert(G == 8 || G == 4, "groups of eight or four lanes");
    // (a group without a ray measures nothing: empty masks)
    const unsigned m = live ? cornell_cell_mask_fast<CORNELL_FINE_N>(pos, fine) : 0u;      // in flight while the previous candidates are measured
    if (!live) prev = 0u;
    float best = 998001.0f;                                                    // 999^2, the loop's start value (fragment.shd:400)
    unsigned set = prev;
#pragma unroll 1
    for (int round = 0; round < 2; round++) {
        // this lane's triangles of `set`: ranks sub, sub + 8, ...
        int n = __builtin_popcount(set);
        for (int k = sub; __ballot(k < n) != 0ull; k += G) {
            if (k < n) {
                const int i = nth_set_bit32(set, k);
                const float x = cornell_tri_dist2(pos, rows + i * CORNELL_STRIDE);
                best = (x < best) ? x : best;
            }
        }
        set = m & ~prev;                                                       // what the cell's own mask adds
        if (round == 0 && __ballot(set != 0u) == 0ull) break;
    }
    prev = m == 0xffffffffu ? 0u : m;          // (a point outside the grid measures all 32: nothing to carry over)
    return sqrt_rn(G == 8 ? group8_min(best) : group4_min(best));
}

__device__ __forceinline__ float de_cornell_box_table(v3 pos, const float *__restrict__ tab, int prune, int &hint, const unsigned *grid = nullptr)
{
    float dist2 = 998001.0f;                                   // 999^2
    // The table is read-only for the whole launch and every index below is wave-uniform: address it through the
    // constant address space so that the rows come in by scalar loads (s_load into SGPRs, used as instruction
    // operands).  Through a plain global pointer the compiler issues vector loads -- 9 global_load_dwordx4 and 27
    // VGPRs per triangle, with the memory latency exposed in front of every distance (measured: 0.23 G VALU
    // instructions/s/SIMD, waves waiting 68 % of the time).
    cfloat *ctab = (cfloat *)tab;
    if (!prune) {
        for (int i = 0; i < 32; i++) {
            const float x = cornell_tri_dist2(pos, ctab + i * CORNELL_STRIDE);
            dist2 = (x < dist2) ? x : dist2;
        }
        return sqrt_rn(dist2);
    }
    // candidates of this wave's cells (all 32 without a grid)
    unsigned cand = grid ? wave_or_active(cornell_cell_mask(pos, grid)) : 0xffffffffu;
    if (cand == 0u) cand = 0xffffffffu;                        // cannot happen with a well-formed grid (the nearest triangle is always in)
    // the triangle that was nearest last time first, unconditionally (if it is a candidate; else the first candidate)
    int g = RMDF_READFIRSTLANE_HERE(hint) & 31;
    if (!((cand >> g) & 1u)) g = (int)__builtin_ctz(cand);
    dist2 = cornell_tri_dist2(pos, ctab + g * CORNELL_STRIDE);
    float dmax = __builtin_amdgcn_sqrtf(dist2) * 1.001f + 1e-5f;
    hint = g;
    // then the rest in table order, four at a time: the bounds of four triangles (32 floats of the compact bounds table
    // behind the rows) arrive with two wide scalar loads, so the tests do not each wait for their own load
    cfloat *btab = ctab + 32 * CORNELL_STRIDE;
#pragma unroll 1
    for (int c = 0; c < 8; c++) {
        if (((cand >> (4 * c)) & 15u & ~((1u << g) >> (4 * c))) == 0u) continue;      // no candidate but possibly g in this chunk
        cfloat *b = btab + c * 32;
        float bb[32];
#pragma unroll
        for (int k = 0; k < 32; k++) bb[k] = b[k];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int i = c * 4 + k;
            if (i == g || !((cand >> i) & 1u)) continue;
            const float pd = fabsf(((bb[8 * k] * pos.x + bb[8 * k + 1] * pos.y) + bb[8 * k + 2] * pos.z) - bb[8 * k + 3]);
            const v3 dc = mk3(pos.x - bb[8 * k + 4], pos.y - bb[8 * k + 5], pos.z - bb[8 * k + 6]);
            const float rs = bb[8 * k + 7] + dmax;
            if ((pd > dmax) || (dot3(dc, dc) > rs * rs)) continue;
            const float x = cornell_tri_dist2(pos, ctab + i * CORNELL_STRIDE);
            if (x < dist2) {
                dist2 = x;
                dmax = __builtin_amdgcn_sqrtf(x) * 1.001f + 1e-5f;
                hint = i;
            }
        }
    }
    return sqrt_rn(dist2);
}

// fragment.shd:694-719.  FAST: the two quotients as div_known_range -- for the shader's eta = 0.4, k = 0.8 and |cosi| <= 1 (a dot
// product of two unit vectors) every numerator and denominator lies in [0.64, 2.6]; rmdf_selftest_shading_math compares the two
// forms for every cosi in [-2, 2] and NaN.  The alternative schedules of librmdf_xcheck keep the compiler's division.
template <bool FAST = RMDF_SHADE_FAST>
__device__ __forceinline__ float fresnel_conductor(float cosi, float eta, float k)
{
    float tmp = (eta * eta + k * k) * cosi * cosi;
    const float pn = tmp - (2.0f * eta * cosi) + 1.0f, pd = tmp + (2.0f * eta * cosi) + 1.0f;
    float r_parallel_2 = FAST ? RMDF_FAST_DIV(pn, pd) : pn / pd;
    float tmp_f = eta * eta + k * k;
    const float qn = tmp_f - (2.0f * eta * cosi) + cosi * cosi, qd = tmp_f + (2.0f * eta * cosi) + cosi * cosi;
    float r_perpend_2 = FAST ? RMDF_FAST_DIV(qn, qd) : qn / qd;
    return (r_parallel_2 + r_perpend_2) / 2.0f;
}

// fragment.shd:595-616, spherePos = 0
template <bool FAST = RMDF_SHADE_FAST>
__device__ __forceinline__ bool ray_sphere(v3 origin, v3 dir, float R, float &tmin, float &tmax)
{
    v3 rs = mk3(0.0f - origin.x, 0.0f - origin.y, 0.0f - origin.z);
    float t = dot3(dir, rs);
    float a = dot3(rs, rs) - t * t;
    float r2 = R * R;
    if (a > r2) return false;
    float h = FAST ? sqrt_rn(r2 - a) : sqrtf(r2 - a);        // sqrt_rn: the same bits for every input (rmdf_selftest_exact_math)
    tmin = t - h;
    tmax = t + h;
    return true;
}

// ---- samplerCube: RGB16F texels, min NEAREST / mag LINEAR, seamless (padded) -------

struct CubeDev {
    const uint2 *texels;   // 6 * (W+2)^2 texels of 4 halfs (r,g,b,0)
    int W;
};

__device__ __forceinline__ void cube_coords(v3 r, int &face, float &sc, float &tc, float &ma)
{
    float ax = fabsf(r.x), ay = fabsf(r.y), az = fabsf(r.z);
    if (ax >= ay && ax >= az) {
        if (r.x > 0.0f) { face = 0; sc = -r.z; tc = -r.y; ma = r.x; }
        else            { face = 1; sc =  r.z; tc = -r.y; ma = r.x; }
    } else if (ay >= az) {
        if (r.y > 0.0f) { face = 2; sc =  r.x; tc =  r.z; ma = r.y; }
        else            { face = 3; sc =  r.x; tc = -r.z; ma = r.y; }
    } else {
        if (r.z > 0.0f) { face = 4; sc =  r.x; tc = -r.y; ma = r.z; }
        else            { face = 5; sc = -r.x; tc = -r.y; ma = r.z; }
    }
}

__device__ __forceinline__ void cube_project(int face, v3 q, float &sc, float &tc, float &ma)
{
    switch (face) {
    case 0:  sc = -q.z; tc = -q.y; ma = q.x; break;
    case 1:  sc =  q.z; tc = -q.y; ma = q.x; break;
    case 2:  sc =  q.x; tc =  q.z; ma = q.y; break;
    case 3:  sc =  q.x; tc = -q.z; ma = q.y; break;
    case 4:  sc =  q.x; tc = -q.y; ma = q.z; break;
    default: sc = -q.x; tc = -q.y; ma = q.z; break;
    }
}

__device__ __forceinline__ float cube_texcoord(float c, float ama, float W) { return (0.5f * (c / ama + 1.0f)) * W; }
// The same with y = rcp_core(ama) (two coordinates share one major axis).  The quotient is the IEEE one whenever ama is in
// [2^-40, 2^40] and |c| >= 2^-102 (no underflow in the remainder); a smaller |c| gives a quotient below 2^-60 either way, which
// the "+ 1" absorbs.  The lane's own direction is a unit vector (ama in [0.57, 1]) or NaN.  A NEIGHBOUR's direction projected
// on this lane's face can have any major-axis component > 0: below 2^-40 one of the two quotients is beyond 2^38 (or inf / NaN
// from the reciprocal), its texel distance fails "<= 1" in both forms, and the quotients feed nothing else (cube_texture).
__device__ __forceinline__ float cube_texcoord_rcp(float c, float ama, float y, float W) { return (0.5f * (div_with_rcp(c, ama, y) + 1.0f)) * W; }

// RMDF_AB_NO_TEXEL_FETCH (tools/abtest only, never defined in the product build): every texel read is replaced by a value
// made from its address -- no memory access at all.  The frame is wrong, of course; the build exists to measure an UPPER
// BOUND of what any staging of env-map texels (LDS or otherwise) could save: NOTEBOOK.md A.1.
#ifdef RMDF_AB_NO_TEXEL_FETCH
#define RMDF_TEXEL(ptr) make_uint2((unsigned)(size_t)(ptr) & 0x3bff3bffu, 0x3c00u)
#else
#define RMDF_TEXEL(ptr) (*(ptr))
#endif

__device__ __forceinline__ v3 texel_rgb(uint2 t)
{
    __half2 rg = *reinterpret_cast<__half2 *>(&t.x);
    __half2 b0 = *reinterpret_cast<__half2 *>(&t.y);
    return mk3(__low2float(rg), __high2float(rg), __low2float(b0));
}

__device__ __forceinline__ v3 cube_fetch_nearest(const CubeDev &c, int face, float u, float v)
{
    float Wm1 = (float)(c.W - 1);
    float fi = floorf(u), fj = floorf(v);
    if (!(fi >= 0.0f)) fi = 0.0f;
    if (fi > Wm1) fi = Wm1;
    if (!(fj >= 0.0f)) fj = 0.0f;
    if (fj > Wm1) fj = Wm1;
    int P = c.W + 2;
    return texel_rgb(RMDF_TEXEL(&c.texels[(face * P + ((int)fj + 1)) * P + ((int)fi + 1)]));
}

__device__ __forceinline__ v3 cube_fetch_linear(const CubeDev &c, int face, float u, float v)
{
    float Wm1 = (float)(c.W - 1);
    float ub = u - 0.5f, vb = v - 0.5f;
    float fi = floorf(ub), fj = floorf(vb);
    if (!(fi >= -1.0f)) fi = -1.0f;
    if (fi > Wm1) fi = Wm1;
    if (!(fj >= -1.0f)) fj = -1.0f;
    if (fj > Wm1) fj = Wm1;
    float fu = ub - fi, fv = vb - fj;
    float gu = 1.0f - fu, gv = 1.0f - fv;
    int P = c.W + 2;
    const uint2 *row0 = c.texels + (face * P + ((int)fj + 1)) * P + ((int)fi + 1);
    v3 t00 = texel_rgb(RMDF_TEXEL(row0)), t10 = texel_rgb(RMDF_TEXEL(row0 + 1));
    v3 t01 = texel_rgb(RMDF_TEXEL(row0 + P)), t11 = texel_rgb(RMDF_TEXEL(row0 + P + 1));
    v3 o;
    o.x = (t00.x * gu + t10.x * fu) * gv + (t01.x * gu + t11.x * fu) * fv;
    o.y = (t00.y * gu + t10.y * fu) * gv + (t01.y * gu + t11.y * fu) * fv;
    o.z = (t00.z * gu + t10.z * fu) * gv + (t01.z * gu + t11.z * fu) * fv;
    return o;
}

// texture(samplerCube, r) inside a 2x2 quad; rh / rv = the same expression in the
// horizontal / vertical quad neighbour, valid_* = that neighbour evaluated it.
template <bool FAST = RMDF_SHADE_FAST>
__device__ __forceinline__ v3 cube_texture(const CubeDev &c, v3 r, bool valid_h, v3 rh, bool valid_v, v3 rv)
{
    int face; float sc, tc, ma;
    cube_coords(r, face, sc, tc, ma);
    float W = (float)c.W, ama = fabsf(ma);
    float u, v;
    if (FAST) { const float y = rcp_core(ama); u = cube_texcoord_rcp(sc, ama, y, W); v = cube_texcoord_rcp(tc, ama, y, W); }
    else { u = cube_texcoord(sc, ama, W); v = cube_texcoord(tc, ama, W); }
    bool linear = false;
    if (valid_h && valid_v) {
        float sh, th, mh, sv, tv, mv;
        cube_project(face, rh, sh, th, mh);
        cube_project(face, rv, sv, tv, mv);
        bool pos = (face & 1) == 0;
        bool okh = pos ? (mh > 0.0f) : (mh < 0.0f);
        bool okv = pos ? (mv > 0.0f) : (mv < 0.0f);
        if (okh && okv) {
            float amh = fabsf(mh), amv = fabsf(mv);
            float dux, dvx, duy, dvy;
            if (FAST) {
                const float yh = rcp_core(amh), yv = rcp_core(amv);
                dux = cube_texcoord_rcp(sh, amh, yh, W) - u; dvx = cube_texcoord_rcp(th, amh, yh, W) - v;
                duy = cube_texcoord_rcp(sv, amv, yv, W) - u; dvy = cube_texcoord_rcp(tv, amv, yv, W) - v;
            } else {
                dux = cube_texcoord(sh, amh, W) - u; dvx = cube_texcoord(th, amh, W) - v;
                duy = cube_texcoord(sv, amv, W) - u; dvy = cube_texcoord(tv, amv, W) - v;
            }
            float rx = dux * dux + dvx * dvx;
            float ry = duy * duy + dvy * dvy;
            linear = (rx <= 1.0f) && (ry <= 1.0f);
        }
    }
    return linear ? cube_fetch_linear(c, face, u, v) : cube_fetch_nearest(c, face, u, v);
}

__device__ __forceinline__ uint32_t to_unorm8(float g)
{
    if (!(g == g)) return 0u;
    return (uint32_t)rintf(gclamp(g, 0.0f, 1.0f) * 255.0f);
}

}  // namespace rmdf
