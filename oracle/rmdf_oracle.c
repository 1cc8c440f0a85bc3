/*
 * rmdf_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE).
 * See rmdf_oracle.h for status ("parity unpinned" by the reference's own
 * vectors) and for who may load this file.  Citations are file:line into the
 * reference tree (blitzcode/ray-marching-distance-fields).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no -ffast-math).
 */
#include "rmdf_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

/* ============================================================================
 * 0. Pinned float32 semantics of the GLSL built-ins (implementation-defined in
 *    GLSL; see DESIGN.md "spec pins").  Everything is one IEEE rounding per op.
 * ==========================================================================*/

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }

/* GLSL min(x,y) = y < x ? y : x ; max(x,y) = x < y ? y : x ; clamp = min(max(x,lo),hi) */
static inline float rm_min(float x, float y) { return (y < x) ? y : x; }
static inline float rm_max(float x, float y) { return (x < y) ? y : x; }
static inline float rm_clamp(float x, float lo, float hi) { return rm_min(rm_max(x, lo), hi); }

static inline float rm_dot(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline float rm_length(v3 a) { return sqrtf(rm_dot(a, a)); }
/* inversesqrt(x) := 1/sqrt(x), both correctly rounded */
static inline float rm_rsqrt(float x) { return 1.0f / sqrtf(x); }
/* normalize(v) := v * inversesqrt(dot(v,v)) */
static inline v3 rm_normalize(v3 a)
{
    float s = rm_rsqrt(rm_dot(a, a));
    return V3(a.x * s, a.y * s, a.z * s);
}
static inline v3 rm_cross(v3 a, v3 b)
{
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline v3 rm_sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 rm_add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 rm_scale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline v3 rm_neg(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* reflect(I,N) = I - 2*dot(N,I)*N */
static inline v3 rm_reflect(v3 i, v3 n)
{
    float k = 2.0f * rm_dot(n, i);
    return V3(i.x - k * n.x, i.y - k * n.y, i.z - k * n.z);
}

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

/* log(x): the classic fdlibm float algorithm (argument reduction to
 * [sqrt(1/2), sqrt(2)), s = f/(2+f), degree-4 even polynomial), written with a
 * fixed operation order so that a second implementation can reproduce it bit
 * for bit with IEEE +,-,*,/ only.  < 1 ulp. */
static float rm_logf(float x)
{
    const float ln2_hi = 6.9313812256e-01f;  /* 0x3f317180 */
    const float ln2_lo = 9.0580006145e-06f;  /* 0x3717f7d1 */
    const float Lg1 = 0.66666662693f, Lg2 = 0.40000972152f, Lg3 = 0.28498786688f, Lg4 = 0.24279078841f;
    int32_t ix = (int32_t)f2u(x);
    int32_t k = 0;
    if (ix < 0x00800000) {                       /* x < 2^-126, zero or negative */
        if ((ix & 0x7fffffff) == 0) return -INFINITY;
        if (ix < 0) return NAN;
        k = -25;
        x = x * 33554432.0f;                     /* 2^25 */
        ix = (int32_t)f2u(x);
    }
    if (ix >= 0x7f800000) return x + x;          /* inf or NaN */
    k += (ix >> 23) - 127;
    ix &= 0x007fffff;
    int32_t i = (ix + 0x4afb20) & 0x800000;      /* mantissa >= sqrt(2) ? */
    x = u2f((uint32_t)(ix | (i ^ 0x3f800000)));  /* x or x/2 in [sqrt(1/2), sqrt(2)) */
    k += (i >> 23);
    float f = x - 1.0f;
    float s = f / (2.0f + f);
    float dk = (float)k;
    float z = s * s;
    float w = z * z;
    float t1 = w * (Lg2 + w * Lg4);
    float t2 = z * (Lg1 + w * Lg3);
    float R = t2 + t1;
    float hfsq = (0.5f * f) * f;
    return dk * ln2_hi - ((hfsq - (s * (hfsq + R) + dk * ln2_lo)) - f);
}

/* exp(x): fdlibm float algorithm, single code path, fixed operation order.
 * Domain pinned: x < -87 -> 0, x > 88.5 -> +inf. */
static float rm_expf(float x)
{
    const float ln2_hi = 6.9314575195e-01f;  /* 0x3f317200 */
    const float ln2_lo = 1.4286067653e-06f;  /* 0x35bfbe8e */
    const float invln2 = 1.4426950216e+00f;  /* 0x3fb8aa3b */
    const float P1 = 1.6666625440e-1f, P2 = -2.7667332906e-3f;
    if (x != x) return x;
    if (x > 88.5f) return INFINITY;
    if (x < -87.0f) return 0.0f;
    float kf = x * invln2 + ((x < 0.0f) ? -0.5f : 0.5f);
    int32_t k = (int32_t)kf;                 /* truncation toward zero */
    float t = (float)k;
    float hi = x - t * ln2_hi;
    float lo = t * ln2_lo;
    float r = hi - lo;
    float tt = r * r;
    float c = r - tt * (P1 + tt * P2);
    float y = 1.0f - ((lo - (r * c) / (2.0f - c)) - hi);
    /* scale by 2^k, k in [-126, 128]: two exact-power multiplications */
    int32_t k1 = k / 2, k2 = k - k1;
    y = y * u2f((uint32_t)(k1 + 127) << 23);
    y = y * u2f((uint32_t)(k2 + 127) << 23);
    return y;
}

/* pow(x,y) := exp(y*log(x)); GLSL leaves x<0 undefined and x==0 needs y>0:
 * pinned to 0 for x <= 0 and for NaN. */
static float rm_powf(float x, float y)
{
    if (!(x > 0.0f)) return 0.0f;
    return rm_expf(y * rm_logf(x));
}


/* ---- pinned sin / cos / acos / atan (scenes FSMBGeneralShader and FSDETestShader) ----
 * fdlibm-style float algorithms with a fixed operation order, < 2 ulp on the ranges the shader uses.
 * Argument reduction: x - k*(pi/2) with a 4-part constant whose leading parts have <= 11 significant
 * bits, so k*c is exact for |k| < 2^13 (|x| < ~1.2e4; beyond that the result is pinned to the same
 * formula, merely less accurate). */
static void rm_rem_pio2(float x, float *r, int *q)
{
    const float invpio2 = 0.6366197466850281f;
    const float c1 = 1.5703125f, c2 = 4.837512969970703e-4f, c3 = 7.549533620476723e-8f, c4 = 2.5633440682570896e-12f;
    float kf = rintf(x * invpio2);
    *q = (int)kf;
    float t = x - kf * c1;
    t = t - kf * c2;
    t = t - kf * c3;
    t = t - kf * c4;
    *r = t;
}
static float rm_ksin(float x)
{
    const float S1 = -1.6666667163e-01f, S2 = 8.3333337680e-03f, S3 = -1.9841270114e-04f, S4 = 2.7557314297e-06f,
                S5 = -2.5050759689e-08f, S6 = 1.5896910177e-10f;
    float z = x * x;
    float v = z * x;
    float r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return x + v * (S1 + z * r);
}
static float rm_kcos(float x)
{
    const float C1 = 4.1666667908e-02f, C2 = -1.3888889225e-03f, C3 = 2.4801587642e-05f, C4 = -2.7557314297e-07f,
                C5 = 2.0875723372e-09f, C6 = -1.1359647598e-11f;
    float z = x * x;
    float r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    return 1.0f - (0.5f * z - z * r);
}
static float rm_sinf(float x)
{
    if (!(fabsf(x) <= 3.4e38f)) return x - x;      /* inf, NaN -> NaN */
    float r; int q;
    rm_rem_pio2(x, &r, &q);
    switch (q & 3) {
    case 0:  return rm_ksin(r);
    case 1:  return rm_kcos(r);
    case 2:  return -rm_ksin(r);
    default: return -rm_kcos(r);
    }
}
static float rm_cosf(float x)
{
    if (!(fabsf(x) <= 3.4e38f)) return x - x;
    float r; int q;
    rm_rem_pio2(x, &r, &q);
    switch (q & 3) {
    case 0:  return rm_kcos(r);
    case 1:  return -rm_ksin(r);
    case 2:  return -rm_kcos(r);
    default: return rm_ksin(r);
    }
}
static float rm_acosf(float x)
{
    const float pio2_hi = 1.5707962513e+00f, pio2_lo = 7.5497894159e-08f, pi = 3.1415925026e+00f;
    const float pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f, pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f,
                pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f;
    const float qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
    float ax = fabsf(x);
    if (!(ax <= 1.0f)) return (x - x) / (x - x);            /* |x| > 1 or NaN -> NaN */
    if (ax == 1.0f) return (x > 0.0f) ? 0.0f : pi + 2.0f * pio2_lo;
    if (ax < 0.5f) {
        if (ax <= 1.4901161e-8f) return pio2_hi + pio2_lo;   /* 2^-26 */
        float z = x * x;
        float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        float r = p / q;
        return pio2_hi - (x - (pio2_lo - x * r));
    } else if (x < 0.0f) {
        float z = (1.0f + x) * 0.5f;
        float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        float s = sqrtf(z);
        float r = p / q;
        float w = r * s - pio2_lo;
        return pi - 2.0f * (s + w);
    } else {
        float z = (1.0f - x) * 0.5f;
        float s = sqrtf(z);
        float df = u2f(f2u(s) & 0xfffff000u);
        float c = (z - df * df) / (s + df);
        float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
        float q = 1.0f + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
        float r = p / q;
        float w = r * s + c;
        return 2.0f * (df + w);
    }
}
static float rm_atanf(float x)
{
    static const float atanhi[4] = { 4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f };
    static const float atanlo[4] = { 5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f };
    static const float aT[11] = { 3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f,
                                  9.0908870101e-02f, -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f,
                                  4.9768779427e-02f, -3.6531571299e-02f, 1.6285819933e-02f };
    if (x != x) return x;
    float ax = fabsf(x);
    int neg = (f2u(x) >> 31) != 0;
    int id;
    float t;
    if (ax >= 67108864.0f) {                                /* 2^26 */
        float z = atanhi[3] + atanlo[3];
        return neg ? -z : z;
    }
    if (ax < 0.4375f) {
        if (ax < 2.44140625e-4f) return x;                   /* 2^-12 */
        id = -1; t = x;
    } else if (ax < 1.1875f) {
        if (ax < 0.6875f) { id = 0; t = (2.0f * ax - 1.0f) / (2.0f + ax); }
        else              { id = 1; t = (ax - 1.0f) / (ax + 1.0f); }
    } else {
        if (ax < 2.4375f) { id = 2; t = (ax - 1.5f) / (1.0f + 1.5f * ax); }
        else              { id = 3; t = -1.0f / ax; }
    }
    float z = t * t;
    float w = z * z;
    float s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
    float s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
    if (id < 0) return t - t * (s1 + s2);
    float zz = atanhi[id] - ((t * (s1 + s2) - atanlo[id]) - t);
    return neg ? -zz : zz;
}
/* GLSL atan(y, x) */
static float rm_atan2f(float y, float x)
{
    const float pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f, pio2 = 1.5707963705e+00f;
    if (x != x || y != y) return x + y;
    int m = (int)((f2u(y) >> 31) | ((f2u(x) >> 30) & 2u));   /* 2*sign(x) + sign(y) */
    float ax = fabsf(x), ay = fabsf(y);
    if (ay == 0.0f) {
        switch (m) { case 0: case 1: return y; case 2: return pi; default: return -pi; }
    }
    if (ax == 0.0f) return (m & 1) ? -pio2 : pio2;
    if (ax == INFINITY) {
        if (ay == INFINITY) {
            switch (m) { case 0: return 0.25f * pi; case 1: return -0.25f * pi; case 2: return 0.75f * pi; default: return -0.75f * pi; }
        }
        switch (m) { case 0: return 0.0f; case 1: return -0.0f; case 2: return pi; default: return -pi; }
    }
    if (ay == INFINITY) return (m & 1) ? -pio2 : pio2;
    float z = rm_atanf(ay / ax);                              /* |y/x| */
    switch (m) {
    case 0:  return z;
    case 1:  return -z;
    case 2:  return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
    }
}
/* GLSL mod(x, y) = x - y*floor(x/y) */
static inline float rm_mod(float x, float y) { return x - y * floorf(x / y); }

float orc_sinf(float x) { return rm_sinf(x); }
float orc_cosf(float x) { return rm_cosf(x); }
float orc_acosf(float x) { return rm_acosf(x); }
float orc_atan2f(float y, float x) { return rm_atan2f(y, x); }

float orc_logf(float x) { return rm_logf(x); }
float orc_expf(float x) { return rm_expf(x); }
float orc_powf(float x, float y) { return rm_powf(x, y); }

/* ============================================================================
 * 1. Distance estimators
 * ==========================================================================*/

/* fragment.shd:74-99 -- closed-form w^8, evaluation order exactly as parsed */
static v3 triplex_pow8(v3 w)
{
    float x = w.x; float x2 = x * x; float x4 = x2 * x2;
    float y = w.y; float y2 = y * y; float y4 = y2 * y2;
    float z = w.z; float z2 = z * z; float z4 = z2 * z2;

    float k3 = y2 + x2;
    float k2 = rm_rsqrt(k3 * k3 * k3 * k3 * k3 * k3 * k3);
    float k1 = y4 + z4 + x4 - 6.0f * z2 * x2 - 6.0f * y2 * z2 + 2.0f * x2 * y2;
    float k4 = y2 - z2 + x2;

    return V3(-8.0f * z * k4 * (y4 * y4 - 28.0f * y4 * y2 * x2 + 70.0f * y4 * x4 - 28.0f * y2 * x2 * x4 + x4 * x4) * k1 * k2,
              64.0f * y * z * x * (y2 - x2) * k4 * (y4 - 6.0f * y2 * x2 + x4) * k1 * k2,
              -16.0f * z2 * k3 * k4 * k4 + k1 * k1);
}

void orc_triplex_pow8(const float w[3], float out[3])
{
    v3 r = triplex_pow8(V3(w[0], w[1], w[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* The constants fragment.shd states, by name (value, shader line).  The restatement below uses these names; orc_shader_constants
 * hands the table out and tests/test_reference_pins.py compares it with the values a script extracted from the reference itself
 * (tests/golden/reference_pins.json). */
#define ORC_SHADER_CONSTANTS(X)                                                                                              \
    X(mb_bailout, 4.0f) X(mb_iterations, 25.0f)                         /* fragment.shd:121-122 */                            \
    X(march_max_steps_default, 128.0f) X(march_min_dist, 0.001f)         /* :634-635 */                                        \
    X(bsphere_r_power8, 1.15f) X(bsphere_r_general, 1.5f) X(bsphere_r_other, 1.0f)   /* :643-648 */                            \
    X(ao_w0, 0.5f) X(ao_d0, 0.016f) X(ao_w1, 0.25f) X(ao_d1, 0.081f) X(ao_bias, 0.29f) X(ao_gain, 3.5f)   /* :548-559 */       \
    X(cornell_ao_w0, 0.1f) X(cornell_ao_d0, 0.1f) X(cornell_ao_w1, 0.2f) X(cornell_ao_d1, 0.2f)          /* :571-576 */       \
    X(cornell_ao_w2, 0.125f) X(cornell_ao_d2, 0.4f) X(cornell_ao_w3, 0.0625f) X(cornell_ao_d3, 0.5f)     /* :579-584 */       \
    X(normal_eps, 0.00001f) X(isec_step_back, 0.00001f)                  /* :466, :751 */                                      \
    X(fresnel_eta, 0.4f) X(fresnel_k, 0.8f) X(diff_weight, 0.5f)         /* :799, :801 */                                      \
    X(diff_r, 1.0f) X(diff_g, 0.8f) X(diff_b, 0.8f) X(spec_r, 0.8f) X(spec_g, 0.8f) X(spec_b, 1.0f)      /* :802-803 */       \
    X(spec_weight_one_minus, 1.0f) X(phong_lobe_n, 8.0f) X(refl_weight, 0.1f) X(exposure, 3.0f)          /* :804-810 */       \
    X(phong_lobe_plus, 2.0f) X(phong_lobe_div, 2.0f)                     /* :723 */                                            \
    X(camera_distance, 2.414213562373095f) X(camera_cornell_radius, 0.4f) X(camera_cornell_z, -2.0f)     /* :888-897 */       \
    X(hfov_deg_a, 45.0f) X(hfov_deg_b, 1.5f) X(gamma, 2.2f)              /* :910, :959 */
#define ORC_X(name, value) static const float K_##name = value;
ORC_SHADER_CONSTANTS(ORC_X)
#undef ORC_X

int orc_shader_constants(const char **names, float *values, int cap)
{
    static const char *const k_names[] = {
#define ORC_X(name, value) #name,
        ORC_SHADER_CONSTANTS(ORC_X)
#undef ORC_X
    };
    const float k_values[] = {          /* the variables the code below uses, not a second copy of the literals */
#define ORC_X(name, value) K_##name,
        ORC_SHADER_CONSTANTS(ORC_X)
#undef ORC_X
    };
    const int n = (int)(sizeof k_values / sizeof k_values[0]);
    for (int i = 0; i < n && i < cap; i++) { if (names) names[i] = k_names[i]; if (values) values[i] = k_values[i]; }
    return n;
}

typedef struct {
    int      scene;
    float    time;
    const float *cornell;    /* 96*3 */
    uint64_t de_evals, triplex_iters, tri_inside;
    float    power;          /* FSMBGeneralShader: fragment.shd:116-119 */
} de_ctx;

/* fragment.shd:101-158 with POWER8: pow(r, power-1) pinned as the multiply chain
 * r2=r*r, r4=r2*r2, r7=(r4*r2)*r */
static float de_mandelbulb8(v3 pos, de_ctx *c)
{
    const float bailout = K_mb_bailout;
    const int iterations = (int)K_mb_iterations;
    pos = V3(pos.z, pos.x, pos.y);          /* pos.zxy, :125 */
    v3 w = pos;
    float dr = 1.0f;
    float r = 0.0f;
    for (int i = 0; i < iterations; i++) {
        r = rm_length(w);
        if (r > bailout) break;
        w = triplex_pow8(w);
        w = rm_add(w, pos);
        float r2 = r * r, r4 = r2 * r2, r7 = (r4 * r2) * r;
        dr = r7 * 8.0f * dr + 1.0f;         /* :148 */
        c->triplex_iters++;
    }
    return 0.5f * rm_logf(r) * r / dr;      /* :157 */
}


/* fragment.shd:116-119: the animated power of FSMBGeneralShader, uniform per frame */
static float general_power(float time)
{
    float pow_offs = rm_mod(time / 2.0f, 9.0f);
    if (pow_offs > 4.5f) pow_offs = 9.0f - pow_offs;
    return pow_offs + 2.0f;
}
float orc_general_power(float time) { return general_power(time); }

/* fragment.shd:42-72 */
static v3 triplex_pow(v3 w, float power)
{
    float r = rm_length(w);
    float theta = rm_acosf(w.z / r);
    float phi = rm_atan2f(w.y, w.x);
    float zr = rm_powf(r, power);
    theta = theta * power;
    phi = phi * power;
    float st = rm_sinf(theta), ct = rm_cosf(theta), sp = rm_sinf(phi), cp = rm_cosf(phi);
    return V3(zr * (st * cp), zr * (st * sp), zr * ct);
}

void orc_triplex_pow(const float w[3], float power, float out[3])
{
    v3 r = triplex_pow(V3(w[0], w[1], w[2]), power);
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* fragment.shd:101-158 without POWER8 */
static float de_mandelbulb_general(v3 pos, de_ctx *c)
{
    const float bailout = K_mb_bailout;
    const float power = c->power;
    pos = V3(pos.z, pos.x, pos.y);
    v3 w = pos;
    float dr = 1.0f;
    float r = 0.0f;
    for (int i = 0; i < (int)K_mb_iterations; i++) {
        r = rm_length(w);
        if (r > bailout) break;
        w = triplex_pow(w, power);
        w = rm_add(w, pos);
        dr = rm_powf(r, power - 1.0f) * power * dr + 1.0f;
        c->triplex_iters++;
    }
    return 0.5f * rm_logf(r) * r / dr;
}

/* fragment.shd:21-33, 413-418, 447-456 */
static inline float length2(float x, float y) { return sqrtf(x * x + y * y); }
static float de_torus(v3 p, float size, float r) { return length2(length2(p.x, p.y) - size, p.z) - r; }
static float de_rounded_box(v3 p, v3 b, float r)
{
    v3 q = V3(rm_max(fabsf(p.x) - b.x, 0.0f), rm_max(fabsf(p.y) - b.y, 0.0f), rm_max(fabsf(p.z) - b.z, 0.0f));
    return rm_length(q) - r;
}
static float smin(float a, float b, float k)
{
    float res = rm_expf(-k * a) + rm_expf(-k * b);
    return -rm_logf(res) / k;
}
static float de_test_scene(v3 pos)
{
    float d_sphere = rm_length(pos) - 0.4f;
    float d_torus = smin(smin(de_torus(pos, 0.85f, 0.1f), de_torus(V3(pos.z, pos.x, pos.y), 0.85f, 0.1f), 64.0f),
                         de_torus(V3(pos.y, pos.z, pos.x), 0.85f, 0.1f), 64.0f);
    float d_box = smin(smin(de_rounded_box(pos, V3(0.8f, 0.06f, 0.06f), 0.03f), de_rounded_box(pos, V3(0.06f, 0.8f, 0.06f), 0.03f), 64.0f),
                       de_rounded_box(pos, V3(0.06f, 0.06f, 0.8f), 0.03f), 64.0f);
    return smin(d_box, rm_min(d_sphere, d_torus), 64.0f);
}

/* fragment.shd:312-321 */
static float line_seg_min_dist_sq(v3 a, v3 b, v3 p)
{
    v3 ab = rm_sub(b, a);
    float len_sq = rm_dot(ab, ab);
    float t = rm_dot(rm_sub(p, a), ab) / len_sq;
    t = rm_clamp(t, 0.0f, 1.0f);
    v3 proj = V3(a.x + t * ab.x, a.y + t * ab.y, a.z + t * ab.z);
    v3 d = rm_sub(p, proj);
    return rm_dot(d, d);
}

/* fragment.shd:323-346 */
static int compute_barycentric(v3 pos, v3 v0, v3 v1, v3 v2, float *u, float *v)
{
    v3 e0 = rm_sub(v2, v0);
    v3 e1 = rm_sub(v1, v0);
    v3 e2 = rm_sub(pos, v0);
    float dot00 = rm_dot(e0, e0);
    float dot01 = rm_dot(e0, e1);
    float dot02 = rm_dot(e0, e2);
    float dot11 = rm_dot(e1, e1);
    float dot12 = rm_dot(e1, e2);
    float inv_denom = 1.0f / (dot00 * dot11 - dot01 * dot01);
    *u = (dot11 * dot02 - dot01 * dot12) * inv_denom;
    *v = (dot00 * dot12 - dot01 * dot02) * inv_denom;
    return (*u >= 0.0f) && (*v >= 0.0f) && (*u + *v < 1.0f);
}

/* fragment.shd:348-372 */
static float de_triangle(v3 pos, v3 v0, v3 v1, v3 v2, uint64_t *n_inside)
{
    float u, v;
    if (compute_barycentric(pos, v0, v1, v2, &u, &v)) {
        (*n_inside)++;                          /* the prism branch (roofline operation count: bench.py) */
        float k = 1.0f - (u + v);
        /* v2*u + v1*v + v0*(1-(u+v)) */
        v3 pp = V3(v2.x * u + v1.x * v + v0.x * k,
                   v2.y * u + v1.y * v + v0.y * k,
                   v2.z * u + v1.z * v + v0.z * k);
        return rm_length(rm_sub(pos, pp));  /* distance() */
    } else {
        return sqrtf(rm_min(line_seg_min_dist_sq(v0, v1, pos),
                            rm_min(line_seg_min_dist_sq(v0, v2, pos),
                                   line_seg_min_dist_sq(v1, v2, pos))));
    }
}

/* fragment.shd:374-411 */
static float de_cornell_box(v3 pos, de_ctx *c)
{
    float dist = 999.0f;
    for (int i = 0; i < 32; i++) {
        const float *t = c->cornell + i * 9;
        dist = rm_min(dist, de_triangle(pos, V3(t[0], t[1], t[2]), V3(t[3], t[4], t[5]), V3(t[6], t[7], t[8]), &c->tri_inside));
    }
    return dist;
}

/* fragment.shd:420-458 */
static float distance_estimator(v3 pos, de_ctx *c)
{
    c->de_evals++;
    switch (c->scene) {
    case ORC_SCENE_MB_POWER8: return de_mandelbulb8(pos, c);
    case ORC_SCENE_CORNELL:   return de_cornell_box(pos, c);
    case ORC_SCENE_MB_GENERAL: return de_mandelbulb_general(pos, c);
    default:                  return de_test_scene(pos);
    }
}

/* ============================================================================
 * 2. Cornell geometry (CornellBox.hs:21-129)
 * ==========================================================================*/

static const float cornell_quads[64][3] = {
    /* floor */
    { 552.8f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 559.2f }, { 549.6f, 0.0f, 559.2f },
    /* ceiling */
    { 556.0f, 548.8f, 0.0f }, { 556.0f, 548.8f, 559.2f }, { 0.0f, 548.8f, 559.2f }, { 0.0f, 548.8f, 0.0f },
    /* back wall */
    { 549.6f, 0.0f, 559.2f }, { 0.0f, 0.0f, 559.2f }, { 0.0f, 548.8f, 559.2f }, { 556.0f, 548.8f, 559.2f },
    /* right wall */
    { 0.0f, 0.0f, 559.2f }, { 0.0f, 0.0f, 0.0f }, { 0.0f, 548.8f, 0.0f }, { 0.0f, 548.8f, 559.2f },
    /* left wall */
    { 552.8f, 0.0f, 0.0f }, { 549.6f, 0.0f, 559.2f }, { 556.0f, 548.8f, 559.2f }, { 556.0f, 548.8f, 0.0f },
    /* light: y = 548.8 - 0.1 evaluated in Float (CornellBox.hs:81-84) */
    { 343.0f, 0.0f, 227.0f }, { 343.0f, 0.0f, 332.0f }, { 213.0f, 0.0f, 332.0f }, { 213.0f, 0.0f, 227.0f },
    /* short block */
    { 130.0f, 165.0f, 65.0f }, { 82.0f, 165.0f, 225.0f }, { 240.0f, 165.0f, 272.0f }, { 290.0f, 165.0f, 114.0f },
    { 290.0f, 0.0f, 114.0f }, { 290.0f, 165.0f, 114.0f }, { 240.0f, 165.0f, 272.0f }, { 240.0f, 0.0f, 272.0f },
    { 130.0f, 0.0f, 65.0f }, { 130.0f, 165.0f, 65.0f }, { 290.0f, 165.0f, 114.0f }, { 290.0f, 0.0f, 114.0f },
    { 82.0f, 0.0f, 225.0f }, { 82.0f, 165.0f, 225.0f }, { 130.0f, 165.0f, 65.0f }, { 130.0f, 0.0f, 65.0f },
    { 240.0f, 0.0f, 272.0f }, { 240.0f, 165.0f, 272.0f }, { 82.0f, 165.0f, 225.0f }, { 82.0f, 0.0f, 225.0f },
    /* tall block */
    { 423.0f, 330.0f, 247.0f }, { 265.0f, 330.0f, 296.0f }, { 314.0f, 330.0f, 456.0f }, { 472.0f, 330.0f, 406.0f },
    { 423.0f, 0.0f, 247.0f }, { 423.0f, 330.0f, 247.0f }, { 472.0f, 330.0f, 406.0f }, { 472.0f, 0.0f, 406.0f },
    { 472.0f, 0.0f, 406.0f }, { 472.0f, 330.0f, 406.0f }, { 314.0f, 330.0f, 456.0f }, { 314.0f, 0.0f, 456.0f },
    { 314.0f, 0.0f, 456.0f }, { 314.0f, 330.0f, 456.0f }, { 265.0f, 330.0f, 296.0f }, { 265.0f, 0.0f, 296.0f },
    { 265.0f, 0.0f, 296.0f }, { 265.0f, 330.0f, 296.0f }, { 423.0f, 330.0f, 247.0f }, { 423.0f, 0.0f, 247.0f },
};

/* mkCornellBoxVerticesTex, CornellBox.hs:23-38: per quad (q0,q1,q3),(q3,q1,q2);
 * vertex = (v / toUnit - 1) ^* scale in Float */
void orc_cornell_vertices(float out[96 * 3])
{
    const float to_unit = 559.2f / 2.0f;
    const float scale = 1.0f / (sqrtf(2.0f * 2.0f + 2.0f * 2.0f + 2.0f * 2.0f) / 2.0f) * 0.99f;
    static const int order[6] = { 0, 1, 3, 3, 1, 2 };
    for (int q = 0; q < 16; q++)
        for (int k = 0; k < 6; k++) {
            const float *v = cornell_quads[q * 4 + order[k]];
            for (int a = 0; a < 3; a++) {
                float c = v[a];
                if (q == 5 && a == 1) c = 548.8f - 0.1f;   /* light quad, :81-84 */
                out[(q * 6 + k) * 3 + a] = (c / to_unit - 1.0f) * scale;
            }
        }
}

static const float *cornell_table(void)
{
    static float tab[96 * 3];
    static int init = 0;
    if (!init) { orc_cornell_vertices(tab); __sync_synchronize(); init = 1; }
    return tab;
}

float orc_de(int scene, float time, const float pos[3])
{
    de_ctx c = { scene, time, cornell_table(), 0, 0, 0, general_power(time) };
    return distance_estimator(V3(pos[0], pos[1], pos[2]), &c);
}

/* ============================================================================
 * 3. Ray set-up, marching, normals, AO, Fresnel (fragment.shd:463-470,542-724)
 * ==========================================================================*/

/* fragment.shd:595-616 with spherePos = 0 */
static int ray_sphere(v3 origin, v3 dir, float R, float *tmin, float *tmax)
{
    v3 rs = V3(0.0f - origin.x, 0.0f - origin.y, 0.0f - origin.z);
    float t = rm_dot(dir, rs);
    float a = rm_dot(rs, rs) - t * t;
    float r2 = R * R;
    if (a > r2) return 0;
    float h = sqrtf(r2 - a);
    *tmin = t - h;
    *tmax = t + h;
    return 1;
}

int orc_ray_sphere(const float o[3], const float d[3], float R, float *tmin, float *tmax)
{
    return ray_sphere(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), R, tmin, tmax);
}

static float scene_bsphere(int scene)
{
    /* fragment.shd:640-649 */
    switch (scene) {
    case ORC_SCENE_MB_POWER8:  return K_bsphere_r_power8;
    case ORC_SCENE_MB_GENERAL: return K_bsphere_r_general;
    default:                   return K_bsphere_r_other;
    }
}

/* fragment.shd:618-676; max_steps is a parameter (the reference hard-codes 128).
 * *entered: the ray intersected the bounding sphere; *steps: loop counter at exit. */
static int ray_march(v3 origin, v3 dir, int max_steps, de_ctx *c, float *t_out, int *steps_out, int *entered,
                     uint64_t *march_steps)
{
    const float MIN_DIST = K_march_min_dist;
    float tmin, tmax;
    *steps_out = 0;
    *entered = 0;
    if (!ray_sphere(origin, dir, scene_bsphere(c->scene), &tmin, &tmax)) return 0;
    *entered = 1;
    float t = rm_max(0.0f, tmin);
    int steps;
    for (steps = 0; steps < max_steps; steps++) {
        v3 pos = V3(origin.x + t * dir.x, origin.y + t * dir.y, origin.z + t * dir.z);
        float dist = distance_estimator(pos, c);
        (*march_steps)++;
        t += dist;
        if (t > tmax) { *steps_out = steps; return 0; }
        if (dist < MIN_DIST) { *steps_out = steps; *t_out = t; return 1; }
    }
    *steps_out = steps;
    return 0;
}

/* fragment.shd:463-470 */
static v3 normal_backward_difference(v3 pos, de_ctx *c)
{
    const float eps = K_normal_eps;
    float d0 = distance_estimator(pos, c);
    float dx = distance_estimator(V3(pos.x - eps, pos.y - 0.0f, pos.z - 0.0f), c);
    float dy = distance_estimator(V3(pos.x - 0.0f, pos.y - eps, pos.z - 0.0f), c);
    float dz = distance_estimator(V3(pos.x - 0.0f, pos.y - 0.0f, pos.z - eps), c);
    return rm_normalize(V3(d0 - dx, d0 - dy, d0 - dz));
}

static float ao_tap(v3 p, v3 n, float delta, de_ctx *c)
{
    float d = distance_estimator(V3(p.x + n.x * delta, p.y + n.y * delta, p.z + n.z * delta), c);
    return rm_clamp(1.0f - d / delta, 0.0f, 1.0f);
}

/* fragment.shd:542-591 */
static float distance_ao(v3 p, v3 n, de_ctx *c)
{
    float occl_sum = 0.0f;
    if (c->scene != ORC_SCENE_CORNELL) {
        occl_sum += K_ao_w0 * ao_tap(p, n, K_ao_d0, c);
        occl_sum += K_ao_w1 * ao_tap(p, n, K_ao_d1, c);
        occl_sum = 1.0f - occl_sum;
        occl_sum -= K_ao_bias;
        occl_sum *= K_ao_gain;
        occl_sum *= occl_sum;
        occl_sum = rm_clamp(occl_sum, 0.0f, 1.0f);
        return occl_sum;
    } else {
        occl_sum += K_cornell_ao_w0 * ao_tap(p, n, K_cornell_ao_d0, c);
        occl_sum += K_cornell_ao_w1 * ao_tap(p, n, K_cornell_ao_d1, c);
        occl_sum += K_cornell_ao_w2 * ao_tap(p, n, K_cornell_ao_d2, c);
        occl_sum += K_cornell_ao_w3 * ao_tap(p, n, K_cornell_ao_d3, c);
        occl_sum = 1.0f - occl_sum;
        return occl_sum;
    }
}

/* fragment.shd:694-719 */
static float fresnel_conductor(float cosi, float eta, float k)
{
    float tmp = (eta * eta + k * k) * cosi * cosi;
    float r_parallel_2 = (tmp - (2.0f * eta * cosi) + 1.0f) / (tmp + (2.0f * eta * cosi) + 1.0f);
    float tmp_f = eta * eta + k * k;
    float r_perpend_2 = (tmp_f - (2.0f * eta * cosi) + cosi * cosi) / (tmp_f + (2.0f * eta * cosi) + cosi * cosi);
    return (r_parallel_2 + r_perpend_2) / 2.0f;
}

float orc_fresnel_conductor(float cosi, float eta, float k) { return fresnel_conductor(cosi, eta, k); }

/* ============================================================================
 * 4. samplerCube semantics: RGB16F texels,
 *    min filter NEAREST / mag filter LINEAR (GLHelpers.hs:105-106), seamless
 *    edges (HDREnvMap.hs:126).
 * ==========================================================================*/

uint16_t orc_f32_to_f16(float f)
{
    /* round-to-nearest-even, overflow -> inf, subnormals supported */
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t em = x & 0x7fffffffu;
    if (em >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((em > 0x7f800000u) ? 0x200u : 0u));
    if (em >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);      /* >= 65520 -> inf */
    if (em < 0x38800000u) {                                          /* < 2^-14: subnormal half */
        if (em < 0x33000000u) return (uint16_t)sign;                 /* < 2^-25 -> 0 */
        int e = (int)(em >> 23);
        uint32_t m = (em & 0x7fffffu) | 0x800000u;
        int shift = 126 - e;                                         /* 14..24 */
        uint32_t hm = m >> shift;
        uint32_t rem = m & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (hm & 1u))) hm++;
        return (uint16_t)(sign | hm);
    }
    uint32_t h = ((em - 0x38000000u) >> 13);
    uint32_t rem = em & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) h++;
    return (uint16_t)(sign | h);
}

float orc_f16_to_f32(uint16_t hb)
{
    uint32_t sign = ((uint32_t)hb & 0x8000u) << 16;
    uint32_t e = (hb >> 10) & 0x1fu;
    uint32_t m = hb & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        float v = (float)m * 5.9604644775390625e-08f;  /* 2^-24 */
        return (sign ? -v : v);
    }
    if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}

/* GL face selection (Appendix A1); ties x before y before z; NaN falls to z */
static void cube_coords(v3 r, int *face, float *sc, float *tc, float *ma)
{
    float ax = fabsf(r.x), ay = fabsf(r.y), az = fabsf(r.z);
    if (ax >= ay && ax >= az) {
        if (r.x > 0.0f) { *face = 0; *sc = -r.z; *tc = -r.y; *ma = r.x; }
        else            { *face = 1; *sc =  r.z; *tc = -r.y; *ma = r.x; }
    } else if (ay >= az) {
        if (r.y > 0.0f) { *face = 2; *sc =  r.x; *tc =  r.z; *ma = r.y; }
        else            { *face = 3; *sc =  r.x; *tc = -r.z; *ma = r.y; }
    } else {
        if (r.z > 0.0f) { *face = 4; *sc =  r.x; *tc = -r.y; *ma = r.z; }
        else            { *face = 5; *sc = -r.x; *tc = -r.y; *ma = r.z; }
    }
}

/* coordinates of an arbitrary direction q on the plane of a given face */
static void cube_project(int face, v3 q, float *sc, float *tc, float *ma)
{
    switch (face) {
    case 0:  *sc = -q.z; *tc = -q.y; *ma = q.x; break;
    case 1:  *sc =  q.z; *tc = -q.y; *ma = q.x; break;
    case 2:  *sc =  q.x; *tc =  q.z; *ma = q.y; break;
    case 3:  *sc =  q.x; *tc = -q.z; *ma = q.y; break;
    case 4:  *sc =  q.x; *tc = -q.y; *ma = q.z; break;
    default: *sc = -q.x; *tc = -q.y; *ma = q.z; break;
    }
}

static inline float cube_texcoord(float c, float ama, float W) { return (0.5f * (c / ama + 1.0f)) * W; }

static inline const uint16_t *cube_texel(const orc_cube *c, int face, int X, int Y)
{
    int P = c->W + 2;
    return c->padded + ((size_t)(face * P + Y) * P + X) * 4;
}

static void cube_fetch_nearest(const orc_cube *c, int face, float u, float v, float rgb[3])
{
    float Wm1 = (float)(c->W - 1);
    float fi = floorf(u), fj = floorf(v);
    if (!(fi >= 0.0f)) fi = 0.0f;
    if (fi > Wm1) fi = Wm1;
    if (!(fj >= 0.0f)) fj = 0.0f;
    if (fj > Wm1) fj = Wm1;
    const uint16_t *t = cube_texel(c, face, (int)fi + 1, (int)fj + 1);
    rgb[0] = orc_f16_to_f32(t[0]); rgb[1] = orc_f16_to_f32(t[1]); rgb[2] = orc_f16_to_f32(t[2]);
}

static void cube_fetch_linear(const orc_cube *c, int face, float u, float v, float rgb[3])
{
    float Wm1 = (float)(c->W - 1);
    float ub = u - 0.5f, vb = v - 0.5f;
    float fi = floorf(ub), fj = floorf(vb);
    if (!(fi >= -1.0f)) fi = -1.0f;
    if (fi > Wm1) fi = Wm1;
    if (!(fj >= -1.0f)) fj = -1.0f;
    if (fj > Wm1) fj = Wm1;
    float fu = ub - fi, fv = vb - fj;
    float gu = 1.0f - fu, gv = 1.0f - fv;
    int X = (int)fi + 1, Y = (int)fj + 1;
    const uint16_t *t00 = cube_texel(c, face, X, Y), *t10 = cube_texel(c, face, X + 1, Y);
    const uint16_t *t01 = cube_texel(c, face, X, Y + 1), *t11 = cube_texel(c, face, X + 1, Y + 1);
    for (int k = 0; k < 3; k++) {
        float top = orc_f16_to_f32(t00[k]) * gu + orc_f16_to_f32(t10[k]) * fu;
        float bot = orc_f16_to_f32(t01[k]) * gu + orc_f16_to_f32(t11[k]) * fu;
        rgb[k] = top * gv + bot * fv;
    }
}

void orc_cube_sample(const orc_cube *c, const float dir[3], int linear, float rgb[3])
{
    int face; float sc, tc, ma;
    cube_coords(V3(dir[0], dir[1], dir[2]), &face, &sc, &tc, &ma);
    float W = (float)c->W, ama = fabsf(ma);
    float u = cube_texcoord(sc, ama, W), v = cube_texcoord(tc, ama, W);
    if (linear) cube_fetch_linear(c, face, u, v, rgb); else cube_fetch_nearest(c, face, u, v, rgb);
}

/* texture(samplerCube, r) inside a 2x2 fragment quad: rh / rv are the same
 * expression evaluated by the horizontal / vertical quad neighbour (valid_* = 0
 * when that neighbour did not execute the expression).  The neighbours'
 * directions are projected onto THIS pixel's face; rho^2 <= 1 on both axes ->
 * magnified -> LINEAR, otherwise (or if a neighbour is unusable) NEAREST. */
static void cube_texture(const orc_cube *c, v3 r, int valid_h, v3 rh, int valid_v, v3 rv, float rgb[3])
{
    int face; float sc, tc, ma;
    cube_coords(r, &face, &sc, &tc, &ma);
    float W = (float)c->W, ama = fabsf(ma);
    float u = cube_texcoord(sc, ama, W), v = cube_texcoord(tc, ama, W);
    int linear = 0;
    if (valid_h && valid_v) {
        float sh, th, mh, sv, tv, mv;
        cube_project(face, rh, &sh, &th, &mh);
        cube_project(face, rv, &sv, &tv, &mv);
        int pos = (face & 1) == 0;
        int okh = pos ? (mh > 0.0f) : (mh < 0.0f);
        int okv = pos ? (mv > 0.0f) : (mv < 0.0f);
        if (okh && okv) {
            float amh = fabsf(mh), amv = fabsf(mv);
            float dux = cube_texcoord(sh, amh, W) - u, dvx = cube_texcoord(th, amh, W) - v;
            float duy = cube_texcoord(sv, amv, W) - u, dvy = cube_texcoord(tv, amv, W) - v;
            float rx = dux * dux + dvx * dvx;
            float ry = duy * duy + dvy * dvy;
            linear = (rx <= 1.0f) && (ry <= 1.0f);
        }
    }
    if (linear) cube_fetch_linear(c, face, u, v, rgb); else cube_fetch_nearest(c, face, u, v, rgb);
}

/* integer cube direction of texel centre (x,y) on a face, in units of 1/W,
 * with cw = 2x+1-W, ch = 2y+1-W  (cubeMapPixelToDir, HDREnvMap.hs:82-87) */
static void cube_int_dir(int face, int W, int cw, int ch, int d[3])
{
    switch (face) {
    case 0:  d[0] =  W;  d[1] = -ch; d[2] = -cw; break;
    case 1:  d[0] = -W;  d[1] = -ch; d[2] =  cw; break;
    case 2:  d[0] =  cw; d[1] =  W;  d[2] =  ch; break;
    case 3:  d[0] =  cw; d[1] = -W;  d[2] = -ch; break;
    case 4:  d[0] =  cw; d[1] = -ch; d[2] =  W;  break;
    default: d[0] = -cw; d[1] = -ch; d[2] = -W;  break;
    }
}

/* texel (x,y) of `face` with exactly one coordinate out of range by one ->
 * the texel that is adjacent across the cube edge */
static void cube_fold(int face, int W, int x, int y, int *nf, int *nx, int *ny)
{
    int d[3];
    cube_int_dir(face, W, 2 * x + 1 - W, 2 * y + 1 - W, d);
    int major = face >> 1;
    for (int a = 0; a < 3; a++) {
        if (a == major) continue;
        if (d[a] > W)  { d[a] =  W; d[major] = (d[major] > 0) ? W - 1 : -(W - 1); major = a; break; }
        if (d[a] < -W) { d[a] = -W; d[major] = (d[major] > 0) ? W - 1 : -(W - 1); major = a; break; }
    }
    int f = major * 2 + (d[major] > 0 ? 0 : 1);
    int cw, ch;
    switch (f) {
    case 0:  ch = -d[1]; cw = -d[2]; break;
    case 1:  ch = -d[1]; cw =  d[2]; break;
    case 2:  cw =  d[0]; ch =  d[2]; break;
    case 3:  cw =  d[0]; ch = -d[2]; break;
    case 4:  cw =  d[0]; ch = -d[1]; break;
    default: cw = -d[0]; ch = -d[1]; break;
    }
    *nf = f; *nx = (cw + W - 1) / 2; *ny = (ch + W - 1) / 2;
}

/* f32 faces (6*cw*cw*3) -> RGB16F (RNE) with a seamless one-texel border.
 * Corner border texels = RNE16(((a+b)+c)/3) of the three texels that meet at
 * the cube corner (GL 3.3 core, seamless cube map filtering). */
void orc_cube_pad_f16(const float *faces, int W, uint16_t *padded)
{
    int P = W + 2;
#define SRC(f, x, y, k) orc_f32_to_f16(faces[(((size_t)(f) * W + (y)) * W + (x)) * 3 + (k)])
    for (int f = 0; f < 6; f++)
        for (int Y = 0; Y < P; Y++)
            for (int X = 0; X < P; X++) {
                int x = X - 1, y = Y - 1;
                int ox = (x < 0 || x >= W), oy = (y < 0 || y >= W);
                uint16_t *dst = padded + ((size_t)(f * P + Y) * P + X) * 4;
                dst[3] = 0;
                if (!ox && !oy) {
                    for (int k = 0; k < 3; k++) dst[k] = SRC(f, x, y, k);
                } else if (ox != oy) {
                    int nf, nx, ny;
                    cube_fold(f, W, x, y, &nf, &nx, &ny);
                    for (int k = 0; k < 3; k++) dst[k] = SRC(nf, nx, ny, k);
                } else {
                    int cx = x < 0 ? 0 : W - 1, cy = y < 0 ? 0 : W - 1;
                    int f1, x1, y1, f2, x2, y2;
                    cube_fold(f, W, x, cy, &f1, &x1, &y1);
                    cube_fold(f, W, cx, y, &f2, &x2, &y2);
                    for (int k = 0; k < 3; k++) {
                        float a = orc_f16_to_f32(SRC(f, cx, cy, k));
                        float b = orc_f16_to_f32(SRC(f1, x1, y1, k));
                        float c = orc_f16_to_f32(SRC(f2, x2, y2, k));
                        dst[k] = orc_f32_to_f16(((a + b) + c) / 3.0f);
                    }
                }
            }
#undef SRC
}

/* ============================================================================
 * 5. Camera, per-pixel pipeline (fragment.shd:726-966)
 * ==========================================================================*/

float orc_fov_xs(void)
{
    /* radians(45.0 * 1.5), tan(hfov / 2)  -- fragment.shd:866-867,910 */
    float hfov = (K_hfov_deg_a * K_hfov_deg_b) * 0.017453292519943295f;
    return tanf(hfov / 2.0f);
}

void orc_camera(int scene, float time, float out[12])
{
    v3 cam;
    if (scene == ORC_SCENE_CORNELL) {
        /* fragment.shd:888-890 */
        cam = V3(sinf(time / 2.0f) * K_camera_cornell_radius, cosf(time / 2.0f) * K_camera_cornell_radius, K_camera_cornell_z);
    } else {
        /* fragment.shd:892-897 */
        cam = V3(sinf(time / 3.0f), cosf(time / 4.0f), cosf(time / 3.0f));
        cam = rm_scale(rm_normalize(cam), K_camera_distance);
    }
    /* lookat(cam, 0, (0,1,0)), fragment.shd:829-838 */
    v3 zaxis = rm_normalize(rm_sub(cam, V3(0.0f, 0.0f, 0.0f)));
    v3 xaxis = rm_normalize(rm_cross(V3(0.0f, 1.0f, 0.0f), zaxis));
    v3 yaxis = rm_cross(zaxis, xaxis);
    out[0] = xaxis.x; out[1] = xaxis.y; out[2] = xaxis.z;
    out[3] = yaxis.x; out[4] = yaxis.y; out[5] = yaxis.z;
    out[6] = zaxis.x; out[7] = zaxis.y; out[8] = zaxis.z;
    out[9] = cam.x;   out[10] = cam.y;  out[11] = cam.z;
}

typedef struct {
    v3       dir;
    int      hit, steps, entered;
    unsigned iters, iters_march;
    v3       n, refl;
    float    ao, fresnel;
} px_state;

typedef struct {
    const orc_frame *f;
    int      x0, y0, x1, y1;
    float   *rgba_f32;
    uint32_t *rgba8;
    uint16_t *steps, *iters, *iters_march;
    float    cam[12];
    float    fov_xs;
    int      qrow_lo, qrow_hi;      /* quad rows [lo,hi), in units of 2 pixel rows */
    orc_counters ctr;
} render_job;

/* generate_ray, perspective branch, sample_offs = 0 (fragment.shd:840-871) */
static v3 generate_ray_dir(const render_job *j, int px, int py)
{
    float wf = (float)j->f->w, hf = (float)j->f->h;
    float ndcx = ((float)px + 0.5f) / wf * 2.0f - 1.0f;
    float ndcy = ((float)py + 0.5f) / hf * 2.0f - 1.0f;
    float aspect = wf / hf;
    v3 d = rm_normalize(V3(ndcx * j->fov_xs, ndcy * j->fov_xs / aspect, -1.0f));
    const float *c = j->cam;
    /* mat3(camera) * d */
    return V3(c[0] * d.x + c[3] * d.y + c[6] * d.z,
              c[1] * d.x + c[4] * d.y + c[7] * d.z,
              c[2] * d.x + c[5] * d.y + c[8] * d.z);
}

static void trace_pixel(render_job *j, int px, int py, px_state *s)
{
    const orc_frame *f = j->f;
    de_ctx c = { f->scene, f->time, cornell_table(), 0, 0, 0, general_power(f->time) };
    v3 origin = V3(j->cam[9], j->cam[10], j->cam[11]);
    uint64_t march_steps = 0;
    float t = 0.0f;
    memset(s, 0, sizeof *s);
    s->dir = generate_ray_dir(j, px, py);
    s->hit = ray_march(origin, s->dir, f->max_steps, &c, &t, &s->steps, &s->entered, &march_steps);
    s->iters_march = (unsigned)c.triplex_iters;      /* iterations spent inside ray_march alone (normal / AO taps follow) */
    if (s->hit) {
        /* render_ray, fragment.shd:743-799 */
        v3 isec = V3(origin.x + s->dir.x * t, origin.y + s->dir.y * t, origin.z + s->dir.z * t);
        v3 np = V3(isec.x - s->dir.x * K_isec_step_back, isec.y - s->dir.y * K_isec_step_back, isec.z - s->dir.z * K_isec_step_back);
        s->n = normal_backward_difference(np, &c);
        s->ao = distance_ao(isec, s->n, &c);
        s->fresnel = fresnel_conductor(rm_dot(rm_neg(s->dir), s->n), K_fresnel_eta, K_fresnel_k);
        s->refl = rm_reflect(s->dir, s->n);
    }
    s->iters = (unsigned)c.triplex_iters;
    j->ctr.de_evals += c.de_evals;
    j->ctr.triplex_iters += c.triplex_iters;
    j->ctr.tri_inside += c.tri_inside;
    j->ctr.march_steps += march_steps;
    j->ctr.hit_pixels += (uint64_t)s->hit;
    j->ctr.sphere_pixels += (uint64_t)s->entered;
    j->ctr.pixels++;
}

static void shade_pixel(const render_job *j, const px_state q[4], int k, float rgb[3])
{
    const orc_frame *f = j->f;
    const px_state *s = &q[k], *sh = &q[k ^ 1], *sv = &q[k ^ 2];
    if (s->hit) {
        /* fragment.shd:799-810; neighbours count only if they took the hit branch too */
        float t1[3], t8[3], tr[3];
        cube_texture(&f->env_cos_1, s->n, sh->hit, sh->n, sv->hit, sv->n, t1);
        cube_texture(&f->env_cos_8, s->refl, sh->hit, sh->refl, sv->hit, sv->refl, t8);
        cube_texture(&f->env_reflection, s->refl, sh->hit, sh->refl, sv->hit, sv->refl, tr);
        const float diff_col[3] = { K_diff_r, K_diff_g, K_diff_b }, spec_col[3] = { K_spec_r, K_spec_g, K_spec_b };
        const float diff_weight = K_diff_weight, spec_weight = K_spec_weight_one_minus - K_diff_weight;
        const float npl = (K_phong_lobe_n + K_phong_lobe_plus) / K_phong_lobe_div;
        for (int c = 0; c < 3; c++)
            rgb[c] = (t1[c] * diff_col[c] * diff_weight
                      + t8[c] * spec_col[c] * npl * s->fresnel * spec_weight
                      + tr[c] * spec_weight * s->fresnel * K_refl_weight) * K_exposure * s->ao;
    } else {
        /* fragment.shd:823.  The lookup sits in the miss branch: a quad neighbour that took the hit
         * branch leaves the derivative undefined (GLSL non-uniform control flow) -> pinned as minified,
         * which is also what the reference shader does on SwiftShader (tests/test_oracle_vs_glsl.py) */
        cube_texture(&f->env_reflection, s->dir, !sh->hit, sh->dir, !sv->hit, sv->dir, rgb);
    }
}

static inline uint32_t to_unorm8(float g)
{
    if (!(g == g)) return 0u;                          /* NaN -> 0 (pinned) */
    return (uint32_t)rintf(rm_clamp(g, 0.0f, 1.0f) * 255.0f);  /* round-half-even */
}

static void *render_worker(void *arg)
{
    render_job *j = (render_job *)arg;
    const orc_frame *f = j->f;
    int qx_lo = j->x0 >> 1, qx_hi = (j->x1 + 1) >> 1;
    for (int qy = j->qrow_lo; qy < j->qrow_hi; qy++)
        for (int qx = qx_lo; qx < qx_hi; qx++) {
            px_state q[4];
            for (int k = 0; k < 4; k++) trace_pixel(j, qx * 2 + (k & 1), qy * 2 + (k >> 1), &q[k]);
            for (int k = 0; k < 4; k++) {
                int px = qx * 2 + (k & 1), py = qy * 2 + (k >> 1);
                if (px < j->x0 || px >= j->x1 || py < j->y0 || py >= j->y1) continue;
                float rgb[3];
                shade_pixel(j, q, k, rgb);
                size_t idx = (size_t)px + (size_t)py * f->w;
                /* fragment.shd:959-960: pow(color, 1/2.2), alpha 1 */
                float g[3];
                for (int c = 0; c < 3; c++) g[c] = rm_powf(rgb[c], 1.0f / K_gamma);
                if (j->rgba_f32) {
                    j->rgba_f32[idx * 4 + 0] = g[0]; j->rgba_f32[idx * 4 + 1] = g[1];
                    j->rgba_f32[idx * 4 + 2] = g[2]; j->rgba_f32[idx * 4 + 3] = 1.0f;
                }
                if (j->rgba8)
                    j->rgba8[idx] = to_unorm8(g[0]) | (to_unorm8(g[1]) << 8) | (to_unorm8(g[2]) << 16) | 0xff000000u;
                if (j->steps) j->steps[idx] = (uint16_t)(q[k].steps | (q[k].hit << 15));
                if (j->iters) j->iters[idx] = (uint16_t)(q[k].iters > 65535u ? 65535u : q[k].iters);
                if (j->iters_march) j->iters_march[idx] = (uint16_t)(q[k].iters_march > 65535u ? 65535u : q[k].iters_march);
            }
        }
    return NULL;
}

/* Shading alone (fragment.shd:788-823, 956-960): the colour of every pixel of the frame given, per pixel, the hit flag and
 * the (normal, ao) pair the hit branch computed -- e.g. as exported from the reference shader itself run on SwiftShader
 * (tests/golden/make_swiftshader_vectors.py, "gbuffer" program).  Everything downstream of the normal is done here exactly as
 * orc_render does it: ray direction, Fresnel, reflect, the three cube-map lookups with the quad filter rule, gamma. */
int orc_shade_gbuffer(const orc_frame *f, const float *nao /* w*h*4 */, const uint8_t *hit /* w*h */, float *rgba_f32)
{
    if (!f || !nao || !hit || !rgba_f32 || f->w <= 0 || f->h <= 0 || (f->w & 1) || (f->h & 1)) return -1;
    render_job j;
    memset(&j, 0, sizeof j);
    j.f = f;
    orc_camera(f->scene, f->time, j.cam);
    j.fov_xs = orc_fov_xs();
    for (int qy = 0; qy < f->h / 2; qy++)
        for (int qx = 0; qx < f->w / 2; qx++) {
            px_state q[4];
            for (int k = 0; k < 4; k++) {
                const int px = qx * 2 + (k & 1), py = qy * 2 + (k >> 1);
                const size_t idx = (size_t)px + (size_t)py * f->w;
                memset(&q[k], 0, sizeof q[k]);
                q[k].dir = generate_ray_dir(&j, px, py);
                q[k].hit = hit[idx] != 0;
                if (q[k].hit) {
                    q[k].n = V3(nao[idx * 4], nao[idx * 4 + 1], nao[idx * 4 + 2]);
                    q[k].ao = nao[idx * 4 + 3];
                    q[k].fresnel = fresnel_conductor(rm_dot(rm_neg(q[k].dir), q[k].n), K_fresnel_eta, K_fresnel_k);
                    q[k].refl = rm_reflect(q[k].dir, q[k].n);
                }
            }
            for (int k = 0; k < 4; k++) {
                const int px = qx * 2 + (k & 1), py = qy * 2 + (k >> 1);
                const size_t idx = (size_t)px + (size_t)py * f->w;
                float rgb[3];
                shade_pixel(&j, q, k, rgb);
                for (int c = 0; c < 3; c++) rgba_f32[idx * 4 + c] = rm_powf(rgb[c], 1.0f / K_gamma);
                rgba_f32[idx * 4 + 3] = 1.0f;
            }
        }
    return 0;
}

int orc_num_processors(void)
{
    long n = sysconf(_SC_NPROCESSORS_ONLN);
    return n < 1 ? 1 : (int)n;
}

/* ConcurrentSegments.hs:14-23 */
int orc_make_n_segments(int nseg, int low, int high, int *out)
{
    if (low >= high) return 0;
    int nsegc = nseg < (high - low) ? nseg : (high - low);
    if (nsegc <= 0) return 0;
    if (nsegc == 1) { out[0] = low; out[1] = high; return 1; }
    int segl = (high - low) / nsegc;
    for (int i = 0; i < nsegc - 1; i++) { out[2 * i] = low + i * segl; out[2 * i + 1] = low + (i + 1) * segl; }
    out[2 * (nsegc - 1)] = low + (nsegc - 1) * segl;
    out[2 * (nsegc - 1) + 1] = high;
    return nsegc;
}

typedef void *(*worker_fn)(void *);

/* forSegmentsConcurrently (ConcurrentSegments.hs:25-28) over pthreads */
static void run_segments(int nthreads, int low, int high, worker_fn fn, void *jobs, size_t job_size,
                         void (*set_range)(void *job, int lo, int hi))
{
    if (nthreads <= 0) nthreads = orc_num_processors();
    if (nthreads > 256) nthreads = 256;
    int seg[512];
    int n = orc_make_n_segments(nthreads, low, high, seg);
    pthread_t th[256];
    for (int i = 0; i < n; i++) {
        void *job = (char *)jobs + (size_t)i * job_size;
        if (i > 0) memcpy(job, jobs, job_size);
    }
    for (int i = 0; i < n; i++) set_range((char *)jobs + (size_t)i * job_size, seg[2 * i], seg[2 * i + 1]);
    for (int i = 1; i < n; i++) pthread_create(&th[i], NULL, fn, (char *)jobs + (size_t)i * job_size);
    if (n > 0) fn(jobs);
    for (int i = 1; i < n; i++) pthread_join(th[i], NULL);
}

static void render_set_range(void *job, int lo, int hi)
{
    render_job *j = (render_job *)job;
    j->qrow_lo = lo; j->qrow_hi = hi;
}

int orc_render(const orc_frame *f, int x0, int y0, int x1, int y1, float *rgba_f32, uint32_t *rgba8,
               uint16_t *steps, uint16_t *iters, orc_counters *ctr, int nthreads)
{
    return orc_render_ex(f, x0, y0, x1, y1, rgba_f32, rgba8, steps, iters, NULL, ctr, nthreads);
}

int orc_render_ex(const orc_frame *f, int x0, int y0, int x1, int y1, float *rgba_f32, uint32_t *rgba8,
                  uint16_t *steps, uint16_t *iters, uint16_t *iters_march, orc_counters *ctr, int nthreads)
{
    if (!f || f->w <= 0 || f->h <= 0 || f->max_steps < 0 || f->max_steps > 32767) return -1;
    if (f->scene < 0 || f->scene > 3) return -2;
    if (x0 < 0 || y0 < 0 || x1 > f->w || y1 > f->h || x0 > x1 || y0 > y1) return -3;
    render_job *jobs = (render_job *)calloc(256, sizeof(render_job));
    if (!jobs) return -4;
    jobs[0].f = f;
    jobs[0].x0 = x0; jobs[0].y0 = y0; jobs[0].x1 = x1; jobs[0].y1 = y1;
    jobs[0].rgba_f32 = rgba_f32; jobs[0].rgba8 = rgba8; jobs[0].steps = steps; jobs[0].iters = iters; jobs[0].iters_march = iters_march;
    orc_camera(f->scene, f->time, jobs[0].cam);
    jobs[0].fov_xs = orc_fov_xs();
    (void)cornell_table();
    int nt = nthreads <= 0 ? orc_num_processors() : nthreads;
    if (nt > 256) nt = 256;
    int qlo = y0 >> 1, qhi = (y1 + 1) >> 1;
    run_segments(nt, qlo, qhi, render_worker, jobs, sizeof(render_job), render_set_range);
    if (ctr) {
        memset(ctr, 0, sizeof *ctr);
        int seg[512];
        int n = orc_make_n_segments(nt, qlo, qhi, seg);
        for (int i = 0; i < n; i++) {
            ctr->de_evals += jobs[i].ctr.de_evals;
            ctr->triplex_iters += jobs[i].ctr.triplex_iters;
            ctr->tri_inside += jobs[i].ctr.tri_inside;
            ctr->march_steps += jobs[i].ctr.march_steps;
            ctr->hit_pixels += jobs[i].ctr.hit_pixels;
            ctr->sphere_pixels += jobs[i].ctr.sphere_pixels;
            ctr->pixels += jobs[i].ctr.pixels;
        }
    }
    free(jobs);
    return 0;
}

/* ============================================================================
 * 6. Radiance RGBE (third-party JuicyPixels arithmetic -- unpinned, restated
 *    from Ward's colr/color conversions which JuicyPixels mirrors:
 *    decode (m + 0.5) * 2^(e-136); encode significand(d)*255.9999/d, truncate;
 *    max component <= 1e-32 -> (0,0,0,0)).  HDREnvMap.hs:33, ShaderRendering.hs:147
 * ==========================================================================*/

void orc_rgbe_decode(const uint8_t *rgbe, long npix, float *rgb)
{
    for (long i = 0; i < npix; i++) {
        const uint8_t *p = rgbe + i * 4;
        float f = ldexpf(1.0f, (int)p[3] - (128 + 8));
        rgb[i * 3 + 0] = ((float)p[0] + 0.5f) * f;
        rgb[i * 3 + 1] = ((float)p[1] + 0.5f) * f;
        rgb[i * 3 + 2] = ((float)p[2] + 0.5f) * f;
    }
}

void orc_rgbe_encode(const float *rgb, long npix, uint8_t *rgbe)
{
    for (long i = 0; i < npix; i++) {
        float r = rgb[i * 3], g = rgb[i * 3 + 1], b = rgb[i * 3 + 2];
        float d = r;
        if (g > d) d = g;
        if (b > d) d = b;
        uint8_t *p = rgbe + i * 4;
        if (!(d > 1e-32f)) { p[0] = p[1] = p[2] = p[3] = 0; continue; }
        int e;
        float sig = frexpf(d, &e);
        float coeff = sig * 255.9999f / d;
        p[0] = (uint8_t)(int)(r * coeff);
        p[1] = (uint8_t)(int)(g * coeff);
        p[2] = (uint8_t)(int)(b * coeff);
        p[3] = (uint8_t)(e + 128);
    }
}

static long hdr_find_resolution(const uint8_t *file, long len, int *w, int *h)
{
    /* header: lines until an empty line, then "-Y <h> +X <w>\n" */
    long pos = 0;
    if (len < 11) return -1;
    while (pos < len) {
        long eol = pos;
        while (eol < len && file[eol] != '\n') eol++;
        if (eol >= len) return -1;
        if (eol == pos) { pos = eol + 1; break; }
        pos = eol + 1;
    }
    long eol = pos;
    while (eol < len && file[eol] != '\n') eol++;
    if (eol >= len) return -1;
    char line[128];
    long n = eol - pos;
    if (n <= 0 || n >= (long)sizeof line) return -1;
    memcpy(line, file + pos, (size_t)n);
    line[n] = 0;
    int hh = 0, ww = 0;
    if (sscanf(line, "-Y %d +X %d", &hh, &ww) != 2 || hh <= 0 || ww <= 0) return -1;
    *w = ww; *h = hh;
    return eol + 1;
}

int orc_hdr_decode(const uint8_t *file, long len, int *w, int *h, float *out)
{
    long pos = hdr_find_resolution(file, len, w, h);
    if (pos < 0) return -1;
    if (!out) return 0;
    int W = *w, H = *h;
    uint8_t *scan = (uint8_t *)malloc((size_t)W * 4);
    if (!scan) return -2;
    for (int y = 0; y < H; y++) {
        if (pos + 4 <= len && W >= 8 && W < 32768 && file[pos] == 2 && file[pos + 1] == 2 &&
            ((file[pos + 2] << 8) | file[pos + 3]) == W) {
            pos += 4;                                /* new-style RLE: 4 planes */
            for (int ch = 0; ch < 4; ch++) {
                int x = 0;
                while (x < W) {
                    if (pos >= len) { free(scan); return -3; }
                    int cnt = file[pos++];
                    if (cnt > 128) {
                        cnt -= 128;
                        if (pos >= len || x + cnt > W) { free(scan); return -3; }
                        uint8_t v = file[pos++];
                        for (int k = 0; k < cnt; k++) scan[(x++) * 4 + ch] = v;
                    } else {
                        if (cnt == 0 || pos + cnt > len || x + cnt > W) { free(scan); return -3; }
                        for (int k = 0; k < cnt; k++) scan[(x++) * 4 + ch] = file[pos++];
                    }
                }
            }
        } else {
            if (pos + (long)W * 4 > len) { free(scan); return -3; }
            memcpy(scan, file + pos, (size_t)W * 4);
            pos += (long)W * 4;
        }
        orc_rgbe_decode(scan, W, out + (size_t)y * W * 3);
    }
    free(scan);
    return 0;
}

long orc_hdr_encode(const float *rgb, int w, int h, uint8_t *file)
{
    char hdr[64];
    int n = 0;
    const char *magic = "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n";
    n = (int)strlen(magic);
    memcpy(hdr, magic, (size_t)n);
    char res[32];
    int m = 0;
    /* "-Y h +X w\n" without stdio */
    char tmp[16];
    res[m++] = '-'; res[m++] = 'Y'; res[m++] = ' ';
    int k = 0, v = h; do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v); while (k) res[m++] = tmp[--k];
    res[m++] = ' '; res[m++] = '+'; res[m++] = 'X'; res[m++] = ' ';
    k = 0; v = w; do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v); while (k) res[m++] = tmp[--k];
    res[m++] = '\n';
    memcpy(file, hdr, (size_t)n);
    memcpy(file + n, res, (size_t)m);
    orc_rgbe_encode(rgb, (long)w * h, file + n + m);
    return (long)n + m + (long)w * h * 4;
}

/* ============================================================================
 * 7. Lat/long -> cube map (HDREnvMap.hs:76-163, CoordTransf.hs:35-70)
 * ==========================================================================*/

static const float PI_F = 3.14159265358979323846f;

/* Linear.normalize (third-party `linear`, unpinned): unchanged when the squared
 * length is within 1e-6 of 0 or 1, else component-wise division by sqrt */
static v3 linear_normalize(v3 a)
{
    float l = a.x * a.x + a.y * a.y + a.z * a.z;
    if (fabsf(l) <= 1e-6f || fabsf(1.0f - l) <= 1e-6f) return a;
    float s = sqrtf(l);
    return V3(a.x / s, a.y / s, a.z / s);
}

/* HDREnvMap.hs:76-87 */
static v3 cube_pixel_to_dir(int face, int w, int x, int y)
{
    float vw = ((float)x + 0.5f) / (float)w * 2.0f - 1.0f;
    float vh = ((float)y + 0.5f) / (float)w * 2.0f - 1.0f;
    v3 d;
    switch (face) {
    case 0:  d = V3(1.0f, -vh, -vw); break;
    case 1:  d = V3(-1.0f, -vh, vw); break;
    case 2:  d = V3(vw, 1.0f, vh); break;
    case 3:  d = V3(vw, -1.0f, -vh); break;
    case 4:  d = V3(vw, -vh, 1.0f); break;
    default: d = V3(-vw, -vh, -1.0f); break;
    }
    return linear_normalize(d);
}

void orc_cube_pixel_to_dir(int face, int w, int x, int y, float dir[3])
{
    v3 d = cube_pixel_to_dir(face, w, x, y);
    dir[0] = d.x; dir[1] = d.y; dir[2] = d.z;
}

/* GHC's class-default atan2 for Float (GHC.Float, RealFloat default method) */
static float hs_atan2f(float y, float x)
{
    if (x > 0.0f) return atanf(y / x);
    if (x == 0.0f && y > 0.0f) return PI_F / 2.0f;
    if (x < 0.0f && y > 0.0f) return PI_F + atanf(y / x);
    if ((x <= 0.0f && y < 0.0f) || (x < 0.0f && y == 0.0f && signbit(y)) ||
        (x == 0.0f && signbit(x) && y == 0.0f && signbit(y)))
        return -hs_atan2f(-y, x);
    if (y == 0.0f && (x < 0.0f || (x == 0.0f && signbit(x)))) return PI_F;
    if (x == 0.0f && y == 0.0f) return y;
    return x + y;
}

/* CoordTransf.hs:35-44 after worldToLocal (46-50): local = (x, -z, y) */
static void cartesian_to_spherical_world(v3 world, float *theta, float *phi)
{
    v3 l = V3((world.x * 1.0f + world.y * 0.0f) + world.z * 0.0f,
              (world.x * 0.0f + world.y * 0.0f) + world.z * -1.0f,
              (world.x * 0.0f + world.y * 1.0f) + world.z * 0.0f);
    float cz = l.z;
    if (cz > 1.0f) cz = 1.0f;            /* max mi $ min ma v */
    if (cz < -1.0f) cz = -1.0f;
    *theta = acosf(cz);
    float p2 = hs_atan2f(l.y, l.x);
    float p1 = (p2 < 0.0f) ? p2 + 2.0f * PI_F : p2;
    *phi = (p1 == 2.0f * PI_F) ? 0.0f : p1;
}

/* CoordTransf.hs:60-70 */
static void spherical_to_env_uv(float theta, float phi, float *u, float *v)
{
    float p1 = phi + PI_F / 2.0f;
    float p2 = (p1 > 2.0f * PI_F) ? p1 - 2.0f * PI_F : p1;
    float p3 = 2.0f * PI_F - p2;
    *u = p3 / (PI_F * 2.0f);
    *v = theta / PI_F;
}

/* HDREnvMap.hs:91-113, including its `mod (w-1)` and min(h-1) quirks.
 * The reference fetches with unsafePixelAt (HDREnvMap.hs:106-109) and reads past the pixel vector when (u, v) leaves [0, 1] --
 * resizeHDRImage (169-195) does for maps that are not 2:1 (1024x510 -> 256: dsth = round(127.5) = 128, last tap row = source
 * row 511).  Undefined in the reference; pinned here (and in the device kernel) by clamping the integer texel into the image
 * before the fetch, weights unchanged.  Inside the image nothing changes. */
static void pixel_at_bilinear(const float *img, int w, int h, float u, float v, float rgb[3])
{
    float upx = u * ((float)w - 1.0f);
    float upy = v * ((float)h - 1.0f);
    int xf = (int)floorf(upx), yf = (int)floorf(upy);
    int x = xf < 0 ? 0 : (xf > w - 1 ? w - 1 : xf), y = yf < 0 ? 0 : (yf > h - 1 ? h - 1 : yf);
    int m = w - 1;
    int xp1 = (x + 1) % m;
    if (xp1 < 0) xp1 += m;                /* Haskell mod is floored */
    int yp1 = (y + 1 < h - 1) ? y + 1 : h - 1;
    float ur = upx - (float)xf, vr = upy - (float)yf;
    float uo = 1.0f - ur, vo = 1.0f - vr;
    const float *a = img + ((size_t)x + (size_t)y * w) * 3, *b = img + ((size_t)xp1 + (size_t)y * w) * 3;
    const float *c = img + ((size_t)x + (size_t)yp1 * w) * 3, *d = img + ((size_t)xp1 + (size_t)yp1 * w) * 3;
    for (int k = 0; k < 3; k++)
        rgb[k] = (a[k] * uo + b[k] * ur) * vo + (c[k] * uo + d[k] * ur) * vr;
}

void orc_pixel_at_bilinear(const float *img, int w, int h, float u, float v, float rgb[3])
{
    pixel_at_bilinear(img, w, h, u, v, rgb);
}

typedef struct { const float *latlong; int w, h, cw, face; float *faces; int lo, hi; } cube_job;

static void *cube_worker(void *arg)
{
    cube_job *j = (cube_job *)arg;
    for (int y = j->lo; y < j->hi; y++)
        for (int x = 0; x < j->cw; x++) {
            v3 dir = cube_pixel_to_dir(j->face, j->cw, x, y);
            float theta, phi, u, v;
            cartesian_to_spherical_world(dir, &theta, &phi);
            spherical_to_env_uv(theta, phi, &u, &v);
            pixel_at_bilinear(j->latlong, j->w, j->h, u, v,
                              j->faces + (((size_t)j->face * j->cw + y) * j->cw + x) * 3);
        }
    return NULL;
}

static void cube_set_range(void *job, int lo, int hi) { ((cube_job *)job)->lo = lo; ((cube_job *)job)->hi = hi; }

/* HDREnvMap.hs:118-163 (debugFaceColorize = False) */
void orc_latlong_to_cube(const float *latlong, int w, int h, float *faces, int nthreads)
{
    int cw = w / 3;
    cube_job *jobs = (cube_job *)calloc(256, sizeof(cube_job));
    for (int face = 0; face < 6; face++) {
        jobs[0].latlong = latlong; jobs[0].w = w; jobs[0].h = h; jobs[0].cw = cw; jobs[0].face = face;
        jobs[0].faces = faces;
        run_segments(nthreads, 0, cw, cube_worker, jobs, sizeof(cube_job), cube_set_range);
    }
    free(jobs);
}

/* buildTestLatLongEnvMap, HDREnvMap.hs:55-73 (environmentPxToSpherical,
 * CoordTransf.hs:80-91; localToWorld 52-58) */
void orc_build_test_latlong(float *rgb)
{
    const int w = 512, h = 256;
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int height = w / 2;
            float u = (float)x / (float)(w - 1), v = (float)y / (float)(height - 1);
            float theta = v * PI_F;
            float p2 = u * PI_F * 2.0f + PI_F / 2.0f;
            float p1 = (p2 >= PI_F * 2.0f) ? p2 - PI_F * 2.0f : p2;
            float phi = 2.0f * PI_F - p1;
            v3 l = V3(sinf(theta) * cosf(phi), sinf(theta) * sinf(phi), cosf(theta));
            /* localToWorld with n=(0,1,0), x=(1,0,0), y=(0,0,-1) */
            v3 d = V3(l.x * 1.0f + l.y * 0.0f + l.z * 0.0f,
                      l.x * 0.0f + l.y * 0.0f + l.z * 1.0f,
                      l.x * 0.0f + l.y * -1.0f + l.z * 0.0f);
            float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
            float c[3];
            if (ax >= ay && ax >= az) { if (d.x > 0.0f) { c[0] = 1; c[1] = 0; c[2] = 0; } else { c[0] = 0; c[1] = 1; c[2] = 0; } }
            else if (ay >= ax && ay >= az) { if (d.y > 0.0f) { c[0] = 0; c[1] = 0; c[2] = 1; } else { c[0] = 1; c[1] = 0; c[2] = 1; } }
            else { if (d.z < 0.0f) { c[0] = 1; c[1] = 1; c[2] = 0; } else { c[0] = 0; c[1] = 1; c[2] = 1; } }
            float *p = rgb + ((size_t)x + (size_t)y * w) * 3;
            p[0] = c[0]; p[1] = c[1]; p[2] = c[2];
        }
}

/* resizeHDRImage, HDREnvMap.hs:169-195 */
int orc_resize_hdr(const float *src, int sw, int sh, int dstw, float *out)
{
    int dsth = (int)rintf((float)sh / (float)sw * (float)dstw);   /* Haskell round = half-even */
    if (!out) return dsth;
    float scale = (float)sw / (float)dstw;
    int taps = (int)ceilf(scale);
    float ntaps = (float)(taps * taps);
    float step = scale / (float)taps;
    for (int dy = 0; dy < dsth; dy++)
        for (int dx = 0; dx < dstw; dx++) {
            float srcx1 = (float)dx * scale, srcy1 = (float)dy * scale;
            float ar = 0.0f, ag = 0.0f, ab = 0.0f;
            for (int y = 0; y < taps; y++)
                for (int x = 0; x < taps; x++) {
                    float sx = srcx1 + (float)x * step, sy = srcy1 + (float)y * step;
                    float u = sx / ((float)sw - 1.0f), v = sy / ((float)sh - 1.0f);
                    float c[3];
                    pixel_at_bilinear(src, sw, sh, u, v, c);
                    ar = ar + c[0]; ag = ag + c[1]; ab = ab + c[2];
                }
            float *p = out + ((size_t)dx + (size_t)dy * dstw) * 3;
            p[0] = ar / ntaps; p[1] = ag / ntaps; p[2] = ab / ntaps;
        }
    return dsth;
}

typedef struct { const float *src; int w, h; float power; float *out; int lo, hi; int pow_mode; } conv_job;

/* cos^p of cosineConvolveHDREnvMap (`cosAngle ** power`, HDREnvMap.hs:246).
 * pow_mode 0: the literal call -- GHC's (**) :: Float is libm powf, whose result is NOT correctly rounded
 *             (glibc 2.35: differs from the correctly rounded power in ~0.1 % of arguments) and depends on
 *             the libm build, so the reference's own cache files are only defined up to that.
 * pow_mode 1: SPEC PIN for the reference's four powers 1, 8, 64, 512 = 2^k: k squarings in
 *             binary64 rounded once to binary32 (relative error <= 2^k * 2^-53 before the rounding).  Other
 *             powers have no pin and use powf.  tests/test_oracle_kat.py bounds the distance between the modes. */
static inline float conv_pow(float c, float power, int pow_mode)
{
    if (pow_mode == 1) {
        int k = -1;
        if (power == 1.0f) k = 0; else if (power == 8.0f) k = 3; else if (power == 64.0f) k = 6; else if (power == 512.0f) k = 9;
        if (k >= 0) {
            double d = (double)c;
            for (int i = 0; i < k; i++) d = d * d;
            return (float)d;
        }
    }
    return powf(c, power);
}

/* cosineConvolveHDREnvMap, HDREnvMap.hs:217-254; libm cosf/sinf stand in for GHC's Float cos/sin, which call
 * the same libm */
static void *conv_worker(void *arg)
{
    conv_job *j = (conv_job *)arg;
    int w = j->w, h = j->h;
    float *lut = (float *)malloc((size_t)w * sizeof(float));
    float *tcos = (float *)malloc((size_t)h * sizeof(float));
    float *tsin = (float *)malloc((size_t)h * sizeof(float));
    for (int y = 0; y < h; y++) {
        float th = (float)y / (float)(h - 1) * PI_F;
        tcos[y] = cosf(th); tsin[y] = sinf(th);
    }
    for (int dy = j->lo; dy < j->hi; dy++)
        for (int dx = 0; dx < w; dx++) {
            float theta_l = (float)dy / (float)(h - 1) * PI_F;
            float lc = cosf(theta_l), ls = sinf(theta_l);
            float phi_l = (float)dx / (float)(w - 1) * 2.0f * PI_F;
            for (int x = 0; x < w; x++) lut[x] = cosf(fabsf(phi_l - (float)x / (float)(w - 1) * 2.0f * PI_F));
            float ar = 0.0f, ag = 0.0f, ab = 0.0f, n = 0.0f;
            for (int y = 0; y < h; y++) {
                float pc = tcos[y], ps = tsin[y];
                const float *row = j->src + (size_t)y * w * 3;
                for (int x = 0; x < w; x++) {
                    float cos_angle = lc * pc + ls * ps * lut[x];
                    if (cos_angle > 0.0f) {
                        float fac = ps * conv_pow(cos_angle, j->power, j->pow_mode);
                        ar = ar + row[x * 3] * fac; ag = ag + row[x * 3 + 1] * fac; ab = ab + row[x * 3 + 2] * fac;
                        n = n + 1.0f;
                    }
                }
            }
            float *p = j->out + ((size_t)dx + (size_t)dy * w) * 3;
            p[0] = ar / n; p[1] = ag / n; p[2] = ab / n;
        }
    free(lut); free(tcos); free(tsin);
    return NULL;
}

static void conv_set_range(void *job, int lo, int hi) { ((conv_job *)job)->lo = lo; ((conv_job *)job)->hi = hi; }

void orc_cosine_convolve(const float *src, int w, int h, float power, float *out, int nthreads, int pow_mode)
{
    conv_job *jobs = (conv_job *)calloc(256, sizeof(conv_job));
    jobs[0].src = src; jobs[0].w = w; jobs[0].h = h; jobs[0].power = power; jobs[0].out = out; jobs[0].pow_mode = pow_mode;
    run_segments(nthreads, 0, h, conv_worker, jobs, sizeof(conv_job), conv_set_range);
    free(jobs);
}

/* ============================================================================
 * 8. Fractal2D.hs
 * ==========================================================================*/

static inline uint32_t iter_to_green(float v) { return ((uint32_t)(int64_t)v) << 8; }  /* truncate, shiftL 8 */

static inline float fractional_iter_cnt(int iter, float zr, float zi)
{
    /* Fractal2D.hs:24-25 */
    float v = (float)iter - logf(logf(zr * zr + zi * zi)) / logf(2.0f);
    return (0.0f <= v) ? v : 0.0f;            /* Haskell max 0 v = if 0 <= v then v else 0 */
}

typedef struct { int w, h; uint32_t *fb; int smooth; float jr, ji; int lo, hi; } julia_job;

static void *julia_worker(void *arg)
{
    julia_job *j = (julia_job *)arg;
    const int max_iter = 40;
    float fw = (float)j->w, fh = (float)j->h;
    float ratio = fw / fh;
    float xshift = 1.45f * ratio;
    for (int py = j->lo; py < j->hi; py++)
        for (int px = 0; px < j->w; px++) {
            float y = ((float)py / fh) * 2.9f - 1.45f;
            float x = ((float)px / fw) * 2.9f * ratio - xshift;
            float zr = x, zi = y;
            int iter = 0;
            for (;;) {
                if (iter == max_iter || zr * zr + zi * zi > 4.0f * 4.0f) break;
                float nr = (zr * zr - zi * zi) + j->jr;
                float ni = (zr * zi + zi * zr) + j->ji;
                if (nr == zr && ni == zi) { iter = max_iter; break; }
                zr = nr; zi = ni; iter++;
            }
            float cont = (iter == max_iter) ? (float)max_iter : fractional_iter_cnt(iter, zr, zi);
            float val = j->smooth ? cont / (float)max_iter * 255.0f : (float)iter / (float)max_iter * 255.0f;
            j->fb[(size_t)px + (size_t)py * j->w] = iter_to_green(val);
        }
    return NULL;
}

static void julia_set_range(void *job, int lo, int hi) { ((julia_job *)job)->lo = lo; ((julia_job *)job)->hi = hi; }

/* Fractal2D.hs:63-98 */
void orc_julia_animated(int w, int h, uint32_t *fb, int smooth, double tick, int nthreads)
{
    float ft = (float)tick;
    float a = ft / 17.0f, b = ft / 61.0f, c = ft / 71.0f;
    float s1 = a - truncf(a), s2 = b - truncf(b), s3 = c - truncf(c);   /* snd . properFraction */
    float two_pi = s1 * 2.0f * PI_F;
    julia_job *jobs = (julia_job *)calloc(256, sizeof(julia_job));
    jobs[0].w = w; jobs[0].h = h; jobs[0].fb = fb; jobs[0].smooth = smooth;
    jobs[0].jr = sinf(two_pi) * ((0.7f < s2) ? s2 : 0.7f);
    jobs[0].ji = cosf(two_pi) * ((0.7f < s3) ? s3 : 0.7f);
    run_segments(nthreads, 0, h, julia_worker, jobs, sizeof(julia_job), julia_set_range);
    free(jobs);
}

/* Fractal2D.hs:30-57 (single-threaded in the reference) */
void orc_mandelbrot(int w, int h, uint32_t *fb, int smooth)
{
    const int max_iter = 40;
    float fw = (float)w, fh = (float)h;
    float ratio = fw / fh;
    for (int py = 0; py < h; py++)
        for (int px = 0; px < w; px++) {
            float y = ((float)py / fh) * 2.0f - 1.0f;
            float xshift = (-2.0f) - ((2.0f * ratio - 2.5f) * 0.5f);
            float x = ((float)px / fw) * 2.0f * ratio + xshift;
            float zr = 0.0f, zi = 0.0f;
            int iter = 0;
            for (;;) {
                if (iter == max_iter || zr * zr + zi * zi > 4.0f * 4.0f) break;
                float nr = (zr * zr - zi * zi) + x;
                float ni = (zr * zi + zi * zr) + y;
                if (nr == zr && ni == zi) { iter = max_iter; break; }
                zr = nr; zi = ni; iter++;
            }
            float cont = (iter == max_iter) ? (float)max_iter : fractional_iter_cnt(iter, zr, zi);
            float val = smooth ? cont / (float)max_iter * 255.0f : (float)iter / (float)max_iter * 255.0f;
            fb[(size_t)px + (size_t)py * w] = iter_to_green(val);
        }
}

/* ============================================================================
 * 9. Super-sampling resolve: one level of the RGBA8 mip chain that
 *    fillFrameBuffer / drawIntoFrameBuffer build with glGenerateMipmap
 *    (FrameBuffer.hs:153-154,187-195) -- 2x2 box filter per channel; the rounding
 *    of the 8-bit average is driver-defined, pinned as round-half-up
 *    ((a+b+c+d+2) >> 2).  Even source sizes only.
 * ==========================================================================*/
int orc_resolve_box2(const uint32_t *src, int sw, int sh, uint32_t *dst)
{
    if (sw <= 0 || sh <= 0 || (sw & 1) || (sh & 1)) return -1;
    int dw = sw / 2, dh = sh / 2;
    for (int y = 0; y < dh; y++)
        for (int x = 0; x < dw; x++) {
            const uint32_t a = src[(size_t)(2 * y) * sw + 2 * x], b = src[(size_t)(2 * y) * sw + 2 * x + 1];
            const uint32_t c = src[(size_t)(2 * y + 1) * sw + 2 * x], d = src[(size_t)(2 * y + 1) * sw + 2 * x + 1];
            uint32_t o = 0;
            for (int k = 0; k < 4; k++) {
                const uint32_t sum = ((a >> (8 * k)) & 255u) + ((b >> (8 * k)) & 255u) + ((c >> (8 * k)) & 255u) + ((d >> (8 * k)) & 255u);
                o |= ((sum + 2u) >> 2) << (8 * k);
            }
            dst[(size_t)y * dw + x] = o;
        }
    return 0;
}
