/* tests/doh_shim/hip/hip_runtime.h -- TEST INFRASTRUCTURE.  What csrc/rmdf_device.hpp needs from <hip/hip_runtime.h> when it is compiled
 * for the CPU, one "lane" at a time (tests/device_on_host.cpp): the qualifiers as nothing, the vector structs, the bit casts, a
 * one-lane __ballot, and the three hardware approximations the device code seeds its exact roots / reciprocals with.
 *
 * The approximations: gfx950's v_rsq_f32 / v_rcp_f32 / v_sqrt_f32 are accurate to 1 ulp and their exact bits are not reproducible here.
 * doh_seed_mode chooses what they return: 0 the correctly rounded value, 1 / 2 one ulp above / below it, 3 a pseudo-random one of the
 * three per call.  Device code whose result must not depend on which 1-ulp-accurate seed the hardware hands it (every "correctly
 * rounded" sequence of rmdf_device.hpp) has to return the same bits in all four modes; where it provably does only for the hardware's
 * own values (the exhaustive GPU test is the authority there), the harness reports how many inputs differ under perturbation. */
#pragma once
#include <math.h>
#include <stdint.h>
#include <string.h>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))

struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline float2 make_float2(float x, float y) { float2 r = { x, y }; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = { x, y, z, w }; return r; }
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r = { x, y }; return r; }

static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }

/* one lane: the wave-uniform "does any lane need the slow path" tests become "does this lane" */
static inline unsigned long long __ballot(int pred) { return pred ? 1ull : 0ull; }

extern thread_local int doh_seed_mode;
extern thread_local unsigned doh_seed_rng;
static inline float doh_perturb(float y)
{
    int m = doh_seed_mode;
    if (m == 3) { doh_seed_rng = doh_seed_rng * 1664525u + 1013904223u; m = (int)((doh_seed_rng >> 24) % 3u); }
    if (m == 0 || !(y == y) || y == 0.0f || isinf(y)) return y;
    unsigned u = __float_as_uint(y);
    u = (m == 1) ? u + 1u : u - 1u;           /* one ulp away from zero / towards zero: same for both signs' magnitudes */
    return __uint_as_float(u);
}
static inline float doh_rsq(float x) { return doh_perturb((float)(1.0 / sqrt((double)x))); }
static inline float doh_rcp(float x) { return doh_perturb((float)(1.0 / (double)x)); }
static inline float doh_sqrt(float x) { return doh_perturb((float)sqrt((double)x)); }
#define __builtin_amdgcn_rsqf(x) doh_rsq(x)
#define __builtin_amdgcn_rcpf(x) doh_rcp(x)
#define __builtin_amdgcn_sqrtf(x) doh_sqrt(x)
/* cross-lane reads with one lane: the lane itself */
#define __builtin_amdgcn_readlane(v, l) (v)
#define __builtin_amdgcn_readfirstlane(v) (v)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rmask, bmask, bc) (src)
