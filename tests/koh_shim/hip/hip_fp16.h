/* tests/doh_shim/hip/hip_fp16.h -- TEST INFRASTRUCTURE: the two half-precision reads rmdf_device.hpp's texel fetch uses, on the CPU */
#pragma once
#include <stdint.h>
#include <string.h>
struct __half2 { uint16_t lo, hi; };
static inline float doh_half_to_float(uint16_t h)
{
    const unsigned s = (unsigned)(h >> 15) << 31, e = (h >> 10) & 31u, m = h & 1023u;
    unsigned u;
    if (e == 0) {
        if (m == 0) u = s;
        else { int k = 0; unsigned mm = m; while (!(mm & 1024u)) { mm <<= 1; k++; } u = s | ((unsigned)(113 - k) << 23) | ((mm & 1023u) << 13); }
    } else if (e == 31) u = s | 0x7f800000u | (m << 13);
    else u = s | ((e + 112u) << 23) | (m << 13);
    float f; memcpy(&f, &u, 4); return f;
}
static inline float __low2float(__half2 h) { return doh_half_to_float(h.lo); }
static inline float __high2float(__half2 h) { return doh_half_to_float(h.hi); }

/* (kernel_on_host only) float -> half, round to nearest even, the conversion k_cube_upload makes with v_cvt_f16_f32 */
struct __half { uint16_t bits; };
static inline __half __float2half_rn(float f)
{
    unsigned u; memcpy(&u, &f, 4);
    const unsigned s = (u >> 16) & 0x8000u, a = u & 0x7fffffffu;
    __half h;
    if (a > 0x7f800000u) { h.bits = (uint16_t)(s | 0x7e00u | ((a >> 13) & 0x1ffu)); return h; }        /* NaN */
    if (a >= 0x47800000u) { h.bits = (uint16_t)(s | 0x7c00u); return h; }                                 /* >= 65536 (and inf): inf.  65520 rounds up below */
    if (a < 0x33000000u) { h.bits = (uint16_t)s; return h; }                                              /* < 2^-25: zero */
    int e = (int)(a >> 23) - 127;
    unsigned m = (a & 0x7fffffu) | 0x800000u;
    int shift = e < -14 ? 13 + (-14 - e) : 13;
    unsigned q = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    unsigned out = e < -14 ? q : (((unsigned)(e + 15) << 10) + (q - 0x400u));                             /* a carry out of the mantissa moves into the exponent */
    h.bits = (uint16_t)(s | out);
    return h;
}
static inline float __half2float(__half h) { return doh_half_to_float(h.bits); }
static inline unsigned short __half_as_ushort(__half h) { return h.bits; }
static inline __half2 __floats2half2_rn(float a, float b) { __half2 r; r.lo = __float2half_rn(a).bits; r.hi = __float2half_rn(b).bits; return r; }
