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
