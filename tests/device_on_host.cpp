// device_on_host.cpp -- TEST INFRASTRUCTURE (CPU tier, round 6): csrc/rmdf_device.hpp -- the per-ray arithmetic of the HIP kernels, the very
// header the product compiles for gfx950 -- compiled for the CPU with -ffp-contract=off, one "lane" at a time (tests/doh_shim/: qualifiers
// as nothing, a one-lane __ballot, the three hardware approximations emulated to 1 ulp), and held against the oracle (liboracle.so, the C
// restatement of fragment.shd) on the same inputs, bit for bit.  What this says: the device SOURCE computes what the oracle computes --
// every distance estimator (power-8 Mandelbulb folded and as written, general power, test scene, Cornell box full / table / per-lane
// pruned), the pinned log / exp / pow / acos / atan2 / sin / cos, the exact roots and quotients, Fresnel, ray-sphere, the cube-map lookup,
// gamma + UNORM8 -- independently of which 1-ulp-accurate seed the hardware's v_rsq / v_rcp / v_sqrt return.  What it does NOT say: anything
// about the code generator, the cross-lane schedule (pooling, DPP minima, LDS queues) or the hardware -- the -m gpu tier is the authority
// there.  Built and driven by tests/test_device_source_on_host.py; never part of the product.
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include <thread>
#include <vector>

#define RMDF_HOST_EMULATION 1
#include "rmdf_device.hpp"          // (-I csrc; <hip/hip_runtime.h> and <hip/hip_fp16.h> resolve to tests/doh_shim/)

extern "C" {
#include "../oracle/rmdf_oracle.h"
}

thread_local int doh_seed_mode = 0;
thread_local unsigned doh_seed_rng = 12345u;

using namespace rmdf;

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) { }
    uint32_t u32() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); }
    float uni() { return (float)(u32() >> 8) * (1.0f / 16777216.0f); }                  // [0, 1)
    float range(float a, float b) { return a + (b - a) * uni(); }
};

inline uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
inline bool same(float a, float b) { return bits(a) == bits(b) || (a != a && b != b); }

struct Stats {          // mirrored in the Python test (ctypes)
    long long n, mismatches, folded_vs_written, iters_mismatches, guard_trips;
    float first_in[4];
    float first_got, first_want;
};

void note(Stats &st, float got, float want, float a, float b = 0.0f, float c = 0.0f, float d = 0.0f)
{
    if (st.mismatches++ == 0) { st.first_in[0] = a; st.first_in[1] = b; st.first_in[2] = c; st.first_in[3] = d; st.first_got = got; st.first_want = want; }
}

template <typename F>
void parallel(long long n, int threads, F fn)            // fn(thread index, lo, hi)
{
    if (threads < 1) threads = 1;
    std::vector<std::thread> ts;
    for (int t = 0; t < threads; t++) ts.emplace_back([=] { fn(t, n * t / threads, n * (t + 1) / threads); });
    for (auto &t : ts) t.join();
}

void merge(Stats &dst, const Stats &s)
{
    if (dst.mismatches == 0 && s.mismatches) { memcpy(dst.first_in, s.first_in, sizeof s.first_in); dst.first_got = s.first_got; dst.first_want = s.first_want; }
    dst.n += s.n; dst.mismatches += s.mismatches; dst.folded_vs_written += s.folded_vs_written; dst.iters_mismatches += s.iters_mismatches; dst.guard_trips += s.guard_trips;
}

// the Cornell tables of the product (librmdf_xcheck.so: rmdf_debug_cornell_table / rmdf_debug_cornell_masks), handed in by the test
const float *g_ctab = nullptr;         // CORNELL_TAB_FLOATS
const unsigned *g_fine = nullptr;      // CORNELL_FINE_N^3
const unsigned *g_coarse = nullptr;    // CORNELL_GRID_N^3
float g_tri[96 * 3];

float device_de(int scene, v3 p, float power, unsigned &iters, int variant, int &hint)
{
    switch (scene) {
    case 2: return variant == 1 ? de_mandelbulb8_written_inl(p, iters) : de_mandelbulb8(p, iters);
    case 3: return de_mandelbulb_general(p, power, iters);
    case 1: return de_test_scene(p);
    default:
        if (variant == 0) return de_cornell_box(p, g_tri);                                     // the reference's loop over 32 triangles
        if (variant == 1) return de_cornell_box_table(p, g_ctab, 0, hint, nullptr);            // table, no pruning
        if (variant == 2) return de_cornell_box_table(p, g_ctab, 1, hint, g_coarse);           // table, bounds + coarse grid (rounds 1-2)
        return de_cornell_box_lanes(p, g_ctab, g_fine, hint);                                  // per-lane pruned estimate (the product's)
    }
}

}  // namespace

extern "C" {

void doh_set_cornell(const float *tab, const unsigned *fine, const unsigned *coarse)
{
    g_ctab = tab; g_fine = fine; g_coarse = coarse;
    orc_cornell_vertices(g_tri);
}

int doh_sizes(int *stride, int *bounds, int *tab_floats, int *fine_n, int *coarse_n)
{
    *stride = CORNELL_STRIDE; *bounds = CORNELL_BOUNDS; *tab_floats = CORNELL_TAB_FLOATS; *fine_n = CORNELL_FINE_N; *coarse_n = CORNELL_GRID_N;
    return (int)sizeof(Stats);
}

// Distance estimates along sphere-traced rays: `rays` rays start on a sphere around the scene, aim at a random point near the origin and
// march with the ORACLE's estimate (max `steps` steps, the reference's hit / miss rules); at every position visited the device source's
// estimate must have the oracle's bits.  variant: scene 2: 0 folded passes + guard + fall-back (the product), 1 as written; scene 0: 0
// reference loop, 1 table, 2 table + bounds + coarse grid, 3 per-lane pruned (the product).  Plus `extra` uniformly random points of the
// scene's bounding cube and, for the Mandelbulbs, points on / next to the axes and coordinate planes (where the guards trip).
void doh_check_de(int scene, int variant, float time, long long rays, int steps, long long extra, unsigned seed, int seed_mode, int threads, Stats *out)
{
    memset(out, 0, sizeof *out);
    const float power = orc_general_power(time);
    std::vector<Stats> per((size_t)(threads < 1 ? 1 : threads));
    parallel(rays + extra, threads, [&](int t, long long lo, long long hi) {
        doh_seed_mode = seed_mode; doh_seed_rng = seed * 2654435761u + (unsigned)t;
        Stats st; memset(&st, 0, sizeof st);
        int hint = 0;
        auto probe = [&](v3 p) {
            const float q[3] = { p.x, p.y, p.z };
            const float want = orc_de(scene, time, q);
            unsigned it = 0;
            const float got = device_de(scene, p, power, it, variant, hint);
            st.n++;
            if (!same(got, want)) note(st, got, want, p.x, p.y, p.z);
            if (scene == 2 && variant == 0) {
                unsigned itw = 0, redone = 0;
                const float w = de_mandelbulb8_written_inl(p, itw);
                unsigned it2 = 0;
                (void)de_mandelbulb8(p, it2, RMDF_MB8_FOLD_MIN, &redone);
                if (!same(got, w)) st.folded_vs_written++;
                if (it != itw) st.iters_mismatches++;
                st.guard_trips += redone;
            }
            return want;
        };
        for (long long i = lo; i < hi; i++) {
            Rng r((uint64_t)seed * 1000003ull + (uint64_t)i);
            if (i < rays) {
                // a ray as the renderer shoots them: from a camera-like distance towards the scene
                const float R = scene == 0 ? 2.2f : 2.4142f;
                float ox, oy, oz, n;
                do { ox = r.range(-1, 1); oy = r.range(-1, 1); oz = r.range(-1, 1); n = ox * ox + oy * oy + oz * oz; } while (n < 0.01f || n > 1.0f);
                n = R / sqrtf(n); ox *= n; oy *= n; oz *= n;
                const float aim = scene == 0 ? 0.9f : 0.8f;
                float dx = r.range(-aim, aim) - ox, dy = r.range(-aim, aim) - oy, dz = r.range(-aim, aim) - oz;
                n = 1.0f / sqrtf(dx * dx + dy * dy + dz * dz); dx *= n; dy *= n; dz *= n;
                float tt = scene == 0 ? 0.0f : fmaxf(0.0f, R - 1.6f);
                for (int s = 0; s < steps; s++) {
                    const v3 p = mk3(ox + dx * tt, oy + dy * tt, oz + dz * tt);
                    const float d = probe(p);
                    if (!(d == d) || d < 0.001f || tt > 2.0f * R) break;
                    tt += d;
                }
            } else {
                const long long k = i - rays;
                const float h = scene == 0 ? 1.3f : 1.3f;
                v3 p = mk3(r.range(-h, h), r.range(-h, h), r.range(-h, h));
                if (scene >= 2) {
                    // every eighth point on an axis / coordinate plane, or within 2^-20 of one (zero divisors, guards, NaN handling)
                    const unsigned pick = (unsigned)(k & 63);
                    const float tiny = ldexpf(r.range(-1, 1), -20 - (int)(r.u32() % 40u));
                    if (pick == 0) { p.x = 0.0f; p.y = 0.0f; }
                    else if (pick == 1) { p.x = tiny; p.y = ldexpf(r.range(-1, 1), -30); }
                    else if (pick == 2) p.x = 0.0f;
                    else if (pick == 3) p.y = tiny;
                    else if (pick == 4) p.z = 0.0f;
                    else if (pick == 5) { p.x = p.y = p.z = 0.0f; }
                    else if (pick == 6) p.z = tiny;
                    else if (pick == 7) { p.y = 0.0f; p.z = 0.0f; }
                }
                probe(p);
            }
        }
        per[(size_t)t] = st;
    });
    for (auto &s : per) merge(*out, s);
}

// One-argument functions at n pseudo-random inputs of the ranges the shader feeds them (plus special values):
// 0 sqrt_rn vs sqrtf, 1 rcp_rn vs 1/x, 2 rsqrt_ieee vs 1/sqrtf, 3 log_pinned vs orc_logf, 4 exp_pinned vs orc_expf, 5 acos_pinned vs orc_acosf,
// 6 sin, 7 cos (sincos_pinned) vs orc_sinf / orc_cosf, 8 to_unorm8(pow_pinned(x, 1/2.2)) vs the oracle's gamma + rounding
void doh_check_unary(int fn, long long n, unsigned seed, int seed_mode, int threads, Stats *out)
{
    memset(out, 0, sizeof *out);
    std::vector<Stats> per((size_t)(threads < 1 ? 1 : threads));
    parallel(n, threads, [&](int t, long long lo, long long hi) {
        doh_seed_mode = seed_mode; doh_seed_rng = seed * 2654435761u + (unsigned)t;
        Stats st; memset(&st, 0, sizeof st);
        static const float special[] = { 0.0f, -0.0f, 1.0f, -1.0f, 0.5f, 2.0f, 4.0f, INFINITY, -INFINITY, NAN, 1e-45f, 1.17549435e-38f, 3.4028235e38f,
                                         7.8886e-31f /* 2^-100 */, 1.2676506e30f /* 2^100 */, 0.99999994f, 1.0000001f, 1e-30f, 1e30f };
        for (long long i = lo; i < hi; i++) {
            Rng r((uint64_t)seed * 7919ull + (uint64_t)i);
            float x;
            const unsigned kind = r.u32() % 16u;
            if (i < (long long)(sizeof special / sizeof special[0])) x = special[i];
            else if (kind == 0) x = __uint_as_float(r.u32());                                     // any bit pattern
            else if (kind < 4) x = ldexpf(r.range(1.0f, 2.0f), (int)(r.u32() % 200u) - 100);        // the core range of the exact sequences
            else if (fn == 5) x = r.range(-1.0f, 1.0f);
            else if (fn == 4) x = r.range(-90.0f, 90.0f);
            else if (fn == 6 || fn == 7) x = r.range(-40.0f, 40.0f);
            else if (fn == 8) x = r.range(-0.1f, 1.6f);
            else x = ldexpf(r.range(1.0f, 2.0f), (int)(r.u32() % 60u) - 30) * ((fn == 1 && (r.u32() & 1u)) ? -1.0f : 1.0f);
            float got, want;
            switch (fn) {
            case 0: got = sqrt_rn(x); want = sqrtf(x); break;
            case 1: got = rcp_rn(x); want = 1.0f / x; break;
            case 2: got = rsqrt_ieee(x); want = 1.0f / sqrtf(x); break;
            case 3: got = log_pinned(x); want = orc_logf(x); break;
            case 4: got = exp_pinned(x); want = orc_expf(x); break;
            case 5: got = acos_pinned(x); want = orc_acosf(x); break;
            case 6: { float s, c; sincos_pinned(x, s, c); got = s; want = orc_sinf(x); break; }
            case 7: { float s, c; sincos_pinned(x, s, c); got = c; want = orc_cosf(x); break; }
            default: {
                const float g = pow_pinned(x, 1.0f / shk::gamma);
                got = (float)to_unorm8(g);
                const float w = orc_powf(x, 1.0f / shk::gamma);
                want = !(w == w) ? 0.0f : rintf(fminf(fmaxf(w, 0.0f), 1.0f) * 255.0f);
                if (!same(g, w)) { got = g; want = w; }
                break; }
            }
            st.n++;
            if (!same(got, want)) note(st, got, want, x);
        }
        per[(size_t)t] = st;
    });
    for (auto &s : per) merge(*out, s);
}

// sqrt_rn / rcp_rn over EVERY float of the exact sequences' core range [2^-100, 2^100] whose low `skip_bits` mantissa bits are a fixed
// pattern (skip_bits = 0: all 1.68 G of them), against the CPU's correctly rounded sqrtf / division
void doh_check_exact_exhaustive(int fn, int skip_bits, unsigned low_pattern, int seed_mode, int threads, Stats *out)
{
    memset(out, 0, sizeof *out);
    const uint32_t lo_bits = 0x0d800000u, hi_bits = 0x71800000u;
    const long long total = ((long long)(hi_bits - lo_bits) >> skip_bits) + 1;
    std::vector<Stats> per((size_t)(threads < 1 ? 1 : threads));
    parallel(total, threads, [&](int t, long long lo, long long hi) {
        doh_seed_mode = seed_mode; doh_seed_rng = 99u + (unsigned)t;
        Stats st; memset(&st, 0, sizeof st);
        for (long long i = lo; i < hi; i++) {
            const uint32_t u = lo_bits + (((uint32_t)i << skip_bits) | (low_pattern & ((1u << skip_bits) - 1u)));
            if (u > hi_bits) continue;
            const float x = __uint_as_float(u);
            float got, want;
            if (fn == 0) { got = sqrt_rn(x); want = sqrtf(x); }
            else if (fn == 1) { got = rcp_rn(x); want = 1.0f / x; }
            else { got = rsqrt_ieee(x); want = 1.0f / sqrtf(x); }
            st.n++;
            if (!same(got, want)) note(st, got, want, x);
        }
        per[(size_t)t] = st;
    });
    for (auto &s : per) merge(*out, s);
}

// 0 pow_pinned(x, y) vs orc_powf, 1 atan2_pinned vs orc_atan2f, 2 div_known_range(a, b) vs a / b (operands inside its stated range),
// 3 fresnel_conductor(cosi, 0.4, 0.8) vs the oracle's, 4 triplex_pow8 (three outputs), 5 ray_sphere
void doh_check_binary(int fn, long long n, unsigned seed, int seed_mode, int threads, Stats *out)
{
    memset(out, 0, sizeof *out);
    std::vector<Stats> per((size_t)(threads < 1 ? 1 : threads));
    parallel(n, threads, [&](int t, long long lo, long long hi) {
        doh_seed_mode = seed_mode; doh_seed_rng = seed * 2654435761u + (unsigned)t;
        Stats st; memset(&st, 0, sizeof st);
        for (long long i = lo; i < hi; i++) {
            Rng r((uint64_t)seed * 104729ull + (uint64_t)i);
            st.n++;
            if (fn == 0) {
                const float x = (r.u32() & 7u) ? r.range(0.0f, 4.0f) : ldexpf(r.range(1, 2), (int)(r.u32() % 60u) - 30), y = (r.u32() & 3u) ? r.range(-9.0f, 9.0f) : (float)((int)(r.u32() % 17u) - 8);
                const float got = pow_pinned(x, y), want = orc_powf(x, y);
                if (!same(got, want)) note(st, got, want, x, y);
            } else if (fn == 1) {
                float y = r.range(-2, 2), x = r.range(-2, 2);
                if ((r.u32() & 31u) == 0u) x = 0.0f;
                if ((r.u32() & 31u) == 0u) y = 0.0f;
                if ((r.u32() & 63u) == 0u) y = ldexpf(y, -40);
                const float got = atan2_pinned(y, x), want = orc_atan2f(y, x);
                if (!same(got, want)) note(st, got, want, y, x);
            } else if (fn == 2) {
                const float b = ldexpf(r.range(1, 2), (int)(r.u32() % 80u) - 40) * ((r.u32() & 1u) ? -1.0f : 1.0f);
                const float a = (r.u32() & 15u) ? ldexpf(r.range(1, 2), (int)(r.u32() % 80u) - 40) * ((r.u32() & 1u) ? -1.0f : 1.0f) : 0.0f;
                const float got = div_known_range(a, b), want = a / b;
                if (!same(got, want)) note(st, got, want, a, b);
            } else if (fn == 3) {
                const float c = (r.u32() & 15u) ? r.range(0.0f, 1.0f) : r.range(-0.2f, 1.2f);
                const float got = fresnel_conductor<RMDF_SHADE_FAST>(c, shk::fresnel_eta, shk::fresnel_k), want = orc_fresnel_conductor(c, shk::fresnel_eta, shk::fresnel_k);
                if (!same(got, want)) note(st, got, want, c);
            } else if (fn == 4) {
                v3 w = mk3(r.range(-1.3f, 1.3f), r.range(-1.3f, 1.3f), r.range(-1.3f, 1.3f));
                if ((r.u32() & 31u) == 0u) { w.x = 0.0f; w.y = 0.0f; }
                const v3 got = triplex_pow8(w);
                const float wi[3] = { w.x, w.y, w.z };
                float want[3];
                orc_triplex_pow8(wi, want);
                if (!same(got.x, want[0])) note(st, got.x, want[0], w.x, w.y, w.z, 0);
                else if (!same(got.y, want[1])) note(st, got.y, want[1], w.x, w.y, w.z, 1);
                else if (!same(got.z, want[2])) note(st, got.z, want[2], w.x, w.y, w.z, 2);
            } else {
                const float o[3] = { r.range(-3, 3), r.range(-3, 3), r.range(-3, 3) };
                float d[3] = { r.range(-1, 1), r.range(-1, 1), r.range(-1, 1) };
                const float n = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2] + 1e-12f);
                d[0] *= n; d[1] *= n; d[2] *= n;
                const float R = (r.u32() & 1u) ? 1.15f : 1.5f;
                float t0 = 0, t1 = 0, w0 = 0, w1 = 0;
                const bool hit = ray_sphere<RMDF_SHADE_FAST>(mk3(o[0], o[1], o[2]), mk3(d[0], d[1], d[2]), R, t0, t1);
                const int whit = orc_ray_sphere(o, d, R, &w0, &w1);
                if ((int)hit != whit) note(st, (float)hit, (float)whit, o[0], o[1], o[2], R);
                else if (hit && !(same(t0, w0) && same(t1, w1))) note(st, t0, w0, o[0], o[1], o[2], R);
            }
        }
        per[(size_t)t] = st;
    });
    for (auto &s : per) merge(*out, s);
}

// texture(samplerCube, dir): `padded` = 6 x (W+2)^2 texels of four halfs (the layout rmdf_set_env_cube / orc_cube_pad_f16 produce).
// linear = 0: no quad neighbours (NEAREST); 1: neighbours identical to the lane's direction (footprint 0: LINEAR); both against
// orc_cube_sample with the same explicit choice.
void doh_check_cube(const uint16_t *padded, int W, long long n, unsigned seed, int seed_mode, int threads, Stats *out)
{
    memset(out, 0, sizeof *out);
    CubeDev c; c.texels = (const uint2 *)padded; c.W = W;
    orc_cube oc; oc.W = W; oc.padded = padded;
    std::vector<Stats> per((size_t)(threads < 1 ? 1 : threads));
    parallel(n, threads, [&](int t, long long lo, long long hi) {
        doh_seed_mode = seed_mode; doh_seed_rng = seed * 2654435761u + (unsigned)t;
        Stats st; memset(&st, 0, sizeof st);
        for (long long i = lo; i < hi; i++) {
            Rng r((uint64_t)seed * 31337ull + (uint64_t)i);
            v3 d = mk3(r.range(-1, 1), r.range(-1, 1), r.range(-1, 1));
            if ((r.u32() & 15u) == 0u) { const float m = fmaxf(fabsf(d.x), fmaxf(fabsf(d.y), fabsf(d.z))); d.x = d.x < 0 ? -m : m; }     // on a cube edge
            const int linear = (int)(r.u32() & 1u);
            const v3 got = cube_texture<RMDF_SHADE_FAST>(c, d, linear != 0, d, linear != 0, d);
            const float di[3] = { d.x, d.y, d.z };
            float want[3];
            orc_cube_sample(&oc, di, linear, want);
            st.n++;
            if (!(same(got.x, want[0]) && same(got.y, want[1]) && same(got.z, want[2]))) note(st, got.x, want[0], d.x, d.y, d.z, (float)linear);
        }
        per[(size_t)t] = st;
    });
    for (auto &s : per) merge(*out, s);
}

}  // extern "C"
