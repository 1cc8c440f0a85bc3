// rmdf_march.hip -- the Mandelbulb power-8 hot path as two gfx950 kernels.
//
//   k_march_mb8   persistent waves.  Every lane carries ONE ray through a flattened state machine:
//                 each pass of the wave-level loop runs one Mandelbulb iteration (fragment.shd:134-149)
//                 for every lane that is inside a distance estimate; lanes whose estimate finished park
//                 until enough of them wait, then the "tail" block finishes the estimate
//                 (fragment.shd:157), advances the march (659-673) or the normal / AO taps (463-470,
//                 542-562) and starts the lane's next estimate.  A lane whose pixel is complete takes the
//                 next pixel from a wave-local range that is refilled from one global atomic counter, so
//                 divergence in march length, in escape-iteration count and between hit / miss pixels does
//                 not idle lanes (the 8x8 packet efficiency of a naive nested loop is ~0.25 here).
//                 Per-ray arithmetic and its order are exactly those of the nested formulation, so step
//                 counts and escape-iteration counts stay bit-exact.
//   k_shade       one lane per pixel in 2x2-quad order: reads the G-buffer the march kernel wrote
//                 (normal, AO, hit, steps), selects min/mag filtering from the quad neighbours, samples the
//                 three cube maps, applies Fresnel / gamma and writes the output planes coalesced.
//
// Compile with -ffp-contract=off, no fast-math (see rmdf_device.hpp).
#include <stdlib.h>
#include <string.h>

#include "rmdf_internal.hpp"

namespace rmdf {

#define MODE_IDLE 0
#define MODE_ITER 1
#define MODE_TAIL 2

#define PH_MARCH 0
#define PH_N0 1
#define PH_NX 2
#define PH_NY 3
#define PH_NZ 4
#define PH_AO0 5
#define PH_AO1 6

// scheduling knobs (FrameParams.tail_t / refill_t / chunk; defaults set by the launcher):
//   refill_t  refill when at least this many lanes are idle
//   tail_t    run the march tail when at least this many lanes wait for it (or nobody iterates)
//   shade_t   same for the normal / AO tail
//   chunk     pixels taken from the global counter per atomic

__device__ __forceinline__ int popc64(unsigned long long m) { return __popcll(m); }
__device__ __forceinline__ int lane_rank(unsigned long long m)
{
    // number of set bits of m below this lane
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

// linear work item -> pixel.  Items enumerate 8x8 tiles of the even-aligned rectangle, 64 items per
// tile in quad order (item&1 = x bit, item&2 = y bit), so the lanes that start together are coherent.
__device__ __forceinline__ bool item_to_pixel(const FrameParams &p, int item, int &px, int &py)
{
    int rx0, ry0, rx1, ry1, local = item;
    if (p.n_shard_tiles > 0) {
        const int slot = item / p.items_per_shard_tile;
        local = item - slot * p.items_per_shard_tile;
        const int midx = (int)p.shard_tile[slot];
        const int tx = midx % 8, ty = midx / 8;
        rx0 = (2 * tx * p.w + 7) / 16; rx1 = (2 * (tx + 1) * p.w + 7) / 16;
        ry0 = (2 * ty * p.h + 7) / 16; ry1 = (2 * (ty + 1) * p.h + 7) / 16;
    } else {
        rx0 = p.x0; ry0 = p.y0; rx1 = p.x1; ry1 = p.y1;
    }
    const int ex0 = rx0 & ~1, ey0 = ry0 & ~1, ex1 = (rx1 + 1) & ~1, ey1 = (ry1 + 1) & ~1;
    const int tiles_x = (ex1 - ex0 + 7) >> 3;
    const int tile = local >> 6, l = local & 63;
    const int lx = (l & 1) | (((l >> 2) & 3) << 1);
    const int ly = ((l >> 1) & 1) | (((l >> 4) & 3) << 1);
    px = ex0 + (tile % tiles_x) * 8 + lx;
    py = ey0 + (tile / tiles_x) * 8 + ly;
    return (px < ex1) && (py < ey1);
}

#define ST_IDLE   0      // lane has no pixel
#define ST_ITER   1      // inside a distance estimate: w, squares of w, r = |w| <= 4, dr, it < 25 are live
#define ST_WAIT_M 2      // estimate finished, ray_march bookkeeping pending
#define ST_WAIT_S 3      // estimate finished, normal / AO bookkeeping pending

// fragment.shd:74-99 with the squares of the components passed in (they are the same products the
// radius test just formed, so they are computed once per iteration)
__device__ __forceinline__ v3 triplex_pow8_sq(float x, float y, float z, float x2, float y2, float z2)
{
    const float x4 = x2 * x2, y4 = y2 * y2, z4 = z2 * z2;
    const float k3 = y2 + x2;
    const float k2 = rsqrt_ieee(k3 * k3 * k3 * k3 * k3 * k3 * k3);
    const float k1 = y4 + z4 + x4 - 6.0f * z2 * x2 - 6.0f * y2 * z2 + 2.0f * x2 * y2;
    const float k4 = y2 - z2 + x2;
    return mk3(-8.0f * z * k4 * (y4 * y4 - 28.0f * y4 * y2 * x2 + 70.0f * y4 * x4 - 28.0f * y2 * x2 * x4 + x4 * x4) * k1 * k2,
               64.0f * y * z * x * (y2 - x2) * k4 * (y4 - 6.0f * y2 * x2 + x4) * k1 * k2,
               -16.0f * z2 * k3 * k4 * k4 + k1 * k1);
}

__global__ __launch_bounds__(256) void k_march_mb8(const FrameParams p)
{
    const v3 origin = mk3(p.cam[9], p.cam[10], p.cam[11]);
    const int max_steps = p.max_steps;
    const int total = p.total_items;
    const int refill_t = p.refill_t, tail_t = p.tail_t, shade_t = p.shade_t, chunk = p.chunk;

    // wave-uniform work range handed out lane by lane
    int next = 0, end = 0;
    bool exhausted = false;
    // diagnostics (p.dbg != null only in measurement runs): pass counters, active-lane sums, clocks
    unsigned long long c_iter = 0, c_mtail = 0, c_stail = 0, c_refill = 0, a_iter = 0, a_mtail = 0;
    unsigned long long y_iter = 0, y_mtail = 0, y_stail = 0, y_refill = 0, y0 = 0;
#define STAMP() (p.dbg ? __builtin_amdgcn_s_memtime() : 0ull)
    const unsigned long long t_begin = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;

    // per-lane ray state
    int   st = ST_IDLE, phase = PH_MARCH, steps = 0, it = 0, pix = 0;
    unsigned iters = 0;
    float dirx = 0, diry = 0, dirz = 0, t = 0, tmax = 0;
    float posx = 0, posy = 0, posz = 0, wx = 0, wy = 0, wz = 0, x2 = 0, y2 = 0, z2 = 0, dr = 1.0f, r = 0.0f;
    // hit bookkeeping: intersection point, backed-off point for the normal, the 3 stored estimates,
    // the normal and the running occlusion sum
    float isx = 0, isy = 0, isz = 0, npx = 0, npy = 0, npz = 0;
    float d0 = 0, d1 = 0, d2 = 0, nx = 0, ny = 0, nz = 0, occl = 0;

    // start a distance estimate at world point q (fragment.shd:125-132 and the first radius test of :137)
#define START_DE(qx, qy, qz, wait_state)                                                    \
    do {                                                                                    \
        posx = (qz); posy = (qx); posz = (qy);              /* pos.zxy */                   \
        wx = posx; wy = posy; wz = posz; dr = 1.0f; it = 0;                                 \
        x2 = wx * wx; y2 = wy * wy; z2 = wz * wz;                                           \
        r = sqrt_rn((x2 + y2) + z2);                                                        \
        st = (r > 4.0f) ? (wait_state) : ST_ITER;                                           \
    } while (0)

    for (;;) {
        // ---------------- refill idle lanes ------------------------------------------------------
        unsigned long long idle = __ballot(st == ST_IDLE);
        int n_idle = popc64(idle);
        if (!exhausted && n_idle >= refill_t) {
            y0 = STAMP();
            while (n_idle > 0 && !exhausted) {
                if (next >= end) {
                    int base = 0;
                    if ((threadIdx.x & 63) == 0) base = atomicAdd(p.work_counter, chunk);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= total) { exhausted = true; break; }
                    next = base;
                    end = base + chunk < total ? base + chunk : total;
                }
                c_refill++;
                const int avail = end - next;
                const int rank = lane_rank(idle);
                const bool take = (st == ST_IDLE) && (rank < avail);
                const int item = next + rank;
                next += (n_idle < avail) ? n_idle : avail;
                if (take) {
                    int px, py;
                    const bool valid = item_to_pixel(p, item, px, py);
                    if (valid) {
                        pix = px + py * p.gw;
                        // generate_ray (fragment.shd:840-871)
                        const float ndcx = ((float)px + 0.5f) / p.wf * 2.0f - 1.0f;
                        const float ndcy = ((float)py + 0.5f) / p.hf * 2.0f - 1.0f;
                        const v3 dc = normalize3(mk3(ndcx * p.fov_xs, ndcy * p.fov_xs / p.aspect, -1.0f));
                        dirx = p.cam[0] * dc.x + p.cam[3] * dc.y + p.cam[6] * dc.z;
                        diry = p.cam[1] * dc.x + p.cam[4] * dc.y + p.cam[7] * dc.z;
                        dirz = p.cam[2] * dc.x + p.cam[5] * dc.y + p.cam[8] * dc.z;
                        float tmin;
                        steps = 0; iters = 0;
                        if (ray_sphere<false>(origin, mk3(dirx, diry, dirz), 1.15f, tmin, tmax) && max_steps > 0) {
                            t = gmax(0.0f, tmin);
                            phase = PH_MARCH;
                            START_DE(origin.x + t * dirx, origin.y + t * diry, origin.z + t * dirz, ST_WAIT_M);
                        } else {
                            p.gbuf_meta[pix] = 0u;           // no march: hit 0, steps 0
                        }
                    }
                }
                idle = __ballot(st == ST_IDLE);
                n_idle = popc64(idle);
                if (n_idle < refill_t) break;
            }
            y_refill += STAMP() - y0;
        }

        const unsigned long long iterating = __ballot(st == ST_ITER);

        // ---------------- one Mandelbulb iteration (fragment.shd:134-149) -------------------------
        // entry invariant for ST_ITER lanes: r = |w| <= 4 already tested, it < 25
        if (iterating != 0ull) {
            c_iter++; a_iter += popc64(iterating);
            y0 = STAMP();
            if (st == ST_ITER) {
                const v3 nw = triplex_pow8_sq(wx, wy, wz, x2, y2, z2);
                wx = nw.x + posx; wy = nw.y + posy; wz = nw.z + posz;
                const float r2 = r * r, r4 = r2 * r2, r7 = (r4 * r2) * r;
                dr = r7 * 8.0f * dr + 1.0f;
                iters++;
                it++;
                const int ws = (phase == PH_MARCH) ? ST_WAIT_M : ST_WAIT_S;
                if (it == 25) {
                    st = ws;                                  // loop exhausted: r keeps the last tested radius
                } else {
                    x2 = wx * wx; y2 = wy * wy; z2 = wz * wz;
                    r = sqrt_rn((x2 + y2) + z2);              // the next iteration's radius test (:137-139)
                    if (r > 4.0f) st = ws;
                }
            }
            y_iter += STAMP() - y0;
        }

        const unsigned long long wait_m = __ballot(st == ST_WAIT_M);
        const unsigned long long wait_s = __ballot(st == ST_WAIT_S);
        const bool none_iter = __ballot(st == ST_ITER) == 0ull;
        if ((wait_m | wait_s) == 0ull) {
            if (none_iter && exhausted && __ballot(st != ST_IDLE) == 0ull) break;
            continue;
        }

        // ---------------- march tail: ray_march loop body after the DE call (fragment.shd:663-672) ----
        if (wait_m != 0ull && (popc64(wait_m) >= tail_t || none_iter)) {
            c_mtail++; a_mtail += popc64(wait_m);
            y0 = STAMP();
            if (st == ST_WAIT_M) {
                const float dist = 0.5f * log_pinned(r) * r / dr;   // fragment.shd:157
                t += dist;
                bool out = t > tmax;
                const bool hit = !out && (dist < 0.001f);
                if (!out && !hit) { steps++; out = steps >= max_steps; }
                if (out) {
                    p.gbuf_meta[pix] = (unsigned)steps | ((iters > 65535u ? 65535u : iters) << 16);
                    st = ST_IDLE;
                } else {
                    float qx, qy, qz;
                    int ws = ST_WAIT_M;
                    if (hit) {
                        // intersection and the backed-off point for the normal (fragment.shd:743-751)
                        isx = origin.x + dirx * t; isy = origin.y + diry * t; isz = origin.z + dirz * t;
                        npx = isx - dirx * 0.00001f; npy = isy - diry * 0.00001f; npz = isz - dirz * 0.00001f;
                        qx = npx; qy = npy; qz = npz;
                        phase = PH_N0;
                        ws = ST_WAIT_S;
                    } else {
                        qx = origin.x + t * dirx; qy = origin.y + t * diry; qz = origin.z + t * dirz;
                    }
                    START_DE(qx, qy, qz, ws);
                }
            }
            y_mtail += STAMP() - y0;
        }

        // ---------------- shade tail: normal taps (fragment.shd:463-470) and AO taps (542-562) ----------
        if (wait_s != 0ull && (popc64(wait_s) >= shade_t || none_iter)) {
            c_stail++;
            y0 = STAMP();
            if (st == ST_WAIT_S) {
                const float dist = 0.5f * log_pinned(r) * r / dr;   // fragment.shd:157
                const float eps = 0.00001f;
                float qx = npx, qy = npy, qz = npz;
                bool done = false;
                if (phase <= PH_NY) {
                    // N0: c = DE(p); then DE(p - eps*x), DE(p - eps*y), DE(p - eps*z)
                    d0 = (phase == PH_N0) ? dist : d0;
                    d1 = (phase == PH_NX) ? dist : d1;
                    d2 = (phase == PH_NY) ? dist : d2;
                    qx = npx - ((phase == PH_N0) ? eps : 0.0f);
                    qy = npy - ((phase == PH_NX) ? eps : 0.0f);
                    qz = npz - ((phase == PH_NY) ? eps : 0.0f);
                } else if (phase == PH_NZ) {
                    const v3 n = normalize3(mk3(d0 - d1, d0 - d2, d0 - dist));
                    nx = n.x; ny = n.y; nz = n.z;
                    qx = isx + nx * 0.016f; qy = isy + ny * 0.016f; qz = isz + nz * 0.016f;   // AO tap 1
                } else if (phase == PH_AO0) {
                    occl = 0.0f;
                    occl += 0.5f * gclamp(1.0f - dist / 0.016f, 0.0f, 1.0f);
                    qx = isx + nx * 0.081f; qy = isy + ny * 0.081f; qz = isz + nz * 0.081f;   // AO tap 2
                } else {
                    occl += 0.25f * gclamp(1.0f - dist / 0.081f, 0.0f, 1.0f);
                    occl = 1.0f - occl;
                    occl -= 0.29f;
                    occl *= 3.5f;
                    occl *= occl;
                    const float ao = gclamp(occl, 0.0f, 1.0f);
                    p.gbuf_nao[pix] = make_float4(nx, ny, nz, ao);
                    p.gbuf_meta[pix] = (unsigned)steps | 0x8000u | ((iters > 65535u ? 65535u : iters) << 16);
                    done = true;
                }
                if (done) {
                    st = ST_IDLE;
                } else {
                    phase++;
                    START_DE(qx, qy, qz, ST_WAIT_S);
                }
            }
            y_stail += STAMP() - y0;
        }
    }
#undef START_DE
    if (p.dbg && (threadIdx.x & 63) == 0) {
        unsigned long long *d = p.dbg + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
        d[0] = c_iter; d[1] = c_mtail; d[2] = c_stail; d[3] = c_refill; d[4] = a_iter; d[5] = a_mtail;
        d[6] = t_begin; d[7] = __builtin_amdgcn_s_memrealtime();
        d[8] = y_iter; d[9] = y_mtail; d[10] = y_stail; d[11] = y_refill;
    }
}

// ------------------------------------------------------------------------------------------------------
// shading pass
// ------------------------------------------------------------------------------------------------------
struct GPix {
    bool hit;
    v3 dir, n, refl;
    float ao, fresnel;
    unsigned meta;
};

__device__ __forceinline__ v3 pixel_dir(const FrameParams &p, int px, int py)
{
    const float ndcx = ((float)px + 0.5f) / p.wf * 2.0f - 1.0f;
    const float ndcy = ((float)py + 0.5f) / p.hf * 2.0f - 1.0f;
    const v3 dc = normalize3(mk3(ndcx * p.fov_xs, ndcy * p.fov_xs / p.aspect, -1.0f));
    return mk3(p.cam[0] * dc.x + p.cam[3] * dc.y + p.cam[6] * dc.z,
               p.cam[1] * dc.x + p.cam[4] * dc.y + p.cam[7] * dc.z,
               p.cam[2] * dc.x + p.cam[5] * dc.y + p.cam[8] * dc.z);
}

__global__ __launch_bounds__(256) void k_shade(const FrameParams p)
{
    int rx0, ry0, rx1, ry1, pitch, ox, oy;
    size_t obase;
    if (p.n_shard_tiles > 0) {
        const int slot = blockIdx.z;
        const int midx = (int)p.shard_tile[slot];
        const int tx = midx % 8, ty = midx / 8;
        rx0 = (2 * tx * p.w + 7) / 16; rx1 = (2 * (tx + 1) * p.w + 7) / 16;
        ry0 = (2 * ty * p.h + 7) / 16; ry1 = (2 * (ty + 1) * p.h + 7) / 16;
        pitch = rx1 - rx0; ox = rx0; oy = ry0;
        obase = (size_t)slot * (size_t)(p.w / 8) * (size_t)(p.h / 8);
    } else {
        rx0 = p.x0; ry0 = p.y0; rx1 = p.x1; ry1 = p.y1;
        pitch = p.w; ox = 0; oy = 0; obase = 0;
    }
    // one wave = 32 x 2 pixels (16 quads side by side): lane&1 = x bit, lane&2 = y bit; rows of 32 pixels
    // are contiguous -> 128 B RGBA8 / 512 B float4 segments per store instruction
    const int ex0 = rx0 & ~1, ey0 = ry0 & ~1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = ex0 + blockIdx.x * 32 + ((lane & 1) | ((lane >> 2) << 1));
    const int py = ey0 + (blockIdx.y * 4 + wave) * 2 + ((lane >> 1) & 1);
    const int ex1 = (rx1 + 1) & ~1, ey1 = (ry1 + 1) & ~1;
    const bool in_ext = (px < ex1) && (py < ey1);

    // own G-buffer record
    const v3 origin = mk3(p.cam[9], p.cam[10], p.cam[11]);
    (void)origin;
    const v3 dir = pixel_dir(p, px, py);
    unsigned meta = 0u;
    float4 nao = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (in_ext) {
        meta = p.gbuf_meta[px + py * p.gw];
        if (meta & 0x8000u) nao = p.gbuf_nao[px + py * p.gw];
    }
    const bool hit = (meta & 0x8000u) != 0u;
    const v3 n = mk3(nao.x, nao.y, nao.z);
    float fresnel = 0.0f;
    v3 refl = mk3(0.0f, 0.0f, 0.0f);
    if (hit) {
        fresnel = fresnel_conductor<false>(dot3(mk3(-dir.x, -dir.y, -dir.z), n), 0.4f, 0.8f);
        refl = reflect3(dir, n);
    }
    // quad neighbours: lane^1 horizontal, lane^2 vertical
    const int hit_i = hit ? 1 : 0;
    const bool hit_h = __shfl_xor(hit_i, 1, 64) != 0, hit_v = __shfl_xor(hit_i, 2, 64) != 0;
    const v3 n_h = mk3(__shfl_xor(n.x, 1, 64), __shfl_xor(n.y, 1, 64), __shfl_xor(n.z, 1, 64));
    const v3 n_v = mk3(__shfl_xor(n.x, 2, 64), __shfl_xor(n.y, 2, 64), __shfl_xor(n.z, 2, 64));
    const v3 refl_h = mk3(__shfl_xor(refl.x, 1, 64), __shfl_xor(refl.y, 1, 64), __shfl_xor(refl.z, 1, 64));
    const v3 refl_v = mk3(__shfl_xor(refl.x, 2, 64), __shfl_xor(refl.y, 2, 64), __shfl_xor(refl.z, 2, 64));
    const v3 dir_h = mk3(__shfl_xor(dir.x, 1, 64), __shfl_xor(dir.y, 1, 64), __shfl_xor(dir.z, 1, 64));
    const v3 dir_v = mk3(__shfl_xor(dir.x, 2, 64), __shfl_xor(dir.y, 2, 64), __shfl_xor(dir.z, 2, 64));

    v3 color;
    if (hit) {
        const float ao = nao.w;
        const v3 t1 = cube_texture<false>(p.env_cos1, n, hit_h, n_h, hit_v, n_v);
        const v3 t8 = cube_texture<false>(p.env_cos8, refl, hit_h, refl_h, hit_v, refl_v);
        const v3 tr = cube_texture<false>(p.env_refl, refl, hit_h, refl_h, hit_v, refl_v);
        const float diff_weight = 0.5f, spec_weight = 1.0f - 0.5f, npl = (8.0f + 2.0f) / 2.0f;
        color.x = (t1.x * 1.0f * diff_weight + t8.x * 0.8f * npl * fresnel * spec_weight + tr.x * spec_weight * fresnel * 0.1f) * 3.0f * ao;
        color.y = (t1.y * 0.8f * diff_weight + t8.y * 0.8f * npl * fresnel * spec_weight + tr.y * spec_weight * fresnel * 0.1f) * 3.0f * ao;
        color.z = (t1.z * 0.8f * diff_weight + t8.z * 1.0f * npl * fresnel * spec_weight + tr.z * spec_weight * fresnel * 0.1f) * 3.0f * ao;
    } else {
        color = cube_texture<false>(p.env_refl, dir, !hit_h, dir_h, !hit_v, dir_v);   // neighbours in the hit branch: undefined derivative -> minified
    }
    const float inv_gamma = 1.0f / 2.2f;
    const float gr = pow_pinned(color.x, inv_gamma), gg = pow_pinned(color.y, inv_gamma), gb = pow_pinned(color.z, inv_gamma);
    if (in_ext && px >= rx0 && px < rx1 && py >= ry0 && py < ry1) {
        const size_t idx = obase + (size_t)(px - ox) + (size_t)(py - oy) * (size_t)pitch;
        if (p.rgba8) p.rgba8[idx] = to_unorm8(gr) | (to_unorm8(gg) << 8) | (to_unorm8(gb) << 16) | 0xff000000u;
        if (p.rgba_f32) p.rgba_f32[idx] = make_float4(gr, gg, gb, 1.0f);
        if (p.steps) p.steps[idx] = (uint16_t)(meta & 0xffffu);
        if (p.iters) p.iters[idx] = (uint16_t)(meta >> 16);
    }
}

hipError_t launch_shade(const FrameParams &p, int ew, int eh, int nz, hipStream_t stream)
{
    dim3 grid((ew + 31) / 32, (eh + 7) / 8, nz);
    hipLaunchKernelGGL(k_shade, grid, dim3(256), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_render_mb8(const FrameParams &p_in, hipStream_t stream, int num_cus)
{
    FrameParams p = p_in;
    int ew, eh, nz = 1;
    if (p.n_shard_tiles > 0) {
        ew = p.w / 8 + 2; eh = p.h / 8 + 2;              // up to one helper column/row per side
        ew = (ew + 1) & ~1; eh = (eh + 1) & ~1;
        p.items_per_shard_tile = ((ew + 7) / 8) * ((eh + 7) / 8) * 64;
        p.total_items = p.items_per_shard_tile * p.n_shard_tiles;
        nz = p.n_shard_tiles;
    } else {
        const int ex0 = p.x0 & ~1, ey0 = p.y0 & ~1, ex1 = (p.x1 + 1) & ~1, ey1 = (p.y1 + 1) & ~1;
        ew = ex1 - ex0; eh = ey1 - ey0;
        if (ew <= 0 || eh <= 0) return hipSuccess;
        p.items_per_shard_tile = 0;
        p.total_items = ((ew + 7) / 8) * ((eh + 7) / 8) * 64;
    }
    hipError_t e = hipMemsetAsync(p.work_counter, 0, sizeof(int), stream);
    if (e != hipSuccess) return e;
    // persistent grid: `wps` waves per SIMD on every CU.  The march loop is VALU-bound and two waves
    // per SIMD already saturate VALU issue while letting every wave run at the single-wave issue rate,
    // which matters because the longest ray of the frame is a serial chain (DESIGN.md "critical path")
    static int wps = 0, env_tail = 0, env_shade = 0, env_refill = 0, env_chunk = 0;
    if (!wps) {
        const char *e;
        wps = (e = getenv("RMDF_WAVES_PER_SIMD")) ? atoi(e) : 2;
        env_tail = (e = getenv("RMDF_TAIL_T")) ? atoi(e) : 1;
        env_shade = (e = getenv("RMDF_SHADE_T")) ? atoi(e) : 16;
        env_refill = (e = getenv("RMDF_REFILL_T")) ? atoi(e) : 16;
        env_chunk = (e = getenv("RMDF_CHUNK")) ? atoi(e) : 128;
        if (wps < 1) wps = 1;
        if (wps > 8) wps = 8;
    }
    p.tail_t = env_tail; p.shade_t = env_shade; p.refill_t = env_refill; p.chunk = env_chunk;
    int blocks = num_cus * wps;
    const int max_useful = (p.total_items + 255) / 256;
    if (blocks > max_useful) blocks = max_useful;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_march_mb8, dim3(blocks), dim3(256), 0, stream, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_shade(p, ew, eh, nz, stream);
}

}  // namespace rmdf
