// rmdf_kernels.hip -- gfx950 kernels of the sphere tracer and its env-map data prep.
//
// Compile with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no fast-math).
// Geometry of the render kernel: one 64-lane wavefront = one 8x8 pixel packet
// (16 GL-style 2x2 quads, 4 consecutive lanes = one quad so that the quad
// neighbours needed by the cube-map min/mag decision are lane^1 and lane^2),
// one 256-thread workgroup = a 32x8 pixel strip.  No MFMA: the path is scalar
// per ray.  See DESIGN.md for the roofline that bounds it.
#include <stdlib.h>

#include "rmdf_internal.hpp"

namespace rmdf {

// ------------------------------------------------------------------------------------
// tile -> pixel rectangle, ShaderRendering.hs:183-193 (centre-inside rasterisation of
// the NDC rect; exact when 8 divides w and h)
// ------------------------------------------------------------------------------------
__host__ __device__ inline void tile_rect(int tile_idx, int w, int h, int &x0, int &y0, int &x1, int &y1)
{
    int midx = tile_idx % 64;
    int tx = midx % 8, ty = midx / 8;
    // pixel centre x+0.5 in [tx*w/8, (tx+1)*w/8)  <=>  x in [ceil(tx*w/8 - 0.5), ceil((tx+1)*w/8 - 0.5))
    x0 = (2 * tx * w + 7) / 16;        // ceil((2*tx*w - 8) / 16) = floor((2*tx*w + 7) / 16)
    x1 = (2 * (tx + 1) * w + 7) / 16;
    y0 = (2 * ty * h + 7) / 16;
    y1 = (2 * (ty + 1) * h + 7) / 16;
}

template <typename T>
__device__ __forceinline__ T shfl_xor_w(T v, int mask) { return __shfl_xor(v, mask, 64); }

__device__ __forceinline__ v3 shfl_xor3(v3 a, int mask)
{
    return mk3(__shfl_xor(a.x, mask, 64), __shfl_xor(a.y, mask, 64), __shfl_xor(a.z, mask, 64));
}

template <int SCENE>
// hint: Cornell only -- the triangle that was nearest in this lane's previous estimate (evaluation order, not a result)
__device__ __forceinline__ float distance_estimator(v3 pos, const FrameParams &p, unsigned &iters, int &hint)
{
    if (SCENE == 2)      return de_mandelbulb8(pos, iters);
    else if (SCENE == 3) return de_mandelbulb_general(pos, p.power, iters);
    else if (SCENE == 1) return de_test_scene(pos);
    else                 return de_cornell_box_table(pos, p.cornell_tab, p.cornell_prune, hint);
}

template <int SCENE>
__device__ __forceinline__ float bsphere_r() { return SCENE == 2 ? 1.15f : (SCENE == 3 ? 1.5f : 1.0f); }

// ------------------------------------------------------------------------------------
// v1 render kernel: per-lane nested loops (march loop around the DE loop)
// ------------------------------------------------------------------------------------
// MERGE = true: the four waves of a workgroup pool their last few rays.  A packet's march loop keeps running until
// its slowest ray is done, and 29 % of all iteration passes of the headline frame run with <= 16 of 64 lanes.  So
// the first wave of a workgroup that is down to <= MERGE_T active rays becomes the "host"; every other wave that
// gets down to MERGE_T hands its remaining rays (strip pixel, t, steps, iterations) over through an LDS mailbox and
// leaves the march; the host adopts them into its idle lanes and marches everything to the end.  Rays never wait:
// they march in their own wave until handed over, then in the host.  Handed-over rays are all late, near-surface
// rays with similar escape-iteration counts, so the packet coherence the nested loop lives on is kept.  Results go
// through an LDS table indexed by strip pixel; ray arithmetic is untouched (bit-identical output).
#define MERGE_T 32

template <int SCENE, bool MERGE, int WPB>
__global__ __launch_bounds__(WPB * 64) void k_render(const FrameParams p)
{
    // Which strip this workgroup renders: raster order over (slot, row, column), or most expensive first
    // (block_order, a permutation of the launch's linear workgroup ids -- it spans all tiles of a shard launch)
    const unsigned strips_per_slot = gridDim.x * gridDim.y;
    unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (p.block_order) {
        // the first workgroups carry last frame's most expensive strips: the launch cannot end before their
        // longest ray has finished its serial chain, so let their waves win the issue arbitration on the SIMD
        if (lin < (unsigned)p.prio_strips) __builtin_amdgcn_s_setprio(3);
        lin = p.block_order[lin];
    }
    const unsigned strip = lin % strips_per_slot;
    // rectangle of this launch / shard slot
    int rx0, ry0, rx1, ry1, pitch, ox, oy;
    size_t obase;
    if (p.n_shard_tiles > 0) {
        const int slot = (int)(lin / strips_per_slot);
        tile_rect((int)p.shard_tile[slot], p.w, p.h, rx0, ry0, rx1, ry1);
        pitch = rx1 - rx0; ox = rx0; oy = ry0;
        obase = (size_t)slot * (size_t)(p.w / 8) * (size_t)(p.h / 8);
    } else {
        rx0 = p.x0; ry0 = p.y0; rx1 = p.x1; ry1 = p.y1;
        pitch = p.w; ox = 0; oy = 0; obase = 0;
    }
    // GL quads are aligned to even window coordinates: helper pixels outside the
    // rectangle are computed (not written) so that derivatives match a full-frame render
    const int ex0 = rx0 & ~1, ey0 = ry0 & ~1;
    const int ex1 = (rx1 + 1) & ~1, ey1 = (ry1 + 1) & ~1;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = (lane & 1) | (((lane >> 2) & 3) << 1);
    const int ly = ((lane >> 1) & 1) | (((lane >> 4) & 3) << 1);
    const int bx = strip % gridDim.x, by = strip / gridDim.x;
    const int px = ex0 + bx * (WPB * 8) + wave * 8 + lx;
    const int py = ey0 + by * 8 + ly;
    const bool active = (px < ex1) && (py < ey1);
    const unsigned long long dbg_t0 = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;

    // generate_ray, perspective branch (fragment.shd:840-871)
    const float ndcx = ((float)px + 0.5f) / p.wf * 2.0f - 1.0f;
    const float ndcy = ((float)py + 0.5f) / p.hf * 2.0f - 1.0f;
    const v3 dcam = normalize3(mk3(ndcx * p.fov_xs, ndcy * p.fov_xs / p.aspect, -1.0f));
    const v3 dir = mk3(p.cam[0] * dcam.x + p.cam[3] * dcam.y + p.cam[6] * dcam.z,
                       p.cam[1] * dcam.x + p.cam[4] * dcam.y + p.cam[7] * dcam.z,
                       p.cam[2] * dcam.x + p.cam[5] * dcam.y + p.cam[8] * dcam.z);
    const v3 origin = mk3(p.cam[9], p.cam[10], p.cam[11]);

    // ray_march (fragment.shd:618-676)
    bool hit = false;
    int steps = 0;
    unsigned iters = 0;
    int tri_hint = 0;               // Cornell: evaluation-order hint of the distance estimate (never a result)
    float t = 0.0f;
    float tmin, tmax;
    if (!MERGE) {
        if (active && ray_sphere(origin, dir, bsphere_r<SCENE>(), tmin, tmax)) {
            t = gmax(0.0f, tmin);
            for (steps = 0; steps < p.max_steps; steps++) {
                v3 pos = mk3(origin.x + t * dir.x, origin.y + t * dir.y, origin.z + t * dir.z);
                float dist = distance_estimator<SCENE>(pos, p, iters, tri_hint);
                t += dist;
                if (t > tmax) break;
                if (dist < 0.001f) { hit = true; break; }
            }
        }
    } else {
        __shared__ int    s_host, s_nreported;
        __shared__ int    s_mb_n[WPB], s_mb_ready[WPB];
        __shared__ float4 s_mb[WPB][MERGE_T];
        __shared__ float4 s_res[WPB * 64];          // per strip pixel: t, steps | hit << 15, iterations
        if (threadIdx.x == 0) { s_host = -1; s_nreported = 0; }
        if (threadIdx.x < WPB) { s_mb_ready[threadIdx.x] = 0; s_mb_n[threadIdx.x] = 0; }
        const int my_sp = ly * (WPB * 8) + wave * 8 + lx;
        s_res[my_sp] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        __syncthreads();

        // the ray this lane is marching right now (its own, or an adopted one when this wave is the host)
        bool act = false;
        int cur_sp = my_sp, st = 0;
        unsigned it = 0;
        float dx = dir.x, dy = dir.y, dz = dir.z, tt = 0.0f, tmx = 0.0f;
        if (active && ray_sphere(origin, dir, bsphere_r<SCENE>(), tmin, tmax) && p.max_steps > 0) {
            tt = gmax(0.0f, tmin); tmx = tmax; act = true;
        }
        bool is_host = false;
        unsigned taken = 0u;
        unsigned long long cursor = 0ull;      // per-mailbox: fully adopted flag, entries adopted so far (8 bits each)
        const int merge_t = p.merge_stragglers < MERGE_T ? p.merge_stragglers : MERGE_T;
        for (;;) {
            unsigned long long am = __ballot(act);
            int n_act = __popcll(am);
            if (!is_host) {
                if (n_act == 0) {
                    if (lane == 0) __hip_atomic_fetch_add(&s_nreported, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                    break;
                }
                if (n_act <= merge_t) {
                    int h = 0;
                    if (lane == 0) h = atomicCAS(&s_host, -1, wave);
                    h = __builtin_amdgcn_readfirstlane(h);
                    if (h == -1) {
                        is_host = true;
                    } else {
                        if (act) {
                            const int r = __builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0));
                            s_mb[wave][r] = make_float4(__int_as_float(cur_sp), tt, __int_as_float(st), __uint_as_float(it));
                        }
                        if (lane == 0) {
                            s_mb_n[wave] = n_act;
                            __hip_atomic_store(&s_mb_ready[wave], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                            __hip_atomic_fetch_add(&s_nreported, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                        }
                        act = false;
                        break;
                    }
                }
            }
            if (is_host) {
                // adopt what the other waves have handed over so far, as far as idle lanes allow
                for (int w = 0; w < WPB; w++) {
                    if (w == wave || ((taken >> w) & 1u)) continue;
                    if (__hip_atomic_load(&s_mb_ready[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 0) continue;
                    const int n = s_mb_n[w], first = (int)((cursor >> (8 * w)) & 255ull);
                    const unsigned long long idle = __ballot(!act);
                    const int n_idle = __popcll(idle);
                    const int r = __builtin_amdgcn_mbcnt_hi((unsigned)(idle >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)idle, 0));
                    const int take = (n - first) < n_idle ? (n - first) : n_idle;
                    if (!act && r < take) {
                        const float4 e = s_mb[w][first + r];
                        cur_sp = __float_as_int(e.x); tt = e.y; st = __float_as_int(e.z); it = __float_as_uint(e.w);
                        const int qx = ex0 + bx * (WPB * 8) + (cur_sp % (WPB * 8)), qy = ey0 + by * 8 + (cur_sp / (WPB * 8));
                        const float nx_ = ((float)qx + 0.5f) / p.wf * 2.0f - 1.0f, ny_ = ((float)qy + 0.5f) / p.hf * 2.0f - 1.0f;
                        const v3 dc = normalize3(mk3(nx_ * p.fov_xs, ny_ * p.fov_xs / p.aspect, -1.0f));
                        dx = p.cam[0] * dc.x + p.cam[3] * dc.y + p.cam[6] * dc.z;
                        dy = p.cam[1] * dc.x + p.cam[4] * dc.y + p.cam[7] * dc.z;
                        dz = p.cam[2] * dc.x + p.cam[5] * dc.y + p.cam[8] * dc.z;
                        float tmin2;
                        (void)ray_sphere(origin, mk3(dx, dy, dz), bsphere_r<SCENE>(), tmin2, tmx);
                        act = true;
                    }
                    cursor += (unsigned long long)take << (8 * w);
                    if (first + take >= n) taken |= 1u << w;
                }
                am = __ballot(act);
                n_act = __popcll(am);
                if (n_act == 0) {
                    // nothing to march: done once the three other waves have left their march and all mail is taken
                    bool all = __hip_atomic_load(&s_nreported, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == WPB - 1;
                    if (all) {
                        for (int w = 0; w < WPB; w++)
                            if (w != wave && !((taken >> w) & 1u) &&
                                __hip_atomic_load(&s_mb_ready[w], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != 0) all = false;
                    }
                    if (all) break;
                    __builtin_amdgcn_s_sleep(4);
                    continue;
                }
            }
            // one march step (fragment.shd:661-672) for the rays in flight
            if (act) {
                const v3 pos = mk3(origin.x + tt * dx, origin.y + tt * dy, origin.z + tt * dz);
                const float dist = distance_estimator<SCENE>(pos, p, it, tri_hint);
                tt += dist;
                const bool out = tt > tmx;
                const bool h2 = !out && (dist < 0.001f);
                bool done = out || h2;
                if (!done) { st++; done = st >= p.max_steps; }
                if (done) {
                    s_res[cur_sp] = make_float4(tt, __int_as_float(st | (h2 ? 0x8000 : 0)), __uint_as_float(it), 0.0f);
                    act = false;
                }
            }
        }
        __syncthreads();
        const float4 rr = s_res[my_sp];
        t = rr.x;
        const int sb = __float_as_int(rr.y);
        steps = sb & 0x7fff;
        hit = (sb >> 15) != 0;
        iters = __float_as_uint(rr.z);
    }

    // render_ray hit branch up to the texture lookups (fragment.shd:743-799)
    v3 n = mk3(0.0f, 0.0f, 0.0f), refl = mk3(0.0f, 0.0f, 0.0f);
    float ao = 0.0f, fresnel = 0.0f;
    if (hit && p.dbg_skip != 0) { n = mk3(0.0f, 1.0f, 0.0f); ao = 1.0f; refl = reflect3(dir, n); fresnel = 0.5f; }
    if (hit && p.dbg_skip == 0) {
        v3 isec = mk3(origin.x + dir.x * t, origin.y + dir.y * t, origin.z + dir.z * t);
        v3 np = mk3(isec.x - dir.x * 0.00001f, isec.y - dir.y * 0.00001f, isec.z - dir.z * 0.00001f);
        const float eps = 0.00001f;
        float d0 = distance_estimator<SCENE>(np, p, iters, tri_hint);
        float dx = distance_estimator<SCENE>(mk3(np.x - eps, np.y - 0.0f, np.z - 0.0f), p, iters, tri_hint);
        float dy = distance_estimator<SCENE>(mk3(np.x - 0.0f, np.y - eps, np.z - 0.0f), p, iters, tri_hint);
        float dz = distance_estimator<SCENE>(mk3(np.x - 0.0f, np.y - 0.0f, np.z - eps), p, iters, tri_hint);
        n = normalize3(mk3(d0 - dx, d0 - dy, d0 - dz));
        // distance_ao (fragment.shd:542-591)
        float occl = 0.0f;
        if (SCENE != 0) {
            const float w0 = 0.5f, e0 = 0.016f, w1 = 0.25f, e1 = 0.081f;
            occl += w0 * gclamp(1.0f - distance_estimator<SCENE>(mk3(isec.x + n.x * e0, isec.y + n.y * e0, isec.z + n.z * e0), p, iters, tri_hint) / e0, 0.0f, 1.0f);
            occl += w1 * gclamp(1.0f - distance_estimator<SCENE>(mk3(isec.x + n.x * e1, isec.y + n.y * e1, isec.z + n.z * e1), p, iters, tri_hint) / e1, 0.0f, 1.0f);
            occl = 1.0f - occl;
            occl -= 0.29f;
            occl *= 3.5f;
            occl *= occl;
            ao = gclamp(occl, 0.0f, 1.0f);
        } else {
            const float wt[4] = { 0.1f, 0.2f, 0.125f, 0.0625f }, dl[4] = { 0.1f, 0.2f, 0.4f, 0.5f };
#pragma unroll
            for (int k = 0; k < 4; k++)
                occl += wt[k] * gclamp(1.0f - distance_estimator<SCENE>(mk3(isec.x + n.x * dl[k], isec.y + n.y * dl[k], isec.z + n.z * dl[k]), p, iters, tri_hint) / dl[k], 0.0f, 1.0f);
            ao = 1.0f - occl;
        }
        fresnel = fresnel_conductor(dot3(mk3(-dir.x, -dir.y, -dir.z), n), 0.4f, 0.8f);
        refl = reflect3(dir, n);
    }

    // quad neighbours (lane^1 = horizontal, lane^2 = vertical)
    const int hit_i = hit ? 1 : 0;
    const bool hit_h = shfl_xor_w(hit_i, 1) != 0, hit_v = shfl_xor_w(hit_i, 2) != 0;
    const v3 n_h = shfl_xor3(n, 1), n_v = shfl_xor3(n, 2);
    const v3 refl_h = shfl_xor3(refl, 1), refl_v = shfl_xor3(refl, 2);
    const v3 dir_h = shfl_xor3(dir, 1), dir_v = shfl_xor3(dir, 2);

    v3 color;
    if (p.dbg_skip == 1) {
        color = mk3(t, (float)steps, 0.0f);
    } else if (hit) {
        // fragment.shd:799-810
        v3 t1 = cube_texture(p.env_cos1, n, hit_h, n_h, hit_v, n_v);
        v3 t8 = cube_texture(p.env_cos8, refl, hit_h, refl_h, hit_v, refl_v);
        v3 tr = cube_texture(p.env_refl, refl, hit_h, refl_h, hit_v, refl_v);
        const float diff_weight = 0.5f, spec_weight = 1.0f - 0.5f, npl = (8.0f + 2.0f) / 2.0f;
        color.x = (t1.x * 1.0f * diff_weight + t8.x * 0.8f * npl * fresnel * spec_weight + tr.x * spec_weight * fresnel * 0.1f) * 3.0f * ao;
        color.y = (t1.y * 0.8f * diff_weight + t8.y * 0.8f * npl * fresnel * spec_weight + tr.y * spec_weight * fresnel * 0.1f) * 3.0f * ao;
        color.z = (t1.z * 0.8f * diff_weight + t8.z * 1.0f * npl * fresnel * spec_weight + tr.z * spec_weight * fresnel * 0.1f) * 3.0f * ao;
    } else {
        // fragment.shd:823
        color = cube_texture(p.env_refl, dir, !hit_h, dir_h, !hit_v, dir_v);   // neighbours in the hit branch: undefined derivative -> minified
    }

    // fragment.shd:959-960 and the RGBA8 conversion of the colour attachment
    const float inv_gamma = 1.0f / 2.2f;
    const float gr = pow_pinned(color.x, inv_gamma), gg = pow_pinned(color.y, inv_gamma), gb = pow_pinned(color.z, inv_gamma);
    // Stage the 32x8 strip in LDS so that every store instruction writes whole 128-byte lines (a wave's own
    // 8x8 packet would write 32-byte pieces of 8 different rows).  Thread t stores pixel (t % 32, t / 32).
    __shared__ uint32_t s_rgba8[8][WPB * 8];
    __shared__ float4   s_f32[8][WPB * 8];
    __shared__ uint32_t s_meta[8][WPB * 8];
    {
        const int sx = wave * 8 + lx;
        s_rgba8[ly][sx] = to_unorm8(gr) | (to_unorm8(gg) << 8) | (to_unorm8(gb) << 16) | 0xff000000u;
        if (p.rgba_f32) s_f32[ly][sx] = make_float4(gr, gg, gb, 1.0f);
        s_meta[ly][sx] = (uint32_t)(steps | (hit_i << 15)) | ((iters > 65535u ? 65535u : iters) << 16);
    }
    __syncthreads();
    {
        const int ox_ = threadIdx.x % (WPB * 8), oy_ = threadIdx.x / (WPB * 8);
        const int qx = ex0 + bx * (WPB * 8) + ox_, qy = ey0 + by * 8 + oy_;
        if (qx >= rx0 && qx < rx1 && qy >= ry0 && qy < ry1) {
            const size_t idx = obase + (size_t)(qx - ox) + (size_t)(qy - oy) * (size_t)pitch;
            if (p.rgba8) p.rgba8[idx] = s_rgba8[oy_][ox_];
            if (p.rgba8_mirror) p.rgba8_mirror[idx] = s_rgba8[oy_][ox_];
            if (p.rgba_f32) p.rgba_f32[idx] = s_f32[oy_][ox_];
            const uint32_t m = s_meta[oy_][ox_];
            if (p.steps) p.steps[idx] = (uint16_t)(m & 0xffffu);
            if (p.iters) p.iters[idx] = (uint16_t)(m >> 16);
        }
    }
    if (p.block_cost) {
        // cost of the strip = the largest escape-iteration total of one of its pixels (proxy of its longest
        // serial chain); only steers next frame's dispatch order, never the image
        __shared__ unsigned s_cost;
        if (threadIdx.x == 0) s_cost = 0u;
        __syncthreads();
        unsigned c = iters + (unsigned)steps;
        for (int o = 32; o > 0; o >>= 1) { const unsigned v = __shfl_xor(c, o, 64); c = v > c ? v : c; }
        if (lane == 0) atomicMax(&s_cost, c);
        __syncthreads();
        if (threadIdx.x == 0) p.block_cost[lin] = s_cost;
    }
    if (p.dbg && lane == 0) {
        const unsigned wid = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
        if (wid < 32768u * 2u) { p.dbg[wid * 8 + 6] = dbg_t0; p.dbg[wid * 8 + 7] = __builtin_amdgcn_s_memrealtime(); p.dbg[wid * 8] = (unsigned long long)steps; }
    }
}

// measurement aid: the march loop of k_render<2> alone, with wave-level divergence counters
__global__ __launch_bounds__(256) void k_march_stats(const FrameParams p)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = (lane & 1) | (((lane >> 2) & 3) << 1);
    const int ly = ((lane >> 1) & 1) | (((lane >> 4) & 3) << 1);
    const int px = blockIdx.x * 32 + wave * 8 + lx, py = blockIdx.y * 8 + ly;
    const bool active = (px < p.w) && (py < p.h);
    const float ndcx = ((float)px + 0.5f) / p.wf * 2.0f - 1.0f;
    const float ndcy = ((float)py + 0.5f) / p.hf * 2.0f - 1.0f;
    const v3 dcam = normalize3(mk3(ndcx * p.fov_xs, ndcy * p.fov_xs / p.aspect, -1.0f));
    const v3 dir = mk3(p.cam[0] * dcam.x + p.cam[3] * dcam.y + p.cam[6] * dcam.z,
                       p.cam[1] * dcam.x + p.cam[4] * dcam.y + p.cam[7] * dcam.z,
                       p.cam[2] * dcam.x + p.cam[5] * dcam.y + p.cam[8] * dcam.z);
    const v3 origin = mk3(p.cam[9], p.cam[10], p.cam[11]);
    unsigned long long passes = 0, lanes = 0, wsteps = 0, wlanes = 0;
    unsigned iters = 0;
    float t = 0.0f, tmin, tmax;
    int steps = 0;
    bool hit = false;
    if (active && ray_sphere(origin, dir, 1.15f, tmin, tmax)) {
        t = gmax(0.0f, tmin);
        for (steps = 0; steps < p.max_steps; steps++) {
            wsteps++; wlanes += __popcll(__ballot(true));
            v3 pos = mk3(origin.x + t * dir.x, origin.y + t * dir.y, origin.z + t * dir.z);
            float dist = de_mandelbulb8_dbg(pos, iters, passes, lanes, p.dbg ? p.dbg + 8 : nullptr);
            t += dist;
            if (t > tmax) break;
            if (dist < 0.001f) { hit = true; break; }
        }
    }
    // wave totals: every lane counted the passes it took part in; the wave-level count is the max
    unsigned long long wp = passes, ws = wsteps;
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long a = __shfl_xor(wp, o, 64), b = __shfl_xor(ws, o, 64);
        wp = a > wp ? a : wp; ws = b > ws ? b : ws;
    }
    unsigned long long li = iters, lh = hit ? 1 : 0, lst = (unsigned long long)(steps);
    for (int o = 32; o > 0; o >>= 1) { li += __shfl_xor(li, o, 64); lh += __shfl_xor(lh, o, 64); lst += __shfl_xor(lst, o, 64); }
    if (lane == 0 && p.dbg) {
        atomicAdd(&p.dbg[0], wp);       // wave-level inner passes (max over lanes is a lower bound of the true count)
        atomicAdd(&p.dbg[1], li);       // lane iterations
        atomicAdd(&p.dbg[2], ws);       // wave-level march steps
        atomicAdd(&p.dbg[3], lst);      // lane steps (sum of loop counters)
        atomicAdd(&p.dbg[4], lh);
        atomicAdd(&p.dbg[5], 1ull);
    }
}

hipError_t launch_march_stats(const FrameParams &p, hipStream_t stream)
{
    dim3 grid((p.w + 31) / 32, (p.h + 7) / 8), block(256);
    hipLaunchKernelGGL(k_march_stats, grid, block, 0, stream, p);
    return hipGetLastError();
}

static int render_wpb(int scene)
{
    // waves per workgroup of k_render: RMDF_WPB (measurement knob) for the power-8 scene, 4 elsewhere
    static int wpb = 0;
    if (!wpb) { const char *e = getenv("RMDF_WPB"); wpb = (e && atoi(e) == 8) ? 8 : 4; }
    return scene == 2 ? wpb : 4;
}

static void render_grid(const FrameParams &p, dim3 &grid, int wpb)
{
    int rx0, ry0, rx1, ry1, nz = 1;
    if (p.n_shard_tiles > 0) {
        rx0 = 0; ry0 = 0; rx1 = p.w / 8 + ((p.w / 8) & 1); ry1 = p.h / 8 + ((p.h / 8) & 1);   // odd-sized tiles: one helper column / row, on one side only
        nz = p.n_shard_tiles;
    } else {
        rx0 = p.x0; ry0 = p.y0; rx1 = p.x1; ry1 = p.y1;
    }
    const int ex0 = rx0 & ~1, ey0 = ry0 & ~1, ex1 = (rx1 + 1) & ~1, ey1 = (ry1 + 1) & ~1;
    if (ex1 <= ex0 || ey1 <= ey0) { grid = dim3(0, 0, 0); return; }
    grid = dim3((ex1 - ex0 + wpb * 8 - 1) / (wpb * 8), (ey1 - ey0 + 7) / 8, nz);
}

int render_grid_blocks(int scene, const FrameParams &p)
{
    dim3 g;
    render_grid(p, g, render_wpb(scene));
    return (int)(g.x * g.y * g.z);
}

// Counting sort of the strips by descending cost (256 logarithmic-ish bins): order[rank] = strip.
// One workgroup; ~n/1024 elements per thread.  Longest-processing-time-first dispatch needs no exact order.
__global__ __launch_bounds__(1024) void k_order_blocks(const unsigned *__restrict__ cost, int n, unsigned *__restrict__ order)
{
    __shared__ unsigned hist[256], base[256];
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0u;
    __syncthreads();
    auto bin_of = [](unsigned c) -> unsigned {
        // 8 sub-bins per power of two: monotone in c, 0..255
        if (c < 8u) return c;
        const int e = 31 - __builtin_clz(c);              // >= 3
        const unsigned b = (unsigned)(e - 2) * 8u + ((c >> (e - 3)) & 7u);
        return b > 255u ? 255u : b;
    };
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[bin_of(cost[i])], 1u);
    __syncthreads();
    if (tid == 0) {
        unsigned acc = 0u;
        for (int b = 255; b >= 0; b--) { base[b] = acc; acc += hist[b]; }   // descending cost
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) order[atomicAdd(&base[bin_of(cost[i])], 1u)] = (unsigned)i;
}

hipError_t launch_order_blocks(const unsigned *d_cost, int n, unsigned *d_order, hipStream_t stream)
{
    hipLaunchKernelGGL(k_order_blocks, dim3(1), dim3(1024), 0, stream, d_cost, n, d_order);
    return hipGetLastError();
}

hipError_t launch_render(int scene, const FrameParams &p, hipStream_t stream)
{
    int rx0, ry0, rx1, ry1, nz = 1;
    if (p.n_shard_tiles > 0) {
        // all tiles have the same size when 8 | w and 8 | h (required in shard mode)
        // a tile may start and end on odd coordinates: up to one helper column/row per side
        rx0 = 0; ry0 = 0; rx1 = p.w / 8 + ((p.w / 8) & 1); ry1 = p.h / 8 + ((p.h / 8) & 1);   // odd-sized tiles: one helper column / row, on one side only
        nz = p.n_shard_tiles;
    } else {
        rx0 = p.x0; ry0 = p.y0; rx1 = p.x1; ry1 = p.y1;
    }
    const int ex0 = rx0 & ~1, ey0 = ry0 & ~1, ex1 = (rx1 + 1) & ~1, ey1 = (ry1 + 1) & ~1;
    if (ex1 <= ex0 || ey1 <= ey0) return hipSuccess;
    const int wpb = render_wpb(scene);
    dim3 grid((ex1 - ex0 + wpb * 8 - 1) / (wpb * 8), (ey1 - ey0 + 7) / 8, nz);
    // RMDF_OCC_LDS (measurement knob): dynamic LDS bytes per workgroup, only to cap waves per SIMD
    static int occ_lds = -1;
    if (occ_lds < 0) { const char *e = getenv("RMDF_OCC_LDS"); occ_lds = e ? atoi(e) : 0; }
    const bool merge = p.merge_stragglers != 0;
#define RMDF_LAUNCH(SC, W)                                                                                   \
    do {                                                                                                     \
        if (merge) hipLaunchKernelGGL((k_render<SC, true, W>), grid, dim3(W * 64), occ_lds, stream, p);      \
        else       hipLaunchKernelGGL((k_render<SC, false, W>), grid, dim3(W * 64), occ_lds, stream, p);     \
    } while (0)
    if (scene == 2)      { if (wpb == 8) RMDF_LAUNCH(2, 8); else RMDF_LAUNCH(2, 4); }
    else if (scene == 0) RMDF_LAUNCH(0, 4);
    else if (scene == 1) RMDF_LAUNCH(1, 4);
    else if (scene == 3) RMDF_LAUNCH(3, 4);
#undef RMDF_LAUNCH
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// Self-test of the short correctly-rounded sequences of rmdf_device.hpp against hipcc's own IEEE expansions,
// over ALL 2^32 float bit patterns: sqrt_rn vs sqrtf, rcp_rn vs 1.0f/x, log_pinned (whose internal quotient uses
// div_known_range) vs the same algorithm with the compiler's division.  counts[k] = number of differing inputs
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
    unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (uint64_t i = tid; i < (1ull << 32); i += stride) {
        const float x = __uint_as_float((uint32_t)i);
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
        c[5] += differ(atan2_pinned(x, y), atan2_full(x, y));
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

hipError_t launch_selftest_exact_math(unsigned long long *d_counts, const float *d_cornell_tab, hipStream_t stream)
{
    hipLaunchKernelGGL(k_selftest_exact_math, dim3(8192), dim3(256), 0, stream, d_counts);
    hipLaunchKernelGGL(k_selftest_cornell_div, dim3(256, 96), dim3(256), 0, stream, d_counts, d_cornell_tab);
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

// ------------------------------------------------------------------------------------
// RGB16F upload with seamless border (the texImage2D RGB16F of HDREnvMap.hs:160-161 +
// GL_TEXTURE_CUBE_MAP_SEAMLESS, :126).  Border rule: DESIGN.md "spec pins".
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void cube_int_dir(int face, int W, int cw, int ch, int d[3])
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

// texel (x,y) of `face` with exactly one coordinate out of range by one -> the texel
// adjacent across the cube edge
__device__ __forceinline__ void cube_fold(int face, int W, int x, int y, int &nf, int &nx, int &ny)
{
    int d[3];
    cube_int_dir(face, W, 2 * x + 1 - W, 2 * y + 1 - W, d);
    int major = face >> 1;
    int over = -1;
#pragma unroll
    for (int a = 0; a < 3; a++)
        if (a != (face >> 1) && over < 0 && (d[a] > W || d[a] < -W)) over = a;
    if (over >= 0) {
        int keep = (d[major] > 0) ? W - 1 : -(W - 1);
        int nd[3] = { d[0], d[1], d[2] };
        nd[over] = (d[over] > 0) ? W : -W;
        nd[major] = keep;
        d[0] = nd[0]; d[1] = nd[1]; d[2] = nd[2];
        major = over;
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
    nf = f; nx = (cw + W - 1) / 2; ny = (ch + W - 1) / 2;
}

__device__ __forceinline__ __half src_half(const float *faces, int W, int f, int x, int y, int k)
{
    return __float2half_rn(faces[(((size_t)f * W + y) * W + x) * 3 + k]);
}

__global__ void k_cube_upload(const float *__restrict__ faces, int W, uint2 *__restrict__ padded)
{
    const int P = W + 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 6 * P * P) return;
    const int X = i % P, Y = (i / P) % P, f = i / (P * P);
    const int x = X - 1, y = Y - 1;
    const bool ox = (x < 0 || x >= W), oy = (y < 0 || y >= W);
    __half c[3];
    if (!ox && !oy) {
        for (int k = 0; k < 3; k++) c[k] = src_half(faces, W, f, x, y, k);
    } else if (ox != oy) {
        int nf, nx, ny;
        cube_fold(f, W, x, y, nf, nx, ny);
        for (int k = 0; k < 3; k++) c[k] = src_half(faces, W, nf, nx, ny, k);
    } else {
        const int cx = x < 0 ? 0 : W - 1, cy = y < 0 ? 0 : W - 1;
        int f1, x1, y1, f2, x2, y2;
        cube_fold(f, W, x, cy, f1, x1, y1);
        cube_fold(f, W, cx, y, f2, x2, y2);
        for (int k = 0; k < 3; k++) {
            float a = __half2float(src_half(faces, W, f, cx, cy, k));
            float b = __half2float(src_half(faces, W, f1, x1, y1, k));
            float cc = __half2float(src_half(faces, W, f2, x2, y2, k));
            c[k] = __float2half_rn(((a + b) + cc) / 3.0f);
        }
    }
    uint2 t;
    t.x = (uint32_t)__half_as_ushort(c[0]) | ((uint32_t)__half_as_ushort(c[1]) << 16);
    t.y = (uint32_t)__half_as_ushort(c[2]);
    padded[i] = t;
}

hipError_t launch_cube_upload(const float *d_faces_f32, int W, uint2 *d_padded, hipStream_t stream)
{
    const int n = 6 * (W + 2) * (W + 2);
    hipLaunchKernelGGL(k_cube_upload, dim3((n + 255) / 256), dim3(256), 0, stream, d_faces_f32, W, d_padded);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// latLongHDREnvMapToCubeMap (HDREnvMap.hs:118-163) on the device: one thread per texel.
// acosf / atanf are the device libm's (the reference calls glibc's through GHC), so
// this kernel is tolerance-checked, not bit-checked.
// ------------------------------------------------------------------------------------
#define RMDF_PI_F 3.14159265358979323846f

// pixelAtBilinear, HDREnvMap.hs:91-113 (keeps the `mod (w-1)` / `min (h-1)` quirks)
__device__ __forceinline__ v3 pixel_at_bilinear(const float *__restrict__ img, int w, int h, float u, float v)
{
    const float upx = u * ((float)w - 1.0f), upy = v * ((float)h - 1.0f);
    const int x = (int)floorf(upx), y = (int)floorf(upy);
    const int m = w - 1;
    int xp1 = (x + 1) % m;
    if (xp1 < 0) xp1 += m;
    const int yp1 = (y + 1 < h - 1) ? y + 1 : h - 1;
    const float ur = upx - (float)x, vr = upy - (float)y;
    const float uo = 1.0f - ur, vo = 1.0f - vr;
    const float *a = img + ((size_t)x + (size_t)y * w) * 3, *b = img + ((size_t)xp1 + (size_t)y * w) * 3;
    const float *c = img + ((size_t)x + (size_t)yp1 * w) * 3, *d = img + ((size_t)xp1 + (size_t)yp1 * w) * 3;
    return mk3((a[0] * uo + b[0] * ur) * vo + (c[0] * uo + d[0] * ur) * vr,
               (a[1] * uo + b[1] * ur) * vo + (c[1] * uo + d[1] * ur) * vr,
               (a[2] * uo + b[2] * ur) * vo + (c[2] * uo + d[2] * ur) * vr);
}

// GHC's class-default RealFloat atan2 (Float has no specialised one)
__device__ float hs_atan2f(float y, float x)
{
    if (x > 0.0f) return atanf(y / x);
    if (x == 0.0f && y > 0.0f) return RMDF_PI_F / 2.0f;
    if (x < 0.0f && y > 0.0f) return RMDF_PI_F + atanf(y / x);
    if ((x <= 0.0f && y < 0.0f) || (x < 0.0f && y == 0.0f && signbit(y)) ||
        (x == 0.0f && signbit(x) && y == 0.0f && signbit(y))) {
        // -atan2 (-y) x : one level of recursion suffices (-y > 0 or -y == +0)
        const float ny = -y;
        if (x == 0.0f && ny > 0.0f) return -(RMDF_PI_F / 2.0f);
        if (x < 0.0f && ny > 0.0f) return -(RMDF_PI_F + atanf(ny / x));
        if (ny == 0.0f && (x < 0.0f || (x == 0.0f && signbit(x)))) return -RMDF_PI_F;
        return -ny;
    }
    if (y == 0.0f && (x < 0.0f || (x == 0.0f && signbit(x)))) return RMDF_PI_F;
    if (x == 0.0f && y == 0.0f) return y;
    return x + y;
}

__global__ void k_latlong_to_cube(const float *__restrict__ latlong, int w, int h, int cw, float *__restrict__ faces)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 6 * cw * cw) return;
    const int x = i % cw, y = (i / cw) % cw, face = i / (cw * cw);
    // cubeMapPixelToDir, HDREnvMap.hs:76-87
    const float vw = ((float)x + 0.5f) / (float)cw * 2.0f - 1.0f;
    const float vh = ((float)y + 0.5f) / (float)cw * 2.0f - 1.0f;
    v3 d;
    switch (face) {
    case 0:  d = mk3(1.0f, -vh, -vw); break;
    case 1:  d = mk3(-1.0f, -vh, vw); break;
    case 2:  d = mk3(vw, 1.0f, vh); break;
    case 3:  d = mk3(vw, -1.0f, -vh); break;
    case 4:  d = mk3(vw, -vh, 1.0f); break;
    default: d = mk3(-vw, -vh, -1.0f); break;
    }
    // Linear.normalize: unchanged if |l| or |1-l| <= 1e-6
    const float l = d.x * d.x + d.y * d.y + d.z * d.z;
    if (!(fabsf(l) <= 1e-6f || fabsf(1.0f - l) <= 1e-6f)) {
        const float s = sqrtf(l);
        d = mk3(d.x / s, d.y / s, d.z / s);
    }
    // worldToLocal (CoordTransf.hs:46-50), cartesianToSpherical (35-44)
    const v3 loc = mk3((d.x * 1.0f + d.y * 0.0f) + d.z * 0.0f,
                       (d.x * 0.0f + d.y * 0.0f) + d.z * -1.0f,
                       (d.x * 0.0f + d.y * 1.0f) + d.z * 0.0f);
    float cz = loc.z;
    if (cz > 1.0f) cz = 1.0f;
    if (cz < -1.0f) cz = -1.0f;
    const float theta = acosf(cz);
    const float p2 = hs_atan2f(loc.y, loc.x);
    const float p1 = (p2 < 0.0f) ? p2 + 2.0f * RMDF_PI_F : p2;
    const float phi = (p1 == 2.0f * RMDF_PI_F) ? 0.0f : p1;
    // sphericalToEnvironmentUV (CoordTransf.hs:60-70)
    const float q1 = phi + RMDF_PI_F / 2.0f;
    const float q2 = (q1 > 2.0f * RMDF_PI_F) ? q1 - 2.0f * RMDF_PI_F : q1;
    const float q3 = 2.0f * RMDF_PI_F - q2;
    const float u = q3 / (RMDF_PI_F * 2.0f), v = theta / RMDF_PI_F;
    const v3 c = pixel_at_bilinear(latlong, w, h, u, v);
    faces[(size_t)i * 3 + 0] = c.x; faces[(size_t)i * 3 + 1] = c.y; faces[(size_t)i * 3 + 2] = c.z;
}

hipError_t launch_latlong_to_cube(const float *d_latlong, int w, int h, float *d_faces_f32, hipStream_t stream)
{
    const int cw = w / 3, n = 6 * cw * cw;
    if (n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_latlong_to_cube, dim3((n + 255) / 256), dim3(256), 0, stream, d_latlong, w, h, cw, d_faces_f32);
    return hipGetLastError();
}

// resizeHDRImage (HDREnvMap.hs:169-195): one thread per destination pixel
__global__ void k_resize_latlong(const float *__restrict__ src, int sw, int sh, int dstw, int dsth, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dstw * dsth) return;
    const int dx = i % dstw, dy = i / dstw;
    const float scale = (float)sw / (float)dstw;
    const int taps = (int)ceilf(scale);
    const float ntaps = (float)(taps * taps);
    const float step = scale / (float)taps;
    const float srcx1 = (float)dx * scale, srcy1 = (float)dy * scale;
    float ar = 0.0f, ag = 0.0f, ab = 0.0f;
    for (int y = 0; y < taps; y++)
        for (int x = 0; x < taps; x++) {
            const float sx = srcx1 + (float)x * step, sy = srcy1 + (float)y * step;
            const v3 c = pixel_at_bilinear(src, sw, sh, sx / ((float)sw - 1.0f), sy / ((float)sh - 1.0f));
            ar = ar + c.x; ag = ag + c.y; ab = ab + c.z;
        }
    out[(size_t)i * 3 + 0] = ar / ntaps; out[(size_t)i * 3 + 1] = ag / ntaps; out[(size_t)i * 3 + 2] = ab / ntaps;
}

hipError_t launch_resize_latlong(const float *d_src, int sw, int sh, int dstw, int dsth, float *d_out, hipStream_t stream)
{
    const int n = dstw * dsth;
    if (n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_resize_latlong, dim3((n + 255) / 256), dim3(256), 0, stream, d_src, sw, sh, dstw, dsth, d_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// cosineConvolveHDREnvMap (HDREnvMap.hs:217-254): O(n^4) -- every destination texel sums sin(theta)*cos^p over
// all source texels with a positive cosine.  Workgroup = 64 destination columns x 8 row groups (512 threads):
// lane = destination column dx, wave g sums the source rows [g*h/8, (g+1)*h/8) in the reference's order (y outer,
// x inner); the 8 partial sums are added in row order, so the result differs from a single serial sum only by the
// re-association at 7 points (tolerance parity).  The source row and per-row cos/sin are wave-uniform
// (scalar loads); cos|phiL - phi_x| is staged once per workgroup in LDS, transposed so lanes read consecutive
// words.  LOG2P >= 0: power = 2^LOG2P <= 8 by repeated squaring
// (float for the reference's 1 and 8, double for its 64 and 512); -1: powf (any other power).
// ------------------------------------------------------------------------------------
template <int LOG2P>
__global__ __launch_bounds__(512) void k_prefilter(const float *__restrict__ src, int w, int h, float power,
                                                   float *__restrict__ out)
{
    extern __shared__ float lut[];            // [w][64]: lut[x*64 + lane] = cos|phiL(lane) - phi(x)|, then 8x64x4 partials
    float *part = lut + (size_t)w * 64;
    // the wave index is wave-uniform, but the compiler cannot know that of threadIdx.x >> 6: say so, or every address
    // derived from it (the source row) is treated as divergent and read with per-lane vector loads
    const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int dx = blockIdx.x * 64 + lane, dy = blockIdx.y;
    const int dxc = dx < w ? dx : w - 1;
    const float theta_l = (float)dy / (float)(h - 1) * RMDF_PI_F;
    const float lc = cosf(theta_l), ls = sinf(theta_l);
    const float phi_l = (float)dxc / (float)(w - 1) * 2.0f * RMDF_PI_F;
    for (int x = g; x < w; x += 8) lut[x * 64 + lane] = cosf(fabsf(phi_l - (float)x / (float)(w - 1) * 2.0f * RMDF_PI_F));
    __syncthreads();
    float ar = 0.0f, ag = 0.0f, ab = 0.0f, n = 0.0f;
    const int y0 = (int)((long long)g * h / 8), y1 = (int)((long long)(g + 1) * h / 8);
    for (int y = y0; y < y1; y++) {
        const float th = (float)y / (float)(h - 1) * RMDF_PI_F;
        const float pc = cosf(th), ps = sinf(th);
        // wave-uniform, read-only: through the constant address space these become scalar loads (see de_cornell_box_table)
        typedef const float __attribute__((address_space(4))) cfloat;
        cfloat *row = (cfloat *)(src + (size_t)y * w * 3);
        if (LOG2P >= 0) {
            // Branch-free and unrolled: the texels with a non-positive cosine contribute a selected +0 (x + 0 == x: the
            // same bits as skipping them) and the sample count grows by a selected 1 or 0, so eight iterations' LDS
            // reads, scalar row loads and multiplies can be in flight at once instead of one load -> wait -> branch per
            // texel (measured: 17.1 -> 5 ms at 512x256).  Non-finite source texels would differ from the skipping form
            // (0 * inf); Radiance RGBE cannot encode them.
#pragma unroll 8
            for (int x = 0; x < w; x++) {
                const float cos_angle = lc * pc + ls * ps * lut[x * 64 + lane];
                const bool pos = cos_angle > 0.0f;
                float cp;
                if (LOG2P <= 3) {
                    cp = cos_angle;
#pragma unroll
                    for (int k = 0; k < LOG2P; k++) cp = cp * cp;
                } else {
                    // 64, 512: float squaring would double the rounding error LOG2P times; in double the chain is
                    // exact to ~2^LOG2P * 1e-16, i.e. the float result is the correctly rounded power (FP64 vector
                    // multiplies issue at the FP32 rate on gfx950), for a fifth of powf's instructions
                    double cd = (double)cos_angle;
#pragma unroll
                    for (int k = 0; k < LOG2P; k++) cd = cd * cd;
                    cp = (float)cd;
                }
                const float fac = ps * cp;
                const float tr = row[x * 3] * fac, tg = row[x * 3 + 1] * fac, tb = row[x * 3 + 2] * fac;
                ar = ar + (pos ? tr : 0.0f); ag = ag + (pos ? tg : 0.0f); ab = ab + (pos ? tb : 0.0f);
                n = n + (pos ? 1.0f : 0.0f);
            }
        } else {
            for (int x = 0; x < w; x++) {
                const float cos_angle = lc * pc + ls * ps * lut[x * 64 + lane];
                if (cos_angle > 0.0f) {
                    const float fac = ps * powf(cos_angle, power);
                    ar = ar + row[x * 3] * fac; ag = ag + row[x * 3 + 1] * fac; ab = ab + row[x * 3 + 2] * fac;
                    n = n + 1.0f;
                }
            }
        }
    }
    float *pp = part + (g * 64 + lane) * 4;
    pp[0] = ar; pp[1] = ag; pp[2] = ab; pp[3] = n;
    __syncthreads();
    if (g == 0 && dx < w) {
        float sr = 0.0f, sg = 0.0f, sb = 0.0f, sn = 0.0f;
        for (int k = 0; k < 8; k++) {
            const float *q = part + (k * 64 + lane) * 4;
            sr = sr + q[0]; sg = sg + q[1]; sb = sb + q[2]; sn = sn + q[3];
        }
        float *o = out + ((size_t)dx + (size_t)dy * w) * 3;
        o[0] = sr / sn; o[1] = sg / sn; o[2] = sb / sn;
    }
}

hipError_t launch_prefilter(const float *d_src, int w, int h, float power, float *d_out, hipStream_t stream)
{
    if (w < 2 || h < 2) return hipErrorInvalidValue;
    const size_t lds = (size_t)w * 64 * sizeof(float) + 8 * 64 * 4 * sizeof(float);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    int log2p = -1;
    // repeated squaring: in float up to 2^3 (the error doubles per step: <= 5e-7), in double for 64 and 512
    for (int k = 0; k <= 3; k++) if (power == (float)(1 << k)) log2p = k;
    if (power == 64.0f) log2p = 6;
    if (power == 512.0f) log2p = 9;
    const void *fn = log2p == 0 ? (const void *)k_prefilter<0> : log2p == 3 ? (const void *)k_prefilter<3> :
                     log2p == 6 ? (const void *)k_prefilter<6> : log2p == 9 ? (const void *)k_prefilter<9> : (const void *)k_prefilter<-1>;
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    dim3 grid((w + 63) / 64, h), block(512);
    if (log2p == 0)      hipLaunchKernelGGL(k_prefilter<0>, grid, block, lds, stream, d_src, w, h, power, d_out);
    else if (log2p == 3) hipLaunchKernelGGL(k_prefilter<3>, grid, block, lds, stream, d_src, w, h, power, d_out);
    else if (log2p == 6) hipLaunchKernelGGL(k_prefilter<6>, grid, block, lds, stream, d_src, w, h, power, d_out);
    else if (log2p == 9) hipLaunchKernelGGL(k_prefilter<9>, grid, block, lds, stream, d_src, w, h, power, d_out);
    else                 hipLaunchKernelGGL(k_prefilter<-1>, grid, block, lds, stream, d_src, w, h, power, d_out);
    return hipGetLastError();
}

}  // namespace rmdf
