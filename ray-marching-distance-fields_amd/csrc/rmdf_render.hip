// rmdf_render.hip -- the sphere-tracing render kernel of librmdf.so (gfx950).
//
// Compile with: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off (no fast-math).
// Geometry of the render kernel: one 64-lane wavefront = one 8x8 pixel packet
// (16 GL-style 2x2 quads, 4 consecutive lanes = one quad so that the quad
// neighbours needed by the cube-map min/mag decision are lane^1 and lane^2),
// one 256-thread workgroup = a 32x8 pixel strip.  No MFMA: the path is scalar
// per ray.  See DESIGN.md for the roofline that bounds it.
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

// The underflow bound of the folded Mandelbulb passes (rmdf_device.hpp).  A constant in the product; the cross-check build reads it
// from the frame parameters so that a test can force every estimate through its written fall-back (RMDF_FLAG_FORCE_WRITTEN).
__device__ __forceinline__ float fold_min_of(const FrameParams &p)
{
#ifdef RMDF_XCHECK
    return p.fold_min;
#else
    return RMDF_MB8_FOLD_MIN;
#endif
}

template <int SCENE>
// hint: Cornell only -- the triangle that was nearest in this lane's previous estimate (evaluation order, not a result)
__device__ __forceinline__ float distance_estimator(v3 pos, const FrameParams &p, unsigned &iters, int &hint, const unsigned *cgrid = nullptr, const float *lds_rows = nullptr)
{
    // Cornell box inside k_render: the per-lane form on the LDS copy of the table, candidates from the fine grid (cgrid = its global
    // address); the wave-uniform form (de_cornell_box_table) stays the estimate of the cross-check schedules and of RMDF_FLAG_NO_PRUNE
    if (SCENE == 0 && cgrid && lds_rows) return de_cornell_box_lanes(pos, lds_rows, cgrid, hint);
    if (SCENE == 2)      return de_mandelbulb8(pos, iters, fold_min_of(p));
    else if (SCENE == 3) return de_mandelbulb_general(pos, p.power, iters);
    else if (SCENE == 1) return de_test_scene(pos);
    else                 return de_cornell_box_table(pos, p.cornell_tab, p.cornell_prune, hint, cgrid);
}

template <int SCENE>
__device__ __forceinline__ float bsphere_r() { return SCENE == 2 ? shk::bsphere_r_power8 : (SCENE == 3 ? shk::bsphere_r_general : shk::bsphere_r_other); }

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
#ifdef RMDF_AB_MERGE_T              // A/B switch (tools/abtest/mt48.so; tools/emulated_schedule.py PREDICTS -2.8 % issued instructions for 48: not yet timed)
#define MERGE_T RMDF_AB_MERGE_T
#else
#define MERGE_T 32
#endif
// Distance-AO estimates of the power-8 Mandelbulb (MERGE variant): a sample point 0.016 / 0.081 off the surface lands INSIDE another
// part of the set for 0.7 % / 4.2 % of the hit pixels and then runs all 25 iterations, while the mean is 3.8 -- so nearly every
// wave's two AO estimates ran ~21 passes for ~7.5 lanes' worth of work (tools/ubench/surface_k.hip).  Every lane therefore
// iterates at most AO_CUT passes in place; estimates still iterating then are set aside in an LDS queue (their state: w, pos,
// dr, r) and the workgroup finishes them together, 64 per wave, with the same code (mb8_iterate resumes anywhere: bit-identical).
// A full queue is not an error: those lanes finish in place.
#ifdef RMDF_AB_AO_CUT               // A/B switch: passes an AO estimate runs in place before it is queued (tools/emulated_schedule.py sweeps it)
#define AO_CUT RMDF_AB_AO_CUT
#else
#define AO_CUT 6
#endif
#define AO_CAP 256

// OUT selects the planes an instantiation writes: OUT_RGBA8 = the product path (RGBA8 frame only), OUT_MIRROR = RGBA8 +
// the same rows into a registered host buffer (rmdf_register_host_buffer), OUT_PLANES = RGBA8 + the float / steps /
// iteration planes of rmdf_render_tile_ex (parity tests, cost probe).
enum { OUT_RGBA8 = 0, OUT_MIRROR = 1, OUT_PLANES = 2 };
#define WPB 4                       // waves per workgroup: a 32x8 strip
#ifdef RMDF_AB_NO_XL                // A/B switch (tools/abtest): the Cornell march without the eight-lanes-per-ray tail
#define XL_TAIL false
#else
#define XL_TAIL true
#endif
#ifndef RMDF_AB_XL_G                // lanes per ray of that tail: 8 (a wave takes it up at <= 8 live rays); A/B: 4 (at <= 16; not yet run on hardware)
#define RMDF_AB_XL_G 8
#endif

// Everything a lane derives from (strip, thread id): the rectangle of its launch / shard slot, its pixel, its primary ray.
// (Deriving it a second time after the march from laundered inputs, so that none of it occupies registers across the march
// loop, was measured in round 3: 1.5 % slower -- the kernel stays under the 64 VGPRs of 8 waves per SIMD (profiles/*_kernel_resources.txt), nothing is short.)
struct PixelGeom {
    int rx0, ry0, rx1, ry1, pitch, ox, oy;
    size_t obase;
    int ex0, ey0, bx, by;
    int lane, wave, lx, ly;
    bool active;
    v3 dir;
};
__device__ __forceinline__ v3 primary_dir(const FrameParams &p, int px, int py)
{
    // generate_ray, perspective branch (fragment.shd:840-871)
    // frame sizes are 1 .. 32768 per side, times at most 8 rays (rmdf_api.cpp), so pixel centres, sizes and the aspect ratio are all inside div_known_range's
    // range; ndcy * fov_xs is 0 or at least 2^-24 in magnitude
    const float ndcx = RMDF_FAST_DIV((float)px + 0.5f, p.wf) * 2.0f - 1.0f;
    const float ndcy = RMDF_FAST_DIV((float)py + 0.5f, p.hf) * 2.0f - 1.0f;
    const v3 dcam = normalize3(mk3(ndcx * p.fov_xs, RMDF_FAST_DIV(ndcy * p.fov_xs, p.aspect), -1.0f));
    return mk3(p.cam[0] * dcam.x + p.cam[3] * dcam.y + p.cam[6] * dcam.z,
               p.cam[1] * dcam.x + p.cam[4] * dcam.y + p.cam[7] * dcam.z,
               p.cam[2] * dcam.x + p.cam[5] * dcam.y + p.cam[8] * dcam.z);
}
__device__ __forceinline__ PixelGeom pixel_geom(const FrameParams &p, unsigned lin, unsigned tid)
{
    PixelGeom g;
    const unsigned strips_per_slot = gridDim.x * gridDim.y;
    const unsigned strip = lin % strips_per_slot;
    // rectangle of this launch / shard slot
    if (p.n_shard_tiles > 0) {
        const int slot = (int)(lin / strips_per_slot);
        tile_rect((int)p.shard_tile[slot], p.w, p.h, g.rx0, g.ry0, g.rx1, g.ry1);
        g.pitch = g.rx1 - g.rx0; g.ox = g.rx0; g.oy = g.ry0;
        g.obase = (size_t)slot * (size_t)(p.w / 8) * (size_t)(p.h / 8);
    } else {
        g.rx0 = p.x0; g.ry0 = p.y0; g.rx1 = p.x1; g.ry1 = p.y1;
        g.pitch = p.w; g.ox = 0; g.oy = 0; g.obase = 0;
    }
    // GL quads are aligned to even window coordinates: helper pixels outside the
    // rectangle are computed (not written) so that derivatives match a full-frame render
    g.ex0 = g.rx0 & ~1; g.ey0 = g.ry0 & ~1;
    const int ex1 = (g.rx1 + 1) & ~1, ey1 = (g.ry1 + 1) & ~1;
    g.lane = (int)(tid & 63u); g.wave = (int)(tid >> 6);
    g.lx = (g.lane & 1) | (((g.lane >> 2) & 3) << 1);
    g.ly = ((g.lane >> 1) & 1) | (((g.lane >> 4) & 3) << 1);
    g.bx = (int)(strip % gridDim.x); g.by = (int)(strip / gridDim.x);
    const int px = g.ex0 + g.bx * (WPB * 8) + g.wave * 8 + g.lx;
    const int py = g.ey0 + g.by * 8 + g.ly;
    g.active = (px < ex1) && (py < ey1);
    g.dir = primary_dir(p, px, py);
    return g;
}

template <int SCENE, bool MERGE, int OUT>
__device__ __forceinline__ void render_body(const FrameParams &p)
{
    constexpr bool AO_POOL = (SCENE == 2) && MERGE;
    // Cornell box: a copy of the triangle rows lives in LDS for the whole launch (every lane reads the rows of ITS candidates:
    // de_cornell_box_lanes); the candidate grid itself stays in global memory (cgrid, 1 MB, L2-resident)
    __shared__ float  s_ctab[SCENE == 0 ? 32 * CORNELL_STRIDE : 1];
    const unsigned *cgrid = nullptr;
    if (SCENE == 0 && p.cornell_prune && p.cornell_grid) {
        for (int i = threadIdx.x; i < 32 * CORNELL_STRIDE; i += WPB * 64) s_ctab[i] = p.cornell_tab[i];
        __syncthreads();
        cgrid = p.cornell_grid;
    }
    __shared__ unsigned s_ao_cnt;
    __shared__ float4   s_ao_q[AO_POOL ? AO_CAP : 1][2];      // w.xyz, dr | pos.xyz, r
    __shared__ float2   s_ao_out[AO_POOL ? AO_CAP : 1];       // distance, iterations run after the hand-over
    // Which strip this workgroup renders: raster order over (slot, row, column), or most expensive first
    // (block_order, a permutation of the launch's linear workgroup ids -- it spans all tiles of a shard launch)
    unsigned lin = (blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    if (p.block_order) lin = p.block_order[lin];
    const PixelGeom g = pixel_geom(p, lin, threadIdx.x);
#ifdef RMDF_XCHECK
    const unsigned long long dbg_t0 = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;   // wave timeline (tools/nested_timeline.py)
    const unsigned long long dbg_c0 = p.dbg ? __builtin_amdgcn_s_memtime() : 0ull;       // shader cycles: the clock the kernel really runs at
#endif
    const v3 origin = mk3(p.cam[9], p.cam[10], p.cam[11]);
#ifdef RMDF_XCHECK
    unsigned dbg_t_march = 0u;          // when this wave's march ended (100 MHz ticks since its start): tools/nested_timeline.py
#endif

    // ray_march (fragment.shd:618-676)
    bool hit = false;
    int steps = 0;
    unsigned iters = 0;
    int tri_hint = 0;               // Cornell: evaluation-order hint of the distance estimate (never a result)
    float t = 0.0f;
    float tmin, tmax;
    if (!MERGE && SCENE == 0 && XL_TAIL) {
        // Cornell box, pruned estimate: the wave marches lane by lane while more than eight of its rays are live, then moves every
        // remaining ray into a group of eight lanes (rmdf_device.hpp: de_cornell_box_group8).  The march is the one below, statement
        // for statement; a ray's (t, steps, hit) do not depend on where it was marched.
        __shared__ float2 s_xl[WPB * 64];                     // results of rays finished in a group, by home lane: t, steps | hit << 15
        bool act = g.active && ray_sphere(origin, g.dir, bsphere_r<SCENE>(), tmin, tmax) && p.max_steps > 0;
        t = act ? gmax(0.0f, tmin) : 0.0f;
        if (!act) tmax = 0.0f;
        const int lane = g.lane, wave = __builtin_amdgcn_readfirstlane(g.wave);
        bool moved = false;
        for (;;) {
            const unsigned long long am = __ballot(act);
            if (am == 0ull) break;
            if (cgrid && __popcll(am) <= 64 / RMDF_AB_XL_G) {
                // group gi = the gi-th live lane's ray: every lane of the group takes a copy of its state
                const int gi = lane / RMDF_AB_XL_G, sub = lane & (RMDF_AB_XL_G - 1);
                unsigned long long mm = am;
                for (int i = 0; i < gi; i++) mm &= mm - 1ull;
                const bool gact0 = mm != 0ull;
                const int home = gact0 ? (int)__builtin_ctzll(mm) : lane;
                moved = act;
                float xt = __shfl(t, home, 64), xtmax = __shfl(tmax, home, 64);
                int xs = __shfl(steps, home, 64);
                const float xdx = __shfl(g.dir.x, home, 64), xdy = __shfl(g.dir.y, home, 64), xdz = __shfl(g.dir.z, home, 64);
                bool gact = gact0;
                unsigned prev = 0u;
                while (__ballot(gact) != 0ull) {
                    const v3 pos = mk3(origin.x + xt * xdx, origin.y + xt * xdy, origin.z + xt * xdz);
                    const float dist = de_cornell_box_group8<RMDF_AB_XL_G>(gact, pos, s_ctab, cgrid, sub, prev);
                    if (gact) {
                        xt += dist;
                        const bool out = xt > xtmax;
                        const bool h2 = !out && (dist < shk::march_min_dist);
                        bool done = out || h2;
                        if (!done) { xs++; done = xs >= p.max_steps; }
                        if (done) {
                            if (sub == 0) s_xl[wave * 64 + home] = make_float2(xt, __int_as_float(xs | (h2 ? 0x8000 : 0)));
                            gact = false;
                        }
                    }
                }
                break;
            }
            if (act) {
                const v3 pos = mk3(origin.x + t * g.dir.x, origin.y + t * g.dir.y, origin.z + t * g.dir.z);
                const float dist = distance_estimator<SCENE>(pos, p, iters, tri_hint, cgrid, s_ctab);
                t += dist;
                if (t > tmax) act = false;
                else if (dist < shk::march_min_dist) { hit = true; act = false; }
                else { steps++; if (steps >= p.max_steps) act = false; }
            }
        }
        if (moved) {
            const float2 r2 = s_xl[wave * 64 + lane];       // written by this wave (LDS operations of one wave complete in order)
            t = r2.x;
            const int sb = __float_as_int(r2.y);
            steps = sb & 0x7fff;
            hit = (sb >> 15) != 0;
        }
#ifdef RMDF_XCHECK
        dbg_t_march = p.dbg ? (unsigned)((__builtin_amdgcn_s_memrealtime() - dbg_t0)) : 0u;
#endif
    } else if (!MERGE) {
        if (g.active && ray_sphere(origin, g.dir, bsphere_r<SCENE>(), tmin, tmax)) {
            t = gmax(0.0f, tmin);
            for (steps = 0; steps < p.max_steps; steps++) {
                v3 pos = mk3(origin.x + t * g.dir.x, origin.y + t * g.dir.y, origin.z + t * g.dir.z);
                float dist = distance_estimator<SCENE>(pos, p, iters, tri_hint, cgrid, s_ctab);
                t += dist;
                if (t > tmax) break;
                if (dist < shk::march_min_dist) { hit = true; break; }
            }
        }
#ifdef RMDF_XCHECK
        dbg_t_march = p.dbg ? (unsigned)((__builtin_amdgcn_s_memrealtime() - dbg_t0)) : 0u;
#endif
    } else {
        __shared__ int    s_host, s_nreported;
        __shared__ int    s_mb_n[WPB], s_mb_ready[WPB];
        __shared__ float4 s_mb[WPB][MERGE_T];
        __shared__ float4 s_res[WPB * 64];          // per strip pixel: t, steps | hit << 15, iterations
        if (threadIdx.x == 0) { s_host = -1; s_nreported = 0; s_ao_cnt = 0u; }
        if (threadIdx.x < WPB) { s_mb_ready[threadIdx.x] = 0; s_mb_n[threadIdx.x] = 0; }
        {
            const int my_sp = g.ly * (WPB * 8) + g.wave * 8 + g.lx;
            s_res[my_sp] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        }
        __syncthreads();

        // the ray this lane is marching right now (its own, or an adopted one when this wave is the host)
        bool act = false;
        int cur_sp = g.ly * (WPB * 8) + g.wave * 8 + g.lx, st = 0;
        unsigned it = 0;
        float dx = g.dir.x, dy = g.dir.y, dz = g.dir.z, tt = 0.0f, tmx = 0.0f;
        const int lane = g.lane, wave = __builtin_amdgcn_readfirstlane(g.wave);
        const int sx0 = g.ex0 + g.bx * (WPB * 8), sy0 = g.ey0 + g.by * 8;        // strip origin (wave-uniform)
        if (g.active && ray_sphere(origin, g.dir, bsphere_r<SCENE>(), tmin, tmax) && p.max_steps > 0) {
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
                        const v3 dq = primary_dir(p, sx0 + (cur_sp % (WPB * 8)), sy0 + (cur_sp / (WPB * 8)));
                        dx = dq.x; dy = dq.y; dz = dq.z;
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
                const float dist = distance_estimator<SCENE>(pos, p, it, tri_hint, cgrid, s_ctab);
                tt += dist;
                const bool out = tt > tmx;
                const bool h2 = !out && (dist < shk::march_min_dist);
                bool done = out || h2;
                if (!done) { st++; done = st >= p.max_steps; }
                if (done) {
                    s_res[cur_sp] = make_float4(tt, __int_as_float(st | (h2 ? 0x8000 : 0)), __uint_as_float(it), 0.0f);
                    act = false;
                }
            }
        }
        __syncthreads();
        const float4 rr = s_res[g.ly * (WPB * 8) + g.wave * 8 + g.lx];
        t = rr.x;
        const int sb = __float_as_int(rr.y);
        steps = sb & 0x7fff;
        hit = (sb >> 15) != 0;
        iters = __float_as_uint(rr.z);
    }

#ifdef RMDF_AB_TAIL_PRIO
    __builtin_amdgcn_s_setprio(RMDF_AB_TAIL_PRIO);           // everything after the march: finish the workgroup and free its slots
#endif
    // render_ray hit branch up to the texture lookups (fragment.shd:743-799)
    const v3 dir = g.dir;
    const int lane = g.lane, wave = g.wave, lx = g.lx, ly = g.ly;
    v3 n = mk3(0.0f, 0.0f, 0.0f), refl = mk3(0.0f, 0.0f, 0.0f);
    float ao = 0.0f, fresnel = 0.0f;
    v3 isec = mk3(0.0f, 0.0f, 0.0f);
    if (SCENE == 0 && hit) {
        // Cornell box: the normal's four estimates and the four AO taps as ONE loop around one copy of the estimate (inlined nine
        // times the kernel was 130 KB of code, twice the instruction cache)
        isec = mk3(origin.x + dir.x * t, origin.y + dir.y * t, origin.z + dir.z * t);
        const v3 np = mk3(isec.x - dir.x * shk::isec_step_back, isec.y - dir.y * shk::isec_step_back, isec.z - dir.z * shk::isec_step_back);
        const float eps = shk::normal_eps;
        float d0 = 0.0f, ddx = 0.0f, ddy = 0.0f, ddz = 0.0f, occl = 0.0f;
#ifdef RMDF_AB_SHARED_BOUNDS
        unsigned kept_tris = 0u;
#endif
#pragma unroll 1
        for (int k = 0; k < 8; k++) {
            if (k == 4) n = normalize3(mk3(ddx, ddy, ddz));
            const float dlk = k == 4 ? shk::cornell_ao_d0 : (k == 5 ? shk::cornell_ao_d1 : (k == 6 ? shk::cornell_ao_d2 : shk::cornell_ao_d3));
            const float wtk = k == 4 ? shk::cornell_ao_w0 : (k == 5 ? shk::cornell_ao_w1 : (k == 6 ? shk::cornell_ao_w2 : shk::cornell_ao_w3));
            const v3 pos = k < 4 ? mk3(np.x - (k == 1 ? eps : 0.0f), np.y - (k == 2 ? eps : 0.0f), np.z - (k == 3 ? eps : 0.0f))
                                 : mk3(isec.x + n.x * dlk, isec.y + n.y * dlk, isec.z + n.z * dlk);
#ifdef RMDF_AB_SHARED_BOUNDS
            // (A/B only, not yet run on hardware: the normal's four sample points lie within 1e-5 of each other -- one pass of bound tests,
            // its margin widened accordingly, serves all four: rmdf_device.hpp de_cornell_box_lanes `keep`)
            // (one inlined copy of each form: the AO samples take the wider margin too -- a superset of survivors, the same minimum)
            float d;
            if (cgrid) {
                unsigned kept_now = 0u;
                if (k == 0 || k >= 4) d = de_cornell_box_lanes(pos, s_ctab, cgrid, tri_hint, &kept_now);
                else                  d = de_cornell_box_lanes_kept(pos, s_ctab, kept_tris);
                if (k == 0) kept_tris = kept_now;
            } else
                d = distance_estimator<SCENE>(pos, p, iters, tri_hint, cgrid, s_ctab);
#else
            const float d = distance_estimator<SCENE>(pos, p, iters, tri_hint, cgrid, s_ctab);
#endif
            if (k == 0) d0 = d;
            else if (k == 1) ddx = d0 - d;
            else if (k == 2) ddy = d0 - d;
            else if (k == 3) ddz = d0 - d;
            else {
                const float ylk = k == 4 ? 1.0f / shk::cornell_ao_d0 : (k == 5 ? 1.0f / shk::cornell_ao_d1 : (k == 6 ? 1.0f / shk::cornell_ao_d2 : 1.0f / shk::cornell_ao_d3));
                occl += wtk * ao_term<RMDF_SHADE_FAST>(d, dlk, ylk);
            }
        }
        ao = 1.0f - occl;
    }
    // distance_ao (fragment.shd:542-591)
    const float w0 = shk::ao_w0, e0 = shk::ao_d0, w1 = shk::ao_w1, e1 = shk::ao_d1;
    float ao_dist[2] = { 0.0f, 0.0f };
    if ((SCENE == 1 || SCENE == 3) && hit) {
        // the test scene and the general-power Mandelbulb: the normal's four estimates and the two AO taps as ONE loop around one copy of
        // the estimate, like the Cornell box above (inlined seven times the general-power kernel was 12 000 lines of assembly, most of it
        // transcendental functions: round 5).  x - 0.0f == x for every x, so the sample points are the written ones bit for bit.
        isec = mk3(origin.x + dir.x * t, origin.y + dir.y * t, origin.z + dir.z * t);
        const v3 np = mk3(isec.x - dir.x * shk::isec_step_back, isec.y - dir.y * shk::isec_step_back, isec.z - dir.z * shk::isec_step_back);
        const float eps = shk::normal_eps;
        float d0 = 0.0f, ddx = 0.0f, ddy = 0.0f, ddz = 0.0f;
#pragma unroll 1
        for (int k = 0; k < 6; k++) {
            if (k == 4) n = normalize3(mk3(ddx, ddy, ddz));
            const float e = k == 4 ? e0 : e1;
            const v3 pos = k < 4 ? mk3(np.x - (k == 1 ? eps : 0.0f), np.y - (k == 2 ? eps : 0.0f), np.z - (k == 3 ? eps : 0.0f))
                                 : mk3(isec.x + n.x * e, isec.y + n.y * e, isec.z + n.z * e);
            const float d = distance_estimator<SCENE>(pos, p, iters, tri_hint, cgrid);
            if (k == 0) d0 = d;
            else if (k == 1) ddx = d0 - d;
            else if (k == 2) ddy = d0 - d;
            else if (k == 3) ddz = d0 - d;
            else if (k == 4) ao_dist[0] = d;
            else ao_dist[1] = d;
        }
    }
    if (SCENE == 2 && hit) {
        isec = mk3(origin.x + dir.x * t, origin.y + dir.y * t, origin.z + dir.z * t);
        v3 np = mk3(isec.x - dir.x * shk::isec_step_back, isec.y - dir.y * shk::isec_step_back, isec.z - dir.z * shk::isec_step_back);
        const float eps = shk::normal_eps;
        float d0 = distance_estimator<SCENE>(np, p, iters, tri_hint, cgrid);
        float dx = distance_estimator<SCENE>(mk3(np.x - eps, np.y - 0.0f, np.z - 0.0f), p, iters, tri_hint, cgrid);
        float dy = distance_estimator<SCENE>(mk3(np.x - 0.0f, np.y - eps, np.z - 0.0f), p, iters, tri_hint, cgrid);
        float dz = distance_estimator<SCENE>(mk3(np.x - 0.0f, np.y - 0.0f, np.z - eps), p, iters, tri_hint, cgrid);
        n = normalize3(mk3(d0 - dx, d0 - dy, d0 - dz));
    }
    if (AO_POOL) {
        // the two estimates with their stragglers set aside (see AO_CUT above); every wave of the workgroup comes through here
        int slot[2] = { -1, -1 };
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const float e = k == 0 ? e0 : e1;
            v3 pos = mk3(isec.x + n.x * e, isec.y + n.y * e, isec.z + n.z * e);
            pos = mk3(pos.z, pos.x, pos.y);                     // de_mandelbulb8's own first statement
            v3 w = pos;
            float dr = 1.0f, r = 0.0f, d = 0.0f;
            bool pend = false;
            if (hit) {
                float m = 1.0f;
                unsigned n = 0u;
                mb8_iterate_t<true>(w, pos, dr, r, d, 0, AO_CUT, n, m);
                const bool redo = mb8_fold_failed(m, fold_min_of(p));   // rmdf_device.hpp: the folded passes need the written ones' range
                if (__builtin_expect(RMDF_LANES_HERE(redo) != 0ull, 0)) {
                    if (redo) { w = pos; dr = 1.0f; r = 0.0f; d = 0.0f; n = 0u; mb8_iterate_t<false>(w, pos, dr, r, d, 0, AO_CUT, n, m); }
                }
                iters += n;
                pend = !(d > RMDF_MB8_D4);
            }
            // one LDS atomic per wave: queue slots for its pending estimates
            const unsigned long long pm = __ballot(pend);
            if (pm != 0ull) {
                unsigned base = 0u;
                if (lane == 0) base = atomicAdd(&s_ao_cnt, (unsigned)__popcll(pm));
                base = __builtin_amdgcn_readfirstlane(base);
                const unsigned mine = base + (unsigned)__builtin_amdgcn_mbcnt_hi((unsigned)(pm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)pm, 0));
                if (pend) {
                    if (mine < (unsigned)AO_CAP) {
                        slot[k] = (int)mine;
                        s_ao_q[mine][0] = make_float4(w.x, w.y, w.z, dr);
                        s_ao_q[mine][1] = make_float4(pos.x, pos.y, pos.z, r);
                    } else {
                        float m = 1.0f;
                        mb8_iterate_t<false>(w, pos, dr, r, d, AO_CUT, shk::mb_iterations_i, iters, m);      // queue full: finish in place (written form)
                    }
                }
            }
            if (hit && slot[k] < 0) ao_dist[k] = mb8_finish(dr, r, d);
        }
        __syncthreads();
        const unsigned n_q = s_ao_cnt < (unsigned)AO_CAP ? s_ao_cnt : (unsigned)AO_CAP;     // the same for every wave
        if (n_q != 0u) {
            for (unsigned tsk = threadIdx.x; tsk < n_q; tsk += WPB * 64) {
                const float4 a = s_ao_q[tsk][0], b = s_ao_q[tsk][1];
                v3 w = mk3(a.x, a.y, a.z);
                const v3 pos = mk3(b.x, b.y, b.z);
                float dr = a.w, r = b.w, d = 0.0f;
                unsigned it2 = 0u;
                float m = 1.0f;
                mb8_iterate_t<true>(w, pos, dr, r, d, AO_CUT, shk::mb_iterations_i, it2, m);
                const bool redo = mb8_fold_failed(m, fold_min_of(p));
                if (__builtin_expect(RMDF_LANES_HERE(redo) != 0ull, 0)) {
                    if (redo) { w = mk3(a.x, a.y, a.z); dr = a.w; r = b.w; d = 0.0f; it2 = 0u; mb8_iterate_t<false>(w, pos, dr, r, d, AO_CUT, shk::mb_iterations_i, it2, m); }
                }
                s_ao_out[tsk] = make_float2(mb8_finish(dr, r, d), __uint_as_float(it2));
            }
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 2; k++)
                if (slot[k] >= 0) { const float2 o = s_ao_out[slot[k]]; ao_dist[k] = o.x; iters += __float_as_uint(o.y); }
        }
    }
    if (hit) {
        float occl = 0.0f;
        if (SCENE != 0) {
            if (!AO_POOL && SCENE == 2) {
                ao_dist[0] = distance_estimator<SCENE>(mk3(isec.x + n.x * e0, isec.y + n.y * e0, isec.z + n.z * e0), p, iters, tri_hint, cgrid);
                ao_dist[1] = distance_estimator<SCENE>(mk3(isec.x + n.x * e1, isec.y + n.y * e1, isec.z + n.z * e1), p, iters, tri_hint, cgrid);
            }
            occl += w0 * ao_term<RMDF_SHADE_FAST>(ao_dist[0], e0, 1.0f / e0);      // rmdf_device.hpp: the quotient by a constant offset
            occl += w1 * ao_term<RMDF_SHADE_FAST>(ao_dist[1], e1, 1.0f / e1);
            occl = 1.0f - occl;
            occl -= shk::ao_bias;
            occl *= shk::ao_gain;
            occl *= occl;
            ao = gclamp(occl, 0.0f, 1.0f);
        }   // Cornell box: the four taps (fragment.shd:568-589) ran above, in the loop with the normal
        fresnel = fresnel_conductor(dot3(mk3(-dir.x, -dir.y, -dir.z), n), shk::fresnel_eta, shk::fresnel_k);
        refl = reflect3(dir, n);
    }

    // quad neighbours (lane^1 = horizontal, lane^2 = vertical)
    const int hit_i = hit ? 1 : 0;
    const bool hit_h = shfl_xor_w(hit_i, 1) != 0, hit_v = shfl_xor_w(hit_i, 2) != 0;
    const v3 n_h = shfl_xor3(n, 1), n_v = shfl_xor3(n, 2);
    const v3 refl_h = shfl_xor3(refl, 1), refl_v = shfl_xor3(refl, 2);
    const v3 dir_h = shfl_xor3(dir, 1), dir_v = shfl_xor3(dir, 2);

    v3 color;
    if (hit) {
        // fragment.shd:799-810
        v3 t1 = cube_texture(p.env_cos1, n, hit_h, n_h, hit_v, n_v);
        v3 t8 = cube_texture(p.env_cos8, refl, hit_h, refl_h, hit_v, refl_v);
        v3 tr = cube_texture(p.env_refl, refl, hit_h, refl_h, hit_v, refl_v);
        const float diff_weight = shk::diff_weight, spec_weight = shk::spec_weight_one_minus - shk::diff_weight;
        const float npl = (shk::phong_lobe_n + shk::phong_lobe_plus) / shk::phong_lobe_div;
        color.x = (t1.x * shk::diff_r * diff_weight + t8.x * shk::spec_r * npl * fresnel * spec_weight + tr.x * spec_weight * fresnel * shk::refl_weight) * shk::exposure * ao;
        color.y = (t1.y * shk::diff_g * diff_weight + t8.y * shk::spec_g * npl * fresnel * spec_weight + tr.y * spec_weight * fresnel * shk::refl_weight) * shk::exposure * ao;
        color.z = (t1.z * shk::diff_b * diff_weight + t8.z * shk::spec_b * npl * fresnel * spec_weight + tr.z * spec_weight * fresnel * shk::refl_weight) * shk::exposure * ao;
    } else {
        // fragment.shd:823
        color = cube_texture(p.env_refl, dir, !hit_h, dir_h, !hit_v, dir_v);   // neighbours in the hit branch: undefined derivative -> minified
    }

    // fragment.shd:959-960 and the RGBA8 conversion of the colour attachment
    const float inv_gamma = 1.0f / shk::gamma;
    const float gr = pow_pinned(color.x, inv_gamma), gg = pow_pinned(color.y, inv_gamma), gb = pow_pinned(color.z, inv_gamma);
    // Stage the 32x8 strip in LDS so that every store instruction writes whole 128-byte lines (a wave's own
    // 8x8 packet would write 32-byte pieces of 8 different rows).  Thread t stores pixel (t % 32, t / 32).
    __shared__ __attribute__((aligned(16))) uint32_t s_rgba8[8][WPB * 8];
    __shared__ float4   s_f32[OUT == OUT_PLANES ? 8 : 1][WPB * 8];
    __shared__ uint32_t s_meta[OUT == OUT_PLANES ? 8 : 1][WPB * 8];
    {
        const int sx = wave * 8 + lx;
        s_rgba8[ly][sx] = to_unorm8(gr) | (to_unorm8(gg) << 8) | (to_unorm8(gb) << 16) | 0xff000000u;
        if (OUT == OUT_PLANES) {
            s_f32[ly][sx] = make_float4(gr, gg, gb, 1.0f);
            s_meta[ly][sx] = (uint32_t)(steps | (hit_i << 15)) | ((iters > 65535u ? 65535u : iters) << 16);
        }
    }
    __syncthreads();
    {
        const int ox_ = threadIdx.x % (WPB * 8), oy_ = threadIdx.x / (WPB * 8);
        const int qx = g.ex0 + g.bx * (WPB * 8) + ox_, qy = g.ey0 + g.by * 8 + oy_;
        if (qx >= g.rx0 && qx < g.rx1 && qy >= g.ry0 && qy < g.ry1) {
            const size_t idx = g.obase + (size_t)(qx - g.ox) + (size_t)(qy - g.oy) * (size_t)g.pitch;
            if (OUT != OUT_PLANES || p.rgba8) p.rgba8[idx] = s_rgba8[oy_][ox_];
#ifndef RMDF_AB_MIRROR16
            if (OUT == OUT_MIRROR) p.rgba8_mirror[idx] = s_rgba8[oy_][ox_];
#endif
            if (OUT == OUT_PLANES) {
                if (p.rgba_f32) p.rgba_f32[idx] = s_f32[oy_][ox_];
                const uint32_t m = s_meta[oy_][ox_];
                if (p.steps) p.steps[idx] = (uint16_t)(m & 0xffffu);
                if (p.iters) p.iters[idx] = (uint16_t)(m >> 16);
            }
        }
    }
#ifdef RMDF_AB_MIRROR16
    // (A/B build only -- written after GPU access closed in round 5, NOT yet run on hardware: NOTEBOOK.md A.5 "whole-frame host calls")
    if (OUT == OUT_MIRROR && threadIdx.x < 64) {
        // The same rows into host memory over PCIe (a registered caller buffer, the page-locked shadow frame, a tile job's host tile): ONE
        // wave, sixteen bytes per lane -- the strip's eight 128-byte rows leave in one store instruction instead of four (round 5: the
        // mirror variant of the headline kernel took 0.45 ms against 0.38 without the second store).  Groups of four pixels that are
        // not whole, or not 16-byte aligned in the destination (a caller's buffer may sit anywhere), go out pixel by pixel.
        const int row = (int)threadIdx.x >> 3, c4 = ((int)threadIdx.x & 7) * 4;
        const int qx = g.ex0 + g.bx * (WPB * 8) + c4, qy = g.ey0 + g.by * 8 + row;
        if (qy >= g.ry0 && qy < g.ry1 && qx + 3 >= g.rx0 && qx < g.rx1) {
            // (qx - g.ox can be -1 .. -3 when the group straddles the rectangle's left edge: signed arithmetic; those elements are not stored)
            const ptrdiff_t idx = (ptrdiff_t)g.obase + (ptrdiff_t)(qx - g.ox) + (ptrdiff_t)(qy - g.oy) * (ptrdiff_t)g.pitch;
            uint32_t *dst = p.rgba8_mirror + idx;
            if (qx >= g.rx0 && qx + 3 < g.rx1 && (((uintptr_t)dst) & 15u) == 0u) {
                *reinterpret_cast<uint4 *>(dst) = *reinterpret_cast<const uint4 *>(&s_rgba8[row][c4]);
            } else {
                for (int k = 0; k < 4; k++)
                    if (qx + k >= g.rx0 && qx + k < g.rx1) dst[k] = s_rgba8[row][c4 + k];
            }
        }
    }
#endif
#ifdef RMDF_XCHECK
    // (librmdf_xcheck.so only: the one-launch band hand-over has never run on hardware -- the product's OUT_MIRROR kernels stay the ones the GPU tier ran)
    if (OUT == OUT_MIRROR && p.band_flag) {
        // every wave waits until ITS stores have landed in host memory; behind the barrier lane 0 speaks for the workgroup
        __threadfence_system();
        __syncthreads();
        if (threadIdx.x == 0) {
            const int band = g.by / p.band_strip_rows;
            const int rows = (int)gridDim.y - band * p.band_strip_rows;
            const unsigned n_band = gridDim.x * (unsigned)(rows < p.band_strip_rows ? rows : p.band_strip_rows);
            const unsigned before = __hip_atomic_fetch_add(&p.band_count[band], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
            if (before == n_band - 1u) {                             // every other workgroup of the band has passed its own fence
                p.band_count[band] = 0u;                             // ready for the next frame
                __hip_atomic_store((unsigned *)&p.band_flag[band], p.band_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
    }
#endif
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
#ifdef RMDF_XCHECK
    if (p.dbg && lane == 0) {
        const unsigned wid = (blockIdx.y * gridDim.x + blockIdx.x) * 4 + wave;
        if (wid < 32768u * 2u) { p.dbg[wid * 8 + 4] = dbg_c0; p.dbg[wid * 8 + 5] = __builtin_amdgcn_s_memtime(); p.dbg[wid * 8 + 6] = dbg_t0; p.dbg[wid * 8 + 7] = __builtin_amdgcn_s_memrealtime(); p.dbg[wid * 8] = (unsigned long long)steps; p.dbg[wid * 8 + 2] = dbg_t_march; }
    }
#endif
}
// The power-8 Mandelbulb's product variants are held to 64 VGPRs = 8 waves per SIMD (the kernel is latency-sensitive: the eighth
// wave is worth 3 %).  The same bound caps the SGPRs at 80 (8 x (80 + the 16 of the trap handler) <= the 800 of a SIMD -- stating
// 96 through amdgpu_num_sgpr instead was measured: the hardware then runs 7 waves); what does not fit lives in lanes of a spare
// VGPR (v_writelane, prologue and epilogue only).  No scratch: `make resources`.
// The Cornell box's product variants are bound to 6 waves per SIMD (80 VGPRs; 7 and 8 spill and were measured slower).
template <int SCENE, bool MERGE, int OUT>
__global__ __launch_bounds__(WPB * 64, (SCENE == 2 && OUT != OUT_PLANES) ? 8 : ((SCENE == 0 && OUT != OUT_PLANES) ? 6 : 1)) void k_render(const FrameParams p) { render_body<SCENE, MERGE, OUT>(p); }

// One source, one object -- or five (csrc/Makefile, `make SPLIT=1`: the build the GPU-memory-fault hunt of rounds 4 and 5 needs, NOTEBOOK.md A.5).
// Compiled with -DRMDF_RENDER_SCENE=N this file yields the kernels of FragmentShader N and their launcher launch_render_scene_N, with
// -DRMDF_RENDER_SPLIT the host side they share (grid, strip order, dispatch); with neither, everything.  Same kernels byte for byte.
#define RMDF_CAT2(a, b) a##b
#define RMDF_CAT(a, b) RMDF_CAT2(a, b)
#ifndef RMDF_RENDER_SPLIT
template <int SCENE>
static void launch_render_scene(const FrameParams &p, dim3 grid, hipStream_t stream)
{
    // The Cornell box never pools its last rays (rmdf_api.cpp fill_params: measured 6 % slower), so its MERGE kernels are not built: they
    // would be the only kernels of the library with scratch (8 bytes, two spilled VGPRs).  -DRMDF_AB_CORNELL_MERGE=<threshold> builds bring them back.
#ifdef RMDF_AB_CORNELL_MERGE
    constexpr bool has_merge = true;
#else
    constexpr bool has_merge = SCENE != 0;
#endif
    const bool merge = has_merge && p.merge_stragglers != 0;
    const int out = (p.rgba_f32 || p.steps || p.iters) ? OUT_PLANES : (p.rgba8_mirror ? OUT_MIRROR : OUT_RGBA8);
#define RMDF_LAUNCH(O)                                                                                                      \
    do {                                                                                                                    \
        if (merge) hipLaunchKernelGGL((k_render<SCENE, has_merge, O>), grid, dim3(WPB * 64), 0, stream, p);                  \
        else       hipLaunchKernelGGL((k_render<SCENE, false, O>), grid, dim3(WPB * 64), 0, stream, p);                      \
    } while (0)
    if (out == OUT_PLANES)      RMDF_LAUNCH(OUT_PLANES);
    else if (out == OUT_MIRROR) RMDF_LAUNCH(OUT_MIRROR);
    else                        RMDF_LAUNCH(OUT_RGBA8);
#undef RMDF_LAUNCH
}
#endif
#ifdef RMDF_RENDER_SCENE
void RMDF_CAT(launch_render_scene_, RMDF_RENDER_SCENE)(const FrameParams &p, dim3 grid, hipStream_t stream) { launch_render_scene<RMDF_RENDER_SCENE>(p, grid, stream); }
#else
#ifdef RMDF_RENDER_SPLIT
void launch_render_scene_0(const FrameParams &p, dim3 grid, hipStream_t stream);
void launch_render_scene_1(const FrameParams &p, dim3 grid, hipStream_t stream);
void launch_render_scene_2(const FrameParams &p, dim3 grid, hipStream_t stream);
void launch_render_scene_3(const FrameParams &p, dim3 grid, hipStream_t stream);
#else
static void launch_render_scene_0(const FrameParams &p, dim3 grid, hipStream_t stream) { launch_render_scene<0>(p, grid, stream); }
static void launch_render_scene_1(const FrameParams &p, dim3 grid, hipStream_t stream) { launch_render_scene<1>(p, grid, stream); }
static void launch_render_scene_2(const FrameParams &p, dim3 grid, hipStream_t stream) { launch_render_scene<2>(p, grid, stream); }
static void launch_render_scene_3(const FrameParams &p, dim3 grid, hipStream_t stream) { launch_render_scene<3>(p, grid, stream); }
#endif
static void render_grid(const FrameParams &p, dim3 &grid)
{
    int rx0, ry0, rx1, ry1, nz = 1;
    if (p.n_shard_tiles > 0) {
        // all tiles have the same size when 8 | w and 8 | h (required in shard mode); odd-sized tiles get one helper
        // column / row, on one side only
        rx0 = 0; ry0 = 0; rx1 = p.w / 8 + ((p.w / 8) & 1); ry1 = p.h / 8 + ((p.h / 8) & 1);
        nz = p.n_shard_tiles;
    } else {
        rx0 = p.x0; ry0 = p.y0; rx1 = p.x1; ry1 = p.y1;
    }
    const int ex0 = rx0 & ~1, ey0 = ry0 & ~1, ex1 = (rx1 + 1) & ~1, ey1 = (ry1 + 1) & ~1;
    if (ex1 <= ex0 || ey1 <= ey0) { grid = dim3(0, 0, 0); return; }
    grid = dim3((ex1 - ex0 + WPB * 8 - 1) / (WPB * 8), (ey1 - ey0 + 7) / 8, nz);
}

int render_grid_blocks(const FrameParams &p)
{
    dim3 g;
    render_grid(p, g);
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

#ifdef RMDF_XCHECK
// (librmdf_xcheck.so only, never run on hardware: the same sort with the band geometry of a one-launch whole-frame host call in the key)
__global__ __launch_bounds__(1024) void k_order_blocks_bands(const unsigned *__restrict__ cost, int n, unsigned *__restrict__ order, int gx, int band_strip_rows, int nbands)
{
    __shared__ unsigned hist[256], base[256], s_maxbin;
    const int tid = threadIdx.x;
    if (tid < 256) hist[tid] = 0u;
    if (tid == 0) s_maxbin = 0u;
    __syncthreads();
    auto bin_of = [](unsigned c) -> unsigned {
        // 8 sub-bins per power of two: monotone in c, 0..255
        if (c < 8u) return c;
        const int e = 31 - __builtin_clz(c);              // >= 3
        const unsigned b = (unsigned)(e - 2) * 8u + ((c >> (e - 3)) & 7u);
        return b > 255u ? 255u : b;
    };
    if (nbands > 0) {
        for (int i = tid; i < n; i += 1024) atomicMax(&s_maxbin, bin_of(cost[i]));
        __syncthreads();
    }
    const unsigned long_bin = s_maxbin > 8u ? s_maxbin - 8u : 0u;          // within a factor two of the costliest strip
    // sort key, descending: plain LPT = the cost bin; with bands: the long strips by cost bin (keys 128 ..), then the bands, outer first
    auto key_of = [&](int i) -> unsigned {
        const unsigned b = bin_of(cost[i]);
        if (nbands <= 0) return b;
        if (b >= long_bin && s_maxbin > 16u) return 128u + (b >> 1);
        const int band = (i / gx) / band_strip_rows;
        const int from_edge = band < nbands - 1 - band ? 2 * band : 2 * (nbands - 1 - band) + 1;        // 0, 1 = the two outer bands, ...
        return (unsigned)(127 - (from_edge < 127 ? from_edge : 127));
    };
    for (int i = tid; i < n; i += 1024) atomicAdd(&hist[key_of(i)], 1u);
    __syncthreads();
    if (tid == 0) {
        unsigned acc = 0u;
        for (int b = 255; b >= 0; b--) { base[b] = acc; acc += hist[b]; }   // descending key
    }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) order[atomicAdd(&base[key_of(i)], 1u)] = (unsigned)i;
}
#endif

hipError_t launch_order_blocks(const unsigned *d_cost, int n, unsigned *d_order, hipStream_t stream, int gx, int band_strip_rows, int nbands)
{
#ifdef RMDF_XCHECK
    if (nbands > 0) {
        hipLaunchKernelGGL(k_order_blocks_bands, dim3(1), dim3(1024), 0, stream, d_cost, n, d_order, gx, band_strip_rows, nbands);
        return hipGetLastError();
    }
#endif
    (void)gx; (void)band_strip_rows; (void)nbands;
    hipLaunchKernelGGL(k_order_blocks, dim3(1), dim3(1024), 0, stream, d_cost, n, d_order);
    return hipGetLastError();
}

hipError_t launch_render(int scene, const FrameParams &p, hipStream_t stream)
{
    dim3 grid;
    render_grid(p, grid);
    if (grid.x == 0) return hipSuccess;
    if (!p.rgba8 && !(p.rgba_f32 || p.steps || p.iters)) return hipErrorInvalidValue;
    if (scene == 2)      launch_render_scene_2(p, grid, stream);
    else if (scene == 0) launch_render_scene_0(p, grid, stream);
    else if (scene == 1) launch_render_scene_1(p, grid, stream);
    else if (scene == 3) launch_render_scene_3(p, grid, stream);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}
#endif

}  // namespace rmdf
