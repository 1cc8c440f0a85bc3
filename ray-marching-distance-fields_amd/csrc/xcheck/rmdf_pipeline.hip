// rmdf_pipeline.hip -- three-kernel schedule of the sphere tracer (all four scenes).
//
//   k_march_refill<SCENE>   persistent waves; every lane marches one ray with the plain nested loops of the
//                           reference (march loop around the distance-estimator loop), but a lane whose ray
//                           ended takes the next pixel of the wave's current 8x8 tile at the next step boundary
//                           instead of idling until the slowest ray of its packet is done (the 8x8 packets of the
//                           single-kernel schedule keep only ~0.69 of their lanes busy in the march loop).  Tiles
//                           come from one global counter, optionally through last frame's cost order.  Hits are
//                           appended to a compact hit list (wave-aggregated: one atomic per 64 hits).
//   k_normal_ao<SCENE>      one lane per HIT pixel of that list: the 4 + 2 (Cornell: 4 + 4) extra distance estimates
//                           of normal_backward_difference and distance_ao run in full waves.
//   k_shade (rmdf_march.hip) reads the G-buffer, does the quad min/mag decision, cube lookups, gamma, stores.
//
// Per-ray arithmetic and its order are those of the nested single-kernel schedule: bit-identical output.
#include <stdlib.h>
#include <string.h>

#include "rmdf_internal.hpp"

namespace rmdf {

__device__ __forceinline__ int pl_popc(unsigned long long m) { return __popcll(m); }
__device__ __forceinline__ int pl_rank(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

template <int SCENE>
__device__ __forceinline__ float pl_de(v3 pos, const FrameParams &p, unsigned &iters)
{
    if (SCENE == 2)      return de_mandelbulb8(pos, iters);
    else if (SCENE == 3) return de_mandelbulb_general(pos, p.power, iters);
    else if (SCENE == 1) return de_test_scene(pos);
    else                 { int hint = 0; return de_cornell_box_table(pos, p.cornell_tab, p.cornell_prune, hint); }   // no order hint in this schedule
}

template <int SCENE>
__device__ __forceinline__ float pl_bsphere() { return SCENE == 2 ? 1.15f : (SCENE == 3 ? 1.5f : 1.0f); }

// item -> pixel: items enumerate the 8x8 tiles of the even-aligned rectangle (or of the shard's tiles), 64 per tile
__device__ __forceinline__ bool pl_item_to_pixel(const FrameParams &p, int item, int &px, int &py)
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

__device__ __forceinline__ v3 pl_pixel_dir(const FrameParams &p, int px, int py)
{
    const float ndcx = ((float)px + 0.5f) / p.wf * 2.0f - 1.0f;
    const float ndcy = ((float)py + 0.5f) / p.hf * 2.0f - 1.0f;
    const v3 dc = normalize3(mk3(ndcx * p.fov_xs, ndcy * p.fov_xs / p.aspect, -1.0f));
    return mk3(p.cam[0] * dc.x + p.cam[3] * dc.y + p.cam[6] * dc.z,
               p.cam[1] * dc.x + p.cam[4] * dc.y + p.cam[7] * dc.z,
               p.cam[2] * dc.x + p.cam[5] * dc.y + p.cam[8] * dc.z);
}

template <int SCENE>
__global__ __launch_bounds__(256) void k_march_refill(const FrameParams p)
{
    __shared__ int s_hits[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int *hb = s_hits[wave];
    const v3 origin = mk3(p.cam[9], p.cam[10], p.cam[11]);
    const int max_steps = p.max_steps;
    const int n_tiles = p.total_items >> 6;
    const int refill_t = p.refill_t;

    // wave-uniform: current tile's unassigned items [next, end), hit-buffer fill, tile supply state
    int next = 0, end = 0, hb_count = 0;
    bool exhausted = false;

    // per-lane ray
    bool active = false;
    int pix = 0, steps = 0;
    unsigned iters = 0;
    float dirx = 0, diry = 0, dirz = 0, t = 0, tmax = 0;

    for (;;) {
        // ---------------- give idle lanes the next pixels ---------------------------------------------
        unsigned long long idle = __ballot(!active);
        int n_idle = pl_popc(idle);
        if (!exhausted && n_idle >= refill_t) {
            while (n_idle > 0) {
                if (next >= end) {
                    int k = 0;
                    if (lane == 0) k = atomicAdd(p.work_counter, 1);
                    k = __builtin_amdgcn_readfirstlane(k);
                    if (k >= n_tiles) { exhausted = true; break; }
                    const int tile = p.tile_order ? (int)p.tile_order[k] : k;
                    next = tile << 6;
                    end = next + 64;
                }
                const int avail = end - next;
                const int rank = pl_rank(idle);
                const bool take = !active && rank < avail;
                const int item = next + rank;
                next += (n_idle < avail) ? n_idle : avail;
                if (take) {
                    int px, py;
                    if (pl_item_to_pixel(p, item, px, py)) {
                        pix = px + py * p.gw;
                        const v3 d = pl_pixel_dir(p, px, py);
                        dirx = d.x; diry = d.y; dirz = d.z;
                        float tmin;
                        steps = 0; iters = 0;
                        if (ray_sphere<false>(origin, d, pl_bsphere<SCENE>(), tmin, tmax) && max_steps > 0) {
                            t = gmax(0.0f, tmin);
                            active = true;
                        } else {
                            p.gbuf_meta[pix] = 0u;               // no march: hit 0, steps 0
                        }
                    }
                }
                idle = __ballot(!active);
                n_idle = pl_popc(idle);
                if (n_idle < refill_t) break;
            }
        }
        if (__ballot(active) == 0ull) {
            if (exhausted) break;
            continue;
        }

        // ---------------- one march step for every active lane (fragment.shd:661-672) -----------------
        bool hit = false;
        if (active) {
            const v3 pos = mk3(origin.x + t * dirx, origin.y + t * diry, origin.z + t * dirz);
            const float dist = pl_de<SCENE>(pos, p, iters);
            t += dist;
            const bool out = t > tmax;
            hit = !out && (dist < 0.001f);
            bool done = out || hit;
            if (!done) { steps++; done = steps >= max_steps; }
            if (done) {
                p.gbuf_meta[pix] = (unsigned)steps | (hit ? 0x8000u : 0u) | ((iters > 65535u ? 65535u : iters) << 16);
                if (hit) p.gbuf_nao[pix] = make_float4(t, 0.0f, 0.0f, 0.0f);    // k_normal_ao picks t up here
                active = false;
            }
        }
        // ---------------- append hits to the global hit list, 64 at a time ----------------------------
        const unsigned long long m_hit = __ballot(hit);
        if (m_hit != 0ull) {
            const int n = pl_popc(m_hit);
            if (hb_count + n > 64) {
                int base = 0;
                if (lane == 0) base = atomicAdd(p.hit_count, hb_count);
                base = __builtin_amdgcn_readfirstlane(base);
                if (lane < hb_count) p.hit_list[base + lane] = hb[lane];
                hb_count = 0;
            }
            if (hit) hb[hb_count + pl_rank(m_hit)] = pix;
            hb_count += n;
        }
    }
    if (hb_count > 0) {
        int base = 0;
        if (lane == 0) base = atomicAdd(p.hit_count, hb_count);
        base = __builtin_amdgcn_readfirstlane(base);
        if (lane < hb_count) p.hit_list[base + lane] = hb[lane];
    }
}

// normal_backward_difference (fragment.shd:463-470) + distance_ao (542-591) for the hit pixels
template <int SCENE>
__global__ __launch_bounds__(256) void k_normal_ao(const FrameParams p)
{
    const int n_hits = *p.hit_count;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_hits) return;
    const int pix = p.hit_list[i];
    const int px = pix % p.gw, py = pix / p.gw;
    const v3 origin = mk3(p.cam[9], p.cam[10], p.cam[11]);
    const v3 dir = pl_pixel_dir(p, px, py);
    const float t = p.gbuf_nao[pix].x;
    unsigned iters = 0;
    const v3 isec = mk3(origin.x + dir.x * t, origin.y + dir.y * t, origin.z + dir.z * t);
    const v3 np = mk3(isec.x - dir.x * 0.00001f, isec.y - dir.y * 0.00001f, isec.z - dir.z * 0.00001f);
    const float eps = 0.00001f;
    const float d0 = pl_de<SCENE>(np, p, iters);
    const float dx = pl_de<SCENE>(mk3(np.x - eps, np.y - 0.0f, np.z - 0.0f), p, iters);
    const float dy = pl_de<SCENE>(mk3(np.x - 0.0f, np.y - eps, np.z - 0.0f), p, iters);
    const float dz = pl_de<SCENE>(mk3(np.x - 0.0f, np.y - 0.0f, np.z - eps), p, iters);
    const v3 n = normalize3(mk3(d0 - dx, d0 - dy, d0 - dz));
    float occl = 0.0f, ao;
    if (SCENE != 0) {
        const float w0 = 0.5f, e0 = 0.016f, w1 = 0.25f, e1 = 0.081f;
        occl += w0 * gclamp(1.0f - pl_de<SCENE>(mk3(isec.x + n.x * e0, isec.y + n.y * e0, isec.z + n.z * e0), p, iters) / e0, 0.0f, 1.0f);
        occl += w1 * gclamp(1.0f - pl_de<SCENE>(mk3(isec.x + n.x * e1, isec.y + n.y * e1, isec.z + n.z * e1), p, iters) / e1, 0.0f, 1.0f);
        occl = 1.0f - occl;
        occl -= 0.29f;
        occl *= 3.5f;
        occl *= occl;
        ao = gclamp(occl, 0.0f, 1.0f);
    } else {
        const float wt[4] = { 0.1f, 0.2f, 0.125f, 0.0625f }, dl[4] = { 0.1f, 0.2f, 0.4f, 0.5f };
#pragma unroll
        for (int k = 0; k < 4; k++)
            occl += wt[k] * gclamp(1.0f - pl_de<SCENE>(mk3(isec.x + n.x * dl[k], isec.y + n.y * dl[k], isec.z + n.z * dl[k]), p, iters) / dl[k], 0.0f, 1.0f);
        ao = 1.0f - occl;
    }
    p.gbuf_nao[pix] = make_float4(n.x, n.y, n.z, ao);
    const unsigned meta = p.gbuf_meta[pix];
    unsigned it = (meta >> 16) + iters;
    if (it > 65535u) it = 65535u;
    p.gbuf_meta[pix] = (meta & 0xffffu) | (it << 16);
}

hipError_t launch_shade(const FrameParams &p, int ew, int eh, int nz, hipStream_t stream);   // rmdf_march.hip

template <int SCENE>
static hipError_t launch_pipeline_t(const FrameParams &p, int blocks, int hit_blocks, int ew, int eh, int nz, hipStream_t stream)
{
    hipLaunchKernelGGL(k_march_refill<SCENE>, dim3(blocks), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_normal_ao<SCENE>, dim3(hit_blocks), dim3(256), 0, stream, p);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_shade(p, ew, eh, nz, stream);
}

hipError_t launch_render_pipeline(int scene, const FrameParams &p_in, hipStream_t stream, int num_cus)
{
    FrameParams p = p_in;
    int ew, eh, nz = 1;
    if (p.n_shard_tiles > 0) {
        ew = p.w / 8 + 2; eh = p.h / 8 + 2;
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
    static int wps = 0, refill = 0;
    if (!wps) {
        const char *e;
        wps = (e = getenv("RMDF_PIPE_WPS")) ? atoi(e) : 6;
        refill = (e = getenv("RMDF_PIPE_REFILL_T")) ? atoi(e) : 16;
        if (wps < 1) wps = 1;
        if (wps > 8) wps = 8;
    }
    p.refill_t = refill;
    // work counter and hit counter live side by side: one memset
    hipError_t e = hipMemsetAsync(p.work_counter, 0, 2 * sizeof(int), stream);
    if (e != hipSuccess) return e;
    int blocks = num_cus * wps;
    const int max_useful = (p.total_items / 64 + 3) / 4;
    if (blocks > max_useful) blocks = max_useful;
    if (blocks < 1) blocks = 1;
    const int hit_blocks = (p.total_items + 255) / 256;
    switch (scene) {
    case 0:  return launch_pipeline_t<0>(p, blocks, hit_blocks, ew, eh, nz, stream);
    case 1:  return launch_pipeline_t<1>(p, blocks, hit_blocks, ew, eh, nz, stream);
    case 2:  return launch_pipeline_t<2>(p, blocks, hit_blocks, ew, eh, nz, stream);
    case 3:  return launch_pipeline_t<3>(p, blocks, hit_blocks, ew, eh, nz, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace rmdf
