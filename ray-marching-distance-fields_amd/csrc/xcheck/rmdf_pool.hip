// rmdf_pool.hip -- Mandelbulb power-8 march kernel with a wave-local ray pool staged in LDS.
//
// Why: a distance estimate (DE) is k Mandelbulb iterations (k = 1..25, mean 3.2) followed by a "tail"
// (log, divide, march / normal / AO bookkeeping) of about the weight of two iterations.  A wave that keeps
// one ray per lane must run the tail block whenever ANY lane needs it -- at ~25 % lane utilisation -- or let
// finished lanes wait -- idling the iteration block.  Here every 64-lane wave owns a pool of R = 128 rays
// in LDS and four LDS ring buffers:
//     REQ     distance estimates ready to start      (slot, pos.zxy, squares, |pos|)
//     DONE_M  finished estimates of marching rays    (slot, r, dr, iterations)
//     DONE_S  finished estimates of normal / AO taps (same)
//     FREE    ray slots without a pixel
// Lanes only ever run the iteration block: a lane whose estimate ends pushes the result to DONE_x and pops the
// next REQ in the same pass, so the iteration block stays ~full.  The tail blocks run on 64 DONE entries at a
// time, i.e. at full lane utilisation, reading and writing the ray records in LDS.  Nothing is shared between
// waves (no barriers, no atomics except the global pixel counter), all queue heads are wave-uniform scalars.
// Per-ray arithmetic and its order are exactly those of the nested formulation (fragment.shd:101-158,
// 463-470, 542-562, 618-676): step counts and escape-iteration counts stay bit-exact.
//
// Compile with -ffp-contract=off, no fast-math (see rmdf_device.hpp).
#include <stdlib.h>

#include "rmdf_internal.hpp"

namespace rmdf {

#define POOL_R 128
#define POOL_MASK (POOL_R - 1)

#define PH_MARCH 0
#define PH_N0 1
#define PH_NX 2
#define PH_NY 3
#define PH_NZ 4
#define PH_AO0 5
#define PH_AO1 6

struct PoolLDS {
    // ray records
    float    t[POOL_R], tmax[POOL_R], dx[POOL_R], dy[POOL_R], dz[POOL_R];
    int      pix[POOL_R];
    unsigned steps[POOL_R], iters[POOL_R], phase[POOL_R];
    float    isx[POOL_R], isy[POOL_R], isz[POOL_R], nx[POOL_R], ny[POOL_R], nz[POOL_R];
    float    d0[POOL_R], d1[POOL_R], d2[POOL_R], occl[POOL_R];
    // REQ ring: bit 8 of req_slot = estimate belongs to a normal / AO tap
    int      req_slot[POOL_R];
    float    req_px[POOL_R], req_py[POOL_R], req_pz[POOL_R], req_x2[POOL_R], req_y2[POOL_R], req_z2[POOL_R], req_r[POOL_R];
    // DONE rings
    int      dm_slot[POOL_R]; float dm_r[POOL_R], dm_dr[POOL_R]; unsigned dm_it[POOL_R];
    int      ds_slot[POOL_R]; float ds_r[POOL_R], ds_dr[POOL_R]; unsigned ds_it[POOL_R];
    int      free_slot[POOL_R];
};

__device__ __forceinline__ int pool_popc(unsigned long long m) { return __popcll(m); }
__device__ __forceinline__ int pool_rank(unsigned long long m)
{
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

__device__ __forceinline__ bool pool_item_to_pixel(const FrameParams &p, int item, int &px, int &py)
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

// fragment.shd:74-99 with the squares of the components passed in
__device__ __forceinline__ v3 pool_triplex_pow8(float x, float y, float z, float x2, float y2, float z2)
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

__global__ __launch_bounds__(256) void k_march_pool(const FrameParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char pool_smem[];
    PoolLDS &L = *reinterpret_cast<PoolLDS *>(pool_smem + (size_t)(threadIdx.x >> 6) * sizeof(PoolLDS));
    const int lane = threadIdx.x & 63;
    const v3 origin = mk3(p.cam[9], p.cam[10], p.cam[11]);
    const int max_steps = p.max_steps;
    const int total = p.total_items;
    const int chunk = p.chunk, low = p.pool_low;

    // wave-uniform ring-buffer cursors (monotonic; index = cursor & POOL_MASK)
    int req_h = 0, req_t = 0, dm_h = 0, dm_t = 0, ds_h = 0, ds_t = 0, free_h = 0, free_t = POOL_R;
    L.free_slot[lane] = lane;
    L.free_slot[lane + 64] = lane + 64;
    int next = 0, end = 0;
    bool px_exhausted = false;

    // diagnostics
    unsigned long long c_iter = 0, c_mtail = 0, c_stail = 0, c_init = 0, a_iter = 0, a_mtail = 0, a_stail = 0;
    const unsigned long long t_begin = p.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull;
    unsigned long long y_refill = 0, y_iter = 0, y_push = 0, y_mtail = 0, y_stail = 0, y_init = 0, y0 = 0;
#define PSTAMP() (p.dbg ? __builtin_amdgcn_s_memrealtime() : 0ull)

    // lane-resident distance-estimate state
    bool  active = false;
    int   slot = 0, kind = 0, it = 0;
    float posx = 0, posy = 0, posz = 0, wx = 0, wy = 0, wz = 0, x2 = 0, y2 = 0, z2 = 0, r = 0, dr = 1.0f;

    // enqueue "estimate the distance at world point q for ray slot s": pos = q.zxy, its squares and |pos|
#define PUSH_REQ(mask_var, cond, s, knd, qx, qy, qz)                                                 \
    do {                                                                                             \
        if (cond) {                                                                                  \
            const int i_ = (req_t + pool_rank(mask_var)) & POOL_MASK;                                \
            const float ppx = (qz), ppy = (qx), ppz = (qy);                                          \
            const float sx = ppx * ppx, sy = ppy * ppy, sz = ppz * ppz;                              \
            L.req_slot[i_] = (s) | ((knd) << 8);                                                     \
            L.req_px[i_] = ppx; L.req_py[i_] = ppy; L.req_pz[i_] = ppz;                              \
            L.req_x2[i_] = sx; L.req_y2[i_] = sy; L.req_z2[i_] = sz;                                 \
            L.req_r[i_] = sqrt_rn((sx + sy) + sz);                                                   \
        }                                                                                            \
        req_t += pool_popc(mask_var);                                                                \
    } while (0)

    for (;;) {
        // ---------------- lanes without an estimate take the next request --------------------------
        y0 = PSTAMP();
        {
            const unsigned long long m_empty = __ballot(!active);
            const int n_req = req_t - req_h;
            if (m_empty != 0ull && n_req > 0) {
                const int rank = pool_rank(m_empty);
                if (!active && rank < n_req) {
                    const int i = (req_h + rank) & POOL_MASK;
                    const int sk = L.req_slot[i];
                    slot = sk & 0xff; kind = sk >> 8;
                    posx = L.req_px[i]; posy = L.req_py[i]; posz = L.req_pz[i];
                    x2 = L.req_x2[i]; y2 = L.req_y2[i]; z2 = L.req_z2[i];
                    r = L.req_r[i];
                    wx = posx; wy = posy; wz = posz; dr = 1.0f; it = 0;
                    active = true;
                }
                const int n_empty = pool_popc(m_empty);
                req_h += (n_empty < n_req) ? n_empty : n_req;
            }
        }
        y_refill += PSTAMP() - y0;

        // ---------------- one Mandelbulb iteration for every lane (fragment.shd:134-149) -----------
        const unsigned long long m_act = __ballot(active);
        if (m_act != 0ull) {
            c_iter++; a_iter += pool_popc(m_act);
            y0 = PSTAMP();
            bool fin = false;
            if (active) {
                if (r > 4.0f) {
                    fin = true;                               // bailout on the radius test (:138-139)
                } else {
                    const v3 nw = pool_triplex_pow8(wx, wy, wz, x2, y2, z2);
                    wx = nw.x + posx; wy = nw.y + posy; wz = nw.z + posz;
                    const float r2 = r * r, r4 = r2 * r2, r7 = (r4 * r2) * r;
                    dr = r7 * 8.0f * dr + 1.0f;
                    it++;
                    if (it == 25) {
                        fin = true;                           // loop exhausted: r keeps the last tested radius
                    } else {
                        x2 = wx * wx; y2 = wy * wy; z2 = wz * wz;
                        r = sqrt_rn((x2 + y2) + z2);          // next radius test
                        fin = r > 4.0f;
                    }
                }
            }
            y_iter += PSTAMP() - y0; y0 = PSTAMP();
            const unsigned long long m_fm = __ballot(fin && kind == 0), m_fs = __ballot(fin && kind != 0);
            if (m_fm != 0ull) {
                if (fin && kind == 0) {
                    const int i = (dm_t + pool_rank(m_fm)) & POOL_MASK;
                    L.dm_slot[i] = slot; L.dm_r[i] = r; L.dm_dr[i] = dr; L.dm_it[i] = (unsigned)it;
                    active = false;
                }
                dm_t += pool_popc(m_fm);
            }
            if (m_fs != 0ull) {
                if (fin && kind != 0) {
                    const int i = (ds_t + pool_rank(m_fs)) & POOL_MASK;
                    L.ds_slot[i] = slot; L.ds_r[i] = r; L.ds_dr[i] = dr; L.ds_it[i] = (unsigned)it;
                    active = false;
                }
                ds_t += pool_popc(m_fs);
            }
            y_push += PSTAMP() - y0;
        }

        // ---------------- bulk stages ----------------------------------------------------------------
        const int n_act = pool_popc(__ballot(active));
        const bool starving = (n_act < low) && (req_t == req_h);
        const int n_dm = dm_t - dm_h, n_ds = ds_t - ds_h;

        // march tail: ray_march loop body after the DE call (fragment.shd:663-672), 64 rays at a time
        if (n_dm >= 64 || (starving && n_dm > 0)) {
            const int n = n_dm < 64 ? n_dm : 64;
            c_mtail++; a_mtail += n;
            y0 = PSTAMP();
            const bool mine = lane < n;
            bool cont = false, done = false;
            int s = 0, knd = 0;
            float qx = 0, qy = 0, qz = 0;
            if (mine) {
                const int i = (dm_h + lane) & POOL_MASK;
                s = L.dm_slot[i];
                const float rr = L.dm_r[i], ddr = L.dm_dr[i];
                const unsigned its = L.dm_it[i];
                const float dist = 0.5f * log_pinned(rr) * rr / ddr;        // fragment.shd:157
                float tt = L.t[s];
                const float tmx = L.tmax[s];
                const float ddx = L.dx[s], ddy = L.dy[s], ddz = L.dz[s];
                unsigned stp = L.steps[s];
                const unsigned itr = L.iters[s] + its;
                L.iters[s] = itr;
                tt += dist;
                bool out = tt > tmx;
                const bool hit = !out && (dist < 0.001f);
                if (!out && !hit) { stp++; out = (int)stp >= max_steps; }
                if (out) {
                    p.gbuf_meta[L.pix[s]] = stp | ((itr > 65535u ? 65535u : itr) << 16);
                    done = true;
                } else {
                    cont = true;
                    L.t[s] = tt; L.steps[s] = stp;
                    if (hit) {
                        // intersection and the backed-off point for the normal (fragment.shd:743-751)
                        const float ix = origin.x + ddx * tt, iy = origin.y + ddy * tt, iz = origin.z + ddz * tt;
                        L.isx[s] = ix; L.isy[s] = iy; L.isz[s] = iz;
                        qx = ix - ddx * 0.00001f; qy = iy - ddy * 0.00001f; qz = iz - ddz * 0.00001f;
                        L.phase[s] = PH_N0;
                        knd = 1;
                    } else {
                        qx = origin.x + tt * ddx; qy = origin.y + tt * ddy; qz = origin.z + tt * ddz;
                    }
                }
            }
            dm_h += n;
            const unsigned long long m_cont = __ballot(cont), m_done = __ballot(done);
            PUSH_REQ(m_cont, cont, s, knd, qx, qy, qz);
            if (m_done != 0ull) {
                if (done) L.free_slot[(free_t + pool_rank(m_done)) & POOL_MASK] = s;
                free_t += pool_popc(m_done);
            }
            y_mtail += PSTAMP() - y0;
        }

        // shade tail: normal taps (fragment.shd:463-470) and AO taps (542-562)
        if (n_ds >= 64 || (starving && n_ds > 0)) {
            const int n = n_ds < 64 ? n_ds : 64;
            c_stail++; a_stail += n;
            y0 = PSTAMP();
            const bool mine = lane < n;
            bool cont = false, done = false;
            int s = 0;
            float qx = 0, qy = 0, qz = 0;
            if (mine) {
                const int i = (ds_h + lane) & POOL_MASK;
                s = L.ds_slot[i];
                const float rr = L.ds_r[i], ddr = L.ds_dr[i];
                const unsigned its = L.ds_it[i];
                const float dist = 0.5f * log_pinned(rr) * rr / ddr;        // fragment.shd:157
                const unsigned itr = L.iters[s] + its;
                L.iters[s] = itr;
                const unsigned ph = L.phase[s];
                const float ix = L.isx[s], iy = L.isy[s], iz = L.isz[s];
                const float eps = 0.00001f;
                if (ph <= PH_NY) {
                    // N0: c = DE(p); then DE(p - eps*x), DE(p - eps*y), DE(p - eps*z), p = isec - dir*1e-5
                    const float npx = ix - L.dx[s] * 0.00001f, npy = iy - L.dy[s] * 0.00001f, npz = iz - L.dz[s] * 0.00001f;
                    if (ph == PH_N0) L.d0[s] = dist; else if (ph == PH_NX) L.d1[s] = dist; else L.d2[s] = dist;
                    qx = npx - ((ph == PH_N0) ? eps : 0.0f);
                    qy = npy - ((ph == PH_NX) ? eps : 0.0f);
                    qz = npz - ((ph == PH_NY) ? eps : 0.0f);
                    cont = true;
                } else if (ph == PH_NZ) {
                    const float c0 = L.d0[s];
                    const v3 nn = normalize3(mk3(c0 - L.d1[s], c0 - L.d2[s], c0 - dist));
                    L.nx[s] = nn.x; L.ny[s] = nn.y; L.nz[s] = nn.z;
                    qx = ix + nn.x * 0.016f; qy = iy + nn.y * 0.016f; qz = iz + nn.z * 0.016f;   // AO tap 1
                    cont = true;
                } else if (ph == PH_AO0) {
                    float oc = 0.0f;
                    oc += 0.5f * gclamp(1.0f - dist / 0.016f, 0.0f, 1.0f);
                    L.occl[s] = oc;
                    qx = ix + L.nx[s] * 0.081f; qy = iy + L.ny[s] * 0.081f; qz = iz + L.nz[s] * 0.081f;   // AO tap 2
                    cont = true;
                } else {
                    float oc = L.occl[s];
                    oc += 0.25f * gclamp(1.0f - dist / 0.081f, 0.0f, 1.0f);
                    oc = 1.0f - oc;
                    oc -= 0.29f;
                    oc *= 3.5f;
                    oc *= oc;
                    const float ao = gclamp(oc, 0.0f, 1.0f);
                    const int pixi = L.pix[s];
                    p.gbuf_nao[pixi] = make_float4(L.nx[s], L.ny[s], L.nz[s], ao);
                    p.gbuf_meta[pixi] = L.steps[s] | 0x8000u | ((itr > 65535u ? 65535u : itr) << 16);
                    done = true;
                }
                if (cont) L.phase[s] = ph + 1u;
            }
            ds_h += n;
            const unsigned long long m_cont = __ballot(cont), m_done = __ballot(done);
            PUSH_REQ(m_cont, cont, s, 1, qx, qy, qz);
            if (m_done != 0ull) {
                if (done) L.free_slot[(free_t + pool_rank(m_done)) & POOL_MASK] = s;
                free_t += pool_popc(m_done);
            }
            y_stail += PSTAMP() - y0;
        }

        // new rays for free slots: generate_ray + bounding sphere (fragment.shd:840-871, 595-616, 651-657)
        const int n_free = free_t - free_h;
        if (!px_exhausted && (n_free >= 64 || (starving && n_free > 0))) {
            y0 = PSTAMP();
            if (next >= end) {
                int base = 0;
                if (lane == 0) base = atomicAdd(p.work_counter, chunk);
                base = __builtin_amdgcn_readfirstlane(base);
                if (base >= total) {
                    px_exhausted = true;
                } else {
                    next = base;
                    end = base + chunk < total ? base + chunk : total;
                }
            }
            if (!px_exhausted) {
                int n = n_free < 64 ? n_free : 64;
                if (n > end - next) n = end - next;
                c_init++;
                const bool mine = lane < n;
                bool cont = false, back = false;
                int s = 0;
                float qx = 0, qy = 0, qz = 0;
                if (mine) {
                    s = L.free_slot[(free_h + lane) & POOL_MASK];
                    int px, py;
                    back = true;
                    if (pool_item_to_pixel(p, next + lane, px, py)) {
                        const int pixi = px + py * p.gw;
                        const float ndcx = ((float)px + 0.5f) / p.wf * 2.0f - 1.0f;
                        const float ndcy = ((float)py + 0.5f) / p.hf * 2.0f - 1.0f;
                        const v3 dc = normalize3(mk3(ndcx * p.fov_xs, ndcy * p.fov_xs / p.aspect, -1.0f));
                        const float ddx = p.cam[0] * dc.x + p.cam[3] * dc.y + p.cam[6] * dc.z;
                        const float ddy = p.cam[1] * dc.x + p.cam[4] * dc.y + p.cam[7] * dc.z;
                        const float ddz = p.cam[2] * dc.x + p.cam[5] * dc.y + p.cam[8] * dc.z;
                        float tmin, tmx;
                        if (ray_sphere<false>(origin, mk3(ddx, ddy, ddz), 1.15f, tmin, tmx) && max_steps > 0) {
                            const float tt = gmax(0.0f, tmin);
                            L.t[s] = tt; L.tmax[s] = tmx; L.dx[s] = ddx; L.dy[s] = ddy; L.dz[s] = ddz;
                            L.pix[s] = pixi; L.steps[s] = 0u; L.iters[s] = 0u; L.phase[s] = PH_MARCH;
                            qx = origin.x + tt * ddx; qy = origin.y + tt * ddy; qz = origin.z + tt * ddz;
                            cont = true; back = false;
                        } else {
                            p.gbuf_meta[pixi] = 0u;           // no march: hit 0, steps 0
                        }
                    }
                }
                free_h += n;
                next += n;
                const unsigned long long m_cont = __ballot(cont), m_back = __ballot(back);
                PUSH_REQ(m_cont, cont, s, 0, qx, qy, qz);
                if (m_back != 0ull) {
                    if (back) L.free_slot[(free_t + pool_rank(m_back)) & POOL_MASK] = s;
                    free_t += pool_popc(m_back);
                }
            }
            y_init += PSTAMP() - y0;
        }

        if (px_exhausted && n_act == 0 && req_t == req_h && dm_t == dm_h && ds_t == ds_h) break;
    }
#undef PUSH_REQ
    if (p.dbg && lane == 0) {
        unsigned long long *d = p.dbg + (size_t)(blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
        d[0] = c_iter; d[1] = c_mtail; d[2] = c_stail; d[3] = c_init; d[4] = a_iter; d[5] = a_mtail;
        d[6] = t_begin; d[7] = __builtin_amdgcn_s_memrealtime();
        d[8] = a_stail; d[9] = y_refill; d[10] = y_iter; d[11] = y_push; d[12] = y_mtail; d[13] = y_stail; d[14] = y_init;
    }
}

hipError_t launch_march_pool(const FrameParams &p, int blocks, hipStream_t stream)
{
    const size_t lds = 4 * sizeof(PoolLDS);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute((const void *)k_march_pool, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL(k_march_pool, dim3(blocks), dim3(256), lds, stream, p);
    return hipGetLastError();
}

}  // namespace rmdf
