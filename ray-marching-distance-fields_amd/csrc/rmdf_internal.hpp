// rmdf_internal.hpp -- structures shared by the C-ABI host code and the kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>

#include "rmdf_device.hpp"

namespace rmdf {

// Everything one frame's kernels need; passed by value (kernarg segment).
struct FrameParams {
    // camera block of main() (fragment.shd:883-902) computed once on the host:
    // xaxis, yaxis, zaxis, eye of lookat() (829-838)
    float cam[12];
    float fov_xs;             // tan(radians(67.5)/2), fragment.shd:866-867
    float power;              // FSMBGeneralShader: animated power (fragment.shd:116-119), uniform per frame
    float wf, hf, aspect;     // in_screen_wdh, in_screen_hgt, wdh/hgt
    int   w, h;
    int   max_steps;          // fragment.shd:634
    // pixel rectangle to produce (single-rect mode, n_shard_tiles == 0)
    int   x0, y0, x1, y1;
    // tile-shard mode: blockIdx.z = slot; tile idx = shard_tile[slot] (shard_tiles_of_rank below);
    // shard_key identifies (rank, nranks) for the dispatch-order cache
    int   n_shard_tiles, shard_key;
    unsigned char shard_tile[64];
    // outputs: element (px,py) lives at out_base + (px-ox) + (py-oy)*pitch, where for
    // single-rect mode ox=oy=0,pitch=w and for shard mode the slot's packed tile
    CubeDev env_refl, env_cos1, env_cos8;
    const float *cornell;     // 96 vertices
    const float *cornell_tab; // 32 x CORNELL_STRIDE per-triangle constants (rmdf_device.hpp: de_cornell_box_table)
    const unsigned *cornell_grid;   // CORNELL_FINE_N^3 candidate masks in global memory (rmdf_device.hpp: de_cornell_box_lanes); null = none
    int   cornell_prune;      // skip triangles that provably cannot undercut the running minimum (bit-identical result)
    uint32_t *rgba8;
    uint32_t *rgba8_mirror;   // optional second destination of the RGBA8 frame: a registered (GPU-mapped) host buffer, written
                              // by the render kernel itself so that the copy over PCIe hides behind the frame (k_render only)
    float4   *rgba_f32;
    uint16_t *steps;
    uint16_t *iters;
    // cost-ordered dispatch of the render kernel (DESIGN.md 'critical path'): workgroup b renders strip block_order[b]
    // (null = raster order) and writes its cost (largest escape-iteration + step total of one of its pixels) to block_cost[]
    int merge_stragglers;     // k_render: pool the last rays of a workgroup's four packets in one wave (MERGE variant)
    const unsigned *block_order;
    unsigned *block_cost;
#ifdef RMDF_XCHECK
    // librmdf_xcheck.so only -- the one-launch band hand-over (rmdf_config.reserved[3] = 2, 3; never run on hardware, so not part of the
    // product: the product's FrameParams, and with it every product kernel, stays what the GPU tier last ran).  Whole-frame calls into
    // host memory (rmdf_api.cpp: render_whole_frame_one_launch, OUT_MIRROR variant only): the frame's strip rows are
    // grouped into bands of band_strip_rows rows of strips; the workgroup that completes a band (band_count: device counters, one per
    // band, left at zero) writes band_seq to band_flag[band] (host-mapped) once every mirror store of the band has landed in host
    // memory, and the host copies that band to the caller while the rest of the frame is still rendering.  null = no bands.
    unsigned *band_count;
    volatile unsigned *band_flag;
    unsigned band_seq;
    int band_strip_rows;
    // librmdf_xcheck.so only: the alternative schedule (xcheck/rmdf_march.hip) and the measurement aids
    // G-buffer written by k_march_mb8, read by k_shade; indexed px + py*gw over the frame padded to even
    // dimensions (helper pixels of odd sizes)
    float4   *gbuf_nao;       // normal.xyz, ao (hit pixels only)
    unsigned *gbuf_meta;      // bits 0..14 steps, bit 15 hit, bits 16..31 escape iterations (saturated)
    int       gw;
    int      *work_counter;   // zeroed before every march launch
    int       total_items, items_per_shard_tile;
    int       tail_t, shade_t, refill_t, chunk;   // scheduling thresholds of k_march_mb8 (rmdf_march.hip)
    unsigned long long *dbg;  // optional per-wave counters (8 or 16 x u64 per wave), may be null
    float     fold_min;       // underflow bound of the folded Mandelbulb passes (RMDF_MB8_FOLD_MIN; +inf under RMDF_FLAG_FORCE_WRITTEN)
#endif
};

#ifdef RMDF_XCHECK
// (the CPU tier's stand-in kernels read the band fields as a tail behind the product's FrameParams)
static_assert(offsetof(FrameParams, band_count) == offsetof(FrameParams, block_cost) + sizeof(unsigned *), "the band fields follow block_cost");
#endif

// Which of the reference's 64 tiles rank `rank` of `nranks` renders, in slot order; returns how many (<= ceil(64/n)).
// cost == nullptr: the static deal.  The scenes sit in the middle of the frame (the camera looks at the origin), so the
// tiles are sorted by their squared distance from the frame centre (nearest first; idx order among equals) and dealt
// to the ranks boustrophedon (0..n-1, n-1..0, ...); a plain idx mod n would hand rank r whole tile COLUMNS for n = 8.
// cost != nullptr (64 per-tile costs, e.g. rmdf_probe_tile_costs): longest-processing-time-first -- tiles in
// descending cost order (idx order among equals), each to the least loaded rank that still has a free slot (lowest
// rank among equals).  Deterministic: ranks that hold the same costs compute the same deal without talking.
// root_handicap: see rmdf_set_shard_root_handicap (rmdf.h).
inline int shard_tiles_of_rank(int rank, int nranks, unsigned char tiles[64], const float *cost = nullptr, float root_handicap = 0.0f)
{
    int order[64], n = 0;
    if (!cost) {
        for (int d2 = 2; d2 <= 98; d2 += 2)
            for (int idx = 0; idx < 64; idx++) {
                const int ax = 2 * (idx % 8) - 7, ay = 2 * (idx / 8) - 7;
                if (ax * ax + ay * ay == d2) order[n++] = idx;
            }
        int cnt = 0;
        for (int j = 0; j < 64; j++) {
            const int round = j / nranks, pos = j % nranks;
            if (((round & 1) ? nranks - 1 - pos : pos) == rank) tiles[cnt++] = (unsigned char)order[j];
        }
        return cnt;
    }
    for (int i = 0; i < 64; i++) order[i] = i;
    for (int i = 1; i < 64; i++) {                       // stable insertion sort, descending cost
        const int t = order[i];
        int j = i;
        while (j > 0 && cost[order[j - 1]] < cost[t]) { order[j] = order[j - 1]; j--; }
        order[j] = t;
    }
    const int cap = (64 + nranks - 1) / nranks;
    double load[64];
    int used[64], cnt = 0;
    for (int r = 0; r < nranks; r++) { load[r] = 0.0; used[r] = 0; }
    // rank 0 also receives and assembles every frame: it starts the deal with that much load (a fraction of a rank's fair share)
    { double total = 0.0; for (int i = 0; i < 64; i++) total += (double)cost[i]; load[0] = (double)root_handicap * total / nranks; }
    for (int j = 0; j < 64; j++) {
        int best = -1;
        for (int r = 0; r < nranks; r++)
            if (used[r] < cap && (best < 0 || load[r] < load[best])) best = r;
        load[best] += (double)cost[order[j]];
        used[best]++;
        if (best == rank) tiles[cnt++] = (unsigned char)order[j];
    }
    return cnt;
}

// tile idx -> pixel rectangle (ShaderRendering.hs:183-193), host copy in rmdf_api.cpp
void tile_rect_host(int tile_idx, int w, int h, int *x0, int *y0, int *x1, int *y1);

// rmdf_render.hip
hipError_t launch_render(int scene, const FrameParams &p, hipStream_t stream);
int render_grid_blocks(const FrameParams &p);              // number of 32x8 strips launch_render() uses for p
// Longest-processing-time-first order of the strips.  (librmdf_xcheck.so: gx / band_strip_rows / nbands = the band geometry of a one-launch
// whole-frame host call -- the order is then "the strips within a factor two of the costliest first, the others band by band, outer bands
// first"; the product library has no such launch and takes the plain order whatever it is handed.)
hipError_t launch_order_blocks(const unsigned *d_cost, int n, unsigned *d_order, hipStream_t stream, int gx = 0, int band_strip_rows = 0, int nbands = 0);
// rmdf_util.hip
hipError_t launch_resolve_box2(const uint32_t *d_src, int sw, int sh, uint32_t *d_dst, hipStream_t stream);
hipError_t launch_selftest_exact_math(unsigned long long *d_counts, const float *d_cornell_tab, hipStream_t stream);
hipError_t launch_selftest_pinned_math(unsigned long long *d_counts, hipStream_t stream);
hipError_t launch_selftest_shading_math(unsigned long long *d_counts, void *d_texels, int face_w, float fov_xs, hipStream_t stream);
#define RMDF_MAX_FRAME_SIDE 32768     // fill_params: 1 <= w, h <= this (rays per side, super-sampling included)
hipError_t launch_fill_u32(uint32_t *dst, uint32_t value, size_t n, hipStream_t stream);
hipError_t launch_clock_probe(unsigned long long *d_out, unsigned long long ticks, int busy, hipStream_t stream);
// rmdf_env.hip: d_uv = per-texel environment (u, v) of the faces (host-built, cube_uv_table_host); d_lutT / d_tcs = the
// prefilter's host-built cosine tables (prefilter_tables_host)
hipError_t launch_cube_upload(const float *d_faces_f32, int W, uint2 *d_padded, hipStream_t stream);
hipError_t launch_latlong_to_cube(const float *d_latlong, int w, int h, const float2 *d_uv, float *d_faces_f32, hipStream_t stream);
hipError_t launch_resize_latlong(const float *d_src, int sw, int sh, int dstw, int dsth, float *d_out, hipStream_t stream);
hipError_t launch_prefilter_fused4(const float *d_src, int w, int h, const float *d_lutT, const float2 *d_tcs, float *const d_out[4],
                                   hipStream_t stream);
int prefilter_log2p(float power);
hipError_t launch_prefilter(const float *d_src, int w, int h, float power, const float *d_lutT, const float2 *d_tcs,
                            float *d_out, hipStream_t stream, bool split_ok);
#ifdef RMDF_XCHECK
hipError_t launch_march_stats(const FrameParams &p, hipStream_t stream);                               // xcheck/rmdf_stats.hip
hipError_t launch_render_mb8(const FrameParams &p, hipStream_t stream, int num_cus);                   // xcheck/rmdf_march.hip
#endif
// where[tile idx] = rank << 8 | slot under the deal in effect (built by the caller, cached per ctx)
struct ShardWhere { unsigned short v[64]; };
hipError_t launch_assemble_shards(const uint32_t *d_gathered, uint32_t *d_frame, int w, int h, int nranks, const ShardWhere &where,
                                  hipStream_t stream);


}  // namespace rmdf
