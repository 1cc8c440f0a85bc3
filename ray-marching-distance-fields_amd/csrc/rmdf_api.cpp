// rmdf_api.cpp -- the C ABI of librmdf.so (include/rmdf.h): host-side counterpart of
// ShaderRendering.hs (withShaderRenderer / drawShaderTile) for the HIP renderer.
// No CPU rendering path exists here: every pixel comes from the gfx950 kernels.
#include <sys/stat.h>
#include <dlfcn.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <functional>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include <rccl/rccl.h>          // types and prototypes only: the library is dlopen()ed by rmdf_comm_init

#include "../../include/rmdf.h"
#ifdef RMDF_XCHECK
#include "../../include/rmdf_xcheck.h"
#endif
#include "rmdf_internal.hpp"
#include "rmdf_host.hpp"

using namespace rmdf;

namespace {

// message of the last failed call that had no ctx (rmdf_create, rmdf_save_png, ...).  Process-wide and mutex-guarded, not
// thread-local: a host whose runtime migrates its green threads between OS threads (GHC without a bound thread) may ask for
// the message on another OS thread than the one the call failed on.  rmdf_last_error(NULL) hands out a per-thread copy.
std::mutex  g_error_mutex;
std::string g_global_error;

struct CubeSlot {
    uint2 *d_texels = nullptr;
    int    W = 0;
};

#define RMDF_MAX_ORDER_STREAMS 32
struct OrderState {
    hipStream_t stream = nullptr;
    unsigned   *d_cost = nullptr, *d_order = nullptr;
    int         cap = 0, n = 0;
    int         key[12] = { -1, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    bool        used = false, valid = false;
    unsigned    last_use = 0;
};

// host-built tables of the env-map kernels (rmdf_env.hip), cached per size
struct UvTable  { int cw = 0; float2 *d_uv = nullptr; };
struct LobeTable { int w = 0, h = 0; float *d_lutT = nullptr; float2 *d_tcs = nullptr; };

}  // namespace

struct rmdf_ctx {
    int          device = 0;
    hipStream_t  stream = nullptr;
    hipStream_t  pstream[4] = { nullptr, nullptr, nullptr, nullptr };   // the four lobe powers run concurrently (ShaderRendering.hs:142)
    hipEvent_t   ev_fork = nullptr, ev_join[4] = { nullptr, nullptr, nullptr, nullptr };
    float       *d_cornell = nullptr;
    float       *d_cornell_tab = nullptr;
    uint32_t    *d_cornell_grid = nullptr;
    hipStream_t  probe_stream = nullptr;   // rmdf_probe_shader_clock: highest priority, created on first use
    unsigned long long *probe_host = nullptr;   // ... and its two counters, in mapped host memory
    CubeSlot     env[RMDF_ENV_SLOTS];
    std::vector<UvTable>   uv_tables;
    std::vector<LobeTable> lobe_tables;
    // device scratch of the host-buffer env entry points (rmdf_prefilter_env_powers: source + one map per power): kept between
    // calls for maps up to the reference's own 256x128 (a hipMalloc / hipFree pair per call cost more than the kernels of such a
    // map); larger maps get per-call buffers, so that one call at the accepted maximum (8192x4096, 16 powers: 6.8 GB) holds nothing
    // afterwards.  One set per ctx: the env entry points are not re-entrant per ctx (rmdf.h: one caller thread per ctx).
    struct Scratch { void *p = nullptr; size_t bytes = 0; };
    static constexpr size_t kEnvScratchKeepBytes = (size_t)256 * 128 * 12;
    Scratch      env_scratch[17];
    // frame latched on the first tile (ShaderRendering.hs:162-176)
    int          w = 0, h = 0, max_steps = 128;
    float        time = 0.0f;
    bool         latched = false;
    // accumulating frame (device); the float / steps / iteration planes exist only once a caller asked for them
    uint32_t    *d_rgba8 = nullptr;
    float4      *d_rgba_f32 = nullptr;
    uint16_t    *d_steps = nullptr;
    uint16_t    *d_iters = nullptr;
    size_t       cap_px = 0, cap_planes_px = 0;
    // tile mode (rmdf_render_tile with tile_idx >= 0): a page-locked host copy of the accumulating frame.  A tile call moves only the
    // rows the tile touched over PCIe and hands the caller its whole frame from here (render_common).  shadow_valid: the copy equals
    // the device frame.
    uint32_t    *h_shadow = nullptr, *h_shadow_dev = nullptr;
    size_t       shadow_px = 0;
    bool         shadow_valid = false;
    WorkPool     pool;                 // host threads: frame copies, staging copies, table builders (ctx_pool() starts them)
    Staging      staging;              // the page-locked chunks between caller memory and the device (rmdf_host.hpp)
    // whole-frame calls into pageable memory (render_whole_frame_host): row bands on streams of their own
#define RMDF_WF_MAX_BANDS 16
#define RMDF_WF_DEFAULT_BANDS 2
#define RMDF_WF_DEFAULT_MODE 1
    hipStream_t  wf_stream[RMDF_WF_MAX_BANDS] = { nullptr };
    hipEvent_t   wf_done[RMDF_WF_MAX_BANDS] = { nullptr };
    hipEvent_t   wf_fork = nullptr;
    int          wf_bands = 0, wf_mirror = 0;      // rmdf_config.reserved[2], [3]
    volatile unsigned *wf_flags = nullptr, *wf_flags_dev = nullptr;       // one-launch hand-over: band k of frame `seq` is in the shadow when wf_flags[k] == seq
    unsigned    *d_wf_count = nullptr;
    unsigned     wf_seq = 0;
    long long    wf_last_us = 0;       // how long the previous whole-frame host call waited for its bands (bounds the copy threads' spin window)
    unsigned     spec_dropped = 0;     // tile jobs that could not be issued ahead of their call (render_tile_fast)
    unsigned     env_gen = 0;          // bumped whenever a cube-map slot changes: tile jobs rendered ahead belong to ONE environment
    // ... and the tile jobs of that mode: a tile is rendered into a device scratch tile AND, by the kernel's mirror store, into a
    // page-locked host tile (both at the frame's row pitch), on a stream of its own.  RMDF_TILE_JOBS sets of those: the call for tile
    // i issues the jobs of tiles i + 1 .. i + RMDF_TILE_JOBS - 1 of the same frame ahead of their calls, so they run side by side (a
    // tile's kernel lasts as long as its longest ray, however few rays it has) while the calls before them copy.  A job is used only
    // by the call whose (scene, tile, latched frame) it was issued for; anything else ignores it.
#define RMDF_TILE_JOBS 4
    struct TileJob {
        uint32_t *d_tile = nullptr, *h_tile = nullptr, *h_tile_dev = nullptr;
        size_t    px = 0;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr, copied = nullptr;     // the kernel has finished; the scratch tile has been copied into the frame
        bool      issued = false;
        int       scene = 0, idx = 0, w = 0, h = 0, max_steps = 0;
        float     time = 0.0f;
        unsigned  env_gen = 0;
    };
    TileJob      tile_job[RMDF_TILE_JOBS];
    // host buffers the caller declared with rmdf_register_host_buffer (bookkeeping only since round 5)
    struct HostReg { char *host; size_t bytes; };
    std::vector<HostReg> host_regs;
    // per-tile costs that steer the deal of tiles to ranks (rmdf_set_shard_costs); unset = static deal
    float        shard_cost[64];
    bool         shard_cost_set = false;
    unsigned     shard_cost_gen = 0;
    float        shard_root_handicap = 0.0f;
    // the deal of the last (nranks, cost generation) asked for: per-frame calls must not redo the sort
    int          deal_nranks = 0;
    unsigned     deal_gen = ~0u;
    unsigned char deal_tiles[64][64];
    int          deal_count[64];
    ShardWhere   deal_where;
    // cost-ordered dispatch state of the render kernel (previous frame's per-strip costs), one per stream that renders:
    // frames in flight on different streams (pipelined rendering) must not share the tables
    OrderState   orders[RMDF_MAX_ORDER_STREAMS];
    unsigned     order_tick = 0;
#ifdef RMDF_XCHECK
    // G-buffer + work counter of the alternative schedules (one set: they support the ctx stream only)
    float4      *d_gbuf_nao = nullptr;
    unsigned    *d_gbuf_meta = nullptr;
    int         *d_work_counter = nullptr;
    size_t       gbuf_cap = 0;
    unsigned long long *d_dbg = nullptr;   // per-wave march diagnostics (rmdf_debug_march_stats)
#endif
    // the job's RCCL communicator (rmdf_comm_init): one rank per GPU, this ctx is rank comm_rank of comm_nranks
    ncclComm_t   comm = nullptr;
    int          comm_rank = 0, comm_nranks = 1;
    uint64_t    *d_verify = nullptr;   // rmdf_comm_verify_deal: one fingerprint per rank + the verdict (allocated with the communicator)
    int          flags = 0;            // rmdf_config.reserved[0]
    int          copy_threads = 0;     // rmdf_config.reserved[1]: host threads of the tile-mode frame copy (0 = by core count)
    std::string  err;
    char         dev_name[256] = { 0 };
    int          cus = 0;
};

namespace {

int fail(rmdf_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) { ctx->err = msg; return code; }
    std::lock_guard<std::mutex> lock(g_error_mutex);
    g_global_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(ctx, RMDF_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

// the ctx's host threads, started on first use BY WORK THAT CAN USE THEM (a copy of at least 1 MB, a table builder): a ctx that only ever
// moves a few kilobytes (rmdf_create's tables, shard calls on device pointers) never starts a thread -- an unstarted pool runs every job
// on the calling thread.  rmdf_config.reserved[1] threads with the caller's, or by the CPUs this process may use (affinity mask and
// container quota, not the machine's core count: rmdf_host.hpp usable_cpus).
WorkPool &ctx_pool(rmdf_ctx *ctx, size_t bytes = (size_t)-1)
{
    if (ctx->pool.workers() == 0 && bytes >= ((size_t)1 << 20)) {
        const int hc = usable_cpus();
        ctx->pool.start(ctx->copy_threads ? ctx->copy_threads - 1 : (hc >= 64 ? 15 : (hc > 8 ? 7 : (hc > 1 ? hc - 1 : 0))));
    }
    return ctx->pool;
}

// Host memory <-> device memory, always through the ctx's page-locked staging (rmdf_host.hpp): the HIP runtime never sees a pointer
// into memory the library did not page-lock itself.  upload() may return before the DMA ends (h_src is not read after it returns);
// download() returns with the bytes in h_dst.
int upload(rmdf_ctx *ctx, void *d_dst, const void *h_src, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return RMDF_OK;
    const hipError_t e = ctx->staging.upload(ctx_pool(ctx, bytes), d_dst, h_src, bytes, st);
    return e == hipSuccess ? RMDF_OK : fail(ctx, RMDF_E_HIP, std::string("upload through staging: ") + hipGetErrorString(e));
}
int download(rmdf_ctx *ctx, void *h_dst, const void *d_src, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return RMDF_OK;
    const hipError_t e = ctx->staging.download(ctx_pool(ctx, bytes), h_dst, d_src, bytes, st);
    return e == hipSuccess ? RMDF_OK : fail(ctx, RMDF_E_HIP, std::string("download through staging: ") + hipGetErrorString(e));
}
#define RMDF_TRY(expr) do { const int rc_ = (expr); if (rc_ != RMDF_OK) return rc_; } while (0)

// Nothing may unwind through the C ABI into a foreign host (the Haskell viewer): every entry point that can allocate wraps
// its body in these.
#define RMDF_GUARD_BEGIN try {
#define RMDF_GUARD_END(ctx)                                                                                   \
    } catch (const std::bad_alloc &) { return fail(ctx, RMDF_E_NOMEM, "out of host memory");                  \
    } catch (const std::exception &e_) { return fail(ctx, RMDF_E_INVALID, std::string("exception: ") + e_.what()); \
    } catch (...) { return fail(ctx, RMDF_E_INVALID, "unknown exception"); }


// one IEEE rounding per operation on the host too (this file is built with -ffp-contract=off)
struct hv3 { float x, y, z; };
inline float hdot(hv3 a, hv3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline hv3 hnormalize(hv3 a)
{
    float s = 1.0f / sqrtf(hdot(a, a));
    return hv3{ a.x * s, a.y * s, a.z * s };
}
inline hv3 hcross(hv3 a, hv3 b) { return hv3{ a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x }; }

// camera block of main() (fragment.shd:883-902) + lookat (829-838), evaluated once per
// frame on the host instead of once per pixel; sinf/cosf/tanf are the host libm's
void host_camera(int scene, float time, float cam[12])
{
    hv3 c;
    if (scene == RMDF_FS_DE_CORNELL_BOX) {
        c = hv3{ sinf(time / 2.0f) * shk::camera_cornell_radius, cosf(time / 2.0f) * shk::camera_cornell_radius, shk::camera_cornell_z };
    } else {
        c = hv3{ sinf(time / 3.0f), cosf(time / 4.0f), cosf(time / 3.0f) };
        hv3 nrm = hnormalize(c);
        c = hv3{ nrm.x * shk::camera_distance, nrm.y * shk::camera_distance, nrm.z * shk::camera_distance };
    }
    hv3 zaxis = hnormalize(hv3{ c.x - 0.0f, c.y - 0.0f, c.z - 0.0f });
    hv3 xaxis = hnormalize(hcross(hv3{ 0.0f, 1.0f, 0.0f }, zaxis));
    hv3 yaxis = hcross(zaxis, xaxis);
    cam[0] = xaxis.x; cam[1] = xaxis.y; cam[2] = xaxis.z;
    cam[3] = yaxis.x; cam[4] = yaxis.y; cam[5] = yaxis.z;
    cam[6] = zaxis.x; cam[7] = zaxis.y; cam[8] = zaxis.z;
    cam[9] = c.x; cam[10] = c.y; cam[11] = c.z;
}

float host_fov_xs()
{
    float hfov = (shk::hfov_deg_a * shk::hfov_deg_b) * 0.017453292519943295f;   // radians(45.0 * 1.5)
    return tanf(hfov / 2.0f);
}

// ---- Cornell box geometry, CornellBox.hs:48-129 (data) and :21-46 (triangulation) ----
const float kCornellQuads[64][3] = {
    { 552.8f, 0.0f, 0.0f }, { 0.0f, 0.0f, 0.0f }, { 0.0f, 0.0f, 559.2f }, { 549.6f, 0.0f, 559.2f },
    { 556.0f, 548.8f, 0.0f }, { 556.0f, 548.8f, 559.2f }, { 0.0f, 548.8f, 559.2f }, { 0.0f, 548.8f, 0.0f },
    { 549.6f, 0.0f, 559.2f }, { 0.0f, 0.0f, 559.2f }, { 0.0f, 548.8f, 559.2f }, { 556.0f, 548.8f, 559.2f },
    { 0.0f, 0.0f, 559.2f }, { 0.0f, 0.0f, 0.0f }, { 0.0f, 548.8f, 0.0f }, { 0.0f, 548.8f, 559.2f },
    { 552.8f, 0.0f, 0.0f }, { 549.6f, 0.0f, 559.2f }, { 556.0f, 548.8f, 559.2f }, { 556.0f, 548.8f, 0.0f },
    { 343.0f, 548.8f - 0.1f, 227.0f }, { 343.0f, 548.8f - 0.1f, 332.0f }, { 213.0f, 548.8f - 0.1f, 332.0f }, { 213.0f, 548.8f - 0.1f, 227.0f },
    { 130.0f, 165.0f, 65.0f }, { 82.0f, 165.0f, 225.0f }, { 240.0f, 165.0f, 272.0f }, { 290.0f, 165.0f, 114.0f },
    { 290.0f, 0.0f, 114.0f }, { 290.0f, 165.0f, 114.0f }, { 240.0f, 165.0f, 272.0f }, { 240.0f, 0.0f, 272.0f },
    { 130.0f, 0.0f, 65.0f }, { 130.0f, 165.0f, 65.0f }, { 290.0f, 165.0f, 114.0f }, { 290.0f, 0.0f, 114.0f },
    { 82.0f, 0.0f, 225.0f }, { 82.0f, 165.0f, 225.0f }, { 130.0f, 165.0f, 65.0f }, { 130.0f, 0.0f, 65.0f },
    { 240.0f, 0.0f, 272.0f }, { 240.0f, 165.0f, 272.0f }, { 82.0f, 165.0f, 225.0f }, { 82.0f, 0.0f, 225.0f },
    { 423.0f, 330.0f, 247.0f }, { 265.0f, 330.0f, 296.0f }, { 314.0f, 330.0f, 456.0f }, { 472.0f, 330.0f, 406.0f },
    { 423.0f, 0.0f, 247.0f }, { 423.0f, 330.0f, 247.0f }, { 472.0f, 330.0f, 406.0f }, { 472.0f, 0.0f, 406.0f },
    { 472.0f, 0.0f, 406.0f }, { 472.0f, 330.0f, 406.0f }, { 314.0f, 330.0f, 456.0f }, { 314.0f, 0.0f, 456.0f },
    { 314.0f, 0.0f, 456.0f }, { 314.0f, 330.0f, 456.0f }, { 265.0f, 330.0f, 296.0f }, { 265.0f, 0.0f, 296.0f },
    { 265.0f, 0.0f, 296.0f }, { 265.0f, 330.0f, 296.0f }, { 423.0f, 330.0f, 247.0f }, { 423.0f, 0.0f, 247.0f },
};

void cornell_triangles(float out[96 * 3])
{
    const float to_unit = 559.2f / 2.0f;
    const float scale = 1.0f / (sqrtf(2.0f * 2.0f + 2.0f * 2.0f + 2.0f * 2.0f) / 2.0f) * 0.99f;
    static const int order[6] = { 0, 1, 3, 3, 1, 2 };   // (q0,q1,q3),(q3,q1,q2)
    for (int q = 0; q < 16; q++)
        for (int k = 0; k < 6; k++)
            for (int a = 0; a < 3; a++)
                out[(q * 6 + k) * 3 + a] = (kCornellQuads[q * 4 + order[k]][a] / to_unit - 1.0f) * scale;
}

// per-triangle constants of de_cornell_box_table: the operation order of de_triangle / line_seg_min_dist_sq
// (fragment.shd:312-372), evaluated once here instead of once per lane and distance estimate
void cornell_table(const float tri[96 * 3], float tab[CORNELL_TAB_FLOATS])
{
    for (int i = 0; i < 32; i++) {
        const float *v0 = tri + i * 9, *v1 = v0 + 3, *v2 = v0 + 6;
        float *t = tab + i * CORNELL_STRIDE;
        for (int k = 0; k < 9; k++) t[k] = v0[k];
        hv3 e0{ v2[0] - v0[0], v2[1] - v0[1], v2[2] - v0[2] }, e1{ v1[0] - v0[0], v1[1] - v0[1], v1[2] - v0[2] };
        hv3 e12{ v2[0] - v1[0], v2[1] - v1[1], v2[2] - v1[2] };
        const float dot00 = hdot(e0, e0), dot01 = hdot(e0, e1), dot11 = hdot(e1, e1);
        t[9] = e0.x; t[10] = e0.y; t[11] = e0.z; t[12] = e1.x; t[13] = e1.y; t[14] = e1.z;
        t[15] = dot00; t[16] = dot01; t[17] = dot11;
        t[18] = 1.0f / (dot00 * dot11 - dot01 * dot01);
        t[19] = e12.x; t[20] = e12.y; t[21] = e12.z;
        const float len12 = hdot(e12, e12);
        t[22] = len12;
        t[23] = 1.0f / dot00; t[24] = 1.0f / dot11; t[25] = 1.0f / len12;
        // pruning bounds (double arithmetic, rounded once).  Every one is a plane the whole triangle lies on one side of, so the
        // distance of a point to the triangle is at least its distance to that plane (on the far side): the triangle's own plane
        // (both sides: |n.p - o|) and, new in round 4, the three planes through its edges perpendicular to it (outward unit
        // normal n_e in the triangle's plane: n_e.q <= o_e for every q of the triangle, the offset rounded up by 1e-6).  t[26..27]
        // pad; t[28..31] plane; t[32..43] edges.  The wave-uniform estimate (de_cornell_box_table) keeps plane + bounding sphere
        // about the centroid, in the compact copy behind the rows.
        {
            const double a[3] = { e0.x, e0.y, e0.z }, b[3] = { e1.x, e1.y, e1.z };
            double n[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
            const double nl = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            for (int k = 0; k < 3; k++) n[k] /= nl;
            t[26] = t[27] = 0.0f;
            t[28] = (float)n[0]; t[29] = (float)n[1]; t[30] = (float)n[2];
            t[31] = (float)(n[0] * v0[0] + n[1] * v0[1] + n[2] * v0[2]);
            const float *vs[3] = { v0, v1, v2 };
            for (int j = 0; j < 3; j++) {
                const float *pa = vs[j], *pb = vs[(j + 1) % 3], *pc = vs[(j + 2) % 3];       // edge pa -> pb, pc opposite
                const double e[3] = { (double)pb[0] - pa[0], (double)pb[1] - pa[1], (double)pb[2] - pa[2] };
                double m[3] = { e[1] * n[2] - e[2] * n[1], e[2] * n[0] - e[0] * n[2], e[0] * n[1] - e[1] * n[0] };
                const double ml = sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
                for (int k = 0; k < 3; k++) m[k] /= ml;
                const double side = m[0] * ((double)pc[0] - pa[0]) + m[1] * ((double)pc[1] - pa[1]) + m[2] * ((double)pc[2] - pa[2]);
                if (side > 0.0) for (int k = 0; k < 3; k++) m[k] = -m[k];                      // away from the opposite vertex
                double o = -1e30;
                for (int q = 0; q < 3; q++) { const double d = m[0] * vs[q][0] + m[1] * vs[q][1] + m[2] * vs[q][2]; if (d > o) o = d; }
                float *eb = t + 32 + 4 * j;
                eb[0] = (float)m[0]; eb[1] = (float)m[1]; eb[2] = (float)m[2]; eb[3] = (float)(o + 1e-6);
            }
            // compact copy behind the rows (read four triangles at a time by de_cornell_box_table): plane, bounding sphere
            float *cb = tab + 32 * CORNELL_STRIDE + i * 8;
            for (int k = 0; k < 4; k++) cb[k] = t[28 + k];
            double c[3], R = 0.0;
            for (int k = 0; k < 3; k++) c[k] = ((double)v0[k] + v1[k] + v2[k]) / 3.0;
            for (int k = 0; k < 3; k++) cb[4 + k] = (float)c[k];
            for (int j = 0; j < 3; j++) {
                double d2 = 0.0;
                for (int k = 0; k < 3; k++) { const double d = (double)vs[j][k] - (double)cb[4 + k]; d2 += d * d; }
                if (sqrt(d2) > R) R = sqrt(d2);
            }
            cb[7] = (float)(R * (1.0 + 1e-6));
        }
    }
}

// Candidate grid of de_cornell_box_table (rmdf_device.hpp: cornell_cell_mask).  Per cell: the triangles t with
//   d(centre, t) <= min_s d(centre, s) + 2 * rho,   rho = half diagonal of the cell + margin,
// in double arithmetic on the float vertices.  |d(p, t) - d(centre, t)| <= |p - centre| <= rho for every p of the cell, so a triangle
// outside the mask is farther than the nearest one everywhere in the cell; the margin (1e-4, scene size ~1) covers the float rounding of
// the cell index at cell borders and of the reference's own distance formulas (1e-7).
static double point_triangle_distance(const double p[3], const double a[3], const double b[3], const double c[3])
{
    // closest point on triangle abc to p (Ericson, Real-Time Collision Detection 5.1.5)
    double ab[3], ac[3], ap[3];
    for (int k = 0; k < 3; k++) { ab[k] = b[k] - a[k]; ac[k] = c[k] - a[k]; ap[k] = p[k] - a[k]; }
    auto dot = [](const double *u, const double *v) { return u[0] * v[0] + u[1] * v[1] + u[2] * v[2]; };
    double q[3];
    const double d1 = dot(ab, ap), d2 = dot(ac, ap);
    double bp[3], cp[3];
    for (int k = 0; k < 3; k++) { bp[k] = p[k] - b[k]; cp[k] = p[k] - c[k]; }
    const double d3 = dot(ab, bp), d4 = dot(ac, bp), d5 = dot(ab, cp), d6 = dot(ac, cp);
    const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
    if (d1 <= 0.0 && d2 <= 0.0) { for (int k = 0; k < 3; k++) q[k] = a[k]; }
    else if (d3 >= 0.0 && d4 <= d3) { for (int k = 0; k < 3; k++) q[k] = b[k]; }
    else if (vc <= 0.0 && d1 >= 0.0 && d3 <= 0.0) { const double v = d1 / (d1 - d3); for (int k = 0; k < 3; k++) q[k] = a[k] + v * ab[k]; }
    else if (d6 >= 0.0 && d5 <= d6) { for (int k = 0; k < 3; k++) q[k] = c[k]; }
    else if (vb <= 0.0 && d2 >= 0.0 && d6 <= 0.0) { const double w = d2 / (d2 - d6); for (int k = 0; k < 3; k++) q[k] = a[k] + w * ac[k]; }
    else if (va <= 0.0 && (d4 - d3) >= 0.0 && (d5 - d6) >= 0.0) {
        const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        for (int k = 0; k < 3; k++) q[k] = b[k] + w * (c[k] - b[k]);
    } else {
        const double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den;
        for (int k = 0; k < 3; k++) q[k] = a[k] + ab[k] * v + ac[k] * w;
    }
    double s2 = 0.0;
    for (int k = 0; k < 3; k++) s2 += (p[k] - q[k]) * (p[k] - q[k]);
    return sqrt(s2);
}

// masks[cell] = the triangles that can be the nearest one for SOME point of the cell: d(centre, t) <= min_s d(centre, s) + 2 rho,
// rho = the cell's half diagonal + 1e-4 (float rounding of the cell index at cell borders and of the shader's formulas).  Double
// arithmetic.  `parent` (a grid of N / 2 cells per axis, or null): a child's candidates are a subset of its parent's (the centres
// are half a child diagonal apart and 2 rho_parent = 4 half-diagonals + 2e-4), and the triangle nearest to the child's centre is one
// of them, so only the parent's candidates are measured -- the 64^3 grid costs 0.1 s instead of a second.
void cornell_grid(const float tri[96 * 3], int N, uint32_t *masks /* N^3 */, const uint32_t *parent = nullptr)
{
    const double H = (double)CORNELL_GRID_H, cell = 2.0 * H / N, rho = 0.5 * cell * sqrt(3.0) + 1e-4;
    double v[32][3][3];
    for (int t = 0; t < 32; t++) for (int j = 0; j < 3; j++) for (int k = 0; k < 3; k++) v[t][j][k] = (double)tri[t * 9 + j * 3 + k];
    for (int iz = 0; iz < N; iz++) for (int iy = 0; iy < N; iy++) for (int ix = 0; ix < N; ix++) {
        const double c[3] = { -H + (ix + 0.5) * cell, -H + (iy + 0.5) * cell, -H + (iz + 0.5) * cell };
        const uint32_t pm = parent ? parent[((iz / 2) * (N / 2) + iy / 2) * (N / 2) + ix / 2] : 0xffffffffu;
        double d[32], dmin = 1e30;
        for (int t = 0; t < 32; t++) {
            if (!((pm >> t) & 1u)) continue;
            d[t] = point_triangle_distance(c, v[t][0], v[t][1], v[t][2]);
            if (d[t] < dmin) dmin = d[t];
        }
        uint32_t m = 0u;
        for (int t = 0; t < 32; t++) if (((pm >> t) & 1u) && d[t] <= dmin + 2.0 * rho) m |= 1u << t;
        masks[(iz * N + iy) * N + ix] = m;
    }
}

// the grid k_render reads from global memory (CORNELL_FINE_N^3 masks), reached from the CORNELL_GRID_N^3 one by halving cells; the
// geometry is a constant, so it is built once per process
const std::vector<uint32_t> &cornell_grids(const float tri[96 * 3])
{
    static std::vector<uint32_t> grids;
    static std::once_flag once;
    std::call_once(once, [&] {
        static_assert(CORNELL_FINE_N % CORNELL_GRID_N == 0 && ((CORNELL_FINE_N / CORNELL_GRID_N) & (CORNELL_FINE_N / CORNELL_GRID_N - 1)) == 0,
                      "the fine grid is reached from the coarse one by halving cells");
        const size_t nc = (size_t)CORNELL_GRID_N * CORNELL_GRID_N * CORNELL_GRID_N, nf = (size_t)CORNELL_FINE_N * CORNELL_FINE_N * CORNELL_FINE_N;
        std::vector<uint32_t> cur(nc), next;
        cornell_grid(tri, CORNELL_GRID_N, cur.data());
        for (int n = CORNELL_GRID_N * 2; n <= CORNELL_FINE_N; n *= 2) {
            next.resize((size_t)n * n * n);
            cornell_grid(tri, n, next.data(), cur.data());
            cur.swap(next);
        }
        grids.assign(cur.begin(), cur.begin() + nf);
    });
    return grids;
}

int ensure_frame(rmdf_ctx *ctx, int w, int h, bool planes)
{
    const size_t npx = (size_t)w * (size_t)h;
    if (!(npx <= ctx->cap_px && ctx->d_rgba8)) {
        if (ctx->d_rgba8) (void)dev_free(ctx->d_rgba8);
        ctx->d_rgba8 = nullptr; ctx->cap_px = 0; ctx->shadow_valid = false;
        HIP_TRY(ctx, dev_malloc((void **)&ctx->d_rgba8, npx * 4));
        ctx->cap_px = npx;
    }
    if (planes && !(npx <= ctx->cap_planes_px && ctx->d_rgba_f32)) {
        if (ctx->d_rgba_f32) (void)dev_free(ctx->d_rgba_f32);
        if (ctx->d_steps) (void)dev_free(ctx->d_steps);
        if (ctx->d_iters) (void)dev_free(ctx->d_iters);
        ctx->d_rgba_f32 = nullptr; ctx->d_steps = nullptr; ctx->d_iters = nullptr; ctx->cap_planes_px = 0;
        HIP_TRY(ctx, dev_malloc((void **)&ctx->d_rgba_f32, npx * 16));
        HIP_TRY(ctx, dev_malloc((void **)&ctx->d_steps, npx * 2));
        HIP_TRY(ctx, dev_malloc((void **)&ctx->d_iters, npx * 2));
        ctx->cap_planes_px = npx;
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_rgba_f32, 0, npx * 16, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_steps, 0, npx * 2, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_iters, 0, npx * 2, ctx->stream));
    }
    return RMDF_OK;
}

// resizeFrameBuffer clears the new texture to opaque black (FrameBuffer.hs:109-111)
int clear_frame(rmdf_ctx *ctx, int w, int h)
{
    const size_t npx = (size_t)w * (size_t)h;
    ctx->shadow_valid = false;
    HIP_TRY(ctx, launch_fill_u32(ctx->d_rgba8, 0xff000000u, npx, ctx->stream));
    if (ctx->d_rgba_f32 && npx <= ctx->cap_planes_px) {
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_rgba_f32, 0, npx * 16, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_steps, 0, npx * 2, ctx->stream));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_iters, 0, npx * 2, ctx->stream));
    }
    return RMDF_OK;
}

#ifdef RMDF_XCHECK
int ensure_gbuf(rmdf_ctx *ctx, int w, int h)
{
    const size_t gw = (size_t)((w + 1) & ~1), gh = (size_t)((h + 1) & ~1);
    const size_t need = gw * gh;
    if (!ctx->d_work_counter) HIP_TRY(ctx, dev_malloc((void **)&ctx->d_work_counter, 256));
    if (need <= ctx->gbuf_cap) return RMDF_OK;
    if (ctx->d_gbuf_nao) (void)dev_free(ctx->d_gbuf_nao);
    if (ctx->d_gbuf_meta) (void)dev_free(ctx->d_gbuf_meta);
    ctx->d_gbuf_nao = nullptr; ctx->d_gbuf_meta = nullptr; ctx->gbuf_cap = 0;
    HIP_TRY(ctx, dev_malloc((void **)&ctx->d_gbuf_nao, need * sizeof(float4)));
    HIP_TRY(ctx, dev_malloc((void **)&ctx->d_gbuf_meta, need * sizeof(unsigned)));
    ctx->gbuf_cap = need;
    return RMDF_OK;
}
#endif

// after_render (optional): queued on `stream` right behind the render kernel, before the sort of next frame's strip order -- what a caller
// waits for (an event, a copy of the rows) must not wait for that sort too
// order_bands > 0 (with p.band_strip_rows): dispatch order of a whole-frame host call that hands bands over as they complete --
// the costliest strips first, then band by band from the outside in (k_order_blocks)
int launch_scene(rmdf_ctx *ctx, int scene, const FrameParams &p, hipStream_t stream, const std::function<int()> *after_render = nullptr, int order_bands = 0)
{
#ifdef RMDF_XCHECK
    // librmdf_xcheck.so: the alternative schedule of the same per-ray arithmetic (cross-check tests, A/B measurements).
    // It keeps ONE G-buffer / work counter per ctx, so it runs on the ctx stream only.
    if (scene == RMDF_FS_MB_POWER8 && ctx->d_dbg && getenv("RMDF_NESTED_STATS")) {
        HIP_TRY(ctx, launch_march_stats(p, stream));
        if (after_render) RMDF_TRY((*after_render)());
        return RMDF_OK;
    }
    if (scene == RMDF_FS_MB_POWER8 && (ctx->flags & RMDF_FLAG_FLAT_MARCH)) {
        if (stream != ctx->stream)
            return fail(ctx, RMDF_E_UNSUPPORTED, "the alternative schedules keep one scratch set per ctx: use the ctx stream (stream = NULL)");
        FrameParams q = p;
        int rc = ensure_gbuf(ctx, p.w, p.h);
        if (rc != RMDF_OK) return rc;
        q.gbuf_nao = ctx->d_gbuf_nao; q.gbuf_meta = ctx->d_gbuf_meta; q.gw = (p.w + 1) & ~1;
        q.work_counter = ctx->d_work_counter;
        HIP_TRY(ctx, launch_render_mb8(q, stream, ctx->cus));
        if (after_render) RMDF_TRY((*after_render)());
        return RMDF_OK;
    }
#endif
    // The run time of a launch is set by the strips that hold the longest rays (a 226-step ray is a ~0.4 ms serial
    // chain), so large launches dispatch the strips that were most expensive in the previous frame of the same
    // configuration first (temporal coherence; `time` is deliberately not part of the key).  The table only permutes
    // which workgroup renders which strip: the image does not depend on it.
    FrameParams q = p;
    const int nblk = render_grid_blocks(p);
    const bool want = nblk >= 1024 && !(ctx->flags & RMDF_FLAG_RASTER_ORDER);
    if (want) {
        // the table set of this stream (the least recently used one is recycled when more than RMDF_MAX_ORDER_STREAMS render)
        OrderState *os = nullptr;
        for (auto &o : ctx->orders) if (o.used && o.stream == stream) { os = &o; break; }
        if (!os) {
            for (auto &o : ctx->orders) if (!o.used) { os = &o; break; }
            if (!os) {
                os = &ctx->orders[0];
                for (auto &o : ctx->orders) if (o.last_use < os->last_use) os = &o;
                // its last launches may still be running on the stream it served (which may no longer exist)
                HIP_TRY(ctx, hipDeviceSynchronize());
            }
            os->used = true; os->stream = stream; os->valid = false;
        }
        os->last_use = ++ctx->order_tick;
        if (nblk > os->cap) {
            if (os->cap) HIP_TRY(ctx, hipStreamSynchronize(stream));
            if (os->d_cost) (void)dev_free(os->d_cost);
            if (os->d_order) (void)dev_free(os->d_order);
            os->d_cost = os->d_order = nullptr; os->cap = 0; os->valid = false;
            HIP_TRY(ctx, dev_malloc((void **)&os->d_cost, (size_t)nblk * 4));
            HIP_TRY(ctx, dev_malloc((void **)&os->d_order, (size_t)nblk * 4));
            os->cap = nblk;
        }
#ifdef RMDF_XCHECK
        const int band_rows = p.band_flag ? p.band_strip_rows : 0;         // (one-launch band hand-over: librmdf_xcheck.so only)
#else
        const int band_rows = 0;
#endif
        const int key[12] = { scene, p.w, p.h, p.x0, p.y0, p.x1, p.y1, p.max_steps,
                              p.n_shard_tiles, p.shard_key, band_rows, order_bands };
        const bool same = os->valid && os->n == nblk && memcmp(key, os->key, sizeof key) == 0;
        q.block_cost = os->d_cost;
        q.block_order = same ? os->d_order : nullptr;
        HIP_TRY(ctx, launch_render(scene, q, stream));
        if (after_render) RMDF_TRY((*after_render)());
        if (order_bands > 0) {
            // (single-rectangle launches only: strip index = column + row * columns)
            const int gx = (((p.x1 + 1) & ~1) - (p.x0 & ~1) + 31) / 32;
            HIP_TRY(ctx, launch_order_blocks(os->d_cost, nblk, os->d_order, stream, gx, band_rows, order_bands));
        } else {
            HIP_TRY(ctx, launch_order_blocks(os->d_cost, nblk, os->d_order, stream));
        }
        memcpy(os->key, key, sizeof key);
        os->n = nblk; os->valid = true;
    } else {
        HIP_TRY(ctx, launch_render(scene, q, stream));
        if (after_render) RMDF_TRY((*after_render)());
    }
    return RMDF_OK;
}


// the deal in effect for `nranks` ranks, computed once per (nranks, set of costs)
void ensure_deal(rmdf_ctx *ctx, int nranks)
{
    if (ctx->deal_nranks == nranks && ctx->deal_gen == ctx->shard_cost_gen) return;
    for (int r = 0; r < nranks; r++) {
        ctx->deal_count[r] = shard_tiles_of_rank(r, nranks, ctx->deal_tiles[r], ctx->shard_cost_set ? ctx->shard_cost : nullptr,
                                                 ctx->shard_root_handicap);
        for (int s = 0; s < ctx->deal_count[r]; s++) ctx->deal_where.v[ctx->deal_tiles[r][s]] = (unsigned short)((r << 8) | s);
    }
    ctx->deal_nranks = nranks; ctx->deal_gen = ctx->shard_cost_gen;
}


int fill_params(rmdf_ctx *ctx, int scene, int w, int h, float time, int max_steps, FrameParams &p)
{
    if (scene < RMDF_FS_DE_CORNELL_BOX || scene > RMDF_FS_MB_GENERAL)
        return fail(ctx, RMDF_E_INVALID, "unknown FragmentShader value");
    // (primary_dir's short quotients are checked for exactly this range: rmdf_selftest_shading_math [4])
    if (w <= 0 || h <= 0 || w > RMDF_MAX_FRAME_SIDE || h > RMDF_MAX_FRAME_SIDE) return fail(ctx, RMDF_E_INVALID, "bad frame size");
    if (max_steps > 32767) return fail(ctx, RMDF_E_INVALID, "max_steps > 32767");
    for (int s = RMDF_ENV_REFLECTION; s <= RMDF_ENV_COS_8; s++)
        if (!ctx->env[s].d_texels)
            return fail(ctx, RMDF_E_NO_ENV, "environment cube map slot " + std::to_string(s) + " not set");
    memset(&p, 0, sizeof p);
    host_camera(scene, time, p.cam);
    p.fov_xs = host_fov_xs();
    {
        // fragment.shd:116-119: pow_offs = mod(in_time / 2, 9); if (pow_offs > 4.5) pow_offs = 9 - pow_offs; power = pow_offs + 2
        const float a = time / 2.0f;
        float pow_offs = a - 9.0f * floorf(a / 9.0f);
        if (pow_offs > 4.5f) pow_offs = 9.0f - pow_offs;
        p.power = pow_offs + 2.0f;
    }
    p.wf = (float)w; p.hf = (float)h; p.aspect = p.wf / p.hf;
    p.w = w; p.h = h;
    p.max_steps = max_steps <= 0 ? (int)shk::march_max_steps_default : max_steps;
    p.env_refl = CubeDev{ ctx->env[RMDF_ENV_REFLECTION].d_texels, ctx->env[RMDF_ENV_REFLECTION].W };
    p.env_cos1 = CubeDev{ ctx->env[RMDF_ENV_COS_1].d_texels, ctx->env[RMDF_ENV_COS_1].W };
    p.env_cos8 = CubeDev{ ctx->env[RMDF_ENV_COS_8].d_texels, ctx->env[RMDF_ENV_COS_8].W };
    p.cornell = ctx->d_cornell;
    // pooling the last rays of a workgroup pays for both Mandelbulbs (+5 %, +8 %) and the test scene (+8 %);
    // it costs 6 % for the Cornell box, whose distance estimate has the same cost for every ray
#ifdef RMDF_AB_MERGE_T              // (A/B builds: the pooling threshold, with the kernel's mailbox size of the same name in rmdf_render.hip)
    const int merge_t = RMDF_AB_MERGE_T;
#else
    const int merge_t = 32;
#endif
#ifdef RMDF_AB_CORNELL_MERGE
    p.merge_stragglers = (ctx->flags & RMDF_FLAG_NO_MERGE) ? 0 : (scene == RMDF_FS_DE_CORNELL_BOX ? RMDF_AB_CORNELL_MERGE : merge_t);
#else
    p.merge_stragglers = ((ctx->flags & RMDF_FLAG_NO_MERGE) || scene == RMDF_FS_DE_CORNELL_BOX) ? 0 : merge_t;
#endif
    p.cornell_tab = ctx->d_cornell_tab;
    p.cornell_grid = ctx->d_cornell_grid;
    p.cornell_prune = (ctx->flags & RMDF_FLAG_NO_PRUNE) ? 0 : 1;
#ifdef RMDF_XCHECK
    p.dbg = ctx->d_dbg;
    p.fold_min = (ctx->flags & RMDF_FLAG_FORCE_WRITTEN) ? __builtin_inff() : RMDF_MB8_FOLD_MIN;
#endif
    return RMDF_OK;
}

bool read_file(const char *path, std::vector<uint8_t> &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    // a regular file of a sane size: fopen() opens a directory too, and ftell() on one answers LONG_MAX (found by tools/asan_host.sh)
    struct stat sb;
    if (fstat(fileno(f), &sb) != 0 || !S_ISREG(sb.st_mode) || sb.st_size < 0 || sb.st_size > ((off_t)1 << 31)) { fclose(f); return false; }
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0 || n > ((long)1 << 31)) { fclose(f); return false; }
    out.resize((size_t)n);
    size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}

bool file_exists(const std::string &p)
{
    FILE *f = fopen(p.c_str(), "rb");
    if (!f) return false;
    fclose(f);
    return true;
}


// Radiance RGBE <-> float as JuicyPixels does it (third-party arithmetic behind
// JP.readImage / JP.saveRadianceImage, HDREnvMap.hs:33, ShaderRendering.hs:147)
void rgbe_to_float(const uint8_t *p, float *rgb)
{
    float f = ldexpf(1.0f, (int)p[3] - (128 + 8));
    rgb[0] = ((float)p[0] + 0.5f) * f;
    rgb[1] = ((float)p[1] + 0.5f) * f;
    rgb[2] = ((float)p[2] + 0.5f) * f;
}

void float_to_rgbe(const float *rgb, uint8_t *p)
{
    float d = rgb[0];
    if (rgb[1] > d) d = rgb[1];
    if (rgb[2] > d) d = rgb[2];
    if (!(d > 1e-32f)) { p[0] = p[1] = p[2] = p[3] = 0; return; }
    int e;
    float sig = frexpf(d, &e);
    float coeff = sig * 255.9999f / d;
    p[0] = (uint8_t)(int)(rgb[0] * coeff);
    p[1] = (uint8_t)(int)(rgb[1] * coeff);
    p[2] = (uint8_t)(int)(rgb[2] * coeff);
    p[3] = (uint8_t)(e + 128);
}


// loadHDRImage (HDREnvMap.hs:31-52): flat or new-style-RLE Radiance files
bool decode_hdr(const uint8_t *file, size_t len, int &w, int &h, std::vector<float> &rgb, std::string &why)
{
    size_t pos = 0;
    bool blank = false;
    while (pos < len) {                        // header lines up to the empty line
        size_t eol = pos;
        while (eol < len && file[eol] != '\n') eol++;
        if (eol >= len) { why = "truncated header"; return false; }
        bool empty = (eol == pos);
        pos = eol + 1;
        if (empty) { blank = true; break; }
    }
    if (!blank) { why = "no header terminator"; return false; }
    size_t eol = pos;
    while (eol < len && file[eol] != '\n') eol++;
    if (eol >= len || eol - pos > 100) { why = "no resolution line"; return false; }
    std::string line((const char *)&file[pos], eol - pos);
    if (sscanf(line.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0) { why = "unsupported resolution line '" + line + "'"; return false; }
    // the size comes from the file: bound it before it sizes an allocation (and by what the file can hold: >= 1 byte per 128 texels even fully run-length coded)
    if (w > 65536 || h > 32768 || (size_t)w * (size_t)h / 128 > len) { why = "implausible image size in resolution line '" + line + "'"; return false; }
    pos = eol + 1;
    rgb.resize((size_t)w * h * 3);
    std::vector<uint8_t> scan((size_t)w * 4);
    for (int y = 0; y < h; y++) {
        if (pos + 4 <= len && w >= 8 && w < 32768 && file[pos] == 2 && file[pos + 1] == 2 &&
            ((file[pos + 2] << 8) | file[pos + 3]) == w) {
            pos += 4;
            for (int ch = 0; ch < 4; ch++) {
                int x = 0;
                while (x < w) {
                    if (pos >= len) { why = "truncated RLE scanline"; return false; }
                    int cnt = file[pos++];
                    if (cnt > 128) {
                        cnt -= 128;
                        if (pos >= len || x + cnt > w) { why = "bad RLE run"; return false; }
                        uint8_t v = file[pos++];
                        for (int k = 0; k < cnt; k++) scan[(size_t)(x++) * 4 + ch] = v;
                    } else {
                        if (cnt == 0 || pos + cnt > len || x + cnt > w) { why = "bad RLE literal"; return false; }
                        for (int k = 0; k < cnt; k++) scan[(size_t)(x++) * 4 + ch] = file[pos++];
                    }
                }
            }
        } else {
            if (pos + (size_t)w * 4 > len) { why = "truncated pixel data"; return false; }
            memcpy(scan.data(), &file[pos], (size_t)w * 4);
            pos += (size_t)w * 4;
        }
        for (int x = 0; x < w; x++) rgbe_to_float(&scan[(size_t)x * 4], &rgb[((size_t)y * w + x) * 3]);
    }
    return true;
}

// JP.saveRadianceImage (ShaderRendering.hs:147) as a flat Radiance file image in memory
void encode_hdr(const std::vector<float> &rgb, int w, int h, std::vector<uint8_t> &file)
{
    char hdr[96];
    const int n = snprintf(hdr, sizeof hdr, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", h, w);
    file.resize((size_t)n + (size_t)w * h * 4);
    memcpy(file.data(), hdr, (size_t)n);
    for (size_t i = 0; i < (size_t)w * h; i++) float_to_rgbe(&rgb[i * 3], &file[(size_t)n + i * 4]);
}

// Several ranks of one node may build the same cache at once (bench.py, any multi-process host): write under a private
// name and rename() into place, so a reader sees either no file or a complete one.
bool write_file_atomic(const std::string &path, const std::vector<uint8_t> &data)
{
    static std::atomic<unsigned> serial{ 0 };                 // several renderers of one process may build the same cache
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid()) + "." + std::to_string(serial.fetch_add(1));
    FILE *f = fopen(tmp.c_str(), "wb");
    if (!f) return false;
    bool ok = fwrite(data.data(), 1, data.size(), f) == data.size();
    ok = (fclose(f) == 0) && ok;
    if (ok) ok = rename(tmp.c_str(), path.c_str()) == 0;
    if (!ok) remove(tmp.c_str());              // removeFile on failure, ShaderRendering.hs:146-148
    return ok;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)dev_free(p); }
};

const float kPi = 3.14159265358979323846f;       // `pi :: Float`

// GHC's class-default RealFloat atan2 (Float has no specialised one), atan = libm atanf
float hs_atan2f(float y, float x)
{
    if (x > 0.0f) return atanf(y / x);
    if (x == 0.0f && y > 0.0f) return kPi / 2.0f;
    if (x < 0.0f && y > 0.0f) return kPi + atanf(y / x);
    if ((x <= 0.0f && y < 0.0f) || (x < 0.0f && y == 0.0f && signbit(y)) || (x == 0.0f && signbit(x) && y == 0.0f && signbit(y)))
        return -hs_atan2f(-y, x);
    if (y == 0.0f && (x < 0.0f || (x == 0.0f && signbit(x)))) return kPi;
    if (x == 0.0f && y == 0.0f) return y;
    return x + y;
}

// Environment (u, v) of every texel of the six cw x cw faces: cubeMapPixelToDir (HDREnvMap.hs:76-87, Linear.normalize's
// near-unit short cut), worldToLocal (CoordTransf.hs:46-50), cartesianToSpherical (35-44), sphericalToEnvironmentUV
// (60-70).  A function of the face size alone, so it is evaluated once per size on the host -- with glibc's acosf / atanf,
// the very functions GHC's Float acos / atan call in the reference -- and k_latlong_to_cube only gathers.  Row segments on the ctx's
// host threads, as forSegmentsConcurrently does for the reference's resampler (HDREnvMap.hs:139).
void cube_uv_table_host(WorkPool &pool, int cw, std::vector<float> &uv)
{
    uv.resize((size_t)6 * cw * cw * 2);
    pool.segments(6 * cw, [&](int lo, int hi) {
        for (int r = lo; r < hi; r++) {
            const int face = r / cw, y = r % cw;
            for (int x = 0; x < cw; x++) {
                const float vw = ((float)x + 0.5f) / (float)cw * 2.0f - 1.0f;
                const float vh = ((float)y + 0.5f) / (float)cw * 2.0f - 1.0f;
                hv3 d;
                switch (face) {
                case 0:  d = hv3{ 1.0f, -vh, -vw }; break;
                case 1:  d = hv3{ -1.0f, -vh, vw }; break;
                case 2:  d = hv3{ vw, 1.0f, vh }; break;
                case 3:  d = hv3{ vw, -1.0f, -vh }; break;
                case 4:  d = hv3{ vw, -vh, 1.0f }; break;
                default: d = hv3{ -vw, -vh, -1.0f }; break;
                }
                // Linear.normalize: unchanged if the squared length is within 1e-6 of 0 or 1
                const float l = d.x * d.x + d.y * d.y + d.z * d.z;
                if (!(fabsf(l) <= 1e-6f || fabsf(1.0f - l) <= 1e-6f)) {
                    const float s = sqrtf(l);
                    d = hv3{ d.x / s, d.y / s, d.z / s };
                }
                const hv3 loc{ (d.x * 1.0f + d.y * 0.0f) + d.z * 0.0f,
                               (d.x * 0.0f + d.y * 0.0f) + d.z * -1.0f,
                               (d.x * 0.0f + d.y * 1.0f) + d.z * 0.0f };
                float cz = loc.z;
                if (cz > 1.0f) cz = 1.0f;
                if (cz < -1.0f) cz = -1.0f;
                const float theta = acosf(cz);
                const float p2 = hs_atan2f(loc.y, loc.x);
                const float p1 = (p2 < 0.0f) ? p2 + 2.0f * kPi : p2;
                const float phi = (p1 == 2.0f * kPi) ? 0.0f : p1;
                const float q1 = phi + kPi / 2.0f;
                const float q2 = (q1 > 2.0f * kPi) ? q1 - 2.0f * kPi : q1;
                const float q3 = 2.0f * kPi - q2;
                float *o = &uv[((size_t)r * cw + x) * 2];
                o[0] = q3 / (kPi * 2.0f);
                o[1] = theta / kPi;
            }
        }
    });
}

int get_uv_table(rmdf_ctx *ctx, int cw, const float2 **d_uv)
{
    for (auto &t : ctx->uv_tables) if (t.cw == cw) { *d_uv = t.d_uv; return RMDF_OK; }
    std::vector<float> uv;
    cube_uv_table_host(ctx_pool(ctx), cw, uv);
    UvTable t;
    t.cw = cw;
    HIP_TRY(ctx, dev_malloc((void **)&t.d_uv, uv.size() * sizeof(float)));
    const int urc = upload(ctx, t.d_uv, uv.data(), uv.size() * sizeof(float), ctx->stream);     // every user of the table runs on the ctx stream or after an event of it
    if (urc != RMDF_OK) { (void)dev_free(t.d_uv); return urc; }
    if (ctx->uv_tables.size() >= 8) { (void)hipDeviceSynchronize(); (void)dev_free(ctx->uv_tables[0].d_uv); ctx->uv_tables.erase(ctx->uv_tables.begin()); }
    ctx->uv_tables.push_back(t);
    *d_uv = t.d_uv;
    return RMDF_OK;
}

// Cosine tables of cosineConvolveHDREnvMap (HDREnvMap.hs:222-239) for a w x h map, host glibc cosf / sinf:
//   lutT[(blk*w + x)*64 + lane] = cos |pxToPhi(blk*64+lane) - pxToPhi(x)|   (absPhiDiffCosLookup, per destination column)
//   tcs[2y], tcs[2y+1]          = cos, sin of pxToTheta(y)
void lobe_tables_host(WorkPool &pool, int w, int h, std::vector<float> &lutT, std::vector<float> &tcs)
{
    const int nblk = (w + 63) / 64;
    lutT.resize((size_t)nblk * w * 64);
    pool.segments(nblk * w, [&](int lo, int hi) {
        for (int r = lo; r < hi; r++) {
            const int blk = r / w, x = r % w;
            const float phi_x = (float)x / (float)(w - 1) * 2.0f * kPi;
            for (int lane = 0; lane < 64; lane++) {
                int dx = blk * 64 + lane;
                if (dx > w - 1) dx = w - 1;
                const float phi_l = (float)dx / (float)(w - 1) * 2.0f * kPi;
                lutT[(size_t)r * 64 + lane] = cosf(fabsf(phi_l - phi_x));
            }
        }
    });
    tcs.resize((size_t)h * 2);
    for (int y = 0; y < h; y++) {
        const float th = (float)y / (float)(h - 1) * kPi;
        tcs[2 * y] = cosf(th); tcs[2 * y + 1] = sinf(th);
    }
}

int get_lobe_tables(rmdf_ctx *ctx, int w, int h, const float **d_lutT, const float2 **d_tcs)
{
    for (auto &t : ctx->lobe_tables) if (t.w == w && t.h == h) { *d_lutT = t.d_lutT; *d_tcs = t.d_tcs; return RMDF_OK; }
    std::vector<float> lutT, tcs;
    lobe_tables_host(ctx_pool(ctx), w, h, lutT, tcs);
    LobeTable t;
    t.w = w; t.h = h;
    HIP_TRY(ctx, dev_malloc((void **)&t.d_lutT, lutT.size() * sizeof(float)));
    hipError_t e = dev_malloc((void **)&t.d_tcs, tcs.size() * sizeof(float));
    int urc = e == hipSuccess ? RMDF_OK : fail(ctx, RMDF_E_HIP, std::string("lobe table: ") + hipGetErrorString(e));
    // (a caller's stream may launch the prefilter, rmdf_prefilter_env_device: the tables are complete before this returns)
    if (urc == RMDF_OK) urc = upload(ctx, t.d_lutT, lutT.data(), lutT.size() * sizeof(float), ctx->stream);
    if (urc == RMDF_OK) urc = upload(ctx, t.d_tcs, tcs.data(), tcs.size() * sizeof(float), ctx->stream);
    if (urc == RMDF_OK && (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) urc = fail(ctx, RMDF_E_HIP, std::string("lobe table upload: ") + hipGetErrorString(e));
    if (urc != RMDF_OK) {
        (void)dev_free(t.d_lutT); if (t.d_tcs) (void)dev_free(t.d_tcs);
        return urc;
    }
    if (ctx->lobe_tables.size() >= 4) {
        (void)hipDeviceSynchronize();
        (void)dev_free(ctx->lobe_tables[0].d_lutT); (void)dev_free(ctx->lobe_tables[0].d_tcs);
        ctx->lobe_tables.erase(ctx->lobe_tables.begin());
    }
    ctx->lobe_tables.push_back(t);
    *d_lutT = t.d_lutT; *d_tcs = t.d_tcs;
    return RMDF_OK;
}

int set_env_from_device_faces(rmdf_ctx *ctx, int slot, const float *d_faces, int W)
{
    uint2 *d_padded = nullptr;
    size_t n = (size_t)6 * (W + 2) * (W + 2);
    HIP_TRY(ctx, dev_malloc((void **)&d_padded, n * sizeof(uint2)));
    hipError_t e = launch_cube_upload(d_faces, W, d_padded, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)dev_free(d_padded); return fail(ctx, RMDF_E_HIP, std::string("cube upload: ") + hipGetErrorString(e)); }
    if (ctx->env[slot].d_texels) { (void)hipDeviceSynchronize(); (void)dev_free(ctx->env[slot].d_texels); }
    ctx->env[slot].d_texels = d_padded;
    ctx->env[slot].W = W;
    ctx->env_gen++;                 // tile jobs rendered ahead of their calls saw the old map: they are not for this environment
    return RMDF_OK;
}

// the lobe prefilter of `n` powers of one device-resident map, concurrently on the ctx's four power streams
int prefilter_powers_device(rmdf_ctx *ctx, const float *d_src, int w, int h, const float *powers, int n, float *const *d_out)
{
    const float *d_lutT = nullptr; const float2 *d_tcs = nullptr;
    int rc = get_lobe_tables(ctx, w, h, &d_lutT, &d_tcs);
    if (rc != RMDF_OK) return rc;
    // the reference's own job -- powers 1, 8, 64 and 512 of one map -- is one launch (rmdf_env.hip: k_prefilter_fused4)
#ifdef RMDF_XCHECK
    static const bool no_fused = getenv("RMDF_PREFILTER_NO_FUSED") != nullptr;       // A/B switch of the cross-check build (tools/)
#else
    constexpr bool no_fused = false;
#endif
    // (two or three of the four too: 0.9 ms for the launch against 1.1 / 1.8 for two / three split launches; the other sums are computed
    // and dropped)
    if (n >= 2 && n <= 4 && w <= 256 && w % 4 == 0 && !no_fused) {
        float *by_k[4] = { nullptr, nullptr, nullptr, nullptr };
        bool ok = true;
        for (int i = 0; i < n; i++) {
            const int l2 = prefilter_log2p(powers[i]);
            if (l2 < 0 || by_k[l2 / 3]) { ok = false; break; }
            by_k[l2 / 3] = d_out[i];
        }
        if (ok) {
            HIP_TRY(ctx, launch_prefilter_fused4(d_src, w, h, d_lutT, d_tcs, by_k, ctx->stream));
            return RMDF_OK;
        }
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    const int ns = n < 4 ? n : 4;
    for (int k = 0; k < ns; k++) HIP_TRY(ctx, hipStreamWaitEvent(ctx->pstream[k], ctx->ev_fork, 0));
    for (int i = 0; i < n; i++)
        HIP_TRY(ctx, launch_prefilter(d_src, w, h, powers[i], d_lutT, d_tcs, d_out[i], ctx->pstream[i % 4], n <= 3));
    for (int k = 0; k < ns; k++) {
        HIP_TRY(ctx, hipEventRecord(ctx->ev_join[k], ctx->pstream[k]));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_join[k], 0));
    }
    return RMDF_OK;
}


// ---- the exchange step: RCCL over xGMI, one process per GPU ----------------------------------------------------------------
// librccl is opened on first use, by soname: a single-GPU host never needs it, and a process that already carries an RCCL
// (PyTorch ships one with the same soname) gets that copy instead of a second one.
struct RcclApi {
    void *handle = nullptr;
    decltype(&ncclGetUniqueId)   GetUniqueId = nullptr;
    decltype(&ncclCommInitRank)  CommInitRank = nullptr;
    decltype(&ncclCommDestroy)   CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclSend)          Send = nullptr;
    decltype(&ncclRecv)          Recv = nullptr;
    decltype(&ncclGroupStart)    GroupStart = nullptr;
    decltype(&ncclGroupEnd)      GroupEnd = nullptr;
};
RcclApi    g_rccl;
std::mutex g_rccl_mutex;

int load_rccl(rmdf_ctx *ctx)
{
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (g_rccl.handle) return RMDF_OK;
    void *h = nullptr;
#ifdef RMDF_XCHECK
    // the cross-check build only: a test double of the eight entry points below (tests/fake_rccl.c), so that the N > 1 branches of the
    // exchange run with N processes on ONE GPU.  librmdf.so never looks at the variable.
    if (const char *over = getenv("RMDF_RCCL_LIB")) {
        h = dlopen(over, RTLD_NOW | RTLD_LOCAL);
        if (!h) return fail(ctx, RMDF_E_UNSUPPORTED, std::string("RMDF_RCCL_LIB: cannot load ") + over + ": " + dlerror());
    }
#endif
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return fail(ctx, RMDF_E_UNSUPPORTED, std::string("cannot load librccl.so.1: ") + dlerror());
    RcclApi a;
    a.handle = h;
#define RMDF_SYM(field, name) a.field = (decltype(a.field))dlsym(h, name); if (!a.field) { dlclose(h); return fail(ctx, RMDF_E_UNSUPPORTED, std::string("librccl lacks ") + name); }
    RMDF_SYM(GetUniqueId, "ncclGetUniqueId") RMDF_SYM(CommInitRank, "ncclCommInitRank") RMDF_SYM(CommDestroy, "ncclCommDestroy")
    RMDF_SYM(GetErrorString, "ncclGetErrorString") RMDF_SYM(Send, "ncclSend") RMDF_SYM(Recv, "ncclRecv")
    RMDF_SYM(GroupStart, "ncclGroupStart") RMDF_SYM(GroupEnd, "ncclGroupEnd")
#undef RMDF_SYM
    g_rccl = a;
    return RMDF_OK;
}

#define RCCL_TRY(ctx, expr)                                                                               \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess)                                                                            \
            return fail(ctx, RMDF_E_COMM, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));         \
    } while (0)

// between ncclGroupStart and ncclGroupEnd: a failing call still closes the group before the error is returned (an open group would
// swallow every later call on this thread)
#define RCCL_GROUP_TRY(ctx, expr)                                                                         \
    do {                                                                                                  \
        ncclResult_t r_ = (expr);                                                                         \
        if (r_ != ncclSuccess) {                                                                          \
            (void)g_rccl.GroupEnd();                                                                      \
            return fail(ctx, RMDF_E_COMM, std::string(#expr) + ": " + g_rccl.GetErrorString(r_));         \
        }                                                                                                 \
    } while (0)

// one tile job: render tile `idx` of the latched frame into job buffer `b` on the job's stream, mirrored to the host tile.  The scratch
// tile keeps the FRAME's row pitch (rows y0 .. y1 of the frame, the tile's columns first): the kernel's rectangle form then needs
// nothing but a base pointer moved back by the rectangle's origin, and any frame size works -- the packed shard form of the first
// version needed sides 8 divides.
int issue_tile_job(rmdf_ctx *ctx, int b, int scene, int idx)
{
    rmdf_ctx::TileJob &j = ctx->tile_job[b];
    int x0, y0, x1, y1;
    tile_rect_host(idx, ctx->w, ctx->h, &x0, &y0, &x1, &y1);
    if (x1 <= x0 || y1 <= y0) { j.issued = false; return RMDF_OK; }                 // a tile without pixels (frames smaller than 8 x 8)
    const size_t tpx = (size_t)((ctx->h + 7) / 8 + 1) * (size_t)ctx->w;                // rows of the tallest tile, at the frame's pitch
    if (!j.stream) HIP_TRY(ctx, hipStreamCreateWithFlags(&j.stream, hipStreamNonBlocking));
    if (!j.done) HIP_TRY(ctx, hipEventCreateWithFlags(&j.done, hipEventDisableTiming));
    if (!j.copied) HIP_TRY(ctx, hipEventCreateWithFlags(&j.copied, hipEventDisableTiming));
    if (j.px < tpx) {
        HIP_TRY(ctx, hipStreamSynchronize(j.stream));
        if (j.d_tile) (void)dev_free(j.d_tile);
        if (j.h_tile) (void)hipHostFree(j.h_tile);
        j.d_tile = j.h_tile = j.h_tile_dev = nullptr; j.px = 0;
        HIP_TRY(ctx, dev_malloc((void **)&j.d_tile, tpx * 4));
        HIP_TRY(ctx, hipHostMalloc((void **)&j.h_tile, tpx * 4, hipHostMallocMapped));
        HIP_TRY(ctx, hipHostGetDevicePointer((void **)&j.h_tile_dev, j.h_tile, 0));
        j.px = tpx;
    }
    j.issued = false;
    FrameParams q;
    int rc = fill_params(ctx, scene, ctx->w, ctx->h, ctx->time, ctx->max_steps, q);
    if (rc != RMDF_OK) return rc;
    q.x0 = x0; q.y0 = y0; q.x1 = x1; q.y1 = y1;
    // pixel (qx, qy) of the frame goes to base[qx + qy * w]: with the base moved back by the rectangle's origin that is
    // tile[(qx - x0) + (qy - y0) * w]; only addresses inside the scratch tile are ever formed into accesses
    const ptrdiff_t origin = (ptrdiff_t)y0 * ctx->w + x0;
    q.rgba8 = j.d_tile - origin;
    q.rgba8_mirror = j.h_tile_dev - origin;
    rc = launch_scene(ctx, scene, q, j.stream);
    if (rc != RMDF_OK) return rc;
    HIP_TRY(ctx, hipEventRecord(j.done, j.stream));
    j.issued = true; j.scene = scene; j.idx = idx % 64; j.w = ctx->w; j.h = ctx->h; j.max_steps = ctx->max_steps; j.time = ctx->time;
    j.env_gen = ctx->env_gen;
    return RMDF_OK;
}

// the page-locked host copy of the accumulating frame (tile mode, whole-frame calls into pageable memory)
int ensure_shadow(rmdf_ctx *ctx, size_t npx)
{
    if (ctx->shadow_px == npx && ctx->h_shadow) return RMDF_OK;
    if (ctx->h_shadow) {
        HIP_TRY(ctx, hipDeviceSynchronize());            // a band's copy or mirror store of an earlier, failed call may still target it
        (void)hipHostFree(ctx->h_shadow); ctx->h_shadow = nullptr; ctx->h_shadow_dev = nullptr; ctx->shadow_px = 0;
    }
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_shadow, npx * 4, hipHostMallocMapped));
    HIP_TRY(ctx, hipHostGetDevicePointer((void **)&ctx->h_shadow_dev, ctx->h_shadow, 0));
    ctx->shadow_px = npx; ctx->shadow_valid = false;
    return RMDF_OK;
}

int render_tile_fast(rmdf_ctx *ctx, int scene, int tile_idx, const FrameParams &p, uint32_t *out_rgba8)
{
    const size_t npx = (size_t)ctx->w * ctx->h;
    const int midx = tile_idx % 64, b = midx % RMDF_TILE_JOBS;
    RMDF_TRY(ensure_shadow(ctx, npx));
    WorkPool &pool = ctx_pool(ctx, npx * 4);
    PoolJobGuard job_guard(pool);                       // whatever leaves this function -- an error return between begin() and finish(), an exception
                                                        // from a string or std::function allocation -- the frame copy is joined first
    // this call's job: the one issued speculatively by the previous call if it is for exactly this tile of this frame, else now
    rmdf_ctx::TileJob &j = ctx->tile_job[b];
    auto is_for = [&](const rmdf_ctx::TileJob &t, int idx) {
        return t.issued && t.scene == scene && t.idx == idx && t.w == ctx->w && t.h == ctx->h && t.max_steps == ctx->max_steps &&
               t.env_gen == ctx->env_gen && memcmp(&t.time, &ctx->time, sizeof(float)) == 0;
    };
    if (!is_for(j, midx)) { int rc = issue_tile_job(ctx, b, scene, midx); if (rc != RMDF_OK) return rc; }
    char *sh = (char *)ctx->h_shadow, *dst = (char *)out_rgba8;
    if (!ctx->shadow_valid) {
        // first tile call after a whole-frame render, a clear or a resize: the shadow is stale everywhere
        HIP_TRY(ctx, hipMemcpyAsync(sh, ctx->d_rgba8, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        ctx->shadow_valid = true;
    }
    // the frame as it was goes to the caller on the worker threads ...
    {
        const int parts = npx * 4 >= ((size_t)1 << 20) ? pool.workers() + 1 : 1;           // small frames: not worth a wake-up
        const size_t bytes = npx * 4;
        pool.begin(parts, [=](int part) {
            size_t lo, hi;
            WorkPool::slice(bytes, part, parts, 4096, lo, hi);
            if (hi > lo) memcpy(dst + lo, sh + lo, hi - lo);
        });
    }
    // ... while this thread issues the next tiles of the same frame ahead of their calls (tile 63 is followed by a frame with another
    // time: nothing to guess), joins the copy, and waits for its own tile
    // (a job that cannot be issued ahead is issued -- and fails, with its message -- by its own call: speculation leaves ctx->err alone)
    for (int t = midx + 1; t < midx + RMDF_TILE_JOBS && t < 64; t++)
        if (!is_for(ctx->tile_job[t % RMDF_TILE_JOBS], t)) {
            std::string keep;
            keep.swap(ctx->err);
            if (issue_tile_job(ctx, t % RMDF_TILE_JOBS, scene, t) != RMDF_OK) ctx->spec_dropped++;
            ctx->err.swap(keep);
        }
    pool.finish();
    HIP_TRY(ctx, hipEventSynchronize(j.done));
    j.issued = false;
    // ... then the tile: host tile -> shadow and caller (rows of the tile's width), scratch tile -> device frame behind everything
    // queued on the ctx stream so far
    const int tw = p.x1 - p.x0, th = p.y1 - p.y0;
    for (int y = 0; y < th; y++) {
        const size_t off = ((size_t)(p.y0 + y) * ctx->w + p.x0) * 4;
        memcpy(sh + off, j.h_tile + (size_t)y * ctx->w, (size_t)tw * 4);
        memcpy(dst + off, j.h_tile + (size_t)y * ctx->w, (size_t)tw * 4);
    }
    HIP_TRY(ctx, hipMemcpy2DAsync(ctx->d_rgba8 + (size_t)p.y0 * ctx->w + p.x0, (size_t)ctx->w * 4, j.d_tile, (size_t)ctx->w * 4,
                                  (size_t)tw * 4, (size_t)th, hipMemcpyDeviceToDevice, ctx->stream));
    // the scratch tile may be overwritten by its next job only once this copy has read it
    HIP_TRY(ctx, hipEventRecord(j.copied, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(j.stream, j.copied, 0));
    return RMDF_OK;
}

#ifdef RMDF_XCHECK
// (librmdf_xcheck.so only: never run on hardware; librmdf.so's rmdf_create answers reserved[3] >= 2 with RMDF_E_UNSUPPORTED)
// The same hand-over with ONE launch (rmdf_config.reserved[3] = 2, 3): the kernel stores the frame's rows into the page-locked shadow itself
// (mirror stores) and, band by band, tells the host when all rows of a band have landed (FrameParams::band_flag); the host threads copy
// each band to the caller while the launch is still running.  No per-band launches, copies or events: a band costs the host nothing
// until it is complete.  reserved[3] = 3 also dispatches the strips band by band behind the costliest ones, outer bands first, so
// that the bands do complete one after the other (plain longest-first order finishes all of them at the end).
int render_whole_frame_one_launch(rmdf_ctx *ctx, int scene, const FrameParams &p, uint32_t *out_rgba8, int nb, int mode)
{
    const int w = ctx->w, h = ctx->h;
    WorkPool &pool = ctx_pool(ctx);
    const int strip_rows = (h + 7) / 8;
    int bsr = (strip_rows + nb - 1) / nb;                            // rows of strips per band
    if (bsr < 1) bsr = 1;
    nb = (strip_rows + bsr - 1) / bsr;
    if (!ctx->wf_flags) {
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->wf_flags, RMDF_WF_MAX_BANDS * sizeof(unsigned), hipHostMallocMapped));
        memset((void *)ctx->wf_flags, 0, RMDF_WF_MAX_BANDS * sizeof(unsigned));
        HIP_TRY(ctx, hipHostGetDevicePointer((void **)&ctx->wf_flags_dev, (void *)ctx->wf_flags, 0));
        HIP_TRY(ctx, dev_malloc((void **)&ctx->d_wf_count, RMDF_WF_MAX_BANDS * sizeof(unsigned)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->d_wf_count, 0, RMDF_WF_MAX_BANDS * sizeof(unsigned), ctx->stream));
    }
    const unsigned seq = ++ctx->wf_seq ? ctx->wf_seq : ++ctx->wf_seq;      // never 0: the flags start there
    FrameParams q = p;
    q.x0 = 0; q.x1 = w; q.y0 = 0; q.y1 = h;
    q.rgba8 = ctx->d_rgba8;
    q.rgba8_mirror = ctx->h_shadow_dev;
    q.band_count = ctx->d_wf_count; q.band_flag = ctx->wf_flags_dev; q.band_seq = seq; q.band_strip_rows = bsr;
    int rc = launch_scene(ctx, scene, q, ctx->stream, nullptr, mode >= 3 ? nb : 0);
    if (rc != RMDF_OK) { (void)hipDeviceSynchronize(); ctx->shadow_valid = false; return rc; }
    char *sh = (char *)ctx->h_shadow, *dst = (char *)out_rgba8;
    unsigned pending = nb >= 32 ? 0xffffffffu : ((1u << nb) - 1u);
    unsigned spins = 0;
    PoolPrimeGuard hot(pool, 2000);                                    // the copy threads spin from now until the last band is through (or any other way out)
    while (pending) {
        for (int k = 0; k < nb; k++) {
            if (!((pending >> k) & 1u)) continue;
            if (__atomic_load_n(&ctx->wf_flags[k], __ATOMIC_ACQUIRE) != seq) continue;
            const int y0 = k * bsr * 8, y1 = (k + 1) * bsr * 8 < h ? (k + 1) * bsr * 8 : h;
            const size_t off = (size_t)y0 * w * 4, bytes = (size_t)(y1 - y0) * w * 4;
            pool.copy(dst + off, sh + off, bytes);
            pending &= ~(1u << k);
        }
        if (pending) {
            __builtin_ia32_pause();
            // a launch that died leaves its flags unset: look at the stream now and then instead of spinning for ever
            if ((++spins & 0xffffu) == 0u) {
                const hipError_t e = hipStreamQuery(ctx->stream);
                if (e != hipSuccess && e != hipErrorNotReady) { ctx->shadow_valid = false; return fail(ctx, RMDF_E_HIP, std::string("whole-frame launch: ") + hipGetErrorString(e)); }
                if (e == hipSuccess) {
                    // the stream is idle and a flag is still missing: cannot happen with a healthy launch -- finish from the device frame
                    bool all = true;
                    for (int k = 0; k < nb; k++) if (((pending >> k) & 1u) && __atomic_load_n(&ctx->wf_flags[k], __ATOMIC_ACQUIRE) != seq) all = false;
                    if (!all) {
                        ctx->shadow_valid = false;
                        HIP_TRY(ctx, hipMemsetAsync(ctx->d_wf_count, 0, RMDF_WF_MAX_BANDS * sizeof(unsigned), ctx->stream));
                        return download(ctx, out_rgba8, ctx->d_rgba8, (size_t)w * h * 4, ctx->stream);
                    }
                }
            }
        }
    }
    ctx->shadow_valid = true;
    return RMDF_OK;
}
#endif

// Whole frame into PAGEABLE caller memory: `rmdf_render_tile(tile_idx = -1, ptr)`, the call the reference's viewer makes every frame
// (Main.hs:67 starts it with tiling off; App.hs:154-166 -> fillFrameBuffer, FrameBuffer.hs:117-158).  Launch, copy 8.3 MB, return
// costs 0.59 ms at 1080p against 0.38 ms for the kernel: the copy (and the runtime's page-locking of the caller's pages for it) stands
// behind the kernel.  Here the frame is rendered as ROW BANDS on streams of their own, all in flight together; each band's rows go into
// the page-locked shadow frame as soon as ITS kernel is over (an SDMA copy queued behind it -- or the kernel's own mirror store,
// wf_mirror), and from there into the caller's frame by the ctx's host threads while the other bands are still rendering.  What stays
// exposed is the last band's copy.  The bands are cut at multiples of eight rows (whole strips, whole GL quads: same pixels as one
// launch -- tests/test_gpu_parity.py), launched middle first: the scenes sit in the middle of the frame, where the long rays are.
int render_whole_frame_host(rmdf_ctx *ctx, int scene, const FrameParams &p, uint32_t *out_rgba8)
{
    const int w = ctx->w, h = ctx->h;
    const size_t npx = (size_t)w * h;
    RMDF_TRY(ensure_shadow(ctx, npx));
    WorkPool &pool = ctx_pool(ctx, npx * 4);
    // the library's choice (neither knob given): RMDF_WF_DEFAULT_* -- measured on the headline frame and the Cornell box, tools/whole_frame_sweep.py
    const int mode = (ctx->wf_bands == 0 && ctx->wf_mirror == 0) ? RMDF_WF_DEFAULT_MODE : ctx->wf_mirror;
    // (the library's choice: two bands from 6 MB of frame on -- 1920x1080: 0.516 ms against 0.552 with one; the 3.7 MB Cornell frame of
    // config 2 is faster in one piece, 0.249 against 0.273 ms)
    int nb = ctx->wf_bands > 0 ? ctx->wf_bands : (npx * 4 >= ((size_t)6 << 20) ? RMDF_WF_DEFAULT_BANDS : 1);
    if (npx * 4 < ((size_t)2 << 20)) nb = 1;                         // small frames: one launch, one copy
    if (nb > h / 64) nb = h / 64 > 0 ? h / 64 : 1;                   // a band is at least 64 rows
    const bool mirror = mode != 0;
#ifdef RMDF_XCHECK
    if (mode >= 2) return render_whole_frame_one_launch(ctx, scene, p, out_rgba8, nb, mode);
#endif
    int ys[RMDF_WF_MAX_BANDS + 1];
    for (int k = 0; k <= nb; k++) ys[k] = k == nb ? h : (int)(((long long)h * k / nb) & ~7ll);
    for (int k = 0; k < nb; k++) {
        if (!ctx->wf_stream[k]) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->wf_stream[k], hipStreamNonBlocking));
        if (!ctx->wf_done[k]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->wf_done[k], hipEventDisableTiming));
    }
    if (!ctx->wf_fork) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->wf_fork, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventRecord(ctx->wf_fork, ctx->stream));          // behind whatever the ctx stream still holds (a clear, an env upload)
    char *sh = (char *)ctx->h_shadow, *dst = (char *)out_rgba8;
    int order[RMDF_WF_MAX_BANDS];
    for (int lo = (nb - 1) / 2, hi = lo + 1, n = 0; n < nb;) {        // middle first, then outwards
        if (lo >= 0) order[n++] = lo--;
        if (hi < nb && n < nb) order[n++] = hi++;
    }
    int rc = RMDF_OK;
    for (int i = 0; i < nb && rc == RMDF_OK; i++) {
        const int k = order[i];
        hipStream_t st = ctx->wf_stream[k];
        FrameParams q = p;
        q.x0 = 0; q.x1 = w; q.y0 = ys[k]; q.y1 = ys[k + 1];
        q.rgba8 = ctx->d_rgba8;
        if (mirror) q.rgba8_mirror = ctx->h_shadow_dev;
        const size_t off = (size_t)ys[k] * w * 4, bytes = (size_t)(ys[k + 1] - ys[k]) * w * 4;
        const std::function<int()> after = [&]() -> int {
            if (!mirror) HIP_TRY(ctx, hipMemcpyAsync(sh + off, (const char *)ctx->d_rgba8 + off, bytes, hipMemcpyDeviceToHost, st));
            HIP_TRY(ctx, hipEventRecord(ctx->wf_done[k], st));
            return RMDF_OK;
        };
        hipError_t e = hipStreamWaitEvent(st, ctx->wf_fork, 0);
        if (e != hipSuccess) { rc = fail(ctx, RMDF_E_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(e)); break; }
        rc = launch_scene(ctx, scene, q, st, &after);
    }
    if (rc != RMDF_OK) { (void)hipDeviceSynchronize(); ctx->shadow_valid = false; return rc; }   // bands already issued still write frame and shadow
    // the bands to the caller as they land
    unsigned pending = nb >= 32 ? 0xffffffffu : ((1u << nb) - 1u);
    // the copy threads spin while the bands land: twice the previous frame's measured time, within [200 us, 2 ms]; the guard puts
    // them back to sleep on every way out of this function
    const auto t_wait0 = std::chrono::steady_clock::now();
    PoolPrimeGuard hot(pool, ctx->wf_last_us > 0 ? std::min<long long>(2000, std::max<long long>(200, 2 * ctx->wf_last_us)) : 2000);
    while (pending) {
        for (int k = 0; k < nb; k++) {
            if (!((pending >> k) & 1u)) continue;
            const hipError_t e = hipEventQuery(ctx->wf_done[k]);
            if (e == hipErrorNotReady) continue;
            if (e != hipSuccess) { (void)hipDeviceSynchronize(); ctx->shadow_valid = false; return fail(ctx, RMDF_E_HIP, std::string("whole-frame band: ") + hipGetErrorString(e)); }
            const size_t off = (size_t)ys[k] * w * 4, bytes = (size_t)(ys[k + 1] - ys[k]) * w * 4;
            pool.copy(dst + off, sh + off, bytes);
            pending &= ~(1u << k);
        }
        if (pending) __builtin_ia32_pause();
    }
    ctx->wf_last_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_wait0).count();
    ctx->shadow_valid = true;                                          // (every band is over: the device frame is complete as well)
    return RMDF_OK;
}

int render_common(rmdf_ctx *ctx, int scene, int tile_idx, int w, int h, double time, int max_steps,
                  uint32_t *out_rgba8, float *out_rgba_f32, uint16_t *out_steps, uint16_t *out_iters)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool whole = tile_idx < 0;
    const bool first = whole || rmdf_is_tile_idx_first_tile(tile_idx);
    const bool planes = out_rgba_f32 || out_steps || out_iters;
    // latch on the first tile (ShaderRendering.hs:162-176).  A size change in the middle
    // of a tiled frame re-latches (the reference would resize its FBO texture and clear it).
    if (first || !ctx->latched || w != ctx->w || h != ctx->h) {
        FrameParams probe;
        int rc = fill_params(ctx, scene, w, h, (float)time, max_steps, probe);
        if (rc != RMDF_OK) return rc;
        const bool resized = (w != ctx->w || h != ctx->h || !ctx->d_rgba8);
        rc = ensure_frame(ctx, w, h, planes);
        if (rc != RMDF_OK) return rc;
        if (resized) { rc = clear_frame(ctx, w, h); if (rc != RMDF_OK) return rc; }
        ctx->w = w; ctx->h = h; ctx->time = (float)time; ctx->max_steps = max_steps <= 0 ? (int)shk::march_max_steps_default : max_steps;
        ctx->latched = true;
    }
    FrameParams p;
    int rc = fill_params(ctx, scene, ctx->w, ctx->h, ctx->time, ctx->max_steps, p);
    if (rc != RMDF_OK) return rc;
    rc = ensure_frame(ctx, ctx->w, ctx->h, planes);     // the extra planes appear the first time a caller asks for them
    if (rc != RMDF_OK) return rc;
    if (whole) { p.x0 = 0; p.y0 = 0; p.x1 = ctx->w; p.y1 = ctx->h; }
    else tile_rect_host(tile_idx, ctx->w, ctx->h, &p.x0, &p.y0, &p.x1, &p.y1);
    p.rgba8 = ctx->d_rgba8;
    if (planes) { p.rgba_f32 = ctx->d_rgba_f32; p.steps = ctx->d_steps; p.iters = ctx->d_iters; }
    const size_t npx = (size_t)ctx->w * ctx->h;
    // (Until round 5 a whole-frame call into a buffer registered with rmdf_register_host_buffer was written by the render kernel directly:
    // hipHostRegister on the caller's pages.  Every GPU memory fault of the round-4 / round-5 hunt fell on a heap range that such a
    // registration had covered -- registered, used, unregistered, and minutes later part of a larger buffer the HIP runtime page-locked
    // for one of its own copies (NOTEBOOK.md A.5, profiles/r05_fault_hunt.txt).  The library no longer asks the driver for a GPU mapping
    // of memory it does not own: registration is bookkeeping, every buffer takes the staged path.)
    // Tile mode hands back the WHOLE accumulating frame on every call (the reference maps a freshly orphaned PBO each time:
    // FrameBuffer.hs:129,207-213), 64 times per frame, and every call has to wait for its tile's kernel -- whose run time is its
    // longest ray's, not 1/64 of the frame's.  What the call does instead of `launch, copy 8.3 MB over PCIe, wait`:
    //  * tiles are rendered as JOBS on streams of their own, into a device scratch tile and (the kernel's mirror store) a
    //    page-locked host tile; the job of tile i + 1 is issued speculatively as soon as call i has found its own,
    //    so it runs while call i copies and while the caller is between calls;
    //  * the library keeps a page-locked shadow of the frame; a few host threads copy it into the caller's buffer while the job
    //    finishes; then the tile goes from the host tile into the shadow and the caller's buffer (130 KB), and from the scratch
    //    tile into the device frame (asynchronously: nothing waits for it but later renders).
    // Frames beyond 1 GiB keep the plain path below (launch in place, whole frame copied): nobody page-locks a shadow of that size.
    if (!whole && out_rgba8 && !planes && p.y1 > p.y0 && p.x1 > p.x0 && npx * 4 <= ((size_t)1 << 30)
#ifdef RMDF_XCHECK
        && !(ctx->flags & RMDF_FLAG_FLAT_MARCH)
#endif
    )
        return render_tile_fast(ctx, scene, tile_idx, p, out_rgba8);
    // whole frame into caller memory that is not registered: row bands through the page-locked shadow (render_whole_frame_host)
    if (whole && out_rgba8 && !planes && npx * 4 <= ((size_t)1 << 30)
#ifdef RMDF_XCHECK
        && !(ctx->flags & RMDF_FLAG_FLAT_MARCH)
#endif
    )
        return render_whole_frame_host(ctx, scene, p, out_rgba8);
    rc = launch_scene(ctx, scene, p, ctx->stream);
    if (rc != RMDF_OK) return rc;
    // (every other case -- the test planes, frames too large for a page-locked shadow, the cross-check schedule, no output at all --
    // copies what was asked for behind the launch, through the staging chunks)
    ctx->shadow_valid = false;                               // the device frame moves on without the shadow
    if (out_rgba8) RMDF_TRY(download(ctx, out_rgba8, ctx->d_rgba8, npx * 4, ctx->stream));
    if (out_rgba_f32) RMDF_TRY(download(ctx, out_rgba_f32, ctx->d_rgba_f32, npx * 16, ctx->stream));
    if (out_steps) RMDF_TRY(download(ctx, out_steps, ctx->d_steps, npx * 2, ctx->stream));
    if (out_iters) RMDF_TRY(download(ctx, out_iters, ctx->d_iters, npx * 2, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RMDF_OK;
}

}  // namespace


namespace rmdf {
void tile_rect_host(int tile_idx, int w, int h, int *x0, int *y0, int *x1, int *y1)
{
    // ShaderRendering.hs:183-193, centre-inside rasterisation (see tile_rect in rmdf_render.hip)
    int midx = tile_idx % 64;
    int tx = midx % 8, ty = midx / 8;
    *x0 = (2 * tx * w + 7) / 16;
    *x1 = (2 * (tx + 1) * w + 7) / 16;
    *y0 = (2 * ty * h + 7) / 16;
    *y1 = (2 * (ty + 1) * h + 7) / 16;
}
}  // namespace rmdf


extern "C" {

int rmdf_create_ex(rmdf_ctx **out, const rmdf_config *cfg, char *err, size_t err_len)
{
    const int rc = rmdf_create(out, cfg);
    if (err && err_len > 0) snprintf(err, err_len, "%s", rc == RMDF_OK ? "" : rmdf_last_error(nullptr));
    return rc;
}

int rmdf_create(rmdf_ctx **out, const rmdf_config *cfg)
{
    if (!out) return fail(nullptr, RMDF_E_INVALID, "null out pointer");
    *out = nullptr;
    RMDF_GUARD_BEGIN
    // Frames in flight live on separate HIP streams (rmdf_render_rect_device / rmdf_render_shard_device): with the
    // runtime's default of 4 hardware queues more than four streams share queues and serialise (measured: 8 frames in
    // flight run at 0.11 ms per frame with 8 queues, 0.069 ms with 12 or more).  The variable is read when the HIP runtime
    // initialises, so it only takes effect if this is the process's first HIP call; a value the host set itself is kept.
    setenv("GPU_MAX_HW_QUEUES", "16", 0);
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, RMDF_E_NO_DEVICE, std::string("no HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0"));
    int dev = cfg ? cfg->device : 0;
    if (dev < 0 || dev >= ndev) return fail(nullptr, RMDF_E_INVALID, "device ordinal out of range");
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, dev);
    if (e != hipSuccess) return fail(nullptr, RMDF_E_HIP, std::string("hipGetDeviceProperties: ") + hipGetErrorString(e));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, RMDF_E_NO_DEVICE, std::string("librmdf is built for gfx950 only, found ") + prop.gcnArchName);
    const int flags = cfg ? cfg->reserved[0] : 0;
#ifndef RMDF_XCHECK
    if (flags & ~(RMDF_FLAG_RASTER_ORDER | RMDF_FLAG_NO_MERGE | RMDF_FLAG_NO_PRUNE))
        return fail(nullptr, RMDF_E_UNSUPPORTED, "unknown flag bits (the alternative schedules live in librmdf_xcheck.so, include/rmdf_xcheck.h)");
#endif
    rmdf_ctx *ctx = new (std::nothrow) rmdf_ctx();
    if (!ctx) return fail(nullptr, RMDF_E_NOMEM, "out of host memory");
    ctx->device = dev;
    ctx->flags = flags;
    ctx->copy_threads = cfg ? cfg->reserved[1] : 0;
    if (ctx->copy_threads < 0 || ctx->copy_threads > 64) { delete ctx; return fail(nullptr, RMDF_E_INVALID, "rmdf_config.reserved[1] (host copy threads): 0 .. 64"); }
    ctx->cus = prop.multiProcessorCount;
    snprintf(ctx->dev_name, sizeof ctx->dev_name, "%s (%s)", prop.name, prop.gcnArchName);
    float tri[96 * 3];
    cornell_triangles(tri);
    float tab[CORNELL_TAB_FLOATS];
    cornell_table(tri, tab);
    const std::vector<uint32_t> &cgrid = cornell_grids(tri);
    if ((e = hipSetDevice(dev)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = dev_malloc((void **)&ctx->d_cornell, sizeof tri)) != hipSuccess ||
        (e = dev_malloc((void **)&ctx->d_cornell_tab, sizeof tab)) != hipSuccess ||
        (e = dev_malloc((void **)&ctx->d_cornell_grid, cgrid.size() * 4)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming)) != hipSuccess) {
        std::string msg = std::string("device init: ") + hipGetErrorString(e);
        rmdf_destroy(ctx);
        return fail(nullptr, RMDF_E_HIP, msg);
    }
    // the tables go up through the staging chunks like everything else (caller streams may render from them: complete before return)
    if (upload(ctx, ctx->d_cornell, tri, sizeof tri, ctx->stream) != RMDF_OK || upload(ctx, ctx->d_cornell_tab, tab, sizeof tab, ctx->stream) != RMDF_OK ||
        upload(ctx, ctx->d_cornell_grid, cgrid.data(), cgrid.size() * 4, ctx->stream) != RMDF_OK ||
        (e = hipStreamSynchronize(ctx->stream)) != hipSuccess) {
        std::string msg = "device init: " + (e != hipSuccess ? std::string(hipGetErrorString(e)) : ctx->err);
        rmdf_destroy(ctx);
        return fail(nullptr, RMDF_E_HIP, msg);
    }
    ctx->wf_bands = cfg ? cfg->reserved[2] : 0;
    ctx->wf_mirror = cfg ? cfg->reserved[3] : 0;
    if (ctx->wf_bands < 0 || ctx->wf_bands > RMDF_WF_MAX_BANDS || ctx->wf_mirror < 0 || ctx->wf_mirror > 3) {
        rmdf_destroy(ctx);
        return fail(nullptr, RMDF_E_INVALID, "rmdf_config.reserved[2] (whole-frame row bands): 0 .. 16; reserved[3] (how the bands reach the host): 0 .. 3");
    }
#ifndef RMDF_XCHECK
    if (ctx->wf_mirror >= 2) {
        // the one-launch hand-over (a host spin on flags the kernel writes) has not had a green run on hardware: not in the product
        rmdf_destroy(ctx);
        return fail(nullptr, RMDF_E_UNSUPPORTED, "rmdf_config.reserved[3] = 2, 3 (one-launch band hand-over) lives in librmdf_xcheck.so only");
    }
#endif
    for (int k = 0; k < 4; k++)
        if ((e = hipStreamCreateWithFlags(&ctx->pstream[k], hipStreamNonBlocking)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&ctx->ev_join[k], hipEventDisableTiming)) != hipSuccess) {
            std::string msg = std::string("device init: ") + hipGetErrorString(e);
            rmdf_destroy(ctx);
            return fail(nullptr, RMDF_E_HIP, msg);
        }
    *out = ctx;
    return RMDF_OK;
    RMDF_GUARD_END(nullptr)
}

void rmdf_destroy(rmdf_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();              // caller streams may still be running launches that use the ctx's tables
    rmdf_comm_destroy(ctx);
    for (auto &s : ctx->env) if (s.d_texels) (void)dev_free(s.d_texels);
    for (auto &t : ctx->uv_tables) (void)dev_free(t.d_uv);
    for (auto &t : ctx->lobe_tables) { (void)dev_free(t.d_lutT); (void)dev_free(t.d_tcs); }
    for (auto &sc : ctx->env_scratch) if (sc.p) (void)dev_free(sc.p);
    if (ctx->d_cornell) (void)dev_free(ctx->d_cornell);
    if (ctx->d_cornell_tab) (void)dev_free(ctx->d_cornell_tab);
    if (ctx->d_cornell_grid) (void)dev_free(ctx->d_cornell_grid);
    if (ctx->probe_stream) (void)hipStreamDestroy(ctx->probe_stream);
    if (ctx->probe_host) (void)hipHostFree(ctx->probe_host);
    if (ctx->d_rgba8) (void)dev_free(ctx->d_rgba8);
    if (ctx->h_shadow) (void)hipHostFree(ctx->h_shadow);
    ctx->staging.destroy();
    for (int k = 0; k < RMDF_WF_MAX_BANDS; k++) {
        if (ctx->wf_done[k]) (void)hipEventDestroy(ctx->wf_done[k]);
        if (ctx->wf_stream[k]) (void)hipStreamDestroy(ctx->wf_stream[k]);
    }
    if (ctx->wf_fork) (void)hipEventDestroy(ctx->wf_fork);
    if (ctx->wf_flags) (void)hipHostFree((void *)ctx->wf_flags);
    if (ctx->d_wf_count) (void)dev_free(ctx->d_wf_count);
    for (auto &j : ctx->tile_job) {
        if (j.d_tile) (void)dev_free(j.d_tile);
        if (j.h_tile) (void)hipHostFree(j.h_tile);
        if (j.done) (void)hipEventDestroy(j.done);
        if (j.copied) (void)hipEventDestroy(j.copied);
        if (j.stream) (void)hipStreamDestroy(j.stream);
    }
    if (ctx->d_rgba_f32) (void)dev_free(ctx->d_rgba_f32);
    if (ctx->d_steps) (void)dev_free(ctx->d_steps);
    if (ctx->d_iters) (void)dev_free(ctx->d_iters);
#ifdef RMDF_XCHECK
    if (ctx->d_gbuf_nao) (void)dev_free(ctx->d_gbuf_nao);
    if (ctx->d_gbuf_meta) (void)dev_free(ctx->d_gbuf_meta);
    if (ctx->d_work_counter) (void)dev_free(ctx->d_work_counter);
    if (ctx->d_dbg) (void)dev_free(ctx->d_dbg);
#endif
    for (auto &o : ctx->orders) {
        if (o.d_cost) (void)dev_free(o.d_cost);
        if (o.d_order) (void)dev_free(o.d_order);
    }
    for (int k = 0; k < 4; k++) {
        if (ctx->pstream[k]) (void)hipStreamDestroy(ctx->pstream[k]);
        if (ctx->ev_join[k]) (void)hipEventDestroy(ctx->ev_join[k]);
    }
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *rmdf_last_error(const rmdf_ctx *ctx)
{
    if (ctx) return ctx->err.c_str();
    // a per-thread copy of the process-wide message: the pointer stays valid while other threads fail and overwrite it
    static thread_local char buf[1024];
    std::lock_guard<std::mutex> lock(g_error_mutex);
    snprintf(buf, sizeof buf, "%s", g_global_error.c_str());
    return buf;
}

int rmdf_is_tile_idx_first_tile(int idx) { return idx % RMDF_N_TILES == 0; }
int rmdf_is_tile_idx_last_tile(int idx) { return idx % RMDF_N_TILES == RMDF_N_TILES - 1; }

int rmdf_set_env_cube(rmdf_ctx *ctx, int slot, const float *faces_rgb, int face_w)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (slot < 0 || slot >= RMDF_ENV_SLOTS || !faces_rgb || face_w < 1 || face_w > 8192)
        return fail(ctx, RMDF_E_INVALID, "rmdf_set_env_cube: bad argument");
    RMDF_GUARD_BEGIN
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf faces;
    size_t bytes = (size_t)6 * face_w * face_w * 3 * sizeof(float);
    HIP_TRY(ctx, dev_malloc(&faces.p, bytes));
    RMDF_TRY(upload(ctx, faces.p, faces_rgb, bytes, ctx->stream));
    return set_env_from_device_faces(ctx, slot, (const float *)faces.p, face_w);
    RMDF_GUARD_END(ctx)
}

int rmdf_set_env_latlong(rmdf_ctx *ctx, int slot, const float *rgb, int w, int h)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (slot < 0 || slot >= RMDF_ENV_SLOTS || !rgb || w < 6 || h < 2 || w > 65536 || h > 32768)
        return fail(ctx, RMDF_E_INVALID, "rmdf_set_env_latlong: bad argument");
    RMDF_GUARD_BEGIN
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int cw = w / 3;
    const float2 *d_uv = nullptr;
    int rc = get_uv_table(ctx, cw, &d_uv);
    if (rc != RMDF_OK) return rc;
    DevBuf ll, faces;
    size_t ll_bytes = (size_t)w * h * 3 * sizeof(float), f_bytes = (size_t)6 * cw * cw * 3 * sizeof(float);
    HIP_TRY(ctx, dev_malloc(&ll.p, ll_bytes));
    HIP_TRY(ctx, dev_malloc(&faces.p, f_bytes));
    RMDF_TRY(upload(ctx, ll.p, rgb, ll_bytes, ctx->stream));
    HIP_TRY(ctx, launch_latlong_to_cube((const float *)ll.p, w, h, d_uv, (float *)faces.p, ctx->stream));
    return set_env_from_device_faces(ctx, slot, (const float *)faces.p, cw);
    RMDF_GUARD_END(ctx)
}

int rmdf_get_env_cube_padded(rmdf_ctx *ctx, int slot, uint16_t *out, int *face_w)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (slot < 0 || slot >= RMDF_ENV_SLOTS) return fail(ctx, RMDF_E_INVALID, "bad slot");
    if (!ctx->env[slot].d_texels) return fail(ctx, RMDF_E_NO_ENV, "slot not set");
    const int W = ctx->env[slot].W;
    if (face_w) *face_w = W;
    if (!out) return RMDF_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    return download(ctx, out, ctx->env[slot].d_texels, (size_t)6 * (W + 2) * (W + 2) * 8, ctx->stream);
}

int rmdf_resize_latlong(rmdf_ctx *ctx, const float *rgb, int w, int h, int dstw, float *out, int *dsth)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!rgb || w < 2 || h < 2 || w > 32768 || h > 16384 || dstw < 1 || dstw > 32768 || !dsth)
        return fail(ctx, RMDF_E_INVALID, "rmdf_resize_latlong: bad argument (2 <= w <= 32768, 2 <= h <= 16384, 1 <= dstw <= 32768)");
    // dsth = round (srch / srcw * dstw) in Float, Haskell round = half-to-even (HDREnvMap.hs:173)
    const int dh = (int)rintf((float)h / (float)w * (float)dstw);
    *dsth = dh;
    if (!out) return RMDF_OK;
    if (dh < 1) return fail(ctx, RMDF_E_INVALID, "destination height < 1");
    if ((long long)dstw * dh > (1ll << 28)) return fail(ctx, RMDF_E_INVALID, "rmdf_resize_latlong: destination larger than 2^28 texels");
    RMDF_GUARD_BEGIN
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf src, dst;
    size_t sb = (size_t)w * h * 12, db = (size_t)dstw * dh * 12;
    HIP_TRY(ctx, dev_malloc(&src.p, sb));
    HIP_TRY(ctx, dev_malloc(&dst.p, db));
    RMDF_TRY(upload(ctx, src.p, rgb, sb, ctx->stream));
    HIP_TRY(ctx, launch_resize_latlong((const float *)src.p, w, h, dstw, dh, (float *)dst.p, ctx->stream));
    return download(ctx, out, dst.p, db, ctx->stream);
    RMDF_GUARD_END(ctx)
}

int rmdf_prefilter_env_powers(rmdf_ctx *ctx, const float *rgb, int w, int h, const float *powers, int npowers, float *out)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!rgb || !out || !powers || npowers < 1 || npowers > 16 || w < 2 || h < 2 || w > 8192 || h > 4096)
        return fail(ctx, RMDF_E_INVALID, "rmdf_prefilter_env_powers: bad argument (2 <= w <= 8192, 2 <= h <= 4096, 1 <= npowers <= 16)");
    RMDF_GUARD_BEGIN
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<float *> d_out((size_t)npowers);
    const size_t b = (size_t)w * h * 12;
    const bool keep = b <= rmdf_ctx::kEnvScratchKeepBytes;
    std::vector<DevBuf> once(keep ? 0 : (size_t)npowers + 1);   // larger maps: freed when the call returns (hipFree waits for the device)
    void *d_src = nullptr;
    for (int i = 0; i <= npowers; i++) {                   // slot 0 = the source, 1 + i = power i
        void *p = nullptr;
        if (keep) {
            rmdf_ctx::Scratch &sc = ctx->env_scratch[i];
            if (sc.bytes < b) {
                if (sc.p) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)dev_free(sc.p); sc.p = nullptr; sc.bytes = 0; }
                HIP_TRY(ctx, dev_malloc(&sc.p, b));
                sc.bytes = b;
            }
            p = sc.p;
        } else {
            HIP_TRY(ctx, dev_malloc(&once[(size_t)i].p, b));
            p = once[(size_t)i].p;
        }
        if (i > 0) d_out[i - 1] = (float *)p; else d_src = p;
    }
    // (the kept scratch is reused by the next call and the per-call buffers are freed on the way out -- hipFree waits for the device --
    // so an error exit needs no wait of its own: nothing the device still touches belongs to the caller, only to the staging chunks)
    RMDF_TRY(upload(ctx, d_src, rgb, b, ctx->stream));
    int rc = prefilter_powers_device(ctx, (const float *)d_src, w, h, powers, npowers, d_out.data());
    if (rc != RMDF_OK) return rc;
    for (int i = 0; i < npowers; i++)
        RMDF_TRY(download(ctx, out + (size_t)i * w * h * 3, d_out[i], b, ctx->stream));
    return RMDF_OK;
    RMDF_GUARD_END(ctx)
}

int rmdf_prefilter_env(rmdf_ctx *ctx, const float *rgb, int w, int h, float power, float *out)
{
    return rmdf_prefilter_env_powers(ctx, rgb, w, h, &power, 1, out);
}

int rmdf_prefilter_env_device(rmdf_ctx *ctx, const void *d_rgb, int w, int h, float power, void *d_out, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!d_rgb || !d_out || w < 2 || h < 2 || w > 8192 || h > 4096) return fail(ctx, RMDF_E_INVALID, "rmdf_prefilter_env_device: bad argument");
    RMDF_GUARD_BEGIN
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const float *d_lutT = nullptr; const float2 *d_tcs = nullptr;
    int rc = get_lobe_tables(ctx, w, h, &d_lutT, &d_tcs);
    if (rc != RMDF_OK) return rc;
    HIP_TRY(ctx, launch_prefilter((const float *)d_rgb, w, h, power, d_lutT, d_tcs, (float *)d_out, stream ? (hipStream_t)stream : ctx->stream, true));
    return RMDF_OK;
    RMDF_GUARD_END(ctx)
}

int rmdf_load_env_hdr(rmdf_ctx *ctx, const char *path)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!path) return fail(ctx, RMDF_E_INVALID, "null path");
    RMDF_GUARD_BEGIN
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<uint8_t> file;
    if (!read_file(path, file)) return fail(ctx, RMDF_E_IO, std::string("cannot read ") + path);
    int w = 0, h = 0;
    std::vector<float> refl;
    std::string why;
    if (!decode_hdr(file.data(), file.size(), w, h, refl, why)) return fail(ctx, RMDF_E_IO, std::string(path) + ": " + why);
    if (w < 6 || h < 2) return fail(ctx, RMDF_E_IO, std::string(path) + ": image too small for an environment map");
    // powers / cache file names, ShaderRendering.hs:71-75 (`show pow` of a Float: "1.0")
    static const struct { int slot; const char *suffix; float power; } kPow[4] = {
        { RMDF_ENV_COS_1, "1.0", 1.0f }, { RMDF_ENV_COS_8, "8.0", 8.0f },
        { RMDF_ENV_COS_64, "64.0", 64.0f }, { RMDF_ENV_COS_512, "512.0", 512.0f } };
    std::string stem(path);
    size_t dot = stem.find_last_of('.'), slash = stem.find_last_of('/');
    if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) stem = stem.substr(0, dot);
    // buildPreConvolvedHDREnvMapCache (ShaderRendering.hs:131-149): the missing maps are convolved from the reflection map
    // resized to 256 texels, all missing powers concurrently, each written as a Radiance file
    std::vector<uint8_t> cache_img[4];
    int missing[4], nmiss = 0;
    float mpow[4];
    for (int k = 0; k < 4; k++)
        if (!file_exists(stem + "_cache_pow_" + kPow[k].suffix + ".hdr")) { mpow[nmiss] = kPow[k].power; missing[nmiss++] = k; }
    if (nmiss > 0) {
        const int rw = 256;
        int rh = 0;
        int rc = rmdf_resize_latlong(ctx, refl.data(), w, h, rw, nullptr, &rh);
        if (rc != RMDF_OK) return rc;
        if (rh < 2) return fail(ctx, RMDF_E_IO, std::string(path) + ": aspect ratio leaves no rows at 256 texels width");
        std::vector<float> resized((size_t)rw * rh * 3), conv((size_t)nmiss * rw * rh * 3);
        rc = rmdf_resize_latlong(ctx, refl.data(), w, h, rw, resized.data(), &rh);
        if (rc != RMDF_OK) return rc;
        rc = rmdf_prefilter_env_powers(ctx, resized.data(), rw, rh, mpow, nmiss, conv.data());
        if (rc != RMDF_OK) return rc;
        for (int m = 0; m < nmiss; m++) {
            const int k = missing[m];
            std::vector<float> one(conv.begin() + (size_t)m * rw * rh * 3, conv.begin() + (size_t)(m + 1) * rw * rh * 3);
            encode_hdr(one, rw, rh, cache_img[k]);
            // A directory that cannot be written (read-only install) is not an error here: the file image just built is
            // used from memory -- the same bytes the reload would read -- and the cache is rebuilt next time.
            (void)write_file_atomic(stem + "_cache_pow_" + kPow[k].suffix + ".hdr", cache_img[k]);
        }
    }
    for (int k = 0; k < 4; k++) {
        const std::string fn = stem + "_cache_pow_" + kPow[k].suffix + ".hdr";
        if (cache_img[k].empty() && !read_file(fn.c_str(), cache_img[k])) return fail(ctx, RMDF_E_IO, "cannot read cache file " + fn);
        int cw = 0, chh = 0;
        std::vector<float> cimg;
        if (!decode_hdr(cache_img[k].data(), cache_img[k].size(), cw, chh, cimg, why)) return fail(ctx, RMDF_E_IO, fn + ": " + why);
        if (cw < 6 || chh < 2) return fail(ctx, RMDF_E_IO, fn + ": image too small for an environment map");
        int rc = rmdf_set_env_latlong(ctx, kPow[k].slot, cimg.data(), cw, chh);
        if (rc != RMDF_OK) return rc;
    }
    return rmdf_set_env_latlong(ctx, RMDF_ENV_REFLECTION, refl.data(), w, h);
    RMDF_GUARD_END(ctx)
}

int rmdf_render_tile(rmdf_ctx *ctx, int scene, int tile_idx, int w, int h, double time, int max_steps,
                     uint32_t *out_rgba8)
{
    RMDF_GUARD_BEGIN
    return render_common(ctx, scene, tile_idx, w, h, time, max_steps, out_rgba8, nullptr, nullptr, nullptr);
    RMDF_GUARD_END(ctx)
}

int rmdf_render_tile_ex(rmdf_ctx *ctx, int scene, int tile_idx, int w, int h, double time, int max_steps,
                        uint32_t *out_rgba8, float *out_rgba_f32, uint16_t *out_steps, uint16_t *out_iters)
{
    RMDF_GUARD_BEGIN
    return render_common(ctx, scene, tile_idx, w, h, time, max_steps, out_rgba8, out_rgba_f32, out_steps, out_iters);
    RMDF_GUARD_END(ctx)
}

int rmdf_render_rect_device(rmdf_ctx *ctx, int scene, int w, int h, double time, int max_steps,
                            int x0, int y0, int x1, int y1,
                            void *d_rgba8, void *d_rgba_f32, void *d_steps, void *d_iters, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    RMDF_GUARD_BEGIN
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    FrameParams p;
    int rc = fill_params(ctx, scene, w, h, (float)time, max_steps, p);
    if (rc != RMDF_OK) return rc;
    if (x0 < 0 || y0 < 0 || x1 > w || y1 > h || x0 > x1 || y0 > y1) return fail(ctx, RMDF_E_INVALID, "bad rectangle");
    if (!d_rgba8 && !d_rgba_f32 && !d_steps && !d_iters) return fail(ctx, RMDF_E_INVALID, "rmdf_render_rect_device: no output buffer");
    p.x0 = x0; p.y0 = y0; p.x1 = x1; p.y1 = y1;
    p.rgba8 = (uint32_t *)d_rgba8; p.rgba_f32 = (float4 *)d_rgba_f32; p.steps = (uint16_t *)d_steps; p.iters = (uint16_t *)d_iters;
    return launch_scene(ctx, scene, p, stream ? (hipStream_t)stream : ctx->stream);
    RMDF_GUARD_END(ctx)
}

int rmdf_render_shard_device(rmdf_ctx *ctx, int scene, int w, int h, double time, int max_steps,
                             int rank, int nranks, void *d_packed_rgba8, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks || !d_packed_rgba8)
        return fail(ctx, RMDF_E_INVALID, "rmdf_render_shard_device: bad rank / nranks / buffer");
    if (w % 8 || h % 8) return fail(ctx, RMDF_E_INVALID, "tile sharding needs w and h divisible by 8");
    RMDF_GUARD_BEGIN
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    FrameParams p;
    int rc = fill_params(ctx, scene, w, h, (float)time, max_steps, p);
    if (rc != RMDF_OK) return rc;
    ensure_deal(ctx, nranks);
    p.n_shard_tiles = ctx->deal_count[rank];
    memcpy(p.shard_tile, ctx->deal_tiles[rank], sizeof p.shard_tile);
    p.shard_key = (int)(ctx->shard_cost_gen << 16) + rank * 256 + nranks;
    p.rgba8 = (uint32_t *)d_packed_rgba8;
    return launch_scene(ctx, scene, p, stream ? (hipStream_t)stream : ctx->stream);
    RMDF_GUARD_END(ctx)
}

int rmdf_get_cornell_vertices(float out[96 * 3])
{
    if (!out) return RMDF_E_INVALID;
    cornell_triangles(out);
    return RMDF_OK;
}

int rmdf_get_shader_constants(const char **names, float *values, int cap)
{
    static const char *const k_names[] = {
#define RMDF_X(name, value) #name,
        RMDF_SHADER_CONSTANTS(RMDF_X)
#undef RMDF_X
    };
    static const float k_values[] = {
#define RMDF_X(name, value) shk::name,
        RMDF_SHADER_CONSTANTS(RMDF_X)
#undef RMDF_X
    };
    const int n = (int)(sizeof k_values / sizeof k_values[0]);
    for (int i = 0; i < n && i < cap; i++) { if (names) names[i] = k_names[i]; if (values) values[i] = k_values[i]; }
    return n;
}

int rmdf_shard_tiles(int rank, int nranks, int tiles[64])
{
    if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks || !tiles) return RMDF_E_INVALID;
    unsigned char t[64];
    const int cnt = shard_tiles_of_rank(rank, nranks, t);
    for (int i = 0; i < cnt; i++) tiles[i] = t[i];
    return cnt;
}

int rmdf_set_shard_costs(rmdf_ctx *ctx, const float cost[64])
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (cost) {
        for (int i = 0; i < 64; i++)
            if (!(cost[i] >= 0.0f) || cost[i] > 3.0e38f) return fail(ctx, RMDF_E_INVALID, "rmdf_set_shard_costs: costs must be finite and >= 0");
        memcpy(ctx->shard_cost, cost, sizeof ctx->shard_cost);
    }
    ctx->shard_cost_set = cost != nullptr;
    ctx->shard_cost_gen++;
    return RMDF_OK;
}

int rmdf_set_shard_root_handicap(rmdf_ctx *ctx, float fraction)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!(fraction >= 0.0f) || fraction > 1.0f) return fail(ctx, RMDF_E_INVALID, "rmdf_set_shard_root_handicap: 0 <= fraction <= 1");
    ctx->shard_root_handicap = fraction;
    ctx->shard_cost_gen++;
    return RMDF_OK;
}

int rmdf_get_shard_tiles(rmdf_ctx *ctx, int rank, int nranks, int tiles[64])
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks || !tiles) return fail(ctx, RMDF_E_INVALID, "rmdf_get_shard_tiles: bad argument");
    ensure_deal(ctx, nranks);
    for (int i = 0; i < ctx->deal_count[rank]; i++) tiles[i] = ctx->deal_tiles[rank][i];
    return ctx->deal_count[rank];
}


int rmdf_probe_tile_costs(rmdf_ctx *ctx, int scene, int w, int h, double time, int max_steps, float cost[64])
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!cost || w <= 0 || h <= 0) return fail(ctx, RMDF_E_INVALID, "rmdf_probe_tile_costs: bad argument");
    RMDF_GUARD_BEGIN
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the same view at 256 x 144 (aspect ratio of the real frame kept to within a pixel): 32 x 18 rays per tile
    const int pw = 256;
    int ph = (int)((long long)pw * h / w);
    ph = (ph + 7) & ~7;
    if (ph < 8) ph = 8;
    if (ph > 4096) ph = 4096;
    FrameParams p;
    int rc = fill_params(ctx, scene, pw, ph, (float)time, max_steps, p);
    if (rc != RMDF_OK) return rc;
    // aspect of the real frame, so that the probe rays cover the real frame's field of view
    p.aspect = (float)w / (float)h;
    DevBuf steps, iters;
    const size_t npx = (size_t)pw * ph;
    HIP_TRY(ctx, dev_malloc(&steps.p, npx * 2));
    HIP_TRY(ctx, dev_malloc(&iters.p, npx * 2));
    p.x0 = 0; p.y0 = 0; p.x1 = pw; p.y1 = ph;
    p.steps = (uint16_t *)steps.p; p.iters = (uint16_t *)iters.p;
    HIP_TRY(ctx, launch_render(scene, p, ctx->stream));
    std::vector<uint16_t> hs(npx), hi(npx);
    RMDF_TRY(download(ctx, hs.data(), steps.p, npx * 2, ctx->stream));
    RMDF_TRY(download(ctx, hi.data(), iters.p, npx * 2, ctx->stream));
    // cost of a ray ~ escape iterations (march + normal + AO) + march steps, plus a constant per pixel
    double acc[64] = { 0 };
    const int tw = pw / 8, th = ph / 8;
    for (int y = 0; y < ph; y++)
        for (int x = 0; x < pw; x++) {
            const size_t i = (size_t)y * pw + x;
            acc[(x / tw) + 8 * (y / th)] += 1.0 + (double)hi[i] + (double)(hs[i] & 0x7fff);
        }
    for (int i = 0; i < 64; i++) cost[i] = (float)acc[i];
    return RMDF_OK;
    RMDF_GUARD_END(ctx)
}

int rmdf_assemble_shards_device(rmdf_ctx *ctx, int w, int h, int nranks, const void *d_gathered,
                                void *d_frame_rgba8, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (nranks < 1 || nranks > 64 || !d_gathered || !d_frame_rgba8 || w % 8 || h % 8 || w <= 0 || h <= 0)
        return fail(ctx, RMDF_E_INVALID, "rmdf_assemble_shards_device: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ensure_deal(ctx, nranks);
    HIP_TRY(ctx, launch_assemble_shards((const uint32_t *)d_gathered, (uint32_t *)d_frame_rgba8, w, h, nranks, ctx->deal_where,
                                        stream ? (hipStream_t)stream : ctx->stream));
    return RMDF_OK;
}


// ---- multi-GPU: the single exchange of the path behind the C ABI ------------------------------------------

int rmdf_comm_get_unique_id(void *id)
{
    if (!id) return fail(nullptr, RMDF_E_INVALID, "rmdf_comm_get_unique_id: null id");
    RMDF_GUARD_BEGIN
    int rc = load_rccl(nullptr);
    if (rc != RMDF_OK) return rc;
    static_assert(sizeof(ncclUniqueId) == RMDF_COMM_ID_BYTES, "RMDF_COMM_ID_BYTES must equal sizeof(ncclUniqueId)");
    ncclUniqueId uid;
    RCCL_TRY(nullptr, g_rccl.GetUniqueId(&uid));
    memcpy(id, &uid, sizeof uid);
    return RMDF_OK;
    RMDF_GUARD_END(nullptr)
}

int rmdf_comm_init(rmdf_ctx *ctx, const void *id, int rank, int nranks)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!id || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return fail(ctx, RMDF_E_INVALID, "rmdf_comm_init: bad id / rank / nranks (<= 64: one rank per tile at most)");
    if (ctx->comm) return fail(ctx, RMDF_E_INVALID, "rmdf_comm_init: this ctx already has a communicator (rmdf_comm_destroy first)");
    RMDF_GUARD_BEGIN
    int rc = load_rccl(ctx);
    if (rc != RMDF_OK) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    ncclComm_t comm = nullptr;
    if (!ctx->d_verify) HIP_TRY(ctx, dev_malloc((void **)&ctx->d_verify, (size_t)64 * 8 + 8));
    RCCL_TRY(ctx, g_rccl.CommInitRank(&comm, nranks, uid, rank));
    ctx->comm = comm; ctx->comm_rank = rank; ctx->comm_nranks = nranks;
    return RMDF_OK;
    RMDF_GUARD_END(ctx)
}

int rmdf_comm_destroy(rmdf_ctx *ctx)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!ctx->comm) return RMDF_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    ncclComm_t c = ctx->comm;
    ctx->comm = nullptr; ctx->comm_rank = 0; ctx->comm_nranks = 1;
    if (ctx->d_verify) { (void)dev_free(ctx->d_verify); ctx->d_verify = nullptr; }
    RCCL_TRY(ctx, g_rccl.CommDestroy(c));
    return RMDF_OK;
}

int rmdf_comm_info(rmdf_ctx *ctx, int *rank, int *nranks)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (rank) *rank = ctx->comm_rank;
    if (nranks) *nranks = ctx->comm ? ctx->comm_nranks : 0;
    return RMDF_OK;
}

int rmdf_gather_shards_device(rmdf_ctx *ctx, int w, int h, const void *d_shard, void *d_gathered, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!ctx->comm) return fail(ctx, RMDF_E_COMM, "rmdf_gather_shards_device: no communicator (rmdf_comm_init)");
    if (w <= 0 || h <= 0 || w % 8 || h % 8 || !d_shard) return fail(ctx, RMDF_E_INVALID, "rmdf_gather_shards_device: bad argument");
    const int n = ctx->comm_nranks, rank = ctx->comm_rank;
    if (rank == 0 && !d_gathered) return fail(ctx, RMDF_E_INVALID, "rmdf_gather_shards_device: rank 0 needs the gather buffer");
    RMDF_GUARD_BEGIN
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    const size_t slots = (size_t)((64 + n - 1) / n);
    const size_t tile_bytes = (size_t)(w / 8) * (size_t)(h / 8) * 4;
    const size_t bytes = slots * tile_bytes;                                       // the stride of a rank's region in d_gathered
    // Every rank sends its WHOLE fixed-size region (ceil(64 / n) tile slots), whatever the deal in effect: the size on the wire is a
    // function of (w, h, n) alone, so no state a rank may hold alone -- costs set on one rank, a handicap, a verification another rank
    // has not seen -- can make a send and its receive disagree (round 4 sent exact tile counts once rmdf_comm_verify_deal had
    // succeeded; a rank that changed its costs afterwards would have left the root waiting.  For the rank counts that divide 64 the two
    // are the same bytes anyway; for the others the difference is at most one tile per rank).  A rank whose deal differs mis-assembles a
    // frame -- which rmdf_comm_verify_deal exists to detect -- and nothing hangs.  Unused slots stay where they are.
    ensure_deal(ctx, n);
    auto count_of = [&](int) -> size_t { return bytes; };
    if (rank == 0) {
        // fan-in: one receive per peer, grouped so that they progress together over the seven xGMI links of the root
        if (n > 1) {
            RCCL_TRY(ctx, g_rccl.GroupStart());
            for (int r = 1; r < n; r++)
                if (count_of(r) > 0)
                    RCCL_GROUP_TRY(ctx, g_rccl.Recv((char *)d_gathered + (size_t)r * bytes, count_of(r), ncclChar, r, ctx->comm, st));
            RCCL_TRY(ctx, g_rccl.GroupEnd());
        }
        if ((const char *)d_shard != (const char *)d_gathered)     // the root may render straight into its own slot
            HIP_TRY(ctx, hipMemcpyAsync(d_gathered, d_shard, (size_t)ctx->deal_count[0] * tile_bytes, hipMemcpyDeviceToDevice, st));
    } else if (count_of(rank) > 0) {
        RCCL_TRY(ctx, g_rccl.Send(d_shard, count_of(rank), ncclChar, 0, ctx->comm, st));
    }
    return RMDF_OK;
    RMDF_GUARD_END(ctx)
}

// FNV-1a over everything the deal is a function of
static uint64_t deal_fingerprint(rmdf_ctx *ctx, int n)
{
    ensure_deal(ctx, n);
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](const void *p, size_t len) { for (size_t i = 0; i < len; i++) { h ^= ((const unsigned char *)p)[i]; h *= 1099511628211ull; } };
    mix(&n, sizeof n);
    for (int r = 0; r < n; r++) { mix(&ctx->deal_count[r], sizeof(int)); mix(ctx->deal_tiles[r], (size_t)ctx->deal_count[r]); }
    return h;
}

int rmdf_comm_verify_deal(rmdf_ctx *ctx, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!ctx->comm) return fail(ctx, RMDF_E_COMM, "rmdf_comm_verify_deal: no communicator (rmdf_comm_init)");
    RMDF_GUARD_BEGIN
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    const int n = ctx->comm_nranks, rank = ctx->comm_rank;
    const uint64_t mine = deal_fingerprint(ctx, n);
    uint32_t verdict = 1;
    if (n > 1) {
        // peers -> root: 8 bytes each; root -> peers: the verdict.  Fixed sizes: nothing here depends on the deal.  The device words were
        // allocated with the communicator (rmdf_comm_init): no allocation -- an implicit device-wide wait -- in the middle of a collective.
        uint64_t *d_fp = ctx->d_verify;
        uint32_t *d_verdict = (uint32_t *)(d_fp + n);
        RMDF_TRY(upload(ctx, d_fp + rank, &mine, 8, st));
        if (rank == 0) {
            RCCL_TRY(ctx, g_rccl.GroupStart());
            for (int r = 1; r < n; r++) RCCL_GROUP_TRY(ctx, g_rccl.Recv(d_fp + r, 8, ncclChar, r, ctx->comm, st));
            RCCL_TRY(ctx, g_rccl.GroupEnd());
            std::vector<uint64_t> all((size_t)n);
            RMDF_TRY(download(ctx, all.data(), d_fp, (size_t)n * 8, st));
            for (int r = 1; r < n; r++) if (all[(size_t)r] != mine) verdict = 0;
            RMDF_TRY(upload(ctx, d_verdict, &verdict, 4, st));
            RCCL_TRY(ctx, g_rccl.GroupStart());
            for (int r = 1; r < n; r++) RCCL_GROUP_TRY(ctx, g_rccl.Send(d_verdict, 4, ncclChar, r, ctx->comm, st));
            RCCL_TRY(ctx, g_rccl.GroupEnd());
            HIP_TRY(ctx, hipStreamSynchronize(st));
        } else {
            RCCL_TRY(ctx, g_rccl.Send(d_fp + rank, 8, ncclChar, 0, ctx->comm, st));
            RCCL_TRY(ctx, g_rccl.Recv(d_verdict, 4, ncclChar, 0, ctx->comm, st));
            RMDF_TRY(download(ctx, &verdict, d_verdict, 4, st));
        }
    }
    if (!verdict)
        return fail(ctx, RMDF_E_COMM, "rmdf_comm_verify_deal: the ranks hold different tile deals (rmdf_set_shard_costs / rmdf_set_shard_root_handicap "
                                      "must be called with the same values on every rank): frames would be assembled from the wrong tiles");
    return RMDF_OK;
    RMDF_GUARD_END(ctx)
}

int rmdf_comm_selftest_loopback(rmdf_ctx *ctx, size_t bytes, void *stream, uint64_t *mismatches)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (bytes < 4 || bytes > ((size_t)1 << 28) || (bytes & 3)) return fail(ctx, RMDF_E_INVALID, "rmdf_comm_selftest_loopback: 4 <= bytes <= 2^28, a multiple of 4");
    if (mismatches) *mismatches = 0;
    RMDF_GUARD_BEGIN
    int rc = load_rccl(ctx);
    if (rc != RMDF_OK) return rc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    // the ctx's communicator if it has one (any size: a rank talks to itself), else a private one-rank communicator
    ncclComm_t comm = ctx->comm;
    int me = ctx->comm_rank;
    struct TmpComm {
        ncclComm_t c = nullptr;
        ~TmpComm() { if (c) { (void)hipDeviceSynchronize(); g_rccl.CommDestroy(c); } }     // an error path may leave its operations in flight
    } tmp;
    if (!comm) {
        ncclUniqueId uid;
        RCCL_TRY(ctx, g_rccl.GetUniqueId(&uid));
        RCCL_TRY(ctx, g_rccl.CommInitRank(&tmp.c, 1, uid, 0));
        comm = tmp.c; me = 0;
    }
    const size_t nw = bytes / 4;
    std::vector<uint32_t> host(nw), back(nw);
    for (size_t i = 0; i < nw; i++) { uint32_t h = (uint32_t)i * 2654435761u + 0x9e3779b9u; h ^= h >> 15; h *= 2246822519u; host[i] = h ^ (h >> 13); }
    DevBuf src, dst;
    HIP_TRY(ctx, dev_malloc(&src.p, bytes));
    HIP_TRY(ctx, dev_malloc(&dst.p, bytes));
    RMDF_TRY(upload(ctx, src.p, host.data(), bytes, st));
    HIP_TRY(ctx, hipMemsetAsync(dst.p, 0, bytes, st));
    // exactly the calls of the exchange step: a grouped receive (the root's side) and a send (a peer's side), on the caller's stream
    RCCL_TRY(ctx, g_rccl.GroupStart());
    RCCL_GROUP_TRY(ctx, g_rccl.Recv(dst.p, bytes, ncclChar, me, comm, st));
    RCCL_GROUP_TRY(ctx, g_rccl.Send(src.p, bytes, ncclChar, me, comm, st));
    RCCL_TRY(ctx, g_rccl.GroupEnd());
    RMDF_TRY(download(ctx, back.data(), dst.p, bytes, st));
    uint64_t bad = 0;
    for (size_t i = 0; i < nw; i++) bad += back[i] != host[i];
    if (mismatches) *mismatches = bad;
    if (bad) return fail(ctx, RMDF_E_COMM, "rmdf_comm_selftest_loopback: " + std::to_string(bad) + " of " + std::to_string(nw) + " words differ after ncclSend/ncclRecv to self");
    return RMDF_OK;
    RMDF_GUARD_END(ctx)
}

int rmdf_render_frame_sharded_device(rmdf_ctx *ctx, int scene, int w, int h, double time, int max_steps,
                                     void *d_shard, void *d_gathered, void *d_frame_rgba8, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!ctx->comm) return fail(ctx, RMDF_E_COMM, "rmdf_render_frame_sharded_device: no communicator (rmdf_comm_init)");
    if (ctx->comm_rank == 0 && (!d_gathered || !d_frame_rgba8)) return fail(ctx, RMDF_E_INVALID, "rmdf_render_frame_sharded_device: rank 0 needs the gather buffer and the frame");
    int rc = rmdf_render_shard_device(ctx, scene, w, h, time, max_steps, ctx->comm_rank, ctx->comm_nranks, d_shard, stream);
    if (rc != RMDF_OK) return rc;
    rc = rmdf_gather_shards_device(ctx, w, h, d_shard, d_gathered, stream);
    if (rc != RMDF_OK) return rc;
    if (ctx->comm_rank == 0) rc = rmdf_assemble_shards_device(ctx, w, h, ctx->comm_nranks, d_gathered, d_frame_rgba8, stream);
    return rc;
}

int rmdf_resolve_box2_device(rmdf_ctx *ctx, const void *d_src_rgba8, int sw, int sh, void *d_dst_rgba8, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!d_src_rgba8 || !d_dst_rgba8 || sw <= 0 || sh <= 0 || (sw & 1) || (sh & 1))
        return fail(ctx, RMDF_E_INVALID, "rmdf_resolve_box2_device: needs even, positive source sizes");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_resolve_box2((const uint32_t *)d_src_rgba8, sw, sh, (uint32_t *)d_dst_rgba8,
                                     stream ? (hipStream_t)stream : ctx->stream));
    return RMDF_OK;
}

int rmdf_render_supersampled(rmdf_ctx *ctx, int scene, int w, int h, int levels, double time, int max_steps,
                             uint32_t *out_rgba8)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!out_rgba8 || levels < 0 || levels > 3 || w <= 0 || h <= 0)
        return fail(ctx, RMDF_E_INVALID, "rmdf_render_supersampled: bad argument (levels 0..3)");
    RMDF_GUARD_BEGIN
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int sw = w << levels, sh = h << levels;
    FrameParams p;
    int rc = fill_params(ctx, scene, sw, sh, (float)time, max_steps, p);
    if (rc != RMDF_OK) return rc;
    DevBuf a, b;
    HIP_TRY(ctx, dev_malloc(&a.p, (size_t)sw * sh * 4));
    if (levels > 0) HIP_TRY(ctx, dev_malloc(&b.p, (size_t)(sw / 2) * (sh / 2) * 4));
    p.x0 = 0; p.y0 = 0; p.x1 = sw; p.y1 = sh;
    p.rgba8 = (uint32_t *)a.p;
    rc = launch_scene(ctx, scene, p, ctx->stream);
    if (rc != RMDF_OK) return rc;
    void *cur = a.p, *other = b.p;
    int cw = sw, ch = sh;
    for (int l = 0; l < levels; l++) {
        HIP_TRY(ctx, launch_resolve_box2((const uint32_t *)cur, cw, ch, (uint32_t *)other, ctx->stream));
        void *t = cur; cur = other; other = t;
        cw /= 2; ch /= 2;
    }
    return download(ctx, out_rgba8, cur, (size_t)w * h * 4, ctx->stream);
    RMDF_GUARD_END(ctx)
}

int rmdf_selftest_exact_math(rmdf_ctx *ctx, uint64_t mismatches[10])
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!mismatches) return fail(ctx, RMDF_E_INVALID, "null output");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf d;
    HIP_TRY(ctx, dev_malloc(&d.p, 10 * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(d.p, 0, 10 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, launch_selftest_exact_math((unsigned long long *)d.p, ctx->d_cornell_tab, ctx->stream));
    return download(ctx, mismatches, d.p, 10 * sizeof(unsigned long long), ctx->stream);
}

int rmdf_probe_shader_clock(rmdf_ctx *ctx, double spin_us, double *mhz)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!mhz || !(spin_us >= 1.0) || spin_us > 1e6) return fail(ctx, RMDF_E_INVALID, "rmdf_probe_shader_clock: null output or spin outside 1 us .. 1 s");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // a stream of its own (highest priority) and a buffer kept for the ctx's life: no allocation, no free -- both may wait for the
    // device to drain -- so that the probe runs BESIDE whatever the caller has in flight on other streams
    if (!ctx->probe_stream) {
        int least = 0, greatest = 0;
        HIP_TRY(ctx, hipDeviceGetStreamPriorityRange(&least, &greatest));
        HIP_TRY(ctx, hipStreamCreateWithPriority(&ctx->probe_stream, hipStreamNonBlocking, greatest));
        HIP_TRY(ctx, hipHostMalloc((void **)&ctx->probe_host, 4 * sizeof(unsigned long long), hipHostMallocMapped));
    }
    hipStream_t st = ctx->probe_stream;
    volatile unsigned long long *h = ctx->probe_host;
    h[0] = 0; h[1] = 0;
    HIP_TRY(ctx, launch_clock_probe((unsigned long long *)ctx->probe_host, (unsigned long long)(spin_us * 100.0), 1, st));   // the wave writes host memory itself
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (h[1] == 0) return fail(ctx, RMDF_E_HIP, "rmdf_probe_shader_clock: the probe did not run");
    *mhz = (double)h[0] / (double)h[1] * 100.0;
    return RMDF_OK;
}

int rmdf_selftest_pinned_math(rmdf_ctx *ctx, uint64_t mismatches[7])
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!mismatches) return fail(ctx, RMDF_E_INVALID, "null output");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf d;
    HIP_TRY(ctx, dev_malloc(&d.p, 8 * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(d.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, launch_selftest_pinned_math((unsigned long long *)d.p, ctx->stream));
    return download(ctx, mismatches, d.p, 7 * sizeof(unsigned long long), ctx->stream);
}

int rmdf_selftest_shading_math(rmdf_ctx *ctx, uint64_t mismatches[5])
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!mismatches) return fail(ctx, RMDF_E_INVALID, "null output");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int face_w = 64;                                  // a cube map of its own: 6 faces of 64 x 64 texels plus the seam padding
    DevBuf d, t;
    HIP_TRY(ctx, dev_malloc(&d.p, 5 * sizeof(unsigned long long)));
    HIP_TRY(ctx, dev_malloc(&t.p, (size_t)6 * (face_w + 2) * (face_w + 2) * 8));
    HIP_TRY(ctx, hipMemsetAsync(d.p, 0, 5 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, launch_selftest_shading_math((unsigned long long *)d.p, t.p, face_w, host_fov_xs(), ctx->stream));
    return download(ctx, mismatches, d.p, 5 * sizeof(unsigned long long), ctx->stream);
}


#ifdef RMDF_XCHECK
int rmdf_debug_cornell_masks(int n, int brute_force, uint32_t *out)
{
    // host-only (no ctx, no device): the candidate grid of n^3 cells, built directly (brute_force != 0: every triangle measured in
    // every cell) or by halving cells from the 16^3 grid as rmdf_create does (tests compare the two)
    if (!out || n < CORNELL_GRID_N || n > 128 || n % CORNELL_GRID_N || ((n / CORNELL_GRID_N) & (n / CORNELL_GRID_N - 1))) return RMDF_E_INVALID;
    float tri[96 * 3];
    cornell_triangles(tri);
    if (brute_force) { cornell_grid(tri, n, out); return RMDF_OK; }
    std::vector<uint32_t> cur((size_t)CORNELL_GRID_N * CORNELL_GRID_N * CORNELL_GRID_N), next;
    cornell_grid(tri, CORNELL_GRID_N, cur.data());
    for (int m = CORNELL_GRID_N * 2; m <= n; m *= 2) { next.resize((size_t)m * m * m); cornell_grid(tri, m, next.data(), cur.data()); cur.swap(next); }
    memcpy(out, cur.data(), cur.size() * 4);
    return RMDF_OK;
}

int rmdf_debug_cornell_table(float *out, int *stride, int *bounds)
{
    if (stride) *stride = CORNELL_STRIDE;
    if (bounds) *bounds = CORNELL_BOUNDS;
    if (!out) return RMDF_OK;
    float tri[96 * 3], tab[CORNELL_TAB_FLOATS];
    cornell_triangles(tri);
    cornell_table(tri, tab);
    memcpy(out, tab, sizeof(float) * 32 * CORNELL_STRIDE);
    return RMDF_OK;
}

int rmdf_debug_cornell_bounds(float *out)
{
    // host-only: the compact bounds table behind the rows (32 x 8 floats: plane, bounding sphere) that the wave-uniform pruned estimate reads
    if (!out) return RMDF_E_INVALID;
    float tri[96 * 3], tab[CORNELL_TAB_FLOATS];
    cornell_triangles(tri);
    cornell_table(tri, tab);
    memcpy(out, tab + 32 * CORNELL_STRIDE, sizeof(float) * 32 * 8);
    return RMDF_OK;
}

int rmdf_debug_camera(int scene, float time, float cam[12], float *fov_xs)
{
    // host-only: the camera block every frame's kernel arguments carry (host_camera, host_fov_xs)
    if (!cam || scene < RMDF_FS_DE_CORNELL_BOX || scene > RMDF_FS_MB_GENERAL) return RMDF_E_INVALID;
    host_camera(scene, time, cam);
    if (fov_xs) *fov_xs = host_fov_xs();
    return RMDF_OK;
}

int rmdf_debug_hdr_decode(const uint8_t *file, size_t len, int *w, int *h, float *out, size_t cap_floats)
{
    // host-only: the Radiance reader behind rmdf_load_env_hdr (decode_hdr).  out may be null (size query).
    if (!file || !w || !h) return RMDF_E_INVALID;
    try {
        std::vector<float> rgb;
        std::string why;
        if (!decode_hdr(file, len, *w, *h, rgb, why)) return RMDF_E_IO;
        if (out) {
            if (cap_floats < rgb.size()) return RMDF_E_INVALID;
            memcpy(out, rgb.data(), rgb.size() * sizeof(float));
        }
    } catch (...) { return RMDF_E_NOMEM; }
    return RMDF_OK;
}

long rmdf_debug_hdr_encode(const float *rgb, int w, int h, uint8_t *out, size_t cap)
{
    // host-only: the cache-file image rmdf_load_env_hdr writes (encode_hdr); returns its length, or a negative error code
    if (!rgb || !out || w <= 0 || h <= 0) return RMDF_E_INVALID;
    try {
        std::vector<float> v(rgb, rgb + (size_t)w * h * 3);
        std::vector<uint8_t> file;
        encode_hdr(v, w, h, file);
        if (file.size() > cap) return RMDF_E_INVALID;
        memcpy(out, file.data(), file.size());
        return (long)file.size();
    } catch (...) { return RMDF_E_NOMEM; }
}

int rmdf_debug_cube_uv_table(int cw, float *out)
{
    // host-only: the table k_latlong_to_cube gathers through (cube_uv_table_host), 6 * cw * cw (u, v) pairs
    if (!out || cw < 1 || cw > 4096) return RMDF_E_INVALID;
    try {
        WorkPool none;                                         // no workers: evaluated on the calling thread
        std::vector<float> uv;
        cube_uv_table_host(none, cw, uv);
        memcpy(out, uv.data(), uv.size() * sizeof(float));
    } catch (...) { return RMDF_E_NOMEM; }
    return RMDF_OK;
}

int rmdf_debug_lobe_tables(int w, int h, float *lutT, float *tcs)
{
    // host-only: the cosine tables the lobe prefilter kernels read (lobe_tables_host): ceil(w / 64) * w * 64 and 2 * h floats
    if (!lutT || !tcs || w < 2 || h < 2 || w > 8192 || h > 4096) return RMDF_E_INVALID;
    try {
        WorkPool none;
        std::vector<float> a, b;
        lobe_tables_host(none, w, h, a, b);
        memcpy(lutT, a.data(), a.size() * sizeof(float));
        memcpy(tcs, b.data(), b.size() * sizeof(float));
    } catch (...) { return RMDF_E_NOMEM; }
    return RMDF_OK;
}

int rmdf_debug_march_stats(rmdf_ctx *ctx, int enable, uint64_t *out, int max_waves)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    const size_t cap = 32768;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (enable && !ctx->d_dbg) {
        HIP_TRY(ctx, dev_malloc((void **)&ctx->d_dbg, cap * 16 * sizeof(unsigned long long)));
        HIP_TRY(ctx, hipMemset(ctx->d_dbg, 0, cap * 16 * sizeof(unsigned long long)));
    }
    if (out && ctx->d_dbg) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        size_t n = (size_t)(max_waves < 0 ? 0 : max_waves);
        if (n > cap) n = cap;
        RMDF_TRY(download(ctx, out, ctx->d_dbg, n * 16 * sizeof(unsigned long long), ctx->stream));
        HIP_TRY(ctx, hipMemset(ctx->d_dbg, 0, cap * 16 * sizeof(unsigned long long)));
    }
    if (!enable && ctx->d_dbg) { (void)dev_free(ctx->d_dbg); ctx->d_dbg = nullptr; }
    return RMDF_OK;
}
#endif

int rmdf_save_png(const char *path, const uint32_t *fb_rgba8, int w, int h)
{
    // saveFrameBufferToPNG (FrameBuffer.hs:215-228): rows flipped (the frame buffer's row 0 is the bottom row, PNG
    // stores top-down) and alpha forced to 0xFF; 8-bit RGBA, no interlace, filter type 0 on every scanline
    if (!path || !fb_rgba8 || w <= 0 || h <= 0 || w > 65535 || h > 65535) return fail(nullptr, RMDF_E_INVALID, "rmdf_save_png: bad argument");
    const size_t stride = (size_t)w * 4 + 1;
    std::vector<uint8_t> raw(stride * (size_t)h);
    for (int y = 0; y < h; y++) {
        uint8_t *row = &raw[stride * (size_t)y];
        const uint8_t *src = (const uint8_t *)(fb_rgba8 + (size_t)(h - 1 - y) * w);
        row[0] = 0;
        memcpy(row + 1, src, (size_t)w * 4);
        for (int x = 0; x < w; x++) row[1 + 4 * x + 3] = 0xFF;
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) return fail(nullptr, RMDF_E_IO, "rmdf_save_png: deflate failed");
    FILE *f = fopen(path, "wb");
    if (!f) return fail(nullptr, RMDF_E_IO, std::string("cannot write ") + path);
    bool ok = true;
    auto chunk = [&](const char type[4], const uint8_t *data, size_t n) {
        uint8_t hd[8] = { (uint8_t)(n >> 24), (uint8_t)(n >> 16), (uint8_t)(n >> 8), (uint8_t)n,
                          (uint8_t)type[0], (uint8_t)type[1], (uint8_t)type[2], (uint8_t)type[3] };
        uLong crc = crc32(0L, hd + 4, 4);
        if (n) crc = crc32(crc, data, (uInt)n);
        const uint8_t tl[4] = { (uint8_t)(crc >> 24), (uint8_t)(crc >> 16), (uint8_t)(crc >> 8), (uint8_t)crc };
        ok = ok && fwrite(hd, 1, 8, f) == 8 && (n == 0 || fwrite(data, 1, n, f) == n) && fwrite(tl, 1, 4, f) == 4;
    };
    static const uint8_t sig[8] = { 0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A };
    ok = fwrite(sig, 1, 8, f) == 8;
    const uint8_t ihdr[13] = { (uint8_t)(w >> 24), (uint8_t)(w >> 16), (uint8_t)(w >> 8), (uint8_t)w,
                               (uint8_t)(h >> 24), (uint8_t)(h >> 16), (uint8_t)(h >> 8), (uint8_t)h, 8, 6, 0, 0, 0 };
    chunk("IHDR", ihdr, 13);
    chunk("IDAT", z.data(), (size_t)zlen);
    chunk("IEND", nullptr, 0);
    ok = (fclose(f) == 0) && ok;
    if (!ok) { remove(path); return fail(nullptr, RMDF_E_IO, std::string("short write to ") + path); }
    return RMDF_OK;
}

int rmdf_register_host_buffer(rmdf_ctx *ctx, void *ptr, size_t bytes)
{
    // Kept for hosts built against the round-2 .. round-4 header.  It used to hipHostRegister the range so that the render kernel could
    // store into it; since round 5 it only remembers the range (the library creates no GPU mapping of caller memory: render_common), and
    // a whole-frame call into it takes the same staged path as any other pointer.
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!ptr || bytes == 0) return fail(ctx, RMDF_E_INVALID, "rmdf_register_host_buffer: bad argument");
    RMDF_GUARD_BEGIN
    for (auto &r : ctx->host_regs) if (r.host == (char *)ptr && r.bytes == bytes) return RMDF_OK;
    ctx->host_regs.push_back(rmdf_ctx::HostReg{ (char *)ptr, bytes });
    return RMDF_OK;
    RMDF_GUARD_END(ctx)
}

int rmdf_unregister_host_buffer(rmdf_ctx *ctx, void *ptr)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    for (size_t i = 0; i < ctx->host_regs.size(); i++)
        if (ctx->host_regs[i].host == (char *)ptr) {
            ctx->host_regs.erase(ctx->host_regs.begin() + (long)i);
            return RMDF_OK;
        }
    return fail(ctx, RMDF_E_INVALID, "rmdf_unregister_host_buffer: not registered");
}

int rmdf_device_malloc(rmdf_ctx *ctx, size_t bytes, void **d_ptr)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!d_ptr || bytes == 0) return fail(ctx, RMDF_E_INVALID, "rmdf_device_malloc: bad argument");
    *d_ptr = nullptr;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, dev_malloc(d_ptr, bytes));
    return RMDF_OK;
}

int rmdf_device_free(rmdf_ctx *ctx, void *d_ptr)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!d_ptr) return RMDF_OK;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, dev_free(d_ptr));
    return RMDF_OK;
}

int rmdf_copy_to_host(rmdf_ctx *ctx, void *host_dst, const void *d_src, size_t bytes, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!host_dst || !d_src) return fail(ctx, RMDF_E_INVALID, "rmdf_copy_to_host: null pointer");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = stream ? (hipStream_t)stream : ctx->stream;
    return download(ctx, host_dst, d_src, bytes, st);
}

int rmdf_synchronize(rmdf_ctx *ctx, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(stream ? (hipStream_t)stream : ctx->stream));
    return RMDF_OK;
}

int rmdf_device_info(rmdf_ctx *ctx, char *name, int name_len, int *compute_units)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (name && name_len > 0) snprintf(name, (size_t)name_len, "%s", ctx->dev_name);
    if (compute_units) *compute_units = ctx->cus;
    return RMDF_OK;
}

}  // extern "C"
