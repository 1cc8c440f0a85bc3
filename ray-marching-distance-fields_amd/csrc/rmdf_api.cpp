// rmdf_api.cpp -- the C ABI of librmdf.so (include/rmdf.h): host-side counterpart of
// ShaderRendering.hs (withShaderRenderer / drawShaderTile) for the HIP renderer.
// No CPU rendering path exists here: every pixel comes from the gfx950 kernels.
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <new>
#include <string>
#include <vector>

#include "../../include/rmdf.h"
#include "rmdf_internal.hpp"

using namespace rmdf;

namespace {

thread_local std::string g_create_error;

struct CubeSlot {
    uint2 *d_texels = nullptr;
    int    W = 0;
};

#define RMDF_MAX_ORDER_STREAMS 32
struct OrderState {
    hipStream_t stream = nullptr;
    unsigned   *d_cost = nullptr, *d_order = nullptr;
    int         cap = 0, n = 0;
    int         key[10] = { -1, 0, 0, 0, 0, 0, 0, 0, 0, 0 };
    bool        used = false, valid = false;
    unsigned    last_use = 0;
};

}  // namespace

struct rmdf_ctx {
    int          device = 0;
    hipStream_t  stream = nullptr;
    float       *d_cornell = nullptr;
    float       *d_cornell_tab = nullptr;
    CubeSlot     env[RMDF_ENV_SLOTS];
    // frame latched on the first tile (ShaderRendering.hs:162-176)
    int          w = 0, h = 0, max_steps = 128;
    float        time = 0.0f;
    bool         latched = false;
    // accumulating frame (device)
    uint32_t    *d_rgba8 = nullptr;
    float4      *d_rgba_f32 = nullptr;
    uint16_t    *d_steps = nullptr;
    uint16_t    *d_iters = nullptr;
    size_t       cap_px = 0;
    // G-buffer + work counter of the two-kernel Mandelbulb path
    float4      *d_gbuf_nao = nullptr;
    unsigned    *d_gbuf_meta = nullptr;
    int         *d_work_counter = nullptr;
    int         *d_hit_list = nullptr;
    // cost-ordered dispatch state of the nested-loop kernel (previous frame's per-strip costs), one per stream
    // that renders: frames in flight on different streams (pipelined rendering) must not share the tables
    // host buffers registered for direct GPU writes (rmdf_register_host_buffer)
    struct HostReg { char *host; size_t bytes; char *dev; };
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
    OrderState   orders[RMDF_MAX_ORDER_STREAMS];
    unsigned     order_tick = 0;
    unsigned long long *d_dbg = nullptr;   // per-wave march diagnostics (rmdf_debug_march_stats)
    size_t       gbuf_cap = 0;
    int          flags = 0;            // rmdf_config.reserved[0]
    std::string  err;
    char         dev_name[256] = { 0 };
    int          cus = 0;
};

namespace {

int fail(rmdf_ctx *ctx, int code, const std::string &msg)
{
    if (ctx) ctx->err = msg; else g_create_error = msg;
    return code;
}

#define HIP_TRY(ctx, expr)                                                                        \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess)                                                                     \
            return fail(ctx, RMDF_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
    } while (0)

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
        c = hv3{ sinf(time / 2.0f) * 0.4f, cosf(time / 2.0f) * 0.4f, -2.0f };
    } else {
        c = hv3{ sinf(time / 3.0f), cosf(time / 4.0f), cosf(time / 3.0f) };
        hv3 nrm = hnormalize(c);
        c = hv3{ nrm.x * 2.414213562373095f, nrm.y * 2.414213562373095f, nrm.z * 2.414213562373095f };
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
    float hfov = (45.0f * 1.5f) * 0.017453292519943295f;   // radians(45.0 * 1.5)
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
        // pruning bounds of de_cornell_box_table (double arithmetic, rounded once): unit normal, plane offset,
        // bounding sphere about the centroid
        {
            const double a[3] = { e0.x, e0.y, e0.z }, b[3] = { e1.x, e1.y, e1.z };
            double n[3] = { a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0] };
            const double nl = sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            for (int k = 0; k < 3; k++) n[k] /= nl;
            t[26] = (float)n[0]; t[27] = (float)n[1]; t[28] = (float)n[2];
            t[29] = (float)(n[0] * v0[0] + n[1] * v0[1] + n[2] * v0[2]);
            double c[3], R = 0.0;
            for (int k = 0; k < 3; k++) c[k] = ((double)v0[k] + v1[k] + v2[k]) / 3.0;
            for (int k = 0; k < 3; k++) t[30 + k] = (float)c[k];
            const float *vs[3] = { v0, v1, v2 };
            for (int j = 0; j < 3; j++) {
                double d2 = 0.0;
                for (int k = 0; k < 3; k++) { const double d = (double)vs[j][k] - (double)t[30 + k]; d2 += d * d; }
                if (sqrt(d2) > R) R = sqrt(d2);
            }
            t[33] = (float)(R * (1.0 + 1e-6));
            t[34] = t[35] = 0.0f;
        }
    }
    // compact copy of the pruning bounds behind the rows (read four triangles at a time)
    for (int i = 0; i < 32; i++)
        for (int k = 0; k < 8; k++) tab[32 * CORNELL_STRIDE + i * 8 + k] = tab[i * CORNELL_STRIDE + 26 + k];
}

int ensure_frame(rmdf_ctx *ctx, int w, int h)
{
    size_t npx = (size_t)w * (size_t)h;
    if (npx <= ctx->cap_px && ctx->d_rgba8) return RMDF_OK;
    if (ctx->d_rgba8) (void)hipFree(ctx->d_rgba8);
    if (ctx->d_rgba_f32) (void)hipFree(ctx->d_rgba_f32);
    if (ctx->d_steps) (void)hipFree(ctx->d_steps);
    if (ctx->d_iters) (void)hipFree(ctx->d_iters);
    ctx->d_rgba8 = nullptr; ctx->d_rgba_f32 = nullptr; ctx->d_steps = nullptr; ctx->d_iters = nullptr;
    ctx->cap_px = 0;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_rgba8, npx * 4));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_rgba_f32, npx * 16));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_steps, npx * 2));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_iters, npx * 2));
    ctx->cap_px = npx;
    return RMDF_OK;
}

// resizeFrameBuffer clears the new texture to opaque black (FrameBuffer.hs:109-111)
int clear_frame(rmdf_ctx *ctx, int w, int h)
{
    size_t npx = (size_t)w * (size_t)h;
    HIP_TRY(ctx, launch_fill_u32(ctx->d_rgba8, 0xff000000u, npx, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_rgba_f32, 0, npx * 16, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_steps, 0, npx * 2, ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_iters, 0, npx * 2, ctx->stream));
    return RMDF_OK;
}

int ensure_gbuf(rmdf_ctx *ctx, int w, int h)
{
    const size_t gw = (size_t)((w + 1) & ~1), gh = (size_t)((h + 1) & ~1);
    const size_t need = gw * gh;
    if (!ctx->d_work_counter) HIP_TRY(ctx, hipMalloc((void **)&ctx->d_work_counter, 256));
    if (need <= ctx->gbuf_cap) return RMDF_OK;
    if (ctx->d_gbuf_nao) (void)hipFree(ctx->d_gbuf_nao);
    if (ctx->d_gbuf_meta) (void)hipFree(ctx->d_gbuf_meta);
    if (ctx->d_hit_list) (void)hipFree(ctx->d_hit_list);
    ctx->d_gbuf_nao = nullptr; ctx->d_gbuf_meta = nullptr; ctx->d_hit_list = nullptr; ctx->gbuf_cap = 0;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_gbuf_nao, need * sizeof(float4)));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_gbuf_meta, need * sizeof(unsigned)));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_hit_list, need * sizeof(int)));
    ctx->gbuf_cap = need;
    return RMDF_OK;
}

// scene dispatch: the Mandelbulb has two schedules of the same per-ray arithmetic; the default is the
// fastest measured one, the other stays selectable for A/B measurements and tests
int launch_scene(rmdf_ctx *ctx, int scene, const FrameParams &p, hipStream_t stream)
{
    if (scene == RMDF_FS_MB_POWER8 && ctx->d_dbg && getenv("RMDF_NESTED_STATS")) {
        HIP_TRY(ctx, launch_march_stats(p, stream));
        return RMDF_OK;
    }
    if (ctx->flags & RMDF_FLAG_PIPELINE) {
        HIP_TRY(ctx, launch_render_pipeline(scene, p, stream, ctx->cus));
        return RMDF_OK;
    }
    if (scene == RMDF_FS_MB_POWER8 && (ctx->flags & RMDF_FLAG_FLAT_MARCH)) {
        HIP_TRY(ctx, launch_render_mb8(p, stream, ctx->cus));
        return RMDF_OK;
    }
    // Nested-loop kernel.  Its run time is set by the strips that hold the longest rays (a 226-step ray is a
    // ~0.4 ms serial chain), so large launches dispatch the strips that were most expensive in the previous
    // frame of the same configuration first (temporal coherence; `time` is deliberately not part of the key).
    // The table only permutes which workgroup renders which strip: the image does not depend on it.
    FrameParams q = p;
    const int nblk = render_grid_blocks(scene, p);
    const bool want = nblk >= 1024 && !(ctx->flags & RMDF_FLAG_RASTER_ORDER);
    if (want) {
        // the table set of this stream (least recently used one is recycled when more than 8 streams render)
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
            if (os->d_cost) (void)hipFree(os->d_cost);
            if (os->d_order) (void)hipFree(os->d_order);
            os->d_cost = os->d_order = nullptr; os->cap = 0; os->valid = false;
            HIP_TRY(ctx, hipMalloc((void **)&os->d_cost, (size_t)nblk * 4));
            HIP_TRY(ctx, hipMalloc((void **)&os->d_order, (size_t)nblk * 4));
            os->cap = nblk;
        }
        const int key[10] = { scene, p.w, p.h, p.x0, p.y0, p.x1, p.y1, p.max_steps,
                              p.n_shard_tiles, p.shard_key };
        const bool same = os->valid && os->n == nblk && memcmp(key, os->key, sizeof key) == 0;
        q.block_cost = os->d_cost;
        q.block_order = same ? os->d_order : nullptr;
        { static int ps = -1; if (ps < 0) { const char *e = getenv("RMDF_PRIO_STRIPS"); ps = e ? atoi(e) : 256; } q.prio_strips = ps; }
        HIP_TRY(ctx, launch_render(scene, q, stream));
        HIP_TRY(ctx, launch_order_blocks(os->d_cost, nblk, os->d_order, stream));
        memcpy(os->key, key, sizeof key);
        os->n = nblk; os->valid = true;
    } else {
        HIP_TRY(ctx, launch_render(scene, q, stream));
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
    if (w <= 0 || h <= 0 || w > 32768 || h > 32768) return fail(ctx, RMDF_E_INVALID, "bad frame size");
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
    p.max_steps = max_steps <= 0 ? 128 : max_steps;
    p.env_refl = CubeDev{ ctx->env[RMDF_ENV_REFLECTION].d_texels, ctx->env[RMDF_ENV_REFLECTION].W };
    p.env_cos1 = CubeDev{ ctx->env[RMDF_ENV_COS_1].d_texels, ctx->env[RMDF_ENV_COS_1].W };
    p.env_cos8 = CubeDev{ ctx->env[RMDF_ENV_COS_8].d_texels, ctx->env[RMDF_ENV_COS_8].W };
    p.cornell = ctx->d_cornell;
    // measurement knobs are read once per process
    static int mg = -1, mg_forced = 0;
    if (mg < 0) { const char *e = getenv("RMDF_MERGE"); mg_forced = e != nullptr; mg = e ? atoi(e) : 32; }
    p.merge_stragglers = (ctx->flags & RMDF_FLAG_NO_MERGE) ? 0 : mg;
    // measured (current build): pooling pays for both Mandelbulbs (+5 %, +8 %) and the test scene (+8 %); it costs 6 % for
    // the Cornell box, whose distance estimate has the same cost for every ray
    if (scene == RMDF_FS_DE_CORNELL_BOX && !mg_forced) p.merge_stragglers = 0;
    p.cornell_tab = ctx->d_cornell_tab;
    p.cornell_prune = (ctx->flags & RMDF_FLAG_NO_PRUNE) ? 0 : 1;
    { static int skip = -1; if (skip < 0) { const char *e = getenv("RMDF_DBG_SKIP"); skip = e ? atoi(e) : 0; } p.dbg_skip = skip; }
    int rc = ensure_gbuf(ctx, w, h);
    if (rc != RMDF_OK) return rc;
    p.gbuf_nao = ctx->d_gbuf_nao; p.gbuf_meta = ctx->d_gbuf_meta; p.gw = (w + 1) & ~1;
    p.work_counter = ctx->d_work_counter;
    p.hit_count = ctx->d_work_counter + 1;
    p.hit_list = ctx->d_hit_list;
    p.tile_order = nullptr;
    p.dbg = ctx->d_dbg;
    return RMDF_OK;
}

bool read_file(const char *path, std::vector<uint8_t> &out)
{
    FILE *f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return false; }
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
bool decode_hdr(const std::vector<uint8_t> &file, int &w, int &h, std::vector<float> &rgb, std::string &why)
{
    size_t pos = 0, len = file.size();
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

bool write_hdr(const std::string &path, const std::vector<float> &rgb, int w, int h)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", h, w);
    std::vector<uint8_t> px((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) float_to_rgbe(&rgb[i * 3], &px[i * 4]);
    bool ok = fwrite(px.data(), 1, px.size(), f) == px.size();
    ok = (fclose(f) == 0) && ok;
    if (!ok) remove(path.c_str());            // removeFile on failure, ShaderRendering.hs:146-148
    return ok;
}

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

int set_env_from_device_faces(rmdf_ctx *ctx, int slot, const float *d_faces, int W)
{
    uint2 *d_padded = nullptr;
    size_t n = (size_t)6 * (W + 2) * (W + 2);
    HIP_TRY(ctx, hipMalloc((void **)&d_padded, n * sizeof(uint2)));
    hipError_t e = launch_cube_upload(d_faces, W, d_padded, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { (void)hipFree(d_padded); return fail(ctx, RMDF_E_HIP, std::string("cube upload: ") + hipGetErrorString(e)); }
    if (ctx->env[slot].d_texels) (void)hipFree(ctx->env[slot].d_texels);
    ctx->env[slot].d_texels = d_padded;
    ctx->env[slot].W = W;
    return RMDF_OK;
}

int render_common(rmdf_ctx *ctx, int scene, int tile_idx, int w, int h, double time, int max_steps,
                  uint32_t *out_rgba8, float *out_rgba_f32, uint16_t *out_steps, uint16_t *out_iters)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const bool whole = tile_idx < 0;
    const bool first = whole || rmdf_is_tile_idx_first_tile(tile_idx);
    // latch on the first tile (ShaderRendering.hs:162-176).  A size change in the middle
    // of a tiled frame re-latches (the reference would resize its FBO texture and clear it).
    if (first || !ctx->latched || w != ctx->w || h != ctx->h) {
        FrameParams probe;
        int rc = fill_params(ctx, scene, w, h, (float)time, max_steps, probe);
        if (rc != RMDF_OK) return rc;
        const bool resized = (w != ctx->w || h != ctx->h || !ctx->d_rgba8);
        rc = ensure_frame(ctx, w, h);
        if (rc != RMDF_OK) return rc;
        if (resized) { rc = clear_frame(ctx, w, h); if (rc != RMDF_OK) return rc; }
        ctx->w = w; ctx->h = h; ctx->time = (float)time; ctx->max_steps = max_steps <= 0 ? 128 : max_steps;
        ctx->latched = true;
    }
    FrameParams p;
    int rc = fill_params(ctx, scene, ctx->w, ctx->h, ctx->time, ctx->max_steps, p);
    if (rc != RMDF_OK) return rc;
    if (whole) { p.x0 = 0; p.y0 = 0; p.x1 = ctx->w; p.y1 = ctx->h; }
    else tile_rect_host(tile_idx, ctx->w, ctx->h, &p.x0, &p.y0, &p.x1, &p.y1);
    p.rgba8 = ctx->d_rgba8; p.rgba_f32 = ctx->d_rgba_f32; p.steps = ctx->d_steps; p.iters = ctx->d_iters;
    const size_t npx = (size_t)ctx->w * ctx->h;
    // Whole-frame call into a registered host buffer: the render kernel stores the RGBA8 rows into it directly (next to
    // the library's own accumulating frame), so the PCIe transfer overlaps the render instead of following it.  Only the
    // default nested-loop kernel knows the mirror pointer.
    bool direct = false;
    if (whole && out_rgba8 && !(ctx->flags & (RMDF_FLAG_PIPELINE | RMDF_FLAG_FLAT_MARCH))) {
        for (auto &r : ctx->host_regs)
            if ((char *)out_rgba8 >= r.host && (char *)out_rgba8 + npx * 4 <= r.host + r.bytes) {
                p.rgba8_mirror = (uint32_t *)(r.dev + ((char *)out_rgba8 - r.host));
                direct = true;
                break;
            }
    }
    rc = launch_scene(ctx, scene, p, ctx->stream);
    if (rc != RMDF_OK) return rc;
    if (out_rgba8 && !direct) HIP_TRY(ctx, hipMemcpyAsync(out_rgba8, ctx->d_rgba8, npx * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (out_rgba_f32) HIP_TRY(ctx, hipMemcpyAsync(out_rgba_f32, ctx->d_rgba_f32, npx * 16, hipMemcpyDeviceToHost, ctx->stream));
    if (out_steps) HIP_TRY(ctx, hipMemcpyAsync(out_steps, ctx->d_steps, npx * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (out_iters) HIP_TRY(ctx, hipMemcpyAsync(out_iters, ctx->d_iters, npx * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RMDF_OK;
}

}  // namespace

namespace rmdf {
void tile_rect_host(int tile_idx, int w, int h, int *x0, int *y0, int *x1, int *y1)
{
    // ShaderRendering.hs:183-193, centre-inside rasterisation (see rmdf_kernels.hip)
    int midx = tile_idx % 64;
    int tx = midx % 8, ty = midx / 8;
    *x0 = (2 * tx * w + 7) / 16;
    *x1 = (2 * (tx + 1) * w + 7) / 16;
    *y0 = (2 * ty * h + 7) / 16;
    *y1 = (2 * (ty + 1) * h + 7) / 16;
}
}  // namespace rmdf

extern "C" {

int rmdf_create(rmdf_ctx **out, const rmdf_config *cfg)
{
    if (!out) return fail(nullptr, RMDF_E_INVALID, "null out pointer");
    *out = nullptr;
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
    rmdf_ctx *ctx = new (std::nothrow) rmdf_ctx();
    if (!ctx) return fail(nullptr, RMDF_E_NOMEM, "out of host memory");
    ctx->device = dev;
    ctx->flags = cfg ? cfg->reserved[0] : 0;
    ctx->cus = prop.multiProcessorCount;
    snprintf(ctx->dev_name, sizeof ctx->dev_name, "%s (%s)", prop.name, prop.gcnArchName);
    float tri[96 * 3];
    cornell_triangles(tri);
    float tab[CORNELL_TAB_FLOATS];
    cornell_table(tri, tab);
    if ((e = hipSetDevice(dev)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipMalloc((void **)&ctx->d_cornell, sizeof tri)) != hipSuccess ||
        (e = hipMemcpy(ctx->d_cornell, tri, sizeof tri, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMalloc((void **)&ctx->d_cornell_tab, sizeof tab)) != hipSuccess ||
        (e = hipMemcpy(ctx->d_cornell_tab, tab, sizeof tab, hipMemcpyHostToDevice)) != hipSuccess) {
        std::string msg = std::string("device init: ") + hipGetErrorString(e);
        rmdf_destroy(ctx);
        return fail(nullptr, RMDF_E_HIP, msg);
    }
    *out = ctx;
    return RMDF_OK;
}

void rmdf_destroy(rmdf_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (auto &s : ctx->env) if (s.d_texels) (void)hipFree(s.d_texels);
    if (ctx->d_cornell) (void)hipFree(ctx->d_cornell);
    if (ctx->d_cornell_tab) (void)hipFree(ctx->d_cornell_tab);
    if (ctx->d_rgba8) (void)hipFree(ctx->d_rgba8);
    if (ctx->d_rgba_f32) (void)hipFree(ctx->d_rgba_f32);
    if (ctx->d_steps) (void)hipFree(ctx->d_steps);
    if (ctx->d_iters) (void)hipFree(ctx->d_iters);
    if (ctx->d_gbuf_nao) (void)hipFree(ctx->d_gbuf_nao);
    if (ctx->d_gbuf_meta) (void)hipFree(ctx->d_gbuf_meta);
    if (ctx->d_hit_list) (void)hipFree(ctx->d_hit_list);
    if (ctx->d_work_counter) (void)hipFree(ctx->d_work_counter);
    if (ctx->d_dbg) (void)hipFree(ctx->d_dbg);
    (void)hipDeviceSynchronize();              // caller streams may still be running launches that use the tables
    for (auto &r : ctx->host_regs) (void)hipHostUnregister(r.host);
    for (auto &o : ctx->orders) {
        if (o.d_cost) (void)hipFree(o.d_cost);
        if (o.d_order) (void)hipFree(o.d_order);
    }
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *rmdf_last_error(const rmdf_ctx *ctx)
{
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

int rmdf_is_tile_idx_first_tile(int idx) { return idx % RMDF_N_TILES == 0; }
int rmdf_is_tile_idx_last_tile(int idx) { return idx % RMDF_N_TILES == RMDF_N_TILES - 1; }

int rmdf_set_env_cube(rmdf_ctx *ctx, int slot, const float *faces_rgb, int face_w)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (slot < 0 || slot >= RMDF_ENV_SLOTS || !faces_rgb || face_w < 1 || face_w > 8192)
        return fail(ctx, RMDF_E_INVALID, "rmdf_set_env_cube: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf faces;
    size_t bytes = (size_t)6 * face_w * face_w * 3 * sizeof(float);
    HIP_TRY(ctx, hipMalloc(&faces.p, bytes));
    HIP_TRY(ctx, hipMemcpyAsync(faces.p, faces_rgb, bytes, hipMemcpyHostToDevice, ctx->stream));
    return set_env_from_device_faces(ctx, slot, (const float *)faces.p, face_w);
}

int rmdf_set_env_latlong(rmdf_ctx *ctx, int slot, const float *rgb, int w, int h)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (slot < 0 || slot >= RMDF_ENV_SLOTS || !rgb || w < 6 || h < 2 || w > 65536 || h > 32768)
        return fail(ctx, RMDF_E_INVALID, "rmdf_set_env_latlong: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int cw = w / 3;
    DevBuf ll, faces;
    size_t ll_bytes = (size_t)w * h * 3 * sizeof(float), f_bytes = (size_t)6 * cw * cw * 3 * sizeof(float);
    HIP_TRY(ctx, hipMalloc(&ll.p, ll_bytes));
    HIP_TRY(ctx, hipMalloc(&faces.p, f_bytes));
    HIP_TRY(ctx, hipMemcpyAsync(ll.p, rgb, ll_bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, launch_latlong_to_cube((const float *)ll.p, w, h, (float *)faces.p, ctx->stream));
    return set_env_from_device_faces(ctx, slot, (const float *)faces.p, cw);
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
    HIP_TRY(ctx, hipMemcpy(out, ctx->env[slot].d_texels, (size_t)6 * (W + 2) * (W + 2) * 8, hipMemcpyDeviceToHost));
    return RMDF_OK;
}

int rmdf_resize_latlong(rmdf_ctx *ctx, const float *rgb, int w, int h, int dstw, float *out, int *dsth)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!rgb || w < 2 || h < 2 || dstw < 1 || !dsth) return fail(ctx, RMDF_E_INVALID, "rmdf_resize_latlong: bad argument");
    // dsth = round (srch / srcw * dstw) in Float, Haskell round = half-to-even (HDREnvMap.hs:173)
    const int dh = (int)rintf((float)h / (float)w * (float)dstw);
    *dsth = dh;
    if (!out) return RMDF_OK;
    if (dh < 1) return fail(ctx, RMDF_E_INVALID, "destination height < 1");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf src, dst;
    size_t sb = (size_t)w * h * 12, db = (size_t)dstw * dh * 12;
    HIP_TRY(ctx, hipMalloc(&src.p, sb));
    HIP_TRY(ctx, hipMalloc(&dst.p, db));
    HIP_TRY(ctx, hipMemcpyAsync(src.p, rgb, sb, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, launch_resize_latlong((const float *)src.p, w, h, dstw, dh, (float *)dst.p, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(out, dst.p, db, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RMDF_OK;
}

int rmdf_prefilter_env(rmdf_ctx *ctx, const float *rgb, int w, int h, float power, float *out)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!rgb || !out || w < 2 || h < 2 || w > 600) return fail(ctx, RMDF_E_INVALID, "rmdf_prefilter_env: bad argument (2 <= w <= 600)");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf src, dst;
    size_t b = (size_t)w * h * 12;
    HIP_TRY(ctx, hipMalloc(&src.p, b));
    HIP_TRY(ctx, hipMalloc(&dst.p, b));
    HIP_TRY(ctx, hipMemcpyAsync(src.p, rgb, b, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, launch_prefilter((const float *)src.p, w, h, power, (float *)dst.p, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(out, dst.p, b, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RMDF_OK;
}

int rmdf_load_env_hdr(rmdf_ctx *ctx, const char *path)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!path) return fail(ctx, RMDF_E_INVALID, "null path");
    std::vector<uint8_t> file;
    if (!read_file(path, file)) return fail(ctx, RMDF_E_IO, std::string("cannot read ") + path);
    int w = 0, h = 0;
    std::vector<float> refl;
    std::string why;
    if (!decode_hdr(file, w, h, refl, why)) return fail(ctx, RMDF_E_IO, std::string(path) + ": " + why);
    // powers / cache file names, ShaderRendering.hs:71-75 (`show pow` of a Float: "1.0")
    static const struct { int slot; const char *suffix; float power; } kPow[4] = {
        { RMDF_ENV_COS_1, "1.0", 1.0f }, { RMDF_ENV_COS_8, "8.0", 8.0f },
        { RMDF_ENV_COS_64, "64.0", 64.0f }, { RMDF_ENV_COS_512, "512.0", 512.0f } };
    std::string stem(path);
    size_t dot = stem.find_last_of('.'), slash = stem.find_last_of('/');
    if (dot != std::string::npos && (slash == std::string::npos || dot > slash)) stem = stem.substr(0, dot);
    std::vector<float> resized;
    int rw = 256, rh = 0;
    for (const auto &pw : kPow) {
        std::string fn = stem + "_cache_pow_" + pw.suffix + ".hdr";
        if (!file_exists(fn)) {
            // buildPreConvolvedHDREnvMapCache, ShaderRendering.hs:131-149
            if (resized.empty()) {
                int rc = rmdf_resize_latlong(ctx, refl.data(), w, h, rw, nullptr, &rh);
                if (rc != RMDF_OK) return rc;
                resized.resize((size_t)rw * rh * 3);
                rc = rmdf_resize_latlong(ctx, refl.data(), w, h, rw, resized.data(), &rh);
                if (rc != RMDF_OK) return rc;
            }
            std::vector<float> conv((size_t)rw * rh * 3);
            int rc = rmdf_prefilter_env(ctx, resized.data(), rw, rh, pw.power, conv.data());
            if (rc != RMDF_OK) return rc;
            if (!write_hdr(fn, conv, rw, rh)) return fail(ctx, RMDF_E_IO, "cannot write cache file " + fn);
        }
        std::vector<uint8_t> cf;
        if (!read_file(fn.c_str(), cf)) return fail(ctx, RMDF_E_IO, "cannot read cache file " + fn);
        int cw = 0, chh = 0;
        std::vector<float> cimg;
        if (!decode_hdr(cf, cw, chh, cimg, why)) return fail(ctx, RMDF_E_IO, fn + ": " + why);
        int rc = rmdf_set_env_latlong(ctx, pw.slot, cimg.data(), cw, chh);
        if (rc != RMDF_OK) return rc;
    }
    return rmdf_set_env_latlong(ctx, RMDF_ENV_REFLECTION, refl.data(), w, h);
}

int rmdf_render_tile(rmdf_ctx *ctx, int scene, int tile_idx, int w, int h, double time, int max_steps,
                     uint32_t *out_rgba8)
{
    return render_common(ctx, scene, tile_idx, w, h, time, max_steps, out_rgba8, nullptr, nullptr, nullptr);
}

int rmdf_render_tile_ex(rmdf_ctx *ctx, int scene, int tile_idx, int w, int h, double time, int max_steps,
                        uint32_t *out_rgba8, float *out_rgba_f32, uint16_t *out_steps, uint16_t *out_iters)
{
    return render_common(ctx, scene, tile_idx, w, h, time, max_steps, out_rgba8, out_rgba_f32, out_steps, out_iters);
}

int rmdf_render_rect_device(rmdf_ctx *ctx, int scene, int w, int h, double time, int max_steps,
                            int x0, int y0, int x1, int y1,
                            void *d_rgba8, void *d_rgba_f32, void *d_steps, void *d_iters, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    FrameParams p;
    int rc = fill_params(ctx, scene, w, h, (float)time, max_steps, p);
    if (rc != RMDF_OK) return rc;
    if (x0 < 0 || y0 < 0 || x1 > w || y1 > h || x0 > x1 || y0 > y1) return fail(ctx, RMDF_E_INVALID, "bad rectangle");
    p.x0 = x0; p.y0 = y0; p.x1 = x1; p.y1 = y1;
    p.rgba8 = (uint32_t *)d_rgba8; p.rgba_f32 = (float4 *)d_rgba_f32; p.steps = (uint16_t *)d_steps; p.iters = (uint16_t *)d_iters;
    return launch_scene(ctx, scene, p, stream ? (hipStream_t)stream : ctx->stream);
}

int rmdf_render_shard_device(rmdf_ctx *ctx, int scene, int w, int h, double time, int max_steps,
                             int rank, int nranks, void *d_packed_rgba8, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks || !d_packed_rgba8)
        return fail(ctx, RMDF_E_INVALID, "rmdf_render_shard_device: bad rank / nranks / buffer");
    if (w % 8 || h % 8) return fail(ctx, RMDF_E_INVALID, "tile sharding needs w and h divisible by 8");
    FrameParams p;
    int rc = fill_params(ctx, scene, w, h, (float)time, max_steps, p);
    if (rc != RMDF_OK) return rc;
    ensure_deal(ctx, nranks);
    p.n_shard_tiles = ctx->deal_count[rank];
    memcpy(p.shard_tile, ctx->deal_tiles[rank], sizeof p.shard_tile);
    p.shard_key = (int)(ctx->shard_cost_gen << 16) + rank * 256 + nranks;
    p.rgba8 = (uint32_t *)d_packed_rgba8;
    return launch_scene(ctx, scene, p, stream ? (hipStream_t)stream : ctx->stream);
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
    HIP_TRY(ctx, hipMalloc(&steps.p, npx * 2));
    HIP_TRY(ctx, hipMalloc(&iters.p, npx * 2));
    p.x0 = 0; p.y0 = 0; p.x1 = pw; p.y1 = ph;
    p.steps = (uint16_t *)steps.p; p.iters = (uint16_t *)iters.p;
    FrameParams q = p;
    q.block_cost = nullptr; q.block_order = nullptr;
    HIP_TRY(ctx, launch_render(scene, q, ctx->stream));
    std::vector<uint16_t> hs(npx), hi(npx);
    HIP_TRY(ctx, hipMemcpyAsync(hs.data(), steps.p, npx * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(hi.data(), iters.p, npx * 2, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
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
}

int rmdf_assemble_shards_device(rmdf_ctx *ctx, int w, int h, int nranks, const void *d_gathered,
                                void *d_frame_rgba8, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (nranks < 1 || nranks > 64 || !d_gathered || !d_frame_rgba8 || w % 8 || h % 8 || w <= 0 || h <= 0)
        return fail(ctx, RMDF_E_INVALID, "rmdf_assemble_shards_device: bad argument");
    ensure_deal(ctx, nranks);
    HIP_TRY(ctx, launch_assemble_shards((const uint32_t *)d_gathered, (uint32_t *)d_frame_rgba8, w, h, nranks, ctx->deal_where,
                                        stream ? (hipStream_t)stream : ctx->stream));
    return RMDF_OK;
}

int rmdf_resolve_box2_device(rmdf_ctx *ctx, const void *d_src_rgba8, int sw, int sh, void *d_dst_rgba8, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!d_src_rgba8 || !d_dst_rgba8 || sw <= 0 || sh <= 0 || (sw & 1) || (sh & 1))
        return fail(ctx, RMDF_E_INVALID, "rmdf_resolve_box2_device: needs even, positive source sizes");
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
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int sw = w << levels, sh = h << levels;
    FrameParams p;
    int rc = fill_params(ctx, scene, sw, sh, (float)time, max_steps, p);
    if (rc != RMDF_OK) return rc;
    DevBuf a, b;
    HIP_TRY(ctx, hipMalloc(&a.p, (size_t)sw * sh * 4));
    if (levels > 0) HIP_TRY(ctx, hipMalloc(&b.p, (size_t)(sw / 2) * (sh / 2) * 4));
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
    HIP_TRY(ctx, hipMemcpyAsync(out_rgba8, cur, (size_t)w * h * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RMDF_OK;
}

int rmdf_selftest_exact_math(rmdf_ctx *ctx, uint64_t mismatches[5])
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!mismatches) return fail(ctx, RMDF_E_INVALID, "null output");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf d;
    HIP_TRY(ctx, hipMalloc(&d.p, 8 * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(d.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, launch_selftest_exact_math((unsigned long long *)d.p, ctx->d_cornell_tab, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(mismatches, d.p, 5 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RMDF_OK;
}

int rmdf_selftest_pinned_math(rmdf_ctx *ctx, uint64_t mismatches[7])
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!mismatches) return fail(ctx, RMDF_E_INVALID, "null output");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevBuf d;
    HIP_TRY(ctx, hipMalloc(&d.p, 8 * sizeof(unsigned long long)));
    HIP_TRY(ctx, hipMemsetAsync(d.p, 0, 8 * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx, launch_selftest_pinned_math((unsigned long long *)d.p, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(mismatches, d.p, 7 * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return RMDF_OK;
}

int rmdf_debug_march_stats(rmdf_ctx *ctx, int enable, uint64_t *out, int max_waves)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    const size_t cap = 32768;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (enable && !ctx->d_dbg) {
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_dbg, cap * 16 * sizeof(unsigned long long)));
        HIP_TRY(ctx, hipMemset(ctx->d_dbg, 0, cap * 16 * sizeof(unsigned long long)));
    }
    if (out && ctx->d_dbg) {
        HIP_TRY(ctx, hipDeviceSynchronize());
        size_t n = (size_t)(max_waves < 0 ? 0 : max_waves);
        if (n > cap) n = cap;
        HIP_TRY(ctx, hipMemcpy(out, ctx->d_dbg, n * 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
        HIP_TRY(ctx, hipMemset(ctx->d_dbg, 0, cap * 16 * sizeof(unsigned long long)));
    }
    if (!enable && ctx->d_dbg) { (void)hipFree(ctx->d_dbg); ctx->d_dbg = nullptr; }
    return RMDF_OK;
}

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
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    if (!ptr || bytes == 0) return fail(ctx, RMDF_E_INVALID, "rmdf_register_host_buffer: bad argument");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (auto &r : ctx->host_regs) if (r.host == (char *)ptr && r.bytes == bytes) return RMDF_OK;
    HIP_TRY(ctx, hipHostRegister(ptr, bytes, hipHostRegisterMapped));
    void *dev = nullptr;
    hipError_t e = hipHostGetDevicePointer(&dev, ptr, 0);
    if (e != hipSuccess) {
        (void)hipHostUnregister(ptr);
        return fail(ctx, RMDF_E_HIP, std::string("hipHostGetDevicePointer: ") + hipGetErrorString(e));
    }
    ctx->host_regs.push_back(rmdf_ctx::HostReg{ (char *)ptr, bytes, (char *)dev });
    return RMDF_OK;
}

int rmdf_unregister_host_buffer(rmdf_ctx *ctx, void *ptr)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
    for (size_t i = 0; i < ctx->host_regs.size(); i++)
        if (ctx->host_regs[i].host == (char *)ptr) {
            HIP_TRY(ctx, hipSetDevice(ctx->device));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipHostUnregister(ptr));
            ctx->host_regs.erase(ctx->host_regs.begin() + (long)i);
            return RMDF_OK;
        }
    return fail(ctx, RMDF_E_INVALID, "rmdf_unregister_host_buffer: not registered");
}

int rmdf_synchronize(rmdf_ctx *ctx, void *stream)
{
    if (!ctx) return fail(nullptr, RMDF_E_INVALID, "null ctx");
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
