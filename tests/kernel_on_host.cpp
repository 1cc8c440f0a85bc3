// kernel_on_host.cpp -- TEST INFRASTRUCTURE (CPU tier, round 6): the render kernel's SOURCE -- csrc/rmdf_render.hip as it is, render_body with its
// march loops, the workgroup pooling of stragglers through LDS mailboxes, the AO queue, the eight-lanes-per-ray Cornell tail with its DPP
// minima, the quad exchanges, the LDS-staged stores, the strip-cost reduction, and the library's own launch code (launch_render: grid,
// variant choice, block order) -- compiled for the CPU and executed by the SIMT emulator of tests/koh_shim/hip/hip_runtime.h (one fiber per
// lane, wave collectives and workgroup barriers with the hardware's meaning).  tests/test_kernel_source_on_host.py renders small frames
// with it and holds every plane to the oracle's: step counts, hit mask and escape-iteration counts bit-exact, float colour and RGBA8
// bit-exact too (same source arithmetic, correctly rounded seeds).  What it cannot see: the code generator, the hardware's memory model
// (the mailbox protocol's acquire / release pairs are plain accesses here), timing.  Never part of the product.
#include <atomic>
#include <thread>
#include <vector>

#define RMDF_HOST_EMULATION 1
#include "rmdf_render.hip"           // (-I csrc; <hip/hip_runtime.h>, <hip/hip_fp16.h> resolve to tests/koh_shim/)
#ifndef KOH_RENDER_ONLY               // (the A/B builds of the render kernel compile the render source alone: a third of the time)
#include "rmdf_env.hip"              // ... and the env-map kernels (cube upload, lat/long -> cube, resize, the lobe prefilter's forms)
#include "rmdf_util.hip"             // ... and the small ones (box resolve, shard assembly, fill; the GPU self-tests of the exact arithmetic)
#endif
#if defined(RMDF_XCHECK) && !defined(KOH_RENDER_ONLY)
#include "xcheck/rmdf_march.hip"     // librmdf_xcheck.so's alternative schedule of the power-8 Mandelbulb (persistent waves fed from a global counter, G-buffer,
#include "xcheck/rmdf_stats.hip"     //   separate shade kernel) and its march statistics: the cross-check of the product's schedule, emulated as well
#endif

thread_local int doh_seed_mode = 0;
thread_local unsigned doh_seed_rng = 12345u;

namespace koh {

thread_local Lane *cur = nullptr;
thread_local Grid *grd = nullptr;

namespace {

// Context switch between fibers: callee-saved registers, stack pointer, MXCSR and the x87 control word -- glibc's swapcontext does the same
// plus a signal-mask system call per switch, which was half of the emulator's run time.
struct Ctx { void *sp; };
extern "C" void koh_ctx_switch(Ctx *from, Ctx *to);
#if !defined(__x86_64__)
#error "tests/kernel_on_host.cpp: the fiber switch is written for x86-64"
#endif
asm(R"(
    .text
    .globl koh_ctx_switch
    .type koh_ctx_switch,@function
koh_ctx_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq (%rsi), %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size koh_ctx_switch, .-koh_ctx_switch
)");

constexpr size_t kStack = 256 * 1024;
struct Fiber { Ctx ctx; Lane lane; bool done; };
struct WaveState { int arrived = 0; unsigned gen = 0; uint64_t val[2][64]; const char *what[2] = { nullptr, nullptr }; unsigned long long chain = 0; };
struct Block {
    std::vector<Fiber> fib;
    std::vector<WaveState> waves;
    int sync_arrived = 0; unsigned sync_gen = 0;
    Ctx sched;
    void (*tramp)(void *) = nullptr; void *closure = nullptr;
    char *stacks = nullptr; size_t nstacks = 0;
    void *dyn = nullptr;
    unsigned long long idle_spins = 0;
};
thread_local Block *blk = nullptr;
std::atomic<unsigned long long> g_counts[6];
// lane utilisation of the Mandelbulb estimates under the schedule being emulated, in the cost model of tools/ubench/sched_sim (round 4): an
// estimate segment costs 87 vector instructions per iteration pass + 100; a wave pays the MAXIMUM over its lanes, 64 lanes wide
std::atomic<unsigned long long> g_useful{ 0 }, g_slots{ 0 }, g_wave_passes{ 0 }, g_lane_passes{ 0 };
// the Cornell box's cost hooks (RMDF_EMU_COST): per wave the serial chain -- the slowest lane between collectives, summed -- and over the launch the
// longest wave's chain (what one frame at a time waits for), the sum of all chains (what the issue ports see) and the lanes' own work
std::atomic<unsigned long long> g_chain_max{ 0 }, g_chain_sum{ 0 }, g_chain_useful{ 0 };
constexpr unsigned kPassCost = 87, kEstimateCost = 100;
std::atomic<int> g_seed_mode{ 0 };
std::atomic<int> g_threads{ 1 };

void flush_wave(Block &b, int wave);

void fiber_main()
{
    blk->tramp(blk->closure);
    Fiber *f = (Fiber *)((char *)cur - offsetof(Fiber, lane));
    f->done = true;
    koh_ctx_switch(&f->ctx, &blk->sched);
    abort();                                                        // a finished fiber is never resumed
}

// a new fiber's stack, as koh_ctx_switch expects to find a suspended one: control words, six callee-saved registers, the address to "return" to
void fiber_init(Fiber &f, char *stack_base)
{
    uintptr_t top = ((uintptr_t)stack_base + kStack) & ~(uintptr_t)15;
    uint64_t *sp = (uint64_t *)top;
    *--sp = 0;                                                      // (entry is reached by `ret`: the stack looks as after a call)
    *--sp = (uint64_t)(uintptr_t)&fiber_main;
    for (int i = 0; i < 6; i++) *--sp = 0;                          // rbp, rbx, r12 .. r15
    *--sp = 0x037f00001f80ull;                                      // MXCSR 0x1f80 (round to nearest, exceptions masked), x87 control word 0x037f
    f.ctx.sp = sp;
}

void run_block(Block &b, Grid &g, unsigned bx, unsigned by, unsigned bz)
{
    const unsigned nt = g.block.x * g.block.y * g.block.z;
    if (nt % 64u) { fprintf(stderr, "koh: workgroup of %u threads is not whole waves\n", nt); abort(); }
    g.bid = uint3{ bx, by, bz };
    b.fib.resize(nt);
    b.waves.assign(nt / 64u, WaveState());
    b.sync_arrived = 0;
    if (b.nstacks < nt) { free(b.stacks); b.stacks = (char *)malloc((size_t)nt * kStack); b.nstacks = nt; }
    for (unsigned t = 0; t < nt; t++) {
        Fiber &f = b.fib[t];
        f.done = false;
        f.lane.tid = uint3{ t % g.block.x, (t / g.block.x) % g.block.y, t / (g.block.x * g.block.y) };
        f.lane.lane = (int)(t & 63u); f.lane.wave = (int)(t >> 6); f.lane.nseg = 0; f.lane.cur_passes = 0; f.lane.cost = 0;
        fiber_init(f, b.stacks + (size_t)t * kStack);
    }
    blk = &b; grd = &g;
    unsigned long long rounds = 0;
    for (;;) {
        bool any = false;
        for (unsigned t = 0; t < nt; t++) {
            Fiber &f = b.fib[t];
            if (f.done) continue;
            any = true;
            cur = &f.lane;
            koh_ctx_switch(&b.sched, &f.ctx);
        }
        if (!any) {
            for (int wv = 0; wv < (int)b.waves.size(); wv++) {
                flush_wave(b, wv);
                const unsigned long long c = b.waves[(size_t)wv].chain;
                g_chain_sum += c;
                unsigned long long m = g_chain_max.load();
                while (c > m && !g_chain_max.compare_exchange_weak(m, c)) { }
            }
            break;
        }
        if (++rounds > 20000000ull) {
            fprintf(stderr, "koh: workgroup (%u, %u, %u) makes no progress: lanes wait in", bx, by, bz);
            for (auto &w : b.waves) fprintf(stderr, " [%s: %d of 64 arrived]", w.what[w.gen & 1u] ? w.what[w.gen & 1u] : "-", w.arrived);
            fprintf(stderr, " [__syncthreads: %d of %u]\n", b.sync_arrived, nt);
            abort();
        }
    }
    cur = nullptr;
}

}  // namespace

void count(int kind) { g_counts[kind].fetch_add(1, std::memory_order_relaxed); }

// the wave's lanes have all arrived at a collective (or the workgroup at a barrier, or the fibers at their end): price what they did since the last one
namespace {
void flush_wave(Block &b, int wave)
{
    unsigned long long useful = 0, slots = 0, wp = 0, lp = 0;
    int maxseg = 0;
    for (int l = 0; l < 64; l++) { const Lane &L = b.fib[(size_t)wave * 64 + l].lane; if (L.nseg > maxseg) maxseg = L.nseg; }
    for (int k = 0; k < maxseg; k++) {
        unsigned mx = 0, mxp = 0;
        for (int l = 0; l < 64; l++) {
            const Lane &L = b.fib[(size_t)wave * 64 + l].lane;
            if (L.nseg <= k) continue;
            const unsigned np = L.seg[k] & 0x7fffu;
            const unsigned c = kPassCost * np + ((L.seg[k] & 0x8000u) ? 0u : kEstimateCost);
            useful += c; lp += np;
            if (c > mx) mx = c;
            if (np > mxp) mxp = np;
        }
        slots += 64ull * mx; wp += mxp;
    }
    unsigned cmax = 0; unsigned long long csum = 0;
    for (int l = 0; l < 64; l++) { Lane &L = b.fib[(size_t)wave * 64 + l].lane; if (L.cost > cmax) cmax = L.cost; csum += L.cost; L.cost = 0; L.nseg = 0; }
    if (cmax) { b.waves[(size_t)wave].chain += cmax; g_chain_useful += csum; }
    if (slots) { g_useful += useful; g_slots += slots; g_wave_passes += wp; g_lane_passes += lp; }
}
}  // namespace

void yield_lane()
{
    Fiber *f = (Fiber *)((char *)cur - offsetof(Fiber, lane));
    koh_ctx_switch(&f->ctx, &blk->sched);
}

const uint64_t *wave_gather(uint64_t v, const char *what)
{
    WaveState &w = blk->waves[(size_t)cur->wave];
    const unsigned g = w.gen, slot = g & 1u;
    if (w.arrived == 0) w.what[slot] = what;
    else if (w.what[slot] != what) {
        fprintf(stderr, "koh: lanes of one wave meet in DIFFERENT collectives (%s vs %s): a collective under divergent control flow\n", w.what[slot], what);
        abort();
    }
    w.val[slot][cur->lane] = v;
    if (++w.arrived == 64) { flush_wave(*blk, cur->wave); w.arrived = 0; w.gen = g + 1u; }
    else while (w.gen == g) yield_lane();
    return w.val[slot];
}

void block_barrier()
{
    Block &b = *blk;
    const unsigned g = b.sync_gen;
    if (++b.sync_arrived == (int)b.fib.size()) { for (int wv = 0; wv < (int)b.waves.size(); wv++) flush_wave(b, wv); b.sync_arrived = 0; b.sync_gen = g + 1u; }
    else while (b.sync_gen == g) yield_lane();
}

void *dyn_lds() { return blk->dyn; }

void launch(dim3 grid, dim3 block, size_t dyn_lds_bytes, void (*tramp)(void *), void *closure)
{
    const unsigned long long nblk = (unsigned long long)grid.x * grid.y * grid.z;
    if (nblk == 0) return;
    std::atomic<unsigned long long> next{ 0 };
    const int nthreads = (int)std::min<unsigned long long>((unsigned long long)std::max(1, g_threads.load()), nblk);
    auto worker = [&] {
        doh_seed_mode = g_seed_mode.load(); doh_seed_rng = 777u;
        Block b; Grid g; g.grid = grid; g.block = block;
        b.tramp = tramp; b.closure = closure;
        b.dyn = dyn_lds_bytes ? calloc(1, dyn_lds_bytes + 64) : nullptr;
        for (;;) {
            const unsigned long long i = next.fetch_add(1);
            if (i >= nblk) break;
            run_block(b, g, (unsigned)(i % grid.x), (unsigned)((i / grid.x) % grid.y), (unsigned)(i / ((unsigned long long)grid.x * grid.y)));
        }
        free(b.stacks); free(b.dyn);
        blk = nullptr; grd = nullptr;
    };
    if (nthreads == 1) { worker(); return; }
    std::vector<std::thread> ts;
    for (int t = 0; t < nthreads; t++) ts.emplace_back(worker);
    for (auto &t : ts) t.join();
}

}  // namespace koh

extern "C" {

struct KohFrame {          // mirrored in the Python test (ctypes)
    int scene, w, h, x0, y0, x1, y1, max_steps;
    float cam[12], fov_xs, time;
    int no_merge, no_prune;
    const void *env_refl, *env_cos1, *env_cos8;      // padded RGB16F texels (6 x (W+2)^2 x 4 halfs)
    int w_refl, w_cos1, w_cos8;
    const float *cornell_tri, *cornell_tab;
    const unsigned *cornell_grid;
    uint32_t *rgba8, *rgba8_mirror;
    float *rgba_f32;
    uint16_t *steps, *iters;
    const unsigned *block_order;
    unsigned *block_cost;
    int n_shard_tiles;
    unsigned char shard_tile[64];
    int threads, seed_mode;
    // librmdf_xcheck.so's one-launch band hand-over (FrameParams' band fields: -DRMDF_XCHECK builds of this harness only)
    unsigned *band_count, *band_flag;
    unsigned band_seq;
    int band_strip_rows;
};

int koh_frame_size(void) { return (int)sizeof(KohFrame); }

// lane-calls of every collective kind since the last call (ballot, shuffles, readfirstlane, DPP, polled loads, __syncthreads): a test can tell
// that a frame DID go through the pooled march's mailboxes or the Cornell tail's DPP minima
void koh_take_counts(unsigned long long out[6]) { for (int k = 0; k < 6; k++) out[k] = koh::g_counts[k].exchange(0); }
// since the last call: useful lane-slots and issued lane-slots of the Mandelbulb estimates (87 per pass + 100 per estimate, a wave pays its slowest
// lane 64 wide), wave-level iteration passes, lane-level iteration passes
// the Cornell box's chains since the last call: longest wave's, sum over waves, the lanes' own instructions (RMDF_EMU_COST units)
void koh_take_chains(unsigned long long out[3]) { out[0] = koh::g_chain_max.exchange(0); out[1] = koh::g_chain_sum.exchange(0); out[2] = koh::g_chain_useful.exchange(0); }
void koh_take_schedule(unsigned long long out[4])
{
    out[0] = koh::g_useful.exchange(0); out[1] = koh::g_slots.exchange(0); out[2] = koh::g_wave_passes.exchange(0); out[3] = koh::g_lane_passes.exchange(0);
}

// the frame parameters as rmdf_api.cpp's fill_params + its callers set them, then the library's own launch_render
int koh_render(const KohFrame *f)
{
    using namespace rmdf;
    FrameParams p;
    memset(&p, 0, sizeof p);
    memcpy(p.cam, f->cam, sizeof p.cam);
    p.fov_xs = f->fov_xs;
    {
        const float a = f->time / 2.0f;                              // fragment.shd:116-119 (rmdf_api.cpp: fill_params)
        float pow_offs = a - 9.0f * floorf(a / 9.0f);
        if (pow_offs > 4.5f) pow_offs = 9.0f - pow_offs;
        p.power = pow_offs + 2.0f;
    }
    p.wf = (float)f->w; p.hf = (float)f->h; p.aspect = p.wf / p.hf;
    p.w = f->w; p.h = f->h;
    p.max_steps = f->max_steps <= 0 ? (int)shk::march_max_steps_default : f->max_steps;
    p.x0 = f->x0; p.y0 = f->y0; p.x1 = f->x1; p.y1 = f->y1;
    p.n_shard_tiles = f->n_shard_tiles;
    memcpy(p.shard_tile, f->shard_tile, 64);
    p.env_refl = CubeDev{ (const uint2 *)f->env_refl, f->w_refl };
    p.env_cos1 = CubeDev{ (const uint2 *)f->env_cos1, f->w_cos1 };
    p.env_cos8 = CubeDev{ (const uint2 *)f->env_cos8, f->w_cos8 };
    p.cornell = f->cornell_tri; p.cornell_tab = f->cornell_tab; p.cornell_grid = f->cornell_grid;
    p.cornell_prune = f->no_prune ? 0 : 1;
    p.merge_stragglers = f->scene == 0 ? 0 : (f->no_merge == 0 ? MERGE_T : (f->no_merge == 1 ? 0 : f->no_merge));       // (no_merge >= 2: that pooling threshold, <= MERGE_T)
    p.rgba8 = f->rgba8; p.rgba8_mirror = f->rgba8_mirror; p.rgba_f32 = (float4 *)f->rgba_f32; p.steps = f->steps; p.iters = f->iters;
    p.block_order = f->block_order; p.block_cost = f->block_cost;
#ifdef RMDF_XCHECK
    p.band_count = f->band_count; p.band_flag = f->band_flag; p.band_seq = f->band_seq; p.band_strip_rows = f->band_strip_rows;
    p.fold_min = RMDF_MB8_FOLD_MIN;
#endif
    koh::g_seed_mode = f->seed_mode; koh::g_threads = f->threads;
    return (int)launch_render(f->scene, p, nullptr);
}

int koh_grid_blocks(const KohFrame *f)
{
    rmdf::FrameParams p;
    memset(&p, 0, sizeof p);
    p.w = f->w; p.h = f->h; p.x0 = f->x0; p.y0 = f->y0; p.x1 = f->x1; p.y1 = f->y1; p.n_shard_tiles = f->n_shard_tiles;
    return rmdf::render_grid_blocks(p);
}

// k_order_blocks itself (1024 lanes, LDS histogram): the order it leaves for `cost`
int koh_order_blocks(const unsigned *cost, int n, unsigned *order, int threads)
{
    koh::g_threads = threads;
    return (int)rmdf::launch_order_blocks(cost, n, order, nullptr);
}
// (-DRMDF_XCHECK builds: k_order_blocks_bands when nbands > 0)
int koh_order_blocks_bands(const unsigned *cost, int n, unsigned *order, int gx, int band_strip_rows, int nbands, int threads)
{
    koh::g_threads = threads;
    return (int)rmdf::launch_order_blocks(cost, n, order, nullptr, gx, band_strip_rows, nbands);
}

#ifndef KOH_RENDER_ONLY
// ---- the env-map and utility kernels through the library's own launchers ----
int koh_cube_upload(const float *faces, int W, void *padded, int threads) { koh::g_threads = threads; return (int)rmdf::launch_cube_upload(faces, W, (uint2 *)padded, nullptr); }
int koh_latlong_to_cube(const float *latlong, int w, int h, const float *uv, float *faces, int threads)
{
    koh::g_threads = threads;
    return (int)rmdf::launch_latlong_to_cube(latlong, w, h, (const float2 *)uv, faces, nullptr);
}
int koh_resize_latlong(const float *src, int sw, int sh, int dstw, int dsth, float *out, int threads)
{
    koh::g_threads = threads;
    return (int)rmdf::launch_resize_latlong(src, sw, sh, dstw, dsth, out, nullptr);
}
int koh_prefilter(const float *src, int w, int h, float power, const float *lutT, const float *tcs, float *out, int split_ok, int threads)
{
    koh::g_threads = threads;
    return (int)rmdf::launch_prefilter(src, w, h, power, lutT, (const float2 *)tcs, out, nullptr, split_ok != 0);
}
int koh_prefilter_fused4(const float *src, int w, int h, const float *lutT, const float *tcs, float *o0, float *o1, float *o2, float *o3, int threads)
{
    koh::g_threads = threads;
    float *const outs[4] = { o0, o1, o2, o3 };
    return (int)rmdf::launch_prefilter_fused4(src, w, h, lutT, (const float2 *)tcs, outs, nullptr);
}
int koh_resolve_box2(const uint32_t *src, int sw, int sh, uint32_t *dst, int threads) { koh::g_threads = threads; return (int)rmdf::launch_resolve_box2(src, sw, sh, dst, nullptr); }
int koh_fill_u32(uint32_t *dst, uint32_t value, size_t n, int threads) { koh::g_threads = threads; return (int)rmdf::launch_fill_u32(dst, value, n, nullptr); }
int koh_assemble_shards(const uint32_t *gathered, uint32_t *frame, int w, int h, int nranks, const unsigned short *where64, int threads)
{
    koh::g_threads = threads;
    rmdf::ShardWhere wh;
    for (int i = 0; i < 64; i++) wh.v[i] = where64[i];
    return (int)rmdf::launch_assemble_shards(gathered, frame, w, h, nranks, wh, nullptr);
}
#endif

}  // extern "C"

// ---- launches by NAME: the HIP double (tests/fake_hip.cpp, FAKE_HIP_EMULATE=1) hands every hipLaunchKernel of librmdf here, so that the library's
// own host code drives the emulated kernels through the C ABI -- the GPU tier's tests then run on a box without a GPU.  The table is keyed by the
// kernels' mangled names, which are the same on both sides (Itanium ABI): dladdr of the host-compiled kernel gives the name the device code
// object registered.
#include <dlfcn.h>
#include <functional>
#include <map>
#include <string>
#include <tuple>

namespace {

typedef std::function<std::function<void()> *(void **)> Prepare;
std::map<std::string, Prepare> &table() { static std::map<std::string, Prepare> t; return t; }

template <typename... A, size_t... I>
std::function<void()> *bind_args(void (*k)(A...), void **args, std::index_sequence<I...>)
{
    auto tup = std::make_tuple(*(typename std::decay<A>::type *)args[I]...);          // copied NOW: the array belongs to the caller's frame
    return new std::function<void()>([k, tup] { std::apply(k, tup); });
}
template <typename... A>
void reg(void (*k)(A...))
{
    Dl_info di;
    if (!dladdr((void *)k, &di) || !di.dli_sname) { fprintf(stderr, "koh: a kernel without a dynamic symbol\n"); abort(); }
    table()[di.dli_sname] = [k](void **args) { return bind_args(k, args, std::index_sequence_for<A...>{}); };
}
template <int S> void reg_render()
{
    reg(&rmdf::k_render<S, false, 0>); reg(&rmdf::k_render<S, false, 1>); reg(&rmdf::k_render<S, false, 2>);
    reg(&rmdf::k_render<S, true, 0>); reg(&rmdf::k_render<S, true, 1>); reg(&rmdf::k_render<S, true, 2>);
}
#ifndef KOH_RENDER_ONLY
template <int L> void reg_prefilter() { reg(&rmdf::k_prefilter<L, false>); reg(&rmdf::k_prefilter<L, true>); }
#endif

void build_table()
{
    using namespace rmdf;
    reg_render<0>(); reg_render<1>(); reg_render<2>(); reg_render<3>();
    reg(&k_order_blocks);
#ifdef RMDF_XCHECK
    reg(&k_order_blocks_bands);
#endif
#ifndef KOH_RENDER_ONLY
    reg(&k_cube_upload); reg(&k_latlong_to_cube); reg(&k_resize_latlong);
    reg_prefilter<-1>(); reg_prefilter<0>(); reg_prefilter<3>(); reg_prefilter<6>(); reg_prefilter<9>();
    reg(&k_prefilter_chan<0>); reg(&k_prefilter_chan<3>); reg(&k_prefilter_chan<6>); reg(&k_prefilter_chan<9>);
    reg(&k_prefilter_fused4);
#ifdef RMDF_XCHECK
    reg(&k_march_mb8); reg(&k_shade); reg(&k_march_stats);
    reg(&k_prefilter_ring<0>); reg(&k_prefilter_ring<3>); reg(&k_prefilter_ring<6>); reg(&k_prefilter_ring<9>);
#endif
    reg(&k_resolve_box2); reg(&k_assemble_shards); reg(&k_assemble_shards_x4); reg(&k_fill_u32); reg(&k_clock_probe);
    reg(&k_selftest_cornell_div); reg(&k_selftest_exact_math); reg(&k_selftest_mb8_folds); reg(&k_selftest_pinned_math);
    reg(&k_selftest_shading_math); reg(&k_selftest_fill_cube); reg(&k_selftest_frame_quotients);
#endif
}

}  // namespace

extern "C" {

// closure for one launch of the kernel registered under `mangled_name` with its arguments copied, or NULL (a kernel this build does not hold)
void *koh_prepare(const char *mangled_name, void **args)
{
    static const bool once = (build_table(), true);
    (void)once;
    auto it = table().find(mangled_name);
    return it == table().end() ? nullptr : (void *)it->second(args);
}
void koh_run(void *closure, const unsigned grid[3], const unsigned block[3], size_t dyn_lds_bytes, int threads)
{
    std::function<void()> *fn = (std::function<void()> *)closure;
    koh::g_threads = threads;
    koh::launch(dim3(grid[0], grid[1], grid[2]), dim3(block[0], block[1], block[2]), dyn_lds_bytes, [](void *c) { (*(std::function<void()> *)c)(); }, fn);
    delete fn;
}
int koh_kernels(void) { (void)koh_prepare("", nullptr); return (int)table().size(); }

}  // extern "C"
