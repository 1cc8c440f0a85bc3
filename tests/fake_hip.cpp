// fake_hip.cpp -- a TEST DOUBLE of the HIP runtime entry points librmdf imports, for running the library's HOST code -- context set-up,
// staging, whole-frame bands, tile jobs, shard gather / assembly, env-map pipeline and cache files, error paths -- on a box WITHOUT a GPU,
// under AddressSanitizer if wanted (tools/asan_host.sh).  Test infrastructure only: it is LD_PRELOADed into a test's child process
// (tests/test_host_logic.py: test_host_paths_against_the_hip_double, tests/fake_hip_workload.py); the product never loads it, and
// librmdf without a real HIP runtime and device fails in rmdf_create as before.
//
// What it is: "device memory" is malloc'd host memory (so every copy between caller memory, staging and device buffers is bounds-checked
// by the allocator's red zones under ASan, and a device pointer can be read by the checks below); kernels are NOT run -- each launch is
// replaced by a small host routine that writes what
// the kernel's CONTRACT says where the kernel would write it (k_render: a pixel value that is a hash of everything that must
// distinguish one pixel from another -- position, frame size, scene, camera, step limit, the contents of the three cube maps -- into the
// frame, the mirror, the planes, the shard slots; k_assemble_shards, k_resolve_box2, k_fill_u32, k_order_blocks exactly; the env-map
// kernels as nearest-texel stand-ins that read and write every element the real ones do).
// Two modes.  Default: streams and events are tokens and every operation completes before its call returns.  FAKE_HIP_ASYNC=1: every
// stream is a worker thread with an in-order queue -- launches, hipMemcpyAsync between device and page-locked memory, hipMemsetAsync and
// event records are queued and return at once; hipStreamWaitEvent makes a stream wait for the record it saw; hipEventQuery /
// hipStreamQuery answer hipErrorNotReady while work is pending; hipFree / hipHostFree / hipDeviceSynchronize drain every stream first, as
// the real runtime does; a copy that touches PAGEABLE memory (neither hipMalloc'ed nor hipHostMalloc'ed) drains its stream and runs in
// the caller's thread.  FAKE_HIP_JITTER_US=n delays every queued operation by a random 0..n microseconds.  In this mode a host that
// reads a buffer before the event that guards it, or re-uses a staging chunk before its DMA has run, computes with stale data -- the
// workload's comparisons fail -- and ThreadSanitizer (tools/asan_host.sh tsan) sees the two accesses without a happens-before edge.
// What it is NOT: evidence about any kernel or about the driver.  The pixels it produces mean nothing; that two paths produce the SAME
// pixels, that nothing outside a buffer is touched and that nothing leaks is what the tests look at.
//
// Build (hipcc for the structure definitions of rmdf_internal.hpp; host code only):
//   hipcc -O1 -g -std=c++17 -fPIC --cuda-host-only -x hip -I ray-marching-distance-fields_amd/csrc -shared tests/fake_hip.cpp -o tests/libfake_hip.so
#include <hip/hip_runtime_api.h>
#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <random>
#include <thread>
#include <unistd.h>
#include <malloc.h>
#include <sys/mman.h>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <vector>
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rmdf_internal.hpp"

using rmdf::FrameParams;

namespace {

// (the globals are built on first use and never torn down: librmdf.so's static constructors register its kernels, and in a program that
// links librmdf directly they may run before this library's own)
struct Globals {
    std::mutex mu;
    std::map<const void *, std::string> kernels;        // host stub address -> device name
    std::set<void *> dev, host;
    std::string last_unknown;
};
Globals &G() { static Globals *g = new Globals; return *g; }
#define g_mu (G().mu)
#define g_kernels (G().kernels)
#define g_dev (G().dev)
#define g_host (G().host)
#define g_last_unknown (G().last_unknown)
std::atomic<long long> g_launches{ 0 }, g_unknown{ 0 }, g_copied{ 0 }, g_fail_malloc_in{ 0 };
struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local std::vector<CallCfg> t_cfg;

uint32_t mix(uint32_t h, uint32_t v) { h ^= v + 0x9e3779b9u + (h << 6) + (h >> 2); h *= 0x85ebca6bu; h ^= h >> 13; return h; }
uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

void tile_rect(int idx, int w, int h, int &x0, int &y0, int &x1, int &y1)
{
    const int tx = (idx % 64) % 8, ty = (idx % 64) / 8;          // ShaderRendering.hs:183-193 as rmdf_api.cpp: tile_rect_host states it
    x0 = (2 * tx * w + 7) / 16; x1 = (2 * (tx + 1) * w + 7) / 16; y0 = (2 * ty * h + 7) / 16; y1 = (2 * (ty + 1) * h + 7) / 16;
}

// ---- k_render<SCENE, MERGE, OUT> ------------------------------------------------------------------------------------------------------
// librmdf_xcheck.so's FrameParams continues behind the product's with the one-launch band hand-over (rmdf_internal.hpp: the first fields of
// its RMDF_XCHECK block); the product library has no such launch and its FrameParams ends where this file's copy of the struct ends
struct BandTail { unsigned *band_count; volatile unsigned *band_flag; unsigned band_seq; int band_strip_rows; };

void fake_render(const std::string &name, const FrameParams &p, dim3 grid, const BandTail &bt)
{
    int scene = 0, merge = 0, out = 0;
    const size_t at = name.find("k_renderILi");
    if (at == std::string::npos || sscanf(name.c_str() + at, "k_renderILi%dELb%dELi%dE", &scene, &merge, &out) != 3) abort();
    // what distinguishes one frame from another: everything the kernel's arguments carry that reaches a pixel
    uint32_t base = mix(0x1234567u, (uint32_t)scene);
    for (int k = 0; k < 12; k++) base = mix(base, fbits(p.cam[k]));
    base = mix(mix(mix(base, (uint32_t)p.w), (uint32_t)p.h), (uint32_t)p.max_steps);
    base = mix(base, fbits(p.power) * (scene == 3 ? 1u : 0u));
    for (const rmdf::CubeDev *c : { &p.env_refl, &p.env_cos1, &p.env_cos8 }) {
        const size_t n = (size_t)6 * (c->W + 2) * (c->W + 2);
        uint32_t t = (uint32_t)c->W;
        for (size_t i = 0; i < n; i += (n / 61 + 1)) t = mix(mix(t, c->texels[i].x), c->texels[i].y);   // reads first .. last texel
        t = mix(mix(t, c->texels[n - 1].x), c->texels[n - 1].y);
        base = mix(base, t);
    }
    const size_t nblk = (size_t)grid.x * grid.y * grid.z;
    if (p.block_order) {                                       // must be a permutation of the launch's strips
        std::vector<char> seen(nblk, 0);
        for (size_t i = 0; i < nblk; i++) { const unsigned s = p.block_order[i]; if (s >= nblk || seen[s]) { fprintf(stderr, "fake_hip: block_order is not a permutation\n"); abort(); } seen[s] = 1; }
    }
    auto put = [&](size_t idx, int px, int py) {
        const uint32_t h = mix(mix(base, (uint32_t)px), (uint32_t)py);
        const uint32_t v = h | 0xff000000u;
        if (out != 2 || p.rgba8) p.rgba8[idx] = v;
        if (out == 1) p.rgba8_mirror[idx] = v;
        if (out == 2) {
            if (p.rgba_f32) p.rgba_f32[idx] = make_float4((float)(v & 255u) / 255.0f, (float)((v >> 8) & 255u) / 255.0f, (float)((v >> 16) & 255u) / 255.0f, 1.0f);
            const unsigned steps = (h >> 3) % (unsigned)(p.max_steps > 0 ? p.max_steps : 1), hit = (h >> 1) & 1u;
            if (p.steps) p.steps[idx] = (uint16_t)(steps | (hit << 15));
            if (p.iters) p.iters[idx] = (uint16_t)((h >> 9) & 0x3ffu);
        }
    };
    if (p.n_shard_tiles > 0) {
        if ((int)grid.z != p.n_shard_tiles) abort();
        for (int slot = 0; slot < p.n_shard_tiles; slot++) {
            int x0, y0, x1, y1;
            tile_rect(p.shard_tile[slot], p.w, p.h, x0, y0, x1, y1);
            const size_t obase = (size_t)slot * (size_t)(p.w / 8) * (size_t)(p.h / 8);
            for (int py = y0; py < y1; py++)
                for (int px = x0; px < x1; px++) put(obase + (size_t)(px - x0) + (size_t)(py - y0) * (size_t)(x1 - x0), px, py);
        }
    } else {
        for (int py = p.y0; py < p.y1; py++)
            for (int px = p.x0; px < p.x1; px++) put((size_t)px + (size_t)py * (size_t)p.w, px, py);
    }
    if (p.block_cost)
        for (size_t i = 0; i < nblk; i++) p.block_cost[i] = 1u + (mix(base, (uint32_t)i) >> 20);
    if (out == 1 && bt.band_flag) {
        const int nb = ((int)grid.y + bt.band_strip_rows - 1) / bt.band_strip_rows;
        for (int b = 0; b < nb; b++) { if (bt.band_count[b] != 0u) abort(); __atomic_store_n((unsigned *)&bt.band_flag[b], bt.band_seq, __ATOMIC_RELEASE); }
    }
}

template <typename T> T arg(void **args, int i) { T v; memcpy(&v, args[i], sizeof v); return v; }

float src_checksum(const float *src, size_t n)                  // reads every element
{
    float s = 0.0f;
    for (size_t i = 0; i < n; i++) s += src[i] * (float)((i % 7) + 1);
    return s;
}

// the arguments are decoded NOW (the args array belongs to the caller's stack frame); what is returned runs when the stream gets to it
// FAKE_HIP_EMULATE=1: instead of the stand-ins below, every kernel the emulator holds (tests/kernel_on_host.cpp: the library's kernel SOURCE
// compiled for the CPU, one fiber per lane) is RUN -- the double then is a functional model of the device, the pixels are the real ones,
// and the GPU tier's own tests can be run through the C ABI without a GPU (small frames: the emulator is ~10^4 x slower than the GPU).
// libkernel_on_host.so serves librmdf.so's launches, libkernel_on_host_xcheck.so librmdf_xcheck.so's (its FrameParams is longer).
struct Emulator {
    void *(*prepare)(const char *, void **) = nullptr;
    void (*run)(void *, const unsigned *, const unsigned *, size_t, int) = nullptr;
};
Emulator *emulator_for(bool from_xcheck)
{
    static const bool on = getenv("FAKE_HIP_EMULATE") && atoi(getenv("FAKE_HIP_EMULATE")) != 0;
    if (!on) return nullptr;
    // (function-local statics, not std::call_once: in a plain C host that does not link libpthread itself, call_once's gthread probe fails)
    auto load = [](int k) {
        Emulator e;
        Dl_info di;
        std::string dir = ".";
        if (dladdr((void *)&emulator_for, &di) && di.dli_fname) { dir = di.dli_fname; const size_t sl = dir.rfind('/'); dir = sl == std::string::npos ? "." : dir.substr(0, sl); }
        const std::string path = dir + (k ? "/libkernel_on_host_xcheck.so" : "/libkernel_on_host.so");
        void *h = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
        if (!h) { fprintf(stderr, "fake_hip: FAKE_HIP_EMULATE: cannot load %s: %s\n", path.c_str(), dlerror()); abort(); }
        e.prepare = (void *(*)(const char *, void **))dlsym(h, "koh_prepare");
        e.run = (void (*)(void *, const unsigned *, const unsigned *, size_t, int))dlsym(h, "koh_run");
        if (!e.prepare || !e.run) { fprintf(stderr, "fake_hip: %s lacks koh_prepare / koh_run\n", path.c_str()); abort(); }
        return e;
    };
    const int k = from_xcheck ? 1 : 0;
    static Emulator emu0 = load(0);
    if (k == 0) return &emu0;
    static Emulator emu1 = load(1);
    return &emu1;
}

std::atomic<unsigned long long> g_emulated{ 0 };

std::function<void()> make_task(const std::string &name, dim3 grid, dim3 block, void **args, bool from_xcheck, size_t shmem)
{
    if (Emulator *e = emulator_for(from_xcheck)) {
        if (void *closure = e->prepare(name.c_str(), args)) {
            static const int threads = getenv("FAKE_HIP_EMULATE_THREADS") ? atoi(getenv("FAKE_HIP_EMULATE_THREADS")) : (int)std::min(8u, std::max(1u, std::thread::hardware_concurrency()));
            g_emulated++;
            return [=] { const unsigned g[3] = { grid.x, grid.y, grid.z }, b[3] = { block.x, block.y, block.z }; e->run(closure, g, b, shmem, threads); };
        }
        // (a kernel the emulator does not hold -- the cross-check build's alternative schedules -- keeps its stand-in: say so, once per name)
        static std::mutex mu; static std::set<std::string> told;
        std::lock_guard<std::mutex> lk(mu);
        if (told.insert(name).second) fprintf(stderr, "fake_hip: FAKE_HIP_EMULATE: no emulated kernel %s (stand-in used)\n", name.c_str());
    }
    (void)block;
    auto has = [&](const char *s) { return name.find(s) != std::string::npos; };
    if (has("k_renderILi")) {
        const FrameParams p = *(const FrameParams *)args[0];
        BandTail bt = { nullptr, nullptr, 0u, 0 };
        if (from_xcheck) memcpy(&bt, (const char *)args[0] + sizeof(FrameParams), sizeof bt);
        return [=] { fake_render(name, p, grid, bt); };
    }
    if (has("k_order_blocks")) {
        const unsigned *cost = arg<const unsigned *>(args, 0); const int n = arg<int>(args, 1); unsigned *order = arg<unsigned *>(args, 2);
        return [=] {
        std::vector<unsigned> idx((size_t)n);
        for (int i = 0; i < n; i++) idx[(size_t)i] = (unsigned)i;
        std::stable_sort(idx.begin(), idx.end(), [&](unsigned a, unsigned b) { return cost[a] > cost[b]; });
        for (int i = 0; i < n; i++) order[i] = idx[(size_t)i];
        };
    }
    if (has("k_fill_u32")) {
        uint32_t *dst = arg<uint32_t *>(args, 0); const uint32_t v = arg<uint32_t>(args, 1); const size_t n = arg<size_t>(args, 2);
        return [=] { for (size_t i = 0; i < n; i++) dst[i] = v; };
    }
    if (has("k_assemble_shards")) {                               // both forms: the same mapping (rmdf_util.hip)
        const uint32_t *gathered = arg<const uint32_t *>(args, 0); uint32_t *frame = arg<uint32_t *>(args, 1);
        const int w = arg<int>(args, 2), h = arg<int>(args, 3), nranks = arg<int>(args, 4);
        const rmdf::ShardWhere where = arg<rmdf::ShardWhere>(args, 5);
        return [=] {
        const int tw = w / 8, th = h / 8, slots = (64 + nranks - 1) / nranks;
        for (int py = 0; py < h; py++)
            for (int px = 0; px < w; px++) {
                const int tx = px / tw, ty = py / th, rs = where.v[tx + ty * 8];
                frame[(size_t)py * w + px] = gathered[((size_t)(rs >> 8) * slots + (rs & 255)) * (size_t)tw * th + (size_t)(px - tx * tw) + (size_t)(py - ty * th) * tw];
            }
        };
    }
    if (has("k_resolve_box2")) {
        const uint32_t *src = arg<const uint32_t *>(args, 0); const int dw = arg<int>(args, 1), dh = arg<int>(args, 2); uint32_t *dst = arg<uint32_t *>(args, 3);
        return [=] {
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++) {
                uint32_t o = 0;
                for (int c = 0; c < 4; c++) {
                    unsigned s = 2;
                    for (int k = 0; k < 4; k++) s += (src[(size_t)(2 * y + (k >> 1)) * (2 * dw) + 2 * x + (k & 1)] >> (8 * c)) & 255u;
                    o |= ((s >> 2) & 255u) << (8 * c);
                }
                dst[(size_t)y * dw + x] = o;
            }
        };
    }
    if (has("k_cube_upload")) {                                   // stand-in: clamped copy as four 16-bit quantities per texel
        const float *faces = arg<const float *>(args, 0); const int W = arg<int>(args, 1); uint2 *padded = arg<uint2 *>(args, 2);
        return [=] {
        const int P = W + 2;
        for (int f = 0; f < 6; f++)
            for (int Y = 0; Y < P; Y++)
                for (int X = 0; X < P; X++) {
                    const int x = std::min(std::max(X - 1, 0), W - 1), y = std::min(std::max(Y - 1, 0), W - 1);
                    const float *t = &faces[(((size_t)f * W + y) * W + x) * 3];
                    uint2 o;
                    o.x = (fbits(t[0]) >> 16) | (fbits(t[1]) & 0xffff0000u);
                    o.y = fbits(t[2]) >> 16;
                    padded[((size_t)f * P + Y) * P + X] = o;
                }
        };
    }
    if (has("k_latlong_to_cube")) {                               // stand-in: nearest texel at the table's (u, v)
        const float *ll = arg<const float *>(args, 0); const int w = arg<int>(args, 1), h = arg<int>(args, 2), cw = arg<int>(args, 3);
        const float2 *uv = arg<const float2 *>(args, 4); float *faces = arg<float *>(args, 5);
        return [=] {
        for (size_t i = 0; i < (size_t)6 * cw * cw; i++) {
            const int x = std::min(w - 1, std::max(0, (int)(uv[i].x * (float)(w - 1)))), y = std::min(h - 1, std::max(0, (int)(uv[i].y * (float)(h - 1))));
            for (int k = 0; k < 3; k++) faces[i * 3 + k] = ll[((size_t)y * w + x) * 3 + k];
        }
        };
    }
    if (has("k_resize_latlong")) {                                // stand-in: nearest texel
        const float *src = arg<const float *>(args, 0); const int sw = arg<int>(args, 1), sh = arg<int>(args, 2), dw = arg<int>(args, 3), dh = arg<int>(args, 4);
        float *out = arg<float *>(args, 5);
        return [=] {
        for (int y = 0; y < dh; y++)
            for (int x = 0; x < dw; x++)
                for (int k = 0; k < 3; k++) out[((size_t)y * dw + x) * 3 + k] = src[((size_t)std::min(sh - 1, y * sh / dh) * sw + std::min(sw - 1, x * sw / dw)) * 3 + k];
        };
    }
    if (has("k_prefilter")) {                                     // stand-in: source scaled by a function of the power; reads both tables end to end
        const float *src = arg<const float *>(args, 0); const int w = arg<int>(args, 1), h = arg<int>(args, 2);
        struct Out { float *p; float power; };
        std::vector<Out> outs;
        const float *lutT; const float2 *tcs;
        if (has("k_prefilter_fused4")) {
            lutT = arg<const float *>(args, 3); tcs = arg<const float2 *>(args, 4);
            const float pw[4] = { 1.0f, 8.0f, 64.0f, 512.0f };
            for (int k = 0; k < 4; k++) outs.push_back(Out{ arg<float *>(args, 5 + k), pw[k] });
        } else if (has("k_prefilter_chan") || has("k_prefilter_ring")) {
            int l2 = 0;
            sscanf(name.c_str() + name.find("ILi") + 3, "%d", &l2);
            lutT = arg<const float *>(args, 3); tcs = arg<const float2 *>(args, 4);
            outs.push_back(Out{ arg<float *>(args, 5), (float)(1 << l2) });
        } else {                                                  // k_prefilter<LOG2P, LUT_IN_LDS>(src, w, h, power, lutT, tcs, out, ..)
            lutT = arg<const float *>(args, 4); tcs = arg<const float2 *>(args, 5);
            outs.push_back(Out{ arg<float *>(args, 6), arg<float>(args, 3) });
        }
        return [=] {
            const size_t n = (size_t)w * h * 3, nl = (size_t)((w + 63) / 64) * w * 64;
            const float cs = src_checksum(src, n), tb = lutT[0] + lutT[nl - 1] + tcs[0].x + tcs[h - 1].y;
            for (const Out &o : outs)
                if (o.p) for (size_t i = 0; i < n; i++) o.p[i] = src[i] / (1.0f + o.power) + 0.0f * (cs + tb);
        };
    }
    if (has("k_selftest") || has("k_clock_probe")) return [] { };  // their result buffers were cleared by the host: "no mismatch"
    g_unknown++;
    std::lock_guard<std::mutex> lk(g_mu);
    g_last_unknown = name;
    return [] { };
}

// ---- streams and events ------------------------------------------------------------------------------------------------------------------------
bool async_mode() { static const bool v = getenv("FAKE_HIP_ASYNC") && atoi(getenv("FAKE_HIP_ASYNC")) != 0; return v; }
int jitter_us() { static const int v = getenv("FAKE_HIP_JITTER_US") ? atoi(getenv("FAKE_HIP_JITTER_US")) : 0; return v; }
#define g_async (async_mode())
#define g_jitter_us (jitter_us())
// FAKE_HIP_SABOTAGE=events: hipEventQuery / hipEventSynchronize claim completion at once -- what a host that forgot to wait would see.
// For the test of the tests: with it the asynchronous workload must FAIL (tests/test_host_logic.py).
bool sabotage_events() { static const bool v = getenv("FAKE_HIP_SABOTAGE") && strstr(getenv("FAKE_HIP_SABOTAGE"), "events"); return v; }
#define g_sabotage_events (sabotage_events())

struct Stream {
    std::mutex mu;
    std::condition_variable cv_work, cv_idle;
    std::deque<std::function<void()>> q;
    bool busy = false, stop = false;
    std::thread th;
    std::minstd_rand rng{ 12345u };
    void start() { th = std::thread([this] { run(); }); }
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || !q.empty(); });
                if (q.empty()) return;
                f = std::move(q.front()); q.pop_front(); busy = true;
            }
            if (g_jitter_us > 0) usleep((useconds_t)(rng() % (unsigned)(g_jitter_us + 1)));
            f();
            { std::lock_guard<std::mutex> lk(mu); busy = false; if (q.empty()) cv_idle.notify_all(); }
        }
    }
    void push(std::function<void()> f) { { std::lock_guard<std::mutex> lk(mu); q.push_back(std::move(f)); } cv_work.notify_one(); }
    void drain() { std::unique_lock<std::mutex> lk(mu); cv_idle.wait(lk, [&] { return q.empty() && !busy; }); }
    bool idle() { std::lock_guard<std::mutex> lk(mu); return q.empty() && !busy; }
    void finish() { drain(); { std::lock_guard<std::mutex> lk(mu); stop = true; } cv_work.notify_one(); if (th.joinable()) th.join(); }
};
struct Event {
    std::mutex mu;
    std::condition_variable cv;
    unsigned long long recorded = 0, completed = 0;       // generations: a record is complete when completed >= its generation
};
struct Handles {
    std::map<Stream *, std::shared_ptr<Stream>> streams;  // (drain_all holds references while it waits: another thread may destroy a stream meanwhile)
    std::map<struct Event *, std::shared_ptr<struct Event>> events;   // queued operations hold a reference: an event may be destroyed while they are pending
    std::shared_ptr<Stream> null_stream;                  // the legacy default stream (hipMemset, stream 0)
};
Handles &H() { static Handles *h = new Handles; return *h; }
#define g_streams (H().streams)
#define g_events (H().events)
#define g_null_stream (H().null_stream)

std::shared_ptr<Stream> stream_of(hipStream_t s)
{
    std::lock_guard<std::mutex> lk(g_mu);
    if (s) { auto it = g_streams.find((Stream *)s); return it == g_streams.end() ? nullptr : it->second; }
    if (!g_null_stream) { g_null_stream = std::make_shared<Stream>(); if (g_async) g_null_stream->start(); }
    return g_null_stream;
}
void submit(hipStream_t s, std::function<void()> f)
{
    if (!g_async) { f(); return; }
    auto st = stream_of(s);
    if (!st) { fprintf(stderr, "fake_hip: work submitted to a stream that does not exist\n"); abort(); }
    st->push(std::move(f));
}
void drain_all()
{
    if (!g_async) return;
    std::vector<std::shared_ptr<Stream>> all;
    { std::lock_guard<std::mutex> lk(g_mu); for (auto &kv : g_streams) all.push_back(kv.second); if (g_null_stream) all.push_back(g_null_stream); }
    for (auto &st : all) st->drain();
}
bool library_owned(const void *p)                          // inside a hipMalloc / hipHostMalloc block?  (pageable memory is not)
{
    std::lock_guard<std::mutex> lk(g_mu);
    for (const std::set<void *> *set : { &g_dev, &g_host }) {
        auto it = set->upper_bound((void *)p);
        if (it == set->begin()) continue;
        --it;
        if ((const char *)p < (const char *)*it + malloc_usable_size(*it)) return true;
    }
    return false;
}

void *take(std::set<void *> &s, size_t bytes)
{
    void *p = malloc(bytes ? bytes : 1);
    if (!p) return nullptr;
    std::lock_guard<std::mutex> lk(g_mu);
    s.insert(p);
    return p;
}
bool give(std::set<void *> &s, void *p)
{
    { std::lock_guard<std::mutex> lk(g_mu); if (!s.erase(p)) return false; }
    free(p);
    return true;
}

}  // namespace

extern "C" {

// per-entry-point call counts (fake_hip_calls): what a frame costs the host in runtime calls is a figure of its own -- each is microseconds
struct Calls { std::mutex mu; std::map<std::string, long long> n; };
static Calls &CALLS() { static Calls *c = new Calls; return *c; }
static void count_call(const char *name) { Calls &c = CALLS(); std::lock_guard<std::mutex> lk(c.mu); c.n[name]++; }
#define COUNT() count_call(__func__)

// ---- test hooks ---------------------------------------------------------------------------------------------------------------------------
// "name=count;name=count;..." of every runtime entry point called since the last reset (reset != 0: clear after reading)
int fake_hip_calls(char *buf, int cap, int reset)
{
    Calls &c = CALLS();
    std::lock_guard<std::mutex> lk(c.mu);
    std::string out;
    for (auto &kv : c.n) out += kv.first + "=" + std::to_string(kv.second) + ";";
    if (reset) c.n.clear();
    snprintf(buf, (size_t)cap, "%s", out.c_str());
    return (int)out.size();
}
// out[0..7]: live device allocations, live page-locked allocations, live streams, live events, launches, launches of kernels the double
// does not know, bytes moved by memcpy calls, registered kernels
void fake_hip_counters(long long out[8])
{
    std::lock_guard<std::mutex> lk(g_mu);
    out[0] = (long long)g_dev.size(); out[1] = (long long)g_host.size(); out[2] = (long long)g_streams.size(); out[3] = (long long)g_events.size();
    out[4] = g_launches; out[5] = g_unknown; out[6] = g_copied; out[7] = (long long)g_kernels.size();
}
int fake_hip_last_unknown(char *buf, int cap)
{
    std::lock_guard<std::mutex> lk(g_mu);
    snprintf(buf, (size_t)cap, "%s", g_last_unknown.c_str());
    return (int)g_last_unknown.size();
}
void fake_hip_fail_malloc_in(long long n) { g_fail_malloc_in = n; }       // the n-th device / page-locked allocation from now fails (0 = none)

// ---- registration and launch ---------------------------------------------------------------------------------------------------------------
void **__hipRegisterFatBinary(const void *) { static void *token[1]; return token; }
void __hipUnregisterFatBinary(void **) { }
void __hipRegisterFunction(void **, const void *host_fn, char *, const char *device_name, unsigned, void *, void *, void *, void *, int *)
{
    std::lock_guard<std::mutex> lk(g_mu);
    g_kernels[host_fn] = device_name;
}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) { t_cfg.push_back(CallCfg{ grid, block, shmem, stream }); return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream)
{
    if (t_cfg.empty()) return hipErrorInvalidValue;
    const CallCfg c = t_cfg.back(); t_cfg.pop_back();
    *grid = c.grid; *block = c.block; *shmem = c.shmem; *stream = c.stream;
    return hipSuccess;
}
hipError_t hipLaunchKernel(const void *fn, dim3 grid, dim3 block, void **args, size_t shmem, hipStream_t stream)
{ COUNT();
    std::string name;
    { std::lock_guard<std::mutex> lk(g_mu); auto it = g_kernels.find(fn); if (it == g_kernels.end()) return hipErrorInvalidDeviceFunction; name = it->second; }
    g_launches++;
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x == 0) return hipErrorInvalidConfiguration;
    // which build launches: the cross-check library's FrameParams is longer (BandTail).  Told by a symbol only that build exports, not by the
    // file's name (tools/asan_host.sh builds it as librmdf_asan.so)
    Dl_info di;
    bool from_xcheck = false;
    if (dladdr(fn, &di) && di.dli_fname) {
        static std::mutex mu; static std::map<std::string, bool> known;
        std::lock_guard<std::mutex> lk(mu);
        auto it = known.find(di.dli_fname);
        if (it == known.end()) {
            void *h = dlopen(di.dli_fname, RTLD_NOW | RTLD_NOLOAD);
            const bool x = h && dlsym(h, "rmdf_debug_cornell_table") != nullptr;
            if (h) dlclose(h);
            it = known.emplace(di.dli_fname, x).first;
        }
        from_xcheck = it->second;
    }
    submit(stream, make_task(name, grid, block, args, from_xcheck, shmem));
    return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }

// ---- device, streams, events -----------------------------------------------------------------------------------------------------------------
// FAKE_HIP_DEVICES=n: n identical devices (one process per "GPU" hosts, examples/c_host_multi.c); all of them share the double's one heap
static int device_count() { static const int v = getenv("FAKE_HIP_DEVICES") && atoi(getenv("FAKE_HIP_DEVICES")) > 0 ? atoi(getenv("FAKE_HIP_DEVICES")) : 1; return v; }
static thread_local int t_device = 0;
hipError_t hipGetDeviceCount(int *n) { *n = device_count(); return hipSuccess; }
hipError_t hipSetDevice(int d) { COUNT(); if (d < 0 || d >= device_count()) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipGetDeviceProperties(hipDeviceProp_t *prop, int d)
{
    if (d < 0 || d >= device_count()) return hipErrorInvalidDevice;
    memset(prop, 0, sizeof *prop);
    snprintf(prop->name, sizeof prop->name, "no GPU: tests/fake_hip.cpp");
    snprintf(prop->gcnArchName, sizeof prop->gcnArchName, "gfx950:fake");
    prop->multiProcessorCount = 256;
    prop->totalGlobalMem = (size_t)8 << 30;
    prop->warpSize = 64;
    return hipSuccess;
}
hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
hipError_t hipDeviceSynchronize(void) { COUNT(); drain_all(); return hipSuccess; }
hipError_t hipGetLastError(void) { COUNT(); return hipSuccess; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorOutOfMemory ? "out of memory (fake_hip)" : (e == hipErrorNotReady ? "not ready" : "error (fake_hip)")); }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned)
{ COUNT();
    auto st = std::make_shared<Stream>();
    if (g_async) st->start();
    { std::lock_guard<std::mutex> lk(g_mu); g_streams[st.get()] = st; }
    *s = (hipStream_t)st.get();
    return hipSuccess;
}
hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned f, int) { return hipStreamCreateWithFlags(s, f); }
hipError_t hipStreamDestroy(hipStream_t s)
{
    std::shared_ptr<Stream> st;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_streams.find((Stream *)s);
        if (it == g_streams.end()) return hipErrorInvalidHandle;
        st = it->second;
        g_streams.erase(it);
    }
    if (g_async) st->finish();                                // (the object goes when the last drain_all that saw it lets go)
    return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t s) { COUNT(); auto st = stream_of(s); if (!st) return hipErrorInvalidHandle; return !g_async || st->idle() ? hipSuccess : hipErrorNotReady; }
hipError_t hipStreamSynchronize(hipStream_t s) { COUNT(); auto st = stream_of(s); if (!st) return hipErrorInvalidHandle; if (g_async) st->drain(); return hipSuccess; }
static std::shared_ptr<Event> event_of(hipEvent_t e)
{
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_events.find((Event *)e);
    return it == g_events.end() ? nullptr : it->second;
}
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned)
{ COUNT();
    auto ev = std::make_shared<Event>();
    { std::lock_guard<std::mutex> lk(g_mu); g_events[ev.get()] = ev; }
    *e = (hipEvent_t)ev.get();
    return hipSuccess;
}
hipError_t hipEventDestroy(hipEvent_t e)
{
    std::lock_guard<std::mutex> lk(g_mu);
    return g_events.erase((Event *)e) ? hipSuccess : hipErrorInvalidHandle;
}
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s)
{ COUNT();
    auto ev = event_of(e);
    if (!ev) return hipErrorInvalidHandle;
    unsigned long long gen;
    { std::lock_guard<std::mutex> lk(ev->mu); gen = ++ev->recorded; }
    submit(s, [ev, gen] { { std::lock_guard<std::mutex> lk(ev->mu); if (ev->completed < gen) ev->completed = gen; } ev->cv.notify_all(); });
    return hipSuccess;
}
hipError_t hipEventQuery(hipEvent_t e) { COUNT(); auto ev = event_of(e); if (!ev) return hipErrorInvalidHandle; if (g_sabotage_events) return hipSuccess; std::lock_guard<std::mutex> lk(ev->mu); return ev->completed >= ev->recorded ? hipSuccess : hipErrorNotReady; }
hipError_t hipEventSynchronize(hipEvent_t e)
{ COUNT();
    auto ev = event_of(e);
    if (!ev) return hipErrorInvalidHandle;
    if (g_sabotage_events) return hipSuccess;
    std::unique_lock<std::mutex> lk(ev->mu);
    ev->cv.wait(lk, [&] { return ev->completed >= ev->recorded; });
    return hipSuccess;
}
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned)
{ COUNT();
    auto ev = event_of(e);
    if (!ev) return hipErrorInvalidHandle;
    unsigned long long gen;
    { std::lock_guard<std::mutex> lk(ev->mu); gen = ev->recorded; }      // the record the call saw (none: nothing to wait for)
    submit(s, [ev, gen] { std::unique_lock<std::mutex> lk(ev->mu); ev->cv.wait(lk, [&] { return ev->completed >= gen; }); });
    return hipSuccess;
}

// ---- memory ---------------------------------------------------------------------------------------------------------------------------------
static bool failing()
{
    long long n = g_fail_malloc_in.load();
    while (n > 0 && !g_fail_malloc_in.compare_exchange_weak(n, n - 1)) { }
    return n == 1;
}
hipError_t hipMalloc(void **p, size_t bytes) { COUNT(); if (failing()) { *p = nullptr; return hipErrorOutOfMemory; } *p = take(g_dev, bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { COUNT(); if (!p) return hipSuccess; drain_all(); return give(g_dev, p) ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { COUNT(); if (failing()) { *p = nullptr; return hipErrorOutOfMemory; } *p = take(g_host, bytes); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { COUNT(); if (!p) return hipSuccess; drain_all(); return give(g_host, p) ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
hipError_t hipMemcpyAsync(void *dst, const void *src, size_t n, hipMemcpyKind, hipStream_t s)
{ COUNT();
    g_copied += (long long)n;
    if (g_async && library_owned(dst) && library_owned(src)) { submit(s, [=] { memmove(dst, src, n); }); return hipSuccess; }
    if (g_async) { auto st = stream_of(s); if (st) st->drain(); }   // pageable memory: behind the stream's work, in the caller's thread
    memmove(dst, src, n);
    return hipSuccess;
}
hipError_t hipMemcpy(void *dst, const void *src, size_t n, hipMemcpyKind) { drain_all(); memmove(dst, src, n); g_copied += (long long)n; return hipSuccess; }   // (tests/fake_rccl.c uses it)
hipError_t hipMemcpy2DAsync(void *dst, size_t dpitch, const void *src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t s)
{ COUNT();
    if (width > dpitch || width > spitch) return hipErrorInvalidPitchValue;
    g_copied += (long long)(width * height);
    auto run = [=] { for (size_t y = 0; y < height; y++) memmove((char *)dst + y * dpitch, (const char *)src + y * spitch, width); };
    if (g_async && library_owned(dst) && library_owned(src)) { submit(s, run); return hipSuccess; }
    if (g_async) { auto st = stream_of(s); if (st) st->drain(); }
    run();
    return hipSuccess;
}
hipError_t hipMemset(void *p, int v, size_t n) { COUNT(); drain_all(); memset(p, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t s) { COUNT(); submit(s, [=] { memset(p, v, n); }); return hipSuccess; }
// The virtual-memory calls of the cross-check build's electric-fence allocator (rmdf_host.hpp: GuardAlloc): an address range is an
// mmap(PROT_NONE), mapping a handle into it makes that part readable and writable, everything else stays a fence -- with 4 KiB pages where
// the GPU has 2 MiB ones.  A stand-in "kernel" (or a host copy) that steps outside a guarded buffer dies of SIGSEGV here.
struct VmHandle { size_t bytes; };
hipError_t hipMemGetAllocationGranularity(size_t *g, const hipMemAllocationProp *, hipMemAllocationGranularity_flags) { *g = 4096; return hipSuccess; }
hipError_t hipMemCreate(hipMemGenericAllocationHandle_t *h, size_t bytes, const hipMemAllocationProp *, unsigned long long)
{
    if (failing()) return hipErrorOutOfMemory;
    *h = (hipMemGenericAllocationHandle_t) new VmHandle{ bytes };
    return hipSuccess;
}
hipError_t hipMemRelease(hipMemGenericAllocationHandle_t h) { delete (VmHandle *)h; return hipSuccess; }
hipError_t hipMemAddressReserve(void **p, size_t bytes, size_t, void *, unsigned long long)
{
    void *m = mmap(nullptr, bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) return hipErrorOutOfMemory;
    *p = m;
    return hipSuccess;
}
hipError_t hipMemAddressFree(void *p, size_t bytes) { return munmap(p, bytes) == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipMemMap(void *p, size_t bytes, size_t, hipMemGenericAllocationHandle_t h, unsigned long long)
{
    if (!h || ((VmHandle *)h)->bytes < bytes) return hipErrorInvalidValue;
    return mprotect(p, bytes, PROT_READ | PROT_WRITE) == 0 ? hipSuccess : hipErrorInvalidValue;
}
hipError_t hipMemUnmap(void *p, size_t bytes) { drain_all(); return mprotect(p, bytes, PROT_NONE) == 0 && madvise(p, bytes, MADV_DONTNEED) == 0 ? hipSuccess : hipErrorInvalidValue; }
hipError_t hipMemSetAccess(void *, size_t, const hipMemAccessDesc *, size_t) { return hipSuccess; }

}  // extern "C"
