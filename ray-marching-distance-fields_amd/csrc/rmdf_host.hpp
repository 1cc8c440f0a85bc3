// rmdf_host.hpp -- host-side plumbing of the C ABI (rmdf_api.cpp): the ctx's worker threads and the page-locked staging through which
// EVERY byte travels between caller memory and the device.
//
// Why staging (round 5).  A copy between device memory and pageable host memory makes the HIP runtime page-lock the caller's pages on
// the fly (hsa_amd_memory_lock: a userptr mapping of those pages into the GPU's address space, created and torn down per call, the
// queue drained around it).  The library does not want the driver to build GPU mappings of memory it does not own -- a numpy array, a
// Haskell storable vector, a std::vector of its own that is freed a line later: the intermittent GPU memory fault of round 4 was first
// caught under exactly such a call (NOTEBOOK.md A.5) -- and it does not want to pay for them: 0.59 ms for a 1080p frame into pageable memory
// against 0.38 ms for the kernel.  So the runtime only ever sees device memory and memory the library page-locked itself, once:
//   upload():   caller -> staging (host threads) -> device (async: the call may return before the DMA ends, the caller's memory is
//               not read after return);
//   download(): device -> staging -> caller (host threads), chunk by chunk, the copy of chunk i overlapping the DMA of chunk i + 1.
#pragma once

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#include <immintrin.h>
#endif
#include <hip/hip_runtime.h>
#if defined(__linux__)
#include <sched.h>
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace rmdf {

// A few host threads owned by the ctx.  run(parts, fn): fn(part) for part = 0 .. parts - 1, part 0 on the calling thread, the others
// on the workers; begin() / finish() split that so that the caller can issue device work in between.  One job at a time, one caller
// thread (rmdf.h: a ctx is used by one thread at a time).  A worker that has just finished a part spins for the next job for a few
// dozen microseconds before it goes to sleep on the condition variable: the band-by-band frame copy (render_whole_frame_host) hands out
// several jobs per frame, and a futex wake-up per job would cost more than the copy.
class WorkPool {
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_work_, cv_done_;
    std::function<void(int)> fn_;
    std::atomic<unsigned> gen_{ 0 };
    std::atomic<bool> stop_{ false };
    std::atomic<int> pending_{ 0 };
    std::atomic<long long> prime_until_{ 0 };
    std::atomic<unsigned> prime_gen_{ 0 };
    int parts_ = 1;
    bool open_ = false;

    static long long now_us()
    {
        return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    }
    void worker(int idx)
    {
        unsigned seen = 0, seen_prime = 0;
        long long hot_until = now_us();
        for (;;) {
            bool got = false;
            for (;;) {
                const long long t = now_us(), prime = prime_until_.load(std::memory_order_relaxed);
                if (t >= hot_until && t >= prime) break;
                if (gen_.load(std::memory_order_acquire) != seen || stop_.load(std::memory_order_relaxed)) { got = true; break; }
                __builtin_ia32_pause();
            }
            std::function<void(int)> fn;
            int np;
            {
                std::unique_lock<std::mutex> lk(m_);
                if (!got) cv_work_.wait(lk, [&] { return stop_.load() || gen_.load() != seen || prime_gen_.load() != seen_prime; });
                if (stop_.load()) return;
                seen_prime = prime_gen_.load();
                if (gen_.load() == seen) continue;            // woken to spin (prime()), not for a job
                seen = gen_.load(); fn = fn_; np = parts_;
            }
            if (idx + 1 < np) fn(idx + 1);
            if (pending_.fetch_sub(1, std::memory_order_acq_rel) == 1) { std::lock_guard<std::mutex> lk(m_); cv_done_.notify_one(); }
            hot_until = now_us() + 60;
        }
    }

public:
    // n worker threads (besides the calling thread); idempotent.  Fewer threads than asked for (thread creation failed) is still correct.
    void start(int n)
    {
        if (!threads_.empty() || n < 1) return;
        try { for (int i = 0; i < n; i++) threads_.emplace_back([this, i] { worker(i); }); } catch (...) { }
    }
    int workers() const { return (int)threads_.size(); }
    // jobs are about to arrive within the next `us` microseconds (a frame's bands landing one by one): the workers leave their sleep now
    // and spin that long, so that each job starts within a microsecond instead of a futex wake-up
    void prime(long long us)
    {
        if (threads_.empty()) return;
        prime_until_.store(now_us() + us, std::memory_order_relaxed);
        { std::lock_guard<std::mutex> lk(m_); prime_gen_.fetch_add(1, std::memory_order_release); }
        cv_work_.notify_all();
    }
    void relax() { prime_until_.store(0, std::memory_order_relaxed); }     // the frame is through: back to sleep (after the usual 60 us)
    // parts is clamped to workers() + 1
    void begin(int parts, std::function<void(int)> fn)
    {
        if (parts > workers() + 1) parts = workers() + 1;
        if (parts < 1) parts = 1;
        {
            std::lock_guard<std::mutex> lk(m_);
            fn_ = std::move(fn); parts_ = parts; open_ = true;
            if (parts > 1) { pending_.store(workers(), std::memory_order_relaxed); gen_.fetch_add(1, std::memory_order_release); }
        }
        if (parts > 1) cv_work_.notify_all();
    }
    void finish()
    {
        if (!open_) return;
        fn_(0);
        if (parts_ > 1) {
            // the parts are short: look a few times before sleeping
            for (int i = 0; i < 2000 && pending_.load(std::memory_order_acquire) != 0; i++) __builtin_ia32_pause();
            if (pending_.load(std::memory_order_acquire) != 0) {
                std::unique_lock<std::mutex> lk(m_);
                cv_done_.wait(lk, [&] { return pending_.load() == 0; });
            }
        }
        open_ = false;
    }
    void run(int parts, std::function<void(int)> fn) { begin(parts, std::move(fn)); finish(); }
    // part `part` of `parts` of [0, n), cut at multiples of `align`
    static void slice(size_t n, int part, int parts, size_t align, size_t &lo, size_t &hi)
    {
        lo = (n * (size_t)part / (size_t)parts) / align * align;
        hi = part == parts - 1 ? n : (n * (size_t)(part + 1) / (size_t)parts) / align * align;
    }
    // memcpy(dst, src, bytes) on all threads (small copies: on the caller's)
    void copy(void *dst, const void *src, size_t bytes)
    {
        if (bytes < ((size_t)1 << 20) || workers() == 0) { memcpy(dst, src, bytes); return; }
        const int parts = workers() + 1;
        // RMDF_COPY_NT=1 (A/B knob of librmdf_xcheck.so, read once; written after GPU access closed in round 5, not yet measured): the slices with streaming
        // stores -- a slice is below glibc's non-temporal threshold, so plain memcpy reads every destination line before overwriting it
        static const bool stream = copy_stream_wanted();
        run(parts, [=](int part) {
            size_t lo, hi;
            slice(bytes, part, parts, 4096, lo, hi);
            if (hi <= lo) return;
            if (stream) copy_stream((char *)dst + lo, (const char *)src + lo, hi - lo);
            else memcpy((char *)dst + lo, (const char *)src + lo, hi - lo);
        });
    }
    static bool copy_stream_wanted()
    {
        // (the cross-check build and the pool's unit test only: unmeasured, so the product library neither reads the variable nor streams)
#if (defined(RMDF_XCHECK) || defined(RMDF_HOST_POOL_TEST)) && defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
        const char *v = getenv("RMDF_COPY_NT");
        return v && v[0] == '1' && __builtin_cpu_supports("avx2");
#else
        return false;
#endif
    }
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    __attribute__((target("avx2"))) static void copy_stream(char *dst, const char *src, size_t n)
    {
        size_t head = (32 - ((uintptr_t)dst & 31)) & 31;               // streaming stores want 32-byte aligned destinations
        if (head > n) head = n;
        memcpy(dst, src, head);
        dst += head; src += head; n -= head;
        const size_t blocks = n / 128;
        for (size_t i = 0; i < blocks; i++) {
            const __m256i a = _mm256_loadu_si256((const __m256i *)(src + 128 * i)), b = _mm256_loadu_si256((const __m256i *)(src + 128 * i + 32));
            const __m256i c = _mm256_loadu_si256((const __m256i *)(src + 128 * i + 64)), d = _mm256_loadu_si256((const __m256i *)(src + 128 * i + 96));
            _mm256_stream_si256((__m256i *)(dst + 128 * i), a); _mm256_stream_si256((__m256i *)(dst + 128 * i + 32), b);
            _mm256_stream_si256((__m256i *)(dst + 128 * i + 64), c); _mm256_stream_si256((__m256i *)(dst + 128 * i + 96), d);
        }
        memcpy(dst + 128 * blocks, src + 128 * blocks, n - 128 * blocks);
        _mm_sfence();                                                   // the streamed lines are globally visible before the job counts as done
    }
#else
    static void copy_stream(char *dst, const char *src, size_t n) { memcpy(dst, src, n); }
#endif
    // fn(lo, hi) over [0, n) in row segments (forSegmentsConcurrently, ConcurrentSegments.hs:14-28)
    template <typename F>
    void segments(int n, F fn)
    {
        int parts = workers() + 1;
        if (parts > n / 16) parts = n / 16;
        if (parts <= 1) { fn(0, n); return; }
        run(parts, [&](int part) {
            size_t lo, hi;
            slice((size_t)n, part, parts, 1, lo, hi);
            if (hi > lo) fn((int)lo, (int)hi);
        });
    }
    ~WorkPool()
    {
        { std::lock_guard<std::mutex> lk(m_); stop_.store(true); }
        cv_work_.notify_all();
        for (auto &t : threads_) if (t.joinable()) t.join();
    }
};

// Scope guards of the two states a caller can leave a pool in: an open job (begin() without finish(): the workers may still be copying
// into the caller's buffer, open_ / pending_ stale for the next job) and primed workers (spinning until relax() or the window ends).
// Every exit of the scope -- an error return, an exception on its way to the C ABI's catch -- closes them.
struct PoolJobGuard {
    WorkPool &pool;
    explicit PoolJobGuard(WorkPool &p) : pool(p) { }
    ~PoolJobGuard() { pool.finish(); }                  // idempotent: nothing to do after an explicit finish()
    PoolJobGuard(const PoolJobGuard &) = delete;
    PoolJobGuard &operator=(const PoolJobGuard &) = delete;
};
struct PoolPrimeGuard {
    WorkPool &pool;
    PoolPrimeGuard(WorkPool &p, long long us) : pool(p) { pool.prime(us); }
    ~PoolPrimeGuard() { pool.relax(); }
    PoolPrimeGuard(const PoolPrimeGuard &) = delete;
    PoolPrimeGuard &operator=(const PoolPrimeGuard &) = delete;
};

// CPUs this process may actually use: hardware_concurrency() counts the machine's, not the affinity mask's and not a container's CFS
// quota (cgroup v2 cpu.max / v1 cpu.cfs_quota_us) -- with several ranks per node or a CPU-limited container, threads sized by the
// machine would spin against the very thread they are meant to serve.
inline int usable_cpus()
{
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
#if defined(__linux__) && !defined(__HIP_DEVICE_COMPILE__)
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int a = CPU_COUNT(&set); if (a >= 1 && a < n) n = a; }
    long long quota = -1, period = -1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = { 0 };
        if (fscanf(f, "%31s %lld", q, &period) == 2 && strcmp(q, "max") != 0) quota = atoll(q);
        fclose(f);
    } else {
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &quota) != 1) quota = -1; fclose(g); }
        if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &period) != 1) period = -1; fclose(g); }
    }
    if (quota > 0 && period > 0) { const int c = (int)((quota + period - 1) / period); if (c >= 1 && c < n) n = c; }
#endif
    return n;
}

// Two page-locked chunks and their events.  Not a cache of anything: a chunk is reused as soon as the DMA that read or wrote it is over.
struct Staging {
    static constexpr size_t kChunk = (size_t)8 << 20;
    char      *buf[2] = { nullptr, nullptr };
    size_t     cap = 0;
    hipEvent_t ev[2] = { nullptr, nullptr };
    bool       busy[2] = { false, false };     // an upload's DMA may still be reading the chunk
    unsigned   next = 0;

    hipError_t reserve(size_t want)
    {
        if (want > kChunk) want = kChunk;
        want = (want + 65535) & ~(size_t)65535;
        hipError_t e;
        for (int b = 0; b < 2; b++) if (!ev[b] && (e = hipEventCreateWithFlags(&ev[b], hipEventDisableTiming)) != hipSuccess) return e;
        if (want <= cap) return hipSuccess;
        if ((e = drain()) != hipSuccess) return e;
        for (int b = 0; b < 2; b++) { if (buf[b]) (void)hipHostFree(buf[b]); buf[b] = nullptr; }
        cap = 0;
        for (int b = 0; b < 2; b++) if ((e = hipHostMalloc((void **)&buf[b], want, hipHostMallocDefault)) != hipSuccess) return e;
        cap = want;
        return hipSuccess;
    }
    hipError_t wait(int b)
    {
        if (!busy[b]) return hipSuccess;
        busy[b] = false;
        return hipEventSynchronize(ev[b]);
    }
    hipError_t drain() { hipError_t e = wait(0), f = wait(1); return e != hipSuccess ? e : f; }
    hipError_t upload(WorkPool &pool, void *d_dst, const void *h_src, size_t bytes, hipStream_t st)
    {
        hipError_t e = reserve(bytes);
        for (size_t off = 0; e == hipSuccess && off < bytes; off += cap) {
            const size_t n = bytes - off < cap ? bytes - off : cap;
            const int b = (int)(next++ & 1u);
            if ((e = wait(b)) != hipSuccess) break;
            pool.copy(buf[b], (const char *)h_src + off, n);
            if ((e = hipMemcpyAsync((char *)d_dst + off, buf[b], n, hipMemcpyHostToDevice, st)) != hipSuccess) break;
            if ((e = hipEventRecord(ev[b], st)) != hipSuccess) break;
            busy[b] = true;
        }
        return e;
    }
    // returns with the bytes in h_dst (it waits for ITS copies, not for the stream)
    hipError_t download(WorkPool &pool, void *h_dst, const void *d_src, size_t bytes, hipStream_t st)
    {
        hipError_t e = reserve(bytes);
        int pb = -1;
        size_t poff = 0, pn = 0;
        for (size_t off = 0; e == hipSuccess && off < bytes; off += cap) {
            const size_t n = bytes - off < cap ? bytes - off : cap;
            const int b = (int)(next++ & 1u);
            if ((e = wait(b)) != hipSuccess) break;
            if ((e = hipMemcpyAsync(buf[b], (const char *)d_src + off, n, hipMemcpyDeviceToHost, st)) != hipSuccess) break;
            if ((e = hipEventRecord(ev[b], st)) != hipSuccess) break;
            busy[b] = true;
            if (pb >= 0) { if ((e = wait(pb)) != hipSuccess) break; pool.copy((char *)h_dst + poff, buf[pb], pn); }
            pb = b; poff = off; pn = n;
        }
        if (e == hipSuccess && pb >= 0 && (e = wait(pb)) == hipSuccess) pool.copy((char *)h_dst + poff, buf[pb], pn);
        if (e != hipSuccess) (void)drain();
        return e;
    }
    void destroy()
    {
        (void)drain();
        for (int b = 0; b < 2; b++) { if (buf[b]) (void)hipHostFree(buf[b]); if (ev[b]) (void)hipEventDestroy(ev[b]); buf[b] = nullptr; ev[b] = nullptr; }
        cap = 0;
    }
};

// ---- device allocations -------------------------------------------------------------------------------------------------------------
// dev_malloc / dev_free: hipMalloc / hipFree in the product.  In the cross-check build (librmdf_xcheck.so) the environment variable
// RMDF_GUARD_ALLOC = "end" | "start" turns every device allocation of the library into an ELECTRIC-FENCE allocation: its own physical
// pages (hipMemCreate) mapped into a reserved address range with one UNMAPPED granule before and one behind, the buffer flush against the
// end (or the start) of the mapped part.  A kernel that reads or writes one element past a buffer then takes a GPU memory fault on the
// spot -- every kernel, every access, the shipped code paths, no instrumentation -- instead of reading a neighbour's bytes that happen to
// be mapped (GPU AddressSanitizer is not available on this pool).  tests/test_gpu_guard.py runs the library's workloads both ways.
#ifdef RMDF_XCHECK
struct GuardAlloc {
    struct Rec { void *user; void *va; size_t va_bytes, map_bytes; hipMemGenericAllocationHandle_t handle; };
    std::mutex m;
    std::vector<Rec> recs;
    int mode = -1;                     // -1 unread, 0 off, 1 end, 2 start
    static GuardAlloc &get() { static GuardAlloc g; return g; }
    int read_mode()
    {
        if (mode < 0) { const char *v = getenv("RMDF_GUARD_ALLOC"); mode = !v ? 0 : (v[0] == 'e' ? 1 : (v[0] == 's' ? 2 : 0)); }
        return mode;
    }
    hipError_t alloc(void **p, size_t bytes)
    {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        hipMemAllocationProp prop = {};
        prop.type = hipMemAllocationTypePinned;
        prop.location.type = hipMemLocationTypeDevice;
        prop.location.id = dev;
        size_t gran = 0;
        if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
        if (gran == 0) return hipErrorInvalidValue;
        Rec r = {};
        r.map_bytes = (bytes + gran - 1) / gran * gran;
        r.va_bytes = r.map_bytes + 2 * gran;
        if ((e = hipMemCreate(&r.handle, r.map_bytes, &prop, 0)) != hipSuccess) return e;
        if ((e = hipMemAddressReserve(&r.va, r.va_bytes, gran, nullptr, 0)) != hipSuccess) { (void)hipMemRelease(r.handle); return e; }
        char *mapped = (char *)r.va + gran;
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        if ((e = hipMemMap(mapped, r.map_bytes, 0, r.handle, 0)) != hipSuccess ||
            (e = hipMemSetAccess(mapped, r.map_bytes, &acc, 1)) != hipSuccess) {
            (void)hipMemUnmap(mapped, r.map_bytes); (void)hipMemAddressFree(r.va, r.va_bytes); (void)hipMemRelease(r.handle);
            return e;
        }
        // flush against the end: the last byte of the buffer is the last mapped byte whenever 16 divides the size (the alignment the
        // kernels' widest accesses need); flush against the start: hipMalloc's own alignment
        r.user = mode == 1 ? mapped + r.map_bytes - (bytes + 15) / 16 * 16 : mapped;
        std::lock_guard<std::mutex> lk(m);
        recs.push_back(r);
        *p = r.user;
        return hipSuccess;
    }
    bool free_if_mine(void *p, hipError_t &e)
    {
        Rec r;
        {
            std::lock_guard<std::mutex> lk(m);
            size_t i = 0;
            for (; i < recs.size(); i++) if (recs[i].user == p) break;
            if (i == recs.size()) return false;
            r = recs[i];
            recs.erase(recs.begin() + (long)i);
        }
        e = hipDeviceSynchronize();                             // hipFree's implicit wait
        char *mapped = (char *)r.va + (r.va_bytes - r.map_bytes) / 2;
        (void)hipMemUnmap(mapped, r.map_bytes);
        (void)hipMemRelease(r.handle);
        (void)hipMemAddressFree(r.va, r.va_bytes);
        return true;
    }
};
inline hipError_t dev_malloc(void **p, size_t bytes)
{
    GuardAlloc &g = GuardAlloc::get();
    if (g.read_mode() == 0 || bytes == 0) return hipMalloc(p, bytes);
    return g.alloc(p, bytes);
}
inline hipError_t dev_free(void *p)
{
    if (!p) return hipSuccess;
    hipError_t e = hipSuccess;
    if (GuardAlloc::get().mode > 0 && GuardAlloc::get().free_if_mine(p, e)) return e;
    return hipFree(p);
}
#else
inline hipError_t dev_malloc(void **p, size_t bytes) { return hipMalloc(p, bytes); }
inline hipError_t dev_free(void *p) { return hipFree(p); }
#endif

}  // namespace rmdf
