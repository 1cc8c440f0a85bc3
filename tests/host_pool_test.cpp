// host_pool_test.cpp -- CPU-only exercise of the ctx's worker pool (csrc/rmdf_host.hpp: WorkPool), compiled and run by
// tests/test_host_logic.py: jobs of every part count, begin / finish with work in between, copy() against memcpy, segments(), prime() / relax()
// around jobs and on an idle pool, pools of 0, 1, 3 and 15 workers, thousands of back-to-back jobs (the hot-spin hand-over) and jobs after a sleep
// (the condition-variable hand-over).  Exit code 0 and "pool ok" = every part of every job ran exactly once and every byte arrived.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include <atomic>
#include <vector>

#include "../ray-marching-distance-fields_amd/csrc/rmdf_host.hpp"

using rmdf::WorkPool;

static int check_pool(int nworkers)
{
    WorkPool pool;
    pool.start(nworkers);
    if (pool.workers() != nworkers) { printf("pool of %d: %d workers\n", nworkers, pool.workers()); return 1; }
    std::vector<std::atomic<int>> hits(64);
    unsigned rnd = 12345u + (unsigned)nworkers;
    for (int job = 0; job < 4000; job++) {
        rnd = rnd * 1664525u + 1013904223u;
        int parts = 1 + (int)((rnd >> 8) % 20u);
        const int eff = parts > nworkers + 1 ? nworkers + 1 : parts;
        for (auto &h : hits) h.store(0);
        if ((rnd >> 3) % 7u == 0u) pool.prime(200);
        if ((rnd >> 5) % 97u == 0u) usleep(300);                       // let the workers fall asleep
        pool.begin(parts, [&](int part) { hits[(size_t)part].fetch_add(1); });
        if ((rnd >> 7) % 3u == 0u) for (volatile int spin = 0; spin < 2000; spin++) { }
        pool.finish();
        if ((rnd >> 9) % 11u == 0u) pool.relax();
        for (int p = 0; p < 64; p++)
            if (hits[(size_t)p].load() != (p < eff ? 1 : 0)) { printf("pool of %d, job %d, %d parts: part %d ran %d times\n", nworkers, job, parts, p, hits[(size_t)p].load()); return 1; }
    }
    // copy(): sizes around the threshold and the 4 KiB cut
    std::vector<unsigned char> src((size_t)9 << 20), dst(src.size());
    for (size_t i = 0; i < src.size(); i++) src[i] = (unsigned char)(i * 2654435761u >> 24);
    for (size_t n : { (size_t)0, (size_t)1, (size_t)4095, (size_t)4097, (size_t)1 << 20, ((size_t)1 << 20) + 1, (size_t)8294400, src.size() }) {
        memset(dst.data(), 0xAB, dst.size());
        pool.copy(dst.data() + 3, src.data() + 5, n > 8 ? n - 8 : n);
        const size_t m = n > 8 ? n - 8 : n;
        if (memcmp(dst.data() + 3, src.data() + 5, m) != 0 || dst[2] != 0xAB || dst[3 + m] != 0xAB) { printf("pool of %d: copy of %zu bytes wrong\n", nworkers, m); return 1; }
    }
    // segments(): every index exactly once, for lengths below and above the split threshold
    for (int n : { 0, 1, 15, 16, 17, 100, 1000, 6 * 170 }) {
        std::vector<std::atomic<int>> seen((size_t)n + 1);
        for (auto &s : seen) s.store(0);
        pool.segments(n, [&](int lo, int hi) { for (int i = lo; i < hi; i++) seen[(size_t)i].fetch_add(1); });
        for (int i = 0; i < n; i++) if (seen[(size_t)i].load() != 1) { printf("pool of %d: segments(%d) index %d seen %d times\n", nworkers, n, i, seen[(size_t)i].load()); return 1; }
    }
    pool.prime(100000);                                                   // destroyed while primed: the workers must still leave
    return 0;
}

// the scope guards (round 6): a job left open by an exception is joined -- every part exactly once -- and the pool takes the next job;
// primed workers are relaxed on the way out; usable_cpus() stays within [1, hardware_concurrency]
static int check_guards(int nworkers)
{
    WorkPool pool;
    pool.start(nworkers);
    std::vector<std::atomic<int>> hits(16);
    for (int round = 0; round < 200; round++) {
        for (auto &h : hits) h.store(0);
        const int parts = 1 + round % (nworkers + 1);
        try {
            rmdf::PoolPrimeGuard hot(pool, 500);
            rmdf::PoolJobGuard guard(pool);
            pool.begin(parts, [&](int part) { hits[(size_t)part].fetch_add(1); });
            if (round % 3 == 0) throw std::bad_alloc();
            pool.finish();
        } catch (const std::bad_alloc &) { }
        for (int p = 0; p < 16; p++)
            if (hits[(size_t)p].load() != (p < parts ? 1 : 0)) { printf("guard, pool of %d, round %d: part %d ran %d times\n", nworkers, round, p, hits[(size_t)p].load()); return 1; }
        int ran = 0;
        pool.run(parts, [&](int) { __atomic_fetch_add(&ran, 1, __ATOMIC_RELAXED); });
        if (ran != parts) { printf("guard, pool of %d, round %d: the job after the guard ran %d of %d parts\n", nworkers, round, ran, parts); return 1; }
    }
    const int c = rmdf::usable_cpus();
    if (c < 1 || c > (int)std::thread::hardware_concurrency()) { printf("usable_cpus() = %d\n", c); return 1; }
    return 0;
}

int main()
{
    for (int n : { 0, 1, 3, 15 })
        if (check_pool(n) || check_guards(n)) return 1;
    { WorkPool idle; idle.prime(10); idle.relax(); idle.copy(nullptr, nullptr, 0); }      // no workers at all
    printf("pool ok\n");
    return 0;
}
