// faultlog.c -- LD_PRELOAD recorder for the hunt of the intermittent GPU memory fault (NOTEBOOK.md A.5).
// Keeps a ring of every device / pinned-host allocation, free, page lock and large copy the PROCESS makes through the HIP and HSA entry
// points (librmdf, RCCL, torch and the HIP runtime's own calls into libhsa-runtime64 alike), and when the ROCr fault handler abort()s
// writes the ring, the call that was running and /proc/self/maps to $RMDF_FAULTLOG_DIR/fault_<pid>.txt.  ROCr prints the faulting
// address on stderr just before; tools/faultlog/resolve.py puts the two together.
// Build: gcc -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include faultlog.c -o libfaultlog.so -ldl
#define _GNU_SOURCE
#include <dlfcn.h>
#include <fcntl.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#define RING (1u << 17)
struct Ent { uint64_t ns; const char *what; uint64_t a, b, c; void *ra; };
static struct Ent g_ring[RING];
static volatile uint64_t g_n;

static uint64_t now_ns(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return (uint64_t)t.tv_sec * 1000000000ull + (uint64_t)t.tv_nsec; }
static void rec(const char *what, uint64_t a, uint64_t b, uint64_t c, void *ra)
{
    const uint64_t i = __atomic_fetch_add(&g_n, 1, __ATOMIC_RELAXED);
    struct Ent *e = &g_ring[i % RING];
    e->ns = now_ns(); e->what = what; e->a = a; e->b = b; e->c = c; e->ra = ra;
}

static void dump(int sig)
{
    char fn[512], line[256];
    const char *dir = getenv("RMDF_FAULTLOG_DIR");
    snprintf(fn, sizeof fn, "%s/fault_%d.txt", dir ? dir : ".", (int)getpid());
    const int fd = open(fn, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd >= 0) {
        const uint64_t n = g_n, lo = n > RING ? n - RING : 0;
        int k = snprintf(line, sizeof line, "signal %d pid %d entries %llu now_ns %llu\n", sig, (int)getpid(), (unsigned long long)n, (unsigned long long)now_ns());
        if (write(fd, line, (size_t)k) < 0) {}
        for (uint64_t i = lo; i < n; i++) {
            const struct Ent *e = &g_ring[i % RING];
            k = snprintf(line, sizeof line, "%llu %s 0x%llx 0x%llx 0x%llx ra=%p\n", (unsigned long long)e->ns, e->what ? e->what : "?",
                         (unsigned long long)e->a, (unsigned long long)e->b, (unsigned long long)e->c, e->ra);
            if (write(fd, line, (size_t)k) < 0) {}
        }
        if (write(fd, "== maps\n", 8) < 0) {}
        const int m = open("/proc/self/maps", O_RDONLY);
        if (m >= 0) { char buf[65536]; ssize_t r; while ((r = read(m, buf, sizeof buf)) > 0) if (write(fd, buf, (size_t)r) < 0) break; close(m); }
        close(fd);
    }
    signal(sig, SIG_DFL);
    raise(sig);
}

__attribute__((constructor)) static void init(void)
{
    if (!getenv("RMDF_FAULTLOG_DIR")) return;
    struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_handler = dump; sigaction(SIGABRT, &sa, NULL); sigaction(SIGSEGV, &sa, NULL); sigaction(SIGBUS, &sa, NULL);
}

// The real entry point: from the copy of the library that is ALREADY in the process (python imports torch's bundled ROCm libraries with
// RTLD_LOCAL: RTLD_NEXT does not see them), first in load order; RTLD_NEXT as the fall-back for plain hosts.
#include <link.h>
struct find_lib { const char *part; char path[512]; };
static int find_lib_cb(struct dl_phdr_info *info, size_t size, void *data)
{
    struct find_lib *f = (struct find_lib *)data;
    (void)size;
    if (!f->path[0] && info->dlpi_name && strstr(info->dlpi_name, f->part) && !strstr(info->dlpi_name, "libfaultlog")) snprintf(f->path, sizeof f->path, "%s", info->dlpi_name);
    return 0;
}
static void *real_sym(const char *lib_part, const char *name)
{
    struct find_lib f; f.part = lib_part; f.path[0] = 0;
    dl_iterate_phdr(find_lib_cb, &f);
    void *p = NULL;
    if (f.path[0]) { void *h = dlopen(f.path, RTLD_NOW | RTLD_NOLOAD); if (h) p = dlsym(h, name); }
    if (!p) p = dlsym(RTLD_NEXT, name);
    if (!p) { fprintf(stderr, "faultlog: cannot resolve %s\n", name); abort(); }
    return p;
}
#define NEXT(name) static __typeof__(&name) real; if (!real) real = (__typeof__(&name))real_sym(#name[1] == 's' ? "libhsa-runtime64" : "libamdhip64", #name)
#define RA __builtin_return_address(0)

hipError_t hipMalloc(void **p, size_t n) { NEXT(hipMalloc); hipError_t e = real(p, n); rec("hipMalloc", p ? (uint64_t)*p : 0, n, e, RA); return e; }
hipError_t hipFree(void *p) { NEXT(hipFree); rec("hipFree>", (uint64_t)p, 0, 0, RA); hipError_t e = real(p); rec("hipFree<", (uint64_t)p, 0, e, RA); return e; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned f) { NEXT(hipHostMalloc); hipError_t e = real(p, n, f); rec("hipHostMalloc", p ? (uint64_t)*p : 0, n, f, RA); return e; }
hipError_t hipHostFree(void *p) { NEXT(hipHostFree); rec("hipHostFree>", (uint64_t)p, 0, 0, RA); hipError_t e = real(p); rec("hipHostFree<", (uint64_t)p, 0, e, RA); return e; }
hipError_t hipHostRegister(void *p, size_t n, unsigned f) { NEXT(hipHostRegister); hipError_t e = real(p, n, f); rec("hipHostRegister", (uint64_t)p, n, e, RA); return e; }
hipError_t hipHostUnregister(void *p) { NEXT(hipHostUnregister); hipError_t e = real(p); rec("hipHostUnregister", (uint64_t)p, 0, e, RA); return e; }
hipError_t hipExtMallocWithFlags(void **p, size_t n, unsigned f) { NEXT(hipExtMallocWithFlags); hipError_t e = real(p, n, f); rec("hipExtMallocWithFlags", p ? (uint64_t)*p : 0, n, f, RA); return e; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st)
{
    NEXT(hipMemcpyAsync);
    rec(k == hipMemcpyHostToDevice ? "hipMemcpyAsync.H2D>" : k == hipMemcpyDeviceToHost ? "hipMemcpyAsync.D2H>" : "hipMemcpyAsync.other>", (uint64_t)d, (uint64_t)s, n, RA);
    hipError_t e = real(d, s, n, k, st);
    rec("hipMemcpyAsync<", (uint64_t)d, (uint64_t)st, e, RA);
    return e;
}
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k)
{
    NEXT(hipMemcpy);
    rec(k == hipMemcpyHostToDevice ? "hipMemcpy.H2D>" : k == hipMemcpyDeviceToHost ? "hipMemcpy.D2H>" : "hipMemcpy.other>", (uint64_t)d, (uint64_t)s, n, RA);
    hipError_t e = real(d, s, n, k);
    rec("hipMemcpy<", (uint64_t)d, 0, e, RA);
    return e;
}
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned f) { NEXT(hipStreamCreateWithFlags); hipError_t e = real(s, f); rec("hipStreamCreateWithFlags", s ? (uint64_t)*s : 0, f, e, RA); return e; }
hipError_t hipStreamDestroy(hipStream_t s) { NEXT(hipStreamDestroy); rec("hipStreamDestroy>", (uint64_t)s, 0, 0, RA); hipError_t e = real(s); rec("hipStreamDestroy<", (uint64_t)s, 0, e, RA); return e; }
hipError_t hipStreamSynchronize(hipStream_t s) { NEXT(hipStreamSynchronize); rec("hipStreamSynchronize>", (uint64_t)s, 0, 0, RA); hipError_t e = real(s); rec("hipStreamSynchronize<", (uint64_t)s, 0, e, RA); return e; }
hipError_t hipDeviceSynchronize(void) { NEXT(hipDeviceSynchronize); rec("hipDeviceSynchronize>", 0, 0, 0, RA); hipError_t e = real(); rec("hipDeviceSynchronize<", 0, 0, e, RA); return e; }
hipError_t hipLaunchKernel(const void *f, dim3 g, dim3 b, void **args, size_t shm, hipStream_t st)
{
    NEXT(hipLaunchKernel);
    rec("hipLaunchKernel", (uint64_t)f, ((uint64_t)g.x << 32) | ((uint64_t)g.y << 16) | g.z, (uint64_t)st, RA);
    return real(f, g, b, args, shm, st);
}

// the HIP runtime's own traffic with ROCr (device memory, page locks of pageable host memory)
hsa_status_t hsa_amd_memory_pool_allocate(hsa_amd_memory_pool_t pool, size_t size, uint32_t flags, void **ptr)
{
    NEXT(hsa_amd_memory_pool_allocate);
    hsa_status_t s = real(pool, size, flags, ptr);
    rec("hsa_pool_allocate", ptr ? (uint64_t)*ptr : 0, size, pool.handle, RA);
    return s;
}
hsa_status_t hsa_amd_memory_pool_free(void *ptr) { NEXT(hsa_amd_memory_pool_free); rec("hsa_pool_free>", (uint64_t)ptr, 0, 0, RA); hsa_status_t s = real(ptr); rec("hsa_pool_free<", (uint64_t)ptr, 0, s, RA); return s; }
hsa_status_t hsa_amd_memory_lock(void *host, size_t size, hsa_agent_t *agents, int n, void **agent_ptr)
{
    NEXT(hsa_amd_memory_lock);
    hsa_status_t s = real(host, size, agents, n, agent_ptr);
    rec("hsa_memory_lock", (uint64_t)host, size, agent_ptr ? (uint64_t)*agent_ptr : 0, RA);
    return s;
}
hsa_status_t hsa_amd_memory_lock_to_pool(void *host, size_t size, hsa_agent_t *agents, int n, hsa_amd_memory_pool_t pool, uint32_t flags, void **agent_ptr)
{
    NEXT(hsa_amd_memory_lock_to_pool);
    hsa_status_t s = real(host, size, agents, n, pool, flags, agent_ptr);
    rec("hsa_memory_lock_to_pool", (uint64_t)host, size, agent_ptr ? (uint64_t)*agent_ptr : 0, RA);
    return s;
}
hsa_status_t hsa_amd_memory_unlock(void *host) { NEXT(hsa_amd_memory_unlock); rec("hsa_memory_unlock>", (uint64_t)host, 0, 0, RA); hsa_status_t s = real(host); rec("hsa_memory_unlock<", (uint64_t)host, 0, s, RA); return s; }
hsa_status_t hsa_executable_freeze(hsa_executable_t ex, const char *opt) { NEXT(hsa_executable_freeze); hsa_status_t s = real(ex, opt); rec("hsa_executable_freeze", ex.handle, 0, s, RA); return s; }
hsa_status_t hsa_executable_destroy(hsa_executable_t ex) { NEXT(hsa_executable_destroy); rec("hsa_executable_destroy", ex.handle, 0, 0, RA); return real(ex); }
