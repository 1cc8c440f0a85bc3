/* fake_rccl.c -- a TEST DOUBLE of the eight RCCL entry points librmdf uses (rmdf_api.cpp: load_rccl), for exercising the N > 1 branches of
 * the exchange step with N processes that share ONE GPU (the builder's lease has one; a real communicator needs one GPU per rank).
 * Loaded only by librmdf_xcheck.so, and only when RMDF_RCCL_LIB names it.  It moves the same bytes between the same buffers in the
 * same order as ncclSend / ncclRecv / ncclGroupStart / ncclGroupEnd would -- through files in /dev/shm instead of xGMI -- and it is
 * stricter than RCCL where that helps a test: a receive whose size differs from the matching send fails (ncclInvalidArgument) instead
 * of hanging, and a receive nobody answers fails after FAKE_RCCL_TIMEOUT_S seconds (default 60) with ncclSystemError.
 * What it does NOT do: run asynchronously on the stream (every operation completes before the call -- or ncclGroupEnd -- returns, after
 * a hipStreamSynchronize of its stream), use xGMI, or tell anything about bandwidth.  It is readiness evidence, not a scaling number.
 *
 * Message (src -> dst, sequence number q of that ordered pair on that communicator): the sender writes
 * /dev/shm/fakerccl_<uid>_<src>_<dst>_<q>.tmp and renames it to ...msg; the receiver waits for the file, checks its size, copies it to
 * the device and unlinks it.  Sends never wait (the gather's peers send, the root receives: no rendezvous needed).
 * Build: gcc -O2 -fPIC -shared -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include fake_rccl.c -o libfake_rccl.so -L/opt/rocm/lib -lamdhip64 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#ifdef FAKE_RCCL_NO_GPU
/* CPU-tier self-test of the double itself (tests/test_host_logic.py): "device" buffers are host memory, streams do not exist */
#define hipStreamSynchronize(s) ((void)(s), hipSuccess)
#define hipMemcpy(d, s, n, k) (memcpy((d), (s), (n)), hipSuccess)
#define hipHostMalloc(pp, n, f) ((*(pp) = malloc(n)) ? hipSuccess : hipErrorOutOfMemory)
#define hipHostFree(p) (free(p), hipSuccess)
#endif

struct ncclComm {
    char uid[33];
    int rank, nranks;
    unsigned long long seq_out[64], seq_in[64];
    /* ONE page-locked staging buffer per communicator, grown on demand, freed by ncclCommDestroy: the double never hands pageable
     * memory to the HIP runtime (a short-lived malloc()ed block copied with hipMemcpy is page-locked on the fly by the runtime -- the
     * pattern round 5's GPU memory fault was traced to, NOTEBOOK.md A.5) */
    void *stage;
    size_t stage_cap;
};

static void *stage_of(struct ncclComm *c, size_t bytes)
{
    if (bytes <= c->stage_cap && c->stage) return c->stage;
    if (c->stage) { (void)hipHostFree(c->stage); c->stage = NULL; c->stage_cap = 0; }
    size_t cap = bytes < 4096 ? 4096 : bytes;
    void *p = NULL;
    if (hipHostMalloc(&p, cap, 0) != hipSuccess || !p) return NULL;
    c->stage = p; c->stage_cap = cap;
    return p;
}

enum { OP_SEND, OP_RECV };
struct op { int kind; void *buf; size_t bytes; int peer; struct ncclComm *comm; hipStream_t stream; };
static __thread int g_depth;
static __thread struct op g_ops[256];
static __thread int g_nops;
static __thread char g_err[512];

static size_t type_size(ncclDataType_t t)
{
    switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 1;
    }
}

static void path_of(char *out, size_t n, const struct ncclComm *c, int src, int dst, unsigned long long q, const char *ext)
{
    snprintf(out, n, "/dev/shm/fakerccl_%s_%d_%d_%llu.%s", c->uid, src, dst, q, ext);
}

static ncclResult_t do_send(const struct op *o)
{
    struct ncclComm *c = o->comm;
    if (hipStreamSynchronize(o->stream) != hipSuccess) { snprintf(g_err, sizeof g_err, "fake ncclSend: stream synchronize failed"); return ncclUnhandledCudaError; }
    void *host = stage_of(c, o->bytes);
    if (!host) { snprintf(g_err, sizeof g_err, "fake ncclSend: no page-locked staging buffer of %zu bytes", o->bytes); return ncclSystemError; }
    if (o->bytes && hipMemcpy(host, o->buf, o->bytes, hipMemcpyDeviceToHost) != hipSuccess) { snprintf(g_err, sizeof g_err, "fake ncclSend: device read failed"); return ncclUnhandledCudaError; }
    char tmp[256], msg[256];
    const unsigned long long q = c->seq_out[o->peer]++;
    path_of(tmp, sizeof tmp, c, c->rank, o->peer, q, "tmp");
    path_of(msg, sizeof msg, c, c->rank, o->peer, q, "msg");
    const int fd = open(tmp, O_WRONLY | O_CREAT | O_TRUNC, 0600);
    if (fd < 0) { snprintf(g_err, sizeof g_err, "fake ncclSend: cannot create %s", tmp); return ncclSystemError; }
    size_t done = 0;
    while (done < o->bytes) { const ssize_t w = write(fd, (char *)host + done, o->bytes - done); if (w <= 0) break; done += (size_t)w; }
    close(fd);
    if (done != o->bytes || rename(tmp, msg) != 0) { unlink(tmp); snprintf(g_err, sizeof g_err, "fake ncclSend: short write to %s", tmp); return ncclSystemError; }
    return ncclSuccess;
}

static ncclResult_t do_recv(const struct op *o)
{
    struct ncclComm *c = o->comm;
    char msg[256];
    const unsigned long long q = c->seq_in[o->peer]++;
    path_of(msg, sizeof msg, c, o->peer, c->rank, q, "msg");
    const char *ts = getenv("FAKE_RCCL_TIMEOUT_S");
    const double limit = ts ? atof(ts) : 60.0;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    struct stat sb;
    for (;;) {
        if (stat(msg, &sb) == 0) break;
        clock_gettime(CLOCK_MONOTONIC, &t1);
        if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > limit) {
            snprintf(g_err, sizeof g_err, "fake ncclRecv: rank %d waited %.0f s for message %llu of rank %d", c->rank, limit, q, o->peer);
            return ncclSystemError;
        }
        usleep(200);
    }
    if ((size_t)sb.st_size != o->bytes) {
        unlink(msg);
        snprintf(g_err, sizeof g_err, "fake ncclRecv: rank %d expects %zu bytes from rank %d, which sent %zu", c->rank, o->bytes, o->peer, (size_t)sb.st_size);
        return ncclInvalidArgument;
    }
    void *host = stage_of(c, o->bytes);
    if (!host) { unlink(msg); snprintf(g_err, sizeof g_err, "fake ncclRecv: no page-locked staging buffer of %zu bytes", o->bytes); return ncclSystemError; }
    const int fd = open(msg, O_RDONLY);
    size_t done = 0;
    while (fd >= 0 && done < o->bytes) { const ssize_t r = read(fd, (char *)host + done, o->bytes - done); if (r <= 0) break; done += (size_t)r; }
    if (fd >= 0) close(fd);
    unlink(msg);
    ncclResult_t res = ncclSuccess;
    if (done != o->bytes) { snprintf(g_err, sizeof g_err, "fake ncclRecv: short read of %s", msg); res = ncclSystemError; }
    else if (hipStreamSynchronize(o->stream) != hipSuccess || (o->bytes && hipMemcpy(o->buf, host, o->bytes, hipMemcpyHostToDevice) != hipSuccess)) {
        snprintf(g_err, sizeof g_err, "fake ncclRecv: device write failed"); res = ncclUnhandledCudaError;
    }
    return res;
}

static ncclResult_t submit(int kind, void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream)
{
    if (!comm || peer < 0 || peer >= comm->nranks) { snprintf(g_err, sizeof g_err, "fake rccl: bad communicator or peer %d", peer); return ncclInvalidArgument; }
    struct op o = { kind, buf, count * type_size(t), peer, comm, stream };
    if (g_depth > 0) {
        if (g_nops >= 256) return ncclInternalError;
        g_ops[g_nops++] = o;
        return ncclSuccess;
    }
    return kind == OP_SEND ? do_send(&o) : do_recv(&o);
}

ncclResult_t ncclGetUniqueId(ncclUniqueId *id)
{
    memset(id, 0, sizeof *id);
    unsigned char r[16];
    const int fd = open("/dev/urandom", O_RDONLY);
    if (fd < 0 || read(fd, r, sizeof r) != (ssize_t)sizeof r) { if (fd >= 0) close(fd); return ncclSystemError; }
    close(fd);
    for (int i = 0; i < 16; i++) snprintf(id->internal + 2 * i, 3, "%02x", r[i]);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank)
{
    if (!comm || nranks < 1 || nranks > 64 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
    struct ncclComm *c = (struct ncclComm *)calloc(1, sizeof *c);
    if (!c) return ncclSystemError;
    memcpy(c->uid, id.internal, 32);
    c->uid[32] = 0;
    for (int i = 0; i < 32; i++) if (!((c->uid[i] >= '0' && c->uid[i] <= '9') || (c->uid[i] >= 'a' && c->uid[i] <= 'f'))) c->uid[i] = 'x';
    c->rank = rank; c->nranks = nranks;
    *comm = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm)
{
    if (!comm) return ncclSuccess;
    /* messages this rank was sent and never received (a test that stopped half way): remove them */
    for (int src = 0; src < comm->nranks; src++)
        for (unsigned long long q = comm->seq_in[src]; q < comm->seq_in[src] + 64; q++) {
            char msg[256];
            path_of(msg, sizeof msg, comm, src, comm->rank, q, "msg");
            if (unlink(msg) != 0) break;
        }
    if (comm->stage) (void)hipHostFree(comm->stage);
    free(comm);
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r)
{
    if (r == ncclSuccess) return "no error";
    return g_err[0] ? g_err : "fake rccl error";
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream)
{
    return submit(OP_SEND, (void *)buf, count, t, peer, comm, stream);
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t stream)
{
    return submit(OP_RECV, buf, count, t, peer, comm, stream);
}

ncclResult_t ncclGroupStart(void) { g_depth++; return ncclSuccess; }

ncclResult_t ncclGroupEnd(void)
{
    if (g_depth <= 0) return ncclInvalidUsage;
    if (--g_depth > 0) return ncclSuccess;
    ncclResult_t res = ncclSuccess;
    /* the sends first (they never wait), then the receives */
    for (int pass = 0; pass < 2; pass++)
        for (int i = 0; i < g_nops; i++) {
            if ((g_ops[i].kind == OP_SEND) != (pass == 0)) continue;
            const ncclResult_t r = g_ops[i].kind == OP_SEND ? do_send(&g_ops[i]) : do_recv(&g_ops[i]);
            if (r != ncclSuccess && res == ncclSuccess) res = r;
        }
    g_nops = 0;
    return res;
}
