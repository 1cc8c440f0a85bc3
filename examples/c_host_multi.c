/* c_host_multi.c -- the multi-GPU path from a host in plain C: one PROCESS per GPU, no Python, no torch, no HIP binding in
 * the host.  What bench.py does for N > 1, through include/rmdf.h alone:
 *
 *   every rank:  rmdf_create(device = rank), rmdf_load_env_hdr, rmdf_probe_tile_costs + rmdf_set_shard_costs (every rank
 *                computes the same deal of the reference's 64 tiles by itself)
 *   rank 0:      rmdf_comm_get_unique_id -> shared page -> the other ranks         (any channel would do)
 *   every rank:  rmdf_comm_init (RCCL, one communicator over the node's GPUs)
 *   per frame:   rmdf_render_frame_sharded_device = this rank's shard + the ONE gather over xGMI + rank 0's assembly
 *   rank 0:      copies the frame out, compares it with its own single-launch render of the whole frame, writes the PNG
 *
 *   gcc -O2 -I include examples/c_host_multi.c -o c_host_multi -L ray-marching-distance-fields_amd -lrmdf \
 *       -Wl,-rpath,$PWD/ray-marching-distance-fields_amd
 *   ./c_host_multi <latlong.hdr> <out.png> <nranks> [w h frames]
 * The ranks are forked before anything touches the GPU.  On a 1-GPU box only nranks = 1 can run (RCCL cannot put two
 * ranks on one device).
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>

#include "rmdf.h"

typedef struct { volatile int ready; unsigned char id[RMDF_COMM_ID_BYTES]; } shared_page;

#define TRY(ctx, call) do { if ((call) != RMDF_OK) { fprintf(stderr, "rank %d: %s: %s\n", rank, #call, rmdf_last_error(ctx)); return 1; } } while (0)

static double now_ms(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e3 + t.tv_nsec * 1e-6; }

static int run_rank(int rank, int nranks, const char *hdr, const char *png, int w, int h, int frames, shared_page *sh)
{
    const int scene = RMDF_FS_MB_POWER8, ms = 256;
    rmdf_ctx *ctx = NULL;
    rmdf_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = rank;
    if (rmdf_create(&ctx, &cfg) != RMDF_OK) { fprintf(stderr, "rank %d: rmdf_create: %s\n", rank, rmdf_last_error(NULL)); return 1; }
    TRY(ctx, rmdf_load_env_hdr(ctx, hdr));
    float cost[64];
    TRY(ctx, rmdf_probe_tile_costs(ctx, scene, w, h, 0.0, ms, cost));
    TRY(ctx, rmdf_set_shard_costs(ctx, cost));
    if (rank == 0) {
        if (rmdf_comm_get_unique_id(sh->id) != RMDF_OK) { fprintf(stderr, "rank 0: unique id: %s\n", rmdf_last_error(NULL)); return 1; }
        __sync_synchronize();
        sh->ready = 1;
    } else {
        while (!sh->ready) usleep(1000);
        __sync_synchronize();
    }
    const size_t slots = (size_t)((64 + nranks - 1) / nranks), tile = (size_t)(w / 8) * (h / 8) * 4;
    /* the exchange's send / receive calls against this rank itself, on a private one-rank communicator (the ctx has none yet) */
    TRY(ctx, rmdf_comm_selftest_loopback(ctx, slots * tile, NULL, NULL));
    TRY(ctx, rmdf_comm_init(ctx, sh->id, rank, nranks));
    /* collective: every rank probed its costs by itself (the kernels are bit-reproducible); compare the resulting deals once.  On a
     * mismatch fall back to the static deal on this rank -- the verdict is the same on every rank, so all of them do. */
    if (rmdf_comm_verify_deal(ctx, NULL) != RMDF_OK) {
        fprintf(stderr, "rank %d: %s -- using the static deal\n", rank, rmdf_last_error(ctx));
        TRY(ctx, rmdf_set_shard_costs(ctx, NULL));
    }
    void *d_shard = NULL, *d_gathered = NULL, *d_frame = NULL;
    if (rank == 0) {
        TRY(ctx, rmdf_device_malloc(ctx, (size_t)nranks * slots * tile, &d_gathered));
        TRY(ctx, rmdf_device_malloc(ctx, (size_t)w * h * 4, &d_frame));
        d_shard = d_gathered;                            /* the root renders straight into its own slot */
    } else {
        TRY(ctx, rmdf_device_malloc(ctx, slots * tile, &d_shard));
    }
    TRY(ctx, rmdf_render_frame_sharded_device(ctx, scene, w, h, 0.0, ms, d_shard, d_gathered, d_frame, NULL));   /* warm-up */
    TRY(ctx, rmdf_synchronize(ctx, NULL));
    const double t0 = now_ms();
    for (int f = 0; f < frames; f++)
        TRY(ctx, rmdf_render_frame_sharded_device(ctx, scene, w, h, 0.0, ms, d_shard, d_gathered, d_frame, NULL));
    TRY(ctx, rmdf_synchronize(ctx, NULL));
    const double dt = (now_ms() - t0) / frames;
    int rc = 0;
    if (rank == 0) {
        uint32_t *fb = (uint32_t *)malloc((size_t)w * h * 4), *ref = (uint32_t *)malloc((size_t)w * h * 4);
        if (!fb || !ref) return 1;
        TRY(ctx, rmdf_copy_to_host(ctx, fb, d_frame, (size_t)w * h * 4, NULL));
        TRY(ctx, rmdf_render_tile(ctx, scene, -1, w, h, 0.0, ms, ref));
        const int same = memcmp(fb, ref, (size_t)w * h * 4) == 0;
        if (rmdf_save_png(png, fb, w, h) != RMDF_OK) { fprintf(stderr, "png: %s\n", rmdf_last_error(NULL)); rc = 1; }
        int r = -1, n = -1;
        rmdf_comm_info(ctx, &r, &n);
        printf("%d rank(s) (RCCL communicator: rank %d of %d): %dx%d, %.3f ms per frame (one frame at a time), sharded == single launch: %s, wrote %s\n",
               nranks, r, n, w, h, dt, same ? "yes" : "NO", png);
        if (!same) rc = 1;
        free(fb); free(ref);
        rmdf_device_free(ctx, d_frame);
        rmdf_device_free(ctx, d_gathered);
    } else {
        rmdf_device_free(ctx, d_shard);
    }
    rmdf_comm_destroy(ctx);
    rmdf_destroy(ctx);
    return rc;
}

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: %s <latlong.hdr> <out.png> <nranks> [w h frames]\n", argv[0]); return 2; }
    const int nranks = atoi(argv[3]);
    const int w = argc > 4 ? atoi(argv[4]) : 1920, h = argc > 5 ? atoi(argv[5]) : 1080, frames = argc > 6 ? atoi(argv[6]) : 20;
    if (nranks < 1 || nranks > 64 || w % 8 || h % 8 || w <= 0 || h <= 0 || frames < 1) { fprintf(stderr, "bad arguments (w, h divisible by 8)\n"); return 2; }
    shared_page *sh = (shared_page *)mmap(NULL, sizeof *sh, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    if (sh == MAP_FAILED) { perror("mmap"); return 1; }
    memset(sh, 0, sizeof *sh);
    /* fork the ranks BEFORE the first HIP call (rmdf_create makes it): a process that has initialised the GPU must not fork */
    pid_t pids[64];
    for (int r = 1; r < nranks; r++) {
        pids[r] = fork();
        if (pids[r] < 0) { perror("fork"); return 1; }
        if (pids[r] == 0) _exit(run_rank(r, nranks, argv[1], argv[2], w, h, frames, sh));
    }
    int rc = run_rank(0, nranks, argv[1], argv[2], w, h, frames, sh);
    for (int r = 1; r < nranks; r++) {
        int st = 0;
        waitpid(pids[r], &st, 0);
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = 1;
    }
    return rc;
}
