/* c_host.c -- a host in plain C on top of include/rmdf.h: what the reference's App.draw does for a shader mode
 * (drawShaderTile into the frame buffer that fillFrameBuffer hands out, tile by tile, then the screenshot key),
 * with no Python and no torch in the process.
 *
 *   gcc -O2 -I include examples/c_host.c -o c_host -L ray-marching-distance-fields_amd -lrmdf \
 *       -Wl,-rpath,$PWD/ray-marching-distance-fields_amd
 *   ./c_host ray-marching-distance-fields_amd/data/latlong_envmaps/uffizi_512.hdr out.png [scene] [w] [h]
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rmdf.h"

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s <latlong.hdr> <out.png> [scene 0..3] [w] [h]\n", argv[0]); return 2; }
    const int scene = argc > 3 ? atoi(argv[3]) : RMDF_FS_MB_POWER8;
    const int w = argc > 4 ? atoi(argv[4]) : 640, h = argc > 5 ? atoi(argv[5]) : 360;
    rmdf_ctx *ctx = NULL;
    rmdf_config cfg;
    memset(&cfg, 0, sizeof cfg);
    if (rmdf_create(&ctx, &cfg) != RMDF_OK) { fprintf(stderr, "rmdf_create: %s\n", rmdf_last_error(NULL)); return 1; }
    if (rmdf_load_env_hdr(ctx, argv[1]) != RMDF_OK) { fprintf(stderr, "env: %s\n", rmdf_last_error(ctx)); rmdf_destroy(ctx); return 1; }
    uint32_t *fb = (uint32_t *)malloc((size_t)w * h * 4);             /* the MVector Word32 of fillFrameBuffer */
    if (!fb) { rmdf_destroy(ctx); return 1; }
    /* tiled mode as the viewer runs it: one tile per displayed frame, time latched on the first tile */
    int idx = 0;
    do {
        if (rmdf_render_tile(ctx, scene, idx, w, h, 1.5, scene == RMDF_FS_MB_POWER8 ? 256 : 128, fb) != RMDF_OK) {
            fprintf(stderr, "tile %d: %s\n", idx, rmdf_last_error(ctx));
            free(fb); rmdf_destroy(ctx); return 1;
        }
    } while (!rmdf_is_tile_idx_last_tile(idx++));
    /* and once untiled (Nothing): must give the same frame */
    uint32_t *fb2 = (uint32_t *)malloc((size_t)w * h * 4);
    int same = 0;
    if (fb2 && rmdf_render_tile(ctx, scene, -1, w, h, 1.5, scene == RMDF_FS_MB_POWER8 ? 256 : 128, fb2) == RMDF_OK)
        same = memcmp(fb, fb2, (size_t)w * h * 4) == 0;
    if (rmdf_save_png(argv[2], fb, w, h) != RMDF_OK) { fprintf(stderr, "png: %s\n", rmdf_last_error(NULL)); free(fb); free(fb2); rmdf_destroy(ctx); return 1; }
    char name[256];
    int cus = 0;
    rmdf_device_info(ctx, name, (int)sizeof name, &cus);
    printf("%s (%d CUs): scene %d %dx%d, 64 tiles == untiled: %s, wrote %s\n", name, cus, scene, w, h, same ? "yes" : "NO", argv[2]);
    free(fb); free(fb2);
    rmdf_destroy(ctx);
    return same ? 0 : 1;
}
