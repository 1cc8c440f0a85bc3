/*
 * rmdf_xcheck.h -- additions of librmdf_xcheck.so, the cross-check / measurement build of the renderer.
 *
 * librmdf_xcheck.so is built from the same sources as librmdf.so (make -C .../csrc xcheck) plus ONE alternative
 * schedule of the same per-ray arithmetic (csrc/xcheck/rmdf_march.hip: both loops flattened into a per-lane state machine, shading
 * in a second kernel behind a G-buffer, the compiler's divisions everywhere): slower than the product kernel
 * and kept so that an independently scheduled implementation can be compared bit for bit in the tests.  (Rounds 1-3 carried two
 * more -- a wave-local ray pool and a march-with-refill pipeline; both measured slower, removed in round 4, in the history.)
 * Nothing here is part of the drop-in boundary; the product library rejects these flag bits.
 * The alternative schedule keeps ONE scratch set (G-buffer, work counter) per ctx: calls must use the ctx stream
 * (stream = NULL) -- another stream returns RMDF_E_UNSUPPORTED -- so frames in flight are not available with it.
 */
#ifndef RMDF_XCHECK_H
#define RMDF_XCHECK_H

#include "rmdf.h"

#ifdef __cplusplus
extern "C" {
#endif

/* rmdf_config.reserved[0] */
#define RMDF_FLAG_NESTED_LOOPS 1   /* no-op (the default render kernel)                                                      */
#define RMDF_FLAG_FLAT_MARCH   2   /* Mandelbulb power 8: flattened march kernel + shade kernel (xcheck/rmdf_march.hip)     */
#define RMDF_FLAG_FORCE_WRITTEN 64 /* power-8 Mandelbulb: the folded iteration passes' underflow guard always trips, so every ray, normal and
                                      AO estimate of the product kernel takes its written fall-back (tests: the frame must not change) */

/* Measurement aid: per-wave counters of the Mandelbulb march kernels.  enable != 0 switches collection on
 * (off: frees the buffer); out (may be NULL) receives 16 uint64 per wave for the launches since the last read:
 * iteration passes, march-tail passes, shade-tail passes, refill rounds, sum of iterating lanes over iteration
 * passes, sum of waiting lanes over march tails, begin / end timestamps (100 MHz s_memrealtime ticks). */
int rmdf_debug_march_stats(rmdf_ctx *ctx, int enable, uint64_t *out, int max_waves);

/* Host-only check aid: the Cornell box's candidate grid of n^3 cells (n = 16, 32, 64, 128; rmdf_device.hpp: the set of triangles
 * that can be nearest somewhere in a cell), built with every triangle measured in every cell (brute_force != 0) or by halving
 * cells from the 16^3 grid the way rmdf_create builds its 64^3 one.  out: n^3 uint32 masks. */
int rmdf_debug_cornell_masks(int n, int brute_force, uint32_t *out);

/* Host-only check aid: the per-triangle table of the Cornell box's distance estimate as rmdf_create uploads it (rmdf_device.hpp:
 * CORNELL_STRIDE floats per triangle -- the reference's constants, then the pruning planes of the per-lane estimate: the triangle's
 * plane and its three edge planes, four floats each from float CORNELL_BOUNDS on).  out: 32 * stride floats; *stride and *bounds
 * (either may be NULL) receive the two offsets.  tests/test_host_logic.py holds every plane to "a lower bound of the distance". */
int rmdf_debug_cornell_table(float *out, int *stride, int *bounds);
/* ... and the 32 x 8 floats that follow the rows in device memory: per triangle its plane (normal, offset) and a bounding sphere (centre,
 * radius) -- the bounds of the wave-uniform pruned estimate (rmdf_device.hpp: de_cornell_box_table; cross-check schedules).
 * tests/device_on_host.cpp runs that estimate on the CPU with them. */
int rmdf_debug_cornell_bounds(float *out);

/* Host-only check aids: the tables the env-map kernels only gather through, as the library builds them on the host with glibc's
 * acosf / atanf / cosf / sinf (the functions GHC's Float instances call in the reference).
 * rmdf_debug_cube_uv_table: the environment (u, v) of every texel of the six cw x cw cube faces (cubeMapPixelToDir ->
 * worldToLocal -> cartesianToSpherical -> sphericalToEnvironmentUV, HDREnvMap.hs:76-87,139-147, CoordTransf.hs:35-70);
 * out: 6 * cw * cw * 2 floats, face-major, rows, columns.
 * rmdf_debug_lobe_tables: cosineConvolveHDREnvMap's cos|phi_L - phi_x| per (destination column, source column) and cos / sin theta
 * per row (HDREnvMap.hs:222-239); lutT: ceil(w / 64) * w * 64 floats indexed (block * w + x) * 64 + lane, destination column =
 * min(block * 64 + lane, w - 1); tcs: 2 * h floats.  tests/test_host_logic.py holds both to the oracle / to the formulas. */
int rmdf_debug_cube_uv_table(int cw, float *out);
/* ... and the camera block of main() + lookat (fragment.shd:829-838, 883-902), which the library evaluates once per frame on the host
 * (host libm sinf / cosf / tanf) instead of once per pixel: xaxis, yaxis, zaxis, eye (12 floats) and tan(hfov / 2). */
int rmdf_debug_camera(int scene, float time, float cam[12], float *fov_xs);
/* ... and the Radiance (.hdr) reader and writer behind rmdf_load_env_hdr -- loadHDRImage (HDREnvMap.hs:31-52) and JP.saveRadianceImage
 * (ShaderRendering.hs:147): flat and new-style run-length coded scanlines, JuicyPixels' RGBE <-> Float arithmetic.  The reader parses
 * bytes from disk; tests/test_host_logic.py compares both with the oracle's and feeds the reader truncated and corrupted files.
 * rmdf_debug_hdr_decode: out may be NULL (size query: *w, *h); cap_floats >= 3 * w * h.  rmdf_debug_hdr_encode returns the file
 * image's length (header + 4 bytes per texel), or a negative RMDF_E_* code. */
int  rmdf_debug_hdr_decode(const uint8_t *file, size_t len, int *w, int *h, float *out, size_t cap_floats);
long rmdf_debug_hdr_encode(const float *rgb, int w, int h, uint8_t *out, size_t cap);
int rmdf_debug_lobe_tables(int w, int h, float *lutT, float *tcs);

#ifdef __cplusplus
}
#endif
#endif /* RMDF_XCHECK_H */
