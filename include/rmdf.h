/*
 * rmdf.h -- C ABI of librmdf.so, the MI355X (gfx950) sphere-tracing renderer.
 *
 * This is the drop-in boundary for ONE path of blitzcode/ray-marching-distance-
 * fields: the per-pixel ray march that the reference runs as a GLSL fragment
 * shader behind `ShaderRendering.drawShaderTile` (ShaderRendering.hs:151-196),
 * delivered through the host-pointer slot the reference already has for CPU
 * renderers, `FrameBuffer.fillFrameBuffer` (FrameBuffer.hs:117-158).  Every
 * entry point cites the reference interface it replaces.  INTEGRATION.md shows
 * the Haskell `foreign import ccall` binding a maintainer would add.
 *
 * Conventions
 *   - plain C types only; the caller owns every host buffer it passes, the
 *     library owns all device memory; nothing is retained across calls except
 *     through the rmdf_ctx.
 *   - every function returns 0 on success or a negative RMDF_E_* code; it never
 *     throws or aborts across the ABI.  rmdf_last_error() returns the message
 *     (the `String` of the reference's `Either String` / `ExceptT String`,
 *     ShaderRendering.hs:63,110; App.hs:246-256).  A failed render leaves the
 *     previously accumulated frame intact.
 *   - a ctx is used by one thread at a time (the reference does all GL work on
 *     its bound main thread, App.hs:287-307).  Host-pointer calls block until
 *     the output is complete: import them `safe` in Haskell.
 *   - pixels: little-endian uint32 = bytes R,G,B,A; index px + py*w; row 0 is the
 *     BOTTOM row (gl_FragCoord origin; FrameBuffer.saveFrameBufferToPNG flips,
 *     FrameBuffer.hs:222-227).
 *   - there is no CPU fallback: without a usable HIP device rmdf_create fails
 *     with RMDF_E_NO_DEVICE.
 */
#ifndef RMDF_H
#define RMDF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rmdf_ctx rmdf_ctx;

/* error codes */
#define RMDF_OK              0
#define RMDF_E_INVALID      -1   /* bad argument                                   */
#define RMDF_E_NO_DEVICE    -2   /* no HIP device / wrong architecture             */
#define RMDF_E_HIP          -3   /* a HIP runtime call failed                      */
#define RMDF_E_IO           -4   /* file missing / unreadable / malformed          */
#define RMDF_E_NO_ENV       -5   /* a cube map the scene samples has not been set  */
#define RMDF_E_UNSUPPORTED  -6   /* not available in this build / on this stream   */
#define RMDF_E_NOMEM        -7
#define RMDF_E_COMM         -8   /* an RCCL call failed / no communicator          */

/* `data FragmentShader = FSDECornellBoxShader | FSDETestShader | FSMBPower8Shader |
 * FSMBGeneralShader deriving Enum` (ShaderRendering.hs:46-47) */
#define RMDF_FS_DE_CORNELL_BOX 0
#define RMDF_FS_DE_TEST        1
#define RMDF_FS_MB_POWER8      2
#define RMDF_FS_MB_GENERAL     3

/* cube-map slots = the shader's samplerCube uniforms (fragment.shd:10-14;
 * srEnvCubeMaps, ShaderRendering.hs:83-91) */
#define RMDF_ENV_REFLECTION 0
#define RMDF_ENV_COS_1      1
#define RMDF_ENV_COS_8      2
#define RMDF_ENV_COS_64     3
#define RMDF_ENV_COS_512    4
#define RMDF_ENV_SLOTS      5

/* tilesX, tilesY, nTiles (ShaderRendering.hs:49-52) */
#define RMDF_TILES_X 8
#define RMDF_TILES_Y 8
#define RMDF_N_TILES 64

/* rmdf_config.reserved[0] flags: each switches ONE optimisation of the render kernel off, so that tests can show it does
 * not change a bit of the output.  (Bits 1, 2 and 8 select alternative schedules that only librmdf_xcheck.so contains, see
 * include/rmdf_xcheck.h; librmdf.so rejects them with RMDF_E_UNSUPPORTED.) */
#define RMDF_FLAG_RASTER_ORDER 4    /* always dispatch strips in raster order (no cost feedback from the previous frame)    */
#define RMDF_FLAG_NO_MERGE     16   /* do NOT pool the last rays of a workgroup's four packets in one wave                  */
#define RMDF_FLAG_NO_PRUNE     32   /* Cornell box: evaluate all 32 triangles per distance estimate (no bound-based skipping) */

typedef struct {
    int device;      /* HIP device ordinal                                              */
    int reserved[7]; /* [0] = RMDF_FLAG_* bits; [1] = host threads that copy frames to the caller, the calling thread included
                        (0 = chosen by core count: 16 on >= 64 cores, 8 on > 8); [2] = row bands a whole-frame call into host
                        memory keeps in flight (0 = the library's choice, 1 = one launch; at most 16); [3] = how those bands reach the host: 0 = the
                        library's choice; 1 = one launch per band, the kernel storing the rows into page-locked host memory itself;
                        (0 with [2] given: one launch per band and a copy behind it); 2, 3 = ONE launch that flags every completed
                        band -- librmdf_xcheck.so only (never run on hardware): librmdf.so answers RMDF_E_UNSUPPORTED; rest zero */
} rmdf_config;

/* ---- lifetime: withShaderRenderer (ShaderRendering.hs:60-110) --------------------- */

/* Opens the resource bracket: picks the device, creates the stream, uploads the
 * Cornell-box vertex table (mkCornellBoxVerticesTex, CornellBox.hs:21-46).
 * cfg may be NULL (device 0).
 * Hardware queues: frames in flight on more than four HIP streams need GPU_MAX_HW_QUEUES > 4 (the runtime's default; with it,
 * streams share queues and serialise).  The variable is read when the HIP runtime initialises: rmdf_create sets it to 16 unless
 * the host already set a value, which takes effect only if rmdf_create is the process's first HIP call -- a host that uses HIP
 * before creating the renderer must export GPU_MAX_HW_QUEUES itself.
 * rmdf_create_ex also copies the failure message into err (for hosts whose threads are not bound to one OS thread). */
int rmdf_create(rmdf_ctx **out, const rmdf_config *cfg);
int rmdf_create_ex(rmdf_ctx **out, const rmdf_config *cfg, char *err, size_t err_len);
/* Closes the bracket; frees every device object (ResourceT release, :63-99). */
void rmdf_destroy(rmdf_ctx *ctx);
/* Message of the last failure on ctx.  ctx == NULL: of the last failed call that takes no ctx (rmdf_create, rmdf_save_png, ...)
 * in this PROCESS, whichever thread it failed on; the returned pointer is a per-thread copy. */
const char *rmdf_last_error(const rmdf_ctx *ctx);

/* ---- environment maps (ShaderRendering.hs:65-91, HDREnvMap.hs) --------------------- */

/* The whole env pipeline of withShaderRenderer for one latlong .hdr file
 * (ShaderRendering.hs:67-91): load reflMapFn, build any missing
 * `<name>_cache_pow_<p>.hdr` (p = 1.0, 8.0, 64.0, 512.0) with the device
 * prefilter (buildPreConvolvedHDREnvMapCache, :131-149: resize to 256 texels, the missing powers concurrently, written as
 * Radiance RGBE), reload the caches, convert all five maps to cube maps on the device.  Cache files are written under a
 * private name and renamed into place (several ranks may build them at once); if the directory cannot be written the
 * file images are used from memory. */
int rmdf_load_env_hdr(rmdf_ctx *ctx, const char *latlong_hdr_path);

/* latLongHDREnvMapToCubeMap (HDREnvMap.hs:118-163) for one slot: rgb = w*h*3
 * floats, first scanline first; builds 6 faces of (w div 3)^2 RGB16F texels. */
int rmdf_set_env_latlong(rmdf_ctx *ctx, int slot, const float *rgb, int w, int h);

/* Upload ready-made faces (the texImage2D RGB16F upload of HDREnvMap.hs:160-161):
 * faces = 6*face_w*face_w*3 floats, order +X,-X,+Y,-Y,+Z,-Z, row 0 first. */
int rmdf_set_env_cube(rmdf_ctx *ctx, int slot, const float *faces_rgb, int face_w);

/* Read a slot back as the padded RGB16F array the kernels sample
 * (6*(W+2)*(W+2)*4 uint16; out may be NULL to query *face_w). */
int rmdf_get_env_cube_padded(rmdf_ctx *ctx, int slot, uint16_t *out, int *face_w);

/* resizeHDRImage (HDREnvMap.hs:169-195) on the device.  out = dstw * *dsth * 3
 * floats (may be NULL to query *dsth). */
int rmdf_resize_latlong(rmdf_ctx *ctx, const float *rgb, int w, int h, int dstw, float *out, int *dsth);

/* cosineConvolveHDREnvMap (HDREnvMap.hs:217-254) on the device: out = w*h*3 floats.  Every destination texel is summed in the
 * reference's order (source rows outer, columns inner); sin / cos come from host-built tables (the libm the reference calls);
 * cos^power for the reference's powers 1, 8, 64, 512 = 2^k is k squarings in binary64 rounded once (DESIGN.md "spec pins"),
 * other powers use the device powf.  2 <= w <= 8192, 2 <= h <= 4096.
 * rmdf_prefilter_env_powers: `npowers` powers of one map, the job of the reference's mapConcurrently (ShaderRendering.hs:142);
 * out = npowers * w*h*3 floats.  The reference's own set 1, 8, 64, 512 (or three of it) of a map up to 256 texels wide is ONE
 * kernel launch whose squaring chains share their prefix; other sets run their powers side by side on four streams.
 * rmdf_prefilter_env_device: one power, device-resident source and destination, asynchronous on `stream`.
 * The host-buffer entry points share one set of device scratch per ctx (kept between calls for maps up to 256x128, per call above
 * that): like every call on a ctx they must not run concurrently on the same ctx. */
int rmdf_prefilter_env(rmdf_ctx *ctx, const float *rgb, int w, int h, float power, float *out);
int rmdf_prefilter_env_powers(rmdf_ctx *ctx, const float *rgb, int w, int h, const float *powers, int npowers, float *out);
int rmdf_prefilter_env_device(rmdf_ctx *ctx, const void *d_rgb, int w, int h, float power, void *d_out, void *stream);

/* ---- rendering: drawShaderTile (ShaderRendering.hs:151-196) ------------------------ */

/* isTileIdxFirstTile / isTileIdxLastTile (ShaderRendering.hs:54-58) */
int rmdf_is_tile_idx_first_tile(int idx);
int rmdf_is_tile_idx_last_tile(int idx);

/* drawShaderTile sr shd tileIdx w h time, delivered through fillFrameBuffer's
 * `MVector Word32` (FrameBuffer.hs:117-158).
 *   tile_idx < 0  = `Nothing`: the whole frame.
 *   tile_idx >= 0 = `Just idx`: tile idx mod 64 of the 8x8 grid, tx = midx mod 8,
 *                   ty = midx div 8 counted from the bottom (:183-193).
 * (w, h, time, max_steps) are latched on the first tile of a frame and frozen for
 * the other 63 (:162-176).  The library keeps the accumulating frame and writes
 * ALL w*h pixels to out_rgba8 on every call (the PBO is orphaned per call,
 * FrameBuffer.hs:129,207-213).  max_steps is fragment.shd:634's MAX_STEPS (128 in
 * the reference; <= 0 selects 128). */
int rmdf_render_tile(rmdf_ctx *ctx, int scene, int tile_idx, int w, int h, double time, int max_steps,
                     uint32_t *out_rgba8);

/* Same, with the extra planes parity tests need (any pointer may be NULL; the library allocates and accumulates these
 * planes only from the first call that asks for one, and in tile mode they hold the tiles rendered through this entry):
 *   out_rgba_f32  w*h*4 floats, the shader's frag_color before RGBA8 conversion
 *   out_steps     w*h uint16: bits 0..14 ray_march loop counter at exit
 *                 (fragment.shd:659-673), bit 15 = hit
 *   out_iters     w*h uint16: Mandelbulb escape iterations the pixel spent
 *                 (march + normal + AO distance estimates; 0 for other scenes) */
int rmdf_render_tile_ex(rmdf_ctx *ctx, int scene, int tile_idx, int w, int h, double time, int max_steps,
                        uint32_t *out_rgba8, float *out_rgba_f32, uint16_t *out_steps, uint16_t *out_iters);

/* ---- device-resident forms (benchmarks, multi-GPU; no PCIe in the timed region) ---- */

/* Render the pixel rectangle [x0,x1) x [y0,y1) of a w x h frame into caller-owned
 * DEVICE buffers laid out as full frames (any may be NULL).  `stream` is a
 * hipStream_t (NULL = the ctx stream); the call is asynchronous on it.  Calls on different streams may overlap (frames in
 * flight): the library keeps its dispatch-order tables per stream (up to 32).  For more than about four concurrent streams
 * set GPU_MAX_HW_QUEUES (e.g. 16) in the environment before the HIP runtime initialises, or streams share hardware queues
 * and serialise. */
int rmdf_render_rect_device(rmdf_ctx *ctx, int scene, int w, int h, double time, int max_steps,
                            int x0, int y0, int x1, int y1,
                            void *d_rgba8, void *d_rgba_f32, void *d_steps, void *d_iters, void *stream);

/* Multi-GPU sharding of the reference's 64 tiles (ShaderRendering.hs:49-52,183-193 renders them one per frame;
 * here they are the units dealt to the GPUs).  rmdf_shard_tiles: the tile indices rank `rank` of `nranks`
 * renders, in slot order; returns their number (<= ceil(64/nranks)) or a negative error code.  Host-only
 * arithmetic, no device needed.  The deal is balanced for scenes centred in the frame: tiles sorted by distance from
 * the frame centre and dealt to the ranks boustrophedon, so every rank gets near and far tiles.
 * rmdf_render_shard_device renders those tiles packed back to back in slot order into d_packed_rgba8
 * (tile = (w/8)*(h/8) uint32, rows bottom-up).  Requires w mod 8 == 0 and h mod 8 == 0. */
int rmdf_shard_tiles(int rank, int nranks, int tiles[64]);
/* Cost-aware deal.  rmdf_probe_tile_costs renders the view at 256 x ~144 and returns, per tile, the work its rays
 * took (escape iterations + march steps + 1 per ray): the kernels are bit-reproducible, so every rank that probes
 * the same (scene, w, h, time, max_steps) gets the same 64 numbers without any exchange.  rmdf_set_shard_costs
 * (NULL = back to the static deal) makes this ctx deal the tiles longest-processing-time-first on those costs:
 * tiles in descending cost order, each to the least loaded rank that still has a free slot.  It applies to
 * rmdf_render_shard_device and rmdf_assemble_shards_device of this ctx; ALL ranks of a job must set the same costs.
 * rmdf_get_shard_tiles: the deal in effect on this ctx (same contract as rmdf_shard_tiles). */
int rmdf_probe_tile_costs(rmdf_ctx *ctx, int scene, int w, int h, double time, int max_steps, float cost[64]);
int rmdf_set_shard_costs(rmdf_ctx *ctx, const float cost[64]);
int rmdf_get_shard_tiles(rmdf_ctx *ctx, int rank, int nranks, int tiles[64]);
/* Rank 0 also receives the other ranks' shards and assembles the frame.  With a handicap the cost-aware deal starts rank 0 at
 * fraction x (total cost / nranks) instead of 0, so it is dealt correspondingly less to render (0 = off, the default; only
 * effective together with rmdf_set_shard_costs; ALL ranks must set the same value). */
int rmdf_set_shard_root_handicap(rmdf_ctx *ctx, float fraction);
int rmdf_render_shard_device(rmdf_ctx *ctx, int scene, int w, int h, double time, int max_steps,
                             int rank, int nranks, void *d_packed_rgba8, void *stream);
/* Rank 0 after the gather: d_gathered holds the nranks shards back to back (rank
 * order, each ceil(64/nranks) tile slots); scatter them to frame positions. */
int rmdf_assemble_shards_device(rmdf_ctx *ctx, int w, int h, int nranks, const void *d_gathered,
                                void *d_frame_rgba8, void *stream);

/* Super-sampling (the viewer's frame-buffer scale, App.hs:105-106,124-133, shown through the mip chain that
 * glGenerateMipmap builds, FrameBuffer.hs:153-154,187-195).  rmdf_resolve_box2_device: one mip level of an RGBA8
 * image on the device: 2x2 box per channel, (a+b+c+d+2)>>2; sw, sh even; dst = (sw/2)*(sh/2) uint32.  It also
 * works on a packed tile shard (tiles stacked: width w/8, height slots*h/8).
 * rmdf_render_supersampled: render scene at (w<<levels) x (h<<levels), resolve `levels` times, copy the w x h
 * result to the host buffer. */
int rmdf_resolve_box2_device(rmdf_ctx *ctx, const void *d_src_rgba8, int sw, int sh, void *d_dst_rgba8, void *stream);
int rmdf_render_supersampled(rmdf_ctx *ctx, int scene, int w, int h, int levels, double time, int max_steps,
                             uint32_t *out_rgba8);

/* ---- multi-GPU exchange: one process per GPU, RCCL over xGMI ------------------------------------------------
 * The reference renders its 64 tiles one per frame (ShaderRendering.hs:49-52,183-193); here they are dealt to the GPUs of a
 * node and ONE gather per frame brings the packed shards to rank 0.  librccl.so.1 is dlopen()ed by the first rmdf_comm_* call
 * (a single-GPU host does not need it).
 *   rank 0:      rmdf_comm_get_unique_id(id); ship the RMDF_COMM_ID_BYTES bytes to the other ranks by any channel
 *   every rank:  rmdf_comm_init(ctx, id, rank, nranks)            -- collective (ncclCommInitRank on the ctx's device)
 *   per frame:   rmdf_render_frame_sharded_device(...)            -- = rmdf_render_shard_device + rmdf_gather_shards_device
 *                                                                    + (rank 0) rmdf_assemble_shards_device, all on `stream`
 * rmdf_gather_shards_device: every rank's packed shard (ceil(64/nranks) tile slots of (w/8)*(h/8) uint32) lands in
 * d_gathered[rank] on rank 0 (grouped ncclRecv fan-in; the other ranks ncclSend).  ALL ranks must have set the same costs /
 * handicap, or none (rmdf_set_shard_costs).  Every rank always sends its WHOLE fixed-size region: the size on the wire depends on
 * (w, h, nranks) alone, so that a rank whose costs differ mis-assembles a frame (which rmdf_comm_verify_deal detects) instead of
 * hanging the job on mismatched sizes; for the rank counts that divide 64 that is exactly the tiles a rank owns.  Rank 0 may pass
 * d_shard == d_gathered (it rendered straight into its own slot).  d_gathered is ignored on the other ranks.
 * rmdf_comm_verify_deal: COLLECTIVE over the ctx's communicator (every rank calls it, after its last rmdf_set_shard_costs /
 * rmdf_set_shard_root_handicap): the peers send a fingerprint of the deal they hold to rank 0 (8 bytes), rank 0 compares and
 * answers; RMDF_OK on every rank iff all deals are equal, RMDF_E_COMM on every rank otherwise.  Blocks until `stream` (NULL = ctx
 * stream) has drained.
 * rmdf_comm_selftest_loopback: the exchange's own calls against this rank itself -- a grouped ncclRecv from self + ncclSend to
 * self of `bytes` bytes on `stream` (NULL = ctx stream), compared word for word; on the ctx's communicator, or on a private
 * one-rank communicator when the ctx has none (a single-GPU box can run it).  *mismatches (may be NULL) = differing words. */
#define RMDF_COMM_ID_BYTES 128
int rmdf_comm_get_unique_id(void *id);
int rmdf_comm_init(rmdf_ctx *ctx, const void *id, int rank, int nranks);
int rmdf_comm_destroy(rmdf_ctx *ctx);
/* *nranks = 0 when the ctx has no communicator */
int rmdf_comm_info(rmdf_ctx *ctx, int *rank, int *nranks);
int rmdf_gather_shards_device(rmdf_ctx *ctx, int w, int h, const void *d_shard, void *d_gathered, void *stream);
int rmdf_comm_verify_deal(rmdf_ctx *ctx, void *stream);
int rmdf_comm_selftest_loopback(rmdf_ctx *ctx, size_t bytes, void *stream, uint64_t *mismatches);
int rmdf_render_frame_sharded_device(rmdf_ctx *ctx, int scene, int w, int h, double time, int max_steps,
                                     void *d_shard, void *d_gathered, void *d_frame_rgba8, void *stream);

/* The constant tables the kernels are built from, for checking against the reference (host-only: no ctx, no device).
 * rmdf_get_cornell_vertices: the 96 triangle vertices mkCornellBoxVerticesTex uploads (CornellBox.hs:21-46,48-129), 96*3 floats.
 * rmdf_get_shader_constants: the named constants of fragment.shd the kernels use (bailout, MIN_DIST, the distance-AO taps, the
 * shading weights, ...: csrc/rmdf_device.hpp RMDF_SHADER_CONSTANTS); fills up to `cap` entries of names[] (static strings) and
 * values[] (either may be NULL) and returns how many there are.  tests/test_reference_pins.py compares both with values a script
 * extracted from the reference (tests/golden/reference_pins.json). */
int rmdf_get_cornell_vertices(float out[96 * 3]);
int rmdf_get_shader_constants(const char **names, float *values, int cap);

/* Self-test of the kernels' short correctly-rounded sequences (sqrt, reciprocal, 1/sqrt, and the known-range
 * division inside log) against the compiler's IEEE expansions for ALL 2^32 float inputs on the device.
 * mismatches[0..3] = sqrt, reciprocal, log, 1/sqrt; mismatches[4] = the table-driven division of the Cornell
 * distance estimator against the compiler's for every numerator and each of its 96 divisors; mismatches[5..7] = the
 * Mandelbulb loop's forms: the bailout test taken on the squared radius together with the estimate's final division
 * (Markstein on the reciprocal of dr, 2^33 operand pairs), the in-loop root of the radius and the in-loop
 * 1/sqrt(k3^7) (one transcendental each, one shared guard) against the written sqrt / inversesqrt; mismatches[8] = whole
 * Mandelbulb estimates with the power-of-two scalings folded into FMAs (behind their underflow guard, written passes as the
 * fall-back) against the written loop on 2^28 points, half of them with coordinates down to 2^-150.  [0..8] must be 0;
 * mismatches[9] = how many of those estimates took the fall-back (informational: the test must reach it, so > 0). */
int rmdf_selftest_exact_math(rmdf_ctx *ctx, uint64_t mismatches[10]);
/* Self-test of the straight-line device forms of the pinned GLSL built-ins (exp, acos, atan, sin, cos: all 2^32 inputs;
 * atan(y,x) and pow(x,y): 2^32 operand pairs) against the branchy fdlibm-style forms they restate.
 * mismatches[0..6] = exp, acos, atan, sin, cos, atan2, pow.  All must be 0. */
int rmdf_selftest_pinned_math(rmdf_ctx *ctx, uint64_t mismatches[7]);
/* Self-test of the short quotients of the shading tail (generate_ray's pixel-centre divisions, the ambient-occlusion terms,
 * fresnel_conductor, the cube-map texture coordinates: Markstein quotients on a correctly rounded reciprocal instead of the
 * compiler's IEEE expansion) against the compiler's division: mismatches[0] = the quotient itself on 2^33 operand pairs inside its
 * range, [1] = clamp(1 - d / e) for every distance d (inf and NaN included) and each tap offset e, [2] = fresnel_conductor for every cosine
 * in [-2, 2], inf and NaN, [3] = whole cube-map lookups on 2^30 (direction, neighbour, neighbour) triples of the kind normalize() can
 * produce, degenerate ones included; [4] = generate_ray's own quotients EXHAUSTIVELY over the frames the library accepts (1 <= w, h <=
 * 32768): (px + 0.5) / w for every side and every pixel centre of it, and for every (w, h) the reciprocal of the aspect ratio plus
 * ndc.y * fov / aspect on three rows.  All must be 0.  A few seconds on an MI355X. */
int rmdf_selftest_shading_math(rmdf_ctx *ctx, uint64_t mismatches[5]);

/* The clock the shader engines run at right now: one wave (issuing vector instructions) stamps the shader-cycle counter against the
 * 100 MHz real-time counter over `spin_us` microseconds, on a highest-priority stream of the library's own -- launched while the
 * caller's frames run on other streams it reports the clock under that load.  MI355X, measured with this renderer: 2.4 GHz idle and
 * under sustained load, but 2.06-2.1 GHz for the first milliseconds of a load that starts from idle (it ramps back up over ~20 ms):
 * a lone frame rendered from idle runs at the lower clock.  Blocks until the probe has finished. */
int rmdf_probe_shader_clock(rmdf_ctx *ctx, double spin_us, double *mhz);

/* Optional, and since round 5 a hint only: declare a host buffer the caller reuses from frame to frame.  Rounds 2-4 page-locked the
 * range and mapped it into the GPU's address space (hipHostRegister) so that the render kernel could store a whole frame into it
 * directly.  Round 5 found every GPU memory fault of its hunt on heap ranges that such a registration had covered earlier (the
 * ROCm user-mode stack kept resolving addresses inside them to the old, shorter mapping: NOTEBOOK.md A.5), so the library no longer
 * creates GPU mappings of memory it does not own: the call records the range, rmdf_unregister_host_buffer forgets it, and a whole-
 * frame rmdf_render_tile into it takes the same path as into any other pointer (row bands through page-locked staging, rmdf_api.cpp).
 * Same results, same blocking behaviour; registering twice is harmless; unregistering an unknown pointer is RMDF_E_INVALID. */
int rmdf_register_host_buffer(rmdf_ctx *ctx, void *ptr, size_t bytes);
int rmdf_unregister_host_buffer(rmdf_ctx *ctx, void *ptr);

/* Screenshot: saveFrameBufferToPNG (FrameBuffer.hs:215-228) for a frame buffer in the boundary's layout (w*h
 * little-endian Word32 = R,G,B,A bytes, row 0 = bottom): rows flipped to top-down, alpha forced to 0xFF, written
 * as an 8-bit RGBA PNG.  Host-only (no ctx, no device); errors are reported through rmdf_last_error(NULL). */
int rmdf_save_png(const char *path, const uint32_t *fb_rgba8, int w, int h);

/* Device memory for hosts without a HIP binding of their own (the Haskell viewer, a plain-C host): buffers for the
 * device-resident entry points above.  rmdf_device_malloc / rmdf_device_free: hipMalloc / hipFree on the ctx's device.
 * rmdf_copy_to_host: asynchronous device-to-host copy on `stream` (NULL = ctx stream) followed by a wait for it. */
int rmdf_device_malloc(rmdf_ctx *ctx, size_t bytes, void **d_ptr);
int rmdf_device_free(rmdf_ctx *ctx, void *d_ptr);
int rmdf_copy_to_host(rmdf_ctx *ctx, void *host_dst, const void *d_src, size_t bytes, void *stream);

/* Block until everything queued on `stream` (NULL = ctx stream) has finished. */
int rmdf_synchronize(rmdf_ctx *ctx, void *stream);

/* Name of the HIP device the ctx runs on and its compute-unit count. */
int rmdf_device_info(rmdf_ctx *ctx, char *name, int name_len, int *compute_units);

#ifdef __cplusplus
}
#endif
#endif /* RMDF_H */
