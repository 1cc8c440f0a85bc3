/*
 * rmdf_oracle.h -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE)
 *
 * A plain-C restatement of the per-pixel sphere-tracing hot path of
 * blitzcode/ray-marching-distance-fields (fragment.shd) and of the CPU paths
 * around it (HDREnvMap.hs, CoordTransf.hs, CornellBox.hs, Fractal2D.hs,
 * ConcurrentSegments.hs).  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (librmdf.so) never does.
 *
 * PARITY STATUS.  The reference ships no tests, golden vectors or fixtures
 *  and its host code is Haskell (no GHC here), so nothing of
 * the reference's own pins this file: the CPU paths restated here (HDREnvMap,
 * CoordTransf, Fractal2D, the JuicyPixels / linear arithmetic behind them) are
 * **parity unpinned**.  The shader path IS pinned against outputs of the
 * reference itself run in the build container: fragment.shd, mechanically patched
 * for GLSL ES, executed on the SwiftShader GLES3 software rasteriser
 * (tests/golden/make_swiftshader_vectors.py wrote tests/golden/swiftshader_*.npz;
 * tests/test_oracle_vs_glsl.py): hit masks identical, march step counts identical
 * on every pixel of 12 frames, the escape-iteration counts of the march (read from
 * the shader's own de_mandelbulb loop) identical per pixel, background colour equal
 * to ~1e-6, the shading -- given the normals / AO the shader itself computed --
 * equal to its colour to ~1e-7 (within 1e-4 on >= 99.9 % of hit pixels); the
 * surface colour of the whole pipeline only statistically (the shader
 * differentiates a fractal with eps = 1e-5 in float32, so two correct
 * implementations agree only in distribution).  A patched shader on a stand-in GL
 * is not the unmodified reference: by the build's rules the float colour stays
 * "parity unpinned"; the integer planes are pinned by the above.  Further
 * pins: analytic known-answer tests (tests/test_oracle_kat.py) and committed
 * golden frames of this oracle (tests/golden/make_fixtures.py).  GLSL leaves
 * inversesqrt/pow/log precision, FMA contraction, f16 texel rounding and LOD
 * selection implementation-defined; every such choice is pinned below and listed
 * in DESIGN.md ("spec pins").
 *
 * All arithmetic is IEEE-754 binary32, round-to-nearest-even, one rounding per
 * written operation (compile with -ffp-contract=off), left-to-right
 * evaluation exactly as the GLSL text parses.
 */
#ifndef RMDF_ORACLE_H
#define RMDF_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* FragmentShader enum order, ShaderRendering.hs:46-47 */
enum { ORC_SCENE_CORNELL = 0, ORC_SCENE_DETEST = 1, ORC_SCENE_MB_POWER8 = 2, ORC_SCENE_MB_GENERAL = 3 };

/* A cube map as the shader sees it: 6 faces of W x W RGB16F texels, stored
 * PADDED to (W+2) x (W+2) with a one-texel seamless border, 4 halfs per texel
 * (r,g,b,0).  Face order +X,-X,+Y,-Y,+Z,-Z (HDREnvMap.hs:131-136). */
typedef struct {
    int             W;
    const uint16_t *padded;   /* 6 * (W+2)*(W+2) * 4 halfs */
} orc_cube;

typedef struct {
    int      scene;          /* ORC_SCENE_* */
    int      w, h;           /* in_screen_wdh / in_screen_hgt (ShaderRendering.hs:169-170) */
    float    time;           /* in_time (ShaderRendering.hs:171) */
    int      max_steps;      /* fragment.shd:634 (128 in the reference) */
    orc_cube env_reflection; /* fragment.shd:10 */
    orc_cube env_cos_1;      /* fragment.shd:11 */
    orc_cube env_cos_8;      /* fragment.shd:12 */
} orc_frame;

typedef struct {
    uint64_t de_evals;       /* distance_estimator() calls                       */
    uint64_t triplex_iters;  /* Mandelbulb iterations that ran triplex_pow       */
    uint64_t march_steps;    /* DE calls made from ray_march                      */
    uint64_t hit_pixels;
    uint64_t sphere_pixels;  /* rays that entered the bounding sphere             */
    uint64_t pixels;
    uint64_t tri_inside;     /* Cornell box: (estimate, triangle) pairs that took de_triangle's prism branch (the other 32 * de_evals - this took the edge branch) */
} orc_counters;

/* ---- hot path (fragment.shd) ------------------------------------------------ */

/* Camera block of main() (fragment.shd:883-902) + lookat (829-838).
 * out12 = xaxis, yaxis, zaxis, eye (column-major mat4 minus the last row). */
void orc_camera(int scene, float time, float out12[12]);

/* tan(radians(45*1.5)/2), fragment.shd:866-867,910 */
float orc_fov_xs(void);

/* Render the pixel rectangle [x0,x1) x [y0,y1) of a w x h frame.  Output arrays
 * are full-frame, index px + py*w, row 0 = bottom (gl_FragCoord origin).  Any
 * output pointer may be NULL.  steps: bits 0..14 = loop counter at exit
 * (fragment.shd:659-673), bit 15 = hit.  iters: total Mandelbulb iterations the
 * pixel spent (march + normal + AO).  nthreads <= 0 -> all cores; rows are split
 * like ConcurrentSegments.makeNSegments. */
int orc_render(const orc_frame *f, int x0, int y0, int x1, int y1,
               float *rgba_f32, uint32_t *rgba8, uint16_t *steps, uint16_t *iters,
               orc_counters *ctr, int nthreads);

/* Same, plus iters_march (may be NULL): the Mandelbulb iterations the pixel spent inside ray_march alone (the `iters`
 * plane adds the four normal taps and the two AO taps of hit pixels).  Used to pin the escape-iteration counts against the
 * reference shader run on SwiftShader, whose near-surface normal / AO taps are chaotic (tests/test_oracle_vs_glsl.py). */
int orc_render_ex(const orc_frame *f, int x0, int y0, int x1, int y1,
                  float *rgba_f32, uint32_t *rgba8, uint16_t *steps, uint16_t *iters, uint16_t *iters_march,
                  orc_counters *ctr, int nthreads);

/* Shading alone: colour of every pixel given its hit flag and the (normal, ao) the hit branch produced (w*h*4 floats);
 * w, h even.  Used to compare the shading with the reference shader's, its own normals fed back (tests/test_oracle_vs_glsl.py). */
int orc_shade_gbuffer(const orc_frame *f, const float *nao, const uint8_t *hit, float *rgba_f32);

/* Single-point probes for known-answer tests */
float orc_de(int scene, float time, const float pos[3]);
void  orc_triplex_pow8(const float w[3], float out[3]);
void  orc_triplex_pow(const float w[3], float power, float out[3]);
float orc_logf(float x);
float orc_expf(float x);
float orc_powf(float x, float y);
float orc_sinf(float x);
float orc_cosf(float x);
float orc_acosf(float x);
float orc_atan2f(float y, float x);
float orc_general_power(float time);   /* fragment.shd:116-119 */
float orc_fresnel_conductor(float cosi, float eta, float k);
int   orc_ray_sphere(const float o[3], const float d[3], float R, float *tmin, float *tmax);
/* texture(samplerCube, dir) with an explicit filter choice (0 NEAREST, 1 LINEAR) */
void  orc_cube_sample(const orc_cube *c, const float dir[3], int linear, float rgb[3]);

/* ---- env-map data prep (HDREnvMap.hs / CoordTransf.hs) ---------------------- */

/* Radiance .hdr reader (flat or new-RLE), JuicyPixels-style RGBE -> float.
 * Returns 0 on success; *w,*h receive the size; out (may be NULL to query the
 * size) receives w*h*3 floats, first scanline first. */
int  orc_hdr_decode(const uint8_t *file, long len, int *w, int *h, float *out);
void orc_rgbe_encode(const float *rgb, long npix, uint8_t *rgbe);   /* toRGBE  */
void orc_rgbe_decode(const uint8_t *rgbe, long npix, float *rgb);   /* toFloat */
/* Flat (un-RLE'd) Radiance file writer; returns bytes written (buffer >= 64 + 4*w*h). */
long orc_hdr_encode(const float *rgb, int w, int h, uint8_t *file);

void orc_pixel_at_bilinear(const float *img, int w, int h, float u, float v, float rgb[3]);
void orc_cube_pixel_to_dir(int face, int w, int x, int y, float dir[3]);
/* latLongHDREnvMapToCubeMap: faces_f32 = 6*cw*cw*3 floats, cw = w/3 */
void orc_latlong_to_cube(const float *latlong, int w, int h, float *faces_f32, int nthreads);
void orc_build_test_latlong(float *rgb /* 512*256*3 */);
/* RGB16F upload (round-to-nearest-even) + seamless padding */
uint16_t orc_f32_to_f16(float f);
float    orc_f16_to_f32(uint16_t hbits);
void orc_cube_pad_f16(const float *faces_f32, int cw, uint16_t *padded);
/* resizeHDRImage: returns dst height; out may be NULL to query */
int  orc_resize_hdr(const float *src, int sw, int sh, int dstw, float *out);
/* pow_mode: 0 = cos^p through libm powf (the reference's literal `**`), 1 = the spec pin for powers 1/8/64/512
 * (binary64 squaring chain rounded once); see conv_pow in rmdf_oracle.c */
void orc_cosine_convolve(const float *src, int w, int h, float power, float *out, int nthreads, int pow_mode);

/* ---- geometry (CornellBox.hs) ------------------------------------------------ */
void orc_cornell_vertices(float out[96 * 3]);
/* the named constants of fragment.shd this restatement uses (ORC_SHADER_CONSTANTS): fills up to cap entries, returns the count */
int  orc_shader_constants(const char **names, float *values, int cap);

/* ---- 2-D fractals (Fractal2D.hs) and segments (ConcurrentSegments.hs) -------- */
void orc_julia_animated(int w, int h, uint32_t *fb, int smooth, double tick, int nthreads);
void orc_mandelbrot(int w, int h, uint32_t *fb, int smooth);
int  orc_make_n_segments(int nseg, int low, int high, int *out_pairs /* 2*nseg ints */);

/* one glGenerateMipmap level of an RGBA8 frame (FrameBuffer.hs:153-154): 2x2 box, round-half-up */
int  orc_resolve_box2(const uint32_t *src, int sw, int sh, uint32_t *dst);

int  orc_num_processors(void);

#ifdef __cplusplus
}
#endif
#endif
