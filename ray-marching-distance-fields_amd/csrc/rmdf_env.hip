// rmdf_env.hip -- gfx950 kernels of the environment-map preparation that the reference runs on the CPU
// (HDREnvMap.hs): RGB16F cube upload with seamless border, lat/long -> cube faces, resizeHDRImage and the
// cosine-lobe prefilter.  Every kernel here is BIT-EXACT against the oracle: the only transcendental values the
// reference's formulas need (acos / atan of a face texel's direction, sin / cos of the row and column angles)
// depend on image SIZES only, so the host computes them once with the libm the reference itself calls through GHC
// (glibc) and the kernels do IEEE float arithmetic in the reference's operation order (-ffp-contract=off).
#include "rmdf_internal.hpp"

namespace rmdf {

// ------------------------------------------------------------------------------------
// RGB16F upload with seamless border (the texImage2D RGB16F of HDREnvMap.hs:160-161 +
// GL_TEXTURE_CUBE_MAP_SEAMLESS, :126).  Border rule: DESIGN.md "spec pins".
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void cube_int_dir(int face, int W, int cw, int ch, int d[3])
{
    switch (face) {
    case 0:  d[0] =  W;  d[1] = -ch; d[2] = -cw; break;
    case 1:  d[0] = -W;  d[1] = -ch; d[2] =  cw; break;
    case 2:  d[0] =  cw; d[1] =  W;  d[2] =  ch; break;
    case 3:  d[0] =  cw; d[1] = -W;  d[2] = -ch; break;
    case 4:  d[0] =  cw; d[1] = -ch; d[2] =  W;  break;
    default: d[0] = -cw; d[1] = -ch; d[2] = -W;  break;
    }
}

// texel (x,y) of `face` with exactly one coordinate out of range by one -> the texel
// adjacent across the cube edge
__device__ __forceinline__ void cube_fold(int face, int W, int x, int y, int &nf, int &nx, int &ny)
{
    int d[3];
    cube_int_dir(face, W, 2 * x + 1 - W, 2 * y + 1 - W, d);
    int major = face >> 1;
    int over = -1;
#pragma unroll
    for (int a = 0; a < 3; a++)
        if (a != (face >> 1) && over < 0 && (d[a] > W || d[a] < -W)) over = a;
    if (over >= 0) {
        int keep = (d[major] > 0) ? W - 1 : -(W - 1);
        int nd[3] = { d[0], d[1], d[2] };
        nd[over] = (d[over] > 0) ? W : -W;
        nd[major] = keep;
        d[0] = nd[0]; d[1] = nd[1]; d[2] = nd[2];
        major = over;
    }
    int f = major * 2 + (d[major] > 0 ? 0 : 1);
    int cw, ch;
    switch (f) {
    case 0:  ch = -d[1]; cw = -d[2]; break;
    case 1:  ch = -d[1]; cw =  d[2]; break;
    case 2:  cw =  d[0]; ch =  d[2]; break;
    case 3:  cw =  d[0]; ch = -d[2]; break;
    case 4:  cw =  d[0]; ch = -d[1]; break;
    default: cw = -d[0]; ch = -d[1]; break;
    }
    nf = f; nx = (cw + W - 1) / 2; ny = (ch + W - 1) / 2;
}

__device__ __forceinline__ __half src_half(const float *faces, int W, int f, int x, int y, int k)
{
    return __float2half_rn(faces[(((size_t)f * W + y) * W + x) * 3 + k]);
}

__global__ void k_cube_upload(const float *__restrict__ faces, int W, uint2 *__restrict__ padded)
{
    const int P = W + 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 6 * P * P) return;
    const int X = i % P, Y = (i / P) % P, f = i / (P * P);
    const int x = X - 1, y = Y - 1;
    const bool ox = (x < 0 || x >= W), oy = (y < 0 || y >= W);
    __half c[3];
    if (!ox && !oy) {
        for (int k = 0; k < 3; k++) c[k] = src_half(faces, W, f, x, y, k);
    } else if (ox != oy) {
        int nf, nx, ny;
        cube_fold(f, W, x, y, nf, nx, ny);
        for (int k = 0; k < 3; k++) c[k] = src_half(faces, W, nf, nx, ny, k);
    } else {
        const int cx = x < 0 ? 0 : W - 1, cy = y < 0 ? 0 : W - 1;
        int f1, x1, y1, f2, x2, y2;
        cube_fold(f, W, x, cy, f1, x1, y1);
        cube_fold(f, W, cx, y, f2, x2, y2);
        for (int k = 0; k < 3; k++) {
            float a = __half2float(src_half(faces, W, f, cx, cy, k));
            float b = __half2float(src_half(faces, W, f1, x1, y1, k));
            float cc = __half2float(src_half(faces, W, f2, x2, y2, k));
            c[k] = __float2half_rn(((a + b) + cc) / 3.0f);
        }
    }
    uint2 t;
    t.x = (uint32_t)__half_as_ushort(c[0]) | ((uint32_t)__half_as_ushort(c[1]) << 16);
    t.y = (uint32_t)__half_as_ushort(c[2]);
    padded[i] = t;
}

hipError_t launch_cube_upload(const float *d_faces_f32, int W, uint2 *d_padded, hipStream_t stream)
{
    const int n = 6 * (W + 2) * (W + 2);
    hipLaunchKernelGGL(k_cube_upload, dim3((n + 255) / 256), dim3(256), 0, stream, d_faces_f32, W, d_padded);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// pixelAtBilinear, HDREnvMap.hs:91-113 (keeps the `mod (w-1)` / `min (h-1)` quirks)
// The reference fetches with unsafePixelAt and steps past the image when (u, v) leaves [0, 1]: resizeHDRImage does that for maps
// that are not 2:1 (1024x510 -> 256: dsth rounds up to 128 and the last tap row asks for source row 511), where the reference
// reads whatever follows the pixel vector.  Pin (oracle and device alike): the integer texel (x, y) is clamped into the image
// before the fetch -- the weights keep their values; inside the image nothing changes.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ v3 pixel_at_bilinear(const float *__restrict__ img, int w, int h, float u, float v)
{
    const float upx = u * ((float)w - 1.0f), upy = v * ((float)h - 1.0f);
    const int xf = (int)floorf(upx), yf = (int)floorf(upy);
    const int x = xf < 0 ? 0 : (xf > w - 1 ? w - 1 : xf), y = yf < 0 ? 0 : (yf > h - 1 ? h - 1 : yf);
    const int m = w - 1;
    int xp1 = (x + 1) % m;
    if (xp1 < 0) xp1 += m;
    const int yp1 = (y + 1 < h - 1) ? y + 1 : h - 1;
    const float ur = upx - (float)xf, vr = upy - (float)yf;
    const float uo = 1.0f - ur, vo = 1.0f - vr;
    const float *a = img + ((size_t)x + (size_t)y * w) * 3, *b = img + ((size_t)xp1 + (size_t)y * w) * 3;
    const float *c = img + ((size_t)x + (size_t)yp1 * w) * 3, *d = img + ((size_t)xp1 + (size_t)yp1 * w) * 3;
    return mk3((a[0] * uo + b[0] * ur) * vo + (c[0] * uo + d[0] * ur) * vr,
               (a[1] * uo + b[1] * ur) * vo + (c[1] * uo + d[1] * ur) * vr,
               (a[2] * uo + b[2] * ur) * vo + (c[2] * uo + d[2] * ur) * vr);
}

// ------------------------------------------------------------------------------------
// latLongHDREnvMapToCubeMap (HDREnvMap.hs:118-163): one thread per face texel.  The environment (u, v) of a face
// texel -- cubeMapPixelToDir, worldToLocal, cartesianToSpherical (acos, atan2), sphericalToEnvironmentUV -- is a
// function of (face, x, y, face size) alone: the host evaluates it once per face size with glibc's acosf / atanf
// (cube_uv_table_host in rmdf_api.cpp) and the kernel only does the bilinear gather, so the faces carry the same
// bits as the reference's CPU loop on this machine.
// ------------------------------------------------------------------------------------
__global__ void k_latlong_to_cube(const float *__restrict__ latlong, int w, int h, int cw,
                                  const float2 *__restrict__ uv, float *__restrict__ faces)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 6 * cw * cw) return;
    const float2 t = uv[i];
    const v3 c = pixel_at_bilinear(latlong, w, h, t.x, t.y);
    faces[(size_t)i * 3 + 0] = c.x; faces[(size_t)i * 3 + 1] = c.y; faces[(size_t)i * 3 + 2] = c.z;
}

hipError_t launch_latlong_to_cube(const float *d_latlong, int w, int h, const float2 *d_uv, float *d_faces_f32, hipStream_t stream)
{
    const int cw = w / 3, n = 6 * cw * cw;
    if (n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_latlong_to_cube, dim3((n + 255) / 256), dim3(256), 0, stream, d_latlong, w, h, cw, d_uv, d_faces_f32);
    return hipGetLastError();
}

// resizeHDRImage (HDREnvMap.hs:169-195): one thread per destination pixel (no libm in it: bit-exact as written)
__global__ void k_resize_latlong(const float *__restrict__ src, int sw, int sh, int dstw, int dsth, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= dstw * dsth) return;
    const int dx = i % dstw, dy = i / dstw;
    const float scale = (float)sw / (float)dstw;
    const int taps = (int)ceilf(scale);
    const float ntaps = (float)(taps * taps);
    const float step = scale / (float)taps;
    const float srcx1 = (float)dx * scale, srcy1 = (float)dy * scale;
    float ar = 0.0f, ag = 0.0f, ab = 0.0f;
    for (int y = 0; y < taps; y++)
        for (int x = 0; x < taps; x++) {
            const float sx = srcx1 + (float)x * step, sy = srcy1 + (float)y * step;
            const v3 c = pixel_at_bilinear(src, sw, sh, sx / ((float)sw - 1.0f), sy / ((float)sh - 1.0f));
            ar = ar + c.x; ag = ag + c.y; ab = ab + c.z;
        }
    out[(size_t)i * 3 + 0] = ar / ntaps; out[(size_t)i * 3 + 1] = ag / ntaps; out[(size_t)i * 3 + 2] = ab / ntaps;
}

hipError_t launch_resize_latlong(const float *d_src, int sw, int sh, int dstw, int dsth, float *d_out, hipStream_t stream)
{
    const int n = dstw * dsth;
    if (n <= 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_resize_latlong, dim3((n + 255) / 256), dim3(256), 0, stream, d_src, sw, sh, dstw, dsth, d_out);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------
// cosineConvolveHDREnvMap (HDREnvMap.hs:217-254): O(n^4) -- every destination texel sums sin(theta) * cos^p over all
// source texels with a positive cosine.  Float addition is not associative, so the only parallelism that keeps the
// reference's bits is across DESTINATION texels: one lane per destination texel walks all source texels in the
// reference's order (y outer, x inner) with its four accumulators (r, g, b, n).  One wavefront = 64 neighbouring
// destination columns of one row; a workgroup = four wavefronts = four rows of the same columns, which share the staged
// cosine table and land on the four SIMDs of a CU (single-wave workgroups were all placed on the same SIMD of their CU and
// shared its issue slots: 1.16 ms per power at 256x128 instead of 0.6).  At 256x128 one power is 512 waves, half the
// machine's SIMDs; rmdf_prefilter_env_powers runs the four powers concurrently (the reference's mapConcurrently), which
// fills it.
//
//   lutT[(blk * w + x) * 64 + lane] = cos |phi_L(blk*64+lane) - phi(x)|      (absPhiDiffCosLookup, host glibc cosf)
//   tcs[y] = (cos theta_y, sin theta_y)                                        (host glibc cosf / sinf)
// The wave's slice of lutT is staged in LDS when two workgroups per CU fit (w <= 256: 64 KiB each) and read from global
// memory (L2-resident) otherwise; the source row, tcs[y] and the loop bounds are wave-uniform and arrive by scalar loads.
//
// cos^p: LOG2P >= 0 -> p = 2^LOG2P by LOG2P squarings in binary64 rounded once to binary32 -- the spec pin for the
// reference's powers 1, 8, 64, 512 (FP64 vector multiplies issue at the FP32 rate on gfx950);
// LOG2P = -1 -> device powf (any other power: tolerance parity only).
// The inner loop is branch-free (see the comment in the kernel), so eight iterations' loads and multiplies are in flight at once.
// ------------------------------------------------------------------------------------
#define PREFILTER_WAVES 4            // destination rows (waves) per workgroup; they share the staged cosine table

// Round 3.  What bounds this kernel is not arithmetic: at the reference's 256x128 one power is 512 waves on 1024 SIMDs, a wave issues
// one vector instruction per ~6 cycles, and four powers side by side (2 waves per SIMD) took TWICE as long each as alone -- the
// waves were waiting for the source row, which every wave of the machine read for itself, texel by texel: with scalar loads
// (round 2; they return out of order, so every use waits for all that are in flight, the prefetch included) or with vector loads
// of a wave-uniform address (16 cycles of the CU's address unit per instruction, for 8 waves).  Now the workgroup stages the
// source row in LDS once, cooperatively (row y + 1 is loaded while row y is summed: two buffers, one barrier per row), and the lanes
// read it by broadcast ds_read_b128; with the cosine table that makes the inner loop LDS-only, LDS returns in order, and the
// loads of the next sixteen texels are in flight while sixteen compute.  Same operations in the same order per destination
// texel: bit-equal (tests/test_gpu_env.py).
// Measured and dropped: one lane per (destination texel, channel) -- r, g, b are independent sums, so the split is exact and gives
// three times the waves -- 0.87 ms for p = 1 either way and slower for the other powers (every lane repeats the binary64 chain).
template <int LOG2P, bool LUT_IN_LDS>
__global__ __launch_bounds__(64 * PREFILTER_WAVES) void k_prefilter(const float *__restrict__ src, int w, int h, float power,
                                                  const float *__restrict__ lutT, const float2 *__restrict__ tcs,
                                                  float *__restrict__ out, int nbuf)
{
#ifdef RMDF_HOST_EMULATION
    float *lds_dyn = (float *)koh::dyn_lds();
#else
    extern __shared__ float lds_dyn[];
#endif                  // [nbuf][row_stride] source rows, then [w][64] cosine table when LUT_IN_LDS
    // nbuf = 2: the next row is written while this one is still being read by slower waves; nbuf = 1: one more barrier per row
    // separates the two.  The launcher takes one buffer as soon as two would exceed 80 KB (w > 3413), so that two workgroups
    // still fit a CU; tests/test_gpu_env.py runs a width on each side of that boundary
    const int row_stride = (w * 3 + 3) & ~3;            // floats per staged row, a multiple of 4 (ds_read_b128)
    float *lds_row = lds_dyn, *lds_lut = lds_dyn + nbuf * row_stride;
    // the wave index is wave-uniform, but the compiler cannot know that of threadIdx.x >> 6: say so, or every address
    // derived from it is treated as divergent
    const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int blk = blockIdx.x, dy_raw = blockIdx.y * PREFILTER_WAVES + g;
    const int dy = dy_raw < h ? dy_raw : h - 1;         // waves past the last row keep the barriers company and store nothing
    const int dx = blk * 64 + lane;
    const float *glut = lutT + (size_t)blk * w * 64;
    if (LUT_IN_LDS)
        for (int i = threadIdx.x; i < w * 64; i += 64 * PREFILTER_WAVES) lds_lut[i] = glut[i];
    // stage source row 0
    const int nrow = w * 3;
    for (int i = threadIdx.x; i < nrow; i += 64 * PREFILTER_WAVES) lds_row[i] = src[i];
    __syncthreads();
    typedef const float __attribute__((address_space(4))) cfloat;      // wave-uniform, read-only: scalar loads
    const float lc = ((cfloat *)tcs)[2 * dy], ls = ((cfloat *)tcs)[2 * dy + 1];
    // The inner loop is written for few instructions:
    //   * the branch `if cosAngle > 0` becomes a clamp: c0 = max(cosAngle, 0) makes cos^p, the factor and the three products
    //     +-0, and x + (+-0) == x -- the same bits as skipping the texel (finite texels: Radiance RGBE cannot encode others);
    //   * (r, g) accumulate as one packed pair (v_pk_mul_f32 / v_pk_add_f32: two IEEE operations per instruction);
    //   * the sample count n is an integer (the reference counts in Float: exact below 2^24, where a Float counter stops growing
    //     -- restored at the end): cos_angle > 0 <=> clamp(bits as int32, 0, 1) == 1, one v_med3_i32, and two of them go into the
    //     count with one v_add3_u32 (inline asm: written in C the compiler turns it back into compare + select + add-with-carry).
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 arg = { 0.0f, 0.0f };
    float ab = 0.0f;
    unsigned ni = 0u;
    constexpr int NPF = 3;                               // floats of the next row a thread carries: 64 * 4 * 3 >= 768 = 256 texels; wider rows loop
    for (int y = 0; y < h; y++) {
        const float pc = ((cfloat *)tcs)[2 * y], ps = ((cfloat *)tcs)[2 * y + 1];
        const float lcpc = lc * pc, lsps = ls * ps;
        const float *row = lds_row + (nbuf == 2 ? (y & 1) : 0) * row_stride;
        // the next source row: loaded into registers now, written to the other buffer after this row's sums
        const bool more = y + 1 < h;
        const float *nsrc = src + (size_t)(more ? y + 1 : y) * nrow;
        float pf[NPF];
#pragma unroll
        for (int k = 0; k < NPF; k++) { const int i = threadIdx.x + k * 64 * PREFILTER_WAVES; pf[k] = i < nrow ? nsrc[i] : 0.0f; }
        if (LOG2P >= 0) {
            // one texel: the reference's arithmetic, operation for operation
            auto texel = [&](float l, float r, float g, float b) {
                const float cos_angle = lcpc + lsps * l;
                unsigned ind;
#ifdef RMDF_HOST_EMULATION
                ind = __float_as_int(cos_angle) < 0 ? 0u : (__float_as_int(cos_angle) > 1 ? 1u : (unsigned)__float_as_int(cos_angle));   // (med3 of the bits as int32, 0, 1)
#else
                asm("v_med3_i32 %0, %1, 0, 1" : "=v"(ind) : "v"(__float_as_int(cos_angle)));
#endif
                ni += ind;
                const float c0 = __builtin_fmaxf(cos_angle, 0.0f);
                float cp = c0;
                if (LOG2P > 0) {
                    double cd = (double)c0;
#pragma unroll
                    for (int k = 0; k < LOG2P; k++) cd = cd * cd;
                    cp = (float)cd;
                }
                const float fac = ps * cp;
                const f2 rg = { r, g };
                const f2 fac2 = { fac, fac };
                arg = arg + rg * fac2;
                ab = ab + b * fac;
            };
            // Sixteen texels per block over TWO register sets (A, B) that alternate without being copied: the LDS reads of block
            // i + 1 are in flight while block i computes (LDS returns in order: the compiler waits with lgkmcnt(n) for exactly the
            // older block).
            constexpr int PB = 16;
            const int wb = w - w % PB;
            float ra[3 * PB], la[PB], rb[3 * PB], lb[PB];
            auto load_block = [&](float (&r)[3 * PB], float (&l)[PB], int x) {
                const float4 *q = (const float4 *)(row + x * 3);           // 16-byte aligned: 192 bytes per block, row_stride % 4 == 0
#pragma unroll
                for (int k = 0; k < 3 * PB / 4; k++) { const float4 v = q[k]; r[4 * k] = v.x; r[4 * k + 1] = v.y; r[4 * k + 2] = v.z; r[4 * k + 3] = v.w; }
#pragma unroll
                for (int k = 0; k < PB; k++) l[k] = LUT_IN_LDS ? lds_lut[(x + k) * 64 + lane] : glut[(x + k) * 64 + lane];
            };
            auto compute_block = [&](const float (&r)[3 * PB], const float (&l)[PB]) {
#pragma unroll
                for (int k = 0; k < PB; k++) texel(l[k], r[3 * k], r[3 * k + 1], r[3 * k + 2]);
            };
            if (wb > 0) load_block(ra, la, 0);
            int x = 0;
            for (; x + 2 * PB <= wb; x += 2 * PB) {
                load_block(rb, lb, x + PB);
                compute_block(ra, la);
                load_block(ra, la, (x + 2 * PB < wb) ? x + 2 * PB : 0);      // past the last block: block 0 again (valid, unused) -- no branch, no copies
                compute_block(rb, lb);
            }
            if (x < wb) compute_block(ra, la);                  // an odd number of blocks: the last one is loaded already
            for (int xt = wb; xt < w; xt++)
                texel(LUT_IN_LDS ? lds_lut[xt * 64 + lane] : glut[xt * 64 + lane], row[xt * 3], row[xt * 3 + 1], row[xt * 3 + 2]);
        } else {
            for (int x = 0; x < w; x++) {
                const float l = LUT_IN_LDS ? lds_lut[x * 64 + lane] : glut[x * 64 + lane];
                const float cos_angle = lcpc + lsps * l;
                if (cos_angle > 0.0f) {
                    const float fac = ps * powf(cos_angle, power);
                    arg.x = arg.x + row[x * 3] * fac; arg.y = arg.y + row[x * 3 + 1] * fac; ab = ab + row[x * 3 + 2] * fac;
                    ni++;
                }
            }
        }
        // hand the next row over (its buffer was last read for row y - 1: every wave is past the barrier that ended that row)
        if (nbuf == 1) __syncthreads();
        if (more) {
            float *nrowbuf = lds_row + (nbuf == 2 ? ((y + 1) & 1) : 0) * row_stride;
#pragma unroll
            for (int k = 0; k < NPF; k++) { const int i = threadIdx.x + k * 64 * PREFILTER_WAVES; if (i < nrow) nrowbuf[i] = pf[k]; }
            for (int i = threadIdx.x + NPF * 64 * PREFILTER_WAVES; i < nrow; i += 64 * PREFILTER_WAVES) nrowbuf[i] = nsrc[i];      // rows wider than 256 texels
        }
        __syncthreads();
    }
    const float ar = arg.x, ag = arg.y;
    const float n = (float)(ni < 16777216u ? ni : 16777216u);       // a Float counter: n + 1 == n from 2^24 on
    if (dx < w && dy_raw < h) {
        float *o = out + ((size_t)dx + (size_t)dy * w) * 3;
        o[0] = ar / n; o[1] = ag / n; o[2] = ab / n;
    }
}

// Source texels for a summing wave without broadcast LDS reads: the wave reads sixteen consecutive floats of the staged row ONCE into the
// sixteen lanes of each of its four rows (ds_read_b32, 256 bytes returned, where a broadcast ds_read_b128 returns 1 KB for 16 distinct
// bytes) and multiplies straight out of that register with v_mul_f32_dpp row_newbcast:k -- the DPP operand fetch does the broadcast:
// no LDS traffic, no extra instruction (hipcc does not fold update_dpp into the multiply, hence the inline asm); same IEEE product.
// The DPP operand must not have been written by a vector instruction in the two slots before (a gfx9 hazard the compiler does not see
// inside inline asm): here it always comes from an LDS read; tests/test_host_logic.py checks the generated assembly for it.
template <int K>
__device__ __forceinline__ float mul_row_bcast(float row16, float f)       // (lane K of each 16-lane row of row16) * f
{
    float r;
#ifdef RMDF_HOST_EMULATION       // (tests/kernel_on_host.cpp: the broadcast is a wave collective of the emulator, the product the same IEEE multiply)
    r = koh_row_newbcast(row16, K) * f;
#else
    asm("v_mul_f32_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(row16), "v"(f), "n"(K));
#endif
    return r;
}
// texel T of a chunk (static): its three floats sit at 3T .. 3T+2 of the chunk's floats, held sixteen per register
template <int T, int NR>
__device__ __forceinline__ void f4_sum_texel(const float (&rr)[NR], float f, float &ar, float &ag, float &ab)
{
    ar = ar + mul_row_bcast<(3 * T) % 16>(rr[(3 * T) / 16], f);
    ag = ag + mul_row_bcast<(3 * T + 1) % 16>(rr[(3 * T + 1) / 16], f);
    ab = ab + mul_row_bcast<(3 * T + 2) % 16>(rr[(3 * T + 2) / 16], f);
}
template <int G, int NR>
__device__ __forceinline__ void f4_sum_group(const float (&rr)[NR], const float4 &f, float &ar, float &ag, float &ab)
{
    f4_sum_texel<4 * G, NR>(rr, f.x, ar, ag, ab);
    f4_sum_texel<4 * G + 1, NR>(rr, f.y, ar, ag, ab);
    f4_sum_texel<4 * G + 2, NR>(rr, f.z, ar, ag, ab);
    f4_sum_texel<4 * G + 3, NR>(rr, f.w, ar, ag, ab);
}

// Round 3, second form (w <= 256, w % 4 == 0, the reference's powers): the per-destination chain is split between waves WITHOUT
// touching the order of a single addition.  One lane per destination texel is all the parallelism the sums allow, and at 256x128
// that is 512 waves, each issuing one vector instruction per ~5 cycles for 32768 x ~12 instructions.  But of those twelve only the
// multiply-adds of r, g and b belong to the chain; the rest computes the factor sin(theta) * cos^p, which depends on nothing the lane
// has summed.  So SUMMING waves (64 destination texels of one row, as before) get PRODUCER waves on the other SIMDs of their CU: these
// evaluate the factors of groups of four source texels for the same 64 destination lanes and hand them over through an LDS ring, one
// chunk of source texels per workgroup barrier, double-buffered; the summing waves read factor and texel and do the reference's
// operations in the reference's order.  The sample count n is an integer (see k_prefilter), so the producers count.  Two kernels are
// built this way: k_prefilter_fused4 (the reference's four powers at once) and k_prefilter_chan (one power).  The first kernel of
// the kind (k_prefilter_split: two summing waves + six producers per workgroup, cosine table in LDS, packed (r, g) sums on broadcast
// row reads: 0.54 / 0.62 / 0.72 / 0.82 ms per power) is in the history of this file; NOTEBOOK.md A.3 has its measurements.

// Round 3, third form: the reference's FOUR powers (1, 8, 64, 512: buildPreConvolvedHDREnvMapCache, ShaderRendering.hs:131-149) in one
// launch.  The four maps differ only in the exponent, and the binary64 squaring chains nest: c^8 is three squarings, c^64 three more
// ON THE SAME VALUE, c^512 three more -- nine multiplications where four separate launches do eighteen, one cosine, one table read,
// one sample count.  A workgroup is four summing waves (one per power, one per SIMD) fed by eight producer waves through four LDS rings,
// 32 source texels per barrier.  With four summing waves per CU the LDS return path becomes the scarce unit, and three quarters of what
// the first kernel of this kind moved through it was the SOURCE ROW, broadcast to 64 lanes by ds_read_b128 (1 KB per instruction for 16 distinct
// bytes).  Here a summing wave reads sixteen consecutive floats of the row ONCE into the sixteen lanes of each row of the wave
// (ds_read_b32, 256 bytes) and multiplies straight out of that register with v_mul_f32_dpp row_newbcast:k -- the DPP operand fetch
// does the broadcast, no LDS traffic, no extra instruction; the products and sums are the reference's, in its order.
#define F4_PROD 8                     // producer waves: one group of four source texels each per chunk
#define F4_CHUNK 32                   // source texels per hand-over
#define F4_STRIDE 36                  // floats per destination lane in a chunk buffer: 16-byte aligned, 4 banks apart
#define F4_MAXCH 8                     // chunks per source row at most (w <= 256)
__global__ __launch_bounds__(64 * (4 + F4_PROD), 6) void k_prefilter_fused4(const float *__restrict__ src, int w, int h,
        const float *__restrict__ lutT, const float2 *__restrict__ tcs, float *__restrict__ out0, float *__restrict__ out1,
        float *__restrict__ out2, float *__restrict__ out3)
{
    // LDS: the four rings and two staged source rows, 78 KB -- two workgroups per CU.  The cosine table stays out of it: a producer
    // meets the same 32 entries (its group of each of the row's <= 8 chunks, for its lane's destination column) in every source row
    // and keeps them in registers; the chunk loop is unrolled so that every index into them is static.
#ifdef RMDF_HOST_EMULATION
    float *lds_dyn = (float *)koh::dyn_lds();
#else
    extern __shared__ float lds_dyn[];
#endif
    const int nch = (w + F4_CHUNK - 1) / F4_CHUNK;                      // chunks per source row
    const int row_stride = nch * F4_CHUNK * 3;                           // floats per staged row, zero beyond w * 3
    float *lds_row = lds_dyn;                                            // [2][row_stride]
    float *lds_ring = lds_row + 2 * row_stride;                          // [4 powers][2][64][F4_STRIDE]
    unsigned *lds_cnt = (unsigned *)lds_ring;                            // [F4_PROD][64], after the last chunk has been summed
    unsigned *lds_flag = (unsigned *)(lds_ring + 4 * 2 * 64 * F4_STRIDE); // [2][F4_PROD]: does any lane have a positive cosine in this group?
    const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool consumer = g < 4;
    const int pidx = consumer ? 0 : g - 4;
    const int blk = blockIdx.x, dy = blockIdx.y;
    const int dx = blk * 64 + lane;
    const float *glut = lutT + (size_t)blk * w * 64;
    constexpr int NT = 64 * (4 + F4_PROD), NP = 64 * F4_PROD, NPF = 2;  // 512 producer threads x 2 floats >= 768 = 256 texels
    for (int i = threadIdx.x; i < 2 * row_stride; i += NT) lds_row[i] = 0.0f;
    const int ptid = (int)threadIdx.x - 64 * 4;
    const int nrow = w * 3;
    float pf[NPF] = { 0.0f, 0.0f };
    float lutreg[F4_MAXCH][4];
    if (!consumer) {
#pragma unroll
        for (int k = 0; k < NPF; k++) { const int i = ptid + k * NP; pf[k] = i < nrow ? src[i] : 0.0f; }
#pragma unroll
        for (int j = 0; j < F4_MAXCH; j++)
#pragma unroll
            for (int t = 0; t < 4; t++) { const int x = j * F4_CHUNK + 4 * pidx + t; lutreg[j][t] = x < w ? glut[x * 64 + lane] : 0.0f; }
    }
    typedef const float __attribute__((address_space(4))) cfloat;
    const float lc = ((cfloat *)tcs)[2 * dy], ls = ((cfloat *)tcs)[2 * dy + 1];
    float ar = 0.0f, ag = 0.0f, ab = 0.0f;
    unsigned ni = 0u;
    float *ring_c = lds_ring + g * 2 * 64 * F4_STRIDE + lane * F4_STRIDE;      // (consumers: their power's ring)
    float *ring_p = lds_ring + lane * F4_STRIDE;
    __syncthreads();
    // Step s: the producers fill chunk s (source row y, chunk j of the row) into ring buffer s & 1 while the summing waves take chunk
    // s - 1 out of the other one; one barrier per step; h * nch + 1 steps.
    int s = 0;
    for (int y = 0; y <= h; y++) {
        float lcpc = 0.0f, lsps = 0.0f, ps = 0.0f;
        if (!consumer && y < h) {
            const float pc = ((cfloat *)tcs)[2 * y];
            ps = ((cfloat *)tcs)[2 * y + 1];
            lcpc = lc * pc; lsps = ls * ps;
        }
#pragma unroll
        for (int j = 0; j < F4_MAXCH; j++) {
            if (j >= nch || (y == h && j > 0)) continue;                // (y == h: the one step that only sums the last chunk)
            if (!consumer) {
                if (y < h) {
                    if (j == 0) {
                        float *rb = lds_row + (y & 1) * row_stride;
#pragma unroll
                        for (int k = 0; k < NPF; k++) { const int i = ptid + k * NP; if (i < nrow) rb[i] = pf[k]; }
                    }
                    if (j == nch - 1 && y + 1 < h) {
                        const float *nsrc = src + (size_t)(y + 1) * nrow;
#pragma unroll
                        for (int k = 0; k < NPF; k++) { const int i = ptid + k * NP; pf[k] = i < nrow ? nsrc[i] : 0.0f; }
                    }
                    const int x0 = j * F4_CHUNK;
                    const int ng = ((w - x0 < F4_CHUNK ? w - x0 : F4_CHUNK) + 3) >> 2;
                    if (pidx < ng) {
                        // A group of four source texels that lies behind ALL 64 destination texels of this workgroup (every cosine <= 0:
                        // a quarter to a half of the sphere, depending on the row) contributes +-0 to every sum, and x + (+-0) == x: the
                        // producers skip its squaring chains and say so, the summing waves skip its twelve products and sums.
                        float c0[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            const float cos_angle = lcpc + lsps * lutreg[j][t];
                            unsigned ind;
#ifdef RMDF_HOST_EMULATION
                            ind = __float_as_int(cos_angle) < 0 ? 0u : (__float_as_int(cos_angle) > 1 ? 1u : (unsigned)__float_as_int(cos_angle));   // (med3 of the bits as int32, 0, 1)
#else
                            asm("v_med3_i32 %0, %1, 0, 1" : "=v"(ind) : "v"(__float_as_int(cos_angle)));
#endif
                            ni += ind;
                            c0[t] = __builtin_fmaxf(cos_angle, 0.0f);
                        }
                        const bool live = __ballot(__builtin_fmaxf(__builtin_fmaxf(c0[0], c0[1]), __builtin_fmaxf(c0[2], c0[3])) > 0.0f) != 0ull;
                        if (lane == 0) lds_flag[(s & 1) * F4_PROD + pidx] = live ? 1u : 0u;
                        if (live) {
                        float f1[4], f8[4], f64[4], f512[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            double cd = (double)c0[t];
                            cd = cd * cd; cd = cd * cd; cd = cd * cd;
                            const float c8 = (float)cd;
                            cd = cd * cd; cd = cd * cd; cd = cd * cd;
                            const float c64 = (float)cd;
                            cd = cd * cd; cd = cd * cd; cd = cd * cd;
                            const float c512 = (float)cd;
                            f1[t] = ps * c0[t]; f8[t] = ps * c8; f64[t] = ps * c64; f512[t] = ps * c512;
                        }
                        float *dst = ring_p + (s & 1) * 64 * F4_STRIDE + 4 * pidx;
                        *(float4 *)(dst) = make_float4(f1[0], f1[1], f1[2], f1[3]);
                        *(float4 *)(dst + 1 * 2 * 64 * F4_STRIDE) = make_float4(f8[0], f8[1], f8[2], f8[3]);
                        *(float4 *)(dst + 2 * 2 * 64 * F4_STRIDE) = make_float4(f64[0], f64[1], f64[2], f64[3]);
                        *(float4 *)(dst + 3 * 2 * 64 * F4_STRIDE) = make_float4(f512[0], f512[1], f512[2], f512[3]);
                        }
                    }
                }
            } else if (s > 0) {
                const int sc = s - 1, yc = j == 0 ? y - 1 : y, jc = j == 0 ? nch - 1 : j - 1;      // the chunk filled in the step before
                const int x0 = jc * F4_CHUNK;
                const int ng = ((w - x0 < F4_CHUNK ? w - x0 : F4_CHUNK) + 3) >> 2;
                const float *fsrc = ring_c + (sc & 1) * 64 * F4_STRIDE;
                const float *rrow = lds_row + (yc & 1) * row_stride + x0 * 3 + (lane & 15);
                float rr[F4_CHUNK * 3 / 16];
                float4 f[F4_CHUNK / 4];
#pragma unroll
                for (int m = 0; m < F4_CHUNK * 3 / 16; m++) rr[m] = rrow[16 * m];
#pragma unroll
                for (int k = 0; k < F4_CHUNK / 4; k++) f[k] = *(const float4 *)(fsrc + 4 * k);        // (groups past ng: stale, unused)
                // groups the producers declared dead (and groups past the row's end) are skipped: their ring slots hold stale factors
                const unsigned fl = lds_flag[(sc & 1) * F4_PROD + (lane & (F4_PROD - 1))];
                const unsigned live = (unsigned)__ballot(fl != 0u) & ((ng >= F4_CHUNK / 4) ? 0xffu : ((1u << ng) - 1u));
                if (live & 1u)   f4_sum_group<0>(rr, f[0], ar, ag, ab);
                if (live & 2u)   f4_sum_group<1>(rr, f[1], ar, ag, ab);
                if (live & 4u)   f4_sum_group<2>(rr, f[2], ar, ag, ab);
                if (live & 8u)   f4_sum_group<3>(rr, f[3], ar, ag, ab);
                if (live & 16u)  f4_sum_group<4>(rr, f[4], ar, ag, ab);
                if (live & 32u)  f4_sum_group<5>(rr, f[5], ar, ag, ab);
                if (live & 64u)  f4_sum_group<6>(rr, f[6], ar, ag, ab);
                if (live & 128u) f4_sum_group<7>(rr, f[7], ar, ag, ab);
            }
            __syncthreads();
            s++;
        }
    }
    if (!consumer) lds_cnt[pidx * 64 + lane] = ni;                       // (the rings are free: the barrier above ended the last sum)
    __syncthreads();
    if (consumer) {
        unsigned nt = 0u;
#pragma unroll
        for (int k = 0; k < F4_PROD; k++) nt += lds_cnt[k * 64 + lane];
        const float n = (float)(nt < 16777216u ? nt : 16777216u);        // a Float counter: n + 1 == n from 2^24 on
        float *ob = g == 0 ? out0 : g == 1 ? out1 : g == 2 ? out2 : out3;      // null: a power the caller did not ask for
        if (dx < w && ob) {
            float *o = ob + ((size_t)dx + (size_t)dy * w) * 3;
            o[0] = ar / n; o[1] = ag / n; o[2] = ab / n;
        }
    }
}

// Round 3, fourth form: ONE power with the fused kernel's machinery and the sum split by CHANNEL.  r, g and b are three independent
// chains (the channel split of k_prefilter lost because every lane repeated the weight; here the weight comes from the producers,
// once): three summing waves -- one per channel, two instructions per source texel each (v_mul_f32_dpp + v_add_f32) -- share one ring
// of factors written by eight producer waves.  Same ring, row registers, table in registers, dead-group skipping and barrier scheme as
// k_prefilter_fused4, with 64-texel chunks (one power's ring is small): 41 KB of LDS, two workgroups of eleven waves per CU.
#define CH_PROD 8                     // producer waves: two groups of four source texels each per chunk
#define CH_CHUNK 64                   // source texels per hand-over (one power's ring is small: half as many barriers as the fused kernel)
#define CH_STRIDE 68                  // floats per destination lane in a chunk buffer
#define CH_MAXCH 4                    // chunks per source row at most (w <= 256)
#define CH_GPP (CH_CHUNK / 4 / CH_PROD)
template <int G, int CH, int NR>
__device__ __forceinline__ void ch_sum_group(const float (&rr)[NR], const float4 &f, float &a)
{
    // the four products first (independent), then the chain of sums: a sum never waits for a product issued just before it
    const float p0 = mul_row_bcast<(12 * G + CH) % 16>(rr[(12 * G + CH) / 16], f.x);
    const float p1 = mul_row_bcast<(12 * G + 3 + CH) % 16>(rr[(12 * G + 3 + CH) / 16], f.y);
    const float p2 = mul_row_bcast<(12 * G + 6 + CH) % 16>(rr[(12 * G + 6 + CH) / 16], f.z);
    const float p3 = mul_row_bcast<(12 * G + 9 + CH) % 16>(rr[(12 * G + 9 + CH) / 16], f.w);
    a = a + p0; a = a + p1; a = a + p2; a = a + p3;
}
// eight groups (half a chunk) starting at group G0; `live` holds their flags in its low eight bits
template <int G0, int CH, int NR>
__device__ __forceinline__ void ch_sum_half(const float (&rr)[NR], const float4 (&f)[CH_CHUNK / 4], unsigned live, float &a)
{
    if (live & 1u)   ch_sum_group<G0 + 0, CH>(rr, f[G0 + 0], a);
    if (live & 2u)   ch_sum_group<G0 + 1, CH>(rr, f[G0 + 1], a);
    if (live & 4u)   ch_sum_group<G0 + 2, CH>(rr, f[G0 + 2], a);
    if (live & 8u)   ch_sum_group<G0 + 3, CH>(rr, f[G0 + 3], a);
    if (live & 16u)  ch_sum_group<G0 + 4, CH>(rr, f[G0 + 4], a);
    if (live & 32u)  ch_sum_group<G0 + 5, CH>(rr, f[G0 + 5], a);
    if (live & 64u)  ch_sum_group<G0 + 6, CH>(rr, f[G0 + 6], a);
    if (live & 128u) ch_sum_group<G0 + 7, CH>(rr, f[G0 + 7], a);
}

template <int LOG2P>
__global__ __launch_bounds__(64 * (3 + CH_PROD), 6) void k_prefilter_chan(const float *__restrict__ src, int w, int h,
        const float *__restrict__ lutT, const float2 *__restrict__ tcs, float *__restrict__ out)
{
#ifdef RMDF_HOST_EMULATION
    float *lds_dyn = (float *)koh::dyn_lds();
#else
    extern __shared__ float lds_dyn[];
#endif
    const int nch = (w + CH_CHUNK - 1) / CH_CHUNK;                      // chunks per source row
    const int row_stride = nch * CH_CHUNK * 3;                           // floats per staged row, zero beyond w * 3
    float *lds_row = lds_dyn;                                            // [2][row_stride]
    float *lds_ring = lds_row + 2 * row_stride;                          // [2][64][CH_STRIDE]
    unsigned *lds_cnt = (unsigned *)lds_ring;                            // [CH_PROD][64], after the last chunk has been summed
    unsigned *lds_flag = (unsigned *)(lds_ring + 2 * 64 * CH_STRIDE);    // [2][CH_CHUNK / 4]
    const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool consumer = g < 3;                                         // summing wave g = channel g
    const int pidx = consumer ? 0 : g - 3;
    const int blk = blockIdx.x, dy = blockIdx.y;
    const int dx = blk * 64 + lane;
    const float *glut = lutT + (size_t)blk * w * 64;
    constexpr int NT = 64 * (3 + CH_PROD), NP = 64 * CH_PROD, NPF = 2;  // 512 producer threads x 2 floats >= 768 = 256 texels
    for (int i = threadIdx.x; i < 2 * row_stride; i += NT) lds_row[i] = 0.0f;
    const int ptid = (int)threadIdx.x - 64 * 3;
    const int nrow = w * 3;
    float pf[NPF] = { 0.0f, 0.0f };
    float lutreg[CH_MAXCH][CH_GPP][4];
    if (!consumer) {
#pragma unroll
        for (int k = 0; k < NPF; k++) { const int i = ptid + k * NP; pf[k] = i < nrow ? src[i] : 0.0f; }
#pragma unroll
        for (int j = 0; j < CH_MAXCH; j++)
#pragma unroll
            for (int k = 0; k < CH_GPP; k++)
#pragma unroll
                for (int t = 0; t < 4; t++) { const int x = j * CH_CHUNK + 4 * (pidx + k * CH_PROD) + t; lutreg[j][k][t] = x < w ? glut[x * 64 + lane] : 0.0f; }
    }
    typedef const float __attribute__((address_space(4))) cfloat;
    const float lc = ((cfloat *)tcs)[2 * dy], ls = ((cfloat *)tcs)[2 * dy + 1];
    float acc = 0.0f;
    unsigned ni = 0u;
    float *ring = lds_ring + lane * CH_STRIDE;
    __syncthreads();
    int s = 0;
    float pcn = ((cfloat *)tcs)[0], psn = ((cfloat *)tcs)[1];           // (cos, sin) of the NEXT source row: loaded a row ahead
    for (int y = 0; y <= h; y++) {
        const float pc = pcn, ps = psn;
        const float lcpc = lc * pc, lsps = ls * ps;
        if (y + 1 < h) { pcn = ((cfloat *)tcs)[2 * (y + 1)]; psn = ((cfloat *)tcs)[2 * (y + 1) + 1]; }
#pragma unroll
        for (int j = 0; j < CH_MAXCH; j++) {
            if (j >= nch || (y == h && j > 0)) continue;                // (y == h: the one step that only sums the last chunk)
            if (!consumer) {
                if (y < h) {
                    if (j == 0) {
                        // hand row y to the summing waves, then fetch row y + 1 at once: it has the row's remaining steps to arrive
                        float *rb = lds_row + (y & 1) * row_stride;
#pragma unroll
                        for (int k = 0; k < NPF; k++) { const int i = ptid + k * NP; if (i < nrow) rb[i] = pf[k]; }
                        if (y + 1 < h) {
                            const float *nsrc = src + (size_t)(y + 1) * nrow;
#pragma unroll
                            for (int k = 0; k < NPF; k++) { const int i = ptid + k * NP; pf[k] = i < nrow ? nsrc[i] : 0.0f; }
                        }
                    }
                    const int x0 = j * CH_CHUNK;
                    const int ng = ((w - x0 < CH_CHUNK ? w - x0 : CH_CHUNK) + 3) >> 2;
#pragma unroll
                    for (int k = 0; k < CH_GPP; k++) {
                        const int gi = pidx + k * CH_PROD;
                        if (gi < ng) {
                            float c0[4];
#pragma unroll
                            for (int t = 0; t < 4; t++) {
                                const float cos_angle = lcpc + lsps * lutreg[j][k][t];
                                unsigned ind;
#ifdef RMDF_HOST_EMULATION
                                ind = __float_as_int(cos_angle) < 0 ? 0u : (__float_as_int(cos_angle) > 1 ? 1u : (unsigned)__float_as_int(cos_angle));   // (med3 of the bits as int32, 0, 1)
#else
                                asm("v_med3_i32 %0, %1, 0, 1" : "=v"(ind) : "v"(__float_as_int(cos_angle)));
#endif
                                ni += ind;
                                c0[t] = __builtin_fmaxf(cos_angle, 0.0f);
                            }
                            const bool live = __ballot(__builtin_fmaxf(__builtin_fmaxf(c0[0], c0[1]), __builtin_fmaxf(c0[2], c0[3])) > 0.0f) != 0ull;
                            if (lane == 0) lds_flag[(s & 1) * (CH_CHUNK / 4) + gi] = live ? 1u : 0u;
                            if (live) {
                                float fac[4];
#pragma unroll
                                for (int t = 0; t < 4; t++) {
                                    float cp = c0[t];
                                    if (LOG2P > 0) {
                                        double cd = (double)c0[t];
#pragma unroll
                                        for (int q = 0; q < LOG2P; q++) cd = cd * cd;
                                        cp = (float)cd;
                                    }
                                    fac[t] = ps * cp;
                                }
                                *(float4 *)(ring + (s & 1) * 64 * CH_STRIDE + 4 * gi) = make_float4(fac[0], fac[1], fac[2], fac[3]);
                            }
                        }
                    }
                }
            } else if (s > 0) {
                const int sc = s - 1, yc = j == 0 ? y - 1 : y, jc = j == 0 ? nch - 1 : j - 1;      // the chunk filled in the step before
                const int x0 = jc * CH_CHUNK;
                const int ng = ((w - x0 < CH_CHUNK ? w - x0 : CH_CHUNK) + 3) >> 2;
                const float *fsrc = ring + (sc & 1) * 64 * CH_STRIDE;
                const float *rrow = lds_row + (yc & 1) * row_stride + x0 * 3 + (lane & 15);
                float rr[CH_CHUNK * 3 / 16];
                float4 f[CH_CHUNK / 4];
                const unsigned fl = lds_flag[(sc & 1) * (CH_CHUNK / 4) + (lane & (CH_CHUNK / 4 - 1))];
                const unsigned live = (unsigned)__ballot(fl != 0u) & ((ng >= CH_CHUNK / 4) ? 0xffffu : ((1u << ng) - 1u));
                // two halves of eight groups (32 texels = six row registers): the second half's LDS reads are issued after the first half's sums
#pragma unroll
                for (int m = 0; m < 6; m++) rr[m] = rrow[16 * m];
#pragma unroll
                for (int k = 0; k < 8; k++) f[k] = *(const float4 *)(fsrc + 4 * k);                   // (dead groups: stale, unused)
                if (g == 0)      ch_sum_half<0, 0>(rr, f, live & 0xffu, acc);
                else if (g == 1) ch_sum_half<0, 1>(rr, f, live & 0xffu, acc);
                else             ch_sum_half<0, 2>(rr, f, live & 0xffu, acc);
                if (live >> 8) {
#pragma unroll
                    for (int m = 6; m < 12; m++) rr[m] = rrow[16 * m];
#pragma unroll
                    for (int k = 8; k < 16; k++) f[k] = *(const float4 *)(fsrc + 4 * k);
                    if (g == 0)      ch_sum_half<8, 0>(rr, f, live >> 8, acc);
                    else if (g == 1) ch_sum_half<8, 1>(rr, f, live >> 8, acc);
                    else             ch_sum_half<8, 2>(rr, f, live >> 8, acc);
                }
            }
            __syncthreads();
            s++;
        }
    }
    if (!consumer) lds_cnt[pidx * 64 + lane] = ni;                       // (the ring is free: the barrier above ended the last sum)
    __syncthreads();
    if (consumer) {
        unsigned nt = 0u;
#pragma unroll
        for (int k = 0; k < CH_PROD; k++) nt += lds_cnt[k * 64 + lane];
        const float n = (float)(nt < 16777216u ? nt : 16777216u);        // a Float counter: n + 1 == n from 2^24 on
        if (dx < w) out[((size_t)dx + (size_t)dy * w) * 3 + g] = acc / n;
    }
}

#ifdef RMDF_XCHECK        // librmdf_xcheck.so only, until it has had a green run on hardware
// Round 5, fifth form (A/B only: RMDF_PREFILTER_RING=1; written after GPU access closed, NOT yet run on hardware).  k_prefilter_chan's
// counters say no unit is busy -- LDS array 0.31, vector issue 0.34, scalar issue 0.37 of their peaks -- and its waves wait 53 % of their
// cycles: producers and summing waves meet at one s_barrier per 64 source texels, 513 times per workgroup, and whoever arrives first waits.
// Here they never meet.  The factors go through a ring of RING_SLOTS chunk buffers with two counters per slot in LDS: a producer wave adds
// one to `filled` when its part of a chunk is written (all eight do, every chunk), a summing wave adds one to `drained` when it has read
// the chunk out; chunk s lives in slot s % RING_SLOTS, is readable when filled == (s / RING_SLOTS + 1) * producers and writable when the
// chunk RING_SLOTS before it is drained by all three summing waves.  So the producers run up to three chunks ahead and the stalls of one
// side are absorbed by the ring instead of being handed to the other.  Same factors, same order of sums per destination texel and channel:
// the arithmetic is k_prefilter_chan's line for line.  32-texel chunks (four slots fit the LDS budget of two workgroups per CU); needs at
// least RING_SLOTS chunks per source row (w >= 128), so that a staged source row is never overwritten while a summing wave still reads it.
#define RING_PROD 8
#define RING_CHUNK 32
#define RING_STRIDE 36
#define RING_SLOTS 4
#define RING_MAXCH 8                  // chunks per source row at most (w <= 256)
template <int G, int CH, int NR>
__device__ __forceinline__ void ring_sum_group(const float (&rr)[NR], const float4 &f, float &a)
{
    const float p0 = mul_row_bcast<(12 * G + CH) % 16>(rr[(12 * G + CH) / 16], f.x);
    const float p1 = mul_row_bcast<(12 * G + 3 + CH) % 16>(rr[(12 * G + 3 + CH) / 16], f.y);
    const float p2 = mul_row_bcast<(12 * G + 6 + CH) % 16>(rr[(12 * G + 6 + CH) / 16], f.z);
    const float p3 = mul_row_bcast<(12 * G + 9 + CH) % 16>(rr[(12 * G + 9 + CH) / 16], f.w);
    a = a + p0; a = a + p1; a = a + p2; a = a + p3;
}
template <int CH, int NR>
__device__ __forceinline__ void ring_sum_chunk(const float (&rr)[NR], const float4 (&f)[RING_CHUNK / 4], unsigned live, float &a)
{
    if (live & 1u)   ring_sum_group<0, CH>(rr, f[0], a);
    if (live & 2u)   ring_sum_group<1, CH>(rr, f[1], a);
    if (live & 4u)   ring_sum_group<2, CH>(rr, f[2], a);
    if (live & 8u)   ring_sum_group<3, CH>(rr, f[3], a);
    if (live & 16u)  ring_sum_group<4, CH>(rr, f[4], a);
    if (live & 32u)  ring_sum_group<5, CH>(rr, f[5], a);
    if (live & 64u)  ring_sum_group<6, CH>(rr, f[6], a);
    if (live & 128u) ring_sum_group<7, CH>(rr, f[7], a);
}
__device__ __forceinline__ void ring_wait_at_least(unsigned *counter, unsigned target)
{
    // the whole wave waits on one LDS word (wave-uniform address: a broadcast read); s_sleep keeps the spinning wave off the issue ports
    while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
}

template <int LOG2P>
__global__ __launch_bounds__(64 * (3 + RING_PROD), 6) void k_prefilter_ring(const float *__restrict__ src, int w, int h,
        const float *__restrict__ lutT, const float2 *__restrict__ tcs, float *__restrict__ out)
{
#ifdef RMDF_HOST_EMULATION
    float *lds_dyn = (float *)koh::dyn_lds();
#else
    extern __shared__ float lds_dyn[];
#endif
    const int nch = (w + RING_CHUNK - 1) / RING_CHUNK;                   // chunks per source row, >= RING_SLOTS (launcher)
    const int row_stride = nch * RING_CHUNK * 3;                          // floats per staged row, zero beyond w * 3
    float *lds_row = lds_dyn;                                             // [2][row_stride]
    float *lds_ring = lds_row + 2 * row_stride;                           // [RING_SLOTS][64][RING_STRIDE]
    unsigned *lds_flag = (unsigned *)(lds_ring + RING_SLOTS * 64 * RING_STRIDE);     // [RING_SLOTS][8]: live groups of the chunk in the slot
    unsigned *lds_filled = lds_flag + RING_SLOTS * 8;                     // [RING_SLOTS]
    unsigned *lds_drained = lds_filled + RING_SLOTS;                      // [RING_SLOTS]
    unsigned *lds_cnt = lds_drained + RING_SLOTS;                         // [RING_PROD][64]: the producers' sample counts, at the end
    const int lane = threadIdx.x & 63, g = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const bool consumer = g < 3;                                          // summing wave g = channel g
    const int pidx = consumer ? 0 : g - 3;
    const int blk = blockIdx.x, dy = blockIdx.y;
    const int dx = blk * 64 + lane;
    const float *glut = lutT + (size_t)blk * w * 64;
    constexpr int NT = 64 * (3 + RING_PROD), NP = 64 * RING_PROD, NPF = 2;
    for (int i = threadIdx.x; i < 2 * row_stride; i += NT) lds_row[i] = 0.0f;
    if (threadIdx.x < 2 * RING_SLOTS) lds_filled[threadIdx.x] = 0u;       // filled[] and drained[] are adjacent
    const int ptid = (int)threadIdx.x - 64 * 3;
    const int nrow = w * 3;
    float pf[NPF] = { 0.0f, 0.0f };
    float lutreg[RING_MAXCH][4];
    if (!consumer) {
#pragma unroll
        for (int k = 0; k < NPF; k++) { const int i = ptid + k * NP; pf[k] = i < nrow ? src[i] : 0.0f; }
#pragma unroll
        for (int j = 0; j < RING_MAXCH; j++)
#pragma unroll
            for (int t = 0; t < 4; t++) { const int x = j * RING_CHUNK + 4 * pidx + t; lutreg[j][t] = x < w ? glut[x * 64 + lane] : 0.0f; }
    }
    typedef const float __attribute__((address_space(4))) cfloat;
    const float lc = ((cfloat *)tcs)[2 * dy], ls = ((cfloat *)tcs)[2 * dy + 1];
    float acc = 0.0f;
    unsigned ni = 0u;
    __syncthreads();                                                      // the only barrier before the end: rows zeroed, counters zeroed
    const unsigned total = (unsigned)h * (unsigned)nch;                   // chunks of this workgroup
    if (!consumer) {
        unsigned s = 0u;
        for (int y = 0; y < h; y++) {
            const float pc = ((cfloat *)tcs)[2 * y], ps = ((cfloat *)tcs)[2 * y + 1];
            const float lcpc = lc * pc, lsps = ls * ps;
#pragma unroll
            for (int j = 0; j < RING_MAXCH; j++) {
                if (j >= nch) continue;
                const unsigned slot = s % RING_SLOTS, round = s / RING_SLOTS;
                // the slot's previous chunk (s - RING_SLOTS) has been read out by all three summing waves.  That also covers the staged row:
                // row y goes into buffer y & 1 at j == 0, which row y - 2 used; its last chunk is s - nch - 1 <= s - RING_SLOTS - 1
                if (round > 0u) ring_wait_at_least(&lds_drained[slot], 3u * round);
                if (j == 0) {
                    float *rb = lds_row + (y & 1) * row_stride;
#pragma unroll
                    for (int k = 0; k < NPF; k++) { const int i = ptid + k * NP; if (i < nrow) rb[i] = pf[k]; }
                    if (y + 1 < h) {
                        const float *nsrc = src + (size_t)(y + 1) * nrow;
#pragma unroll
                        for (int k = 0; k < NPF; k++) { const int i = ptid + k * NP; pf[k] = i < nrow ? nsrc[i] : 0.0f; }
                    }
                }
                const int x0 = j * RING_CHUNK;
                const int ng = ((w - x0 < RING_CHUNK ? w - x0 : RING_CHUNK) + 3) >> 2;
                if (pidx < ng) {
                    float c0[4];
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        const float cos_angle = lcpc + lsps * lutreg[j][t];
                        unsigned ind;
#ifdef RMDF_HOST_EMULATION
                        ind = __float_as_int(cos_angle) < 0 ? 0u : (__float_as_int(cos_angle) > 1 ? 1u : (unsigned)__float_as_int(cos_angle));   // (med3 of the bits as int32, 0, 1)
#else
                        asm("v_med3_i32 %0, %1, 0, 1" : "=v"(ind) : "v"(__float_as_int(cos_angle)));
#endif
                        ni += ind;
                        c0[t] = __builtin_fmaxf(cos_angle, 0.0f);
                    }
                    const bool live = __ballot(__builtin_fmaxf(__builtin_fmaxf(c0[0], c0[1]), __builtin_fmaxf(c0[2], c0[3])) > 0.0f) != 0ull;
                    if (lane == 0) lds_flag[slot * 8 + pidx] = live ? 1u : 0u;
                    if (live) {
                        float fac[4];
#pragma unroll
                        for (int t = 0; t < 4; t++) {
                            float cp = c0[t];
                            if (LOG2P > 0) {
                                double cd = (double)c0[t];
#pragma unroll
                                for (int q = 0; q < LOG2P; q++) cd = cd * cd;
                                cp = (float)cd;
                            }
                            fac[t] = ps * cp;
                        }
                        *(float4 *)(lds_ring + slot * 64 * RING_STRIDE + lane * RING_STRIDE + 4 * pidx) = make_float4(fac[0], fac[1], fac[2], fac[3]);
                    }
                } else if (lane == 0 && pidx < 8) {
                    lds_flag[slot * 8 + pidx] = 0u;                       // a group past the row's end
                }
                // this wave's part of chunk s (row floats, flag, factors) is in LDS: count it in
                if (lane == 0) __hip_atomic_fetch_add(&lds_filled[slot], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                s++;
            }
        }
    } else {
        unsigned s = 0u;
        for (int y = 0; y < h; y++) {
#pragma unroll
            for (int j = 0; j < RING_MAXCH; j++) {
                if (j >= nch) continue;
                const unsigned slot = s % RING_SLOTS, round = s / RING_SLOTS;
                ring_wait_at_least(&lds_filled[slot], (unsigned)RING_PROD * (round + 1u));
                const int x0 = j * RING_CHUNK;
                const int ng = ((w - x0 < RING_CHUNK ? w - x0 : RING_CHUNK) + 3) >> 2;
                const float *fsrc = lds_ring + slot * 64 * RING_STRIDE + lane * RING_STRIDE;
                const float *rrow = lds_row + (y & 1) * row_stride + x0 * 3 + (lane & 15);
                float rr[RING_CHUNK * 3 / 16];
                float4 f[RING_CHUNK / 4];
                const unsigned fl = lds_flag[slot * 8 + (lane & 7)];
                const unsigned live = (unsigned)__ballot(fl != 0u) & ((ng >= 8) ? 0xffu : ((1u << ng) - 1u));
#pragma unroll
                for (int m = 0; m < RING_CHUNK * 3 / 16; m++) rr[m] = rrow[16 * m];
#pragma unroll
                for (int k = 0; k < RING_CHUNK / 4; k++) f[k] = *(const float4 *)(fsrc + 4 * k);      // (dead groups: stale, unused)
                // everything this wave needs of the slot is in registers (the compiler waits for the reads before their first use; the
                // release below is ordered behind them): hand the slot back before the sums
#ifndef RMDF_HOST_EMULATION
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
                if (lane == 0) __hip_atomic_fetch_add(&lds_drained[slot], 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (g == 0)      ring_sum_chunk<0>(rr, f, live, acc);
                else if (g == 1) ring_sum_chunk<1>(rr, f, live, acc);
                else             ring_sum_chunk<2>(rr, f, live, acc);
                s++;
            }
        }
    }
    (void)total;
    if (!consumer) lds_cnt[pidx * 64 + lane] = ni;
    __syncthreads();
    if (consumer) {
        unsigned nt = 0u;
#pragma unroll
        for (int k = 0; k < RING_PROD; k++) nt += lds_cnt[k * 64 + lane];
        const float n = (float)(nt < 16777216u ? nt : 16777216u);        // a Float counter: n + 1 == n from 2^24 on
        if (dx < w) out[((size_t)dx + (size_t)dy * w) * 3 + g] = acc / n;
    }
}

template <int LOG2P>
static hipError_t launch_prefilter_ring_t(const float *d_src, int w, int h, const float *d_lutT, const float2 *d_tcs, float *d_out, hipStream_t stream)
{
    const int nch = (w + RING_CHUNK - 1) / RING_CHUNK;
    const size_t lds = (2 * (size_t)nch * RING_CHUNK * 3 + (size_t)RING_SLOTS * 64 * RING_STRIDE) * sizeof(float) +
                       ((size_t)RING_SLOTS * 8 + 2 * RING_SLOTS + (size_t)RING_PROD * 64) * sizeof(unsigned);
    hipLaunchKernelGGL((k_prefilter_ring<LOG2P>), dim3((w + 63) / 64, h), dim3(64 * (3 + RING_PROD)), lds, stream, d_src, w, h, d_lutT, d_tcs, d_out);
    return hipGetLastError();
}
#endif

template <int LOG2P>
static hipError_t launch_prefilter_chan_t(const float *d_src, int w, int h, const float *d_lutT, const float2 *d_tcs, float *d_out, hipStream_t stream)
{
    const int nch = (w + CH_CHUNK - 1) / CH_CHUNK;
    const size_t lds = (2 * (size_t)nch * CH_CHUNK * 3 + (size_t)2 * 64 * CH_STRIDE + 2 * (CH_CHUNK / 4)) * sizeof(float);
    hipLaunchKernelGGL((k_prefilter_chan<LOG2P>), dim3((w + 63) / 64, h), dim3(64 * (3 + CH_PROD)), lds, stream, d_src, w, h, d_lutT, d_tcs, d_out);
    return hipGetLastError();
}

// d_out[k] = the map of power 8^k (1, 8, 64, 512), or null
hipError_t launch_prefilter_fused4(const float *d_src, int w, int h, const float *d_lutT, const float2 *d_tcs, float *const d_out[4],
                                   hipStream_t stream)
{
    if (w < 2 || h < 2 || w > 256 || w % 4) return hipErrorInvalidValue;
    const int nch = (w + F4_CHUNK - 1) / F4_CHUNK;
    const size_t lds = (2 * (size_t)nch * F4_CHUNK * 3 + (size_t)4 * 2 * 64 * F4_STRIDE + 2 * F4_PROD) * sizeof(float);
    hipError_t e = hipFuncSetAttribute((const void *)k_prefilter_fused4, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k_prefilter_fused4, dim3((w + 63) / 64, h), dim3(64 * (4 + F4_PROD)), lds, stream, d_src, w, h, d_lutT, d_tcs,
                       d_out[0], d_out[1], d_out[2], d_out[3]);
    return hipGetLastError();
}

int prefilter_log2p(float power)
{
    if (power == 1.0f) return 0;
    if (power == 8.0f) return 3;
    if (power == 64.0f) return 6;
    if (power == 512.0f) return 9;
    return -1;
}

template <int LOG2P>
static hipError_t launch_prefilter_t(const float *d_src, int w, int h, float power, const float *d_lutT, const float2 *d_tcs,
                                     float *d_out, hipStream_t stream, bool split_ok)
{
#ifdef RMDF_XCHECK
    // A/B switches of the cross-check build (tools/, tests; read once per process).  The product library reads no environment variable.
    static const bool one_wave = getenv("RMDF_PREFILTER_ONE_WAVE") != nullptr;          // the one-wave kernel at every size
    static const bool ring = getenv("RMDF_PREFILTER_RING") != nullptr;                  // the barrier-free ring form (not yet run on hardware)
    if (LOG2P >= 0 && ring && w >= RING_SLOTS * RING_CHUNK && w <= 256 && w % 4 == 0 && split_ok && !one_wave)
        return launch_prefilter_ring_t<(LOG2P >= 0 ? LOG2P : 0)>(d_src, w, h, d_lutT, d_tcs, d_out, stream);
#else
    constexpr bool one_wave = false;
#endif
    if (LOG2P >= 0 && w <= 256 && w % 4 == 0 && split_ok && !one_wave)                  // the reference's size: factor and sum on different waves
        return launch_prefilter_chan_t<(LOG2P >= 0 ? LOG2P : 0)>(d_src, w, h, d_lutT, d_tcs, d_out, stream);
    const dim3 grid((w + 63) / 64, (h + PREFILTER_WAVES - 1) / PREFILTER_WAVES), block(64 * PREFILTER_WAVES);
    const size_t row = (size_t)((w * 3 + 3) & ~3) * sizeof(float);             // one staged source row
    const size_t lut = (size_t)w * 64 * sizeof(float);
    if (2 * row + lut <= 72 * 1024) {   // two workgroups per CU keep table and rows in LDS (w <= 256: 70 KB each); wider maps read the table through L2
        hipError_t e = hipFuncSetAttribute((const void *)k_prefilter<LOG2P, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(2 * row + lut));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_prefilter<LOG2P, true>), grid, block, 2 * row + lut, stream, d_src, w, h, power, d_lutT, d_tcs, d_out, 2);
    } else {
        const int nbuf = 2 * row <= 80 * 1024 ? 2 : 1;              // w <= 3413: two buffers and still two workgroups per CU
        if ((size_t)nbuf * row > 160 * 1024) return hipErrorInvalidValue;
        hipError_t e = hipFuncSetAttribute((const void *)k_prefilter<LOG2P, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(nbuf * row));
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((k_prefilter<LOG2P, false>), grid, block, (size_t)nbuf * row, stream, d_src, w, h, power, d_lutT, d_tcs, d_out, nbuf);
    }
    return hipGetLastError();
}

// split_ok: the caller runs this power alone (or beside one or two others).  k_prefilter_chan fills the machine by itself (5632 waves,
// two workgroups per CU): 0.52 / 0.60 / 0.62 / 0.65 ms for p = 1, 8, 64, 512 at 256x128 against the one-wave kernel's 0.92 / 1.17 / 1.33 /
// 1.54; with four or more launches side by side the one-wave kernel's, which overlap, finish sooner together, so
// rmdf_prefilter_env_powers keeps it for power sets that are not the reference's (whose two to four powers are ONE launch of
// k_prefilter_fused4).
hipError_t launch_prefilter(const float *d_src, int w, int h, float power, const float *d_lutT, const float2 *d_tcs,
                            float *d_out, hipStream_t stream, bool split_ok)
{
    if (w < 2 || h < 2) return hipErrorInvalidValue;
    switch (prefilter_log2p(power)) {
    case 0:  return launch_prefilter_t<0>(d_src, w, h, power, d_lutT, d_tcs, d_out, stream, split_ok);
    case 3:  return launch_prefilter_t<3>(d_src, w, h, power, d_lutT, d_tcs, d_out, stream, split_ok);
    case 6:  return launch_prefilter_t<6>(d_src, w, h, power, d_lutT, d_tcs, d_out, stream, split_ok);
    case 9:  return launch_prefilter_t<9>(d_src, w, h, power, d_lutT, d_tcs, d_out, stream, split_ok);
    default: return launch_prefilter_t<-1>(d_src, w, h, power, d_lutT, d_tcs, d_out, stream, split_ok);
    }
}

}  // namespace rmdf
