// rmdf_stats.hip -- measurement aid (librmdf_xcheck.so only): the march loop of the power-8 render kernel alone, with
// wave-level divergence counters.
#include "rmdf_internal.hpp"

namespace rmdf {

// same, counting wave-level inner passes and the lanes active in them (measurement builds only)
__device__ __forceinline__ float de_mandelbulb8_dbg(v3 pos, unsigned &iters, unsigned long long &passes, unsigned long long &lanes,
                                                    unsigned long long *hist)
{
    pos = mk3(pos.z, pos.x, pos.y);
    v3 w = pos;
    float dr = 1.0f;
    float r = 0.0f;
    for (int i = 0; i < 25; i++) {
        {
            const unsigned long long am = __ballot(true);
            const int na = __popcll(am);
            passes++; lanes += na;
            // histogram of the active-lane count of this pass, kept by the first active lane only
            if (hist && __builtin_amdgcn_mbcnt_hi((unsigned)(am >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)am, 0)) == 0) {
                const int b = na <= 2 ? 0 : na <= 4 ? 1 : na <= 8 ? 2 : na <= 16 ? 3 : na <= 32 ? 4 : na <= 48 ? 5 : 6;
                atomicAdd(&hist[b], 1ull);
            }
        }
        r = length3(w);
        if (r > 4.0f) break;
        w = triplex_pow8(w);
        w = add3(w, pos);
        float r2 = r * r, r4 = r2 * r2, r7 = (r4 * r2) * r;
        dr = r7 * 8.0f * dr + 1.0f;
        iters++;
    }
    return 0.5f * log_pinned(r) * r / dr;
}

// measurement aid: the march loop of k_render<2> alone, with wave-level divergence counters
__global__ __launch_bounds__(256) void k_march_stats(const FrameParams p)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lx = (lane & 1) | (((lane >> 2) & 3) << 1);
    const int ly = ((lane >> 1) & 1) | (((lane >> 4) & 3) << 1);
    const int px = blockIdx.x * 32 + wave * 8 + lx, py = blockIdx.y * 8 + ly;
    const bool active = (px < p.w) && (py < p.h);
    const float ndcx = ((float)px + 0.5f) / p.wf * 2.0f - 1.0f;
    const float ndcy = ((float)py + 0.5f) / p.hf * 2.0f - 1.0f;
    const v3 dcam = normalize3(mk3(ndcx * p.fov_xs, ndcy * p.fov_xs / p.aspect, -1.0f));
    const v3 dir = mk3(p.cam[0] * dcam.x + p.cam[3] * dcam.y + p.cam[6] * dcam.z,
                       p.cam[1] * dcam.x + p.cam[4] * dcam.y + p.cam[7] * dcam.z,
                       p.cam[2] * dcam.x + p.cam[5] * dcam.y + p.cam[8] * dcam.z);
    const v3 origin = mk3(p.cam[9], p.cam[10], p.cam[11]);
    unsigned long long passes = 0, lanes = 0, wsteps = 0;
    unsigned iters = 0;
    float t = 0.0f, tmin, tmax;
    int steps = 0;
    bool hit = false;
    if (active && ray_sphere<false>(origin, dir, 1.15f, tmin, tmax)) {
        t = gmax(0.0f, tmin);
        for (steps = 0; steps < p.max_steps; steps++) {
            wsteps++;
            v3 pos = mk3(origin.x + t * dir.x, origin.y + t * dir.y, origin.z + t * dir.z);
            float dist = de_mandelbulb8_dbg(pos, iters, passes, lanes, p.dbg ? p.dbg + 8 : nullptr);
            t += dist;
            if (t > tmax) break;
            if (dist < 0.001f) { hit = true; break; }
        }
    }
    // wave totals: every lane counted the passes it took part in; the wave-level count is the max
    unsigned long long wp = passes, ws = wsteps;
    for (int o = 32; o > 0; o >>= 1) {
        unsigned long long a = __shfl_xor(wp, o, 64), b = __shfl_xor(ws, o, 64);
        wp = a > wp ? a : wp; ws = b > ws ? b : ws;
    }
    unsigned long long li = iters, lh = hit ? 1 : 0, lst = (unsigned long long)(steps);
    for (int o = 32; o > 0; o >>= 1) { li += __shfl_xor(li, o, 64); lh += __shfl_xor(lh, o, 64); lst += __shfl_xor(lst, o, 64); }
    if (lane == 0 && p.dbg) {
        atomicAdd(&p.dbg[0], wp);       // wave-level inner passes (max over lanes is a lower bound of the true count)
        atomicAdd(&p.dbg[1], li);       // lane iterations
        atomicAdd(&p.dbg[2], ws);       // wave-level march steps
        atomicAdd(&p.dbg[3], lst);      // lane steps (sum of loop counters)
        atomicAdd(&p.dbg[4], lh);
        atomicAdd(&p.dbg[5], 1ull);
    }
}

hipError_t launch_march_stats(const FrameParams &p, hipStream_t stream)
{
    dim3 grid((p.w + 31) / 32, (p.h + 7) / 8), block(256);
    hipLaunchKernelGGL(k_march_stats, grid, block, 0, stream, p);
    return hipGetLastError();
}

}  // namespace rmdf
