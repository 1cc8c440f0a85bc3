// surface_k.hip -- what would another lane schedule of the ON-SURFACE estimates (normal: 4, distance AO: 2 per hit pixel) save?
// Measurement tool.  A kernel renders the headline frame's march (1920x1080, in_time 0, 256 steps) with the product's arithmetic
// (rmdf_device.hpp) and records, per hit pixel, the escape-iteration count k of each of its six on-surface estimates.  The host
// replays lane schedules over those counts with the cost model  wave-pass-set = A * max_k(lanes in it) + B  (A = instructions of one
// iteration pass, B = per-estimate tail) for every 8x8 packet (= one wave of k_render):
//   current      : six estimates one after the other, each for all hit lanes of the packet
//   sorted-3     : estimate 0 as now; its k is known then, and estimates 1..3 sit within 1e-5 of the same point, so the 3*n tasks are
//                  sorted by k0 and run 64 at a time (cross-lane moves: 3 in + 1 out per task set)
//   + ao sorted  : the two AO estimates likewise, 2*n tasks sorted by k0 (a predictor only: they sit 0.016 / 0.081 off the surface)
//   ideal        : every lane busy in every pass
// Build: make -C tools/ubench surface_k ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
#include "rmdf_device.hpp"
using namespace rmdf;

#define W 1920
#define H 1080
#define MS 256

struct Cam { float c[12]; float fov_xs; };

__global__ void k_surface(Cam cam, unsigned char *ks /* 8 per pixel: hit, k0..k5, 0 */)
{
    const int px = blockIdx.x * blockDim.x + threadIdx.x, py = blockIdx.y;
    if (px >= W) return;
    const float ndcx = ((float)px + 0.5f) / (float)W * 2.0f - 1.0f;
    const float ndcy = ((float)py + 0.5f) / (float)H * 2.0f - 1.0f;
    const float aspect = (float)W / (float)H;
    const v3 d = normalize3(mk3(ndcx * cam.fov_xs, ndcy * cam.fov_xs / aspect, -1.0f));
    const float *c = cam.c;
    const v3 dir = mk3(c[0] * d.x + c[3] * d.y + c[6] * d.z, c[1] * d.x + c[4] * d.y + c[7] * d.z, c[2] * d.x + c[5] * d.y + c[8] * d.z);
    const v3 origin = mk3(c[9], c[10], c[11]);
    float tmin, tmax;
    unsigned char out[8] = { 0, 0, 0, 0, 0, 0, 0, 0 };
    if (ray_sphere(origin, dir, 1.15f, tmin, tmax)) {
        float t = gmax(0.0f, tmin);
        bool hit = false;
        for (int s = 0; s < MS; s++) {
            unsigned it = 0;
            const float dist = de_mandelbulb8(mk3(origin.x + t * dir.x, origin.y + t * dir.y, origin.z + t * dir.z), it);
            t += dist;
            if (t > tmax) break;
            if (dist < 0.001f) { hit = true; break; }
        }
        if (hit) {
            const v3 isec = mk3(origin.x + dir.x * t, origin.y + dir.y * t, origin.z + dir.z * t);
            const v3 np = mk3(isec.x - dir.x * 0.00001f, isec.y - dir.y * 0.00001f, isec.z - dir.z * 0.00001f);
            const float eps = 0.00001f;
            unsigned k0 = 0, k1 = 0, k2 = 0, k3 = 0, k4 = 0, k5 = 0;
            const float d0 = de_mandelbulb8(np, k0);
            const float dx = de_mandelbulb8(mk3(np.x - eps, np.y, np.z), k1);
            const float dy = de_mandelbulb8(mk3(np.x, np.y - eps, np.z), k2);
            const float dz = de_mandelbulb8(mk3(np.x, np.y, np.z - eps), k3);
            const v3 n = normalize3(mk3(d0 - dx, d0 - dy, d0 - dz));
            (void)de_mandelbulb8(mk3(isec.x + n.x * 0.016f, isec.y + n.y * 0.016f, isec.z + n.z * 0.016f), k4);
            (void)de_mandelbulb8(mk3(isec.x + n.x * 0.081f, isec.y + n.y * 0.081f, isec.z + n.z * 0.081f), k5);
            out[0] = 1; out[1] = (unsigned char)k0; out[2] = (unsigned char)k1; out[3] = (unsigned char)k2; out[4] = (unsigned char)k3;
            out[5] = (unsigned char)k4; out[6] = (unsigned char)k5;
        }
    }
    for (int i = 0; i < 8; i++) ks[((size_t)py * W + px) * 8 + i] = out[i];
}

static void host_camera(float cam[12], float *fov_xs, float time = 0.0f)
{
    float cx = sinf(time / 3.0f), cy = cosf(time / 4.0f), cz = cosf(time / 3.0f);
    float s = 1.0f / sqrtf((cx * cx + cy * cy) + cz * cz);
    cx = cx * s * 2.414213562373095f; cy = cy * s * 2.414213562373095f; cz = cz * s * 2.414213562373095f;
    float zl = 1.0f / sqrtf((cx * cx + cy * cy) + cz * cz);
    float zx = cx * zl, zy = cy * zl, zz = cz * zl;
    float xx = 1.0f * zz - 0.0f * zy, xy = 0.0f * zx - 0.0f * zz, xz = 0.0f * zy - 1.0f * zx;
    float xl = 1.0f / sqrtf((xx * xx + xy * xy) + xz * xz);
    xx *= xl; xy *= xl; xz *= xl;
    float yx = zy * xz - zz * xy, yy = zz * xx - zx * xz, yz = zx * xy - zy * xx;
    float v[12] = { xx, xy, xz, yx, yy, yz, zx, zy, zz, cx, cy, cz };
    for (int i = 0; i < 12; i++) cam[i] = v[i];
    *fov_xs = tanf(((45.0f * 1.5f) * 0.017453292519943295f) / 2.0f);
}

int main(int argc, char **argv)
{
    const double A = argc > 1 ? atof(argv[1]) : 91.0, B = argc > 2 ? atof(argv[2]) : 85.0, SORT = argc > 3 ? atof(argv[3]) : 120.0, MOVE = 8.0;
    Cam cam;
    host_camera(cam.c, &cam.fov_xs);
    const size_t npx = (size_t)W * H;
    unsigned char *d_ks;
    if (hipMalloc((void **)&d_ks, npx * 8) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipLaunchKernelGGL(k_surface, dim3((W + 63) / 64, H), dim3(64), 0, 0, cam, d_ks);
    std::vector<unsigned char> ks(npx * 8);
    if (hipMemcpy(ks.data(), d_ks, npx * 8, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }

    double hits = 0, lane_iters[6] = { 0 }, diff_taps = 0;
    long hist0[32] = { 0 }, hist4[32] = { 0 }, hist5[32] = { 0 };
    for (size_t i = 0; i < npx; i++) {
        const unsigned char *p = &ks[i * 8];
        if (!p[0]) continue;
        hits++;
        for (int t = 0; t < 6; t++) lane_iters[t] += p[1 + t];
        if (p[2] != p[1] || p[3] != p[1] || p[4] != p[1]) diff_taps++;
        hist0[p[1] > 31 ? 31 : p[1]]++; hist4[p[5] > 31 ? 31 : p[5]]++; hist5[p[6] > 31 ? 31 : p[6]]++;
    }
    printf("hit pixels %.0f; mean k of the six on-surface estimates: %.2f %.2f %.2f %.2f | %.2f %.2f; pixels whose normal estimates differ in k: %.2f %%\n",
           hits, lane_iters[0] / hits, lane_iters[1] / hits, lane_iters[2] / hits, lane_iters[3] / hits, lane_iters[4] / hits, lane_iters[5] / hits,
           100.0 * diff_taps / hits);
    printf("histogram of k0 (base point):"); for (int k = 0; k < 27; k++) printf(" %d:%.1f%%", k, 100.0 * hist0[k] / hits); printf("\n");
    printf("histogram of k4 (AO 0.016)  :"); for (int k = 0; k < 27; k++) printf(" %d:%.1f%%", k, 100.0 * hist4[k] / hits); printf("\n");
    printf("histogram of k5 (AO 0.081)  :"); for (int k = 0; k < 27; k++) printf(" %d:%.1f%%", k, 100.0 * hist5[k] / hits); printf("\n");

    double cur = 0, sorted3 = 0, sorted3ao = 0, ideal = 0, cur_n = 0, cur_ao = 0, s3_n = 0, s_ao = 0, wg_sorted = 0;
    long packets = 0, full = 0;
    // workgroup-level variant: the four packets of a 32x8 strip pool their tasks (sorted by k0) -- through LDS
    for (int by = 0; by < H / 8; by++) {
        for (int sx = 0; sx < W / 32; sx++) {
            std::vector<int> wg_tasks_n, wg_tasks_ao0, wg_tasks_ao1, wg_k0;   // k values of tasks, paired with predictor k0
            struct T { int k0, k; };
            std::vector<T> wgn, wga;
            double wg_first = 0;
            for (int wv = 0; wv < 4; wv++) {
                const int bx = sx * 4 + wv;
                std::vector<const unsigned char *> px;
                for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
                    const unsigned char *p = &ks[((size_t)(by * 8 + y) * W + bx * 8 + x) * 8];
                    if (p[0]) px.push_back(p);
                }
                const int n = (int)px.size();
                if (!n) continue;
                packets++; if (n == 64) full++;
                int mx[6] = { 0 };
                for (auto p : px) for (int t = 0; t < 6; t++) { if (p[1 + t] > mx[t]) mx[t] = p[1 + t]; ideal += (A * p[1 + t] + B) / 64.0; }
                for (int t = 0; t < 6; t++) cur += A * mx[t] + B;
                for (int t = 0; t < 4; t++) cur_n += A * mx[t] + B;
                for (int t = 4; t < 6; t++) cur_ao += A * mx[t] + B;
                // sorted-3: estimate 0 for all, then 3n tasks sorted by k0 in sets of 64
                std::vector<T> tasks;
                for (auto p : px) for (int t = 1; t < 4; t++) tasks.push_back({ p[1], p[1 + t] });
                std::sort(tasks.begin(), tasks.end(), [](const T &a, const T &b) { return a.k0 > b.k0; });
                double c3 = A * mx[0] + B + SORT;
                for (size_t o = 0; o < tasks.size(); o += 64) {
                    int m = 0; for (size_t j = o; j < std::min(tasks.size(), o + 64); j++) m = std::max(m, tasks[j].k);
                    c3 += A * m + B + MOVE;
                }
                s3_n += c3;
                std::vector<T> ta;
                for (auto p : px) for (int t = 4; t < 6; t++) ta.push_back({ p[1] * 2 + (t == 4), p[1 + t] });   // nearer tap first within equal k0
                std::sort(ta.begin(), ta.end(), [](const T &a, const T &b) { return a.k0 > b.k0; });
                double ca = 0;
                for (size_t o = 0; o < ta.size(); o += 64) {
                    int m = 0; for (size_t j = o; j < std::min(ta.size(), o + 64); j++) m = std::max(m, ta[j].k);
                    ca += A * m + B + MOVE;
                }
                s_ao += ca;
                // workgroup pooling
                wg_first += A * mx[0] + B;
                for (auto p : px) { for (int t = 1; t < 4; t++) wgn.push_back({ p[1], p[1 + t] }); for (int t = 4; t < 6; t++) wga.push_back({ p[1] * 2 + (t == 4), p[1 + t] }); }
            }
            if (!wgn.empty()) {
                std::sort(wgn.begin(), wgn.end(), [](const T &a, const T &b) { return a.k0 > b.k0; });
                std::sort(wga.begin(), wga.end(), [](const T &a, const T &b) { return a.k0 > b.k0; });
                double c = wg_first + 4 * SORT;
                for (size_t o = 0; o < wgn.size(); o += 64) { int m = 0; for (size_t j = o; j < std::min(wgn.size(), o + 64); j++) m = std::max(m, wgn[j].k); c += A * m + B + 2 * MOVE; }
                for (size_t o = 0; o < wga.size(); o += 64) { int m = 0; for (size_t j = o; j < std::min(wga.size(), o + 64); j++) m = std::max(m, wga[j].k); c += A * m + B + 2 * MOVE; }
                wg_sorted += c;
            }
        }
    }
    sorted3 = s3_n + cur_ao; sorted3ao = s3_n + s_ao;
    printf("packets with hits %ld (all 64 lanes hit: %ld)\n", packets, full);
    printf("wave-instructions of the on-surface estimates (A = %.0f, B = %.0f, sort %.0f, moves %.0f per task set), in M:\n", A, B, SORT, MOVE);
    printf("  current                        %8.2f   (normal %.2f, AO %.2f)\n", cur / 1e6, cur_n / 1e6, cur_ao / 1e6);
    printf("  sorted-3 (normal)              %8.2f   (normal %.2f)\n", sorted3 / 1e6, s3_n / 1e6);
    printf("  sorted-3 + AO tasks sorted     %8.2f   (AO %.2f)\n", sorted3ao / 1e6, s_ao / 1e6);
    printf("  workgroup-pooled sorted tasks  %8.2f\n", wg_sorted / 1e6);
    printf("  ideal                          %8.2f\n", ideal / 1e6);
    // straggler split of the two AO estimates: every lane iterates at most CUT passes in place; estimates still iterating then are
    // set aside (9 dwords of state) and finished together, 64 at a time -- per packet (both taps' stragglers of one wave) or per
    // workgroup (the four packets of a 32x8 strip through LDS).  XCH = instructions a wave spends on setting aside / taking back.
    for (int cut : { 4, 5, 6, 8, 10, 12 }) {
        const double XCH = 60.0;
        double v1 = 0, v2 = 0; long sets1 = 0, sets2 = 0, strag = 0;
        for (int by = 0; by < H / 8; by++) for (int sx = 0; sx < W / 32; sx++) {
            std::vector<int> wg;
            int wg_waves = 0;
            for (int wv = 0; wv < 4; wv++) {
                const int bx = sx * 4 + wv;
                std::vector<int> st; int n = 0, m4 = 0, m5 = 0;
                for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) {
                    const unsigned char *p = &ks[((size_t)(by * 8 + y) * W + bx * 8 + x) * 8];
                    if (!p[0]) continue;
                    n++;
                    m4 = std::max(m4, std::min((int)p[5], cut)); m5 = std::max(m5, std::min((int)p[6], cut));
                    if (p[5] > cut) st.push_back(p[5] - cut);
                    if (p[6] > cut) st.push_back(p[6] - cut);
                }
                if (!n) continue;
                wg_waves++;
                const double base = A * (m4 + m5) + 2 * B;
                v1 += base; v2 += base;
                strag += (long)st.size();
                if (!st.empty()) {
                    v1 += XCH;
                    std::sort(st.begin(), st.end(), std::greater<int>());
                    for (size_t o = 0; o < st.size(); o += 64) { v1 += A * st[o] + 20; sets1++; }
                }
                wg.insert(wg.end(), st.begin(), st.end());
            }
            if (!wg.empty()) {
                v2 += XCH * wg_waves;
                std::sort(wg.begin(), wg.end(), std::greater<int>());
                for (size_t o = 0; o < wg.size(); o += 64) { v2 += A * wg[o] + 20; sets2++; }
            }
        }
        printf("  AO with stragglers set aside after %2d passes: per packet %7.2f M (%ld sets), per workgroup %7.2f M (%ld sets); %ld stragglers of %.0f estimates\n",
               cut, v1 / 1e6, sets1, v2 / 1e6, sets2, strag, 2 * hits);
    }
    return 0;
}
