// sched_sim.hip -- how much VALU work could another lane schedule of the power-8 march save?
// Measurement tool.  A kernel marches every ray of the headline frame (1920x1080, in_time 0, 256 steps) with the
// product's arithmetic (rmdf_device.hpp) and records the escape-iteration count k of every distance estimate
// (trace[step][pixel], uint8).  The host then replays lane schedules over that trace with the cost model
//     wave-step = A * max_k(active lanes) + B          (A = instructions of one iteration pass, B = per-estimate tail)
// and prints the wave-instruction totals of the march part:
//   nested      : 8x8 packet per wave, runs until its last ray ends (k_render<.., MERGE=false>)
//   wg-pool T   : + the <= T last rays of each of the 4 packets of a 32x8 strip move to one host wave (MERGE=true)
//   global T    : a wave with <= T rays left hands them to a global pool; pooled rays are marched 64 at a time in
//                 hand-over order, re-pooled at <= T again (what a cross-workgroup queue could reach at best)
//   ideal       : every lane busy in every pass (sum of lane work / 64)
//   flat T      : lanes iterate independently; escaped lanes wait for a tail pass run when >= T wait (no packets' pooling)
// Build: make -C tools/ubench sched_sim ; run on an MI355X.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#include "rmdf_device.hpp"
using namespace rmdf;

#define W 1920
#define H 1080
#define MS 256

struct Cam { float c[12]; float fov_xs; };

__global__ void k_trace(Cam cam, unsigned char *trace, unsigned short *nsteps, unsigned char *k_normal)
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
    int n = 0;
    unsigned char kn = 0;
    if (ray_sphere(origin, dir, 1.15f, tmin, tmax)) {
        float t = gmax(0.0f, tmin);
        for (int s = 0; s < MS; s++) {
            unsigned it = 0;
            const float dist = de_mandelbulb8(mk3(origin.x + t * dir.x, origin.y + t * dir.y, origin.z + t * dir.z), it);
            trace[(size_t)s * W * H + (size_t)py * W + px] = (unsigned char)it;
            n = s + 1;
            t += dist;
            if (t > tmax) break;
            if (dist < 0.001f) {
                // hit: k of the distance estimate at the finite-difference base point (the normal's four estimates share it)
                unsigned it2 = 0;
                const v3 isec = mk3(origin.x + dir.x * t, origin.y + dir.y * t, origin.z + dir.z * t);
                (void)de_mandelbulb8(mk3(isec.x - dir.x * 0.00001f, isec.y - dir.y * 0.00001f, isec.z - dir.z * 0.00001f), it2);
                kn = (unsigned char)(it2 + 1);                  // +1: 0 means "no hit"
                break;
            }
        }
    }
    k_normal[(size_t)py * W + px] = kn;
    nsteps[(size_t)py * W + px] = (unsigned short)n;     // distance estimates taken (= march passes of this ray)
}

static void host_camera(float cam[12], float *fov_xs, float time = 0.0f)
{
    // fragment.shd:892-897, 829-838 at in_time = 0 (same operation order as rmdf_api.cpp: host_camera)
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

struct Ray { int pix; int step; };                 // a ray in flight: its pixel and the index of its next estimate

static const unsigned char *g_trace;
static const unsigned short *g_n;
static inline int kof(int pix, int step) { return g_trace[(size_t)step * W * H + pix]; }

// march `rays` (<= 64) in one wave until at most T of them are left (T = 0: to the end); returns the cost, leaves the
// survivors in `rays`
static double run_wave(std::vector<Ray> &rays, int T, double A, double B, double *lane_work)
{
    double cost = 0.0;
    for (;;) {
        int act = 0, mk = 0;
        for (auto &r : rays) if (r.step < g_n[r.pix]) { act++; int k = kof(r.pix, r.step); if (k > mk) mk = k; *lane_work += A * k + B; }
        if (act == 0) break;
        cost += A * mk + B;
        for (auto &r : rays) if (r.step < g_n[r.pix]) r.step++;
        int left = 0;
        for (auto &r : rays) if (r.step < g_n[r.pix]) left++;
        if (left <= T) break;
    }
    std::vector<Ray> s;
    for (auto &r : rays) if (r.step < g_n[r.pix]) s.push_back(r);
    rays.swap(s);
    return cost;
}


// Round 4: lockstep regrouping inside a workgroup.  The NW packets of a strip share a ray pool in LDS; at every march-step boundary
// the live rays are (optionally) sorted by a key and dealt 64 at a time to ceil(live / 64) waves, so dead lanes vanish and -- with a
// key that predicts the next estimate's escape iteration -- a wave's lanes need similar k.  Cost per wave-step: A * max k + B + Q
// (Q = what writing and reading a ray's state through LDS and the sort cost per wave-step).  key: 0 = pool order (compaction only),
// 1 = k of the ray's previous estimate (what a kernel knows), 2 = k of the estimate to come (oracle: the bound of any predictor),
// 3 = previous k, but only rays are exchanged when the step index is a multiple of `every` (regroup every few steps)
static void regroup_sims(const std::vector<unsigned short> &n)
{
    const double A = 87.0, B = 100.0;
    const int PX = (W + 7) / 8, PY = (H + 7) / 8;
    auto packet = [&](int bx, int by, std::vector<Ray> &rays) {
        for (int ly = 0; ly < 8; ly++) for (int lx = 0; lx < 8; lx++) {
            const int x = bx * 8 + lx, y = by * 8 + ly;
            if (x < W && y < H && n[(size_t)y * W + x] > 0) rays.push_back(Ray{ y * W + x, 0 });
        }
    };
    // reference points with the same A, B: nested packets, and ideal
    double lane_work = 0.0, c_nested = 0.0;
    { std::vector<Ray> r; for (int by = 0; by < PY; by++) for (int bx = 0; bx < PX; bx++) { r.clear(); packet(bx, by, r); c_nested += run_wave(r, 0, A, B, &lane_work); } }
    const double ideal = lane_work / 64.0;
    printf("A = %.0f, B = %.0f: ideal %.1f M, nested %.1f M (lane utilisation %.3f)\n", A, B, ideal / 1e6, c_nested / 1e6, ideal / c_nested);
    // idle lane-slot causes of the nested schedule: (a) lanes whose ray has ended (or never started) while the packet marches on,
    // (b) live lanes waiting for the packet's largest k inside an estimate
    {
        double slots = 0, dead = 0, kwait = 0, useful = 0;
        std::vector<Ray> r;
        for (int by = 0; by < PY; by++) for (int bx = 0; bx < PX; bx++) {
            r.clear(); packet(bx, by, r);
            for (;;) {
                int act = 0, mk = 0, sumk = 0;
                for (auto &x : r) if (x.step < g_n[x.pix]) { act++; const int k = kof(x.pix, x.step); sumk += k; if (k > mk) mk = k; }
                if (!act) break;
                slots += 64.0 * (A * mk + B);
                useful += A * sumk + B * act;
                kwait += A * ((double)mk * act - sumk);
                dead += (64.0 - act) * (A * mk + B);
                for (auto &x : r) if (x.step < g_n[x.pix]) x.step++;
            }
        }
        printf("nested packets, lane-slots of the march: useful %.3f, live lanes waiting for the packet's largest k %.3f, lanes whose ray has ended %.3f\n",
               useful / slots, kwait / slots, dead / slots);
    }
    for (int NW : { 4, 8 }) for (int key : { 0, 1, 2 }) for (double Q : { 0.0, 40.0 }) {
        double c = 0.0, steps = 0.0;
        for (int by = 0; by < PY; by++) for (int sx = 0; sx < (PX + NW - 1) / NW; sx++) {
            std::vector<Ray> pool;
            for (int q = 0; q < NW; q++) { const int bx = sx * NW + q; if (bx >= PX) break; packet(bx, by, pool); }
            std::vector<int> prevk(pool.size(), 0), idx;
            for (;;) {
                idx.clear();
                for (size_t i = 0; i < pool.size(); i++) if (pool[i].step < g_n[pool[i].pix]) idx.push_back((int)i);
                if (idx.empty()) break;
                if (key == 1) std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return prevk[a] < prevk[b]; });
                if (key == 2) std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return kof(pool[a].pix, pool[a].step) < kof(pool[b].pix, pool[b].step); });
                for (size_t j = 0; j < idx.size(); j += 64) {
                    int mk = 0;
                    for (size_t jj = j; jj < std::min(idx.size(), j + 64); jj++) { const int k = kof(pool[idx[jj]].pix, pool[idx[jj]].step); if (k > mk) mk = k; }
                    c += A * mk + B + Q; steps += 1.0;
                }
                for (int i : idx) { prevk[i] = kof(pool[i].pix, pool[i].step); pool[i].step++; }
            }
        }
        printf("lockstep regroup, %d waves per workgroup, key %-22s Q = %2.0f: %8.1f M wave-instr (x%.3f of nested) lane utilisation %.3f, %.2f M wave-steps\n",
               NW, key == 0 ? "none (compaction)" : key == 1 ? "previous k" : "next k (oracle)", Q, c / 1e6, c / c_nested, ideal / c, steps / 1e6);
    }
    // the product's schedule with the same A, B for comparison: event-driven host pooling, T = 32
    {
        const int T = 32;
        double c = 0.0, by_live[5] = { 0, 0, 0, 0, 0 }, longest = 0.0, quad_steps = 0, quad_iter_cost = 0, quad_cost = 0, quad_entries = 0, quad_cost_any = 0;
        for (int by = 0; by < PY; by++) for (int sx = 0; sx < (PX + 3) / 4; sx++) {
            bool in_quad[1] = { false };
            std::vector<Ray> wv[4], mail;
            double clk[4] = { 0, 0, 0, 0 };
            bool alive[4] = { false, false, false, false };
            int host = -1, nalive = 0;
            for (int q = 0; q < 4; q++) { const int bx = sx * 4 + q; if (bx >= PX) break; packet(bx, by, wv[q]); alive[q] = !wv[q].empty(); nalive += alive[q]; }
            while (nalive > 0) {
                int w = -1;
                for (int q = 0; q < 4; q++) if (alive[q] && (w < 0 || clk[q] < clk[w])) w = q;
                std::vector<Ray> &r = wv[w];
                if ((int)r.size() <= T && host < 0) host = w;
                if (w == host) { while (r.size() < 64 && !mail.empty()) { r.push_back(mail.back()); mail.pop_back(); } }
                else if ((int)r.size() <= T && host >= 0) { mail.insert(mail.end(), r.begin(), r.end()); r.clear(); }
                if (r.empty()) {
                    if (w == host && nalive > 1) { double nxt = 1e300; for (int q = 0; q < 4; q++) if (alive[q] && q != w && clk[q] < nxt) nxt = clk[q]; clk[w] = nxt + 1e-9; continue; }
                    alive[w] = false; nalive--; continue;
                }
                int mk = 0;
                for (auto &x : r) { const int k = kof(x.pix, x.step); if (k > mk) mk = k; }
                const double cost = A * mk + B;
                c += cost; clk[w] += cost;
                const int live = (int)r.size();
                by_live[live <= 4 ? 0 : live <= 8 ? 1 : live <= 16 ? 2 : live <= 32 ? 3 : 4] += cost;
                // lane groups as they could be built: only the host, only once no mail can arrive any more (it is the last wave marching)
                if (w == host && nalive == 1 && mail.empty() && live <= 16) { quad_steps += 1.0; quad_iter_cost += A * mk; quad_cost += cost; if (!in_quad[0]) { in_quad[0] = true; quad_entries += 1.0; } }
                if (w == host && live <= 16) quad_cost_any += cost;
                std::vector<Ray> nr;
                for (auto &x : r) { x.step++; if (x.step < g_n[x.pix]) nr.push_back(x); }
                r.swap(nr);
            }
            for (int q = 0; q < 4; q++) if (clk[q] > longest) longest = clk[q];
        }
        printf("product schedule (host pooling, T = 32) %8.1f M wave-instr (x%.3f of nested) lane utilisation %.3f\n", c / 1e6, c / c_nested, ideal / c);
        printf("  of which wave-steps with <= 4 / 5-8 / 9-16 / 17-32 / > 32 live rays: %.1f / %.1f / %.1f / %.1f / %.1f M; longest wave: %.0f instructions\n",
               by_live[0] / 1e6, by_live[1] / 1e6, by_live[2] / 1e6, by_live[3] / 1e6, by_live[4] / 1e6, longest);
        printf("  host alone with <= 16 rays and no mail to come: %.1f M wave-instr in %.2f M wave-steps (%.1f M of it iteration passes), %.0f workgroups enter; "
               "any host step with <= 16 rays: %.1f M\n", quad_cost / 1e6, quad_steps / 1e6, quad_iter_cost / 1e6, quad_entries, quad_cost_any / 1e6);
        for (double aq : { 40.0, 45.0, 50.0 }) {
            const double saved = quad_iter_cost * (1.0 - aq / A) - 150.0 * quad_entries;
            printf("  4 lanes per ray there, %.0f instructions per pass, 150 per switch: %8.1f M wave-instr (x%.3f of the product schedule)\n", aq, (c - saved) / 1e6, (c - saved) / c);
        }
        // the same schedule with G lanes per ray once a wave is down to 64 / G rays (the lanes of a group evaluate one ray's iteration
        // together: a pass costs A / speedup(G), the tail B / speedup_tail(G))
        for (double sp4 : { 2.0, 2.5 }) for (double sp2 : { 1.0, 1.5 }) {
            const double tot = by_live[0] / sp4 + by_live[1] / sp4 + by_live[2] / sp4 + by_live[3] / sp2 + by_live[4];
            printf("  lane groups: <= 16 live rays x%.1f faster, 17-32 x%.1f: %8.1f M wave-instr (x%.3f of the product schedule)\n", sp4, sp2, tot / 1e6, tot / c);
        }
    }
}

// Round 4: hand-over ACROSS workgroups without a G-buffer: a host wave that is alone in its workgroup with <= T2 rays (no mail to
// come) either joins an ACCEPTOR -- another such host anywhere on the chip with enough idle lanes: it hands its rays over, WAITS for
// their results and then shades as before -- or, if nobody accepts, becomes an acceptor itself.  Acceptors never wait for anybody.
// Replay: (1) every workgroup alone (product schedule) -> its duration, the clock at which its host is alone with <= T2 rays, and
// those rays; (2) workgroups dispatched longest-first over `slots` workgroup slots, each running at speed 1 (instructions = time);
// (3) the tails replayed in global time order with the pairing rule.  Reported: instructions of the march part with and without.
static void cross_wg_sim(const std::vector<unsigned short> &n)
{
    const double A = 87.0, B = 100.0;
    const int PX = (W + 7) / 8, PY = (H + 7) / 8, T = 32;
    auto packet = [&](int bx, int by, std::vector<Ray> &rays) {
        for (int ly = 0; ly < 8; ly++) for (int lx = 0; lx < 8; lx++) {
            const int x = bx * 8 + lx, y = by * 8 + ly;
            if (x < W && y < H && n[(size_t)y * W + x] > 0) rays.push_back(Ray{ y * W + x, 0 });
        }
    };
    struct WG { double dur = 0, head = 0, tail_start = 0; std::vector<Ray> tail; double start = 0; };
    for (int T2 : { 8, 16, 24, 32 }) {
        std::vector<WG> wgs;
        double total = 0;
        for (int by = 0; by < PY; by++) for (int sx = 0; sx < (PX + 3) / 4; sx++) {
            WG g;
            std::vector<Ray> wv[4], mail;
            double clk[4] = { 0, 0, 0, 0 }, cost_sum = 0;
            bool alive[4] = { false, false, false, false };
            int host = -1, nalive = 0;
            bool in_tail = false;
            for (int q = 0; q < 4; q++) { const int bx = sx * 4 + q; if (bx >= PX) break; packet(bx, by, wv[q]); alive[q] = !wv[q].empty(); nalive += alive[q]; }
            while (nalive > 0) {
                int w = -1;
                for (int q = 0; q < 4; q++) if (alive[q] && (w < 0 || clk[q] < clk[w])) w = q;
                std::vector<Ray> &r = wv[w];
                if ((int)r.size() <= T && host < 0) host = w;
                if (w == host) { while (r.size() < 64 && !mail.empty()) { r.push_back(mail.back()); mail.pop_back(); } }
                else if ((int)r.size() <= T && host >= 0) { mail.insert(mail.end(), r.begin(), r.end()); r.clear(); }
                if (r.empty()) {
                    if (w == host && nalive > 1) { double nxt = 1e300; for (int q = 0; q < 4; q++) if (alive[q] && q != w && clk[q] < nxt) nxt = clk[q]; clk[w] = nxt + 1e-9; continue; }
                    alive[w] = false; nalive--; continue;
                }
                if (w == host && nalive == 1 && mail.empty() && (int)r.size() <= T2) { in_tail = true; g.tail_start = clk[w]; g.tail = r; break; }
                int mk = 0;
                for (auto &x : r) { const int k = kof(x.pix, x.step); if (k > mk) mk = k; }
                const double cost = A * mk + B;
                cost_sum += cost; clk[w] += cost;
                std::vector<Ray> nr;
                for (auto &x : r) { x.step++; if (x.step < g_n[x.pix]) nr.push_back(x); }
                r.swap(nr);
            }
            g.head = cost_sum;
            double mx = 0; for (int q = 0; q < 4; q++) if (clk[q] > mx) mx = clk[q];
            if (!in_tail) g.tail_start = mx;
            // the tail marched alone (product schedule)
            double tc = 0; { std::vector<Ray> t = g.tail; tc = run_wave(t, 0, A, B, &total); }
            g.dur = g.tail_start + tc;
            total = 0;
            wgs.push_back(g);
            (void)cost_sum;
        }
        // (2) longest-first dispatch over the workgroup slots
        const int slots = 2048;
        std::vector<size_t> order(wgs.size());
        for (size_t i = 0; i < order.size(); i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return wgs[a].dur > wgs[b].dur; });
        std::vector<double> freeat(slots, 0.0);
        for (size_t i : order) { auto it = std::min_element(freeat.begin(), freeat.end()); wgs[i].start = *it; *it += wgs[i].dur; }
        // (3) tails in global time order
        struct Acc { std::vector<Ray> rays; double clk; bool open; };
        std::vector<size_t> tl;
        for (size_t i = 0; i < wgs.size(); i++) if (!wgs[i].tail.empty()) tl.push_back(i);
        std::sort(tl.begin(), tl.end(), [&](size_t a, size_t b) { return wgs[a].start + wgs[a].tail_start < wgs[b].start + wgs[b].tail_start; });
        std::vector<Acc> accs;
        double base = 0, alone_tails = 0, merged_tails = 0; size_t donors = 0;
        for (auto &g : wgs) base += g.head;
        // advance an acceptor to time t (marching its rays), return cost spent
        auto advance = [&](Acc &a, double t) {
            double c = 0;
            while (a.clk < t && !a.rays.empty()) {
                int mk = 0;
                for (auto &x : a.rays) { const int k = kof(x.pix, x.step); if (k > mk) mk = k; }
                const double cost = A * mk + B;
                c += cost; a.clk += cost;
                std::vector<Ray> nr;
                for (auto &x : a.rays) { x.step++; if (x.step < g_n[x.pix]) nr.push_back(x); }
                a.rays.swap(nr);
            }
            if (a.rays.empty()) a.open = false;
            return c;
        };
        for (size_t i : tl) {
            const double t = wgs[i].start + wgs[i].tail_start;
            { double d = 0; std::vector<Ray> tt = wgs[i].tail; alone_tails += run_wave(tt, 0, A, B, &d); }
            // bring the open acceptors up to now; pick the one with the most rays that still has room
            Acc *best = nullptr;
            for (auto &a : accs) if (a.open) { merged_tails += advance(a, t); if (a.open && a.rays.size() + wgs[i].tail.size() <= 64 && (!best || a.rays.size() > best->rays.size())) best = &a; }
            if (best) { best->rays.insert(best->rays.end(), wgs[i].tail.begin(), wgs[i].tail.end()); merged_tails += 150.0; donors++; }
            else { Acc a; a.rays = wgs[i].tail; a.clk = t; a.open = true; accs.push_back(a); }
            // forget closed acceptors now and then
            if (accs.size() > 4096) { std::vector<Acc> keep; for (auto &a : accs) if (a.open) keep.push_back(a); accs.swap(keep); }
        }
        for (auto &a : accs) if (a.open) merged_tails += advance(a, 1e300);
        printf("cross-workgroup hand-over with waiting donors, T2 = %2d: %zu of %zu workgroups have such a tail, %zu donate; march %.1f M alone -> %.1f M (x%.3f)\n",
               T2, tl.size(), wgs.size(), donors, (base + alone_tails) / 1e6, (base + merged_tails) / 1e6, (base + merged_tails) / (base + alone_tails));
    }
}

int main(int argc, char **argv)
{
    Cam cam;
    host_camera(cam.c, &cam.fov_xs);
    const size_t npx = (size_t)W * H;
    unsigned char *d_trace, *d_kn; unsigned short *d_n;
    if (hipMalloc((void **)&d_trace, npx * MS) != hipSuccess || hipMalloc((void **)&d_n, npx * 2) != hipSuccess ||
        hipMalloc((void **)&d_kn, npx) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(d_trace, 0, npx * MS);
    hipLaunchKernelGGL(k_trace, dim3((W + 63) / 64, H), dim3(64), 0, 0, cam, d_trace, d_n, d_kn);
    std::vector<unsigned char> trace(npx * MS);
    std::vector<unsigned short> n(npx);
    if (hipMemcpy(trace.data(), d_trace, npx * MS, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
    hipMemcpy(n.data(), d_n, npx * 2, hipMemcpyDeviceToHost);
    std::vector<unsigned char> kn(npx);
    hipMemcpy(kn.data(), d_kn, npx, hipMemcpyDeviceToHost);
    g_trace = trace.data(); g_n = n.data();
    // step counts of the same scene a moment earlier (what a viewer's previous frame would have measured)
    std::vector<unsigned short> n_prev60(npx), n_prev10(npx);
    {
        Cam cam2;
        host_camera(cam2.c, &cam2.fov_xs, -1.0f / 60.0f);
        hipLaunchKernelGGL(k_trace, dim3((W + 63) / 64, H), dim3(64), 0, 0, cam2, d_trace, d_n, d_kn);
        hipMemcpy(n_prev60.data(), d_n, npx * 2, hipMemcpyDeviceToHost);
        host_camera(cam2.c, &cam2.fov_xs, -0.1f);
        hipLaunchKernelGGL(k_trace, dim3((W + 63) / 64, H), dim3(64), 0, 0, cam2, d_trace, d_n, d_kn);
        hipMemcpy(n_prev10.data(), d_n, npx * 2, hipMemcpyDeviceToHost);
    }
    if (argc > 1 && !strcmp(argv[1], "regroup")) { regroup_sims(n); return 0; }
    if (argc > 1 && !strcmp(argv[1], "crosswg")) { cross_wg_sim(n); return 0; }
    double evals = 0, iters = 0; int maxn = 0;
    for (size_t i = 0; i < npx; i++) { evals += n[i]; if (n[i] > maxn) maxn = n[i]; for (int s = 0; s < n[i]; s++) iters += kof((int)i, s); }
    printf("rays with estimates: march estimates %.4e  escape iterations %.4e  mean k %.2f  longest ray %d estimates\n", evals, iters, iters / evals, maxn);

    const double A = 117.0, B = 110.0;
    auto packet = [&](int bx, int by, std::vector<Ray> &rays) {
        rays.clear();
        for (int ly = 0; ly < 8; ly++) for (int lx = 0; lx < 8; lx++) {
            const int x = bx * 8 + lx, y = by * 8 + ly;
            if (x < W && y < H && n[(size_t)y * W + x] > 0) rays.push_back(Ray{ y * W + x, 0 });
        }
    };
    const int PX = (W + 7) / 8, PY = (H + 7) / 8;
    double lane_work = 0.0, dummy = 0.0;
    // nested
    double c_nested = 0.0;
    { std::vector<Ray> r; for (int by = 0; by < PY; by++) for (int bx = 0; bx < PX; bx++) { packet(bx, by, r); c_nested += run_wave(r, 0, A, B, &lane_work); } }
    const double ideal = lane_work / 64.0;
    printf("cost model: wave-step = %.0f * max k + %.0f wave instructions\n", A, B);
    printf("%-28s %10.1f M wave-instr  (x%.3f of nested)\n", "ideal (all lanes busy)", ideal / 1e6, ideal / c_nested);
    printf("%-28s %10.1f M wave-instr  lane utilisation %.3f\n", "nested 8x8 packets", c_nested / 1e6, ideal / c_nested);
    // Upper bound of what setting aside long ESTIMATES of the march could save (what k_render does for the distance-AO estimates since
    // round 2): every lane iterates at most `cut` passes per wave-step in place; the iterations beyond that are assumed to run elsewhere at
    // full lane utilisation, for `xch` instructions per wave-step that has any.  Nested packets, no ray pooling.  Ignores that a ray
    // whose estimate is set aside cannot take its next step until it returns -- a bound, not a schedule.
    {
        const double A2 = 91.0, B2 = 100.0, xch = 40.0;
        std::vector<Ray> r;
        double base = 0.0, bound[5] = { 0, 0, 0, 0, 0 };
        const int cuts[5] = { 3, 4, 5, 6, 8 };
        for (int by = 0; by < PY; by++) for (int bx = 0; bx < PX; bx++) {
            packet(bx, by, r);
            for (;;) {
                int act = 0, mk = 0; int over[5] = { 0, 0, 0, 0, 0 };
                for (auto &x : r) if (x.step < g_n[x.pix]) { act++; const int k = kof(x.pix, x.step); if (k > mk) mk = k; for (int c = 0; c < 5; c++) if (k > cuts[c]) over[c] += k - cuts[c]; }
                if (!act) break;
                base += A2 * mk + B2;
                for (int c = 0; c < 5; c++) bound[c] += A2 * std::min(mk, cuts[c]) + B2 + (over[c] ? xch + A2 * over[c] / 64.0 : 0.0);
                for (auto &x : r) if (x.step < g_n[x.pix]) x.step++;
            }
        }
        printf("march estimates with long ones set aside (bound; A = %.0f, B = %.0f): nested %.1f M;", A2, B2, base / 1e6);
        for (int c = 0; c < 5; c++) printf("  cut %d: %.1f M (x%.3f)", cuts[c], bound[c] / 1e6, bound[c] / base);
        printf("\n");
    }
    for (int T : { 16, 24, 32, 40 }) {
        // workgroup pooling: the four packets of a 32x8 strip; survivors of each (<= T) go to one host wave, which
        // takes them 64 at a time in hand-over order and runs them to the end
        double c = 0.0;
        std::vector<Ray> r, pool;
        for (int by = 0; by < PY; by++) for (int sx = 0; sx < (PX + 3) / 4; sx++) {
            pool.clear();
            for (int q = 0; q < 4; q++) { const int bx = sx * 4 + q; if (bx >= PX) break; packet(bx, by, r); c += run_wave(r, T, A, B, &dummy); pool.insert(pool.end(), r.begin(), r.end()); }
            // host wave: refill idle lanes from the pool at step boundaries
            std::vector<Ray> host;
            size_t next = 0;
            for (;;) {
                while (host.size() < 64 && next < pool.size()) host.push_back(pool[next++]);
                if (host.empty()) break;
                int mk = 0;
                for (auto &x : host) { int k = kof(x.pix, x.step); if (k > mk) mk = k; }
                c += A * mk + B;
                std::vector<Ray> s;
                for (auto &x : host) { x.step++; if (x.step < g_n[x.pix]) s.push_back(x); }
                host.swap(s);
            }
        }
        printf("wg-pool T=%-2d %26.1f M wave-instr  (x%.3f of nested)  lane utilisation %.3f\n", T, c / 1e6, c / c_nested, ideal / c);
    }
    // Generations (what a march split into launches with a global hand-over between them could reach): generation 0 = the product's
    // workgroup pooling (T = 32), but a host wave that has taken all mail and is down to <= T2 rays appends them to a global list and
    // stops; generation g + 1 packs that list 64 rays per wave in list order (4 waves per workgroup, the same pooling) and does the same;
    // the last generation runs to the end.  Q = instructions charged per handed-over ray (queue write + read, ray set-up again), per 64.
    for (int T2 : { 8, 16, 24, 32 }) for (int ngen : { 2, 3 }) {
        const double A2 = 91.0, B2 = 100.0, Q = 120.0;
        const int T = 32;
        double c = 0.0; size_t moved = 0;
        // one workgroup: up to 4 waves' rays; returns the survivors handed to the global list
        auto run_wg = [&](std::vector<Ray> wv[4], int nw, bool last, std::vector<Ray> &out) {
            std::vector<Ray> pool;
            for (int q = 0; q < nw; q++) { c += run_wave(wv[q], T, A2, B2, &dummy); pool.insert(pool.end(), wv[q].begin(), wv[q].end()); }
            std::vector<Ray> host; size_t next = 0;
            for (;;) {
                while (host.size() < 64 && next < pool.size()) host.push_back(pool[next++]);
                if (host.empty()) break;
                if (!last && next >= pool.size() && (int)host.size() <= T2) { out.insert(out.end(), host.begin(), host.end()); break; }
                int mk = 0;
                for (auto &x : host) { int k = kof(x.pix, x.step); if (k > mk) mk = k; }
                c += A2 * mk + B2;
                std::vector<Ray> s2;
                for (auto &x : host) { x.step++; if (x.step < g_n[x.pix]) s2.push_back(x); }
                host.swap(s2);
            }
        };
        std::vector<Ray> glob, nextglob, wv[4];
        for (int by = 0; by < PY; by++) for (int sx = 0; sx < (PX + 3) / 4; sx++) {
            int nw = 0;
            for (int q = 0; q < 4; q++) { const int bx = sx * 4 + q; if (bx >= PX) break; packet(bx, by, wv[q]); nw++; }
            run_wg(wv, nw, ngen == 1, glob);
        }
        for (int g = 1; g < ngen; g++) {
            moved += glob.size(); c += Q * glob.size() / 64.0;
            nextglob.clear();
            for (size_t i = 0; i < glob.size(); i += 256) {
                int nw = 0;
                for (int q = 0; q < 4 && i + 64 * q < glob.size(); q++) { wv[q].assign(glob.begin() + i + 64 * q, glob.begin() + std::min(glob.size(), i + 64 * q + 64)); nw++; }
                run_wg(wv, nw, g == ngen - 1, nextglob);
            }
            glob.swap(nextglob);
        }
        printf("generations: T2=%-2d %d launches %14.1f M wave-instr (A = 91, B = 100; %zu rays handed over)\n", T2, ngen, c / 1e6, moved);
    }
    for (int T : { 16, 24, 32, 40, 48 }) {
        // global pool: generations; a generation's waves are formed of 64 consecutive pooled rays (hand-over order keeps
        // neighbours together), marched until <= T are left, survivors re-pooled; the last generation runs to the end
        double c = 0.0;
        std::vector<Ray> r, pool, nextpool;
        for (int by = 0; by < PY; by++) for (int bx = 0; bx < PX; bx++) { packet(bx, by, r); c += run_wave(r, T, A, B, &dummy); pool.insert(pool.end(), r.begin(), r.end()); }
        int gen = 0;
        const size_t first_pool = pool.size();
        while (!pool.empty()) {
            nextpool.clear();
            const bool last = pool.size() <= 64 * 256 || gen >= 12;       // fewer rays than a quarter of the wave slots: just finish
            for (size_t i = 0; i < pool.size(); i += 64) {
                std::vector<Ray> w(pool.begin() + i, pool.begin() + std::min(pool.size(), i + 64));
                c += run_wave(w, last ? 0 : T, A, B, &dummy);
                nextpool.insert(nextpool.end(), w.begin(), w.end());
            }
            pool.swap(nextpool);
            gen++;
        }
        printf("global T=%-2d %27.1f M wave-instr  (x%.3f of nested)  lane utilisation %.3f  pooled rays %zu, %d generations\n",
               T, c / 1e6, c / c_nested, ideal / c, first_pool, gen);
    }
    // flat state machine: every lane iterates on its own; a lane whose estimate has escaped waits for a tail pass
    // (log, division, next position: B instructions), which the wave runs when >= T lanes wait or nobody iterates.
    // Packets as in `nested` (no pooling): isolates the effect of decoupling the lanes' estimates from each other.
    for (int T : { 1, 8, 16, 24, 32, 48 }) {
        double c = 0.0;
        std::vector<Ray> r;
        for (int by = 0; by < PY; by++) for (int bx = 0; bx < PX; bx++) {
            packet(bx, by, r);
            const int nr = (int)r.size();
            if (!nr) continue;
            std::vector<int> left(nr);                 // iterations left in the current estimate; -1 = waiting for the tail; -2 = done
            for (int i = 0; i < nr; i++) left[i] = kof(r[i].pix, 0);
            for (;;) {
                int iterating = 0, waiting = 0;
                for (int i = 0; i < nr; i++) { if (left[i] > 0) iterating++; else if (left[i] == -1 || left[i] == 0) waiting++; }
                if (!iterating && !waiting) break;
                if (waiting >= T || !iterating) {
                    c += B;
                    for (int i = 0; i < nr; i++) if (left[i] == -1 || left[i] == 0) {
                        r[i].step++;
                        left[i] = r[i].step < g_n[r[i].pix] ? kof(r[i].pix, r[i].step) : -2;
                    }
                } else {
                    c += A;
                    for (int i = 0; i < nr; i++) if (left[i] > 0) { left[i]--; if (left[i] == 0) left[i] = -1; }
                }
            }
        }
        printf("flat, tail when >= %-2d wait %10.1f M wave-instr  (x%.3f of nested)  lane utilisation %.3f\n", T, c / 1e6, c / c_nested, ideal / c);
    }
    // Event-driven replay of the four waves of a workgroup running concurrently (each wave's clock advances by the cost
    // of its wave-steps; the wave with the smallest clock moves next), for two pooling policies at threshold T = 32:
    //   host     : what k_render<.., MERGE> does -- the first wave down to <= T rays becomes the host and keeps marching;
    //              every other wave that gets down to <= T hands its rays to the host's mailbox and leaves; the host adopts
    //              mail into idle lanes at step boundaries
    //   exchange : a wave down to <= T rays adopts posted mail if there is any (as far as idle lanes allow), otherwise it
    //              posts its own rays and leaves; a wave that runs out of rays adopts posted mail too; rays may wait in a
    //              mailbox until the next wave comes by (the last wave standing takes everything)
    for (int policy = 0; policy < 2; policy++) {
        const int T = 32;
        double c = 0.0, wait_cost = 0.0;
        for (int by = 0; by < PY; by++) for (int sx = 0; sx < (PX + 3) / 4; sx++) {
            std::vector<Ray> wv[4], mail;            // rays in flight per wave; posted rays
            double clk[4] = { 0, 0, 0, 0 };
            bool alive[4] = { false, false, false, false };
            int host = -1, nalive = 0;
            for (int q = 0; q < 4; q++) { const int bx = sx * 4 + q; if (bx >= PX) break; packet(bx, by, wv[q]); alive[q] = !wv[q].empty(); nalive += alive[q]; }
            while (nalive > 0) {
                int w = -1;
                for (int q = 0; q < 4; q++) if (alive[q] && (w < 0 || clk[q] < clk[w])) w = q;
                std::vector<Ray> &r = wv[w];
                // step boundary decisions
                if (policy == 0) {
                    if ((int)r.size() <= T && host < 0) host = w;
                    if (w == host) { while (r.size() < 64 && !mail.empty()) { r.push_back(mail.back()); mail.pop_back(); } }
                    else if ((int)r.size() <= T && host >= 0) { mail.insert(mail.end(), r.begin(), r.end()); r.clear(); }
                    if (r.empty()) {
                        if (w == host && nalive > 1) { // host idles until mail arrives: jump its clock to the next wave's
                            double nxt = 1e300; for (int q = 0; q < 4; q++) if (alive[q] && q != w && clk[q] < nxt) nxt = clk[q];
                            clk[w] = nxt + 1e-9; continue;
                        }
                        alive[w] = false; nalive--; continue;
                    }
                } else {
                    if ((int)r.size() <= T) {
                        if (!mail.empty() || nalive == 1) { while (r.size() < 64 && !mail.empty()) { r.push_back(mail.back()); mail.pop_back(); } }
                        else if (nalive > 1) { mail.insert(mail.end(), r.begin(), r.end()); r.clear(); }
                    }
                    if (r.empty()) { alive[w] = false; nalive--; if (nalive == 0 && !mail.empty()) { alive[w] = true; nalive = 1; r.swap(mail); } continue; }
                }
                // one wave-step
                int mk = 0;
                for (auto &x : r) { const int k = kof(x.pix, x.step); if (k > mk) mk = k; }
                const double cost = A * mk + B;
                c += cost; clk[w] += cost;
                std::vector<Ray> nr;
                for (auto &x : r) { x.step++; if (x.step < g_n[x.pix]) nr.push_back(x); }
                r.swap(nr);
            }
            (void)wait_cost;
        }
        printf("event-driven wg-pool T=32, %-8s %10.1f M wave-instr  (x%.3f of nested)  lane utilisation %.3f\n",
               policy == 0 ? "host" : "exchange", c / 1e6, c / c_nested, ideal / c);
    }
    // Event-driven host pooling with larger workgroups and several hosts: NW waves per workgroup (NW/4 strips of 32x8 side
    // by side), up to NH of them may become hosts (first come), all hosts adopt from one shared mailbox pool.
    for (int cfg = 0; cfg < 6; cfg++) {
        const int NWs[6] = { 4, 8, 8, 16, 16, 8 }, NHs[6] = { 1, 1, 2, 2, 4, 2 }, Ts[6] = { 32, 32, 32, 32, 32, 40 };
        const int NW = NWs[cfg], NH = NHs[cfg], T = Ts[cfg];
        double c = 0.0;
        const int SXN = (PX + NW - 1) / NW;
        for (int by = 0; by < PY; by++) for (int sx = 0; sx < SXN; sx++) {
            std::vector<std::vector<Ray>> wv(NW);
            std::vector<Ray> mail;
            std::vector<double> clk(NW, 0.0);
            std::vector<char> alive(NW, 0), is_host(NW, 0);
            int nalive = 0, nhost = 0;
            for (int q = 0; q < NW; q++) { const int bx = sx * NW + q; if (bx >= PX) break; packet(bx, by, wv[q]); alive[q] = !wv[q].empty(); nalive += alive[q]; }
            while (nalive > 0) {
                int w = -1;
                for (int q = 0; q < NW; q++) if (alive[q] && (w < 0 || clk[q] < clk[w])) w = q;
                std::vector<Ray> &r = wv[w];
                if ((int)r.size() <= T && !is_host[w] && nhost < NH) { is_host[w] = 1; nhost++; }
                if (is_host[w]) { while (r.size() < 64 && !mail.empty()) { r.push_back(mail.back()); mail.pop_back(); } }
                else if ((int)r.size() <= T && nhost > 0) { mail.insert(mail.end(), r.begin(), r.end()); r.clear(); }
                if (r.empty()) {
                    int others = 0; for (int q = 0; q < NW; q++) if (alive[q] && q != w && !is_host[q]) others++;
                    if (is_host[w] && (others > 0 || !mail.empty())) {
                        if (!mail.empty()) continue;
                        double nxt = 1e300; for (int q = 0; q < NW; q++) if (alive[q] && q != w && !is_host[q] && clk[q] < nxt) nxt = clk[q];
                        clk[w] = nxt + 1e-9; continue;
                    }
                    alive[w] = 0; nalive--; continue;
                }
                int mk = 0;
                for (auto &x : r) { const int k = kof(x.pix, x.step); if (k > mk) mk = k; }
                const double cost = A * mk + B;
                c += cost; clk[w] += cost;
                std::vector<Ray> nr;
                for (auto &x : r) { x.step++; if (x.step < g_n[x.pix]) nr.push_back(x); }
                r.swap(nr);
            }
        }
        printf("event-driven pooling, %2d waves per workgroup, %d host(s), T=%d %10.1f M wave-instr  (x%.3f of nested)  lane utilisation %.3f\n",
               NW, NH, T, c / 1e6, c / c_nested, ideal / c);
    }
    // Packets formed from the previous frame's step counts: the 64 quads (2x2 pixels, kept together for the texture
    // derivatives) of a 32x8 strip are sorted by their longest ray and dealt 16 to a wave, so a wave's rays have similar
    // lengths a priori.  Upper bound: sorted by THIS frame's step counts.  Then nested, and with event-driven host pooling.
    for (int pooled = 0; pooled < 2; pooled++) {
        const int T = 32;
        double c = 0.0;
        for (int by = 0; by < PY; by++) for (int sx = 0; sx < (PX + 3) / 4; sx++) {
            struct Quad { int key; int pix[4]; };
            std::vector<Quad> quads;
            for (int qy = 0; qy < 4; qy++) for (int qx = 0; qx < 16; qx++) {
                Quad q; q.key = 0;
                for (int k = 0; k < 4; k++) {
                    const int x = sx * 32 + qx * 2 + (k & 1), y = by * 8 + qy * 2 + (k >> 1);
                    q.pix[k] = (x < W && y < H) ? y * W + x : -1;
                    if (q.pix[k] >= 0 && g_n[q.pix[k]] > q.key) q.key = g_n[q.pix[k]];
                }
                quads.push_back(q);
            }
            std::stable_sort(quads.begin(), quads.end(), [](const Quad &a, const Quad &b) { return a.key > b.key; });
            std::vector<Ray> wv[4], mail;
            for (int j = 0; j < 64; j++) for (int k = 0; k < 4; k++) { const int p = quads[j].pix[k]; if (p >= 0 && g_n[p] > 0) wv[j / 16].push_back(Ray{ p, 0 }); }
            if (!pooled) { for (int q = 0; q < 4; q++) c += run_wave(wv[q], 0, A, B, &dummy); continue; }
            double clk[4] = { 0, 0, 0, 0 };
            bool alive[4]; int host = -1, nalive = 0;
            for (int q = 0; q < 4; q++) { alive[q] = !wv[q].empty(); nalive += alive[q]; }
            while (nalive > 0) {
                int w = -1;
                for (int q = 0; q < 4; q++) if (alive[q] && (w < 0 || clk[q] < clk[w])) w = q;
                std::vector<Ray> &r = wv[w];
                if ((int)r.size() <= T && host < 0) host = w;
                if (w == host) { while (r.size() < 64 && !mail.empty()) { r.push_back(mail.back()); mail.pop_back(); } }
                else if ((int)r.size() <= T && host >= 0) { mail.insert(mail.end(), r.begin(), r.end()); r.clear(); }
                if (r.empty()) {
                    if (w == host && nalive > 1) { double nxt = 1e300; for (int q = 0; q < 4; q++) if (alive[q] && q != w && clk[q] < nxt) nxt = clk[q]; clk[w] = nxt + 1e-9; continue; }
                    alive[w] = false; nalive--; continue;
                }
                int mk = 0;
                for (auto &x : r) { const int k = kof(x.pix, x.step); if (k > mk) mk = k; }
                const double cost = A * mk + B;
                c += cost; clk[w] += cost;
                std::vector<Ray> nr;
                for (auto &x : r) { x.step++; if (x.step < g_n[x.pix]) nr.push_back(x); }
                r.swap(nr);
            }
        }
        printf("waves formed from quads sorted by ray length within a strip%s %10.1f M wave-instr  (x%.3f of nested)  lane utilisation %.3f\n",
               pooled ? " + host pooling" : "               ", c / 1e6, c / c_nested, ideal / c);
    }
    // the same with larger sorting domains: DWxDH packets (8x8 pixels each) per domain, quads sorted by ray length over the
    // whole domain and dealt 16 to a wave; no run-time pooling
    for (int cfg = 0; cfg < 6; cfg++) {
        const int DWs[6] = { 4, 8, 8, 16, 30, 240 }, DHs[6] = { 1, 1, 2, 4, 9, 135 };
        const int DW = DWs[cfg], DH = DHs[cfg];
        double c = 0.0;
        for (int dy = 0; dy < (PY + DH - 1) / DH; dy++) for (int dx = 0; dx < (PX + DW - 1) / DW; dx++) {
            struct Quad { int key; int pix[4]; };
            std::vector<Quad> quads;
            for (int qy = 0; qy < DH * 4; qy++) for (int qx = 0; qx < DW * 4; qx++) {
                Quad q; q.key = 0; bool any = false;
                for (int k = 0; k < 4; k++) {
                    const int x = dx * DW * 8 + qx * 2 + (k & 1), y = dy * DH * 8 + qy * 2 + (k >> 1);
                    q.pix[k] = (x < W && y < H) ? y * W + x : -1;
                    if (q.pix[k] >= 0) { any = true; if (g_n[q.pix[k]] > q.key) q.key = g_n[q.pix[k]]; }
                }
                if (any) quads.push_back(q);
            }
            std::stable_sort(quads.begin(), quads.end(), [](const Quad &a, const Quad &b) { return a.key > b.key; });
            for (size_t j = 0; j < quads.size(); j += 16) {
                std::vector<Ray> wv;
                for (size_t jj = j; jj < std::min(quads.size(), j + 16); jj++)
                    for (int k = 0; k < 4; k++) { const int p = quads[jj].pix[k]; if (p >= 0 && g_n[p] > 0) wv.push_back(Ray{ p, 0 }); }
                c += run_wave(wv, 0, A, B, &dummy);
            }
        }
        printf("quads sorted by ray length over %3dx%-3d packets (%5d waves per domain) %10.1f M wave-instr  (x%.3f of nested)  lane utilisation %.3f\n",
               DW, DH, DW * DH, c / 1e6, c / c_nested, ideal / c);
    }
    // key variants on the 16x4 domain: total escape iterations of the quad's longest ray; step count in coarse bins; the step
    // counts of the frame 1/60 s and 0.1 s earlier (animated camera)
    for (int variant = 0; variant < 4; variant++) {
        const int DW = 16, DH = 4;
        double c = 0.0;
        for (int dy = 0; dy < (PY + DH - 1) / DH; dy++) for (int dx = 0; dx < (PX + DW - 1) / DW; dx++) {
            struct Quad { int key; int pix[4]; };
            std::vector<Quad> quads;
            for (int qy = 0; qy < DH * 4; qy++) for (int qx = 0; qx < DW * 4; qx++) {
                Quad q; q.key = 0; bool any = false;
                for (int k = 0; k < 4; k++) {
                    const int x = dx * DW * 8 + qx * 2 + (k & 1), y = dy * DH * 8 + qy * 2 + (k >> 1);
                    q.pix[k] = (x < W && y < H) ? y * W + x : -1;
                    if (q.pix[k] >= 0) {
                        any = true;
                        int key = 0;
                        if (variant == 0) { for (int st = 0; st < g_n[q.pix[k]]; st++) key += kof(q.pix[k], st); }
                        else if (variant == 1) key = (g_n[q.pix[k]] + 7) / 8;
                        else if (variant == 2) key = n_prev60[q.pix[k]];
                        else key = n_prev10[q.pix[k]];
                        if (key > q.key) q.key = key;
                    }
                }
                if (any) quads.push_back(q);
            }
            std::stable_sort(quads.begin(), quads.end(), [](const Quad &a, const Quad &b) { return a.key > b.key; });
            for (size_t j = 0; j < quads.size(); j += 16) {
                std::vector<Ray> wv;
                for (size_t jj = j; jj < std::min(quads.size(), j + 16); jj++)
                    for (int k = 0; k < 4; k++) { const int p = quads[jj].pix[k]; if (p >= 0 && g_n[p] > 0) wv.push_back(Ray{ p, 0 }); }
                c += run_wave(wv, 0, A, B, &dummy);
            }
        }
        printf("quads sorted over 16x4 packets by %-42s %10.1f M wave-instr  (x%.3f of nested)  lane utilisation %.3f\n",
               variant == 0 ? "total escape iterations of the longest ray" : variant == 1 ? "step count in bins of 8" :
               variant == 2 ? "step counts of the frame 1/60 s earlier" : "step counts of the frame 0.1 s earlier", c / 1e6, c / c_nested, ideal / c);
    }
    // nested packets + deferring straggler estimates: when an estimate has run >= K0 passes and only <= L0 lanes are still
    // iterating while other lanes of the wave wait for their tail, the stragglers are parked (their w, dr, i stay in their
    // registers) and the others go on marching; parked estimates are resumed together once >= R0 are parked or nothing else
    // is left.  Costs: pass A + 2 (per-lane iteration counters), tail B.  No pooling across packets here.
    for (int cfg = 0; cfg < 6; cfg++) {
        const int K0s[6] = { 6, 8, 8, 10, 12, 8 }, L0s[6] = { 8, 8, 16, 8, 8, 4 }, R0s[6] = { 16, 16, 24, 16, 16, 8 };
        const int K0 = K0s[cfg], L0 = L0s[cfg], R0 = R0s[cfg];
        double c = 0.0;
        std::vector<Ray> r;
        for (int by = 0; by < PY; by++) for (int bx = 0; bx < PX; bx++) {
            packet(bx, by, r);
            const int nr = (int)r.size();
            if (!nr) continue;
            std::vector<int> rem(nr, 0);        // remaining iterations of the lane's parked estimate (0 = not parked)
            std::vector<int> done_it(nr, 0);
            for (;;) {
                // lanes that can start an estimate now
                int nstart = 0, nparked = 0;
                for (int i = 0; i < nr; i++) { if (rem[i] > 0) nparked++; else if (r[i].step < g_n[r[i].pix]) nstart++; }
                if (!nstart && !nparked) break;
                if (nparked >= R0 || !nstart) {
                    // resume the parked estimates together, to the end
                    int mk = 0;
                    for (int i = 0; i < nr; i++) if (rem[i] > mk) mk = rem[i];
                    c += (A + 2) * mk + B;
                    for (int i = 0; i < nr; i++) if (rem[i] > 0) { rem[i] = 0; r[i].step++; }
                    continue;
                }
                // one wave-step of the startable lanes, parking stragglers
                std::vector<int> ks(nr, 0);
                for (int i = 0; i < nr; i++) if (rem[i] == 0 && r[i].step < g_n[r[i].pix]) ks[i] = kof(r[i].pix, r[i].step);
                int pass = 0;
                for (;;) {
                    int running = 0;
                    for (int i = 0; i < nr; i++) if (ks[i] > pass) running++;
                    if (!running) break;
                    if (pass >= K0 && running <= L0 && running < nstart) {
                        for (int i = 0; i < nr; i++) if (ks[i] > pass) { rem[i] = ks[i] - pass; ks[i] = -1; }   // parked
                        break;
                    }
                    pass++;
                }
                c += (A + 2) * pass + B;
                for (int i = 0; i < nr; i++) if (ks[i] > 0) r[i].step++;      // estimate finished -> tail -> next step
            }
        }
        printf("nested + parked stragglers (K0=%d L0=%d R0=%d) %10.1f M wave-instr  (x%.3f of nested)  lane utilisation %.3f\n",
               K0, L0, R0, c / 1e6, c / c_nested, ideal / c);
    }
    // the serial chain of the frame's longest rays: instructions a wave issues until that ray is done, when the ray marches in
    // a full 8x8 packet (max k of the packet's active lanes per step), in an 8x2 sub-packet, in its 2x2 quad, or alone
    {
        std::vector<int> top;
        { std::vector<std::pair<int,int>> v; for (size_t i = 0; i < npx; i++) if (g_n[i] > 150) v.push_back({ -(int)g_n[i], (int)i });
          std::sort(v.begin(), v.end()); for (size_t j = 0; j < v.size() && j < 12; j++) top.push_back(v[j].second); }
        for (int pix : top) {
            const int px = pix % W, py = pix / W;
            double chain[4] = { 0, 0, 0, 0 };
            const int bw[4] = { 8, 8, 2, 1 }, bh[4] = { 8, 2, 2, 1 };
            for (int m = 0; m < 4; m++) {
                const int x0 = px / bw[m] * bw[m], y0 = py / bh[m] * bh[m];
                for (int st = 0; st < g_n[pix]; st++) {
                    int mk = 0;
                    for (int y = y0; y < y0 + bh[m] && y < H; y++) for (int x = x0; x < x0 + bw[m] && x < W; x++) {
                        const int q = y * W + x;
                        if (st < g_n[q]) { const int k = kof(q, st); if (k > mk) mk = k; }
                    }
                    chain[m] += A * mk + B;
                }
            }
            printf("ray (%4d,%4d) %3d estimates: chain in 8x8 packet %7.0f instr, 8x2 %7.0f, quad %7.0f, alone %7.0f\n",
                   px, py, (int)g_n[pix], chain[0], chain[1], chain[2], chain[3]);
        }
    }
    // normal estimates: the four estimates of a hit pixel share k; a wave pays max k over its hit lanes.  What would
    // regrouping the hit pixels of a workgroup (32x8 strip, 4 waves) by k save?
    {
        double now = 0, sorted = 0, lanes = 0;
        for (int by = 0; by < PY; by++) for (int sx = 0; sx < (PX + 3) / 4; sx++) {
            std::vector<int> ks;
            for (int q = 0; q < 4; q++) {
                const int bx = sx * 4 + q; if (bx >= PX) break;
                int mk = 0;
                for (int ly = 0; ly < 8; ly++) for (int lx = 0; lx < 8; lx++) {
                    const int x = bx * 8 + lx, y = by * 8 + ly;
                    if (x < W && y < H && kn[(size_t)y * W + x]) { const int k = kn[(size_t)y * W + x] - 1; ks.push_back(k); if (k > mk) mk = k; lanes += k; }
                }
                now += mk;
            }
            std::sort(ks.begin(), ks.end(), [](int a, int b) { return a > b; });
            for (size_t i = 0; i < ks.size(); i += 64) sorted += ks[i];
        }
        printf("normal estimates (per estimate): wave passes now %.3e, hit pixels of a workgroup regrouped by k %.3e (x%.3f); lane utilisation %.3f -> %.3f\n",
               now, sorted, sorted / now, lanes / 64 / now, lanes / 64 / sorted);
    }
    return 0;
}
