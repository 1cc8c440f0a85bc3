/* tests/koh_shim/hip/hip_runtime.h -- TEST INFRASTRUCTURE.  <hip/hip_runtime.h> for tests/kernel_on_host.cpp: csrc/rmdf_render.hip -- the
 * render kernel's SOURCE, launch code included -- compiled for the CPU and executed by a small SIMT emulator:
 *   * one FIBER per lane (a hand-written context switch: callee-saved registers and the stack pointer), 64 lanes per wave, all waves of a workgroup on one OS thread, scheduled round-robin;
 *   * __ballot / __shfl / __shfl_xor / readfirstlane / DPP are true 64-lane collectives: a lane that calls one yields until all 64 lanes
 *     of its wave have called the same kind of collective, then everyone reads the exchanged values.  That is the hardware's meaning for
 *     wave-UNIFORM call sites, which is what the kernel's own collectives are; the per-lane "is anyone here in trouble" tests of divergent
 *     code go through RMDF_LANES_HERE (rmdf_device.hpp) and act on the calling lane alone;
 *   * __syncthreads is a workgroup barrier; __shared__ is static thread_local storage (one workgroup at a time per OS thread); LDS / global
 *     atomics are plain operations (fibers of a workgroup never run concurrently); s_sleep yields;
 *   * threadIdx / blockIdx / gridDim are the running fiber's; hipLaunchKernelGGL runs the grid, workgroups spread over OS threads;
 *   * v_rsq_f32 / v_rcp_f32 / v_sqrt_f32 as in tests/doh_shim (seed modes 0..3).
 * A deadlock (a collective some lane never reaches) aborts with the site of the waiting lanes instead of hanging. */
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __device__
#define __host__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __noinline__ __attribute__((noinline))
#define __launch_bounds__(...)
#define __shared__ static thread_local

/* the vector types under HIP's own names: a kernel's mangled name (what the HIP double looks the emulated kernel up by) spells its parameter types */
template <typename T, unsigned N> struct HIP_vector_type;
template <typename T> struct HIP_vector_type<T, 2u> { T x, y; };
template <typename T> struct HIP_vector_type<T, 3u> { T x, y, z; };
template <typename T> struct HIP_vector_type<T, 4u> { T x, y, z, w; };
typedef HIP_vector_type<float, 2u> float2;
typedef HIP_vector_type<float, 4u> float4;
typedef HIP_vector_type<unsigned, 2u> uint2;
typedef HIP_vector_type<unsigned, 3u> uint3;
typedef HIP_vector_type<unsigned, 4u> uint4;
struct dim3 { unsigned x, y, z; dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) { } };
static inline float2 make_float2(float x, float y) { float2 r = { x, y }; return r; }
static inline float4 make_float4(float x, float y, float z, float w) { float4 r = { x, y, z, w }; return r; }
static inline uint2 make_uint2(unsigned x, unsigned y) { uint2 r = { x, y }; return r; }

static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __popcll(unsigned long long m) { return __builtin_popcountll(m); }

typedef int hipError_t;
typedef void *hipStream_t;
#define hipSuccess 0
#define hipErrorInvalidValue 1
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }   /* (launchers of the cross-check schedule) */

namespace koh {
/* per lane: the Mandelbulb iteration passes of every mb8_iterate_t call ("segment") since the wave's last collective -- the emulator turns them
 * into the wave's lock-step cost there (max over lanes per segment) and the lanes' useful work (their own): rmdf_device.hpp RMDF_EMU_PASS */
struct Lane { uint3 tid; int lane, wave; unsigned short seg[24]; int nseg; unsigned short cur_passes; unsigned cost; };
struct Grid { uint3 bid; dim3 grid, block; };
extern thread_local Lane *cur;
extern thread_local Grid *grd;
void yield_lane();
const uint64_t *wave_gather(uint64_t v, const char *what);      // every lane of the wave calls; returns the 64 contributed values
void count(int kind);                                           // 0 ballot, 1 shfl, 2 readfirstlane, 3 DPP, 4 polled load, 5 __syncthreads (per lane-call)
void block_barrier();
void launch(dim3 grid, dim3 block, size_t dyn_lds_bytes, void (*tramp)(void *), void *closure);
template <typename F> void launch(dim3 grid, dim3 block, size_t dyn_lds_bytes, F fn) { launch(grid, block, dyn_lds_bytes, [](void *c) { (*(F *)c)(); }, &fn); }
void *dyn_lds();                                                // the workgroup's dynamic LDS (`extern __shared__`), sized by the launch
template <typename T> inline uint64_t pack(T v) { uint64_t u = 0; static_assert(sizeof(T) <= 8, ""); memcpy(&u, &v, sizeof v); return u; }
template <typename T> inline T unpack(uint64_t u) { T v; memcpy(&v, &u, sizeof v); return v; }
}  // namespace koh

#define RMDF_EMU_PASS() (koh::cur->cur_passes++)
#define RMDF_EMU_COST(n) (koh::cur->cost += (n))      /* instructions of a lane's serial chain since the wave's last collective (Cornell box) */
/* (bit 15 of an entry: the segment continues an estimate begun earlier -- the AO queue's second half -- and pays no second per-estimate overhead) */
#define RMDF_EMU_SEGMENT_END(i0) do { koh::Lane *l_ = koh::cur; const unsigned short e_ = (unsigned short)(l_->cur_passes | ((i0) != 0 ? 0x8000u : 0u)); \
                                       if (l_->nseg < 24) l_->seg[l_->nseg++] = e_; else l_->seg[23] = (unsigned short)(l_->seg[23] + l_->cur_passes); l_->cur_passes = 0; } while (0)
#define threadIdx (koh::cur->tid)
#define blockIdx (koh::grd->bid)
#define gridDim (koh::grd->grid)
#define blockDim (koh::grd->block)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) koh::launch((grid), (block), (size_t)(shmem), [&] { kernel(__VA_ARGS__); })
typedef int hipFuncAttribute;
#define hipFuncAttributeMaxDynamicSharedMemorySize 8
static inline hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int bytes) { return bytes <= 160 * 1024 ? hipSuccess : hipErrorInvalidValue; }   /* 160 KB of LDS per CU */

static inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __syncthreads() { koh::count(5); koh::block_barrier(); }
static inline unsigned long long __ballot(int pred)
{
    koh::count(0);
    const uint64_t *v = koh::wave_gather(pred ? 1u : 0u, "__ballot");
    unsigned long long m = 0ull;
    for (int i = 0; i < 64; i++) m |= (unsigned long long)(v[i] & 1u) << i;
    return m;
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) { (void)width; koh::count(1); return koh::unpack<T>(koh::wave_gather(koh::pack(v), "__shfl")[src & 63]); }
template <typename T> static inline T __shfl_xor(T v, int mask, int width = 64) { (void)width; koh::count(1); return koh::unpack<T>(koh::wave_gather(koh::pack(v), "__shfl_xor")[(koh::cur->lane ^ mask) & 63]); }
static inline int koh_readfirstlane(int v) { koh::count(2); return koh::unpack<int>(koh::wave_gather(koh::pack(v), "readfirstlane")[0]); }
static inline int koh_update_dpp(int old, int src, int ctrl)
{
    (void)old;
    koh::count(3);
    const int l = koh::cur->lane;
    int from;
    if (ctrl >= 0 && ctrl <= 0xff) from = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3);            /* quad_perm */
    else if (ctrl == 0x141) from = (l & ~7) | (7 - (l & 7));                                      /* row_half_mirror */
    else if (ctrl == 0x140) from = (l & ~15) | (15 - (l & 15));                                   /* row_mirror */
    else { fprintf(stderr, "koh: DPP control 0x%x not emulated\n", ctrl); abort(); }
    return koh::unpack<int>(koh::wave_gather(koh::pack(src), "update_dpp")[from]);
}
/* v_mul_f32_dpp ... row_newbcast:K -- lane K of the caller's 16-lane row (csrc/rmdf_env.hip: mul_row_bcast) */
static inline float koh_row_newbcast(float v, int k)
{
    koh::count(3);
    return koh::unpack<float>(koh::wave_gather(koh::pack(v), "row_newbcast")[(koh::cur->lane & ~15) | (k & 15)]);
}
static inline unsigned long long koh_ticks() { static thread_local unsigned long long t = 0; return t += 100; }
#define __builtin_amdgcn_s_memrealtime() koh_ticks()
#define __builtin_amdgcn_s_memtime() koh_ticks()
#define __builtin_amdgcn_readfirstlane(v) koh_readfirstlane(v)
#define __builtin_amdgcn_update_dpp(old, src, ctrl, rmask, bmask, bc) koh_update_dpp((old), (src), (ctrl))
static inline int koh_mbcnt_lo(unsigned m, int base) { const int l = koh::cur->lane; return base + __builtin_popcount(l >= 32 ? m : (m & ((1u << l) - 1u))); }
static inline int koh_mbcnt_hi(unsigned m, int base) { const int l = koh::cur->lane; return base + (l > 32 ? __builtin_popcount(m & ((1u << (l - 32)) - 1u)) : 0); }
#define __builtin_amdgcn_mbcnt_lo(m, b) koh_mbcnt_lo((m), (b))
#define __builtin_amdgcn_mbcnt_hi(m, b) koh_mbcnt_hi((m), (b))
#define __builtin_amdgcn_s_sleep(n) koh::yield_lane()

/* atomics: real ones (workgroups of one launch run on several OS threads and may meet in global counters) */
static inline int atomicCAS(int *p, int cmp, int val) { __atomic_compare_exchange_n(p, &cmp, val, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return cmp; }
template <typename T, typename V> static inline T atomicAdd(T *p, V v) { return __atomic_fetch_add(p, (T)v, __ATOMIC_SEQ_CST); }
static inline unsigned atomicMax(unsigned *p, unsigned v) { unsigned o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (v > o && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) { } return o; }
/* (__hip_atomic_fetch_add / _store are clang builtins on every target: the kernel's scoped atomics compile as they are.)
 * __hip_atomic_load is the exception: the kernel polls workgroup state with it from wave-uniform code, and on the hardware ONE load
 * instruction gives all 64 lanes the same answer.  Fibers run one after the other, so each lane would see the flag at its own moment and
 * the wave would split: here the load is a wave collective and every lane takes lane 0's value. */
template <typename T> static inline T koh_uniform_load(T *p)
{
    const T mine = *(volatile T *)p;
    koh::count(4);
    return koh::unpack<T>(koh::wave_gather(koh::pack(mine), "__hip_atomic_load (wave-uniform)")[0]);
}
#define __hip_atomic_load(p, order, scope) koh_uniform_load(p)
#ifndef __HIP_MEMORY_SCOPE_WORKGROUP
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#endif

extern thread_local int doh_seed_mode;
extern thread_local unsigned doh_seed_rng;
static inline float doh_perturb(float y)
{
    int m = doh_seed_mode;
    if (m == 3) { doh_seed_rng = doh_seed_rng * 1664525u + 1013904223u; m = (int)((doh_seed_rng >> 24) % 3u); }
    if (m == 0 || !(y == y) || y == 0.0f || isinf(y)) return y;
    unsigned u = __float_as_uint(y);
    u = (m == 1) ? u + 1u : u - 1u;
    return __uint_as_float(u);
}
static inline float doh_rsq(float x) { return doh_perturb((float)(1.0 / sqrt((double)x))); }
static inline float doh_rcp(float x) { return doh_perturb((float)(1.0 / (double)x)); }
static inline float doh_sqrt(float x) { return doh_perturb((float)sqrt((double)x)); }
#define __builtin_amdgcn_rsqf(x) doh_rsq(x)
#define __builtin_amdgcn_rcpf(x) doh_rcp(x)
#define __builtin_amdgcn_sqrtf(x) doh_sqrt(x)
