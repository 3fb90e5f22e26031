// hipemu — a minimal wave64 SIMT emulator for the host CPU.  TEST INFRASTRUCTURE ONLY (like oracle/): nothing under
// compress_amd/ includes or links it.  It lets the wave-synchronous HIP kernels of this repository be compiled by g++
// (`-I tools/hipemu` puts this file in front of the real <hip/hip_runtime.h>) and run lane by lane on the CPU, so the
// device algorithms can be checked against the oracle without a GPU (tests/test_emu_lds.py).
//
// Model: a workgroup is a set of fibers (one per thread), scheduled cooperatively on one OS thread; a fiber runs until
// it reaches a cross-lane operation (__ballot, __shfl*, readfirstlane, KC_WAVE_SYNC, __syncthreads), where it waits for
// the other live lanes of its wave (workgroup).  Lanes therefore run maximally OUT of lockstep between two such
// points: any communication through LDS that is not fenced by KC_WAVE_SYNC() shows up as a wrong result here, which is
// the property a test wants.  Every cross-lane operation checks that all lanes arrive with the same kind of operation
// and the same count of operations executed so far (divergent collectives abort with a message).  Only wave-uniform
// control flow around collectives is supported.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <functional>
#include <utility>

#define KC_HIPEMU 1

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { uint32_t x, y; };
struct uint4 { uint32_t x, y, z, w; } __attribute__((aligned(16)));
static inline uint2 make_uint2(uint32_t x, uint32_t y) { uint2 r; r.x = x; r.y = y; return r; }
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

typedef void* hipStream_t;
typedef int hipError_t;
#define hipSuccess 0

namespace hipemu {
const dim3& thread_idx();
const dim3& block_idx();
const dim3& block_dim();
const dim3& grid_dim();
uint64_t ballot(bool p);
uint64_t shfl64(uint64_t v, int src_lane);           // absolute lane 0..63 of the caller's wave
uint64_t first_lane64(uint64_t v);
void wave_sync();
void block_sync();
int block_or(int p);  // __syncthreads_or
int lane();
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
void set_reverse(bool r);    // the scheduler visits the lanes from 63 down (same-address stores of one instruction: the lowest lane's stays)
void set_group(unsigned g);  // sub-wave groups of g lanes rendezvous among themselves (kernels whose groups diverge); 64: whole waves
uint64_t collectives();   // cross-lane operations executed so far (diagnostics)
void pause();             // s_sleep: the lane gives way without a rendezvous (a wave polling a flag another wave of the block sets)
}  // namespace hipemu

#define threadIdx (hipemu::thread_idx())
#define blockIdx (hipemu::block_idx())
#define blockDim (hipemu::block_dim())
#define gridDim (hipemu::grid_dim())

#define __global__
#define __device__
#define __host__
#define __constant__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#ifndef __noinline__
#define __noinline__ __attribute__((noinline))
#endif

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    hipemu::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); })

// ---- cross-lane operations ----
static inline uint64_t __ballot(int p) { return hipemu::ballot(p != 0); }
static inline int emu_src_lane(int lane, int src, int width) { return (lane & ~(width - 1)) | (src & (width - 1)); }
static inline int __shfl(int v, int src, int width = 64) {
    return (int)(uint32_t)hipemu::shfl64((uint32_t)v, emu_src_lane(hipemu::lane(), src, width));
}
static inline int __shfl_up(int v, unsigned d, int width = 64) {
    const int l = hipemu::lane(), lw = l & (width - 1);
    const int src = lw >= (int)d ? l - (int)d : l;
    return (int)(uint32_t)hipemu::shfl64((uint32_t)v, src);
}
static inline int __shfl_down(int v, unsigned d, int width = 64) {
    const int l = hipemu::lane(), lw = l & (width - 1);
    const int src = lw + (int)d < width ? l + (int)d : l;
    return (int)(uint32_t)hipemu::shfl64((uint32_t)v, src);
}
static inline int __shfl_xor(int v, int m, int width = 64) {
    const int l = hipemu::lane();
    return (int)(uint32_t)hipemu::shfl64((uint32_t)v, emu_src_lane(l, (l & (width - 1)) ^ m, width));
}
static inline void __syncthreads() { hipemu::block_sync(); }
// v_mov_b32 with a DPP control word: quad_perm (below 0x100: two bits per lane of the quad select the source lane), row_shr:n
// (0x111..0x11f), row_bcast:15 (0x142: lane 15 of a row to the whole next row), row_bcast:31 (0x143: lane 31 to rows 2 and 3).  A lane
// whose row / bank is masked off, or that has no source lane, keeps `old` (0 with bound_ctrl when only the source is missing).
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int row_mask, int bank_mask, bool bound_ctrl) {
    const int lane = hipemu::lane();
    int from = -1;
    if (ctrl >= 0 && ctrl < 0x100) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);
    else if (ctrl >= 0x111 && ctrl <= 0x11f) { const int n = ctrl & 15; if ((lane & 15) >= n) from = lane - n; }
    else if (ctrl == 0x142) { if (lane >= 16) from = (lane & ~15) - 1; }
    else if (ctrl == 0x143) { if (lane >= 32) from = 31; }
    else { fprintf(stderr, "hipemu: DPP control 0x%x not emulated\n", ctrl); abort(); }
    const uint32_t v = (uint32_t)hipemu::shfl64((uint32_t)src, from < 0 ? lane : from);  // (every lane takes part in the exchange)
    if (!((row_mask >> (lane >> 4)) & 1) || !((bank_mask >> ((lane >> 2) & 3)) & 1)) return old;
    if (from < 0) return bound_ctrl ? 0 : old;
    return (int)v;
}
static inline void __builtin_amdgcn_fence(int, const char*) {}  // (always next to a wave barrier, which is the rendezvous here)
static inline long long clock64() { return 0; }
static inline void __builtin_amdgcn_s_sleep(int) { hipemu::pause(); }
static inline int __syncthreads_or(int p) { return hipemu::block_or(p); }
static inline int __builtin_amdgcn_readfirstlane(int v) { return (int)(uint32_t)hipemu::first_lane64((uint32_t)v); }
static inline int __builtin_amdgcn_readlane(int v, int src) { return (int)(uint32_t)hipemu::shfl64((uint32_t)v, src & 63); }
static inline void __builtin_amdgcn_wave_barrier() { hipemu::wave_sync(); }
static inline uint32_t __builtin_amdgcn_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) {
    return (uint32_t)((((uint64_t)hi << 32) | lo) >> (8 * (sh & 3)));
}
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
static inline int __popcll(uint64_t v) { return __builtin_popcountll(v); }
static inline unsigned long long __builtin_amdgcn_s_memtime() { return 0; }

// ---- atomics (one OS thread: plain read-modify-write) ----
template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = (T)(o + v); return o; }
template <class T> static inline T atomicOr(T* p, T v) { T o = *p; *p = (T)(o | v); return o; }
template <class T> static inline T atomicAnd(T* p, T v) { T o = *p; *p = (T)(o & v); return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <class T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <class T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <class T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }

// ---- the host API (hipMalloc, streams, events ...) as synchronous stand-ins: only for the whole-library build of
// tools/emu_host_check.py (-DKC_HIPEMU_HOST); the kernel-level emulator tests do not see it ----
#ifdef KC_HIPEMU_HOST
#include "hip_host_emu.h"
#endif
