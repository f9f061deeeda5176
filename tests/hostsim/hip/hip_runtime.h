// A HOST stand-in for <hip/hip_runtime.h> — TEST INFRASTRUCTURE ONLY (tests/test_hostsim_cpu.py): the translation units of
// xclim_amd/csrc that use neither LDS nor wave intrinsics (detrend, window, runlen, reduce, reduce2, spell, elemwise) are compiled
// unchanged with g++ against this header, and hipLaunchKernelGGL runs a kernel THREAD BY THREAD on the CPU.  That pins the
// kernels' arithmetic and the entry points' dispatch logic against the oracle without a GPU.  Nothing here is ever linked into
// libxclimhip.so; the product path has no CPU fallback.
#pragma once
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct SimIdx { unsigned x, y, z; };
extern thread_local SimIdx blockIdx, threadIdx, gridDim, blockDim;

typedef void* hipStream_t;
typedef void* hipEvent_t;
typedef int hipError_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
static inline const char* hipGetErrorString(hipError_t) { return "host simulation"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = malloc(n ? n : 1); return hipSuccess; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
template <typename F> static inline hipError_t hipFuncSetAttribute(F, int, int) { return hipSuccess; }
enum { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

// (a macro, like the real one: kernels with default arguments are launched with fewer than they declare)
#ifdef SIM_FIBERS
#include "../simt.h"
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) sim_launch_fibers([=]() { kern(__VA_ARGS__); }, (grid), (block))
#else
template <typename F>
static inline void sim_launch_loop(F body, dim3 grid, dim3 block) {
  gridDim = {grid.x, grid.y, grid.z};
  blockDim = {block.x, block.y, block.z};
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = {bx, by, bz};
        for (unsigned tz = 0; tz < block.z; ++tz)
          for (unsigned ty = 0; ty < block.y; ++ty)
            for (unsigned tx = 0; tx < block.x; ++tx) {
              threadIdx = {tx, ty, tz};
              body();
            }
      }
}
#define hipLaunchKernelGGL(kern, grid, block, lds, stream, ...) sim_launch_loop([=]() { kern(__VA_ARGS__); }, (grid), (block))
#endif

struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
struct uchar4 { unsigned char x, y, z, w; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline double2 make_double2(double a, double b) { return double2{a, b}; }
static inline uchar4 make_uchar4(unsigned char a, unsigned char b, unsigned char c, unsigned char d) { return uchar4{a, b, c, d}; }
static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline double __longlong_as_double(long long v) { double d; memcpy(&d, &v, 8); return d; }
static inline long long __double_as_longlong(double d) { long long v; memcpy(&v, &d, 8); return v; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
#define __builtin_readcyclecounter() 0ull

// pdoy.h speaks to the wave (readlane, buffer loads): every unit that includes it is built on fibers (simt.h: real exchanges, the
// buffer loads are plain loads); thread by thread these stand-ins only let a header parse and abort if anything reaches them.
typedef struct { const void* p; } __amdgpu_buffer_rsrc_t;
#ifndef SIM_FIBERS
static inline int __builtin_amdgcn_readfirstlane(int) { abort(); }
static inline int __builtin_amdgcn_readlane(int, int) { abort(); }
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void*, short, int, int) { abort(); }
static inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t, int, int, int) { abort(); }
#else
static inline __amdgpu_buffer_rsrc_t __builtin_amdgcn_make_buffer_rsrc(void* p, short, int, int) { return __amdgpu_buffer_rsrc_t{p}; }
static inline unsigned __builtin_amdgcn_raw_buffer_load_b32(__amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
  unsigned v;
  memcpy(&v, (const char*)r.p + (unsigned)voff + (unsigned)soff, 4);
  return v;
}
static inline void __builtin_amdgcn_raw_buffer_store_b32(unsigned v, __amdgpu_buffer_rsrc_t r, int voff, int soff, int) {
  memcpy((char*)const_cast<void*>(r.p) + (unsigned)voff + (unsigned)soff, &v, 4);
}
#endif

// plane.hip: the work list is filled by wave-aggregated appends (ballot + one atomic per wave + shuffle).  One thread at a time IS a
// wave whose only active lane is that thread: the stand-ins below (selected per unit by -D in tests/hostsim/simdevice.py) make the
// same code append one entry per call; lane-private LDS columns become static arrays.
static inline unsigned sim_atomic_add(unsigned* p, unsigned v) { const unsigned old = *p; *p = old + v; return old; }
