// mall_ubench.hip — does the 256 MiB Infinity Cache serve a RE-READ faster than HBM, and how much of it is usable?
// (VERDICT r4 #1a: the two-pass histogram select of select4.hip reads every sample twice; its floor is two HBM streams
// unless the second one comes from the die.)  Rates are by TIME (FETCH_SIZE appears to count Infinity-Cache hits).
//
//   mode "flat":  a buffer of S MiB is streamed twice by two back-to-back launches of a plain coalesced reader
//                 (16 B per lane); reported: rate of the cold pass and of the re-read, S = 32 ... 1024 MiB.
//   mode "tile":  select4's own access pattern.  The field is (T, C) time-major, a TILE = 64 adjacent columns
//                 (256-byte row segments, T rows, stride C floats).  256 persistent workgroups of 1024 threads; P
//                 workgroups share a tile, each streams T / P of its rows `passes` times before the group moves to its
//                 next tile -> live set = (256 / P) tiles x 2.8 MB: P = 1: 717 MB, 2: 359 MB, 4: 179 MB, 8: 90 MB,
//                 16: 45 MB.  time(passes = 2) - time(passes = 1) is what the second pass of a T-split select4 would
//                 cost; P = 1, passes = 1 is today's pass (6.1 TB/s).  `samexcd` puts the P workgroups of a tile on one
//                 XCD (blockIdx % 8 is the XCD).
//   hipcc --offload-arch=gfx950 -O3 tools/mall_ubench.hip -o tools/mall_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x)                                                                     \
  do {                                                                               \
    hipError_t e_ = (x);                                                             \
    if (e_ != hipSuccess) {                                                          \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));      \
      exit(1);                                                                       \
    }                                                                                \
  } while (0)

__global__ void __launch_bounds__(256) k_fill(float4* __restrict__ p, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256)
    p[i] = make_float4(1.0f, 2.0f, 3.0f, 4.0f);
}

// plain coalesced reader, 8 x 16 B in flight per lane, grid-strided
__global__ void __launch_bounds__(256) k_flat(const float4* __restrict__ p, int64_t n4, float* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  float acc = 0.f;
  for (; i + 7 * stride < n4; i += 8 * stride) {
    float4 b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) b[u] = p[i + u * stride];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += b[u].x + b[u].y + b[u].z + b[u].w;
  }
  for (; i < n4; i += stride) acc += p[i].x;
  if (acc == 12345.678f) out[0] = acc;
}

// select4's streaming ring: thread (col, rl) reads rows rl, rl + 16, ... of its column, NSET sets of U loads in flight
constexpr int CW = 64, RL = 16, NT = CW * RL, U = 8, NSET = 5, ROWS = RL * U;

struct Ring {
  float S[NSET][U];
  int st_sign;
  __device__ __forceinline__ void load(float (&dst)[U], const float* __restrict__ x, int64_t st, uint32_t voff, int kb) {
    const float* base = x + (int64_t)(kb * st_sign) * ROWS * st;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
    const uint32_t rowstep = (uint32_t)(st * 4 * RL);
    uint32_t soff = 0u;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      dst[u] = __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
      soff += rowstep;
    }
  }
  template <typename F>
  __device__ __forceinline__ void run(const float* __restrict__ x, int T, int64_t st, int64_t cc, int rl, bool rev, F&& f) {
    const uint32_t voff = (uint32_t)(((int64_t)rl * st + cc) * 4);
    const int nfull = T / ROWS;  // (the tail rows are ignored: a bandwidth test)
    // rev: the batches from the last one down to the first (what was read LAST is re-read FIRST)
    if (rev) { x += (int64_t)(nfull - 1) * ROWS * st; st_sign = -1; } else st_sign = 1;
#pragma unroll
    for (int i = 0; i < NSET - 1; ++i)
      if (i < nfull) load(S[i], x, st, voff, i);
    int done = 0;
    while (done + 2 * NSET - 1 <= nfull) {
#pragma unroll
      for (int i = 0; i < NSET; ++i) {
        load(S[(i + NSET - 1) % NSET], x, st, voff, done + i + NSET - 1);
        f(S[i]);
      }
      done += NSET;
    }
#pragma unroll
    for (int i = 0; i < 2 * NSET - 2; ++i) {
      if (done + i < nfull) {
        if (done + i + NSET - 1 < nfull) load(S[(i + NSET - 1) % NSET], x, st, voff, done + i + NSET - 1);
        f(S[i % NSET]);
      }
    }
  }
};

// group barrier for the P workgroups of a tile (all 256 workgroups are resident: one per CU): monotone counter per group
__device__ __forceinline__ void group_sync(uint32_t* ctr, uint32_t target) {
  __syncthreads();
  if (threadIdx.x == 0) {
    // relaxed polling (an acquire load per poll is a cache invalidate per poll: 11-25 us per sync in the first version)
    __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
  }
  __syncthreads();
}

__global__ void __launch_bounds__(NT, 4)
k_tile(const float* __restrict__ x, int T, int64_t C, int64_t st, int P, int passes, int samexcd, int sync, int lds,
       uint32_t* __restrict__ ctrs, float* __restrict__ out, int rev) {
  extern __shared__ float pad[];
  if (lds < 0) pad[threadIdx.x] = 0.f;
  const int tid = threadIdx.x, col = tid & (CW - 1), rl = tid / CW;
  const int b = blockIdx.x, nb = gridDim.x, ngroups = nb / P;
  int group, part;
  if (samexcd) {  // blockIdx % 8 = XCD: the P parts of a group keep the same residue
    const int xcd = b & 7, k = b >> 3;   // k = 0 .. nb/8 - 1 on this XCD
    part = k % P;
    group = xcd + 8 * (k / P);
  } else {
    group = b / P;
    part = b % P;
  }
  const int64_t ntiles = C / CW;
  const int r0 = (int)((int64_t)T * part / P), r1 = (int)((int64_t)T * (part + 1) / P);
  float acc = 0.f;
  Ring ring;
  uint32_t epoch = 0;
  for (int64_t tile = group; tile < ntiles; tile += ngroups) {
    const int64_t cc = tile * CW + col;
    for (int ps = 0; ps < passes; ++ps) {
      ring.run(x + (int64_t)r0 * st, r1 - r0, st, cc, rl, rev && (ps & 1), [&](const float (&v)[U]) {
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
      });
      if (sync && P > 1) {
        ++epoch;
        group_sync(ctrs + group, epoch * (uint32_t)P);
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

static float time_ms(hipEvent_t a, hipEvent_t b) {
  float ms;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms;
}

int main(int argc, char** argv) {
  const char* mode = argc > 1 ? argv[1] : "all";
  hipEvent_t e0, e1, e2;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventCreate(&e2));
  float* out;
  CHECK(hipMalloc(&out, 256));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount;
  printf("# device %s, %d CUs, L2 %d MiB\n", prop.name, ncu, prop.l2CacheSize >> 20);

  if (!strcmp(mode, "flat") || !strcmp(mode, "all")) {
    // a big "flush" buffer is streamed before every measurement so that pass 1 is cold
    const int64_t flushB = 2048ll << 20;
    float4* fl;
    CHECK(hipMalloc(&fl, flushB));
    hipLaunchKernelGGL(k_fill, dim3(ncu * 8), dim3(256), 0, 0, fl, flushB / 16);
    const int sizes[] = {32, 64, 96, 128, 160, 192, 224, 256, 320, 384, 512, 1024};
    printf("# flat: S MiB | cold pass GB/s | re-read GB/s | re-read after a 3rd pass GB/s\n");
    for (int S : sizes) {
      const int64_t bytes = (int64_t)S << 20;
      float4* p;
      CHECK(hipMalloc(&p, bytes));
      hipLaunchKernelGGL(k_fill, dim3(ncu * 8), dim3(256), 0, 0, p, bytes / 16);
      float best1 = 1e9f, best2 = 1e9f, best3 = 1e9f;
      for (int rep = 0; rep < 5; ++rep) {
        hipLaunchKernelGGL(k_flat, dim3(ncu * 8), dim3(256), 0, 0, fl, flushB / 16, out);
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_flat, dim3(ncu * 8), dim3(256), 0, 0, p, bytes / 16, out);
        CHECK(hipEventRecord(e1, 0));
        hipLaunchKernelGGL(k_flat, dim3(ncu * 8), dim3(256), 0, 0, p, bytes / 16, out);
        CHECK(hipEventRecord(e2, 0));
        CHECK(hipEventSynchronize(e2));
        const float t1 = time_ms(e0, e1), t2 = time_ms(e1, e2);
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_flat, dim3(ncu * 8), dim3(256), 0, 0, p, bytes / 16, out);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        const float t3 = time_ms(e0, e1);
        best1 = t1 < best1 ? t1 : best1;
        best2 = t2 < best2 ? t2 : best2;
        best3 = t3 < best3 ? t3 : best3;
      }
      printf("flat %5d  %8.0f  %8.0f  %8.0f   (ms %.4f %.4f %.4f)\n", S, bytes / best1 * 1e-6, bytes / best2 * 1e-6,
             bytes / best3 * 1e-6, best1, best2, best3);
      CHECK(hipFree(p));
    }
    CHECK(hipFree(fl));
  }

  if (!strcmp(mode, "tile") || !strcmp(mode, "all")) {
    const int T = argc > 2 ? atoi(argv[2]) : 10950;
    const int64_t C = argc > 3 ? atoll(argv[3]) : 1440 * 720;
    const int64_t elems = (int64_t)T * C;
    float* x;
    CHECK(hipMalloc(&x, elems * 4));
    hipLaunchKernelGGL(k_fill, dim3(ncu * 8), dim3(256), 0, 0, (float4*)x, elems / 4);
    uint32_t* ctrs;
    CHECK(hipMalloc(&ctrs, 4096 * 4));
    CHECK(hipFuncSetAttribute((const void*)k_tile, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    printf("# tile: T=%d C=%lld (%.1f GB)  P | samexcd | sync | passes=1 ms (TB/s) | passes=2 ms | second pass ms (TB/s) | live MB\n", T,
           (long long)C, elems * 4e-9);
    const int Ps[] = {1, 2, 4, 8, 16, 32};
    const size_t lds = 150 * 1024;  // one workgroup per CU, as the select kernels
    const int rev = argc > 4 ? atoi(argv[4]) : 0;
    printf("# second pass %s\n", rev ? "in REVERSE row order" : "in the same row order");
    for (int P : Ps) {
      for (int samexcd = 0; samexcd < 2; ++samexcd) {
        for (int sync = 0; sync < 2; ++sync) {
          if (P == 1 && (samexcd || sync)) continue;
          if (P > 1 && !samexcd) continue;
          float tm[3] = {0, 1e9f, 1e9f};
          for (int passes = 1; passes <= 2; ++passes) {
            for (int rep = 0; rep < 2; ++rep) {
              CHECK(hipMemsetAsync(ctrs, 0, 4096 * 4, 0));
              CHECK(hipEventRecord(e0, 0));
              hipLaunchKernelGGL(k_tile, dim3(ncu), dim3(NT), lds, 0, x, T, C, C, P, passes, samexcd, sync, 0, ctrs, out, rev);
              CHECK(hipEventRecord(e1, 0));
              CHECK(hipEventSynchronize(e1));
              const float t = time_ms(e0, e1);
              tm[passes] = t < tm[passes] ? t : tm[passes];
            }
          }
          const double gb = elems * 4e-9;
          printf("tile P=%2d xcd=%d sync=%d  %7.3f (%5.2f)  %7.3f  %7.3f (%5.2f)  %6.0f\n", P, samexcd, sync, tm[1], gb / tm[1], tm[2],
                 tm[2] - tm[1], gb / (tm[2] - tm[1]), (double)(ncu / P) * CW * T * 4e-6);
          fflush(stdout);
        }
      }
    }
    CHECK(hipFree(x));
  }
  return 0;
}
