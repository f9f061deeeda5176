// gather_ubench.hip — the access pattern of the multi-year percentile_doy kernels without their arithmetic:
// a (T = NY * 365) x C fp32 field, time-major; a wave owns 64 * VEC adjacent columns and, for each calendar day of its
// chunk, reads the NY rows (year y, day d) = 256 * VEC bytes from each of NY rows 365 rows apart.  What limits the rate:
// bytes in flight (waves per SIMD x loads per wave), the request size, or lock-step between neighbouring waves?
//   hipcc --offload-arch=gfx950 -O3 tools/gather_ubench.hip -o tools/gather_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int VEC> struct Vec;
template <> struct Vec<1> { typedef float T; };
template <> struct Vec<2> { typedef float2 T; };
template <> struct Vec<4> { typedef float4 T; };
__device__ __forceinline__ float vsum(float v) { return v; }
__device__ __forceinline__ float vsum(float2 v) { return v.x + v.y; }
__device__ __forceinline__ float vsum(float4 v) { return v.x + v.y + v.z + v.w; }

// NT threads per workgroup (NT / 64 waves on adjacent column blocks), DEPTH day-sets in flight per wave, SYNC: barrier per
// day (keeps the waves of a workgroup on the same rows), LDSPAD: bytes of LDS per workgroup to cap the occupancy
template <int NY, int VEC, int NT, int DEPTH, bool SYNC>
__global__ void __launch_bounds__(NT) k_gather(const float* __restrict__ x, int64_t C, int64_t st, int ndoy, int chunk,
                                                float* __restrict__ out, int ldspad) {
  extern __shared__ float pad[];
  typedef typename Vec<VEC>::T V;
  const int64_t c = ((int64_t)blockIdx.x * NT + threadIdx.x) * VEC;
  if (ldspad < 0) pad[threadIdx.x] = 0.f;
  const int d0 = blockIdx.y * chunk;
  int d1 = d0 + chunk;
  if (d1 > ndoy) d1 = ndoy;
  V buf[DEPTH][NY];
  float acc = 0.f;
  auto issue = [&](int slot, int d) {
#pragma unroll
    for (int y = 0; y < NY; ++y) {
      // wave-uniform row pointer + lane offset, as the kernels do it
      const float* rowp = x + (int64_t)(__builtin_amdgcn_readfirstlane(y * ndoy + d)) * st;
      buf[slot][y] = *reinterpret_cast<const V*>(rowp + c);
    }
  };
#pragma unroll
  for (int k = 0; k < DEPTH; ++k) issue(k, d0 + k < d1 ? d0 + k : d1 - 1);
  for (int d = d0; d < d1; d += DEPTH) {
#pragma unroll
    for (int k = 0; k < DEPTH; ++k) {
#pragma unroll
      for (int y = 0; y < NY; ++y) acc += vsum(buf[k][y]);
      const int dn = d + k + DEPTH;
      issue(k, dn < d1 ? dn : d1 - 1);
      if (SYNC) __syncthreads();
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

// the gather exactly as k_pdoy_quad does it: lane y holds the row address (from a table load one step ahead), 32 slots
// (30 years + 2 padding slots that read a constant row), buffer loads through a per-row resource
template <int PADS, bool TABLE>
__global__ void __launch_bounds__(64) k_gather_b(const float* __restrict__ x, int64_t C, int64_t st, int ndoy, int chunk,
                                                  float* __restrict__ out, const int32_t* __restrict__ tbase,
                                                  const float* __restrict__ padrow, int ldspad) {
  extern __shared__ float pad[];
  constexpr int NYP = 32, NY = 30;
  const int lane = threadIdx.x;
  if (ldspad < 0) pad[threadIdx.x] = 0.f;
  const uint32_t coff = (uint32_t)(blockIdx.x * 64 + lane) * 4u;
  const int d0 = blockIdx.y * chunk;
  int d1 = d0 + chunk;
  if (d1 > ndoy) d1 = ndoy;
  float raw[NYP];
  float acc = 0.f;
  auto fetch = [&](int d) -> int {
    const int dd = d < ndoy ? d : ndoy - 1;
    if (TABLE) return lane < NY ? tbase[lane * ndoy + dd] : -1;
    return lane < NY ? lane * ndoy + dd : -1;
  };
  auto resolve = [&](int tp, uint32_t& alo, uint32_t& ahi) {
    const float* p = tp < 0 ? padrow : x + (int64_t)tp * st;
    alo = (uint32_t)(uintptr_t)p;
    ahi = (uint32_t)((uintptr_t)p >> 32);
  };
  auto gather = [&](uint32_t alo, uint32_t ahi) {
#pragma unroll
    for (int y = 0; y < NY + PADS; ++y) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)alo, y), hi = (uint32_t)__builtin_amdgcn_readlane((int)ahi, y);
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, 0x7FFFFFFF, 0x00020000);
      raw[y] = __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)coff, 0, 0));
    }
  };
  uint32_t alo, ahi, nlo, nhi;
  resolve(fetch(d0), alo, ahi);
  gather(alo, ahi);
  resolve(fetch(d0 + 1), nlo, nhi);
  int tbv = fetch(d0 + 2);
  for (int d = d0; d < d1; ++d) {
#pragma unroll
    for (int y = 0; y < NY + PADS; ++y) acc += raw[y];
    uint32_t rlo, rhi;
    resolve(tbv, rlo, rhi);
    gather(nlo, nhi);
    nlo = rlo;
    nhi = rhi;
    tbv = fetch(d + 3);
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int PADS, bool TABLE>
static void run_b(const char* name, const float* d, int64_t C, int ndoy, int chunk, float* out, const int32_t* tb, const float* padrow,
                  int lds) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  dim3 grid((unsigned)(C / 64), (unsigned)((ndoy + chunk - 1) / chunk));
  auto k = k_gather_b<PADS, TABLE>;
  hipLaunchKernelGGL(k, grid, dim3(64), lds, 0, d, C, C, ndoy, chunk, out, tb, padrow, lds);
  hipEventRecord(a);
  const int reps = 3;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(k, grid, dim3(64), lds, 0, d, C, C, ndoy, chunk, out, tb, padrow, lds);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  ms /= reps;
  hipError_t e = hipGetLastError();
  printf("%-64s chunk %3d lds %6d  %7.3f ms  %5.0f GB/s%s\n", name, chunk, lds, ms, 30.0 * ndoy * (double)C * 4 / ms / 1e6,
         e == hipSuccess ? "" : hipGetErrorString(e));
  fflush(stdout);
}

template <int NY, int VEC, int NT, int DEPTH, bool SYNC>
static void run(const char* name, const float* d, int64_t C, int ndoy, int chunk, float* out, int lds) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  dim3 grid((unsigned)(C / ((int64_t)NT * VEC)), (unsigned)((ndoy + chunk - 1) / chunk));
  auto k = k_gather<NY, VEC, NT, DEPTH, SYNC>;
  if (lds > 64 * 1024) hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(k, grid, dim3(NT), lds, 0, d, C, C, ndoy, chunk, out, lds);
  hipEventRecord(a);
  const int reps = 3;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(k, grid, dim3(NT), lds, 0, d, C, C, ndoy, chunk, out, lds);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  ms /= reps;
  hipError_t e = hipGetLastError();
  printf("%-64s chunk %3d lds %6d  %7.3f ms  %5.0f GB/s%s\n", name, chunk, lds, ms, (double)NY * ndoy * ((double)grid.x * NT * VEC) * 4 / ms / 1e6,
         e == hipSuccess ? "" : hipGetErrorString(e));
  fflush(stdout);
}

int main() {
  const int NY = 30, ndoy = 365;
  const int64_t C = 1440 * 720;
  const int64_t T = NY * ndoy + 8;
  float *d, *out;
  if (hipMalloc((void**)&d, sizeof(float) * (size_t)T * C) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc((void**)&out, 64);
  hipMemset(d, 0, sizeof(float) * (size_t)T * C);
  hipDeviceSynchronize();
  {
    int32_t* tb;
    float* padrow;
    hipMalloc((void**)&tb, sizeof(int32_t) * 30 * ndoy);
    hipMalloc((void**)&padrow, sizeof(float) * C);
    hipMemset(padrow, 0, sizeof(float) * C);
    int32_t* h = (int32_t*)malloc(sizeof(int32_t) * 30 * ndoy);
    for (int i = 0; i < 30 * ndoy; ++i) h[i] = i;
    hipMemcpy(tb, h, sizeof(int32_t) * 30 * ndoy, hipMemcpyHostToDevice);
    const int L = 160 * 1024 / 12;
    run_b<0, false>("quad-style gather (readlane + buffer_load), 30 slots, no table", d, C, ndoy, 92, out, tb, padrow, L);
    run_b<2, false>("quad-style gather, 30 + 2 padding slots, no table", d, C, ndoy, 92, out, tb, padrow, L);
    run_b<0, true>("quad-style gather, 30 slots, table load one step ahead", d, C, ndoy, 92, out, tb, padrow, L);
    run_b<2, true>("quad-style gather, 30 + 2 slots, table load one step ahead", d, C, ndoy, 92, out, tb, padrow, L);
  }
  // occupancy caps through LDS: 160 KB / lds per workgroup = workgroups per CU (64 threads: = waves per CU)
  const int L12 = 160 * 1024 / 12, L8 = 160 * 1024 / 8, L16 = 160 * 1024 / 16, L24 = 160 * 1024 / 24, L32 = 160 * 1024 / 32;
  run<30, 1, 64, 1, false>("dword, 1 wave/wg, 1 day-set in flight, 8 waves/CU", d, C, ndoy, 92, out, L8);
  run<30, 1, 64, 1, false>("dword, 1 wave/wg, 1 day-set in flight, 12 waves/CU", d, C, ndoy, 92, out, L12);
  run<30, 1, 64, 1, false>("dword, 1 wave/wg, 1 day-set in flight, 16 waves/CU", d, C, ndoy, 92, out, L16);
  run<30, 1, 64, 1, false>("dword, 1 wave/wg, 1 day-set in flight, 24 waves/CU", d, C, ndoy, 92, out, L24);
  run<30, 1, 64, 1, false>("dword, 1 wave/wg, 1 day-set in flight, 32 waves/CU", d, C, ndoy, 92, out, L32);
  run<30, 1, 64, 2, false>("dword, 1 wave/wg, 2 day-sets in flight, 12 waves/CU", d, C, ndoy, 92, out, L12);
  run<30, 1, 64, 4, false>("dword, 1 wave/wg, 4 day-sets in flight, 12 waves/CU", d, C, ndoy, 92, out, L12);
  run<30, 1, 64, 1, false>("dword, 1 wave/wg, 1 day-set, 12 waves/CU, chunk 24", d, C, ndoy, 24, out, L12);
  run<30, 1, 64, 1, false>("dword, 1 wave/wg, 1 day-set, 12 waves/CU, chunk 365", d, C, ndoy, 365, out, L12);
  run<30, 2, 64, 1, false>("dwordx2 (512 B / row), 1 day-set, 12 waves/CU", d, C, ndoy, 92, out, L12);
  run<30, 4, 64, 1, false>("dwordx4 (1 KB / row), 1 day-set, 12 waves/CU", d, C, ndoy, 92, out, L12);
  run<30, 4, 64, 1, false>("dwordx4 (1 KB / row), 1 day-set, 8 waves/CU", d, C, ndoy, 92, out, L8);
  run<30, 1, 256, 1, true>("dword, 4 waves/wg in lock-step (1 KB / row / wg), 3 wg/CU", d, C, ndoy, 92, out, 160 * 1024 / 3);
  run<30, 1, 256, 1, false>("dword, 4 waves/wg free-running, 3 wg/CU", d, C, ndoy, 92, out, 160 * 1024 / 3);
  run<30, 1, 256, 2, true>("dword, 4 waves/wg in lock-step, 2 day-sets, 3 wg/CU", d, C, ndoy, 92, out, 160 * 1024 / 3);
  run<30, 1, 512, 1, true>("dword, 8 waves/wg in lock-step (2 KB / row / wg), 2 wg/CU", d, C, ndoy, 92, out, 160 * 1024 / 2);
  run<30, 1, 1024, 1, true>("dword, 16 waves/wg in lock-step (4 KB / row / wg), 1 wg/CU", d, C, ndoy, 92, out, 100 * 1024);
  return 0;
}
