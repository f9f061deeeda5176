// LDS histogram micro-benchmark for the select kernels' design ("quantile_series"): how fast can a CU
// count keys into LDS bins?  Build: hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_ubench.hip -o tools/lds_atomic_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// MODE 0: shared histogram, atomicAdd (ds_add_u32)      MODE 1: lane-private slots [bin][lane], plain read-add-write
// MODE 2: lane-private slots, atomicAdd                 MODE 3: shared histogram, atomicAdd with the result used (ds_add_rtn)
// MODE 4: no LDS at all (hash only: the floor)
template <int MODE, int NB>
__global__ void k(const uint32_t* __restrict__ seed, uint32_t* __restrict__ out, int iters) {
  extern __shared__ uint32_t h[];
  const int n = MODE == 1 || MODE == 2 ? NB * 64 : NB;
  for (int i = threadIdx.x; i < n; i += blockDim.x) h[i] = 0;
  __syncthreads();
  uint32_t s = seed[0] + blockIdx.x * 7919u + threadIdx.x * 104729u, acc = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < iters; ++it) {
    s = mix(s + it);
    uint32_t b = s & (NB - 1);
    if (MODE == 0) atomicAdd(&h[b], 1u);
    if (MODE == 1) { uint32_t* p = &h[b * 64 + lane]; *p = *p + 1; (void)wave; }
    if (MODE == 2) atomicAdd(&h[b * 64 + lane], 1u);
    if (MODE == 3) acc += atomicAdd(&h[b], 1u);
    if (MODE == 4) acc += b;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) acc += h[i];
  if (acc == 0xdeadbeefu) out[0] = acc;
}

template <int MODE, int NB>
static void run(const char* name, int threads, uint32_t* seed, uint32_t* out) {
  const int iters = 4096;
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int blocks = p.multiProcessorCount * (threads == 64 ? 8 : 2);
  size_t lds = sizeof(uint32_t) * (size_t)((MODE == 1 || MODE == 2) ? NB * 64 : NB);
  hipFuncSetAttribute((const void*)k<MODE, NB>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  k<MODE, NB><<<blocks, threads, lds>>>(seed, out, iters);
  hipEventRecord(a);
  k<MODE, NB><<<blocks, threads, lds>>>(seed, out, iters);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ops = (double)blocks * threads * iters;
  double clk = ms * 1e-3 * p.clockRate * 1e3;  // clockRate in kHz
  printf("%-44s threads=%4d bins=%5d lds=%6zu B : %8.3f ms  %6.2f lane-ops/clk/CU  (%s)\n", name, threads, NB, lds, ms,
         ops / clk / p.multiProcessorCount, hipGetErrorString(hipGetLastError()));
}

int main() {
  uint32_t *seed, *out;
  hipMalloc(&seed, 4); hipMalloc(&out, 4); hipMemset(seed, 1, 4);
  run<4, 2048>("hash only (floor)", 64, seed, out);
  run<4, 2048>("hash only (floor)", 1024, seed, out);
  run<0, 2048>("shared hist, ds_add_u32", 64, seed, out);
  run<0, 2048>("shared hist, ds_add_u32", 1024, seed, out);
  run<0, 64>("shared hist, ds_add_u32", 64, seed, out);
  run<3, 2048>("shared hist, ds_add_rtn_u32", 64, seed, out);
  run<3, 2048>("shared hist, ds_add_rtn_u32", 1024, seed, out);
  run<1, 64>("lane-private [bin][lane], read-add-write", 64, seed, out);
  run<2, 64>("lane-private [bin][lane], ds_add_u32", 64, seed, out);
  run<1, 256>("lane-private [bin][lane], read-add-write", 64, seed, out);
  return 0;
}
