// colread_ubench.hip — how fast can workgroups stream whole columns (44 KB contiguous each) of a 532 MB buffer?
// The access pattern of k_select_lean phase A without any of its arithmetic.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>

template <int NT, int KPL, int VEC>
__global__ void __launch_bounds__(NT) k_colread(const float* __restrict__ x, int64_t ncols, int64_t cs, float* __restrict__ out) {
  float acc = 0.f;
  for (int64_t col = blockIdx.x; col < ncols; col += gridDim.x) {
    const float* xc = x + col * cs;
    if (VEC == 1) {
      float v[KPL];
#pragma unroll
      for (int k = 0; k < KPL; ++k) v[k] = xc[threadIdx.x + k * NT];
#pragma unroll
      for (int k = 0; k < KPL; ++k) acc += v[k];
    } else {
      float4 v[KPL / 4];
#pragma unroll
      for (int k = 0; k < KPL / 4; ++k) v[k] = *reinterpret_cast<const float4*>(xc + 4 * (threadIdx.x + k * NT));
#pragma unroll
      for (int k = 0; k < KPL / 4; ++k) acc += v[k].x + v[k].y + v[k].z + v[k].w;
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int NT, int KPL, int VEC>
static void run(const char* name, const float* d, int64_t ncols, int64_t cs, float* out, int grid) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k_colread<NT, KPL, VEC>), dim3(grid), dim3(NT), 0, 0, d, ncols, cs, out);
  hipEventRecord(a);
  const int reps = 5;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k_colread<NT, KPL, VEC>), dim3(grid), dim3(NT), 0, 0, d, ncols, cs, out);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
  printf("%-40s grid %5d  %.3f ms  %.0f GB/s\n", name, grid, ms, (double)ncols * KPL * NT * 4 / ms / 1e6);
}

int main() {
  const int64_t cs = 11072, ncols = 12096;
  float *d, *out;
  hipMalloc((void**)&d, sizeof(float) * cs * (ncols + 2));
  hipMalloc((void**)&out, 64);
  hipMemset(d, 0, sizeof(float) * cs * (ncols + 2));
  run<512, 20, 1>("512 thr, 20 dword loads", d, ncols, cs, out, 512);
  run<512, 20, 1>("512 thr, 20 dword loads", d, ncols, cs, out, 1024);
  run<512, 20, 1>("512 thr, 20 dword loads", d, ncols, cs, out, 4096);
  run<512, 20, 4>("512 thr, 5 dwordx4 loads", d, ncols, cs, out, 512);
  run<512, 20, 4>("512 thr, 5 dwordx4 loads", d, ncols, cs, out, 1024);
  run<512, 20, 4>("512 thr, 5 dwordx4 loads", d, ncols, cs, out, 4096);
  run<256, 40, 1>("256 thr, 40 dword loads", d, ncols, cs, out, 1024);
  run<256, 40, 1>("256 thr, 40 dword loads", d, ncols, cs, out, 4096);
  run<256, 40, 4>("256 thr, 10 dwordx4 loads", d, ncols, cs, out, 2048);
  run<1024, 8, 1>("1024 thr, 8 dword loads (quarter column)", d, ncols, cs, out, 2048);
  return 0;
}
