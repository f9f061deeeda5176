// hbm_ubench.hip — achievable HBM bandwidth ceilings on this MI355X for the access mixes the kernels use:
// read-only (16 B/lane), write-only (16 B/lane), copy 1:1, read 4 B : write 8 B per element (percentile_doy shape),
// and the time-marching row pattern (each lane walks T rows of a (T, C) array).  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_read(const float4* __restrict__ a, size_t n, float* sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  float s = 0.f;
  for (; i < n; i += stride) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 123.456f) *sink = s;
}
__global__ void k_write(float4* __restrict__ a, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) a[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) b[i] = a[i];
}
// read float4 (4 cells), write 4 doubles
__global__ void k_r4w8(const float4* __restrict__ a, double2* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    float4 v = a[i];
    b[2 * i] = make_double2((double)v.x, (double)v.y);
    b[2 * i + 1] = make_double2((double)v.z, (double)v.w);
  }
}
// time-marching: lane owns 4 cells, walks rows t0..t1 (chunked over blockIdx.y), unroll U
template <int U>
__global__ void k_march(const float* __restrict__ x, int64_t T, int64_t C, float* sink) {
  int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (c >= C) return;
  int64_t chunk = (T + gridDim.y - 1) / gridDim.y, t0 = blockIdx.y * chunk, t1 = t0 + chunk;
  if (t1 > T) t1 = T;
  float s = 0.f;
  int64_t t = t0;
  for (; t + U <= t1; t += U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = *reinterpret_cast<const float4*>(x + (t + u) * C + c);
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  for (; t < t1; ++t) { float4 v = *reinterpret_cast<const float4*>(x + t * C + c); s += v.x + v.y + v.z + v.w; }
  if (s == 123.456f) *sink = s;
}

template <typename F>
float timeit(F f, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const int64_t T = 365, C = 1440 * 720;
  size_t nf = (size_t)T * C;  // floats
  float *a, *sink;
  double* b;
  CK(hipMalloc(&a, nf * 4)); CK(hipMalloc(&b, nf * 8)); CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 0, nf * 4)); CK(hipMemset(b, 0, nf * 8));
  size_t n4 = nf / 4;
  for (int blocks : {2048, 8192, 32768}) {
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, (const float4*)a, n4, sink); }, 10);
    printf("read   blocks=%6d  %.3f ms  %.0f GB/s\n", blocks, ms, nf * 4 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, (float4*)b, n4); }, 10);
    printf("write  blocks=%6d  %.3f ms  %.0f GB/s\n", blocks, ms, nf * 4 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, (const float4*)a, (float4*)b, n4); }, 10);
    printf("copy   blocks=%6d  %.3f ms  %.0f GB/s (r+w)\n", blocks, ms, nf * 8 / ms / 1e6);
    ms = timeit([&] { hipLaunchKernelGGL(k_r4w8, dim3(blocks), dim3(256), 0, 0, (const float4*)a, (double2*)b, n4); }, 10);
    printf("r4w8   blocks=%6d  %.3f ms  %.0f GB/s (r+w)\n", blocks, ms, nf * 12 / ms / 1e6);
  }
  for (int gy : {1, 2, 4, 8, 16}) {
    dim3 grid((unsigned)((C / 4 + 255) / 256), gy);
    float ms = timeit([&] { hipLaunchKernelGGL((k_march<4>), grid, dim3(256), 0, 0, (const float*)a, T, C, sink); }, 10);
    float ms8 = timeit([&] { hipLaunchKernelGGL((k_march<8>), grid, dim3(256), 0, 0, (const float*)a, T, C, sink); }, 10);
    float ms16 = timeit([&] { hipLaunchKernelGGL((k_march<16>), grid, dim3(256), 0, 0, (const float*)a, T, C, sink); }, 10);
    printf("march  gy=%2d  U4 %.3f ms %.0f GB/s | U8 %.3f ms %.0f GB/s | U16 %.3f ms %.0f GB/s\n", gy, ms, nf * 4 / ms / 1e6,
           ms8, nf * 4 / ms8 / 1e6, ms16, nf * 4 / ms16 / 1e6);
  }
  return 0;
}
