// regsort_ubench.hip — the load pattern of k_select_regsort / k_qdm_regsort (select3.hip, qdm2.hip) without their arithmetic,
// to calibrate rocprofv3's FETCH_SIZE on it (the T = 365 EQM leg carried no traffic figure because the x2
// correction of the streaming kernels had not been checked for this pattern).  A wave owns 32 adjacent columns, two lanes
// per column: lane A reads rows 0 .. N-1, lane B rows T-N .. T-1 (N = 183, T = 365: one row is read by both), 4 bytes per
// load, 183 loads in flight per lane, row stride C floats.  Bytes read by construction: 2 N C 4.
//   hipcc --offload-arch=gfx950 -O3 tools/regsort_ubench.hip -o tools/regsort_ubench
//   rocprofv3 --pmc FETCH_SIZE -- tools/regsort_ubench      (FETCH_SIZE is in KiB)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int N = 183;

__global__ void __launch_bounds__(256, 2) k_regsort_loads(const float* __restrict__ x, int T, int64_t C, float* __restrict__ out) {
  const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
  const int64_t ntiles = (C + 31) / 32, nwaves = (int64_t)gridDim.x * 4;
  const uint32_t strideB = (uint32_t)(C * 4);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)0xFFFFFFFFu, 0x00020000);
  float acc = 0.f;
  for (int64_t tile = (int64_t)blockIdx.x * 4 + w; tile < ntiles; tile += nwaves) {
    const uint32_t h = lane & 1u, c32 = lane >> 1;
    int64_t col = tile * 32 + c32;
    col = col < C ? col : C - 1;
    const uint32_t voff = (uint32_t)(col * 4) + (h ? (uint32_t)(T - N) * strideB : 0u);
    float k[N];
    uint32_t soff = 0u;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      k[i] = __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
      soff += strideB;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) acc += k[i];
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const int T = 365;
  const int64_t C = 1440 * 720;
  float *x, *out;
  CHECK(hipMalloc(&x, (size_t)T * C * 4));
  CHECK(hipMalloc(&out, 256));
  CHECK(hipMemset(x, 0, (size_t)T * C * 4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int rep = 0; rep < 5; ++rep) {
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(k_regsort_loads, dim3(512), dim3(256), 0, 0, x, T, C, out);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    printf("k_regsort_loads: %.4f ms, %.3f GB by construction (2 x 183 rows x C x 4), %.0f GB/s\n", ms, 2.0 * N * C * 4 / 1e9, 2.0 * N * C * 4 / ms * 1e-6);
  }
  return 0;
}
