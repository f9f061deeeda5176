// dpp_test.hip — which DPP controls give lane ^ m inside a wave64 on gfx950 (used by hs_lane_xor in select4.hip)
//   hipcc --offload-arch=gfx950 -O3 tools/dpp_test.hip -o tools/dpp_test && tools/dpp_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ uint32_t lane_xor(uint32_t v, int m) {
  switch (m) {
    case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2,3,0,1]
    case 4: {
      const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);          // row_shl:4 into banks 0, 2
      return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);       // row_shr:4 into banks 1, 3
    }
    case 8: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);  // row_ror:8
    default: return (uint32_t)__shfl_xor((int)v, m);
  }
}

__global__ void k(uint32_t* out) {
  const uint32_t lane = threadIdx.x, v = lane * 7u + 3u;
  out[0 * 64 + lane] = lane_xor(v, 1);
  out[1 * 64 + lane] = lane_xor(v, 2);
  out[2 * 64 + lane] = lane_xor(v, 4);
  out[3 * 64 + lane] = lane_xor(v, 8);
  out[4 * 64 + lane] = lane_xor(v, 16);
  out[5 * 64 + lane] = lane_xor(v, 32);
}

int main() {
  uint32_t* d;
  hipMalloc(&d, 6 * 64 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  uint32_t h[6 * 64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const int ms[6] = {1, 2, 4, 8, 16, 32};
  int bad = 0;
  for (int i = 0; i < 6; ++i) {
    int nb = 0;
    for (int l = 0; l < 64; ++l) nb += h[i * 64 + l] != (uint32_t)((l ^ ms[i]) * 7 + 3);
    printf("xor %2d: %s (%d wrong lanes); lane 0..7 got from lanes:", ms[i], nb ? "WRONG" : "ok", nb);
    for (int l = 0; l < 8; ++l) printf(" %d", (int)(h[i * 64 + l] - 3) / 7);
    printf("\n");
    bad += nb;
  }
  return bad ? 1 : 0;
}
