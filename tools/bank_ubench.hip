// bank_ubench.hip — does the VGPR bank of the two sources of v_min_u32 / v_max_u32 / v_add_u32 change the issue cost?
// Explicit registers: 8 independent instructions per block, sources (v[20+i], v[20+i+D]) for D = 4 (same bank, index mod 4)
// or D = 5 / 6 / 7 (different banks), destinations v[60+i].
//   hipcc --offload-arch=gfx950 -O3 tools/bank_ubench.hip -o tools/bank_ubench && tools/bank_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITER 8192
#define STR2(x) #x
#define STR(x) STR2(x)
#define ONE(OP, d, a, b) OP " v" STR(d) ", v" STR(a) ", v" STR(b) "\n\t"
#define CLOB "v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39", \
             "v60","v61","v62","v63","v64","v65","v66","v67"
#define KERNEL(name, ASM)                                                       \
  __global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) {   \
    for (int it = 0; it < ITER; ++it) asm volatile(ASM ::: CLOB);               \
    uint32_t r;                                                                 \
    asm volatile("v_mov_b32 %0, v60" : "=v"(r));                                \
    out[blockIdx.x * 256 + threadIdx.x] = r + seed;                             \
  }
// (asm strings generated: the preprocessor cannot add register numbers)
KERNEL(k_min_d4, "v_min_u32 v60, v20, v24\n\tv_min_u32 v61, v21, v25\n\tv_min_u32 v62, v22, v26\n\tv_min_u32 v63, v23, v27\n\tv_min_u32 v64, v24, v28\n\tv_min_u32 v65, v25, v29\n\tv_min_u32 v66, v26, v30\n\tv_min_u32 v67, v27, v31\n\t")
KERNEL(k_min_d5, "v_min_u32 v60, v20, v25\n\tv_min_u32 v61, v21, v26\n\tv_min_u32 v62, v22, v27\n\tv_min_u32 v63, v23, v28\n\tv_min_u32 v64, v24, v29\n\tv_min_u32 v65, v25, v30\n\tv_min_u32 v66, v26, v31\n\tv_min_u32 v67, v27, v32\n\t")
KERNEL(k_min_d6, "v_min_u32 v60, v20, v26\n\tv_min_u32 v61, v21, v27\n\tv_min_u32 v62, v22, v28\n\tv_min_u32 v63, v23, v29\n\tv_min_u32 v64, v24, v30\n\tv_min_u32 v65, v25, v31\n\tv_min_u32 v66, v26, v32\n\tv_min_u32 v67, v27, v33\n\t")
KERNEL(k_min_d8, "v_min_u32 v60, v20, v28\n\tv_min_u32 v61, v21, v29\n\tv_min_u32 v62, v22, v30\n\tv_min_u32 v63, v23, v31\n\tv_min_u32 v64, v24, v32\n\tv_min_u32 v65, v25, v33\n\tv_min_u32 v66, v26, v34\n\tv_min_u32 v67, v27, v35\n\t")
KERNEL(k_add_d4, "v_add_u32 v60, v20, v24\n\tv_add_u32 v61, v21, v25\n\tv_add_u32 v62, v22, v26\n\tv_add_u32 v63, v23, v27\n\tv_add_u32 v64, v24, v28\n\tv_add_u32 v65, v25, v29\n\tv_add_u32 v66, v26, v30\n\tv_add_u32 v67, v27, v31\n\t")
KERNEL(k_add_d5, "v_add_u32 v60, v20, v25\n\tv_add_u32 v61, v21, v26\n\tv_add_u32 v62, v22, v27\n\tv_add_u32 v63, v23, v28\n\tv_add_u32 v64, v24, v29\n\tv_add_u32 v65, v25, v30\n\tv_add_u32 v66, v26, v31\n\tv_add_u32 v67, v27, v32\n\t")
KERNEL(k_minip_d4, "v_min_u32 v20, v20, v24\n\tv_min_u32 v21, v21, v25\n\tv_min_u32 v22, v22, v26\n\tv_min_u32 v23, v23, v27\n\tv_min_u32 v32, v32, v36\n\tv_min_u32 v33, v33, v37\n\tv_min_u32 v34, v34, v38\n\tv_min_u32 v35, v35, v39\n\t")
KERNEL(k_minip_d5, "v_min_u32 v20, v20, v25\n\tv_min_u32 v21, v21, v26\n\tv_min_u32 v22, v22, v27\n\tv_min_u32 v23, v23, v28\n\tv_min_u32 v32, v32, v37\n\tv_min_u32 v33, v33, v38\n\tv_min_u32 v34, v34, v39\n\tv_min_u32 v35, v35, v40\n\t")
KERNEL(k_minf_d4, "v_min_f32 v60, v20, v24\n\tv_min_f32 v61, v21, v25\n\tv_min_f32 v62, v22, v26\n\tv_min_f32 v63, v23, v27\n\tv_min_f32 v64, v24, v28\n\tv_min_f32 v65, v25, v29\n\tv_min_f32 v66, v26, v30\n\tv_min_f32 v67, v27, v31\n\t")
KERNEL(k_minf_d5, "v_min_f32 v60, v20, v25\n\tv_min_f32 v61, v21, v26\n\tv_min_f32 v62, v22, v27\n\tv_min_f32 v63, v23, v28\n\tv_min_f32 v64, v24, v29\n\tv_min_f32 v65, v25, v30\n\tv_min_f32 v66, v26, v31\n\tv_min_f32 v67, v27, v32\n\t")

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Ent { const char* name; kern_t k; };
int main() {
  uint32_t* out; hipMalloc(&out, 1 << 24);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  Ent ents[] = {{"v_min_u32 src banks equal (D=4)", k_min_d4}, {"v_min_u32 D=5", k_min_d5}, {"v_min_u32 D=6", k_min_d6}, {"v_min_u32 D=8", k_min_d8},
                {"v_add_u32 D=4", k_add_d4}, {"v_add_u32 D=5", k_add_d5}, {"v_min_u32 in place D=4", k_minip_d4}, {"v_min_u32 in place D=5", k_minip_d5},
                {"v_min_f32 D=4", k_minf_d4}, {"v_min_f32 D=5", k_minf_d5}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wps = 1; wps <= 4; wps *= 2)
    for (auto& e : ents) {
      dim3 grid(ncu * wps);
      hipLaunchKernelGGL(e.k, grid, dim3(256), 0, 0, out, 1u); hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(e.k, grid, dim3(256), 0, 0, out, 1u + r);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      const double ns = ms * 1e6 / ((double)ITER * 8 * wps);
      printf("waves/SIMD %d  %-34s %8.3f ms  %5.2f cycles per wave-instruction at 2.4 GHz\n", wps, e.name, ms, ns * 2.4);
    }
  return 0;
}
