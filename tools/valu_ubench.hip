// valu_ubench.hip — issue cost of the VALU instructions the sorting-network kernels are made of, on gfx950.
// Each kernel runs ITER x 64 independent instructions of one opcode per wave (8 accumulators, no memory traffic);
// cycles per wave-instruction = wall cycles * waves_per_SIMD / instructions per wave.
//   hipcc --offload-arch=gfx950 -O3 tools/valu_ubench.hip -o tools/valu_ubench && tools/valu_ubench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define ITER 4096
#define OPS8(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)

#define KERNEL(name, body)                                                            \
  __global__ void __launch_bounds__(256) name(uint32_t* out, uint32_t seed) {         \
    uint32_t a0 = threadIdx.x ^ seed, a1 = a0 * 3u, a2 = a0 * 5u, a3 = a0 * 7u, a4 = a0 * 11u, a5 = a0 * 13u, a6 = a0 * 17u, a7 = a0 * 19u; \
    uint32_t b = seed * 2654435761u + threadIdx.x;                                    \
    for (int it = 0; it < ITER; ++it) {                                               \
      _Pragma("unroll") for (int r = 0; r < 8; ++r) { body }                          \
    }                                                                                 \
    out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;      \
  }
#define A1(op) asm volatile(op " %0, %0, %1" : "+v"(a0) : "v"(b)); asm volatile(op " %0, %0, %1" : "+v"(a1) : "v"(b)); \
  asm volatile(op " %0, %0, %1" : "+v"(a2) : "v"(b)); asm volatile(op " %0, %0, %1" : "+v"(a3) : "v"(b));               \
  asm volatile(op " %0, %0, %1" : "+v"(a4) : "v"(b)); asm volatile(op " %0, %0, %1" : "+v"(a5) : "v"(b));               \
  asm volatile(op " %0, %0, %1" : "+v"(a6) : "v"(b)); asm volatile(op " %0, %0, %1" : "+v"(a7) : "v"(b));
#define A3(op) asm volatile(op " %0, %0, %1, %1" : "+v"(a0) : "v"(b)); asm volatile(op " %0, %0, %1, %1" : "+v"(a1) : "v"(b)); \
  asm volatile(op " %0, %0, %1, %1" : "+v"(a2) : "v"(b)); asm volatile(op " %0, %0, %1, %1" : "+v"(a3) : "v"(b));               \
  asm volatile(op " %0, %0, %1, %1" : "+v"(a4) : "v"(b)); asm volatile(op " %0, %0, %1, %1" : "+v"(a5) : "v"(b));               \
  asm volatile(op " %0, %0, %1, %1" : "+v"(a6) : "v"(b)); asm volatile(op " %0, %0, %1, %1" : "+v"(a7) : "v"(b));

#define A1M(op) asm volatile(op " %0, %1" : "+v"(a0) : "v"(b)); asm volatile(op " %0, %1" : "+v"(a1) : "v"(b)); asm volatile(op " %0, %1" : "+v"(a2) : "v"(b)); \
  asm volatile(op " %0, %1" : "+v"(a3) : "v"(b)); asm volatile(op " %0, %1" : "+v"(a4) : "v"(b)); asm volatile(op " %0, %1" : "+v"(a5) : "v"(b)); \
  asm volatile(op " %0, %1" : "+v"(a6) : "v"(b)); asm volatile(op " %0, %1" : "+v"(a7) : "v"(b));
#define ADDF64 { double d0 = __hiloint2double(a0, a1), d1 = __hiloint2double(a2, a3), d2 = __hiloint2double(a4, a5), d3 = __hiloint2double(a6, a7), e = (double)b; \
  asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(e)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d1) : "v"(e)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d2) : "v"(e)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d3) : "v"(e)); \
  asm volatile("v_add_f64 %0, %0, %1" : "+v"(d0) : "v"(e)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d1) : "v"(e)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d2) : "v"(e)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(d3) : "v"(e)); \
  a0 = __double2hiint(d0); a1 = __double2loint(d0); a2 = __double2hiint(d1); a3 = __double2loint(d1); a4 = __double2hiint(d2); a5 = __double2loint(d2); a6 = __double2hiint(d3); a7 = __double2loint(d3); }
KERNEL(k_min_u32, A1("v_min_u32"))
KERNEL(k_max_u32, A1("v_max_u32"))
KERNEL(k_min_i32, A1("v_min_i32"))
KERNEL(k_min_f32, A1("v_min_f32"))
KERNEL(k_max_f32, A1("v_max_f32"))
KERNEL(k_add_u32, A1("v_add_u32"))
KERNEL(k_xor_b32, A1("v_xor_b32"))
KERNEL(k_and_b32, A1("v_and_b32"))
KERNEL(k_add_f32, A1("v_add_f32"))
KERNEL(k_mul_f32, A1("v_mul_f32"))
KERNEL(k_cndmask, A1("v_cndmask_b32"))
KERNEL(k_min_u16, A1("v_min_u16"))
KERNEL(k_pk_min_u16, A1("v_pk_min_u16"))
KERNEL(k_pk_max_i16, A1("v_pk_max_i16"))
KERNEL(k_fma_f32, A3("v_fma_f32"))
KERNEL(k_min3_f32, A3("v_min3_f32"))
KERNEL(k_med3_f32, A3("v_med3_f32"))
KERNEL(k_min3_u32, A3("v_min3_u32"))
KERNEL(k_med3_u32, A3("v_med3_u32"))
KERNEL(k_lshl_add, A3("v_lshl_add_u32"))
KERNEL(k_bfi, A3("v_bfi_b32"))
KERNEL(k_perm, A3("v_perm_b32"))

// compare + select forms: one v_cmp (writes VCC) feeding TWO v_cndmask = a compare-exchange without v_min / v_max
#define CE_CMP(ai, aj) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %2, %1, %0, vcc\n\tv_cndmask_b32 %1, %0, %1, vcc\n\tv_mov_b32 %0, %2" : "+v"(ai), "+v"(aj), "=&v"(t));
KERNEL(k_ce_cmp, uint32_t t; CE_CMP(a0, a1) CE_CMP(a2, a3) CE_CMP(a4, a5) CE_CMP(a6, a7) CE_CMP(a0, a2) CE_CMP(a1, a3) CE_CMP(a4, a6) CE_CMP(a5, a7))
#define CE_MM(ai, aj) asm volatile("v_min_u32 %2, %0, %1\n\tv_max_u32 %1, %0, %1\n\tv_mov_b32 %0, %2" : "+v"(ai), "+v"(aj), "=&v"(t));
KERNEL(k_ce_minmax, uint32_t t; CE_MM(a0, a1) CE_MM(a2, a3) CE_MM(a4, a5) CE_MM(a6, a7) CE_MM(a0, a2) CE_MM(a1, a3) CE_MM(a4, a6) CE_MM(a5, a7))
// compare-exchange from the borrow of a subtraction: v_sub_co_u32 (is it in the fast class?) + two v_cndmask
#define CE_SUB(ai, aj) asm volatile("v_sub_co_u32 %2, vcc, %0, %1\n\tv_cndmask_b32 %2, %1, %0, vcc\n\tv_cndmask_b32 %1, %0, %1, vcc\n\tv_mov_b32 %0, %2" : "+v"(ai), "+v"(aj), "=&v"(t) : : "vcc");
KERNEL(k_ce_subco, uint32_t t; CE_SUB(a0, a1) CE_SUB(a2, a3) CE_SUB(a4, a5) CE_SUB(a6, a7) CE_SUB(a0, a2) CE_SUB(a1, a3) CE_SUB(a4, a6) CE_SUB(a5, a7))
#define SUBCO(ai) asm volatile("v_sub_co_u32 %0, vcc, %0, %1" : "+v"(ai) : "v"(b) : "vcc");
KERNEL(k_sub_co, SUBCO(a0) SUBCO(a1) SUBCO(a2) SUBCO(a3) SUBCO(a4) SUBCO(a5) SUBCO(a6) SUBCO(a7))
KERNEL(k_cvt_u32_f32, A1M("v_cvt_u32_f32"))
#define CMP1(ai) asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(ai), "v"(b) : "vcc");
KERNEL(k_cmp_u32, CMP1(a0) CMP1(a1) CMP1(a2) CMP1(a3) CMP1(a4) CMP1(a5) CMP1(a6) CMP1(a7))
#define CMPF(ai) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(ai), "v"(b) : "vcc");
KERNEL(k_cmp_f32, CMPF(a0) CMPF(a1) CMPF(a2) CMPF(a3) CMPF(a4) CMPF(a5) CMPF(a6) CMPF(a7))
#define CND1(ai) asm volatile("v_cmp_lt_u32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(ai) : "v"(b) : "vcc");
KERNEL(k_cmp_cnd, CND1(a0) CND1(a1) CND1(a2) CND1(a3) CND1(a4) CND1(a5) CND1(a6) CND1(a7))
KERNEL(k_mov, A1M("v_mov_b32"))
KERNEL(k_ashr, A1("v_ashrrev_i32"))
KERNEL(k_or, A1("v_or_b32"))
KERNEL(k_sub, A1("v_sub_u32"))
KERNEL(k_max_u16, A1("v_max_u16"))
KERNEL(k_mul_lo, A1("v_mul_lo_u32"))
KERNEL(k_mul_u24, A1("v_mul_u32_u24"))
KERNEL(k_add_f64, uint32_t dummy = 0; (void)dummy; ADDF64)

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Ent { const char* name; kern_t k; };

int main() {
  uint32_t* out;
  hipMalloc(&out, 1 << 24);
  hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
  const int ncu = p.multiProcessorCount;
  Ent ents[] = {{"v_min_u32", k_min_u32}, {"v_max_u32", k_max_u32}, {"v_min_i32", k_min_i32}, {"v_min_f32", k_min_f32}, {"v_max_f32", k_max_f32},
                {"v_add_u32", k_add_u32}, {"v_xor_b32", k_xor_b32}, {"v_and_b32", k_and_b32}, {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32},
                {"v_cndmask_b32", k_cndmask}, {"v_min_u16", k_min_u16}, {"v_pk_min_u16", k_pk_min_u16}, {"v_pk_max_i16", k_pk_max_i16},
                {"v_fma_f32", k_fma_f32}, {"v_min3_f32", k_min3_f32}, {"v_med3_f32", k_med3_f32}, {"v_min3_u32", k_min3_u32},
                {"v_med3_u32", k_med3_u32}, {"v_lshl_add_u32", k_lshl_add}, {"v_bfi_b32", k_bfi}, {"v_perm_b32", k_perm},
                {"v_cmp_lt_u32", k_cmp_u32}, {"v_cmp_lt_f32", k_cmp_f32}, {"cmp+cndmask (2 inst)", k_cmp_cnd}, {"CE cmp+2cnd+mov (4)", k_ce_cmp},
                {"CE min+max+mov (3)", k_ce_minmax}, {"CE sub_co+2cnd+mov (4)", k_ce_subco}, {"v_sub_co_u32", k_sub_co}, {"v_cvt_u32_f32", k_cvt_u32_f32}, {"v_mov_b32", k_mov}, {"v_ashrrev_i32", k_ashr}, {"v_or_b32", k_or}, {"v_sub_u32", k_sub},
                {"v_max_u16", k_max_u16}, {"v_mul_lo_u32", k_mul_lo}, {"v_mul_u32_u24", k_mul_u24}, {"v_add_f64 (+cvt)", k_add_f64}};
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  int clk = 0; hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
  printf("device %s, %d CUs, clock attr %d kHz\n", p.gcnArchName, ncu, clk);
  for (int wps = 1; wps <= 2; ++wps) {   // waves per SIMD: one or two 256-thread workgroups per CU
    for (auto& e : ents) {
      dim3 grid(ncu * wps);
      hipLaunchKernelGGL(e.k, grid, dim3(256), 0, 0, out, 1u);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(e.k, grid, dim3(256), 0, 0, out, 1u + r);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
      const double insts = (double)ITER * 64;         // per wave
      const double ns_per_inst = ms * 1e6 / (insts * wps);  // SIMD time per wave-instruction
      printf("waves/SIMD %d  %-22s %8.3f ms   %6.2f ns per wave-instruction per SIMD  (= %5.2f cycles at 2.4 GHz, %5.2f at 2.0)\n", wps, e.name, ms,
             ns_per_inst, ns_per_inst * 2.4, ns_per_inst * 2.0);
    }
  }
  return 0;
}
