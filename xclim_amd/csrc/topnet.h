// topnet.h — register comparator networks for "the 16 outermost of ..." (pdoy_top.hip, pdoy_quad.hip).
// The networks run on the floats themselves with v_max_f32 / v_min_f32.  BOT = false keeps the 16 LARGEST, ordered
// from the largest down; BOT = true keeps the 16 SMALLEST, ordered from the smallest up: the same wiring with the two
// instructions swapped, so that low percentiles need no mirrored copy of the data.  "Outer" below = larger (BOT =
// false) / smaller (BOT = true).  NaN never enters: the callers replace it by the INNERMOST value (-inf / +inf) and
// count it, so the IEEE-mode quieting rules of v_max / v_min do not apply.  Inline assembly because fmaxf() makes the
// compiler canonicalise every input first (one extra v_max per operand of the first layer).
#pragma once
#include "common.h"

template <bool BOT>
__device__ __forceinline__ float tn_outer(float a, float b) {
  float r;
  if constexpr (BOT) asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <bool BOT>
__device__ __forceinline__ float tn_inner(float a, float b) {
  return tn_outer<!BOT>(a, b);
}
template <bool BOT>
__device__ __forceinline__ void tn_ce(float& a, float& b) {  // a <- the outer one
  const float o = tn_outer<BOT>(a, b), i = tn_inner<BOT>(a, b);
  a = o;
  b = i;
}
// the innermost value: what a NaN / an absent day / a padding slot becomes
template <bool BOT>
__device__ __forceinline__ float tn_sentinel() {
  return __uint_as_float(BOT ? 0x7F800000u : 0xFF800000u);
}

// 16 values sorted outer-first with the 60-comparator, 10-layer optimal network (verified exhaustively with the 0-1
// principle by tools/gen_sortnet.py's checker; the bitonic sorter needs 80)
template <bool BOT>
__device__ __forceinline__ void tn_sort16(float (&k)[16]) {
#define XH_C(i, j) tn_ce<BOT>(k[i], k[j]);
  XH_C(0, 13) XH_C(1, 12) XH_C(2, 15) XH_C(3, 14) XH_C(4, 8) XH_C(5, 6) XH_C(7, 11) XH_C(9, 10)
  XH_C(0, 5) XH_C(1, 7) XH_C(2, 9) XH_C(3, 4) XH_C(6, 13) XH_C(8, 14) XH_C(10, 15) XH_C(11, 12)
  XH_C(0, 1) XH_C(2, 3) XH_C(4, 5) XH_C(6, 8) XH_C(7, 9) XH_C(10, 11) XH_C(12, 13) XH_C(14, 15)
  XH_C(0, 2) XH_C(1, 3) XH_C(4, 10) XH_C(5, 11) XH_C(6, 7) XH_C(8, 9) XH_C(12, 14) XH_C(13, 15)
  XH_C(1, 2) XH_C(3, 12) XH_C(4, 6) XH_C(5, 7) XH_C(8, 10) XH_C(9, 11) XH_C(13, 14)
  XH_C(1, 4) XH_C(2, 6) XH_C(5, 8) XH_C(7, 10) XH_C(9, 13) XH_C(11, 14)
  XH_C(2, 4) XH_C(3, 6) XH_C(9, 12) XH_C(11, 13)
  XH_C(3, 5) XH_C(6, 8) XH_C(7, 9) XH_C(10, 12)
  XH_C(3, 4) XH_C(5, 6) XH_C(7, 8) XH_C(9, 10) XH_C(11, 12)
  XH_C(6, 7) XH_C(8, 9)
#undef XH_C
}

// o <- the 16 outermost of (a u b), sorted; a and b sorted.  outer(a[i], b[15 - i]) is bitonic and holds exactly those
// 16; four compare-exchange stages re-sort it (16 + 32 comparators).  o may alias a.
template <bool BOT>
__device__ __forceinline__ void tn_merge16(float (&o)[16], const float (&a)[16], const float (&b)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) o[i] = tn_outer<BOT>(a[i], b[15 - i]);
#pragma unroll
  for (int stride = 8; stride > 0; stride >>= 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if ((i & stride) == 0) tn_ce<BOT>(o[i], o[i + stride]);
  }
}

// 32 values sorted outer-first: two sort-16 blocks + a full bitonic merge (60 + 60 + 80 comparators)
template <bool BOT>
__device__ __forceinline__ void tn_sort32(float (&k)[32]) {
  float a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = k[i];
    b[i] = k[16 + i];
  }
  tn_sort16<BOT>(a);
  tn_sort16<BOT>(b);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    tn_ce<BOT>(a[i], b[15 - i]);  // a: the 16 outermost of all 32, b: the rest — both bitonic
  }
#pragma unroll
  for (int stride = 8; stride > 0; stride >>= 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if ((i & stride) == 0) {
        tn_ce<BOT>(a[i], a[i + stride]);
        tn_ce<BOT>(b[i], b[i + stride]);
      }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    k[i] = a[i];
    k[16 + i] = b[i];
  }
}

// positions 15 and 14 of the 16 outermost of (a u b) without sorting them: the inner half of a half-cleaner holds the
// inner half of a bitonic sequence and is bitonic again (16 + 8 + 4 + 2 + 2 single instructions instead of 16 + 64)
template <bool BOT>
__device__ __forceinline__ void tn_last2(const float (&a)[16], const float (&b)[16], float& p15, float& p14) {
  float c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = tn_outer<BOT>(a[i], b[15 - i]);
#pragma unroll
  for (int n = 8; n >= 2; n >>= 1) {
#pragma unroll
    for (int i = 0; i < n; ++i) c[i] = tn_inner<BOT>(c[i], c[i + n]);
  }
  p15 = tn_inner<BOT>(c[0], c[1]);
  p14 = tn_outer<BOT>(c[0], c[1]);
}

// key <- raw with NaN replaced by `sentinel`, nn += the NaNs.  One asm block per sample so that the compare result stays
// in VCC: left to the compiler, the results are hoisted into SGPR pairs and spill to VGPR lanes.
__device__ __forceinline__ void tn_denan(float& key, int& nn, float raw, float sentinel) {
  asm("v_cmp_u_f32 vcc, %2, %2\n\tv_cndmask_b32 %0, %2, %3, vcc\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "=&v"(key), "+v"(nn)
      : "v"(raw), "v"(sentinel)
      : "vcc");
}
// the same in place
__device__ __forceinline__ void tn_denan_inplace(float& v, int& nn, float sentinel) {
  asm("v_cmp_u_f32 vcc, %0, %0\n\tv_cndmask_b32 %0, %0, %2, vcc\n\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(v), "+v"(nn)
      : "v"(sentinel)
      : "vcc");
}
