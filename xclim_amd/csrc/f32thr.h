// f32thr.h — an fp64 threshold as the ONE fp32 value that gives the same compare against fp32 samples.
#pragma once
#include "common.h"

// Largest float <= r / smallest float >= r: with them the fp64 compare `(double)x OP r` of an fp32 sample against an fp64
// threshold (numpy promotion of tx90p's compare, gen:301-361) is exactly ONE fp32 compare:
//   x > r  <=>  x > below(r)      x <= r  <=>  x <= below(r)      x < r  <=>  x < above(r)      x >= r  <=>  x >= above(r)
// (every float above below(r) is above r).  NaN stays NaN (every compare False).
__device__ __forceinline__ float f32_step(float f, bool up) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7FFFFFFFu) == 0u) return __uint_as_float(up ? 0x00000001u : 0x80000001u);  // +-0 -> smallest subnormal
  const bool away = up == !(u >> 31);  // moving away from zero
  return __uint_as_float(away ? u + 1u : u - 1u);
}
__device__ __forceinline__ float f32_threshold(double r, int op) {
  float f = (float)r;  // round to nearest
  if (f != f || f == __uint_as_float(0x7F800000u) || f == __uint_as_float(0xFF800000u)) {
    // overflow of a finite r to +-inf: step back inside when the direction demands it
    if (r == r && (double)f != r) {
      const bool want_below = op == XH_OP_GT || op == XH_OP_LE;
      if (want_below && f > 0.0f) return __uint_as_float(0x7F7FFFFFu);
      if (!want_below && f < 0.0f) return __uint_as_float(0xFF7FFFFFu);
    }
    return f;
  }
  if (op == XH_OP_GT || op == XH_OP_LE) return (double)f > r ? f32_step(f, false) : f;  // below(r)
  return (double)f < r ? f32_step(f, true) : f;                                          // above(r)
}
