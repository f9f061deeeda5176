// rowstream.h — the row-lane streaming loop of select4.hip without the key conversion, for kernels that march a
// (CW columns x RLT row lanes) workgroup down a time-major field: thread (col, rl) visits rows rl, rl + RLT, ... in batches
// of U rows per lane (U * RLT rows per workgroup and batch); f(values, kb) is called once per batch, kb = batch index.
// Rows past T arrive as NaN.  Loads: buffer loads with a descriptor re-based per batch, the row of load u as a scalar
// offset and one 32-bit per-lane byte offset; two register sets in ping-pong (see select4.hip for the measurements).
#pragma once
#include "common.h"

template <int U, int RLT, typename F>
__device__ __forceinline__ void xh_row_stream(const float* __restrict__ x, int T, int64_t st, int64_t cc, int rl, F&& f) {
  constexpr int ROWS = RLT * U;
  const uint32_t voff = (uint32_t)(((int64_t)rl * st + cc) * 4);
  const uint32_t rowstep = (uint32_t)(st * 4 * RLT);  // host: ROWS rows < 4 GiB
  const int nfull = T / ROWS;
  auto load = [&](float (&dst)[U], int kb) {
    const float* base = x + (int64_t)kb * ROWS * st;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
    uint32_t soff = 0u;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      dst[u] = __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
      soff += rowstep;
    }
  };
  if (nfull > 0) {
    float A[U], B[U];
    load(A, 0);
    int kb = 0;
    while (kb + 2 < nfull) {  // no conditional loads in here: the compiler counts the outstanding loads exactly
      load(B, kb + 1);
      f(A, kb);
      load(A, kb + 2);
      f(B, kb + 1);
      kb += 2;
    }
    if (kb + 1 < nfull) {
      load(B, kb + 1);
      f(A, kb);
      f(B, kb + 1);
    } else {
      f(A, kb);
    }
  }
  const int t0 = nfull * ROWS;
  if (t0 < T) {  // tail rows: clamped per-lane rows, validity applied afterwards
    int rlo = rl;
    asm volatile("" : "+v"(rlo));
    const int rem = T - t0;
    const float* base = x + (int64_t)t0 * st;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
    float buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int r = u * RLT + rlo;
      r = r < rem ? r : rem - 1;
      const uint32_t vo = (uint32_t)r * (uint32_t)(st * 4) + (uint32_t)(cc * 4);
      buf[u] = __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)vo, 0, 0));
    }
#pragma unroll
    for (int u = 0; u < U; ++u) buf[u] = u * RLT + rlo < rem ? buf[u] : __uint_as_float(0x7FC00000u);
    f(buf, nfull);
  }
}
