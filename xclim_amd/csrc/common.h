// common.h — internal helpers shared by the HIP translation units of libxclimhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/xclim_hip.h"

struct xh_ctx {
  int device;
  hipStream_t stream;
  hipEvent_t ev0, ev1;
  // second stream + events: transposes of the next column batch run next to the selection of the current one
  hipStream_t stream2;
  hipEvent_t ev_ready[2], ev_done[2];
  // copy lanes of the block adapter (xclim_amd/blocks.py): lane 1 = host -> device, lane 2 = device -> host; lane 0 is
  // `stream`.  ev_lane is the fence event (record on one lane, wait on another).
  hipStream_t copy_in, copy_out;
  hipEvent_t ev_lane;
  // small device scratch for tables (seg_off, quantiles ...) uploaded per call: a RING shared by consecutive calls
  // with a pinned host mirror, so that uploads are asynchronous (no stream synchronisation per call)
  void* scratch;
  size_t scratch_bytes;
  char* scratch_host;   // pinned, same size
  size_t scratch_head;  // next free offset
  void* retired[16];    // rings replaced by a larger one: kept until xh_destroy (tables handed out earlier in the same
  char* retired_host[16];  //   call may still point into them)
  int nretired;
  // large scratch (transposes), grown on demand
  void* big;
  size_t big_bytes;
  // three constant rows (NaN | -inf | +inf), grown on demand and never written afterwards: the gather kernels of
  // pdoy_top.hip / pdoy_quad.hip read absent days and padding slots from them, which keeps the mask arithmetic out of
  // their loops (xh_const_rows)
  void* nanrow;
  size_t nanrow_bytes;  // of ONE row
  int num_cu;
};

void xh_set_error(const char* fmt, ...);

// Diagnostic / tuning switches (phase ablation, phase timers, kernel-shape overrides) are read from the environment
// ONLY when XH_DIAGNOSTICS=1 is set as well: an ablation switch makes results wrong on purpose and must never be picked
// up by accident in production.  Used by tools/bench_select_abl.sh, tools/pmc_eqm.sh.
const char* xh_diag_env(const char* name);

#define XH_CHECK_HIP(expr)                                                                  \
  do {                                                                                      \
    hipError_t _e = (expr);                                                                 \
    if (_e != hipSuccess) {                                                                 \
      xh_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      (void)hipGetLastError(); /* do not leave a sticky error for the next launch check */  \
      return XH_ERR_HIP;                                                                    \
    }                                                                                       \
  } while (0)

#define XH_REQUIRE(cond, code, ...) \
  do {                              \
    if (!(cond)) {                  \
      xh_set_error(__VA_ARGS__);    \
      return (code);                \
    }                               \
  } while (0)

#define XH_LAUNCH_CHECK()                                                                      \
  do {                                                                                         \
    hipError_t _e = hipGetLastError();                                                         \
    if (_e != hipSuccess) {                                                                    \
      xh_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return XH_ERR_HIP;                                                                       \
    }                                                                                          \
  } while (0)

// Upload a small host table into the context scratch (bump allocated per call via `*cursor`).
int xh_nanmax_fix(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int nq, float* out);  // select.hip
int xh_correction_fix(xh_ctx* ctx, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st, int64_t sc, int nq, int kind,
                      float* af, float* hist_q);  // select.hip
int xh_scratch_upload(xh_ctx* ctx, size_t* cursor, const void* host, size_t bytes, void** dptr);
int xh_big_scratch(xh_ctx* ctx, size_t bytes, void** dptr);
// >= elems floats each of NaN, -inf, +inf (persistent; any of the three pointers may be NULL)
int xh_const_rows(xh_ctx* ctx, int64_t elems, const float** nan_row, const float** ninf_row, const float** pinf_row);
// tcount.hip: XH_OK launched, XH_ERR_NOTIMPL (no error text) = outside the tile kernel's domain
int xh_launch_tcount_doy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int op, const double* table,
                         int64_t tstride, const int32_t* tidx, const int64_t* d_seg, const int64_t* h_seg, int P, int ndoy,
                         int32_t* count_out, int32_t* valid_out);
// qdm.hip / qdm2.hip
int xh_qdm_columns(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride, const float* af, int64_t af_qs,
                   const double* d_q, int nq, int kind, int interp, int extrap, float* out, int64_t out_cs);
// qdm3.hip: exact ranks through a global (rocPRIM) sort, any T; workspace query + run on time-minor columns
int xh_qdm_sorted_ws(int64_t T, int64_t ncols, size_t* bytes);
int xh_qdm_sorted(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride, const float* af, int64_t af_qs,
                  const double* d_q, int nq, int kind, int interp, int extrap, float* out, int64_t out_cs, void* ws);
int xh_qdm_regsort(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, const float* af, int64_t af_qs,
                   const double* d_q, int nq, int kind, int extrap, float* scen, int64_t ost);
int xh_cut_classify(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, const float* gcut, const float* gfac, int ntest,
                    int kind, float* scen, int64_t ost);
// select4.hip: QDM "nearest" on long time-major series (1024 < T <= 65535): class boundaries as order statistics from the
// two-pass histogram selection, then one streaming classification pass; XH_ERR_NOTIMPL when the shape does not fit
int xh_qdm_hist(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, const float* af, int64_t af_qs, const double* d_q,
                int nq, int kind, int extrap, float* scen, int64_t ost);
int xh_tcount_plan(int64_t T, int64_t C, int64_t st, int op, int P, int ndoy, int64_t longest, size_t* lds, int* narrow);
int64_t xh_tcount_meta_slots(int64_t T);
int64_t xh_tcount_slot_of_row(int64_t t);
int xh_tcount_run(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int op, const double* table, int64_t tstride,
                  const uint32_t* meta, int P, int ndoy, int narrow, size_t lds, int32_t* count_out, int32_t* valid_out);
// select.hip: per-column exact multi-quantile selection on a time-minor view (d_q: device pointer)
int xh_select_columns(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride, const double* d_q,
                      int nq, float* out, int64_t out_cstride, int64_t out_qstride);
// select2.hip: long series (1024 < T <= 16384) without a per-column LDS copy; XH_ERR_NOTIMPL outside that range
int xh_select_columns_lean(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride,
                           const double* d_q, int nq, float* out, int64_t out_cstride, int64_t out_qstride);
// select5.hip: radix select, one workgroup per column, any T < 2^31 (the fallback above 32768 samples per column)
int xh_select_columns_radix(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride, const double* d_q,
                            int nq, float* out, int64_t out_cstride, int64_t out_qstride);
// select3.hip: one-year daily series (360 <= T <= 366) from a time-major view, column in registers + sorting network;
// XH_ERR_NOTIMPL when the shape does not fit
int xh_select_regsort(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const double* d_q, int nq, float* out,
                      int64_t out_cstride, int64_t out_qstride);
// select4.hip: long series (1024 < T <= 65535) straight from a time-major view: two streaming passes with per-column
// histograms in LDS; XH_ERR_NOTIMPL when the shape does not fit or too many columns need the column kernels
int xh_select_hist(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const double* d_q, int nq, float* out,
                   int64_t out_cstride, int64_t out_qstride);
// short series read straight from a time-major view; XH_ERR_NOTIMPL when the shape does not fit
int xh_select_time_major(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const double* d_q, int nq,
                         float* out, int64_t out_cstride, int64_t out_qstride);

__host__ __device__ static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// ---- device helpers ---------------------------------------------------------------------------
#define XH_BLOCK 256

// Run-time operator as four wave-uniform masks (greater / less / equal / unordered) instead of a switch: the masks are
// loop invariant scalars, every element costs the same three or four v_cmp whatever the operator, and the compiler can
// keep the loop bodies branch-free.  NaN compares False except for != (numpy semantics): only NE sets `un`.
__device__ __forceinline__ bool xh_cmp_f32(float a, int op, float b) {
  const bool gt = op == XH_OP_GT || op == XH_OP_GE || op == XH_OP_NE;
  const bool lt = op == XH_OP_LT || op == XH_OP_LE || op == XH_OP_NE;
  const bool eq = op == XH_OP_GE || op == XH_OP_LE || op == XH_OP_EQ;
  const bool un = op == XH_OP_NE;
  return (gt & (a > b)) | (lt & (a < b)) | (eq & (a == b)) | (un & ((a != a) | (b != b)));
}
// fp64 compares are half rate: here the switch is kept (the compiler unswitches the loops on the uniform operator and
// every element costs ONE compare; measured 0.75 vs 0.80 ms for the per-doy threshold_count of the tx90p chain)
__device__ __forceinline__ bool xh_cmp_f64(double a, int op, double b) {
  switch (op) {
    case XH_OP_GT: return a > b;
    case XH_OP_LT: return a < b;
    case XH_OP_GE: return a >= b;
    case XH_OP_LE: return a <= b;
    case XH_OP_EQ: return a == b;
    default: return a != b;
  }
}

// One-compare form of a run-time ordering operator: for > < >= <= against a FINITE float threshold t there are (sgn, t')
// with   x OP t  <=>  sgn * x > t'   for every x (NaN included: both sides are False):
//   >: (+1, t)   <: (-1, -t)   >=: (+1, pred(t))   <=: (-1, pred(-t)),   pred = the next float towards -inf.
// One multiply (full rate) + one v_cmp per element instead of the three compares + mask logic of xh_cmp_f32 (v_cmp
// issues at half rate on gfx950, tools/valu_ubench.hip).  `ok` = 0 for == / != and non-finite thresholds: use xh_cmp_f32.
struct XhOneCmp {
  float sgn, thr;
  int ok;
};
static inline XhOneCmp xh_one_cmp(int op, float t) {
  XhOneCmp r = {1.0f, t, 0};
  if (!(t == t) || t > 3.4028234e38f || t < -3.4028234e38f) return r;
  auto pred = [](float v) -> float {
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7FFFFFFFu) == 0u) u = 0x80000001u;  // +-0 -> the smallest negative subnormal
    else u = (u >> 31) ? u + 1u : u - 1u;           // away from zero for negatives, towards zero for positives
    memcpy(&v, &u, 4);
    return v;
  };
  switch (op) {
    case XH_OP_GT: r.sgn = 1.0f; r.thr = t; r.ok = 1; break;
    case XH_OP_LT: r.sgn = -1.0f; r.thr = -t; r.ok = 1; break;
    case XH_OP_GE: r.sgn = 1.0f; r.thr = pred(t); r.ok = 1; break;
    case XH_OP_LE: r.sgn = -1.0f; r.thr = pred(-t); r.ok = 1; break;
    default: break;
  }
  return r;
}

template <int OP>
__device__ __forceinline__ bool xh_cmp_t(float a, float b) {
  if (OP == XH_OP_GT) return a > b;
  if (OP == XH_OP_LT) return a < b;
  if (OP == XH_OP_GE) return a >= b;
  if (OP == XH_OP_LE) return a <= b;
  if (OP == XH_OP_EQ) return a == b;
  return a != b;
}

// vector-of-cells loads: VEC consecutive cells per lane (16 B per lane when VEC == 4)
template <int VEC>
struct VecF;
template <>
struct VecF<1> {
  float v[1];
};
template <>
struct VecF<2> {
  float v[2];
};
template <>
struct VecF<4> {
  float v[4];
};

template <int VEC>
__device__ __forceinline__ VecF<VEC> xh_load(const float* __restrict__ p) {
  VecF<VEC> r;
  if (VEC == 4) {
    float4 t = *reinterpret_cast<const float4*>(p);
    r.v[0] = t.x; r.v[1 % VEC] = t.y; r.v[2 % VEC] = t.z; r.v[3 % VEC] = t.w;
  } else if (VEC == 2) {
    float2 t = *reinterpret_cast<const float2*>(p);
    r.v[0] = t.x; r.v[1 % VEC] = t.y;
  } else {
    r.v[0] = *p;
  }
  return r;
}

// March along time over rows [t0, t1) of the lane's VEC cells with double-buffered batches of U rows: the loads of
// batch b+1 are in flight while batch b is consumed, so a lane keeps U..2U independent 16-byte loads outstanding
// (the serial state machines of the time-marching kernels are otherwise HBM-latency bound).
template <int VEC, int U, typename F>
__device__ __forceinline__ void xh_march_rows(const float* __restrict__ p, int64_t st, int64_t t0, int64_t t1, F&& f) {
  int64_t t = t0;
  int64_t nfull = (t1 - t0) / U;
  if (nfull > 0) {
    VecF<VEC> buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) buf[u] = xh_load<VEC>(p + (t + u) * st);
    for (int64_t b = 0; b < nfull; ++b) {
      VecF<VEC> nxt[U];
      const bool more = b + 1 < nfull;
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = xh_load<VEC>(p + (t + U + u) * st);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) f(t + u, buf[u]);
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) buf[u] = nxt[u];
      }
      t += U;
    }
  }
  for (; t < t1; ++t) f(t, xh_load<VEC>(p + t * st));
}

// Two arrays of the same layout marched together (q may alias p).
template <int VEC, int U, typename F>
__device__ __forceinline__ void xh_march_rows2(const float* __restrict__ p, const float* __restrict__ q, int64_t st,
                                               int64_t sq, int64_t t0, int64_t t1, F&& f) {
  int64_t t = t0;
  int64_t nfull = (t1 - t0) / U;
  if (nfull > 0) {
    VecF<VEC> bp[U], bq[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { bp[u] = xh_load<VEC>(p + (t + u) * st); bq[u] = xh_load<VEC>(q + (t + u) * sq); }
    for (int64_t b = 0; b < nfull; ++b) {
      VecF<VEC> np[U], nq[U];
      const bool more = b + 1 < nfull;
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) { np[u] = xh_load<VEC>(p + (t + U + u) * st); nq[u] = xh_load<VEC>(q + (t + U + u) * sq); }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) f(t + u, bp[u], bq[u]);
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) { bp[u] = np[u]; bq[u] = nq[u]; }
      }
      t += U;
    }
  }
  for (; t < t1; ++t) f(t, xh_load<VEC>(p + t * st), xh_load<VEC>(q + t * sq));
}

// Same march in REVERSE time order (t1-1 down to t0) for the backward state machines.
template <int VEC, int U, typename F>
__device__ __forceinline__ void xh_march_rows_rev(const float* __restrict__ p, int64_t st, int64_t t0, int64_t t1, F&& f) {
  int64_t t = t1;  // rows [t - U, t) are the next batch
  int64_t nfull = (t1 - t0) / U;
  if (nfull > 0) {
    VecF<VEC> buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) buf[u] = xh_load<VEC>(p + (t - 1 - u) * st);
    for (int64_t b = 0; b < nfull; ++b) {
      VecF<VEC> nxt[U];
      const bool more = b + 1 < nfull;
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = xh_load<VEC>(p + (t - U - 1 - u) * st);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) f(t - 1 - u, buf[u]);
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) buf[u] = nxt[u];
      }
      t -= U;
    }
  }
  for (t = t - 1; t >= t0; --t) f(t, xh_load<VEC>(p + t * st));
}

// order-preserving float <-> uint32 key (ascending; NaN maps above +inf so it sorts last like numpy)
__device__ __forceinline__ uint32_t xh_f2key(float f) {
  uint32_t u = __float_as_uint(f);
  if (f != f) return 0xFFFFFFFFu;
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float xh_key2f(uint32_t k) {
  if (k == 0xFFFFFFFFu) return __uint_as_float(0x7FC00000u);
  uint32_t u = (k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k;
  return __uint_as_float(u);
}

// Histogram bins of the selection / rank kernels.  They divide [kmin2, kmax] linearly in KEY space (a power-of-two shift) — log-like
// in the value, which suits skewed positive data (precipitation: the wet days spread over the bins).  But when the keys STRADDLE
// ZERO (temperatures in degrees Celsius, anomalies, a series normalised by its mean) the key space in between is almost all
// exponent range: a bin is a whole binade, eight bins hold every sample and the exact search inside a bin (select_in_bin; the
// O(m) scan per key of k_qdm_columns) turns quadratic — measured round 6: eqm_train on 930 rows x 1440 x 90 in kelvin 1.36 ms, the
// same field in degrees Celsius 19.7 ms.  Then the bins are linear in the VALUE instead (fp32 subtraction, multiplication by a
// positive scale and truncation are all monotone, which is all the exact selection needs of a bin index).
struct XhValueBins {
  int on;
  float lo, scale;
};
__device__ __forceinline__ XhValueBins xh_value_bins(uint32_t kmin2, uint32_t kmax, bool any, int nb) {
  XhValueBins v{0, 0.f, 0.f};
  if (any && kmin2 < 0x80000000u && kmax >= 0x80000000u && kmax != 0xFFFFFFFFu) {
    const float lo = xh_key2f(kmin2), hi = xh_key2f(kmax);
    const float span = hi - lo;
    if (span > 0.f && span - span == 0.f) {   // (finite)
      v.on = 1;
      v.lo = lo;
      v.scale = (float)(nb - 1) / span;
    }
  }
  return v;
}
// bin index in [0, nb - 1] (callers clamp); garbage for a NaN key or a key below kmin2, which callers never use
__device__ __forceinline__ uint32_t xh_value_bin(const XhValueBins& v, uint32_t kk) {
  const float d = (xh_key2f(kk) - v.lo) * v.scale;
  return d > 0.f ? (uint32_t)d : 0u;
}

// s / n for a small positive integer n, given inv = 1.0 / n: q = RN(s * inv), r = s - n * q (exact with FMA),
// q' = RN(q + r * inv) is the correctly rounded quotient (Markstein's theorem; inv is the correctly rounded
// reciprocal) — bit-identical to the division at 3 fp64 FMAs instead of the ~12-instruction IEEE sequence, which
// matters in the rolling kernels where it runs once per cell-timestep.  Non-finite s: r is NaN, keep q.
__device__ __forceinline__ double xh_div_int(double s, double n, double inv) {
  const double q = s * inv;
  const double r = fma(-q, n, s);
  return (r == r) ? fma(r, inv, q) : q;
}

__device__ __forceinline__ double xh_nan64() { return __longlong_as_double(0x7FF8000000000000LL); }
__device__ __forceinline__ float xh_nan32() { return __uint_as_float(0x7FC00000u); }

// choose cells-per-lane: 4 when rows are 16-byte aligned, else 1
static inline int xh_pick_vec(const void* p, int64_t C, int64_t st) {
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && (C % 4) == 0 && (st % 4) == 0) return 4;
  return 1;
}
