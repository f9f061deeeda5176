// reduce.hip — threshold/count family, segmented reductions, missing mask, rolling reductions.
//
// Layout: time-major (T, C), one lane owns VEC consecutive cells and marches along time; a wave reads
// 64*VEC*4 contiguous bytes per time step (1 KiB at VEC = 4).  Periods (resample segments) are mapped to
// blockIdx.y so that P periods give P-fold more workgroups.  All kernels are HBM-bound.
#include "common.h"
#include "window.h"

// ---- threshold_count ------------------------------------------------------------------------------
// Reference: threshold_count (indices/generic.py:329-361) + compare (gen:301-326) + resample.sum, fused
// with MissingBase.is_valid / MissingAny (core/missing.py:201-220, 318-322).
template <int VEC, int KIND>
__global__ void __launch_bounds__(XH_BLOCK)
k_threshold_count(const float* __restrict__ x, int64_t C, int64_t st, int op, float thr32, double thr64,
                  const void* __restrict__ table, int64_t tstride, const int32_t* __restrict__ tidx,
                  const int64_t* __restrict__ seg_off, int P, int32_t* __restrict__ count_out,
                  int32_t* __restrict__ valid_out, int period_fast) {
  // period_fast: periods on blockIdx.x, cell tiles on blockIdx.y — workgroups that are dispatched together then
  // work on the SAME cell tile in different years and share the per-doy threshold rows through L2 / Infinity Cache
  // (otherwise a multi-year tx90p re-reads the (D, C) fp64 table from HBM once per year).
  const unsigned tile = period_fast ? blockIdx.y : blockIdx.x;
  const int pstart = period_fast ? blockIdx.x : blockIdx.y, pstep = period_fast ? gridDim.x : gridDim.y;
  int64_t c = ((int64_t)tile * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  for (int p = pstart; p < P; p += pstep) {
    int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    int cnt[VEC], val[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) cnt[i] = 0, val[i] = 0;
    xh_march_rows<VEC, 8>(x + c, st, t0, t1, [&](int64_t t, const VecF<VEC>& xv) {
      if (KIND == XH_THR_SCALAR_F32) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) cnt[i] += xh_cmp_f32(xv.v[i], op, thr32) ? 1 : 0;
      } else if (KIND == XH_THR_SCALAR_F64) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) cnt[i] += xh_cmp_f64((double)xv.v[i], op, thr64) ? 1 : 0;
      } else if (KIND == XH_THR_DOY_F64 || KIND == XH_THR_FULL_F64) {
        int64_t row = (KIND == XH_THR_DOY_F64) ? (int64_t)tidx[t] : t;
        const double* tp = reinterpret_cast<const double*>(table) + row * tstride + c;
        double th[VEC];
        if (VEC == 4) {
          double2 a = *reinterpret_cast<const double2*>(tp);
          double2 b = *reinterpret_cast<const double2*>(tp + 2);
          th[0] = a.x; th[1 % VEC] = a.y; th[2 % VEC] = b.x; th[3 % VEC] = b.y;
        } else {
#pragma unroll
          for (int i = 0; i < VEC; ++i) th[i] = tp[i];
        }
#pragma unroll
        for (int i = 0; i < VEC; ++i) cnt[i] += xh_cmp_f64((double)xv.v[i], op, th[i]) ? 1 : 0;
      } else {
        int64_t row = (KIND == XH_THR_DOY_F32) ? (int64_t)tidx[t] : t;
        VecF<VEC> th = xh_load<VEC>(reinterpret_cast<const float*>(table) + row * tstride + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) cnt[i] += xh_cmp_f32(xv.v[i], op, th.v[i]) ? 1 : 0;
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) val[i] += (xv.v[i] == xv.v[i]) ? 1 : 0;
    });
    int64_t o = (int64_t)p * C + c;
#pragma unroll
    for (int i = 0; i < VEC; ++i) count_out[o + i] = cnt[i];
    if (valid_out) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) valid_out[o + i] = val[i];
    }
  }
}

template <int VEC>
static int launch_threshold_count(xh_ctx* ctx, int kind, dim3 grid, const float* x, int64_t C, int64_t st, int op,
                                  double thr, const void* table, int64_t tstride, const int32_t* tidx,
                                  const int64_t* seg, int P, int32_t* count_out, int32_t* valid_out, int period_fast) {
#define XH_TC(K)                                                                                                      \
  case K:                                                                                                             \
    hipLaunchKernelGGL((k_threshold_count<VEC, K>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, op, (float)thr, thr, \
                       table, tstride, tidx, seg, P, count_out, valid_out, period_fast);                              \
    break;
  switch (kind) {
    XH_TC(XH_THR_SCALAR_F32)
    XH_TC(XH_THR_SCALAR_F64)
    XH_TC(XH_THR_DOY_F64)
    XH_TC(XH_THR_DOY_F32)
    XH_TC(XH_THR_FULL_F64)
    XH_TC(XH_THR_FULL_F32)
    default:
      xh_set_error("xh_threshold_count: unknown thr_kind %d", kind);
      return XH_ERR_ARG;
  }
#undef XH_TC
  XH_LAUNCH_CHECK();
  return XH_OK;
}

static inline unsigned period_grid(int P) { return (unsigned)(P < 1 ? 1 : (P > 4096 ? 4096 : P)); }

// ---- domain_count ----------------------------------------------------------------------------------
// FAST: both conditions in the one-compare form of xh_one_cmp (sgn * x > thr): 2 multiplies + 2 compares per element
// instead of two run-time operators (PMC: 28 VALU per element, 0.39 ms at 365 x 1440 x 720 before).
template <int VEC, bool FAST = false>
__global__ void __launch_bounds__(XH_BLOCK)
k_domain_count(const float* __restrict__ x, int64_t C, int64_t st, int op1, float thr1, int op2, float thr2, int combine,
               const int64_t* __restrict__ seg_off, int P, int32_t* __restrict__ count_out,
               int32_t* __restrict__ valid_out) {
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    int cnt[VEC], val[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) cnt[i] = 0, val[i] = 0;
    xh_march_rows<VEC, 8>(x + c, st, t0, t1, [&](int64_t, const VecF<VEC>& xv) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        bool a, b;
        if (FAST) {  // op1 / op2 carry the signs here (see the launcher)
          a = xv.v[i] * __int_as_float(op1) > thr1;
          b = xv.v[i] * __int_as_float(op2) > thr2;
        } else {
          a = xh_cmp_f32(xv.v[i], op1, thr1);
          b = xh_cmp_f32(xv.v[i], op2, thr2);
        }
        cnt[i] += ((combine == 1) ? (a && b) : (a || b)) ? 1 : 0;
        val[i] += (xv.v[i] == xv.v[i]) ? 1 : 0;
      }
    });
    int64_t o = (int64_t)p * C + c;
#pragma unroll
    for (int i = 0; i < VEC; ++i) count_out[o + i] = cnt[i];
    if (valid_out) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) valid_out[o + i] = val[i];
    }
  }
}

// ---- select_resample_op ------------------------------------------------------------------------------
// Reference: select_resample_op (indices/generic.py:83-125) -> da.resample(time=freq).<op>(dim="time").
// xarray reduces floats with skipna=True: nansum/nanmean/nanmin/nanmax/nanstd/nanvar (ddof 0), count = notnull.
// Accumulation here is fp64 in time order (stated in DESIGN.md; the reference stack accumulates in fp32 or
// fp64 depending on bottleneck/numpy, differing by ~1e-7 relative).
template <int VEC, int RED>
__global__ void __launch_bounds__(XH_BLOCK)
k_resample_reduce(const float* __restrict__ x, int64_t C, int64_t st, int skipna, const int64_t* __restrict__ seg_off,
                  int P, void* __restrict__ out_v, int32_t* __restrict__ valid_out) {
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    double s1[VEC], s2[VEC];
    float ext[VEC];
    int n[VEC], arg[VEC], nanseen[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      s1[i] = 0.0; s2[i] = 0.0; n[i] = 0; arg[i] = -1; nanseen[i] = 0;
      ext[i] = 0.f;
    }
    if (RED == XH_RED_STD || RED == XH_RED_VAR) {
      // two passes for a numerically faithful nanvar: mean first (second pass is an L2/MALL re-read of the segment)
#pragma unroll 4
      for (int64_t t = t0; t < t1; ++t) {
        VecF<VEC> xv = xh_load<VEC>(x + t * st + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          bool ok = xv.v[i] == xv.v[i];
          if (ok) { s1[i] += (double)xv.v[i]; n[i]++; } else nanseen[i] = 1;
        }
      }
      double mean[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) mean[i] = n[i] > 0 ? s1[i] / (double)n[i] : 0.0;
#pragma unroll 4
      for (int64_t t = t0; t < t1; ++t) {
        VecF<VEC> xv = xh_load<VEC>(x + t * st + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          if (xv.v[i] == xv.v[i]) { double d = (double)xv.v[i] - mean[i]; s2[i] += d * d; }
        }
      }
    } else {
      xh_march_rows<VEC, 8>(x + c, st, t0, t1, [&](int64_t t, const VecF<VEC>& xv) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          float v = xv.v[i];
          bool ok = v == v;
          if (ok) {
            if (RED == XH_RED_SUM || RED == XH_RED_MEAN) s1[i] += (double)v;
            if (RED == XH_RED_MIN || RED == XH_RED_ARGMIN) {
              if (n[i] == 0 || v < ext[i]) { ext[i] = v; arg[i] = (int)(t - t0); }
            }
            if (RED == XH_RED_MAX || RED == XH_RED_ARGMAX) {
              if (n[i] == 0 || v > ext[i]) { ext[i] = v; arg[i] = (int)(t - t0); }
            }
            n[i]++;
          } else {
            if (!nanseen[i] && !skipna && (RED == XH_RED_ARGMIN || RED == XH_RED_ARGMAX)) arg[i] = (int)(t - t0);
            nanseen[i] = 1;
          }
        }
      });
    }
    int64_t o = (int64_t)p * C + c;
    int len = (int)(t1 - t0);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      bool poisoned = (!skipna && nanseen[i]);
      if (RED == XH_RED_COUNT) {
        reinterpret_cast<int32_t*>(out_v)[o + i] = n[i];
      } else if (RED == XH_RED_ARGMIN || RED == XH_RED_ARGMAX) {
        // numpy nanarg* raise on all-NaN slices; we return -1 there.  skipna=False: first NaN position.
        reinterpret_cast<int32_t*>(out_v)[o + i] = arg[i];
      } else {
        float r;
        if (RED == XH_RED_SUM) r = poisoned ? xh_nan32() : (float)s1[i];
        else if (RED == XH_RED_MEAN) r = (poisoned || n[i] == 0) ? xh_nan32() : (float)(s1[i] / (double)n[i]);
        else if (RED == XH_RED_MIN || RED == XH_RED_MAX) r = (poisoned || n[i] == 0) ? xh_nan32() : ext[i];
        else if (RED == XH_RED_VAR) r = (poisoned || n[i] == 0) ? xh_nan32() : (float)(s2[i] / (double)n[i]);
        else r = (poisoned || n[i] == 0) ? xh_nan32() : (float)sqrt(s2[i] / (double)n[i]);
        if (len == 0 && RED != XH_RED_SUM) r = xh_nan32();
        reinterpret_cast<float*>(out_v)[o + i] = r;
      }
    }
    if (valid_out) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) valid_out[o + i] = n[i];
    }
  }
}

template <int VEC>
static int launch_resample_reduce(xh_ctx* ctx, int reducer, dim3 grid, const float* x, int64_t C, int64_t st, int skipna,
                                  const int64_t* seg, int P, void* out, int32_t* valid_out) {
#define XH_RR(R)                                                                                                   \
  case R:                                                                                                          \
    hipLaunchKernelGGL((k_resample_reduce<VEC, R>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, skipna, seg, P, out, \
                       valid_out);                                                                                 \
    break;
  switch (reducer) {
    XH_RR(XH_RED_SUM) XH_RR(XH_RED_MEAN) XH_RR(XH_RED_MIN) XH_RR(XH_RED_MAX) XH_RR(XH_RED_STD) XH_RR(XH_RED_VAR)
    XH_RR(XH_RED_COUNT) XH_RR(XH_RED_ARGMIN) XH_RR(XH_RED_ARGMAX)
    default:
      xh_set_error("xh_resample_reduce: reducer %d not recognized", reducer);
      return XH_ERR_OP;
  }
#undef XH_RR
  XH_LAUNCH_CHECK();
  return XH_OK;
}

// ---- missing mask ----------------------------------------------------------------------------------
__global__ void __launch_bounds__(XH_BLOCK)
k_apply_missing_mask(const void* __restrict__ value, int kind, const int32_t* __restrict__ valid,
                     const int32_t* __restrict__ expected, int P, int64_t C, double* __restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  int64_t n = (int64_t)P * C;
  if (i >= n) return;
  int p = (int)(i / C);
  double v = kind == 0 ? (double)reinterpret_cast<const int32_t*>(value)[i]
                       : (double)reinterpret_cast<const float*>(value)[i];
  out[i] = (valid[i] != expected[p]) ? xh_nan64() : v;
}

// ---- rolling reductions --------------------------------------------------------------------------------
// Reference: select_rolling_resample_op (indices/generic.py:128-174): da.rolling(time=w, center=c).<op>() with
// xarray defaults min_periods = window: any NaN or an incomplete window gives NaN ("count" is the exception:
// xarray's rolling count uses min_periods=0 -> number of valid values of the partial window, NaN never).
// Window staged in registers: each lane re-reads its last `w` values from L2 (w*4 B/step) — kept simple; the
// HBM traffic stays 4 B in + 4 B out per cell-timestep.
template <int RED>
__global__ void __launch_bounds__(XH_BLOCK)
k_rolling_reduce(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, int window, int left, int right,
                 float* __restrict__ out, int64_t out_st) {
  int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  // time chunking over blockIdx.y
  int64_t chunk = cdiv64(T, (int64_t)gridDim.y);
  int64_t ta = (int64_t)blockIdx.y * chunk, tb = ta + chunk;
  if (tb > T) tb = T;
  for (int64_t t = ta; t < tb; ++t) {
    int64_t a = t - left, b = t + right;  // inclusive window [a, b]
    float r;
    if (RED == XH_RED_COUNT) {
      int n = 0;
      for (int64_t k = (a < 0 ? 0 : a); k <= b && k < T; ++k) {
        float v = x[k * st + c];
        n += (v == v) ? 1 : 0;
      }
      r = (float)n;
    } else if (a < 0 || b >= T) {
      r = xh_nan32();
    } else {
      double s = 0.0;
      float e = x[a * st + c];
      bool nan = false;
      for (int64_t k = a; k <= b; ++k) {
        float v = x[k * st + c];
        nan |= (v != v);
        if (RED == XH_RED_MIN) e = v < e ? v : e;
        else if (RED == XH_RED_MAX) e = v > e ? v : e;
        else s += (double)v;
      }
      if (RED == XH_RED_SUM) r = (float)s;
      else if (RED == XH_RED_MEAN) r = (float)(s / (double)window);
      else if (RED == XH_RED_MIN || RED == XH_RED_MAX) r = e;
      else {
        double m = s / (double)window, s2 = 0.0;
        for (int64_t k = a; k <= b; ++k) {
          double d = (double)x[k * st + c] - m;
          s2 += d * d;
        }
        r = (RED == XH_RED_VAR) ? (float)(s2 / (double)window) : (float)sqrt(s2 / (double)window);
      }
      if (nan) r = xh_nan32();
    }
    out[t * out_st + c] = r;
  }
}

static int check_tc(const char* fn, xh_ctx* ctx, const void* x, int64_t T, int64_t C, int64_t st, int64_t sc) {
  XH_REQUIRE(ctx && x, XH_ERR_ARG, "%s: NULL argument", fn);
  XH_REQUIRE(T >= 0 && C >= 0, XH_ERR_ARG, "%s: negative shape", fn);
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT,
             "%s: streaming kernels need a time-major view (sc == 1, st >= C); got st=%lld sc=%lld — transpose first",
             fn, (long long)st, (long long)sc);
  return XH_OK;
}

static int upload_segments(xh_ctx* ctx, size_t* cur, const int64_t* seg_off, int P, int64_t T, const char* fn,
                           const int64_t** d_seg) {
  XH_REQUIRE(seg_off && P >= 1, XH_ERR_ARG, "%s: seg_off NULL or P < 1", fn);
  for (int p = 0; p < P; ++p)
    XH_REQUIRE(seg_off[p] <= seg_off[p + 1] && seg_off[p] >= 0 && seg_off[p + 1] <= T, XH_ERR_ARG,
               "%s: seg_off must be non-decreasing within [0, T]", fn);
  void* d = nullptr;
  int rc = xh_scratch_upload(ctx, cur, seg_off, sizeof(int64_t) * (size_t)(P + 1), &d);
  if (rc) return rc;
  *d_seg = (const int64_t*)d;
  return XH_OK;
}

extern "C" {

static int threshold_count_impl(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op, int thr_kind,
                                double thr_scalar, const void* thr_table, int64_t thr_stride, const int32_t* tidx,
                                const int64_t* seg_off, int P, int32_t* count_out, int32_t* valid_out, int ndoy) {
  int rc = check_tc("xh_threshold_count", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(op >= XH_OP_GT && op <= XH_OP_NE, XH_ERR_OP, "Operation `%d` not recognized.", op);
  XH_REQUIRE(count_out, XH_ERR_ARG, "xh_threshold_count: count_out is NULL");
  if (thr_kind >= XH_THR_DOY_F64) {
    XH_REQUIRE(thr_table && thr_stride >= C, XH_ERR_ARG, "xh_threshold_count: threshold table missing or stride < C");
    if (thr_kind == XH_THR_DOY_F64 || thr_kind == XH_THR_DOY_F32)
      XH_REQUIRE(tidx, XH_ERR_ARG, "xh_threshold_count: tidx required for per-doy thresholds");
  }
  size_t cur = 0;
  const int64_t* d_seg = nullptr;
  rc = upload_segments(ctx, &cur, seg_off, P, T, "xh_threshold_count", &d_seg);
  if (rc) return rc;
  if (C == 0) return XH_OK;
  if (thr_kind == XH_THR_DOY_F64 && ndoy > 0 && !xh_diag_env("XH_TCOUNT_LEGACY")) {
    // multi-year series: one workgroup per column tile with the table slice in LDS (tcount.hip)
    rc = xh_launch_tcount_doy(ctx, x, T, C, st, op, static_cast<const double*>(thr_table), thr_stride, tidx, d_seg, seg_off, P,
                              ndoy, count_out, valid_out);
    if (rc != XH_ERR_NOTIMPL) return rc;
  }
  int vec = xh_pick_vec(x, C, st);
  if (thr_kind >= XH_THR_DOY_F64) {
    size_t esz = (thr_kind == XH_THR_DOY_F64 || thr_kind == XH_THR_FULL_F64) ? 8 : 4;
    if ((reinterpret_cast<uintptr_t>(thr_table) & 15) != 0 || (thr_stride * esz) % 16 != 0) vec = 1;
  }
  unsigned tiles = (unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK);
  const bool doy = thr_kind == XH_THR_DOY_F64 || thr_kind == XH_THR_DOY_F32;
  const int period_fast = (doy && P > 1 && tiles <= 65535u) ? 1 : 0;
  dim3 grid = period_fast ? dim3(period_grid(P), tiles) : dim3(tiles, period_grid(P));
  if (vec == 4)
    return launch_threshold_count<4>(ctx, thr_kind, grid, x, C, st, op, thr_scalar, thr_table, thr_stride, tidx, d_seg, P,
                                     count_out, valid_out, period_fast);
  return launch_threshold_count<1>(ctx, thr_kind, grid, x, C, st, op, thr_scalar, thr_table, thr_stride, tidx, d_seg, P,
                                   count_out, valid_out, period_fast);
}

int xh_threshold_count(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op, int thr_kind,
                       double thr_scalar, const void* thr_table, int64_t thr_stride, const int32_t* tidx,
                       const int64_t* seg_off, int P, int32_t* count_out, int32_t* valid_out) {
  return threshold_count_impl(ctx, x, T, C, st, sc, op, thr_kind, thr_scalar, thr_table, thr_stride, tidx, seg_off, P, count_out,
                              valid_out, 0);
}

int xh_threshold_count_doy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op,
                           const double* thr_table, int64_t thr_stride, int ndoy, const int32_t* tidx, const int64_t* seg_off,
                           int P, int32_t* count_out, int32_t* valid_out) {
  XH_REQUIRE(ndoy >= 1, XH_ERR_ARG, "xh_threshold_count_doy: the table needs at least one row (ndoy = %d)", ndoy);
  return threshold_count_impl(ctx, x, T, C, st, sc, op, XH_THR_DOY_F64, 0.0, thr_table, thr_stride, tidx, seg_off, P, count_out,
                              valid_out, ndoy);
}

int xh_domain_count(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op1, double thr1,
                    int op2, double thr2, int combine, const int64_t* seg_off, int P, int32_t* count_out,
                    int32_t* valid_out) {
  int rc = check_tc("xh_domain_count", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(op1 >= XH_OP_GT && op1 <= XH_OP_NE && op2 >= XH_OP_GT && op2 <= XH_OP_NE, XH_ERR_OP,
             "Operation `%d/%d` not recognized.", op1, op2);
  XH_REQUIRE(combine == 1 || combine == 2, XH_ERR_ARG, "xh_domain_count: combine must be 1 (and) or 2 (or)");
  XH_REQUIRE(count_out, XH_ERR_ARG, "xh_domain_count: count_out is NULL");
  size_t cur = 0;
  const int64_t* d_seg = nullptr;
  rc = upload_segments(ctx, &cur, seg_off, P, T, "xh_domain_count", &d_seg);
  if (rc) return rc;
  if (C == 0) return XH_OK;
  int vec = xh_pick_vec(x, C, st);
  dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), period_grid(P));
  const XhOneCmp c1 = xh_one_cmp(op1, (float)thr1), c2 = xh_one_cmp(op2, (float)thr2);
  if (vec == 4 && c1.ok && c2.ok) {
    int s1, s2;
    memcpy(&s1, &c1.sgn, 4);
    memcpy(&s2, &c2.sgn, 4);
    hipLaunchKernelGGL((k_domain_count<4, true>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, s1, c1.thr, s2, c2.thr, combine,
                       d_seg, P, count_out, valid_out);
  } else if (vec == 4)
    hipLaunchKernelGGL((k_domain_count<4>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, op1, (float)thr1, op2,
                       (float)thr2, combine, d_seg, P, count_out, valid_out);
  else
    hipLaunchKernelGGL((k_domain_count<1>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, op1, (float)thr1, op2,
                       (float)thr2, combine, d_seg, P, count_out, valid_out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_resample_reduce(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int reducer, int skipna,
                       const int64_t* seg_off, int P, void* out, int32_t* valid_out) {
  int rc = check_tc("xh_resample_reduce", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(out, XH_ERR_ARG, "xh_resample_reduce: out is NULL");
  size_t cur = 0;
  const int64_t* d_seg = nullptr;
  rc = upload_segments(ctx, &cur, seg_off, P, T, "xh_resample_reduce", &d_seg);
  if (rc) return rc;
  if (C == 0) return XH_OK;
  int vec = xh_pick_vec(x, C, st);
  dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), period_grid(P));
  if (vec == 4) return launch_resample_reduce<4>(ctx, reducer, grid, x, C, st, skipna, d_seg, P, out, valid_out);
  return launch_resample_reduce<1>(ctx, reducer, grid, x, C, st, skipna, d_seg, P, out, valid_out);
}

int xh_apply_missing_mask(xh_ctx* ctx, const void* value, int value_kind, const int32_t* valid, const int32_t* expected,
                          int P, int64_t C, double* out64) {
  XH_REQUIRE(ctx && value && valid && expected && out64, XH_ERR_ARG, "xh_apply_missing_mask: NULL argument");
  XH_REQUIRE(value_kind == 0 || value_kind == 1, XH_ERR_ARG, "xh_apply_missing_mask: value_kind must be 0 or 1");
  XH_REQUIRE(P >= 1 && C >= 0, XH_ERR_ARG, "xh_apply_missing_mask: bad shape");
  size_t cur = 0;
  void* d_exp = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, expected, sizeof(int32_t) * (size_t)P, &d_exp);
  if (rc) return rc;
  int64_t n = (int64_t)P * C;
  if (n == 0) return XH_OK;
  hipLaunchKernelGGL(k_apply_missing_mask, dim3((unsigned)cdiv64(n, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, value,
                     value_kind, valid, (const int32_t*)d_exp, P, C, out64);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_rolling_reduce(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window, int center,
                      int reducer, float* out, int64_t out_st) {
  int rc = check_tc("xh_rolling_reduce", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(out && out_st >= C, XH_ERR_ARG, "xh_rolling_reduce: out NULL or out_st < C");
  XH_REQUIRE(window >= 1, XH_ERR_ARG, "xh_rolling_reduce: window must be >= 1");
  if (T == 0 || C == 0) return XH_OK;
  // xarray: center=True -> window covers [t - w//2, t + w - 1 - w//2]; else trailing [t - w + 1, t]
  int left = center ? window / 2 : window - 1;
  int right = window - 1 - left;
  {  // windows of up to 8 steps: register ring, every row read once (window.hip)
    int rr = xh_launch_rolling_ring(ctx, x, T, C, st, window, left, right, reducer, out, out_st);
    if (rr != XH_ERR_NOTIMPL) return rr;
  }
  unsigned ny = (unsigned)(T < 64 ? T : 64);
  dim3 grid((unsigned)cdiv64(C, XH_BLOCK), ny);
#define XH_RO(R)                                                                                                      \
  case R:                                                                                                             \
    hipLaunchKernelGGL((k_rolling_reduce<R>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, window, left, right, out, \
                       out_st);                                                                                       \
    break;
  switch (reducer) {
    XH_RO(XH_RED_SUM) XH_RO(XH_RED_MEAN) XH_RO(XH_RED_MIN) XH_RO(XH_RED_MAX) XH_RO(XH_RED_STD) XH_RO(XH_RED_VAR)
    XH_RO(XH_RED_COUNT)
    default:
      xh_set_error("xh_rolling_reduce: reducer %d not recognized", reducer);
      return XH_ERR_OP;
  }
#undef XH_RO
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // extern "C"
