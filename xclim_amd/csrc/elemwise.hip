// elemwise.hip — two-variable range reductions and the elementwise compare map.
//
// Reference (xclim/indices/generic.py):
//   diurnal_temperature_range           gen:1076-1105   reducer((high - low).resample(time=freq))
//   extreme_temperature_range           gen:1388-1414   high.resample.max() - low.resample.min()
//   interday_diurnal_temperature_range  gen:1360-1385   abs((high - low).diff("time")).resample(time=freq).mean()
//   compare / get_daily_events          gen:301-326, 395-431
// Time-major (T, C), one lane owns VEC consecutive cells, periods on blockIdx.y.  (high - low) and the day-to-day
// difference are formed in fp32 like the reference (fp32 arrays), sums accumulate in fp64.  All reducers skip NaN
// (xarray's default for floats): empty / all-NaN period -> NaN, except sum -> 0.
#include "common.h"

template <int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_range_reduce(const float* __restrict__ lo, const float* __restrict__ hi, int64_t C, int64_t st_lo, int64_t st_hi, int mode,
               int reducer, const int64_t* __restrict__ seg_off, int P, float* __restrict__ out,
               int32_t* __restrict__ valid_out) {
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    const int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    double s[VEC];
    float e1[VEC], e2[VEC], prev[VEC];
    int n[VEC], n2[VEC], val[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      s[i] = 0.0; n[i] = 0; n2[i] = 0; val[i] = 0; e1[i] = 0.f; e2[i] = 0.f;
      prev[i] = xh_nan32();
    }
    if (mode == 1 && t0 > 0) {  // the difference at the first day of the period uses the last day of the previous one
      VecF<VEC> a = xh_load<VEC>(lo + (t0 - 1) * st_lo + c), b = xh_load<VEC>(hi + (t0 - 1) * st_hi + c);
#pragma unroll
      for (int i = 0; i < VEC; ++i) prev[i] = b.v[i] - a.v[i];
    }
#pragma unroll 4
    for (int64_t t = t0; t < t1; ++t) {
      VecF<VEC> a = xh_load<VEC>(lo + t * st_lo + c), b = xh_load<VEC>(hi + t * st_hi + c);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float l = a.v[i], h = b.v[i];
        val[i] += (l == l && h == h) ? 1 : 0;
        if (mode == 2) {  // extreme range: max(high), min(low) tracked separately
          if (h == h) { e1[i] = (n[i] == 0 || h > e1[i]) ? h : e1[i]; n[i]++; }
          if (l == l) { e2[i] = (n2[i] == 0 || l < e2[i]) ? l : e2[i]; n2[i]++; }
        } else {
          const float d = h - l;
          float v = d;
          if (mode == 1) {
            v = fabsf(d - prev[i]);  // NaN on the very first day of the series (diff drops it) and next to NaNs
            prev[i] = d;
          }
          if (v == v) {
            s[i] += (double)v;
            e1[i] = (n[i] == 0 || (reducer == XH_RED_MIN ? v < e1[i] : v > e1[i])) ? v : e1[i];
            n[i]++;
          }
        }
      }
    }
    const int64_t o = (int64_t)p * C + c;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float r;
      if (mode == 2) r = (n[i] == 0 || n2[i] == 0) ? xh_nan32() : e1[i] - e2[i];
      else if (mode == 1 || reducer == XH_RED_MEAN) r = n[i] == 0 ? xh_nan32() : (float)(s[i] / (double)n[i]);
      else if (reducer == XH_RED_SUM) r = (float)s[i];
      else r = n[i] == 0 ? xh_nan32() : e1[i];
      out[o + i] = r;
      if (valid_out) valid_out[o + i] = val[i];
    }
  }
}

// out_kind 0: uint8 mask (compare)        1: float 1/0, NaN where a is NaN (get_daily_events)
//          2: float a where the condition holds, NaN elsewhere (da.where(cond))     3: float 1/0 (bool mask as float)
// A lane owns VEC consecutive cells (16-byte loads / stores when the views are aligned), the time axis is cut into
// chunks over blockIdx.y.
template <bool F64, int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_compare_map(const float* __restrict__ a, int64_t T, int64_t C, int64_t st, int op, double thr, const float* __restrict__ b,
              int64_t st_b, int out_kind, void* __restrict__ out_v, int64_t st_out) {
  const int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const int64_t chunk = cdiv64(T, (int64_t)gridDim.y);
  int64_t ta = (int64_t)blockIdx.y * chunk, tb = ta + chunk;
  if (tb > T) tb = T;
  const float thr32 = (float)thr;
#pragma unroll 4
  for (int64_t t = ta; t < tb; ++t) {
    const VecF<VEC> v = xh_load<VEC>(a + t * st + c);
    VecF<VEC> w;
    if (b) w = xh_load<VEC>(b + t * st_b + c);
    float r[VEC];
    uint8_t m[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      bool cond;
      if (b) cond = xh_cmp_f32(v.v[i], op, w.v[i]);
      else cond = F64 ? xh_cmp_f64((double)v.v[i], op, thr) : xh_cmp_f32(v.v[i], op, thr32);
      m[i] = cond ? 1 : 0;
      if (out_kind == 1) r[i] = (v.v[i] == v.v[i]) ? (cond ? 1.f : 0.f) : xh_nan32();
      else if (out_kind == 2) r[i] = cond ? v.v[i] : xh_nan32();
      else if (out_kind == 4) {  // (a - thr).clip(0): NaN stays NaN (hot_spell_max_magnitude, _threshold.py:2056-2057)
        const float dlt = v.v[i] - (b ? w.v[i] : thr32);
        r[i] = (dlt != dlt) ? dlt : (dlt > 0.f ? dlt : 0.f);
      }
      else r[i] = cond ? 1.f : 0.f;
    }
    if (out_kind == 0) {
      uint8_t* o = reinterpret_cast<uint8_t*>(out_v) + t * st_out + c;
      if (VEC == 4) *reinterpret_cast<uchar4*>(o) = make_uchar4(m[0], m[1 % VEC], m[2 % VEC], m[3 % VEC]);
      else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = m[i];
      }
    } else {
      float* o = reinterpret_cast<float*>(out_v) + t * st_out + c;
      if (VEC == 4) *reinterpret_cast<float4*>(o) = make_float4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
      else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) o[i] = r[i];
      }
    }
  }
}

// ---- per-doy tables broadcast onto the time axis (core/calendar.py) ----------------------------------------------
//   resample_doy     cal:763-790   out[t] = table[tidx[t]]  (fp64; the fused kernels never materialise this field)
//   within_bnds_doy  cal:934-954   (low[tidx[t]] < x[t]) && (x[t] < high[tidx[t]]), compared in fp64
__global__ void __launch_bounds__(XH_BLOCK)
k_doy_broadcast(const double* __restrict__ table, int64_t C, const int32_t* __restrict__ tidx, int64_t T,
                double* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  const int64_t chunk = cdiv64(T, (int64_t)gridDim.y);
  int64_t ta = (int64_t)blockIdx.y * chunk, tb = ta + chunk;
  if (tb > T) tb = T;
#pragma unroll 4
  for (int64_t t = ta; t < tb; ++t) {
    const int r = tidx[t];
    out[t * C + c] = table[(int64_t)(r < 0 ? 0 : r) * C + c];
  }
}

__global__ void __launch_bounds__(XH_BLOCK)
k_within_bnds_doy(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, const double* __restrict__ low,
                  const double* __restrict__ high, const int32_t* __restrict__ tidx, uint8_t* __restrict__ out) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  const int64_t chunk = cdiv64(T, (int64_t)gridDim.y);
  int64_t ta = (int64_t)blockIdx.y * chunk, tb = ta + chunk;
  if (tb > T) tb = T;
#pragma unroll 4
  for (int64_t t = ta; t < tb; ++t) {
    const int64_t r = tidx[t];
    const double v = (double)x[t * st + c];
    out[t * C + c] = (low[r * C + c] < v && v < high[r * C + c]) ? 1 : 0;
  }
}

// mask of `x op table[tidx[t]]` (fp64 compare against a per-doy table): the `compare(da, op, resample_doy(per, da))` step of
// the percentile-spell indices (warm/cold_spell_duration_index, indices/_multivariate.py:66-152, 1693-1793) without the
// (T, C) fp64 threshold field.  out float32 1/0.
__global__ void __launch_bounds__(XH_BLOCK)
k_compare_doy(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, int op, const double* __restrict__ table,
              const int32_t* __restrict__ tidx, float* __restrict__ out, int64_t st_out) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  const int64_t chunk = cdiv64(T, (int64_t)gridDim.y);
  int64_t ta = (int64_t)blockIdx.y * chunk, tb = ta + chunk;
  if (tb > T) tb = T;
#pragma unroll 4
  for (int64_t t = ta; t < tb; ++t) {
    const int64_t r = tidx[t];
    out[t * st_out + c] = xh_cmp_f64((double)x[t * st + c], op, table[r * C + c]) ? 1.f : 0.f;
  }
}

// days_over_precip_thresh / fraction_over_precip_thresh (indices/_multivariate.py:1220-1232, 1281-1296) in one pass:
//   tp      = table[tidx[t]] > thr ? table[tidx[t]] : thr        (fp64; NaN percentile -> thr)
//   n_over  = #(x op tp)  [fp64 compare],  over = sum of x where x op tp,  total = sum of x where x op (float)thr [fp32]
//   frac    = (float)over / (float)total                         (0/0 = NaN like the reference)
// The reference materialises tp as a (T, C) fp64 field and makes five passes; here x and the (D, C) table rows are read
// once, in batches of 8 rows whose loads are issued before any use.
__global__ void __launch_bounds__(XH_BLOCK)
k_precip_over_doy(const float* __restrict__ x, int64_t C, int64_t st, int op, double thr, const double* __restrict__ table,
                  const int32_t* __restrict__ tidx, const int64_t* __restrict__ seg_off, int P, float* __restrict__ frac,
                  int32_t* __restrict__ n_over, int32_t* __restrict__ valid) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  const float thr32 = (float)thr;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    const int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    double over = 0.0, total = 0.0;
    int cnt = 0, nv = 0;
    auto step = [&](float xv, double tv) {
      const double tp = tv > thr ? tv : thr;
      const bool o = xh_cmp_f64((double)xv, op, tp);
      const bool w = xh_cmp_f32(xv, op, thr32);
      over += o ? (double)xv : 0.0;
      total += w ? (double)xv : 0.0;
      cnt += o ? 1 : 0;
      nv += (xv == xv) ? 1 : 0;
    };
    int64_t t = t0;
    for (; t + 8 <= t1; t += 8) {
      int r[8];
      float xv[8];
      double tv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = tidx[t + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = x[(t + u) * st + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) tv[u] = table[(int64_t)r[u] * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) step(xv[u], tv[u]);
    }
    for (; t < t1; ++t) step(x[t * st + c], table[(int64_t)tidx[t] * C + c]);
    const int64_t o = (int64_t)p * C + c;
    if (frac) frac[o] = (float)over / (float)total;
    if (n_over) n_over[o] = cnt;
    if (valid) valid[o] = nv;
  }
}

// bool / uint8 mask -> the float32 1 / 0 mask the run-length kernels read: host masks cross PCIe as bytes (a quarter of
// the float form, and no host-side astype of the whole field)
__global__ void __launch_bounds__(XH_BLOCK)
k_mask_u8_to_f32(const uint8_t* __restrict__ m, int64_t n, float* __restrict__ out) {
  const int64_t i = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * 4;
  if (i + 4 <= n) {
    const uchar4 v = *reinterpret_cast<const uchar4*>(m + i);
    *reinterpret_cast<float4*>(out + i) = make_float4(v.x ? 1.f : 0.f, v.y ? 1.f : 0.f, v.z ? 1.f : 0.f, v.w ? 1.f : 0.f);
  } else {
    for (int64_t k = i; k < n; ++k) out[k] = m[k] ? 1.f : 0.f;
  }
}

// select_time (core/calendar.py:1259-1378): out row i = x row idx[i], or NaN when idx[i] < 0.  da.where(mask) is
// idx[t] = mask[t] ? t : -1 over all rows; drop=True lists the selected rows only.  Row-uniform choice, 16-byte moves.
template <int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_select_rows(const float* __restrict__ x, int64_t C, int64_t st, const int64_t* __restrict__ idx, int64_t n,
              float* __restrict__ out, int64_t st_out) {
  const int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const int64_t chunk = cdiv64(n, (int64_t)gridDim.y);
  int64_t ia = (int64_t)blockIdx.y * chunk, ib = ia + chunk;
  if (ib > n) ib = n;
#pragma unroll 4
  for (int64_t i = ia; i < ib; ++i) {
    const int64_t r = idx[i];
    VecF<VEC> v = xh_load<VEC>(x + (r < 0 ? 0 : r) * st + c);  // unconditional load, masked afterwards
    float o[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) o[k] = r < 0 ? xh_nan32() : v.v[k];
    if (VEC == 4) *reinterpret_cast<float4*>(out + i * st_out + c) = make_float4(o[0], o[1 % VEC], o[2 % VEC], o[3 % VEC]);
    else {
#pragma unroll
      for (int k = 0; k < VEC; ++k) out[i * st_out + c + k] = o[k];
    }
  }
}

static dim3 time_chunk_grid(xh_ctx* ctx, int64_t T, int64_t C) {
  const int64_t cblocks = cdiv64(C, XH_BLOCK);
  int64_t gy = cdiv64((int64_t)ctx->num_cu * 16, cblocks);
  if (gy < 1) gy = 1;
  if (gy > T) gy = T;
  if (gy > 1024) gy = 1024;
  return dim3((unsigned)cblocks, (unsigned)gy);
}

static int upload_tidx(xh_ctx* ctx, const char* fn, const int32_t* tidx, int64_t T, int D, const int32_t** d_tidx) {
  XH_REQUIRE(tidx, XH_ERR_ARG, "%s: tidx is NULL", fn);
  for (int64_t t = 0; t < T; ++t)
    XH_REQUIRE(tidx[t] >= 0 && tidx[t] < D, XH_ERR_ARG, "%s: tidx[%lld] = %d outside the table (D = %d)", fn, (long long)t,
               tidx[t], D);
  size_t cur = 0;
  void* d = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, tidx, sizeof(int32_t) * (size_t)T, &d);
  if (rc) return rc;
  *d_tidx = (const int32_t*)d;
  return XH_OK;
}

// select_time(doy_bounds=(start, end)) with per-cell bounds (mask_between_doys, core/calendar.py:1244-1257, the
// "spatial dims only" case): out = x where the day of year of the step lies inside the cell's [start, end] — a span that
// wraps over the new year when start > end — else NaN.  start / end are float32 (C,), already shifted for exclusive
// bounds; NaN bounds default to 1 / 366.
template <int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_mask_doy_cells(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, const int32_t* __restrict__ doy,
                 const float* __restrict__ start, const float* __restrict__ end, float* __restrict__ out, int64_t out_st) {
  const int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const int64_t chunk = cdiv64(T, (int64_t)gridDim.y);
  const int64_t ta = (int64_t)blockIdx.y * chunk;
  int64_t tb = ta + chunk;
  if (tb > T) tb = T;
  if (ta >= tb) return;
  float s[VEC], e[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const float a = start[c + i], b = end[c + i];
    s[i] = a == a ? a : 1.0f;
    e[i] = b == b ? b : 366.0f;
  }
  xh_march_rows<VEC, 8>(x + c, st, ta, tb, [&](int64_t t, const VecF<VEC>& xv) {
    const float d = (float)doy[t];
    float r[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const bool inside = s[i] <= e[i] ? (d >= s[i] && d <= e[i]) : !(d > e[i] && d < s[i]);
      r[i] = inside ? xv.v[i] : xh_nan32();
    }
    float* dst = out + t * out_st + c;
    if (VEC == 4) *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
    else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) dst[i] = r[i];
    }
  });
}

// Trailing weighted window sum: out[t] = sum_k w[k] * x[t - W + 1 + k] in float64 (k ascending), rounded once to float32;
// NaN while the window is incomplete or holds a NaN.  The "spell value" of spell_mask with `weights` (indices/generic.py:
// 523-524: data_pad.rolling(time=window).construct("window").dot(weights)) as a field of its own — needed when the
// threshold differs per cell (the fused spell kernels of window.hip take one scalar threshold).  Rarely used: every
// output re-reads its W inputs through L2.
__global__ void __launch_bounds__(XH_BLOCK)
k_rolling_dot(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, int W, const double* __restrict__ w,
              float* __restrict__ out, int64_t out_st) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  const int64_t chunk = cdiv64(T, (int64_t)gridDim.y);
  const int64_t ta = (int64_t)blockIdx.y * chunk;
  int64_t tb = ta + chunk;
  if (tb > T) tb = T;
  for (int64_t t = ta; t < tb; ++t) {
    float r = xh_nan32();
    if (t >= W - 1) {
      double s = 0.0;
      for (int k = 0; k < W; ++k) s += w[k] * (double)x[(t - W + 1 + k) * st + c];  // a NaN sample poisons the sum
      r = (float)s;
    }
    out[t * out_st + c] = r;
  }
}

// select_time(da, doy_bounds=(start, end)) with bounds that carry a TIME dimension (mask_between_doys, core/calendar.py:
// 1211-1246): one pair of bounds per period and cell, already converted by the host to "days since the period's first
// step" (doy_to_days_since; NaN -> 0 / 366; a period without bounds: lo = +inf).  out = x where lo <= t - t0 <= hi.
template <int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_mask_days_cells(const float* __restrict__ x, int64_t C, int64_t st, const int64_t* __restrict__ seg_off, int P,
                  const float* __restrict__ lo, const float* __restrict__ hi, float* __restrict__ out, int64_t out_st) {
  const int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    const int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    float a[VEC], b[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) a[i] = lo[(int64_t)p * C + c + i], b[i] = hi[(int64_t)p * C + c + i];
    xh_march_rows<VEC, 8>(x + c, st, t0, t1, [&](int64_t t, const VecF<VEC>& xv) {
      const float d = (float)(t - t0);
      float* dst = out + t * out_st + c;
#pragma unroll
      for (int i = 0; i < VEC; ++i) dst[i] = (d >= a[i] && d <= b[i]) ? xv.v[i] : xh_nan32();
    });
  }
}

extern "C" {

int xh_range_reduce(xh_ctx* ctx, const float* low, const float* high, int64_t T, int64_t C, int64_t st_low, int64_t st_high,
                    int mode, int reducer, const int64_t* seg_off, int P, float* out, int32_t* valid_out) {
  XH_REQUIRE(ctx && low && high && out, XH_ERR_ARG, "xh_range_reduce: NULL argument");
  XH_REQUIRE(T >= 0 && C >= 0, XH_ERR_ARG, "xh_range_reduce: negative shape");
  XH_REQUIRE(st_low >= C && st_high >= C, XH_ERR_LAYOUT, "xh_range_reduce: needs time-major views (row strides >= C)");
  XH_REQUIRE(mode >= 0 && mode <= 2, XH_ERR_ARG, "xh_range_reduce: mode must be 0 (range), 1 (interday) or 2 (extreme)");
  XH_REQUIRE(mode != 0 || (reducer >= XH_RED_SUM && reducer <= XH_RED_MAX), XH_ERR_OP,
             "xh_range_reduce: reducer %d not recognized", reducer);
  XH_REQUIRE(seg_off && P >= 1, XH_ERR_ARG, "xh_range_reduce: seg_off NULL or P < 1");
  for (int p = 0; p < P; ++p)
    XH_REQUIRE(seg_off[p] <= seg_off[p + 1] && seg_off[p] >= 0 && seg_off[p + 1] <= T, XH_ERR_ARG,
               "xh_range_reduce: seg_off must be non-decreasing within [0, T]");
  size_t cur = 0;
  void* d_seg = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, seg_off, sizeof(int64_t) * (size_t)(P + 1), &d_seg);
  if (rc) return rc;
  if (C == 0) return XH_OK;
  const int vec = (xh_pick_vec(low, C, st_low) == 4 && xh_pick_vec(high, C, st_high) == 4) ? 4 : 1;
  dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), (unsigned)(P > 4096 ? 4096 : P));
  if (vec == 4)
    hipLaunchKernelGGL((k_range_reduce<4>), grid, dim3(XH_BLOCK), 0, ctx->stream, low, high, C, st_low, st_high, mode, reducer,
                       (const int64_t*)d_seg, P, out, valid_out);
  else
    hipLaunchKernelGGL((k_range_reduce<1>), grid, dim3(XH_BLOCK), 0, ctx->stream, low, high, C, st_low, st_high, mode, reducer,
                       (const int64_t*)d_seg, P, out, valid_out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_compare_map(xh_ctx* ctx, const float* a, int64_t T, int64_t C, int64_t st, int op, double thr, int thr_is_f64,
                   const float* b, int64_t st_b, int out_kind, void* out, int64_t st_out) {
  XH_REQUIRE(ctx && a && out, XH_ERR_ARG, "xh_compare_map: NULL argument");
  XH_REQUIRE(T >= 0 && C >= 0, XH_ERR_ARG, "xh_compare_map: negative shape");
  XH_REQUIRE(st >= C && st_out >= C && (!b || st_b >= C), XH_ERR_LAYOUT, "xh_compare_map: needs time-major views");
  XH_REQUIRE(op >= XH_OP_GT && op <= XH_OP_NE, XH_ERR_OP, "Operation `%d` not recognized.", op);
  XH_REQUIRE(out_kind >= 0 && out_kind <= 4, XH_ERR_ARG, "xh_compare_map: out_kind must be 0 .. 4");
  if (T == 0 || C == 0) return XH_OK;
  // 16-byte path: every view aligned (the uint8 output needs 4-byte alignment of its rows)
  int vec = xh_pick_vec(a, C, st);
  if (b && xh_pick_vec(b, C, st_b) != 4) vec = 1;
  if (out_kind == 0) {
    if ((reinterpret_cast<uintptr_t>(out) & 3) != 0 || (st_out & 3) != 0) vec = 1;
  } else if (xh_pick_vec((const float*)out, C, st_out) != 4) vec = 1;
  const int64_t cblocks = cdiv64(cdiv64(C, vec), XH_BLOCK);
  int64_t gy = cdiv64((int64_t)ctx->num_cu * 16, cblocks);
  if (gy < 1) gy = 1;
  if (gy > T) gy = T;
  if (gy > 1024) gy = 1024;
  dim3 grid((unsigned)cblocks, (unsigned)gy);
#define XH_CM(F, V)                                                                                                   \
  hipLaunchKernelGGL((k_compare_map<F, V>), grid, dim3(XH_BLOCK), 0, ctx->stream, a, T, C, st, op, thr, b, st_b, out_kind, \
                     out, st_out)
  if (thr_is_f64 && !b) {
    if (vec == 4) XH_CM(true, 4); else XH_CM(true, 1);
  } else {
    if (vec == 4) XH_CM(false, 4); else XH_CM(false, 1);
  }
#undef XH_CM
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_doy_broadcast(xh_ctx* ctx, const double* table, int D, int64_t C, const int32_t* tidx, int64_t T, double* out) {
  XH_REQUIRE(ctx && table && out, XH_ERR_ARG, "xh_doy_broadcast: NULL argument");
  XH_REQUIRE(D >= 1 && C >= 0 && T >= 0, XH_ERR_ARG, "xh_doy_broadcast: bad shape");
  if (T == 0 || C == 0) return XH_OK;
  const int32_t* d_tidx = nullptr;
  int rc = upload_tidx(ctx, "xh_doy_broadcast", tidx, T, D, &d_tidx);
  if (rc) return rc;
  hipLaunchKernelGGL(k_doy_broadcast, time_chunk_grid(ctx, T, C), dim3(XH_BLOCK), 0, ctx->stream, table, C, d_tidx, T, out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_within_bnds_doy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const double* low,
                       const double* high, int D, const int32_t* tidx, uint8_t* out) {
  XH_REQUIRE(ctx && x && low && high && out, XH_ERR_ARG, "xh_within_bnds_doy: NULL argument");
  XH_REQUIRE(D >= 1 && C >= 0 && T >= 0, XH_ERR_ARG, "xh_within_bnds_doy: bad shape");
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT, "xh_within_bnds_doy: needs a time-major view (sc == 1, st >= C)");
  if (T == 0 || C == 0) return XH_OK;
  const int32_t* d_tidx = nullptr;
  int rc = upload_tidx(ctx, "xh_within_bnds_doy", tidx, T, D, &d_tidx);
  if (rc) return rc;
  hipLaunchKernelGGL(k_within_bnds_doy, time_chunk_grid(ctx, T, C), dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, low, high,
                     d_tidx, out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_compare_doy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op, const double* table,
                   int D, const int32_t* tidx, float* out, int64_t st_out) {
  XH_REQUIRE(ctx && x && table && out, XH_ERR_ARG, "xh_compare_doy: NULL argument");
  XH_REQUIRE(D >= 1 && C >= 0 && T >= 0, XH_ERR_ARG, "xh_compare_doy: bad shape");
  XH_REQUIRE(sc == 1 && st >= C && st_out >= C, XH_ERR_LAYOUT, "xh_compare_doy: needs time-major views (sc == 1, st >= C)");
  XH_REQUIRE(op >= XH_OP_GT && op <= XH_OP_NE, XH_ERR_OP, "Operation `%d` not recognized.", op);
  if (T == 0 || C == 0) return XH_OK;
  const int32_t* d_tidx = nullptr;
  int rc = upload_tidx(ctx, "xh_compare_doy", tidx, T, D, &d_tidx);
  if (rc) return rc;
  hipLaunchKernelGGL(k_compare_doy, time_chunk_grid(ctx, T, C), dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, op, table, d_tidx,
                     out, st_out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_precip_over_doy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op, double thr,
                       const double* table, int D, const int32_t* tidx, const int64_t* seg_off, int P, float* frac,
                       int32_t* n_over, int32_t* valid_out) {
  XH_REQUIRE(ctx && x && table, XH_ERR_ARG, "xh_precip_over_doy: NULL argument");
  XH_REQUIRE(frac || n_over, XH_ERR_ARG, "xh_precip_over_doy: at least one of frac / n_over is required");
  XH_REQUIRE(T >= 0 && C >= 0 && D >= 1, XH_ERR_ARG, "xh_precip_over_doy: bad shape");
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT, "xh_precip_over_doy: needs a time-major view (cell stride 1)");
  XH_REQUIRE(op == XH_OP_GT || op == XH_OP_GE, XH_ERR_OP, "xh_precip_over_doy: operator must be > or >=");
  XH_REQUIRE(seg_off && P >= 1, XH_ERR_ARG, "xh_precip_over_doy: seg_off NULL or P < 1");
  for (int p = 0; p < P; ++p)
    XH_REQUIRE(seg_off[p] <= seg_off[p + 1] && seg_off[p] >= 0 && seg_off[p + 1] <= T, XH_ERR_ARG,
               "xh_precip_over_doy: seg_off must be non-decreasing within [0, T]");
  XH_REQUIRE(tidx, XH_ERR_ARG, "xh_precip_over_doy: tidx is NULL");
  for (int64_t t = 0; t < T; ++t)
    XH_REQUIRE(tidx[t] >= 0 && tidx[t] < D, XH_ERR_ARG, "xh_precip_over_doy: tidx[%lld] = %d outside the table (D = %d)",
               (long long)t, tidx[t], D);
  size_t cur = 0;
  void *d_tidx = nullptr, *d_seg = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, tidx, sizeof(int32_t) * (size_t)T, &d_tidx);
  if (rc) return rc;
  rc = xh_scratch_upload(ctx, &cur, seg_off, sizeof(int64_t) * (size_t)(P + 1), &d_seg);
  if (rc) return rc;
  if (C == 0) return XH_OK;
  dim3 grid((unsigned)cdiv64(C, XH_BLOCK), (unsigned)(P > 4096 ? 4096 : P));
  hipLaunchKernelGGL(k_precip_over_doy, grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, op, thr, table, (const int32_t*)d_tidx,
                     (const int64_t*)d_seg, P, frac, n_over, valid_out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_mask_u8_to_f32(xh_ctx* ctx, const uint8_t* mask, int64_t n, float* out) {
  XH_REQUIRE(ctx && (n == 0 || (mask && out)) && n >= 0, XH_ERR_ARG, "xh_mask_u8_to_f32: bad arguments");
  XH_REQUIRE(((uintptr_t)mask & 3u) == 0 && ((uintptr_t)out & 15u) == 0, XH_ERR_LAYOUT,
             "xh_mask_u8_to_f32: mask must be 4-byte and out 16-byte aligned");
  if (n == 0) return XH_OK;
  hipLaunchKernelGGL(k_mask_u8_to_f32, dim3((unsigned)cdiv64(cdiv64(n, 4), XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, mask, n, out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_select_rows(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const int64_t* idx, int64_t n,
                   float* out, int64_t st_out) {
  XH_REQUIRE(ctx && x && (n == 0 || (idx && out)), XH_ERR_ARG, "xh_select_rows: NULL argument");
  XH_REQUIRE(T >= 1 && C >= 0 && n >= 0, XH_ERR_ARG, "xh_select_rows: bad shape (T >= 1)");
  XH_REQUIRE(sc == 1 && st >= C && st_out >= C, XH_ERR_LAYOUT, "xh_select_rows: needs time-major views (cell stride 1)");
  for (int64_t i = 0; i < n; ++i)
    XH_REQUIRE(idx[i] < T, XH_ERR_ARG, "xh_select_rows: idx[%lld] = %lld outside [0, T)", (long long)i, (long long)idx[i]);
  if (n == 0 || C == 0) return XH_OK;
  size_t cur = 0;
  void* d_idx = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, idx, sizeof(int64_t) * (size_t)n, &d_idx);
  if (rc) return rc;
  const int vec = (xh_pick_vec(x, C, st) == 4 && xh_pick_vec(out, C, st_out) == 4) ? 4 : 1;
  const int64_t cblocks = cdiv64(cdiv64(C, vec), XH_BLOCK);
  int64_t gy = cdiv64((int64_t)ctx->num_cu * 16, cblocks);
  if (gy < 1) gy = 1;
  if (gy > n) gy = n;
  if (gy > 1024) gy = 1024;
  dim3 grid((unsigned)cblocks, (unsigned)gy);
  if (vec == 4)
    hipLaunchKernelGGL((k_select_rows<4>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, (const int64_t*)d_idx, n, out, st_out);
  else
    hipLaunchKernelGGL((k_select_rows<1>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, (const int64_t*)d_idx, n, out, st_out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_mask_doy_cells(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const int32_t* doy,
                      const float* start, const float* end, float* out, int64_t out_st) {
  XH_REQUIRE(ctx && x && doy && start && end && out, XH_ERR_ARG, "xh_mask_doy_cells: NULL argument");
  XH_REQUIRE(T >= 0 && C >= 0, XH_ERR_ARG, "xh_mask_doy_cells: bad shape");
  XH_REQUIRE(sc == 1 && st >= C && out_st >= C, XH_ERR_LAYOUT, "xh_mask_doy_cells: needs time-major views (sc == 1)");
  if (T == 0 || C == 0) return XH_OK;
  size_t cur = 0;
  void* d_doy = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, doy, sizeof(int32_t) * (size_t)T, &d_doy);
  if (rc) return rc;
  const int vec = (xh_pick_vec(x, C, st) == 4 && xh_pick_vec(out, C, out_st) == 4 &&
                   ((reinterpret_cast<uintptr_t>(start) | reinterpret_cast<uintptr_t>(end)) & 15) == 0) ? 4 : 1;
  const int64_t cblocks = cdiv64(cdiv64(C, vec), XH_BLOCK);
  int64_t gy = cdiv64((int64_t)ctx->num_cu * 12, cblocks);
  if (gy < 1) gy = 1;
  if (gy > cdiv64(T, 32)) gy = cdiv64(T, 32);
  if (gy < 1) gy = 1;
  const dim3 grid((unsigned)cblocks, (unsigned)gy);
  if (vec == 4)
    hipLaunchKernelGGL((k_mask_doy_cells<4>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, (const int32_t*)d_doy, start, end, out, out_st);
  else
    hipLaunchKernelGGL((k_mask_doy_cells<1>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, (const int32_t*)d_doy, start, end, out, out_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_mask_days_cells(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const int64_t* seg_off, int P,
                       const float* lo, const float* hi, float* out, int64_t out_st) {
  XH_REQUIRE(ctx && x && seg_off && lo && hi && out, XH_ERR_ARG, "xh_mask_days_cells: NULL argument");
  XH_REQUIRE(T >= 0 && C >= 0 && P >= 1, XH_ERR_ARG, "xh_mask_days_cells: bad shape");
  XH_REQUIRE(sc == 1 && st >= C && out_st >= C, XH_ERR_LAYOUT, "xh_mask_days_cells: needs time-major views (sc == 1)");
  XH_REQUIRE(seg_off[0] == 0 && seg_off[P] == T, XH_ERR_ARG, "xh_mask_days_cells: the periods must cover [0, T)");
  for (int p = 0; p < P; ++p)
    XH_REQUIRE(seg_off[p] <= seg_off[p + 1], XH_ERR_ARG, "xh_mask_days_cells: seg_off must be non-decreasing");
  if (T == 0 || C == 0) return XH_OK;
  size_t cur = 0;
  void* d_seg = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, seg_off, sizeof(int64_t) * (size_t)(P + 1), &d_seg);
  if (rc) return rc;
  const int vec = (xh_pick_vec(x, C, st) == 4 && xh_pick_vec(out, C, out_st) == 4) ? 4 : 1;
  const dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), (unsigned)(P > 4096 ? 4096 : P));
  if (vec == 4)
    hipLaunchKernelGGL((k_mask_days_cells<4>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, (const int64_t*)d_seg, P, lo, hi, out, out_st);
  else
    hipLaunchKernelGGL((k_mask_days_cells<1>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, (const int64_t*)d_seg, P, lo, hi, out, out_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_rolling_dot(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window, const double* weights,
                   float* out, int64_t out_st) {
  XH_REQUIRE(ctx && x && weights && out, XH_ERR_ARG, "xh_rolling_dot: NULL argument");
  XH_REQUIRE(T >= 0 && C >= 0 && window >= 1 && window <= 4096, XH_ERR_ARG, "xh_rolling_dot: bad shape (1 <= window <= 4096)");
  XH_REQUIRE(sc == 1 && st >= C && out_st >= C, XH_ERR_LAYOUT, "xh_rolling_dot: needs time-major views (sc == 1)");
  if (T == 0 || C == 0) return XH_OK;
  size_t cur = 0;
  void* d_w = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, weights, sizeof(double) * (size_t)window, &d_w);
  if (rc) return rc;
  const int64_t cblocks = cdiv64(C, XH_BLOCK);
  int64_t gy = cdiv64((int64_t)ctx->num_cu * 16, cblocks);
  if (gy > cdiv64(T, 16)) gy = cdiv64(T, 16);
  if (gy < 1) gy = 1;
  hipLaunchKernelGGL(k_rolling_dot, dim3((unsigned)cblocks, (unsigned)gy), dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, window,
                     (const double*)d_w, out, out_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // extern "C"
