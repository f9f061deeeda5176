// detrend.hip — the two elementwise / reduction pieces DetrendedQuantileMapping adds around the EQM kernels
// (xsdba._adjustment.dqm_train / dqm_adjust, xsdba.detrending.PolyDetrend).  xsdba is not in the
// reference tree (src/xclim/sdba.py:10 re-exports it): PARITY UNPINNED, restated in oracle/sdba.py.
//
//   xh_poly_trend   per-cell least-squares polynomial of degree 0 or 1 over the valid samples of a series
//                   (DataArray.polyfit(dim="time", deg) with NaNs skipped): p0 + p1 (t - tc), tc = (T - 1) / 2
//   xh_trend_apply  out = x OP (p0[c] + p1[c] (t - tc)), OP in {+, -, *, /}: apply_correction with a per-cell constant
//                   (p1 NULL: the scaling / normalisation of dqm_train) or with the trend (detrend / retrend)
// Both stream the time-major (T, C) field once: a lane owns VEC cells and marches along time.
#include "common.h"

namespace {

// sums in fp64 over the valid samples with the CENTRED step index u = t - tc (|u| <= T / 2: u and u^2 are exact):
// n, Su, Suu, Sx, Sux;  slope = (n Sux - Su Sx) / (n Suu - Su^2), intercept at u = 0: (Sx - slope Su) / n
template <int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_poly_trend(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, int degree, double tc, double* __restrict__ p0,
             double* __restrict__ p1, int32_t* __restrict__ nvalid, const double* __restrict__ ucoord) {
  const int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  double n[VEC], su[VEC], suu[VEC], sx[VEC], sux[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) n[i] = su[i] = suu[i] = sx[i] = sux[i] = 0.0;
  xh_march_rows<VEC, 8>(x + c, st, 0, T, [&](int64_t t, const VecF<VEC>& xv) {
    const double u = ucoord ? ucoord[t] : (double)t - tc;  // (ucoord: the rows' own time coordinate, e.g. a group's steps)
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float f = xv.v[i];
      const bool ok = f == f;
      const double v = ok ? (double)f : 0.0, w = ok ? 1.0 : 0.0;
      n[i] += w;
      su[i] += w * u;
      suu[i] += w * u * u;
      sx[i] += v;
      sux[i] += u * v;
    }
  });
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    double a = xh_nan64(), b = 0.0;
    if (n[i] > 0.0) {
      if (degree == 0) a = sx[i] / n[i];
      else {
        const double den = n[i] * suu[i] - su[i] * su[i];
        if (den > 0.0) {
          b = (n[i] * sux[i] - su[i] * sx[i]) / den;
          a = (sx[i] - b * su[i]) / n[i];
        } else a = sx[i] / n[i];  // one valid step (or all at the same step): the fit degenerates to the mean, slope 0
      }
    }
    p0[c + i] = a;
    if (p1) p1[c + i] = (n[i] > 0.0) ? b : xh_nan64();
    if (nvalid) nvalid[c + i] = (int32_t)n[i];
  }
}

template <int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_trend_apply(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, const double* __restrict__ p0,
              const double* __restrict__ p1, double tc, int mode, float* __restrict__ out, int64_t out_st,
              const double* __restrict__ ucoord) {
  const int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const int64_t chunk = cdiv64(T, (int64_t)gridDim.y);
  const int64_t ta = (int64_t)blockIdx.y * chunk;
  int64_t tb = ta + chunk;
  if (tb > T) tb = T;
  if (ta >= tb) return;
  double a[VEC], b[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { a[i] = p0[c + i]; b[i] = p1 ? p1[c + i] : 0.0; }
  xh_march_rows<VEC, 8>(x + c, st, ta, tb, [&](int64_t t, const VecF<VEC>& xv) {
    const double u = ucoord ? ucoord[t] : (double)t - tc;
    float r[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const double tr = a[i] + b[i] * u, v = (double)xv.v[i];
      const double o = mode == 0 ? v + tr : (mode == 1 ? v - tr : (mode == 2 ? v * tr : v / tr));
      r[i] = (float)o;
    }
    float* dst = out + t * out_st + c;
    if (VEC == 4) *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
    else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) dst[i] = r[i];
    }
  });
}

// ---- the same two pieces over GROUPS of rows in ONE launch (DetrendedQuantileMapping.adjust with a sub-grouping: 365 day-of-year
// groups took 365 x 4 launches on gathered blocks — launch-bound, 87 ms for a 30-year 1440 x 90 band).  The rows stay where they
// are: a group is a list of row numbers (its steps in time order), the coefficients are (G, C) tables.
//   k_poly_trend_groups   thread = VEC cells of one group: the sums of k_poly_trend over the group's rows, in the list's order
//                         (bit-identical to xh_poly_trend_u on the gathered block), u[t] = the row's own coordinate
//   k_trend_apply_groups  thread = VEC cells of one group: out[t, c] = x[t, c] OP (p0[g, c] + p1[g, c] u[t]) for the group's rows t
//                         (the coefficients are read once per group, not once per row; rows in no group are not written)
template <int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_poly_trend_groups(const float* __restrict__ x, int64_t C, int64_t st, const int32_t* __restrict__ rows,
                    const int64_t* __restrict__ offs, const double* __restrict__ ucoord, int degree, double* __restrict__ p0,
                    double* __restrict__ p1) {
  const int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const int64_t g = blockIdx.y;
  const int64_t k0 = offs[g], k1 = offs[g + 1];
  double n[VEC], su[VEC], suu[VEC], sx[VEC], sux[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) n[i] = su[i] = suu[i] = sx[i] = sux[i] = 0.0;
  auto take = [&](double u, const VecF<VEC>& xv) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float f = xv.v[i];
      const bool ok = f == f;
      const double v = ok ? (double)f : 0.0, w = ok ? 1.0 : 0.0;
      n[i] += w;
      su[i] += w * u;
      suu[i] += w * u * u;
      sx[i] += v;
      sux[i] += u * v;
    }
  };
  int64_t k = k0;
  for (; k + 4 <= k1; k += 4) {   // four rows in flight
    const int32_t t0 = rows[k], t1 = rows[k + 1], t2 = rows[k + 2], t3 = rows[k + 3];
    const VecF<VEC> a0 = xh_load<VEC>(x + (int64_t)t0 * st + c), a1 = xh_load<VEC>(x + (int64_t)t1 * st + c),
                    a2 = xh_load<VEC>(x + (int64_t)t2 * st + c), a3 = xh_load<VEC>(x + (int64_t)t3 * st + c);
    take(ucoord[t0], a0);
    take(ucoord[t1], a1);
    take(ucoord[t2], a2);
    take(ucoord[t3], a3);
  }
  for (; k < k1; ++k) {
    const int32_t t = rows[k];
    take(ucoord[t], xh_load<VEC>(x + (int64_t)t * st + c));
  }
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    double a = xh_nan64(), b = 0.0;
    if (n[i] > 0.0) {
      if (degree == 0) a = sx[i] / n[i];
      else {
        const double den = n[i] * suu[i] - su[i] * su[i];
        if (den > 0.0) {
          b = (n[i] * sux[i] - su[i] * sx[i]) / den;
          a = (sx[i] - b * su[i]) / n[i];
        } else a = sx[i] / n[i];
      }
    }
    p0[g * C + c + i] = a;
    if (p1) p1[g * C + c + i] = (n[i] > 0.0) ? b : xh_nan64();
  }
}

template <int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_trend_apply_groups(const float* x, int64_t C, int64_t st, const int32_t* __restrict__ rows,
                     const int64_t* __restrict__ offs, const double* __restrict__ ucoord, const double* __restrict__ p0,
                     const double* __restrict__ p1, int mode, float* out, int64_t out_st) {   // (x may be out: no __restrict__)
  const int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const int64_t g = blockIdx.y;
  const int64_t k0 = offs[g], k1 = offs[g + 1];
  double a[VEC], b[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) { a[i] = p0[g * C + c + i]; b[i] = p1 ? p1[g * C + c + i] : 0.0; }
  auto put = [&](int32_t t, const VecF<VEC>& xv) {
    const double u = ucoord ? ucoord[t] : 0.0;
    float r[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const double tr = a[i] + b[i] * u, v = (double)xv.v[i];
      const double o = mode == 0 ? v + tr : (mode == 1 ? v - tr : (mode == 2 ? v * tr : v / tr));
      r[i] = (float)o;
    }
    float* dst = out + (int64_t)t * out_st + c;
    if (VEC == 4) *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
    else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) dst[i] = r[i];
    }
  };
  int64_t k = k0;
  for (; k + 4 <= k1; k += 4) {
    const int32_t t0 = rows[k], t1 = rows[k + 1], t2 = rows[k + 2], t3 = rows[k + 3];
    const VecF<VEC> a0 = xh_load<VEC>(x + (int64_t)t0 * st + c), a1 = xh_load<VEC>(x + (int64_t)t1 * st + c),
                    a2 = xh_load<VEC>(x + (int64_t)t2 * st + c), a3 = xh_load<VEC>(x + (int64_t)t3 * st + c);
    put(t0, a0);
    put(t1, a1);
    put(t2, a2);
    put(t3, a3);
  }
  for (; k < k1; ++k) {
    const int32_t t = rows[k];
    put(t, xh_load<VEC>(x + (int64_t)t * st + c));
  }
}

}  // namespace

// Centred window mean over the valid samples (xsdba.detrending._polydetrend_get_trend with a windowed Grouper:
// rolling(time=window, center=True).construct("window") pads the ends with NaN, then da.mean over the window dimension skips
// NaN): out[t, c] = mean { x[s, c] : |s - t| <= window / 2, 0 <= s < T, x[s, c] valid }, NaN when the window holds no valid
// sample.  A thread = one column x one stretch of WM_ROWS output rows: a running float64 sum and count (float32 samples: every
// add / subtract is exact to 2^-53 of the sum), the window's first rows summed up front.
constexpr int WM_ROWS = 128;
__global__ void __launch_bounds__(XH_BLOCK)
k_window_nanmean(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, int half, float* __restrict__ out, int64_t ost) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  const int64_t t0 = (int64_t)blockIdx.y * WM_ROWS;
  int64_t t1 = t0 + WM_ROWS;
  if (t1 > T) t1 = T;
  double sum = 0.0;
  int cnt = 0;
  for (int64_t s = t0 - half < 0 ? 0 : t0 - half; s < t0 + half && s < T; ++s) {  // rows [t0 - half, t0 + half): the next one enters in the loop
    const float v = x[s * st + c];
    if (v == v) { sum += (double)v; ++cnt; }
  }
  for (int64_t t = t0; t < t1; ++t) {
    const int64_t sin = t + half, sout = t - half - 1;
    if (sin < T) {
      const float v = x[sin * st + c];
      if (v == v) { sum += (double)v; ++cnt; }
    }
    if (sout >= 0 && sout >= t0 - half) {
      const float v = x[sout * st + c];
      if (v == v) { sum -= (double)v; --cnt; }
    }
    // (a window that lost all its samples: the running sum is rounding residue, not data)
    if (cnt == 0) sum = 0.0;
    out[t * ost + c] = cnt > 0 ? (float)(sum / (double)cnt) : xh_nan32();
  }
}

extern "C" {

static int poly_trend_impl(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int degree, double* p0,
                           double* p1, int32_t* nvalid, const double* ucoord) {
  XH_REQUIRE(ctx && x && p0, XH_ERR_ARG, "xh_poly_trend: NULL argument");
  XH_REQUIRE(T >= 1 && C >= 0, XH_ERR_ARG, "xh_poly_trend: bad shape");
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT, "xh_poly_trend: needs a time-major view (sc == 1)");
  XH_REQUIRE(degree == 0 || degree == 1, XH_ERR_NOTIMPL, "xh_poly_trend: degree must be 0 or 1");
  XH_REQUIRE(degree == 0 || p1, XH_ERR_ARG, "xh_poly_trend: p1 is NULL");
  if (C == 0) return XH_OK;
  const double tc = 0.5 * (double)(T - 1);
  const bool v4 = xh_pick_vec(x, C, st) == 4 && cdiv64(cdiv64(C, 4), XH_BLOCK) >= 2 * (int64_t)ctx->num_cu;
  if (v4)
    hipLaunchKernelGGL((k_poly_trend<4>), dim3((unsigned)cdiv64(cdiv64(C, 4), XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st,
                       degree, tc, p0, p1, nvalid, ucoord);
  else
    hipLaunchKernelGGL((k_poly_trend<1>), dim3((unsigned)cdiv64(C, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, degree,
                       tc, p0, p1, nvalid, ucoord);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_poly_trend(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int degree, double* p0,
                  double* p1, int32_t* nvalid) {
  return poly_trend_impl(ctx, x, T, C, st, sc, degree, p0, p1, nvalid, nullptr);
}

// the same with the rows' own coordinate u[t] (DEVICE float64, T; e.g. days since the mean date of a group's steps) instead
// of the centred row number: PolyDetrend fitted per group on a gathered block of rows (xsdba.detrending.PolyDetrend with a
// sub-grouping: DataArray.polyfit over the group's time coordinate)
int xh_poly_trend_u(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int degree, const double* u,
                    double* p0, double* p1, int32_t* nvalid) {
  XH_REQUIRE(u, XH_ERR_ARG, "xh_poly_trend_u: NULL coordinate");
  return poly_trend_impl(ctx, x, T, C, st, sc, degree, p0, p1, nvalid, u);
}

static int trend_apply_impl(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const double* p0,
                            const double* p1, int mode, float* out, int64_t out_st, const double* ucoord) {
  XH_REQUIRE(ctx && x && p0 && out, XH_ERR_ARG, "xh_trend_apply: NULL argument");
  XH_REQUIRE(T >= 0 && C >= 0, XH_ERR_ARG, "xh_trend_apply: bad shape");
  XH_REQUIRE(sc == 1 && st >= C && out_st >= C, XH_ERR_LAYOUT, "xh_trend_apply: needs time-major views (sc == 1)");
  XH_REQUIRE(mode >= 0 && mode <= 3, XH_ERR_ARG, "xh_trend_apply: mode must be 0 (+), 1 (-), 2 (*) or 3 (/)");
  if (T == 0 || C == 0) return XH_OK;
  const double tc = 0.5 * (double)(T - 1);
  const int vec = (xh_pick_vec(x, C, st) == 4 && xh_pick_vec(out, C, out_st) == 4) ? 4 : 1;
  const int64_t cblocks = cdiv64(cdiv64(C, vec), XH_BLOCK);
  int64_t gy = cdiv64((int64_t)ctx->num_cu * 12, cblocks);
  if (gy < 1) gy = 1;
  if (gy > cdiv64(T, 32)) gy = cdiv64(T, 32);
  if (gy < 1) gy = 1;
  const dim3 grid((unsigned)cblocks, (unsigned)gy);
  if (vec == 4)
    hipLaunchKernelGGL((k_trend_apply<4>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, p0, p1, tc, mode, out, out_st, ucoord);
  else
    hipLaunchKernelGGL((k_trend_apply<1>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, p0, p1, tc, mode, out, out_st, ucoord);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_trend_apply(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const double* p0,
                   const double* p1, int mode, float* out, int64_t out_st) {
  return trend_apply_impl(ctx, x, T, C, st, sc, p0, p1, mode, out, out_st, nullptr);
}

int xh_trend_apply_u(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const double* u, const double* p0,
                     const double* p1, int mode, float* out, int64_t out_st) {
  XH_REQUIRE(u, XH_ERR_ARG, "xh_trend_apply_u: NULL coordinate");
  return trend_apply_impl(ctx, x, T, C, st, sc, p0, p1, mode, out, out_st, u);
}

// PolyDetrend with a windowed sub-grouping fits the trend on the WINDOW MEAN of every step (see k_window_nanmean)
int xh_window_nanmean(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window, float* out, int64_t out_st) {
  XH_REQUIRE(ctx && x && out, XH_ERR_ARG, "xh_window_nanmean: NULL argument");
  XH_REQUIRE(T >= 0 && C >= 0, XH_ERR_ARG, "xh_window_nanmean: bad shape");
  XH_REQUIRE(sc == 1 && st >= C && out_st >= C, XH_ERR_LAYOUT, "xh_window_nanmean: needs time-major views (sc == 1)");
  XH_REQUIRE(window >= 1 && (window & 1), XH_ERR_ARG, "xh_window_nanmean: window must be a positive odd number of steps");
  XH_REQUIRE(x != out, XH_ERR_ARG, "xh_window_nanmean: not in place");
  if (T == 0 || C == 0) return XH_OK;
  const dim3 grid((unsigned)cdiv64(C, XH_BLOCK), (unsigned)cdiv64(T, WM_ROWS));
  hipLaunchKernelGGL(k_window_nanmean, grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, window / 2, out, out_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

// PolyDetrend per group in one launch: rows (host, offs[G] entries): the row numbers of group 0, then of group 1, ... (a group's
// rows in the order its sums are taken), offs (host, G + 1): where each group's rows start; u (DEVICE float64, T): the
// coordinate of every ROW of x (e.g. days since the mean date of the row's group).  p0, p1 (G, C) float64 (p1 NULL for degree 0).
// Bit-identical to xh_poly_trend_u on each group's gathered rows.
int xh_poly_trend_groups(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* rows, const int64_t* offs, int G,
                         const double* u, int degree, double* p0, double* p1) {
  XH_REQUIRE(ctx && x && rows && offs && u && p0, XH_ERR_ARG, "xh_poly_trend_groups: NULL argument");
  XH_REQUIRE(T >= 1 && C >= 0 && G >= 1 && st >= C, XH_ERR_ARG, "xh_poly_trend_groups: bad shape");
  XH_REQUIRE(degree == 0 || degree == 1, XH_ERR_NOTIMPL, "xh_poly_trend_groups: degree must be 0 or 1");
  XH_REQUIRE(degree == 0 || p1, XH_ERR_ARG, "xh_poly_trend_groups: p1 is NULL");
  XH_REQUIRE(offs[0] == 0, XH_ERR_ARG, "xh_poly_trend_groups: offs[0] must be 0");
  for (int g = 0; g < G; ++g) XH_REQUIRE(offs[g + 1] >= offs[g], XH_ERR_ARG, "xh_poly_trend_groups: offs must not decrease");
  const int64_t nr = offs[G];
  XH_REQUIRE(nr < (1ll << 31), XH_ERR_ARG, "xh_poly_trend_groups: too many rows");
  for (int64_t k = 0; k < nr; ++k) XH_REQUIRE(rows[k] >= 0 && rows[k] < T, XH_ERR_ARG, "xh_poly_trend_groups: row %lld out of range", (long long)k);
  if (C == 0) return XH_OK;
  size_t cur = 0;
  void *d_rows = nullptr, *d_offs = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, offs, sizeof(int64_t) * (size_t)(G + 1), &d_offs);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, rows, sizeof(int32_t) * (size_t)(nr > 0 ? nr : 1), &d_rows);
  if (rc) return rc;
  if (xh_pick_vec(x, C, st) == 4)
    hipLaunchKernelGGL((k_poly_trend_groups<4>), dim3((unsigned)cdiv64(cdiv64(C, 4), XH_BLOCK), (unsigned)G), dim3(XH_BLOCK), 0, ctx->stream, x,
                       C, st, (const int32_t*)d_rows, (const int64_t*)d_offs, u, degree, p0, p1);
  else
    hipLaunchKernelGGL((k_poly_trend_groups<1>), dim3((unsigned)cdiv64(C, XH_BLOCK), (unsigned)G), dim3(XH_BLOCK), 0, ctx->stream, x, C, st,
                       (const int32_t*)d_rows, (const int64_t*)d_offs, u, degree, p0, p1);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

// apply_correction with the coefficients of every row's GROUP: out[t, c] = x[t, c] OP (p0[g, c] + p1[g, c] u[t]) for the rows t of
// group g (rows / offs as in xh_poly_trend_groups; rows in no group are left as they are; a row listed twice is written twice with
// the same value only if it is in one group — do not list a row in two groups); u (DEVICE float64, T) or NULL with p1 NULL (a
// per-group constant: the scaling of dqm_adjust); p0, p1 (G, C) float64.  x == out is allowed (elementwise).  Bit-identical to
// xh_trend_apply_u on each group's gathered rows.
int xh_trend_apply_groups(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* rows, const int64_t* offs, int G,
                          const double* u, const double* p0, const double* p1, int mode, float* out, int64_t out_st) {
  XH_REQUIRE(ctx && x && rows && offs && p0 && out, XH_ERR_ARG, "xh_trend_apply_groups: NULL argument");
  XH_REQUIRE(T >= 1 && C >= 0 && G >= 1 && st >= C && out_st >= C, XH_ERR_ARG, "xh_trend_apply_groups: bad shape");
  XH_REQUIRE(mode >= 0 && mode <= 3, XH_ERR_ARG, "xh_trend_apply_groups: mode must be 0 (+), 1 (-), 2 (*) or 3 (/)");
  XH_REQUIRE(u || !p1, XH_ERR_ARG, "xh_trend_apply_groups: a slope table needs the rows' coordinate");
  XH_REQUIRE(offs[0] == 0, XH_ERR_ARG, "xh_trend_apply_groups: offs[0] must be 0");
  for (int g = 0; g < G; ++g) XH_REQUIRE(offs[g + 1] >= offs[g], XH_ERR_ARG, "xh_trend_apply_groups: offs must not decrease");
  const int64_t nr = offs[G];
  XH_REQUIRE(nr < (1ll << 31), XH_ERR_ARG, "xh_trend_apply_groups: too many rows");
  for (int64_t k = 0; k < nr; ++k) XH_REQUIRE(rows[k] >= 0 && rows[k] < T, XH_ERR_ARG, "xh_trend_apply_groups: row %lld out of range", (long long)k);
  if (C == 0 || nr == 0) return XH_OK;
  size_t cur = 0;
  void *d_rows = nullptr, *d_offs = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, offs, sizeof(int64_t) * (size_t)(G + 1), &d_offs);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, rows, sizeof(int32_t) * (size_t)nr, &d_rows);
  if (rc) return rc;
  const int vec = (xh_pick_vec(x, C, st) == 4 && xh_pick_vec(out, C, out_st) == 4) ? 4 : 1;
  const dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), (unsigned)G);
  if (vec == 4)
    hipLaunchKernelGGL((k_trend_apply_groups<4>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, (const int32_t*)d_rows, (const int64_t*)d_offs, u,
                       p0, p1, mode, out, out_st);
  else
    hipLaunchKernelGGL((k_trend_apply_groups<1>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, (const int32_t*)d_rows, (const int64_t*)d_offs, u,
                       p0, p1, mode, out, out_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // extern "C"
