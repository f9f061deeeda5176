// reduce2.hip — two-variable counts, thresholded reductions, day-of-year climatology (rows G3, G4, Q5 of SURVEY.md §8a).
#include <stdlib.h>

#include "common.h"
#include "pdoy.h"

// ---- bivariate counts ---------------------------------------------------------------------------------------
// count_level_crossings (gen:913-957): ((low op_low thr) & (high op_high thr)).resample.sum
// bivariate_count_occurrences (gen:1002-1073): cond1 [all -> & | any -> |] cond2, then resample.sum
// valid = days on which BOTH variables are non-NaN (MissingAny checks every input, core/missing.py:253-298).
template <int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_bivariate_count(const float* __restrict__ x1, const float* __restrict__ x2, int64_t C, int64_t st1, int64_t st2, int op1,
                  float thr1, int op2, float thr2, int combine, const int64_t* __restrict__ seg_off, int P,
                  int32_t* __restrict__ count_out, int32_t* __restrict__ valid_out) {
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    int cnt[VEC], val[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) cnt[i] = 0, val[i] = 0;
#pragma unroll 4
    for (int64_t t = t0; t < t1; ++t) {
      VecF<VEC> a = xh_load<VEC>(x1 + t * st1 + c), b = xh_load<VEC>(x2 + t * st2 + c);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        bool ca = xh_cmp_f32(a.v[i], op1, thr1), cb = xh_cmp_f32(b.v[i], op2, thr2);
        cnt[i] += ((combine == 1) ? (ca && cb) : (ca || cb)) ? 1 : 0;
        val[i] += (a.v[i] == a.v[i] && b.v[i] == b.v[i]) ? 1 : 0;
      }
    }
    int64_t o = (int64_t)p * C + c;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      count_out[o + i] = cnt[i];
      if (valid_out) valid_out[o + i] = val[i];
    }
  }
}

// ---- thresholded reductions ----------------------------------------------------------------------------------
// mode 0: thresholded_statistics (gen:1278-1320): reducer(data.where(cond)) per period (sum/mean/min/max)
// mode 1: temperature_sum (gen:1323-1357): direction * sum((data - thr).where(cond))
// mode 2: cumulative_difference (gen:1514-1552): sum(clip(data - thr, 0)) for > / >=, sum(clip(thr - data, 0)) for < / <=
// (data - thr) is formed in fp32 like the reference (fp32 array minus python float), sums accumulate in fp64.
template <int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_thresholded_reduce(const float* __restrict__ x, int64_t C, int64_t st, int op, float thr, int mode, int reducer,
                     const int64_t* __restrict__ seg_off, int P, float* __restrict__ out, int32_t* __restrict__ valid_out) {
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const bool below = (op == XH_OP_LT || op == XH_OP_LE);
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    double s[VEC];
    float ext[VEC];
    int n[VEC], val[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s[i] = 0.0, ext[i] = 0.f, n[i] = 0, val[i] = 0;
    xh_march_rows<VEC, 8>(x + c, st, t0, t1, [&](int64_t, const VecF<VEC>& xv) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float v = xv.v[i];
        val[i] += (v == v) ? 1 : 0;
        if (mode == 2) {
          float d = below ? (thr - v) : (v - thr);
          d = d < 0.0f ? 0.0f : d;  // clip(0); NaN stays NaN and is skipped by the sum
          if (d == d) s[i] += (double)d;
        } else if (xh_cmp_f32(v, op, thr)) {
          float d = (mode == 1) ? (v - thr) : v;
          s[i] += (double)d;
          if (n[i] == 0) ext[i] = d;
          if (reducer == XH_RED_MIN) ext[i] = d < ext[i] ? d : ext[i];
          if (reducer == XH_RED_MAX) ext[i] = d > ext[i] ? d : ext[i];
          n[i]++;
        }
      }
    });
    int64_t o = (int64_t)p * C + c;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float r;
      if (mode == 2) r = (float)s[i];
      else if (mode == 1) r = (float)(below ? -s[i] : s[i]);
      else if (reducer == XH_RED_SUM) r = (float)s[i];
      else if (n[i] == 0) r = xh_nan32();
      else if (reducer == XH_RED_MEAN) r = (float)(s[i] / (double)n[i]);
      else r = ext[i];
      out[o + i] = r;
      if (valid_out) valid_out[o + i] = val[i];
    }
  }
}

// ---- climatological_mean_doy (cal:907-931) ---------------------------------------------------------------------
// Same sample set as percentile_doy (all years x centred window, NaN outside the series): nanmean and nanstd (ddof 0).
// Loads are unconditional (clamped row, validity applied afterwards): a load under a condition is followed by
// s_waitcnt vmcnt(0) and the `window` samples of a year would arrive one memory latency after the other.
template <int W>
__global__ void __launch_bounds__(XH_BLOCK)
k_doy_mean_std(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, const int32_t* __restrict__ tbase, int nyears,
               int ndoy, int window, float* __restrict__ mean_out, float* __restrict__ std_out,
               const int32_t* __restrict__ doy_list, int ndl) {
  int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  const int w = W > 0 ? W : window;  // W == 0: run-time window
  const int half = w / 2;
  const int nloop = doy_list ? ndl : ndoy;  // doy_list: only the listed doys (the irregular ones)
  for (int di = blockIdx.y; di < nloop; di += gridDim.y) {
    const int d = doy_list ? doy_list[di] : di;
    double s = 0.0;
    int n = 0;
    auto pass = [&](auto&& f) {
      for (int y = 0; y < nyears; ++y) {
        const int64_t tb = tbase[(int64_t)y * ndoy + d];
        if (W > 0) {
          float v[W > 0 ? W : 1];
#pragma unroll
          for (int k = 0; k < W; ++k) {
            const int64_t t = tb - half + k;
            v[k] = x[(t < 0 ? 0 : (t >= T ? T - 1 : t)) * st + c];
          }
#pragma unroll
          for (int k = 0; k < W; ++k) {
            const int64_t t = tb - half + k;
            if (tb >= 0 && t >= 0 && t < T && v[k] == v[k]) f(v[k]);
          }
        } else {
          for (int k = 0; k < w; ++k) {
            const int64_t t = tb - half + k;
            const float v = x[(t < 0 ? 0 : (t >= T ? T - 1 : t)) * st + c];
            if (tb >= 0 && t >= 0 && t < T && v == v) f(v);
          }
        }
      }
    };
    pass([&](float v) { s += (double)v; n++; });
    const double m = n > 0 ? s / (double)n : 0.0;
    double s2 = 0.0;
    pass([&](float v) { const double dv = (double)v - m; s2 += dv * dv; });
    mean_out[(int64_t)d * C + c] = n > 0 ? (float)m : xh_nan32();
    std_out[(int64_t)d * C + c] = n > 0 ? (float)sqrt(s2 / (double)n) : xh_nan32();
  }
}

// ---- row-range masking (da.where(t in range), fillna(0)) used by the date-bounded run functions (rl:1148-1331) ----
__global__ void __launch_bounds__(XH_BLOCK)
k_mask_rows(const float* __restrict__ x, int64_t C, int64_t st, const int64_t* __restrict__ seg_off, int P,
            const int32_t* __restrict__ lo, const int32_t* __restrict__ hi, int invert, float* __restrict__ out,
            int64_t out_st) {
  int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    for (int64_t t = t0; t < t1; ++t) {
      int i = (int)(t - t0);
      float v = x[t * st + c];
      bool on = v > 0.0f;
      if (invert) on = !on;  // (~da)
      out[t * out_st + c] = (i >= lo[p] && i < hi[p] && on) ? 1.0f : 0.0f;
    }
  }
}

static int chk2(const char* fn, xh_ctx* ctx, const void* x, int64_t T, int64_t C, int64_t st, int64_t sc) {
  XH_REQUIRE(ctx && x, XH_ERR_ARG, "%s: NULL argument", fn);
  XH_REQUIRE(T >= 0 && C >= 0, XH_ERR_ARG, "%s: negative shape", fn);
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT, "%s: needs a time-major view (sc == 1, st >= C)", fn);
  return XH_OK;
}

static int up_seg(xh_ctx* ctx, size_t* cur, const int64_t* seg_off, int P, int64_t T, const char* fn, const int64_t** d_seg) {
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

int xh_bivariate_count(xh_ctx* ctx, const float* x1, const float* x2, int64_t T, int64_t C, int64_t st1, int64_t st2,
                       int op1, double thr1, int op2, double thr2, int combine, const int64_t* seg_off, int P,
                       int32_t* count_out, int32_t* valid_out) {
  int rc = chk2("xh_bivariate_count", ctx, x1, T, C, st1, 1);
  if (rc) return rc;
  XH_REQUIRE(x2 && st2 >= C, XH_ERR_ARG, "xh_bivariate_count: x2 NULL or st2 < C");
  XH_REQUIRE(op1 >= XH_OP_GT && op1 <= XH_OP_NE && op2 >= XH_OP_GT && op2 <= XH_OP_NE, XH_ERR_OP,
             "Operation `%d/%d` not recognized.", op1, op2);
  XH_REQUIRE(combine == 1 || combine == 2, XH_ERR_ARG, "xh_bivariate_count: combine must be 1 (all) or 2 (any)");
  XH_REQUIRE(count_out, XH_ERR_ARG, "xh_bivariate_count: count_out is NULL");
  size_t cur = 0;
  const int64_t* d_seg = nullptr;
  rc = up_seg(ctx, &cur, seg_off, P, T, "xh_bivariate_count", &d_seg);
  if (rc) return rc;
  if (C == 0) return XH_OK;
  int vec = (xh_pick_vec(x1, C, st1) == 4 && xh_pick_vec(x2, C, st2) == 4) ? 4 : 1;
  dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), (unsigned)(P > 4096 ? 4096 : P));
  if (vec == 4)
    hipLaunchKernelGGL((k_bivariate_count<4>), grid, dim3(XH_BLOCK), 0, ctx->stream, x1, x2, C, st1, st2, op1, (float)thr1,
                       op2, (float)thr2, combine, d_seg, P, count_out, valid_out);
  else
    hipLaunchKernelGGL((k_bivariate_count<1>), grid, dim3(XH_BLOCK), 0, ctx->stream, x1, x2, C, st1, st2, op1, (float)thr1,
                       op2, (float)thr2, combine, d_seg, P, count_out, valid_out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_thresholded_reduce(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op, double thr,
                          int mode, int reducer, const int64_t* seg_off, int P, float* out, int32_t* valid_out) {
  int rc = chk2("xh_thresholded_reduce", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(op >= XH_OP_GT && op <= XH_OP_NE, XH_ERR_OP, "Operation `%d` not recognized.", op);
  XH_REQUIRE(mode >= 0 && mode <= 2, XH_ERR_ARG, "xh_thresholded_reduce: mode must be 0, 1 or 2");
  XH_REQUIRE(mode != 0 || (reducer >= XH_RED_SUM && reducer <= XH_RED_MAX), XH_ERR_OP,
             "xh_thresholded_reduce: reducer %d not recognized", reducer);
  XH_REQUIRE(mode == 0 || op <= XH_OP_LE, XH_ERR_OP, "Condition not supported: '%d'.", op);
  XH_REQUIRE(out, XH_ERR_ARG, "xh_thresholded_reduce: out is NULL");
  size_t cur = 0;
  const int64_t* d_seg = nullptr;
  rc = up_seg(ctx, &cur, seg_off, P, T, "xh_thresholded_reduce", &d_seg);
  if (rc) return rc;
  if (C == 0) return XH_OK;
  int vec = xh_pick_vec(x, C, st);
  dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), (unsigned)(P > 4096 ? 4096 : P));
  if (vec == 4)
    hipLaunchKernelGGL((k_thresholded_reduce<4>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, op, (float)thr, mode,
                       reducer, d_seg, P, out, valid_out);
  else
    hipLaunchKernelGGL((k_thresholded_reduce<1>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, op, (float)thr, mode,
                       reducer, d_seg, P, out, valid_out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_mask_rows(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const int64_t* seg_off, int P,
                 const int32_t* lo, const int32_t* hi, int invert, float* out, int64_t out_st) {
  int rc = chk2("xh_mask_rows", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(lo && hi && out && out_st >= C, XH_ERR_ARG, "xh_mask_rows: NULL argument or out_st < C");
  size_t cur = 0;
  const int64_t* d_seg = nullptr;
  rc = up_seg(ctx, &cur, seg_off, P, T, "xh_mask_rows", &d_seg);
  if (rc) return rc;
  void *d_lo = nullptr, *d_hi = nullptr;
  rc = xh_scratch_upload(ctx, &cur, lo, sizeof(int32_t) * (size_t)P, &d_lo);
  if (rc) return rc;
  rc = xh_scratch_upload(ctx, &cur, hi, sizeof(int32_t) * (size_t)P, &d_hi);
  if (rc) return rc;
  if (C == 0) return XH_OK;
  dim3 grid((unsigned)cdiv64(C, XH_BLOCK), (unsigned)(P > 4096 ? 4096 : P));
  hipLaunchKernelGGL(k_mask_rows, grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, d_seg, P, (const int32_t*)d_lo,
                     (const int32_t*)d_hi, invert, out, out_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_doy_mean_std(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const int32_t* tbase,
                    int nyears, int ndoy, int window, float* mean_out, float* std_out) {
  int rc = chk2("xh_doy_mean_std", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(tbase && mean_out && std_out, XH_ERR_ARG, "xh_doy_mean_std: NULL argument");
  XH_REQUIRE(nyears >= 1 && ndoy >= 1 && window >= 1, XH_ERR_ARG, "xh_doy_mean_std: bad shape");
  if (C == 0) return XH_OK;
  size_t cur = 0;
  void* d_tb = nullptr;
  rc = xh_scratch_upload(ctx, &cur, tbase, sizeof(int32_t) * (size_t)nyears * ndoy, &d_tb);
  if (rc) return rc;
  // regular doys: per-day-set partial sums (doystats.hip); irregular doys (calendar gaps) and unusual windows: the generic
  // per-doy kernel below
  const int32_t* d_list = nullptr;
  int ndl = 0;
  bool generic_all = true;
  if ((window == 3 || window == 5 || window == 7) && nyears <= 64) {
    uint8_t* regular = (uint8_t*)malloc((size_t)ndoy);
    int32_t* irregular = (int32_t*)malloc(sizeof(int32_t) * (size_t)ndoy);
    if (!regular || !irregular) {
      free(regular); free(irregular);
      xh_set_error("xh_doy_mean_std: out of host memory");
      return XH_ERR_ARG;
    }
    ndl = pdoy_regular_flags(tbase, nyears, ndoy, window, T, nullptr, T, regular, irregular);
    void *d_reg = nullptr, *d_irr = nullptr;
    rc = xh_scratch_upload(ctx, &cur, regular, (size_t)ndoy, &d_reg);
    if (!rc && ndl) rc = xh_scratch_upload(ctx, &cur, irregular, sizeof(int32_t) * (size_t)ndl, &d_irr);
    free(regular); free(irregular);
    if (rc) return rc;
    // one contiguous year (row of doy index i = tbase[0] + i): the rolling form of doystats.hip
    int64_t year_t0 = -1;
    if (nyears == 1 && tbase[0] >= 0) {
      year_t0 = tbase[0];
      for (int i = 1; i < ndoy; ++i)
        if ((int64_t)tbase[i] != year_t0 + i) { year_t0 = -1; break; }
    }
    rc = xh_launch_doy_stats_sets(ctx, x, T, C, st, (const int32_t*)d_tb, nyears, ndoy, window, (const uint8_t*)d_reg,
                                  mean_out, std_out, year_t0);
    if (rc && rc != XH_ERR_NOTIMPL) return rc;
    if (rc == XH_OK) {
      generic_all = false;
      d_list = (const int32_t*)d_irr;
      if (ndl == 0) return XH_OK;
    }
  }
  const int nloop = generic_all ? ndoy : ndl;
  dim3 grid((unsigned)cdiv64(C, XH_BLOCK), (unsigned)(nloop > 1024 ? 1024 : nloop));
#define XH_DMS(W)                                                                                                        \
  hipLaunchKernelGGL((k_doy_mean_std<W>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, (const int32_t*)d_tb, nyears, ndoy, \
                     window, mean_out, std_out, generic_all ? (const int32_t*)nullptr : d_list, ndl)
  if (window == 5) XH_DMS(5); else if (window == 3) XH_DMS(3); else if (window == 7) XH_DMS(7); else XH_DMS(0);
#undef XH_DMS
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // extern "C"
