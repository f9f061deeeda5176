// wquantile.hip — weighted quantiles over the realization axis: ensemble_percentiles(ens, weights=...)
// (reference: src/xclim/ensembles/_base.py:346-356, which hands the work to xarray's DataArrayWeighted.quantile).
//
// PARITY UNPINNED: the arithmetic lives in xarray (xarray/core/weighted.py, `_weighted_quantile_1d`, not under
// /root/reference and not installable here).  It is restated from its published form (Akinshin 2023, "Weighted quantile
// estimators", the Hyndman-Fan type 7 member; xarray's only method when called like this):
//   drop NaN samples and zero weights; n_eff = (sum w)^2 / sum w^2 (Kish); sort by value; W_i = cumulative normalised
//   weights (W_0 = 0); h = (n_eff - 1) q + 1; u_i = max((h - 1) / n_eff, min(h / n_eff, W_i)); v_i = u_i n_eff - h + 1;
//   result = sum_i x_i (v_{i+1} - v_i).
// With equal weights it reduces to the unweighted type-7 quantile (checked against xh_nan_quantile in the tests).
//
// Layout: (N, C) members on the slow axis, cells contiguous: one lane per cell reads its N samples coalesced, keeps
// (value, member) pairs in a lane-private LDS column (stride 64 words: conflict-free) and insertion-sorts them — N is the
// ensemble size (tens of members), the kernel is a handful of passes over 4 N C bytes.
#include "common.h"

namespace {

constexpr int WQ_MAXN = 128;

template <int MAXN>
__global__ void __launch_bounds__(64)
k_weighted_quantile(const float* __restrict__ x, int N, int64_t C, int64_t sn, const double* __restrict__ w,
                    const double* __restrict__ qs, int nq, double* __restrict__ out) {
  __shared__ float sv[MAXN * 64];
  __shared__ unsigned char si[MAXN * 64];  // member index of each sorted sample (its float64 weight is read from w[])
  const int lane = threadIdx.x;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  if (c >= C) return;
  int n = 0;
  double sum_w = 0.0, sum_w2 = 0.0;
  for (int i = 0; i < N; ++i) {
    const float v = x[(int64_t)i * sn + c];
    const double wi = w[i];
    if (v != v || wi == 0.0) continue;  // skipna + nonzero weights (weighted.py)
    // insertion into the sorted prefix (stable for ties: equal values keep member order)
    int j = n;
    while (j > 0 && sv[(j - 1) * 64 + lane] > v) {
      sv[j * 64 + lane] = sv[(j - 1) * 64 + lane];
      si[j * 64 + lane] = si[(j - 1) * 64 + lane];
      --j;
    }
    sv[j * 64 + lane] = v;
    si[j * 64 + lane] = (unsigned char)i;
    sum_w += wi;
    sum_w2 += wi * wi;
    ++n;
  }
  for (int jq = 0; jq < nq; ++jq) {
    double r = xh_nan64();
    if (n > 0) {
      const double nw = sum_w * sum_w / sum_w2;
      const double h = (nw - 1.0) * qs[jq] + 1.0;
      const double lo = (h - 1.0) / nw, hi = h / nw;
      double cum = 0.0, vprev, acc = 0.0;
      {
        const double u0 = fmax(lo, fmin(hi, 0.0));
        vprev = u0 * nw - h + 1.0;
      }
      for (int i = 0; i < n; ++i) {
        cum += w[si[i * 64 + lane]] / sum_w;
        const double u = fmax(lo, fmin(hi, cum));
        const double v = u * nw - h + 1.0;
        acc += (double)sv[i * 64 + lane] * (v - vprev);
        vprev = v;
      }
      r = acc;
    }
    out[(int64_t)jq * C + c] = r;
  }
}

}  // namespace

extern "C" int xh_weighted_quantile(xh_ctx* ctx, const float* x, int64_t N, int64_t C, int64_t sn, int64_t sc,
                                    const double* weights, const double* q, int nq, double* out) {
  XH_REQUIRE(ctx && x && weights && q && out, XH_ERR_ARG, "xh_weighted_quantile: NULL argument");
  XH_REQUIRE(N >= 1 && C >= 0 && nq >= 1 && nq <= 64, XH_ERR_ARG, "xh_weighted_quantile: bad shape (N >= 1, 1 <= nq <= 64)");
  XH_REQUIRE(N <= WQ_MAXN, XH_ERR_LIMIT, "xh_weighted_quantile: N = %lld members exceed %d", (long long)N, WQ_MAXN);
  XH_REQUIRE(sc == 1 && sn >= C, XH_ERR_LAYOUT, "xh_weighted_quantile: needs a member-major view (sc == 1, sn >= C)");
  for (int64_t i = 0; i < N; ++i)
    XH_REQUIRE(weights[i] >= 0.0 && weights[i] == weights[i], XH_ERR_ARG, "xh_weighted_quantile: weights must be >= 0 and not NaN");
  for (int j = 0; j < nq; ++j) XH_REQUIRE(q[j] >= 0.0 && q[j] <= 1.0, XH_ERR_ARG, "xh_weighted_quantile: quantile outside [0, 1]");
  if (C == 0) return XH_OK;
  size_t cur = 0;
  void *d_w = nullptr, *d_q = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, weights, sizeof(double) * (size_t)N, &d_w);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, q, sizeof(double) * (size_t)nq, &d_q);
  if (rc) return rc;
  hipLaunchKernelGGL((k_weighted_quantile<WQ_MAXN>), dim3((unsigned)cdiv64(C, 64)), dim3(64), 0, ctx->stream, x, (int)N, C, sn,
                     (const double*)d_w, (const double*)d_q, nq, out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}
