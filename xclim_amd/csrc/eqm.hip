// eqm.hip — sdba empirical quantile mapping: full-series quantiles per cell (train) and node search + correction
// (adjust).  The algorithm lives in the third-party package xsdba (>= 0.4.0; pyproject.toml:111 of the reference;
// only src/xclim/sdba.py:10 and tests/test_xsdba.py reference it).  Spec adopted (parity "unpinned"):
//   nbutils.quantile        : NaN-aware Hyndman-Fan type 7 (alpha = beta = 1, same formula as core/utils.py:395)
//   utils.get_correction    : af = ref_q - hist_q ("+")  |  ref_q / hist_q ("*")
//   utils.interp_on_quantiles (1-D, group="time"): scipy.interpolate.interp1d(hist_q, af, kind=nearest|linear,
//                             bounds_error=False, fill_value=(af[0], af[-1]) | nan) on the non-NaN nodes
//   utils.apply_correction  : scen = sim + af_t  |  sim * af_t
#include "common.h"

// Per-column quantiles: exact multi-select kernels in select.hip (xh_select_columns).

// af from ref_q / hist_q  (get_correction), one thread per COLUMN, with the nanmax rule of utl:552-554 folded in (see
// k_nanmax_fix in select.hip): a NaN node of a series that has valid samples is that series' largest valid sample.  The thread
// reads the column's 2 * nq nodes — the bytes an elementwise correction reads anyway — so that EQM training pays no separate
// pass for the rule; the series themselves are scanned only for columns that hold such a node (infinities in the data).
// (k_correction_fix: select.hip, next to k_nanmax_fix — its column scans are a workgroup's work since round 6)

// ---- adjust -------------------------------------------------------------------------------------------------
// One lane per cell, marching along time (time-major).  The nq nodes of the cell stay in registers; the node
// search is a fully unrolled scan (static register indices).  NaN nodes are skipped (mask_old).
template <int NQMAX, int INTERP>
__global__ void __launch_bounds__(XH_BLOCK)
k_eqm_adjust(const float* __restrict__ sim, int64_t T, int64_t C, int64_t st, const float* __restrict__ af,
             const float* __restrict__ hq, int nq, int kind, int extrap, float* __restrict__ scen, int64_t scen_st) {
  int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  float nx[NQMAX], ny[NQMAX], mid[NQMAX];
  const float inf = __uint_as_float(0x7F800000u);
  float firstx = 0.f, firsty = xh_nan32(), lastx = 0.f, lasty = xh_nan32();
  int m = 0;
  // every node of the cell is requested before the first one is used: unconditional loads from a clamped row into register
  // arrays (a load inside `if (j < nq)`, consumed in the same iteration, was followed by s_waitcnt vmcnt(0) — the 2 x nq node
  // loads of a cell were nq dependent round trips at the start of every block: DESIGN.md §7, the compiler rule)
  float xl[NQMAX], yl[NQMAX];
#pragma unroll
  for (int j = 0; j < NQMAX; ++j) {
    const int jj = j < nq ? j : nq - 1;
    xl[j] = hq[(int64_t)jj * C + c];
    yl[j] = af[(int64_t)jj * C + c];
  }
#pragma unroll
  for (int j = 0; j < NQMAX; ++j) {
    const float xj = j < nq ? xl[j] : xh_nan32(), yj = j < nq ? yl[j] : xh_nan32();
    bool valid = (xj == xj) && (yj == yj);
    nx[j] = valid ? xj : xh_nan32();
    ny[j] = yj;
    if (INTERP == 0) {
      // scipy nearest: x_bds = x/2; x_bds = x_bds[1:] + x_bds[:-1]; idx = searchsorted(x_bds, x_new, side="left")
      mid[j] = valid ? (m == 0 ? -inf : (xj * 0.5f + lastx * 0.5f)) : inf;
    }
    if (valid) {
      if (m == 0) { firstx = xj; firsty = yj; }
      lastx = xj; lasty = yj;
      m++;
    }
  }
  // linear: the valid nodes compacted to the front (padded with x = +inf) and the slope of every interval, once per cell:
  // the per-element work is then one compare and three selects per node instead of a scan that re-derives the valid
  // pairs, and the division leaves the time loop.  Same fp32 operations as scipy's slope * (x - x_lo) + y_lo.
  float cx[INTERP == 1 ? NQMAX : 1], cy[INTERP == 1 ? NQMAX : 1], sl[INTERP == 1 ? NQMAX : 1];
  if (INTERP == 1) {
    if (__all(m == nq)) {  // no NaN node in the wave (the usual case): nothing to compact
#pragma unroll
      for (int j = 0; j < NQMAX; ++j) { cx[j] = j < nq ? nx[j] : inf; cy[j] = ny[j]; }
    } else {
      int rank[NQMAX], r = 0;
#pragma unroll
      for (int j = 0; j < NQMAX; ++j) { rank[j] = r; r += (nx[j] == nx[j]) ? 1 : 0; }
#pragma unroll
      for (int i = 0; i < NQMAX; ++i) {
        cx[i] = inf; cy[i] = 0.f;
#pragma unroll
        for (int j = i; j < NQMAX; ++j) {
          const bool pick = (nx[j] == nx[j]) && rank[j] == i;
          cx[i] = pick ? nx[j] : cx[i];
          cy[i] = pick ? ny[j] : cy[i];
        }
      }
    }
#pragma unroll
    for (int j = 0; j + 1 < NQMAX; ++j) sl[j] = (cy[j + 1] - cy[j]) / (cx[j + 1] - cx[j]);
    sl[NQMAX - 1] = 0.f;
  }
  int64_t chunk = cdiv64(T, (int64_t)gridDim.y);
  int64_t ta = (int64_t)blockIdx.y * chunk, tb = ta + chunk;
  if (tb > T) tb = T;
  // batches of 8 rows, the loads are issued before any use (8 independent rows in flight per lane instead of 2)
  auto adjust_one = [&](int64_t t, float x) {
    float a = xh_nan32();
    if (m >= 2 && x == x) {  // fewer than two valid nodes: NaN (scipy's interp1d refuses them; the oracle returns NaN)
      bool below = x < firstx, above = x > lastx;
      if (INTERP == 0) {
        a = firsty;
#pragma unroll
        for (int j = 0; j < NQMAX; ++j) a = (mid[j] < x) ? ny[j] : a;
      } else {
        // lo = the last node strictly below x (searchsorted side="left", clipped to the first interval); a padded node
        // (+inf) is never below x, and the last valid node only when x is above the range (overridden below)
        float lx = cx[0], ly = cy[0], ls = sl[0];
#pragma unroll
        for (int j = 1; j + 1 < NQMAX; ++j) {
          const bool take = cx[j] < x;
          lx = take ? cx[j] : lx; ly = take ? cy[j] : ly; ls = take ? sl[j] : ls;
        }
        a = ls * (x - lx) + ly;
      }
      if (below) a = extrap == 0 ? firsty : xh_nan32();
      if (above) a = extrap == 0 ? lasty : xh_nan32();
    }
    scen[t * scen_st + c] = kind == 0 ? (x + a) : (kind == 1 ? (x * a) : a);  // kind 2: the interpolated factor itself
  };
  int64_t t = ta;
  for (; t + 8 <= tb; t += 8) {
    float xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = sim[(t + u) * st + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) adjust_one(t + u, xv[u]);
  }
  for (; t < tb; ++t) adjust_one(t, sim[t * st + c]);
}

// ---- adjust with a sub-grouping, interp = "nearest" the way xsdba does it ----------------------------------------------
// xsdba.utils.interp_on_quantiles with a month / day-of-year Grouper does NOT interpolate inside the step's own group: it
// calls _interp_on_quantiles_2D -> scipy.interpolate.griddata((hist_q, group coordinate), af, (sim, group of the step),
// method="nearest") on the nodes of ALL groups (after add_cyclic_bounds: coordinates 0 .. G + 1, 0 = a copy of the last
// group, G + 1 = a copy of the first), then _extrapolate_on_quantiles puts the step's OWN group's first / last factor
// (or NaN) where the value lies outside that group's nodes.  The nearest point in the (value, group) plane is a node of
// the own group unless that node is more than one unit away — then a neighbouring group's node at distance
// sqrt(dx^2 + dg^2) can win (precipitation in mm/day, temperature tails).  Round 3 always took the own group's node.
// One lane per cell over the rows of ONE group's block: the own nodes in registers, the neighbours' nodes read on demand
// (group distance k while k^2 < the best squared distance so far); distances in float64 like the cKDTree's.
template <int NQMAX>
__global__ void __launch_bounds__(XH_BLOCK)
k_eqm_adjust_g2d(const float* __restrict__ sim, int64_t n, int64_t C, int64_t st, const float* __restrict__ af_all,
                 const float* __restrict__ hq_all, int G, int nq, int gcoord, int kind, int extrap, float* __restrict__ scen,
                 int64_t scen_st) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  const int64_t plane = (int64_t)nq * C;
  const float* __restrict__ hq = hq_all + (int64_t)(gcoord - 1) * plane;
  const float* __restrict__ af = af_all + (int64_t)(gcoord - 1) * plane;
  float nx[NQMAX], ny[NQMAX];
  float firstx = 0.f, firsty = xh_nan32(), lastx = 0.f, lasty = xh_nan32();
  int m = 0;
  // every node of the cell is requested before the first one is used: unconditional loads from a clamped row into register
  // arrays (a load inside `if (j < nq)`, consumed in the same iteration, was followed by s_waitcnt vmcnt(0) — the 2 x nq node
  // loads of a cell were nq dependent round trips at the start of every block: DESIGN.md §7, the compiler rule)
  float xl[NQMAX], yl[NQMAX];
#pragma unroll
  for (int j = 0; j < NQMAX; ++j) {
    const int jj = j < nq ? j : nq - 1;
    xl[j] = hq[(int64_t)jj * C + c];
    yl[j] = af[(int64_t)jj * C + c];
  }
#pragma unroll
  for (int j = 0; j < NQMAX; ++j) {
    const float xj = j < nq ? xl[j] : xh_nan32(), yj = j < nq ? yl[j] : xh_nan32();
    const bool valid = (xj == xj) && (yj == yj);
    nx[j] = valid ? xj : xh_nan32();
    ny[j] = yj;
    // utils._first_and_last_nonnull works on hist_q and on af SEPARATELY: the bounds are the first / last non-null node
    // VALUE, the constants the first / last non-null FACTOR (a 0 / 0 factor at a dry node does not move the bound)
    if (xj == xj) {
      if (m == 0) firstx = xj;
      lastx = xj;
      m++;
    }
    if (yj == yj) {
      if (firsty != firsty) firsty = yj;
      lasty = yj;
    }
  }
  const int64_t chunk = cdiv64(n, (int64_t)gridDim.y);
  const int64_t ta = (int64_t)blockIdx.y * chunk;
  int64_t tb = ta + chunk;
  if (tb > n) tb = n;
  for (int64_t t = ta; t < tb; ++t) {
    const float x = sim[t * st + c];
    float a = xh_nan32();
    if (x == x) {
      const double xd = (double)x;
      double best = __longlong_as_double(0x7FF0000000000000LL);  // +inf
#pragma unroll
      for (int j = 0; j < NQMAX; ++j) {
        const double dx = xd - (double)nx[j];
        const double d2 = dx * dx;  // (NaN node: the compare is false)
        if (d2 < best) { best = d2; a = ny[j]; }
      }
      for (int k = 1; (double)k * (double)k < best && k <= G + 1; ++k) {
#pragma unroll 1
        for (int sgn = -1; sgn <= 1; sgn += 2) {
          const int gg = gcoord + sgn * k;  // coordinate 0 .. G + 1 (cyclic copies at the ends)
          if (gg < 0 || gg > G + 1) continue;
          const int ti = gg == 0 ? G - 1 : (gg == G + 1 ? 0 : gg - 1);
          const float* __restrict__ hx = hq_all + (int64_t)ti * plane + c;
          const float* __restrict__ hy = af_all + (int64_t)ti * plane + c;
          for (int j = 0; j < nq; ++j) {
            const float xj = hx[(int64_t)j * C], yj = hy[(int64_t)j * C];
            if (xj == xj && yj == yj) {
              const double dx = xd - (double)xj;
              const double d2 = dx * dx + (double)k * (double)k;
              if (d2 < best) { best = d2; a = yj; }
            }
          }
        }
      }
      // _extrapolate_on_quantiles: outside the OWN group's nodes -> its first / last factor, or NaN
      if (m > 0) {
        if (x < firstx) a = extrap == 0 ? firsty : xh_nan32();
        if (x > lastx) a = extrap == 0 ? lasty : xh_nan32();
      }
    }
    scen[t * scen_st + c] = kind == 0 ? (x + a) : (kind == 1 ? (x * a) : a);
  }
}

// ---- adjust, interp = "cubic" ----------------------------------------------------------------------------------
// scipy.interpolate.interp1d(kind="cubic") = interpolating cubic spline with not-a-knot end conditions
// (make_interp_spline(k=3)).  Restated in its classical form: second derivatives M_i from the tridiagonal system
//   h_{i-1} M_{i-1} + 2 (h_{i-1} + h_i) M_i + h_i M_{i+1} = 6 (d_i - d_{i-1}),   d_i = (y_{i+1} - y_i) / h_i,
// with M_0 and M_{m-1} eliminated through the not-a-knot conditions (continuous third derivative at x_1 and x_{m-2}).
// Everything in fp64 like scipy (which converts the float32 nodes first); the value is rounded to fp32 once.
// Set-up kernel: one lane per cell drops the NaN nodes, solves the system (Thomas) in a (rows, C) fp64 workspace
// (coalesced across lanes) and leaves x, y, M per node; fewer than 4 valid nodes -> count 0 (result NaN).
__global__ void __launch_bounds__(XH_BLOCK)
k_cubic_setup(const float* __restrict__ af, const float* __restrict__ hq, int nq, int64_t C, double* __restrict__ wx,
              double* __restrict__ wy, double* __restrict__ wM, double* __restrict__ wc, double* __restrict__ wd,
              int32_t* __restrict__ wm) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  int m = 0;
  for (int j = 0; j < nq; ++j) {
    const float xj = hq[(int64_t)j * C + c], yj = af[(int64_t)j * C + c];
    if (xj == xj && yj == yj) {
      wx[(int64_t)m * C + c] = (double)xj;
      wy[(int64_t)m * C + c] = (double)yj;
      m++;
    }
  }
  for (int j = m; j < nq; ++j) {
    wx[(int64_t)j * C + c] = __longlong_as_double(0x7FF0000000000000LL);  // +inf: never selected
    wy[(int64_t)j * C + c] = 0.0;
    wM[(int64_t)j * C + c] = 0.0;
  }
  for (int j = 0; j < nq; ++j) { wc[(int64_t)j * C + c] = 0.0; wd[(int64_t)j * C + c] = 0.0; }
  if (m < 4) { wm[c] = 0; return; }
  wm[c] = m;
  {
    // a non-finite factor among the nodes (x / 0 of a multiplicative mapping): scipy's banded solve spreads inf / NaN over the
    // spline coefficients in a pattern that depends on LAPACK's elimination order and on the dtype it is handed (float64: NaN
    // everywhere inside the node range).  Rule here: NaN everywhere inside the node range, the end factors outside it
    // (interp1d's fill_value) — deterministic, and what scipy answers for float64 factors
    bool finite = true;
    for (int j = 0; j < m; ++j) {
      const double yj = wy[(int64_t)j * C + c];
      finite = finite && (yj - yj == 0.0);
    }
    if (!finite) {
      for (int j = 0; j < m; ++j) {
        wM[(int64_t)j * C + c] = xh_nan64();
        wc[(int64_t)j * C + c] = xh_nan64();
        wd[(int64_t)j * C + c] = xh_nan64();
      }
      return;
    }
  }
  auto X = [&](int i) { return wx[(int64_t)i * C + c]; };
  auto Y = [&](int i) { return wy[(int64_t)i * C + c]; };
  auto H = [&](int i) { return X(i + 1) - X(i); };
  auto D = [&](int i) { return (Y(i + 1) - Y(i)) / H(i); };
  // forward sweep over the unknowns M_1 .. M_{m-2}
  double cp_prev = 0.0, rp_prev = 0.0;
  for (int i = 1; i <= m - 2; ++i) {
    const double hm = H(i - 1), hi = H(i);
    double a = hm, b = 2.0 * (hm + hi), cc = hi;
    const double r = 6.0 * (D(i) - D(i - 1));
    if (i == 1) {  // M_0 = ((h_0 + h_1) M_1 - h_0 M_2) / h_1
      b += hm * (hm + hi) / hi;
      cc -= hm * hm / hi;
      a = 0.0;
    }
    if (i == m - 2) {  // M_{m-1} = ((h_{m-2} + h_{m-3}) M_{m-2} - h_{m-2} M_{m-3}) / h_{m-3}
      a -= hi * hi / hm;
      b += hi * (hi + hm) / hm;
      cc = 0.0;
    }
    const double den = b - a * cp_prev;
    cp_prev = cc / den;
    rp_prev = (r - a * rp_prev) / den;
    wc[(int64_t)i * C + c] = cp_prev;
    wM[(int64_t)i * C + c] = rp_prev;
  }
  // back substitution
  double Mn = wM[(int64_t)(m - 2) * C + c];
  for (int i = m - 3; i >= 1; --i) {
    Mn = wM[(int64_t)i * C + c] - wc[(int64_t)i * C + c] * Mn;
    wM[(int64_t)i * C + c] = Mn;
  }
  {
    const double h0 = H(0), h1 = H(1), M1 = wM[(int64_t)1 * C + c], M2 = wM[(int64_t)2 * C + c];
    wM[c] = ((h0 + h1) * M1 - h0 * M2) / h1;
    const double ha = H(m - 2), hb = H(m - 3), Ma = wM[(int64_t)(m - 2) * C + c], Mb = wM[(int64_t)(m - 3) * C + c];
    wM[(int64_t)(m - 1) * C + c] = ((ha + hb) * Ma - ha * Mb) / hb;
  }
  // per-interval polynomial coefficients (the divisions leave the per-element loop; same operations, same results):
  //   S(x) = y_i + t c1_i + t^2 (M_i / 2) + t^3 c3_i,  t = x - x_i
  for (int i = 0; i <= m - 2; ++i) {
    const double h = H(i), d = D(i), Mi = wM[(int64_t)i * C + c], M1 = wM[(int64_t)(i + 1) * C + c];
    wc[(int64_t)i * C + c] = d - h * (2.0 * Mi + M1) / 6.0;
    wd[(int64_t)i * C + c] = (M1 - Mi) / (6.0 * h);
  }
  wc[(int64_t)(m - 1) * C + c] = 0.0;  // (row m - 2 of the temporary c' is overwritten above, m - 1 was never used)
}

template <int NQMAX>
__global__ void __launch_bounds__(XH_BLOCK)
k_eqm_adjust_cubic(const float* __restrict__ sim, int64_t T, int64_t C, int64_t st, const double* __restrict__ wx,
                   const double* __restrict__ wy, const double* __restrict__ wM, const double* __restrict__ wc,
                   const double* __restrict__ wd, const int32_t* __restrict__ wm, int nq, int kind, int extrap,
                   float* __restrict__ scen, int64_t scen_st) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  // nodes as float (they ARE float32 values: the interval search compares in fp32, exactly), values and coefficients in fp64
  float xf[NQMAX];
  double yn[NQMAX], c1[NQMAX], Mn[NQMAX], c3[NQMAX];
  const int m = wm[c];
#pragma unroll
  for (int j = 0; j < NQMAX; ++j) {
    const bool in = j < nq;
    xf[j] = in ? (float)wx[(int64_t)j * C + c] : __uint_as_float(0x7F800000u);
    yn[j] = in ? wy[(int64_t)j * C + c] : 0.0;
    c1[j] = in ? wc[(int64_t)j * C + c] : 0.0;
    Mn[j] = in ? wM[(int64_t)j * C + c] : 0.0;
    c3[j] = in ? wd[(int64_t)j * C + c] : 0.0;
  }
  float xlast = xf[0];
  double ylast = yn[0];
#pragma unroll
  for (int j = 1; j < NQMAX; ++j) {
    const bool v = j < m;
    xlast = v ? xf[j] : xlast;
    ylast = v ? yn[j] : ylast;
  }
  const int64_t chunk = cdiv64(T, (int64_t)gridDim.y);
  int64_t ta = (int64_t)blockIdx.y * chunk, tb = ta + chunk;
  if (tb > T) tb = T;
  auto adjust_one = [&](int64_t t, float xs) {
    float a = xh_nan32();
    if (m >= 4 && xs == xs) {
      // interval index by counting (the nodes ascend, padded nodes are +inf), then ONE dynamic fetch per coefficient
      // from the lane's private arrays (scratch, L1-resident) instead of a 5 x 18 select chain per element
      int idx = 0;
#pragma unroll
      for (int j = 1; j < NQMAX - 1; ++j) idx += ((xs >= xf[j]) && (j <= m - 2)) ? 1 : 0;
      const float xi = xf[idx];
      const double yi = yn[idx], ci = c1[idx], Mi = Mn[idx], di = c3[idx];
      const double tt = (double)xs - (double)xi;
      double S = yi + tt * ci + tt * tt * (Mi * 0.5) + tt * tt * tt * di;
      if (xs < xf[0]) S = extrap == 0 ? yn[0] : xh_nan64();
      if (xs > xlast) S = extrap == 0 ? ylast : xh_nan64();
      a = (float)S;
    }
    scen[t * scen_st + c] = kind == 0 ? (xs + a) : (kind == 1 ? (xs * a) : a);
  };
  int64_t t = ta;
  for (; t + 8 <= tb; t += 8) {  // 8 rows in flight per lane
    float xv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) xv[u] = sim[(t + u) * st + c];
#pragma unroll
    for (int u = 0; u < 8; ++u) adjust_one(t + u, xv[u]);
  }
  for (; t < tb; ++t) adjust_one(t, sim[t * st + c]);
}

static int quantile_series_core(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc,
                                const double* d_q, int nq, float* out);

static int quantile_series_impl(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc,
                                const double* d_q, int nq, float* out) {
  const int rc = quantile_series_core(ctx, x, T, C, st, sc, d_q, nq, out);
  if (rc) return rc;
  return xh_nanmax_fix(ctx, x, T, C, st, sc, nq, out);   // (select.hip)
}

static int quantile_series_core(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc,
                                const double* d_q, int nq, float* out) {
  if (st == 1 && sc >= T) {
    return xh_select_columns(ctx, x, T, C, sc, d_q, nq, out, 1, C);
  }
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT, "quantile_series: one of the strides must be 1 (st=%lld sc=%lld)",
             (long long)st, (long long)sc);
  // short series: 8-lane groups read the time-major rows directly (no transpose)
  int rc0 = xh_select_time_major(ctx, x, T, C, st, d_q, nq, out, 1, C);
  if (rc0 != XH_ERR_NOTIMPL) return rc0;
  // long series: two streaming passes over the rows as they lie in memory (select4.hip)
  rc0 = xh_select_hist(ctx, x, T, C, st, d_q, nq, out, 1, C);
  if (rc0 != XH_ERR_NOTIMPL) return rc0;
  // time-major: transpose batches of columns into scratch (not counted as algorithmic bytes).  Two scratch
  // buffers and two streams: batch k+1 is transposed on stream2 while batch k is selected on the main stream (the
  // selection kernels are latency bound and leave the memory pipe mostly idle).
  // padded column stride: every 64-float segment the transpose writes is one aligned 256-byte block (a misaligned segment
  // touches three 128-byte lines instead of two), and an ODD number of 256-byte blocks per column: the selection
  // workgroups stream hundreds of columns in lockstep, and with an even count (11008 floats = 172 blocks at T = 10950)
  // the same offset of consecutive columns falls onto a quarter of the HBM channels.
  int64_t Tp = (T + 63) & ~(int64_t)63;
  if (((Tp / 64) & 1) == 0) Tp += 64;
  int64_t batch = (int64_t)((1ull << 29) / (sizeof(float) * (size_t)Tp));  // 512 MB batches (smaller ones only add tails)
  batch = (batch / 64) * 64;
  if (batch < 64) batch = 64;
  if (batch > C) batch = C;
  const size_t buf_elems = (size_t)batch * (size_t)Tp;
  void* tmp = nullptr;
  int rc = xh_big_scratch(ctx, 2 * sizeof(float) * buf_elems, &tmp);
  if (rc) return rc;
  float* bufs[2] = {(float*)tmp, (float*)tmp + buf_elems};
  hipStream_t s_main = ctx->stream, s_tr = ctx->stream2;
  // stream2 starts after everything already queued on the main stream (the producer of x)
  XH_CHECK_HIP(hipEventRecord(ctx->ev_done[0], s_main));
  XH_CHECK_HIP(hipStreamWaitEvent(s_tr, ctx->ev_done[0], 0));
  const int64_t nbatch = cdiv64(C, batch);
  auto transpose_batch = [&](int64_t k) -> int {
    const int64_t c0 = k * batch, nb = C - c0 < batch ? C - c0 : batch;
    if (k >= 2) XH_CHECK_HIP(hipStreamWaitEvent(s_tr, ctx->ev_done[k & 1], 0));  // buffer free again
    ctx->stream = s_tr;
    int r = xh_transpose_f32(ctx, x + c0, T, nb, st, bufs[k & 1], Tp);
    ctx->stream = s_main;
    if (r) return r;
    XH_CHECK_HIP(hipEventRecord(ctx->ev_ready[k & 1], s_tr));
    return XH_OK;
  };
  rc = transpose_batch(0);
  if (rc) return rc;
  for (int64_t k = 0; k < nbatch; ++k) {
    const int64_t c0 = k * batch, nb = C - c0 < batch ? C - c0 : batch;
    if (k + 1 < nbatch) {
      rc = transpose_batch(k + 1);
      if (rc) return rc;
    }
    XH_CHECK_HIP(hipStreamWaitEvent(s_main, ctx->ev_ready[k & 1], 0));
    rc = xh_select_columns(ctx, bufs[k & 1], T, nb, Tp, d_q, nq, out + c0, 1, C);
    if (rc) return rc;
    XH_CHECK_HIP(hipEventRecord(ctx->ev_done[k & 1], s_main));
  }
  return XH_OK;
}

// utils.apply_correction on two fields of one shape: out = base + fac | base * fac (the factor came out of an interpolation
// whose abscissa is not the field itself: QDM "cubic", where it is the percentage rank)
__global__ void __launch_bounds__(XH_BLOCK)
k_apply_factor(const float* __restrict__ base, const float* __restrict__ fac, int64_t T, int64_t C, int64_t st, int64_t fst, int kind,
               float* __restrict__ out, int64_t ost) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  for (int64_t t = blockIdx.y; t < T; t += gridDim.y) {
    const float b = base[t * st + c], a = fac[t * fst + c];
    out[t * ost + c] = kind == 0 ? b + a : b * a;
  }
}

extern "C" {

int xh_quantile_series(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const double* q, int nq,
                       float* out) {
  XH_REQUIRE(ctx && x && q && out, XH_ERR_ARG, "xh_quantile_series: NULL argument");
  XH_REQUIRE(T >= 1 && C >= 0 && nq >= 1 && nq <= 64, XH_ERR_ARG, "xh_quantile_series: bad shape (T >= 1, 1 <= nq <= 64)");
  if (C == 0) return XH_OK;
  size_t cur = 0;
  void* d_q = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, q, sizeof(double) * nq, &d_q);
  if (rc) return rc;
  return quantile_series_impl(ctx, x, T, C, st, sc, (const double*)d_q, nq, out);
}

int xh_eqm_train(xh_ctx* ctx, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st, int64_t sc,
                 const double* q, int nq, int kind, float* af, float* hist_q) {
  XH_REQUIRE(ctx && ref && hist && q && af && hist_q, XH_ERR_ARG, "xh_eqm_train: NULL argument");
  XH_REQUIRE(T >= 1 && C >= 0 && nq >= 1 && nq <= 64, XH_ERR_ARG, "xh_eqm_train: bad shape (T >= 1, 1 <= nq <= 64)");
  XH_REQUIRE(kind == 0 || kind == 1, XH_ERR_ARG, "xh_eqm_train: kind must be 0 (+) or 1 (*)");
  if (C == 0) return XH_OK;
  size_t cur = 0;
  void* d_q = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, q, sizeof(double) * nq, &d_q);
  if (rc) return rc;
  // ref_q goes to `af` first, then af = correction(ref_q, hist_q) in place
  rc = quantile_series_core(ctx, ref, T, C, st, sc, (const double*)d_q, nq, af);
  if (rc) return rc;
  rc = quantile_series_core(ctx, hist, T, C, st, sc, (const double*)d_q, nq, hist_q);
  if (rc) return rc;
  return xh_correction_fix(ctx, ref, hist, T, C, st, sc, nq, kind, af, hist_q);
}

// Grouped adjustment with xsdba's 2-D "nearest" (see k_eqm_adjust_g2d): sim (n, C) = the steps of ONE group (coordinate
// gcoord in 1 .. G, the g-th label of a contiguous month / day-of-year grouping), af_all / hq_all (G, nq, C).
int xh_eqm_adjust_g2d(xh_ctx* ctx, const float* sim, int64_t n, int64_t C, int64_t st, const float* af_all, const float* hq_all,
                      int G, int nq, int gcoord, int kind, int extrap, float* scen, int64_t scen_st) {
  XH_REQUIRE(ctx && sim && af_all && hq_all && scen, XH_ERR_ARG, "xh_eqm_adjust_g2d: NULL argument");
  XH_REQUIRE(n >= 0 && C >= 0 && nq >= 1 && nq <= 32 && G >= 1 && gcoord >= 1 && gcoord <= G, XH_ERR_ARG,
             "xh_eqm_adjust_g2d: bad shape (1 <= nq <= 32, 1 <= gcoord <= G)");
  XH_REQUIRE(st >= C && scen_st >= C, XH_ERR_LAYOUT, "xh_eqm_adjust_g2d: needs time-major views");
  XH_REQUIRE(kind >= 0 && kind <= 2, XH_ERR_ARG, "xh_eqm_adjust_g2d: kind must be 0 (+), 1 (*) or 2 (the factor only)");
  XH_REQUIRE(extrap == 0 || extrap == 1, XH_ERR_ARG, "xh_eqm_adjust_g2d: extrap must be 0 (constant) or 1 (nan)");
  if (n == 0 || C == 0) return XH_OK;
  const int64_t cblocks = cdiv64(C, XH_BLOCK);
  int64_t gy = cdiv64((int64_t)ctx->num_cu * 16, cblocks);
  if (gy < 1) gy = 1;
  if (gy > n) gy = n;
  if (gy > 1024) gy = 1024;
  const dim3 grid((unsigned)cblocks, (unsigned)gy);
  if (nq <= 20)
    hipLaunchKernelGGL((k_eqm_adjust_g2d<20>), grid, dim3(XH_BLOCK), 0, ctx->stream, sim, n, C, st, af_all, hq_all, G, nq, gcoord, kind,
                       extrap, scen, scen_st);
  else
    hipLaunchKernelGGL((k_eqm_adjust_g2d<32>), grid, dim3(XH_BLOCK), 0, ctx->stream, sim, n, C, st, af_all, hq_all, G, nq, gcoord, kind,
                       extrap, scen, scen_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_eqm_adjust(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, int64_t sc, const float* af,
                  const float* hist_q, int nq, int kind, int interp, int extrap, float* scen, int64_t scen_st) {
  XH_REQUIRE(ctx && sim && af && hist_q && scen, XH_ERR_ARG, "xh_eqm_adjust: NULL argument");
  XH_REQUIRE(T >= 0 && C >= 0 && nq >= 1 && nq <= 64, XH_ERR_ARG, "xh_eqm_adjust: bad shape (1 <= nq <= 64)");
  XH_REQUIRE(sc == 1 && st >= C && scen_st >= C, XH_ERR_LAYOUT, "xh_eqm_adjust: needs time-major views (sc == 1)");
  XH_REQUIRE(kind >= 0 && kind <= 2, XH_ERR_ARG, "xh_eqm_adjust: kind must be 0 (+), 1 (*) or 2 (the interpolated factor only)");
  XH_REQUIRE(interp >= 0 && interp <= 2, XH_ERR_ARG, "xh_eqm_adjust: interp must be 0 (nearest), 1 (linear) or 2 (cubic)");
  XH_REQUIRE(extrap == 0 || extrap == 1, XH_ERR_ARG, "xh_eqm_adjust: extrap must be 0 (constant) or 1 (nan)");
  if (T == 0 || C == 0) return XH_OK;
  int64_t cblocks = cdiv64(C, XH_BLOCK);
  int64_t want = (int64_t)ctx->num_cu * 16;
  int64_t gy = cdiv64(want, cblocks);
  if (gy < 1) gy = 1;
  if (gy > T) gy = T;
  if (gy > 1024) gy = 1024;
  dim3 grid((unsigned)cblocks, (unsigned)gy);
  if (interp == 2) {
    XH_REQUIRE(nq <= 32, XH_ERR_LIMIT, "xh_eqm_adjust: cubic interpolation supports at most 32 quantile nodes (got %d)", nq);
    // workspace: x, y, M, c1 (c' during the solve), c3 as (nq, C) float64 + the valid-node count per cell
    const size_t plane = sizeof(double) * (size_t)nq * (size_t)C;
    void* ws = nullptr;
    int rc = xh_big_scratch(ctx, 5 * plane + sizeof(int32_t) * (size_t)C, &ws);
    if (rc) return rc;
    double *wx = (double*)ws, *wy = wx + (size_t)nq * C, *wM = wy + (size_t)nq * C, *wc = wM + (size_t)nq * C;
    double* wd = wc + (size_t)nq * C;
    int32_t* wm = (int32_t*)(wd + (size_t)nq * C);
    hipLaunchKernelGGL(k_cubic_setup, dim3((unsigned)cblocks), dim3(XH_BLOCK), 0, ctx->stream, af, hist_q, nq, C, wx, wy, wM, wc, wd,
                       wm);
    if (nq <= 20)
      hipLaunchKernelGGL((k_eqm_adjust_cubic<20>), grid, dim3(XH_BLOCK), 0, ctx->stream, sim, T, C, st, wx, wy, wM, wc, wd, wm, nq, kind,
                         extrap, scen, scen_st);
    else
      hipLaunchKernelGGL((k_eqm_adjust_cubic<32>), grid, dim3(XH_BLOCK), 0, ctx->stream, sim, T, C, st, wx, wy, wM, wc, wd, wm, nq, kind,
                         extrap, scen, scen_st);
    XH_LAUNCH_CHECK();
    return XH_OK;
  }
#define XH_ADJ(NQM, IP)                                                                                              \
  hipLaunchKernelGGL((k_eqm_adjust<NQM, IP>), grid, dim3(XH_BLOCK), 0, ctx->stream, sim, T, C, st, af, hist_q, nq, kind, \
                     extrap, scen, scen_st)
  if (nq <= 10) {
    if (interp == 0) XH_ADJ(10, 0); else XH_ADJ(10, 1);
  } else if (nq <= 20) {
    if (interp == 0) XH_ADJ(20, 0); else XH_ADJ(20, 1);
  } else if (nq <= 32) {
    if (interp == 0) XH_ADJ(32, 0); else XH_ADJ(32, 1);
  } else {
    if (interp == 0) XH_ADJ(64, 0); else XH_ADJ(64, 1);
  }
#undef XH_ADJ
  XH_LAUNCH_CHECK();
  return XH_OK;
}

// apply_correction(base, fac, kind) for two (T, C) fields (row strides st, fst, out_st)
int xh_apply_factor(xh_ctx* ctx, const float* base, const float* fac, int64_t T, int64_t C, int64_t st, int64_t fst, int kind,
                    float* out, int64_t out_st) {
  XH_REQUIRE(ctx && base && fac && out, XH_ERR_ARG, "xh_apply_factor: NULL argument");
  XH_REQUIRE(T >= 0 && C >= 0 && st >= C && fst >= C && out_st >= C, XH_ERR_LAYOUT, "xh_apply_factor: needs time-major views");
  XH_REQUIRE(kind == 0 || kind == 1, XH_ERR_ARG, "xh_apply_factor: kind must be 0 (+) or 1 (*)");
  if (T == 0 || C == 0) return XH_OK;
  const int64_t cblocks = cdiv64(C, XH_BLOCK);
  int64_t gy = cdiv64((int64_t)ctx->num_cu * 16, cblocks);
  gy = gy < 1 ? 1 : (gy > T ? T : (gy > 4096 ? 4096 : gy));
  hipLaunchKernelGGL(k_apply_factor, dim3((unsigned)cblocks, (unsigned)gy), dim3(XH_BLOCK), 0, ctx->stream, base, fac, T, C, st, fst, kind,
                     out, out_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // extern "C"
