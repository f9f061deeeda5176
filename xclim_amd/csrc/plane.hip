// plane.hip — interp="linear" over the (quantile, group) PLANE: what xsdba.utils.interp_on_quantiles does for a month /
// day-of-year Grouper when the method is not "nearest" (upstream xsdba >= 0.4, re-exported by
// /root/reference/src/xclim/sdba.py:10; documented as the standard configuration: /root/reference/docs/sdba.rst:64-65,
// /root/reference/CHANGELOG.rst:338).  Upstream, per cell:
//   add_cyclic_bounds            the G group rows get coordinates 1 .. G, the last row is copied to 0, the first to G + 1
//   _interp_on_quantiles_2D      scipy.interpolate.griddata((oldx, oldg), oldy, (newx, newg), method="linear") on the non-NaN
//                                nodes: barycentric interpolation on Qhull's DELAUNAY triangulation of the nodes, no rescaling
//                                of the axes (a temperature in K and a month number are the same unit)
//   _extrapolate_on_quantiles    newx below / above np.interp(newg, rows, first / last non-null node): the (row-interpolated)
//                                first / last factor
// newg = Grouper.get_index(interp=True): month - 0.5 + day / days_in_month (fractional), or the integer day of year.
//
// No triangulation is built.  The nodes lie on G + 2 horizontal lines, sorted along each: for a query q
//   1. two non-empty rows around q give a starting triangle that contains q (two neighbours of one row + one of the other);
//   2. the Delaunay triangle that contains q is the optimum of a tiny linear programme — over all convex combinations of
//      sites that reproduce q, minimise sum lambda_i |p_i - q|^2 (the lower hull of the lifted sites) — solved by the dual
//      simplex method: while some site lies INSIDE the circumcircle of the current triangle, bring in the one deepest
//      inside (per row that is the node nearest to the circle's centre: one binary search per row the circle crosses), drop
//      the vertex the ratio test names (barycentric coordinates of q and of the entering site); the new triangle still
//      contains q and the objective falls, so the walk ends at the empty-circle triangle.  0.5-6 pivots per query
//      (tools/experiments/r05/proto_plane.py: agrees with scipy.griddata to 3e-15 on 40 random node sets, value ranges from 0.05
//      to 30 units per group step, i.e. triangles spanning one to a dozen rows);
//   3. the factor is the barycentric combination of the three node factors (fp64), rounded to fp32 like upstream's output.
// What cannot be reproduced: node sets with four or more COCIRCULAR nodes (a regular grid: QDM, whose abscissa is the
// quantile itself in every group) have no unique Delaunay triangulation — scipy's answer there depends on the order Qhull
// happens to visit facets in; this walk keeps the triangle it holds when no site is strictly inside (tolerance 1e-10 R^2).
// On a group row itself (every day-of-year query) both diagonals give the same value.  Duplicated nodes of one row (tied
// quantiles) collapse to the first of them (Qhull keeps one it chooses).
//
// Precision: fp64 throughout (Qhull and LinearNDInterpolator are fp64).  One lane per (time step, cell); node tables packed
// per cell by k_plane_pack (NaN nodes dropped, strictly increasing abscissa) — gathers that hit L2; functional, not tuned.
#include <math.h>
#include <stdlib.h>

#include <vector>

#include "common.h"

namespace {

struct PlaneTabs {
  const float* __restrict__ px;   // (G, nq, C) packed abscissa of the valid nodes of every row
  const float* __restrict__ py;   // (G, nq, C) their factors
  const uint8_t* __restrict__ cnt;  // (G, C) valid nodes per row
  const float* __restrict__ fx;   // (G, C) first / last non-null ABSCISSA (utils._first_and_last_nonnull on oldx alone)
  const float* __restrict__ lx;
  const float* __restrict__ fy;   // (G, C) first / last non-null FACTOR (... on oldy alone)
  const float* __restrict__ ly;
  int G, nq;
  int64_t C;
};

// one thread per (group row, cell): drop the nodes whose abscissa or factor is NaN, collapse runs of equal abscissae
__global__ void __launch_bounds__(XH_BLOCK)
k_plane_pack(const float* __restrict__ xq_all, const double* __restrict__ xq_common, const float* __restrict__ yq_all, int G, int nq,
             int64_t C, float* __restrict__ px, float* __restrict__ py, uint8_t* __restrict__ cnt, float* __restrict__ fx,
             float* __restrict__ lx, float* __restrict__ fy, float* __restrict__ ly) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  const int g = blockIdx.y;
  if (c >= C) return;
  const int64_t base = (int64_t)g * nq * C + c;
  int m = 0;
  float last = 0.f, firstx = xh_nan32(), lastx = xh_nan32(), firsty = xh_nan32(), lasty = xh_nan32();
  for (int k = 0; k < nq; ++k) {
    const float x = xq_all ? xq_all[base + (int64_t)k * C] : (float)xq_common[k];
    const float y = yq_all[base + (int64_t)k * C];
    if (x == x) {
      if (firstx != firstx) firstx = x;
      lastx = x;
    }
    if (y == y) {
      if (firsty != firsty) firsty = y;
      lasty = y;
    }
    if (x == x && y == y && (m == 0 || x > last)) {
      px[base + (int64_t)m * C] = x;
      py[base + (int64_t)m * C] = y;
      last = x;
      ++m;
    }
  }
  cnt[(int64_t)g * C + c] = (uint8_t)m;
  fx[(int64_t)g * C + c] = firstx;
  lx[(int64_t)g * C + c] = lastx;
  fy[(int64_t)g * C + c] = firsty;
  ly[(int64_t)g * C + c] = lasty;
}

// [host-testable: begin] — plain C++ up to the matching end marker: tests/test_plane_host_build.py compiles this stretch with g++
// (no GPU) and checks plane_locate / plane_nearest against scipy.griddata
// row coordinate r in 0 .. G + 1 -> table row (cyclic copies at both ends)
__device__ __forceinline__ int plane_row(int r, int G) { return r == 0 ? G - 1 : (r == G + 1 ? 0 : r - 1); }

struct PlaneCell {
  const PlaneTabs& t;
  int64_t c;
  __device__ __forceinline__ int count(int r) const { return (int)t.cnt[(int64_t)plane_row(r, t.G) * t.C + c]; }
  __device__ __forceinline__ double x(int r, int k) const { return (double)t.px[((int64_t)plane_row(r, t.G) * t.nq + k) * t.C + c]; }
  __device__ __forceinline__ double y(int r, int k) const { return (double)t.py[((int64_t)plane_row(r, t.G) * t.nq + k) * t.C + c]; }
  // number of nodes of row r with abscissa <= v (searchsorted side="right"), n = count(r)
  __device__ __forceinline__ int upper(int r, int n, double v) const {
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (x(r, mid) <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
  }
  // ... with abscissa < v (side="left")
  __device__ __forceinline__ int lower(int r, int n, double v) const {
    int lo = 0, hi = n;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (x(r, mid) < v) lo = mid + 1; else hi = mid;
    }
    return lo;
  }
};

// np.interp between two rows at fraction f in [0, 1): the node value itself on a row (numpy: xp[j] == x short-cut)
__device__ __forceinline__ double plane_lerp(double v0, double v1, double f) {
  if (f == 0.0) return v0;
  const double slope = v1 - v0;
  double r = slope * f + v0;
  if (r != r) {  // numpy's repair of a non-finite product
    r = slope * (f - 1.0) + v1;
    if (r != r && v0 == v1) r = v0;
  }
  return r;
}

struct PlaneTri {
  int r[3], k[3];
  double x[3];
};

// barycentric coordinates of (qx, qy) in the triangle
__device__ __forceinline__ void plane_bary(const PlaneTri& T, double qx, double qy, double (&lam)[3]) {
  const double x0 = T.x[0], y0 = (double)T.r[0];
  const double ax = T.x[1] - x0, ay = (double)T.r[1] - y0, bx = T.x[2] - x0, by = (double)T.r[2] - y0;
  const double d = ax * by - bx * ay;
  const double l1 = ((qx - x0) * by - bx * (qy - y0)) / d;
  const double l2 = (ax * (qy - y0) - (qx - x0) * ay) / d;
  lam[0] = 1.0 - l1 - l2;
  lam[1] = l1;
  lam[2] = l2;
}

__device__ __forceinline__ bool plane_try(const PlaneCell& P, PlaneTri& T, int ra, int ka, int rb, int kb, int rc, int kc, double qx,
                                          double qy) {
  T.r[0] = ra; T.k[0] = ka; T.x[0] = P.x(ra, ka);
  T.r[1] = rb; T.k[1] = kb; T.x[1] = P.x(rb, kb);
  T.r[2] = rc; T.k[2] = kc; T.x[2] = P.x(rc, kc);
  double lam[3];
  plane_bary(T, qx, qy, lam);
  const double mn = fmin(lam[0], fmin(lam[1], lam[2]));
  return mn >= -1e-12;
}

// The factor at (qx, qy), qy in [0, G + 1]; NaN when q lies outside the strip polygon of its two rows (the caller's bounds
// test catches those first) or no two non-empty rows surround it.
__device__ double plane_locate(const PlaneCell& P, double qx, double qy) {
  const int R = P.t.G + 2;
  int r0 = (int)floor(qy);
  r0 = r0 < 0 ? 0 : (r0 > R - 1 ? R - 1 : r0);
  const bool onrow = (double)r0 == qy;
  int r1 = r0 + 1;
  while (r0 >= 0 && P.count(r0) == 0) --r0;
  if (r0 < 0) return xh_nan64();
  while (r1 < R && P.count(r1) == 0) ++r1;
  if (r1 >= R) {  // nothing above: only a query ON row r0 can still be served, by an apex row below
    if (!(onrow && (double)r0 == qy)) return xh_nan64();
    r1 = r0 - 1;
    while (r1 >= 0 && P.count(r1) == 0) --r1;
    if (r1 < 0) return xh_nan64();
  }
  const int nA = P.count(r0), nB = P.count(r1);
  const double f = (qy - (double)r0) / (double)(r1 - r0);
  // s(i, j) = (1 - f) A_i + f B_j: the abscissa of segment A_i - B_j at the height of q
  const double A0 = P.x(r0, 0), B0 = P.x(r1, 0), AL = P.x(r0, nA - 1), BL = P.x(r1, nB - 1);
  if (qx < (1.0 - f) * A0 + f * B0 || qx > (1.0 - f) * AL + f * BL) return xh_nan64();
  PlaneTri T;
  bool have = false;
  {
    // A good start matters more than a fast step: neighbouring rows are shifted against each other by several node gaps
    // (the seasonal cycle between two months), and from a fan triangle the walk needs many pivots, each scanning every row
    // its huge circle crosses.  The Delaunay triangulation of the TWO rows alone is known in closed form — the apex of a
    // row's gap is the other row's node nearest to the gap's midpoint — and one of its triangles contains q; it is looked
    // for among the gaps around q's brackets (alternating sides, both base rows), and it usually IS the answer.
    auto nearest = [&](int r, int n, double m) -> int {
      int k = P.lower(r, n, m);
      if (k >= n) return n - 1;
      if (k > 0 && m - P.x(r, k - 1) <= P.x(r, k) - m) --k;
      return k;
    };
    const int iA = P.upper(r0, nA, qx) - 1, jB = P.upper(r1, nB, qx) - 1;
#pragma unroll 1
    for (int k = 0; k <= 16 && !have; ++k) {
      const int dk = (k & 1) ? (k + 1) / 2 : -(k / 2);   // 0, +1, -1, +2, -2 ...
      const int i = iA + dk, j = jB + dk;
      if (nA >= 2 && i >= 0 && i <= nA - 2) {
        const double a0 = P.x(r0, i), a1 = P.x(r0, i + 1);
        const int jj = nearest(r1, nB, 0.5 * (a0 + a1));
        const double b = P.x(r1, jj);
        if ((1.0 - f) * a0 + f * b <= qx && qx <= (1.0 - f) * a1 + f * b) have = plane_try(P, T, r0, i, r0, i + 1, r1, jj, qx, qy);
      }
      if (!have && nB >= 2 && j >= 0 && j <= nB - 2) {
        const double b0 = P.x(r1, j), b1 = P.x(r1, j + 1);
        const int ii = nearest(r0, nA, 0.5 * (b0 + b1));
        const double a = P.x(r0, ii);
        if ((1.0 - f) * a + f * b0 <= qx && qx <= (1.0 - f) * a + f * b1) have = plane_try(P, T, r0, ii, r1, j, r1, j + 1, qx, qy);
      }
    }
  }
  if (!have) {
    int i = P.upper(r0, nA, qx) - 1, j = P.upper(r1, nB, qx) - 1;
    const int imax = nA >= 2 ? nA - 2 : 0, jmax = nB >= 2 ? nB - 2 : 0;
    i = i < 0 ? 0 : (i > imax ? imax : i);
    j = j < 0 ? 0 : (j > jmax ? jmax : j);
    if (nA >= 2) have = plane_try(P, T, r0, i, r0, i + 1, r1, j, qx, qy);
    if (!have && nA >= 2 && nB >= 2) have = plane_try(P, T, r0, i, r0, i + 1, r1, j + 1, qx, qy);
    if (!have && nB >= 2) have = plane_try(P, T, r0, i, r1, j, r1, j + 1, qx, qy);
    if (!have && nA >= 2 && nB >= 2) have = plane_try(P, T, r0, i + 1, r1, j, r1, j + 1, qx, qy);
  }
  if (!have) {
    // the fan from A_0 over B, then from B_last over A: j = last index with s(0, j) <= qx
    int lo = 0, hi = nB;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if ((1.0 - f) * A0 + f * P.x(r1, mid) <= qx) lo = mid + 1; else hi = mid;
    }
    const int jj = lo - 1;
    if (jj < nB - 1) {
      plane_try(P, T, r0, 0, r1, jj < 0 ? 0 : jj, r1, (jj < 0 ? 0 : jj) + 1, qx, qy);
    } else {
      lo = 0; hi = nA;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((1.0 - f) * P.x(r0, mid) + f * BL <= qx) lo = mid + 1; else hi = mid;
      }
      int ii = lo - 1;
      ii = ii < 0 ? 0 : (ii > nA - 2 ? nA - 2 : ii);
      if (nA < 2) return xh_nan64();  // (one node in each row and q not on their segment's end: caught by the bounds)
      plane_try(P, T, r0, ii, r0, ii + 1, r1, nB - 1, qx, qy);
    }
  }
  // ---- dual simplex walk to the empty-circle triangle
  double lam[3];
#pragma unroll 1
  for (int it = 0; it < 96; ++it) {
    const double x0 = T.x[0], y0 = (double)T.r[0];
    const double ax = T.x[1] - x0, ay = (double)T.r[1] - y0, bx = T.x[2] - x0, by = (double)T.r[2] - y0;
    const double d = 2.0 * (ax * by - ay * bx);
    const double a2 = ax * ax + ay * ay, b2 = bx * bx + by * by;
    const double ux = (by * a2 - ay * b2) / d, uy = (ax * b2 - bx * a2) / d;
    const double cx = x0 + ux, cy = y0 + uy, R2 = ux * ux + uy * uy;
    if (!(R2 < 1e300)) break;  // degenerate start (collinear): keep it
    const double Rr = sqrt(R2);
    int ra = (int)ceil(cy - Rr), rb = (int)floor(cy + Rr);
    ra = ra < 0 ? 0 : ra;
    rb = rb > R - 1 ? R - 1 : rb;
    double bestp = -1e-10 * fmax(R2, 1.0);
    int br = -1, bk = 0;
    double bxv = 0.0;
#pragma unroll 1
    for (int r = ra; r <= rb; ++r) {
      const int n = P.count(r);
      if (n == 0) continue;
      const double h = (double)r - cy;
      const double w2 = R2 - h * h;
      if (!(w2 > 0.0)) continue;
      const int kk = P.lower(r, n, cx);
#pragma unroll 1
      for (int k = kk - 1; k <= kk; ++k) {
        if (k < 0 || k >= n) continue;
        if ((r == T.r[0] && k == T.k[0]) || (r == T.r[1] && k == T.k[1]) || (r == T.r[2] && k == T.k[2])) continue;
        const double xv = P.x(r, k);
        const double pw = (xv - cx) * (xv - cx) - w2;
        if (pw < bestp) { bestp = pw; br = r; bk = k; bxv = xv; }
      }
    }
    if (br < 0) break;
    double mu[3];
    plane_bary(T, qx, qy, lam);
    plane_bary(T, bxv, (double)br, mu);
    int kout = -1;
    double best = 0.0;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
      if (mu[v] > 1e-14) {
        const double ratio = lam[v] / mu[v];
        if (kout < 0 || ratio < best) { kout = v; best = ratio; }
      }
    }
    if (kout < 0) break;
    // (static indexing: a dynamic index into the register arrays becomes a select chain anyway)
#pragma unroll
    for (int v = 0; v < 3; ++v)
      if (v == kout) { T.r[v] = br; T.k[v] = bk; T.x[v] = bxv; }
  }
  plane_bary(T, qx, qy, lam);
  return lam[0] * P.y(T.r[0], T.k[0]) + lam[1] * P.y(T.r[1], T.k[1]) + lam[2] * P.y(T.r[2], T.k[2]);
}

// method "nearest" (scipy griddata over the nodes of all groups: a cKDTree query in the (abscissa, group coordinate) plane, no
// rescaling): the own row first, then the rows k = 1, 2 ... away (below before above) while a node there could still be
// nearer (k^2 < best squared distance), the first of equally near nodes wins — the order of k_eqm_adjust_g2d (eqm.hip).
// Exact ties are NOT rare with windowed day-of-year groups: the groups d - 1 and d + 1 share most of their samples, so their
// extreme nodes are often the same value, equally far from a query on row d; scipy's cKDTree returns either of them (230 : 170
// in 400 constructed ties, tests/test_gpu_api.py::test_grouper_add_dims_pools_the_members) — not reproducible, documented.
__device__ double plane_nearest(const PlaneCell& P, double x, int r) {
  const int G = P.t.G;
  double best = __longlong_as_double(0x7FF0000000000000LL), a = xh_nan64();
  const int n = P.count(r);
  for (int j = 0; j < n; ++j) {
    const double dx = x - P.x(r, j), d2 = dx * dx;
    if (d2 < best) { best = d2; a = P.y(r, j); }
  }
  for (int k = 1; (double)k * (double)k < best && k <= G + 1; ++k) {
#pragma unroll 1
    for (int sgn = -1; sgn <= 1; sgn += 2) {
      const int gg = r + sgn * k;
      if (gg < 0 || gg > G + 1) continue;
      const int m = P.count(gg);
      for (int j = 0; j < m; ++j) {
        const double dx = x - P.x(gg, j), d2 = dx * dx + (double)k * (double)k;
        if (d2 < best) { best = d2; a = P.y(gg, j); }
      }
    }
  }
  return a;
}

// [host-testable: end]
__global__ void __launch_bounds__(XH_BLOCK)
k_plane_linear(const float* __restrict__ xnew, const float* __restrict__ base, int64_t T, int64_t st, const double* __restrict__ gnew,
               PlaneTabs tabs, int kind, float* __restrict__ scen, int64_t scen_st) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= tabs.C) return;
  const PlaneCell P{tabs, c};
  const int G = tabs.G;
  const int64_t chunk = cdiv64(T, (int64_t)gridDim.y);
  const int64_t ta = (int64_t)blockIdx.y * chunk;
  int64_t tb = ta + chunk;
  if (tb > T) tb = T;
  for (int64_t t = ta; t < tb; ++t) {
    const float xf = xnew[t * st + c];
    const float bf = base ? base[t * st + c] : xf;
    float a = xh_nan32();
    if (xf == xf) {
      const double x = (double)xf;
      double g = gnew[t];
      g = g < 0.0 ? 0.0 : (g > (double)(G + 1) ? (double)(G + 1) : g);
      int r0 = (int)floor(g);
      if (r0 > G) r0 = G;  // (g == G + 1: the last interval, f = 1 — np.interp returns the end value)
      const double f = g - (double)r0;
      const int64_t i0 = (int64_t)plane_row(r0, G) * tabs.C + c, i1 = (int64_t)plane_row(r0 + 1, G) * tabs.C + c;
      const double lo = f == 1.0 ? (double)tabs.fx[i1] : plane_lerp((double)tabs.fx[i0], (double)tabs.fx[i1], f);
      const double hi = f == 1.0 ? (double)tabs.lx[i1] : plane_lerp((double)tabs.lx[i0], (double)tabs.lx[i1], f);
      double ad;
      if (x < lo) ad = f == 1.0 ? (double)tabs.fy[i1] : plane_lerp((double)tabs.fy[i0], (double)tabs.fy[i1], f);
      else if (x > hi) ad = f == 1.0 ? (double)tabs.ly[i1] : plane_lerp((double)tabs.ly[i0], (double)tabs.ly[i1], f);
      else ad = plane_locate(P, x, g);
      a = (float)ad;
    }
    scen[t * scen_st + c] = kind == 0 ? (bf + a) : (kind == 1 ? (bf * a) : a);
  }
}

// Integer group coordinates (the day-of-year grouping: every query lies ON its group's row).  One lane per cell walks the
// rows of its chunk; a row's packed nodes (abscissae + factors) are loaded ONCE into registers — coalesced across the lanes,
// all loads in flight together — and serve every time step of that row (30 of them for 30 years).  Between two neighbouring
// nodes less than 2 group steps apart the row's own edge belongs to every Delaunay triangulation (its diametral circle
// reaches no other row: Gabriel edge), so the plane interpolation is the linear interpolation along the row; wider gaps
// (temperature tails, precipitation in mm/day) take the walk of plane_locate.  rows: CSR by row coordinate 1 .. G
// (roff[r - 1] .. roff[r]) of the time steps with that coordinate.
// NEAREST: the nearest node of the plane instead — the own row's nearest node stands whenever it is at most one group step away
// (any node of another row is at least that far), else the query is listed; bounds: the own row's, constant or NaN.
template <int NQMAX, bool NEAREST>
__global__ void __launch_bounds__(XH_BLOCK)
k_plane_rows(const float* __restrict__ xnew, const float* __restrict__ base, int64_t st, const int32_t* __restrict__ roff,
             const int32_t* __restrict__ rsteps, PlaneTabs tabs, int kind, float* __restrict__ scen, int64_t scen_st, int r_first,
             int r_end, uint2* __restrict__ work, unsigned int* __restrict__ nwork, unsigned long long segcap, int abl, int extrap) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= tabs.C) return;
  // this workgroup's own stretch of the work list and its own counter: one counter for the whole grid was an atomic on ONE
  // address per wave and step
  const unsigned long long seg = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;
  uint2* __restrict__ mywork = work + seg * segcap;
  unsigned int* __restrict__ mycount = nwork + seg;
  const int nq = tabs.nq;
  const int rchunk = (r_end - r_first + (int)gridDim.y - 1) / (int)gridDim.y;
  const int ra = r_first + (int)blockIdx.y * rchunk;
  int rb = ra + rchunk;
  if (rb > r_end) rb = r_end;
  for (int r = ra; r < rb; ++r) {
    const int32_t t0 = roff[r - 1], t1 = roff[r];
    if (t0 == t1) continue;
    const int64_t rowi = (int64_t)(r - 1) * tabs.C + c;
    const int n = (int)tabs.cnt[rowi];
    float nx[NQMAX], ny[NQMAX];
#pragma unroll
    for (int j = 0; j < NQMAX; ++j) {
      const bool in = j < nq;
      nx[j] = in ? tabs.px[((int64_t)(r - 1) * nq + j) * tabs.C + c] : 0.f;
      ny[j] = in ? tabs.py[((int64_t)(r - 1) * nq + j) * tabs.C + c] : 0.f;
    }
    const float PINF = __uint_as_float(0x7F800000u);
#pragma unroll
    for (int j = 0; j < NQMAX; ++j) nx[j] = j < n ? nx[j] : PINF;  // (padding never counts as "<= x")
    const float lo = tabs.fx[rowi], hi = tabs.lx[rowi], flo = tabs.fy[rowi], fhi = tabs.ly[rowi];
    for (int32_t i = t0; i < t1; ++i) {
      const int64_t t = rsteps[i];
      const float xf = xnew[t * st + c];
      const float bf = base ? base[t * st + c] : xf;
      float a = xh_nan32();
      if (xf == xf) {
        bool fast = false;
        if (NEAREST) {
          // (the bounds come AFTER the search upstream, _extrapolate_on_quantiles, but decide alone: outside them the search's
          // result is overwritten)
          if (xf < lo) { a = extrap == 0 ? flo : xh_nan32(); fast = true; }
          else if (xf > hi) { a = extrap == 0 ? fhi : xh_nan32(); fast = true; }
          else {
            double best = __longlong_as_double(0x7FF0000000000000LL);
#pragma unroll
            for (int j = 0; j < NQMAX; ++j) {
              const double dx = (double)xf - (double)nx[j], d2 = dx * dx;  // (padding: +inf, never the minimum)
              if (d2 < best) { best = d2; a = ny[j]; }
            }
            fast = best <= 1.0;
          }
        } else if (xf < lo) { a = flo; fast = true; }
        else if (xf > hi) { a = fhi; fast = true; }
        else {
          int idx = 0;  // nodes <= x, by counting (the nodes ascend)
#pragma unroll
          for (int j = 0; j < NQMAX; ++j) idx += xf >= nx[j] ? 1 : 0;
          if (idx >= 1 && idx < n) {
            float x0 = nx[0], x1 = nx[1], y0 = ny[0], y1 = ny[1];
#pragma unroll
            for (int j = 1; j < NQMAX - 1; ++j) {
              const bool s = j == idx - 1;
              x0 = s ? nx[j] : x0; x1 = s ? nx[j + 1] : x1;
              y0 = s ? ny[j] : y0; y1 = s ? ny[j + 1] : y1;
            }
            if ((double)x1 - (double)x0 < 2.0) {
              const double l = ((double)xf - (double)x0) / ((double)x1 - (double)x0);
              a = (float)((1.0 - l) * (double)y0 + l * (double)y1);
              fast = true;
            }
          } else if (idx >= 1 && idx == n) {  // x on the row's last valid node (beyond it: the walk decides)
            float xl = nx[0], yl = ny[0];
#pragma unroll
            for (int j = 1; j < NQMAX; ++j) {
              xl = j == n - 1 ? nx[j] : xl;
              yl = j == n - 1 ? ny[j] : yl;
            }
            if (xf == xl) {
              a = yl;
              fast = true;
            }
          }
        }
        {
          if (!fast) {
            // the walk is a chain of dependent gathers: done here, a handful of lanes would hold the whole wave for tens of
            // microseconds at EVERY step (measured: 211 ms against 442 for the walk alone) — the query goes on a list that
            // k_plane_work runs through with full waves (one atomic per wave: the lanes' slots by ballot)
            if (abl & 1) continue;  // diagnostics: the row kernel alone (unresolved queries stay unwritten)
            const unsigned long long m = __ballot(1);
            const int lane = threadIdx.x & 63;
            unsigned int b0 = 0;
            if (lane == __ffsll((long long)m) - 1) b0 = atomicAdd(mycount, (unsigned int)__popcll(m));
            b0 = __shfl(b0, __ffsll((long long)m) - 1);
            mywork[b0 + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = make_uint2((uint32_t)t, (uint32_t)c);
            continue;
          }
        }
      }
      scen[t * scen_st + c] = kind == 0 ? (bf + a) : (kind == 1 ? (bf * a) : a);
    }
  }
}

// Fractional group coordinates (the month grouping: newg = month - 0.5 + day / days_in_month).  One lane per cell walks the
// steps of its chunk in time order; the coordinate is the same for every cell of a step, so the two rows around it (r0 =
// floor(g), r0 + 1) are the same for the ~15 consecutive days of half a month: their packed nodes are loaded ONCE — abscissae
// into registers, abscissae and factors into lane-private LDS columns (20 KB per wave: 2 waves per SIMD; with the rows and
// the apex tables all in LDS or all in registers it was one wave per SIMD and 3.2 x slower: the fp64 chains of a single wave
// issue an instruction every 7 cycles).  The Delaunay triangulation of two rows ALONE is known in closed form — every gap of
// a row is the base of exactly one triangle, whose apex is the other row's node nearest to the gap's midpoint (its
// circumcentre lies on the gap's bisector, so that apex leaves the circle empty of both rows' nodes) — and these triangles
// tile the strip.  Per pair of rows the apex of every gap is found once (38 binary searches over the LDS columns in lockstep,
// fp64, the tie rule of plane_locate's start) and kept as its abscissa (registers) + its 5-bit index (packed); per query the
// cuts of the triangles at the query's height, [(1 - f) a_i + f b, (1 - f) a_{i+1} + f b], ascend with the gap's index, so the
// triangle is the LAST one whose cut starts at or left of x: a count over registers, then 8 reads of the LDS columns for its
// vertices.  That triangle is a Delaunay triangle of the WHOLE node set iff its circumcircle reaches no third row
// (ceil(cy - R) >= r0 and floor(cy + R) <= r0 + 1); else — wide gaps (precipitation in mm/day, the tails of a temperature
// distribution) or neighbouring months shifted by several kelvin (a seasonal cycle: ~30 % of the queries of bench.py's
// extra.eqm_month_linear) — the query goes on the work list for the walk of plane_locate.
// 30 years x 1440 x 90, 12 groups x 20 nodes: 300 ms through k_plane_linear -> 125 ms (pair kernel 35 + walk 85 + lists).
constexpr int PP_BLOCK = 64;  // (20 KB of LDS per wave at 20 nodes: 8 waves per CU)
template <int NQMAX>
__global__ void __launch_bounds__(PP_BLOCK)
k_plane_pair(const float* __restrict__ xnew, const float* __restrict__ base, int64_t st, const double* __restrict__ gnew, int64_t t_first,
             int64_t t_end, PlaneTabs tabs, int kind, float* __restrict__ scen, int64_t scen_st, uint2* __restrict__ work,
             unsigned int* __restrict__ nwork, unsigned long long segcap, int abl) {
  const int64_t c = (int64_t)blockIdx.x * PP_BLOCK + threadIdx.x;
  if (c >= tabs.C) return;
  const unsigned long long seg = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;
  uint2* __restrict__ mywork = work + seg * segcap;
  unsigned int* __restrict__ mycount = nwork + seg;
  const int nq = tabs.nq, G = tabs.G;
  const int64_t chunk = cdiv64(t_end - t_first, (int64_t)gridDim.y);
  const int64_t ta = t_first + (int64_t)blockIdx.y * chunk;
  int64_t tb = ta + chunk;
  if (tb > t_end) tb = t_end;
  const float PINF = __uint_as_float(0x7F800000u);
  // lane-private LDS columns: the two rows ([0] = r0, [1] = r0 + 1) — whatever needs a run-time index; the abscissae the cut
  // test runs over also sit in registers, the apex of a gap as its 5-bit index into the other row (5 gaps per word)
  __shared__ float sx_[2][NQMAX][PP_BLOCK], sy_[2][NQMAX][PP_BLOCK];
  static_assert(NQMAX <= 32 && (NQMAX - 1 + 4) / 5 <= 4, "apex indices: 5 bits each, four words per row");
  float ax[NQMAX], bx[NQMAX];
  float pbx[NQMAX - 1], pax[NQMAX - 1];  // apex abscissa of A's gap i (a node of B), of B's gap j (a node of A)
  uint32_t kA[4] = {0u, 0u, 0u, 0u}, kB[4] = {0u, 0u, 0u, 0u};
  int nA = 0, nB = 0, cur = -1;
  float fxA = 0.f, lxA = 0.f, fyA = 0.f, lyA = 0.f, fxB = 0.f, lxB = 0.f, fyB = 0.f, lyB = 0.f;
  float xf_next = ta < tb ? xnew[ta * st + c] : 0.f, bf_next = (base && ta < tb) ? base[ta * st + c] : 0.f;
  for (int64_t t = ta; t < tb; ++t) {
    double g = gnew[t];
    g = g < 0.0 ? 0.0 : (g > (double)(G + 1) ? (double)(G + 1) : g);
    int r0 = (int)floor(g);
    if (r0 > G) r0 = G;  // (g == G + 1: the last interval, f = 1)
    const double f = g - (double)r0;
    if (r0 != cur) {  // (the same for every lane: the coordinate belongs to the step)
      cur = r0;
      const int rowA = plane_row(r0, G), rowB = plane_row(r0 + 1, G);
      const int64_t iA = (int64_t)rowA * tabs.C + c, iB = (int64_t)rowB * tabs.C + c;
      nA = (int)tabs.cnt[iA];
      nB = (int)tabs.cnt[iB];
      // (entries past nq re-read the last one: the counts below pad them; a select on `j < nq` per entry made the compiler keep
      //  80 uniform masks alive across the whole kernel)
      const float* __restrict__ pxa = tabs.px + (int64_t)rowA * nq * tabs.C + c;
      const float* __restrict__ pya = tabs.py + (int64_t)rowA * nq * tabs.C + c;
      const float* __restrict__ pxb = tabs.px + (int64_t)rowB * nq * tabs.C + c;
      const float* __restrict__ pyb = tabs.py + (int64_t)rowB * nq * tabs.C + c;
#pragma unroll
      for (int j = 0; j < NQMAX; ++j) {
        const int64_t o = (int64_t)(j < nq ? j : nq - 1) * tabs.C;
        ax[j] = pxa[o];
        bx[j] = pxb[o];
        sy_[0][j][threadIdx.x] = pya[o];
        sy_[1][j][threadIdx.x] = pyb[o];
      }
#pragma unroll
      for (int j = 0; j < NQMAX; ++j) {
        ax[j] = j < nA ? ax[j] : PINF;
        bx[j] = j < nB ? bx[j] : PINF;
        sx_[0][j][threadIdx.x] = ax[j];
        sx_[1][j][threadIdx.x] = bx[j];
      }
      fxA = tabs.fx[iA]; lxA = tabs.lx[iA]; fyA = tabs.fy[iA]; lyA = tabs.ly[iA];
      fxB = tabs.fx[iB]; lxB = tabs.lx[iB]; fyB = tabs.fy[iB]; lyB = tabs.ly[iB];
      // apex of every gap: the other row's node nearest to the gap's midpoint, the lower one of two equally near (the rule of
      // plane_locate's start); NaN for a gap past the row's last node: every comparison with its cut fails.
      // all 38 searches in lockstep (5 halvings, the reads of one level in flight together): lo = nodes of the other row < the
      // gap's midpoint; then plane_locate's rule: the lower of two equally near nodes
      int la[NQMAX - 1], ha[NQMAX - 1], lb[NQMAX - 1], hb[NQMAX - 1];
#pragma unroll
      for (int i = 0; i < NQMAX - 1; ++i) { la[i] = 0; ha[i] = nB; lb[i] = 0; hb[i] = nA; }
#pragma unroll 1
      for (int it = 0; it < 5; ++it) {
#pragma unroll
        for (int i = 0; i < NQMAX - 1; ++i) {
          const double mA = 0.5 * ((double)ax[i] + (double)ax[i + 1]), mB = 0.5 * ((double)bx[i] + (double)bx[i + 1]);
          const int ma = (la[i] + ha[i]) >> 1, mb = (lb[i] + hb[i]) >> 1;
          const float va = sx_[1][ma < NQMAX ? ma : NQMAX - 1][threadIdx.x], vb = sx_[0][mb < NQMAX ? mb : NQMAX - 1][threadIdx.x];
          const bool oa = la[i] < ha[i], ob = lb[i] < hb[i];
          const bool ga = oa && (double)va < mA, gb = ob && (double)vb < mB;
          la[i] = ga ? ma + 1 : la[i];
          ha[i] = (oa && !ga) ? ma : ha[i];
          lb[i] = gb ? mb + 1 : lb[i];
          hb[i] = (ob && !gb) ? mb : hb[i];
        }
      }
#pragma unroll
      for (int i = 0; i < NQMAX - 1; ++i) {
        const double mA = 0.5 * ((double)ax[i] + (double)ax[i + 1]), mB = 0.5 * ((double)bx[i] + (double)bx[i + 1]);
        auto pick = [&](int row, int lo, int n, double m, float& vx) -> uint32_t {
          const int l1 = lo > 0 ? lo - 1 : 0, l0 = lo < NQMAX ? lo : NQMAX - 1;
          const double below = (double)sx_[row][l1][threadIdx.x], at = (double)sx_[row][l0][threadIdx.x];
          int k = lo >= n ? n - 1 : lo;
          if (lo < n && lo > 0 && m - below <= at - m) k = lo - 1;
          k = k < 0 ? 0 : k;
          vx = sx_[row][k][threadIdx.x];
          return (uint32_t)k;
        };
        float vx, wx;
        const uint32_t ka = pick(1, la[i], nB, mA, vx), kb = pick(0, lb[i], nA, mB, wx);
        pbx[i] = (i + 1 < nA && nB >= 1) ? vx : xh_nan32();
        pax[i] = (i + 1 < nB && nA >= 1) ? wx : xh_nan32();
        if (i % 5 == 0) { kA[i / 5] = 0u; kB[i / 5] = 0u; }
        kA[i / 5] |= ka << ((i % 5) * 5);
        kB[i / 5] |= kb << ((i % 5) * 5);
      }
    }
    const float xf = xf_next;
    const float bf = base ? bf_next : xf;
    if (t + 1 < tb) {  // (one wave per SIMD: the next step's sample is on its way while this one is located)
      xf_next = xnew[(t + 1) * st + c];
      if (base) bf_next = base[(t + 1) * st + c];
    }
    float a = xh_nan32();
    if (xf == xf) {
      const double x = (double)xf;
      const double lo = f == 1.0 ? (double)fxB : plane_lerp((double)fxA, (double)fxB, f);
      const double hi = f == 1.0 ? (double)lxB : plane_lerp((double)lxA, (double)lxB, f);
      bool done = false;
      if (x < lo) { a = (float)(f == 1.0 ? (double)fyB : plane_lerp((double)fyA, (double)fyB, f)); done = true; }
      else if (x > hi) { a = (float)(f == 1.0 ? (double)lyB : plane_lerp((double)lyA, (double)lyB, f)); done = true; }
      else if (nA >= 1 && nB >= 1 && f < 1.0) {
        // the triangle of the two-row tiling whose cut at height f holds x.  The left ends of the cuts ascend with the gap's
        // index (the nodes ascend and so do the apexes), so the candidate is the LAST gap whose cut starts at or left of x: a
        // count over the registers; its vertices come from the LDS columns.
        const double w0 = 1.0 - f;
        int cA = 0, cB = 0;
#pragma unroll
        for (int i = 0; i < NQMAX - 1; ++i) {
          cA += (w0 * (double)ax[i] + f * (double)pbx[i] <= x) ? 1 : 0;   // (NaN apex: a gap that does not exist)
          cB += (w0 * (double)pax[i] + f * (double)bx[i] <= x) ? 1 : 0;
        }
        double x0 = 0.0, x1 = 0.0, x2 = 0.0;
        float y0 = 0.f, y1 = 0.f, y2 = 0.f;
        int typ = 0;  // 1: base on A (P0, P1 on row r0, P2 on r0 + 1), 2: base on B (P0 on r0, P1, P2 on r0 + 1)
        auto apex_of = [&](const uint32_t (&kw)[4], int i) -> int {
          const int w = i / 5;
          const uint32_t word = w == 0 ? kw[0] : (w == 1 ? kw[1] : (w == 2 ? kw[2] : kw[3]));
          return (int)((word >> (uint32_t)((i - w * 5) * 5)) & 31u);
        };
        if (cA >= 1) {
          const int i = cA - 1, k = apex_of(kA, i);
          const double a1 = (double)sx_[0][i + 1][threadIdx.x], pb = (double)sx_[1][k][threadIdx.x];
          if (x <= w0 * a1 + f * pb) {
            typ = 1;
            x0 = (double)sx_[0][i][threadIdx.x]; x1 = a1; x2 = pb;
            y0 = sy_[0][i][threadIdx.x]; y1 = sy_[0][i + 1][threadIdx.x]; y2 = sy_[1][k][threadIdx.x];
          }
        }
        if (typ == 0 && cB >= 1) {
          const int j = cB - 1, k = apex_of(kB, j);
          const double b1 = (double)sx_[1][j + 1][threadIdx.x], pa = (double)sx_[0][k][threadIdx.x];
          if (x <= w0 * pa + f * b1) {
            typ = 2;
            x0 = pa; x1 = (double)sx_[1][j][threadIdx.x]; x2 = b1;
            y0 = sy_[0][k][threadIdx.x]; y1 = sy_[1][j][threadIdx.x]; y2 = sy_[1][j + 1][threadIdx.x];
          }
        }
        if (typ != 0) {
          // circumcircle (the formulas of plane_locate); rows are r0 (y = 0) and r0 + 1 (y = 1) here
          const double yy1 = typ == 1 ? 0.0 : 1.0;  // P1's row; P0 on row 0, P2 on row 1
          const double ax_ = x1 - x0, ay_ = yy1, bx_ = x2 - x0, by_ = 1.0;
          const double d = 2.0 * (ax_ * by_ - ay_ * bx_);
          const double a2 = ax_ * ax_ + ay_ * ay_, b2 = bx_ * bx_ + by_ * by_;
          const double ux = (by_ * a2 - ay_ * b2) / d, uy = (ax_ * b2 - bx_ * a2) / d;
          const double R2 = ux * ux + uy * uy;
          const double Rr = sqrt(R2);
          if (R2 < 1e300 && ceil(uy - Rr) >= 0.0 && floor(uy + Rr) <= 1.0) {
            const double l1 = ((x - x0) * by_ - bx_ * f) / (0.5 * d);
            const double l2 = (ax_ * f - (x - x0) * ay_) / (0.5 * d);
            a = (float)((1.0 - l1 - l2) * (double)y0 + l1 * (double)y1 + l2 * (double)y2);
            done = true;
          }
        }
      }
      if (!done) {
        if (abl & 1) continue;  // diagnostics: the pair kernel alone
        const unsigned long long m = __ballot(1);
        const int lane = threadIdx.x & 63;
        unsigned int b0 = 0;
        if (lane == __ffsll((long long)m) - 1) b0 = atomicAdd(mycount, (unsigned int)__popcll(m));
        b0 = __shfl(b0, __ffsll((long long)m) - 1);
        mywork[b0 + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = make_uint2((uint32_t)t, (uint32_t)c);
        continue;
      }
    }
    scen[t * scen_st + c] = kind == 0 ? (bf + a) : (kind == 1 ? (bf * a) : a);
  }
}

// the queries the row kernel could not decide from the row alone: every workgroup runs through the stretch of the list its
// twin of the row kernel filled, one lane per entry, the Delaunay walk of plane_locate
__global__ void __launch_bounds__(XH_BLOCK)
k_plane_work(const float* __restrict__ xnew, const float* __restrict__ base, int64_t st, const double* __restrict__ gnew, PlaneTabs tabs,
             int kind, float* __restrict__ scen, int64_t scen_st, const uint2* __restrict__ work, const unsigned int* __restrict__ nwork,
             unsigned long long segcap, int nearest) {
  const unsigned long long seg = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;
  const unsigned int n = nwork[seg];
  const uint2* __restrict__ mywork = work + seg * segcap;
  for (unsigned int i = threadIdx.x; i < n; i += XH_BLOCK) {
    const uint2 e = mywork[i];
    const int64_t t = e.x, c = e.y;
    const PlaneCell P{tabs, c};
    const float xf = xnew[t * st + c];
    const float bf = base ? base[t * st + c] : xf;
    // (listed queries lie inside their row's bounds: the row kernel decided the others)
    const float a = (float)(nearest ? plane_nearest(P, (double)xf, (int)gnew[t]) : plane_locate(P, (double)xf, gnew[t]));
    scen[t * scen_st + c] = kind == 0 ? (bf + a) : (kind == 1 ? (bf * a) : a);
  }
}

// method 1 = "linear" (Delaunay interpolation), 0 = "nearest" (integer coordinates only; extrap 0 constant | 1 nan)
int plane_run(xh_ctx* ctx, const float* xnew, const float* base, int64_t T, int64_t C, int64_t st, const double* gnew,
              const float* xq_all, const double* xq_common, const float* yq_all, int G, int nq, int kind, float* scen,
              int64_t scen_st, int method, int extrap) {
  XH_REQUIRE(ctx && xnew && gnew && yq_all && scen, XH_ERR_ARG, "xh_plane_linear: NULL argument");
  XH_REQUIRE((xq_all != nullptr) != (xq_common != nullptr), XH_ERR_ARG, "xh_plane_linear: give xq_all (G, nq, C) OR xq_common (nq)");
  XH_REQUIRE(T >= 0 && C >= 0 && nq >= 1 && nq <= 255 && G >= 1, XH_ERR_ARG, "xh_plane_linear: bad shape (1 <= nq <= 255, G >= 1)");
  XH_REQUIRE(st >= C && scen_st >= C, XH_ERR_LAYOUT, "xh_plane_linear: needs time-major views");
  XH_REQUIRE(kind >= 0 && kind <= 2, XH_ERR_ARG, "xh_plane_linear: kind must be 0 (+), 1 (*) or 2 (the factor only)");
  if (T == 0 || C == 0) return XH_OK;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t b_tab = al(4 * (size_t)G * nq * C), b_cnt = al((size_t)G * C), b_row = al(4 * (size_t)G * C), b_q = al(8 * (size_t)nq);
  // work list of the row kernel: (step, cell) pairs, at most 2^27 of them (1 GiB) or every query of the call
  size_t wcap = (size_t)1 << 27;
  {
    const unsigned long long all = (unsigned long long)T * XH_BLOCK * (unsigned long long)cdiv64(C, XH_BLOCK);  // every query, cells padded to whole workgroups
    if (all < wcap) wcap = (size_t)all;
  }
  const size_t b_work = al(8 * wcap) + 4 * 65536;
  void* ws = nullptr;
  int rc = xh_big_scratch(ctx, 2 * b_tab + b_cnt + 4 * b_row + b_q + b_work, &ws);
  if (rc) return rc;
  char* p = (char*)ws;
  float* px = (float*)p; p += b_tab;
  float* py = (float*)p; p += b_tab;
  uint8_t* cnt = (uint8_t*)p; p += b_cnt;
  float* fx = (float*)p; p += b_row;
  float* lx = (float*)p; p += b_row;
  float* fy = (float*)p; p += b_row;
  float* ly = (float*)p; p += b_row;
  double* dq = (double*)p; p += b_q;
  unsigned int* nwork = (unsigned int*)p;   // one counter per workgroup of the row kernel (<= 65536)
  uint2* work = (uint2*)(p + 4 * 65536);
  if (xq_common) XH_CHECK_HIP(hipMemcpyAsync(dq, xq_common, 8 * (size_t)nq, hipMemcpyHostToDevice, ctx->stream));
  const int64_t cblocks = cdiv64(C, XH_BLOCK);
  hipLaunchKernelGGL(k_plane_pack, dim3((unsigned)cblocks, (unsigned)G), dim3(XH_BLOCK), 0, ctx->stream, xq_all, xq_common ? dq : nullptr,
                     yq_all, G, nq, C, px, py, cnt, fx, lx, fy, ly);
  XH_LAUNCH_CHECK();
  PlaneTabs tabs{px, py, cnt, fx, lx, fy, ly, G, nq, C};
  // integer group coordinates in 1 .. G (day-of-year groupings): the row kernel.  gnew is a device array: T doubles come
  // back once per call (one synchronisation per adjust); the CSR by row goes up through the context's table scratch.
  XH_REQUIRE(method == 1 || (nq <= 32 && T < ((int64_t)1 << 31)), XH_ERR_LIMIT, "xh_plane_nearest: at most 32 nodes per group");
  if (nq <= 32 && T < ((int64_t)1 << 31) && (method == 0 || !xh_diag_env("XH_PLANE_NOROWS"))) {
    std::vector<double> g((size_t)T);
    XH_CHECK_HIP(hipMemcpyAsync(g.data(), gnew, 8 * (size_t)T, hipMemcpyDeviceToHost, ctx->stream));
    XH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    bool integral = true;
    for (int64_t t = 0; t < T && integral; ++t) integral = g[t] >= 1.0 && g[t] <= (double)G && g[t] == floor(g[t]);
    XH_REQUIRE(integral || method == 1, XH_ERR_ARG, "xh_plane_nearest: the group coordinates must be integers in 1 .. G");
    if (integral) {
      std::vector<int32_t> off((size_t)G + 1, 0), steps((size_t)T);
      for (int64_t t = 0; t < T; ++t) off[(size_t)g[t]]++;
      for (int r = 1; r <= G; ++r) off[r] += off[r - 1];
      std::vector<int32_t> cur(off.begin(), off.end() - 1);
      for (int64_t t = 0; t < T; ++t) steps[(size_t)cur[(size_t)g[t] - 1]++] = (int32_t)t;
      size_t cursor = 0;
      void *d_off = nullptr, *d_steps = nullptr;
      rc = xh_scratch_upload(ctx, &cursor, off.data(), 4 * off.size(), &d_off);
      if (rc) return rc;
      rc = xh_scratch_upload(ctx, &cursor, steps.data(), 4 * steps.size(), &d_steps);
      if (rc) return rc;
      // rows in chunks whose queries fit the work list even if every one of them needs the walk; every workgroup of the row
      // kernel owns a stretch of the list (rows of its chunk x the longest row x 256 cells)
      int maxsteps = 1;
      for (int r = 1; r <= G; ++r) maxsteps = off[r] - off[r - 1] > maxsteps ? off[r] - off[r - 1] : maxsteps;
      const char* eab = xh_diag_env("XH_PLANE_ABL");  // diagnostics: 1 = row kernel alone, 2 = no work kernel
      const int abl = eab ? atoi(eab) : 0;
      int r0 = 1;
      while (r0 <= G) {
        // rows per launch: gy row chunks of rchunk rows each, gy * cblocks stretches of rchunk * maxsteps * 256 entries
        const unsigned long long per_row = (unsigned long long)maxsteps * XH_BLOCK * (unsigned long long)cblocks;
        const int64_t slots = (int64_t)((unsigned long long)wcap / per_row);
        XH_REQUIRE(slots >= 1 && cblocks <= 65536, XH_ERR_LIMIT,
                   "xh_plane_linear: the work list (%llu entries) is too small for one group row of this grid (%llu)",
                   (unsigned long long)wcap, per_row);
        int64_t gy = cdiv64((int64_t)ctx->num_cu * 8, cblocks);
        gy = gy < 1 ? 1 : gy;
        if (gy > slots) gy = slots;
        if (gy > G + 1 - r0) gy = G + 1 - r0;
        if (gy * cblocks > 65536) gy = 65536 / cblocks;
        int64_t rchunk = cdiv64((int64_t)(G + 1 - r0), gy);
        if (rchunk > slots / gy) rchunk = slots / gy;
        int r1 = r0 + (int)(rchunk * gy);
        if (r1 > G + 1) r1 = G + 1;
        const unsigned long long segcap = (unsigned long long)rchunk * (unsigned long long)maxsteps * XH_BLOCK;
        XH_CHECK_HIP(hipMemsetAsync(nwork, 0, 4 * (size_t)(gy * cblocks), ctx->stream));
#define XH_PLANE_ROWS(NQM, NEAR)                                                                                                  \
  hipLaunchKernelGGL((k_plane_rows<NQM, NEAR>), dim3((unsigned)cblocks, (unsigned)gy), dim3(XH_BLOCK), 0, ctx->stream, xnew, base, st, \
                     (const int32_t*)d_off, (const int32_t*)d_steps, tabs, kind, scen, scen_st, r0, r1, work, nwork, segcap, abl, extrap)
        if (method == 1) {
          if (nq <= 20) XH_PLANE_ROWS(20, false); else XH_PLANE_ROWS(32, false);
        } else {
          if (nq <= 20) XH_PLANE_ROWS(20, true); else XH_PLANE_ROWS(32, true);
        }
#undef XH_PLANE_ROWS
        if (!(abl & 3))
          hipLaunchKernelGGL(k_plane_work, dim3((unsigned)cblocks, (unsigned)gy), dim3(XH_BLOCK), 0, ctx->stream, xnew, base, st, gnew, tabs,
                             kind, scen, scen_st, work, nwork, segcap, method == 0 ? 1 : 0);
        r0 = r1;
      }
      XH_LAUNCH_CHECK();
      return XH_OK;
    }
    // fractional coordinates (month groupings), <= 20 nodes per group: the two rows around a step in registers (k_plane_pair),
    // the queries whose triangle is not decided by those two rows alone through the work list.  Steps in stretches whose
    // queries fit the list even if every one of them is listed.
    const int64_t pblocks = cdiv64(C, PP_BLOCK);
    if (method == 1 && nq <= 20 && pblocks <= 65536 && !xh_diag_env("XH_PLANE_NOPAIR")) {
      const char* eab = xh_diag_env("XH_PLANE_ABL");  // diagnostics: 1 = pair kernel alone, 2 = no work kernel
      const int abl = eab ? atoi(eab) : 0;
      const int64_t len_cap = (int64_t)((unsigned long long)wcap / ((unsigned long long)PP_BLOCK * (unsigned long long)pblocks));
      XH_REQUIRE(len_cap >= 1, XH_ERR_LIMIT, "xh_plane_linear: the work list is too small for one time step of this grid");
      int64_t t0 = 0;
      while (t0 < T) {
        int64_t len = T - t0 < len_cap ? T - t0 : len_cap;
        int64_t gy = cdiv64((int64_t)ctx->num_cu * 16, pblocks);
        if (gy > len / 64) gy = len / 64;   // (a chunk reloads its two rows at its start: keep chunks long)
        if (gy < 1) gy = 1;
        if (gy * pblocks > 65536) gy = 65536 / pblocks;
        int64_t chunk = cdiv64(len, gy);
        if (chunk * gy > len_cap) { chunk = len_cap / gy; len = chunk * gy; }
        const unsigned long long segcap = (unsigned long long)chunk * PP_BLOCK;
        XH_CHECK_HIP(hipMemsetAsync(nwork, 0, 4 * (size_t)(gy * pblocks), ctx->stream));
        hipLaunchKernelGGL((k_plane_pair<20>), dim3((unsigned)pblocks, (unsigned)gy), dim3(PP_BLOCK), 0, ctx->stream, xnew, base, st, gnew, t0,
                           t0 + len, tabs, kind, scen, scen_st, work, nwork, segcap, abl);
        if (!(abl & 3))
          hipLaunchKernelGGL(k_plane_work, dim3((unsigned)pblocks, (unsigned)gy), dim3(XH_BLOCK), 0, ctx->stream, xnew, base, st, gnew, tabs,
                             kind, scen, scen_st, work, nwork, segcap, 0);
        t0 += len;
      }
      XH_LAUNCH_CHECK();
      return XH_OK;
    }
  }
  int64_t gy = cdiv64((int64_t)ctx->num_cu * 16, cblocks);
  if (gy < 1) gy = 1;
  if (gy > T) gy = T;
  if (gy > 4096) gy = 4096;
  hipLaunchKernelGGL(k_plane_linear, dim3((unsigned)cblocks, (unsigned)gy), dim3(XH_BLOCK), 0, ctx->stream, xnew, base, T, st, gnew, tabs,
                     kind, scen, scen_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // namespace

extern "C" {

int xh_plane_linear(xh_ctx* ctx, const float* xnew, const float* base, int64_t T, int64_t C, int64_t st, const double* gnew,
                    const float* xq_all, const double* xq_common, const float* yq_all, int G, int nq, int kind, float* scen,
                    int64_t scen_st) {
  return plane_run(ctx, xnew, base, T, C, st, gnew, xq_all, xq_common, yq_all, G, nq, kind, scen, scen_st, 1, 0);
}

int xh_plane_nearest(xh_ctx* ctx, const float* xnew, const float* base, int64_t T, int64_t C, int64_t st, const double* gnew,
                     const float* xq_all, const double* xq_common, const float* yq_all, int G, int nq, int kind, int extrap, float* scen,
                     int64_t scen_st) {
  XH_REQUIRE(extrap == 0 || extrap == 1, XH_ERR_ARG, "xh_plane_nearest: extrap must be 0 (constant) or 1 (nan)");
  return plane_run(ctx, xnew, base, T, C, st, gnew, xq_all, xq_common, yq_all, G, nq, kind, scen, scen_st, 0, extrap);
}

}  // extern "C"
