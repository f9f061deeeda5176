// qdmrank.h — the class boundaries of QuantileDeltaMapping.adjust with interp = "nearest" as RANKS (shared by qdm2.hip, the
// one-year register kernel, and the streaming kernels of select4.hip; xsdba._adjustment.qdm_adjust -> utils.rank(pct=True)
// -> interp_on_quantiles(nearest): upstream xsdba, re-exported by /root/reference/src/xclim/sdba.py:10; parity unpinned,
// restated in oracle/sdba.py rank_pct / qdm_adjust).
//
// With nearest-node interpolation the factor of a sample depends only on which of <= nvn + 2 classes its percentage rank
//     pct = mx (r2 / (2 n) - mn) / (mx - mn),   r2 = 2 below + equal + 1  (doubled average rank),
//     mn = ((c0 + 1) / 2) / n  (rank of the minimum, c0 copies),  mx = ((2 n - cmax + 1) / 2) / n  (rank of the maximum)
// falls in: below the first node | nearest to node j | above the last node.  pct is non-decreasing in r2 and r2 is
// non-decreasing in the VALUE, so class boundary t is a cut VALUE: the smallest value whose r2 passes test t
//     t = 0: pct >= x_0 (not below the first node);  0 < t < nvn: pct > x_{t-1} / 2 + x_t / 2 (scipy's nearest bounds);
//     t = nvn: pct > x_{nvn-1}.
// qdm_min_r2 returns R = min { r2 in [1, 2 n] : test(pct(r2)) } (2 n + 1 when none passes) by an analytic guess + exact
// verification in the operation order of qdm.hip / numpy (fp64, the two divisions as exact quotients).  The value with
// that r2: in a column sorted ascending, position p = ceil((R - 2) / 2) lies in the run [a, b) of its value, whose r2 is
// a + b + 1; earlier runs fail (their r2 <= 2 a <= 2 p < R), so the cut is that value if a + b + 1 >= R, else the value
// at position b (its run has r2 >= 2 b + 2 > R).
#pragma once
#include "common.h"

__device__ __forceinline__ uint32_t qdm_min_r2(bool first, double thr, uint32_t nn, uint32_t c0, uint32_t cmax) {
  const double dn = (double)nn;
  const double mn = ((double)(c0 + 1u) / 2.0) / dn;
  const double mx = ((double)(2u * nn - cmax + 1u) / 2.0) / dn;
  const double mxmn = mx - mn;
  const double inv_dn = 1.0 / dn, inv_mxmn = 1.0 / mxmn;
  const double slope = 2.0 * dn * (mxmn / mx);  // d r2 / d pct
  // pct ~ mx (r2 / 2n - mn) / mxmn  =>  r2 ~ 2n mn + thr slope; then exactly: down while R - 1 passes, up while R fails
  const double g = 2.0 * dn * mn + thr * slope;
  uint32_t R = g < 1.0 ? 1u : (g > 2.0 * dn ? 2u * nn : (uint32_t)g);
  bool down = true;
  for (;;) {
    const uint32_t probe = down ? R - 1u : R;
    const bool valid = down ? R > 1u : R <= 2u * nn;
    bool pass = false;
    if (valid) {
      const double rnk = xh_div_int((double)probe * 0.5, dn, inv_dn);
      const double p = xh_div_int(mx * (rnk - mn), mxmn, inv_mxmn);
      pass = first ? !(p < thr) : p > thr;
    }
    if (down) {
      if (pass) --R; else down = false;
    } else {
      if (!valid || pass) break;
      ++R;
    }
  }
  return R;
}

// position (0-based, in the ascending column) whose run decides boundary R
__device__ __forceinline__ uint32_t qdm_pos_of_r2(uint32_t R) { return R <= 2u ? 0u : (R - 1u) >> 1; }
