// runacc.h — accumulator of run-length statistics shared by the run-length kernels (runlen.hip) and the fused
// spell kernels (window.hip).  Reference: rle_statistics (indices/run_length.py:275-335).
#pragma once
#include "common.h"

struct RunAcc {
  int mx, mn, sum, cnt;
  unsigned long long sumsq;  // exact: sum of squared run lengths <= T^2
};

__device__ __forceinline__ void acc_reset(RunAcc& a) {
  a.mx = 0; a.mn = 0x7FFFFFFF; a.sum = 0; a.cnt = 0; a.sumsq = 0ull;
}
__device__ __forceinline__ void acc_add(RunAcc& a, int len) {
  a.mx = len > a.mx ? len : a.mx;
  a.mn = len < a.mn ? len : a.mn;
  a.sum += len;
  a.cnt += 1;
  a.sumsq += (unsigned long long)((unsigned)len) * (unsigned)len;
}
// predicated form: len == 0 means "no run finished here".  SG (stat group, compile time) prunes the fields the
// requested statistic does not read: 0 all, 1 max, 2 sum / count / mean, 3 min.
template <int SG = 0>
__device__ __forceinline__ void acc_add_if(RunAcc& a, int len) {
  if (SG == 0 || SG == 1) a.mx = len > a.mx ? len : a.mx;
  if (SG == 0 || SG == 3) a.mn = (len > 0 && len < a.mn) ? len : a.mn;
  if (SG == 0 || SG == 2) a.sum += len;
  if (SG == 0 || SG == 2 || SG == 3 || SG == 1) a.cnt += len > 0 ? 1 : 0;
  if (SG == 0) a.sumsq += (unsigned long long)((unsigned)len) * (unsigned)len;
}
__device__ __forceinline__ float acc_result(const RunAcc& a, int stat, int plainsum) {
  if (stat == XH_RUN_PLAINSUM) return (float)plainsum;
  if (a.cnt == 0) return 0.0f;  // rl:326: no qualifying run -> 0
  switch (stat) {
    case XH_RUN_MAX: return (float)a.mx;
    case XH_RUN_MIN: return (float)a.mn;
    case XH_RUN_SUM: return (float)a.sum;
    case XH_RUN_COUNT: return (float)a.cnt;
    case XH_RUN_MEAN: return (float)((double)a.sum / (double)a.cnt);
    default: {  // population std (ddof = 0), tests/test_run_length.py:255-256
      double m = (double)a.sum / (double)a.cnt;
      double v = (double)a.sumsq / (double)a.cnt - m * m;
      return (float)sqrt(v > 0.0 ? v : 0.0);
    }
  }
}
