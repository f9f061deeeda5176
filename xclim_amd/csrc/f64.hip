// f64.hip — the float64 FIELD variants of the three entry points whose result depends on the input dtype in the
// reference: the compare of threshold_count happens in the data dtype (indices/generic.py:301-326, 360: float64 data
// against any threshold is a float64 compare), resample(...).<op>() returns the data dtype (gen:83-125), and
// _nan_quantile takes its `diff` in the data dtype (core/utils.py:486).  The float32 kernels elsewhere would have to
// round such a field first — counts next to a threshold can flip; these do not.  All three are
// HBM-bound marches like their float32 twins (16-byte double2 loads, two cells per lane).
#include <stdlib.h>

#include "common.h"

namespace {

template <int VEC>
struct VecD {
  double v[VEC];
};
template <int VEC>
__device__ __forceinline__ VecD<VEC> ld(const double* __restrict__ p) {
  VecD<VEC> r;
  if (VEC == 2) {
    const double2 t = *reinterpret_cast<const double2*>(p);
    r.v[0] = t.x;
    r.v[1 % VEC] = t.y;
  } else {
    r.v[0] = *p;
  }
  return r;
}

// rows [t0, t1) in double-buffered batches of 8 (the pattern of xh_march_rows)
template <int VEC, typename F>
__device__ __forceinline__ void march64(const double* __restrict__ p, int64_t st, int64_t t0, int64_t t1, F&& f) {
  constexpr int U = 8;
  int64_t t = t0;
  const int64_t nfull = (t1 - t0) / U;
  if (nfull > 0) {
    VecD<VEC> buf[U];
#pragma unroll
    for (int u = 0; u < U; ++u) buf[u] = ld<VEC>(p + (t + u) * st);
    for (int64_t b = 0; b < nfull; ++b) {
      VecD<VEC> nxt[U];
      const bool more = b + 1 < nfull;
      if (more) {
#pragma unroll
        for (int u = 0; u < U; ++u) nxt[u] = ld<VEC>(p + (t + U + u) * st);
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
  for (; t < t1; ++t) f(t, ld<VEC>(p + t * st));
}

// KIND: XH_THR_SCALAR_F64 | XH_THR_DOY_F64 | XH_THR_FULL_F64 (a float32 threshold is exactly representable: the host
// widens scalars; float32 tables are refused)
template <int VEC, int KIND>
__global__ void __launch_bounds__(XH_BLOCK)
k_threshold_count_f64(const double* __restrict__ x, int64_t C, int64_t st, int op, double thr, const double* __restrict__ table,
                      int64_t tstride, const int32_t* __restrict__ tidx, const int64_t* __restrict__ seg_off, int P,
                      int32_t* __restrict__ count_out, int32_t* __restrict__ valid_out) {
  const int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    const int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    int cnt[VEC], val[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) cnt[i] = 0, val[i] = 0;
    march64<VEC>(x + c, st, t0, t1, [&](int64_t t, const VecD<VEC>& xv) {
      double th[VEC];
      if (KIND == XH_THR_SCALAR_F64) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) th[i] = thr;
      } else {
        const int64_t row = KIND == XH_THR_DOY_F64 ? (int64_t)tidx[t] : t;
        const VecD<VEC> tv = ld<VEC>(table + row * tstride + c);
#pragma unroll
        for (int i = 0; i < VEC; ++i) th[i] = tv.v[i];
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        cnt[i] += xh_cmp_f64(xv.v[i], op, th[i]) ? 1 : 0;
        val[i] += xv.v[i] == xv.v[i] ? 1 : 0;
      }
    });
    const int64_t o = (int64_t)p * C + c;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      count_out[o + i] = cnt[i];
      if (valid_out) valid_out[o + i] = val[i];
    }
  }
}

// resample(time=freq).<op>() of a float64 field: float64 results (int32 for count / argmin / argmax), fp64 accumulation
// in time order, two passes for std / var like k_resample_reduce
template <int VEC, int RED>
__global__ void __launch_bounds__(XH_BLOCK)
k_resample_reduce_f64(const double* __restrict__ x, int64_t C, int64_t st, int skipna, const int64_t* __restrict__ seg_off, int P,
                      void* __restrict__ out_v, int32_t* __restrict__ valid_out) {
  const int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    const int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    double s1[VEC], s2[VEC], ext[VEC];
    int n[VEC], arg[VEC], nanseen[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) s1[i] = 0.0, s2[i] = 0.0, ext[i] = 0.0, n[i] = 0, arg[i] = -1, nanseen[i] = 0;
    march64<VEC>(x + c, st, t0, t1, [&](int64_t t, const VecD<VEC>& xv) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const double v = xv.v[i];
        if (v == v) {
          if (RED == XH_RED_SUM || RED == XH_RED_MEAN || RED == XH_RED_STD || RED == XH_RED_VAR) s1[i] += v;
          if (RED == XH_RED_MIN || RED == XH_RED_ARGMIN) {
            if (n[i] == 0 || v < ext[i]) ext[i] = v, arg[i] = (int)(t - t0);
          }
          if (RED == XH_RED_MAX || RED == XH_RED_ARGMAX) {
            if (n[i] == 0 || v > ext[i]) ext[i] = v, arg[i] = (int)(t - t0);
          }
          n[i]++;
        } else {
          if (!nanseen[i] && !skipna && (RED == XH_RED_ARGMIN || RED == XH_RED_ARGMAX)) arg[i] = (int)(t - t0);
          nanseen[i] = 1;
        }
      }
    });
    if (RED == XH_RED_STD || RED == XH_RED_VAR) {
      double mean[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) mean[i] = n[i] > 0 ? s1[i] / (double)n[i] : 0.0;
      march64<VEC>(x + c, st, t0, t1, [&](int64_t, const VecD<VEC>& xv) {
#pragma unroll
        for (int i = 0; i < VEC; ++i)
          if (xv.v[i] == xv.v[i]) {
            const double d = xv.v[i] - mean[i];
            s2[i] += d * d;
          }
      });
    }
    const int64_t o = (int64_t)p * C + c;
    const int len = (int)(t1 - t0);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const bool poisoned = !skipna && nanseen[i];
      if (RED == XH_RED_COUNT) reinterpret_cast<int32_t*>(out_v)[o + i] = n[i];
      else if (RED == XH_RED_ARGMIN || RED == XH_RED_ARGMAX) reinterpret_cast<int32_t*>(out_v)[o + i] = arg[i];
      else {
        double r;
        if (RED == XH_RED_SUM) r = poisoned ? xh_nan64() : s1[i];
        else if (RED == XH_RED_MEAN) r = (poisoned || n[i] == 0) ? xh_nan64() : s1[i] / (double)n[i];
        else if (RED == XH_RED_MIN || RED == XH_RED_MAX) r = (poisoned || n[i] == 0) ? xh_nan64() : ext[i];
        else if (RED == XH_RED_VAR) r = (poisoned || n[i] == 0) ? xh_nan64() : s2[i] / (double)n[i];
        else r = (poisoned || n[i] == 0) ? xh_nan64() : sqrt(s2[i] / (double)n[i]);
        if (len == 0 && RED != XH_RED_SUM) r = xh_nan64();
        reinterpret_cast<double*>(out_v)[o + i] = r;
      }
      if (valid_out) valid_out[o + i] = n[i];
    }
  }
}

// ---- _nan_quantile on float64 samples (utl:494-557): one wave per slice, the N samples sorted as order-preserving
// 64-bit keys in LDS (bitonic, NaN = largest key), then the Hyndman-Fan lerp with `diff` in float64 (the data dtype)
__device__ __forceinline__ uint64_t d2key(double d) {
  const uint64_t u = (uint64_t)__double_as_longlong(d);
  if (d != d) return ~0ull;
  return (u >> 63) ? ~u : (u | (1ull << 63));
}
__device__ __forceinline__ double key2d(uint64_t k) {
  if (k == ~0ull) return xh_nan64();
  const uint64_t u = (k >> 63) ? (k & ~(1ull << 63)) : ~k;
  return __longlong_as_double((long long)u);
}

__global__ void __launch_bounds__(64)
k_nan_quantile_f64(const double* __restrict__ x, int N, int NP, int64_t C, int64_t sn, int64_t sc, const double* __restrict__ qs,
                   int nq, double alpha, double beta, double* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* key = reinterpret_cast<uint64_t*>(smem);
  const int lane = threadIdx.x;
  for (int64_t c = blockIdx.x; c < C; c += gridDim.x) {
    int nv = 0;
    for (int i = lane; i < NP; i += 64) {
      const uint64_t k = i < N ? d2key(x[(int64_t)i * sn + c * sc]) : ~0ull;
      nv += k != ~0ull ? 1 : 0;
      key[i] = k;
    }
    for (int m = 32; m > 0; m >>= 1) nv += __shfl_xor(nv, m);
    __syncthreads();
    for (int k = 2; k <= NP; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int i = lane; i < NP; i += 64) {
          const int l = i ^ j;
          if (l > i) {
            const bool up = (i & k) == 0;
            const uint64_t a = key[i], b = key[l];
            if ((a > b) == up) {
              key[i] = b;
              key[l] = a;
            }
          }
        }
        __syncthreads();
      }
    }
    for (int j = lane; j < nq; j += 64) {
      const double q = qs[j];
      double r;
      if (N == 1) r = key2d(key[0]);  // utl:508-510
      else if (nv < 2) r = nv == 1 ? key2d(key[0]) : xh_nan64();
      else {
        const double nn = (double)nv;
        const double vi = nn * q + (alpha + q * (1.0 - alpha - beta)) - 1.0;  // utl:395
        if (vi >= nn - 1.0) r = key2d(key[nv - 1]);
        else if (vi < 0.0) r = key2d(key[0]);
        else {
          const double prev = floor(vi);
          const int ip = (int)prev;
          const double gamma = vi - prev;
          const double left = key2d(key[ip]), right = key2d(key[ip + 1]);
          const double diff = right - left;  // utl:486 in the data dtype: float64 here
          r = left + diff * gamma;
          if (gamma >= 0.5) r = right - diff * (1.0 - gamma);  // utl:488
          if (r != r) r = key2d(key[nv - 1]);                  // utl:552-554
        }
      }
      out[(int64_t)j * C + c] = r;
    }
    __syncthreads();
  }
}

int check_f64(const char* fn, xh_ctx* ctx, const void* x, int64_t T, int64_t C, int64_t st, int64_t sc) {
  XH_REQUIRE(ctx && x, XH_ERR_ARG, "%s: NULL argument", fn);
  XH_REQUIRE(T >= 0 && C >= 0, XH_ERR_ARG, "%s: negative shape", fn);
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT, "%s: streaming kernels need a time-major view (sc == 1, st >= C); got st=%lld sc=%lld",
             fn, (long long)st, (long long)sc);
  return XH_OK;
}

int upload_seg(xh_ctx* ctx, size_t* cur, const int64_t* seg_off, int P, int64_t T, const char* fn, const int64_t** d_seg) {
  XH_REQUIRE(seg_off && P >= 1, XH_ERR_ARG, "%s: seg_off NULL or P < 1", fn);
  for (int p = 0; p < P; ++p)
    XH_REQUIRE(seg_off[p] <= seg_off[p + 1] && seg_off[p] >= 0 && seg_off[p + 1] <= T, XH_ERR_ARG,
               "%s: seg_off must be non-decreasing within [0, T]", fn);
  void* d = nullptr;
  const int rc = xh_scratch_upload(ctx, cur, seg_off, sizeof(int64_t) * (size_t)(P + 1), &d);
  if (rc) return rc;
  *d_seg = (const int64_t*)d;
  return XH_OK;
}

inline int pick_vec64(const void* p, int64_t C, int64_t st) {
  return ((reinterpret_cast<uintptr_t>(p) & 15) == 0 && (C % 2) == 0 && (st % 2) == 0) ? 2 : 1;
}

}  // namespace

extern "C" {

int xh_threshold_count_f64(xh_ctx* ctx, const double* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op, int thr_kind,
                           double thr_scalar, const double* thr_table, int64_t thr_stride, const int32_t* tidx,
                           const int64_t* seg_off, int P, int32_t* count_out, int32_t* valid_out) {
  int rc = check_f64("xh_threshold_count_f64", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(op >= XH_OP_GT && op <= XH_OP_NE, XH_ERR_OP, "Operation `%d` not recognized.", op);
  XH_REQUIRE(count_out, XH_ERR_ARG, "xh_threshold_count_f64: count_out is NULL");
  XH_REQUIRE(thr_kind == XH_THR_SCALAR_F64 || thr_kind == XH_THR_DOY_F64 || thr_kind == XH_THR_FULL_F64, XH_ERR_ARG,
             "xh_threshold_count_f64: thr_kind must be XH_THR_SCALAR_F64, XH_THR_DOY_F64 or XH_THR_FULL_F64 (got %d)", thr_kind);
  if (thr_kind != XH_THR_SCALAR_F64) {
    XH_REQUIRE(thr_table && thr_stride >= C, XH_ERR_ARG, "xh_threshold_count_f64: threshold table missing or stride < C");
    if (thr_kind == XH_THR_DOY_F64) XH_REQUIRE(tidx, XH_ERR_ARG, "xh_threshold_count_f64: tidx required for per-doy thresholds");
  }
  size_t cur = 0;
  const int64_t* d_seg = nullptr;
  rc = upload_seg(ctx, &cur, seg_off, P, T, "xh_threshold_count_f64", &d_seg);
  if (rc) return rc;
  if (C == 0) return XH_OK;
  int vec = pick_vec64(x, C, st);
  if (thr_kind != XH_THR_SCALAR_F64 && ((reinterpret_cast<uintptr_t>(thr_table) & 15) != 0 || (thr_stride % 2) != 0)) vec = 1;
  const dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), (unsigned)(P < 1 ? 1 : (P > 4096 ? 4096 : P)));
#define XH_TC64(V, K)                                                                                                       \
  hipLaunchKernelGGL((k_threshold_count_f64<V, K>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, op, thr_scalar, thr_table, \
                     thr_stride, tidx, d_seg, P, count_out, valid_out)
  if (vec == 2) {
    if (thr_kind == XH_THR_SCALAR_F64) XH_TC64(2, XH_THR_SCALAR_F64);
    else if (thr_kind == XH_THR_DOY_F64) XH_TC64(2, XH_THR_DOY_F64);
    else XH_TC64(2, XH_THR_FULL_F64);
  } else {
    if (thr_kind == XH_THR_SCALAR_F64) XH_TC64(1, XH_THR_SCALAR_F64);
    else if (thr_kind == XH_THR_DOY_F64) XH_TC64(1, XH_THR_DOY_F64);
    else XH_TC64(1, XH_THR_FULL_F64);
  }
#undef XH_TC64
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_resample_reduce_f64(xh_ctx* ctx, const double* x, int64_t T, int64_t C, int64_t st, int64_t sc, int reducer, int skipna,
                           const int64_t* seg_off, int P, void* out, int32_t* valid_out) {
  int rc = check_f64("xh_resample_reduce_f64", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(out, XH_ERR_ARG, "xh_resample_reduce_f64: out is NULL");
  size_t cur = 0;
  const int64_t* d_seg = nullptr;
  rc = upload_seg(ctx, &cur, seg_off, P, T, "xh_resample_reduce_f64", &d_seg);
  if (rc) return rc;
  if (C == 0) return XH_OK;
  const int vec = pick_vec64(x, C, st);
  const dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), (unsigned)(P < 1 ? 1 : (P > 4096 ? 4096 : P)));
#define XH_RR64(R)                                                                                                            \
  case R:                                                                                                                     \
    if (vec == 2) hipLaunchKernelGGL((k_resample_reduce_f64<2, R>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, skipna, d_seg, P, out, valid_out); \
    else hipLaunchKernelGGL((k_resample_reduce_f64<1, R>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, skipna, d_seg, P, out, valid_out);          \
    break;
  switch (reducer) {
    XH_RR64(XH_RED_SUM) XH_RR64(XH_RED_MEAN) XH_RR64(XH_RED_MIN) XH_RR64(XH_RED_MAX) XH_RR64(XH_RED_STD) XH_RR64(XH_RED_VAR)
    XH_RR64(XH_RED_COUNT) XH_RR64(XH_RED_ARGMIN) XH_RR64(XH_RED_ARGMAX)
    default:
      xh_set_error("xh_resample_reduce_f64: reducer %d not recognized", reducer);
      return XH_ERR_OP;
  }
#undef XH_RR64
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_nan_quantile_f64(xh_ctx* ctx, const double* x, int64_t N, int64_t C, int64_t sn, int64_t sc, const double* q, int nq,
                        double alpha, double beta, double* out) {
  XH_REQUIRE(ctx && x && q && out, XH_ERR_ARG, "xh_nan_quantile_f64: NULL argument");
  XH_REQUIRE(N >= 1 && C >= 0 && nq >= 1 && nq <= 64, XH_ERR_ARG, "xh_nan_quantile_f64: bad shape (N >= 1, 1 <= nq <= 64)");
  XH_REQUIRE(N <= 4096, XH_ERR_LIMIT, "xh_nan_quantile_f64: N = %lld samples exceed 4096", (long long)N);
  XH_REQUIRE((sc == 1 && sn >= C) || (sn == 1 && sc >= N), XH_ERR_LAYOUT, "xh_nan_quantile_f64: one of the two strides must be 1");
  if (C == 0) return XH_OK;
  size_t cur = 0;
  void* d_q = nullptr;
  const int rc = xh_scratch_upload(ctx, &cur, q, sizeof(double) * nq, &d_q);
  if (rc) return rc;
  int NP = 2;
  while (NP < N) NP <<= 1;
  int64_t nblk = C;
  if (nblk > (int64_t)ctx->num_cu * 64) nblk = (int64_t)ctx->num_cu * 64;
  hipLaunchKernelGGL(k_nan_quantile_f64, dim3((unsigned)nblk), dim3(64), (size_t)NP * 8, ctx->stream, x, (int)N, NP, C, sn, sc,
                     (const double*)d_q, nq, alpha, beta, out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // extern "C"
