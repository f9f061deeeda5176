// qdm3.hip — QuantileDeltaMapping.adjust for series LONGER than 32768 steps (xsdba._adjustment.qdm_adjust; SURVEY.md 8f
// rank 4; group = "time"; reference call site /root/reference/src/xclim/sdba.py:10; PARITY UNPINNED like qdm.hip).
//
// The exact-rank kernel of qdm.hip keeps a column's keys in registers and an LDS list (144 KB at 32768 steps): a
// 1950-2100 daily series (55 152 steps) does not fit.  Here the (key, time index) pairs of every column are SORTED in
// global memory — rocPRIM's segmented radix sort, a library sort is the plain-library part of this path — and one
// workgroup per column turns sorted positions into average ranks:
//   sorted position p, tie run [a, b) of its key  ->  below = a, equal = b - a, doubled rank r2 = 2 below + equal + 1
// and from there the SAME fp64 sequence as k_qdm_columns (qdm.hip phase E): percentage rank, node search, nearest /
// linear factor, correction; the result goes to the sample's own time index.  Not tuned: five passes over the batch
// (keys, sort x ~2, ranks with scattered stores) — this is the "any length" path, the one-year and 30-year shapes never
// come here.
#include <string.h>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace {

constexpr int Q3_MAXQ = 64;
constexpr int Q3_NT = 256;
constexpr uint32_t Q3_NANKEY = 0xFFFFFFFFu;

__global__ void __launch_bounds__(XH_BLOCK)
k_q3_keys(const float* __restrict__ x, int64_t T, int64_t ncols, int64_t col_stride, uint32_t* __restrict__ keys,
          uint32_t* __restrict__ idx, uint32_t* __restrict__ offs) {
  const int64_t i = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (i <= ncols) offs[i] = (uint32_t)(i * T);
  if (i >= ncols * T) return;
  const int64_t c = i / T, t = i - c * T;
  keys[i] = xh_f2key(x[c * col_stride + t] + 0.0f);  // -0.0 + 0.0 = +0.0: the two zeros tie, as they do in rankdata
  idx[i] = (uint32_t)t;
}

// first position in ks[0, n) whose key is >= k (lower) / > k (upper)
__device__ __forceinline__ uint32_t q3_lower(const uint32_t* __restrict__ ks, uint32_t n, uint32_t k) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (ks[mid] < k) lo = mid + 1; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ uint32_t q3_upper(const uint32_t* __restrict__ ks, uint32_t n, uint32_t k) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (ks[mid] <= k) lo = mid + 1; else hi = mid;
  }
  return lo;
}

__global__ void __launch_bounds__(Q3_NT)
k_q3_ranks(const float* __restrict__ x, int64_t T, int64_t ncols, int64_t col_stride, const uint32_t* __restrict__ keys,
           const uint32_t* __restrict__ idx, const float* __restrict__ af, int64_t af_qstride, const double* __restrict__ qnodes,
           int nq, int kind, int interp, int extrap, float* __restrict__ out, int64_t out_cstride) {
  __shared__ double xs[Q3_MAXQ], xb[Q3_MAXQ];  // compacted quantile nodes of the column's non-NaN factors; scipy's nearest bounds
  __shared__ float ys[Q3_MAXQ];
  __shared__ uint32_t s_n, s_c0, s_cm;
  __shared__ int s_nv;
  const int gt = threadIdx.x, lane = gt & 63, w = gt >> 6;
  for (int64_t col = blockIdx.x; col < ncols; col += gridDim.x) {
    const uint32_t* __restrict__ ks = keys + col * T;
    const uint32_t* __restrict__ ix = idx + col * T;
    const float* __restrict__ xc = x + col * col_stride;
    float* __restrict__ oc = out + col * out_cstride;
    if (w == 0) {  // nodes of this column: drop the NaN factors (interp_on_quantiles masks them)
      const bool have = lane < nq;
      const float a = have ? af[(int64_t)lane * af_qstride + col] : 0.f;
      const bool ok = have && (a == a);
      const unsigned long long m = __ballot(ok);
      const int pos = __popcll(m & ((1ull << lane) - 1ull));
      if (ok) { xs[pos] = qnodes[lane]; ys[pos] = a; }
      if (lane == 0) s_nv = __popcll(m);
    }
    if (gt == 64) {  // valid samples (the NaN key sorts last), copies of the minimum and of the maximum
      const uint32_t n = q3_lower(ks, (uint32_t)T, Q3_NANKEY);
      s_n = n;
      s_c0 = n > 0u ? q3_upper(ks, n, ks[0]) : 0u;
      s_cm = n > 0u ? n - q3_lower(ks, n, ks[n - 1u]) : 0u;
    }
    __syncthreads();
    const int nvn = s_nv;
    if (gt + 1 < nvn) xb[gt] = xs[gt] / 2.0 + xs[gt + 1] / 2.0;
    __syncthreads();
    const uint32_t n = s_n, cnt0 = s_c0, cntm = s_cm;
    const double dn = (double)n;
    const double mn = ((double)(cnt0 + 1u) / 2.0) / dn;           // rank of the minimum / count
    const double mx = ((double)(2u * n - cntm + 1u) / 2.0) / dn;  // rank of the maximum / count
    const double mxmn = mx - mn;
    const double inv_dn = 1.0 / dn, inv_mxmn = 1.0 / mxmn;
    for (uint32_t p = (uint32_t)gt; p < (uint32_t)T; p += Q3_NT) {
      const uint32_t t = ix[p];
      float res = xh_nan32();
      if (p < n && nvn >= 2) {
        const uint32_t kk = ks[p];
        uint32_t below = p, equal = 1u;
        const bool tl = p > 0u && ks[p - 1u] == kk, tr = p + 1u < n && ks[p + 1u] == kk;
        if (tl || tr) {  // a tie run: its ends by binary search
          below = tl ? q3_lower(ks, n, kk) : p;
          equal = (tr ? q3_upper(ks, n, kk) : p + 1u) - below;
        }
        const double rnk = xh_div_int((double)(2u * below + equal + 1u) * 0.5, dn, inv_dn);
        const double pnum = mx * (rnk - mn);
        const double pct = mxmn == 0.0 ? xh_nan64() : xh_div_int(pnum, mxmn, inv_mxmn);  // 0 / 0 = NaN: all valid samples equal
        if (pct == pct) {
          const double x0 = xs[0], xl = xs[nvn - 1];
          float a;
          if (pct < x0) a = extrap == 0 ? ys[0] : xh_nan32();
          else if (pct > xl) a = extrap == 0 ? ys[nvn - 1] : xh_nan32();
          else if (interp == 0) {  // searchsorted(x_bds, pct, side="left"), clipped
            int lo = 0, hi = nvn - 1;
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (xb[mid] < pct) lo = mid + 1; else hi = mid;
            }
            a = ys[lo];
          } else {  // searchsorted(x, pct, side="left") clipped to [1, nv - 1]
            int lo = 0, hi = nvn;
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (xs[mid] < pct) lo = mid + 1; else hi = mid;
            }
            lo = lo < 1 ? 1 : (lo > nvn - 1 ? nvn - 1 : lo);
            const float ylo = ys[lo - 1], yhi = ys[lo];
            const double slope = (double)(yhi - ylo) / (xs[lo] - xs[lo - 1]);
            a = (float)(slope * (pct - xs[lo - 1]) + (double)ylo);
          }
          const float raw = xc[t];
          res = kind == 0 ? raw + a : (kind == 1 ? raw * a : a);
        }
      }
      oc[t] = res;
    }
    __syncthreads();  // the nodes are rewritten by the next column
  }
}

// ---- adapt_freq (xsdba.processing.adapt_freq -> _processing._adapt_freq) ------------------------------------------------
// sim_ad = sim.where(dP0 < 0, sim.where((rank < P0_ref) | (rank > P0_sim) | sim.isnull(), (pth - thresh) * U + thresh))
// with rank = sim.rank(dim, pct=True) (average ranks / valid count) from the sorted pairs, per column the float64 P0_ref,
// P0_sim, dP0 and the float32 pth.  U replaces upstream's np.random.random_sample (not reproducible across runs there
// either): a counter-based uniform in [0, 1) keyed by (seed, global time index, global cell), 53 bits, restated bit for
// bit in oracle/sdba.py.  tindex: global time index of every row of the columns (a group's rows), nullptr = row number.
__device__ __forceinline__ uint64_t q3_mix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ULL;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBULL;
  z ^= z >> 31;
  return z;
}

__global__ void __launch_bounds__(Q3_NT)
k_q3_adapt(const float* __restrict__ x, int64_t T, int64_t ncols, int64_t col_stride, const uint32_t* __restrict__ keys,
           const uint32_t* __restrict__ idx, const double* __restrict__ p0_ref, const double* __restrict__ p0_sim,
           const double* __restrict__ dp0, const float* __restrict__ pth, double thresh, uint64_t seed,
           const int64_t* __restrict__ tindex, int64_t cell0, float* __restrict__ out, int64_t out_cstride) {
  __shared__ uint32_t s_n;
  const int gt = threadIdx.x;
  for (int64_t col = blockIdx.x; col < ncols; col += gridDim.x) {
    const uint32_t* __restrict__ ks = keys + col * T;
    const uint32_t* __restrict__ ix = idx + col * T;
    const float* __restrict__ xc = x + col * col_stride;
    float* __restrict__ oc = out + col * out_cstride;
    if (gt == 0) s_n = q3_lower(ks, (uint32_t)T, Q3_NANKEY);
    __syncthreads();
    const uint32_t n = s_n;
    const double dn = (double)n, inv_dn = 1.0 / dn;
    const double pr = p0_ref[col], ps = p0_sim[col], d = dp0[col];
    const double span = (double)pth[col] - thresh;
    const uint64_t cell = (uint64_t)(cell0 + col);
    for (uint32_t p = (uint32_t)gt; p < (uint32_t)T; p += Q3_NT) {
      const uint32_t t = ix[p];
      const float raw = xc[t];
      float res = raw;  // NaN samples (p >= n) and everything that keeps its value
      if (p < n && !(d < 0.0)) {
        const uint32_t kk = ks[p];
        uint32_t below = p, equal = 1u;
        const bool tl = p > 0u && ks[p - 1u] == kk, tr = p + 1u < n && ks[p + 1u] == kk;
        if (tl || tr) {
          below = tl ? q3_lower(ks, n, kk) : p;
          equal = (tr ? q3_upper(ks, n, kk) : p + 1u) - below;
        }
        const double rnk = xh_div_int((double)(2u * below + equal + 1u) * 0.5, dn, inv_dn);
        if (!((rnk < pr) || (rnk > ps))) {
          const uint64_t tg = (uint64_t)(tindex ? tindex[t] : (int64_t)t);
          const uint64_t z = q3_mix64(seed * 0xD1342543DE82EF95ULL + tg * 0x9E3779B97F4A7C15ULL + cell * 0xC2B2AE3D27D4EB4FULL);
          const double u = (double)(z >> 11) * (1.0 / 9007199254740992.0);
          res = (float)(span * u + thresh);
        }
      }
      oc[t] = res;
    }
    __syncthreads();
  }
}

size_t q3_al(size_t b) { return (b + 255) & ~(size_t)255; }

}  // namespace

// bytes of workspace for `ncols` columns of T steps (ncols * T < 2^31)
int xh_qdm_sorted_ws(int64_t T, int64_t ncols, size_t* bytes) {
  XH_REQUIRE(T >= 1 && ncols >= 1 && ncols * T < (1ll << 31), XH_ERR_LIMIT, "xh_qdm_sorted_ws: batch of %lld x %lld samples too large",
             (long long)ncols, (long long)T);
  size_t tmp = 0;
  const uint32_t* k = nullptr;
  uint32_t* ko = nullptr;
  const unsigned* o = nullptr;
  hipError_t e = rocprim::segmented_radix_sort_pairs(nullptr, tmp, k, ko, k, ko, (unsigned)(ncols * T), (unsigned)ncols, o, o + 1, 0, 32,
                                                     (hipStream_t)0);
  if (e != hipSuccess) {
    xh_set_error("xh_qdm_sorted_ws: rocprim size query failed: %s", hipGetErrorString(e));
    return XH_ERR_HIP;
  }
  const size_t n = (size_t)(ncols * T);
  *bytes = 4 * q3_al(4 * n) + q3_al(4 * (size_t)(ncols + 1)) + q3_al(tmp);
  return XH_OK;
}

// exact-rank QDM on time-minor columns of any length through a global sort; `ws`: xh_qdm_sorted_ws(T, ncols) bytes
int xh_qdm_sorted(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride, const float* af, int64_t af_qs,
                  const double* d_q, int nq, int kind, int interp, int extrap, float* out, int64_t out_cs, void* ws) {
  XH_REQUIRE(ncols * T < (1ll << 31) && nq <= Q3_MAXQ, XH_ERR_LIMIT, "xh_qdm_sorted: batch of %lld x %lld samples too large",
             (long long)ncols, (long long)T);
  if (ncols <= 0) return XH_OK;
  const size_t n = (size_t)(ncols * T);
  char* p = (char*)ws;
  uint32_t* kin = (uint32_t*)p; p += q3_al(4 * n);
  uint32_t* kout = (uint32_t*)p; p += q3_al(4 * n);
  uint32_t* iin = (uint32_t*)p; p += q3_al(4 * n);
  uint32_t* iout = (uint32_t*)p; p += q3_al(4 * n);
  uint32_t* offs = (uint32_t*)p; p += q3_al(4 * (size_t)(ncols + 1));
  size_t tmp = 0;
  XH_CHECK_HIP(rocprim::segmented_radix_sort_pairs(nullptr, tmp, kin, kout, iin, iout, (unsigned)n, (unsigned)ncols, offs, offs + 1, 0, 32,
                                                   ctx->stream));
  hipLaunchKernelGGL(k_q3_keys, dim3((unsigned)cdiv64((int64_t)n + 1, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, xcols, T, ncols,
                     col_stride, kin, iin, offs);
  XH_LAUNCH_CHECK();
  XH_CHECK_HIP(rocprim::segmented_radix_sort_pairs((void*)p, tmp, kin, kout, iin, iout, (unsigned)n, (unsigned)ncols, offs, offs + 1, 0, 32,
                                                   ctx->stream));
  int64_t nblk = ncols;
  if (nblk > (int64_t)ctx->num_cu * 8) nblk = (int64_t)ctx->num_cu * 8;
  hipLaunchKernelGGL(k_q3_ranks, dim3((unsigned)nblk), dim3(Q3_NT), 0, ctx->stream, xcols, T, ncols, col_stride, kout, iout, af, af_qs,
                     d_q, nq, kind, interp, extrap, out, out_cs);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

extern "C" {

// The value-replacement step of xsdba.processing.adapt_freq on sim (T, C), element strides (st, sc), one of them 1; per
// cell: p0_ref, p0_sim, dp0 (float64) and pth (float32), device arrays of C; scen in the layout of sim.  The counts
// behind P0 and the quantile behind pth are the caller's (xh_threshold_count, xh_quantile_cells: xclim_amd/sdba.py).
int xh_adapt_freq(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, int64_t sc, const double* p0_ref,
                  const double* p0_sim, const double* dp0, const float* pth, double thresh, uint64_t seed, const int64_t* tindex,
                  int64_t cell0, float* scen) {
  XH_REQUIRE(ctx && sim && p0_ref && p0_sim && dp0 && pth && scen, XH_ERR_ARG, "xh_adapt_freq: NULL argument");
  XH_REQUIRE(T >= 1 && T < (1ll << 27) && C >= 0, XH_ERR_ARG, "xh_adapt_freq: bad shape (1 <= T < 2^27)");
  if (C == 0) return XH_OK;
  auto run = [&](const float* cols, int64_t n, int64_t cs, int64_t c0, float* o, int64_t ocs, void* ws) -> int {
    XH_REQUIRE(n * T < (1ll << 31), XH_ERR_LIMIT, "xh_adapt_freq: batch of %lld x %lld samples too large", (long long)n, (long long)T);
    const size_t ne = (size_t)(n * T);
    char* p = (char*)ws;
    uint32_t* kin = (uint32_t*)p; p += q3_al(4 * ne);
    uint32_t* kout = (uint32_t*)p; p += q3_al(4 * ne);
    uint32_t* iin = (uint32_t*)p; p += q3_al(4 * ne);
    uint32_t* iout = (uint32_t*)p; p += q3_al(4 * ne);
    uint32_t* offs = (uint32_t*)p; p += q3_al(4 * (size_t)(n + 1));
    size_t tmp = 0;
    XH_CHECK_HIP(rocprim::segmented_radix_sort_pairs(nullptr, tmp, kin, kout, iin, iout, (unsigned)ne, (unsigned)n, offs, offs + 1, 0, 32,
                                                     ctx->stream));
    hipLaunchKernelGGL(k_q3_keys, dim3((unsigned)cdiv64((int64_t)ne + 1, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, cols, T, n, cs, kin,
                       iin, offs);
    XH_LAUNCH_CHECK();
    XH_CHECK_HIP(rocprim::segmented_radix_sort_pairs((void*)p, tmp, kin, kout, iin, iout, (unsigned)ne, (unsigned)n, offs, offs + 1, 0, 32,
                                                     ctx->stream));
    int64_t nblk = n < (int64_t)ctx->num_cu * 8 ? n : (int64_t)ctx->num_cu * 8;
    hipLaunchKernelGGL(k_q3_adapt, dim3((unsigned)nblk), dim3(Q3_NT), 0, ctx->stream, cols, T, n, cs, kout, iout, p0_ref + c0, p0_sim + c0,
                       dp0 + c0, pth + c0, thresh, seed, tindex, cell0 + c0, o, ocs);
    XH_LAUNCH_CHECK();
    return XH_OK;
  };
  const bool minor = st == 1 && sc >= T;
  XH_REQUIRE(minor || (sc == 1 && st >= C), XH_ERR_LAYOUT, "xh_adapt_freq: one of the strides must be 1 (st=%lld sc=%lld)",
             (long long)st, (long long)sc);
  const int64_t Tp = (T + 63) & ~(int64_t)63;
  int64_t batch = (1ll << 26) / Tp;  // 64 M samples per batch: ~1.3 GB of sort workspace
  batch = (batch / 128) * 128;
  if (batch < 128) batch = 128;
  // the sort and the rank kernel index a batch with 32-bit offsets: very long series get fewer than 128 columns per batch
  while (batch > 1 && batch * Tp >= (1ll << 31)) batch >>= 1;
  if (batch > C) batch = C;
  size_t wsb = 0;
  int rc = xh_qdm_sorted_ws(T, batch, &wsb);
  if (rc) return rc;
  const size_t bufb = minor ? 0 : 2 * sizeof(float) * (size_t)batch * (size_t)Tp;
  void* tmp = nullptr;
  rc = xh_big_scratch(ctx, bufb + wsb, &tmp);
  if (rc) return rc;
  float* bin = (float*)tmp;
  float* bout = bin + (size_t)batch * (size_t)Tp;
  void* ws = (char*)tmp + bufb;
  for (int64_t c0 = 0; c0 < C; c0 += batch) {
    const int64_t nb = C - c0 < batch ? C - c0 : batch;
    if (minor) {
      rc = run(sim + c0 * sc, nb, sc, c0, scen + c0 * sc, sc, ws);
      if (rc) return rc;
      continue;
    }
    rc = xh_transpose_f32(ctx, sim + c0, T, nb, st, bin, Tp);
    if (rc) return rc;
    rc = run(bin, nb, Tp, c0, bout, Tp, ws);
    if (rc) return rc;
    rc = xh_transpose_f32(ctx, bout, nb, T, Tp, scen + c0, st);
    if (rc) return rc;
  }
  return XH_OK;
}

}  // extern "C"
