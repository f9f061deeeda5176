// eqm.hip — sdba empirical quantile mapping: full-series quantiles per cell (train) and node search + correction
// (adjust).  The algorithm lives in the third-party package xsdba (>= 0.4.0; pyproject.toml:111 of the reference;
// only src/xclim/sdba.py:10 and tests/test_xsdba.py reference it).  Spec adopted (parity "unpinned"):
//   nbutils.quantile        : NaN-aware Hyndman-Fan type 7 (alpha = beta = 1, same formula as core/utils.py:395)
//   utils.get_correction    : af = ref_q - hist_q ("+")  |  ref_q / hist_q ("*")
//   utils.interp_on_quantiles (1-D, group="time"): scipy.interpolate.interp1d(hist_q, af, kind=nearest|linear,
//                             bounds_error=False, fill_value=(af[0], af[-1]) | nan) on the non-NaN nodes
//   utils.apply_correction  : scen = sim + af_t  |  sim * af_t
#include "common.h"

// ---- per-column sort + quantiles ------------------------------------------------------------------------
// Columns are contiguous in memory (time-minor).  WAVE_COLS: each of the 4 waves of a block sorts its own column
// (NP <= 4096 keys) in its LDS slice; otherwise the whole block sorts one column (NP <= 32768).  Bitonic network on
// order-preserving uint32 keys (NaN -> 0xFFFFFFFF sorts last, as numpy).
template <bool WAVE_COLS>
__global__ void __launch_bounds__(XH_BLOCK)
k_colsort_quantile(const float* __restrict__ x, int64_t T, int64_t ncols, int64_t col_stride, int NP,
                   const double* __restrict__ qs, int nq, float* __restrict__ out, int64_t out_cstride,
                   int64_t out_qstride) {
  extern __shared__ uint32_t lds[];
  __shared__ int s_nvalid[4];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int cols_per_block = WAVE_COLS ? 4 : 1;
  const int nthr = WAVE_COLS ? 64 : XH_BLOCK;   // threads cooperating on one column
  const int me = WAVE_COLS ? lane : tid;
  uint32_t* keys = lds + (WAVE_COLS ? (size_t)wave * NP : 0);

  for (int64_t cb = (int64_t)blockIdx.x * cols_per_block; cb < ncols; cb += (int64_t)gridDim.x * cols_per_block) {
    int64_t col = cb + (WAVE_COLS ? wave : 0);
    bool have = col < ncols;
    if (tid < 4) s_nvalid[tid] = 0;
    __syncthreads();
    int nv = 0;
    for (int i = me; i < NP; i += nthr) {
      uint32_t k = 0xFFFFFFFFu;
      if (have && i < T) {
        float v = x[col * col_stride + i];
        k = xh_f2key(v);
        nv += (v == v) ? 1 : 0;
      }
      keys[i] = k;
    }
    if (nv) atomicAdd(&s_nvalid[WAVE_COLS ? wave : 0], nv);
    __syncthreads();
    for (int size = 2; size <= NP; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int idx = me; idx < (NP >> 1); idx += nthr) {
          int i = 2 * idx - (idx & (stride - 1));
          int j = i + stride;
          bool up = ((i & size) == 0);
          uint32_t a = keys[i], b = keys[j];
          uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
          keys[i] = up ? lo : hi;
          keys[j] = up ? hi : lo;
        }
        __syncthreads();
      }
    }
    if (have && me < nq) {
      int n = s_nvalid[WAVE_COLS ? wave : 0];
      double q = qs[me];
      double r;
      // same Hyndman-Fan evaluation as quantile.hip::xh_hf_quantile with alpha = beta = 1
      if (T == 1) r = (double)xh_key2f(keys[0]);
      else if (n < 2) r = n == 1 ? (double)xh_key2f(keys[0]) : xh_nan64();
      else {
        double nn = (double)n;
        double vi = nn * q + (1.0 + q * (1.0 - 1.0 - 1.0)) - 1.0;
        if (vi >= nn - 1.0) r = (double)xh_key2f(keys[n - 1]);
        else if (vi < 0.0) r = (double)xh_key2f(keys[0]);
        else {
          double prev = floor(vi);
          int ip = (int)prev;
          double gamma = vi - prev;
          float left = xh_key2f(keys[ip]), right = xh_key2f(keys[ip + 1]);
          float diff = right - left;
          r = (double)left + (double)diff * gamma;
          if (gamma >= 0.5) r = (double)right - (double)diff * (1.0 - gamma);
          if (r != r) r = (double)xh_key2f(keys[n - 1]);
        }
      }
      out[col * out_cstride + (int64_t)me * out_qstride] = (float)r;
    }
    __syncthreads();
  }
}

// af from ref_q / hist_q  (get_correction)
__global__ void __launch_bounds__(XH_BLOCK)
k_correction(const float* __restrict__ ref_q, const float* __restrict__ hist_q, int64_t n, int kind,
             float* __restrict__ af) {
  int64_t i = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (i >= n) return;
  af[i] = kind == 0 ? (ref_q[i] - hist_q[i]) : (ref_q[i] / hist_q[i]);
}

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
#pragma unroll
  for (int j = 0; j < NQMAX; ++j) {
    float xj = xh_nan32(), yj = xh_nan32();
    if (j < nq) { xj = hq[(int64_t)j * C + c]; yj = af[(int64_t)j * C + c]; }
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
  int64_t chunk = cdiv64(T, (int64_t)gridDim.y);
  int64_t ta = (int64_t)blockIdx.y * chunk, tb = ta + chunk;
  if (tb > T) tb = T;
#pragma unroll 2
  for (int64_t t = ta; t < tb; ++t) {
    float x = sim[t * st + c];
    float a = xh_nan32();
    if (m >= 1 && x == x) {
      bool below = x < firstx, above = x > lastx;
      if (INTERP == 0) {
        a = firsty;
#pragma unroll
        for (int j = 0; j < NQMAX; ++j) a = (mid[j] < x) ? ny[j] : a;
      } else {
        float lox = firstx, loy = firsty, hix = firstx, hiy = firsty, px = 0.f, py = 0.f;
        bool haveprev = false, firstpair = true;
#pragma unroll
        for (int j = 0; j < NQMAX; ++j) {
          bool valid = nx[j] == nx[j];
          if (valid) {
            if (haveprev) {
              bool take = firstpair || (px < x);
              lox = take ? px : lox; loy = take ? py : loy;
              hix = take ? nx[j] : hix; hiy = take ? ny[j] : hiy;
              firstpair = false;
            }
            px = nx[j]; py = ny[j]; haveprev = true;
          }
        }
        float slope = (hiy - loy) / (hix - lox);
        a = slope * (x - lox) + loy;
        if (m < 2) a = xh_nan32();
      }
      if (below) a = extrap == 0 ? firsty : xh_nan32();
      if (above) a = extrap == 0 ? lasty : xh_nan32();
    }
    scen[t * scen_st + c] = kind == 0 ? (x + a) : (x * a);
  }
}

static int next_pow2_i(int64_t n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

// quantiles of all columns of a time-minor view
static int colsort_launch(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride,
                          const double* d_q, int nq, float* out, int64_t out_cstride, int64_t out_qstride) {
  int NP = next_pow2_i(T);
  if (NP < 64) NP = 64;
  XH_REQUIRE(NP <= 32768, XH_ERR_LIMIT, "quantile_series: T = %lld exceeds the 32768-sample LDS column limit", (long long)T);
  bool wave_cols = NP <= 4096;
  size_t lds = (size_t)NP * sizeof(uint32_t) * (wave_cols ? 4 : 1);
  int64_t nblk = wave_cols ? cdiv64(ncols, 4) : ncols;
  int64_t maxblk = (int64_t)ctx->num_cu * 8;
  if (nblk > maxblk) nblk = maxblk;
  if (wave_cols) {
    if (lds > 64 * 1024)
      XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_colsort_quantile<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds));
    hipLaunchKernelGGL((k_colsort_quantile<true>), dim3((unsigned)nblk), dim3(XH_BLOCK), lds, ctx->stream, xcols, T, ncols,
                       col_stride, NP, d_q, nq, out, out_cstride, out_qstride);
  } else {
    if (lds > 64 * 1024)
      XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_colsort_quantile<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds));
    hipLaunchKernelGGL((k_colsort_quantile<false>), dim3((unsigned)nblk), dim3(XH_BLOCK), lds, ctx->stream, xcols, T,
                       ncols, col_stride, NP, d_q, nq, out, out_cstride, out_qstride);
  }
  XH_LAUNCH_CHECK();
  return XH_OK;
}

static int quantile_series_impl(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc,
                                const double* d_q, int nq, float* out) {
  if (st == 1 && sc >= T) {
    return colsort_launch(ctx, x, T, C, sc, d_q, nq, out, 1, C);
  }
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT, "quantile_series: one of the strides must be 1 (st=%lld sc=%lld)",
             (long long)st, (long long)sc);
  // time-major: transpose batches of columns into scratch (not counted as algorithmic bytes)
  int64_t batch = (int64_t)((1ull << 30) / (sizeof(float) * (size_t)T));
  batch = (batch / 64) * 64;
  if (batch < 64) batch = 64;
  if (batch > C) batch = C;
  void* tmp = nullptr;
  int rc = xh_big_scratch(ctx, sizeof(float) * (size_t)batch * (size_t)T, &tmp);
  if (rc) return rc;
  for (int64_t c0 = 0; c0 < C; c0 += batch) {
    int64_t nb = C - c0 < batch ? C - c0 : batch;
    rc = xh_transpose_f32(ctx, x + c0, T, nb, st, (float*)tmp, T);
    if (rc) return rc;
    rc = colsort_launch(ctx, (const float*)tmp, T, nb, T, d_q, nq, out + c0, 1, C);
    if (rc) return rc;
  }
  return XH_OK;
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
  rc = quantile_series_impl(ctx, ref, T, C, st, sc, (const double*)d_q, nq, af);
  if (rc) return rc;
  rc = quantile_series_impl(ctx, hist, T, C, st, sc, (const double*)d_q, nq, hist_q);
  if (rc) return rc;
  int64_t n = (int64_t)nq * C;
  hipLaunchKernelGGL(k_correction, dim3((unsigned)cdiv64(n, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, af, hist_q, n, kind,
                     af);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_eqm_adjust(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, int64_t sc, const float* af,
                  const float* hist_q, int nq, int kind, int interp, int extrap, float* scen, int64_t scen_st) {
  XH_REQUIRE(ctx && sim && af && hist_q && scen, XH_ERR_ARG, "xh_eqm_adjust: NULL argument");
  XH_REQUIRE(T >= 0 && C >= 0 && nq >= 1 && nq <= 64, XH_ERR_ARG, "xh_eqm_adjust: bad shape (1 <= nq <= 64)");
  XH_REQUIRE(sc == 1 && st >= C && scen_st >= C, XH_ERR_LAYOUT, "xh_eqm_adjust: needs time-major views (sc == 1)");
  XH_REQUIRE(kind == 0 || kind == 1, XH_ERR_ARG, "xh_eqm_adjust: kind must be 0 (+) or 1 (*)");
  XH_REQUIRE(interp == 0 || interp == 1, XH_ERR_ARG, "xh_eqm_adjust: interp must be 0 (nearest) or 1 (linear)");
  XH_REQUIRE(extrap == 0 || extrap == 1, XH_ERR_ARG, "xh_eqm_adjust: extrap must be 0 (constant) or 1 (nan)");
  if (T == 0 || C == 0) return XH_OK;
  int64_t cblocks = cdiv64(C, XH_BLOCK);
  int64_t want = (int64_t)ctx->num_cu * 16;
  int64_t gy = cdiv64(want, cblocks);
  if (gy < 1) gy = 1;
  if (gy > T) gy = T;
  if (gy > 1024) gy = 1024;
  dim3 grid((unsigned)cblocks, (unsigned)gy);
#define XH_ADJ(NQM, IP)                                                                                              \
  hipLaunchKernelGGL((k_eqm_adjust<NQM, IP>), grid, dim3(XH_BLOCK), 0, ctx->stream, sim, T, C, st, af, hist_q, nq, kind, \
                     extrap, scen, scen_st)
  if (nq <= 32) {
    if (interp == 0) XH_ADJ(32, 0); else XH_ADJ(32, 1);
  } else {
    if (interp == 0) XH_ADJ(64, 0); else XH_ADJ(64, 1);
  }
#undef XH_ADJ
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // extern "C"
