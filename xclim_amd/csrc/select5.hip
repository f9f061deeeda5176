// select5.hip — exact per-column quantiles of series of ANY length: a most-significant-digit radix select for all the
// order statistics of a column at once (xsdba nbutils.quantile, E1 of SURVEY.md §8a; Hyndman-Fan type 7 =
// /root/reference/src/xclim/core/utils.py:370-395, 464-491 with alpha = beta = 1; reference call site
// /root/reference/src/xclim/sdba.py:10).
//
// Role: the column kernels of select.hip / select2.hip keep a column in registers or LDS and stop at 32768 samples.
// 1950-2100 daily (55 152 steps) and longer series go through the streaming histogram passes of select4.hip; this
// kernel takes what those passes hand back — columns whose target bins hold too many tied or clustered values — and any
// time-minor input longer than 32768.  It is a fallback: exact for every input (ties, NaN, +-inf, any T < 2^31), not
// tuned (five reads of a column that sits in L2 after the first one).
//
// One 256-thread workgroup per column, keys = order-preserving uint32 (NaN = 0xFFFFFFFF, skipped).  Round r (r = 0..3)
// looks at byte 3 - r of the key:
//   round 0: one 256-bin histogram of the top byte; every target (2 per quantile: the order statistics floor(vi) and
//            floor(vi) + 1 of utl:417-461) finds the digit that holds its rank and keeps the rank inside that digit;
//            the distinct digits become SLOTS (map0[digit] -> slot)
//   round r: a key whose prefix belongs to a slot counts into hist[slot][byte 3 - r]; every target finds its digit
//            inside its slot's histogram; the distinct (slot, digit) pairs become the next round's slots
// After round 3 a target's four digits ARE its key.  At most 64 slots (64 targets = 32 quantiles) per sweep; more
// quantiles take another sweep.
#include "common.h"

namespace {

constexpr int RS_NT = 256;
constexpr int RS_MAXS = 64;  // slots = distinct prefixes among the targets of one sweep
constexpr uint32_t RS_NANKEY = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t rs_rank(uint32_t n, double q, int side) {  // utl:395, 417-461 (type 7)
  if (n < 2u) return 0u;
  const double nn = (double)n;
  const double vi = nn * q + (1.0 + q * (1.0 - 1.0 - 1.0)) - 1.0;
  if (vi >= nn - 1.0) return n - 1u;
  if (vi < 0.0) return 0u;
  return (uint32_t)floor(vi) + (uint32_t)side;
}

__global__ void __launch_bounds__(RS_NT)
k_radix_select(const float* __restrict__ xcols, int64_t T, int64_t ncols, int64_t col_stride, const double* __restrict__ qs,
               int nq, float* __restrict__ out, int64_t ocs, int64_t oqs, const double* __restrict__ qcol) {
  // qcol != nullptr: ONE quantile per column, its probability qcol[column] (xsdba.nbutils.vecquantiles; nq = 1)
  // LDS (dynamic: 112.8 KB, one workgroup per CU): hist [64][256] u32 | map [3][64][256] u8: (slot, digit) of round r -> slot
  // of round r + 1 (0xFF: none) | per target: rank inside its current slot, digits found so far, slot
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t (*hist)[256] = reinterpret_cast<uint32_t (*)[256]>(smem);
  uint8_t (*map)[RS_MAXS][256] = reinterpret_cast<uint8_t (*)[RS_MAXS][256]>(smem + RS_MAXS * 256 * 4);
  uint32_t* t_rank = reinterpret_cast<uint32_t*>(smem + RS_MAXS * 256 * 4 + 3 * RS_MAXS * 256);
  uint32_t* t_key = t_rank + RS_MAXS;
  uint32_t* t_slot = t_key + RS_MAXS;
  uint32_t* s_misc = t_slot + RS_MAXS;  // [0] = valid samples, [1] = slots of the next round
  const int tid = threadIdx.x;
  for (int64_t c = blockIdx.x; c < ncols; c += gridDim.x) {
    const float* x = xcols + c * col_stride;
    for (int q0 = 0; q0 < nq; q0 += RS_MAXS / 2) {
      const int nqs = nq - q0 < RS_MAXS / 2 ? nq - q0 : RS_MAXS / 2;
      const int ntgt = 2 * nqs;
      uint32_t nslots = 1;
      for (int r = 0; r < 4; ++r) {
        for (int i = tid; i < (int)nslots * 256; i += RS_NT) (&hist[0][0])[i] = 0u;
        __syncthreads();
        const int shift = 24 - 8 * r;
        for (int64_t t = tid; t < T; t += RS_NT) {
          const uint32_t k = xh_f2key(x[t]);
          if (k == RS_NANKEY) continue;
          uint32_t s = 0;
          bool ok = true;
          for (int rr = 0; rr < r && ok; ++rr) {  // walk the prefix through the slot maps of the earlier rounds
            s = map[rr][s][(k >> (24 - 8 * rr)) & 255u];
            ok = s != 0xFFu;
          }
          if (ok) atomicAdd(&hist[s][(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (r == 0) {
          if (tid < 64) {  // valid samples = the sum of the top-byte histogram (one wave)
            uint32_t a = hist[0][tid] + hist[0][tid + 64] + hist[0][tid + 128] + hist[0][tid + 192];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) a += (uint32_t)__shfl_xor((int)a, d);
            if (tid == 0) s_misc[0] = a;
          }
          __syncthreads();
          if (tid < ntgt) {
            const double qq = qcol ? qcol[c] : qs[q0 + (tid >> 1)];
            t_rank[tid] = qq == qq ? rs_rank(s_misc[0], qq, tid & 1) : 0u;
            t_key[tid] = 0u;
            t_slot[tid] = 0u;
          }
          __syncthreads();
        }
        const uint32_t n = s_misc[0];
        // every target: the digit of its slot's histogram that holds its rank
        if (tid < ntgt && n > 0u) {
          const uint32_t* h = hist[t_slot[tid]];
          uint32_t rk = t_rank[tid], d = 0;
          for (; d < 255u; ++d) {
            const uint32_t cnt = h[d];
            if (rk < cnt) break;
            rk -= cnt;
          }
          t_rank[tid] = rk;
          t_key[tid] |= d << shift;
        }
        if (r < 3) {
          for (int i = tid; i < (int)nslots * 256; i += RS_NT) (&map[r][0][0])[i] = 0xFFu;
          __syncthreads();
          if (tid == 0) {  // distinct (slot, digit) pairs of the targets -> the next round's slots
            uint32_t ns = 0;
            if (n > 0u)
              for (int j = 0; j < ntgt; ++j) {
                const uint32_t s = t_slot[j], d = (t_key[j] >> shift) & 255u;
                uint32_t m = map[r][s][d];
                if (m == 0xFFu) {
                  m = ns++;
                  map[r][s][d] = (uint8_t)m;
                }
                t_slot[j] = m;
              }
            s_misc[1] = ns > 0u ? ns : 1u;
          }
          __syncthreads();
          nslots = s_misc[1];
        } else {
          __syncthreads();
        }
      }
      // Hyndman-Fan lerp (utl:464-491) of the two neighbours of every quantile
      if (tid < nqs) {
        const uint32_t n = s_misc[0];
        const float left = xh_key2f(t_key[2 * tid]), right = xh_key2f(t_key[2 * tid + 1]);
        double r;
        const double qq = qcol ? qcol[c] : qs[q0 + tid];
        if (n == 0u || qq != qq) r = xh_nan64();  // (a NaN probability: vecquantiles of a cell without valid sim samples)
        else if (n < 2u) r = (double)left;
        else {
          const double nn = (double)n;
          const double vi = nn * qq + (1.0 + qq * (1.0 - 1.0 - 1.0)) - 1.0;
          if (vi >= nn - 1.0 || vi < 0.0) r = (double)left;
          else {
            const double gamma = vi - floor(vi);
            const float diff = right - left;
            r = (double)left + (double)diff * gamma;
            if (gamma >= 0.5) r = (double)right - (double)diff * (1.0 - gamma);
          }
        }
        out[c * ocs + (int64_t)(q0 + tid) * oqs] = (float)r;
      }
      __syncthreads();
    }
  }
}

}  // namespace

int xh_select_columns_radix(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride, const double* d_q,
                            int nq, float* out, int64_t out_cstride, int64_t out_qstride) {
  XH_REQUIRE(T >= 1 && T < (1ll << 31) && nq >= 1, XH_ERR_LIMIT, "quantile_series: T = %lld outside [1, 2^31)", (long long)T);
  if (ncols <= 0) return XH_OK;
  int64_t nblk = ncols;
  const int64_t maxblk = (int64_t)ctx->num_cu;
  if (nblk > maxblk) nblk = maxblk;
  constexpr size_t lds = (size_t)RS_MAXS * 256 * 4 + 3 * RS_MAXS * 256 + 3 * RS_MAXS * 4 + 16;
  XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_radix_select, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k_radix_select, dim3((unsigned)nblk), dim3(RS_NT), lds, ctx->stream, xcols, T, ncols, col_stride, d_q, nq, out,
                     out_cstride, out_qstride, (const double*)nullptr);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

extern "C" {

// xsdba.nbutils.vecquantiles(da, rnk, dim): ONE quantile per cell, at that cell's own probability q_cell[c] (device,
// float64; NaN -> NaN) — Hyndman-Fan type 7 over the valid samples.  x (T, C) with element strides (st, sc), one of them
// 1; out (C) float32.  Used by adapt_freq (pth = the value of ref at the dry-day frequency of sim).  Time-major input
// goes through transposed column batches (the radix select reads a column five times).
int xh_quantile_cells(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const double* q_cell, float* out) {
  XH_REQUIRE(ctx && x && q_cell && out, XH_ERR_ARG, "xh_quantile_cells: NULL argument");
  XH_REQUIRE(T >= 1 && T < (1ll << 31) && C >= 0, XH_ERR_ARG, "xh_quantile_cells: bad shape");
  if (C == 0) return XH_OK;
  constexpr size_t lds = (size_t)RS_MAXS * 256 * 4 + 3 * RS_MAXS * 256 + 3 * RS_MAXS * 4 + 16;
  XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_radix_select, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  auto run = [&](const float* cols, int64_t n, int64_t cs, const double* qc, float* o) -> int {
    int64_t nblk = n < (int64_t)ctx->num_cu ? n : (int64_t)ctx->num_cu;
    hipLaunchKernelGGL(k_radix_select, dim3((unsigned)nblk), dim3(RS_NT), lds, ctx->stream, cols, T, n, cs, qc, 1, o, (int64_t)1, (int64_t)0, qc);
    XH_LAUNCH_CHECK();
    return XH_OK;
  };
  if (st == 1 && sc >= T) return run(x, C, sc, q_cell, out);
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT, "xh_quantile_cells: one of the strides must be 1 (st=%lld sc=%lld)", (long long)st,
             (long long)sc);
  const int64_t Tp = (T + 63) & ~(int64_t)63;
  int64_t batch = (int64_t)((1ull << 28) / (sizeof(float) * (size_t)Tp));
  batch = (batch / 128) * 128;
  if (batch < 128) batch = 128;
  if (batch > C) batch = C;
  void* tmp = nullptr;
  int rc = xh_big_scratch(ctx, sizeof(float) * (size_t)batch * (size_t)Tp, &tmp);
  if (rc) return rc;
  for (int64_t c0 = 0; c0 < C; c0 += batch) {
    const int64_t nb = C - c0 < batch ? C - c0 : batch;
    rc = xh_transpose_f32(ctx, x + c0, T, nb, st, (float*)tmp, Tp);
    if (rc) return rc;
    rc = run((const float*)tmp, nb, Tp, q_cell + c0, out + c0);
    if (rc) return rc;
  }
  return XH_OK;
}

}  // extern "C"
