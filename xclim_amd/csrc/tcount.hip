// tcount.hip — threshold_count against a per-doy fp64 table on MULTI-YEAR series (tx90p over 30 years).
//
// Reference: threshold_count (indices/generic.py:329-361) + compare (gen:301-326) on `tasmax > resample_doy(per)`
// (indices/_threshold.py tx90p; core/calendar.py resample_doy), resample(freq).sum, and the valid-sample count of
// MissingBase.is_valid (core/missing.py:201-220).
//
// k_threshold_count (reduce.hip) maps periods to workgroups: every year re-reads the (D, C) fp64 table (8 bytes per doy
// and cell, as much as the samples of two years) and depends on L2 / Infinity Cache for it — 11.8 ms at 30 years x
// 1440 x 720 (4.1 TB/s algorithmic).  Here a workgroup owns a tile of 64 columns for ALL rows:
//   * the tile's table slice is converted ONCE to the fp32 thresholds of f32thr.h (the fp64 compare of the reference
//     is exactly one fp32 compare against below(r) / above(r)) and sits in LDS: (D + 1) x 64 floats (row D = NaN for a
//     step whose doy the table does not hold);
//   * the samples are streamed with the row-lane geometry of select4.hip (64 columns x 16 row lanes, 16 loads per lane
//     in flight, 256-byte row segments): a wave = one row lane, so the doy and the period of a row are wave-uniform —
//     they come as ONE scalar 64-byte load per batch from a table that k_tc_meta lays out in visiting order;
//   * hits and valid samples accumulate in one packed register per lane (hits | valid << 16) and are added to the LDS
//     counters [P + 1][64] when the wave's row crosses into another period (row P collects steps outside every period).
//     NARROW (every period shorter than 256 steps, e.g. monthly): hits | valid << 8 in 16-bit counters, two columns per
//     LDS word — 360 months x 64 columns fit beside the table.
// HBM traffic: the samples once + the table once + the counts.
#include "f32thr.h"
#include "rowstream.h"

namespace {

constexpr int TC_CW = 64, TC_RL = 16, TC_U = 16, TC_NT = TC_CW * TC_RL, TC_ROWS = TC_RL * TC_U;

// meta[(kb * 16 + rl) * 16 + u] = doy index | period << 16 of row kb * 256 + u * 16 + rl (the order xh_row_stream visits them)
__global__ void k_tc_meta(const int32_t* __restrict__ tidx, const int64_t* __restrict__ seg, int P, int64_t T, int64_t nslots,
                          int ndoy, uint32_t* __restrict__ meta) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nslots) return;
  const int64_t kb = i >> 8;
  const int rl = (int)(i >> 4) & 15, u = (int)i & 15;
  const int64_t t = kb * TC_ROWS + u * TC_RL + rl;
  uint32_t dy = (uint32_t)ndoy, per = (uint32_t)P;
  if (t < T) {
    const int d = tidx[t];
    if (d >= 0 && d < ndoy) dy = (uint32_t)d;
    if (t >= seg[0] && t < seg[P]) {
      int lo = 0, hi = P;  // seg[lo] <= t < seg[hi]
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (seg[mid] <= t) lo = mid; else hi = mid;
      }
      per = (uint32_t)lo;
    }
  }
  meta[i] = dy | (per << 16);
}

template <int OP>
__device__ __forceinline__ bool tc_cmp(float x, float thr) {
  return OP == XH_OP_GT ? x > thr : OP == XH_OP_LT ? x < thr : OP == XH_OP_GE ? x >= thr : x <= thr;
}

template <int OP, bool NARROW>
__global__ void __launch_bounds__(TC_NT, 4)
k_tc_doy(const float* __restrict__ x, int T, int64_t C, int64_t st, const double* __restrict__ table, int64_t tstride,
         const uint32_t* __restrict__ meta, int ndoy, int P, int32_t* __restrict__ count_out, int32_t* __restrict__ valid_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* tab = reinterpret_cast<float*>(smem);                        // [ndoy + 1][64]
  uint32_t* cnt = reinterpret_cast<uint32_t*>(tab + (ndoy + 1) * TC_CW);  // [P + 1][64] (NARROW: 16-bit, [P + 1][32] words)
  constexpr int VSH = NARROW ? 8 : 16, CPW = NARROW ? 2 : 1;           // valid-count shift, counters per LDS word
  constexpr uint32_t VONE = 1u << VSH;
  const int tid = threadIdx.x, col = tid & (TC_CW - 1), rl = tid / TC_CW;
  const int rlu = __builtin_amdgcn_readfirstlane(rl);  // a wave is one row lane
  const int64_t ntiles = (C + TC_CW - 1) / TC_CW;
  for (int i = tid; i < (P + 1) * TC_CW / CPW; i += TC_NT) cnt[i] = 0u;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t c = tile * TC_CW + col;
    const int64_t cc = c < C ? c : C - 1;  // a column past C reads the last one and is never stored
    // the tile's table slice: 12 rows per lane in flight (a dependent load -> convert -> store loop costs one HBM latency
    // per 16 rows: 23 of them per tile, a third of the tile's streaming time)
    for (int d0 = 0; d0 < ndoy; d0 += 12 * TC_RL) {
      double buf[12];
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        int d = d0 + j * TC_RL + rl;
        d = d < ndoy ? d : ndoy - 1;
        buf[j] = table[(int64_t)d * tstride + cc];
      }
#pragma unroll
      for (int j = 0; j < 12; ++j) {
        const int d = d0 + j * TC_RL + rl;
        if (d < ndoy) tab[d * TC_CW + col] = f32_threshold(buf[j], OP);
      }
    }
    if (rl == 0) tab[ndoy * TC_CW + col] = __uint_as_float(0x7FC00000u);
    __syncthreads();
    uint32_t acc = 0u;
    uint32_t cur = (uint32_t)P;
    const float* mytab = tab + col;
    uint32_t* mycnt = cnt + col / CPW;
    const int csh = NARROW ? (col & 1) * 16 : 0;
    auto flush = [&](uint32_t period, uint32_t a) { atomicAdd(&mycnt[period * (TC_CW / CPW)], a << csh); };
    xh_row_stream<TC_U, TC_RL>(x, T, st, cc, rl, [&](const float (&v)[TC_U], int kb) {
      // the 16 (doy, period) words of this wave's rows: four scalar 16-byte loads, then 16 LDS reads in flight
      const uint4* mp = reinterpret_cast<const uint4*>(meta + ((int64_t)kb * TC_RL + rlu) * TC_U);
      uint32_t m[TC_U];
#pragma unroll
      for (int j = 0; j < TC_U / 4; ++j) {
        const uint4 w = mp[j];
        m[4 * j] = w.x; m[4 * j + 1] = w.y; m[4 * j + 2] = w.z; m[4 * j + 3] = w.w;
      }
      float thr[TC_U];
#pragma unroll
      for (int u = 0; u < TC_U; ++u) thr[u] = mytab[(m[u] & 0xFFFFu) * TC_CW];
      uint32_t h[TC_U];
#pragma unroll
      for (int u = 0; u < TC_U; ++u) h[u] = (tc_cmp<OP>(v[u], thr[u]) ? 1u : 0u) + (v[u] == v[u] ? VONE : 0u);
      // all 16 rows in the current period (wave-uniform, scalar xor / or of the 16 words).  Testing only the first and the
      // last row is not enough: rows before the first and after the last period share one dummy period, and a period
      // shorter than the 256 rows a batch spans can lie between two of them.
      uint32_t pdiff = 0u;
#pragma unroll
      for (int u = 1; u < TC_U; ++u) pdiff |= m[u] ^ m[0];
      if ((pdiff >> 16) == 0u && (m[0] >> 16) == cur) {
        uint32_t sum = 0u;
#pragma unroll
        for (int u = 0; u < TC_U; ++u) sum += h[u];
        acc += sum;
      } else {
#pragma unroll
        for (int u = 0; u < TC_U; ++u) {
          const uint32_t pp = m[u] >> 16;
          if (pp != cur) {
            flush(cur, acc);
            acc = 0u;
            cur = pp;
          }
          acc += h[u];
        }
      }
    });
    flush(cur, acc);
    __syncthreads();
    for (int i = tid; i < P * TC_CW; i += TC_NT) {
      const int64_t c2 = tile * TC_CW + (i & (TC_CW - 1));
      const uint32_t w = NARROW ? (cnt[i >> 1] >> ((i & 1) * 16)) & 0xFFFFu : cnt[i];
      if (c2 < C) {
        const int64_t o = (int64_t)(i / TC_CW) * C + c2;
        count_out[o] = (int32_t)(w & (VONE - 1u));
        if (valid_out) valid_out[o] = (int32_t)(w >> VSH);
      }
    }
    __syncthreads();
    for (int i = tid; i < (P + 1) * TC_CW / CPW; i += TC_NT) cnt[i] = 0u;
    __syncthreads();
  }
}

}  // namespace

// ---- host side ---------------------------------------------------------------------------------------------------------
// xh_tcount_plan: XH_OK and the launch shape, or XH_ERR_NOTIMPL (no error text) = outside the tile kernel's domain (the
// caller uses k_threshold_count).  longest = steps of the longest period.
int xh_tcount_plan(int64_t T, int64_t C, int64_t st, int op, int P, int ndoy, int64_t longest, size_t* lds, int* narrow) {
  if (op < XH_OP_GT || op > XH_OP_LE) return XH_ERR_NOTIMPL;
  if (ndoy < 1 || ndoy > 0xFFFE || P < 1 || P > 0xFFFE || T > 0x7FFFFFFF) return XH_ERR_NOTIMPL;
  if (longest > 0xFFFF) return XH_ERR_NOTIMPL;  // 16-bit halves of the packed counters
  *narrow = longest <= 0xFF ? 1 : 0;            // 8-bit halves: two columns per LDS word
  *lds = (size_t)(ndoy + 1) * TC_CW * 4 + (size_t)(P + 1) * TC_CW * (*narrow ? 2 : 4);
  if (*lds > 156 * 1024) return XH_ERR_NOTIMPL;
  if (T < 2048 || T < 3 * (int64_t)ndoy || C < TC_CW) return XH_ERR_NOTIMPL;  // one or two years: the table is read about once anyway
  if ((int64_t)TC_ROWS * st * 4 >= ((int64_t)1 << 32) || C * 4 + (int64_t)TC_RL * st * 4 >= ((int64_t)1 << 32)) return XH_ERR_NOTIMPL;
  return XH_OK;
}

int64_t xh_tcount_meta_slots(int64_t T) { return (T + TC_ROWS - 1) / TC_ROWS * TC_ROWS; }

// slot of row t in the visiting order of xh_row_stream (host twin of k_tc_meta's index arithmetic)
int64_t xh_tcount_slot_of_row(int64_t t) {
  const int64_t kb = t / TC_ROWS, r = t % TC_ROWS;
  return (kb * TC_RL + r % TC_RL) * TC_U + r / TC_RL;
}

int xh_tcount_run(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int op, const double* table, int64_t tstride,
                  const uint32_t* meta, int P, int ndoy, int narrow, size_t lds, int32_t* count_out, int32_t* valid_out) {
  const int64_t ntiles = cdiv64(C, TC_CW);
  const unsigned grid = (unsigned)(ntiles < ctx->num_cu ? ntiles : ctx->num_cu);  // one 1024-thread workgroup per CU (LDS)
#define XH_TCD2(OPV, NW)                                                                                                      \
  {                                                                                                                           \
    XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_tc_doy<OPV, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_tc_doy<OPV, NW>), dim3(grid), dim3(TC_NT), lds, ctx->stream, x, (int)T, C, st, table, tstride, meta,  \
                       ndoy, P, count_out, valid_out);                                                                        \
  }
#define XH_TCD(OPV)                         \
  case OPV:                                 \
    if (narrow) XH_TCD2(OPV, true) else XH_TCD2(OPV, false) \
    break;
  switch (op) {
    XH_TCD(XH_OP_GT)
    XH_TCD(XH_OP_LT)
    XH_TCD(XH_OP_GE)
    XH_TCD(XH_OP_LE)
  }
#undef XH_TCD2
#undef XH_TCD
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_launch_tcount_doy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int op, const double* table,
                         int64_t tstride, const int32_t* tidx, const int64_t* d_seg, const int64_t* h_seg, int P, int ndoy,
                         int32_t* count_out, int32_t* valid_out) {
  int64_t longest = 0;
  for (int p = 0; p < P; ++p) longest = h_seg[p + 1] - h_seg[p] > longest ? h_seg[p + 1] - h_seg[p] : longest;
  size_t lds = 0;
  int narrow = 0;
  int rc = xh_tcount_plan(T, C, st, op, P, ndoy, longest, &lds, &narrow);
  if (rc) return rc;
  const int64_t nslots = xh_tcount_meta_slots(T);
  void* scratch = nullptr;
  rc = xh_big_scratch(ctx, (size_t)nslots * 4, &scratch);
  if (rc) return rc;
  uint32_t* meta = static_cast<uint32_t*>(scratch);
  hipLaunchKernelGGL(k_tc_meta, dim3((unsigned)cdiv64(nslots, 256)), dim3(256), 0, ctx->stream, tidx, d_seg, P, T, nslots, ndoy, meta);
  return xh_tcount_run(ctx, x, T, C, st, op, table, tstride, meta, P, ndoy, narrow, lds, count_out, valid_out);
}
