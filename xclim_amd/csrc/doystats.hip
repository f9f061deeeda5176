// doystats.hip — climatological_mean_doy (core/calendar.py:907-931) from per-day-set partial sums.
//
// The sample set of doy d is the union of the W day-sets {(year y, doy d + k)} (same decomposition as the multi-year
// percentile kernels, pdoy.h).  One lane per cell marches over a chunk of doys: every day-set is gathered ONCE (rows
// resolved with one vector load + readlane, unconditional loads), reduced to (sum, sum of squares, count) of the values
// shifted by a per-cell pivot K, kept in a ring of W partials; mean = K + S1 / N, var = (S2 - S1^2 / N) / N in fp64 (the
// shift keeps the one-pass variance well conditioned: |v - K| is of the order of the spread, not of the magnitude).
// The generic kernel (reduce2.hip) re-reads every row W times and twice for its two-pass std; it remains the exact path
// for irregular doys (calendar gaps).
#include <stdlib.h>

#include "pdoy.h"

namespace {

template <int W, int NYP>
__global__ void __launch_bounds__(64)
k_doy_stats_sets(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, const int32_t* __restrict__ tbase, int nyears,
                 int ndoy, int chunk, const uint8_t* __restrict__ regular, float* __restrict__ mean_out,
                 float* __restrict__ std_out) {
  const int lane = threadIdx.x;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  const bool active = c < C;
  const int64_t cc = active ? c : C - 1;
  constexpr int half = W / 2;
  int d0 = blockIdx.y * chunk, d1 = d0 + chunk;
  if (d1 > ndoy) d1 = ndoy;
  double s1[W], s2[W];
  int n[W];
#pragma unroll
  for (int w = 0; w < W; ++w) { s1[w] = 0.0; s2[w] = 0.0; n[w] = 0; }
  float raw[NYP];
  float K = 0.0f;
  bool haveK = false;
  auto rows_of = [&](int dn) { return pdoy_row(lane, nyears, ndoy, dn, 0, tbase, nullptr, T, T); };
  auto reduce_into = [&](double& a1, double& a2, int& an) {
    if (!haveK) {  // pivot: the first valid value this lane meets
#pragma unroll
      for (int y = 0; y < NYP; ++y)
        if (!haveK && raw[y] == raw[y]) { K = raw[y]; haveK = true; }
    }
    double b1 = 0.0, b2 = 0.0;
    int bn = 0;
#pragma unroll
    for (int y = 0; y < NYP; ++y) {
      const float v = raw[y];
      const bool ok = v == v;
      const double dv = ok ? (double)v - (double)K : 0.0;
      b1 += dv;
      b2 += dv * dv;
      bn += ok ? 1 : 0;
    }
    a1 = b1; a2 = b2; an = bn;
  };
  // ring[w] = day-set of doy (d - half + w); prologue fills slots 1 .. W-1 for d = d0 - 1
#pragma unroll
  for (int w = 1; w < W; ++w) {
    pdoy_gather<NYP>(raw, rows_of(d0 - 1 - half + w), x, st, cc);
    reduce_into(s1[w], s2[w], n[w]);
  }
  pdoy_gather<NYP>(raw, rows_of(d0 + half), x, st, cc);
  int rows_next = rows_of(d0 + half + 1);
  // The ring does not shift: the day loop is unrolled by W and the partial of the day-set that enters at step u of a group
  // lands in the static slot (u + W - 1) % W — the slot of the day-set that leaves (20 64-bit moves per doy before).
  // Running window sums: RS += entering partial - leaving partial (fp64 drift over a chunk of doys ~1e-15 relative; the
  // contract is 1e-6).  1 / N is a wave-uniform constant when every lane's window is complete (the usual case), the
  // standard deviation comes from the hardware fp32 square root: the divide / divide / fp64 sqrt per doy was most of
  // the VALU work of this (VALU-bound) kernel on short base periods.
#pragma unroll
  for (int w = 0; w < W - 1; ++w) { s1[w] = s1[w + 1]; s2[w] = s2[w + 1]; n[w] = n[w + 1]; }  // prologue slots 1..W-1 -> 0..W-2
  s1[W - 1] = 0.0; s2[W - 1] = 0.0; n[W - 1] = 0;
  double RS1 = 0.0, RS2 = 0.0;
  int RN = 0;
#pragma unroll
  for (int w = 0; w < W - 1; ++w) { RS1 += s1[w]; RS2 += s2[w]; RN += n[w]; }
  const int nfull = W * nyears;
  const double inv_full = 1.0 / (double)nfull;
  for (int db = d0; db < d1; db += W) {
#pragma unroll
    for (int u = 0; u < W; ++u) {
      const int d = db + u;
      if (d < d1) {
        const int SLOT = (u + W - 1) % W;  // (compile-time after unrolling)
        double b1, b2;
        int bn;
        reduce_into(b1, b2, bn);
        RS1 += b1 - s1[SLOT]; RS2 += b2 - s2[SLOT]; RN += bn - n[SLOT];
        s1[SLOT] = b1; s2[SLOT] = b2; n[SLOT] = bn;
        if (d + 1 < d1) {
          pdoy_gather<NYP>(raw, rows_next, x, st, cc);
          rows_next = rows_of(d + 2 + half);
        }
        if (regular[d]) {
          const bool allfull = __all(RN == nfull ? 1 : 0) != 0;
          float m = xh_nan32(), sd = xh_nan32();
          if (allfull || RN > 0) {
            const double invN = allfull ? inv_full : 1.0 / (double)(RN > 0 ? RN : 1);
            const double mean_s = RS1 * invN;
            double var = (RS2 - RS1 * mean_s) * invN;
            var = var > 0.0 ? var : 0.0;
            m = (float)((double)K + mean_s);
            sd = __builtin_amdgcn_sqrtf((float)var);
          }
          if (active) {
            mean_out[(int64_t)d * C + c] = m;
            std_out[(int64_t)d * C + c] = sd;
          }
        }
      }
    }
  }
}

// ONE contiguous year (row of doy index i = t0 + i): the day-sets are single rows, the climatology is a centred,
// NaN-skipping rolling mean / std over consecutive rows.  Four cells per lane, rows in double-buffered batches of 8
// (xh_march_rows), the window in a register ring with static slots; mean and population std (two passes over the W
// values in fp64 — exact, no pivot) per doy.  The per-day-set kernel above moves 256 bytes per wave and row.
template <int W, int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_doy_stats_year(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, int64_t t0, int ndoy, int chunk,
                 const uint8_t* __restrict__ regular, float* __restrict__ mean_out, float* __restrict__ std_out) {
  const int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  constexpr int half = W / 2;
  const int d0 = blockIdx.y * chunk;
  int d1 = d0 + chunk;
  if (d1 > ndoy) d1 = ndoy;
  if (d0 >= d1) return;
  float ring[VEC][W];
#pragma unroll
  for (int i = 0; i < VEC; ++i)
#pragma unroll
    for (int k = 0; k < W; ++k) ring[i][k] = xh_nan32();
  // rows t0 + d0 - half .. t0 + d1 - 1 + half; rows outside the series are absent days (NaN)
  const int64_t ta = t0 + d0 - half, tb = t0 + d1 - 1 + half + 1;
  const int64_t ra = ta < 0 ? 0 : ta, rb = tb > T ? T : tb;
  auto emit = [&](int64_t tnew) {  // tnew: the row that just entered = the newest of the window of doy d
    const int64_t d = tnew - half - t0;
    if (d < d0 || d >= d1 || !regular[d]) return;
    float m[VEC], sd[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      double sum = 0.0;
      int n = 0;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const float v = ring[i][k];
        const bool ok = v == v;
        sum += ok ? (double)v : 0.0;
        n += ok ? 1 : 0;
      }
      const double inv = 1.0 / (double)(n > 0 ? n : 1);
      const double mean = sum * inv;
      double s2 = 0.0;
#pragma unroll
      for (int k = 0; k < W; ++k) {
        const float v = ring[i][k];
        const double dv = (v == v) ? (double)v - mean : 0.0;
        s2 += dv * dv;
      }
      m[i] = n > 0 ? (float)mean : xh_nan32();
      sd[i] = n > 0 ? __builtin_amdgcn_sqrtf((float)(s2 * inv)) : xh_nan32();
    }
    float* mo = mean_out + d * C + c;
    float* so = std_out + d * C + c;
    if (VEC == 4) {
      *reinterpret_cast<float4*>(mo) = make_float4(m[0], m[1 % VEC], m[2 % VEC], m[3 % VEC]);
      *reinterpret_cast<float4*>(so) = make_float4(sd[0], sd[1 % VEC], sd[2 % VEC], sd[3 % VEC]);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) { mo[i] = m[i]; so[i] = sd[i]; }
    }
  };
  auto push = [&](const VecF<VEC>& xv) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
#pragma unroll
      for (int k = 0; k < W - 1; ++k) ring[i][k] = ring[i][k + 1];
      ring[i][W - 1] = xv.v[i];
    }
  };
  VecF<VEC> nanv;
#pragma unroll
  for (int i = 0; i < VEC; ++i) nanv.v[i] = xh_nan32();
  for (int64_t t = ta; t < ra; ++t) { push(nanv); emit(t); }  // absent rows before the series
  xh_march_rows<VEC, 8>(x + c, st, ra, rb, [&](int64_t t, const VecF<VEC>& xv) {
    push(xv);
    emit(t);
  });
  for (int64_t t = (rb > ra ? rb : ra); t < tb; ++t) { push(nanv); emit(t); }  // absent rows after it
}

}  // namespace

int xh_launch_doy_stats_sets(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* d_tb, int nyears,
                             int ndoy, int window, const uint8_t* d_reg, float* mean_out, float* std_out, int64_t year_t0) {
  if (!(window == 3 || window == 5 || window == 7) || nyears > 64) return XH_ERR_NOTIMPL;
  if (nyears == 1 && year_t0 >= 0) {  // one contiguous year: rolling form
    const int vec = (xh_pick_vec(x, C, st) == 4 && xh_pick_vec(mean_out, C, C) == 4 && xh_pick_vec(std_out, C, C) == 4) ? 4 : 1;
    const int64_t cblocks = cdiv64(cdiv64(C, vec), XH_BLOCK);
    int64_t gy = cdiv64((int64_t)ctx->num_cu * 12, cblocks);
    if (gy < 1) gy = 1;
    if (gy > cdiv64(ndoy, 32)) gy = cdiv64(ndoy, 32);
    if (gy < 1) gy = 1;
    const int ychunk = (int)cdiv64(ndoy, gy);
    const dim3 ygrid((unsigned)cblocks, (unsigned)cdiv64(ndoy, ychunk));
#define XH_DY(W, V) \
  hipLaunchKernelGGL((k_doy_stats_year<W, V>), ygrid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, year_t0, ndoy, ychunk, d_reg, mean_out, std_out)
    if (vec == 4) { if (window == 3) XH_DY(3, 4); else if (window == 5) XH_DY(5, 4); else XH_DY(7, 4); }
    else { if (window == 3) XH_DY(3, 1); else if (window == 5) XH_DY(5, 1); else XH_DY(7, 1); }
#undef XH_DY
    XH_LAUNCH_CHECK();
    return XH_OK;
  }
  int chunk = 24;
  if (const char* e = xh_diag_env("XH_DOYSTATS_CHUNK")) chunk = atoi(e) > 0 ? atoi(e) : chunk;  // diagnostics
  const dim3 grid((unsigned)cdiv64(C, 64), (unsigned)((ndoy + chunk - 1) / chunk));
#define XH_DS(W, NY)                                                                                                     \
  hipLaunchKernelGGL((k_doy_stats_sets<W, NY>), grid, dim3(64), 0, ctx->stream, x, T, C, st, d_tb, nyears, ndoy, chunk, d_reg, \
                     mean_out, std_out)
#define XH_DSW(NY)                                                                   \
  do {                                                                               \
    if (window == 3) XH_DS(3, NY); else if (window == 5) XH_DS(5, NY); else XH_DS(7, NY); \
  } while (0)
  if (nyears == 1) XH_DSW(1);  // (one sample per day-set: no padded second slot to gather and mask)
  else if (nyears <= 2) XH_DSW(2);
  else if (nyears <= 8) XH_DSW(8);
  else if (nyears <= 32) XH_DSW(32);
  else XH_DSW(64);
#undef XH_DSW
#undef XH_DS
  XH_LAUNCH_CHECK();
  return XH_OK;
}
