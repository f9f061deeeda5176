// pdoy_top.hip — percentile_doy on multi-year base periods, register top-16 variant.
#include <stdlib.h>

#include "f32thr.h"
#include "pdoy.h"
#include "topnet.h"

// ---- multi-year path, register variant: top-16 of the W day-sets by bitonic half-merges ---------------------
// For high (or, mirrored, low) percentiles only the 16 largest samples of a doy can be selected (e.g. per = 90 over
// 30 years x 5 days: ranks 134/135 of 150).  Each day-set is sorted once (descending, NaN last) and only its top 16
// are kept — in REGISTERS.  The top 16 of a union of lists come from bitonic half-merges (C[i] = max(A[i], B[15-i]) is
// bitonic and holds the 16 largest of A u B; 4 compare-exchange stages re-sort it): static networks, no LDS, no
// data-dependent loops, lots of independent work per lane.
// Neighbouring windows share their day-sets, so the merges are shared too: pr[k] = top16(L_{d-h+k} u L_{d-h+k+1}) (the
// PAIR of two consecutive day-sets, made once, used by two windows), and window_d = pr[0] u pr[2] u ... u L_{d+h}:
// W/2 + 1 merges per doy instead of W - 1 (W = 5: 3 instead of 4) in the same W x 16 registers.  The last of them need not
// re-sort when the wave's ranks are the two lowest of the top 16 (per = 90 of 150 samples: ranks 134 / 135 = positions
// 15 / 14): the two smallest of the bitonic C take 16 comparators instead of 32 compare-exchanges.
// REV selects within the 16 SMALLEST instead (low percentiles): the same networks with v_max / v_min swapped (topnet.h).
// Round 4: the networks run on the FLOATS themselves, NaN and absent days mapped to the innermost value (-inf / +inf) and
// counted: 3 VALU instructions per sample (compare, select, add-with-carry) instead of the ~11 of the ordered-integer key
// (sign test, two bit operations, select, NaN test, select, validity test, count, mirror, mask), which was a third of the
// kernel's instructions.  An infinity that stands for a NaN and a sample that IS that infinity are the same value, and only
// the positions below the valid count are ever read.

// COUNT = true (xh_percentile_doy_count on a multi-year base period): the percentile of doy d is compared with the
// samples of day d of EVERY year and the exceedances are counted per (year, doy) -> period; the (D, C) fp64 table of the
// unfused chain is neither written nor re-read once per year.  One percentile (nsub == 1), regular doys only.
template <int W, int NYP, bool COUNT = false, bool REV = false>
__global__ void __launch_bounds__(64, (COUNT || W > 5 || NYP > 32) ? 2 : 3)
k_pdoy_top16(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, const int32_t* __restrict__ tbase, int nyears,
             int ndoy, int chunk, const QTab* __restrict__ qtab, const int32_t* __restrict__ jmap, int nsub,
             double* __restrict__ out, const int32_t* __restrict__ vmap, int64_t Tv, const uint8_t* __restrict__ regular,
             const float* __restrict__ nanrow, int op = 0, const int32_t* __restrict__ yd_period = nullptr,
             int32_t* __restrict__ cnt_out = nullptr, int32_t* __restrict__ valid_out = nullptr,
             const uint8_t* __restrict__ newseg = nullptr) {
  constexpr bool rev = REV;
  float SENT = tn_sentinel<REV>();
  asm volatile("" : "+v"(SENT));  // one VGPR, not a literal per use
  const int lane = threadIdx.x;
  int64_t c = (int64_t)blockIdx.x * 64 + lane;
  const bool active = c < C;
  constexpr int half = W / 2;
  constexpr bool FASTSEL = !COUNT;
  const int N = nyears * W;
  float pr[W - 1][16];  // pr[k]: pair of the day-sets of doys d - half + k and d - half + k + 1
  float last[16];       // day-set of doy d + half
  int cnt[W];              // valid samples of the day-sets d - half .. d + half
  float raw[NYP];

  const int64_t cc_ = active ? c : C - 1;  // inactive lanes read a valid cell and never store
  auto rows_of = [&](int dn, int off) { return pdoy_row(lane, nyears, ndoy, dn, off, tbase, vmap, Tv, T); };
  auto gather = [&](int rowv) { pdoy_gather<NYP, true>(raw, rowv, x, st, cc_, nanrow); };
  // a gathered day-set -> values with NaN / absent day / padding mapped to the innermost value, + the valid count
  auto convert = [&](float (&key)[NYP], int& nv) {
    int nn = 0;
#pragma unroll
    for (int y = 0; y < NYP; ++y) {
      tn_denan(key[y], nn, raw[y], SENT);
    }
    nv = NYP - nn;
  };
  // top 16 of the NYP values, sorted (in the possibly mirrored order): blocks of 16 through the optimal network, combined
  // with half-merges (NYP = 32: 60 + 60 + 16 + 32 = 168 comparator-equivalents instead of 240 for a full bitonic-32)
  auto sort_top = [&](const float (&key)[NYP], float (&top)[16]) {
    float blk[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) top[i] = key[i];
    tn_sort16<REV>(top);
#pragma unroll
    for (int b = 1; b < NYP / 16; ++b) {
#pragma unroll
      for (int i = 0; i < 16; ++i) blk[i] = key[b * 16 + i];
      tn_sort16<REV>(blk);
      tn_merge16<REV>(top, top, blk);
    }
  };
  // COUNT: one packed counter per year and lane (low half: exceedances, high half: valid days; a chunk has < 2^16 doys).
  // Period boundaries fall on the same doy for every year (`newseg[d]`, checked by the host), so ALL years are flushed
  // together — one atomic per year and lane — when a new segment starts and at the end of the doy chunk; the period of
  // (year y, doy) comes from the table at flush time.  The samples of day d are re-read (the day-set left the registers
  // W/2 + 1 doys ago) ONE DOY AHEAD of their use: gathered and used in the same step they cost a full L2 / HBM latency
  // per doy (the first version of this kernel: 31.8 ms fused against 18.2 + 11.9 ms for the two-step chain).
  const bool prim_lt = op == XH_OP_LT || op == XH_OP_GE, complement = op == XH_OP_GE || op == XH_OP_LE;
  const float sgn = prim_lt ? -1.0f : 1.0f;
  uint32_t cc[COUNT ? NYP : 1];
  float xv[COUNT ? NYP : 1];
  // the prefetched samples wait in LDS (lane-private column) while the sort networks need the registers
  __shared__ float xs[COUNT ? NYP * 64 : 1];
  if (COUNT) {
#pragma unroll
    for (int y = 0; y < NYP; ++y) cc[y] = 0u;
  }
  auto flush_all = [&](int dlast) {  // dlast: a doy of the segment that ends
    if (COUNT) {
      const int perv = (lane < nyears) ? yd_period[(int64_t)lane * ndoy + dlast] : -1;
#pragma unroll
      for (int y = 0; y < NYP; ++y) {
        const int pp = __builtin_amdgcn_readlane(perv, y);
        if (pp >= 0 && active) {
          const int nvld = (int)(cc[y] >> 16);
          const int nc = complement ? nvld - (int)(cc[y] & 0xFFFFu) : (int)(cc[y] & 0xFFFFu);
          if (nc) atomicAdd(&cnt_out[(int64_t)pp * C + c], nc);
          if (valid_out && nvld) atomicAdd(&valid_out[(int64_t)pp * C + c], nvld);
        }
        cc[y] = 0u;
      }
    }
  };
  auto fetch_day = [&](int d) {  // samples of (year y, doy d) for the count of doy d
    if constexpr (COUNT) pdoy_gather<NYP, true>(xv, rows_of(d, 0), x, st, cc_, nanrow);
  };
  auto stash_day = [&]() {  // registers -> LDS at the top of the step that uses them
    if constexpr (COUNT) {
#pragma unroll
      for (int y = 0; y < NYP; ++y) xs[y * 64 + lane] = xv[y];
    }
  };
  // One primitive, "sample > threshold", serves the four operators: < is > on negated values, and >= / <= are the
  // complements within the valid samples (hits = valid - count(x < r), resp. valid - count(x > r); applied at flush time).
  auto count_day = [&](int d, double r, bool all_valid) {
    if (!COUNT) return;
    const float thr = sgn * f32_threshold(r, prim_lt ? XH_OP_LT : XH_OP_GT);
    if (all_valid) {  // the day-set holds no NaN / absent day (wave-uniform): one multiply, compare and add-with-carry per sample
#pragma unroll
      for (int y = 0; y < NYP; ++y) cc[y] += 0x10000u + ((xs[y * 64 + lane] * sgn > thr) ? 1u : 0u);
    } else {
#pragma unroll
      for (int y = 0; y < NYP; ++y) {  // an absent row was gathered as NaN: never a hit, never valid
        const float v = xs[y * 64 + lane];
        cc[y] += ((v * sgn > thr) ? 1u : 0u) + ((v == v) ? 0x10000u : 0u);
      }
    }
  };
  auto select_and_store = [&](int d) {
    float t16[16];
    int n = cnt[0];
#pragma unroll
    for (int i = 0; i < 16; ++i) t16[i] = pr[0][i];
#pragma unroll
    for (int k = 2; k <= W - 3; k += 2) tn_merge16<REV>(t16, t16, pr[k]);
#pragma unroll
    for (int w = 1; w < W; ++w) n += cnt[w];
    const bool all_valid = COUNT && __all(cnt[half] == nyears ? 1 : 0) != 0;
    // fast path: one percentile, the same sample count in every lane, ranks at positions 14 / 15 of the top 16
    if (FASTSEL && nsub == 1) {
      const int n0 = __builtin_amdgcn_readfirstlane(n);
      if (__all(n == n0 ? 1 : 0)) {
        const int j = jmap[0];
        const QTab e = qtab[j * (N + 1) + n0];  // wave-uniform
        const int plo = rev ? e.lo : (n0 - 1 - e.lo), phi = rev ? e.hi : (n0 - 1 - e.hi);
        if (e.lo >= 0 && plo >= 14 && plo <= 15 && phi >= 14 && phi <= 15) {
          float k16, k15;
          tn_last2<REV>(t16, last, k16, k15);
          const float left = plo == 15 ? k16 : k15, right = phi == 15 ? k16 : k15;
          const float diff = right - left;
          double r = (double)left + (double)diff * e.gamma;
          if (e.gamma >= 0.5) r = (double)right - (double)diff * (1.0 - e.gamma);
          if (!__any(r != r ? 1 : 0)) {  // (+-inf samples: the nanmax rule below needs the whole list)
            if (COUNT) count_day(d, r, all_valid);
            else if (active) out[((int64_t)j * ndoy + d) * C + c] = r;
            return;
          }
        }
      }
    }
    tn_merge16<REV>(t16, t16, last);
    auto get = [&](int idx) -> float {
      uint32_t g = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) g |= (i == idx) ? __float_as_uint(t16[i]) : 0u;
      return __uint_as_float(g);
    };
    for (int jj = 0; jj < nsub; ++jj) {
      const int j = jmap[jj];
      const QTab e = qtab[j * (N + 1) + n];
      double r = xh_nan64();
      bool needmax = false;
      if (e.lo >= 0) {
        // position in the (mirrored) descending top-16: rev -> rank from the bottom, else rank from the top
        const int plo = rev ? e.lo : (n - 1 - e.lo), phi = rev ? e.hi : (n - 1 - e.hi);
        float left = get(plo), right = get(phi);
        float diff = right - left;
        r = (double)left + (double)diff * e.gamma;
        if (e.gamma >= 0.5) r = (double)right - (double)diff * (1.0 - e.gamma);
        if (r != r && n > 0) {  // +-inf samples: nanmax fallback (utl:552-554)
          if (!rev) r = (double)get(0);
          else if (n <= 16) r = (double)get(n - 1);
          else needmax = true;  // -inf at the selected ranks: the largest sample is not among the 16 smallest
        }
      }
      if (rev && __any(needmax ? 1 : 0)) {
        const float wm = pdoy_window_nanmax(d, W, lane, nyears, ndoy, tbase, vmap, Tv, T, x, st, cc_);
        if (needmax) r = (double)wm;
      }
      if (COUNT) count_day(d, r, all_valid);
      else if (active) out[((int64_t)j * ndoy + d) * C + c] = r;
    }
  };

  {
    int d0 = blockIdx.y * chunk, d1 = d0 + chunk;
    if (d1 > ndoy) d1 = ndoy;
    // one step: the pairs move up, the old `last` starts the new pair, the gathered day-set becomes `last` and joins it.
    // The next day-set is requested as soon as the registers of the current one are free (after the conversion, before
    // the sorting networks), so that its loads are in flight for a whole step.  The table load behind the rows of the one
    // after that is issued right AFTER the gather and consumed one step later: its s_waitcnt vmcnt(0) placed behind a fresh
    // gather drained the 32 gather loads every step (first versions), placed before it cost a full round trip of its own.
    int rows_next = 0, tbv = -1;  // tbv: table entry behind the rows of day-set d + 2 + half, fetched one step earlier
    auto advance = [&](int d) {
      float key[NYP];
#pragma unroll
      for (int k = 0; k < W - 2; ++k) {
#pragma unroll
        for (int i = 0; i < 16; ++i) pr[k][i] = pr[k + 1][i];
      }
#pragma unroll
      for (int w = 0; w < W - 1; ++w) cnt[w] = cnt[w + 1];
#pragma unroll
      for (int i = 0; i < 16; ++i) pr[W - 2][i] = last[i];
      convert(key, cnt[W - 1]);
      const int rows_nn = pdoy_row_finish(tbv, 0, vmap, Tv, T);
      gather(rows_next);  // (one day-set beyond the chunk's last: a valid or the NaN row, never used)
      rows_next = rows_nn;
      tbv = pdoy_row_fetch(lane, nyears, ndoy, d + 3 + half, tbase);  // arrives behind the gather, used in the next step
      sort_top(key, last);
    };
    // The ring starts empty W - 1 steps before the chunk: those warm-up steps feed the day-sets d0 - half .. d0 + half - 1
    // (pr[0] then holds an incomplete pair; it leaves with the first live step) and select nothing.
#pragma unroll
    for (int i = 0; i < 16; ++i) last[i] = SENT;
#pragma unroll
    for (int k = 0; k < W - 1; ++k) {
#pragma unroll
      for (int i = 0; i < 16; ++i) pr[k][i] = SENT;
    }
#pragma unroll
    for (int w = 0; w < W; ++w) cnt[w] = 0;
    const int dstart = d0 - (W - 1);
    gather(rows_of(dstart + half, 0));
    rows_next = rows_of(dstart + half + 1, 0);
    tbv = pdoy_row_fetch(lane, nyears, ndoy, dstart + half + 2, tbase);
    auto step = [&](int d) {
      const bool live = d >= d0;  // wave-uniform
      if (live) stash_day();
      advance(d);
      tn_merge16<REV>(pr[W - 2], pr[W - 2], last);
      if (live) {
        if (COUNT && d > d0 && pdoy_flag(newseg, d)) flush_all(d - 1);
        if (pdoy_flag(regular, d)) select_and_store(d);
      }
      if (COUNT && d + 1 >= d0 && d + 1 < d1) fetch_day(d + 1);
    };
    int d = dstart;
    for (; d < d1; ++d) step(d);
    if (COUNT) flush_all(d1 - 1);
  }
}

int xh_launch_pdoy_top16(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* d_tb, int nyears,
                         int ndoy, int window, const QTab* d_tab, const int32_t* d_jmap, int nsub, int rev, double* out,
                         const int32_t* d_vmap, int64_t Tv, const uint8_t* d_reg) {
  {
    const int rq = xh_launch_pdoy_quad(ctx, x, T, C, st, d_tb, nyears, ndoy, window, d_tab, d_jmap, nsub, rev, out, d_vmap, Tv, d_reg);
    if (rq != XH_ERR_NOTIMPL) return rq;
  }
  XH_REQUIRE(C < ((int64_t)1 << 29), XH_ERR_LIMIT, "percentile_doy: more than 2^29 columns per call");
  const float* nanrow = nullptr;
  if (int rc = xh_const_rows(ctx, C, &nanrow, nullptr, nullptr)) return rc;
  int chunk = 24;
  if (const char* e = xh_diag_env("XH_PDOY_CHUNK")) chunk = atoi(e) > 0 ? atoi(e) : chunk;
  const dim3 grid((unsigned)cdiv64(C, 64), (unsigned)((ndoy + chunk - 1) / chunk));
#define XH_TOP16_(W, NY, R)                                                                                                 \
  hipLaunchKernelGGL((k_pdoy_top16<W, NY, false, R>), grid, dim3(64), 0, ctx->stream, x, T, C, st, d_tb, nyears, ndoy, chunk, \
                     d_tab, d_jmap, nsub, out, d_vmap, Tv, d_reg, nanrow)
#define XH_TOP16(W, NY)                \
  do {                                 \
    if (rev) XH_TOP16_(W, NY, true);   \
    else XH_TOP16_(W, NY, false);      \
  } while (0)
  if (nyears <= 32) {
    if (window == 3) XH_TOP16(3, 32); else if (window == 5) XH_TOP16(5, 32); else XH_TOP16(7, 32);
  } else {
    if (window == 3) XH_TOP16(3, 64); else if (window == 5) XH_TOP16(5, 64); else XH_TOP16(7, 64);
  }
#undef XH_TOP16_
#undef XH_TOP16
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_launch_pdoy_top16_count(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* d_tb, int nyears,
                               int ndoy, int window, const QTab* d_tab, const int32_t* d_jmap, int rev, const uint8_t* d_reg,
                               int op, const int32_t* d_period, int32_t* cnt_out, int32_t* valid_out,
                               const uint8_t* d_newseg) {
  XH_REQUIRE(C < ((int64_t)1 << 29), XH_ERR_LIMIT, "percentile_doy_count: more than 2^29 columns per call");
  const float* nanrow = nullptr;
  if (int rc = xh_const_rows(ctx, C, &nanrow, nullptr, nullptr)) return rc;
  int chunk = 92;  // 4 chunks of a 365-day year (24 / 46 / 92 / 183 / 365: 26.3 / 25.1 / 24.6 / 24.5 / 24.7 ms at 30 yr x 1440 x 720)
  if (const char* e = xh_diag_env("XH_PDOY_COUNT_CHUNK")) chunk = atoi(e) > 0 ? atoi(e) : chunk;
  const dim3 grid((unsigned)cdiv64(C, 64), (unsigned)((ndoy + chunk - 1) / chunk));
#define XH_TOP16C_(W, NY, R)                                                                                                     \
  hipLaunchKernelGGL((k_pdoy_top16<W, NY, true, R>), grid, dim3(64), 0, ctx->stream, x, T, C, st, d_tb, nyears, ndoy, chunk, d_tab, \
                     d_jmap, 1, (double*)nullptr, (const int32_t*)nullptr, T, d_reg, nanrow, op, d_period, cnt_out, valid_out, d_newseg)
#define XH_TOP16C(W, NY)               \
  do {                                 \
    if (rev) XH_TOP16C_(W, NY, true);  \
    else XH_TOP16C_(W, NY, false);     \
  } while (0)
  if (nyears <= 32) {
    if (window == 3) XH_TOP16C(3, 32); else if (window == 5) XH_TOP16C(5, 32); else XH_TOP16C(7, 32);
  } else {
    if (window == 3) XH_TOP16C(3, 64); else if (window == 5) XH_TOP16C(5, 64); else XH_TOP16C(7, 64);
  }
#undef XH_TOP16C_
#undef XH_TOP16C
  XH_LAUNCH_CHECK();
  return XH_OK;
}
