// pdoy_top.hip — percentile_doy on multi-year base periods, register top-16 variant.
#include <stdlib.h>

#include "f32thr.h"
#include "pdoy.h"

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
// rev mirrors the key order so that the same code serves low percentiles (bottom-16).  Valid keys are never 0.
__device__ __forceinline__ void ce_desc(uint32_t& a, uint32_t& b) {
  uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
  a = hi;
  b = lo;
}

template <int NP>
__device__ __forceinline__ void bitonic_desc(uint32_t (&k)[NP]) {  // full sort, descending
#pragma unroll
  for (int size = 2; size <= NP; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        int j = i ^ stride;
        if (j > i) {
          bool desc = ((i & size) == 0);
          uint32_t a = k[i], b = k[j];
          uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
          k[i] = desc ? hi : lo;
          k[j] = desc ? lo : hi;
        }
      }
    }
  }
}

// 16 keys sorted descending with the 60-comparator, 10-layer optimal network (verified exhaustively with the 0-1
// principle; the bitonic sorter needs 80)
__device__ __forceinline__ void sort16_desc(uint32_t (&k)[16]) {
#define XH_C(i, j) ce_desc(k[i], k[j]);
  XH_C(0, 13) XH_C(1, 12) XH_C(2, 15) XH_C(3, 14) XH_C(4, 8) XH_C(5, 6) XH_C(7, 11) XH_C(9, 10)
  XH_C(0, 5) XH_C(1, 7) XH_C(2, 9) XH_C(3, 4) XH_C(6, 13) XH_C(8, 14) XH_C(10, 15) XH_C(11, 12)
  XH_C(0, 1) XH_C(2, 3) XH_C(4, 5) XH_C(6, 8) XH_C(7, 9) XH_C(10, 11) XH_C(12, 13) XH_C(14, 15)
  XH_C(0, 2) XH_C(1, 3) XH_C(4, 10) XH_C(5, 11) XH_C(6, 7) XH_C(8, 9) XH_C(12, 14) XH_C(13, 15)
  XH_C(1, 2) XH_C(3, 12) XH_C(4, 6) XH_C(5, 7) XH_C(8, 10) XH_C(9, 11) XH_C(13, 14)
  XH_C(1, 4) XH_C(2, 6) XH_C(5, 8) XH_C(7, 10) XH_C(9, 13) XH_C(11, 14)
  XH_C(2, 4) XH_C(3, 6) XH_C(9, 12) XH_C(11, 13)
  XH_C(3, 5) XH_C(6, 8) XH_C(7, 9) XH_C(10, 12)
  XH_C(3, 4) XH_C(5, 6) XH_C(7, 8) XH_C(9, 10) XH_C(11, 12)
  XH_C(6, 7) XH_C(8, 9)
#undef XH_C
}

// t <- the 16 largest of (t u b), sorted descending; t and b sorted descending
__device__ __forceinline__ void merge_top16(uint32_t (&t)[16], const uint32_t (&b)[16]) {
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = t[i] > b[15 - i] ? t[i] : b[15 - i];
#pragma unroll
  for (int stride = 8; stride > 0; stride >>= 1) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if ((i & stride) == 0) ce_desc(t[i], t[i + stride]);
  }
}

// the two smallest of the 16 largest of (t u b): lo16 = 16th largest, hi15 = 15th largest of the union (t, b sorted descending).
// max(t[i], b[15 - i]) is bitonic; the lower half of a half-cleaner holds the smaller half and is bitonic again.
__device__ __forceinline__ void merge_low2(const uint32_t (&t)[16], const uint32_t (&b)[16], uint32_t& lo16, uint32_t& hi15) {
  uint32_t c[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) c[i] = t[i] > b[15 - i] ? t[i] : b[15 - i];
#pragma unroll
  for (int n = 8; n >= 2; n >>= 1) {
#pragma unroll
    for (int i = 0; i < n; ++i) c[i] = c[i] < c[i + n] ? c[i] : c[i + n];
  }
  lo16 = c[0] < c[1] ? c[0] : c[1];
  hi15 = c[0] < c[1] ? c[1] : c[0];
}

// COUNT = true (xh_percentile_doy_count on a multi-year base period): the percentile of doy d is compared with the
// samples of day d of EVERY year and the exceedances are counted per (year, doy) -> period; the (D, C) fp64 table of the
// unfused chain is neither written nor re-read once per year.  One percentile (nsub == 1), regular doys only.
template <int W, int NYP, bool COUNT = false>
__global__ void __launch_bounds__(64, (COUNT || W > 5 || NYP > 32) ? 2 : 3)
k_pdoy_top16(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, const int32_t* __restrict__ tbase, int nyears,
             int ndoy, int chunk, const QTab* __restrict__ qtab, const int32_t* __restrict__ jmap, int nsub,
             double* __restrict__ out, const int32_t* __restrict__ vmap, int64_t Tv, const uint8_t* __restrict__ regular,
             int rev, int op = 0, const int32_t* __restrict__ yd_period = nullptr, int32_t* __restrict__ cnt_out = nullptr,
             int32_t* __restrict__ valid_out = nullptr, const uint8_t* __restrict__ newseg = nullptr) {
  const uint32_t rmask = rev ? 0xFFFFFFFFu : 0u;  // mirrored key order for the bottom-16 case
  const int lane = threadIdx.x;
  int64_t c = (int64_t)blockIdx.x * 64 + lane;
  const bool active = c < C;
  constexpr int half = W / 2;
  constexpr bool FASTSEL = !COUNT;
  const int N = nyears * W;
  uint32_t pr[W - 1][16];  // pr[k]: pair of the day-sets of doys d - half + k and d - half + k + 1
  uint32_t last[16];       // day-set of doy d + half
  int cnt[W];              // valid samples of the day-sets d - half .. d + half
  float raw[NYP];

  const int64_t cc_ = active ? c : C - 1;  // inactive lanes read a valid cell and never store
  auto rows_of = [&](int dn, int off) { return pdoy_row(lane, nyears, ndoy, dn, off, tbase, vmap, Tv, T); };
  auto gather = [&](int rowv) { pdoy_gather<NYP>(raw, rowv, x, st, cc_); };
  // sort the gathered day-set and return its top 16 (in the possibly mirrored order) + valid count
  auto finish = [&](uint32_t (&top)[16], int& nv) {
    uint32_t key[NYP];
    nv = 0;
#pragma unroll
    for (int y = 0; y < NYP; ++y) {
      uint32_t kk = xh_f2key(raw[y]);
      bool ok = kk != 0xFFFFFFFFu;
      nv += ok ? 1 : 0;
      key[y] = ok ? (kk ^ rmask) : 0u;  // NaN / padding -> 0 = smallest
    }
    // top 16 of the NYP keys: sort blocks of 16 with the optimal network, combine with half-merges
    // (NYP = 32: 60 + 60 + 16 + 32 = 168 comparator-equivalents instead of 240 for a full bitonic-32)
    uint32_t blk[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) top[i] = key[i];
    sort16_desc(top);
#pragma unroll
    for (int b = 1; b < NYP / 16; ++b) {
#pragma unroll
      for (int i = 0; i < 16; ++i) blk[i] = key[b * 16 + i];
      sort16_desc(blk);
      merge_top16(top, blk);
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
    if constexpr (COUNT) pdoy_gather<NYP>(xv, rows_of(d, 0), x, st, cc_);
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
    uint32_t t16[16];
    int n = cnt[0];
#pragma unroll
    for (int i = 0; i < 16; ++i) t16[i] = pr[0][i];
#pragma unroll
    for (int k = 2; k <= W - 3; k += 2) merge_top16(t16, pr[k]);
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
          uint32_t k16, k15;
          merge_low2(t16, last, k16, k15);
          const float left = xh_key2f((plo == 15 ? k16 : k15) ^ rmask), right = xh_key2f((phi == 15 ? k16 : k15) ^ rmask);
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
    merge_top16(t16, last);
    auto get = [&](int idx) -> float {
      uint32_t g = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) g |= (i == idx) ? t16[i] : 0u;
      return xh_key2f(g ^ rmask);
    };
    for (int jj = 0; jj < nsub; ++jj) {
      const int j = jmap[jj];
      const QTab e = qtab[j * (N + 1) + n];
      double r = xh_nan64();
      if (e.lo >= 0) {
        // position in the (mirrored) descending top-16: rev -> rank from the bottom, else rank from the top
        const int plo = rev ? e.lo : (n - 1 - e.lo), phi = rev ? e.hi : (n - 1 - e.hi);
        float left = get(plo), right = get(phi);
        float diff = right - left;
        r = (double)left + (double)diff * e.gamma;
        if (e.gamma >= 0.5) r = (double)right - (double)diff * (1.0 - e.gamma);
        if (r != r && n > 0) r = (double)get(rev ? (n - 1 < 15 ? n - 1 : 15) : 0);  // +-inf: nanmax fallback (utl:552-554)
      }
      if (COUNT) count_day(d, r, all_valid);
      else if (active) out[((int64_t)j * ndoy + d) * C + c] = r;
    }
  };

  {
    int d0 = blockIdx.y * chunk, d1 = d0 + chunk;
    if (d1 > ndoy) d1 = ndoy;
    // one step: the pairs move up, the old `last` starts the new pair, the gathered day-set becomes `last` and joins it
    auto advance = [&]() {
#pragma unroll
      for (int k = 0; k < W - 2; ++k) {
#pragma unroll
        for (int i = 0; i < 16; ++i) pr[k][i] = pr[k + 1][i];
      }
#pragma unroll
      for (int w = 0; w < W - 1; ++w) cnt[w] = cnt[w + 1];
#pragma unroll
      for (int i = 0; i < 16; ++i) pr[W - 2][i] = last[i];
      finish(last, cnt[W - 1]);
    };
    // prologue: the state of doy d0 - 1 = day-sets d0 - half .. d0 - 1 + half fed into an empty ring (pr[0] then holds an
    // incomplete pair; it leaves with the first step)
#pragma unroll
    for (int i = 0; i < 16; ++i) last[i] = 0u;
#pragma unroll
    for (int k = 0; k < W - 1; ++k) {
#pragma unroll
      for (int i = 0; i < 16; ++i) pr[k][i] = 0u;
    }
#pragma unroll
    for (int w = 0; w < W; ++w) cnt[w] = 0;
#pragma unroll
    for (int w = 1; w < W; ++w) {
      gather(rows_of(d0 - 1 - half + w, 0));
      advance();
      merge_top16(pr[W - 2], last);
    }
    gather(rows_of(d0 + half, 0));
    int rows_next = rows_of(d0 + half + 1, 0);
    fetch_day(d0);
    for (int d = d0; d < d1; ++d) {
      stash_day();
      advance();
      if (d + 1 < d1) {
        gather(rows_next);
        rows_next = rows_of(d + 2 + half, 0);
      }
      merge_top16(pr[W - 2], last);
      if (COUNT && d > d0 && newseg[d]) flush_all(d - 1);
      if (regular[d]) select_and_store(d);
      if (COUNT && d + 1 < d1) fetch_day(d + 1);
    }
    if (COUNT) flush_all(d1 - 1);
  }
}

int xh_launch_pdoy_top16(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* d_tb, int nyears,
                         int ndoy, int window, const QTab* d_tab, const int32_t* d_jmap, int nsub, int rev, double* out,
                         const int32_t* d_vmap, int64_t Tv, const uint8_t* d_reg) {
  const int chunk = 24;
  const dim3 grid((unsigned)cdiv64(C, 64), (unsigned)((ndoy + chunk - 1) / chunk));
#define XH_TOP16(W, NY)                                                                                                   \
  hipLaunchKernelGGL((k_pdoy_top16<W, NY>), grid, dim3(64), 0, ctx->stream, x, T, C, st, d_tb, nyears, ndoy, chunk, d_tab,  \
                     d_jmap, nsub, out, d_vmap, Tv, d_reg, rev)
  if (nyears <= 32) {
    if (window == 3) XH_TOP16(3, 32); else if (window == 5) XH_TOP16(5, 32); else XH_TOP16(7, 32);
  } else {
    if (window == 3) XH_TOP16(3, 64); else if (window == 5) XH_TOP16(5, 64); else XH_TOP16(7, 64);
  }
#undef XH_TOP16
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_launch_pdoy_top16_count(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* d_tb, int nyears,
                               int ndoy, int window, const QTab* d_tab, const int32_t* d_jmap, int rev, const uint8_t* d_reg,
                               int op, const int32_t* d_period, int32_t* cnt_out, int32_t* valid_out,
                               const uint8_t* d_newseg) {
  int chunk = 92;  // 4 chunks of a 365-day year (24 / 46 / 92 / 183 / 365: 26.3 / 25.1 / 24.6 / 24.5 / 24.7 ms at 30 yr x 1440 x 720)
  if (const char* e = xh_diag_env("XH_PDOY_COUNT_CHUNK")) chunk = atoi(e) > 0 ? atoi(e) : chunk;
  const dim3 grid((unsigned)cdiv64(C, 64), (unsigned)((ndoy + chunk - 1) / chunk));
#define XH_TOP16C(W, NY)                                                                                                       \
  hipLaunchKernelGGL((k_pdoy_top16<W, NY, true>), grid, dim3(64), 0, ctx->stream, x, T, C, st, d_tb, nyears, ndoy, chunk, d_tab, \
                     d_jmap, 1, (double*)nullptr, (const int32_t*)nullptr, T, d_reg, rev, op, d_period, cnt_out, valid_out, d_newseg)
  if (nyears <= 32) {
    if (window == 3) XH_TOP16C(3, 32); else if (window == 5) XH_TOP16C(5, 32); else XH_TOP16C(7, 32);
  } else {
    if (window == 3) XH_TOP16C(3, 64); else if (window == 5) XH_TOP16C(5, 64); else XH_TOP16C(7, 64);
  }
#undef XH_TOP16C
  XH_LAUNCH_CHECK();
  return XH_OK;
}
