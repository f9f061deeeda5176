// quantile.hip — NaN-aware Hyndman-Fan quantiles per cell: calc_perc / percentile_doy / doy re-gridding.
//
// Reference: core/utils.py:370-557 (_compute_virtual_index, _get_indexes, _get_gamma, _linear_interpolation,
// _nan_quantile) and core/calendar.py:395-494 (percentile_doy), 690-726 (_interpolate_doy_calendar).
//
// Sample set of (doy row d, cell c): for every year y with tb = tbase[y*ndoy + d] >= 0 the `window` values
// x[tb - window/2 + k], k = 0..window-1 (NaN outside [0, T)); years lacking the day contribute NaNs
// (rolling(center=True, min_periods=1).construct + unstack/stack, cal:448-458).  N = nyears * window samples.
// Kernels, chosen on the host (pdoy_impl):
//   one contiguous year, window 3/5/7  : k_pdoy_slide   sliding register ring, every row read once (the tx90p benchmark);
//                                         COUNT variant fuses the exceedance count (xh_percentile_doy_count)
//   N <= 32 samples                    : k_pdoy_reg     gather into registers, bitonic network on uint32 keys
//   multi-year, regular doys           : k_pdoy_quad    (pdoy_quad.hip, window 5) / k_pdoy_top16 (pdoy_top.hip, windows 3, 7):
//                                         register top-16 of the W day-sets, for percentiles whose order statistics lie
//                                         within the 16 largest / smallest samples
//                                        k_pdoy_walk    (pdoy_walk.hip) the other percentiles up to 32 years: sorted day-set
//                                         lists in LDS and a split that walks from day to day
//                                        k_pdoy_merge   per-day sorted lists in a compact LDS ring + W-way tail merge (more
//                                         than 32 years; OFFSET variant: exact window lists for irregular doys, e.g. Feb 29)
//   anything else (N <= 4096)          : k_pdoy_lds     lane-private LDS column, bitonic network in LDS (64 cells per wave
//                                                         up to 512 samples, 32 / 16 / 8 beyond)
// Time-major layout, one lane per cell (VEC cells in the sliding kernel).
#include <stdlib.h>

#include <type_traits>

#include "pdoy.h"

// Hyndman-Fan quantile from a sorted sample (ascending, NaN last).  `get(i)` returns sorted element i as float.
// L = total slots, n = non-NaN count.  Follows utl:494-557 step by step.
template <typename Getter>
__device__ __forceinline__ double xh_hf_quantile(int L, int n, double q, double alpha, double beta, Getter get) {
  if (L == 1) return (double)get(0);  // utl:508-510
  if (n < 2) {                        // utl:519-523: vi = NaN -> last slot (NaN) -> nanmax fallback (utl:552-554)
    return n == 1 ? (double)get(0) : xh_nan64();
  }
  double nn = (double)n;
  // utl:395  n * quantiles + (alpha + quantiles * (1 - alpha - beta)) - 1   (no FMA: built with -ffp-contract=off)
  double vi = nn * q + (alpha + q * (1.0 - alpha - beta)) - 1.0;
  if (vi >= nn - 1.0) return (double)get(n - 1);  // utl:443-447 + nanmax fallback
  if (vi < 0.0) {  // utl:449-452: both neighbours = slot 0, then the lerp of utl:486: inf - inf = NaN -> nanmax (utl:552-554)
    const float v0 = get(0);
    return (v0 - v0 != 0.0f) ? (double)get(n - 1) : (double)v0;
  }
  double prev = floor(vi);
  int ip = (int)prev;
  double gamma = vi - prev;                // utl:412
  float left = get(ip), right = get(ip + 1);
  float diff = right - left;               // utl:486 np.subtract in the data dtype (fp32)
  double r = (double)left + (double)diff * gamma;
  if (gamma >= 0.5) r = (double)right - (double)diff * (1.0 - gamma);  // utl:488
  if (r != r) r = (double)get(n - 1);      // utl:552-554
  return r;
}

// ---- register path --------------------------------------------------------------------------------
template <int NMAX>
__device__ __forceinline__ void bitonic_regs(uint32_t (&k)[NMAX]) {
#pragma unroll
  for (int size = 2; size <= NMAX; size <<= 1) {
#pragma unroll
    for (int stride = size >> 1; stride > 0; stride >>= 1) {
#pragma unroll
      for (int i = 0; i < NMAX; ++i) {
        int j = i ^ stride;
        if (j > i) {
          bool up = ((i & size) == 0);
          uint32_t a = k[i], b = k[j];
          uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
          k[i] = up ? lo : hi;
          k[j] = up ? hi : lo;
        }
      }
    }
  }
}

template <int NMAX, int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_pdoy_reg(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, const int32_t* __restrict__ tbase, int nyears,
           int ndoy, int window, const double* __restrict__ qs, int nper, double alpha, double beta,
           double* __restrict__ out) {
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const int half = window / 2;
  const int N = nyears * window;
  for (int d = blockIdx.y; d < ndoy; d += gridDim.y) {
    uint32_t key[VEC][NMAX];
    int nvalid[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) nvalid[v] = 0;
    // all the loads first, unconditional with a clamped row (a load inside a conditional is followed by s_waitcnt
    // vmcnt(0): the gathers would be serialised), conversion and masking afterwards
    VecF<VEC> raw[NMAX];
    bool okm[NMAX];
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
      // slot i -> (year y, window offset k)
      int y = i / window, k = i - y * window;
      bool inb = i < N;
      int64_t tb = inb ? (int64_t)tbase[(int64_t)y * ndoy + d] : -1;
      int64_t t = tb - half + k;
      okm[i] = inb && tb >= 0 && t >= 0 && t < T;
      raw[i] = xh_load<VEC>(x + (okm[i] ? t : (int64_t)0) * st + c);
    }
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const uint32_t kk = okm[i] ? xh_f2key(raw[i].v[v]) : 0xFFFFFFFFu;
        key[v][i] = kk;
        nvalid[v] += (kk != 0xFFFFFFFFu) ? 1 : 0;
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      bitonic_regs<NMAX>(key[v]);
      auto get = [&](int idx) -> float {  // OR of masked values: keeps the array in registers (see k_pdoy_slide)
        uint32_t r = 0;
#pragma unroll
        for (int i = 0; i < NMAX; ++i) r |= (i == idx) ? key[v][i] : 0u;
        return xh_key2f(r);
      };
      for (int j = 0; j < nper; ++j) {
        double r = xh_hf_quantile(N, nvalid[v], qs[j], alpha, beta, get);
        out[((int64_t)j * ndoy + d) * C + c + v] = r;
      }
    }
  }
}

// ---- sliding-window path (one contiguous year: the 365 x 1440 x 720 benchmark shape) -----------------
// With a single year of contiguous days the sample set of doy d+1 is the set of doy d shifted by one row, so a
// lane keeps the W keys of its VEC cells in registers, loads ONE new row per doy and re-sorts a copy: x is read
// once from HBM (plus a (W-1)-row halo per doy chunk), the (D, C) fp64 result is written once.
// The Hyndman-Fan index arithmetic depends only on (percentile, valid count n): the host evaluates it in fp64
// exactly as utl:395/417-461 do and ships a tiny table (lo, hi, gamma) per (j, n), staged in LDS.
__device__ __forceinline__ void ce(uint32_t& a, uint32_t& b) {
  uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
  a = lo;
  b = hi;
}

template <int W>
__device__ __forceinline__ void sort_small(uint32_t (&k)[W]) {
  if (W == 1) return;
  if (W == 2) { ce(k[0], k[1 % W]); return; }
  if (W == 3) { ce(k[0], k[1 % W]); ce(k[1 % W], k[2 % W]); ce(k[0], k[1 % W]); return; }
  if (W == 5) {  // optimal 9-comparator network
    ce(k[0], k[1 % W]); ce(k[3 % W], k[4 % W]); ce(k[2 % W], k[4 % W]); ce(k[2 % W], k[3 % W]); ce(k[0], k[3 % W]);
    ce(k[0], k[2 % W]); ce(k[1 % W], k[4 % W]); ce(k[1 % W], k[3 % W]); ce(k[1 % W], k[2 % W]);
    return;
  }
  // generic: odd-even transposition (W rounds), fine for the small windows this path accepts
#pragma unroll
  for (int r = 0; r < W; ++r) {
#pragma unroll
    for (int i = (r & 1); i + 1 < W; i += 2) ce(k[i], k[i + 1]);
  }
}

// COUNT = true (xh_percentile_doy_count): the percentile of doy d is compared at once with the day's own value and the
// exceedances are counted per period — the (D, C) fp64 table of the unfused tx90p chain (2/3 of its HBM traffic) is
// never written; partial counts of a doy chunk are combined with global atomics (nper must be 1).
//
// The kernel is VALU-bound (PMC: 331 VALU wave-instructions per 4-cell doy step before this layout), so the window is
// a RING: the day loop is unrolled by W, the row of step k lands in the static slot (k + W - 1) % W — the slot of the
// day that leaves — and nothing is ever shifted (sorting and min/max do not care about the order).  `fast` (host:
// every percentile clips to the window maximum (1) / minimum (2) when the window is full, e.g. per = 90 with 5
// samples, utl:443-452) selects a path without the sorting network when no lane of the wave holds a NaN.
template <int W, int VEC, bool COUNT = false>
__global__ void __launch_bounds__(XH_BLOCK)
k_pdoy_slide(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, int64_t t_first, int ndoy, int chunk,
             const QTab* __restrict__ qtab, int nper, double* __restrict__ out, int fast, int op = 0,
             const int32_t* __restrict__ doy_period = nullptr, int32_t* __restrict__ cnt_out = nullptr,
             int32_t* __restrict__ valid_out = nullptr) {
  __shared__ QTab s_tab[8 * (W + 1)];
  for (int i = threadIdx.x; i < nper * (W + 1); i += XH_BLOCK) s_tab[i] = qtab[i];
  __syncthreads();
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  constexpr int half = W / 2;
  int d0 = blockIdx.y * chunk, d1 = d0 + chunk;
  if (d1 > ndoy) d1 = ndoy;
  uint32_t win[VEC][W];
  int n[VEC];  // valid keys currently in the window (maintained incrementally)
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    n[v] = 0;
#pragma unroll
    for (int k = 0; k < W; ++k) win[v][k] = 0xFFFFFFFFu;
  }
  // rows are fetched unconditionally (clamped row index: a load inside a conditional is followed by s_waitcnt
  // vmcnt(0)) and one doy AHEAD of their use, so the row of doy d+1 is in flight while doy d is selected and stored
  auto fetch = [&](int64_t t) -> VecF<VEC> {
    const int64_t tc = t < 0 ? 0 : (t >= T ? T - 1 : t);
    return xh_load<VEC>(x + tc * st + c);
  };
  int ccnt[VEC], cval[VEC], cper = -1;
#pragma unroll
  for (int v = 0; v < VEC; ++v) { ccnt[v] = 0; cval[v] = 0; }
  auto flush_counts = [&]() {
    if (cper >= 0) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        if (ccnt[v]) atomicAdd(&cnt_out[(int64_t)cper * C + c + v], ccnt[v]);
        if (valid_out && cval[v]) atomicAdd(&valid_out[(int64_t)cper * C + c + v], cval[v]);
        ccnt[v] = 0; cval[v] = 0;
      }
    }
  };
  // result of doy d for percentile j: stored, or compared with the day's own value (window centre) and counted
  auto emit = [&](int d, int j, const double (&r)[VEC], const uint32_t (&centre)[VEC]) {
    if (COUNT) {
      const int p = doy_period[d];
      if (p != cper) { flush_counts(); cper = p; }
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const float xv = xh_key2f(centre[v]);
        ccnt[v] += xh_cmp_f64((double)xv, op, r[v]) ? 1 : 0;
        cval[v] += (xv == xv) ? 1 : 0;
      }
    } else {
      double* optr = out + ((int64_t)j * ndoy + d) * C + c;
      if (VEC == 4) {
        *reinterpret_cast<double2*>(optr) = make_double2(r[0], r[1 % VEC]);
        *reinterpret_cast<double2*>(optr + 2) = make_double2(r[2 % VEC], r[3 % VEC]);
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) optr[v] = r[v];
      }
    }
  };
  // prologue: rows d0-half .. d0+half-1 into slots 0 .. W-2
  {
    VecF<VEC> pre[W > 1 ? W - 1 : 1];
#pragma unroll
    for (int k = 0; k < W - 1; ++k) pre[k] = fetch(t_first + d0 - half + k);
#pragma unroll
    for (int k = 0; k < W - 1; ++k) {
      const int64_t t = t_first + d0 - half + k;
      const bool inside = t >= 0 && t < T;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const uint32_t kk = inside ? xh_f2key(pre[k].v[v]) : 0xFFFFFFFFu;
        win[v][k] = kk;
        n[v] += (kk != 0xFFFFFFFFu) ? 1 : 0;
      }
    }
  }
  VecF<VEC> nxt = fetch(t_first + d0 - half + (W - 1));

  // one doy; U = (d - d0) % W is static
  auto step = [&](int d, auto UC) {
    constexpr int U = decltype(UC)::value;
    constexpr int SLOT = (U + W - 1) % W;   // slot of the row entering (== slot of the row leaving)
    constexpr int CEN = (U + half) % W;     // slot of day d itself
    const VecF<VEC> cur = nxt;
    nxt = fetch(t_first + d + 1 - half + (W - 1));  // (one clamped extra row at the end of the chunk)
    const int64_t tin = t_first + d - half + (W - 1);
    const bool inside = tin >= 0 && tin < T;
    bool full = true;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const uint32_t kk = inside ? xh_f2key(cur.v[v]) : 0xFFFFFFFFu;
      n[v] += ((kk != 0xFFFFFFFFu) ? 1 : 0) - ((win[v][SLOT] != 0xFFFFFFFFu) ? 1 : 0);
      win[v][SLOT] = kk;
      full &= (n[v] == W);
    }
    uint32_t centre[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) centre[v] = win[v][CEN];
    // Wave-uniform fast paths: no NaN in any window of the wave -> the table entry (lo, hi, gamma) is the same for
    // all lanes and selection needs no per-lane index.
    const bool uniform = __all(full ? 1 : 0) != 0;
    if (uniform && fast != 0) {
      // every percentile clips to the sample maximum / minimum (utl:443-452): no sort at all
      double r[VEC];
      float rf[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        uint32_t m = win[v][0];
#pragma unroll
        for (int k = 1; k < W; ++k) m = (fast == 2) ? (win[v][k] < m ? win[v][k] : m) : (win[v][k] > m ? win[v][k] : m);
        rf[v] = xh_key2f(m);
        r[v] = (double)rf[v];
      }
      // An infinite MINIMUM is not the answer: the lerp between slot 0 and itself is inf - inf = NaN and the reference
      // then takes the window's nanmax (utl:552-554) — the general path below does that (an infinite maximum is its own
      // nanmax).
      bool odd = false;
      if (fast == 2) {
#pragma unroll
        for (int v = 0; v < VEC; ++v) odd |= (rf[v] - rf[v]) != 0.0f;
      }
      if (!(fast == 2 && __any(odd ? 1 : 0))) {
        if (COUNT) {
          // the percentile IS a float32 sample here, so the fp64 compare of the general path is exactly an fp32 compare
          // (no conversions, mask-form operator); the window is full: every value is valid
          const int p = doy_period[d];
          if (p != cper) { flush_counts(); cper = p; }
#pragma unroll
          for (int v = 0; v < VEC; ++v) {
            ccnt[v] += xh_cmp_f32(xh_key2f(centre[v]), op, rf[v]) ? 1 : 0;
            cval[v] += 1;
          }
          return;
        }
        for (int j = 0; j < nper; ++j) emit(d, j, r, centre);
        return;
      }
    }
    uint32_t s[VEC][W];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
#pragma unroll
      for (int k = 0; k < W; ++k) s[v][k] = win[v][k];
      sort_small<W>(s[v]);
    }
    for (int j = 0; j < nper; ++j) {
      double r[VEC];
      if (uniform) {
        const QTab e = s_tab[j * (W + 1) + W];
        const int lo = __builtin_amdgcn_readfirstlane(e.lo), hi = __builtin_amdgcn_readfirstlane(e.hi);
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          uint32_t kl = s[v][0], kh = s[v][0];
#pragma unroll
          for (int i = 1; i < W; ++i) {  // lo / hi are wave-uniform: these selects are scalar-predicated moves
            kl = (i == lo) ? s[v][i] : kl;
            kh = (i == hi) ? s[v][i] : kh;
          }
          float left = xh_key2f(kl), right = xh_key2f(kh);
          float diff = right - left;
          double rr = (double)left + (double)diff * e.gamma;
          if (e.gamma >= 0.5) rr = (double)right - (double)diff * (1.0 - e.gamma);
          if (rr != rr) rr = (double)xh_key2f(s[v][W - 1]);
          r[v] = rr;
        }
      } else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          // register-array select written as an OR of masked values: a select chain gets folded back into a
          // dynamically indexed (scratch) array by the optimizer
          auto get = [&](int idx) -> float {
            uint32_t g = 0;
#pragma unroll
            for (int i = 0; i < W; ++i) g |= (i == idx) ? s[v][i] : 0u;
            return xh_key2f(g);
          };
          QTab e = s_tab[j * (W + 1) + n[v]];
          float left = get(e.lo), right = get(e.hi);
          float diff = right - left;
          double rr = (double)left + (double)diff * e.gamma;
          if (e.gamma >= 0.5) rr = (double)right - (double)diff * (1.0 - e.gamma);
          if (rr != rr && n[v] > 0 && W > 1) rr = (double)get(n[v] - 1);
          r[v] = (e.lo < 0) ? xh_nan64() : rr;
        }
      }
      emit(d, j, r, centre);
    }
  };

  for (int d = d0; d < d1; d += W) {
    step(d, std::integral_constant<int, 0>{});
    if (W > 1 && d + 1 < d1) step(d + 1, std::integral_constant<int, 1 % W>{});
    if (W > 2 && d + 2 < d1) step(d + 2, std::integral_constant<int, 2 % W>{});
    if (W > 3 && d + 3 < d1) step(d + 3, std::integral_constant<int, 3 % W>{});
    if (W > 4 && d + 4 < d1) step(d + 4, std::integral_constant<int, 4 % W>{});
    if (W > 5 && d + 5 < d1) step(d + 5, std::integral_constant<int, 5 % W>{});
    if (W > 6 && d + 6 < d1) step(d + 6, std::integral_constant<int, 6 % W>{});
  }
  if (COUNT) flush_counts();
}

// Host side of the table: one entry per (percentile j, valid count n), mirroring xh_hf_quantile.
static void build_qtab(int L, const double* qs, int nper, double alpha, double beta, QTab* tab) {
  for (int j = 0; j < nper; ++j) {
    for (int n = 0; n <= L; ++n) {
      QTab e;
      e.gamma = 0.0;
      if (L == 1) { e.lo = e.hi = 0; }
      else if (n < 2) { e.lo = e.hi = (n == 1 ? 0 : -1); }
      else {
        double nn = (double)n, q = qs[j];
        volatile double a = nn * q;
        volatile double b = q * (1.0 - alpha - beta);
        volatile double cc = alpha + b;
        volatile double s = a + cc;
        double vi = s - 1.0;
        if (vi >= nn - 1.0) { e.lo = e.hi = n - 1; }
        else if (vi < 0.0) { e.lo = e.hi = 0; }
        else {
          double prev = __builtin_floor(vi);
          e.lo = (int)prev;
          e.hi = e.lo + 1;
          e.gamma = vi - prev;
        }
      }
      tab[j * (L + 1) + n] = e;
    }
  }
}

// ---- LDS path -------------------------------------------------------------------------------------
// One wave (64 lanes) per block; lane l owns column l of an [NP][cw] uint32 LDS array (conflict-free: the bank
// is the lane).  All lanes execute the same compare-exchange sequence on their own column, so no barriers.
// cw = cells per wave: 64 up to 512 samples per cell; 32 / 16 / 8 (the other lanes idle) for 1024 / 2048 / 4096 samples
// (e.g. window 31 over 30 years = 930): the exact, slow path of last resort.
__global__ void __launch_bounds__(64)
k_pdoy_lds(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, const int32_t* __restrict__ tbase, int nyears,
           int ndoy, int window, int NP, int cw, const double* __restrict__ qs, int nper, double alpha, double beta,
           double* __restrict__ out, const int32_t* __restrict__ vmap, int64_t Tv, const int32_t* __restrict__ doy_list,
           int ndl) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x;
  if (lane >= cw) return;  // (no barriers in this kernel)
  int64_t c = (int64_t)blockIdx.x * cw + lane;
  const bool active = c < C;
  const int half = window / 2;
  const int N = nyears * window;
  uint32_t* col = lds + lane;  // element i at col[i * cw]
  const int nd = doy_list ? ndl : ndoy;
  for (int di = blockIdx.y; di < nd; di += gridDim.y) {
    const int d = doy_list ? doy_list[di] : di;
    int nvalid = 0;
    for (int y = 0; y < nyears; ++y) {
      int64_t tb = (int64_t)tbase[(int64_t)y * ndoy + d];
      for (int k = 0; k < window; ++k) {
        int64_t t = tb - half + k;  // virtual time index
        float v = xh_nan32();
        if (tb >= 0 && t >= 0 && t < Tv) {
          int64_t tp = vmap ? (int64_t)vmap[t] : t;  // physical row (-1: a day the replica does not have)
          if (active && tp >= 0 && tp < T) v = x[tp * st + c];
        }
        nvalid += (v == v) ? 1 : 0;
        col[(y * window + k) * cw] = xh_f2key(v);
      }
    }
    for (int i = N; i < NP; ++i) col[i * cw] = 0xFFFFFFFFu;
    // bitonic sort of NP keys, lane-private
    for (int size = 2; size <= NP; size <<= 1) {
      for (int stride = size >> 1; stride > 0; stride >>= 1) {
        for (int i = 0; i < NP; ++i) {
          int j = i ^ stride;
          if (j > i) {
            bool up = ((i & size) == 0);
            uint32_t a = col[i * cw], b = col[j * cw];
            uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
            col[i * cw] = up ? lo : hi;
            col[j * cw] = up ? hi : lo;
          }
        }
      }
    }
    if (active) {
      auto get = [&](int idx) -> float { return xh_key2f(col[idx * cw]); };
      for (int j = 0; j < nper; ++j) {
        double r = xh_hf_quantile(N, nvalid, qs[j], alpha, beta, get);
        out[((int64_t)j * ndoy + d) * C + c] = r;
      }
    }
  }
}

// ---- multi-year path: per-day sorted lists + W-way tail merge ---------------------------------------------
// For a base period of NY years the sample set of a day of year is the union over the W window days of "that
// calendar day in every year".  Consecutive doys share W-1 of those day-sets, so each day-set is gathered and
// sorted ONCE (register bitonic on NYP <= 64 keys), kept in an LDS ring of W lists per lane, and the two order
// statistics of a percentile are found by popping from the W sorted lists (from the top or the bottom, whichever
// is closer): ~min(k, N-k) * 25 ops instead of a (N log^2 N) sort of all W*NY samples per doy.
// Days where the union-of-lists identity does not hold (the first/last W/2 days of the year, doy 365/366 in mixed
// leap/non-leap periods) are flagged by the host (regular[d] == 0) and computed by k_pdoy_lds instead.
// One wave per workgroup; lane = cell.  ring[slot][i][lane]; cnt[slot][lane] = number of non-NaN keys of the list.
// Ring layout (compact): only the KT largest and KB smallest valid keys of every day-set can ever be popped (the
// host derives KT / KB from the percentiles: e.g. per = 90, N = 150 -> KT = 16, KB = 0..1), so the LDS footprint per
// wave is W * (KT + KB + 1) * 256 B instead of W * NYP * 256 B and more waves fit on a CU.
//   ringT[w][j][lane] : j-th largest valid key of list w     ringB[w][i][lane] : i-th smallest     cnt[w][lane]
// OFFSET = false: lists are day-sets shared between neighbouring doys (one new list per doy, chunked over blockIdx.y).
// OFFSET = true : doys from `doy_list` whose window does not decompose into day-sets; the W lists are built from the
//                 window offsets themselves (list k = { x[tbase[y][d] - W/2 + k] : y }), no reuse, always exact.
template <int W, int NYP, bool OFFSET>
__global__ void __launch_bounds__(64)
k_pdoy_merge(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, const int32_t* __restrict__ tbase, int nyears,
             int ndoy, int chunk, const QTab* __restrict__ qtab, const int32_t* __restrict__ jmap, int nper,
             double* __restrict__ out, const int32_t* __restrict__ vmap, int64_t Tv, const uint8_t* __restrict__ regular,
             const int32_t* __restrict__ doy_list, int ndl, int KT, int KB, int abl) {
  extern __shared__ uint32_t lds[];
  const int lane = threadIdx.x;
  int64_t c = (int64_t)blockIdx.x * 64 + lane;
  const bool active = c < C;
  constexpr int half = W / 2;
  const int N = nyears * W;
  uint32_t* ringT = lds;                          // [W][KT][64]
  uint32_t* ringB = ringT + W * KT * 64;          // [W][KB][64]
  uint32_t* cnt = ringB + W * KB * 64;            // [W][64]

  // gather one list (virtual day index per year given by `vidx(y)`) into registers ...
  float raw[NYP];
  const int64_t cc = active ? c : C - 1;  // inactive lanes read a valid cell and never store
  auto rows_of = [&](int dn, int off) { return pdoy_row(lane, nyears, ndoy, dn, off, tbase, vmap, Tv, T); };
  auto gather = [&](int rowv) { pdoy_gather<NYP>(raw, rowv, x, st, cc); };
  // ... then sort it and store its two ends in ring slot `slot`
  auto finish = [&](int slot) {
    uint32_t key[NYP];
    int nv = 0;
#pragma unroll
    for (int y = 0; y < NYP; ++y) {
      key[y] = xh_f2key(raw[y]);
      nv += (key[y] != 0xFFFFFFFFu) ? 1 : 0;
    }
    if (!(abl & 1)) bitonic_regs<NYP>(key);
#pragma unroll
    for (int i = 0; i < NYP; ++i) {
      int rt = nv - 1 - i;  // rank from the top of sorted slot i
      if (i < nv && rt < KT) ringT[(slot * KT + rt) * 64 + lane] = key[i];
      if (i < nv && i < KB) ringB[(slot * KB + i) * 64 + lane] = key[i];
    }
    cnt[slot * 64 + lane] = (uint32_t)nv;
  };
  auto build = [&](int slot, int rowv) {
    gather(rowv);
    finish(slot);
  };

  auto select_and_store = [&](int d) {
    int n = 0, cw[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
      cw[w] = (int)cnt[w * 64 + lane];
      n += cw[w];
    }
    for (int jj = 0; jj < nper; ++jj) {
      const int j = jmap[jj];
      QTab e = qtab[j * (N + 1) + n];
      double r = xh_nan64();
      if (e.lo >= 0) {
        const bool from_top = (n - 1 - e.hi) < e.lo;  // pops needed from either end to reach ranks lo <= hi
        uint32_t klo = 0, khi = 0;
        if (from_top) {
          int pt[W];
          uint32_t h[W];
#pragma unroll
          for (int w = 0; w < W; ++w) {
            pt[w] = 0;
            h[w] = (cw[w] > 0 && KT > 0) ? ringT[(w * KT) * 64 + lane] : 0u;
          }
          const int npop = n - e.lo;  // ranks n-1 ... lo
          for (int it = 0; it < npop; ++it) {
            uint32_t m = h[0];
#pragma unroll
            for (int w = 1; w < W; ++w) m = h[w] > m ? h[w] : m;
            if (it == n - 1 - e.hi) khi = m;
            klo = m;  // the last pop is rank lo
            bool done = false;
            int wsel = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) {
              bool is = !done && h[w] == m && pt[w] < cw[w];
              wsel = is ? w : wsel;
              done |= is;
            }
            int psel = 0, csel = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) {
              pt[w] += (w == wsel) ? 1 : 0;
              psel = (w == wsel) ? pt[w] : psel;
              csel = (w == wsel) ? cw[w] : csel;
            }
            bool more = psel < csel && psel < KT;
            uint32_t nh = more ? ringT[(wsel * KT + psel) * 64 + lane] : 0u;
#pragma unroll
            for (int w = 0; w < W; ++w) h[w] = (w == wsel) ? nh : h[w];
          }
        } else {
          int pb[W];
          uint32_t h[W];
#pragma unroll
          for (int w = 0; w < W; ++w) {
            pb[w] = 0;
            h[w] = cw[w] > 0 ? (KB > 0 ? ringB[(w * KB) * 64 + lane] : ringT[(w * KT + cw[w] - 1) * 64 + lane]) : 0xFFFFFFFFu;
          }
          const int npop = e.hi + 1;  // ranks 0 ... hi
          for (int it = 0; it < npop; ++it) {
            uint32_t m = h[0];
#pragma unroll
            for (int w = 1; w < W; ++w) m = h[w] < m ? h[w] : m;
            if (it == e.lo) klo = m;
            khi = m;
            bool done = false;
            int wsel = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) {
              bool is = !done && h[w] == m && pb[w] < cw[w];
              wsel = is ? w : wsel;
              done |= is;
            }
            int psel = 0, csel = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) {
              pb[w] += (w == wsel) ? 1 : 0;
              psel = (w == wsel) ? pb[w] : psel;
              csel = (w == wsel) ? cw[w] : csel;
            }
            bool more = psel < csel && (KB > 0 ? psel < KB : true);
            uint32_t nh = more ? (KB > 0 ? ringB[(wsel * KB + psel) * 64 + lane] : ringT[(wsel * KT + (csel - 1 - psel)) * 64 + lane])
                               : 0xFFFFFFFFu;
#pragma unroll
            for (int w = 0; w < W; ++w) h[w] = (w == wsel) ? nh : h[w];
          }
        }
        float left = xh_key2f(klo), right = xh_key2f(khi);
        float diff = right - left;
        r = (double)left + (double)diff * e.gamma;
        if (e.gamma >= 0.5) r = (double)right - (double)diff * (1.0 - e.gamma);
        if (r != r && n > 0) {  // +-inf samples: nanmax fallback of utl:552-554 (KT >= 1 always)
          uint32_t m = 0;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            uint32_t t0 = cw[w] > 0 ? ringT[(w * KT) * 64 + lane] : 0u;
            m = t0 > m ? t0 : m;
          }
          r = (double)xh_key2f(m);
        }
      }
      if (active) out[((int64_t)j * ndoy + d) * C + c] = r;
    }
  };

  if (OFFSET) {
    for (int di = blockIdx.y; di < ndl; di += gridDim.y) {
      const int d = doy_list[di];
#pragma unroll
      for (int k = 0; k < W; ++k) build(k, rows_of(d, k - half));
      select_and_store(d);
    }
  } else {
    int d0 = blockIdx.y * chunk, d1 = d0 + chunk;
    if (d1 > ndoy) d1 = ndoy;
    for (int dn = d0 - half; dn < d0 + half; ++dn) build(((dn % W) + W) % W, rows_of(dn, 0));
    // software pipeline: the day-set of doy d+half+1 is in flight (and the rows of d+half+2 resolved) while doy d is
    // selected
    gather(rows_of(d0 + half, 0));
    int rows_next = rows_of(d0 + half + 1, 0);
    for (int d = d0; d < d1; ++d) {
      const int dn = d + half;
      finish(((dn % W) + W) % W);
      if (d + 1 < d1) {
        gather(rows_next);
        rows_next = rows_of(dn + 2, 0);
      }
      if (regular[d] && !(abl & 2)) select_and_store(d);
    }
  }
}

// ---- doy re-gridding --------------------------------------------------------------------------------
// _interpolate_doy_calendar (cal:690-726): interpolate_na(dim="dayofyear") [linear in the doy coordinate, no
// extrapolation past the first/last valid point] then interp onto the target doys: host supplies, for each
// output doy j, the bracketing source rows i0[j], i1[j], dxn[j] = x_new - x_lo and dxs[j] = x_hi - x_lo.
__global__ void __launch_bounds__(XH_BLOCK)
k_doy_interp(const double* __restrict__ in, int D_in, int64_t C, const int32_t* __restrict__ i0,
             const int32_t* __restrict__ i1, const double* __restrict__ dxn, const double* __restrict__ dxs, int D_out,
             double* __restrict__ out, double* __restrict__ filled, const double* __restrict__ xsrc) {
  int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  // pass 1: interpolate_na along doy.  xarray interpolates in the dayofyear COORDINATE (use_coordinate=True), which is
  // not uniform when a doy is absent from the series (e.g. 365 days starting on Feb 28 of a leap year never see doy 58):
  // xsrc[d] holds the coordinate of source row d (NULL: row index).
  int last = -1;
  double lastv = 0.0;
  for (int d = 0; d < D_in; ++d) {
    double v = in[(int64_t)d * C + c];
    if (v == v) {
      if (last >= 0 && d - last > 1) {
        const double xl = xsrc ? xsrc[last] : (double)last, xd = xsrc ? xsrc[d] : (double)d;
        for (int g = last + 1; g < d; ++g) {
          // np.interp form: slope * (x - xlo) + ylo
          double slope = (v - lastv) / (xd - xl);
          filled[(int64_t)g * C + c] = slope * ((xsrc ? xsrc[g] : (double)g) - xl) + lastv;
        }
      } else if (last < 0) {
        for (int g = 0; g < d; ++g) filled[(int64_t)g * C + c] = xh_nan64();
      }
      filled[(int64_t)d * C + c] = v;
      last = d;
      lastv = v;
    }
  }
  for (int g = last + 1; g < D_in; ++g) filled[(int64_t)g * C + c] = xh_nan64();
  // pass 2: linear re-grid, scipy interp1d form: slope = (y_hi - y_lo) / (x_hi - x_lo); y = slope * (x - x_lo) + y_lo
  for (int j = 0; j < D_out; ++j) {
    double a = filled[(int64_t)i0[j] * C + c], b = filled[(int64_t)i1[j] * C + c];
    double slope = (b - a) / dxs[j];
    out[(int64_t)j * C + c] = slope * dxn[j] + a;
  }
}

static int next_pow2(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

static int launch_pdoy_lds(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* d_tbase,
                           int nyears, int ndoy, int window, const double* d_q, int nper, double alpha, double beta,
                           double* out, const int32_t* d_vmap, int64_t Tv, const int32_t* d_doy_list, int ndl) {
  int N = nyears * window;
  int NP = next_pow2(N < 2 ? 2 : N);
  XH_REQUIRE(NP <= 4096, XH_ERR_LIMIT, "percentile: %d samples per cell exceed the LDS column capacity (4096)", N);
  const int cw = NP <= 512 ? 64 : (NP == 1024 ? 32 : (NP == 2048 ? 16 : 8));  // cells per wave: NP * cw * 4 B <= 128 KB
  size_t lds = (size_t)NP * cw * sizeof(uint32_t);
  if (lds > 64 * 1024)
    XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_pdoy_lds, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int nd = d_doy_list ? ndl : ndoy;
  dim3 grid((unsigned)cdiv64(C, cw), (unsigned)(nd > 1024 ? 1024 : nd));
  hipLaunchKernelGGL(k_pdoy_lds, grid, dim3(64), lds, ctx->stream, x, T, C, st, d_tbase, nyears, ndoy, window, NP, cw, d_q, nper,
                     alpha, beta, out, d_vmap, Tv, d_doy_list, ndl);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

static int launch_pdoy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* d_tbase, int nyears,
                       int ndoy, int window, const double* d_q, int nper, double alpha, double beta, double* out) {
  int N = nyears * window;
  unsigned gy = (unsigned)(ndoy > 1024 ? 1024 : ndoy);
  if (N <= 8 && xh_pick_vec(x, C, st) == 4) {
    dim3 grid((unsigned)cdiv64(cdiv64(C, 4), XH_BLOCK), gy);
    hipLaunchKernelGGL((k_pdoy_reg<8, 4>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, d_tbase, nyears, ndoy,
                       window, d_q, nper, alpha, beta, out);
  } else if (N <= 8) {
    dim3 grid((unsigned)cdiv64(C, XH_BLOCK), gy);
    hipLaunchKernelGGL((k_pdoy_reg<8, 1>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, d_tbase, nyears, ndoy,
                       window, d_q, nper, alpha, beta, out);
  } else if (N <= 16) {
    dim3 grid((unsigned)cdiv64(C, XH_BLOCK), gy);
    hipLaunchKernelGGL((k_pdoy_reg<16, 1>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, d_tbase, nyears, ndoy,
                       window, d_q, nper, alpha, beta, out);
  } else if (N <= 32) {
    dim3 grid((unsigned)cdiv64(C, XH_BLOCK), gy);
    hipLaunchKernelGGL((k_pdoy_reg<32, 1>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, d_tbase, nyears, ndoy,
                       window, d_q, nper, alpha, beta, out);
  } else {
    return launch_pdoy_lds(ctx, x, T, C, st, d_tbase, nyears, ndoy, window, d_q, nper, alpha, beta, out, nullptr, T, nullptr,
                           0);
  }
  XH_LAUNCH_CHECK();
  return XH_OK;
}

// 1 / 2 when every percentile of the table clips to the maximum / minimum of a FULL window (n == W), else 0
static int slide_fast_mode(const QTab* tab, int nper, int W) {
  bool allmax = W > 1, allmin = W > 1;
  for (int j = 0; j < nper; ++j) {
    const QTab& e = tab[j * (W + 1) + W];
    allmax = allmax && e.lo == e.hi && e.lo == W - 1;
    allmin = allmin && e.lo == e.hi && e.lo == 0;
  }
  return allmax ? 1 : (allmin ? 2 : 0);
}

// Shared implementation of xh_percentile_doy / xh_percentile_doy_mapped.  `vmap` (host, Tv entries, may be NULL) maps
// the virtual time axis the calendar tables refer to onto physical rows of x (-1 = day absent): the bootstrap of
// core/bootstrapping.py:235-282 becomes a different index table per replica instead of a deep copy of the base period.
static int pdoy_impl(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* tbase, int nyears,
                     int ndoy, int window, const double* qh, int nper, double alpha, double beta, const int32_t* vmap,
                     int64_t Tv, double* out) {
  size_t cur = 0;
  void *d_tb = nullptr, *d_q = nullptr, *d_vmap = nullptr;
  const int N = nyears * window;
  // fast path: one year of contiguous days -> sliding register window
  bool contiguous = !vmap && nyears == 1 && nper <= 8 && (window == 3 || window == 5 || window == 7) && tbase[0] >= 0;
  for (int d = 1; contiguous && d < ndoy; ++d) contiguous = tbase[d] == tbase[0] + d;
  if (contiguous) {
    QTab tab[8 * 8];
    build_qtab(window, qh, nper, alpha, beta, tab);
    void* d_tab = nullptr;
    int rc = xh_scratch_upload(ctx, &cur, tab, sizeof(QTab) * (size_t)nper * (window + 1), &d_tab);
    if (rc) return rc;
    int chunk = 32;
    if (const char* e = xh_diag_env("XH_PDOY_SLIDE_CHUNK")) chunk = atoi(e) > 0 ? atoi(e) : chunk;  // diagnostics
    int vec = xh_pick_vec(x, C, st);
    if ((reinterpret_cast<uintptr_t>(out) & 15) != 0) vec = 1;
    dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), (unsigned)((ndoy + chunk - 1) / chunk));
#define XH_SLIDE(W, V)                                                                                              \
  hipLaunchKernelGGL((k_pdoy_slide<W, V>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, (int64_t)tbase[0], ndoy, \
                     chunk, (const QTab*)d_tab, nper, out, slide_fast_mode(tab, nper, window))
    if (vec == 4) {
      if (window == 3) XH_SLIDE(3, 4); else if (window == 5) XH_SLIDE(5, 4); else XH_SLIDE(7, 4);
    } else {
      if (window == 3) XH_SLIDE(3, 1); else if (window == 5) XH_SLIDE(5, 1); else XH_SLIDE(7, 1);
    }
#undef XH_SLIDE
    XH_LAUNCH_CHECK();
    return XH_OK;
  }
  int rc = xh_scratch_upload(ctx, &cur, tbase, sizeof(int32_t) * (size_t)nyears * ndoy, &d_tb);
  if (rc) return rc;
  rc = xh_scratch_upload(ctx, &cur, qh, sizeof(double) * nper, &d_q);
  if (rc) return rc;
  if (vmap) {
    rc = xh_scratch_upload(ctx, &cur, vmap, sizeof(int32_t) * (size_t)Tv, &d_vmap);
    if (rc) return rc;
  }
  if (!vmap && N <= 32)
    return launch_pdoy(ctx, x, T, C, st, (const int32_t*)d_tb, nyears, ndoy, window, (const double*)d_q, nper, alpha, beta,
                       out);
  const bool mergeable = N > 32 && nyears <= 64 && (window == 3 || window == 5 || window == 7) && nper <= 16;
  if (!mergeable)
    return launch_pdoy_lds(ctx, x, T, C, st, (const int32_t*)d_tb, nyears, ndoy, window, (const double*)d_q, nper, alpha,
                           beta, out, (const int32_t*)d_vmap, Tv, nullptr, 0);
  // ---- merge path: flag the doys for which "window sample set == union of the W day-sets" holds exactly
  uint8_t* regular = (uint8_t*)malloc((size_t)ndoy);
  int32_t* irregular = (int32_t*)malloc(sizeof(int32_t) * (size_t)ndoy);
  QTab* tab = (QTab*)malloc(sizeof(QTab) * (size_t)nper * (N + 1));
  if (!regular || !irregular || !tab) {
    free(regular); free(irregular); free(tab);
    xh_set_error("xh_percentile_doy: out of host memory");
    return XH_ERR_ARG;
  }
  const int nirr = pdoy_regular_flags(tbase, nyears, ndoy, window, T, vmap, Tv, regular, irregular);
  build_qtab(N, qh, nper, alpha, beta, tab);
  // Classify the percentiles: "top" / "bottom" when, for every possible valid count n, both order statistics lie within
  // the 16 largest / smallest samples (register kernel k_pdoy_top16), otherwise "rest" (LDS-ring kernel k_pdoy_merge).
  int32_t jtop[64], jbot[64], jrest[64];
  int ntop = 0, nbot = 0, nrest = 0;
  for (int j = 0; j < nper; ++j) {
    int dt = 0, db = 0;
    for (int n = 0; n <= N; ++n) {
      const QTab& e = tab[j * (N + 1) + n];
      if (e.lo < 0) continue;
      if (n - e.lo > dt) dt = n - e.lo;
      if (e.hi + 1 > db) db = e.hi + 1;
    }
    if (dt <= 16) jtop[ntop++] = j;
    else if (db <= 16) jbot[nbot++] = j;
    else jrest[nrest++] = j;
  }
  // deepest pop from either end over a set of percentiles and all valid counts (same from_top rule as the kernel):
  // "rest" for the regular doys, ALL percentiles for the irregular doys (those always take the LDS-ring kernel)
  auto ring_depth = [&](const int32_t* js, int nj, int* pKT, int* pKB) {
    int kt = 1, kb = 0;
    for (int jj = 0; jj < nj; ++jj)
      for (int n = 0; n <= N; ++n) {
        const QTab& e = tab[js[jj] * (N + 1) + n];
        if (e.lo < 0) continue;
        if ((n - 1 - e.hi) < e.lo) { if (n - e.lo > kt) kt = n - e.lo; }
        else { if (e.hi + 1 > kb) kb = e.hi + 1; }
      }
    if (kt > nyears) kt = nyears;
    if (kb > nyears) kb = nyears;
    if (kt + kb >= nyears) { kt = nyears; kb = 0; }  // full mode: whole list in ringT, bottom pops index it from the end
    *pKT = kt; *pKB = kb;
  };
  int32_t jall[64];
  for (int j = 0; j < nper; ++j) jall[j] = j;
  int KT = 1, KB = 0, KTa = 1, KBa = 0;
  ring_depth(jrest, nrest, &KT, &KB);
  ring_depth(jall, nper, &KTa, &KBa);
  void *d_reg = nullptr, *d_tab = nullptr, *d_irr = nullptr, *d_jtop = nullptr, *d_jbot = nullptr, *d_jrest = nullptr,
       *d_jall = nullptr;
  rc = xh_scratch_upload(ctx, &cur, regular, (size_t)ndoy, &d_reg);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, tab, sizeof(QTab) * (size_t)nper * (N + 1), &d_tab);
  if (!rc && nirr) rc = xh_scratch_upload(ctx, &cur, irregular, sizeof(int32_t) * (size_t)nirr, &d_irr);
  if (!rc && ntop) rc = xh_scratch_upload(ctx, &cur, jtop, sizeof(int32_t) * (size_t)ntop, &d_jtop);
  if (!rc && nbot) rc = xh_scratch_upload(ctx, &cur, jbot, sizeof(int32_t) * (size_t)nbot, &d_jbot);
  if (!rc && nrest) rc = xh_scratch_upload(ctx, &cur, jrest, sizeof(int32_t) * (size_t)nrest, &d_jrest);
  if (!rc && nirr) rc = xh_scratch_upload(ctx, &cur, jall, sizeof(int32_t) * (size_t)nper, &d_jall);
  free(regular); free(irregular); free(tab);
  if (rc) return rc;
  const int chunk = 24;
  const char* ea = xh_diag_env("XH_PDOY_ABL");  // diagnostics only (results become wrong)
  const int abl = ea ? atoi(ea) : 0;
  const int NYP = nyears <= 32 ? 32 : 64;
  dim3 grid((unsigned)cdiv64(C, 64), (unsigned)((ndoy + chunk - 1) / chunk));
  dim3 grid_irr((unsigned)cdiv64(C, 64), (unsigned)(nirr > 0 ? nirr : 1));
  // ---- register top-16 kernel (pdoy_top.hip) for the top / bottom groups, regular doys
  if (ntop) {
    rc = xh_launch_pdoy_top16(ctx, x, T, C, st, (const int32_t*)d_tb, nyears, ndoy, window, (const QTab*)d_tab,
                              (const int32_t*)d_jtop, ntop, 0, out, (const int32_t*)d_vmap, Tv, (const uint8_t*)d_reg);
    if (rc) return rc;
  }
  if (nbot) {
    rc = xh_launch_pdoy_top16(ctx, x, T, C, st, (const int32_t*)d_tb, nyears, ndoy, window, (const QTab*)d_tab,
                              (const int32_t*)d_jbot, nbot, 1, out, (const int32_t*)d_vmap, Tv, (const uint8_t*)d_reg);
    if (rc) return rc;
  }
  if (nrest == 0 && nirr == 0) return XH_OK;
  // ---- the remaining percentiles on the regular doys: the split-walk kernel (pdoy_walk.hip) where it applies
  if (nrest) {
    const int rw = xh_launch_pdoy_walk(ctx, x, T, C, st, (const int32_t*)d_tb, nyears, ndoy, window, (const QTab*)d_tab,
                                       (const int32_t*)d_jrest, nrest, out, (const int32_t*)d_vmap, Tv, (const uint8_t*)d_reg);
    if (rw == XH_OK) nrest = 0;
    else if (rw != XH_ERR_NOTIMPL) return rw;
    if (nrest == 0 && nirr == 0) return XH_OK;
  }
  // ---- LDS-ring kernel: what is left of them, and every percentile on the irregular doys
  const size_t lds = ((size_t)window * (KT + KB) * 64 + (size_t)window * 64) * sizeof(uint32_t);
  const size_t lds_a = ((size_t)window * (KTa + KBa) * 64 + (size_t)window * 64) * sizeof(uint32_t);
#define XH_MERGE(W, NY)                                                                                                    \
  do {                                                                                                                     \
    if (nrest) {                                                                                                           \
      if (lds > 64 * 1024)                                                                                                 \
        XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_pdoy_merge<W, NY, false>,                                          \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));                           \
      hipLaunchKernelGGL((k_pdoy_merge<W, NY, false>), grid, dim3(64), lds, ctx->stream, x, T, C, st, (const int32_t*)d_tb, \
                         nyears, ndoy, chunk, (const QTab*)d_tab, (const int32_t*)d_jrest, nrest, out,                      \
                         (const int32_t*)d_vmap, Tv, (const uint8_t*)d_reg, (const int32_t*)nullptr, 0, KT, KB, abl);       \
    }                                                                                                                      \
    if (nirr) {                                                                                                            \
      if (lds_a > 64 * 1024)                                                                                               \
        XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_pdoy_merge<W, NY, true>,                                           \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a));                         \
      hipLaunchKernelGGL((k_pdoy_merge<W, NY, true>), grid_irr, dim3(64), lds_a, ctx->stream, x, T, C, st,                  \
                         (const int32_t*)d_tb, nyears, ndoy, chunk, (const QTab*)d_tab, (const int32_t*)d_jall, nper, out,  \
                         (const int32_t*)d_vmap, Tv, (const uint8_t*)d_reg, (const int32_t*)d_irr, nirr, KTa, KBa, abl);    \
    }                                                                                                                      \
  } while (0)
  if (NYP == 32) {
    if (window == 3) XH_MERGE(3, 32); else if (window == 5) XH_MERGE(5, 32); else XH_MERGE(7, 32);
  } else {
    if (window == 3) XH_MERGE(3, 64); else if (window == 5) XH_MERGE(5, 64); else XH_MERGE(7, 64);
  }
#undef XH_MERGE
  XH_LAUNCH_CHECK();
  return XH_OK;
}

// tx90p-style fused chain for a base period that is ONE contiguous year and also the analysed period
static int pdoy_count_impl(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* tbase, int ndoy,
                           int window, double per, double alpha, double beta, int op, const int32_t* doy_period, int P,
                           int32_t* count_out, int32_t* valid_out) {
  bool contiguous = (window == 3 || window == 5 || window == 7) && tbase[0] >= 0;
  for (int d = 1; contiguous && d < ndoy; ++d) contiguous = tbase[d] == tbase[0] + d;
  if (!contiguous) return XH_ERR_NOTIMPL;
  for (int d = 0; d < ndoy; ++d)
    XH_REQUIRE(doy_period[d] >= 0 && doy_period[d] < P, XH_ERR_ARG, "xh_percentile_doy_count: doy_period[%d] outside [0, P)", d);
  size_t cur = 0;
  QTab tab[8];
  const double q = per / 100.0;
  build_qtab(window, &q, 1, alpha, beta, tab);
  void *d_tab = nullptr, *d_dp = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, tab, sizeof(QTab) * (size_t)(window + 1), &d_tab);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, doy_period, sizeof(int32_t) * (size_t)ndoy, &d_dp);
  if (rc) return rc;
  XH_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int32_t) * (size_t)P * (size_t)C, ctx->stream));
  if (valid_out) XH_CHECK_HIP(hipMemsetAsync(valid_out, 0, sizeof(int32_t) * (size_t)P * (size_t)C, ctx->stream));
  int chunk = 61;  // doys per workgroup (W - 1 halo rows each): 16 / 32 / 46 / 61 / 92 / 183 -> 0.61 / 0.57 / 0.56 / 0.54 / 0.55 / 0.56 ms
  if (const char* e = xh_diag_env("XH_PDOY_SLIDE_COUNT_CHUNK")) chunk = atoi(e) > 0 ? atoi(e) : chunk;  // diagnostics
  const int vec = xh_pick_vec(x, C, st);
  dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), (unsigned)((ndoy + chunk - 1) / chunk));
#define XH_SLIDEC(W, V)                                                                                                   \
  hipLaunchKernelGGL((k_pdoy_slide<W, V, true>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, (int64_t)tbase[0], ndoy, \
                     chunk, (const QTab*)d_tab, 1, (double*)nullptr, slide_fast_mode(tab, 1, window), op,          \
                     (const int32_t*)d_dp, count_out, valid_out)
  if (vec == 4) {
    if (window == 3) XH_SLIDEC(3, 4); else if (window == 5) XH_SLIDEC(5, 4); else XH_SLIDEC(7, 4);
  } else {
    if (window == 3) XH_SLIDEC(3, 1); else if (window == 5) XH_SLIDEC(5, 1); else XH_SLIDEC(7, 1);
  }
#undef XH_SLIDEC
  XH_LAUNCH_CHECK();
  return XH_OK;
}

// tx90p-style fused chain on a multi-year base period that is also the analysed period: every doy regular (no calendar
// gaps: noleap / 360-day calendars, whole years), the percentile within the 16 largest / smallest samples for every valid
// count (register top-16 kernel).  Anything else: XH_ERR_NOTIMPL -> the caller runs xh_percentile_doy + xh_threshold_count.
// doy_period: (nyears, ndoy) period of every (year, doy) day, -1 where tbase is -1.
static int pdoy_count_multi(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* tbase, int nyears,
                            int ndoy, int window, double per, double alpha, double beta, int op, const int32_t* doy_period,
                            int P, int32_t* count_out, int32_t* valid_out) {
  const int N = nyears * window;
  if (!(N > 32 && nyears <= 64 && (window == 3 || window == 5 || window == 7))) return XH_ERR_NOTIMPL;
  if (op != XH_OP_GT && op != XH_OP_GE && op != XH_OP_LT && op != XH_OP_LE) return XH_ERR_NOTIMPL;
  for (int64_t i = 0; i < (int64_t)nyears * ndoy; ++i) {
    XH_REQUIRE(tbase[i] >= -1 && tbase[i] < T, XH_ERR_ARG, "xh_percentile_doy_count: tbase entry out of range");
    XH_REQUIRE(tbase[i] < 0 ? doy_period[i] < 0 : (doy_period[i] >= 0 && doy_period[i] < P), XH_ERR_ARG,
               "xh_percentile_doy_count: doy_period[%lld] must be in [0, P) for present days and < 0 for absent ones",
               (long long)i);
  }
  uint8_t* regular = (uint8_t*)malloc((size_t)ndoy);
  int32_t* irregular = (int32_t*)malloc(sizeof(int32_t) * (size_t)ndoy);
  QTab* tab = (QTab*)malloc(sizeof(QTab) * (size_t)(N + 1));
  if (!regular || !irregular || !tab) {
    free(regular); free(irregular); free(tab);
    xh_set_error("xh_percentile_doy_count: out of host memory");
    return XH_ERR_ARG;
  }
  const int nirr = pdoy_regular_flags(tbase, nyears, ndoy, window, T, nullptr, T, regular, irregular);
  const double q = per / 100.0;
  build_qtab(N, &q, 1, alpha, beta, tab);
  int dt = 0, db = 0;
  for (int n = 0; n <= N; ++n) {
    const QTab& e = tab[n];
    if (e.lo < 0) continue;
    if (n - e.lo > dt) dt = n - e.lo;
    if (e.hi + 1 > db) db = e.hi + 1;
  }
  const bool top = dt <= 16, bot = !top && db <= 16;
  const bool table_ok = nirr == 0 && (top || bot);
  int rc = table_ok ? XH_OK : XH_ERR_NOTIMPL;
  // period boundaries must fall on the same doy in every year (the kernel flushes all years' counters together)
  uint8_t* newseg = (uint8_t*)malloc((size_t)ndoy);
  if (!newseg) rc = XH_ERR_ARG;
  for (int d = 0; !rc && d < ndoy; ++d) {
    int nchg = 0, npres = 0;
    for (int y = 0; y < nyears; ++y) {
      const int32_t p1 = doy_period[(int64_t)y * ndoy + d], p0 = d > 0 ? doy_period[(int64_t)y * ndoy + d - 1] : p1;
      if (p1 < 0 || p0 < 0) continue;
      npres++;
      nchg += p1 != p0 ? 1 : 0;
    }
    if (nchg != 0 && nchg != npres) rc = XH_ERR_NOTIMPL;
    newseg[d] = nchg != 0 ? 1 : 0;
  }
  size_t cur = 0;
  void *d_tb = nullptr, *d_reg = nullptr, *d_tab = nullptr, *d_j = nullptr, *d_dp = nullptr, *d_ns = nullptr;
  const int32_t j0 = 0;
  // Two kernels beat the fused one since tcount.hip: the (ndoy, C) table goes to scratch (top-16 table kernel, 13.7 ms at
  // 30 yr x 1440 x 720) and the tile kernel counts against it (8.3 ms) — 22.0 ms against 25.2 ms fused (which re-reads
  // every sample for the count).  The fused kernel stays for the shapes the tile kernel does not take.
  // (It does not need the period boundaries on the same doy in every year either.)
  if (table_ok && newseg && !xh_diag_env("XH_PDOY_COUNT_FUSED")) {
    int64_t* plen = (int64_t*)calloc((size_t)P, sizeof(int64_t));
    const int64_t nslots = xh_tcount_meta_slots(T);
    uint32_t* hmeta = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)nslots);
    int rc2 = (plen && hmeta) ? XH_OK : XH_ERR_NOTIMPL;
    size_t lds = 0;
    int narrow = 0;
    if (!rc2) {
      for (int64_t i = 0; i < nslots; ++i) hmeta[i] = (uint32_t)ndoy | ((uint32_t)P << 16);  // a step outside the table / every period
      for (int y = 0; y < nyears; ++y)
        for (int d = 0; d < ndoy; ++d) {
          const int64_t i = (int64_t)y * ndoy + d;
          if (tbase[i] < 0) continue;
          hmeta[xh_tcount_slot_of_row(tbase[i])] = (uint32_t)d | ((uint32_t)doy_period[i] << 16);
          plen[doy_period[i]]++;
        }
      int64_t longest = 0;
      for (int p = 0; p < P; ++p) longest = plen[p] > longest ? plen[p] : longest;
      rc2 = xh_tcount_plan(T, C, st, op, P, ndoy, longest, &lds, &narrow);
    }
    void *ws = nullptr, *d_meta = nullptr;
    const size_t b_table = sizeof(double) * (size_t)ndoy * (size_t)C;
    if (!rc2) rc2 = xh_big_scratch(ctx, b_table, &ws);
    if (!rc2) rc2 = xh_scratch_upload(ctx, &cur, tbase, sizeof(int32_t) * (size_t)nyears * ndoy, &d_tb);
    if (!rc2) rc2 = xh_scratch_upload(ctx, &cur, regular, (size_t)ndoy, &d_reg);
    if (!rc2) rc2 = xh_scratch_upload(ctx, &cur, tab, sizeof(QTab) * (size_t)(N + 1), &d_tab);
    if (!rc2) rc2 = xh_scratch_upload(ctx, &cur, &j0, sizeof(int32_t), &d_j);
    if (!rc2) rc2 = xh_scratch_upload(ctx, &cur, hmeta, sizeof(uint32_t) * (size_t)nslots, &d_meta);
    free(plen); free(hmeta);
    if (!rc2) {
      free(regular); free(irregular); free(tab); free(newseg);
      rc2 = xh_launch_pdoy_top16(ctx, x, T, C, st, (const int32_t*)d_tb, nyears, ndoy, window, (const QTab*)d_tab,
                                 (const int32_t*)d_j, 1, bot ? 1 : 0, (double*)ws, nullptr, T, (const uint8_t*)d_reg);
      if (rc2) return rc2;
      return xh_tcount_run(ctx, x, T, C, st, op, (const double*)ws, C, (const uint32_t*)d_meta, P, ndoy, narrow, lds, count_out,
                           valid_out);
    }
    cur = 0;  // not this shape: the fused kernel below
  }
  if (!rc) rc = xh_scratch_upload(ctx, &cur, tbase, sizeof(int32_t) * (size_t)nyears * ndoy, &d_tb);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, regular, (size_t)ndoy, &d_reg);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, tab, sizeof(QTab) * (size_t)(N + 1), &d_tab);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, &j0, sizeof(int32_t), &d_j);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, doy_period, sizeof(int32_t) * (size_t)nyears * ndoy, &d_dp);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, newseg, (size_t)ndoy, &d_ns);
  free(regular); free(irregular); free(tab); free(newseg);
  if (rc) return rc;
  XH_CHECK_HIP(hipMemsetAsync(count_out, 0, sizeof(int32_t) * (size_t)P * (size_t)C, ctx->stream));
  if (valid_out) XH_CHECK_HIP(hipMemsetAsync(valid_out, 0, sizeof(int32_t) * (size_t)P * (size_t)C, ctx->stream));
  return xh_launch_pdoy_top16_count(ctx, x, T, C, st, (const int32_t*)d_tb, nyears, ndoy, window, (const QTab*)d_tab,
                                    (const int32_t*)d_j, bot ? 1 : 0, (const uint8_t*)d_reg, op, (const int32_t*)d_dp,
                                    count_out, valid_out, (const uint8_t*)d_ns);
}

extern "C" {

int xh_percentile_doy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const int32_t* tbase,
                      int nyears, int ndoy, int window, const double* per, int nper, double alpha, double beta,
                      double* out) {
  return xh_percentile_doy_mapped(ctx, x, T, C, st, sc, tbase, nyears, ndoy, window, per, nper, alpha, beta, nullptr, T, out);
}

int xh_percentile_doy_mapped(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc,
                             const int32_t* tbase, int nyears, int ndoy, int window, const double* per, int nper,
                             double alpha, double beta, const int32_t* vmap, int64_t Tv, double* out) {
  XH_REQUIRE(ctx && x && tbase && per && out, XH_ERR_ARG, "xh_percentile_doy: NULL argument");
  XH_REQUIRE(T >= 1 && C >= 0 && nyears >= 1 && ndoy >= 1 && window >= 1 && nper >= 1 && Tv >= 1, XH_ERR_ARG,
             "xh_percentile_doy: bad shape");
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT, "xh_percentile_doy: needs a time-major view (sc == 1, st >= C)");
  XH_REQUIRE(nper <= 64, XH_ERR_LIMIT, "xh_percentile_doy: at most 64 percentiles per call");
  for (int j = 0; j < nper; ++j)
    XH_REQUIRE(per[j] >= 0.0 && per[j] <= 100.0, XH_ERR_ARG, "xh_percentile_doy: percentile %g outside [0, 100]", per[j]);
  for (int64_t i = 0; i < (int64_t)nyears * ndoy; ++i)
    XH_REQUIRE(tbase[i] >= -1 && tbase[i] < Tv, XH_ERR_ARG, "xh_percentile_doy: tbase entry out of range");
  if (vmap)
    for (int64_t i = 0; i < Tv; ++i)
      XH_REQUIRE(vmap[i] >= -1 && vmap[i] < T, XH_ERR_ARG, "xh_percentile_doy: vmap entry out of range");
  if (C == 0) return XH_OK;
  double qh[64];
  for (int j = 0; j < nper; ++j) qh[j] = per[j] / 100.0;  // utl:366
  return pdoy_impl(ctx, x, T, C, st, tbase, nyears, ndoy, window, qh, nper, alpha, beta, vmap, Tv, out);
}

int xh_percentile_doy_count(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const int32_t* tbase,
                            int nyears, int ndoy, int window, double per, double alpha, double beta, int op,
                            const int32_t* doy_period, int P, int32_t* count_out, int32_t* valid_out) {
  XH_REQUIRE(ctx && x && tbase && doy_period && count_out, XH_ERR_ARG, "xh_percentile_doy_count: NULL argument");
  XH_REQUIRE(T >= 1 && C >= 0 && ndoy >= 1 && P >= 1, XH_ERR_ARG, "xh_percentile_doy_count: bad shape");
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT, "xh_percentile_doy_count: needs a time-major view (sc == 1, st >= C)");
  XH_REQUIRE(per >= 0.0 && per <= 100.0, XH_ERR_ARG, "xh_percentile_doy_count: percentile outside [0, 100]");
  XH_REQUIRE(op >= XH_OP_GT && op <= XH_OP_NE, XH_ERR_OP, "Operation `%d` not recognized.", op);
  if (C == 0) return XH_OK;
  if (nyears != 1)
    return pdoy_count_multi(ctx, x, T, C, st, tbase, nyears, ndoy, window, per, alpha, beta, op, doy_period, P, count_out,
                            valid_out);
  return pdoy_count_impl(ctx, x, T, C, st, tbase, ndoy, window, per, alpha, beta, op, doy_period, P, count_out, valid_out);
}

int xh_nan_quantile(xh_ctx* ctx, const float* x, int64_t N, int64_t C, int64_t sn, int64_t sc, const double* q, int nq,
                    double alpha, double beta, double* out) {
  XH_REQUIRE(ctx && x && q && out, XH_ERR_ARG, "xh_nan_quantile: NULL argument");
  XH_REQUIRE(N >= 1 && C >= 0 && nq >= 1 && nq <= 64, XH_ERR_ARG, "xh_nan_quantile: bad shape (N >= 1, 1 <= nq <= 64)");
  XH_REQUIRE(N <= 4096, XH_ERR_LIMIT, "xh_nan_quantile: N = %lld samples exceed 4096 (use xh_quantile_series)",
             (long long)N);
  if (C == 0) return XH_OK;
  const float* xs = x;
  int64_t st = sn;
  if (!(sc == 1 && sn >= C)) {
    // sample-minor input (what apply_ufunc hands calc_perc): transpose into scratch to a sample-major view
    XH_REQUIRE(sn == 1 && sc >= N, XH_ERR_LAYOUT, "xh_nan_quantile: one of the two strides must be 1");
    void* tmp = nullptr;
    int rc = xh_big_scratch(ctx, sizeof(float) * (size_t)N * (size_t)C, &tmp);
    if (rc) return rc;
    rc = xh_transpose_f32(ctx, x, C, N, sc, (float*)tmp, C);
    if (rc) return rc;
    xs = (const float*)tmp;
    st = C;
  }
  // one "doy" row, N "years", window 1, tbase[y] = y
  int32_t tb[4096];
  for (int i = 0; i < (int)N; ++i) tb[i] = i;
  size_t cur = 0;
  void *d_tb = nullptr, *d_q = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, tb, sizeof(int32_t) * (size_t)N, &d_tb);
  if (rc) return rc;
  rc = xh_scratch_upload(ctx, &cur, q, sizeof(double) * nq, &d_q);
  if (rc) return rc;
  return launch_pdoy(ctx, xs, N, C, st, (const int32_t*)d_tb, (int)N, 1, 1, (const double*)d_q, nq, alpha, beta, out);
}

int xh_doy_interp(xh_ctx* ctx, const double* in, int D_in, int64_t C, const int32_t* i0, const int32_t* i1,
                  const double* dxn, const double* dxs, int D_out, double* out, const double* xsrc) {
  XH_REQUIRE(ctx && in && i0 && i1 && dxn && dxs && out, XH_ERR_ARG, "xh_doy_interp: NULL argument");
  XH_REQUIRE(D_in >= 1 && D_out >= 1 && C >= 0, XH_ERR_ARG, "xh_doy_interp: bad shape");
  for (int j = 0; j < D_out; ++j)
    XH_REQUIRE(i0[j] >= 0 && i0[j] < D_in && i1[j] >= 0 && i1[j] < D_in, XH_ERR_ARG, "xh_doy_interp: index out of range");
  if (C == 0) return XH_OK;
  size_t cur = 0;
  void *d_i0 = nullptr, *d_i1 = nullptr, *d_w = nullptr, *d_s = nullptr, *filled = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, i0, sizeof(int32_t) * D_out, &d_i0);
  if (rc) return rc;
  rc = xh_scratch_upload(ctx, &cur, i1, sizeof(int32_t) * D_out, &d_i1);
  if (rc) return rc;
  rc = xh_scratch_upload(ctx, &cur, dxn, sizeof(double) * D_out, &d_w);
  if (rc) return rc;
  rc = xh_scratch_upload(ctx, &cur, dxs, sizeof(double) * D_out, &d_s);
  if (rc) return rc;
  void* d_x = nullptr;
  if (xsrc) {
    rc = xh_scratch_upload(ctx, &cur, xsrc, sizeof(double) * D_in, &d_x);
    if (rc) return rc;
  }
  rc = xh_big_scratch(ctx, sizeof(double) * (size_t)D_in * (size_t)C, &filled);
  if (rc) return rc;
  hipLaunchKernelGGL(k_doy_interp, dim3((unsigned)cdiv64(C, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, in, D_in, C,
                     (const int32_t*)d_i0, (const int32_t*)d_i1, (const double*)d_w, (const double*)d_s, D_out, out,
                     (double*)filled, (const double*)d_x);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // extern "C"
