// select.hip — exact multi-quantile selection per column (xsdba nbutils.quantile; E1 of SURVEY.md §8a).
//
// A full sort is ~30x more work than the 2*nq order statistics need.  The samples of a column are held as
// order-preserving uint32 keys and go through ONE counting pass of an MSD radix sort over the key range [kmin, kmax]:
//   1. histogram of NB linear-in-key bins in LDS (ds atomics), exclusive scan -> bin offsets;
//   2. scatter keys to their bin's slot range (grouped by bin, unordered inside a bin);
//   3. each target rank (prev/next of every quantile) finds its bin by binary search in the offsets and selects
//      exactly inside the bin (typically 1-3 keys; all-equal bins — e.g. dry days — short-circuit).
// Binning uses integer arithmetic on the keys, hence is monotone and exact; duplicates and NaNs (excluded, counted)
// are handled.  Kernels by series length (xh_select_columns / xh_select_time_major):
//   T <= 512          k_select_grp      32 (time-major, rows staged coalesced through an LDS tile) or 64 lanes per column
//   512 < T <= 1024   k_select_quantile one wave per column, column in LDS
//   1024 < T <= 16384 k_select_lean     (select2.hip) one workgroup per column, keys in registers, list-free
//   T > 16384         k_select_quantile 1024-thread workgroups, column in LDS
#include <stdlib.h>

#include "common.h"

template <int NT>
__device__ __forceinline__ void group_sync() {
  __syncthreads();
}

// inclusive scan of one value per thread over a group of NT threads (NT multiple of 64); tmp: NT/64 uints in LDS
template <int NT>
__device__ __forceinline__ uint32_t group_incl_scan(uint32_t v, int tid_in_group, uint32_t* tmp) {
  const int lane = tid_in_group & 63, w = tid_in_group >> 6;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    uint32_t o = __shfl_up(v, off, 64);
    if (lane >= off) v += o;
  }
  if (NT > 64) {
    if (lane == 63) tmp[w] = v;
    __syncthreads();
    uint32_t add = 0;
    for (int i = 0; i < w; ++i) add += tmp[i];
    v += add;
    __syncthreads();
  }
  return v;
}

template <int NT>
__device__ __forceinline__ uint32_t group_reduce_min(uint32_t v, int tid_in_group, uint32_t* tmp) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    uint32_t o = __shfl_xor(v, off, 64);
    v = o < v ? o : v;
  }
  if (NT > 64) {
    const int lane = tid_in_group & 63, w = tid_in_group >> 6;
    if (lane == 0) tmp[w] = v;
    __syncthreads();
    v = tmp[0];
    for (int i = 1; i < NT / 64; ++i) v = tmp[i] < v ? tmp[i] : v;
    __syncthreads();
  }
  return v;
}
template <int NT>
__device__ __forceinline__ uint32_t group_reduce_max(uint32_t v, int tid_in_group, uint32_t* tmp) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    uint32_t o = __shfl_xor(v, off, 64);
    v = o > v ? o : v;
  }
  if (NT > 64) {
    const int lane = tid_in_group & 63, w = tid_in_group >> 6;
    if (lane == 0) tmp[w] = v;
    __syncthreads();
    v = tmp[0];
    for (int i = 1; i < NT / 64; ++i) v = tmp[i] > v ? tmp[i] : v;
    __syncthreads();
  }
  return v;
}
template <int NT>
__device__ __forceinline__ uint32_t group_reduce_sum(uint32_t v, int tid_in_group, uint32_t* tmp) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if (NT > 64) {
    const int lane = tid_in_group & 63, w = tid_in_group >> 6;
    if (lane == 0) tmp[w] = v;
    __syncthreads();
    v = 0;
    for (int i = 0; i < NT / 64; ++i) v += tmp[i];
    __syncthreads();
  }
  return v;
}

// k-th smallest (0-based) key of keys[0..m): quickselect on VALUES, the data is not moved (other lanes read the same
// bin).  The answer stays inside the key interval [lo, hi]; each pass counts #(k < e) and #(k <= e) for a pivot e that
// is itself a key of the interval, and remembers one key on either side of e inside the interval as the next pivot.
// O(m) LDS reads per pass, ~2 ln(m) passes expected; a bin of equal keys (the dry days of a precipitation series) ends
// after one pass.  Replaces a candidate-by-candidate scan that cost O(m^2) on skewed or tied data.
__device__ __forceinline__ uint32_t select_in_bin(const uint32_t* __restrict__ keys, uint32_t m, uint32_t kth) {
  uint32_t lo = 0u, hi = 0xFFFFFFFFu;
  uint32_t e = keys[kth < m ? kth : 0u];
  for (;;) {
    uint32_t less = 0, leq = 0, cl = e, ch = e;
#pragma unroll 4
    for (uint32_t b = 0; b < m; ++b) {
      const uint32_t k = keys[b];
      less += k < e ? 1u : 0u;
      leq += k <= e ? 1u : 0u;
      cl = (k < e && k >= lo) ? k : cl;
      ch = (k > e && k <= hi) ? k : ch;
    }
    if (less <= kth && kth < leq) return e;
    if (kth < less) { hi = e - 1u; e = cl; } else { lo = e + 1u; e = ch; }
  }
}

// NT threads per column, KPL keys per thread (T <= NT*KPL), NB bins (multiple of NT), GROUPS columns per block.
// LDS layout per group: sorted[Tpad] | offs[NB+1] | cursor[NB] | vals[2*64] | tmp[16]
template <int NT, int KPL, int NB>
__global__ void __launch_bounds__(NT <= 256 ? 256 : NT)
k_select_quantile(const float* __restrict__ x, int64_t T, int64_t ncols, int64_t col_stride, const double* __restrict__ qs,
                  int nq, float* __restrict__ out, int64_t out_cstride, int64_t out_qstride) {
  extern __shared__ uint32_t lds[];
  constexpr int BLOCK = NT <= 256 ? 256 : NT;
  constexpr int GROUPS = BLOCK / NT;
  constexpr int BPT = NB / NT;  // bins per thread in the scan
  const int tid = threadIdx.x;
  const int g = tid / NT, gt = tid % NT;
  const int Tpad = (int)((T + 63) & ~(int64_t)63);
  const int per_group = Tpad + (NB + 1) + NB + 128 + 16;
  uint32_t* sorted = lds + (size_t)g * per_group;
  uint32_t* offs = sorted + Tpad;
  uint32_t* cursor = offs + NB + 1;
  float* vals = reinterpret_cast<float*>(cursor + NB);
  uint32_t* tmp = reinterpret_cast<uint32_t*>(vals + 128);

  for (int64_t cb = (int64_t)blockIdx.x * GROUPS; cb < ncols; cb += (int64_t)gridDim.x * GROUPS) {
    const int64_t col = cb + g;
    const bool have = col < ncols;
    // ---- load keys into registers
    uint32_t key[KPL];
    uint32_t nv = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
    // unconditional, clamped loads first (a load inside an `if` is followed by s_waitcnt vmcnt(0): one full memory
    // latency PER element), conversion afterwards
    const int64_t colc = have ? col : ncols - 1;
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      int i = gt + k * NT;
      key[k] = __float_as_uint(x[colc * col_stride + (i < T ? i : (int)T - 1)]);
    }
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      int i = gt + k * NT;
      uint32_t kk = (have && i < T) ? xh_f2key(__uint_as_float(key[k])) : 0xFFFFFFFFu;
      key[k] = kk;
      if (kk != 0xFFFFFFFFu) {
        nv++;
        kmin = kk < kmin ? kk : kmin;
        kmax = kk > kmax ? kk : kmax;
      }
    }
    const uint32_t n = group_reduce_sum<NT>(nv, gt, tmp);
    kmin = group_reduce_min<NT>(kmin, gt, tmp);
    kmax = group_reduce_max<NT>(kmax, gt, tmp);
    // the smallest key owns bin 0 (counted in registers: no atomics, never stored, its targets need no search); the other
    // bins divide [kmin2, kmax], kmin2 = the smallest key above it — see k_select_grp
    uint32_t kmin2 = 0xFFFFFFFFu, cnt0 = 0;
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      kmin2 = (key[k] > kmin && key[k] < kmin2) ? key[k] : kmin2;
      cnt0 += (key[k] == kmin && key[k] != 0xFFFFFFFFu) ? 1u : 0u;
    }
    kmin2 = group_reduce_min<NT>(kmin2, gt, tmp);
    cnt0 = group_reduce_sum<NT>(cnt0, gt, tmp);
    kmin2 = kmin2 == 0xFFFFFFFFu ? kmin : kmin2;
    const uint32_t range = (n > 0) ? (kmax - kmin2) : 0u;
    int shift = 0;
    while ((range >> shift) >= (uint32_t)NB) shift++;
    const XhValueBins vb = xh_value_bins(kmin2, kmax, n > 0, NB - 1);  // (keys on both sides of zero: bins linear in the value)
    auto binof = [&](uint32_t kk) -> uint32_t {  // (never called for kmin copies / NaN keys)
      const uint32_t bb = 1u + (vb.on ? xh_value_bin(vb, kk) : ((kk - kmin2) >> shift));
      return bb < (uint32_t)NB ? bb : (uint32_t)NB - 1u;
    };
    // ---- histogram
#pragma unroll
    for (int b = 0; b < BPT; ++b) offs[gt + b * NT] = 0;
    group_sync<NT>();
#pragma unroll
    for (int k = 0; k < KPL; ++k)
      if (key[k] != 0xFFFFFFFFu && key[k] != kmin) atomicAdd(&offs[binof(key[k])], 1u);
    if (gt == 0) offs[0] = cnt0;
    group_sync<NT>();
    // ---- exclusive scan of NB bins: thread owns BPT consecutive bins
    {
      uint32_t loc[BPT], s = 0;
#pragma unroll
      for (int b = 0; b < BPT; ++b) { loc[b] = offs[gt * BPT + b]; s += loc[b]; }
      uint32_t incl = group_incl_scan<NT>(s, gt, tmp);
      uint32_t run = incl - s;
      group_sync<NT>();
#pragma unroll
      for (int b = 0; b < BPT; ++b) {
        offs[gt * BPT + b] = run;
        cursor[gt * BPT + b] = run;
        run += loc[b];
      }
      if (gt == NT - 1) offs[NB] = run;
    }
    group_sync<NT>();
    // ---- scatter (grouped by bin)
#pragma unroll
    for (int k = 0; k < KPL; ++k)
      if (key[k] != 0xFFFFFFFFu && key[k] != kmin) {
        uint32_t pos = atomicAdd(&cursor[binof(key[k])], 1u);
        sorted[pos] = key[k];
      }
    group_sync<NT>();
    // ---- targets: 2 per quantile (prev, next); thread tgt handles target tgt
    for (int tgt = gt; tgt < 2 * nq; tgt += NT) {
      const int j = tgt >> 1;
      float v = xh_nan32();
      if (n >= 1) {
        int r;
        if (T == 1 || n < 2) r = 0;
        else {
          double nn = (double)n, q = qs[j];
          double vi = nn * q + (1.0 + q * (1.0 - 1.0 - 1.0)) - 1.0;  // utl:395 with alpha = beta = 1
          if (vi >= nn - 1.0) r = (int)n - 1;
          else if (vi < 0.0) r = 0;
          else r = (int)floor(vi) + (tgt & 1);
        }
        // bin containing rank r: largest b with offs[b] <= r  (offs non-decreasing, offs[NB] = n)
        int lo = 0, hi = NB;  // invariant: offs[lo] <= r < offs[hi]
        while (hi - lo > 1) {
          int mid = (lo + hi) >> 1;
          if (offs[mid] <= (uint32_t)r) lo = mid; else hi = mid;
        }
        const uint32_t s0 = offs[lo], s1 = offs[lo + 1];
        const uint32_t kth = (uint32_t)r - s0;
        uint32_t ans = kmin;  // bin 0: the copies of the smallest key (not stored)
        if (lo > 0) ans = (s1 - s0 > 1) ? select_in_bin(sorted + s0, s1 - s0, kth) : sorted[s0];  // exact selection inside the bin
        v = xh_key2f(ans);
      }
      vals[tgt] = v;
    }
    group_sync<NT>();
    for (int j = gt; j < nq; j += NT) {
      if (have) {
        double r;
        if (n == 0) r = xh_nan64();
        else if (T == 1 || n < 2) r = (double)vals[2 * j];
        else {
          double nn = (double)n, q = qs[j];
          double vi = nn * q + (1.0 + q * (1.0 - 1.0 - 1.0)) - 1.0;
          float left = vals[2 * j], right = vals[2 * j + 1];
          if (vi >= nn - 1.0 || vi < 0.0) r = (double)left;
          else {
            double gamma = vi - floor(vi);
            float diff = right - left;
            r = (double)left + (double)diff * gamma;
            if (gamma >= 0.5) r = (double)right - (double)diff * (1.0 - gamma);
          }
        }
        out[col * out_cstride + (int64_t)j * out_qstride] = (float)r;
      }
    }
    group_sync<NT>();
  }
}

// ---- short series (T <= 512): G lanes per column, 64/G columns per wave, 256/G columns per workgroup ----------
// With one wave per column the per-column fixed costs (reductions, scan, target search on a few lanes) dominate
// for T ~ 365; sharing every wave instruction between 64/G columns cuts the issued instructions per column.
// TIME_MAJOR reads x[t * stride + col] directly: a workgroup covers 256/G adjacent cells per row (64..128 bytes),
// so no transpose pass is needed for the common (time, lat, lon) layout.
// Columns never span waves here (G <= 64), so phases only need wave-level ordering of the LDS traffic: the LDS
// executes one wave's instructions in issue order; this fence just stops the compiler from reordering across it.
__device__ __forceinline__ void wave_sync() {
  // compiler-only barrier: no s_waitcnt vmcnt(0) (a fence would drain the prefetched global loads of the next
  // column at every phase boundary); cross-lane visibility inside one wave comes from the in-order LDS pipeline
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// workgroup barrier that only orders LDS traffic (no vmcnt wait: prefetched global loads stay in flight)
__device__ __forceinline__ void block_sync_lds() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// STAGE (time-major input, NT = 512, G = 32 -> 16 columns per workgroup): the rows are loaded COALESCED (a half-wave
// reads 2 rows x 16 adjacent cells = two full 64-byte sectors; the lane-owns-column mapping reads 8 bytes per sector,
// PMC: 3x over-fetch), staged through an LDS tile (row pitch 17 words: conflict-free both ways) and picked up by the
// owner lanes; the tile is then reused as the `sorted` arrays.  The next tile is prefetched into registers meanwhile.
template <int G, int KPL, int NB, bool TIME_MAJOR, int NT = 256, bool STAGE = false>
__global__ void __launch_bounds__(NT)
k_select_grp(const float* __restrict__ x, int64_t T, int64_t ncols, int64_t stride, const double* __restrict__ qs, int nq,
             float* __restrict__ out, int64_t out_cstride, int64_t out_qstride, int abl) {
  constexpr int COLS = NT / G;              // columns per workgroup
  static_assert(!STAGE || (TIME_MAJOR && G == 32 && NT == 512), "staged loads: time-major, 32 lanes per column, 512 threads");
  constexpr int BPL = NB / G;               // bins per lane in the scan
  // LDS words per column: cur | sorted | vals, padded so that PER % 32 == G: the 64/G columns that share a wave
  // then sit on disjoint LDS banks for the lane-structured accesses
  constexpr int PER0 = NB + (STAGE ? 0 : G * KPL) + 128;
  constexpr int PER = PER0 + ((G % 32) - (PER0 % 32) + 32) % 32;
  constexpr int PITCH = COLS + 1;                         // tile row pitch (words)
  constexpr int TILEW = STAGE ? G * KPL * PITCH : 0;      // tile words (>= COLS * G * KPL: reused as sorted[])
  __shared__ uint32_t lds[COLS * PER + TILEW];
  const int tid = threadIdx.x;
  const int l = tid & (G - 1), g = tid / G;
  uint32_t* cur = lds + g * PER;
  uint32_t* tile = lds + COLS * PER;
  uint32_t* sorted = STAGE ? tile + g * (G * KPL) : cur + NB;
  float* vals = reinterpret_cast<float*>(cur + NB + (STAGE ? 0 : G * KPL));

  // software pipeline: the raw samples of the NEXT column are in flight while the current one is processed
  float raw[KPL];
  const int sc = tid % COLS, sr = tid / COLS;  // staged mapping: column / first row of this thread (NT / COLS = G rows per pass)
  auto issue_loads = [&](int64_t cb) {
    if (STAGE) {
      // unconditional, clamped loads (a load under a condition is followed by s_waitcnt vmcnt(0)); masked when used
      const int64_t colc = (cb + sc < ncols) ? cb + sc : ncols - 1;
#pragma unroll
      for (int k = 0; k < KPL; ++k) {
        const int t = sr + k * G;
        raw[k] = x[(int64_t)(t < T ? t : (int)T - 1) * stride + colc];
      }
      return;
    }
    const int64_t col = cb + g;
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      int t = l + k * G;
      raw[k] = xh_nan32();
      if (col < ncols && t < T) raw[k] = TIME_MAJOR ? x[(int64_t)t * stride + col] : x[col * stride + t];
    }
  };
  constexpr int NROUND = (128 + G - 1) / G;  // rounds of the target loop (2 * nq <= 128 targets, G lanes)
  int rank_c[NROUND];
  uint32_t rank_n = 0xFFFFFFFFu;
#pragma unroll
  for (int rr = 0; rr < NROUND; ++rr) rank_c[rr] = 0;
  const int64_t cstep = (int64_t)gridDim.x * COLS;
  int64_t cb = (int64_t)blockIdx.x * COLS;
  if (cb < ncols) issue_loads(cb);
  for (; cb < ncols; cb += cstep) {
    const int64_t col = cb + g;
    const bool have = col < ncols;
    uint32_t key[KPL];
    uint32_t nv = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
    if (STAGE) {
      // registers (coalesced mapping) -> LDS tile -> owner lanes; rows >= T and columns >= ncols become NaN keys
#pragma unroll
      for (int k = 0; k < KPL; ++k) tile[(sr + k * G) * PITCH + sc] = __float_as_uint(raw[k]);
      block_sync_lds();
#pragma unroll
      for (int k = 0; k < KPL; ++k) {
        const int t = l + k * G;
        const float f = __uint_as_float(tile[t * PITCH + g]);
        raw[k] = (have && t < T) ? f : xh_nan32();
      }
      block_sync_lds();  // the tile is dead from here on: it becomes the sorted[] arrays
    }
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      uint32_t kk = xh_f2key(raw[k]);
      key[k] = kk;
      bool ok = kk != 0xFFFFFFFFu;
      nv += ok ? 1u : 0u;
      kmin = (ok && kk < kmin) ? kk : kmin;
      kmax = (ok && kk > kmax) ? kk : kmax;
    }
    if (cb + cstep < ncols) issue_loads(cb + cstep);
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
      nv += __shfl_xor(nv, off, G);
      uint32_t a = __shfl_xor(kmin, off, G), b = __shfl_xor(kmax, off, G);
      kmin = a < kmin ? a : kmin;
      kmax = b > kmax ? b : kmax;
    }
    const uint32_t n = nv;
    // The smallest key gets bin 0 to itself and the other NB - 1 bins divide [kmin2, kmax], kmin2 = the smallest key
    // above it: a precipitation series is ~70 % exact zeros followed by a gap of most of the key range (0 -> the
    // smallest wet amount), which would otherwise squeeze every wet day into a quarter of the bins and leave one bin
    // of hundreds of tied keys to be searched.  Bin 0 is constant by construction: its targets need no search.
    uint32_t kmin2 = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < KPL; ++k) kmin2 = (key[k] > kmin && key[k] < kmin2) ? key[k] : kmin2;  // NaN keys are 0xFFFFFFFF
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) {
      const uint32_t a = __shfl_xor(kmin2, off, G);
      kmin2 = a < kmin2 ? a : kmin2;
    }
    kmin2 = kmin2 == 0xFFFFFFFFu ? kmin : kmin2;  // all valid keys equal
    // copies of the smallest key are counted in registers, never touch the LDS atomics (hundreds of lanes on one
    // address) and are not stored: nobody reads bin 0
    uint32_t cnt0 = 0;
#pragma unroll
    for (int k = 0; k < KPL; ++k) cnt0 += (key[k] == kmin && key[k] != 0xFFFFFFFFu) ? 1u : 0u;
#pragma unroll
    for (int off = G / 2; off > 0; off >>= 1) cnt0 += __shfl_xor(cnt0, off, G);
    const uint32_t range = n > 0 ? kmax - kmin2 : 0u;
    // smallest shift with (range >> shift) < NB; the top value bin is merged into bin NB - 1 when that index is reached
    int shift = 32 - __clz((int)range) - (31 - __clz(NB));  // bits(range) - log2(NB)
    shift = (range == 0u || shift < 0) ? 0 : shift;
    const XhValueBins vb = xh_value_bins(kmin2, kmax, n > 0, NB - 1);  // (keys on both sides of zero: bins linear in the value)
    auto binof = [&](uint32_t kk) -> uint32_t {
      const uint32_t b = 1u + (vb.on ? xh_value_bin(vb, kk) : ((kk - kmin2) >> shift));
      return kk == kmin ? 0u : (b < (uint32_t)NB ? b : (uint32_t)NB - 1u);
    };
#pragma unroll
    for (int b = 0; b < BPL; ++b) cur[l + b * G] = 0;
    wave_sync();
    if (!(abl & 1)) {
#pragma unroll
    for (int k = 0; k < KPL; ++k)
      if (key[k] != 0xFFFFFFFFu && key[k] != kmin) atomicAdd(&cur[binof(key[k])], 1u);
    }
    wave_sync();
    if (l == 0) cur[0] = cnt0;
    wave_sync();
    {
      uint32_t loc[BPL], s = 0;
#pragma unroll
      for (int b = 0; b < BPL; ++b) { loc[b] = cur[l * BPL + b]; s += loc[b]; }
      uint32_t incl = s;
#pragma unroll
      for (int off = 1; off < G; off <<= 1) {
        uint32_t o = __shfl_up(incl, off, G);
        if (l >= off) incl += o;
      }
      uint32_t run = incl - s;
      wave_sync();
#pragma unroll
      for (int b = 0; b < BPL; ++b) { cur[l * BPL + b] = run; run += loc[b]; }
    }
    wave_sync();
    if (!(abl & 2)) {
      // scatter, atomics issued in batches of 8 so that their latencies overlap
#pragma unroll
      for (int k0 = 0; k0 < KPL; k0 += 8) {
        uint32_t pos[8];
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k0 + k < KPL)
            pos[k] = (key[k0 + k] != 0xFFFFFFFFu && key[k0 + k] != kmin) ? atomicAdd(&cur[binof(key[k0 + k])], 1u) : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k)
          if (k0 + k < KPL && key[k0 + k] != 0xFFFFFFFFu && key[k0 + k] != kmin) sorted[pos[k]] = key[k0 + k];
      }
    }
    if (l == 0) cur[0] = cnt0;  // END of bin 0 (its cursor never moved)
    wave_sync();
    // after the scatter cur[b] is the END of bin b (== start of bin b+1)
    int round = 0;
    for (int tgt = l; tgt < 2 * nq && !(abl & 4); tgt += G, ++round) {
      float v = xh_nan32();
      if (n >= 1) {
        // the rank depends on (n, target) only: cached per lane and recomputed when the valid count changes
        if (n != rank_n) {
#pragma unroll
          for (int rr = 0; rr < NROUND; ++rr) {
            const int tg = l + rr * G;
            int r0 = 0;
            if (tg < 2 * nq && !(T == 1 || n < 2)) {
              double nn = (double)n, q = qs[tg >> 1];
              double vi = nn * q + (1.0 + q * (1.0 - 1.0 - 1.0)) - 1.0;  // utl:395 with alpha = beta = 1
              if (vi >= nn - 1.0) r0 = (int)n - 1;
              else if (vi < 0.0) r0 = 0;
              else r0 = (int)floor(vi) + (tg & 1);
            }
            rank_c[rr] = r0;
          }
          rank_n = n;
        }
        int r = 0;
#pragma unroll
        for (int rr = 0; rr < NROUND; ++rr) r = (rr == round) ? rank_c[rr] : r;
        // first bin whose end exceeds r
        int lo = -1, hi = NB - 1;  // invariant: end[lo] <= r < end[hi]   (end[-1] = 0, end[NB-1] = n)
        if (abl & 16) { hi = 1 + (int)(((uint32_t)r * (uint32_t)(NB - 1)) / (n + 1u)); lo = hi - 1; }  // diagnostics: no search
        while (hi - lo > 1) {
          int mid = (lo + hi) >> 1;
          if (cur[mid] <= (uint32_t)r) lo = mid; else hi = mid;
        }
        const uint32_t s0 = hi > 0 ? cur[hi - 1] : 0u, s1 = cur[hi];
        const uint32_t kth = (uint32_t)r - s0, m = s1 - s0;
        uint32_t ans;
        if (hi == 0) ans = kmin;  // bin 0 holds the copies of the smallest key only
        else if (abl & (16 | 32)) ans = sorted[s0 < n ? s0 : 0u];  // diagnostics: no in-bin selection (never loops)
        else if (m <= 8) {
          // exact k-th smallest of <= 8 keys: independent LDS loads, optimal 19-comparator network (0xFFFFFFFF pads
          // sort last), k-th picked with an OR of masked values (no dynamic register indexing)
          uint32_t kk[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) kk[i] = (uint32_t)i < m ? sorted[s0 + i] : 0xFFFFFFFFu;
#define XH_CE8(i, j) { uint32_t a_ = kk[i] < kk[j] ? kk[i] : kk[j], b_ = kk[i] < kk[j] ? kk[j] : kk[i]; kk[i] = a_; kk[j] = b_; }
          XH_CE8(0, 1) XH_CE8(2, 3) XH_CE8(4, 5) XH_CE8(6, 7) XH_CE8(0, 2) XH_CE8(1, 3) XH_CE8(4, 6) XH_CE8(5, 7)
          XH_CE8(1, 2) XH_CE8(5, 6) XH_CE8(0, 4) XH_CE8(3, 7) XH_CE8(1, 5) XH_CE8(2, 6) XH_CE8(1, 4) XH_CE8(3, 6)
          XH_CE8(2, 4) XH_CE8(3, 5) XH_CE8(3, 4)
#undef XH_CE8
          ans = 0u;
#pragma unroll
          for (int i = 0; i < 8; ++i) ans |= ((uint32_t)i == kth) ? kk[i] : 0u;
        } else {
          ans = select_in_bin(sorted + s0, m, kth);
        }
        v = xh_key2f(ans);
      }
      vals[tgt] = v;
    }
    wave_sync();
    for (int j = l; j < nq && !(abl & 8); j += G) {
      if (have) {
        double r;
        if (n == 0) r = xh_nan64();
        else if (T == 1 || n < 2) r = (double)vals[2 * j];
        else {
          double nn = (double)n, q = qs[j];
          double vi = nn * q + (1.0 + q * (1.0 - 1.0 - 1.0)) - 1.0;
          float left = vals[2 * j], right = vals[2 * j + 1];
          if (vi >= nn - 1.0 || vi < 0.0) r = (double)left;
          else {
            double gamma = vi - floor(vi);
            float diff = right - left;
            r = (double)left + (double)diff * gamma;
            if (gamma >= 0.5) r = (double)right - (double)diff * (1.0 - gamma);
          }
        }
        out[col * out_cstride + (int64_t)j * out_qstride] = (float)r;
      }
    }
    if (STAGE) block_sync_lds();  // every wave is done with sorted[] before the next tile overwrites it
    else wave_sync();
  }
}

template <int G, bool TM>
static int launch_select_grp_g(xh_ctx* ctx, const float* x, int64_t T, int64_t ncols, int64_t stride, const double* d_q,
                               int nq, float* out, int64_t out_cstride, int64_t out_qstride) {
  constexpr int COLS = 256 / G;
  int64_t nblk = cdiv64(ncols, COLS);
  int64_t maxblk = (int64_t)ctx->num_cu * 64;
  if (nblk > maxblk) nblk = maxblk;
  const char* ea = xh_diag_env("XH_SELECT_ABL");  // diagnostics: skip phases (results become wrong)
  const int abl = ea ? atoi(ea) : 0;
  if (TM && G == 32 && !xh_diag_env("XH_SELECT_NOSTAGE")) {
    constexpr int SG = 32;  // (fixed: the staged kernel is only instantiated for 32 lanes per column)
    int64_t nb2 = cdiv64(ncols, 512 / SG);
    if (nb2 > (int64_t)ctx->num_cu * 32) nb2 = (int64_t)ctx->num_cu * 32;
    if (T <= 384)
      hipLaunchKernelGGL((k_select_grp<SG, 384 / SG, 256, true, 512, true>), dim3((unsigned)nb2), dim3(512), 0, ctx->stream, x,
                         T, ncols, stride, d_q, nq, out, out_cstride, out_qstride, abl);
    else
      hipLaunchKernelGGL((k_select_grp<SG, 512 / SG, 256, true, 512, true>), dim3((unsigned)nb2), dim3(512), 0, ctx->stream, x,
                         T, ncols, stride, d_q, nq, out, out_cstride, out_qstride, abl);
    XH_LAUNCH_CHECK();
    return XH_OK;
  }
  if (T <= 384)
    hipLaunchKernelGGL((k_select_grp<G, 384 / G, 256, TM>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, x, T, ncols,
                       stride, d_q, nq, out, out_cstride, out_qstride, abl);
  else
    hipLaunchKernelGGL((k_select_grp<G, 512 / G, 256, TM>), dim3((unsigned)nblk), dim3(256), 0, ctx->stream, x, T, ncols,
                       stride, d_q, nq, out, out_cstride, out_qstride, abl);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

// lanes per column for short series; XH_SELECT_G (8 | 16 | 32) overrides the tuned default for experiments
// measured on MI355X, T = 365, 1 036 800 columns: time-major 32 lanes/column 2.2 ms (64: 4.0, 16: 2.6, 8: 6.1);
// time-minor 64 lanes/column 1.4 ms (32: 1.44, 16: 1.8)
static int select_group_size(bool time_major) {
  const char* e = xh_diag_env("XH_SELECT_G");
  int g = e ? atoi(e) : (time_major ? 32 : 64);
  if (g != 8 && g != 16 && g != 32 && g != 64) g = time_major ? 32 : 64;
  return g;
}

template <bool TM>
static int launch_select_grp(xh_ctx* ctx, const float* x, int64_t T, int64_t ncols, int64_t stride, const double* d_q,
                             int nq, float* out, int64_t out_cstride, int64_t out_qstride) {
  switch (select_group_size(TM)) {
    case 8: return launch_select_grp_g<8, TM>(ctx, x, T, ncols, stride, d_q, nq, out, out_cstride, out_qstride);
    case 32: return launch_select_grp_g<32, TM>(ctx, x, T, ncols, stride, d_q, nq, out, out_cstride, out_qstride);
    case 64: return launch_select_grp_g<64, TM>(ctx, x, T, ncols, stride, d_q, nq, out, out_cstride, out_qstride);
    default: return launch_select_grp_g<16, TM>(ctx, x, T, ncols, stride, d_q, nq, out, out_cstride, out_qstride);
  }
}

// quantiles for short series straight from a time-major (T, C) view (no transpose); returns XH_ERR_NOTIMPL when
// the shape does not fit so that the caller can fall back to the transposed path
int xh_select_time_major(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const double* d_q, int nq,
                         float* out, int64_t out_cstride, int64_t out_qstride) {
  if (T > 512 || nq > 64) return XH_ERR_NOTIMPL;
  {  // one-year daily series: register sorting network (select3.hip)
    const int rc = xh_select_regsort(ctx, x, T, C, st, d_q, nq, out, out_cstride, out_qstride);
    if (rc != XH_ERR_NOTIMPL) return rc;
  }
  return launch_select_grp<true>(ctx, x, T, C, st, d_q, nq, out, out_cstride, out_qstride);
}

template <int NT, int KPL, int NB>
static int launch_select(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride, const double* d_q,
                         int nq, float* out, int64_t out_cstride, int64_t out_qstride) {
  constexpr int BLOCK = NT <= 256 ? 256 : NT;
  constexpr int GROUPS = BLOCK / NT;
  int Tpad = (int)((T + 63) & ~(int64_t)63);
  size_t lds = (size_t)GROUPS * (Tpad + (NB + 1) + NB + 128 + 16) * sizeof(uint32_t);
  XH_REQUIRE(lds <= 160 * 1024, XH_ERR_LIMIT, "quantile_series: LDS need %zu exceeds 160 KiB", lds);
  auto kern = k_select_quantile<NT, KPL, NB>;
  if (lds > 64 * 1024)
    XH_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int64_t nblk = cdiv64(ncols, GROUPS);
  int64_t maxblk = (int64_t)ctx->num_cu * 16;
  if (nblk > maxblk) nblk = maxblk;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(BLOCK), lds, ctx->stream, xcols, T, ncols, col_stride, d_q, nq, out,
                     out_cstride, out_qstride);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

// Entry used by eqm.hip: quantiles of `ncols` contiguous columns (time-minor view).
int xh_select_columns(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride, const double* d_q,
                      int nq, float* out, int64_t out_cstride, int64_t out_qstride) {
  XH_REQUIRE(nq <= 64, XH_ERR_LIMIT, "quantile_series: at most 64 quantiles");
  if (T <= 512) return launch_select_grp<false>(ctx, xcols, T, ncols, col_stride, d_q, nq, out, out_cstride, out_qstride);
  // 512 < T <= 1024: one WAVE per column with the column in LDS (the per-column fixed costs of a whole workgroup are
  // 3x slower here: measured 11.1 vs 3.55 ms at T = 800, 1 036 800 columns)
  if (T <= 1024) return launch_select<64, 16, 512>(ctx, xcols, T, ncols, col_stride, d_q, nq, out, out_cstride, out_qstride);
  // longer series: one workgroup per column, keys in registers, list-free selection (select2.hip)
  {
    int rc_lean = xh_select_columns_lean(ctx, xcols, T, ncols, col_stride, d_q, nq, out, out_cstride, out_qstride);
    if (rc_lean != XH_ERR_NOTIMPL) return rc_lean;
  }
  // T > 16384: the whole column lives in LDS (sorted[T]), 1024-thread workgroups
  // beyond 32768 samples (1950-2100 daily = 55 152): the radix select of select5.hip, any length
  if (T > 32768) return xh_select_columns_radix(ctx, xcols, T, ncols, col_stride, d_q, nq, out, out_cstride, out_qstride);
  return launch_select<1024, 32, 2048>(ctx, xcols, T, ncols, col_stride, d_q, nq, out, out_cstride, out_qstride);
}

// ---- the NaN nodes of the selection results (called by eqm.hip after every quantile_series)
// utl:552-554 — "when an interpolation is in NaN range ... clip to the array max value": a NaN node of a column that has
// valid samples becomes the column's largest valid sample.  The selection kernels produce such NaNs only from infinities
// (inf - inf in the lerp between two order statistics), so this is a pass over the (nq, C) nodes — 1/18 of the series'
// bytes at nq = 20, T = 365 — plus one scan of each column that actually holds one; columns without valid samples keep
// their NaN.  out: (nq, C) with unit column stride.
// Round 6: two passes (before: one thread per column scanning its T rows one load at a time — a field with a land / sea mask
// has NaN nodes in 30 % of its columns, every wave held one, and the fix-up took 4.7 ms behind a 0.55 ms selection at
// T = 10950, 0.1 ms behind 0.03 at T = 365).  Pass 1 keeps the elementwise shape and only LISTS the 64-column groups that
// hold a NaN node; pass 2 is a fixed grid of 1024-thread workgroups that finds nothing to do in the common case.
constexpr int NF_WAVES = 16;
// the largest valid sample of the workgroup's 64 columns that ask for it (want; lane = column), rows split over the 16 waves;
// returns (to wave 0's lanes) whether the column has a valid sample, its maximum in m
__device__ __forceinline__ bool nf_colmax(const float* __restrict__ x, int64_t T, int64_t st, int64_t sc, int64_t c, bool want, int lane,
                                          int w, float (*s_max)[64], int (*s_any)[64], float& m) {
  const int64_t chunk = cdiv64(T, NF_WAVES);
  const int64_t ta = (int64_t)w * chunk;
  int64_t tb = ta + chunk;
  if (tb > T) tb = T;
  m = __uint_as_float(0xFF800000u);
  bool any = false;
  if (want) {
    const float* __restrict__ xc = x + c * sc;
    int64_t t = ta;
    for (; t + 8 <= tb; t += 8) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = xc[(t + u) * st];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v[u] == v[u]) { any = true; m = v[u] > m ? v[u] : m; }
    }
    for (; t < tb; ++t) {
      const float v = xc[t * st];
      if (v == v) { any = true; m = v > m ? v : m; }
    }
  }
  __syncthreads();   // (the arrays may still be read from the previous call)
  s_max[w][lane] = m;
  s_any[w][lane] = any ? 1 : 0;
  __syncthreads();
  if (w == 0) {
#pragma unroll
    for (int k = 0; k < NF_WAVES; ++k) {
      any |= s_any[k][lane] != 0;
      m = s_max[k][lane] > m ? s_max[k][lane] : m;
    }
  }
  return any;
}

// Pass 1 (one thread per column, the shape of an elementwise kernel): look at the column's nodes.  CORR (EQM training: af holds
// ref_q on entry): a column without a NaN node gets its correction here — the bytes an elementwise correction reads anyway, so
// the rule costs the common case nothing.  A wave (= 64 adjacent columns) that holds a NaN node appends its group to `list`.
template <bool CORR>
__global__ void __launch_bounds__(XH_BLOCK)
k_nanfix_flag(int64_t C, int nq, int kind, float* __restrict__ af, float* __restrict__ hist_q, uint32_t* __restrict__ list,
              uint32_t* __restrict__ count, unsigned char* __restrict__ flags) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  int bad = 0;
  if (c < C) {
    for (int j = 0; j < nq; ++j) {
      const float r = af[(int64_t)j * C + c];
      bad |= r != r ? 1 : 0;
      if (CORR) {
        const float h = hist_q[(int64_t)j * C + c];
        bad |= h != h ? 2 : 0;
      }
    }
    if (CORR && !bad)
      for (int j = 0; j < nq; ++j) {
        const float r = af[(int64_t)j * C + c], h = hist_q[(int64_t)j * C + c];
        af[(int64_t)j * C + c] = kind == 0 ? (r - h) : (r / h);
      }
  }
  // (pass 2 must see THESE flags: a corrected column of a listed group may hold a NaN of its own — 0 / 0 — that is no node)
  if (__ballot(bad != 0) != 0ull) {
    if (c < C) flags[c] = (unsigned char)bad;
    if ((threadIdx.x & 63) == 0) list[atomicAdd(count, 1u)] = (uint32_t)(c >> 6);
  }
}

// Pass 2 (a fixed grid; workgroup = 64 adjacent columns x 16 row chunks, looping over the listed groups — none in the common
// case): the NaN nodes of a column that has valid samples become its largest valid sample (the series are scanned by all 16
// waves, 8 loads in flight per lane, 256-byte row segments); CORR: then the correction of the group's flagged columns.
template <bool CORR>
__global__ void __launch_bounds__(64 * NF_WAVES)
k_nanfix_scan(const float* __restrict__ ref, const float* __restrict__ hist, int64_t T, int64_t C, int64_t st, int64_t sc, int nq, int kind,
              float* __restrict__ af, float* __restrict__ hist_q, const uint32_t* __restrict__ list, const uint32_t* __restrict__ count,
              const unsigned char* __restrict__ flags) {
  __shared__ float s_max[NF_WAVES][64];
  __shared__ int s_any[NF_WAVES][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const uint32_t n = *count;
  for (uint32_t i = blockIdx.x; i < n; i += gridDim.x) {   // (uniform)
    const int64_t c = (int64_t)list[i] * 64 + lane;
    const int bad = c < C ? (int)flags[c] : 0;   // (pass 1's view of the nodes)
    float mr = 0.f, mh = 0.f;
    bool fixr = nf_colmax(ref, T, st, sc, c, (bad & 1) != 0, lane, w, s_max, s_any, mr) && (bad & 1) != 0;
    bool fixh = false;
    if (CORR) fixh = nf_colmax(hist, T, st, sc, c, (bad & 2) != 0, lane, w, s_max, s_any, mh) && (bad & 2) != 0;
    if (w != 0 || c >= C || bad == 0) continue;   // (the other columns of the group were corrected in pass 1)
    for (int j = 0; j < nq; ++j) {
      float r = af[(int64_t)j * C + c];
      if (fixr && r != r) r = mr;
      if (CORR) {
        float h = hist_q[(int64_t)j * C + c];
        if (fixh && h != h) {
          h = mh;
          hist_q[(int64_t)j * C + c] = h;
        }
        af[(int64_t)j * C + c] = kind == 0 ? (r - h) : (r / h);
      } else af[(int64_t)j * C + c] = r;
    }
  }
}

template <bool CORR>
static int nanfix_launch(xh_ctx* ctx, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st, int64_t sc, int nq, int kind,
                         float* af, float* hist_q) {
  if (C <= 0 || T <= 0) return XH_OK;
  const int64_t groups = cdiv64(C, 64);
  void* ws = nullptr;
  int rc = xh_big_scratch(ctx, sizeof(uint32_t) * (size_t)(groups + 16) + (size_t)groups * 64, &ws);   // (the selection kernels are done with it: stream order)
  if (rc) return rc;
  uint32_t* count = static_cast<uint32_t*>(ws);
  uint32_t* list = count + 16;
  unsigned char* flags = reinterpret_cast<unsigned char*>(list + groups);
  XH_CHECK_HIP(hipMemsetAsync(count, 0, 64, ctx->stream));
  hipLaunchKernelGGL((k_nanfix_flag<CORR>), dim3((unsigned)cdiv64(C, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, C, nq, kind, af, hist_q, list, count, flags);
  XH_LAUNCH_CHECK();
  const int64_t grid = groups < 2048 ? groups : 2048;
  hipLaunchKernelGGL((k_nanfix_scan<CORR>), dim3((unsigned)grid), dim3(64 * NF_WAVES), 0, ctx->stream, ref, hist, T, C, st, sc, nq, kind, af,
                     hist_q, (const uint32_t*)list, (const uint32_t*)count, (const unsigned char*)flags);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_nanmax_fix(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int nq, float* out) {
  return nanfix_launch<false>(ctx, x, nullptr, T, C, st, sc, nq, 0, out, nullptr);
}

int xh_correction_fix(xh_ctx* ctx, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st, int64_t sc, int nq, int kind,
                      float* af, float* hist_q) {
  return nanfix_launch<true>(ctx, ref, hist, T, C, st, sc, nq, kind, af, hist_q);
}
