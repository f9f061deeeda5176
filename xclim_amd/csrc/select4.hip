// select4.hip — exact per-column quantiles of LONG time-major series in TWO STREAMING PASSES, nothing transposed
// (xsdba nbutils.quantile, E1 of SURVEY.md §8a; Hyndman-Fan type 7 = /root/reference/src/xclim/core/utils.py:370-395,
// 464-491 with alpha = beta = 1; reference call site /root/reference/src/xclim/sdba.py:10).
//
// Why: for T ~ 10^4 the column-at-a-time selection of select2.hip needs the time-major (T, C) input transposed through
// HBM first (read 4E + write 4E + read 4E for a 4E kernel) and is VALU-bound on top of that (profiles/r02/
// eqm_c4_anatomy.txt: 33.5 ms per 45.4 GB array).  Here every pass marches along time over the rows as they lie in
// memory, like the run-length kernels do, and the per-column state is a histogram in LDS:
//
//   pass 0  k_hs_sample   every S-th row (~342 rows, 3 % of the bytes): per column the 4th smallest / 4th largest sampled
//                          key -> a value window [lo, hi] that holds ~97.5 % of the column (keys = order-preserving uint32)
//   pass 1  k_hs_hist     all rows: 1024-bin histogram per column over that window (packed u16 counters in LDS,
//                          [bin / 2][column], LDS atomics).  Bins: 0 keys < lo | 1 keys == lo | 2..1021 regular, equally wide in value
//                          | binH keys == hi | binH + 1 keys > hi.  Bins 1 and binH hold ONE value each ("pure": the dry
//                          days of a precipitation series, a saturated maximum) and never need a second look.  At the
//                          end of a tile: prefix sums, the bin and the rank inside its bin of each of the 2 nq order
//                          statistics (Hyndman-Fan, utl:395, 417-461), a bitmap of those target bins, and the position
//                          every target will have among the column's sorted CANDIDATES (= the keys of the target bins).
//   pass 2  k_hs_collect  all rows again: keys whose bin is a target bin (~3 % of them) are appended to the column's
//                          list in LDS; at the end of a tile one wave per column sorts its candidates (<= 2048) in registers
//                          (bitonic, ds_bpermute exchanges), picks the 2 nq order statistics by position, lerps
//                          (utl:464-491) and stores the nq quantiles.
//
// Geometry: a 512-thread workgroup owns 32 adjacent columns (one 128-byte line per row) and 16 "row lanes" per column;
// thread (column c, row lane r) loads rows r, r + 16, r + 32 ... with 16 loads in flight, double buffered; a wave
// instruction reads two full 128-byte row segments.  64 KB of histogram (or candidate list) per workgroup, two
// workgroups per CU.  Algorithmic bytes per pass: 4E; nothing else crosses HBM except ~230 bytes of per-column tables.
//
// Columns whose target bins hold more than 2048 keys, or that do not fit the tile's 16384-key LDS pool (heavily tied or
// clustered values away from the window's ends), are flagged in pass 1 and recomputed by the column kernels of select.hip / select2.hip from a gathered copy.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int HS_RL = 16;              // row lanes per column: a workgroup owns CW columns with CW * 16 threads
constexpr int HS_NT = 512;             // (threads of the diagnostic streaming kernel)
constexpr int HS_UDEF = 16;            // loads in flight per thread and register set (two sets: ping-pong)
constexpr int HS_NB = 1024;            // bins per column
constexpr int HS_NREG = 1020;          // regular bins 2 .. 1021
constexpr int HS_CAPMAX = 2048;        // ... and at most this many for one column (the largest register sort)
constexpr int HS_MAXQ = 32;            // quantiles per call on this path
constexpr uint32_t HS_NANKEY = 0xFFFFFFFFu;
constexpr uint32_t HS_SPEC_LO = 0xFFFEu, HS_SPEC_HI = 0xFFFDu, HS_SPEC_NONE = 0xFFFFu;
constexpr uint32_t HS_FLAGGED = 0xFFFFFFFFu;

struct HsStat {
  uint32_t nflag;   // columns handed to the column kernels
  uint32_t maxm;    // largest candidate count of any column
  uint32_t errors;  // pass 2 met a different candidate count than pass 1 announced (must stay 0)
  uint32_t summ;    // candidates of all columns (diagnostics: the mean per column)
};

// per-column window -> bin arithmetic, identical in pass 1 and pass 2.  Bins are equally wide in VALUE, not in key
// space: keys are logarithmic in |x| (a series that straddles zero — degrees Celsius, anomalies — would put all of its
// samples into a handful of key-space bins around the huge key range of the tiny values).  fp32 subtract, multiply and
// floor are monotone, and monotone + identical in both passes is all the selection needs:
//   key < lo -> bin 0 | key == lo -> 1 | otherwise 2 + min(floor((x - lo) * scale), 1019) | key == hi -> binH | key > hi -> binH + 1
// The class offsets come from saturating key arithmetic (no compares: v_cmp + v_cndmask pairs serialise on VCC).
struct HsScale {
  uint32_t lo1, hi1, binH;
  float lof, scale;
};

__device__ __forceinline__ uint32_t hs_usub_sat(uint32_t a, uint32_t b) { return __builtin_elementwise_sub_sat(a, b); }
__device__ __forceinline__ uint32_t hs_min(uint32_t a, uint32_t b) { return a < b ? a : b; }

__device__ __forceinline__ uint32_t hs_regular(float x, float lof, float scale) {
  // median of (t, 0, 1019): NaN (inf * 0, inf - inf) and negative t give 0, large t the top regular bin
  return (uint32_t)__builtin_amdgcn_fmed3f((x - lof) * scale, 0.0f, (float)(HS_NREG - 1));
}

__device__ __forceinline__ HsScale hs_scale(uint32_t lo, uint32_t hi) {
  HsScale s;
  s.lo1 = lo + 1u;  // lo is a sampled (non-NaN) key: <= 0xFF800000
  s.lof = xh_key2f(lo);
  const float hif = xh_key2f(hi);
  const float d = hif - s.lof;
  float sc = 0.0f;
  if (d > 0.0f) sc = (float)((double)HS_NREG * 1.000001 / (double)d);  // (hi - lo) * scale >= 1020: hi lands in the top regular bin
  if (!(sc <= 3.0e38f)) sc = 0.0f;                                       // d = inf, or a window of denormal width
  s.scale = sc;
  s.hi1 = hi > lo ? hi - 1u : HS_NANKEY;  // one-valued window: no "== hi" class (bin 1 holds the value)
  s.binH = hi > lo ? 3u + hs_regular(hif, s.lof, sc) : HS_SPEC_NONE;
  return s;
}

// order-preserving key WITHOUT the NaN test of xh_f2key: NaN bit patterns land above key(+inf) = 0xFF800000 or below
// key(-inf) = 0x007FFFFF and are recognised by hs_valid
__device__ __forceinline__ uint32_t hs_key(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ bool hs_valid(uint32_t k) { return k - 0x007FFFFFu <= 0xFF000001u; }

__device__ __forceinline__ uint32_t hs_bin(float x, uint32_t k, const HsScale& s) {
  const uint32_t up = hs_min(hs_usub_sat(k, s.hi1), 2u);  // 0 | 1 (key == hi) | 2 (above)
  const uint32_t dn = hs_min(hs_usub_sat(s.lo1, k), 2u);  // 0 | 1 (key == lo) | 2 (below)
  return hs_regular(x, s.lof, s.scale) + (2u + up) - dn;
}

// rank of target (quantile q, side 0 = lower / 1 = upper neighbour) among n valid samples: utl:395, 417-461 (type 7)
__device__ __forceinline__ uint32_t hs_rank(uint32_t n, double q, int side) {
  if (n < 2u) return 0u;
  const double nn = (double)n;
  const double vi = nn * q + (1.0 + q * (1.0 - 1.0 - 1.0)) - 1.0;
  if (vi >= nn - 1.0) return n - 1u;
  if (vi < 0.0) return 0u;
  return (uint32_t)floor(vi) + (uint32_t)side;
}

// ---- pass 0: window from a row sample ------------------------------------------------------------------------------
// smallest-four / largest-four insertion (a0 <= a1 <= a2 <= a3 the four smallest keys so far, b0 >= ... the largest)
struct HsTop4 {
  uint32_t a0, a1, a2, a3, b0, b1, b2, b3, nv;
};
__device__ __forceinline__ void hs_top4_init(HsTop4& s) {
  s.a0 = s.a1 = s.a2 = s.a3 = 0xFFFFFFFFu;
  s.b0 = s.b1 = s.b2 = s.b3 = 0u;
  s.nv = 0u;
}
__device__ __forceinline__ void hs_ce(uint32_t& lo, uint32_t& hi) {
  const uint32_t a = lo, b = hi;
  lo = a < b ? a : b;
  hi = a < b ? b : a;
}
__device__ __forceinline__ void hs_top4_add(HsTop4& s, float f) {
  const uint32_t u = xh_f2key(f);
  const bool ok = u != HS_NANKEY;
  s.nv += ok ? 1u : 0u;
  const uint32_t kmin = u;               // the NaN key is the largest key: never among the four smallest
  const uint32_t kmax = ok ? u : 0u;     // ... and must not count among the largest
  s.a3 = kmin < s.a3 ? kmin : s.a3;
  hs_ce(s.a2, s.a3);
  hs_ce(s.a1, s.a2);
  hs_ce(s.a0, s.a1);
  s.b3 = kmax > s.b3 ? kmax : s.b3;
  hs_ce(s.b3, s.b2);
  hs_ce(s.b2, s.b1);
  hs_ce(s.b1, s.b0);
}

template <int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_hs_sample(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, int64_t S, int64_t ns, uint2* __restrict__ lohi) {
  const int64_t c0 = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c0 >= C) return;
  HsTop4 s[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) hs_top4_init(s[v]);
  const float* p = x + c0 + (S / 2) * st;
  const int64_t step = S * st;
  int64_t i = 0;
  for (; i + 8 <= ns; i += 8) {  // 8 sampled rows in flight per lane
    VecF<VEC> buf[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) buf[u] = xh_load<VEC>(p + (i + u) * step);
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int v = 0; v < VEC; ++v) hs_top4_add(s[v], buf[u].v[v]);
  }
  for (; i < ns; ++i) {
    const VecF<VEC> b = xh_load<VEC>(p + i * step);
#pragma unroll
    for (int v = 0; v < VEC; ++v) hs_top4_add(s[v], b.v[v]);
  }
#pragma unroll
  for (int v = 0; v < VEC; ++v) {
    uint32_t lo, hi;
    if (s[v].nv >= 8u) {  // the 4th smallest and the 4th largest of >= 8 samples: lo <= hi
      lo = s[v].a3;
      hi = s[v].b3;
    } else if (s[v].nv >= 1u) {
      lo = s[v].a0;
      hi = s[v].b0;
    } else {  // no valid sample: any window works (everything lands in the bins "< lo" / "> hi")
      lo = 0x80000000u;
      hi = 0x80000000u;
    }
    lohi[c0 + v] = make_uint2(lo, hi);
  }
}

// ---- the streaming loop shared by pass 1 and pass 2 ------------------------------------------------------------------
// Thread (col, rl) of a tile visits rows rl, rl + 16, ... in batches of HS_U rows; f(values, keys) is called once per batch
// with the order-preserving keys of hs_key (NaN patterns fail hs_valid; 0xFFFFFFFF for rows past the end and for the columns
// past C of a ragged tile).
template <int HS_U, int RLT, typename F>
__device__ __forceinline__ void hs_stream(const float* __restrict__ x, int T, int64_t st, int64_t cc, bool cvalid, int rl, F&& f) {
  constexpr int HS_ROWS = RLT * HS_U;  // rows a workgroup covers per batch
  // full batches: every row of the batch exists for every row lane.  Buffer loads: a descriptor re-based per batch
  // (scalar), the row offset of load u as the scalar offset, ONE 32-bit per-lane byte offset — no vector address
  // arithmetic and no 64-bit address registers per load (16 loads in flight would hold 32 of them).
  const uint32_t voff = (uint32_t)(((int64_t)rl * st + cc) * 4);
  const uint32_t rowstep = (uint32_t)(st * 4 * RLT);  // bytes between the rows of loads u and u + 1 (host: 256 rows < 4 GiB)
  const int nfull = T / HS_ROWS;
  const uint32_t padkey = cvalid ? 0u : HS_NANKEY;  // OR-ed into the key: a column past C only ever shows NaN keys
  auto load = [&](float (&dst)[HS_U], int kb) {
    const float* base = x + (int64_t)kb * HS_ROWS * st;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
    uint32_t soff = 0u;
#pragma unroll
    for (int u = 0; u < HS_U; ++u) {
      dst[u] = __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
      soff += rowstep;
    }
  };
  auto proc = [&](const float (&src)[HS_U]) {
    uint32_t k[HS_U];
#pragma unroll
    for (int u = 0; u < HS_U; ++u) k[u] = hs_key(src[u]) | padkey;
    f(src, k);
  };
  if (nfull > 0) {
    // two register sets, ping-pong (no copies: a copy of set B into set A right after element u is consumed would make
    // element u wait for the load that was issued a moment ago): the loads of one batch fly while the other is consumed
    float A[HS_U], B[HS_U];
    load(A, 0);
    int kb = 0;
    while (kb + 2 < nfull) {  // no conditional loads in here: the compiler counts the outstanding loads exactly
      load(B, kb + 1);
      proc(A);
      load(A, kb + 2);
      proc(B);
      kb += 2;
    }
    if (kb + 1 < nfull) {
      load(B, kb + 1);
      proc(A);
      proc(B);
    } else {
      proc(A);
    }
  }
  // tail rows (fewer than 256): same buffer loads from CLAMPED per-lane rows, validity applied afterwards (a conditional
  // load costs a full s_waitcnt vmcnt(0)).  rl passes through an opaque copy: hoisted out of the tile loop, the 16
  // clamped offsets would stay live (and spill) across the whole kernel.
  const int t0 = nfull * HS_ROWS;
  if (t0 < T) {
    int rlo = rl;
    asm volatile("" : "+v"(rlo));
    const int rem = T - t0;
    const float* base = x + (int64_t)t0 * st;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
    float buf[HS_U];
#pragma unroll
    for (int u = 0; u < HS_U; ++u) {
      int r = u * RLT + rlo;
      r = r < rem ? r : rem - 1;
      const uint32_t vo = (uint32_t)r * (uint32_t)(st * 4) + (uint32_t)(cc * 4);
      buf[u] = __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)vo, 0, 0));
    }
    uint32_t k[HS_U];
#pragma unroll
    for (int u = 0; u < HS_U; ++u) k[u] = u * RLT + rlo < rem ? (hs_key(buf[u]) | padkey) : HS_NANKEY;
    f(buf, k);
  }
}

// Workgroup -> tile map.  The dispatcher places block b on XCD b % 8; with tile = b the workgroups of one XCD touch
// every 8th 128-byte line of a row.  Optional map (XH_HIST_XCD=1, diagnostics): inside each round of gridDim tiles, XCD
// x takes the x-th CONTIGUOUS eighth (64 adjacent tiles = 8 KB of every row).  Measured on config 4: no gain (42.8 ms
// with, 41.3 without), so the identity is the default.  Needs gridDim % 8 == 0, identity otherwise.
__device__ __forceinline__ int64_t hs_tile_of(int64_t round_base, int64_t ntiles, int xcd_map) {
  const int64_t G = gridDim.x, b = blockIdx.x;
  const int64_t local = (xcd_map && (G & 7) == 0) ? (b & 7) * (G >> 3) + (b >> 3) : b;
  const int64_t tile = round_base + local;
  return tile < ntiles ? tile : -1;
}

// ---- pass 1: histogram + target bins -----------------------------------------------------------------------------------
// LDS: hist [512][32] u32 (two u16 counters per word: bins 2d, 2d + 1 of column c at [d][c]) | bm [32][32] target-bin
// bitmap | part [16][32] partial sums | tgt [2 * MAXQ][32] (bin | rank inside the bin << 16) | mcol [32] | cbase [32]
constexpr size_t hs_lds1(int cw) { return (size_t)(HS_NB / 2) * cw * 4 + 32 * cw * 4 + HS_RL * cw * 4 + 2 * HS_MAXQ * cw * 4 + 2 * cw * 4; }

template <int HS_U, int CW>
__global__ void __launch_bounds__(CW * HS_RL, 4)
k_hs_hist(const float* __restrict__ x, int T, int64_t C, int64_t st, const uint2* __restrict__ lohi,
          const double* __restrict__ qs, int nq, uint32_t* __restrict__ meta_n, uint32_t* __restrict__ meta_m,
          uint32_t* __restrict__ meta_base, uint16_t* __restrict__ crank, uint32_t* __restrict__ bitmap_g,
          uint32_t* __restrict__ flist, HsStat* __restrict__ stat, int xcd_map, int abl) {
  constexpr int NT = CW * HS_RL, POOL = CW * 512;  // threads per workgroup; LDS pool of candidate keys in pass 2
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
  uint32_t* bm = hist + (HS_NB / 2) * CW;
  uint32_t* part = bm + 32 * CW;
  uint32_t* tgt = part + HS_RL * CW;
  uint32_t* mcol = tgt + 2 * HS_MAXQ * CW;
  uint32_t* cbase = mcol + CW;
  const int tid = threadIdx.x, col = tid & (CW - 1), rl = tid / CW;
  const int ntgt = 2 * nq;
  const int64_t ntiles = (C + CW - 1) / CW;
  // zero the histogram and the bitmap (again at the end of every tile)
  for (int i = tid; i < (HS_NB / 2) * CW + 32 * CW; i += NT) hist[i] = 0u;
  __syncthreads();
  for (int64_t round_base = 0; round_base < ntiles; round_base += gridDim.x) {
    const int64_t tile = hs_tile_of(round_base, ntiles, xcd_map);
    if (tile < 0) break;  // (block-uniform; only in the last round)
    const int64_t c = tile * CW + col;
    const bool cvalid = c < C;
    const int64_t cc = cvalid ? c : C - 1;
    const uint2 lh = lohi[cc];
    const HsScale s = hs_scale(lh.x, lh.y);
    uint32_t* mycol = hist + col;
    uint32_t dummy = 0;
    hs_stream<HS_U, HS_RL>(x, T, st, cc, cvalid, rl, [&](const float (&v)[HS_U], const uint32_t (&k)[HS_U]) {
      if (abl & 2) {  // diagnostics: loads only
#pragma unroll
        for (int u = 0; u < HS_U; ++u) dummy ^= k[u];
        return;
      }
#pragma unroll
      for (int u = 0; u < HS_U; ++u) {
        const uint32_t b = hs_bin(v[u], k[u], s);
        const uint32_t val = hs_valid(k[u]) ? (1u << ((b & 1u) << 4)) : 0u;
        atomicAdd(mycol + (b >> 1) * CW, val);
      }
    });
    if ((abl & 2) && dummy == 0x12345679u) atomicAdd(&stat->errors, 1u);
    __syncthreads();
    if (abl & 4) continue;  // diagnostics: no tile epilogue (wrong results, the histogram is not even cleared)
    // ---- exclusive prefix sums, in place: thread (col, rl) owns words [rl * 32, rl * 32 + 32) = bins [rl * 64, ...)
    uint32_t ssum = 0;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const uint32_t w = hist[(rl * 32 + i) * CW + col];
      ssum += (w & 0xFFFFu) + (w >> 16);
    }
    part[rl * CW + col] = ssum;
    __syncthreads();
    uint32_t run = 0, n = 0;
#pragma unroll
    for (int r = 0; r < HS_RL; ++r) {
      const uint32_t p = part[r * CW + col];
      run += r < rl ? p : 0u;
      n += p;
    }
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const int idx = (rl * 32 + i) * CW + col;
      const uint32_t w = hist[idx];
      const uint32_t c0 = w & 0xFFFFu, c1 = w >> 16;
      hist[idx] = run | ((run + c0) << 16);  // counts below bin 2d | below bin 2d + 1 (<= n <= T <= 65535)
      run += c0 + c1;
    }
    __syncthreads();
    auto below = [&](uint32_t b) -> uint32_t {  // keys in bins < b, b in [0, 1024]
      if (b >= (uint32_t)HS_NB) return n;
      const uint32_t w = hist[(b >> 1) * CW + col];
      return (b & 1u) ? (w >> 16) : (w & 0xFFFFu);
    };
    // ---- the bin and the rank inside it of every target; mark the bins that need a second look
    for (int j = rl; j < ntgt; j += HS_RL) {
      uint32_t e = HS_SPEC_NONE;
      if (n > 0u) {
        const uint32_t r = hs_rank(n, qs[j >> 1], j & 1);
        uint32_t lo_b = 0u, hi_b = HS_NB - 1;  // largest b with below(b) <= r: that bin holds rank r
#pragma unroll 1
        for (int it = 0; it < 10; ++it) {
          const uint32_t mid = (lo_b + hi_b + 1u) >> 1;
          const bool le = below(mid) <= r;
          lo_b = le ? mid : lo_b;
          hi_b = le ? hi_b : mid - 1u;
        }
        e = lo_b | ((r - below(lo_b)) << 16);
        if (lo_b != 1u && lo_b != s.binH) atomicOr(&bm[(lo_b >> 5) * CW + col], 1u << (lo_b & 31u));
      }
      tgt[j * CW + col] = e;
    }
    __syncthreads();
    // ---- candidates below each target: keys of the marked bins, words [2 rl, 2 rl + 1] of the bitmap per thread
    auto marked_below = [&](int w, uint32_t limit_bit) -> uint32_t {  // keys of the marked bins of word w below bit `limit_bit`
      uint32_t bits = bm[w * CW + col];
      if (limit_bit < 32u) bits &= (1u << limit_bit) - 1u;
      uint32_t acc = 0;
      while (bits) {
        const uint32_t b = (uint32_t)w * 32u + (uint32_t)__ffs((int)bits) - 1u;
        bits &= bits - 1u;
        acc += below(b + 1u) - below(b);
      }
      return acc;
    };
    part[rl * CW + col] = marked_below(2 * rl, 32u) + marked_below(2 * rl + 1, 32u);
    __syncthreads();
    uint32_t m = 0;
#pragma unroll
    for (int r = 0; r < HS_RL; ++r) m += part[r * CW + col];
    // the tile's candidate lists share one LDS pool in pass 2: column offsets in column order; a column that does not
    // fit (or exceeds the largest register sort) is flagged for the column kernels
    if (rl == 0) mcol[col] = cvalid ? m : 0u;
    __syncthreads();
    if (tid == 0) {
      uint32_t runp = 0;
      for (int k = 0; k < CW; ++k) {
        const uint32_t mk = mcol[k];
        const bool fl = mk > (uint32_t)HS_CAPMAX || runp + mk > (uint32_t)POOL;
        cbase[k] = fl ? HS_FLAGGED : runp;
        runp += fl ? 0u : mk;
      }
    }
    __syncthreads();
    const uint32_t mybase = cbase[col];
    const bool flagged = mybase == HS_FLAGGED;
    for (int j = rl; j < ntgt; j += HS_RL) {
      const uint32_t e = tgt[j * CW + col];
      uint32_t cr = HS_SPEC_NONE;
      if (e != HS_SPEC_NONE) {
        const uint32_t b = e & 0xFFFFu, o = e >> 16;
        if (b == 1u) cr = HS_SPEC_LO;
        else if (b == s.binH) cr = HS_SPEC_HI;
        else {
          const int w = (int)(b >> 5);
          uint32_t mp = 0;
          for (int r = 0; r < (w >> 1); ++r) mp += part[r * CW + col];
          if (w & 1) mp += marked_below(w - 1, 32u);
          mp += marked_below(w, b & 31u);
          cr = mp + o;
        }
      }
      if (cvalid) crank[(tile * ntgt + j) * CW + col] = (uint16_t)cr;
    }
    if (rl == 0 && cvalid) {
      meta_n[c] = n;
      meta_m[c] = flagged ? HS_FLAGGED : m;
      meta_base[c] = mybase;
      atomicMax(&stat->maxm, m);
      atomicAdd(&stat->summ, m);
      if (flagged) flist[atomicAdd(&stat->nflag, 1u)] = (uint32_t)c;
    }
    __syncthreads();  // every reader of bm / hist / part is done
    // bitmap of the tile (pass 2 turns the keys of flagged columns into NaN keys: nothing is collected there), then clear
    for (int i = tid; i < 32 * CW; i += NT) bitmap_g[tile * (32 * CW) + i] = bm[i];
    __syncthreads();
    for (int i = tid; i < (HS_NB / 2) * CW + 32 * CW; i += NT) hist[i] = 0u;
    __syncthreads();
  }
}

// diagnostics only (XH_HIST_GEOM): the bare streaming loop with CWT columns x (512 / CWT) row lanes per workgroup — what
// does the access pattern itself sustain?  (32 columns: one 128-byte line per row and workgroup; 64: two; 128: four)
template <int CWT, int HS_U>
__global__ void __launch_bounds__(HS_NT, 4)
k_hs_stream_test(const float* __restrict__ x, int T, int64_t C, int64_t st, HsStat* __restrict__ stat) {
  constexpr int RLT = HS_NT / CWT;
  const int tid = threadIdx.x, col = tid & (CWT - 1), rl = tid / CWT;
  const int64_t ntiles = (C + CWT - 1) / CWT;
  uint32_t dummy = 0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t c = tile * CWT + col;
    const bool cvalid = c < C;
    const int64_t cc = cvalid ? c : C - 1;
    hs_stream<HS_U, RLT>(x, T, st, cc, cvalid, rl, [&](const float (&v)[HS_U], const uint32_t (&k)[HS_U]) {
#pragma unroll
      for (int u = 0; u < HS_U; ++u) dummy ^= k[u];
    });
  }
  if (dummy == 0x12345679u) atomicAdd(&stat->errors, 1u);
}

// ---- wave-wide bitonic sort of 64 * K keys held K per lane (element i = lane * K + r), ascending ---------------------
// value of lane ^ m: DPP inside the VALU for m = 1, 2, 8 (quad_perm / row_ror:8) and two masked DPP moves for m = 4
// (row_shl:4 into banks 0, 2; row_shr:4 into banks 1, 3) — verified lane by lane on gfx950 by tools/dpp_test.hip;
// ds_bpermute (the LDS crossbar) only for m = 16, 32.  With every exchange on ds_bpermute the deferred sort kernel was
// bound by the crossbar (168 of them per 512-key column), not by the VALU.
__device__ __forceinline__ uint32_t hs_lane_xor(uint32_t v, int m) {
  switch (m) {
    case 1: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);
    case 4: {
      const int t = __builtin_amdgcn_update_dpp(0, (int)v, 0x104, 0xF, 0x5, false);
      return (uint32_t)__builtin_amdgcn_update_dpp(t, (int)v, 0x114, 0xF, 0xA, false);
    }
    case 8: return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true);
    default: return (uint32_t)__shfl_xor((int)v, m);
  }
}

template <int K>
__device__ __forceinline__ void hs_wave_sort(uint32_t (&v)[K], int lane) {
  // Where the direction of a comparator depends on the lane, it is ONE compare + selects (v_cmp, the lane condition folded
  // into the compare mask by a scalar xnor, v_cndmask) instead of v_min + v_max + v_cndmask: a compare and a min / max
  // issue at the same (half) rate, so a cross-lane step costs two VALU instructions per key instead of three.
#pragma unroll
  for (int k = 2; k <= 64 * K; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= K) {  // partner element lives in lane ^ (j / K), same register
        const int mlane = j / K;
        const bool up = ((lane * K) & k) == 0;
        const bool lower = (lane & mlane) == 0;
        const bool takemin = lower == up;
#pragma unroll
        for (int r = 0; r < K; ++r) {
          const uint32_t p = hs_lane_xor(v[r], mlane);
          v[r] = ((v[r] < p) == takemin) ? v[r] : p;  // (equal keys: either copy)
        }
      } else {  // inside the lane: registers r and r ^ j
#pragma unroll
        for (int r = 0; r < K; ++r) {
          if ((r & j) == 0) {
            const uint32_t a = v[r], b = v[r | j];
            if (k < K) {  // direction known at compile time
              const bool up = (r & k) == 0;
              const uint32_t mn = a < b ? a : b, mx = a < b ? b : a;
              v[r] = up ? mn : mx;
              v[r | j] = up ? mx : mn;
            } else {
              const bool up = ((lane * K + r) & k) == 0;
              const bool keep = (a < b) == up;
              v[r] = keep ? a : b;
              v[r | j] = keep ? b : a;
            }
          }
        }
      }
    }
  }
}

template <int K>
__device__ __forceinline__ void hs_sort_column(uint32_t* __restrict__ list, uint32_t m, int lane) {
  uint32_t v[K];
#pragma unroll
  for (int r = 0; r < K; ++r) {
    const uint32_t i = (uint32_t)(lane * K + r);
    v[r] = i < m ? list[i] : HS_NANKEY;
  }
  hs_wave_sort<K>(v, lane);
#pragma unroll
  for (int r = 0; r < K; ++r)
    if ((uint32_t)(lane * K + r) < m) list[lane * K + r] = v[r];  // (the next column's list starts at list + m)
}

// One column, its candidates sorted in `list` (LDS): pick the 2 nq order statistics by position, Hyndman-Fan lerp
// (utl:464-491), store the nq quantiles.  One wave; tv = 64 words of LDS scratch.
template <int CW>
__device__ __forceinline__ void hs_pick_store(const uint32_t* list, uint32_t mm, int64_t tile, int k, int64_t ck, int lane, int ntgt,
                                              int nq, const double* __restrict__ qs, const uint32_t* __restrict__ meta_n,
                                              const uint2* __restrict__ lohi, const uint16_t* __restrict__ crank, uint32_t* tv,
                                              float* __restrict__ out, int64_t ocs, int64_t oqs) {
  const uint32_t n = meta_n[ck];
  const uint2 lhk = lohi[ck];
  if (lane < ntgt) {
    const uint32_t cr = crank[(tile * ntgt + lane) * CW + k];
    uint32_t key = HS_NANKEY;
    if (cr == HS_SPEC_LO) key = lhk.x;
    else if (cr == HS_SPEC_HI) key = lhk.y;
    else if (cr < mm) key = list[cr];
    tv[lane] = key;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < nq) {
    const float left = xh_key2f(tv[2 * lane]), right = xh_key2f(tv[2 * lane + 1]);
    double r;
    if (n == 0u) r = xh_nan64();
    else if (n < 2u) r = (double)left;
    else {
      const double nn = (double)n, qq = qs[lane];
      const double vi = nn * qq + (1.0 + qq * (1.0 - 1.0 - 1.0)) - 1.0;
      if (vi >= nn - 1.0 || vi < 0.0) r = (double)left;
      else {
        const double gamma = vi - floor(vi);
        const float diff = right - left;
        r = (double)left + (double)diff * gamma;
        if (gamma >= 0.5) r = (double)right - (double)diff * (1.0 - gamma);
      }
    }
    out[ck * ocs + (int64_t)lane * oqs] = (float)r;
  }
  __builtin_amdgcn_wave_barrier();
}

// ---- pass 2: collect the keys of the target bins, sort them per column, pick + lerp -----------------------------------
// LDS: cand [CW * 512] keys (the columns' lists back to back) | bm [32][CW] | cursor [CW] | tv [waves][64] picked keys | trash
constexpr size_t hs_lds2(int cw) { return (size_t)cw * 512 * 4 + 32 * cw * 4 + cw * 4 + (size_t)(cw * HS_RL / 64) * 64 * 4 + 16; }

template <int HS_U, int CW, bool DEFER>
__global__ void __launch_bounds__(CW * HS_RL, 4)
k_hs_collect(const float* __restrict__ x, int T, int64_t C, int64_t st, const uint2* __restrict__ lohi,
             const double* __restrict__ qs, int nq, const uint32_t* __restrict__ meta_n, const uint32_t* __restrict__ meta_m,
             const uint32_t* __restrict__ meta_base, const uint16_t* __restrict__ crank, const uint32_t* __restrict__ bitmap_g, float* __restrict__ out, int64_t ocs,
             int64_t oqs, HsStat* __restrict__ stat, int xcd_map, int abl, uint32_t* __restrict__ cand_g) {
  constexpr int NT = CW * HS_RL, POOL = CW * 512;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint32_t* cand = reinterpret_cast<uint32_t*>(smem);
  uint32_t* bm = cand + POOL;
  uint32_t* cursor = bm + 32 * CW;
  uint32_t* tvall = cursor + CW;
  uint32_t* trash = tvall + (NT / 64) * 64;
  const int tid = threadIdx.x, col = tid & (CW - 1), rl = tid / CW;
  const int lane = tid & 63, wv = tid >> 6;
  const int ntgt = 2 * nq;
  const int64_t ntiles = (C + CW - 1) / CW;
  if (tid == 0) tvall[0] = 0u;
  for (int64_t round_base = 0; round_base < ntiles; round_base += gridDim.x) {
    const int64_t tile = hs_tile_of(round_base, ntiles, xcd_map);
    if (tile < 0) break;  // (block-uniform; only in the last round)
    const int64_t c = tile * CW + col;
    const bool cvalid = c < C;
    const int64_t cc = cvalid ? c : C - 1;
    const uint2 lh = lohi[cc];
    const HsScale s = hs_scale(lh.x, lh.y);
    const uint32_t mymax = meta_m[cc];
    const bool collect = cvalid && mymax != HS_FLAGGED;
    for (int i = tid; i < 32 * CW; i += NT) bm[i] = bitmap_g[tile * (32 * CW) + i];
    if (tid < CW) cursor[tid] = 0u;
    __syncthreads();
    uint32_t* mylist = cand + (collect ? meta_base[cc] : 0u);
    const uint32_t* mybm = bm + col;
    hs_stream<HS_U, HS_RL>(x, T, st, cc, collect, rl, [&](const float (&v)[HS_U], const uint32_t (&k)[HS_U]) {
      // all bins, then all bitmap words (16 LDS reads in flight), then ONE cursor atomic per lane and batch
      uint32_t b[HS_U], w[HS_U];
#pragma unroll
      for (int u = 0; u < HS_U; ++u) b[u] = hs_bin(v[u], k[u], s);
#pragma unroll
      for (int u = 0; u < HS_U; ++u) w[u] = mybm[(b[u] >> 5) * CW];
      uint32_t hit = 0;
#pragma unroll
      for (int u = 0; u < HS_U; ++u) hit |= (((w[u] >> (b[u] & 31u)) & 1u) & (hs_valid(k[u]) ? 1u : 0u)) << u;
      if (__any(hit != 0u)) {
        // one reservation per lane and batch; the stores are unconditional (a key that is no candidate goes to a trash
        // word) — no exec-mask round trip per key.  pos < the column's count by construction (pass 1 counted the same bins).
        uint32_t pos = atomicAdd(&cursor[col], (uint32_t)__popc(hit));
#pragma unroll
        for (int u = 0; u < HS_U; ++u) {
          const uint32_t bit = (hit >> u) & 1u;
          uint32_t* dst = bit ? mylist + pos : trash;
          *dst = k[u];
          pos += bit;
        }
      }
    });
    __syncthreads();
    if (DEFER) {
      // one workgroup per CU (LDS): a sort in here would run with nothing streaming beside it.  The tile's candidate
      // lists leave for global memory as they lie in the pool (one coalesced copy of ~1 % of the tile's bytes);
      // k_hs_finish sorts them with the whole chip's VALUs.
      uint32_t end = 0;
      if (tid < CW) {
        const int64_t ck = tile * CW + tid;
        if (ck < C && meta_m[ck] != HS_FLAGGED) {
          if (cursor[tid] != meta_m[ck]) atomicAdd(&stat->errors, 1u);
          end = meta_base[ck] + meta_m[ck];
        }
        atomicMax(&tvall[0], end);
      }
      __syncthreads();
      const uint32_t tot = tvall[0];
      uint32_t* dst = cand_g + tile * (int64_t)POOL;
      for (uint32_t i = tid; i < tot; i += NT) dst[i] = cand[i];
      __syncthreads();
      if (tid == 0) tvall[0] = 0u;
      continue;
    }
    // ---- one wave per column: sort, pick, lerp (utl:464-491), store
    uint32_t* tv = tvall + wv * 64;
    for (int k = wv; k < CW; k += NT / 64) {
      const int64_t ck = tile * CW + k;
      if (ck >= C) break;  // (wave-uniform)
      const uint32_t mm = meta_m[ck];
      if (mm == HS_FLAGGED) continue;
      const uint32_t m = cursor[k];
      if (m != mm && lane == 0) atomicAdd(&stat->errors, 1u);
      uint32_t* list = cand + meta_base[ck];
      if (abl & 1) {
      } else if (m > 1024u) hs_sort_column<32>(list, m < mm ? m : mm, lane);
      else if (m > 512u) hs_sort_column<16>(list, m, lane);
      else if (m > 256u) hs_sort_column<8>(list, m, lane);
      else if (m > 128u) hs_sort_column<4>(list, m, lane);
      else if (m > 64u) hs_sort_column<2>(list, m, lane);
      else if (m > 1u) hs_sort_column<1>(list, m, lane);
      __builtin_amdgcn_wave_barrier();
      hs_pick_store<CW>(list, mm, tile, k, ck, lane, ntgt, nq, qs, meta_n, lohi, crank, tv, out, ocs, oqs);
    }
    __syncthreads();  // cand / bm / cursor are rewritten by the next tile
  }
}

// ---- pass 3 (deferred form of the epilogue above): one wave per column sorts its candidates in registers ----------------
template <int K>
__device__ __forceinline__ void hs_sort_global(const uint32_t* __restrict__ src, uint32_t* buf, uint32_t m, int lane) {
  uint32_t v[K];
#pragma unroll
  for (int r = 0; r < K; ++r) {  // element i = r * 64 + lane on the way in (coalesced); the sort does not care
    const uint32_t i = (uint32_t)(r * 64 + lane);
    v[r] = i < m ? src[i] : HS_NANKEY;
  }
  hs_wave_sort<K>(v, lane);
#pragma unroll
  for (int r = 0; r < K; ++r) buf[lane * K + r] = v[r];
}

// BIG = false: the columns with at most 512 candidates (2 KB of LDS per wave: many waves per CU hide the chain of
// dependent loads meta -> list -> positions); BIG = true: the few larger ones.
template <int CW, bool BIG>
__global__ void __launch_bounds__(256)
k_hs_finish(const uint32_t* __restrict__ cand_g, int64_t C, const uint2* __restrict__ lohi, const double* __restrict__ qs, int nq,
            const uint32_t* __restrict__ meta_n, const uint32_t* __restrict__ meta_m, const uint32_t* __restrict__ meta_base,
            const uint16_t* __restrict__ crank, float* __restrict__ out, int64_t ocs, int64_t oqs) {
  constexpr int POOL = CW * 512;
  __shared__ uint32_t sorted[4][BIG ? HS_CAPMAX : 512];
  __shared__ uint32_t tvs[4][64];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int ntgt = 2 * nq;
  for (int64_t ck = (int64_t)blockIdx.x * 4 + wv; ck < C; ck += (int64_t)gridDim.x * 4) {
    const uint32_t mm = meta_m[ck];
    if (mm == HS_FLAGGED || (mm > 512u) != BIG) continue;
    const int64_t tile = ck / CW;
    const int k = (int)(ck - tile * CW);
    const uint32_t* src = cand_g + tile * (int64_t)POOL + meta_base[ck];
    uint32_t* buf = sorted[wv];
    if (BIG) {
      if (mm > 1024u) hs_sort_global<32>(src, buf, mm, lane);
      else hs_sort_global<16>(src, buf, mm, lane);
    } else {
      if (mm > 256u) hs_sort_global<8>(src, buf, mm, lane);
      else if (mm > 128u) hs_sort_global<4>(src, buf, mm, lane);
      else if (mm > 64u) hs_sort_global<2>(src, buf, mm, lane);
      else hs_sort_global<1>(src, buf, mm, lane);
    }
    __builtin_amdgcn_wave_barrier();
    hs_pick_store<CW>(buf, mm, tile, k, ck, lane, ntgt, nq, qs, meta_n, lohi, crank, tvs[wv], out, ocs, oqs);
  }
}

// ---- flagged columns: gathered into column-contiguous scratch for the column kernels, results scattered back ---------
__global__ void __launch_bounds__(XH_BLOCK)
k_hs_gather(const float* __restrict__ x, int64_t T, int64_t st, const uint32_t* __restrict__ flist, float* __restrict__ buf,
            int64_t Tp) {
  const int64_t c = flist[blockIdx.x];
  float* dst = buf + (int64_t)blockIdx.x * Tp;
  for (int64_t t = threadIdx.x; t < Tp; t += XH_BLOCK) dst[t] = t < T ? x[t * st + c] : xh_nan32();
}

__global__ void __launch_bounds__(XH_BLOCK)
k_hs_scatter(const float* __restrict__ tmp, int64_t nf, int nq, const uint32_t* __restrict__ flist, float* __restrict__ out,
             int64_t ocs, int64_t oqs) {
  const int64_t i = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (i >= nf * nq) return;
  const int64_t q = i / nf, f = i - q * nf;
  out[(int64_t)flist[f] * ocs + q * oqs] = tmp[i];
}

}  // namespace

// Quantiles of long series straight from the time-major (T, C) view.  XH_ERR_NOTIMPL when the shape does not fit or too
// many columns would need the column kernels (the caller then takes the transposed pipeline of eqm.hip for everything).
int xh_select_hist(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const double* d_q, int nq, float* out,
                   int64_t out_cstride, int64_t out_qstride) {
  if (T <= 1024 || T > 32768 || nq < 1 || nq > HS_MAXQ || C < 1) return XH_ERR_NOTIMPL;  // (32768: the column kernels' limit)
  if ((unsigned long long)((HS_RL * 32 + HS_RL) * st + C) * 4ull >= (1ull << 32)) return XH_ERR_NOTIMPL;  // 32-bit offsets inside a batch (U <= 32)
  if (xh_diag_env("XH_SELECT_NOHIST")) return XH_ERR_NOTIMPL;  // A/B against the transposed pipeline
  // columns per workgroup: 64 (1024 threads, one workgroup per CU, 256-byte row segments: the bare streaming loop runs
  // at 6.2 TB/s against 5.3 TB/s with 32 columns / 128-byte segments, profiles/r03/select_hist_geometry.txt)
  const char* ecw = xh_diag_env("XH_HIST_CW");
  const int CWH = (ecw && atoi(ecw) == 32) ? 32 : 64;
  const int64_t ntiles = cdiv64(C, CWH);
  const int ntgt = 2 * nq;
  // fallback capacity: flagged columns are recomputed from a gathered copy; more than that -> everything the old way
  const int64_t Tp = (T + 63) & ~(int64_t)63;
  int64_t nfmax = C < 4096 ? C : 4096;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t b_lohi = al(sizeof(uint2) * (size_t)C), b_n = al(4 * (size_t)C), b_m = al(4 * (size_t)C), b_base = al(4 * (size_t)C);
  const size_t b_crank = al(2 * (size_t)ntiles * ntgt * CWH), b_bm = al(4 * (size_t)ntiles * 32 * CWH);
  const size_t b_flist = al(4 * (size_t)C), b_stat = al(sizeof(HsStat));
  const size_t b_gather = al(4 * (size_t)nfmax * (size_t)Tp), b_tmp = al(4 * (size_t)nfmax * (size_t)nq);
  // diagnostics: XH_HIST_DEFER=1 moves the per-column sort out of pass 2 into k_hs_finish (measured on config 4 with the
  // DPP sort: 9.27 + 1.29 ms against 10.71 ms inline — no gain, and it needs 2 GB of scratch: off)
  const char* edf = xh_diag_env("XH_HIST_DEFER");
  const bool defer = CWH == 64 && edf && atoi(edf) == 1;
  const size_t b_cand = defer ? al(4 * (size_t)ntiles * (size_t)CWH * 512) : 0;
  void* ws = nullptr;
  int rc = xh_big_scratch(ctx, b_lohi + b_n + b_m + b_base + b_crank + b_bm + b_flist + b_stat + b_gather + b_tmp + b_cand, &ws);
  if (rc) return rc;
  char* p = (char*)ws;
  uint2* lohi = (uint2*)p; p += b_lohi;
  uint32_t* meta_n = (uint32_t*)p; p += b_n;
  uint32_t* meta_m = (uint32_t*)p; p += b_m;
  uint32_t* meta_base = (uint32_t*)p; p += b_base;
  uint16_t* crank = (uint16_t*)p; p += b_crank;
  uint32_t* bitmap_g = (uint32_t*)p; p += b_bm;
  uint32_t* flist = (uint32_t*)p; p += b_flist;
  HsStat* stat = (HsStat*)p; p += b_stat;
  float* gbuf = (float*)p; p += b_gather;
  float* gtmp = (float*)p; p += b_tmp;
  uint32_t* cand_g = defer ? (uint32_t*)p : nullptr;
  XH_CHECK_HIP(hipMemsetAsync(stat, 0, sizeof(HsStat), ctx->stream));
  // pass 0
  int64_t S = T / 342;
  if (S < 1) S = 1;
  const int64_t ns = T / S;
  const int vec = xh_pick_vec(x, C, st);
  {
    const int64_t nthreads = cdiv64(C, vec);
    dim3 grid((unsigned)cdiv64(nthreads, XH_BLOCK));
    if (vec == 4) hipLaunchKernelGGL((k_hs_sample<4>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, S, ns, lohi);
    else hipLaunchKernelGGL((k_hs_sample<1>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, S, ns, lohi);
    XH_LAUNCH_CHECK();
  }
  int64_t nblk = ntiles;
  const int64_t maxblk = (int64_t)ctx->num_cu * (CWH == 32 ? 2 : 1);  // LDS: two 512-thread or one 1024-thread workgroup per CU
  if (nblk > maxblk) nblk = maxblk;
  if (nblk >= 8) nblk &= ~(int64_t)7;  // the XCD-aware tile map wants a multiple of 8 (the tile loop is grid-strided)
  const int xcd_map = xh_diag_env("XH_HIST_XCD") ? 1 : 0;  // measured on config 4: 42.8 ms with the map, 41.3 without -> off
  const char* eabl = xh_diag_env("XH_HIST_ABL");  // diagnostics: 1 = no candidate sort, 2 = pass 1 loads only, 4 = no pass-1 tile epilogue (wrong results)
  const int abl = eabl ? atoi(eabl) : 0;
  {
    const char* g = xh_diag_env("XH_HIST_GRID");  // diagnostics: workgroups per CU
    if (g && atoi(g) > 0 && (int64_t)ctx->num_cu * atoi(g) < ntiles) nblk = (int64_t)ctx->num_cu * atoi(g);
  }
  if (const char* eg = xh_diag_env("XH_HIST_GEOM")) {  // diagnostics: time the bare streaming loop in another geometry, then go on
    const int g = atoi(eg);
    const int64_t nt = cdiv64(C, g);
    int64_t nb = nt < maxblk ? nt : maxblk;
    if (g == 64) hipLaunchKernelGGL((k_hs_stream_test<64, 16>), dim3((unsigned)nb), dim3(HS_NT), 0, ctx->stream, x, (int)T, C, st, stat);
    else if (g == 128) hipLaunchKernelGGL((k_hs_stream_test<128, 16>), dim3((unsigned)nb), dim3(HS_NT), 0, ctx->stream, x, (int)T, C, st, stat);
    else if (g == 256) hipLaunchKernelGGL((k_hs_stream_test<256, 16>), dim3((unsigned)nb), dim3(HS_NT), 0, ctx->stream, x, (int)T, C, st, stat);
    else hipLaunchKernelGGL((k_hs_stream_test<32, 16>), dim3((unsigned)nb), dim3(HS_NT), 0, ctx->stream, x, (int)T, C, st, stat);
    XH_LAUNCH_CHECK();
  }
  const char* eu = xh_diag_env("XH_HIST_U");  // diagnostics: loads in flight per register set (8 | 16)
  const int U = eu ? atoi(eu) : HS_UDEF;
#define XH_HS_LAUNCH(UU, CC, DD)                                                                                                      \
  {                                                                                                                                  \
    XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_hs_hist<UU, CC>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hs_lds1(CC))); \
    XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_hs_collect<UU, CC, DD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hs_lds2(CC))); \
    hipLaunchKernelGGL((k_hs_hist<UU, CC>), dim3((unsigned)nblk), dim3(CC * HS_RL), hs_lds1(CC), ctx->stream, x, (int)T, C, st, lohi,  \
                       d_q, nq, meta_n, meta_m, meta_base, crank, bitmap_g, flist, stat, xcd_map, abl);                              \
    XH_LAUNCH_CHECK();                                                                                                               \
    hipLaunchKernelGGL((k_hs_collect<UU, CC, DD>), dim3((unsigned)nblk), dim3(CC * HS_RL), hs_lds2(CC), ctx->stream, x, (int)T, C, st, \
                       lohi, d_q, nq, meta_n, meta_m, meta_base, crank, bitmap_g, out, out_cstride, out_qstride, stat, xcd_map, abl,  \
                       cand_g);                                                                                                      \
    XH_LAUNCH_CHECK();                                                                                                               \
  }
  if (CWH == 32 && U == 8) XH_HS_LAUNCH(8, 32, false)
  else if (CWH == 32) XH_HS_LAUNCH(16, 32, false)
  else if (U == 8 && defer) XH_HS_LAUNCH(8, 64, true)
  else if (U == 8) XH_HS_LAUNCH(8, 64, false)
  else if (defer) XH_HS_LAUNCH(16, 64, true)
  else XH_HS_LAUNCH(16, 64, false)
  if (defer) {
    int64_t fb = cdiv64(C, 4);
    if (fb > (int64_t)ctx->num_cu * 64) fb = (int64_t)ctx->num_cu * 64;
    hipLaunchKernelGGL((k_hs_finish<64, false>), dim3((unsigned)fb), dim3(256), 0, ctx->stream, cand_g, C, lohi, d_q, nq, meta_n,
                       meta_m, meta_base, crank, out, out_cstride, out_qstride);
    XH_LAUNCH_CHECK();
    hipLaunchKernelGGL((k_hs_finish<64, true>), dim3((unsigned)fb), dim3(256), 0, ctx->stream, cand_g, C, lohi, d_q, nq, meta_n,
                       meta_m, meta_base, crank, out, out_cstride, out_qstride);
    XH_LAUNCH_CHECK();
  }
#undef XH_HS_LAUNCH
  HsStat h;
  XH_CHECK_HIP(hipMemcpyAsync(&h, stat, sizeof(HsStat), hipMemcpyDeviceToHost, ctx->stream));
  XH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  XH_REQUIRE(h.errors == 0, XH_ERR_HIP, "xh_select_hist: %u columns met another candidate count in pass 2 than in pass 1",
             h.errors);
  if (xh_diag_env("XH_HIST_STATS")) fprintf(stderr, "[xh_select_hist] T=%lld C=%lld flagged=%u candidates: max %u mean %.1f\n", (long long)T, (long long)C, h.nflag, h.maxm, (double)h.summ / (double)C);
  if (h.nflag == 0) return XH_OK;
  if ((int64_t)h.nflag > nfmax) return XH_ERR_NOTIMPL;  // the caller recomputes everything with the transposed pipeline
  const int64_t nf = h.nflag;
  hipLaunchKernelGGL(k_hs_gather, dim3((unsigned)nf), dim3(XH_BLOCK), 0, ctx->stream, x, T, st, flist, gbuf, Tp);
  XH_LAUNCH_CHECK();
  rc = xh_select_columns(ctx, gbuf, T, nf, Tp, d_q, nq, gtmp, 1, nf);
  if (rc) return rc;
  hipLaunchKernelGGL(k_hs_scatter, dim3((unsigned)cdiv64(nf * nq, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, gtmp, nf, nq, flist,
                     out, out_cstride, out_qstride);
  XH_LAUNCH_CHECK();
  return XH_OK;
}
