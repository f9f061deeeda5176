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
//                          value -> a FINITE value window [lo, hi] that holds ~97.5 % of the column
//   pass 1  k_hs_hist     all rows: 1024-bin histogram per column over that window (u16 counters in LDS, [bin][column],
//                          two adjacent columns per 32-bit word, LDS atomics).  Bins (hs_bin): 0 x < lo | 1 x == lo |
//                          2 + f regular, f = floor((x - lo) * scale) clamped to [0, 1019] | 3 + f(hi) x == hi |
//                          4 + f x > hi.  The bins of lo and of hi hold ONE value each ("pure": the dry days of a
//                          precipitation series, a saturated maximum) and never need a second look.  At the end of a
//                          tile: prefix sums, the bin and the rank inside its bin of each of the 2 nq order statistics
//                          (Hyndman-Fan, utl:395, 417-461), a bitmap of those target bins, and the position every
//                          target will have among the column's sorted CANDIDATES (= the keys of the target bins).
//   pass 2  k_hs_collect  all rows again: values whose bin is a target bin (~3 % of them) are appended to the column's
//                          list in LDS; at the end of a tile one wave per column sorts its candidates (<= 2048) in registers
//                          (bitonic, DPP / ds_bpermute exchanges), picks the 2 nq order statistics by position, lerps
//                          (utl:464-491) and stores the nq quantiles.
//
// Round 4: the per-sample instruction diet.  Both passes were VALU-issue bound next to a 6 TB/s stream (profiles/r03/
// valu_busy.txt: 22 + 1 and ~30 instructions per sample).  Now
//   * the bin comes from the VALUE alone: f from sub / mul / med3 / cvt, the five classes from the SIGNS of x - lo and
//     x - hi (v_med3_i32 of the float's bits against -1 / +1; a float difference is zero iff the operands are equal) —
//     no order-preserving integer key, no saturating key arithmetic, no validity compare: 8 instructions;
//   * NaN samples (and the rows past the end of the series, which the loader turns into NaN) are found per 16-row batch
//     by ONE float sum per sample (NaN propagates); only a batch with a NaN in the wave takes the path that tests
//     every sample;
//   * the histogram word of a sample is [bin][column / 2]: its address is one v_lshl_add of the bin, the increment
//     (1 or 1 << 16) is a per-lane constant — the bin-pair layout of round 3 needed five instructions for both;
//   * pass 2 looks the sample's REGULAR index f up in a table of bit pairs (candidate | needs the exact classes) built
//     by pass 1: 4 instructions for f, 5 for the lookup; only batches that touch a flagged f (the window's ends when
//     they are targets) or hold a NaN recompute the exact bin.  Candidates are appended as raw floats under the
//     execution mask and become keys when they are sorted.
//
// Geometry: a 1024-thread workgroup owns 64 adjacent columns (256-byte row segments) and 16 "row lanes" per column;
// thread (column c, row lane r) loads rows r, r + 16, r + 32 ... through a ring of 5 register sets of 8 loads (32 to 40
// in flight per lane).  One workgroup per CU (LDS).  Algorithmic bytes per pass: 4E; nothing else crosses HBM except ~600 bytes of per-column
// tables.
//
// Columns whose target bins hold more than 8192 keys (heavily tied or clustered values away from the window's ends) are
// flagged in pass 1 and recomputed by the column kernels of select.hip / select2.hip / select5.hip from a gathered copy;
// lists of 2049 .. 8192 keys are sorted in place in LDS (hs_sort_lds, round 5), tiles whose lists do not fit the 32768-key
// pool together are collected in up to 8 rounds of pass 2.
//
// Round 5: (1) QDM mode (template flag of the tile functions; xh_qdm_hist): QuantileDeltaMapping "nearest" needs no rank per
// sample, only the <= nq + 1 class boundaries as order statistics — the targets become the boundary ranks of qdmrank.h, pass
// 1 also tracks the column's extremes, pass 2 counts their copies and its epilogue follows ties through the runs of equal
// candidates; a streaming classification (k_cut_classify, qdm2.hip) finishes.  (2) k_hs_fused: both passes of a tile in one
// kernel, the second in reverse row order, to catch the tile's tail in the Infinity Cache — built, bit-identical, slower
// than the two kernels (profiles/r05/select4_fused_ab.txt): diagnostic only.
#include <stdlib.h>

#include "common.h"
#include "qdmrank.h"

namespace {

constexpr int HS_RL = 16;              // row lanes per column
constexpr int HS_CW = 64;              // columns per workgroup
constexpr int HS_NT = HS_CW * HS_RL;   // threads per workgroup

constexpr int HS_NB = 1024;            // bins per column
constexpr int HS_NREG = 1020;          // regular indices f = 0 .. 1019
constexpr int HS_POOL = HS_CW * 512;   // candidate keys of one tile in LDS
constexpr int HS_CAPMAX = 2048;        // ... at most this many for one column in the register sort
constexpr int HS_CAPBIG = 8192;        // ... and this many at all (2049 .. 8192: a bitonic sort in LDS, in place in the column's list)
constexpr int HS_MAXQ = 32;            // quantiles per call on this path
constexpr int HS_QREC_MAX = 2 * (HS_MAXQ + 1);  // QDM: records (collected bins) per column, two per class boundary
constexpr int HS_QWS = 2 * HS_QREC_MAX + 64 + HS_MAXQ + 2 * HS_MAXQ + 4;  // ... and the per-wave scratch of its epilogue: records | valid nodes | factors | quantiles | n, lo, hi (words)
static_assert((HS_NT / 64) * HS_QWS <= 64 * HS_CW + 32 * HS_CW && HS_QWS % 2 == 0, "the epilogue's scratch lives in the dead tables of the streaming loop");
constexpr uint32_t HS_NANKEY = 0xFFFFFFFFu;
constexpr uint32_t HS_KEY_MINF = 0x00800000u, HS_KEY_MAXF = 0xFF7FFFFFu;  // keys of -FLT_MAX / +FLT_MAX
constexpr uint32_t HS_SPEC_LO = 0xFFFEu, HS_SPEC_HI = 0xFFFDu, HS_SPEC_NONE = 0xFFFFu;
constexpr uint32_t HS_FLAGGED = 0xFFFFFFFFu;

struct HsStat {
  uint32_t nflag;   // columns handed to the column kernels
  uint32_t maxm;    // largest candidate count of any column
  uint32_t errors;  // pass 2 met a different candidate count than pass 1 announced (must stay 0)
  uint32_t summ;    // candidates of all columns (diagnostics: the mean per column)
  uint32_t rounds;  // highest collect round any column was put in (pass 2 runs rounds + 1 times)
};

// per-column window -> bin arithmetic, identical in pass 1 and pass 2.  Bins are equally wide in VALUE, not in key
// space: keys are logarithmic in |x| (a series that straddles zero — degrees Celsius, anomalies — would put all of its
// samples into a handful of key-space bins around the huge key range of the tiny values).  fp32 subtract, multiply and
// floor are monotone, and monotone + identical in both passes + pure bins for lo and hi is all the selection needs:
//   bin(x) = f(x) + 2 + sgn(x - lo) + sgn(x - hi),   f(x) = floor(med3((x - lo) * scale, 0, 1019))
//   x < lo -> 0 | x == lo -> 1 | lo < x < hi -> 2 + f | x == hi -> 3 + f(hi) | x > hi -> 4 + f        (lo < hi)
// x - lo is +0 iff x == lo (IEEE: a difference of finite floats never rounds to zero; denormals are kept, the kernel
// descriptor says float_denorm_mode_32 = 3), so the classes are exact whatever the rounding of f.  lo and hi are finite
// (k_hs_sample clamps them to +-FLT_MAX: x = -inf then is "below lo", inf - inf never happens for a non-NaN sample).
// Zeros of both signs are ONE value here (hs_scale turns a zero window end into -0.0: x - (-0.0) = +0 for either zero); the
// integer keys of round 3 ordered -0 below +0.
// NaN samples do not pass through hs_bin in pass 1 (see the NaN sum) and land on f = 0 in pass 2's table look-up.
struct HsScale {
  float lof, hif, scale;
};

__device__ __forceinline__ HsScale hs_scale(uint32_t lo, uint32_t hi) {
  HsScale s;
  s.lof = xh_key2f(lo);
  s.hif = xh_key2f(hi);
  // a window end that is a zero becomes NEGATIVE zero: x - (-0.0) = x + (+0.0) is +0 for both zeros, whereas -0.0 - (+0.0)
  // is -0.0, whose bits read as "negative" in hs_sgn — the two zeros of a column would land in different bins (harmless for
  // a quantile, wrong for the runs of equal values the QDM boundaries follow; found by tools/fuzz_r05.py, round 5)
  if (s.lof == 0.0f) s.lof = -0.0f;
  if (s.hif == 0.0f) s.hif = -0.0f;
  const float d = s.hif - s.lof;
  float sc = 0.0f;
  if (d > 0.0f) sc = (float)((double)HS_NREG * 1.000001 / (double)d);  // (hi - lo) * scale >= 1020: hi lands on f = 1019
  if (!(sc <= 3.0e38f)) sc = 0.0f;                                       // d = inf, or a window of denormal width
  s.scale = sc;
  return s;
}

__device__ __forceinline__ int hs_sgn(float z) {  // -1 | 0 | +1 from the bits of a float difference (one v_med3_i32)
  int r;  // (the compiler turns min / max of the float's bits into two v_cmp + two v_cndmask)
  asm("v_med3_i32 %0, %1, -1, 1" : "=v"(r) : "v"(z));
  return r;
}
__device__ __forceinline__ uint32_t hs_findex(float z, float scale) {
  // median of (t, 0, 1019): NaN (inf * 0, NaN samples) and negative t give 0, large t the top index
  return (uint32_t)__builtin_amdgcn_fmed3f(z * scale, 0.0f, (float)(HS_NREG - 1));
}
// the bin MINUS 2 (a signed number >= -2): the callers fold the 2 into their table base
__device__ __forceinline__ int hs_bin_m2(float x, const HsScale& s) {
  const float z = x - s.lof;
  return (int)hs_findex(z, s.scale) + hs_sgn(z) + hs_sgn(x - s.hif);
}
__device__ __forceinline__ uint32_t hs_bin(float x, const HsScale& s) { return (uint32_t)(hs_bin_m2(x, s) + 2); }

// order-preserving key of a non-NaN float (candidates are never NaN)
__device__ __forceinline__ uint32_t hs_key(float f) {
  const uint32_t u = __float_as_uint(f);
  return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
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
    // a finite window: hs_bin takes float differences against lo and hi
    lo = lo < HS_KEY_MINF ? HS_KEY_MINF : (lo > HS_KEY_MAXF ? HS_KEY_MAXF : lo);
    hi = hi < HS_KEY_MINF ? HS_KEY_MINF : (hi > HS_KEY_MAXF ? HS_KEY_MAXF : hi);
    lohi[c0 + v] = make_uint2(lo, hi);
  }
}

// ---- the streaming loop shared by pass 1 and pass 2 ------------------------------------------------------------------
// Thread (col, rl) of a tile visits rows rl, rl + 16, ... in batches of HS_U rows; f(values) is called once per batch.
// Rows past the end of the series arrive as NaN (the callers skip NaN samples anyway).
//
// A RING of NSET register sets (no copies: a copy of one set into another right after element u is consumed would make
// element u wait for the load that was issued a moment ago): while one set is consumed, the loads of NSET - 1 batches
// fly.  Round 4 measurements on config 4 (profiles/r04/select4_anatomy.txt): the bare loop streams at 6.0 TB/s whatever
// the occupancy, but every microsecond a wave spends away from it (LDS latency chains, the tile epilogues) comes on top —
// one workgroup per CU, nothing else to run.  5 sets of 8 (32 to 40 loads per lane in flight, re-requested every 8
// samples) against round 3's 2 sets of 16: config-4 train 40.5 -> 37 ms.  prime() requests the first NSET - 1 batches
// of a tile; calling it for the NEXT tile before a tile's epilogue (round 4: template flag EARLY, removed in round 5), so
// that the epilogue runs under flying loads — measured: no gain, off.
//
// Round 5: `rev` walks the full batches from the LAST one down to the first.  The fused kernel reads a tile twice in a row
// (histogram pass, then collect pass); with 256 tiles of 2.8 MB in flight the 256 MiB Infinity Cache still holds roughly
// the last third of what the first pass read, and only a second pass that STARTS there finds it (tools/mall_ubench.hip,
// profiles/r05/mall_ubench.txt: second pass 6.46 ms forward, 5.52 ms in reverse, 7.5 ms cold, per 45.4 GB array).
template <int HS_U, int NSET, int RLT = HS_RL>
struct HsRing {
  static constexpr int ROWS = RLT * HS_U;  // rows a workgroup covers per batch
  float S[NSET][HS_U];
  int kb_last = -1;  // >= 0: batch k of the ring is batch kb_last - k of the series (reverse order)

  // Buffer loads: a descriptor re-based per batch (scalar), the row offset of load u as the scalar offset, ONE 32-bit
  // per-lane byte offset — no vector address arithmetic and no 64-bit address registers per load.
  __device__ __forceinline__ void load(float (&dst)[HS_U], const float* __restrict__ x, int64_t st, uint32_t voff, int kb) {
    if (kb_last >= 0) kb = kb_last - kb;
    const float* base = x + (int64_t)kb * ROWS * st;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, (int)0xFFFFFFFFu, 0x00020000);
    const uint32_t rowstep = (uint32_t)(st * 4 * RLT);  // bytes between the rows of loads u and u + 1 (host: 256 rows < 4 GiB)
    uint32_t soff = 0u;
#pragma unroll
    for (int u = 0; u < HS_U; ++u) {
      dst[u] = __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0));
      soff += rowstep;
    }
  }

  __device__ __forceinline__ void prime(const float* __restrict__ x, int T, int64_t st, int64_t cc, int rl, bool rev = false) {
    const uint32_t voff = (uint32_t)(((int64_t)rl * st + cc) * 4);
    const int nfull = T / ROWS;
    kb_last = rev ? nfull - 1 : -1;
#pragma unroll
    for (int i = 0; i < NSET - 1; ++i)
      if (i < nfull) load(S[i], x, st, voff, i);
  }

  // (prime() with the same arguments came first)
  template <typename F>
  __device__ __forceinline__ void run(const float* __restrict__ x, int T, int64_t st, int64_t cc, int rl, F&& f) {
    const uint32_t voff = (uint32_t)(((int64_t)rl * st + cc) * 4);
    const int nfull = T / ROWS;
    int done = 0;
    // steady state: no conditional loads (the compiler counts the outstanding loads exactly)
    while (done + 2 * NSET - 1 <= nfull) {
#pragma unroll
      for (int i = 0; i < NSET; ++i) {
        load(S[(i + NSET - 1) % NSET], x, st, voff, done + i + NSET - 1);
        f(S[i]);
      }
      done += NSET;
    }
    // drain: at most 2 NSET - 2 batches are left, NSET - 1 of them already requested (done % NSET == 0: static sets)
#pragma unroll
    for (int i = 0; i < 2 * NSET - 2; ++i) {
      if (done + i < nfull) {
        if (done + i + NSET - 1 < nfull) load(S[(i + NSET - 1) % NSET], x, st, voff, done + i + NSET - 1);
        f(S[i % NSET]);
      }
    }
    // tail rows (fewer than a batch): same buffer loads from CLAMPED per-lane rows, validity applied afterwards (a
    // conditional load costs a full s_waitcnt vmcnt(0)).  rl passes through an opaque copy: hoisted out of the tile
    // loop, the clamped offsets would stay live (and spill) across the whole kernel.
    const int t0 = nfull * ROWS;
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
#pragma unroll
      for (int u = 0; u < HS_U; ++u) buf[u] = u * RLT + rlo < rem ? buf[u] : xh_nan32();
      f(buf);
    }
  }
};

// NaN anywhere in the batch of this WAVE?  One float add per sample (NaN propagates; +inf and -inf in one lane's batch
// give a false alarm, which only costs the exact path).
template <int HS_U>
__device__ __forceinline__ bool hs_wave_has_nan(const float (&v)[HS_U]) {
  // two sequential chains: a tree makes the compiler pair the operands for v_pk_add_f32 (16 v_mov to line them up)
  float a = v[0], b = v[1];
#pragma unroll
  for (int u = 2; u < HS_U; u += 2) {
    a += v[u];
    b += v[u + 1];
  }
  a += b;
  return __any(a != a) != 0;
}

// Workgroup -> tile map: grid-strided, the identity inside a round (an XCD-aware map was measured in round 3: no gain).
__device__ __forceinline__ int64_t hs_tile_of(int64_t round_base, int64_t ntiles) {
  const int64_t tile = round_base + blockIdx.x;
  return tile < ntiles ? tile : -1;
}

// diagnostics only (XH_HIST_GEOM): the bare streaming loop with 64 columns x RLT row lanes per workgroup and `lds` bytes of
// dynamic LDS (= how many workgroups a CU holds): what does the access pattern itself sustain at which occupancy?
template <int RLT, int NSET, int HS_U = 16>
__global__ void __launch_bounds__(HS_CW * RLT)
k_hs_stream_test(const float* __restrict__ x, int T, int64_t C, int64_t st, HsStat* __restrict__ stat) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, col = tid & (HS_CW - 1), rl = tid / HS_CW;
  const int64_t ntiles = (C + HS_CW - 1) / HS_CW;
  float dummy = 0.0f;
  HsRing<HS_U, NSET, RLT> ring;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t c = tile * HS_CW + col;
    const int64_t cc = c < C ? c : C - 1;
    ring.prime(x, T, st, cc, rl);
    ring.run(x, T, st, cc, rl, [&](const float (&v)[HS_U]) {
#pragma unroll
      for (int u = 0; u < HS_U; ++u) dummy += v[u];
    });
  }
  if (dummy == 0.12345f) {
    atomicAdd(&stat->errors, 1u);
    smem[tid] = 1;
  }
}

// ---- pass 1: histogram + target bins -----------------------------------------------------------------------------------
// LDS: h32 [1024 bins][32 column pairs] u32 (two u16 counters per word: columns 2p, 2p + 1 of bin b at [b][p]) | bm [32][64]
// target-bin bitmap | part [32][32] packed partial sums (later [16][64] candidate counts) | tgt [2 * MAXQ][64] (bin | rank
// inside the bin << 16) | mcol [64] | cbase [64] | ntot [32] packed column totals
constexpr size_t hs_lds1() {
  return (size_t)HS_NB * 32 * 4 + 32 * HS_CW * 4 + 1024 * 4 + 2 * HS_MAXQ * HS_CW * 4 + 2 * HS_CW * 4 + 32 * 4;
}

// 16 bits -> the even bit positions of a word
__device__ __forceinline__ uint32_t hs_spread16(uint32_t x) {
  x = (x | (x << 8)) & 0x00FF00FFu;
  x = (x | (x << 4)) & 0x0F0F0F0Fu;
  x = (x | (x << 2)) & 0x33333333u;
  x = (x | (x << 1)) & 0x55555555u;
  return x;
}

// everything the two passes share (kernel arguments of the round-3/4 kernels, now one struct: the fused kernel runs both)
struct HsArgs {
  const float* __restrict__ x;
  int T;
  int64_t C, st;
  const uint2* __restrict__ lohi;
  const double* __restrict__ qs;
  int nq;
  uint32_t* __restrict__ meta_n;
  uint32_t* __restrict__ meta_m;
  uint32_t* __restrict__ meta_base;
  uint16_t* __restrict__ crank;
  uint32_t* __restrict__ bitmap_g;
  uint32_t* __restrict__ tab_g;
  uint32_t* __restrict__ flist;
  float* __restrict__ out;
  int64_t ocs, oqs;
  HsStat* __restrict__ stat;
  int abl;
  // QDM mode (xh_qdm_hist): the targets are the class boundaries of QuantileDeltaMapping "nearest" (qdmrank.h); pass 1 also
  // tracks the column's minimum and maximum (two VALU per sample), pass 2 counts their copies (four) — the ranks of the
  // boundaries depend on them, and collecting the open-ended tail bins instead (1.2 % of the samples each) doubled the
  // candidates and cost a second collect round; pass 1 leaves two RECORDS per target (its bin and the next non-empty bin: global rank of the
  // bin's first sample | samples in the bin, offset of the bin's keys in the candidate list | kind), pass 2's epilogue turns
  // them into cut values (gcut) and class factors (gfac)
  int qdm;
  const float* __restrict__ af;   // (nq, C) factors, row stride af_qs; a NaN factor drops its node
  int64_t af_qs;
  int extrap;
  uint32_t* __restrict__ qrec;    // [tile][2 * (nq + 1)][2][64]
  float* __restrict__ colmin;     // (C) smallest / largest valid sample of the column (pass 1); pass 2 counts their copies
  float* __restrict__ colmax;
  float* __restrict__ gcut;       // (nq + 1, C)
  float* __restrict__ gfac;       // (nq + 2, C)
};

constexpr uint32_t HS_REC_REG = 0u, HS_REC_LO = 1u, HS_REC_HI = 2u, HS_REC_NONE = 0xFFFFu;

// 128-bit clears of n words (n a multiple of 4 * NT is not required)
__device__ __forceinline__ void hs_clear(uint32_t* p, int n, int tid) {
  uint4* p4 = reinterpret_cast<uint4*>(p);
  for (int i = tid; i < n / 4; i += HS_NT) p4[i] = make_uint4(0u, 0u, 0u, 0u);
}

// One tile of pass 1.  The caller primed the ring for this tile (rows forward); the histogram and the bitmap are zero on
// entry (cleared at the end of the previous tile, or by the caller) and zero again on return when `clear_after`.
template <int HS_U, int NSET, bool QDM>
__device__ __forceinline__ void hs_hist_tile(const HsArgs& A, HsRing<HS_U, NSET>& ring, unsigned char* smem, int64_t tile, bool clear_after) {
  constexpr int CW = HS_CW, NT = HS_NT;
  const float* __restrict__ x = A.x;
  const int T = A.T;
  const int64_t C = A.C, st = A.st;
  const uint2* __restrict__ lohi = A.lohi;
  const double* __restrict__ qs = A.qs;
  const int nq = A.nq, abl = A.abl;
  uint32_t* __restrict__ meta_n = A.meta_n;
  uint32_t* __restrict__ meta_m = A.meta_m;
  uint32_t* __restrict__ meta_base = A.meta_base;
  uint16_t* __restrict__ crank = A.crank;
  uint32_t* __restrict__ bitmap_g = A.bitmap_g;
  uint32_t* __restrict__ tab_g = A.tab_g;
  uint32_t* __restrict__ flist = A.flist;
  HsStat* __restrict__ stat = A.stat;
  uint32_t* h32 = reinterpret_cast<uint32_t*>(smem);
  uint32_t* bm = h32 + HS_NB * 32;
  uint32_t* part = bm + 32 * CW;
  uint32_t* tgt = part + 1024;
  uint32_t* mcol = tgt + 2 * HS_MAXQ * CW;
  uint32_t* cbase = mcol + CW;
  uint32_t* ntot = cbase + CW;
  const int tid = threadIdx.x, col = tid & (CW - 1), rl = tid / CW;
  const int pw = tid & 31, prt = tid >> 5;  // epilogue: column-pair word and 32-bin part of the prefix sums
  const uint32_t sh16 = (uint32_t)(col & 1) << 4;
  const int ntgt = 2 * nq;
  {
    const int64_t c = tile * CW + col;
    const bool cvalid = c < C;
    const int64_t cc = cvalid ? c : C - 1;
    const uint2 lh = lohi[cc];
    const HsScale s = hs_scale(lh.x, lh.y);
    // this lane's counter of bin b is the half `col & 1` of word [b][col >> 1]; a column past C adds zeros.  Row 2 of
    // the table is bin 0 of hs_bin_m2.
    uint32_t* mycol = h32 + 2 * 32 + (col >> 1);
    const uint32_t one = cvalid ? (1u << sh16) : 0u;
    float dummy = 0.0f;
    float vmn = __uint_as_float(0x7F800000u), vmx = __uint_as_float(0xFF800000u);  // QDM: the column's extremes (NaN ignored)
    constexpr bool qdm = QDM;  // (a template parameter: the quantile kernels do not carry the extra code in their loops)
    ring.run(x, T, st, cc, rl, [&](const float (&v)[HS_U]) {
      if (abl & 2) {  // diagnostics: loads only
#pragma unroll
        for (int u = 0; u < HS_U; ++u) dummy += v[u];
        return;
      }
      if (qdm) {  // (block-uniform) v_min / v_max (and their three-operand forms: two samples per instruction at the same issue
                  // cost) return a non-NaN operand when there is one; inline asm: fminf canonicalises both inputs first
#pragma unroll
        for (int u = 0; u + 1 < HS_U; u += 2) {
          asm("v_min3_f32 %0, %0, %1, %2" : "+v"(vmn) : "v"(v[u]), "v"(v[u + 1]));
          asm("v_max3_f32 %0, %0, %1, %2" : "+v"(vmx) : "v"(v[u]), "v"(v[u + 1]));
        }
        if (HS_U % 2 == 1) {
          asm("v_min_f32 %0, %0, %1" : "+v"(vmn) : "v"(v[HS_U - 1]));
          asm("v_max_f32 %0, %0, %1" : "+v"(vmx) : "v"(v[HS_U - 1]));
        }
      }
      // (groups of four samples between scheduling barriers: with all 16 in flight at once the temporaries of the bin
      // arithmetic push the register sets of the streaming ring out into scratch)
      if (!hs_wave_has_nan<HS_U>(v)) {
#pragma unroll
        for (int u = 0; u < HS_U; ++u) {
          atomicAdd(mycol + hs_bin_m2(v[u], s) * 32, one);
          if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int u = 0; u < HS_U; ++u) {
          atomicAdd(mycol + hs_bin_m2(v[u], s) * 32, v[u] == v[u] ? one : 0u);
          if ((u & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      }
    });
    if ((abl & 2) && dummy == 0.12345f) atomicAdd(&stat->errors, 1u);
    __syncthreads();
    if (abl & 4) return;  // diagnostics: no tile epilogue (wrong results, the histogram is not even cleared)
    if (qdm) {  // the extremes of the 16 row lanes of a column -> colmin / colmax (through `part`, which the prefix sums use next)
      float* pf = reinterpret_cast<float*>(part);
#pragma unroll 1
      for (int h = 0; h < 2; ++h) {
        pf[rl * CW + col] = h ? vmx : vmn;
        __syncthreads();
        if (rl == 0 && cvalid) {
          float m = pf[col];
          for (int r = 1; r < HS_RL; ++r) {
            const float o = pf[r * CW + col];
            m = h ? (o > m ? o : m) : (o < m ? o : m);
          }
          (h ? A.colmax : A.colmin)[c] = m;
        }
        __syncthreads();
      }
    }
    // ---- exclusive prefix sums over the bins, in place and PACKED (two columns per word; every half stays <= T <= 65535):
    // thread (pw, prt) owns the words of bins [prt * 32, prt * 32 + 32) of column pair pw
    uint32_t ssum = 0;
    if (!(abl & 1024)) {
#pragma unroll 8
    for (int i = 0; i < 32; ++i) ssum += h32[(prt * 32 + i) * 32 + pw];
    }
    part[prt * 32 + pw] = ssum;
    __syncthreads();
    uint32_t run = 0;
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const uint32_t p = part[r * 32 + pw];
      run += r < prt ? p : 0u;
    }
    if (!(abl & 1024)) {
#pragma unroll 8
    for (int i = 0; i < 32; ++i) {
      const int idx = (prt * 32 + i) * 32 + pw;
      const uint32_t w = h32[idx];
      h32[idx] = run;  // samples in the bins below this one
      run += w;
    }
    }
    if (prt == 31) ntot[pw] = run;
    __syncthreads();
    const uint32_t n = (ntot[col >> 1] >> sh16) & 0xFFFFu;
    const uint32_t* myh = h32 + (col >> 1);
    auto below = [&](uint32_t b) -> uint32_t {  // samples in bins < b, b in [0, 1024]
      if (b >= (uint32_t)HS_NB) return n;
      return (myh[b * 32] >> sh16) & 0xFFFFu;
    };
    const uint32_t binL = hs_bin(s.lof, s), binH = hs_bin(s.hif, s);  // the pure bins
    auto bin_of_rank = [&](uint32_t r) -> uint32_t {  // largest b with below(b) <= r: that bin holds rank r
      uint32_t lo_b = 0u, hi_b = HS_NB - 1;
#pragma unroll 1
      for (int it = 0; it < 10; ++it) {
        const uint32_t mid = (lo_b + hi_b + 1u) >> 1;
        const bool le = below(mid) <= r;
        lo_b = le ? mid : lo_b;
        hi_b = le ? hi_b : mid - 1u;
      }
      return lo_b;
    };
    auto mark = [&](uint32_t b) {
      if (b != binL && b != binH) atomicOr(&bm[(b >> 5) * CW + col], 1u << (b & 31u));
    };
    // ---- the bin and the rank inside it of every target; mark the bins that need a second look
    if (!QDM) {
      for (int j = rl; j < ntgt; j += HS_RL) {
        uint32_t e = HS_SPEC_NONE;
        if (n > 0u && !(abl & 128)) {
          const uint32_t r = hs_rank(n, qs[j >> 1], j & 1);
          const uint32_t lo_b = bin_of_rank(r);
          e = lo_b | ((r - below(lo_b)) << 16);
          mark(lo_b);
        }
        tgt[j * CW + col] = e;
      }
    } else {
      // QDM: targets 0 .. nq = the class boundaries (rank from qdm_min_r2 with the copies of the minimum / maximum ESTIMATED
      // from the histogram: exact when the extreme value has a pure bin to itself — the dry days —, 1 otherwise; pass 2
      // counts them, recomputes, and flags the column if a rank then leaves the collected bins).
      // tgt = bin | successor bin << 16: the next non-empty bin is collected too when the
      // target's bin is pure or the target is its last sample (the cut may be the next distinct value, qdmrank.h).
      uint8_t* idx = reinterpret_cast<uint8_t*>(tgt + 40 * CW);  // [nq][CW] node index of the j-th valid node
      uint32_t* nvc = tgt + 50 * CW;                             // [CW] valid nodes of the column
      if (rl == 0) {
        uint32_t cnt = 0;
#pragma unroll 1
        for (int j0 = 0; j0 < nq; j0 += 8) {
          float a[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] = A.af[(int64_t)(j0 + u < nq ? j0 + u : nq - 1) * A.af_qs + cc];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (j0 + u < nq && a[u] == a[u]) { idx[cnt * CW + col] = (uint8_t)(j0 + u); ++cnt; }
        }
        nvc[col] = cnt;
      }
      __syncthreads();
      const uint32_t nvn = nvc[col];
      uint32_t c0e = 1u, cme = 1u;
      if (n > 0u) {
        const uint32_t fb = bin_of_rank(0u), lb = bin_of_rank(n - 1u);
        if (fb == binL || fb == binH) c0e = below(fb + 1u) - below(fb);
        if (lb == binL || lb == binH) cme = below(lb + 1u) - below(lb);
      }
      const bool stok = n > 0u && nvn >= 2u && c0e < n;
      const int ntest = nq + 1;
      for (int j = rl; j < ntest; j += HS_RL) {
        uint32_t r = 0xFFFFFFFFu;
        if (n > 0u) {
          if (stok && (uint32_t)j <= nvn) {
            double thr;
            if (j == 0) thr = qs[idx[col]];
            else if ((uint32_t)j == nvn) thr = qs[idx[(nvn - 1u) * CW + col]];
            else thr = qs[idx[(j - 1) * CW + col]] / 2.0 + qs[idx[j * CW + col]] / 2.0;
            const uint32_t R = qdm_min_r2(j == 0, thr, n, c0e, cme);
            if (R <= 2u * n) r = qdm_pos_of_r2(R);
            if (r >= n) r = 0xFFFFFFFFu;
          }
        }
        uint32_t e = HS_SPEC_NONE | (HS_SPEC_NONE << 16);
        if (r != 0xFFFFFFFFu) {
          const uint32_t b = bin_of_rank(r);
          mark(b);
          uint32_t sb = HS_SPEC_NONE;
          const uint32_t endb = below(b + 1u);
          if ((b == binL || b == binH || r + 1u == endb) && endb < n) {
            sb = bin_of_rank(endb);
            mark(sb);
          }
          e = b | (sb << 16);
        }
        tgt[j * CW + col] = e;
      }
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
    // the tile's candidate lists share one LDS pool in pass 2: column offsets in column order (one wave, shuffle scan).  A
    // column with more candidates than the largest register sort is flagged for the column kernels.  A tile whose lists do
    // not fit the pool together (long series: the candidates grow with T, ~1400 per column at 55 152 steps) is collected in
    // several ROUNDS of pass 2, each round the columns of one pool-full: round = (inclusive prefix - 1) / (POOL - CAPBIG)
    // — the round's first list may start below the boundary by less than CAPBIG, so a round never overflows the pool.
    if (rl == 0) {  // wave 0: lane = column
      const uint32_t mk = cvalid ? m : 0u;
      const bool big = mk > (uint32_t)HS_CAPBIG;
      const uint32_t val = big ? 0u : mk;
      uint32_t incl = val;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        incl += col >= d ? o : 0u;
      }
      const uint32_t excl = incl - val;
      const uint32_t round = val > 0u ? (incl - 1u) / (uint32_t)(HS_POOL - HS_CAPBIG) : 0u;
      unsigned long long mine = 0ull;  // the lanes (columns) of my round
#pragma unroll 1
      for (uint32_t r = 0; r <= 7u; ++r) {  // 8 rounds handled here (55 152 steps need 4), columns beyond are flagged
        const unsigned long long mr = __ballot(round == r && val > 0u);
        if (round == r) mine = mr;
      }
      const int first = mine ? __ffsll((long long)mine) - 1 : col;
      const uint32_t start = (uint32_t)__shfl((int)excl, first);
      const bool fl = big || round > 7u;
      cbase[col] = fl ? HS_FLAGGED : ((excl - start) | (round << 16));
      if (!fl && val > 0u) atomicMax(&stat->rounds, round);
    }
    __syncthreads();
    const uint32_t mybase = cbase[col];
    const bool flagged = mybase == HS_FLAGGED;
    auto list_offset_of = [&](uint32_t b) -> uint32_t {  // keys of the marked bins below bin b
      const int w = (int)(b >> 5);
      uint32_t mp = 0;
      for (int r = 0; r < (w >> 1); ++r) mp += part[r * CW + col];
      if (w & 1) mp += marked_below(w - 1, 32u);
      mp += marked_below(w, b & 31u);
      return mp;
    };
    if (!QDM) {
      for (int j = rl; j < ntgt; j += HS_RL) {
        const uint32_t e = tgt[j * CW + col];
        uint32_t cr = HS_SPEC_NONE;
        if (e != HS_SPEC_NONE && !(abl & 256)) {
          const uint32_t b = e & 0xFFFFu, o = e >> 16;
          if (b == binL) cr = HS_SPEC_LO;
          else if (b == binH) cr = HS_SPEC_HI;
          else cr = list_offset_of(b) + o;
        }
        if (cvalid) crank[(tile * ntgt + j) * CW + col] = (uint16_t)cr;
      }
    } else {
      const int nrec = 2 * (nq + 1);
      for (int j = rl; j < nq + 1; j += HS_RL) {
        const uint32_t e = tgt[j * CW + col];
#pragma unroll 1
        for (int h = 0; h < 2; ++h) {
          const uint32_t b = h ? e >> 16 : e & 0xFFFFu;
          uint32_t w0 = 0u, w1 = HS_REC_NONE << 16;
          if (b != HS_SPEC_NONE) {
            w0 = below(b) | ((below(b + 1u) - below(b)) << 16);
            if (b == binL) w1 = HS_REC_LO << 16;
            else if (b == binH) w1 = HS_REC_HI << 16;
            else w1 = list_offset_of(b) | (HS_REC_REG << 16);
          }
          if (cvalid) {
            A.qrec[((tile * nrec + 2 * j + h) * 2 + 0) * CW + col] = w0;
            A.qrec[((tile * nrec + 2 * j + h) * 2 + 1) * CW + col] = w1;
          }
        }
      }
    }
    if (rl == 0 && cvalid) {
      meta_n[c] = n;
      meta_m[c] = flagged ? HS_FLAGGED : m;
      meta_base[c] = mybase;
      atomicMax(&stat->maxm, m);
      atomicAdd(&stat->summ, m);
      if (flagged) flist[atomicAdd(&stat->nflag, 1u)] = (uint32_t)c;
    }
    // ---- pass 2's look-up table: one bit pair per regular index f (word f >> 4, bits 2 (f & 15)): bit 0 = "a sample
    // with this f is a candidate" (interior f: 0 < f < f(hi), where the bin is f + 2 whatever the sample), bit 1 =
    // "decide with the exact bin" (f = 0 and f >= f(hi), where one f covers several bins and one of them is marked)
    if (!(abl & 512)) {
      const uint32_t fH = hs_findex(s.hif - s.lof, s.scale);
      auto bmbit = [&](uint32_t b) -> uint32_t { return (bm[(b >> 5) * CW + col] >> (b & 31u)) & 1u; };
#pragma unroll 1
      for (int i = 0; i < 4; ++i) {
        const int wq = rl * 4 + i;
        const uint32_t bin0 = (uint32_t)wq * 16u + 2u;
        const int wi = (int)(bin0 >> 5);
        const uint32_t lo32 = bm[wi * CW + col];
        const uint32_t hi32 = wi + 1 < 32 ? bm[(wi + 1) * CW + col] : 0u;
        const uint64_t both = ((uint64_t)hi32 << 32) | lo32;
        uint32_t word = hs_spread16((uint32_t)(both >> (bin0 & 31u)) & 0xFFFFu);
        if (wq == 0 || (uint32_t)wq * 16u + 15u >= fH) {
#pragma unroll 1
          for (uint32_t e = 0; e < 16u; ++e) {
            const uint32_t f = (uint32_t)wq * 16u + e;
            const bool first = f == 0u, top = f >= fH && f < (uint32_t)HS_NREG;
            if (!first && !top) continue;
            uint32_t fl = 0u;
            if (first) fl |= bmbit(0u) | bmbit(2u);
            if (top) fl |= bmbit(f + 4u) | (f == fH ? bmbit(f + 2u) : 0u);
            word = (word & ~(3u << (2u * e))) | (fl << (2u * e + 1u));
          }
        }
        if (cvalid) tab_g[(tile * 64 + wq) * CW + col] = word;
      }
    }
    __syncthreads();  // every reader of bm / h32 / part is done
    // bitmap of the tile, then clear
    for (int i = tid; i < 32 * CW; i += NT) bitmap_g[tile * (32 * CW) + i] = bm[i];
    __syncthreads();
    if (clear_after) {
      hs_clear(h32, HS_NB * 32 + 32 * CW, tid);
      __syncthreads();
    }
  }
}

template <int HS_U, int NSET, bool QDM>
__global__ void __launch_bounds__(HS_NT, 4)
k_hs_hist(HsArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, col = tid & (HS_CW - 1), rl = tid / HS_CW;
  const int64_t ntiles = (A.C + HS_CW - 1) / HS_CW;
  hs_clear(reinterpret_cast<uint32_t*>(smem), HS_NB * 32 + 32 * HS_CW, tid);
  __syncthreads();
  HsRing<HS_U, NSET> ring;
  for (int64_t round_base = 0; round_base < ntiles; round_base += gridDim.x) {
    const int64_t tile = hs_tile_of(round_base, ntiles);
    if (tile < 0) break;  // (block-uniform; only in the last round)
    const int64_t c = tile * HS_CW + col;
    ring.prime(A.x, A.T, A.st, c < A.C ? c : A.C - 1, rl);
    hs_hist_tile<HS_U, NSET, QDM>(A, ring, smem, tile, true);
  }
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
__device__ __forceinline__ void hs_sort_column(uint32_t* __restrict__ list, uint32_t m, int lane, float zero) {
  uint32_t v[K];
#pragma unroll
  for (int r = 0; r < K; ++r) {
    const uint32_t i = (uint32_t)(lane * K + r);
    // the list holds the raw floats of pass 2.  `zero` = +0.0 (QDM: -0.0 + 0.0 = +0.0, the two zeros are ONE value when
    // ranks are counted, scipy.stats.rankdata) or -0.0 (quantiles: x + -0.0 = x bit for bit)
    v[r] = i < m ? hs_key(__uint_as_float(list[i]) + zero) : HS_NANKEY;
  }
  hs_wave_sort<K>(v, lane);
#pragma unroll
  for (int r = 0; r < K; ++r)
    if ((uint32_t)(lane * K + r) < m) list[lane * K + r] = v[r];  // (the next column's list starts at list + m)
}

// Lists beyond the register sort (2049 .. HS_CAPBIG keys: clustered or tied values, series of 55 152 steps): one wave sorts
// the list IN PLACE in LDS.  Bitonic network with every comparator pointing up (the first step of a merge mirrors, i ^ (k - 1),
// the others compare i with i + j): indices past the list's end stand for +inf, a comparator that touches one is a no-op,
// so no padding is stored — the next column's list starts right behind this one.  ~90 stages of m / 128 compare-exchanges per
// lane; rare, and it replaces the detour through the radix select / the global sort of whole columns (eqm_55k: 62 of 85 ms).
__device__ __forceinline__ void hs_sort_lds(uint32_t* __restrict__ list, uint32_t m, int lane, float zero) {
  for (uint32_t i = (uint32_t)lane; i < m; i += 64u) list[i] = hs_key(__uint_as_float(list[i]) + zero);
  __builtin_amdgcn_wave_barrier();
  uint32_t P = 1u;
  while (P < m) P <<= 1;
#pragma unroll 1
  for (uint32_t k = 2u; k <= P; k <<= 1) {
    const uint32_t hk = k >> 1;
#pragma unroll 1
    for (uint32_t idx = (uint32_t)lane; idx < (P >> 1); idx += 64u) {  // mirror step: i in the lower half of its block
      const uint32_t i = (idx / hk) * k + (idx % hk), p = i ^ (k - 1u);
      if (p < m) {
        const uint32_t a = list[i], b = list[p];
        if (a > b) { list[i] = b; list[p] = a; }
      }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll 1
    for (uint32_t j = hk >> 1; j >= 1u; j >>= 1) {
#pragma unroll 1
      for (uint32_t idx = (uint32_t)lane; idx < (P >> 1); idx += 64u) {
        const uint32_t i = (idx / j) * (2u * j) + (idx % j), p = i + j;
        if (p < m) {
          const uint32_t a = list[i], b = list[p];
          if (a > b) { list[i] = b; list[p] = a; }
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
}

// element at (0-based) position p of the merge of two ascending runs A[0, nA) and B[0, nB) (keys; p < nA + nB): the merge-path
// split — the smallest i with B[p - i - 1] <= A[i] (or i at an end) takes i elements of A and p - i of B in front of it
__device__ __forceinline__ uint32_t hs_merged_at(const uint32_t* A, uint32_t nA, const uint32_t* B, uint32_t nB, uint32_t p) {
  uint32_t lo = p > nB ? p - nB : 0u, hi = p < nA ? p : nA;
  while (lo < hi) {
    const uint32_t i = (lo + hi) >> 1, j = p - i;  // (i < hi <= min(p, nA): j >= 1, A[i] exists; j <= nB by lo's start)
    if (B[j - 1u] <= A[i]) hi = i; else lo = i + 1u;
  }
  const uint32_t j = p - lo;
  const uint32_t a = lo < nA ? A[lo] : HS_NANKEY, b = j < nB ? B[j] : HS_NANKEY;
  return a < b ? a : b;
}

// One column, its candidates sorted in `list` (LDS): pick the 2 nq order statistics by position, Hyndman-Fan lerp
// (utl:464-491), store the nq quantiles.  One wave; tv = 64 words of LDS scratch.
// nA = 0: `list` is sorted as a whole; otherwise two sorted runs, [0, nA) and [nA, mm)
template <int CW>
__device__ __forceinline__ void hs_pick_store(const uint32_t* list, uint32_t mm, uint32_t nA, int k, int64_t ck, int lane, int ntgt,
                                              int nq, double qq, uint32_t n, uint2 lhk, const uint16_t* crank_s, uint32_t* tv,
                                              float* __restrict__ out, int64_t ocs, int64_t oqs) {
  if (lane < ntgt) {
    const uint32_t cr = crank_s[lane * CW + k];  // (the tile's [ntgt][CW] table, in LDS)
    uint32_t key = HS_NANKEY;
    if (cr == HS_SPEC_LO) key = lhk.x;
    else if (cr == HS_SPEC_HI) key = lhk.y;
    else if (cr < mm) key = nA ? hs_merged_at(list, nA, list + nA, mm - nA, cr) : list[cr];
    tv[lane] = key;
  }
  __builtin_amdgcn_wave_barrier();
  if (lane < nq) {
    const float left = xh_key2f(tv[2 * lane]), right = xh_key2f(tv[2 * lane + 1]);
    double r;
    if (n == 0u) r = xh_nan64();
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
    out[ck * ocs + (int64_t)lane * oqs] = (float)r;
  }
  __builtin_amdgcn_wave_barrier();
}

// QDM epilogue of one column (one wave; its candidates sorted in `list`, keys; c0 / cmax = the copies of the column's minimum
// / maximum counted by pass 2): per class boundary the rank that decides it (qdmrank.h), its run of equal values among the candidates (or a pure bin),
// the cut value -> gcut; the class factors -> gfac.  A rank outside the collected bins, or a run longer than the scan cap,
// puts the column on the list for the exact-rank kernels (returns true).  ws: HS_QWS words of LDS scratch.
// What hs_qdm_pick reads from global memory for one column goes to the wave's LDS scratch BEFORE the column's candidates are
// sorted — as plain load -> LDS-store pairs: kept in registers across the sort instead (a first version of this round), the
// nine values per lane were spilled to scratch around it (128-VGPR cap of a 1024-thread workgroup): 5 GB of scratch traffic
// per call in the PMC passes and nothing gained.
template <int CW>
__device__ __forceinline__ void hs_qdm_stage(const HsArgs& A, int64_t tile, int k, int64_t ck, int lane, uint32_t* ws, uint32_t n,
                                             uint32_t lo, uint32_t hi) {
  const int nq = A.nq, nrec = 2 * (nq + 1);
  const uint32_t* __restrict__ rec = A.qrec + (tile * nrec) * 2 * CW + k;
  uint2* recs = reinterpret_cast<uint2*>(ws);
  uint32_t* tv = ws + 2 * HS_QREC_MAX;
  float* afs = reinterpret_cast<float*>(tv + 64);
  double* qss = reinterpret_cast<double*>(afs + HS_MAXQ);
  uint32_t* hdr = reinterpret_cast<uint32_t*>(qss + HS_MAXQ);  // n | lo | hi
  for (int i = lane; i < nrec; i += 64) recs[i] = make_uint2(rec[(i * 2 + 0) * CW], rec[(i * 2 + 1) * CW]);
  if (lane < nq) {
    afs[lane] = A.af[(int64_t)lane * A.af_qs + ck];
    qss[lane] = A.qs[lane];
  }
  if (lane == 0) {
    hdr[0] = n;
    hdr[1] = lo;
    hdr[2] = hi;
  }
}

template <int CW>
__device__ __forceinline__ bool hs_qdm_pick(const HsArgs& A, const uint32_t* list, int64_t ck, int lane, uint32_t* ws,
                                            uint32_t c0, uint32_t cmax) {
  const int nq = A.nq, ntest = nq + 1, nrec = 2 * (nq + 1);
  // The column's records sit in LDS (`ws`: per-wave scratch inside the tables of the streaming loop, which are dead in the
  // epilogue; staged by hs_qdm_stage) and every lane — one class boundary each — scans them with broadcast reads and selects,
  // no branch, no scalar code.  Round 5 kept them in two registers per lane and read them back with v_readlane in a loop that
  // `continue`d per record: 42 iterations of readlane -> scalar compare -> divergent branch per search, two searches per
  // boundary — 7.0 of the 43.8 ms of QuantileDeltaMapping.adjust at 30 years x 1440 x 720 (tools/experiments/r06/qdm_c4_abl.py:
  // XH_HIST_ABL=128 switches this function off), a chain of scalar-to-vector hazards that 16 waves of a CU ran in lock-step.
  uint2* recs = reinterpret_cast<uint2*>(ws);
  uint32_t* tv = ws + 2 * HS_QREC_MAX;
  float* afs = reinterpret_cast<float*>(tv + 64);       // [nq] the column's factors (the class factors below read them by node)
  double* qss = reinterpret_cast<double*>(afs + HS_MAXQ);  // [nq] the quantiles
  const uint32_t* hdr = reinterpret_cast<const uint32_t*>(qss + HS_MAXQ);
  const uint32_t n = hdr[0];
  const uint2 lh = make_uint2(hdr[1], hdr[2]);
  constexpr uint32_t CAP = 256u;
  // the column's valid nodes (NaN factors dropped), compacted through the wave: tv[pos] = node index
  const float myaf = lane < nq ? afs[lane] : xh_nan32();
  const unsigned long long vmask = __ballot(myaf == myaf);
  const uint32_t nvn = (uint32_t)__popcll(vmask);
  if (myaf == myaf) tv[__popcll(vmask & ((1ull << lane) - 1ull))] = (uint32_t)lane;
  __builtin_amdgcn_wave_barrier();
  // a record that holds global rank r: value + run [a, b) of equal values around it.  found: 1 ok | 0 not collected / cap
  auto run_at = [&](uint32_t r, uint32_t& key, uint32_t& a, uint32_t& b) -> bool {
    uint32_t w0 = 0u, w1 = HS_REC_NONE << 16;   // the FIRST record whose rank range holds r (the ranges are disjoint anyway)
    for (int i = 0; i < nrec; ++i) {
      const uint2 w = recs[i];
      const bool hit = (w1 >> 16) == HS_REC_NONE && (w.y >> 16) != HS_REC_NONE && r - (w.x & 0xFFFFu) < (w.x >> 16) && r >= (w.x & 0xFFFFu);
      w0 = hit ? w.x : w0;
      w1 = hit ? w.y : w1;
    }
    const uint32_t kind = w1 >> 16, bel = w0 & 0xFFFFu, cnt = w0 >> 16;
    if (kind == HS_REC_NONE) return false;
    if (kind != HS_REC_REG) {  // a pure bin: one value (as the canonical key the sorted candidates carry: -0.0 -> +0.0)
      key = hs_key(xh_key2f(kind == HS_REC_LO ? lh.x : lh.y) + 0.0f);
      a = bel;
      b = bel + cnt;
      return true;
    }
    const uint32_t cs = w1 & 0xFFFFu, pos = cs + (r - bel);
    key = list[pos];
    uint32_t l = pos, h = pos + 1u, steps = 0u;
    while (l > cs && list[l - 1u] == key && steps < CAP) { --l; ++steps; }
    while (h < cs + cnt && list[h] == key && steps < CAP) { ++h; ++steps; }
    a = bel + (l - cs);
    b = bel + (h - cs);
    return steps < CAP;
  };
  bool flag = false;
  const bool ok = n > 0u && nvn >= 2u && c0 < n;
  if (lane < ntest) {
    uint32_t cut = HS_NANKEY;
    if (ok && (uint32_t)lane <= nvn && !flag) {
      double thr;
      if (lane == 0) thr = qss[tv[0]];
      else if ((uint32_t)lane == nvn) thr = qss[tv[nvn - 1u]];
      else thr = qss[tv[lane - 1]] / 2.0 + qss[tv[lane]] / 2.0;
      const uint32_t R = qdm_min_r2(lane == 0, thr, n, c0, cmax);
      if (R <= 2u * n - cmax + 1u) {  // (the largest doubled rank a sample has: beyond it no sample passes)
        const uint32_t p = qdm_pos_of_r2(R);
        uint32_t key, a, b;
        if (p < n && run_at(p, key, a, b)) {
          if (a + b + 1u >= R) cut = key;
          else if (b < n) {
            uint32_t k2, a2, b2;
            if (run_at(b, k2, a2, b2)) cut = k2; else flag = true;
          }
        } else if (p < n) flag = true;
      }
    }
    A.gcut[(int64_t)lane * A.C + ck] = xh_key2f(cut);
  }
  if (lane <= ntest) {  // class factors: [0] below the first node, [k] node k - 1, [nvn + 1] above the last node
    uint32_t node = lane < 1 ? 0u : (uint32_t)lane - 1u;
    node = node < nvn ? node : (nvn > 0u ? nvn - 1u : 0u);
    const uint32_t jn = nvn > 0u ? tv[node] : 0u;
    const float a = afs[jn];
    const bool inner = lane >= 1 && (uint32_t)lane <= nvn;
    A.gfac[(int64_t)lane * A.C + ck] = (ok && (uint32_t)lane <= nvn + 1u && (inner || A.extrap == 0)) ? a : xh_nan32();
  }
  __builtin_amdgcn_wave_barrier();
  return __any(flag ? 1 : 0) != 0;
}

// ---- pass 2: collect the samples of the target bins, sort them per column, pick + lerp ---------------------------------
// LDS: cand [64 * 512] candidates (the columns' lists back to back) | tab [64][64] bit pairs per regular index | bm [32][64]
// bin bitmap (exact path) | cursor [64] | colok [64] | lbase [64] list offsets | cntmn / cntmx / valmn / valmx [64] (QDM) | tv [waves][64] picked keys
// | cmeta [5][64] the columns' candidate count, list base | round, valid samples, window ends (round 6: the epilogue read them
// from global memory column by column — five dependent L2 round trips per column in front of and behind its sort)
constexpr size_t hs_lds2() {
  return (size_t)HS_POOL * 4 + 64 * HS_CW * 4 + 32 * HS_CW * 4 + 7 * HS_CW * 4 + (size_t)(HS_NT / 64) * 64 * 4 + 5 * HS_CW * 4;
}
static_assert(hs_lds2() <= 160 * 1024, "pass 2: LDS");

// One tile of pass 2 (collect round `round`).  `rev`: the full batches are streamed from the last one down (fused kernel).
// The ring is primed here.  Returns without streaming when no column of the tile belongs to this round.
template <int HS_U, int NSET, bool QDM>
__device__ __forceinline__ void hs_collect_tile(const HsArgs& A, HsRing<HS_U, NSET>& ring, unsigned char* smem, int64_t tile, int round, bool rev) {
  constexpr int CW = HS_CW, NT = HS_NT;
  const float* __restrict__ x = A.x;
  const int T = A.T;
  const int64_t C = A.C, st = A.st;
  const uint2* __restrict__ lohi = A.lohi;
  const double* __restrict__ qs = A.qs;
  const int nq = A.nq, abl = A.abl;
  const uint32_t* __restrict__ meta_n = A.meta_n;
  const uint32_t* __restrict__ meta_m = A.meta_m;
  const uint32_t* __restrict__ meta_base = A.meta_base;
  const uint16_t* __restrict__ crank = A.crank;
  const uint32_t* __restrict__ bitmap_g = A.bitmap_g;
  const uint32_t* __restrict__ tab_g = A.tab_g;
  float* __restrict__ out = A.out;
  const int64_t ocs = A.ocs, oqs = A.oqs;
  HsStat* __restrict__ stat = A.stat;
  uint32_t* cand = reinterpret_cast<uint32_t*>(smem);
  uint32_t* tab = cand + HS_POOL;
  uint32_t* bm = tab + 64 * CW;
  uint32_t* cursor = bm + 32 * CW;
  uint32_t* colok = cursor + CW;
  uint32_t* lbase = colok + CW;
  uint32_t* cntmn = lbase + CW;   // QDM: copies of the column's minimum / maximum ...
  uint32_t* cntmx = cntmn + CW;
  float* valmn = reinterpret_cast<float*>(cntmx + CW);  // ... and the two values (read per batch: they must not live in registers)
  float* valmx = valmn + CW;
  uint32_t* tvall = reinterpret_cast<uint32_t*>(valmx + CW);
  uint32_t* cmeta = tvall + (NT / 64) * 64;  // [0] meta_m | [1] meta_base | [2] meta_n | [3] lo | [4] hi, each [CW]
  const int tid = threadIdx.x, col = tid & (CW - 1), rl = tid / CW;
  const int lane = tid & 63, wv = tid >> 6;
  const int ntgt = 2 * nq;
  {
    const int64_t c = tile * CW + col;
    const bool cvalid = c < C;
    const int64_t cc = cvalid ? c : C - 1;
    const uint2 lh = lohi[cc];
    const HsScale s = hs_scale(lh.x, lh.y);
    const uint32_t mbase = meta_base[cc];
    const bool collect = cvalid && meta_m[cc] != HS_FLAGGED && (mbase >> 16) == (uint32_t)round;
    if (tid < CW) {
      cursor[tid] = 0u;
      colok[tid] = collect ? 0xFFFFFFFFu : 0u;  // (tid < CW: col == tid)
      lbase[tid] = collect ? (mbase & 0xFFFFu) : 0u;
      cntmn[tid] = 0u;
      cntmx[tid] = 0u;
      cmeta[0 * CW + tid] = cvalid ? meta_m[cc] : HS_FLAGGED;
      cmeta[1 * CW + tid] = mbase;
      cmeta[2 * CW + tid] = meta_n[cc];
      cmeta[3 * CW + tid] = lh.x;
      cmeta[4 * CW + tid] = lh.y;
      if (QDM) {
        valmn[tid] = A.colmin[cc];
        valmx[tid] = A.colmax[cc];
      }
    }
    if (!__syncthreads_or(collect ? 1 : 0)) return;  // no column of this tile belongs to this round (block-uniform)
    ring.prime(x, T, st, cc, rl, rev);
    // the tile's tables; the columns nothing is collected for (flagged, other rounds, past C) read all-zero tables
    for (int i = tid; i < 64 * CW; i += NT) tab[i] = tab_g[tile * (64 * CW) + i] & colok[i & (CW - 1)];
    for (int i = tid; i < 32 * CW; i += NT) bm[i] = bitmap_g[tile * (32 * CW) + i] & colok[i & (CW - 1)];
    __syncthreads();
    const uint32_t* mytab = tab + col;
    const uint32_t* mybm = bm + col;
    uint32_t dummy = 0u;
    // QDM: the copies of the column's minimum / maximum (pass 1 found the values).  Nothing of this may live in registers
    // across the streaming loop: with two values and two counters per lane (at the 128-VGPR cap) the allocator copied the ring
    // of register sets at the loop's back edge behind an s_waitcnt vmcnt(0) — every iteration drained the loads in flight
    // (tools/isa_loops.py).  So: the values are read from LDS per batch, the batch's own minimum / maximum (v_min3 / v_max3)
    // decides whether any lane of the wave holds a copy at all, and only then the copies are counted into LDS.
    constexpr bool qdm = QDM;
    ring.run(x, T, st, cc, rl, [&](const float (&v)[HS_U]) {
      if (qdm && !(abl & 256)) {  // (diagnostics bit 256: no counting of the extremes' copies)
        const float vmn = valmn[col], vmx = valmx[col];
        float m = v[0], M = v[0];
#pragma unroll
        for (int u = 1; u + 1 < HS_U; u += 2) {
          asm("v_min3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(v[u]), "v"(v[u + 1]));
          asm("v_max3_f32 %0, %0, %1, %2" : "+v"(M) : "v"(v[u]), "v"(v[u + 1]));
        }
        if (HS_U % 2 == 0) {
          asm("v_min_f32 %0, %0, %1" : "+v"(m) : "v"(v[HS_U - 1]));
          asm("v_max_f32 %0, %0, %1" : "+v"(M) : "v"(v[HS_U - 1]));
        }
        if (__any((m <= vmn || M >= vmx) ? 1 : 0)) {
          uint32_t cn = 0u, cx = 0u;
#pragma unroll
          for (int u = 0; u < HS_U; ++u) {
            cn += v[u] == vmn ? 1u : 0u;
            cx += v[u] == vmx ? 1u : 0u;
          }
          if (cn) atomicAdd(&cntmn[col], cn);
          if (cx) atomicAdd(&cntmx[col], cx);
        }
      }
      // the regular index of every sample and its bit pair (8 LDS reads in flight); bit 1 anywhere in the wave: this
      // batch is decided by the exact bins (NaN samples: f = 0, whose candidate bit is never set).  The hits of the batch
      // are ONE register (bit u = sample u): sixteen 0 / 1 registers next to the ring's register sets spill, and a spill
      // reload inside this loop is an s_waitcnt vmcnt(0) — it waits for the prefetched batch as well (round 4: +1.7 ms).
      uint32_t hit = 0u, flags = 0u;
      if (abl & 64) {  // diagnostics: loads only
#pragma unroll
        for (int u = 0; u < HS_U; ++u) dummy += __float_as_uint(v[u]);
        return;
      }
#pragma unroll
      for (int u = 0; u < HS_U; ++u) {
        const uint32_t f = hs_findex(v[u] - s.lof, s.scale);
        const uint32_t pair = mytab[(f >> 4) * CW] >> ((f << 1) & 31u);
        flags |= pair;
        hit |= (pair & 1u) << u;
        if ((u & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
      if (__any((flags & 2u) != 0u)) {
        hit = 0u;
#pragma unroll
        for (int u = 0; u < HS_U; ++u) {
          const uint32_t b = hs_bin(v[u], s);
          const uint32_t w = (mybm[(b >> 5) * CW] >> (b & 31u)) & 1u;
          hit |= (v[u] == v[u] ? w : 0u) << u;
        }
      }
      if (abl & 32) {  // diagnostics: no appends
        dummy += hit;
        return;
      }
      if (__any(hit != 0u)) {
        // one reservation per lane and batch (pos < the column's count by construction: pass 1 counted the same bins), then
        // sixteen ds_write under the execution mask (v_cmpx, no branch).  The cursor's and the list's LDS addresses are
        // rebuilt from the column here (opaque copy): kept live across the loop they are the values the allocator spills.
        int colo = col;
        asm volatile("" : "+v"(colo));
        typedef __attribute__((address_space(3))) uint32_t lds_u32;
        const uint32_t pos = atomicAdd(&cursor[colo], (uint32_t)__popc(hit));
        uint32_t addr = (uint32_t)(uintptr_t)(lds_u32*)(cand + lbase[colo]) + pos * 4u;
#pragma unroll
        for (int u = 0; u < HS_U; ++u) {
          uint64_t sv;
          uint32_t bit = hit & (1u << u);
          asm volatile(
              "s_mov_b64 %[sv], exec\n\t"
              "v_cmpx_ne_u32_e32 0, %[bit]\n\t"
              "ds_write_b32 %[addr], %[val]\n\t"
              "v_add_u32_e32 %[addr], 4, %[addr]\n\t"
              "s_mov_b64 exec, %[sv]"
              : [addr] "+v"(addr), [sv] "=&s"(sv)
              : [bit] "v"(bit), [val] "v"(v[u])
              : "vcc", "memory");
        }
      }
    });
    if ((abl & 96) && dummy == 0x12345679u) atomicAdd(&stat->errors, 1u);
    __syncthreads();
    if (abl & 96) return;
    // ---- one wave per column: sort, pick, lerp (utl:464-491), store.  Nothing in the column loop reads global memory in the
    // quantile mode: the columns' metadata sit in LDS since the tile's start, the positions of the targets among the sorted
    // candidates (crank, [ntgt][CW] u16 per tile) are copied into the dead tables of the streaming loop by the whole workgroup,
    // the quantiles wait in a register pair.
    uint32_t* tv = tvall + wv * 64;
    const uint16_t* crank_s = reinterpret_cast<const uint16_t*>(tab);
    double qv = 0.0;
    if constexpr (!QDM) {
      const uint32_t* __restrict__ src = reinterpret_cast<const uint32_t*>(crank + tile * ntgt * CW);
      for (int i = tid; i < ntgt * CW / 2; i += NT) tab[i] = src[i];
      qv = qs[lane < nq ? lane : 0];
      __syncthreads();
    }
    for (int k = wv; k < CW; k += NT / 64) {
      const int64_t ck = tile * CW + k;
      if (ck >= C) break;  // (wave-uniform)
      const uint32_t mm = cmeta[0 * CW + k], mb = cmeta[1 * CW + k];
      if (mm == HS_FLAGGED || (mb >> 16) != (uint32_t)round) continue;
      const uint32_t m = cursor[k];
      if (m != mm && lane == 0) atomicAdd(&stat->errors, 1u);
      if constexpr (QDM) hs_qdm_stage<CW>(A, tile, k, ck, lane, tab + wv * HS_QWS, cmeta[2 * CW + k], cmeta[3 * CW + k], cmeta[4 * CW + k]);
      uint32_t* list = cand + (mb & 0xFFFFu);
      const uint32_t ms = m < mm ? m : mm;
      const float zero = QDM ? 0.0f : -0.0f;
      // Quantile mode, 257 .. 512 candidates (the typical column of a 30-year series holds ~275): TWO sorted runs — the first
      // 256 keys through the 256-slot network (36 stages on 4 registers), the rest through the 64 / 128 / 256-slot one — and
      // the <= 2 nq targets picked from their merge by a merge-path search (<= 9 steps of two LDS reads per target, the targets
      // in parallel) instead of one 512-slot network (45 stages on 8 registers) for ~275 keys.
      uint32_t runA = 0u;
      if (abl & 1) {
      } else if (!QDM && ms > 256u && ms <= 512u && m == mm && !(abl & 2048)) {
        runA = 256u;
        hs_sort_column<4>(list, 256u, lane, zero);
        const uint32_t rest = ms - 256u;
        if (rest > 128u) hs_sort_column<4>(list + 256, rest, lane, zero);
        else if (rest > 64u) hs_sort_column<2>(list + 256, rest, lane, zero);
        else hs_sort_column<1>(list + 256, rest, lane, zero);
      } else if (!QDM && ms > 1024u && ms <= 2048u && m == mm && !(abl & 2048)) {
        // the same for 1025 .. 2048 candidates (series of 55 152 steps: ~1230 per column): the 2048-slot network on 32 registers
        // per lane was measured at 3.7 x its instruction count (anatomy of eqm_55k: 8.5 of 67 ms) — two runs on 16 registers
        runA = 1024u;
        hs_sort_column<16>(list, 1024u, lane, zero);
        const uint32_t rest = ms - 1024u;
        if (rest > 512u) hs_sort_column<16>(list + 1024, rest, lane, zero);
        else if (rest > 256u) hs_sort_column<8>(list + 1024, rest, lane, zero);
        else if (rest > 128u) hs_sort_column<4>(list + 1024, rest, lane, zero);
        else if (rest > 64u) hs_sort_column<2>(list + 1024, rest, lane, zero);
        else hs_sort_column<1>(list + 1024, rest, lane, zero);
      } else if (ms > (uint32_t)HS_CAPMAX) { if (!(abl & 8192)) hs_sort_lds(list, ms, lane, zero); }
      else if (ms > 1024u) { if (!(abl & 16384)) hs_sort_column<32>(list, ms, lane, zero); }
      else if (ms > 512u) hs_sort_column<16>(list, ms, lane, zero);
      else if (ms > 256u) hs_sort_column<8>(list, ms, lane, zero);
      else if (ms > 128u) hs_sort_column<4>(list, ms, lane, zero);
      else if (ms > 64u) hs_sort_column<2>(list, ms, lane, zero);
      else if (ms > 0u) hs_sort_column<1>(list, ms, lane, zero);  // (one candidate: only turned into its key)
      __builtin_amdgcn_wave_barrier();
      if (!QDM) hs_pick_store<CW>(list, mm, runA, k, ck, lane, ntgt, nq, qv, cmeta[2 * CW + k], make_uint2(cmeta[3 * CW + k], cmeta[4 * CW + k]), crank_s, tv, out, ocs, oqs);
      else if (abl & 128) {  // diagnostics: no QDM epilogue (results wrong)
      } else if (hs_qdm_pick<CW>(A, list, ck, lane, tab + wv * HS_QWS, cntmn[k], cntmx[k]) && lane == 0)
        A.flist[atomicAdd(&stat->nflag, 1u)] = (uint32_t)ck;  // (behind pass 1's own entries: the host reads the count afterwards)
    }
    __syncthreads();  // cand / tab / bm / cursor are rewritten by the next tile
  }
}

template <int HS_U, int NSET, bool QDM>
__global__ void __launch_bounds__(HS_NT, 4)
k_hs_collect(HsArgs A, int round) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int64_t ntiles = (A.C + HS_CW - 1) / HS_CW;
  HsRing<HS_U, NSET> ring;
  for (int64_t round_base = 0; round_base < ntiles; round_base += gridDim.x) {
    const int64_t tile = hs_tile_of(round_base, ntiles);
    if (tile < 0) break;  // (block-uniform; only in the last round)
    hs_collect_tile<HS_U, NSET, QDM>(A, ring, smem, tile, round, false);
  }
}

// Round 5 EXPERIMENT (diagnostics, XH_HIST_FUSED=1 | 2; bit-identical, tools/fuzz_r05.py): both passes of a tile back to back
// in ONE kernel, the second one in reverse row order (see HsRing), so that what pass 1 read last is still in the Infinity
// Cache when pass 2 starts there.  The micro-benchmark of the access pattern alone promised 7.5 -> 5.5 ms for the second pass
// (profiles/r05/mall_ubench.txt); the kernel LOSES 1.4-1.7 ms per training (39.57 two kernels, 41.25 fused, 40.97 fused +
// reverse, alternating in one process): with every workgroup in its own phase the workgroups of neighbouring tiles no
// longer walk the same rows at the same time (a row of 256 adjacent tiles is one contiguous 64 KB stretch of DRAM), each
// tile epilogue ages the tile's lines in the cache by what the other 255 workgroups stream meanwhile, and the re-read
// gains 0.3 ms.  The tables pass 1 writes for pass 2 (tab_g, bitmap_g, crank, meta_*) go through global memory as before —
// the same workgroup reads them back (__syncthreads orders a workgroup's own global writes and reads); the two LDS
// layouts alias.  Collect rounds > 0 (series beyond 32768 steps) still run k_hs_collect.
template <int HS_U, int NSET>
__global__ void __launch_bounds__(HS_NT, 4)
k_hs_fused(HsArgs A, int rev) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, col = tid & (HS_CW - 1), rl = tid / HS_CW;
  const int64_t ntiles = (A.C + HS_CW - 1) / HS_CW;
  HsRing<HS_U, NSET> ring;
  for (int64_t round_base = 0; round_base < ntiles; round_base += gridDim.x) {
    const int64_t tile = hs_tile_of(round_base, ntiles);
    if (tile < 0) break;  // (block-uniform; only in the last round)
    const int64_t c = tile * HS_CW + col;
    ring.prime(A.x, A.T, A.st, c < A.C ? c : A.C - 1, rl);
    hs_clear(reinterpret_cast<uint32_t*>(smem), HS_NB * 32 + 32 * HS_CW, tid);  // (under the first loads)
    __syncthreads();
    hs_hist_tile<HS_U, NSET, false>(A, ring, smem, tile, false);
    __threadfence_block();
    __syncthreads();  // pass 1's tables are in global memory, its LDS is free
    hs_collect_tile<HS_U, NSET, false>(A, ring, smem, tile, 0, rev != 0);
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

// QDM fallback: the flagged columns' factors (nq, nf) and their adjusted series back into the time-major field
__global__ void __launch_bounds__(XH_BLOCK)
k_hs_gather_af(const float* __restrict__ af, int64_t af_qs, int nq, const uint32_t* __restrict__ flist, int64_t nf, float* __restrict__ gaf) {
  const int64_t i = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (i >= nf * nq) return;
  const int64_t q = i / nf, f = i - q * nf;
  gaf[i] = af[q * af_qs + flist[f]];
}

__global__ void __launch_bounds__(XH_BLOCK)
k_hs_scatter_rows(const float* __restrict__ buf, int64_t T, int64_t Tp, const uint32_t* __restrict__ flist, float* __restrict__ out,
                  int64_t ost) {
  const int64_t c = flist[blockIdx.x];
  const float* src = buf + (int64_t)blockIdx.x * Tp;
  for (int64_t t = threadIdx.x; t < T; t += XH_BLOCK) out[t * ost + c] = src[t];
}

struct HsQdm {  // xh_qdm_hist's part of the call
  const float* af;
  int64_t af_qs;
  int kind, extrap;
  float* scen;
  int64_t ost;
};

int hs_run(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const double* d_q, int nq, float* out, int64_t out_cstride,
           int64_t out_qstride, const HsQdm* qd);

}  // namespace

// Quantiles of long series straight from the time-major (T, C) view.  XH_ERR_NOTIMPL when the shape does not fit or too
// many columns would need the column kernels (the caller then takes the transposed pipeline of eqm.hip for everything).
int xh_select_hist(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const double* d_q, int nq, float* out,
                   int64_t out_cstride, int64_t out_qstride) {
  return hs_run(ctx, x, T, C, st, d_q, nq, out, out_cstride, out_qstride, nullptr);
}

// QuantileDeltaMapping.adjust, interp = "nearest", on long time-major series: three streaming passes — histogram, collect
// (class boundaries as order statistics, ties followed exactly through the runs of equal candidates: qdmrank.h) and
// classification against the cut values (k_cut_classify) — instead of two transposes around a per-column exact ranking
// (T <= 32768: 228 ms at 10950 x 1440 x 720) or a global sort (T > 32768: 779 ms at 55152 x 1440 x 90).  Columns whose
// boundary ranks leave the collected bins go through the exact-rank kernels afterwards.
int xh_qdm_hist(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, const float* af, int64_t af_qs, const double* d_q,
                int nq, int kind, int extrap, float* scen, int64_t ost) {
  if (nq < 2 || scen == sim) return XH_ERR_NOTIMPL;
  if (xh_diag_env("XH_QDM_NOHIST")) return XH_ERR_NOTIMPL;  // A/B against the exact-rank kernels
  const HsQdm qd{af, af_qs, kind, extrap, scen, ost};
  return hs_run(ctx, sim, T, C, st, d_q, nq, nullptr, 0, 0, &qd);
}

namespace {

int hs_run(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const double* d_q, int nq, float* out, int64_t out_cstride,
           int64_t out_qstride, const HsQdm* qd) {
  if (T <= 1024 || T > 65535 || nq < 1 || nq > HS_MAXQ || C < 1) return XH_ERR_NOTIMPL;  // (65535: u16 counters)
  if ((unsigned long long)((HS_RL * 32 + HS_RL) * st + C) * 4ull >= (1ull << 32)) return XH_ERR_NOTIMPL;  // 32-bit offsets inside a batch
  if (xh_diag_env("XH_SELECT_NOHIST")) return XH_ERR_NOTIMPL;  // A/B against the transposed pipeline
  const int64_t ntiles = cdiv64(C, HS_CW);
  const int ntgt = 2 * nq;
  // fallback capacity: flagged columns are recomputed from a gathered copy, at most `nfmax` of them at a time
  const int64_t Tp = (T + 63) & ~(int64_t)63;
  int64_t nfmax = T > 32768 ? 1024 : 4096;
  if (nfmax > C) nfmax = C;
  auto al = [](size_t b) { return (b + 255) & ~(size_t)255; };
  const size_t b_lohi = al(sizeof(uint2) * (size_t)C), b_n = al(4 * (size_t)C), b_m = al(4 * (size_t)C), b_base = al(4 * (size_t)C);
  const size_t b_crank = al(2 * (size_t)ntiles * ntgt * HS_CW), b_bm = al(4 * (size_t)ntiles * 32 * HS_CW);
  const size_t b_tab = al(4 * (size_t)ntiles * 64 * HS_CW);
  const size_t b_flist = al(4 * (size_t)C), b_stat = al(sizeof(HsStat));
  const size_t b_gather = al(4 * (size_t)nfmax * (size_t)Tp), b_tmp = al(4 * (size_t)nfmax * (size_t)nq);
  // QDM: records, cut values, class factors, the fallback's output columns (+ the global sort's workspace beyond 32768 steps)
  const int nrec = 2 * (nq + 1);
  const size_t b_qrec = qd ? al(4 * (size_t)ntiles * nrec * 2 * HS_CW) + 2 * al(4 * (size_t)C) : 0;
  const size_t b_gcut = qd ? al(4 * (size_t)(nq + 1) * (size_t)C) : 0, b_gfac = qd ? al(4 * (size_t)(nq + 2) * (size_t)C) : 0;
  const size_t b_gout = qd ? b_gather : 0;
  size_t b_sorted = 0;
  if (qd && T > 32768) {
    const int rcw = xh_qdm_sorted_ws(T, nfmax, &b_sorted);
    if (rcw) return rcw;
    b_sorted = al(b_sorted);
  }
  void* ws = nullptr;
  int rc = xh_big_scratch(ctx, b_lohi + b_n + b_m + b_base + b_crank + b_bm + b_tab + b_flist + b_stat + b_gather + b_tmp + b_qrec + b_gcut +
                                   b_gfac + b_gout + b_sorted, &ws);
  if (rc) return rc;
  char* p = (char*)ws;
  uint2* lohi = (uint2*)p; p += b_lohi;
  uint32_t* meta_n = (uint32_t*)p; p += b_n;
  uint32_t* meta_m = (uint32_t*)p; p += b_m;
  uint32_t* meta_base = (uint32_t*)p; p += b_base;
  uint16_t* crank = (uint16_t*)p; p += b_crank;
  uint32_t* bitmap_g = (uint32_t*)p; p += b_bm;
  uint32_t* tab_g = (uint32_t*)p; p += b_tab;
  uint32_t* flist = (uint32_t*)p; p += b_flist;
  HsStat* stat = (HsStat*)p; p += b_stat;
  float* gbuf = (float*)p; p += b_gather;
  float* gtmp = (float*)p; p += b_tmp;
  uint32_t* qrec = (uint32_t*)p; p += b_qrec;
  float* colmin = qd ? (float*)((char*)qrec + al(4 * (size_t)ntiles * nrec * 2 * HS_CW)) : nullptr;
  float* colmax = qd ? (float*)((char*)colmin + al(4 * (size_t)C)) : nullptr;
  float* gcut = (float*)p; p += b_gcut;
  float* gfac = (float*)p; p += b_gfac;
  float* gout = (float*)p; p += b_gout;
  void* sws = (void*)p; p += b_sorted;
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
  const int64_t maxblk = (int64_t)ctx->num_cu;  // LDS: one 1024-thread workgroup per CU
  if (nblk > maxblk) nblk = maxblk;
  if (const char* eg = xh_diag_env("XH_HIST_GEOM")) {  // diagnostics: "<row lanes>,<sets>,<lds bytes>,<workgroups per CU>": time the bare loop, then go on
    int grl = 16, gns = 2, glds = 0, gwg = 1;
    sscanf(eg, "%d,%d,%d,%d", &grl, &gns, &glds, &gwg);
    int64_t nb = ntiles < (int64_t)ctx->num_cu * gwg ? ntiles : (int64_t)ctx->num_cu * gwg;
#define XH_HS_GEOM(RL, NS)                                                                                                       \
  {                                                                                                                             \
    XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_hs_stream_test<RL, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
    hipLaunchKernelGGL((k_hs_stream_test<RL, NS>), dim3((unsigned)nb), dim3(HS_CW * RL), (size_t)glds, ctx->stream, x, (int)T, C, st, stat); \
  }
    if (grl == 16 && gns == 2) XH_HS_GEOM(16, 2)
    else if (grl == 8 && gns == 2) XH_HS_GEOM(8, 2)
    else if (grl == 4 && gns == 2) XH_HS_GEOM(4, 2)
#undef XH_HS_GEOM
    XH_LAUNCH_CHECK();
  }
  const char* eabl = xh_diag_env("XH_HIST_ABL");  // diagnostics: 1 = no candidate sort, 2 = pass 1 loads only, 4 = no pass-1 tile epilogue, 32 = no appends, 64 = pass 2 loads only, 128 = no QDM epilogue, 256 = no counting of the extremes (wrong results); 2048 = one 512-slot sort instead of two runs (same results: A/B)
  const int abl = eabl ? atoi(eabl) : 0;
  // the streaming ring: 5 sets of 8 loads (32 to 40 loads per lane in flight).  Diagnostics: XH_HIST_RING=162 = two sets of 16
  // (round 3's ping-pong: 16 to 32 in flight; config-4 train 40.5 against 36.7-38.2 ms on the same box, profiles/r04/)
  const char* ens = xh_diag_env("XH_HIST_RING");
  const int ring = ens ? atoi(ens) : 85;
  HsArgs A;
  A.x = x; A.T = (int)T; A.C = C; A.st = st; A.lohi = lohi; A.qs = d_q; A.nq = nq;
  A.meta_n = meta_n; A.meta_m = meta_m; A.meta_base = meta_base; A.crank = crank; A.bitmap_g = bitmap_g; A.tab_g = tab_g;
  A.flist = flist; A.out = out; A.ocs = out_cstride; A.oqs = out_qstride; A.stat = stat; A.abl = abl;
  A.qdm = qd ? 1 : 0; A.af = qd ? qd->af : nullptr; A.af_qs = qd ? qd->af_qs : 0; A.extrap = qd ? qd->extrap : 0;
  A.qrec = qrec; A.gcut = gcut; A.gfac = gfac; A.colmin = colmin; A.colmax = colmax;
  // XH_HIST_FUSED (diagnostics): "1" = both passes of a tile in one kernel, second pass forward, "2" = ... in reverse row
  // order.  Default 0 = two kernels: measured in one process at config 4 (profiles/r05/select4_fused_ab.txt) 39.57 ms against
  // 41.25 (fused) and 40.97 (fused + reverse) per training — see k_hs_fused.
  const char* efu = xh_diag_env("XH_HIST_FUSED");
  const int fused = efu ? atoi(efu) : 0;
  const size_t lds_f = hs_lds1() > hs_lds2() ? hs_lds1() : hs_lds2();
#define XH_HS_TWO(UU, NS, QD)                                                                                                    \
  {                                                                                                                             \
    if (round == 0) {                                                                                                           \
      XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_hs_hist<UU, NS, QD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hs_lds1())); \
      hipLaunchKernelGGL((k_hs_hist<UU, NS, QD>), dim3((unsigned)nblk), dim3(HS_NT), hs_lds1(), ctx->stream, A);                \
      XH_LAUNCH_CHECK();                                                                                                        \
    }                                                                                                                           \
    XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_hs_collect<UU, NS, QD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)hs_lds2())); \
    hipLaunchKernelGGL((k_hs_collect<UU, NS, QD>), dim3((unsigned)nblk), dim3(HS_NT), hs_lds2(), ctx->stream, A, round);        \
  }
#define XH_HS_LAUNCH(UU, NS)                                                                                                     \
  {                                                                                                                             \
    if (round == 0 && fused && !qd) {                                                                                           \
      XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_hs_fused<UU, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_f)); \
      hipLaunchKernelGGL((k_hs_fused<UU, NS>), dim3((unsigned)nblk), dim3(HS_NT), lds_f, ctx->stream, A, fused == 2 ? 1 : 0);   \
    } else if (qd) XH_HS_TWO(UU, NS, true)                                                                                      \
    else XH_HS_TWO(UU, NS, false)                                                                                               \
  }
  HsStat h;
  for (int round = 0;; ++round) {  // round 0: pass 1 + pass 2; further rounds of pass 2 for tiles whose lists overflow the pool
    if (ring == 162 && !qd) XH_HS_LAUNCH(16, 2)
    else XH_HS_LAUNCH(8, 5)
    XH_LAUNCH_CHECK();
    XH_CHECK_HIP(hipMemcpyAsync(&h, stat, sizeof(HsStat), hipMemcpyDeviceToHost, ctx->stream));
    XH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    if (round >= (int)h.rounds) break;
  }
#undef XH_HS_TWO
#undef XH_HS_LAUNCH
  XH_REQUIRE(h.errors == 0 || abl != 0, XH_ERR_HIP, "xh_select_hist: %u columns met another candidate count in pass 2 than in pass 1",
             h.errors);
  if (xh_diag_env("XH_HIST_STATS")) fprintf(stderr, "[xh_select_hist] T=%lld C=%lld flagged=%u candidates: max %u mean %.1f, %u collect rounds\n", (long long)T, (long long)C, h.nflag, h.maxm, (double)h.summ / (double)C, h.rounds + 1u);
  if (qd) {
    // more than an eighth of the grid on the list (a heavily quantised field): the exact-rank pipeline for everything
    if ((int64_t)h.nflag > C / 8 && (int64_t)h.nflag > 4096) return XH_ERR_NOTIMPL;
    rc = xh_cut_classify(ctx, x, T, C, st, gcut, gfac, nq + 1, qd->kind, qd->scen, qd->ost);
    if (rc) return rc;
    if (xh_diag_env("XH_HIST_STATS")) fprintf(stderr, "[xh_qdm_hist] %u of %lld columns go through the exact-rank kernels\n", h.nflag, (long long)C);
    for (int64_t f0 = 0; f0 < (int64_t)h.nflag; f0 += nfmax) {
      const int64_t nf = (int64_t)h.nflag - f0 < nfmax ? (int64_t)h.nflag - f0 : nfmax;
      hipLaunchKernelGGL(k_hs_gather, dim3((unsigned)nf), dim3(XH_BLOCK), 0, ctx->stream, x, T, st, flist + f0, gbuf, Tp);
      hipLaunchKernelGGL(k_hs_gather_af, dim3((unsigned)cdiv64(nf * nq, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, qd->af, qd->af_qs, nq,
                         flist + f0, nf, gtmp);
      XH_LAUNCH_CHECK();
      rc = T > 32768 ? xh_qdm_sorted(ctx, gbuf, T, nf, Tp, gtmp, nf, d_q, nq, qd->kind, 0, qd->extrap, gout, Tp, sws)
                     : xh_qdm_columns(ctx, gbuf, T, nf, Tp, gtmp, nf, d_q, nq, qd->kind, 0, qd->extrap, gout, Tp);
      if (rc) return rc;
      hipLaunchKernelGGL(k_hs_scatter_rows, dim3((unsigned)nf), dim3(XH_BLOCK), 0, ctx->stream, gout, T, Tp, flist + f0, qd->scen, qd->ost);
      XH_LAUNCH_CHECK();
    }
    return XH_OK;
  }
  if (h.nflag == 0) return XH_OK;
  // up to 32768 steps the transposed pipeline is the better answer when MANY columns are flagged (heavily tied fields)
  if (T <= 32768 && (int64_t)h.nflag > nfmax) return XH_ERR_NOTIMPL;
  for (int64_t f0 = 0; f0 < (int64_t)h.nflag; f0 += nfmax) {
    const int64_t nf = (int64_t)h.nflag - f0 < nfmax ? (int64_t)h.nflag - f0 : nfmax;
    hipLaunchKernelGGL(k_hs_gather, dim3((unsigned)nf), dim3(XH_BLOCK), 0, ctx->stream, x, T, st, flist + f0, gbuf, Tp);
    XH_LAUNCH_CHECK();
    rc = xh_select_columns(ctx, gbuf, T, nf, Tp, d_q, nq, gtmp, 1, nf);
    if (rc) return rc;
    hipLaunchKernelGGL(k_hs_scatter, dim3((unsigned)cdiv64(nf * nq, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, gtmp, nf, nq,
                       flist + f0, out, out_cstride, out_qstride);
    XH_LAUNCH_CHECK();
  }
  return XH_OK;
}

}  // namespace
