// winsel.hip — quantiles of a SLIDING row sample: the training of a quantile mapping with Grouper("time.dayofyear", window=W)
// (xsdba's documented standard configuration, /root/reference/docs/sdba.rst:64-65; upstream per group: nbutils.quantile over
// the windowed block — xsdba >= 0.4 base.Grouper.get_index / apply, re-exported by /root/reference/src/xclim/sdba.py:10).
//
// The sample of day-of-year group g is the rows {day g - W/2 .. g + W/2 of every year}: nyears * W rows (930 for 30 years and
// W = 31), and the sample of g + 1 is the sample of g with ONE row per year replaced.  Rounds 2-5 selected every group from
// scratch — 365 x 2 selection problems of 930 samples per cell, 0.75 TB/s over the samples they select from, 508 ms for a
// 30-year 1440 x 90 band.  Here every cell keeps its window SORTED and updates it:
//
//   one wave per cell; the sorted window (keys) in LDS, rewritten in place
//   step g -> g + 1:  the leaving and the entering values of the cell (<= 64 each: one per year) are loaded one per lane and
//                     sorted through the wave (64-slot network, DPP exchanges); a leaving value's position in the window is a
//                     binary search (+ its index among equal leaving values: exactly one copy goes per leaving sample); every
//                     lane then rewrites a stretch of ~15 window elements to its new position = old position - leaving
//                     positions before it + entering values below it (two short binary searches at the stretch's start, then
//                     pointers that only advance), the entering values go to their rank + the surviving elements <= them
//   after every step: the 2 nq order statistics by POSITION (the window is sorted), Hyndman-Fan type 7 lerp as in every other
//                     selection kernel of this library (utl:395, 417-491 with alpha = beta = 1; fp32 difference, fp64 lerp),
//                     a NaN result from inf - inf becomes the window's largest valid sample (utl:552-554)
//
// NaN samples and absent rows (row index -1: the window reaches beyond the series) are never in the window; the valid count n
// follows the steps.  The 16 waves of a workgroup own 16 ADJACENT cells and stage the rows of a step together (each row
// segment of 64 bytes is read once by the workgroup: thread (row, cell) loads one sample, coalesced over the cells).
// Results are bit-identical to selecting every group from scratch (tests/test_gpu_api.py::test_eqm_doy_window_sliding_*: the
// per-group path stays behind XH_WINSEL=0 under XH_DIAGNOSTICS).
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int WS_CAP = 1024;     // window capacity (keys)
constexpr int WS_WAVES = 16;     // waves = cells per workgroup
constexpr int WS_PER = 64;       // rows leaving / entering per step at most
constexpr int WS_MAXQ = 32;      // quantiles (2 * nq targets <= 64 lanes)
constexpr uint32_t WS_INF = 0xFFFFFFFFu;

__device__ __forceinline__ uint32_t ws_key(float f) {  // order-preserving key of a non-NaN float
  const uint32_t u = __float_as_uint(f);
  return u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
}
__device__ __forceinline__ float ws_unkey(uint32_t k) {
  const uint32_t u = k ^ ((k >> 31) ? 0x80000000u : 0xFFFFFFFFu);
  return __uint_as_float(u);
}

// value of lane ^ m (select4.hip hs_lane_xor: DPP for m = 1, 2, 4, 8; the LDS crossbar for 16, 32)
__device__ __forceinline__ uint32_t ws_lane_xor(uint32_t v, int m) {
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

// wave-wide bitonic sort of 64 * K keys, K per lane (element i = lane * K + r), ascending (select4.hip hs_wave_sort)
template <int K>
__device__ __forceinline__ void ws_wave_sort(uint32_t (&v)[K], int lane) {
#pragma unroll
  for (int k = 2; k <= 64 * K; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (j >= K) {
        const int mlane = j / K;
        const bool up = ((lane * K) & k) == 0;
        const bool lower = (lane & mlane) == 0;
        const bool takemin = lower == up;
#pragma unroll
        for (int r = 0; r < K; ++r) {
          const uint32_t p = ws_lane_xor(v[r], mlane);
          v[r] = ((v[r] < p) == takemin) ? v[r] : p;
        }
      } else {
#pragma unroll
        for (int r = 0; r < K; ++r) {
          if ((r & j) == 0) {
            const uint32_t a = v[r], b = v[r | j];
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

// first index in [0, n) of the ascending keys `a` that is >= key (lower) / > key (upper)
__device__ __forceinline__ uint32_t ws_lower(const uint32_t* a, uint32_t n, uint32_t key) {
  uint32_t lo = 0u, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] < key) lo = mid + 1u; else hi = mid;
  }
  return lo;
}
__device__ __forceinline__ uint32_t ws_upper(const uint32_t* a, uint32_t n, uint32_t key) {
  uint32_t lo = 0u, hi = n;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a[mid] <= key) lo = mid + 1u; else hi = mid;
  }
  return lo;
}

// branch-free forms (fixed step counts, one LDS read per step): the step loop below is bound by instruction issue — written
// with `while (lo < hi)` loops and per-element event loops it ran ~2 500 instructions per cell and step, 900 of them scalar
// branch bookkeeping
__device__ __forceinline__ uint32_t ws_lower_w(const uint32_t* cur, uint32_t n, uint32_t key) {  // window: n <= 1024
  uint32_t pos = 0u;
#pragma unroll
  for (uint32_t sft = 1024u; sft >= 1u; sft >>= 1) {
    const uint32_t t = pos + sft;
    const uint32_t a = cur[(t - 1u) & 1023u];
    pos = (t <= n && a < key) ? t : pos;
  }
  return pos;
}
__device__ __forceinline__ uint32_t ws_lower_64(const uint32_t* a64, uint32_t n, uint32_t key) {  // n <= 64
  uint32_t pos = 0u;
#pragma unroll
  for (uint32_t sft = 64u; sft >= 1u; sft >>= 1) {
    const uint32_t t = pos + sft;
    const uint32_t a = a64[(t - 1u) & 63u];
    pos = (t <= n && a < key) ? t : pos;
  }
  return pos;
}
// 64-slot (SL = 64) or 32-slot sort of one key per lane
template <int SL>
__device__ __forceinline__ uint32_t ws_sort1(uint32_t v, int lane) {
#pragma unroll
  for (int k = 2; k <= SL; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
      const bool takemin = ((lane & j) == 0) == ((lane & k) == 0);
      const uint32_t p = ws_lane_xor(v, j);
      v = ((v < p) == takemin) ? v : p;
    }
  }
  return v;
}

// LDS per wave: the window [1024] | trash [64] | marks [516 words: per window position a byte "entering keys that go in front of the next
// element" and a byte "this element leaves"] | leaving positions [64] | entering keys [64] | picked keys [64]; per workgroup: the
// staged samples of a step [2 * WS_PER][WS_WAVES]
constexpr int ws_words_per_wave() { return WS_CAP + 64 + 516 + 3 * 64; }
// (per = 30 years: 77 824 + 3 840 bytes — two workgroups share a CU's 160 KiB)
inline size_t ws_lds(int per) { return (size_t)WS_WAVES * ws_words_per_wave() * 4 + (size_t)2 * (size_t)per * WS_WAVES * 4; }

// the targets' positions in a sorted sample of n (type 7; utl:395, 417-491): lane t < 2 nq holds the position of target t
// (qt = its quantile), lane j < nq the lerp weight of quantile j (qq)
__device__ __forceinline__ void ws_positions(uint32_t n, double qt, double qq, int lane, uint32_t& ppos, bool& pedge, double& pgamma) {
  ppos = 0u;
  if (n >= 2u) {
    const double nn = (double)n;
    const double vi = nn * qt + (1.0 + qt * (1.0 - 1.0 - 1.0)) - 1.0;
    if (vi >= nn - 1.0) ppos = n - 1u;
    else if (vi < 0.0) ppos = 0u;
    else ppos = (uint32_t)floor(vi) + (uint32_t)(lane & 1);
    const double v2 = nn * qq + (1.0 + qq * (1.0 - 1.0 - 1.0)) - 1.0;
    pedge = v2 >= nn - 1.0 || v2 < 0.0;
    pgamma = v2 - floor(v2);
  }
}

// NORM (the training of a DETRENDED quantile mapping, xsdba._adjustment.dqm_train): the quantiles are those of the sample
// normalised by its own mean — x - mu ("+", nmode 1) or x / mu ("*", nmode 3), in the arithmetic of xh_trend_apply (fp64
// operation, fp32 result) — and the mean of every group's sample goes to mu_out (G, C).  The mean is the fp64 sum over the
// sorted window (16 elements per lane, then over the lanes) / n.  The normalisation keeps the order of the samples when mu is
// finite (and positive for "*"): the picked samples are normalised, nothing else changes.  Otherwise (an infinite sample in the
// window, mu <= 0 for "*") samples can turn NaN and drop out of the sample, or the order reverses: the wave normalises the
// whole window and sorts it again for that group (the 1024-slot network of the first window).
template <bool NORM>
__global__ void __launch_bounds__(WS_WAVES * 64)
k_window_quantiles(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, const int32_t* __restrict__ rows0, int n0,
                   const int32_t* __restrict__ enter, const int32_t* __restrict__ leave, int G, int per,
                   const double* __restrict__ qs, int nq, float* __restrict__ out, int nmode, double* __restrict__ mu_out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  uint32_t* wbase = smem + wv * ws_words_per_wave();
  uint32_t* bufA = wbase;
  uint32_t* marks = wbase + WS_CAP + 64;    // (cur[1024 + lane]: the slot a lane's dropped elements are written to) [516] two bytes per window position: [2 p] entering keys with upper bound p, [2 p + 1] leaves
  uint32_t* lpos = marks + 516;             // sorted positions (in the current window) of the leaving samples
  uint32_t* ekey = lpos + 64;               // sorted keys of the entering samples
  uint32_t* tv = ekey + 64;                 // picked keys
  float* stage = reinterpret_cast<float*>(smem + WS_WAVES * ws_words_per_wave());  // [2 * per][WS_WAVES]
  const int64_t c0 = (int64_t)blockIdx.x * WS_WAVES;
  const int64_t c = c0 + wv;
  const bool cvalid = c < C;  // (wave-uniform; the workgroup's barriers are reached by every wave all the same)
  const double qq = qs[lane < nq ? lane : 0];            // lane j < nq: quantile j (the lerp)
  const double qt = qs[(lane >> 1) < nq ? (lane >> 1) : 0];  // lane t < 2 nq: the quantile of target t (its position)
  const int ntgt = 2 * nq;

  // ---- the first window: gather (one sample per lane and pass), keys with +inf for NaN / absent, 1024-slot sort
  uint32_t n = 0u;
  {
    uint32_t v[16];
    const int64_t cc = cvalid ? c : C - 1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = lane * 16 + r;
      const int32_t row = i < n0 ? rows0[i] : -1;
      const float f = x[(int64_t)(row < 0 ? 0 : row) * st + cc];
      const bool ok = row >= 0 && f == f;
      v[r] = ok ? ws_key(f + (-0.0f)) : WS_INF;
      n += ok ? 1u : 0u;
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) n += (uint32_t)__shfl_xor((int)n, d);
    ws_wave_sort<16>(v, lane);
#pragma unroll
    for (int r = 0; r < 16; ++r) bufA[lane * 16 + r] = v[r];
  }
  __builtin_amdgcn_wave_barrier();
  uint32_t* cur = bufA;  // the window lives in ONE buffer and is rewritten in place (every lane reads its stretch into registers first)

  // the samples that leave and enter at a step are staged by the whole workgroup: thread (row j, cell w) loads one, one step
  // AHEAD of its use (two values per thread at most: 2 * WS_PER * WS_WAVES = 2 * blockDim)
  auto fetch = [&](int g, float (&pf)[2]) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = tid + u * WS_WAVES * 64;
      const int j = i / WS_WAVES, w = i % WS_WAVES;
      const bool in = i < 2 * per * WS_WAVES && g + 1 < G;
      const int64_t o = (int64_t)(g + 1 < G ? g : 0) * per;
      const int32_t row = !in ? -1 : (j < per ? leave[o + j] : enter[o + (j - per)]);
      const int64_t cw = c0 + w < C ? c0 + w : C - 1;
      const float f = x[(int64_t)(row < 0 ? 0 : row) * st + cw];
      pf[u] = row < 0 ? xh_nan32() : f;
    }
  };
  float pf[2];
  fetch(0, pf);

  for (int i = lane; i < 516; i += 64) marks[i] = 0u;
  unsigned char* m8 = reinterpret_cast<unsigned char*>(marks);
  uint32_t nprev = 0xFFFFFFFFu, ppos = 0u;  // the targets' positions only change with the valid count
  double pgamma = 0.0;
  bool pedge = true;
  __builtin_amdgcn_wave_barrier();

  for (int g = 0; g < G; ++g) {
    // ---- the quantiles of the current window (positions: the window is sorted; type 7, utl:395, 417-491)
    if (n != nprev) {  // (wave-uniform)
      nprev = n;
      ws_positions(n, qt, qq, lane, ppos, pedge, pgamma);
    }
    double mu = 0.0;
    bool ordered = true;   // the normalisation keeps the window's order and all of its samples
    if (NORM) {
      const uint4* c4 = reinterpret_cast<const uint4*>(cur + (uint32_t)lane * 16u);
      const uint4 q0 = c4[0], q1 = c4[1], q2 = c4[2], q3 = c4[3];
      const uint32_t e[16] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, q3.x, q3.y, q3.z, q3.w};
      double sum = 0.0;
#pragma unroll
      for (int r = 0; r < 16; ++r) sum += ((uint32_t)lane * 16u + (uint32_t)r < n) ? (double)ws_unkey(e[r]) : 0.0;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) sum += __shfl_xor(sum, d);   // (a + b == b + a: every lane ends with the same bits)
      mu = n > 0u ? sum / (double)n : xh_nan64();
      if (lane == 0 && cvalid) mu_out[(int64_t)g * C + c] = mu;
      const bool fin = mu - mu == 0.0;
      // (an empty window — a masked cell — has nothing to normalise: not the re-sorting path)
      ordered = __builtin_amdgcn_readfirstlane((int)(n == 0u || (fin && (nmode == 1 || mu > 0.0)))) != 0;
    }
    auto norm = [&](float f) -> float {
      if (!NORM) return f;
      const double v = (double)f;
      return (float)(nmode == 1 ? v - mu : v / mu);
    };
    uint32_t nn = n;       // the sample the quantiles are taken from (NORM, not ordered: the normalised window without its NaNs)
    bool edge = pedge;
    double gamma = pgamma;
    uint32_t topkey = 0u;
    if (!NORM || ordered) {
      if (lane < ntgt) tv[lane] = n > 0u ? cur[ppos] : WS_INF;
      topkey = n > 0u ? cur[n - 1u] : WS_INF;
    } else {
      uint32_t v[16];
      nn = 0u;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const uint32_t i = (uint32_t)lane * 16u + (uint32_t)r;
        const float y = norm(ws_unkey(cur[i]));
        const bool ok = i < n && y == y;
        v[r] = ok ? ws_key(y + (-0.0f)) : WS_INF;
        nn += ok ? 1u : 0u;
      }
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) nn += (uint32_t)__shfl_xor((int)nn, d);
      ws_wave_sort<16>(v, lane);
      uint32_t pp = 0u;
      ws_positions(nn, qt, qq, lane, pp, edge, gamma);
      auto at = [&](uint32_t p) -> uint32_t {   // element p of the sorted sample (lane p / 16 holds it in v[p % 16])
        uint32_t val = 0u;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const uint32_t t = (uint32_t)__shfl((int)v[r], (int)(p >> 4));
          val = (p & 15u) == (uint32_t)r ? t : val;
        }
        return val;
      };
      const uint32_t got = at(pp);
      topkey = at(nn > 0u ? nn - 1u : 0u);
      if (lane < ntgt) tv[lane] = nn > 0u ? got : WS_INF;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < nq && cvalid) {
      double r;
      if (nn == 0u) r = xh_nan64();
      else {
        const bool raw = !NORM || ordered;    // the picked keys are window samples still to be normalised
        const float l0 = ws_unkey(tv[2 * lane]), r0 = ws_unkey(tv[2 * lane + 1]), t0 = ws_unkey(topkey);
        const float left = raw ? norm(l0) : l0, right = raw ? norm(r0) : r0;
        if (nn < 2u || edge) r = (double)left;
        else {
          const float diff = right - left;
          r = (double)left + (double)diff * gamma;
          if (gamma >= 0.5) r = (double)right - (double)diff * (1.0 - gamma);
        }
        if (r != r) r = (double)(raw ? norm(t0) : t0);  // inf - inf: the largest valid sample (utl:552-554)
      }
      out[((int64_t)g * nq + lane) * C + c] = (float)r;
    }
    if (g + 1 == G) break;
    __syncthreads();  // (the previous step's staged samples are consumed)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int i = tid + u * WS_WAVES * 64;
      if (i < 2 * per * WS_WAVES) stage[i] = pf[u];   // (i = j * WS_WAVES + w)
    }
    __syncthreads();
    fetch(g + 1, pf);  // the next step's samples fly while this one is worked
    const float fl = lane < per ? stage[lane * WS_WAVES + wv] : xh_nan32();
    const float fe = lane < per ? stage[(per + lane) * WS_WAVES + wv] : xh_nan32();
    uint32_t kl = fl == fl ? ws_key(fl + (-0.0f)) : WS_INF;
    uint32_t ke = fe == fe ? ws_key(fe + (-0.0f)) : WS_INF;
    const uint32_t nl = (uint32_t)__popcll(__ballot(fl == fl)), ne = (uint32_t)__popcll(__ballot(fe == fe));
    if (per <= 32) { kl = ws_sort1<32>(kl, lane); ke = ws_sort1<32>(ke, lane); }
    else { kl = ws_sort1<64>(kl, lane); ke = ws_sort1<64>(ke, lane); }
    // Sorted leaving key i: its position = lower bound in the window + its index among the equal leaving keys before it (exactly
    // one copy goes per leaving sample).  Sorted entering key: ub = the window elements <= it; it lands at its rank among the
    // entering keys + ub - the leaving positions below ub.  A surviving element i moves to i - #{leaving positions < i} +
    // #{entering keys with ub <= i}: both counts are prefix sums over POSITIONS — two bytes of marks per position, 16 positions
    // per lane, one wave scan.
    const unsigned long long upto = lane == 63 ? ~0ull : ((1ull << (lane + 1)) - 1ull);
    uint32_t lp, ub, erun;
    {
      const uint32_t prevl = (uint32_t)__shfl_up((int)kl, 1);
      const unsigned long long sl = __ballot(lane > 0 && prevl == kl);
      const int startl = 63 - __builtin_clzll(~sl & upto);  // (lane 0 starts a run)
      // both searches over the window in one unrolled chain of 11 steps
      uint32_t p1 = 0u, p2 = 0u;
#pragma unroll
      for (uint32_t sft = 1024u; sft >= 1u; sft >>= 1) {
        const uint32_t t1 = p1 + sft, t2 = p2 + sft;
        const uint32_t a1 = cur[(t1 - 1u) & 1023u], a2 = cur[(t2 - 1u) & 1023u];
        p1 = (t1 <= n && a1 < kl) ? t1 : p1;     // lower bound of the leaving key
        p2 = (t2 <= n && a2 <= ke) ? t2 : p2;    // upper bound of the entering key
      }
      lp = p1 + (uint32_t)(lane - startl);
      ub = p2;
      // (equal upper bounds, not equal keys, share a mark: entering keys that differ can still go in front of the same element)
      const uint32_t prevu = (uint32_t)__shfl_up((int)ub, 1);
      const unsigned long long su = __ballot(lane > 0 && prevu == ub && (uint32_t)lane < ne);
      const int startu = 63 - __builtin_clzll(~su & upto);
      const bool lastu = (uint32_t)lane + 1u >= ne || (((su >> ((lane + 1) & 63)) & 1ull) == 0ull) || lane == 63;
      erun = lastu ? (uint32_t)(lane - startu + 1) : 0u;   // the last lane of a run of equal upper bounds marks the run's length
      if ((uint32_t)lane < nl) {
        lpos[lane] = lp;
        m8[2u * lp + 1u] = 1;
      }
      if ((uint32_t)lane < ne && erun && ub < 1024u) m8[2u * ub] = (unsigned char)erun;
    }
    __builtin_amdgcn_wave_barrier();
    {
      // the entering keys' positions (before anything moves)
      uint32_t epos = 0u;
      if ((uint32_t)lane < ne) epos = (uint32_t)lane + ub - ws_lower_64(lpos, nl, ub);
      // this lane's stretch: 16 positions, their elements and marks
      const uint32_t i0 = (uint32_t)lane * 16u;
      uint32_t e[16], mk[8];
      {
        const uint4* c4 = reinterpret_cast<const uint4*>(cur + i0);
        const uint4 q0 = c4[0], q1 = c4[1], q2 = c4[2], q3 = c4[3];
        e[0] = q0.x; e[1] = q0.y; e[2] = q0.z; e[3] = q0.w; e[4] = q1.x; e[5] = q1.y; e[6] = q1.z; e[7] = q1.w;
        e[8] = q2.x; e[9] = q2.y; e[10] = q2.z; e[11] = q2.w; e[12] = q3.x; e[13] = q3.y; e[14] = q3.z; e[15] = q3.w;
        const uint4* m4 = reinterpret_cast<const uint4*>(marks + 8u * (uint32_t)lane);
        const uint4 r0 = m4[0], r1 = m4[1];
        mk[0] = r0.x; mk[1] = r0.y; mk[2] = r0.z; mk[3] = r0.w; mk[4] = r1.x; mk[5] = r1.y; mk[6] = r1.z; mk[7] = r1.w;
      }
      // stretch totals (leaving << 16 | entering) and their exclusive prefix over the lanes below
      uint32_t tot = 0u;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const uint32_t m = mk[w];
        tot += (m & 0xFFu) + ((m >> 16) & 0xFFu) + (((m >> 8) & 1u) << 16) + (((m >> 24) & 1u) << 16);
      }
      uint32_t incl = tot;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
        incl += lane >= d ? o : 0u;
      }
      uint32_t run = incl - tot;   // high half: leaving positions below i0; low half: entering keys with ub below i0
      uint32_t dst[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const uint32_t m = (mk[j >> 1] >> ((j & 1) * 16)) & 0xFFFFu;
        const uint32_t i = i0 + (uint32_t)j;
        run += m & 0xFFu;                                   // entering keys with ub <= i
        const bool gone = (m >> 8) != 0u;
        dst[j] = (i < n && !gone) ? i - (run >> 16) + (run & 0xFFFFu) : 1024u + (uint32_t)lane;   // (dropped: a private trash slot)
        run += gone ? 0x10000u : 0u;                        // leaving positions < the next i
      }
      __builtin_amdgcn_wave_barrier();  // every lane holds its stretch: the window may be overwritten
#pragma unroll
      for (int j = 0; j < 16; ++j) cur[dst[j]] = e[j];   // (unconditional: sixteen stores under one execution mask)
      if ((uint32_t)lane < ne) cur[epos] = ke;
      // the marks go back to zero
      if ((uint32_t)lane < nl) m8[2u * lp + 1u] = 0;
      if ((uint32_t)lane < ne && erun && ub < 1024u) m8[2u * ub] = 0;
    }
    n = n - nl + ne;
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- small groups: the training with a day-of-year grouping WITHOUT a window — 365 groups of one row per year.  Nothing slides
// (every step replaces the whole sample) and the per-group path is launch-bound (365 gathers + selections on 30-row blocks:
// 121 ms for the EQM tables of a 30-year 1440 x 90 band, 470 ms for DQM with its means and normalisations).  ONE launch:
// thread = one cell of one group, the group's rows (<= PER, listed by row number) as keys in registers, a bitonic network over
// the register array, the sorted keys to the thread's LDS column, the quantiles by position (type 7, the lerp of every
// selection kernel: ws_positions).  NORM (dqm_train): the mean over the valid samples in the list's order (bit-identical to
// xh_poly_trend degree 0 on the gathered rows), every sample normalised BEFORE the sort exactly as xh_trend_apply does (a
// sample that turns NaN drops out like any NaN).
template <int PER, int NT, bool NORM>
__global__ void __launch_bounds__(NT)
k_group_quantiles(const float* __restrict__ x, int64_t C, int64_t st, const int32_t* __restrict__ rows, const int64_t* __restrict__ offs,
                  const double* __restrict__ qs, int nq, float* __restrict__ out, int nmode, double* __restrict__ mu_out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t gq_keys[];   // [PER][NT]
  const int tid = threadIdx.x;
  const int64_t c = (int64_t)blockIdx.x * NT + tid;
  if (c >= C) return;
  const int64_t g = blockIdx.y;
  const int64_t k0 = offs[g];
  const int m = (int)(offs[g + 1] - k0);
  float f[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int32_t row = j < m ? rows[k0 + j] : 0;
    const float t = x[(int64_t)row * st + c];
    f[j] = j < m ? t : xh_nan32();
  }
  if (NORM) {
    double sx = 0.0, cnt = 0.0;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const bool ok = f[j] == f[j];
      cnt += ok ? 1.0 : 0.0;
      sx += ok ? (double)f[j] : 0.0;
    }
    const double mu = cnt > 0.0 ? sx / cnt : xh_nan64();
    mu_out[g * C + c] = mu;
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const double v = (double)f[j];
      f[j] = (float)(nmode == 1 ? v - mu : v / mu);
    }
  }
  uint32_t v[PER];
  uint32_t n = 0u;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const bool ok = f[j] == f[j];
    v[j] = ok ? ws_key(f[j] + (-0.0f)) : WS_INF;
    n += ok ? 1u : 0u;
  }
#pragma unroll
  for (int k = 2; k <= PER; k <<= 1) {
#pragma unroll
    for (int j = k >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const uint32_t a = v[i], b = v[l];
          const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
          const bool up = (i & k) == 0;
          v[i] = up ? lo : hi;
          v[l] = up ? hi : lo;
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < PER; ++j) gq_keys[j * NT + tid] = v[j];
  // (the column is the thread's own: no barrier)
  for (int j = 0; j < nq; ++j) {
    double r;
    if (n == 0u) r = xh_nan64();
    else {
      const double q = qs[j], nn = (double)n;
      uint32_t plo = 0u, phi = 0u;
      bool edge = true;
      double gamma = 0.0;
      if (n >= 2u) {
        const double vi = nn * q + (1.0 + q * (1.0 - 1.0 - 1.0)) - 1.0;
        if (vi >= nn - 1.0) plo = phi = n - 1u;
        else if (vi < 0.0) plo = phi = 0u;
        else { plo = (uint32_t)floor(vi); phi = plo + 1u; }
        edge = vi >= nn - 1.0 || vi < 0.0;
        gamma = vi - floor(vi);
      }
      const float left = ws_unkey(gq_keys[plo * NT + tid]), right = ws_unkey(gq_keys[phi * NT + tid]);
      if (n < 2u || edge) r = (double)left;
      else {
        const float diff = right - left;
        r = (double)left + (double)diff * gamma;
        if (gamma >= 0.5) r = (double)right - (double)diff * (1.0 - gamma);
      }
      if (r != r) r = (double)ws_unkey(gq_keys[(n - 1u) * NT + tid]);  // inf - inf: the largest valid sample (utl:552-554)
    }
    out[(g * nq + j) * C + c] = (float)r;
  }
}

__global__ void __launch_bounds__(XH_BLOCK)
k_correction(const float* __restrict__ hq, int64_t n, int kind, float* __restrict__ af) {  // af holds ref_q on entry
  const int64_t i = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (i >= n) return;
  const float r = af[i], h = hq[i];
  af[i] = kind == 0 ? (r - h) : (r / h);
}

__global__ void __launch_bounds__(XH_BLOCK)
k_scaling(const double* __restrict__ mu_h, int64_t n, int kind, double* __restrict__ sc) {  // sc holds mu_ref on entry
  const int64_t i = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (i >= n) return;
  const double r = sc[i], h = mu_h[i];
  sc[i] = kind == 0 ? (r - h) : (r / h);
}

// both entry points: the arguments checked, the schedule on the device, the two window kernels and the correction
int ws_train(xh_ctx* ctx, const char* who, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st, const int32_t* rows0,
             int n0, const int32_t* enter, const int32_t* leave, int G, int per, const double* q, int nq, int kind, float* af,
             float* hist_q, double* scaling, double* mu_hist) {
  XH_REQUIRE(ctx && ref && hist && rows0 && q && af && hist_q && (G == 1 || (enter && leave)), XH_ERR_ARG, "%s: NULL argument", who);
  XH_REQUIRE(T >= 1 && C >= 0 && G >= 1 && n0 >= 1 && per >= 0 && nq >= 1 && st >= C, XH_ERR_ARG, "%s: bad shape", who);
  XH_REQUIRE(kind == 0 || kind == 1, XH_ERR_ARG, "%s: kind must be 0 (+) or 1 (*)", who);
  if (n0 > WS_CAP || per > WS_PER || nq > WS_MAXQ) return XH_ERR_NOTIMPL;
  if (const char* e = xh_diag_env("XH_WINSEL"))
    if (!atoi(e)) return XH_ERR_NOTIMPL;
  for (int i = 0; i < n0; ++i) XH_REQUIRE(rows0[i] >= -1 && rows0[i] < T, XH_ERR_ARG, "%s: rows0[%d] out of range", who, i);
  for (int64_t i = 0; i < (int64_t)(G - 1) * per; ++i)
    XH_REQUIRE(enter[i] >= -1 && enter[i] < T && leave[i] >= -1 && leave[i] < T, XH_ERR_ARG, "%s: step row out of range", who);
  if (C == 0) return XH_OK;
  // the window never outgrows its buffers: the valid samples are at most n0 + sum(entering - leaving) <= the rows present
  {
    int64_t present = 0, most = 0;
    for (int i = 0; i < n0; ++i) present += rows0[i] >= 0;
    most = present;
    for (int g = 0; g + 1 < G; ++g) {
      for (int j = 0; j < per; ++j) present += (enter[(int64_t)g * per + j] >= 0) - (leave[(int64_t)g * per + j] >= 0);
      most = present > most ? present : most;
    }
    if (most > WS_CAP) return XH_ERR_NOTIMPL;
  }
  size_t cur = 0;
  void *d_q = nullptr, *d_r0 = nullptr, *d_en = nullptr, *d_lv = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, q, sizeof(double) * (size_t)nq, &d_q);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, rows0, sizeof(int32_t) * (size_t)n0, &d_r0);
  if (!rc && G > 1) rc = xh_scratch_upload(ctx, &cur, enter, sizeof(int32_t) * (size_t)(G - 1) * (size_t)per, &d_en);
  if (!rc && G > 1) rc = xh_scratch_upload(ctx, &cur, leave, sizeof(int32_t) * (size_t)(G - 1) * (size_t)per, &d_lv);
  if (rc) return rc;
  const dim3 grid((unsigned)cdiv64(C, WS_WAVES)), block(WS_WAVES * 64);
  const int64_t tot = (int64_t)G * nq * C;
  // ref_q goes to `af` first, then af = correction(ref_q, hist_q) in place (as xh_eqm_train)
  if (!scaling) {
    XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_window_quantiles<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_lds(WS_PER)));
    hipLaunchKernelGGL(k_window_quantiles<false>, grid, block, ws_lds(per), ctx->stream, ref, T, C, st, (const int32_t*)d_r0, n0,
                       (const int32_t*)d_en, (const int32_t*)d_lv, G, per, (const double*)d_q, nq, af, 0, (double*)nullptr);
    XH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_window_quantiles<false>, grid, block, ws_lds(per), ctx->stream, hist, T, C, st, (const int32_t*)d_r0, n0,
                       (const int32_t*)d_en, (const int32_t*)d_lv, G, per, (const double*)d_q, nq, hist_q, 0, (double*)nullptr);
    XH_LAUNCH_CHECK();
  } else {
    const int nmode = kind == 0 ? 1 : 3;   // xh_trend_apply's "-" and "/"
    XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_window_quantiles<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ws_lds(WS_PER)));
    hipLaunchKernelGGL(k_window_quantiles<true>, grid, block, ws_lds(per), ctx->stream, ref, T, C, st, (const int32_t*)d_r0, n0,
                       (const int32_t*)d_en, (const int32_t*)d_lv, G, per, (const double*)d_q, nq, af, nmode, scaling);
    XH_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_window_quantiles<true>, grid, block, ws_lds(per), ctx->stream, hist, T, C, st, (const int32_t*)d_r0, n0,
                       (const int32_t*)d_en, (const int32_t*)d_lv, G, per, (const double*)d_q, nq, hist_q, nmode, mu_hist);
    XH_LAUNCH_CHECK();
    const int64_t ng = (int64_t)G * C;
    hipLaunchKernelGGL(k_scaling, dim3((unsigned)cdiv64(ng, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, (const double*)mu_hist, ng, kind, scaling);
    XH_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_correction, dim3((unsigned)cdiv64(tot, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, hist_q, tot, kind, af);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

// both small-group entry points (see k_group_quantiles)
int gq_train(xh_ctx* ctx, const char* who, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st, const int32_t* rows,
             const int64_t* offs, int G, const double* q, int nq, int kind, float* af, float* hist_q, double* scaling, double* mu_hist) {
  XH_REQUIRE(ctx && ref && hist && rows && offs && q && af && hist_q, XH_ERR_ARG, "%s: NULL argument", who);
  XH_REQUIRE(T >= 1 && C >= 0 && G >= 1 && nq >= 1 && st >= C, XH_ERR_ARG, "%s: bad shape", who);
  XH_REQUIRE(kind == 0 || kind == 1, XH_ERR_ARG, "%s: kind must be 0 (+) or 1 (*)", who);
  XH_REQUIRE(offs[0] == 0, XH_ERR_ARG, "%s: offs[0] must be 0", who);
  int64_t most = 0;
  for (int g = 0; g < G; ++g) {
    XH_REQUIRE(offs[g + 1] >= offs[g], XH_ERR_ARG, "%s: offs must not decrease", who);
    most = offs[g + 1] - offs[g] > most ? offs[g + 1] - offs[g] : most;
  }
  const int64_t nr = offs[G];
  for (int64_t k = 0; k < nr; ++k) XH_REQUIRE(rows[k] >= 0 && rows[k] < T, XH_ERR_ARG, "%s: row %lld out of range", who, (long long)k);
  if (most > 64 || nq > 1024) return XH_ERR_NOTIMPL;
  if (const char* e = xh_diag_env("XH_TRAIN_GROUPS"))
    if (!atoi(e)) return XH_ERR_NOTIMPL;
  if (C == 0) return XH_OK;
  size_t cur = 0;
  void *d_q = nullptr, *d_rows = nullptr, *d_offs = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, q, sizeof(double) * (size_t)nq, &d_q);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, offs, sizeof(int64_t) * (size_t)(G + 1), &d_offs);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, rows, sizeof(int32_t) * (size_t)(nr > 0 ? nr : 1), &d_rows);
  if (rc) return rc;
  const int nmode = kind == 0 ? 1 : 3;   // xh_trend_apply's "-" and "/"
#define XH_GQ(PER, NT, NORM, X, OUT, MU)                                                                                              \
  hipLaunchKernelGGL((k_group_quantiles<PER, NT, NORM>), dim3((unsigned)cdiv64(C, NT), (unsigned)G), dim3(NT), (size_t)PER * NT * 4, ctx->stream, X, C, \
                     st, (const int32_t*)d_rows, (const int64_t*)d_offs, (const double*)d_q, nq, OUT, nmode, MU)
  // ref_q goes to `af` first, then af = correction(ref_q, hist_q) in place (as xh_eqm_train)
  if (!scaling) {
    if (most <= 32) { XH_GQ(32, 256, false, ref, af, (double*)nullptr); XH_GQ(32, 256, false, hist, hist_q, (double*)nullptr); }
    else { XH_GQ(64, 128, false, ref, af, (double*)nullptr); XH_GQ(64, 128, false, hist, hist_q, (double*)nullptr); }
  } else {
    if (most <= 32) { XH_GQ(32, 256, true, ref, af, scaling); XH_GQ(32, 256, true, hist, hist_q, mu_hist); }
    else { XH_GQ(64, 128, true, ref, af, scaling); XH_GQ(64, 128, true, hist, hist_q, mu_hist); }
    const int64_t ng = (int64_t)G * C;
    hipLaunchKernelGGL(k_scaling, dim3((unsigned)cdiv64(ng, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, (const double*)mu_hist, ng, kind, scaling);
  }
#undef XH_GQ
  XH_LAUNCH_CHECK();
  const int64_t tot = (int64_t)G * nq * C;
  hipLaunchKernelGGL(k_correction, dim3((unsigned)cdiv64(tot, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, hist_q, tot, kind, af);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // namespace

extern "C" {

// EQM training over a sliding row sample (see the head of this file).  rows0 [n0 <= 1024]: the rows of the first group's sample
// (-1: beyond the series); leave / enter [G - 1][per <= 64]: the rows that leave / enter at the step from group g to g + 1 (-1:
// none).  af, hist_q: (G, nq, C).  XH_ERR_NOTIMPL (no error text): not this kernel's shape — the caller selects every group
// from its gathered sample (xh_eqm_train).
int xh_eqm_train_window(xh_ctx* ctx, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st, const int32_t* rows0,
                        int n0, const int32_t* enter, const int32_t* leave, int G, int per, const double* q, int nq, int kind,
                        float* af, float* hist_q) {
  return ws_train(ctx, "xh_eqm_train_window", ref, hist, T, C, st, rows0, n0, enter, leave, G, per, q, nq, kind, af, hist_q, nullptr,
                  nullptr);
}

// DQM training over the same sliding sample (xsdba._adjustment.dqm_train per day-of-year group with a window): af / hist_q from the
// quantiles of the samples normalised by their own means (x - mean for "+", x / mean for "*"; xh_poly_trend degree 0 +
// xh_trend_apply + xh_eqm_train per group otherwise), scaling (G, C) float64 = mean(ref) - mean(hist) resp. the ratio, mu_hist
// (G, C) float64 = the means of hist (the second mean table the correction needs: an output like the others).
int xh_dqm_train_window(xh_ctx* ctx, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st, const int32_t* rows0,
                        int n0, const int32_t* enter, const int32_t* leave, int G, int per, const double* q, int nq, int kind,
                        float* af, float* hist_q, double* scaling, double* mu_hist) {
  XH_REQUIRE(scaling && mu_hist, XH_ERR_ARG, "xh_dqm_train_window: NULL argument");
  return ws_train(ctx, "xh_dqm_train_window", ref, hist, T, C, st, rows0, n0, enter, leave, G, per, q, nq, kind, af, hist_q, scaling,
                  mu_hist);
}

// EQM / DQM training for ALL groups of a sub-grouping with small groups in one launch per field (a day-of-year grouping without a
// window: one row per year).  rows (host, offs[G] entries): the row numbers of group 0, then of group 1, ... (a group's rows in
// the order its mean is summed); offs (host, G + 1).  af, hist_q: (G, nq, C).  Bit-identical to xh_eqm_train on each group's
// gathered rows; xh_dqm_train_groups: to xh_poly_trend (degree 0) + xh_trend_apply + xh_eqm_train per group, scaling / mu_hist
// (G, C) float64 as xh_dqm_train_window.  XH_ERR_NOTIMPL (no error text): a group of more than 64 rows.
int xh_eqm_train_groups(xh_ctx* ctx, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st, const int32_t* rows,
                        const int64_t* offs, int G, const double* q, int nq, int kind, float* af, float* hist_q) {
  return gq_train(ctx, "xh_eqm_train_groups", ref, hist, T, C, st, rows, offs, G, q, nq, kind, af, hist_q, nullptr, nullptr);
}

int xh_dqm_train_groups(xh_ctx* ctx, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st, const int32_t* rows,
                        const int64_t* offs, int G, const double* q, int nq, int kind, float* af, float* hist_q, double* scaling,
                        double* mu_hist) {
  XH_REQUIRE(scaling && mu_hist, XH_ERR_ARG, "xh_dqm_train_groups: NULL argument");
  return gq_train(ctx, "xh_dqm_train_groups", ref, hist, T, C, st, rows, offs, G, q, nq, kind, af, hist_q, scaling, mu_hist);
}

}  // extern "C"
