// select2.hip — long-series multi-quantile selection WITHOUT a per-column copy of the series in LDS.
//
// One workgroup per column, the column's keys stay in registers (KPL per thread).  Only the keys that fall into the
// <= 2*nq TARGET bins matter:
//   A  load (unconditional clamped loads, all in flight at once), key conversion, (n, kmin, kmax) reduction
//   B  target ranks (Hyndman-Fan type 7, utl:395) -> LDS; histogram of NB linear-in-key bins (ds_add_u32)
//   C  exclusive scan; every thread looks for the target ranks that fall into ITS bins (ranks are broadcast from
//      registers with readlane — no dependent LDS chains), allocates the bin's list region (or a min/max slot for bins
//      with more than BIGM keys, e.g. the "exact zero" bin of a precipitation series) and tags cur[bin] with the cursor
//   D  keys whose bin is tagged are appended to the LDS list / update the big bin's min & max
//   E  exact k-th smallest inside the (tiny) bin lists, (target x candidate) pairs spread over the workgroup; a bin
//      that is big AND not constant falls back to a bisection on the key value with workgroup-wide counting sweeps
//   F  Hyndman-Fan lerp and store
// The phases are separated by LDS-only barriers (s_waitcnt lgkmcnt(0); s_barrier): __syncthreads() would also wait
// for the result stores (vmcnt(0)) — a full memory round trip per column.  Measured phase costs: XH_SELECT_PROF=1.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"

#ifndef XH_LEAN_MINB
#define XH_LEAN_MINB 4
#endif

namespace {

constexpr int BIGM = 64;       // target bins up to this population are listed
constexpr int LISTCAP = 2048;  // LDS list capacity (keys); more -> the affected bins are treated as "big"
constexpr int MAXT = 128;      // targets = 2 * nq <= 128
constexpr uint32_t F_LIST = 0x80000000u, F_BIG = 0x40000000u, F_MASK = 0x3FFFFFFFu;

struct TInfo {
  int bin, kth, m, region;  // region: list offset, or -(slot+1) for big bins
};

// workgroup barrier that only orders LDS traffic (no vmcnt wait)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NT>
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  lds_barrier();
  if (lane == 0) red[w] = v;
  lds_barrier();
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) s += red[i];
  return s;
}

// DIAG = false (production): the ablation switches and phase timers are compiled out — each of them is a uniform
// condition held in scalar registers across the column loop, and the kernel already spills SGPRs into VGPR lanes.
// KSAFE: the first KSAFE key steps are known at compile time to lie inside the series for every T this instantiation
// is used with (no index test, no clamp on the load).
template <int NT, int KPL, int NB, int KSAFE = 0, bool DIAG = false>
__global__ void __launch_bounds__(NT, XH_LEAN_MINB)
k_select_lean(const float* __restrict__ x, int64_t T, int64_t ncols, int64_t col_stride, const double* __restrict__ qs,
              int nq, float* __restrict__ out, int64_t out_cstride, int64_t out_qstride, int abl_,
              unsigned long long* __restrict__ prof_) {
  const int abl = DIAG ? abl_ : 0;
  unsigned long long* __restrict__ prof = DIAG ? prof_ : nullptr;
  constexpr int BPT = NB / NT;
  constexpr int NW = NT / 64;
  __shared__ uint32_t cur[NB + 1];  // [NB]: dummy bin of the NaN keys (keeps the histogram atomics branch-free)
  __shared__ TInfo tinfo[MAXT];
  __shared__ int trank[MAXT];
  __shared__ uint32_t bmin[MAXT], bmax[MAXT];
  __shared__ uint32_t list[LISTCAP + 2 * 8 + 4];  // padded: the in-bin select reads (masked) past a bin's keys
  __shared__ float vals[MAXT];
  __shared__ uint32_t red[5 * NW + 8];
  __shared__ uint32_t texcl[NT];  // exclusive prefix of the bin counts inside each wave (target search)
  __shared__ int s_slow, s_off, s_nslot;
  const int gt = threadIdx.x;
  const int lane = gt & 63, w = gt >> 6;
  const int ntgt = 2 * nq;
  const double tq = gt < ntgt ? qs[gt >> 1] : 0.0;  // quantile of "my" target / of "my" output (gt < nq)
  const double oq = gt < nq ? qs[gt] : 0.0;

  // phase timers (diagnostics, XH_SELECT_PROF=1): thread 0 accumulates cycle-counter deltas per phase
  unsigned long long tprev = prof ? __builtin_readcyclecounter() : 0ull;
#define XH_PHASE(i)                                              \
  if (prof && gt == 0) {                                         \
    unsigned long long tn = __builtin_readcyclecounter();        \
    atomicAdd(&prof[i], tn - tprev);                             \
    tprev = tn;                                                  \
  }
  for (int64_t col = blockIdx.x; col < ncols; col += gridDim.x) {
    // ---- A: load + one-pass reduction of (n, kmin, kmax).  Unconditional, clamped loads first: a load inside a
    //      conditional is followed by s_waitcnt vmcnt(0), i.e. one full memory latency PER element
    uint32_t key[KPL];
    uint32_t nv = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
    const float* __restrict__ xc = x + col * col_stride;  // wave-uniform base, 32-bit lane offsets
    const uint32_t Tm1 = (uint32_t)T - 1u;
    // `g` is made opaque per column: otherwise the KPL clamped offsets and KPL range masks are hoisted out of the column
    // loop and stay live across it (~70 registers -> spills or one workgroup per CU)
    uint32_t g = (uint32_t)gt;
    asm volatile("" : "+v"(g));
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      const uint32_t i = g + (uint32_t)(k * NT);
      key[k] = __float_as_uint(xc[(k < KSAFE || i < Tm1) ? i : Tm1]);
    }
#pragma unroll
    for (int b = 0; b < BPT; ++b) cur[gt + b * NT] = 0;
    if (gt == 0) { s_slow = 0; s_off = 0; s_nslot = 0; cur[NB] = 0; }
    // The kernel is VALU-bound (PMC: the VALU is busy 86 % of the time with 4 workgroups per CU), so the per-key work is
    // counted in instructions.  Key: u ^ ((u >> 31) | 0x80000000) (3 ops); NaN (either sign) and the positions beyond
    // the series become 0xFFFFFFFF with one compare + one select; the valid and minimum counts are wave-uniform (popcount
    // of the compare mask on the scalar unit).
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      const uint32_t u = key[k];
      const uint32_t kk = u ^ ((uint32_t)((int32_t)u >> 31) | 0x80000000u);
      if (k < KSAFE) {
        // compare, count and select in one block on VCC: written in C++ the scheduler batches the KPL compares and their
        // lane masks overflow the scalar registers into VGPR lanes (v_writelane / v_readlane per key)
        uint32_t o, c;
        asm volatile("v_cmp_o_f32 vcc, %3, %3\n\ts_bcnt1_i32_b64 %1, vcc\n\tv_cndmask_b32 %0, -1, %2, vcc"
                     : "=v"(o), "=s"(c)
                     : "v"(kk), "v"(u)
                     : "vcc", "scc");
        key[k] = o;
        nv += c;  // (wave-uniform)
      } else {
        const uint32_t i = g + (uint32_t)(k * NT);
        const float f = __uint_as_float(u);
        const bool valid = i <= Tm1 && (f == f);
        key[k] = valid ? kk : 0xFFFFFFFFu;
        nv += (uint32_t)__popcll(__ballot(valid));
      }
    }
    // NaN key = 0xFFFFFFFF never wins a min; kmax is tracked as (key + 1) so that the NaN key wraps to 0
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      uint32_t kk = key[k];
      kmin = kk < kmin ? kk : kmin;
      kmax = kk + 1u > kmax ? kk + 1u : kmax;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      uint32_t a = __shfl_xor(kmin, off, 64), b = __shfl_xor(kmax, off, 64);
      kmin = a < kmin ? a : kmin;
      kmax = b > kmax ? b : kmax;
    }
    kmax -= 1u;  // (only meaningful when nv > 0)
    // The smallest key gets bin 0 to itself and bins 1 .. NB-1 divide [kmin2, kmax], kmin2 = the smallest key above it:
    // a precipitation series is mostly exact zeros followed by a gap of most of the key range (0 -> the smallest wet
    // amount); with [kmin, kmax] divided evenly every wet day lands in a quarter of the bins.  The copies of kmin are
    // counted in registers: they skip the histogram atomics (thousands of lanes on one address) and are never collected.
    // Wave level first (shuffles only): wave minimum, smallest key above it, copies of it; one LDS round joins the waves.
    // (smallest key above kmin = kmin + 1 + min over keys of (key - kmin - 1): the copies of kmin wrap to 0xFFFFFFFF and
    //  never win — a subtract and a min per key instead of two compares and a select)
    uint32_t wmin2 = 0xFFFFFFFFu, wcnt = 0;
    {
      const uint32_t kmin1 = kmin + 1u;
      uint32_t m2 = 0xFFFFFFFFu;
#pragma unroll
      for (int k = 0; k < KPL; ++k) {
        const uint32_t d1 = key[k] - kmin1;
        m2 = d1 < m2 ? d1 : m2;
        uint32_t c;
        asm volatile("v_cmp_eq_u32 vcc, %1, %2\n\ts_bcnt1_i32_b64 %0, vcc" : "=s"(c) : "v"(key[k]), "v"(kmin) : "vcc", "scc");
        wcnt += c;  // (wave-uniform)
      }
      // NaN keys: 0xFFFFFFFF - kmin1 stays below 0xFFFFFFFF unless kmin1 == 0 (an all-NaN wave); keep the old sentinel
      wmin2 = (m2 == 0xFFFFFFFFu || kmin1 + m2 == 0xFFFFFFFFu) ? 0xFFFFFFFFu : kmin1 + m2;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t a = __shfl_xor(wmin2, off, 64);
      wmin2 = a < wmin2 ? a : wmin2;
    }
    if (lane == 0) { red[w] = nv; red[NW + w] = kmin; red[2 * NW + w] = kmax; red[3 * NW + w] = wmin2; red[4 * NW + w] = wcnt; }
    lds_barrier();
    XH_PHASE(0);
    uint32_t n = 0;
    kmin = 0xFFFFFFFFu; kmax = 0u;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      n += red[i];
      kmin = red[NW + i] < kmin ? red[NW + i] : kmin;
      kmax = red[2 * NW + i] > kmax ? red[2 * NW + i] : kmax;
    }
    uint32_t kmin2 = 0xFFFFFFFFu, cnt0 = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      const bool own = red[NW + i] == kmin;                       // this wave holds copies of the minimum
      const uint32_t c2 = own ? red[3 * NW + i] : red[NW + i];    // its smallest key above kmin
      kmin2 = c2 < kmin2 ? c2 : kmin2;
      cnt0 += own ? red[4 * NW + i] : 0u;
    }
    const bool alleq = kmin2 == 0xFFFFFFFFu;      // all valid keys equal (or none valid)
    kmin2 = alleq ? kmin + 1u : kmin2;            // (kmin + 1: the copies of kmin then wrap to the dummy bin like always)
    cnt0 = n > 0 ? cnt0 : 0u;                     // (all NaN: kmin is the NaN key)
    const uint32_t range = (n > 0 && !alleq) ? kmax - kmin2 : 0u;
    // bins 1 .. NB-1 divide [kmin2, kmax] EVENLY: bin = 1 + floor((key - kmin2) * (NB - 1) / (range + 1)) as one
    // v_mul_hi_u32 with a 32.32 fixed-point scale (monotone in the key, which is all the selection needs).  A power-of-two
    // shift used between half and all of the bins (1009 of 2048 on the benchmark series): twice the keys per target
    // bin, and the in-bin selection is quadratic in that.  Fewer distinct keys than bins: one key value per bin.
    // (fewer distinct key values than bins: the scale saturates at 1 - 2^-32, d -> d - 1: still monotone, still < NB - 1)
    const uint32_t bscale = range < (uint32_t)(NB - 1) ? 0xFFFFFFFFu : (uint32_t)((((uint64_t)(NB - 1)) << 32) / ((uint64_t)range + 1ull));
    // bin of a key: NB (dummy, no tag, no atomics) for NaN and for the copies of kmin — WITHOUT testing for them: a
    // genuine key has d <= range and lands in 1 .. NB-1; the NaN key and the copies of kmin (d wraps) are at least
    // 0x7FFFFF beyond the range (valid keys end at 0xFF800000), which puts the scaled value at NB-1 or more
    auto binof = [&](uint32_t kk) -> uint32_t {
      const uint32_t d = kk - kmin2;
      const uint32_t x = __umulhi(d, bscale);
      return 1u + (x < (uint32_t)(NB - 1) ? x : (uint32_t)(NB - 1));
    };
    // ---- B: target ranks + histogram
    if (gt < ntgt) {
      int r = -1;
      if (n >= 1) {
        if (T == 1 || n < 2) r = 0;
        else {
          double nn = (double)n;
          double vi = nn * tq + (1.0 + tq * (1.0 - 1.0 - 1.0)) - 1.0;  // utl:395 with alpha = beta = 1
          if (vi >= nn - 1.0) r = (int)n - 1;
          else if (vi < 0.0) r = 0;
          else r = (int)floor(vi) + (gt & 1);
        }
      }
      trank[gt] = r;
      tinfo[gt].bin = -1;
    }
    if (!(abl & 1)) {
#pragma unroll
      for (int k = 0; k < KPL; ++k) {
        const uint32_t b = binof(key[k]);
        if (b != (uint32_t)NB) atomicAdd(&cur[b], 1u);
      }
    }
    if (gt == 0) cur[0] = cnt0;  // nobody else touches bin 0
    lds_barrier();
    XH_PHASE(1);
    // ---- C: exclusive scan (thread gt owns bins gt*BPT ... gt*BPT+BPT-1), targets, list regions
    uint32_t loc[BPT], s = 0;
#pragma unroll
    for (int b = 0; b < BPT; ++b) { loc[b] = cur[gt * BPT + b]; s += loc[b]; }
    uint32_t incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      uint32_t o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) red[3 * NW + w] = incl;
    texcl[gt] = incl - s;  // keys in the bins before mine INSIDE this wave
    const int myr0 = lane < ntgt ? trank[lane] : -1, myr1 = lane + 64 < ntgt ? trank[lane + 64] : -1;
    lds_barrier();
    XH_PHASE(2);
    // Targets: every lane of wave 0 resolves the (up to two) target ranks it holds IN PARALLEL — owner wave from the
    // eight wave totals, owner thread by a 6-step binary search over that wave's exclusive prefixes, bin from the owner's
    // BPT counts (one LDS read) — where round 1 walked the targets of each wave one after the other on scalars (~2 K
    // cycles per target: ballot / readlane / LDS-atomic round trips in a serial loop, 20 % of the kernel).
    // A bin hit by several targets is allocated once, by its LEADER (the lowest target index with that bin); the ranks
    // are monotone only up to lo_{j+1} < hi_j, so equal bins need not be neighbours: the leader is found by a uniform
    // loop over the targets (readlane + compare).
    if (w == 0 && !(abl & 8)) {
      uint32_t wp[NW + 1];
      wp[0] = 0;
#pragma unroll
      for (int i = 0; i < NW; ++i) wp[i + 1] = wp[i] + red[3 * NW + i];
      int tbin[2], tm[2], tkth[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int r = h ? myr1 : myr0;
        tbin[h] = -1; tm[h] = 0; tkth[h] = 0;
        if (r >= 0) {
          int wi = 0;
          uint32_t base = 0;
#pragma unroll
          for (int i = 1; i < NW; ++i) {
            const bool in = wp[i] <= (uint32_t)r;  // the LAST wave whose first key index is <= r holds rank r
            wi = in ? i : wi;
            base = in ? wp[i] : base;
          }
          int lo = 0, hi = 63;
#pragma unroll
          for (int it = 0; it < 6; ++it) {
            const int mid = (lo + hi + 1) >> 1;
            const bool le = base + texcl[wi * 64 + mid] <= (uint32_t)r;
            lo = le ? mid : lo;
            hi = le ? hi : mid - 1;
          }
          const int L = wi * 64 + lo;
          uint32_t st0 = base + texcl[L];
          bool found = false;
#pragma unroll
          for (int b = 0; b < BPT; ++b) {
            const uint32_t lb = cur[L * BPT + b];
            if (!found && (uint32_t)r < st0 + lb) {
              found = true;
              tbin[h] = L * BPT + b; tm[h] = (int)lb; tkth[h] = (int)((uint32_t)r - st0);
            }
            st0 += lb;
          }
        }
      }
      // leaders: lowest target index with the same bin
      int lead[2] = {lane, lane + 64};
      for (int j = ntgt - 1; j >= 0; --j) {  // descending: the last assignment that sticks is the lowest index
        const int bj = j < 64 ? __builtin_amdgcn_readlane(tbin[0], j) : __builtin_amdgcn_readlane(tbin[1], j - 64);
        lead[0] = (bj == tbin[0]) ? j : lead[0];
        lead[1] = (bj == tbin[1]) ? j : lead[1];
      }
      uint32_t tag[2] = {0u, 0u};
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int t = lane + 64 * h;
        if (tbin[h] >= 0 && lead[h] == t) {
          const int bin = tbin[h], m = tm[h];
          const bool floor_bin = bin == 0;  // the copies of kmin: a constant bin whose keys are never collected
          int off = LISTCAP + 1;
          if (m <= BIGM && !floor_bin) off = atomicAdd(&s_off, m);
          if (m <= BIGM && !floor_bin && off + m <= LISTCAP) tag[h] = F_LIST | (uint32_t)off;
          else {
            const int slot = atomicAdd(&s_nslot, 1);
            bmin[slot] = floor_bin ? kmin : 0xFFFFFFFFu;
            bmax[slot] = floor_bin ? kmin : 0u;
            tag[h] = F_BIG | (uint32_t)slot;
          }
          cur[bin] = tag[h];  // append cursor / min-max slot of the bin (the count left in cur[] is not needed any more)
        }
      }
      // followers take the leader's tag
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int ld = lead[h];
        const uint32_t t0 = (uint32_t)__shfl((int)tag[0], ld & 63, 64), t1 = (uint32_t)__shfl((int)tag[1], ld & 63, 64);
        const uint32_t tg = ld < 64 ? t0 : t1;
        const int t = lane + 64 * h;
        if (tbin[h] >= 0 && t < ntgt) {
          TInfo ti;
          ti.bin = tbin[h]; ti.kth = tkth[h]; ti.m = tm[h];
          ti.region = (tg & F_LIST) ? (int)(tg & F_MASK) : -((int)(tg & F_MASK) + 1);
          tinfo[t] = ti;
        }
      }
    }
    lds_barrier();
    XH_PHASE(3);
    // ---- D: collect the keys of the target bins (chunks of CH keys: tags, then the append atomics, then the list
    //      writes — independent LDS operations in flight instead of one dependent chain per key)
    if (!(abl & 2)) {
      constexpr int CH = KPL % 8 == 0 ? 8 : 4;
#pragma unroll
      for (int k0 = 0; k0 < KPL; k0 += CH) {
        uint32_t cf[CH], pos[CH], bidx[CH];
        uint32_t bigor = 0;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          bidx[k] = binof(key[k0 + k]);
          cf[k] = cur[bidx[k]];  // cur[NB] (NaN keys, copies of kmin) carries no tag
          bigor |= cf[k];
        }
        const bool anybig = (bigor & F_BIG) != 0u;
        // branch-free: a lane whose key is not in a listed bin appends to a private dummy word (texcl[gt], free until
        // the next column's scan) — every atomic of the chunk is in flight before the first result is needed; with a
        // branch per key the compiler serialised one LDS round trip per key
        uint32_t* ap[CH];
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const bool lst = (cf[k] & F_LIST) != 0u;
          ap[k] = lst ? &cur[bidx[k]] : &texcl[gt];
        }
#pragma unroll
        for (int k = 0; k < CH; ++k) pos[k] = atomicAdd(ap[k], 1u) & F_MASK;
#pragma unroll
        for (int k = 0; k < CH; ++k) {
          const bool lst = (cf[k] & F_LIST) != 0u;
          uint32_t* wp = lst ? &list[pos[k]] : &texcl[gt];
          *wp = key[k0 + k];
        }
        if (__any(anybig)) {
#pragma unroll
          for (int k = 0; k < CH; ++k)
            if (cf[k] & F_BIG) {
              const uint32_t slot = cf[k] & F_MASK;
              atomicMin(&bmin[slot], key[k0 + k]);
              atomicMax(&bmax[slot], key[k0 + k]);
            }
        }
      }
    }
    lds_barrier();
    XH_PHASE(4);
    // ---- E: select inside the bins: (target, candidate) pairs are spread over the whole workgroup — a candidate key
    //      is the answer iff #(keys < e) <= kth < #(keys <= e); O(m) LDS reads per thread, not O(m^2) per target
    if (!(abl & 4)) {
      constexpr int CPT = NT >= 512 ? 8 : NT / 64;  // candidate lanes per target: 64 targets (2 x 32 quantiles) in ONE pass
      for (int t = gt / CPT; t < ntgt; t += NT / CPT) {
        const TInfo ti = tinfo[t];
        const int a0 = gt % CPT;
        if (ti.bin < 0) {  // no valid sample
          if (a0 == 0) vals[t] = xh_nan32();
          continue;
        }
        if (ti.region < 0) {
          if (a0 == 0) {
            const int slot = -ti.region - 1;
            float v = xh_nan32();
            if (bmin[slot] == bmax[slot]) v = xh_key2f(bmin[slot]);  // constant bin (e.g. all the dry days)
            else s_slow = 1;
            vals[t] = v;
          }
          continue;
        }
        // two candidates per lane and pass, four list keys per step: independent LDS reads in flight (a one-key-per-
        // iteration loop waits one LDS latency per key).  Reads past the bin's m keys are masked (list is padded).
        const uint32_t m = (uint32_t)ti.m, kth = (uint32_t)ti.kth;
        const uint32_t* lp = list + ti.region;
        for (uint32_t a2 = a0; a2 < m; a2 += 2 * CPT) {
          const uint32_t e0 = lp[a2];
          uint32_t e1 = lp[a2 + CPT];
          e1 = a2 + CPT < m ? e1 : 0u;  // key 0 never wins (valid keys are > 0)
          uint32_t less0 = 0, leq0 = 0, less1 = 0, leq1 = 0;
          for (uint32_t b2 = 0; b2 < m; b2 += 4) {
            uint32_t k0 = lp[b2], k1 = lp[b2 + 1], k2 = lp[b2 + 2], k3 = lp[b2 + 3];
            k1 = b2 + 1 < m ? k1 : 0xFFFFFFFFu;
            k2 = b2 + 2 < m ? k2 : 0xFFFFFFFFu;
            k3 = b2 + 3 < m ? k3 : 0xFFFFFFFFu;
            less0 += (k0 < e0 ? 1u : 0u) + (k1 < e0 ? 1u : 0u) + (k2 < e0 ? 1u : 0u) + (k3 < e0 ? 1u : 0u);
            leq0 += (k0 <= e0 ? 1u : 0u) + (k1 <= e0 ? 1u : 0u) + (k2 <= e0 ? 1u : 0u) + (k3 <= e0 ? 1u : 0u);
            less1 += (k0 < e1 ? 1u : 0u) + (k1 < e1 ? 1u : 0u) + (k2 < e1 ? 1u : 0u) + (k3 < e1 ? 1u : 0u);
            leq1 += (k0 <= e1 ? 1u : 0u) + (k1 <= e1 ? 1u : 0u) + (k2 <= e1 ? 1u : 0u) + (k3 <= e1 ? 1u : 0u);
          }
          if (less0 <= kth && kth < leq0) vals[t] = xh_key2f(e0);  // every winner writes the same value
          if (e1 != 0u && less1 <= kth && kth < leq1) vals[t] = xh_key2f(e1);
        }
      }
    }
    lds_barrier();
    XH_PHASE(5);
    // ---- rare: big non-constant target bins -> bisection on the key value with counting sweeps
    if (s_slow) {
      for (int t = 0; t < ntgt; ++t) {
        const TInfo ti = tinfo[t];
        if (ti.bin < 0 || ti.region >= 0) continue;
        const int slot = -ti.region - 1;
        uint32_t lo = bmin[slot], hi = bmax[slot];
        if (lo == hi) continue;
        // smallest K in [lo, hi] with #(key <= K, key in this bin) >= kth + 1
        const uint32_t binlo = lo;  // the smallest key of the bin: every key of a lower bin is below it
        while (lo < hi) {
          const uint32_t mid = lo + ((hi - lo) >> 1);
          uint32_t c = 0;
#pragma unroll
          for (int k = 0; k < KPL; ++k) c += (key[k] != 0xFFFFFFFFu && key[k] >= binlo && key[k] <= mid) ? 1u : 0u;
          c = block_sum<NT>(c, red);
          if (c >= (uint32_t)ti.kth + 1u) hi = mid; else lo = mid + 1;
        }
        if (gt == 0) vals[t] = xh_key2f(lo);
        lds_barrier();
      }
    }
    XH_PHASE(6);
    // ---- F: Hyndman-Fan lerp (type 7) and store
    if (gt < nq) {
      const int j = gt;
      double r;
      if (n == 0) r = xh_nan64();
      else if (T == 1 || n < 2) r = (double)vals[2 * j];
      else {
        double nn = (double)n;
        double vi = nn * oq + (1.0 + oq * (1.0 - 1.0 - 1.0)) - 1.0;
        float left = vals[2 * j], right = vals[2 * j + 1];
        if (vi >= nn - 1.0 || vi < 0.0) r = (double)left;
        else {
          double gamma = vi - floor(vi);
          float diff = right - left;
          r = (double)left + (double)diff * gamma;
          if (gamma >= 0.5) r = (double)right - (double)diff * (1.0 - gamma);
        }
      }
      if (!(abl & 16)) out[col * out_cstride + (int64_t)j * out_qstride] = (float)r;
      else if (r == 12345.678) out[0] = (float)r;
    }
    lds_barrier();  // vals / tinfo / red are rewritten by the next column
    XH_PHASE(7);
  }
#undef XH_PHASE
}

template <int NT, int KPL, int NB, int KSAFE = 0>
int launch_lean(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride, const double* d_q, int nq,
                float* out, int64_t out_cstride, int64_t out_qstride) {
  int64_t nblk = ncols;
  int64_t maxblk = (int64_t)ctx->num_cu * 16;
  if (nblk > maxblk) nblk = maxblk;
  const char* ea = xh_diag_env("XH_SELECT_ABL");  // diagnostics only: skip phases (results become wrong)
  const char* ep = xh_diag_env("XH_SELECT_PROF");  // diagnostics only: per-phase cycle counts on stderr
  unsigned long long* d_prof = nullptr;
  if (ep && atoi(ep)) {
    XH_CHECK_HIP(hipMalloc((void**)&d_prof, 16 * sizeof(unsigned long long)));
    XH_CHECK_HIP(hipMemsetAsync(d_prof, 0, 16 * sizeof(unsigned long long), ctx->stream));
  }
  if ((ea && atoi(ea)) || d_prof)
    hipLaunchKernelGGL((k_select_lean<NT, KPL, NB, KSAFE, true>), dim3((unsigned)nblk), dim3(NT), 0, ctx->stream, xcols, T, ncols,
                       col_stride, d_q, nq, out, out_cstride, out_qstride, ea ? atoi(ea) : 0, d_prof);
  else
    hipLaunchKernelGGL((k_select_lean<NT, KPL, NB, KSAFE, false>), dim3((unsigned)nblk), dim3(NT), 0, ctx->stream, xcols, T, ncols,
                       col_stride, d_q, nq, out, out_cstride, out_qstride, 0, nullptr);
  XH_LAUNCH_CHECK();
  if (d_prof) {
    unsigned long long h[16];
    XH_CHECK_HIP(hipMemcpyAsync(h, d_prof, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
    XH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    XH_CHECK_HIP(hipFree(d_prof));
    static const char* names[10] = {"A load+reduce", "B ranks+hist", "C scan", "C targets", "D collect", "E in-bin select",
                                    "slow path", "F lerp+store", "-", "-"};
    unsigned long long tot = 0;
    for (int i = 0; i < 10; ++i) tot += h[i];
    fprintf(stderr, "k_select_lean<%d,%d,%d> phases (cycles per column per workgroup, %lld columns):\n", NT, KPL, NB, (long long)ncols);
    for (int i = 0; i < 10; ++i)
      fprintf(stderr, "  %-14s %10.0f  %5.1f %%\n", names[i], (double)h[i] / (double)ncols, 100.0 * (double)h[i] / (double)(tot ? tot : 1));
  }
  return XH_OK;
}

}  // namespace

// T in (1024, 16384]: keys per thread rounded up to a multiple of 4 to avoid idle register slots
int xh_select_columns_lean(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride,
                           const double* d_q, int nq, float* out, int64_t out_cstride, int64_t out_qstride) {
  if (T <= 1024 || T > 16384 || nq > 64) return XH_ERR_NOTIMPL;
#define XH_LEAN(NT, KPL, NB) return launch_lean<NT, KPL, NB>(ctx, xcols, T, ncols, col_stride, d_q, nq, out, out_cstride, out_qstride)
#define XH_LEANS(NT, KPL, NB, TLOW) \
  return launch_lean<NT, KPL, NB, (TLOW) / (NT)>(ctx, xcols, T, ncols, col_stride, d_q, nq, out, out_cstride, out_qstride)
  // threads per column above 4096 steps: 256 (four workgroups per CU).  With the parallel target search the config-4
  // train (T = 10950, time-major pipeline) takes 85.4 ms with 256 threads and 93.2 ms with 512; T = 7300: 59.1 vs 69.0 ms
  // (with the serial per-wave target loop of round 1 it was 98.5 vs 92.7: four waves walked ten targets each).
  const char* ent = xh_diag_env("XH_LEAN_NT");  // tuning only
  const int nt = ent ? atoi(ent) : 256;
  if (T <= 2048) XH_LEANS(256, 8, 1024, 1024);
  if (T <= 3072) XH_LEANS(256, 12, 1024, 2048);
  if (T <= 4096) XH_LEANS(256, 16, 1024, 3072);
  if (nt == 256) {
    if (T <= 6144) XH_LEANS(256, 24, 2048, 4096);
    if (T <= 8192) XH_LEANS(256, 32, 2048, 6144);
    if (T <= 10240) XH_LEANS(256, 40, 2048, 8192);
    if (T <= 11264) XH_LEANS(256, 44, 2048, 10240);  // 30 years of days: 10950 .. 10958 keys, 3 % idle slots instead of 11 %
    if (T <= 12288) XH_LEANS(256, 48, 2048, 11264);
    if (T <= 14336) XH_LEANS(256, 56, 2048, 12288);
    XH_LEANS(256, 64, 2048, 14336);
  }
  if (nt == 1024) {
    if (T <= 8192) XH_LEAN(1024, 8, 2048);
    if (T <= 12288) XH_LEAN(1024, 12, 2048);
    XH_LEAN(1024, 16, 2048);
  }
  if (T <= 6144) XH_LEAN(512, 12, 2048);
  if (T <= 8192) XH_LEAN(512, 16, 2048);
  if (T <= 10240) XH_LEAN(512, 20, 2048);
  if (T <= 12288) XH_LEAN(512, 24, 2048);
  if (T <= 14336) XH_LEAN(512, 28, 2048);
  XH_LEAN(512, 32, 2048);
#undef XH_LEAN
#undef XH_LEANS
}
