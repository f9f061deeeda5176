// qdm.hip — QuantileDeltaMapping.adjust (xsdba._adjustment.qdm_adjust), group = "time".
//
//   sim_q = rank(sim, dim="time", pct=True)                    xsdba.utils.rank: average ranks of the valid samples,
//                                                              r / n rescaled as mx * (rnk - mn) / (mx - mn) -> [0, mx]
//   af_t  = interp_on_quantiles(sim_q, quantiles, af)          scipy interp1d over the non-NaN nodes, x = the quantile
//                                                              nodes (the same for every cell), nearest | linear,
//                                                              fill (af[0], af[-1]) | NaN outside the nodes
//   scen  = sim + af_t | sim * af_t                            apply_correction
// xsdba is not in the reference tree (src/xclim/sdba.py:10 re-exports it): PARITY UNPINNED, the oracle
// (oracle/sdba.py: qdm_adjust) is a restatement of the same published algorithm with scipy.stats.rankdata.
//
// Kernel: one workgroup per COLUMN (time-minor view; time-major inputs go through the transposed-batch scratch of the
// quantile kernels, both ways).  The column's keys stay in registers; an exact average rank needs, for every key, the
// number of keys below it and equal to it:
//   A  load, order-preserving keys, n / kmin / kmax, copies of kmin and of kmax, smallest key above kmin
//   B  histogram of NB bins linear in key space over [kmin2, kmax] (LDS atomics); the copies of the minimum (the dry
//      days of a precipitation series) are counted in registers and never enter a bin
//   C  exclusive scan -> list offsets;  D  counting-sort scatter of the keys into an LDS list ordered by bin
//   E  every key scans ITS bin (a handful of keys): less / equal counts -> doubled rank r2 = 2 (below) + equal + 1,
//      pct in fp64 exactly as numpy would compute it, node search + interpolation in fp64, correction, store
// A bin that is heavy and not constant costs O(m) per key (O(m^2) per bin): correct, slow; not met on daily series.
#include <stdlib.h>

#include "common.h"

namespace {

__device__ __forceinline__ void qdm_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

constexpr int QDM_MAXQ = 64;

template <int NT, int KPL, int NB>
__global__ void __launch_bounds__(NT)
k_qdm_columns(const float* __restrict__ x, int64_t T, int64_t ncols, int64_t col_stride, const float* __restrict__ af,
              int64_t af_qstride, const double* __restrict__ qnodes, int nq, int kind, int interp, int extrap,
              float* __restrict__ out, int64_t out_cstride) {
  constexpr int BPT = (NB + NT - 1) / NT;
  constexpr int NW = NT / 64;
  extern __shared__ double qdm_lds[];       // (double: 8-byte aligned base)
  double* xs = qdm_lds;                     // compacted quantile nodes of the column's non-NaN factors
  double* xb = xs + QDM_MAXQ;               // scipy's nearest bounds x[j]/2 + x[j+1]/2
  float* ys = reinterpret_cast<float*>(xb + QDM_MAXQ);
  uint32_t* list = reinterpret_cast<uint32_t*>(ys + QDM_MAXQ);  // KPL * NT keys ordered by bin
  uint32_t* cur = list + KPL * NT;          // NB + 1 counters / cursors ([NB]: dummy)
  uint32_t* start = cur + NB + 1;           // NB list offsets
  uint32_t* red = start + NB;               // 6 * NW + 8
  __shared__ int s_nv;
  const int gt = threadIdx.x, lane = gt & 63, w = gt >> 6;
  const uint32_t Tm1 = (uint32_t)T - 1u;

  for (int64_t col = blockIdx.x; col < ncols; col += gridDim.x) {
    const float* __restrict__ xc = x + col * col_stride;
    float raw[KPL];
    uint32_t key[KPL];
    uint32_t g = (uint32_t)gt;
    asm volatile("" : "+v"(g));  // keeps the KPL clamped offsets from being hoisted out of the column loop
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      const uint32_t i = g + (uint32_t)(k * NT);
      raw[k] = xc[i < Tm1 ? i : Tm1];
    }
    // nodes of this column: drop the NaN factors (interp_on_quantiles masks them), first wave only
    if (w == 0) {
      const bool have = lane < nq;
      const float a = have ? af[(int64_t)lane * af_qstride + col] : 0.f;
      const bool ok = have && (a == a);
      const unsigned long long m = __ballot(ok);
      const int pos = __popcll(m & ((1ull << lane) - 1ull));
      if (ok) { xs[pos] = qnodes[lane]; ys[pos] = a; }
      if (lane == 0) s_nv = __popcll(m);
    }
#pragma unroll
    for (int b = 0; b < BPT; ++b)
      if (gt + b * NT <= NB) cur[gt + b * NT] = 0;
    uint32_t nv = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      const uint32_t i = g + (uint32_t)(k * NT);
      const uint32_t kk = xh_f2key(raw[k] + 0.0f);  // -0.0 + 0.0 = +0.0: the two zeros tie, as they do in rankdata
      key[k] = (i <= Tm1) ? kk : 0xFFFFFFFFu;
      nv += key[k] != 0xFFFFFFFFu ? 1u : 0u;
      kmin = key[k] < kmin ? key[k] : kmin;
      kmax = key[k] + 1u > kmax ? key[k] + 1u : kmax;  // (key + 1: the NaN key wraps to 0 and never wins)
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      nv += __shfl_xor(nv, off, 64);
      const uint32_t a = __shfl_xor(kmin, off, 64), b = __shfl_xor(kmax, off, 64);
      kmin = a < kmin ? a : kmin;
      kmax = b > kmax ? b : kmax;
    }
    if (lane == 0) { red[w] = nv; red[NW + w] = kmin; red[2 * NW + w] = kmax; }
    qdm_barrier();
    uint32_t n = 0;
    kmin = 0xFFFFFFFFu; kmax = 0u;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      n += red[i];
      kmin = red[NW + i] < kmin ? red[NW + i] : kmin;
      kmax = red[2 * NW + i] > kmax ? red[2 * NW + i] : kmax;
    }
    kmax -= 1u;
    // second round: copies of the extremes, smallest key above the minimum
    uint32_t c0 = 0, cm = 0, k2 = 0xFFFFFFFFu;
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      c0 += key[k] == kmin ? 1u : 0u;
      cm += key[k] == kmax ? 1u : 0u;
      k2 = (key[k] > kmin && key[k] < k2) ? key[k] : k2;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      c0 += __shfl_xor(c0, off, 64);
      cm += __shfl_xor(cm, off, 64);
      const uint32_t a = __shfl_xor(k2, off, 64);
      k2 = a < k2 ? a : k2;
    }
    if (lane == 0) { red[3 * NW + w] = c0; red[4 * NW + w] = cm; red[5 * NW + w] = k2; }
    qdm_barrier();
    uint32_t cnt0 = 0, cntm = 0, kmin2 = 0xFFFFFFFFu;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      cnt0 += red[3 * NW + i];
      cntm += red[4 * NW + i];
      kmin2 = red[5 * NW + i] < kmin2 ? red[5 * NW + i] : kmin2;
    }
    if (n == 0) { cnt0 = 0; cntm = 0; }
    kmin2 = kmin2 == 0xFFFFFFFFu ? kmin : kmin2;  // all valid keys equal
    const uint32_t range = n > 0 ? kmax - kmin2 : 0u;
    int shift = 32 - __clz((int)range) - (31 - __clz(NB));
    shift = (range == 0u || shift < 0) ? 0 : shift;
    const XhValueBins vb = xh_value_bins(kmin2, kmax, n > 0, NB);  // (keys on both sides of zero: bins linear in the value)
    auto binof = [&](uint32_t kk) -> uint32_t {  // NB: dummy bin (NaN keys and the copies of the minimum)
      const uint32_t b = vb.on ? xh_value_bin(vb, kk) : ((kk - kmin2) >> shift);
      return (kk == 0xFFFFFFFFu || kk == kmin) ? (uint32_t)NB : (b < (uint32_t)NB ? b : (uint32_t)NB - 1u);
    };
    // ---- B: histogram
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      const uint32_t b = binof(key[k]);
      if (b != (uint32_t)NB) atomicAdd(&cur[b], 1u);
    }
    qdm_barrier();
    // ---- C: exclusive scan (thread gt owns bins gt * BPT ...)
    uint32_t loc[BPT], s = 0;
#pragma unroll
    for (int b = 0; b < BPT; ++b) {
      const int bi = gt * BPT + b;
      loc[b] = bi < NB ? cur[bi] : 0u;
      s += loc[b];
    }
    uint32_t incl = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const uint32_t o = __shfl_up(incl, off, 64);
      if (lane >= off) incl += o;
    }
    if (lane == 63) red[w] = incl;
    qdm_barrier();
    {
      uint32_t add = 0;
#pragma unroll
      for (int i = 0; i < NW; ++i) add += (i < w) ? red[i] : 0u;
      uint32_t first = incl - s + add;
#pragma unroll
      for (int b = 0; b < BPT; ++b) {
        const int bi = gt * BPT + b;
        if (bi < NB) { start[bi] = first; cur[bi] = first; }
        first += loc[b];
      }
    }
    // scipy's bounds for kind="nearest" (after the compaction above is visible)
    const int nvn = s_nv;
    if (gt + 1 < nvn) xb[gt] = xs[gt] / 2.0 + xs[gt + 1] / 2.0;
    qdm_barrier();
    // ---- D: scatter by bin
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      const uint32_t b = binof(key[k]);
      if (b != (uint32_t)NB) {
        const uint32_t pos = atomicAdd(&cur[b], 1u);
        list[pos] = key[k];
      }
    }
    qdm_barrier();
    // ---- E: ranks, pct, factors, correction
    const double dn = (double)n;
    const double mn = ((double)(cnt0 + 1u) / 2.0) / dn;                // rank of the minimum / count
    const double mx = ((double)(2u * n - cntm + 1u) / 2.0) / dn;       // rank of the maximum / count
    const double mxmn = mx - mn;
    // the two divisions of every key (rank / count, then the rescaling) as exact quotients from reciprocals computed
    // once per column (xh_div_int: Markstein's correction, bit-identical to the IEEE division)
    const double inv_dn = 1.0 / dn, inv_mxmn = 1.0 / mxmn;
    float* __restrict__ oc = out + col * out_cstride;
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      const uint32_t i = g + (uint32_t)(k * NT);
      const uint32_t kk = key[k];
      float res = xh_nan32();
      if (kk != 0xFFFFFFFFu && nvn >= 2) {
        uint32_t below = 0, equal = cnt0;
        if (kk != kmin) {
          const uint32_t b = binof(kk);
          const uint32_t s0 = start[b], s1 = cur[b];
          uint32_t less = 0, eq = 0;
          for (uint32_t j = s0; j < s1; ++j) {
            const uint32_t o = list[j];
            less += o < kk ? 1u : 0u;
            eq += o == kk ? 1u : 0u;
          }
          below = cnt0 + s0 + less;
          equal = eq;
        }
        const double rnk = xh_div_int((double)(2u * below + equal + 1u) * 0.5, dn, inv_dn);
        const double pnum = mx * (rnk - mn);
        const double pct = mxmn == 0.0 ? xh_nan64() : xh_div_int(pnum, mxmn, inv_mxmn);  // 0 / 0 = NaN: all valid samples equal
        if (pct == pct) {
          const double x0 = xs[0], xl = xs[nvn - 1];
          float a;
          if (pct < x0) a = extrap == 0 ? ys[0] : xh_nan32();
          else if (pct > xl) a = extrap == 0 ? ys[nvn - 1] : xh_nan32();
          else if (interp == 0) {  // searchsorted(x_bds, pct, side="left"), clipped
            int lo = 0, hi = nvn - 1;
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (xb[mid] < pct) lo = mid + 1; else hi = mid;
            }
            a = ys[lo];
          } else {  // searchsorted(x, pct, side="left") clipped to [1, nv - 1]
            int lo = 0, hi = nvn;
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (xs[mid] < pct) lo = mid + 1; else hi = mid;
            }
            lo = lo < 1 ? 1 : (lo > nvn - 1 ? nvn - 1 : lo);
            const float ylo = ys[lo - 1], yhi = ys[lo];
            const double slope = (double)(yhi - ylo) / (xs[lo] - xs[lo - 1]);
            a = (float)(slope * (pct - xs[lo - 1]) + (double)ylo);
          }
          res = kind == 0 ? raw[k] + a : (kind == 1 ? raw[k] * a : a);
        }
      }
      if (i <= Tm1) oc[i] = res;
    }
    qdm_barrier();  // list / cur / nodes are rewritten by the next column
  }
}

template <int NT, int KPL, int NB>
int launch_qdm(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride, const float* af, int64_t af_qs,
               const double* d_q, int nq, int kind, int interp, int extrap, float* out, int64_t out_cs) {
  constexpr int NW = NT / 64;
  const size_t lds = sizeof(uint32_t) * ((size_t)KPL * NT + (NB + 1) + NB + (6 * NW + 8) + 1) + sizeof(double) * 2 * QDM_MAXQ +
                     sizeof(float) * QDM_MAXQ + 16;
  auto kern = k_qdm_columns<NT, KPL, NB>;
  if (lds > 48 * 1024) XH_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int64_t nblk = ncols;
  const int64_t maxblk = (int64_t)ctx->num_cu * (NT <= 64 ? 64 : 16);
  if (nblk > maxblk) nblk = maxblk;
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NT), lds, ctx->stream, xcols, T, ncols, col_stride, af, af_qs, d_q, nq, kind,
                     interp, extrap, out, out_cs);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

// ---- small groups: QuantileDeltaMapping.adjust with a day-of-year grouping ranks every step among the steps of its OWN group —
// one per year, 30 rows for a 30-year series — and 365 groups meant 365 gathers + 365 launches of the column kernels on 30-row
// blocks (launch-bound: 80 ms for a 30-year 1440 x 90 band).  Here ONE launch: thread = one cell of one group (rows <= PER,
// listed by row number — nothing is gathered).  PER stops at 64: the kernel is VALU-bound (4 instructions per key pair + ~150 of
// fp64 per row: 11 ms for 30 rows per group) and quadratic in the group size — instances with 128 / 192 / 256 keys in registers
// were built and measured (round 6): 120 ms for 100 rows per group, 227 ms for 151, against 143 ms for the per-group path.  The group's keys sit in registers; every row is then taken in turn: its key
// against the PER register keys (below / equal counts -> the doubled average rank), pct and the node lookup in the arithmetic of
// stage E above (bit-identical), the result stored at the row's own place.  The nodes of the cell (the non-NaN factors,
// compacted) live in the thread's LDS column: values + their indices into the common quantile nodes.
template <int PER>
__global__ void __launch_bounds__(256)
k_qdm_groups(const float* __restrict__ x, int64_t C, int64_t st, const int32_t* __restrict__ rows, const int64_t* __restrict__ offs,
             const float* __restrict__ af, const double* __restrict__ qnodes, int nq, int kind, int interp, int extrap,
             float* __restrict__ out, int64_t ost) {
  extern __shared__ double qg_lds[];
  double* qn = qg_lds;                                         // [nq] the common nodes
  float* ys = reinterpret_cast<float*>(qn + QDM_MAXQ);         // [nq][256] the cell's valid factors
  unsigned char* xi = reinterpret_cast<unsigned char*>(ys + (size_t)nq * 256);  // [nq][256] their node numbers
  const int tid = threadIdx.x;
  if (tid < nq) qn[tid] = qnodes[tid];
  __syncthreads();
  const int64_t c = (int64_t)blockIdx.x * 256 + tid;
  if (c >= C) return;
  const int64_t g = blockIdx.y;
  const int64_t k0 = offs[g];
  const int m = (int)(offs[g + 1] - k0);                        // rows of the group (<= PER)
  int nvn = 0;
  for (int j = 0; j < nq; ++j) {
    const float a = af[((int64_t)g * nq + j) * C + c];
    if (a == a) { ys[nvn * 256 + tid] = a; xi[nvn * 256 + tid] = (unsigned char)j; ++nvn; }
  }
  uint32_t key[PER];
  uint32_t n = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int32_t row = rows[k0 + (j < m ? j : 0)];
    const float f = x[(int64_t)row * st + c];
    const uint32_t kk = j < m ? xh_f2key(f + 0.0f) : 0xFFFFFFFFu;   // (-0.0 + 0.0 = +0.0: the two zeros tie)
    key[j] = kk;
    n += kk != 0xFFFFFFFFu ? 1u : 0u;
    kmin = kk < kmin ? kk : kmin;
    kmax = kk + 1u > kmax ? kk + 1u : kmax;
  }
  kmax -= 1u;
  uint32_t cnt0 = 0, cntm = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    cnt0 += key[j] == kmin ? 1u : 0u;
    cntm += key[j] == kmax ? 1u : 0u;
  }
  if (n == 0) { cnt0 = 0; cntm = 0; }
  const double dn = (double)n;
  const double mn = ((double)(cnt0 + 1u) / 2.0) / dn;
  const double mx = ((double)(2u * n - cntm + 1u) / 2.0) / dn;
  const double mxmn = mx - mn;
  const double inv_dn = 1.0 / dn, inv_mxmn = 1.0 / mxmn;
  auto xs = [&](int j) -> double { return qn[xi[j * 256 + tid]]; };
  float fnext = m > 0 ? x[(int64_t)rows[k0] * st + c] : 0.f;
  for (int j = 0; j < m; ++j) {
    const int32_t row = rows[k0 + j];
    const float raw = fnext;
    if (j + 1 < m) fnext = x[(int64_t)rows[k0 + j + 1] * st + c];   // (re-read from the cache: the keys are indexed statically only)
    const uint32_t kk = xh_f2key(raw + 0.0f);
    float res = xh_nan32();
    if (kk != 0xFFFFFFFFu && nvn >= 2) {
      uint32_t below = 0, equal = 0;
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        below += key[i] < kk ? 1u : 0u;
        equal += key[i] == kk ? 1u : 0u;
      }
      const double rnk = xh_div_int((double)(2u * below + equal + 1u) * 0.5, dn, inv_dn);
      const double pnum = mx * (rnk - mn);
      const double pct = mxmn == 0.0 ? xh_nan64() : xh_div_int(pnum, mxmn, inv_mxmn);
      if (pct == pct) {
        const double x0 = xs(0), xl = xs(nvn - 1);
        float a;
        if (pct < x0) a = extrap == 0 ? ys[tid] : xh_nan32();
        else if (pct > xl) a = extrap == 0 ? ys[(nvn - 1) * 256 + tid] : xh_nan32();
        else if (interp == 0) {  // searchsorted(x_bds, pct, side="left"), clipped; x_bds[j] = x[j] / 2 + x[j + 1] / 2
          int lo = 0, hi = nvn - 1;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (xs(mid) / 2.0 + xs(mid + 1) / 2.0 < pct) lo = mid + 1; else hi = mid;
          }
          a = ys[lo * 256 + tid];
        } else {
          int lo = 0, hi = nvn;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (xs(mid) < pct) lo = mid + 1; else hi = mid;
          }
          lo = lo < 1 ? 1 : (lo > nvn - 1 ? nvn - 1 : lo);
          const float ylo = ys[(lo - 1) * 256 + tid], yhi = ys[lo * 256 + tid];
          const double slope = (double)(yhi - ylo) / (xs(lo) - xs(lo - 1));
          a = (float)(slope * (pct - xs(lo - 1)) + (double)ylo);
        }
        res = kind == 0 ? raw + a : (kind == 1 ? raw * a : a);
      }
    }
    out[(int64_t)row * ost + c] = res;
  }
}

}  // namespace

// exact-rank kernel on time-minor columns (column c at xcols + c * col_stride, factors af[j * af_qs + c]); also the
// fallback of qdm2.hip for columns with ties
int xh_qdm_columns(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride, const float* af, int64_t af_qs,
                   const double* d_q, int nq, int kind, int interp, int extrap, float* out, int64_t out_cs) {
#define XH_QDM(NT, KPL, NB) return launch_qdm<NT, KPL, NB>(ctx, xcols, T, ncols, col_stride, af, af_qs, d_q, nq, kind, interp, extrap, out, out_cs)
  if (T <= 512) XH_QDM(64, 8, 256);
  if (T <= 2048) XH_QDM(256, 8, 1024);
  if (T <= 4096) XH_QDM(256, 16, 1024);
  if (T <= 8192) XH_QDM(512, 16, 2048);
  if (T <= 12288) XH_QDM(512, 24, 2048);
  if (T <= 16384) XH_QDM(512, 32, 2048);
  XH_QDM(512, 64, 2048);  // up to 32768 steps (~90 years of daily data): 144 KB of LDS, one workgroup per CU
#undef XH_QDM
}

extern "C" {

int xh_qdm_adjust(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, int64_t sc, const float* af,
                  const double* q, int nq, int kind, int interp, int extrap, float* scen) {
  XH_REQUIRE(ctx && sim && af && q && scen, XH_ERR_ARG, "xh_qdm_adjust: NULL argument");
  XH_REQUIRE(T >= 1 && T < (1ll << 27) && C >= 0 && nq >= 1 && nq <= QDM_MAXQ, XH_ERR_ARG,
             "xh_qdm_adjust: bad shape (1 <= T < 2^27, 1 <= nq <= 64)");
  XH_REQUIRE(kind >= 0 && kind <= 2, XH_ERR_ARG, "xh_qdm_adjust: kind must be 0 (+), 1 (*) or 2 (the interpolated factor only)");
  XH_REQUIRE(interp == 0 || interp == 1, XH_ERR_NOTIMPL, "xh_qdm_adjust: interp must be 0 (nearest) or 1 (linear)");
  XH_REQUIRE(extrap == 0 || extrap == 1, XH_ERR_ARG, "xh_qdm_adjust: extrap must be 0 (constant) or 1 (nan)");
  for (int j = 1; j < nq; ++j)
    XH_REQUIRE(q[j] > q[j - 1], XH_ERR_ARG, "xh_qdm_adjust: the quantile nodes must be strictly increasing");
  if (C == 0) return XH_OK;
  size_t cur = 0;
  void* d_q = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, q, sizeof(double) * nq, &d_q);
  if (rc) return rc;
  // beyond the exact-rank kernel's LDS list: ranks through a global sort (qdm3.hip), in batches (diagnostics:
  // XH_QDM_FORCE_SORTED sends every length there — the differential test against the exact-rank kernel)
  const bool longs = T > 32768 || xh_diag_env("XH_QDM_FORCE_SORTED") != nullptr;
  if (st == 1 && sc >= T) {  // time-minor: columns in place, scen in the same layout
    if (!longs) return xh_qdm_columns(ctx, sim, T, C, sc, af, C, (const double*)d_q, nq, kind, interp, extrap, scen, sc);
    int64_t chunk = (1ll << 27) / T;
    chunk = chunk < 1 ? 1 : (chunk > C ? C : chunk);
    size_t wsb = 0;
    rc = xh_qdm_sorted_ws(T, chunk, &wsb);
    if (rc) return rc;
    void* ws = nullptr;
    rc = xh_big_scratch(ctx, wsb, &ws);
    if (rc) return rc;
    for (int64_t c0 = 0; c0 < C; c0 += chunk) {
      const int64_t nb = C - c0 < chunk ? C - c0 : chunk;
      rc = xh_qdm_sorted(ctx, sim + c0 * sc, T, nb, sc, af + c0, C, (const double*)d_q, nq, kind, interp, extrap, scen + c0 * sc, sc, ws);
      if (rc) return rc;
    }
    return XH_OK;
  }
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT, "xh_qdm_adjust: one of the strides must be 1 (st=%lld sc=%lld)", (long long)st,
             (long long)sc);
  if (interp == 0 && !longs) {  // one-year series, nearest: sort in registers + classify by cut values, in place (qdm2.hip)
    rc = xh_qdm_regsort(ctx, sim, T, C, st, af, C, (const double*)d_q, nq, kind, extrap, scen, st);
    if (rc != XH_ERR_NOTIMPL) return rc;
  }
  if (interp == 0 && !xh_diag_env("XH_QDM_FORCE_SORTED")) {
    // long series (1024 < T <= 65535), nearest: class boundaries as order statistics from the two-pass histogram selection,
    // then one streaming classification pass (select4.hip) — nothing transposed, nothing sorted globally
    rc = xh_qdm_hist(ctx, sim, T, C, st, af, C, (const double*)d_q, nq, kind, extrap, scen, st);
    if (rc != XH_ERR_NOTIMPL) return rc;
  }
  // time-major: batches of columns through a transposed scratch, both ways (padded pitch: 256-byte aligned segments)
  int64_t Tp = (T + 63) & ~(int64_t)63;
  int64_t batch = (int64_t)((1ull << 28) / (sizeof(float) * (size_t)Tp));
  batch = (batch / 128) * 128;
  if (batch < 128) batch = 128;
  if (batch > C) batch = C;
  void* tmp = nullptr;
  size_t wsb = 0;
  if (longs) {
    rc = xh_qdm_sorted_ws(T, batch, &wsb);
    if (rc) return rc;
  }
  const size_t bufb = 2 * sizeof(float) * (size_t)batch * (size_t)Tp;
  rc = xh_big_scratch(ctx, bufb + wsb, &tmp);
  if (rc) return rc;
  float* bin = (float*)tmp;
  float* bout = bin + (size_t)batch * (size_t)Tp;
  void* ws = (char*)tmp + bufb;
  for (int64_t c0 = 0; c0 < C; c0 += batch) {
    const int64_t nb = C - c0 < batch ? C - c0 : batch;
    rc = xh_transpose_f32(ctx, sim + c0, T, nb, st, bin, Tp);
    if (rc) return rc;
    rc = longs ? xh_qdm_sorted(ctx, bin, T, nb, Tp, af + c0, C, (const double*)d_q, nq, kind, interp, extrap, bout, Tp, ws)
               : xh_qdm_columns(ctx, bin, T, nb, Tp, af + c0, C, (const double*)d_q, nq, kind, interp, extrap, bout, Tp);
    if (rc) return rc;
    rc = xh_transpose_f32(ctx, bout, nb, T, Tp, scen + c0, st);
    if (rc) return rc;
  }
  return XH_OK;
}

// QuantileDeltaMapping.adjust for ALL groups of a sub-grouping whose groups are small (a day-of-year grouping: one row per year):
// rows (host, offs[G] entries): the row numbers of group 0, then of group 1, ...; offs (host, G + 1); af (G, nq, C): every
// group's factors; the ranks are taken inside each group (xsdba: group.apply(rank, sim, main_only=True)).  scen[t] is written
// for the listed rows only.  Bit-identical to xh_qdm_adjust on each group's gathered rows.  XH_ERR_NOTIMPL (no error text):
// a group of more than 64 rows or more than 32 nodes — gather the groups and call xh_qdm_adjust.
int xh_qdm_adjust_groups(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, const int32_t* rows, const int64_t* offs,
                         int G, const float* af, const double* q, int nq, int kind, int interp, int extrap, float* scen,
                         int64_t scen_st) {
  XH_REQUIRE(ctx && sim && rows && offs && af && q && scen, XH_ERR_ARG, "xh_qdm_adjust_groups: NULL argument");
  XH_REQUIRE(T >= 1 && C >= 0 && G >= 1 && nq >= 1 && st >= C && scen_st >= C, XH_ERR_ARG, "xh_qdm_adjust_groups: bad shape");
  XH_REQUIRE(kind >= 0 && kind <= 2, XH_ERR_ARG, "xh_qdm_adjust_groups: kind must be 0 (+), 1 (*) or 2 (the interpolated factor only)");
  XH_REQUIRE(interp == 0 || interp == 1, XH_ERR_NOTIMPL, "xh_qdm_adjust_groups: interp must be 0 (nearest) or 1 (linear)");
  XH_REQUIRE(extrap == 0 || extrap == 1, XH_ERR_ARG, "xh_qdm_adjust_groups: extrap must be 0 (constant) or 1 (nan)");
  XH_REQUIRE(sim != scen, XH_ERR_ARG, "xh_qdm_adjust_groups: not in place");
  for (int j = 1; j < nq; ++j)
    XH_REQUIRE(q[j] > q[j - 1], XH_ERR_ARG, "xh_qdm_adjust_groups: the quantile nodes must be strictly increasing");
  XH_REQUIRE(offs[0] == 0, XH_ERR_ARG, "xh_qdm_adjust_groups: offs[0] must be 0");
  int64_t most = 0;
  for (int g = 0; g < G; ++g) {
    XH_REQUIRE(offs[g + 1] >= offs[g], XH_ERR_ARG, "xh_qdm_adjust_groups: offs must not decrease");
    most = offs[g + 1] - offs[g] > most ? offs[g + 1] - offs[g] : most;
  }
  const int64_t nr = offs[G];
  for (int64_t k = 0; k < nr; ++k) XH_REQUIRE(rows[k] >= 0 && rows[k] < T, XH_ERR_ARG, "xh_qdm_adjust_groups: row %lld out of range", (long long)k);
  if (most > 64 || nq > 32) return XH_ERR_NOTIMPL;
  if (const char* e = xh_diag_env("XH_QDM_GROUPS"))
    if (!atoi(e)) return XH_ERR_NOTIMPL;
  if (C == 0 || nr == 0) return XH_OK;
  size_t cur = 0;
  void *d_q = nullptr, *d_rows = nullptr, *d_offs = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, q, sizeof(double) * (size_t)nq, &d_q);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, offs, sizeof(int64_t) * (size_t)(G + 1), &d_offs);
  if (!rc) rc = xh_scratch_upload(ctx, &cur, rows, sizeof(int32_t) * (size_t)nr, &d_rows);
  if (rc) return rc;
  const size_t lds = sizeof(double) * QDM_MAXQ + (size_t)nq * 256 * 5;
  const dim3 grid((unsigned)cdiv64(C, 256), (unsigned)G);
#define XH_QG(PER)                                                                                                                 \
  hipLaunchKernelGGL((k_qdm_groups<PER>), grid, dim3(256), lds, ctx->stream, sim, C, st, (const int32_t*)d_rows, (const int64_t*)d_offs, \
                     af, (const double*)d_q, nq, kind, interp, extrap, scen, scen_st)
  if (most <= 32) XH_QG(32);
  else XH_QG(64);
#undef XH_QG
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // extern "C"
