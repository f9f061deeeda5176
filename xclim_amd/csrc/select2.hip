// select2.hip — long-series multi-quantile selection WITHOUT a per-column copy of the series in LDS.
//
// select.hip keeps the whole column (sorted[T]) in LDS, which allows 1-2 workgroups per CU for a 30-year daily series
// and leaves the kernel latency bound (PMC: waves parked 74 % of their residency, 4 waves/SIMD).  Only the keys that
// fall into the <= 2*nq TARGET bins matter, so here:
//   pass 1  histogram of NB linear-in-key bins (keys stay in registers), exclusive scan
//   target  every target rank finds its bin, its rank inside the bin and the bin population m
//   pass 2  keys whose bin is a target bin are appended to a small LDS list (bins with m <= BIGM) or only update the
//           bin's min/max key (bigger bins: e.g. the "exact zero" bin of a precipitation series -> min == max, done)
//   select  exact k-th smallest inside the (tiny) bin list; a bin that is big AND not constant falls back to a
//           32-step bisection on the key value with workgroup-wide counting sweeps (rare, slow, exact)
// LDS per workgroup ~20 KB instead of ~60 KB -> 4 workgroups (32 waves) per CU.
#include <stdlib.h>

#include "common.h"

namespace {

constexpr int BIGM = 64;       // target bins up to this population are listed
constexpr int LISTCAP = 2048;  // LDS list capacity (keys); more -> the affected bins are treated as "big"
constexpr int MAXT = 128;      // targets = 2 * nq <= 128

struct TInfo {
  int bin, kth, m, region;  // region: list offset, or -(slot+1) for big bins (valid on the bin's OWNER target)
  int owner;                // first target that refers to the same bin
};

template <int NT>
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* red) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) s += red[i];
  return s;
}

template <int NT, int KPL, int NB>
__global__ void __launch_bounds__(NT)
k_select_lean(const float* __restrict__ x, int64_t T, int64_t ncols, int64_t col_stride, const double* __restrict__ qs,
              int nq, float* __restrict__ out, int64_t out_cstride, int64_t out_qstride, int abl) {
  constexpr int BPT = NB / NT;
  constexpr int NW = NT / 64;
  __shared__ uint32_t cur[NB];
  __shared__ uint32_t bmL[NB / 32], bmB[NB / 32];  // target-bin bitmaps: listed / big
  __shared__ TInfo tinfo[MAXT];
  __shared__ uint32_t bmin[MAXT], bmax[MAXT];
  __shared__ uint32_t list[LISTCAP];
  __shared__ float vals[MAXT];
  __shared__ uint32_t red[4 * NW + 8];
  __shared__ int s_slow, s_off, s_nslot;
  const int gt = threadIdx.x;
  const int lane = gt & 63, w = gt >> 6;
  const int ntgt = 2 * nq;

  for (int64_t col = blockIdx.x; col < ncols; col += gridDim.x) {
    // ---- load + one-pass reduction of (n, kmin, kmax)
    uint32_t key[KPL];
    uint32_t nv = 0, kmin = 0xFFFFFFFFu, kmax = 0u;
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      int i = gt + k * NT;
      key[k] = (i < T) ? xh_f2key(x[col * col_stride + i]) : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      uint32_t kk = key[k];
      bool ok = kk != 0xFFFFFFFFu;
      nv += ok ? 1u : 0u;
      kmin = (ok && kk < kmin) ? kk : kmin;
      kmax = (ok && kk > kmax) ? kk : kmax;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      nv += __shfl_xor(nv, off, 64);
      uint32_t a = __shfl_xor(kmin, off, 64), b = __shfl_xor(kmax, off, 64);
      kmin = a < kmin ? a : kmin;
      kmax = b > kmax ? b : kmax;
    }
    if (lane == 0) { red[w] = nv; red[NW + w] = kmin; red[2 * NW + w] = kmax; }
#pragma unroll
    for (int b = 0; b < BPT; ++b) cur[gt + b * NT] = 0;
    if (gt < NB / 32) { bmL[gt] = 0; bmB[gt] = 0; }
    if (gt == 0) s_slow = 0;
    __syncthreads();
    uint32_t n = 0;
    kmin = 0xFFFFFFFFu; kmax = 0u;
#pragma unroll
    for (int i = 0; i < NW; ++i) {
      n += red[i];
      kmin = red[NW + i] < kmin ? red[NW + i] : kmin;
      kmax = red[2 * NW + i] > kmax ? red[2 * NW + i] : kmax;
    }
    const uint32_t range = n > 0 ? kmax - kmin : 0u;
    int shift = 32 - __clz((int)range) - (31 - __clz(NB));
    shift = (range == 0u || shift < 0) ? 0 : shift;
    // ---- pass 1: histogram
    if (!(abl & 1)) {
#pragma unroll
    for (int k = 0; k < KPL; ++k)
      if (key[k] != 0xFFFFFFFFu) atomicAdd(&cur[(key[k] - kmin) >> shift], 1u);
    }
    __syncthreads();
    {  // exclusive scan: cur[b] = number of keys in bins < b
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
      __syncthreads();
      uint32_t add = 0;
      for (int i = 0; i < w; ++i) add += red[3 * NW + i];
      uint32_t run = incl - s + add;
#pragma unroll
      for (int b = 0; b < BPT; ++b) { cur[gt * BPT + b] = run; run += loc[b]; }
    }
    __syncthreads();
    // ---- targets: rank -> (bin, rank in bin, bin population)
    if (gt < ntgt && !(abl & 8)) {
      TInfo ti;
      ti.bin = -1; ti.kth = 0; ti.m = 0; ti.region = 0; ti.owner = gt;
      if (n >= 1) {
        const int j = gt >> 1;
        int r;
        if (T == 1 || n < 2) r = 0;
        else {
          double nn = (double)n, q = qs[j];
          double vi = nn * q + (1.0 + q * (1.0 - 1.0 - 1.0)) - 1.0;  // utl:395 with alpha = beta = 1
          if (vi >= nn - 1.0) r = (int)n - 1;
          else if (vi < 0.0) r = 0;
          else r = (int)floor(vi) + (gt & 1);
        }
        int lo = 0, hi = NB;  // largest b with start[b] <= r (empty bins share their start with the next bin)
        while (hi - lo > 1) {
          int mid = (lo + hi) >> 1;
          if (cur[mid] <= (uint32_t)r) lo = mid; else hi = mid;
        }
        const uint32_t s0 = cur[lo], s1 = (lo + 1 < NB) ? cur[lo + 1] : n;
        ti.bin = lo; ti.kth = (int)((uint32_t)r - s0); ti.m = (int)(s1 - s0);
      }
      tinfo[gt] = ti;
    }
    __syncthreads();
    // ---- list regions: the first target of every distinct bin ("owner") allocates the bin's list region (or a
    //      min/max slot for big bins) with an atomic bump; targets are NOT assumed to be sorted by rank
    if (gt == 0) { s_off = 0; s_nslot = 0; }
    __syncthreads();
    if (gt < ntgt) {
      TInfo ti = tinfo[gt];
      int owner = gt;
      if (ti.bin >= 0) {
        for (int s2 = 0; s2 < gt; ++s2)
          if (tinfo[s2].bin == ti.bin) { owner = s2; break; }
        if (owner == gt) {
          int region;
          int off = (ti.m <= BIGM) ? atomicAdd(&s_off, ti.m) : LISTCAP + 1;
          if (ti.m <= BIGM && off + ti.m <= LISTCAP) {
            region = off;
            cur[ti.bin] = (uint32_t)off;  // becomes the append cursor of this bin
            atomicOr(&bmL[ti.bin >> 5], 1u << (ti.bin & 31));
          } else {
            int slot = atomicAdd(&s_nslot, 1);
            region = -(slot + 1);
            cur[ti.bin] = (uint32_t)slot;  // slot of the min/max trackers
            bmin[slot] = 0xFFFFFFFFu; bmax[slot] = 0u;
            atomicOr(&bmB[ti.bin >> 5], 1u << (ti.bin & 31));
          }
          tinfo[gt].region = region;
        }
      }
      tinfo[gt].owner = owner;
    }
    __syncthreads();
    // ---- pass 2: collect the keys of the target bins
    if (!(abl & 2))
#pragma unroll
    for (int k = 0; k < KPL; ++k) {
      if (key[k] == 0xFFFFFFFFu) continue;
      const uint32_t b = (key[k] - kmin) >> shift;
      const uint32_t bit = 1u << (b & 31);
      if (bmL[b >> 5] & bit) {
        uint32_t pos = atomicAdd(&cur[b], 1u);
        list[pos] = key[k];
      } else if (bmB[b >> 5] & bit) {
        const uint32_t slot = cur[b];
        atomicMin(&bmin[slot], key[k]);
        atomicMax(&bmax[slot], key[k]);
      }
    }
    __syncthreads();
    // ---- select inside the bins: (target, candidate) pairs are spread over the whole workgroup — a candidate key is
    //      the answer iff #(keys < e) <= kth < #(keys <= e); O(m) LDS reads per thread instead of O(m^2) per target
    if (gt < ntgt) {
      const TInfo ti = tinfo[gt];
      float v = xh_nan32();
      if (ti.bin >= 0) {
        const int region = tinfo[ti.owner].region;
        if (region < 0) {
          const int slot = -region - 1;
          if (bmin[slot] == bmax[slot]) v = xh_key2f(bmin[slot]);  // constant bin (e.g. all the dry days)
          else atomicOr((unsigned int*)&s_slow, 1u);
        }
      }
      vals[gt] = v;
    }
    __syncthreads();
    if (!(abl & 4)) {
      constexpr int CPT = 16;  // candidate lanes per target
      for (int t = gt / CPT; t < ntgt; t += NT / CPT) {
        const TInfo ti = tinfo[t];
        if (ti.bin < 0) continue;
        const int region = tinfo[ti.owner].region;
        if (region < 0) continue;
        const uint32_t m = (uint32_t)ti.m, kth = (uint32_t)ti.kth;
        for (uint32_t a2 = gt % CPT; a2 < m; a2 += CPT) {
          const uint32_t e = list[region + a2];
          uint32_t less = 0, leq = 0;
          for (uint32_t b2 = 0; b2 < m; ++b2) {
            const uint32_t k2 = list[region + b2];
            less += k2 < e ? 1u : 0u;
            leq += k2 <= e ? 1u : 0u;
          }
          if (less <= kth && kth < leq) vals[t] = xh_key2f(e);  // every winner writes the same value
        }
      }
    }
    __syncthreads();
    // ---- rare: big non-constant target bins -> bisection on the key value with counting sweeps
    if (s_slow) {
      for (int t = 0; t < ntgt; ++t) {
        TInfo ti = tinfo[t];
        if (ti.bin < 0) continue;
        ti.region = tinfo[ti.owner].region;
        if (ti.region >= 0) continue;
        const int slot = -ti.region - 1;
        uint32_t lo = bmin[slot], hi = bmax[slot];
        if (lo == hi) continue;
        // global rank wanted: (#keys in lower bins) + kth ; recover it from the bin start
        // smallest K in [lo, hi] with #(key <= K, key in this bin) >= kth + 1
        const uint32_t binlo = kmin + ((uint32_t)ti.bin << shift);
        while (lo < hi) {
          const uint32_t mid = lo + ((hi - lo) >> 1);
          uint32_t c = 0;
#pragma unroll
          for (int k = 0; k < KPL; ++k) c += (key[k] != 0xFFFFFFFFu && key[k] >= binlo && key[k] <= mid) ? 1u : 0u;
          c = block_sum<NT>(c, red);
          if (c >= (uint32_t)ti.kth + 1u) hi = mid; else lo = mid + 1;
        }
        if (gt == 0) vals[t] = xh_key2f(lo);
        __syncthreads();
      }
    }
    // ---- Hyndman-Fan lerp (type 7) and store
    if (gt < nq) {
      const int j = gt;
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
    __syncthreads();
  }
}

template <int NT, int KPL, int NB>
int launch_lean(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride, const double* d_q, int nq,
                float* out, int64_t out_cstride, int64_t out_qstride) {
  int64_t nblk = ncols;
  int64_t maxblk = (int64_t)ctx->num_cu * 16;
  if (nblk > maxblk) nblk = maxblk;
  const char* ea = getenv("XH_SELECT_ABL");  // diagnostics only: skip phases (results become wrong)
  hipLaunchKernelGGL((k_select_lean<NT, KPL, NB>), dim3((unsigned)nblk), dim3(NT), 0, ctx->stream, xcols, T, ncols, col_stride,
                     d_q, nq, out, out_cstride, out_qstride, ea ? atoi(ea) : 0);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // namespace

// T in (1024, 16384]: keys per thread rounded up to a multiple of 4 to avoid idle register slots
int xh_select_columns_lean(xh_ctx* ctx, const float* xcols, int64_t T, int64_t ncols, int64_t col_stride,
                           const double* d_q, int nq, float* out, int64_t out_cstride, int64_t out_qstride) {
  if (T <= 1024 || T > 16384 || nq > 64) return XH_ERR_NOTIMPL;
#define XH_LEAN(NT, KPL, NB) return launch_lean<NT, KPL, NB>(ctx, xcols, T, ncols, col_stride, d_q, nq, out, out_cstride, out_qstride)
  if (T <= 2048) XH_LEAN(256, 8, 1024);
  if (T <= 3072) XH_LEAN(256, 12, 1024);
  if (T <= 4096) XH_LEAN(256, 16, 1024);
  if (T <= 6144) XH_LEAN(512, 12, 2048);
  if (T <= 8192) XH_LEAN(512, 16, 2048);
  if (T <= 10240) XH_LEAN(512, 20, 2048);
  if (T <= 12288) XH_LEAN(512, 24, 2048);
  if (T <= 14336) XH_LEAN(512, 28, 2048);
  XH_LEAN(512, 32, 2048);
#undef XH_LEAN
}
