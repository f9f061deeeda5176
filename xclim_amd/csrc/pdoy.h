// pdoy.h — pieces shared by the percentile_doy kernels in quantile.hip and pdoy_top.hip.
#pragma once
#include "common.h"

// (lo, hi, gamma) of the Hyndman-Fan estimate for one (percentile j, valid count n): built on the host in fp64 exactly
// as utl:395/417-461 evaluate it (build_qtab in quantile.hip), indexed [j * (N + 1) + n].
struct QTab {
  int lo, hi;     // sorted slots to combine; lo < 0 -> NaN (no valid sample)
  double gamma;   // interpolation weight (0 when lo == hi)
};
// A per-doy byte flag read with a SCALAR load (s_load_dword of the enclosing word; the tables live in the 256-byte aligned
// context scratch).  `flags[d]` itself is a vector byte load, and the s_waitcnt vmcnt(0) before its use drains every gather
// load in flight at that point.
__device__ __forceinline__ uint32_t pdoy_flag(const uint8_t* __restrict__ flags, int d) {
  const uint32_t w = ((const uint32_t*)flags)[__builtin_amdgcn_readfirstlane(d) >> 2];
  return (w >> ((d & 3) * 8)) & 0xFFu;
}
// Gather plumbing shared by the multi-year kernels.  Lane y resolves the physical row of (year y, doy dn, offset off)
// with ONE vector load chain (tbase -> vmap); the unrolled gather then reads the rows with readlane + unconditional,
// clamped loads.  (Resolving the rows with per-year scalar loads serialised every gather behind two s_load latencies:
// ~22k clk per doy, 4x the whole rest of the kernel.)
// A day-set index outside [0, ndoy) wraps into the neighbouring year: day-set -1 is {(y - 1, ndoy - 1)} — the day before
// (y, 0) on a calendar without gaps.  With it the windows of the first / last W/2 doys of a multi-year base period
// decompose into day-sets like every other doy (pdoy_regular_flags verifies exactly that, row by row).
// In two halves, so that a kernel can issue the table load one step before it needs the row (the value arrives behind
// the gather loads issued just before it, at no extra latency): pdoy_row_fetch returns the VIRTUAL day index (or -1).
__device__ __forceinline__ int pdoy_row_fetch(int lane, int nyears, int ndoy, int dn, const int32_t* __restrict__ tbase) {
  const int yy = dn < 0 ? lane - 1 : (dn >= ndoy ? lane + 1 : lane);
  const int dd = dn < 0 ? dn + ndoy : (dn >= ndoy ? dn - ndoy : dn);
  int v = -1;
  if (lane < nyears && yy >= 0 && yy < nyears && dd >= 0 && dd < ndoy) v = tbase[(int64_t)yy * ndoy + dd];
  return v;
}
__device__ __forceinline__ int pdoy_row_finish(int v, int off, const int32_t* __restrict__ vmap, int64_t Tv, int64_t T) {
  int tp = -1;
  const int64_t vv = (int64_t)v + off;
  if (v >= 0 && vv >= 0 && vv < Tv) {
    const int64_t p = vmap ? (int64_t)vmap[vv] : vv;
    if (p >= 0 && p < T) tp = (int)p;
  }
  return tp;
}
__device__ __forceinline__ int pdoy_row(int lane, int nyears, int ndoy, int dn, int off, const int32_t* __restrict__ tbase,
                                        const int32_t* __restrict__ vmap, int64_t Tv, int64_t T) {
  return pdoy_row_finish(pdoy_row_fetch(lane, nyears, ndoy, dn, tbase), off, vmap, Tv, T);
}
// Each row is read through a buffer resource whose base is the wave-uniform row pointer (SGPRs from the readlane) with the
// lane's column as the 32-bit offset: `buffer_load_dword v, v_off, s[rsrc], 0 offen` — the address arithmetic is scalar,
// no 64-bit vector add per sample (C < 2^29: the column offset is a 32-bit byte offset; host-checked by the launchers).
// An absent day (row < 0) reads NaN: with `nanrow` (>= C floats of NaN, xh_const_rows) by a scalar select of the row
// pointer, so that the gather is loads and nothing else — any vector instruction on the loaded values here would have to
// wait for them at the place of the gather instead of where they are consumed one step later; without it by OR-ing a
// wave-uniform mask into the bits (written as `tp < 0 ? NaN : f` the compiler sinks the load into a scalar branch per sample).
template <int NYP, bool NANROW = false>
__device__ __forceinline__ void pdoy_gather(float (&raw)[NYP], int rowv, const float* __restrict__ x, int64_t st, int64_t cc,
                                            const float* __restrict__ nanrow = nullptr) {
  const uint32_t coff = (uint32_t)cc * 4u;  // byte offset of the lane's column
#pragma unroll
  for (int y = 0; y < NYP; ++y) {
    const int tp = __builtin_amdgcn_readlane(rowv, y);
    const float* rowp = x + (int64_t)(tp < 0 ? 0 : tp) * st;
    if (NANROW) rowp = tp < 0 ? nanrow : rowp;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)rowp, 0, 0x7FFFFFFF, 0x00020000);
    const uint32_t f = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)coff, 0, 0);
    raw[y] = __uint_as_float(NANROW ? f : (f | (tp < 0 ? 0x7FFFFFFFu : 0u)));
  }
}

// The largest non-NaN sample of the window of doy d, read again from memory (the day-sets d - W/2 .. d + W/2): the
// nanmax fallback of utl:552-554 for the kernels that keep only the 16 SMALLEST samples in registers.  Reached when a
// low-percentile result is NaN with more than 16 valid samples, i.e. with -inf samples at the selected ranks; the branch
// around the call must be wave-uniform.
__device__ __forceinline__ float pdoy_window_nanmax(int d, int W, int lane, int nyears, int ndoy, const int32_t* __restrict__ tbase,
                                                    const int32_t* __restrict__ vmap, int64_t Tv, int64_t T,
                                                    const float* __restrict__ x, int64_t st, int64_t cc) {
  float m = __uint_as_float(0xFF800000u);
  for (int k = -(W / 2); k <= W / 2; ++k) {
    const int rowv = pdoy_row(lane, nyears, ndoy, d + k, 0, tbase, vmap, Tv, T);
    for (int y = 0; y < nyears; ++y) {
      const int tp = __builtin_amdgcn_readlane(rowv, y);
      if (tp < 0) continue;
      const float f = x[(int64_t)tp * st + cc];
      m = (f == f && f > m) ? f : m;
    }
  }
  return m;
}

// Host: regular[d] = 1 when the window sample set of doy d (time offsets -W/2 .. W/2 around every year's day, NaN outside
// the series) equals the union of the W neighbouring DAY-SETS {(y, d + k)} — true except around calendar gaps (Feb 29 /
// doy 366, partial first or last years); the irregular doys are listed for the exact fallback kernels.  Returns their
// number.  `vmap` / Tv: optional virtual time map (bootstrap replicas), else Tv = T.
static inline int pdoy_regular_flags(const int32_t* tbase, int nyears, int ndoy, int window, int64_t T, const int32_t* vmap,
                                     int64_t Tv, uint8_t* regular, int32_t* irregular) {
  const int half = window / 2;
  int nirr = 0;
  for (int d = 0; d < ndoy; ++d) {
    bool ok = true;
    for (int y = 0; ok && y < nyears; ++y) {
      int v = tbase[(int64_t)y * ndoy + d];
      for (int k = 0; ok && k < window; ++k) {
        int64_t a = -1;  // virtual index the window semantic reads
        if (v >= 0) {
          int64_t t = (int64_t)v - half + k;
          if (t >= 0 && t < Tv) a = t;
        }
        int dn = d - half + k;
        const int yy = dn < 0 ? y - 1 : (dn >= ndoy ? y + 1 : y);              // (same wrap as pdoy_row)
        const int dd = dn < 0 ? dn + ndoy : (dn >= ndoy ? dn - ndoy : dn);
        int64_t b = (yy >= 0 && yy < nyears && dd >= 0 && dd < ndoy) ? (int64_t)tbase[(int64_t)yy * ndoy + dd] : -1;
        // compare the PHYSICAL rows (two virtual days may map to the same / to an absent row)
        int64_t pa = a < 0 ? -1 : (vmap ? vmap[a] : a), pb = b < 0 ? -1 : (vmap ? vmap[b] : b);
        if (pa >= T) pa = -1;
        if (pb >= T) pb = -1;
        ok = pa == pb;
      }
    }
    regular[d] = ok ? 1 : 0;
    if (!ok) irregular[nirr++] = d;
  }
  return nirr;
}

// doystats.hip: climatological mean / std per doy from per-day-set partial sums (regular doys of the chunk grid)
int xh_launch_doy_stats_sets(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* d_tb, int nyears,
                             int ndoy, int window, const uint8_t* d_reg, float* mean_out, float* std_out, int64_t year_t0);

// pdoy_top.hip: register top-16 kernel for the percentiles in jmap[0..nsub) (rev = 0: all of them select within the 16
// largest samples; rev = 1: within the 16 smallest) on the REGULAR doys (d_reg[d] != 0) of the chunk grid; irregular doys
// (window does not decompose into day-sets, e.g. around Feb 29) are left to k_pdoy_merge.
int xh_launch_pdoy_top16(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* d_tb, int nyears,
                         int ndoy, int window, const QTab* d_tab, const int32_t* d_jmap, int nsub, int rev, double* out,
                         const int32_t* d_vmap, int64_t Tv, const uint8_t* d_reg);
// pdoy_quad.hip: the same contract for window 5 with quad sharing (the default there); XH_ERR_NOTIMPL without an error
// text = not its shape, take k_pdoy_top16
int xh_launch_pdoy_quad(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* d_tb, int nyears,
                        int ndoy, int window, const QTab* d_tab, const int32_t* d_jmap, int nsub, int bot, double* out,
                        const int32_t* d_vmap, int64_t Tv, const uint8_t* d_reg);
// pdoy_walk.hip: percentiles anywhere in the distribution on the regular doys (sorted day-set lists in LDS, a split that
// walks from day to day); XH_ERR_NOTIMPL without an error text = not its shape, take k_pdoy_merge
int xh_launch_pdoy_walk(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* d_tb, int nyears,
                        int ndoy, int window, const QTab* d_tab, const int32_t* d_jmap, int nsub, double* out,
                        const int32_t* d_vmap, int64_t Tv, const uint8_t* d_reg);
// COUNT variant (xh_percentile_doy_count, multi-year base period): one percentile, every doy regular; the exceedances
// of (year y, doy d) are added to period d_period[y * ndoy + d] (atomics: cnt_out / valid_out must be zeroed);
// d_newseg[d] = 1 where the period of doy d differs from that of doy d - 1 — for EVERY year at once (host-checked)
int xh_launch_pdoy_top16_count(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* d_tb, int nyears,
                               int ndoy, int window, const QTab* d_tab, const int32_t* d_jmap, int rev, const uint8_t* d_reg,
                               int op, const int32_t* d_period, int32_t* cnt_out, int32_t* valid_out,
                               const uint8_t* d_newseg);
