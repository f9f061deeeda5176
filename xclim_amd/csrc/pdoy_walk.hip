// pdoy_walk.hip — percentile_doy on multi-year base periods for percentiles ANYWHERE in the distribution (the median,
// quartiles ...): the ones the register top-16 kernels (pdoy_quad.hip, pdoy_top.hip) cannot serve.
//
// The W day-sets of a window live as sorted lists in LDS (lane-private columns [slot][i][lane], conflict-free), one new
// list per calendar day, as in k_pdoy_merge (quantile.hip).  k_pdoy_merge finds an order statistic by popping the W lists
// from one end: min(k, N - k) dependent LDS reads per day and cell — 75 for the median of 150 samples — on a ring that
// leaves one wave per CU.  But consecutive days share W - 1 of their W lists, so the answer hardly moves.  This kernel
// keeps, per percentile and cell, a SPLIT of every list into a lower and an upper part,
//     sum_w p[w] = k = (rank lo) + 1,   every lower element <= every upper element,
// so that rank lo = the largest lower head and rank lo + 1 = the smallest upper head.  When a list is replaced, the new
// list is split at the current largest lower value (32 compares in registers, no chain), which keeps the invariant for
// whatever k; then the split WALKS: while sum p > k the largest lower head moves up, while sum p < k the smallest upper
// head moves down — one dependent LDS read per step, and the expected distance is a few steps (30 of 150 samples changed).
// All W lists of 30 - 32 samples in full: 38 - 40 KB per wave, 3 - 4 waves per CU.
// Irregular doys (leap day ...) stay with k_pdoy_merge<OFFSET>; nyears > 32 stays with k_pdoy_merge.
#include <stdio.h>
#include <stdlib.h>

#include "pdoy.h"
#include "topnet.h"

namespace {
constexpr int PW_NS = 4;  // percentiles per launch
}

template <int W>
__global__ void __launch_bounds__(64)
k_pdoy_walk(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, const int32_t* __restrict__ tbase, int nyears,
            int ndoy, int chunk, const QTab* __restrict__ qtab, const int32_t* __restrict__ jmap, int nsub,
            double* __restrict__ out, const int32_t* __restrict__ vmap, int64_t Tv, const uint8_t* __restrict__ regular,
            const float* __restrict__ nanrow, const float* __restrict__ padrow, int abl, unsigned long long* __restrict__ steps) {
  constexpr int NYP = 32, half = W / 2;
  extern __shared__ float lds[];
  const int NL = nyears;           // rows per list (only positions below the valid count are ever read)
  float* Ls = lds;                 // [W][NL][64] ascending, the valid samples first
  const int lane = threadIdx.x;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  const bool active = c < C;
  const uint32_t coff = (uint32_t)(active ? c : C - 1) * 4u;
  const int N = nyears * W;
  const float PINF = __uint_as_float(0x7F800000u), NINF = __uint_as_float(0xFF800000u);
  float SENT = PINF;  // ascending lists: NaN / absent / padding sort last
  asm volatile("" : "+v"(SENT));

  int cw[W];                     // valid samples of the list in slot w
  int p[PW_NS][W];               // lower part of list w for percentile jj
  float lh[PW_NS][W], uh[PW_NS][W];  // its heads: list[p - 1] (or -inf) and list[p] (or +inf)
#pragma unroll
  for (int w = 0; w < W; ++w) {
    cw[w] = 0;
#pragma unroll
    for (int jj = 0; jj < PW_NS; ++jj) {
      p[jj][w] = 0;
      lh[jj][w] = NINF;
      uh[jj][w] = PINF;
    }
  }

  float raw[NYP];
  uint32_t nlo, nhi;
  int tbv;
  auto resolve = [&](int v, uint32_t& alo, uint32_t& ahi) {
    const int tp = pdoy_row_finish(v, 0, vmap, Tv, T);
    const float* pp = lane >= nyears ? padrow : (tp < 0 ? nanrow : x + (int64_t)tp * st);
    alo = (uint32_t)(uintptr_t)pp;
    ahi = (uint32_t)((uintptr_t)pp >> 32);
  };
  auto gather = [&](uint32_t alo, uint32_t ahi) {
#pragma unroll
    for (int y = 0; y < NYP; ++y) {
      const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)alo, y), hi = (uint32_t)__builtin_amdgcn_readlane((int)ahi, y);
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, 0x7FFFFFFF, 0x00020000);
      raw[y] = __uint_as_float((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)coff, 0, 0));
    }
  };
  auto list_at = [&](int w, int i) -> float { return Ls[((w * NL) + i) * 64 + lane]; };

  // the gathered day-set dn enters slot `slot` (wave-uniform): sort, store, re-split for every percentile
  auto enter = [&](int slot, int dn) {
    float s0 = raw[0], s1 = raw[1];
#pragma unroll
    for (int y = 2; y < NYP; y += 2) {
      s0 += raw[y];
      s1 += raw[y + 1];
    }
    s0 += s1;
    int nv = nyears;
    if (__any(s0 != s0 ? 1 : 0)) {  // (the padding slots hold +inf: a -inf sample makes the sum NaN — the careful path is right for it)
      int nn = 0;
#pragma unroll
      for (int y = 0; y < NYP; ++y) tn_denan_inplace(raw[y], nn, SENT);
      nv = nyears - nn;
    }
    float key[NYP];
#pragma unroll
    for (int y = 0; y < NYP; ++y) key[y] = raw[y];
    uint32_t rlo, rhi;
    resolve(tbv, rlo, rhi);
    gather(nlo, nhi);
    nlo = rlo;
    nhi = rhi;
    tbv = pdoy_row_fetch(lane, nyears, ndoy, dn + 3, tbase);
    if (!(abl & 2)) tn_sort32<true>(key);  // ascending; NaN -> +inf and the +inf padding last (a sample that IS +inf ties with them: same value)
#pragma unroll
    for (int i = 0; i < NYP; ++i)
      if (i < NL) Ls[((slot * NL) + i) * 64 + lane] = key[i];
#pragma unroll
    for (int jj = 0; jj < PW_NS; ++jj) {
      if (jj < nsub) {
        // the largest lower value of the OTHER lists
        float lcur = NINF;
        bool has = false;
#pragma unroll
        for (int w = 0; w < W; ++w) {
          const bool use = w != slot && p[jj][w] > 0;
          lcur = (use && (!has || lh[jj][w] > lcur)) ? lh[jj][w] : lcur;
          has |= use;
        }
        int pn = 0;
#pragma unroll
        for (int i = 0; i < NYP; ++i) pn += (has && i < nv && key[i] <= lcur) ? 1 : 0;
        const float nl = pn > 0 ? list_at(slot, pn - 1) : NINF;
        const float nu = pn < nv ? list_at(slot, pn < NL ? pn : NL - 1) : PINF;
#pragma unroll
        for (int w = 0; w < W; ++w) {
          p[jj][w] = w == slot ? pn : p[jj][w];
          lh[jj][w] = w == slot ? nl : lh[jj][w];
          uh[jj][w] = w == slot ? nu : uh[jj][w];
        }
      }
    }
#pragma unroll
    for (int w = 0; w < W; ++w) cw[w] = w == slot ? nv : cw[w];
  };

  auto select_and_store = [&](int d) {
    int n = 0;
#pragma unroll
    for (int w = 0; w < W; ++w) n += cw[w];
    // (one percentile after the other: all of them walking in the same loop iterations — independent chains for a CU that
    //  holds one wave per SIMD — was measured SLOWER, 85 against 60 ms for one percentile, 148 against 136 for three)
#pragma unroll
    for (int jj = 0; jj < PW_NS; ++jj) {
      if (jj < nsub) {
        const int j = jmap[jj];
        const QTab e = qtab[j * (N + 1) + n];
        double r = xh_nan64();
        const int k = e.lo + 1;  // size of the lower part (0 when there is no valid sample: nothing moves)
        int sp = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) sp += p[jj][w];
        unsigned nit = 0;
        // Two single-direction loops (a cell walks down OR up), integer flags and bitwise logic, unconditional clamped LDS
        // reads: written as one loop with && / || and a direction select per operand, the compiler builds the step out of
        // 18 divergent branches (s_and_saveexec / s_cbranch_execz) — 1700 cycles per step instead of ~650.
        const int ok = e.lo >= 0 ? 1 : 0;
        while (__any(ok & (sp > k ? 1 : 0))) {  // the largest lower head becomes an upper head
          ++nit;
          const int act = ok & (sp > k ? 1 : 0);
          int wsel = -1;
          float best = NINF;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            const int take = act & (p[jj][w] > 0 ? 1 : 0) & ((wsel < 0 ? 1 : 0) | (lh[jj][w] > best ? 1 : 0));
            best = take ? lh[jj][w] : best;
            wsel = take ? w : wsel;
          }
          int psel = 1;
#pragma unroll
          for (int w = 0; w < W; ++w) psel = w == wsel ? p[jj][w] : psel;
          const int pn = psel - 1;
          const float got = list_at(wsel < 0 ? 0 : wsel, pn > 0 ? pn - 1 : 0);
          const float nl = pn > 0 ? got : NINF;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            const bool me = w == wsel;
            p[jj][w] = me ? pn : p[jj][w];
            uh[jj][w] = me ? best : uh[jj][w];
            lh[jj][w] = me ? nl : lh[jj][w];
          }
          sp -= act;
        }
        while (__any(ok & (sp < k ? 1 : 0))) {  // the smallest upper head becomes a lower head
          ++nit;
          const int act = ok & (sp < k ? 1 : 0);
          int wsel = -1;
          float best = PINF;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            const int take = act & (p[jj][w] < cw[w] ? 1 : 0) & ((wsel < 0 ? 1 : 0) | (uh[jj][w] < best ? 1 : 0));
            best = take ? uh[jj][w] : best;
            wsel = take ? w : wsel;
          }
          int psel = 0, csel = 0;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            psel = w == wsel ? p[jj][w] : psel;
            csel = w == wsel ? cw[w] : csel;
          }
          const int pn = psel + 1;
          const float got = list_at(wsel < 0 ? 0 : wsel, pn < NL ? pn : NL - 1);
          const float nu = pn < csel ? got : PINF;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            const bool me = w == wsel;
            p[jj][w] = me ? pn : p[jj][w];
            lh[jj][w] = me ? best : lh[jj][w];
            uh[jj][w] = me ? nu : uh[jj][w];
          }
          sp += act;
        }
        if (steps && lane == 0) atomicAdd(steps, (unsigned long long)nit);
        if (e.lo >= 0) {
          float left = NINF, right = PINF;
          bool hl = false, hr = false;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            const bool al = p[jj][w] > 0, ar = p[jj][w] < cw[w];
            left = (al && (!hl || lh[jj][w] > left)) ? lh[jj][w] : left;
            hl |= al;
            right = (ar && (!hr || uh[jj][w] < right)) ? uh[jj][w] : right;
            hr |= ar;
          }
          if (e.hi == e.lo) right = left;
          const float diff = right - left;
          r = (double)left + (double)diff * e.gamma;
          if (e.gamma >= 0.5) r = (double)right - (double)diff * (1.0 - e.gamma);
        }
        if (__any((r != r && n > 0 && e.lo >= 0) ? 1 : 0)) {  // +-inf samples: nanmax fallback (utl:552-554)
          float m = NINF;
#pragma unroll
          for (int w = 0; w < W; ++w) {
            const float t0 = cw[w] > 0 ? list_at(w, cw[w] - 1) : NINF;
            m = t0 > m ? t0 : m;
          }
          if (r != r && n > 0 && e.lo >= 0) r = (double)m;
        }
        if (active) out[((int64_t)j * ndoy + d) * C + c] = r;
      }
    }
  };

  const int d0 = blockIdx.y * chunk;
  int d1 = d0 + chunk;
  if (d1 > ndoy) d1 = ndoy;
  const int dstart = d0 - (W - 1);  // the ring starts empty W - 1 steps before the chunk (those steps select nothing)
  {
    uint32_t alo, ahi;
    resolve(pdoy_row_fetch(lane, nyears, ndoy, dstart + half, tbase), alo, ahi);
    gather(alo, ahi);
    resolve(pdoy_row_fetch(lane, nyears, ndoy, dstart + half + 1, tbase), nlo, nhi);
    tbv = pdoy_row_fetch(lane, nyears, ndoy, dstart + half + 2, tbase);
  }
  for (int d = dstart; d < d1; ++d) {
    const int dn = d + half;
    enter(((dn % W) + W) % W, dn);
    if (d >= d0 && pdoy_flag(regular, d) && !(abl & 1)) select_and_store(d);
  }
}

// XH_ERR_NOTIMPL (no error text): not this kernel's shape — the caller takes k_pdoy_merge
int xh_launch_pdoy_walk(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* d_tb, int nyears,
                        int ndoy, int window, const QTab* d_tab, const int32_t* d_jmap, int nsub, double* out,
                        const int32_t* d_vmap, int64_t Tv, const uint8_t* d_reg) {
  if ((window != 3 && window != 5 && window != 7) || nyears > 32 || C >= ((int64_t)1 << 29) || nsub < 1) return XH_ERR_NOTIMPL;
  if (const char* e = xh_diag_env("XH_PDOY_WALK"))
    if (!atoi(e)) return XH_ERR_NOTIMPL;
  const float *nanrow = nullptr, *pinf = nullptr;
  if (int rc = xh_const_rows(ctx, C, &nanrow, nullptr, &pinf)) return rc;
  int chunk = 92;
  if (const char* e = xh_diag_env("XH_PDOY_CHUNK")) chunk = atoi(e) > 0 ? atoi(e) : chunk;
  const dim3 grid((unsigned)cdiv64(C, 64), (unsigned)((ndoy + chunk - 1) / chunk));
  const size_t lds = (size_t)window * (size_t)nyears * 64 * sizeof(float);
  const char* ea = xh_diag_env("XH_PDOY_ABL");  // diagnostics (results wrong): 1 = no selection, 2 = no sort
  const int abl = ea ? atoi(ea) : 0;
  unsigned long long* d_steps = nullptr;  // diagnostics: XH_PDOY_WALK_STEPS=1 prints the walk steps per (wave, day, percentile)
  if (xh_diag_env("XH_PDOY_WALK_STEPS")) {
    XH_CHECK_HIP(hipMalloc((void**)&d_steps, 8));
    XH_CHECK_HIP(hipMemsetAsync(d_steps, 0, 8, ctx->stream));
  }
  for (int j0 = 0; j0 < nsub; j0 += PW_NS) {
    const int ns = nsub - j0 < PW_NS ? nsub - j0 : PW_NS;
#define XH_WALK(W)                                                                                                      \
  do {                                                                                                                  \
    if (lds > 64 * 1024)                                                                                                \
      XH_CHECK_HIP(hipFuncSetAttribute((const void*)k_pdoy_walk<W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
    hipLaunchKernelGGL((k_pdoy_walk<W>), grid, dim3(64), lds, ctx->stream, x, T, C, st, d_tb, nyears, ndoy, chunk, d_tab, \
                       d_jmap + j0, ns, out, d_vmap, Tv, d_reg, nanrow, pinf, abl, d_steps);                                         \
  } while (0)
    if (window == 3) XH_WALK(3); else if (window == 5) XH_WALK(5); else XH_WALK(7);
#undef XH_WALK
    XH_LAUNCH_CHECK();
  }
  if (d_steps) {
    unsigned long long h = 0;
    XH_CHECK_HIP(hipMemcpyAsync(&h, d_steps, 8, hipMemcpyDeviceToHost, ctx->stream));
    XH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
    fprintf(stderr, "[k_pdoy_walk] %.2f walk steps per (wave, day, percentile)\n",
            (double)h / ((double)grid.x * (double)ndoy * (double)nsub));
    XH_CHECK_HIP(hipFree(d_steps));
  }
  return XH_OK;
}
