// pdoy_quad.hip — percentile_doy on multi-year base periods, window 5, register top-16 with QUAD sharing.
//
// Same problem and the same building blocks as k_pdoy_top16 (pdoy_top.hip): the order statistics of a high (BOT =
// false) or low (BOT = true) percentile lie within the 16 outermost samples of the window, each calendar day's
// year-samples ("day-set" L_k) are gathered once, sorted in registers, and only their 16 outermost kept.  What differs is
// how the five day-sets of a window are combined.  With pairs and quads on an EVEN grid,
//     P_e = L_e u L_{e+1},   Q_e = P_{e-2} u P_e = L_{e-2} .. L_{e+1}            (e even)
// one quad serves TWO windows:
//     window_{e-1} = L_{e-3} u Q_e,        window_e = Q_e u L_{e+2}.
// Per two calendar days: one pair merge, one quad merge and two "last two of the merged 16" selections — 112
// single-instruction comparators per day instead of the 192 of the pairs-only scheme (k_pdoy_top16: a pair merge and a
// merge of two pairs for EVERY day), on 64 state registers instead of 80, and the state does not rotate (16 register
// copies per two days instead of 73 per day).
// Round 4 anatomy behind this kernel (profiles/r04/pdoy_anatomy.txt): k_pdoy_top16 is bound by vector instruction issue
// (v_max / v_min take 4.3 cycles per wave); removing its stalls (loads drained by a byte-flag load and by the row
// resolution placed after the gather) changed nothing, removing instructions did.  Therefore also:
//   * a day-set without NaN (one v_add chain + one compare per wave decides) skips the per-sample compare / select /
//     count; padding slots read -inf / +inf from a constant row and absent days NaN from another (xh_const_rows), so the
//     gather is loads and nothing else;
//   * the row ADDRESSES are resolved per lane (lane y = year y: one 64-bit multiply-add for all samples of the day-set)
//     and broadcast with two v_readlane per sample — 3 scalar instructions per load instead of 12.
#include <stdlib.h>

#include "pdoy.h"
#include "topnet.h"

// ABL (diagnostics, results wrong): 1 = no sorting networks, 2 = no merges / selection — compile-time, so that the
// production instance carries none of it
template <int NYP, bool BOT, int ABL = 0>
__global__ void __launch_bounds__(64, NYP > 32 ? 2 : 3)
k_pdoy_quad(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, const int32_t* __restrict__ tbase, int nyears,
            int ndoy, int chunk, const QTab* __restrict__ qtab, const int32_t* __restrict__ jmap, int nsub,
            double* __restrict__ out, const int32_t* __restrict__ vmap, int64_t Tv, const uint8_t* __restrict__ regular,
            const float* __restrict__ nanrow, const float* __restrict__ padrow) {
  const int lane = threadIdx.x;
  const int64_t c = (int64_t)blockIdx.x * 64 + lane;
  const bool active = c < C;
  const uint32_t coff = (uint32_t)(active ? c : C - 1) * 4u;  // inactive lanes read a valid cell and never store
  const int N = nyears * 5;
  float SENT = tn_sentinel<BOT>();
  asm volatile("" : "+v"(SENT));  // one VGPR, not a literal per use

  float Qp[16], Pp[16], La[16], Lb[16];  // Q_{e-2}, P_{e-2}, L_{e-3}, L_{e-1} at the top of iteration e
  int nQp = 0, nPp = 0, nLa = 0, nLb = 0;
  float raw[NYP];
  uint32_t nlo, nhi;  // lane y: address of the row (year y, NEXT day-set to gather)

  // lane y -> address of the row of (year y, day-set dn): the sample row, the NaN row (absent day) or the padding row
  auto resolve = [&](int v, uint32_t& alo, uint32_t& ahi) {  // v: pdoy_row_fetch of the day-set
    const int tp = pdoy_row_finish(v, 0, vmap, Tv, T);
    const float* p = lane >= nyears ? padrow : (tp < 0 ? nanrow : x + (int64_t)tp * st);
    alo = (uint32_t)(uintptr_t)p;
    ahi = (uint32_t)((uintptr_t)p >> 32);
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
  // The gathered day-set dn -> its 16 outermost, sorted, + its valid count.  Day-set dn + 1 is requested as soon as the
  // registers are free; the table load behind the rows of dn + 3 is issued right after that gather and consumed one call
  // later (`tbv`, for dn + 2 here), where it has arrived with the samples: issued and awaited in the same call, it cost a
  // full memory round trip per day with nothing else in flight (profiles/r04/pdoy_anatomy.txt).
  int tbv;
  auto take = [&](float (&top)[16], int& nv, int dn) {
    float s0 = raw[0], s1 = raw[1];
#pragma unroll
    for (int y = 2; y < NYP; y += 2) {
      s0 += raw[y];
      s1 += raw[y + 1];
    }
    s0 += s1;  // NaN iff a sample is NaN (or +inf meets -inf: the careful path is right for those, too)
    if (__any(s0 != s0 ? 1 : 0)) {
      int nn = 0;
#pragma unroll
      for (int y = 0; y < NYP; ++y) tn_denan_inplace(raw[y], nn, SENT);
      nv = nyears - nn;
    } else {
      nv = nyears;
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
    float blk[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) top[i] = key[i];
    if constexpr ((ABL & 1) != 0) {  // diagnostics: no sorting (results wrong)
#pragma unroll
      for (int i = 0; i < 16; ++i) top[i] = tn_outer<BOT>(key[i], key[i + NYP - 16]);
      return;
    }
    tn_sort16<BOT>(top);
#pragma unroll
    for (int b = 1; b < NYP / 16; ++b) {
#pragma unroll
      for (int i = 0; i < 16; ++i) blk[i] = key[b * 16 + i];
      tn_sort16<BOT>(blk);
      tn_merge16<BOT>(top, top, blk);
    }
  };
  // percentile(s) of doy d from the two sorted 16-lists whose union holds the window's 16 outermost; n = valid samples
  auto select_and_store = [&](int d, const float (&a)[16], const float (&b)[16], int n) {
    // fast path: one percentile, the same sample count in every lane, ranks at positions 14 / 15 of the 16
    if (nsub == 1) {
      const int n0 = __builtin_amdgcn_readfirstlane(n);
      if (__all(n == n0 ? 1 : 0)) {
        const int j = jmap[0];
        const QTab e = qtab[j * (N + 1) + n0];  // wave-uniform
        const int plo = BOT ? e.lo : (n0 - 1 - e.lo), phi = BOT ? e.hi : (n0 - 1 - e.hi);
        if (e.lo >= 0 && plo >= 14 && plo <= 15 && phi >= 14 && phi <= 15) {
          float p15, p14;
          tn_last2<BOT>(a, b, p15, p14);
          const float left = plo == 15 ? p15 : p14, right = phi == 15 ? p15 : p14;
          const float diff = right - left;
          double r = (double)left + (double)diff * e.gamma;
          if (e.gamma >= 0.5) r = (double)right - (double)diff * (1.0 - e.gamma);
          if (!__any(r != r ? 1 : 0)) {  // (+-inf samples: the nanmax rule below needs the whole list)
            if (active) out[((int64_t)j * ndoy + d) * C + c] = r;
            return;
          }
        }
      }
    }
    float t16[16];
    tn_merge16<BOT>(t16, a, b);
    auto get = [&](int idx) -> float {
      uint32_t g = 0;
#pragma unroll
      for (int i = 0; i < 16; ++i) g |= (i == idx) ? __float_as_uint(t16[i]) : 0u;
      return __uint_as_float(g);
    };
    for (int jj = 0; jj < nsub; ++jj) {
      const int j = jmap[jj];
      const QTab e = qtab[j * (N + 1) + n];
      double r = xh_nan64();
      bool needmax = false;
      if (e.lo >= 0) {
        // position in the outer-first 16: BOT -> rank from the bottom, else rank from the top
        const int plo = BOT ? e.lo : (n - 1 - e.lo), phi = BOT ? e.hi : (n - 1 - e.hi);
        const float left = get(plo), right = get(phi);
        const float diff = right - left;
        r = (double)left + (double)diff * e.gamma;
        if (e.gamma >= 0.5) r = (double)right - (double)diff * (1.0 - e.gamma);
        if (r != r && n > 0) {  // +-inf samples: nanmax fallback (utl:552-554)
          if (!BOT) r = (double)get(0);
          else if (n <= 16) r = (double)get(n - 1);
          else needmax = true;  // -inf at the selected ranks: the largest sample is not among the 16 smallest
        }
      }
      if (BOT && __any(needmax ? 1 : 0)) {
        const float wm = pdoy_window_nanmax(d, 5, lane, nyears, ndoy, tbase, vmap, Tv, T, x, st, active ? c : C - 1);
        if (needmax) r = (double)wm;
      }
      if (active) out[((int64_t)j * ndoy + d) * C + c] = r;
    }
  };

  const int d0 = blockIdx.y * chunk;  // even (host)
  int d1 = d0 + chunk;
  if (d1 > ndoy) d1 = ndoy;
#pragma unroll
  for (int i = 0; i < 16; ++i) Qp[i] = Pp[i] = La[i] = Lb[i] = SENT;
  // Iteration e (even) takes L_e and L_{e+1} and completes window_{e-2} and window_{e-1}; the two iterations before the
  // chunk's first window only build the state.
  {
    uint32_t alo, ahi;
    resolve(pdoy_row_fetch(lane, nyears, ndoy, d0 - 2, tbase), alo, ahi);
    gather(alo, ahi);
    resolve(pdoy_row_fetch(lane, nyears, ndoy, d0 - 1, tbase), nlo, nhi);
    tbv = pdoy_row_fetch(lane, nyears, ndoy, d0, tbase);
  }
  for (int e = d0 - 2; e < d1 + 2; e += 2) {
    float A[16], B[16], P[16];
    int nA, nB;
    take(A, nA, e);
    {
      const int d = e - 2;  // = Q_{e-2} u L_e
      if (d >= d0 && pdoy_flag(regular, d)) select_and_store(d, Qp, A, nQp + nA);
    }
    take(B, nB, e + 1);
    if constexpr ((ABL & 2) != 0) {  // diagnostics: no merges, no selection (results wrong)
#pragma unroll
      for (int i = 0; i < 16; ++i) Qp[i] = tn_outer<BOT>(tn_outer<BOT>(Qp[i], A[i]), B[i]);
      if (e + 2 >= d1 + 2 && active) out[(int64_t)d0 * C + c] = Qp[0] + Qp[5] + Qp[15];
      continue;
    }
    tn_merge16<BOT>(P, A, B);
    const int nP = nA + nB;
    tn_merge16<BOT>(Qp, Pp, P);  // Q_e
    nQp = nPp + nP;
    {
      const int d = e - 1;  // = L_{e-3} u Q_e
      if (d >= d0 && d < d1 && pdoy_flag(regular, d)) select_and_store(d, Qp, La, nQp + nLa);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      Pp[i] = P[i];
      La[i] = Lb[i];
      Lb[i] = B[i];
    }
    nPp = nP;
    nLa = nLb;
    nLb = nB;
  }
}

// XH_ERR_NOTIMPL (no error text): not this kernel's shape — the caller falls back to k_pdoy_top16
int xh_launch_pdoy_quad(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* d_tb, int nyears,
                        int ndoy, int window, const QTab* d_tab, const int32_t* d_jmap, int nsub, int bot, double* out,
                        const int32_t* d_vmap, int64_t Tv, const uint8_t* d_reg) {
  if (window != 5 || nyears > 64 || C >= ((int64_t)1 << 29)) return XH_ERR_NOTIMPL;
  if (const char* e = xh_diag_env("XH_PDOY_QUAD"))
    if (!atoi(e)) return XH_ERR_NOTIMPL;
  const float *nanrow = nullptr, *ninf = nullptr, *pinf = nullptr;
  if (int rc = xh_const_rows(ctx, C, &nanrow, &ninf, &pinf)) return rc;
  int chunk = 92;
  if (const char* e = xh_diag_env("XH_PDOY_CHUNK")) chunk = atoi(e) > 0 ? atoi(e) : chunk;
  chunk += chunk & 1;  // the pairs live on the even grid
  const char* ea = xh_diag_env("XH_PDOY_ABL");  // diagnostics only (results become wrong)
  const int abl = ea ? atoi(ea) : 0;
  const char* el = xh_diag_env("XH_PDOY_LDSPAD");  // diagnostics: unused dynamic LDS per wave, caps the waves per CU
  const size_t ldspad = el ? (size_t)atoi(el) : 0;
  const dim3 grid((unsigned)cdiv64(C, 64), (unsigned)((ndoy + chunk - 1) / chunk));
#define XH_QUAD_(NY, B, A)                                                                                                   \
  hipLaunchKernelGGL((k_pdoy_quad<NY, B, A>), grid, dim3(64), ldspad, ctx->stream, x, T, C, st, d_tb, nyears, ndoy, chunk, d_tab, \
                     d_jmap, nsub, out, d_vmap, Tv, d_reg, nanrow, B ? pinf : ninf)
#define XH_QUAD(NY, B) XH_QUAD_(NY, B, 0)
  if (abl && nyears <= 32 && !bot) {
    if (abl == 1) XH_QUAD_(32, false, 1); else if (abl == 2) XH_QUAD_(32, false, 2); else XH_QUAD_(32, false, 3);
  } else if (nyears <= 32) {
    if (bot) XH_QUAD(32, true); else XH_QUAD(32, false);
  } else {
    if (bot) XH_QUAD(64, true); else XH_QUAD(64, false);
  }
#undef XH_QUAD
#undef XH_QUAD_
  XH_LAUNCH_CHECK();
  return XH_OK;
}
