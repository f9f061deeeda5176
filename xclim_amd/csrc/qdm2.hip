// qdm2.hip — QuantileDeltaMapping.adjust, interp = "nearest", on ONE-YEAR daily series (360 <= T <= 366), time-major, read and
// written in place (xsdba._adjustment.qdm_adjust: rank(sim, pct=True) -> interp_on_quantiles(nearest) -> apply_correction;
// not in the reference tree: PARITY UNPINNED, oracle/sdba.py qdm_adjust restates it with scipy.stats.rankdata + interp1d).
//
// With nearest-node interpolation the factor of a sample depends only on which of <= nq + 2 classes its percentage rank
// falls in (below the first node | nearest to node j | above the last node), the percentage rank is a non-decreasing
// function of the sample's doubled average rank r2 = 2·below + equal + 1, and r2 is a non-decreasing function of the VALUE.
// So a column needs no rank per sample: per class boundary the smallest r2 that passes (fp64, the operation order of
// qdm.hip / numpy), the order statistic that first reaches it, and then every sample is classified by <= 6 compares against
// the column's <= nq + 1 cut values.
//
//   1    the column in REGISTERS (two lanes per column, 183 keys each, as select3.hip); the valid count n = T - #NaN falls
//        out of the key conversion
//   3-5  sort by the comparator networks of select3.hip (local sort, split across the lane pair, merge)
//   1b   smallest key = rank 0, its copies cnt0 (dry days: exact): 1 when rank 1 differs, else counted
//   2    per (column, boundary): R = min { r2 : test(pct(r2)) } (qdmrank.h) — looked up in a table by cnt0 when the column has
//        T valid samples and all nq nodes (k_qdm_rank_table, round 5), else an analytic guess + exact verification in fp64
//        (32 of the 183 registers wait in LDS meanwhile) — then the rank p = ceil((R - 2) / 2) (inside the minimum's run:
//        rank 0 if its r2 = cnt0 + 1 passes, else rank cnt0)
//   6-7  the <= nq + 1 order statistics leave the registers through the LDS hand-over of select3.hip (one monotone pass).
//        A pick with an equal neighbour sits in a run [a, b) of equal keys: the cut is that value if a + b + 1 >= R, else
//        the value at rank b (round 5; until then every such column went to the exact-rank kernel).  Runs that cross a
//        hand-over chunk and copies of the MAXIMUM (they change the rank scale of the whole column) still put the column
//        on a list for k_qdm_columns.
//   8    the cut values and class factors leave for k_cut_classify (one streaming pass: read, classify by a branch-free
//        binary search, correct, store); XH_QDM_SPLIT=0: this kernel re-reads its tile and classifies itself.
// HBM traffic: sim twice, scen once, the tables (2 nq + 3 floats per column) once each way.
#include <stdio.h>
#include <stdlib.h>

#include "common.h"
#include "rowstream.h"
#include "qdmrank.h"
#include "sortnet_183.h"

namespace {

constexpr uint32_t PADK = 0xFFFFFFFFu;
constexpr int QR_MAXQ = 36;          // quantile nodes (LDS budget of two workgroups per CU)
constexpr uint32_t QR_SENT = 0x7FFFu;  // rank that no register index ever matches (stored as u16)
constexpr uint32_t QR_OK = 0u, QR_NAN = 1u, QR_TIES = 2u;

__device__ __forceinline__ uint32_t qr_swap1(uint32_t v) {  // partner lane (lane ^ 1): DPP quad_perm [1,0,3,2]
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}
__device__ __forceinline__ void qr_fence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// LDS words per wave for nq nodes: status arrays [4][32] | idx bytes [nq][32] | factor table [nq + 2][32] (its start holds
// the u16 ranks [nq + 1][32] until the picks are done) | cut values [nq + 2][32] (row nq + 1 = NaN) | hand-over buffer [32][64]
__host__ __device__ constexpr int qr_words_per_wave(int nq) {
  return 4 * 32 + (nq * 32 + 3) / 4 + 2 * (nq + 2) * 32 + 32 * 64;
}

template <int N, int TMIN>
__global__ void __launch_bounds__(256, 2)
k_qdm_regsort(const float* __restrict__ x, int T, int64_t C, int64_t st, const float* __restrict__ af, int64_t af_qs,
              const double* __restrict__ qnodes, int nq, int kind, int extrap, float* __restrict__ out, int64_t ost,
              uint32_t* __restrict__ flist, uint32_t* __restrict__ nflag, int abl, float* __restrict__ gcut,
              float* __restrict__ gfac, const uint16_t* __restrict__ rtab, int rtab_lds) {
  // rtab_lds: the rank table fits in LDS behind the waves' regions
  // rtab != nullptr: the boundary ranks R of a column with T valid samples, a single maximum and nq valid nodes, by the copies
  // of its minimum ([T][nq + 1], k_qdm_rank_table): a wave whose 32 columns are all of that kind looks them up
  // gcut != nullptr: the tables leave for k_cut_classify (cut values [nq + 1][C], class factors [nq + 2][C]) and this kernel
  // neither re-reads nor writes the series
  const bool split = gcut != nullptr;
  static_assert(N == XH_SN_N, "sortnet header generated for another N");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* qS = reinterpret_cast<double*>(smem);  // [QR_MAXQ] quantile nodes
  uint32_t* wave0 = reinterpret_cast<uint32_t*>(qS + QR_MAXQ);
  const int tid = threadIdx.x, w = tid >> 6;
  const int ntmax = nq + 1;  // tests per column at most
  if (tid < nq) qS[tid] = qnodes[tid];
  uint16_t* rtabS = reinterpret_cast<uint16_t*>(wave0 + 4 * qr_words_per_wave(nq));
  if (rtab_lds)
    for (int i = tid; i < T * ntmax; i += 256) rtabS[i] = rtab[i];
  __syncthreads();
  uint32_t* ncol = wave0 + w * qr_words_per_wave(nq);
  uint32_t* c0col = ncol + 32;
  uint32_t* stcol = c0col + 32;
  uint32_t* nvcol = stcol + 32;
  uint8_t* idx = reinterpret_cast<uint8_t*>(nvcol + 32);                                  // [nq][32] node index of the j-th valid node
  float* FS = reinterpret_cast<float*>(reinterpret_cast<uint32_t*>(idx) + (nq * 32 + 3) / 4);  // [ntmax + 1][32], after the picks
  uint16_t* rkT = reinterpret_cast<uint16_t*>(FS);                                        // [ntmax + 1][32] until then
  uint16_t* rRT = rkT + (ntmax + 1) * 32;                                                 // [ntmax + 1][32] the boundaries' R (0: the maximum's target)
  uint32_t* vals = reinterpret_cast<uint32_t*>(FS) + (ntmax + 1) * 32;                    // [ntmax + 1][32] keys, then float cuts
  uint32_t* dump = vals + (ntmax + 1) * 32;                                               // [32][64]
  const int64_t ntiles = (C + 31) / 32;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const uint32_t strideB = (uint32_t)(st * 4), strideO = (uint32_t)(ost * 4);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)0xFFFFFFFFu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc(out, 0, (int)0xFFFFFFFFu, 0x00020000);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + w; tile < ntiles; tile += nwaves) {
    uint32_t lane = (uint32_t)tid & 63u;
    asm volatile("" : "+v"(lane));
    const int64_t col0 = tile * 32;
#define XH_DECL(i) uint32_t k##i;
    XH_SN_FOREACH(XH_DECL)
#undef XH_DECL
    {
      const uint32_t h = lane & 1u, c32 = lane >> 1;
      const int64_t col = col0 + c32;
      const int64_t colc = col < C ? col : C - 1;
      // ---- 1. loads (as select3.hip): lane A rows 0 .. N-1, lane B rows T-N .. T-1
      const uint32_t voff = (uint32_t)(colc * 4) + (h ? (uint32_t)(T - N) * strideB : 0u);
      uint32_t soff = 0u;
      asm volatile("" : "+s"(soff));
#define XH_LD(i) k##i = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0); soff += strideB;
      XH_SN_FOREACH(XH_LD)
#undef XH_LD
      uint32_t bcut = h ? (uint32_t)(2 * N - T) : 0u;
      asm volatile("" : "+v"(bcut));
      // keys of x + 0.0f: -0.0 and +0.0 tie (rankdata); NaN / the rows lane A holds too -> pad
      uint32_t nanc = 0u;  // NaN among my rows (lane B: not those lane A holds too)
#define XH_CV(i)                                                           \
  {                                                                        \
    const float f_ = __uint_as_float(k##i) + 0.0f;                         \
    const uint32_t u_ = __float_as_uint(f_);                               \
    uint32_t kk_ = u_ ^ ((uint32_t)((int32_t)u_ >> 31) | 0x80000000u);    \
    if (i < 2 * N - TMIN) {                                                \
      kk_ = (f_ != f_) ? PADK : kk_;                                       \
      kk_ = ((uint32_t)i < bcut) ? PADK : kk_;                             \
      nanc += (f_ != f_ && !((uint32_t)i < bcut)) ? 1u : 0u;               \
    } else {                                                               \
      /* NaN -> pad, counted by the same compare (the mask is consumed at once: no SGPR pair lives on) */ \
      asm volatile("v_cmp_u_f32 vcc, %2, %2\n\tv_cndmask_b32_e64 %0, %0, -1, vcc\n\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" \
                   : "+v"(kk_), "+v"(nanc) : "v"(f_) : "vcc");            \
    }                                                                      \
    k##i = kk_;                                                            \
  }
      XH_SN_FOREACH(XH_CV)
#undef XH_CV
      // valid count of the column: leaves for LDS now (no register lives through the sort for it)
      {
        const uint32_t nn_ = (uint32_t)T - (nanc + qr_swap1(nanc));
        if (h == 0u) ncol[c32] = nn_;
      }
    }
    // ---- 3. local sort
#define XH_CE(i, j)                               \
  {                                               \
    const uint32_t a_ = k##i, b_ = k##j;          \
    k##i = a_ < b_ ? a_ : b_;                     \
    k##j = a_ < b_ ? b_ : a_;                     \
  }
    if (!(abl & 4)) {
    XH_SN_SORT(XH_CE)
    }
    uint32_t lane3 = (uint32_t)tid & 63u;
    asm volatile("" : "+v"(lane3));
    const uint32_t mA3 = (lane3 & 1u) ? 0u : 0xFFFFFFFFu;
    // ---- 4. lane A negates, bitonic split across the lane pair (one asm statement per register pair, see select3.hip)
#define XH_DPP_SWAP1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define XH_SP(i, j)                                                                                              \
  {                                                                                                              \
    uint32_t t0_, t1_;                                                                                           \
    asm volatile(                                                                                                \
        "v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\ts_nop 1\n\t"                                            \
        "v_not_b32_dpp %2, %1 " XH_DPP_SWAP1 "\n\tv_not_b32_dpp %3, %0 " XH_DPP_SWAP1 "\n\t"                    \
        "v_max_u32 %0, %0, %2\n\tv_max_u32 %1, %1, %3"                                                           \
        : "+v"(k##i), "+v"(k##j), "=&v"(t0_), "=&v"(t1_)                                                         \
        : "v"(mA3));                                                                                             \
  }
#define XH_SM(m)                                                                          \
  {                                                                                       \
    uint32_t t0_;                                                                         \
    asm volatile("v_xor_b32 %0, %0, %2\n\ts_nop 1\n\tv_not_b32_dpp %1, %0 " XH_DPP_SWAP1   \
                 "\n\tv_max_u32 %0, %0, %1"                                               \
                 : "+v"(k##m), "=&v"(t0_)                                                 \
                 : "v"(mA3));                                                             \
  }
    if (!(abl & 4)) {
    XH_SN_SPLIT(XH_SP, XH_SM)
    }
#undef XH_SP
#undef XH_SM
    // ---- 5. merge: rank r of the column sits in A at k[N-1-r] (negated) for r < N, in B at k[r-N] otherwise
    if (!(abl & 4)) {
    XH_SN_MERGE(XH_CE)
    }
#undef XH_CE
    uint32_t lane1 = (uint32_t)tid & 63u;
    asm volatile("" : "+v"(lane1));
    uint32_t* mydump = dump + lane1;
#define XH_DW(e, reg) mydump[(e) * 64] = reg;
#define QR_DUMP_CHUNK(ci)                   \
  switch (ci) {                             \
    case 0: XH_SN_DUMP_0(XH_DW) break;      \
    case 1: XH_SN_DUMP_1(XH_DW) break;      \
    case 2: XH_SN_DUMP_2(XH_DW) break;      \
    case 3: XH_SN_DUMP_3(XH_DW) break;      \
    case 4: XH_SN_DUMP_4(XH_DW) break;      \
    default: XH_SN_DUMP_5(XH_DW) break;     \
  }
    static_assert(XH_SN_CHUNKS == 6 && N > 160 && N <= 192, "six hand-over chunks of 32 registers");
    // ---- 1b. smallest key and its copies, from the SORTED column: rank 0 is lane A's k[N-1]; a column whose rank 1 differs has
    //      one copy (every column of a continuous field), else the copies are counted through the hand-over buffer in a
    //      ROLLED loop (dry days; the kernel has to stay inside the 64 KB instruction cache: sort networks 32 KB)
    if (!(abl & 1)) {
      const uint32_t h = lane1 & 1u, c32 = lane1 >> 1;
      const uint32_t mAs = h ? 0u : 0xFFFFFFFFu;
      const uint32_t n = ncol[c32];
      uint32_t kmin = k182 ^ mAs;
      const uint32_t pk = qr_swap1(kmin);
      kmin = h ? pk : kmin;
      static_assert(N == 183, "rank 0 / rank 1 of the column: k182 / k181 of lane A");
      const bool more = !h && (k181 ^ mAs) == kmin && kmin != PADK;
      uint32_t cml = h ? 0u : 1u;
      if (__any(more ? 1 : 0)) {
        cml = 0u;
#pragma nounroll
        for (int ci = 0; ci < 6; ++ci) {
          QR_DUMP_CHUNK(ci)
          const int cnt = ci < 5 ? 32 : N - 160;
#pragma unroll 8
          for (int e = 0; e < cnt; ++e) cml += ((mydump[e * 64] ^ mAs) == kmin) ? 1u : 0u;
        }
      }
      const uint32_t cnt0 = cml + qr_swap1(cml);
      // nothing valid / all valid samples equal: 0 / 0 ranks -> NaN (qdm.hip)
      if (h == 0) { c0col[c32] = cnt0; stcol[c32] = (n == 0u || cnt0 >= n) ? QR_NAN : QR_OK; }
    }
    if (abl & 3) {  // diagnostics only: sane tables for the phases that still run
      if (abl & 1) { if (lane1 < 32u) { ncol[lane1] = (uint32_t)T; c0col[lane1] = 1u; stcol[lane1] = QR_OK; } }
      if (abl & 2) {
        for (int i = (int)lane1; i < (ntmax + 1) * 32; i += 64) { rkT[i] = (uint16_t)QR_SENT; vals[i] = PADK; }
        if (lane1 < 32u) nvcol[lane1] = 2u;
        for (int i = (int)lane1; i < nq * 32; i += 64) idx[i] = 0;
      }
      qr_fence();
    }
    // ---- 2. ranks of the class boundaries.  k0 .. k31 wait in the hand-over buffer while the fp64 arithmetic runs.
    if (!(abl & 2)) {
      QR_DUMP_CHUNK(0)
      qr_fence();
      // ---- 2a. nodes of the column: drop the NaN factors (interp_on_quantiles masks them); lane c < 32 owns column c
      if (lane1 < 32u) {
        const int64_t cc = col0 + lane1 < C ? col0 + lane1 : C - 1;
        uint32_t cnt = 0;
        // 8 factors in flight (one load per iteration would pay a full memory latency nq times)
#pragma nounroll
        for (int j0 = 0; j0 < nq; j0 += 8) {
          float a[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) a[u] = af[(int64_t)(j0 + u < nq ? j0 + u : nq - 1) * af_qs + cc];
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (j0 + u < nq && a[u] == a[u]) { idx[cnt * 32 + lane1] = (uint8_t)(j0 + u); ++cnt; }
        }
        nvcol[lane1] = cnt;
        if (cnt < 2u && stcol[lane1] == QR_OK) stcol[lane1] = QR_NAN;  // fewer than two nodes: nothing to interpolate on
      }
      qr_fence();
      // ---- 2b. per (column, boundary) the rank of the order statistic that opens the class.  Lane: column lane & 31, the
      //      even (lanes < 32) or odd boundaries.
      {
        const uint32_t c = lane1 & 31u;
        const uint32_t nn = ncol[c], c0 = c0col[c], nvn = nvcol[c];
        const bool ok = stcol[c] == QR_OK;
        const bool lut = rtab != nullptr && __all((!ok || (nn == (uint32_t)T && nvn == (uint32_t)nq)) ? 1 : 0);
        if (lut) {
          auto put = [&](uint32_t t, uint32_t R) {
            uint32_t rank = QR_SENT;
            if (ok && R <= 2u * nn) {
              const uint32_t p = qdm_pos_of_r2(R);
              rank = p < c0 ? ((c0 + 1u >= R) ? 0u : c0) : p;
              if (rank >= nn) rank = QR_SENT;
            }
            rkT[t * 32 + c] = (uint16_t)rank;
            rRT[t * 32 + c] = (uint16_t)R;
            vals[t * 32 + c] = PADK;
          };
          const uint32_t roff = (ok ? c0 - 1u : 0u) * (uint32_t)ntmax;
          if (rtab_lds) {
#pragma nounroll
            for (uint32_t t = lane1 >> 5; t < (uint32_t)ntmax; t += 2u) put(t, rtabS[roff + t]);
          } else {
            // 8 table entries in flight per round
            const uint16_t* row = rtab + roff;
#pragma nounroll
            for (uint32_t t0 = lane1 >> 5; t0 < (uint32_t)ntmax; t0 += 16u) {
              uint32_t Rr[8];
#pragma unroll
              for (int u = 0; u < 8; ++u) { const uint32_t t = t0 + 2u * (uint32_t)u; Rr[u] = row[t < (uint32_t)ntmax ? t : (uint32_t)ntmax - 1u]; }
#pragma unroll
              for (int u = 0; u < 8; ++u) {
                const uint32_t t = t0 + 2u * (uint32_t)u;
                if (t < (uint32_t)ntmax) put(t, Rr[u]);
              }
            }
          }
        } else {
#pragma nounroll
        for (uint32_t t = lane1 >> 5; t < (uint32_t)ntmax; t += 2u) {
          uint32_t rank = QR_SENT, R = 0u;
          if (ok && t <= nvn) {
            // t = 0: pct >= x0 (not below the first node); 0 < t < nvn: pct > (x[t-1] / 2 + x[t] / 2), scipy's nearest
            // bounds; t = nvn: pct > the last node
            double thr;
            if (t == 0u) thr = qS[idx[c]];
            else if (t == nvn) thr = qS[idx[(nvn - 1u) * 32 + c]];
            else thr = qS[idx[(t - 1u) * 32 + c]] / 2.0 + qS[idx[t * 32 + c]] / 2.0;
            R = qdm_min_r2(t == 0u, thr, nn, c0, 1u);  // (a single maximum: copies of it put the column on the list)
            if (R <= 2u * nn) {
              const uint32_t p = qdm_pos_of_r2(R);  // first position whose r2 = 2p + 2 reaches R
              if (p < c0) rank = (c0 + 1u >= R) ? 0u : c0;      // inside the minimum's run [0, c0): its r2 is c0 + 1
              else rank = p;
              if (rank >= nn) rank = QR_SENT;
            }
          }
          rkT[t * 32 + c] = (uint16_t)rank;
          rRT[t * 32 + c] = (uint16_t)R;
          vals[t * 32 + c] = PADK;  // a boundary nobody reaches decodes to NaN: every compare fails
        }
        }
        // one more target, for the tie test only: the maximum (rank n - 1).  Copies of the maximum change mx, i.e. the
        // percentage rank of EVERY sample, wherever the class boundaries lie.
        if (lane1 < 32u) {
          vals[ntmax * 32 + lane1] = PADK;
          rkT[ntmax * 32 + lane1] = (uint16_t)(ok && nn >= 1u ? nn - 1u : QR_SENT);
          rRT[ntmax * 32 + lane1] = 0;
        }
      }
      qr_fence();
#define XH_UP(e, reg) reg = mydump[(e) * 64];
      XH_SN_DUMP_0(XH_UP)
#undef XH_UP
    }
    uint32_t lane4 = (uint32_t)tid & 63u;
    asm volatile("" : "+v"(lane4));
    const uint32_t h = lane4 & 1u, c32 = lane4 >> 1;
    const uint32_t mA = h ? 0u : 0xFFFFFFFFu;
    const int64_t mycol = col0 + c32;
    // ---- 6 + 7. picks: one monotone sequence of targets — B (ranks >= N) walks its targets upwards, A downwards — and the
    //      tie test: a picked key that equals one of its two neighbours in the sorted column belongs to a run the kernel
    //      knows nothing about (unless it is the minimum's: rank < cnt0) -> the column goes on the list.  The neighbours are
    //      the hand-over entries next to it, at a chunk's edges the adjacent register, across the lane pair the partner's k0.
    if (!(abl & 16)) {
      const uint32_t c0 = c0col[c32];
      // my targets: ranks < N live in lane A, ranks in [N, 2N) in lane B; unreachable boundaries carry the sentinel and are
      // skipped (they may lie between the last boundary and the maximum's target)
      const int dj = h ? 1 : -1;
      auto next_mine = [&](int j) -> int {
        while (j >= 0 && j <= ntmax) {
          const uint32_t r = rkT[j * 32 + c32];
          if (h ? (r >= (uint32_t)N && r != QR_SENT) : r < (uint32_t)N) break;
          j += dj;
        }
        return j;
      };
      auto local_of = [&](int j) -> uint32_t {
        if (j < 0 || j > ntmax) return 0xFFFFFFFFu;
        const uint32_t r = rkT[j * 32 + c32];
        return h ? r - (uint32_t)N : (uint32_t)(N - 1) - r;
      };
      int jcur = next_mine(h ? 0 : ntmax);
      uint32_t snext = local_of(jcur);
      uint32_t tied = 0u;
      const uint32_t partner0 = qr_swap1(k0 ^ mA);  // the partner lane's k0 as a real key: my local index -1
#pragma nounroll
      for (uint32_t ci = 0; ci < 6u; ++ci) {
        QR_DUMP_CHUNK(ci)
        const uint32_t cnt_ = ci < 5u ? 32u : (uint32_t)(N - 160);
        // lane-local neighbours of the chunk's edges (real keys)
        const uint32_t lo_edge_ = ci == 0u ? partner0 : (ci == 1u ? k31 : ci == 2u ? k63 : ci == 3u ? k95 : ci == 4u ? k127 : k159) ^ mA;
        const uint32_t hi_edge_ = ci == 5u ? PADK : (ci == 0u ? k32 : ci == 1u ? k64 : ci == 2u ? k96 : ci == 3u ? k128 : k160) ^ mA;
        while ((snext >> 5) == ci) {
          const uint32_t e_ = snext & 31u;
          uint32_t v_ = mydump[e_ * 64] ^ mA;
          const uint32_t lo_ = e_ > 0u ? mydump[(e_ - 1u) * 64] ^ mA : lo_edge_;
          const uint32_t hi_ = e_ + 1u < cnt_ ? mydump[(e_ + 1u) * 64] ^ mA : hi_edge_;
          if ((lo_ == v_ || hi_ == v_) && rkT[jcur * 32 + c32] >= c0) {
            // a run of equal keys [a, b) around the pick: the cut is this value if its r2 = a + b + 1 reaches R, else the
            // value at rank b (qdmrank.h).  Runs that leave the chunk and copies of the maximum (which change the ranks of
            // every sample) go to the exact-rank kernel.
            const uint32_t R_ = rRT[jcur * 32 + c32];
            uint32_t s_lo = e_, s_hi = e_ + 1u;
            while (s_lo > 0u && (mydump[(s_lo - 1u) * 64] ^ mA) == v_) --s_lo;
            while (s_hi < cnt_ && (mydump[s_hi * 64] ^ mA) == v_) ++s_hi;
            const uint32_t below_ = s_lo > 0u ? mydump[(s_lo - 1u) * 64] ^ mA : lo_edge_;
            const uint32_t above_ = s_hi < cnt_ ? mydump[s_hi * 64] ^ mA : hi_edge_;
            if (R_ == 0u || below_ == v_ || above_ == v_ || (abl & 64)) tied = 1u;
            else {
              const uint32_t L0 = ci * 32u + s_lo, L1 = ci * 32u + s_hi;  // local [L0, L1): ranks N + L (lane B), N - 1 - L (lane A)
              const uint32_t a_ = h ? (uint32_t)N + L0 : (uint32_t)N - L1, b_ = h ? (uint32_t)N + L1 : (uint32_t)N - L0;
              if (a_ + b_ + 1u < R_) v_ = h ? above_ : below_;
            }
          }
          vals[jcur * 32 + (int)c32] = v_;
          jcur = next_mine(jcur + dj);
          snext = local_of(jcur);
        }
      }
      if (tied && stcol[c32] == QR_OK) stcol[c32] = QR_TIES;
    }
    qr_fence();
    // ---- 8a. the tile's rows once more, all in flight (the sorted keys are dead: same registers, same loads as step 1; L2 /
    //      Infinity Cache serve most of them) — issued before the tables below are built
    const int64_t colc8 = mycol < C ? mycol : C - 1;
    if (!(abl & 32) && !split) {
      const uint32_t voff = (uint32_t)(colc8 * 4) + (h ? (uint32_t)(T - N) * strideB : 0u);
      uint32_t soff = 0u;
      asm volatile("" : "+s"(soff));
#define XH_LD(i) k##i = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0); soff += strideB;
      XH_SN_FOREACH(XH_LD)
#undef XH_LD
    }
    // ---- 8b. cut values as floats; factor table of the column's classes: [0] below the first node, [k] node k - 1,
    //      [nvn + 1] above the last node; columns that are all-NaN or on the tie list get NaN everywhere
    for (int i = (int)lane4; i < (ntmax + 1) * 32; i += 64) vals[i] = __float_as_uint(xh_key2f(vals[i]));
    {
      // lane: column lane & 31, the even (lanes < 32) or odd classes; 8 factor loads in flight
      const uint32_t c = lane4 & 31u, hf = lane4 >> 5;
      const int64_t cc = col0 + c < C ? col0 + c : C - 1;
      const uint32_t nvn = nvcol[c], stt = stcol[c];
      const float nanv = xh_nan32();
#pragma nounroll
      for (uint32_t k0_ = hf; k0_ <= (uint32_t)ntmax; k0_ += 16u) {
        float a[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const uint32_t k = k0_ + 2u * (uint32_t)u;  // class k: node min(max(k, 1), nvn) - 1
          uint32_t node = k < 1u ? 0u : k - 1u;
          node = node < nvn ? node : (nvn > 0u ? nvn - 1u : 0u);
          const uint32_t jn = nvn > 0u ? idx[node * 32 + c] : 0u;  // (no valid node: the table entry is never read as one)
          a[u] = af[(int64_t)jn * af_qs + cc];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const uint32_t k = k0_ + 2u * (uint32_t)u;
          if (k <= (uint32_t)ntmax) {
            const bool inner = k >= 1u && k <= nvn;  // below the first / above the last node: the end factor or NaN
            FS[k * 32 + c] = (stt == QR_OK && k <= nvn + 1u && (inner || extrap == 0)) ? a[u] : nanv;
          }
        }
      }
      if (lane4 < 32u && stt == QR_TIES && col0 + c < C) flist[atomicAdd(nflag, 1u)] = (uint32_t)(col0 + c);
    }
    qr_fence();
    if (split) {  // tables -> global, 32 columns x 4 bytes per row
      for (int i = (int)lane4; i < ntmax * 32; i += 64) {
        const int64_t cg = col0 + (i & 31);
        if (cg < C) gcut[(int64_t)(i >> 5) * C + cg] = __uint_as_float(vals[i]);
      }
      for (int i = (int)lane4; i < (ntmax + 1) * 32; i += 64) {
        const int64_t cg = col0 + (i & 31);
        if (cg < C) gfac[(int64_t)(i >> 5) * C + cg] = FS[i];
      }
    }
    // ---- 8c. classify, correct, store.  The rows leave the registers through the hand-over buffer (a rolled loop needs a
    //      run-time row index); class = number of cuts <= x (cuts non-decreasing, unreachable ones NaN at the end): a
    //      branch-free lower bound, 8 rows level by level (8 independent LDS reads in flight instead of one dependent
    //      chain per sample)
    if (!(abl & 32) && !split) {
      const uint32_t voffO = (uint32_t)(colc8 * 4) + (h ? (uint32_t)(T - N) * strideO : 0u);
      const uint32_t bc = h ? (uint32_t)(2 * N - T) : 0u;  // B's first rows are A's
      const float* cuts = reinterpret_cast<const float*>(vals) + c32;
      const float* fac = FS + c32;
      const bool colok = mycol < C;
      constexpr int QB = 8;
#pragma nounroll
      for (int ci = 0; ci < 6; ++ci) {
        QR_DUMP_CHUNK(ci)
        const int cnt = ci < 5 ? 32 : N - 160;
#pragma nounroll
        for (int e0 = 0; e0 < cnt; e0 += QB) {
          float v[QB];
#pragma unroll
          for (int u = 0; u < QB; ++u) v[u] = __uint_as_float(mydump[(e0 + u < cnt ? e0 + u : cnt - 1) * 64]);
          uint32_t base[QB];
#pragma unroll
          for (int u = 0; u < QB; ++u) base[u] = 0u;
          int len = ntmax;
#pragma nounroll
          while (len > 1) {
            const int half = len >> 1;
            float cv[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u) cv[u] = cuts[(base[u] + (uint32_t)half - 1u) * 32];
            asm volatile("" ::: "memory");  // (all QB reads issued before the first compare waits)
#pragma unroll
            for (int u = 0; u < QB; ++u) base[u] += (v[u] >= cv[u]) ? (uint32_t)half : 0u;
            len -= half;
          }
          {
            float cv[QB];
#pragma unroll
            for (int u = 0; u < QB; ++u) cv[u] = cuts[base[u] * 32];
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < QB; ++u) base[u] += (v[u] >= cv[u]) ? 1u : 0u;
          }
          float a[QB];
#pragma unroll
          for (int u = 0; u < QB; ++u) a[u] = fac[base[u] * 32];
#pragma unroll
          for (int u = 0; u < QB; ++u) {
            const int i = ci * 32 + e0 + u;
            const float r = kind == 0 ? v[u] + a[u] : (kind == 1 ? v[u] * a[u] : a[u]);
            if (e0 + u < cnt && (uint32_t)i >= bc && colok)
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), rsrcO, (int)voffO, (int)((uint32_t)i * strideO), 0);
          }
        }
      }
    }
#undef QR_DUMP_CHUNK
#undef XH_DW
    qr_fence();  // the per-wave tables are rewritten by the next tile
  }
}

// Boundary ranks of a column with n = T valid samples, a single maximum and all nq nodes valid, by the copies c0 of its minimum:
// tab[(c0 - 1) * (nq + 1) + t] = R (qdmrank.h; 2 T + 1 when no rank passes).  Nearly every column of a field without missing
// values is of that kind, and the fp64 search per (column, boundary) was a third of k_qdm_regsort.
__global__ void __launch_bounds__(XH_BLOCK)
k_qdm_rank_table(int T, int nq, const double* __restrict__ qnodes, uint16_t* __restrict__ tab) {
  const int ntmax = nq + 1;
  const int64_t i = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (i >= (int64_t)T * ntmax) return;
  const uint32_t c0 = (uint32_t)(i / ntmax) + 1u, t = (uint32_t)(i % ntmax);
  double thr;
  if (t == 0u) thr = qnodes[0];
  else if (t == (uint32_t)nq) thr = qnodes[nq - 1];
  else thr = qnodes[t - 1u] / 2.0 + qnodes[t] / 2.0;
  tab[i] = (uint16_t)qdm_min_r2(t == 0u, thr, (uint32_t)T, c0, 1u);
}

// ---- classification against per-column cut values (the second half of the QDM path when it is split in two kernels) ----
// scen[t, c] = sim[t, c] (+|*) fac[#{k : cut[k, c] <= sim[t, c]}, c]; cut (ntest, C) non-decreasing per column with NaN for the
// boundaries no sample reaches (every compare fails), fac (ntest + 1, C).  A workgroup = 64 columns x 4 row lanes; the
// tile's two tables sit in LDS; 16 rows per lane in flight (xh_row_stream), their searches run level by level.
constexpr int CC_CW = 64, CC_RL = 4, CC_U = 16, CC_NT = CC_CW * CC_RL;
__global__ void __launch_bounds__(CC_NT)
k_cut_classify(const float* __restrict__ x, int T, int64_t C, int64_t st, const float* __restrict__ gcut,
               const float* __restrict__ gfac, int ntest, int kind, float* __restrict__ out, int64_t ost) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* cutS = reinterpret_cast<float*>(smem);  // [ntest][64]
  float* facS = cutS + ntest * CC_CW;            // [ntest + 1][64]
  const int tid = threadIdx.x, col = tid & (CC_CW - 1), rl = tid / CC_CW;
  const int64_t ntiles = (C + CC_CW - 1) / CC_CW;
  const uint32_t rowstepO = (uint32_t)(ost * 4 * CC_RL);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t c = tile * CC_CW + col;
    const bool cvalid = c < C;
    const int64_t cc = cvalid ? c : C - 1;
    for (int k = rl; k <= ntest; k += CC_RL) {
      if (k < ntest) cutS[k * CC_CW + col] = gcut[(int64_t)k * C + cc];
      facS[k * CC_CW + col] = gfac[(int64_t)k * C + cc];
    }
    __syncthreads();
    const float* cuts = cutS + col;
    const float* fac = facS + col;
    const uint32_t voffO = (uint32_t)(((int64_t)rl * ost + cc) * 4);
    xh_row_stream<CC_U, CC_RL>(x, T, st, cc, rl, [&](const float (&v)[CC_U], int kb) {
      uint32_t base[CC_U];
#pragma unroll
      for (int u = 0; u < CC_U; ++u) base[u] = 0u;
      int len = ntest;
#pragma nounroll
      while (len > 1) {
        const int half = len >> 1;
        float cv[CC_U];
#pragma unroll
        for (int u = 0; u < CC_U; ++u) cv[u] = cuts[(base[u] + (uint32_t)half - 1u) * CC_CW];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < CC_U; ++u) base[u] += (v[u] >= cv[u]) ? (uint32_t)half : 0u;
        len -= half;
      }
      {
        float cv[CC_U];
#pragma unroll
        for (int u = 0; u < CC_U; ++u) cv[u] = cuts[base[u] * CC_CW];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < CC_U; ++u) base[u] += (v[u] >= cv[u]) ? 1u : 0u;
      }
      float a[CC_U];
#pragma unroll
      for (int u = 0; u < CC_U; ++u) a[u] = fac[base[u] * CC_CW];
      float* ob = out + (int64_t)kb * (CC_RL * CC_U) * ost;
      const __amdgpu_buffer_rsrc_t rsrcO = __builtin_amdgcn_make_buffer_rsrc(ob, 0, (int)0xFFFFFFFFu, 0x00020000);
      uint32_t soff = 0u;
#pragma unroll
      for (int u = 0; u < CC_U; ++u) {
        const float r = kind == 0 ? v[u] + a[u] : (kind == 1 ? v[u] * a[u] : a[u]);
        if (cvalid && kb * (CC_RL * CC_U) + u * CC_RL + rl < T)
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(r), rsrcO, (int)voffO, (int)soff, 0);
        soff += rowstepO;
      }
    });
    __syncthreads();  // the tables are rewritten by the next tile
  }
}

// flagged columns -> time-minor scratch (column f at buf + f * Tp), their factors -> (nq, nf)
__global__ void __launch_bounds__(XH_BLOCK)
k_qr_gather(const float* __restrict__ x, int64_t T, int64_t st, const uint32_t* __restrict__ flist, float* __restrict__ buf,
            int64_t Tp, const float* __restrict__ af, int64_t af_qs, int nq, int64_t nf, float* __restrict__ gaf) {
  const int64_t c = flist[blockIdx.x];
  float* dst = buf + (int64_t)blockIdx.x * Tp;
  for (int64_t t = threadIdx.x; t < T; t += XH_BLOCK) dst[t] = x[t * st + c];
  if (threadIdx.x < nq) gaf[(int64_t)threadIdx.x * nf + blockIdx.x] = af[(int64_t)threadIdx.x * af_qs + c];
}

__global__ void __launch_bounds__(XH_BLOCK)
k_qr_scatter(const float* __restrict__ buf, int64_t T, int64_t Tp, const uint32_t* __restrict__ flist, float* __restrict__ out,
             int64_t ost) {
  const int64_t c = flist[blockIdx.x];
  const float* src = buf + (int64_t)blockIdx.x * Tp;
  for (int64_t t = threadIdx.x; t < T; t += XH_BLOCK) out[t * ost + c] = src[t];
}

}  // namespace

// scen = sim (+|*) fac[class], class = number of cut values <= sim: the streaming second half of the QDM "nearest" path, also
// behind the two-pass histogram selection of select4.hip (xh_qdm_hist).  gcut (ntest, C), gfac (ntest + 1, C).
int xh_cut_classify(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, const float* gcut, const float* gfac, int ntest,
                    int kind, float* scen, int64_t ost) {
  const size_t lds2 = sizeof(float) * (size_t)(2 * ntest + 1) * CC_CW;
  const int64_t ct = cdiv64(C, CC_CW);
  const int64_t cg = ct < (int64_t)ctx->num_cu * 8 ? ct : (int64_t)ctx->num_cu * 8;
  hipLaunchKernelGGL(k_cut_classify, dim3((unsigned)cg), dim3(CC_NT), lds2, ctx->stream, sim, (int)T, C, st, gcut, gfac, ntest, kind, scen,
                     ost);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

// XH_OK: scen written.  XH_ERR_NOTIMPL (no error text): not this kernel's shape, the caller runs the exact-rank pipeline.
int xh_qdm_regsort(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, const float* af, int64_t af_qs,
                   const double* d_q, int nq, int kind, int extrap, float* scen, int64_t ost) {
  constexpr int N = XH_SN_N, TMIN = 360;
  if (T < TMIN || T > 2 * N || nq < 2 || nq > QR_MAXQ || C < 1) return XH_ERR_NOTIMPL;
  if (scen == sim) return XH_ERR_NOTIMPL;  // (the tied columns are re-read from sim after scen has been written)
  if ((unsigned long long)T * (unsigned long long)st * 4ull >= (1ull << 32) ||
      (unsigned long long)T * (unsigned long long)ost * 4ull >= (1ull << 32) || C >= ((int64_t)1 << 32))
    return XH_ERR_NOTIMPL;
  if (xh_diag_env("XH_QDM_NOREGSORT")) return XH_ERR_NOTIMPL;  // A/B against the exact-rank kernel
  size_t lds = sizeof(double) * QR_MAXQ + 4 * sizeof(uint32_t) * (size_t)qr_words_per_wave(nq);
  if (lds > 79 * 1024) return XH_ERR_NOTIMPL;  // two workgroups per CU
  const int64_t ntiles = (C + 31) / 32;
  int64_t nblk = (ntiles + 3) / 4;
  const int64_t maxblk = (int64_t)ctx->num_cu * 2;
  if (nblk > maxblk) nblk = maxblk;
  // scratch: counter | tie list (at most C entries) | room for the tied columns' detour through the exact-rank kernel
  // (time-minor copies in and out + their factors), capped at C / 8 columns — more ties than that: exact ranks for everything
  const int64_t Tp = (T + 63) & ~(int64_t)63;
  int64_t nfmax = C / 8 > 4096 ? C / 8 : 4096;
  if (nfmax > C) nfmax = C;
  const size_t b_list = sizeof(uint32_t) * (((size_t)C + 4 + 63) & ~(size_t)63);
  const size_t b_cols = sizeof(float) * (size_t)nfmax * (size_t)Tp;
  void* ws = nullptr;
  const char* esp = xh_diag_env("XH_QDM_SPLIT");  // diagnostics: 0 = one kernel that also classifies and writes (re-reads its tile)
  const bool split = !(esp && atoi(esp) == 0);
  const size_t b_tab = split ? sizeof(float) * (size_t)(2 * nq + 3) * (size_t)C : 0;
  const size_t b_rtab = (sizeof(uint16_t) * (size_t)T * (size_t)(nq + 1) + 63) & ~(size_t)63;
  int rc = xh_big_scratch(ctx, b_rtab + b_list + 2 * b_cols + sizeof(float) * (size_t)nq * (size_t)nfmax + b_tab, &ws);
  if (rc) return rc;
  uint16_t* rtab = static_cast<uint16_t*>(ws);
  uint32_t* nflag = reinterpret_cast<uint32_t*>(static_cast<char*>(ws) + b_rtab);
  uint32_t* flist = nflag + 4;
  float* gbuf = reinterpret_cast<float*>(static_cast<char*>(ws) + b_rtab + b_list);
  float* gout = gbuf + (size_t)nfmax * (size_t)Tp;
  float* gaf = gout + (size_t)nfmax * (size_t)Tp;
  float* gcut = split ? gaf + (size_t)nq * (size_t)nfmax : nullptr;  // (nq + 1, C)
  float* gfac = split ? gcut + (size_t)(nq + 1) * (size_t)C : nullptr;  // (nq + 2, C)
  XH_CHECK_HIP(hipMemsetAsync(nflag, 0, 16, ctx->stream));
  const char* ea = xh_diag_env("XH_QDM_ABL");  // diagnostics: skip phases (1 stats, 2 ranks, 4 sort, 16 picks, 32 apply; wrong results)
  const int abl = ea ? atoi(ea) : 0;
  const bool nolut = xh_diag_env("XH_QDM_NOLUT") != nullptr, norun = xh_diag_env("XH_QDM_NORUN") != nullptr;  // A/B of round 5
  if (!nolut) {
    const int64_t ne = T * (int64_t)(nq + 1);
    hipLaunchKernelGGL(k_qdm_rank_table, dim3((unsigned)cdiv64(ne, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, (int)T, nq, d_q, rtab);
  }
  auto kern = k_qdm_regsort<N, TMIN>;
  const size_t b_lut = sizeof(uint16_t) * (size_t)T * (size_t)(nq + 1);
  const int rtab_lds = (!nolut && lds + b_lut <= 79 * 1024 && !xh_diag_env("XH_QDM_LUT_GLOBAL")) ? 1 : 0;
  if (rtab_lds) lds += b_lut;
  if (lds > 48 * 1024) XH_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, ctx->stream, sim, (int)T, C, st, af, af_qs, d_q, nq, kind, extrap,
                     scen, ost, flist, nflag, abl | (norun ? 64 : 0), gcut, gfac, nolut ? (const uint16_t*)nullptr : rtab, rtab_lds);
  XH_LAUNCH_CHECK();
  if (split && !(abl & 32)) {
    rc = xh_cut_classify(ctx, sim, T, C, st, gcut, gfac, nq + 1, kind, scen, ost);
    if (rc) return rc;
  }
  uint32_t nf = 0;
  XH_CHECK_HIP(hipMemcpyAsync(&nf, nflag, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  XH_CHECK_HIP(hipStreamSynchronize(ctx->stream));
  if (xh_diag_env("XH_QDM_STATS")) fprintf(stderr, "[xh_qdm_regsort] T=%lld C=%lld nq=%d: %u columns with ties -> exact-rank kernel\n", (long long)T, (long long)C, nq, nf);
  if (nf == 0 || abl) return XH_OK;
  if ((int64_t)nf > nfmax) return XH_ERR_NOTIMPL;  // a quantised field: the caller ranks every column exactly
  // columns with ties: exact average ranks (k_qdm_columns, time-minor input: it does not touch the big scratch) on a copy
  hipLaunchKernelGGL(k_qr_gather, dim3(nf), dim3(XH_BLOCK), 0, ctx->stream, sim, T, st, flist, gbuf, Tp, af, af_qs, nq, (int64_t)nf, gaf);
  rc = xh_qdm_columns(ctx, gbuf, T, (int64_t)nf, Tp, gaf, (int64_t)nf, d_q, nq, kind, 0, extrap, gout, Tp);
  if (rc) return rc;
  hipLaunchKernelGGL(k_qr_scatter, dim3(nf), dim3(XH_BLOCK), 0, ctx->stream, gout, T, Tp, flist, scen, ost);
  XH_LAUNCH_CHECK();
  return XH_OK;
}
