// select3.hip — per-column quantiles of ONE-YEAR daily series (360 <= T <= 366) with the column held in REGISTERS
// and sorted by a static comparator network (xsdba nbutils.quantile; E1 of SURVEY.md §8a, Hyndman-Fan type 7 =
// /root/reference/src/xclim/core/utils.py:370-395, 464-491 with alpha = beta = 1).
//
// Why: the histogram / counting-sort selection of select.hip is bound by LDS latency chains (1.27 ms per 365 x 1 036 800
// array, 15 % of the HBM peak).  A data-oblivious network needs no LDS, no atomics, no divergence and no cross-lane
// traffic while it sorts, and its instruction count is known: Batcher's merge-exchange for 183 keys is 2520 comparators
// = 5040 v_min_u32 / v_max_u32.
//
// Layout: the input is time-major (T, C), cells contiguous.  TWO adjacent lanes own one column: lane A (even) loads rows
// 0 .. N-1, lane B (odd) rows T-N .. T-1 (the 2N - T rows both would hold become pad keys in B), N = 183, so
// one wave instruction serves 32 columns and every load of a wave reads two full 128-byte row segments.  A column's
// 2N keys fit the 256-VGPR budget of two waves per SIMD (one wave sorts while the other waits for its loads).
//   1. load (unconditional, 32-bit lane offsets from the uniform base), order-preserving keys (NaN / pad = 0xFFFFFFFF)
//   2. each lane sorts its N keys                                   XH_SN_SORT   (tools/gen_sortnet.py)
//   3. lane A negates its keys (its order flips), then the bitonic split across the pair
//          k[i] = max(k[i], ~partner.k[N-1-i])      (v_not_b32 with a DPP quad_perm [1,0,3,2] operand + v_max_u32)
//      leaves the N smallest samples in A (negated) and the N largest in B, both as "down then up" sequences
//   4. the same ascending bitonic merge in both lanes                XH_SN_MERGE  (Lang's merge for arbitrary N)
//      -> rank r of the column sits in A at k[N-1-r] (negated) for r < N and in B at k[r-N] otherwise
//   5. the 2*nq order statistics are picked from the registers (per-lane index -> 5-level select tree over 32 registers
//      at a time) into a small LDS table, then Hyndman-Fan lerp + coalesced stores
// Columns whose valid count is neither T nor 0 (some NaN samples) have lane-dependent ranks: in such ("irregular") tiles
// the valid counts are counted and every lane evaluates its own ranks instead of reading the shared rank table.
// The tile loop is 46 KB of straight-line code (inside the 64 KB instruction cache).  Measured on MI355X, 365 x 1 036 800
// (profiles/r02/regsort_anatomy.txt): 0.78 ms per array against 1.27 ms for the histogram kernel, and VALU-bound — with
// the loads replaced by register fills it still takes 0.75 ms: sort 0.36, split + merge 0.11, picks 0.11, keys / lerp /
// stores 0.17.  v_min_u32 / v_max_u32 (like v_cmp, v_fma_f32 and every 3-operand op) issue at HALF the rate of
// v_add / v_xor / v_mov on gfx950 (tools/valu_ubench.hip: 3.7 vs 2.1 clocks per wave64 instruction with two waves per
// SIMD), so a comparator costs ~8 clocks; delaying one wave of each SIMD pair (to make one load while the other sorts)
// changes nothing for the same reason.
#include <stdlib.h>

#include "common.h"
#include "sortnet_183.h"

namespace {

constexpr uint32_t PADK = 0xFFFFFFFFu;
constexpr int MAXQ = 64;
constexpr int RK_SENTINEL = 0x40000000;  // rank that no register index ever matches

// partner lane (lane ^ 1) through DPP quad_perm [1,0,3,2]
__device__ __forceinline__ uint32_t swap1(uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);
}

__device__ __forceinline__ void wave_fence() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_wave_barrier();
  asm volatile("" ::: "memory");
}

// rank of target (quantile q, side 0 = lower / 1 = upper neighbour) among n valid samples: utl:395, 417-461 (type 7)
__device__ __forceinline__ int hf7_rank(uint32_t n, double q, int side) {
  if (n < 2) return 0;
  const double nn = (double)n;
  const double vi = nn * q + (1.0 + q * (1.0 - 1.0 - 1.0)) - 1.0;
  if (vi >= nn - 1.0) return (int)n - 1;
  if (vi < 0.0) return 0;
  return (int)floor(vi) + side;
}

template <int N, int TMIN>
__global__ void __launch_bounds__(256, 2)
k_select_regsort(const float* __restrict__ x, int T, int64_t C, int64_t st, const double* __restrict__ qs, int nq,
                 float* __restrict__ out, int64_t ocs, int64_t oqs, int force_irregular, int abl) {
  static_assert(N == XH_SN_N, "sortnet header generated for another N");
  // LDS (dynamic, 16-byte aligned carve): per block q / gamma tables and the n = T rank table; per wave the valid counts
  // of the tile's columns, the picked keys [target][column] and a 32-register hand-over buffer [register][lane]
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  double* qS = reinterpret_cast<double*>(smem);                 // [MAXQ] quantiles
  double* gS = qS + MAXQ;                                       // [MAXQ] gamma of quantile q when n = T
  double* g1S = gS + MAXQ;                                      // [MAXQ] 1 - gamma
  int* modeS = reinterpret_cast<int*>(g1S + MAXQ);              // [MAXQ] 0: left only, 1: left + d*g, 2: right - d*(1-g)
  int* rkS = modeS + MAXQ;                                      // [2*MAXQ + 4] rkS[j + 2] = rank of target j when n = T
  uint32_t* wave0 = reinterpret_cast<uint32_t*>(rkS + 2 * MAXQ + 4);
  const int per_wave = 32 + 2 * nq * 32 + 32 * 64;              // words
  const int tid = threadIdx.x, w = tid >> 6;
  const int ntgt = 2 * nq;
  if (tid < nq) {
    const double qq = qs[tid], nn = (double)T;
    const double vi = nn * qq + (1.0 + qq * (1.0 - 1.0 - 1.0)) - 1.0;  // utl:395, alpha = beta = 1
    const double gamma = vi - floor(vi);
    qS[tid] = qq;
    gS[tid] = gamma;
    g1S[tid] = 1.0 - gamma;
    modeS[tid] = (vi >= nn - 1.0 || vi < 0.0) ? 0 : (gamma >= 0.5 ? 2 : 1);  // utl:464-491
  }
  if (tid < ntgt + 4)
    rkS[tid] = (tid < 2 || tid >= ntgt + 2) ? RK_SENTINEL : hf7_rank((uint32_t)T, qs[(tid - 2) >> 1], (tid - 2) & 1);
  __syncthreads();
  // Targets 2k / 2k + 1 are the lower / upper neighbour of quantile k.  Each parity class has non-decreasing ranks (the
  // whole sequence does not when quantiles lie closer than one rank: lower(k + 1) < upper(k)), so the picks below run
  // once per parity.  jA[p]: targets of parity p that live in lane A (rank < N) when n = T.
  int jA[2] = {0, 0};
  for (int j = 0; j < ntgt; ++j) jA[j & 1] += rkS[j + 2] < N ? 1 : 0;

  uint32_t* ncol = wave0 + w * per_wave;
  uint32_t* vals = ncol + 32;
  uint32_t* dump = vals + 2 * nq * 32;
  const int64_t ntiles = (C + 31) / 32;
  const int64_t nwaves = (int64_t)gridDim.x * 4;
  const uint32_t strideB = (uint32_t)(st * 4);
  // one-year series: the whole (T, C) view lies within 4 GiB of x (checked by the host) -> 32-bit buffer offsets
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)0xFFFFFFFFu, 0x00020000);
  for (int64_t tile = (int64_t)blockIdx.x * 4 + w; tile < ntiles; tile += nwaves) {
    // lane constants are re-derived per tile from an opaque copy of the lane id: hoisted out of the tile loop they (and
    // everything computed from them) would stay live across the sort, which needs every register it can get
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
    // ---- 1. loads: register i of lane A holds row i, of lane B row T - N + i: one uniform row base per instruction
    //      (scalar address arithmetic) plus ONE per-lane 32-bit byte offset
    const uint32_t voff = (uint32_t)(colc * 4) + (h ? (uint32_t)(T - N) * strideB : 0u);
    // (buffer loads: descriptor base + scalar row offset + per-lane offset.  The scalar offset is advanced inside the
    //  tile and made opaque per tile: hoisted out of the tile loop, N loop-invariant offsets would spill)
    uint32_t soff = 0u;
    asm volatile("" : "+s"(soff));
#define XH_LD(i) k##i = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, (int)soff, 0); soff += strideB;
    if (!(abl & 8)) {
    XH_SN_FOREACH(XH_LD)
    } else {  // diagnostics: no memory traffic, pseudo-random keys
#define XH_FK(i) k##i = (voff * 2654435761u + (uint32_t)i * 40503u) >> 9 | 0x40000000u;
    XH_SN_FOREACH(XH_FK)
#undef XH_FK
    }
#undef XH_LD
    // keys; B's first 2N - T registers repeat rows that lane A holds -> pads (only i < 2N - TMIN can be affected)
    uint32_t bcut = h ? (uint32_t)(2 * N - T) : 0u;
    asm volatile("" : "+v"(bcut));
#define XH_CV(i)                                                           \
  {                                                                        \
    const uint32_t u_ = k##i;                                              \
    const float f_ = __uint_as_float(u_);                                  \
    uint32_t kk_ = u_ ^ ((uint32_t)((int32_t)u_ >> 31) | 0x80000000u);    \
    kk_ = (f_ != f_) ? PADK : kk_;                                         \
    if (i < 2 * N - TMIN) kk_ = ((uint32_t)i < bcut) ? PADK : kk_;         \
    k##i = kk_;                                                            \
  }
    XH_SN_FOREACH(XH_CV)
#undef XH_CV
    }
    // ---- 2. local sort
#define XH_CE(i, j)                               \
  {                                               \
    const uint32_t a_ = k##i, b_ = k##j;          \
    k##i = a_ < b_ ? a_ : b_;                     \
    k##j = a_ < b_ ? b_ : a_;                     \
  }
    if (!(abl & 1)) {
    XH_SN_SORT(XH_CE)
    }
    // (lane constants are re-derived after each long phase instead of being kept in registers across it)
    uint32_t lane3 = (uint32_t)tid & 63u;
    asm volatile("" : "+v"(lane3));
    const uint32_t mA3 = (lane3 & 1u) ? 0u : 0xFFFFFFFFu;  // lane A keeps its keys negated from the split on
    // ---- 3. lane A negates, bitonic split across the lane pair.  One asm statement per register pair: written with the
    //      DPP builtin, instruction selection parks the DPP moves of ALL pairs in registers long before the v_max that
    //      consumes them and ~140 of them spill.
#define XH_DPP_SWAP1 "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1"
#define XH_SP(i, j)                                                                                              \
  {                                                                                                              \
    uint32_t t0_, t1_;                                                                                           \
    asm volatile(                                                                                                \
        "v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\ts_nop 1\n\t" /* VALU write -> DPP read: 2 wait states */  \
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
    if (!(abl & 2)) {
    XH_SN_SPLIT(XH_SP, XH_SM)
    }
#undef XH_SP
#undef XH_SM
    // ---- 4. merge
    if (!(abl & 2)) {
    XH_SN_MERGE(XH_CE)
    }
#undef XH_CE
    uint32_t lane4 = (uint32_t)tid & 63u;
    asm volatile("" : "+v"(lane4));
    const uint32_t h = lane4 & 1u, c32 = lane4 >> 1;
    const uint32_t mA = h ? 0u : 0xFFFFFFFFu;
    // ---- valid counts: the largest real sample (rank T-1) is B's k[T-1-N], the smallest (rank 0) A's k[N-1]
    uint32_t top = 0u, low = 0u;
    uint32_t tsel = (uint32_t)(T - 1 - N);
    asm volatile("" : "+v"(tsel));  // (per-lane compare: as scalar conditions the 7 select masks are hoisted and spilled)
#define XH_TP(i)                                                          \
  if (i >= TMIN - 1 - N) top = ((uint32_t)i == tsel) ? k##i : top;        \
  if (i == N - 1) low = k##i;
    XH_SN_FOREACH(XH_TP)
#undef XH_TP
    const uint32_t fl = h ? (top == PADK ? 1u : 0u) : ((low ^ mA) == PADK ? 2u : 0u);
    const uint32_t both = fl | swap1(fl);  // bit 0: the column has NaN samples, bit 1: it has nothing else
    const bool partial = (both & 1u) && !(both & 2u);
    const bool irregular = __any(partial ? 1 : 0) || force_irregular;
    // ---- 5. pick the 2*nq order statistics.  Valid count n of my column: T (or 0: every key is the pad then and every
    //      pick decodes to NaN) unless some column of the tile has NaN samples — then n is counted and every lane
    //      evaluates its own ranks (utl:395) instead of reading the shared table.
    uint32_t n = (both & 2u) ? 0u : (uint32_t)T;
    int myA[2] = {jA[0], jA[1]};  // my column's targets with rank < N, per parity
    if (irregular) {
      const uint32_t pad_stored = PADK ^ mA;
      uint32_t nv = 0;
#define XH_CN(i) nv += k##i != pad_stored ? 1u : 0u;
      XH_SN_FOREACH(XH_CN)
#undef XH_CN
      n = nv + swap1(nv);
      myA[0] = 0;
      myA[1] = 0;
#pragma nounroll
      for (int j = 0; j < ntgt; ++j) {
        const int below = hf7_rank(n, qS[j >> 1], j & 1) < N ? 1 : 0;
        myA[0] += (j & 1) ? 0 : below;
        myA[1] += (j & 1) ? below : 0;
      }
    }
    if (h == 0) ncol[c32] = n;
    // register index (in MY lane) of target j's order statistic; a sentinel that matches no index past either end
    auto local_of = [&](int j) -> uint32_t {
      int r;
      if (irregular) r = (j >= 0 && j < ntgt) ? hf7_rank(n, qS[j >> 1], j & 1) : RK_SENTINEL;
      else r = rkS[j + 2];
      return h ? (uint32_t)(r - N) : (uint32_t)(N - 1 - r);
    };
    // Both lanes meet the targets of one parity in order of INCREASING register index: B walks them upwards, A (whose
    // local order is reversed) downwards.  The registers are handed to LDS 32 at a time ([register][lane]: conflict-free
    // stores with immediate offsets) and the wanted one is read back with a per-lane address — no VALU work per
    // register.  (Earlier forms: a compare-and-branch per register = 13 KB of code, out of the instruction cache; a
    // 31-v_cndmask select tree per pick = 0.11 of the kernel's 0.78 ms.)
    int jcur0 = h ? 2 * myA[0] : 2 * (myA[0] - 1), jcur1 = (h ? 2 * myA[1] : 2 * (myA[1] - 1)) + 1;
    const int dj = h ? 2 : -2;
    uint32_t snext0 = local_of(jcur0), snext1 = local_of(jcur1);
    uint32_t* mydump = dump + lane4;
#define XH_DW(e, reg) mydump[(e) * 64] = reg;
#define XH_PICK(c)                                                     \
  if (!(abl & 4)) {                                                    \
    XH_SN_DUMP_##c(XH_DW)                                              \
    while ((snext0 >> 5) == (uint32_t)(c)) {                           \
      vals[jcur0 * 32 + (int)c32] = mydump[(snext0 & 31u) * 64] ^ mA;  \
      jcur0 += dj;                                                     \
      snext0 = local_of(jcur0);                                        \
    }                                                                  \
    while ((snext1 >> 5) == (uint32_t)(c)) {                           \
      vals[jcur1 * 32 + (int)c32] = mydump[(snext1 & 31u) * 64] ^ mA;  \
      jcur1 += dj;                                                     \
      snext1 = local_of(jcur1);                                        \
    }                                                                  \
  }
    XH_PICK(0) XH_PICK(1) XH_PICK(2) XH_PICK(3) XH_PICK(4) XH_PICK(5)
    static_assert(XH_SN_CHUNKS == 6, "XH_PICK chunks");
#undef XH_PICK
#undef XH_DW
    wave_fence();
    // ---- Hyndman-Fan lerp (utl:464-491) and stores: 32 columns x nq quantiles spread over the wave
    uint32_t lane2 = (uint32_t)tid & 63u;
    asm volatile("" : "+v"(lane2));
    for (int it = 0; it * 64 < nq * 32; ++it) {
      const int idx = it * 64 + (int)lane2;
      const int q = idx >> 5, c = idx & 31;
      if (q < nq && col0 + c < C) {
        const uint32_t n = ncol[c];
        const float left = xh_key2f(vals[(2 * q) * 32 + c]), right = xh_key2f(vals[(2 * q + 1) * 32 + c]);
        double r;
        if (n == 0) r = xh_nan64();
        else if (n < 2) r = (double)left;
        else if (n == (uint32_t)T) {  // the usual case: gamma and the branch of utl:464-491 come from the per-q tables
          const int mode = modeS[q];
          const float diff = right - left;
          r = (double)left;
          if (mode == 1) r = (double)left + (double)diff * gS[q];
          if (mode == 2) r = (double)right - (double)diff * g1S[q];
        } else {
          const double nn = (double)n, qq = qS[q];
          const double vi = nn * qq + (1.0 + qq * (1.0 - 1.0 - 1.0)) - 1.0;
          if (vi >= nn - 1.0 || vi < 0.0) r = (double)left;
          else {
            const double gamma = vi - floor(vi);
            const float diff = right - left;
            r = (double)left + (double)diff * gamma;
            if (gamma >= 0.5) r = (double)right - (double)diff * (1.0 - gamma);
          }
        }
        out[(col0 + c) * ocs + (int64_t)q * oqs] = (float)r;
      }
    }
    wave_fence();  // vals / ncol are rewritten by the next tile
  }
}

}  // namespace

// Quantiles of one-year daily series straight from the time-major (T, C) view; XH_ERR_NOTIMPL when the shape does not fit
// (the caller then takes the histogram kernels of select.hip).
int xh_select_regsort(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const double* d_q, int nq, float* out,
                      int64_t out_cstride, int64_t out_qstride) {
  constexpr int N = XH_SN_N, TMIN = 360;
  if (T < TMIN || T > 2 * N || nq < 1 || nq > MAXQ || C < 1) return XH_ERR_NOTIMPL;
  if ((unsigned long long)T * (unsigned long long)st * 4ull >= (1ull << 32)) return XH_ERR_NOTIMPL;  // 32-bit lane offsets
  if (xh_diag_env("XH_SELECT_NOREGSORT")) return XH_ERR_NOTIMPL;  // A/B against the histogram kernels
  const char* fi = xh_diag_env("XH_REGSORT_IRREGULAR");  // tests: send clean data through the NaN path as well
  const char* ea = xh_diag_env("XH_REGSORT_ABL");  // diagnostics: skip phases (1 sort, 2 split + merge, 4 picks, 8 loads; wrong results)
  const int64_t ntiles = (C + 31) / 32;
  int64_t nblk = (ntiles + 3) / 4;
  const int64_t maxblk = (int64_t)ctx->num_cu * 2;  // two 256-thread workgroups per CU (256 VGPRs per lane)
  if (nblk > maxblk) nblk = maxblk;
  const size_t lds = 3 * MAXQ * sizeof(double) + (MAXQ + 2 * MAXQ + 4) * sizeof(int) +
                     4 * (size_t)(32 + 2 * nq * 32 + 32 * 64) * sizeof(uint32_t);
  auto kern = k_select_regsort<N, TMIN>;
  if (lds > 64 * 1024) XH_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), lds, ctx->stream, x, (int)T, C, st, d_q, nq, out, out_cstride,
                     out_qstride, fi ? atoi(fi) : 0, ea ? atoi(ea) : 0);
  XH_LAUNCH_CHECK();
  return XH_OK;
}
