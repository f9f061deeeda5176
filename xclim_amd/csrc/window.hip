// window.hip — rolling-window kernels with the window in a REGISTER ring (window <= 8 time steps).
//
//   k_rolling_ring   select_rolling_resample_op's rolling step (indices/generic.py:128-174): da.rolling(time=w, center)
//   k_spell_ring     spell_mask for one variable (indices/generic.py:434-540)
// The generic kernels (reduce.hip: k_rolling_reduce, spell.hip: k_spell_mask) re-read the window from L2 at every step
// (w dependent loads per output); here every input row is loaded exactly once, in double-buffered batches of 8 rows
// (xh_march_rows), shifted through ring[WMAX] and reduced in registers in the SAME order (oldest -> newest, fp64
// accumulation) so the results are bit-identical to the generic kernels.  A lane owns VEC consecutive cells; the time
// axis is cut into chunks over blockIdx.y, each chunk re-reads its (w - 1)-row halo.
#include "common.h"
#include "runacc.h"
#include "window.h"

namespace {

constexpr int WMAX = 8;

// WT > 0: the window length is a compile-time constant (3 and 5 are instantiated: the windows of the spell / rolling
// indices); the ring then holds exactly WT rows — with the generic 8-slot ring a 3-day window still shifted 8 registers
// and walked 8 predicated slots per cell-step (PMC: 64 VALU per cell-step for spell_mask w = 3).  WT = 0: run-time w <= 8.
template <int WT>
struct RingN {
  static constexpr int N = WT > 0 ? WT : WMAX;
};

template <int VEC, int WT = 0>
struct Ring {
  static constexpr int N = RingN<WT>::N;
  float v[VEC][N];
  __device__ __forceinline__ void fill_nan() {
#pragma unroll
    for (int i = 0; i < VEC; ++i)
#pragma unroll
      for (int k = 0; k < N; ++k) v[i][k] = xh_nan32();
  }
  __device__ __forceinline__ void push(const VecF<VEC>& x) {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
#pragma unroll
      for (int k = 0; k < N - 1; ++k) v[i][k] = v[i][k + 1];
      v[i][N - 1] = x.v[i];
    }
  }
};

// statistic of the last `w` ring entries (oldest first).  RED: XH_RED_* ; returns NaN when any entry is NaN (xarray
// rolling with min_periods = window), except COUNT (number of valid entries).
template <int RED, int WT = 0>
__device__ __forceinline__ float ring_stat(const float (&r)[RingN<WT>::N], int w, double inv_w, const float* wts) {
  constexpr int WMAX = RingN<WT>::N;  // (shadows the file constant: the loops below walk this ring's slots)
  if (WT > 0) w = WT;
  double s = 0.0;
  float e = 0.f;
  bool nan = false, first = true;
  int n = 0;
#pragma unroll
  for (int k = 0; k < WMAX; ++k) {
    if (k >= WMAX - w) {  // uniform predicate
      const float x = r[k];
      if (RED == XH_RED_MIN || RED == XH_RED_MAX) nan |= (x != x);  // the fp64 sums propagate NaN by themselves
      if (RED == XH_RED_COUNT) n += (x == x) ? 1 : 0;
      if (RED == XH_RED_MIN) e = (first || x < e) ? x : e;
      else if (RED == XH_RED_MAX) e = (first || x > e) ? x : e;
      else if (RED == 100) s += (double)x * (double)wts[k];  // weighted mean (spell_mask weights)
      else s += (double)x;
      first = false;
    }
  }
  if (RED == XH_RED_COUNT) return (float)n;
  float out;
  if (RED == XH_RED_MIN || RED == XH_RED_MAX) out = e;
  else if (RED == XH_RED_MEAN) out = (float)xh_div_int(s, (double)w, inv_w);
  else if (RED == XH_RED_SUM || RED == 100) out = (float)s;
  else {  // var / std: two passes over the registers, population (ddof = 0)
    const double m = xh_div_int(s, (double)w, inv_w);
    double s2 = 0.0;
#pragma unroll
    for (int k = 0; k < WMAX; ++k)
      if (k >= WMAX - w) { const double d = (double)r[k] - m; s2 += d * d; }
    const double v = xh_div_int(s2, (double)w, inv_w);
    out = (RED == XH_RED_VAR) ? (float)v : (float)sqrt(v);
  }
  return nan ? xh_nan32() : out;
}

// Compile-time windows with a sum / mean reducer: the spell condition `float(sum / w) OP thr` is decided on the fp64 SUM
// itself, `s * sgn64 > thr64` with the threshold moved into sum space on the host (spell_sum_threshold below: float
// rounding and the division by w are monotone, so the set of sums that meet the condition is a half line) — three
// converts, two adds, one multiply and one compare per cell-step instead of the exact division (3 fp64 FMAs), the convert
// back and the fp32 compare.  NaN anywhere in the window makes the sum NaN and the compare false.
template <int WT>
__device__ __forceinline__ bool ring_sum_cond(const float (&r)[RingN<WT>::N], double sgn64, double thr64) {
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < WT; ++k) s += (double)r[k];  // oldest -> newest, as ring_stat
  return s * sgn64 > thr64;
}

template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&r)[VEC]) {
  if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
  else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) p[i] = r[i];
  }
}

template <int VEC, int RED, int WT = 0>
__global__ void __launch_bounds__(XH_BLOCK)
k_rolling_ring(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, int w_, int left, int right,
               float* __restrict__ out, int64_t out_st) {
  const int w = WT > 0 ? WT : w_;
  const int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const int64_t chunk = cdiv64(T, (int64_t)gridDim.y);
  const int64_t ta = (int64_t)blockIdx.y * chunk;
  int64_t tb = ta + chunk;
  if (tb > T) tb = T;
  if (ta >= tb) return;
  const double inv_w = 1.0 / (double)w;
  Ring<VEC, WT> ring;
  ring.fill_nan();
  // rows needed: [ta - left, tb - 1 + right] clipped to [0, T); a row tp completes the window of t = tp - right
  const int64_t r0 = ta - left < 0 ? 0 : ta - left;
  const int64_t r1 = tb + right > T ? T : tb + right;
  auto emit = [&](int64_t tp) {
    const int64_t t = tp - right;
    if (t < ta || t >= tb) return;
    float r[VEC];
    const bool whole = (t - left >= 0) && (t + right < T);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float s = ring_stat<RED, WT>(ring.v[i], w, inv_w, nullptr);
      r[i] = (RED == XH_RED_COUNT || whole) ? s : xh_nan32();
    }
    store_vec<VEC>(out + t * out_st + c, r);
  };
  xh_march_rows<VEC, 8>(x + c, st, r0, r1, [&](int64_t tp, const VecF<VEC>& xv) {
    ring.push(xv);
    emit(tp);
  });
  // windows that reach past the end of the series: NaN rows keep sliding in
  VecF<VEC> nanrow;
#pragma unroll
  for (int i = 0; i < VEC; ++i) nanrow.v[i] = xh_nan32();
  for (int64_t tp = r1; tp < tb + right; ++tp) {
    ring.push(nanrow);
    emit(tp);
  }
}

// out[t] = any cond[t'] for t' in [t, t + w - 1], cond[t'] = stat(x[t'-w+1 .. t']) op thr (False for incomplete / NaN windows)
template <int VEC, int RED, int WT = 0>
__global__ void __launch_bounds__(XH_BLOCK)
k_spell_ring(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, int w_, int op, float thr, float sgn,
             double thr64, double sgn64,
             const float* __restrict__ weights, float* __restrict__ out, int64_t out_st) {
  const int w = WT > 0 ? WT : w_;
  const int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const int64_t chunk = cdiv64(T, (int64_t)gridDim.y);
  const int64_t ta = (int64_t)blockIdx.y * chunk;
  int64_t tb = ta + chunk;
  if (tb > T) tb = T;
  if (ta >= tb) return;
  constexpr int RN = RingN<WT>::N;
  float wts[RN];
#pragma unroll
  for (int k = 0; k < RN; ++k) wts[k] = (RED == 100 && k >= RN - w) ? weights[k - (RN - w)] : 0.f;
  const double inv_w = 1.0 / (double)w;
  Ring<VEC, WT> ring;
  ring.fill_nan();
  int since[VEC];  // steps since the last window that met the condition (spell_mask: any window covering t)
#pragma unroll
  for (int i = 0; i < VEC; ++i) since[i] = WMAX;
  // outputs t in [ta, tb) need cond[t'] for t' in [ta, tb + w - 2], i.e. rows [ta - (w - 1), tb + w - 2]
  const int64_t r0 = ta - (w - 1) < 0 ? 0 : ta - (w - 1);
  const int64_t r1 = tb + w - 1 > T ? T : tb + w - 1;
  auto emit = [&](int64_t tp, bool have_row) {
    float r[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      bool cond = false;
      if (have_row && tp >= w - 1) {
        if (WT > 0 && (RED == XH_RED_SUM || RED == XH_RED_MEAN)) cond = ring_sum_cond<WT>(ring.v[i], sgn64, thr64);
        else {
          if (WT > 0 && (RED == XH_RED_SUM || RED == XH_RED_MEAN)) cond = ring_sum_cond<WT>(ring.v[i], sgn64, thr64);
          else {
            const float s = ring_stat<RED, WT>(ring.v[i], w, inv_w, wts);
            cond = WT > 0 ? (s * sgn > thr) : ((s == s) && xh_cmp_f32(s, op, thr));
          }  // WT > 0: xh_one_cmp form, NaN -> false
        }
      }
      since[i] = cond ? 0 : since[i] + 1;
      r[i] = (since[i] < w) ? 1.0f : 0.0f;
    }
    const int64_t t = tp - (w - 1);
    if (t >= ta && t < tb) store_vec<VEC>(out + t * out_st + c, r);
  };
  xh_march_rows<VEC, 8>(x + c, st, r0, r1, [&](int64_t tp, const VecF<VEC>& xv) {
    ring.push(xv);
    emit(tp, true);
  });
  for (int64_t tp = r1; tp < tb + w - 1; ++tp) emit(tp, false);
}

// spell_length_statistics with a window > 1 in ONE pass (generic.py:543-585 with resample_before_rl): the spell mask of
// k_spell_ring is not written — the mask value of step t = tp - (w - 1) feeds the run-length accumulator of its period
// directly (rle_statistics with window = 1 on the mask, runs cut at the period edges).  The mask near a period edge
// depends on the rows of the neighbouring periods (the reference builds it on the whole series first): every period
// re-reads a (w - 1)-row halo on both sides.  One workgroup row per period.
template <int VEC, int RED, int WT = 0, int SG = 0>
__global__ void __launch_bounds__(XH_BLOCK)
k_spell_runs(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, int w_, int op, float thr, float sgn,
             double thr64, double sgn64,
             const float* __restrict__ weights, int stat, const int64_t* __restrict__ seg_off, int P, float* __restrict__ out,
             int32_t* __restrict__ valid_out) {
  const int w = WT > 0 ? WT : w_;
  const int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  constexpr int RN = RingN<WT>::N;
  float wts[RN];
#pragma unroll
  for (int k = 0; k < RN; ++k) wts[k] = (RED == 100 && k >= RN - w) ? weights[k - (RN - w)] : 0.f;
  const double inv_w = 1.0 / (double)w;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    const int64_t ta = seg_off[p], tb = seg_off[p + 1];
    Ring<VEC, WT> ring;
    ring.fill_nan();
    int since[VEC];
    RunAcc acc[VEC];
    int run[VEC], nvalid[VEC], days[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) { since[i] = WMAX; acc_reset(acc[i]); run[i] = 0; nvalid[i] = 0; days[i] = 0; }
    const int64_t r0 = ta - (w - 1) < 0 ? 0 : ta - (w - 1);
    const int64_t r1 = tb + w - 1 > T ? T : tb + w - 1;
    auto emit = [&](int64_t tp, bool have_row) {
      const int64_t t = tp - (w - 1);
      const bool inside = t >= ta && t < tb;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        bool cond = false;
        if (have_row && tp >= w - 1) {
          if (WT > 0 && (RED == XH_RED_SUM || RED == XH_RED_MEAN)) cond = ring_sum_cond<WT>(ring.v[i], sgn64, thr64);
          else {
            const float s = ring_stat<RED, WT>(ring.v[i], w, inv_w, wts);
            cond = WT > 0 ? (s * sgn > thr) : ((s == s) && xh_cmp_f32(s, op, thr));
          }
        }
        since[i] = cond ? 0 : since[i] + 1;
        const bool on = inside && (since[i] < w);
        const int len = (inside && !on) ? run[i] : 0;  // a spell ended at t - 1
        acc_add_if<SG>(acc[i], len);
        run[i] = on ? run[i] + 1 : (inside ? 0 : run[i]);
        days[i] += on ? 1 : 0;
      }
    };
    if (ta < tb) {
      xh_march_rows<VEC, 8>(x + c, st, r0, r1, [&](int64_t tp, const VecF<VEC>& xv) {
        ring.push(xv);
        if (tp >= ta && tp < tb) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) nvalid[i] += (xv.v[i] == xv.v[i]) ? 1 : 0;
        }
        emit(tp, true);
      });
      for (int64_t tp = r1; tp < tb + w - 1; ++tp) emit(tp, false);
    }
    const int64_t o = (int64_t)p * C + c;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      acc_add_if<SG>(acc[i], run[i]);  // spell cut by the period end
      out[o + i] = acc_result(acc[i], stat, days[i]);
      if (valid_out) valid_out[o + i] = nvalid[i];
    }
  }
}

static dim3 window_grid(xh_ctx* ctx, int64_t T, int64_t C, int vec) {
  const int64_t cblocks = cdiv64(cdiv64(C, vec), XH_BLOCK);
  int64_t gy = cdiv64((int64_t)ctx->num_cu * 12, cblocks);
  if (gy < 1) gy = 1;
  if (gy > cdiv64(T, 32)) gy = cdiv64(T, 32);  // chunks of at least 32 steps: the halo stays below 25 %
  if (gy < 1) gy = 1;
  return dim3((unsigned)cblocks, (unsigned)gy);
}

}  // namespace

int xh_launch_rolling_ring(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int window, int left, int right,
                           int reducer, float* out, int64_t out_st) {
  if (window > WMAX) return XH_ERR_NOTIMPL;
  const int vec = (xh_pick_vec(x, C, st) == 4 && xh_pick_vec(out, C, out_st) == 4) ? 4 : 1;
  const dim3 grid = window_grid(ctx, T, C, vec);
#define XH_RR(R)                                                                                                          \
  case R:                                                                                                                 \
    if (vec == 4 && window == 3)                                                                                          \
      hipLaunchKernelGGL((k_rolling_ring<4, R, 3>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, window, left, right, out, \
                         out_st);                                                                                         \
    else if (vec == 4 && window == 5)                                                                                     \
      hipLaunchKernelGGL((k_rolling_ring<4, R, 5>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, window, left, right, out, \
                         out_st);                                                                                         \
    else if (vec == 4)                                                                                                    \
      hipLaunchKernelGGL((k_rolling_ring<4, R>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, window, left, right, out, \
                         out_st);                                                                                         \
    else                                                                                                                  \
      hipLaunchKernelGGL((k_rolling_ring<1, R>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, window, left, right, out, \
                         out_st);                                                                                         \
    break;
  switch (reducer) {
    XH_RR(XH_RED_SUM) XH_RR(XH_RED_MEAN) XH_RR(XH_RED_MIN) XH_RR(XH_RED_MAX) XH_RR(XH_RED_STD) XH_RR(XH_RED_VAR)
    XH_RR(XH_RED_COUNT)
    default:
      return XH_ERR_NOTIMPL;
  }
#undef XH_RR
  XH_LAUNCH_CHECK();
  return XH_OK;
}

namespace {

struct SpellArgs {
  const float* x;
  int64_t T, C, st;
  int window, op;
  float thr, sgn;
  double thr64, sgn64;  // the same condition on the fp64 window sum (sum / mean reducers, compile-time windows)
  const float* d_weights;
};

template <int VEC, int RED, int WT>
void launch_spell_ring_t(xh_ctx* ctx, dim3 grid, const SpellArgs& a, float* out, int64_t out_st) {
  hipLaunchKernelGGL((k_spell_ring<VEC, RED, WT>), grid, dim3(XH_BLOCK), 0, ctx->stream, a.x, a.T, a.C, a.st, a.window, a.op,
                     a.thr, a.sgn, a.thr64, a.sgn64, a.d_weights, out, out_st);
}

template <int RED>
void launch_spell_ring_r(xh_ctx* ctx, dim3 grid, int vec, int wt, const SpellArgs& a, float* out, int64_t out_st) {
  if (vec == 4 && wt == 3) launch_spell_ring_t<4, RED, 3>(ctx, grid, a, out, out_st);
  else if (vec == 4 && wt == 5) launch_spell_ring_t<4, RED, 5>(ctx, grid, a, out, out_st);
  else if (vec == 4) launch_spell_ring_t<4, RED, 0>(ctx, grid, a, out, out_st);
  else launch_spell_ring_t<1, RED, 0>(ctx, grid, a, out, out_st);
}

template <int VEC, int RED, int WT, int SG>
void launch_spell_runs_t(xh_ctx* ctx, dim3 grid, const SpellArgs& a, int stat, const int64_t* d_seg, int P, float* out,
                         int32_t* valid_out) {
  hipLaunchKernelGGL((k_spell_runs<VEC, RED, WT, SG>), grid, dim3(XH_BLOCK), 0, ctx->stream, a.x, a.T, a.C, a.st, a.window,
                     a.op, a.thr, a.sgn, a.thr64, a.sgn64, a.d_weights, stat, d_seg, P, out, valid_out);
}

template <int RED>
void launch_spell_runs_r(xh_ctx* ctx, dim3 grid, int vec, int wt, int sg, const SpellArgs& a, int stat, const int64_t* d_seg,
                         int P, float* out, int32_t* valid_out) {
#define XH_SRN(V, W)                                                                      \
  do {                                                                                    \
    if (sg == 1) launch_spell_runs_t<V, RED, W, 1>(ctx, grid, a, stat, d_seg, P, out, valid_out);      \
    else if (sg == 2) launch_spell_runs_t<V, RED, W, 2>(ctx, grid, a, stat, d_seg, P, out, valid_out); \
    else launch_spell_runs_t<V, RED, W, 0>(ctx, grid, a, stat, d_seg, P, out, valid_out);              \
  } while (0)
  if (vec == 1 && wt == 3) XH_SRN(1, 3);
  else if (vec == 1 && wt == 5) XH_SRN(1, 5);
  else if (vec == 4) launch_spell_runs_t<4, RED, 0, 0>(ctx, grid, a, stat, d_seg, P, out, valid_out);
  else launch_spell_runs_t<1, RED, 0, 0>(ctx, grid, a, stat, d_seg, P, out, valid_out);
#undef XH_SRN
}

// Threshold of the one-compare condition `F(s) * sgn > t` (F(s) = float(s / w) for the mean, float(s) for the sum, s the
// fp64 window sum) moved into sum space: sgn = +1: F(s) > t  <=>  s > S+,  S+ = the largest double with F(S+) <= t;
// sgn = -1: F(s) < -t  <=>  s < S-,  S- = the smallest double with F(S-) >= -t.  F is monotone, so a bisection over the
// ordered doubles (64 steps) finds the boundary; the arithmetic of F is the device's (IEEE division and round-to-nearest
// conversion: xh_div_int is bit-identical to s / w).  Returns thr64 for `s * sgn > thr64`.
double spell_sum_threshold(float t, float sgn, int w, bool mean) {
  auto F = [&](double s) -> float { return (float)(mean ? s / (double)w : s); };
  auto ord = [](double d) -> int64_t {  // monotone map double -> int64
    int64_t u;
    memcpy(&u, &d, 8);
    return u < 0 ? (int64_t)(0x8000000000000000ull - (uint64_t)u) : u;
  };
  auto unord = [](int64_t k) -> double {
    const int64_t u = k < 0 ? (int64_t)(0x8000000000000000ull - (uint64_t)k) : k;
    double d;
    memcpy(&d, &u, 8);
    return d;
  };
  int64_t lo = ord(-1.7976931348623157e308), hi = ord(1.7976931348623157e308);
  if (sgn > 0.0f) {  // largest s with F(s) <= t  (F(lo) = -inf <= t, F(hi) = +inf > t for a finite t)
    while (lo + 1 < hi) {
      const int64_t mid = (lo >> 1) + (hi >> 1) + (lo & hi & 1);  // (hi - lo overflows int64 over the whole double range)
      if (F(unord(mid)) <= t) lo = mid; else hi = mid;
    }
    return unord(lo);
  }
  const float nt = -t;  // smallest s with F(s) >= -t
  while (lo + 1 < hi) {
    const int64_t mid = (lo >> 1) + (hi >> 1) + (lo & hi & 1);
    if (F(unord(mid)) >= nt) hi = mid; else lo = mid;
  }
  return -unord(hi);
}

// The compile-time windows (3, 5) take the one-compare form of the condition (xh_one_cmp): ordered op, finite threshold.
SpellArgs spell_args(const float* x, int64_t T, int64_t C, int64_t st, int window, int win_red, int op, float thr,
                     const float* d_weights, bool wt_built, int* wt) {
  SpellArgs a = {x, T, C, st, window, op, thr, 1.0f, 0.0, 1.0, d_weights};
  const XhOneCmp one = xh_one_cmp(op, thr);
  *wt = (wt_built && one.ok && (window == 3 || window == 5)) ? window : 0;
  if (*wt) {
    a.thr = one.thr; a.sgn = one.sgn;
    a.sgn64 = (double)one.sgn;
    if (win_red == 0 || win_red == 1) a.thr64 = spell_sum_threshold(one.thr, one.sgn, window, win_red == 1);
  }
  return a;
}

}  // namespace

int xh_launch_spell_ring(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int window, int win_red, int op,
                         float thr, const float* d_weights, float* out, int64_t out_st) {
  if (window > WMAX) return XH_ERR_NOTIMPL;
  const int vec = (xh_pick_vec(x, C, st) == 4 && xh_pick_vec(out, C, out_st) == 4) ? 4 : 1;
  const dim3 grid = window_grid(ctx, T, C, vec);
  int wt;
  const SpellArgs a = spell_args(x, T, C, st, window, win_red, op, thr, d_weights, vec == 4, &wt);
  switch (win_red) {  // win_red (spell.hip): 0 sum, 1 mean, 2 min, 3 max, 4 weighted mean
    case 0: launch_spell_ring_r<XH_RED_SUM>(ctx, grid, vec, wt, a, out, out_st); break;
    case 1: launch_spell_ring_r<XH_RED_MEAN>(ctx, grid, vec, wt, a, out, out_st); break;
    case 2: launch_spell_ring_r<XH_RED_MIN>(ctx, grid, vec, wt, a, out, out_st); break;
    case 3: launch_spell_ring_r<XH_RED_MAX>(ctx, grid, vec, wt, a, out, out_st); break;
    case 4: launch_spell_ring_r<100>(ctx, grid, vec, wt, a, out, out_st); break;
    default: return XH_ERR_NOTIMPL;
  }
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_launch_spell_runs(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int window, int win_red, int op, float thr,
                         const float* d_weights, int stat, const int64_t* d_seg, int P, float* out, int32_t* valid_out) {
  if (window > WMAX) return XH_ERR_NOTIMPL;
  const unsigned py = (unsigned)(P > 4096 ? 4096 : P);
  // four cells per lane only when that still leaves >= 8 workgroups per CU (a period cannot be cut into time chunks)
  const int vec = (xh_pick_vec(x, C, st) == 4 && cdiv64(cdiv64(C, 4), XH_BLOCK) * py >= 8 * (int64_t)ctx->num_cu) ? 4 : 1;
  const dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), py);
  int wt;
  const SpellArgs a = spell_args(x, T, C, st, window, win_red, op, thr, d_weights, vec == 1, &wt);
  // fields of the run accumulator the statistic reads (runacc.h): 1 max, 2 sum / count / mean / plain sum, 0 all
  const int sg = stat == XH_RUN_MAX ? 1 : (stat == XH_RUN_SUM || stat == XH_RUN_COUNT || stat == XH_RUN_MEAN || stat == XH_RUN_PLAINSUM) ? 2 : 0;
  switch (win_red) {
    case 0: launch_spell_runs_r<XH_RED_SUM>(ctx, grid, vec, wt, sg, a, stat, d_seg, P, out, valid_out); break;
    case 1: launch_spell_runs_r<XH_RED_MEAN>(ctx, grid, vec, wt, sg, a, stat, d_seg, P, out, valid_out); break;
    case 2: launch_spell_runs_r<XH_RED_MIN>(ctx, grid, vec, wt, sg, a, stat, d_seg, P, out, valid_out); break;
    case 3: launch_spell_runs_r<XH_RED_MAX>(ctx, grid, vec, wt, sg, a, stat, d_seg, P, out, valid_out); break;
    case 4: launch_spell_runs_r<100>(ctx, grid, vec, wt, sg, a, stat, d_seg, P, out, valid_out); break;
    default: return XH_ERR_NOTIMPL;
  }
  XH_LAUNCH_CHECK();
  return XH_OK;
}
