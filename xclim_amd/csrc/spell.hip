// spell.hip — spell masks, seasons, hysteresis runs: the remaining state machines of the run-length family.
//
// Reference: indices/generic.py:434-540 (spell_mask), indices/run_length.py:844-888 (runs_with_holes), 805-841
// (keep_longest_run), 891-1145 (season_start / season_end / season / season_length), 491-540 (windowed_max_run_sum).
// Same layout as runlen.hip: time-major (T, C), one lane per cell marching along time, periods on blockIdx.y.
#include "common.h"
#include "window.h"

// ---- spell_mask ------------------------------------------------------------------------------------------
// A day is in a spell iff it belongs to ANY window of `window` consecutive days whose statistic satisfies the
// condition (gen:519-535; the min/max fast path of gen:503-518 gives the same mask).  cond[t'] is evaluated on the
// trailing window [t'-w+1, t'] (NaN anywhere or an incomplete window -> False, xarray rolling min_periods = w);
// out[t] = any cond[t'] for t' in [t, t+w-1].  The window is re-read from L2 (w loads per step).
// win_red: 0 sum, 1 mean, 2 min, 3 max, 4 weighted mean (dot with weights[w]).
// Several variables (gen:434-540 with a list of DataArrays): the per-variable conditions are combined with all (1) /
// any (2) before the second rolling step; xs / thrs hold nvar device pointers / thresholds.
struct SpellVars {
  const float* x[8];
  float thr[8];
};

__global__ void __launch_bounds__(XH_BLOCK)
k_spell_mask(SpellVars vars, int nvar, int combine, int64_t T, int64_t C, int64_t st, int window, int win_red, int op,
             const float* __restrict__ weights, float* __restrict__ out, int64_t out_st) {
  int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  int64_t last_true = -1;
  for (int64_t tp = 0; tp < T + window - 1; ++tp) {
    bool cond = false;
    if (tp < T && tp >= window - 1) {
      cond = combine == 1;
      for (int iv = 0; iv < nvar; ++iv) {
        const float* __restrict__ x = vars.x[iv];
        double s = 0.0;
        float e = x[(tp - window + 1) * st + c];
        bool nan = false;
        for (int k = 0; k < window; ++k) {
          float v = x[(tp - window + 1 + k) * st + c];
          nan |= (v != v);
          if (win_red == 2) e = v < e ? v : e;
          else if (win_red == 3) e = v > e ? v : e;
          else if (win_red == 4) s += (double)v * (double)weights[k];
          else s += (double)v;
        }
        float stat = (win_red == 2 || win_red == 3) ? e : (win_red == 1 ? (float)(s / (double)window) : (float)s);
        const bool cv = !nan && xh_cmp_f32(stat, op, vars.thr[iv]);
        cond = combine == 1 ? (cond && cv) : (cond || cv);
      }
    }
    if (cond) last_true = tp;
    int64_t t = tp - (window - 1);
    if (t >= 0) out[t * out_st + c] = (last_true >= t) ? 1.0f : 0.0f;
  }
}

// ---- runs_with_holes -------------------------------------------------------------------------------------
// state[t] = 1 where a run of >= w_start True of `start` begins/continues with >= w_start elements remaining,
//            0 where the same holds for `stop` with w_stop (stop wins), else the previous state; initial 0.
// Backward sweep computes the remaining-run lengths and writes 1 / 0 / NaN marks, forward sweep forward-fills.
__global__ void __launch_bounds__(XH_BLOCK)
k_runs_with_holes(const float* __restrict__ a, const float* __restrict__ b, int64_t T, int64_t C, int64_t sa, int64_t sb,
                  int w_start, int w_stop, int stop_is_not_start, float* __restrict__ out, int64_t out_st) {
  int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  int ra = 0, rb = 0;
  for (int64_t t = T - 1; t >= 0; --t) {
    float va = a[t * sa + c];
    bool on_a = va > 0.0f;  // astype(int).fillna(0): NaN -> 0
    bool on_b = stop_is_not_start ? !(on_a) : (b[t * sb + c] > 0.0f);
    ra = on_a ? ra + 1 : 0;
    rb = on_b ? rb + 1 : 0;
    float mark = xh_nan32();
    if (ra >= w_start) mark = 1.0f;
    if (rb >= w_stop) mark = 0.0f;  // combine_first: stop positions take precedence
    out[t * out_st + c] = mark;
  }
  float state = 0.0f;
  for (int64_t t = 0; t < T; ++t) {
    float m = out[t * out_st + c];
    if (m == m) state = m;
    out[t * out_st + c] = state;
  }
}

// Single forward pass for windows up to 64 steps: "a run of >= w True starts at t" is known w - 1 steps later (the
// trailing run at t + w - 1 reaches w), so the two mark series are kept in per-lane 64-bit shift registers and the
// state is emitted D = max(w_start, w_stop) - 1 steps behind the read position: 4 B in (8 B with a stop mask) + 4 B out
// per step instead of the 20 B of the two-pass kernel.
template <int VEC, bool TWO>
__global__ void __launch_bounds__(XH_BLOCK)
k_runs_with_holes_fwd(const float* __restrict__ a, const float* __restrict__ b, int64_t T, int64_t C, int64_t sa, int64_t sb,
                      int w_start, int w_stop, float* __restrict__ out, int64_t out_st) {
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const int D = (w_start > w_stop ? w_start : w_stop) - 1;
  const int bit_s = D - (w_start - 1), bit_t = D - (w_stop - 1);
  int ra[VEC], rb[VEC];
  unsigned long long ms[VEC], mt[VEC];
  float state[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { ra[v] = 0; rb[v] = 0; ms[v] = 0ull; mt[v] = 0ull; state[v] = 0.0f; }
  auto step = [&](int64_t tp, const VecF<VEC>& xa, const VecF<VEC>& xb, bool inside) {
    float r[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const bool on_a = inside && xa.v[v] > 0.0f;  // astype(int).fillna(0): NaN -> 0
      const bool on_b = inside && (TWO ? (xb.v[v] > 0.0f) : !(xa.v[v] > 0.0f));
      ra[v] = on_a ? ra[v] + 1 : 0;
      rb[v] = on_b ? rb[v] + 1 : 0;
      ms[v] = (ms[v] << 1) | (ra[v] >= w_start ? 1ull : 0ull);
      mt[v] = (mt[v] << 1) | (rb[v] >= w_stop ? 1ull : 0ull);
      const bool m1 = (ms[v] >> bit_s) & 1ull, m0 = (mt[v] >> bit_t) & 1ull;
      state[v] = m0 ? 0.0f : (m1 ? 1.0f : state[v]);  // combine_first: stop positions take precedence
      r[v] = state[v];
    }
    const int64_t t = tp - D;
    if (t >= 0) {
      if (VEC == 4) *reinterpret_cast<float4*>(out + t * out_st + c) = make_float4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
      else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) out[t * out_st + c + v] = r[v];
      }
    }
  };
  if (TWO) xh_march_rows2<VEC, 8>(a + c, b + c, sa, sb, 0, T, [&](int64_t tp, const VecF<VEC>& xa, const VecF<VEC>& xb) { step(tp, xa, xb, true); });
  else xh_march_rows<VEC, 8>(a + c, sa, 0, T, [&](int64_t tp, const VecF<VEC>& xa) { step(tp, xa, xa, true); });
  VecF<VEC> z;
#pragma unroll
  for (int v = 0; v < VEC; ++v) z.v[v] = 0.f;
  for (int64_t tp = T; tp < T + D; ++tp) step(tp, z, z, false);
}

// ---- keep_longest_run (per period) -----------------------------------------------------------------------------
template <int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_keep_longest_run(const float* __restrict__ x, int64_t C, int64_t st, const int64_t* __restrict__ seg_off, int P,
                   float* __restrict__ out, int64_t out_st) {
  // rle() runs over the WHOLE series (rl:823), then each period keeps, among the runs that START inside it, the
  // first one with the largest FULL length (it may extend past the period end; only its part inside the period is
  // marked); the leading part of a run that started in an earlier period is never marked (NaN == max is False).
  // Quirk restated (rl:826-833): a period without any run start marks its first non-run element.
  // One backward march over the whole series (rows loaded once, in double-buffered batches).  Whether step t + 1 starts
  // a run is known when step t has been seen, so every step is resolved one step late; a period is written out (vector
  // stores, no loads) as soon as its first step has been resolved.
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  int rem[VEC], prem[VEC], best[VEC];
  int64_t best_start[VEC], first_zero[VEC];
  bool pon[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) { rem[v] = 0; prem[v] = 0; best[v] = 0; best_start[v] = -1; first_zero[v] = -1; pon[v] = false; }
  int pc = P - 1;          // period of the pending step
  int64_t pt = -1;         // pending step (-1: none)
  int64_t pc0 = seg_off[pc];  // first step of period pc (kept in a register: a scalar load per step would stall)
  auto prev_period = [&]() { pc--; pc0 = pc >= 0 ? seg_off[pc] : -1; };
  auto write_period = [&](int p) {
    const int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    for (int64_t t = t0; t < t1; ++t) {
      float r[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const bool m = best[v] > 0 ? (t >= best_start[v] && t < best_start[v] + best[v]) : (t == first_zero[v]);
        r[v] = m ? 1.0f : 0.0f;
      }
      if (VEC == 4) *reinterpret_cast<float4*>(out + t * out_st + c) = make_float4(r[0], r[1 % VEC], r[2 % VEC], r[3 % VEC]);
      else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) out[t * out_st + c + v] = r[v];
      }
    }
#pragma unroll
    for (int v = 0; v < VEC; ++v) { best[v] = 0; best_start[v] = -1; first_zero[v] = -1; }
  };
  // resolve the pending step given whether the step before it is on; close its period when it was the period's first step
  auto resolve = [&](const bool (&before_on)[VEC]) {
    if (pt < 0) return;
#pragma unroll
    for (int v = 0; v < VEC; ++v)
      if (pon[v] && !before_on[v] && prem[v] >= best[v]) { best[v] = prem[v]; best_start[v] = pt; }  // >=: earlier start wins ties
    if (pt == pc0) { write_period(pc); prev_period(); }
  };
  xh_march_rows_rev<VEC, 8>(x + c, st, seg_off[0], seg_off[P], [&](int64_t t, const VecF<VEC>& xv) {
    bool on[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) on[v] = xv.v[v] > 0.0f;
    resolve(on);
    while (pc >= 0 && t < pc0) { write_period(pc); prev_period(); }  // empty periods
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      rem[v] = on[v] ? rem[v] + 1 : 0;
      if (!on[v]) first_zero[v] = t;
      pon[v] = on[v]; prem[v] = rem[v];
    }
    pt = t;
  });
  bool none[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) none[v] = false;
  resolve(none);
  while (pc >= 0) { write_period(pc); prev_period(); }
}

// ---- season (per period) -------------------------------------------------------------------------------------
// start = first_run_before_date(da, window, mid): first t with `window` consecutive True, all inside t < mid+window-1
// end   = first t >= max(start, mid) with `window` consecutive False (runs cut at that lower bound);
// length: 0 if no start; T - start if no end; else end - start.  Reported end = T-1 when none, NaN when no start.
// mid_idx[p] < 0 : the date is not in the group -> start NaN (rl:1319-1321) ; mid_idx == INT_MAX-ish : date=None.
// One forward march per period (rows loaded once, in double-buffered batches), branch-free state machine:
//   start search while i < limit and no start yet; end search from lb = max(start, mid) on.  The start is known at
//   i = start + window - 1 >= start, and the steps start .. start + window - 1 are all True, so the end search that
//   formally begins at lb can be switched on as soon as the start is known without changing its counters.
template <int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_season(const float* __restrict__ x, int64_t C, int64_t st, int window, const int64_t* __restrict__ seg_off,
         const int32_t* __restrict__ mid_idx, int has_date, int P, float* __restrict__ start_out,
         float* __restrict__ end_out, float* __restrict__ len_out) {
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    const int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    const int len = (int)(t1 - t0);
    const int mid = has_date ? mid_idx[p] : 0;  // relative to the period start
    const bool nodate = has_date && mid < 0;
    int limit = has_date ? (mid + window - 1) : len;  // da.where(t < mid + window - 1)
    if (limit > len) limit = len;
    int run[VEC], beg[VEC], ones[VEC], runf[VEC], end[VEC], onesf[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { run[v] = 0; beg[v] = -1; ones[v] = 0; runf[v] = 0; end[v] = -1; onesf[v] = 0; }
    xh_march_rows<VEC, 8>(x + c, st, t0, t1, [&](int64_t t, const VecF<VEC>& xv) {
      const int i = (int)(t - t0);
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const bool on = xv.v[v] > 0.0f;
        // phase 1: start (window == 1 keeps counting the Trues for the argmax == argmin quirk)
        const bool p1 = i < limit && (beg[v] < 0 || window == 1);
        run[v] = p1 ? (on ? run[v] + 1 : 0) : run[v];
        ones[v] += (p1 && on) ? 1 : 0;
        beg[v] = (p1 && beg[v] < 0 && run[v] >= window) ? i - window + 1 : beg[v];
        // phase 2: end — window consecutive False at i >= max(beg, mid)
        const int lb = beg[v] > mid ? beg[v] : mid;
        const bool p2 = beg[v] >= 0 && i >= lb && (end[v] < 0 || window == 1);
        const bool off = !on;
        runf[v] = p2 ? (off ? runf[v] + 1 : 0) : runf[v];
        onesf[v] += (p2 && off) ? 1 : 0;
        end[v] = (p2 && end[v] < 0 && runf[v] >= window) ? i - window + 1 : end[v];
      }
    });
    const int64_t o = (int64_t)p * C + c;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float fs = xh_nan32(), fe = xh_nan32(), fl = 0.0f;
      int b = beg[v], e = end[v];
      if (window == 1 && limit == len && ones[v] == len) b = -1;  // argmax == argmin quirk (rl:603-605)
      if (!nodate && b >= 0) {
        const int lb = b > mid ? b : mid;
        if (window == 1 && lb == 0 && onesf[v] == len) e = -1;
        fs = (float)b;
        fl = e < 0 ? (float)(len - b) : (float)(e - b);
        fe = e < 0 ? (float)(len - 1) : (float)e;
      }
      start_out[o + v] = fs;
      end_out[o + v] = fe;
      len_out[o + v] = fl;
    }
  }
}

// window >= 2 (no argmax == argmin quirk): ONE counter per cell and two mode flags instead of the two counters, two
// True-counts and ~8 compares per cell-step of k_season (v_cmp issues at half rate on gfx950, tools/valu_ubench.hip).
//   mode 0 (m0): looking for the start, k = consecutive True;   valid while i < limit
//   mode 1 (m1): looking for the end,   k = consecutive False;  valid from i >= mid, the count restarts at i == mid
// The end search formally begins at lb = max(start, mid): steps start .. start + window - 1 are True, so counting the
// False steps from the step the start became known (k = 0 there) is the same count when start >= mid, and the restart
// at i == mid (uniform) cuts the runs there when mid > start.  2 compares + 6 selects / adds per cell-step; the mode
// flags live in lane masks (SALU).
template <int VEC>
__global__ void __launch_bounds__(XH_BLOCK)
k_season_w(const float* __restrict__ x, int64_t C, int64_t st, int window, const int64_t* __restrict__ seg_off,
           const int32_t* __restrict__ mid_idx, int has_date, int P, float* __restrict__ start_out,
           float* __restrict__ end_out, float* __restrict__ len_out) {
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    const int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    const int len = (int)(t1 - t0);
    const int mid = has_date ? mid_idx[p] : 0;
    const bool nodate = has_date && mid < 0;
    int limit = has_date ? (mid + window - 1) : len;
    if (limit > len) limit = len;
    int k[VEC], beg[VEC], end[VEC];
    bool m0[VEC], m1[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) { k[v] = 0; beg[v] = -1; end[v] = -1; m0[v] = true; m1[v] = false; }
    xh_march_rows<VEC, 8>(x + c, st, t0, t1, [&](int64_t t, const VecF<VEC>& xv) {
      const int i = (int)(t - t0);
      const bool a_ok = i < limit, b_ok = i >= mid, at_mid = i == mid;
      const int pos = i - window + 1;
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const bool on = xv.v[v] > 0.0f;
        const bool hit = on != m1[v];
        const int kp = (at_mid && m1[v]) ? 0 : k[v];
        const int kn = hit ? kp + 1 : 0;
        const bool ge = kn >= window;
        const bool d0 = ge && m0[v] && a_ok;
        const bool d1 = ge && m1[v] && b_ok;
        beg[v] = d0 ? pos : beg[v];
        end[v] = d1 ? pos : end[v];
        k[v] = d0 ? 0 : kn;
        m0[v] = m0[v] && !d0;
        m1[v] = (m1[v] || d0) && !d1;
      }
    });
    const int64_t o = (int64_t)p * C + c;
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      float fs = xh_nan32(), fe = xh_nan32(), fl = 0.0f;
      const int b = beg[v], e = end[v];
      if (!nodate && b >= 0) {
        fs = (float)b;
        fl = e < 0 ? (float)(len - b) : (float)(e - b);
        fe = e < 0 ? (float)(len - 1) : (float)e;
      }
      start_out[o + v] = fs;
      end_out[o + v] = fe;
      len_out[o + v] = fl;
    }
  }
}

// ---- windowed_max_run_sum (cut at segments / whole series) ------------------------------------------------------
// rl:491-540: d_rse = reset-cumsum of the VALUES from the run's first element to the next exact zero (NaN adds 0
// and does not reset), kept where rle(da > 0) >= window, max over the period.  Backward march.
template <int VEC, bool CUT>
__global__ void __launch_bounds__(XH_BLOCK)
k_max_run_sum(const float* __restrict__ x, int64_t C, int64_t st, int window, const int64_t* __restrict__ seg_off, int P,
              float* __restrict__ out) {
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  // the reference's arithmetic, restated so that results are bit-identical: cs = running fp32 cumsum of the
  // reversed series (NaN adds 0, never reset), csr = cs at the latest exact zero, d_rse = cs - csr (rl:154-169).
  // Whether step t is the FIRST of its run is known one step later in the backward march, so the candidate of
  // step t is held back until t - 1 has been seen (rows are loaded once, in double-buffered batches).
  float cs[VEC], csr[VEC], best[VEC], pacc[VEC];
  int run[VEC], prun[VEC];
  bool pon[VEC];
  auto reset = [&]() {
#pragma unroll
    for (int v = 0; v < VEC; ++v) { cs[v] = 0.f; csr[v] = 0.f; best[v] = 0.f; pacc[v] = 0.f; run[v] = 0; prun[v] = 0; pon[v] = false; }
  };
  auto fold_pending = [&](const bool (&on_now)[VEC]) {  // the pending step is the first of its run iff this one is off
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float d = (pon[v] && !on_now[v] && prun[v] >= window) ? pacc[v] : 0.0f;
      best[v] = d > best[v] ? d : best[v];
    }
  };
  auto advance = [&](const VecF<VEC>& xv, const bool (&on_now)[VEC]) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      const float val = xv.v[v];
      cs[v] = cs[v] + ((val == val) ? val : 0.0f);
      if (val == 0.0f) csr[v] = cs[v];
      run[v] = on_now[v] ? run[v] + 1 : 0;
      pon[v] = on_now[v]; pacc[v] = cs[v] - csr[v]; prun[v] = run[v];
    }
  };
  auto flush = [&](int p, bool nonempty) {
#pragma unroll
    for (int v = 0; v < VEC; ++v) { out[(int64_t)p * C + c + v] = nonempty ? best[v] : xh_nan32(); best[v] = 0.f; }
  };
  bool off[VEC];
#pragma unroll
  for (int v = 0; v < VEC; ++v) off[v] = false;
  if (CUT) {
    for (int p = blockIdx.y; p < P; p += gridDim.y) {
      const int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
      reset();
      xh_march_rows_rev<VEC, 8>(x + c, st, t0, t1, [&](int64_t, const VecF<VEC>& xv) {
        bool on[VEC];
#pragma unroll
        for (int v = 0; v < VEC; ++v) on[v] = xv.v[v] > 0.0f;
        fold_pending(on);
        advance(xv, on);
      });
      fold_pending(off);  // a run that starts on the first step
      flush(p, t1 > t0);
    }
  } else {
    // resample AFTER (rl:533-538): one march over the whole series — the cumsum and the run lengths cross the period
    // edges — and a run's sum goes to the period of its first step
    reset();
    int pq = P - 1;                   // period of the pending step
    int64_t q0 = seg_off[pq];
    bool any = false;                 // period pq holds at least one step
    xh_march_rows_rev<VEC, 8>(x + c, st, seg_off[0], seg_off[P], [&](int64_t t, const VecF<VEC>& xv) {
      bool on[VEC];
#pragma unroll
      for (int v = 0; v < VEC; ++v) on[v] = xv.v[v] > 0.0f;
      fold_pending(on);
      while (t < q0) { flush(pq, any); any = false; pq--; q0 = seg_off[pq]; }  // the pending step closed its period(s)
      any = true;
      advance(xv, on);
    });
    fold_pending(off);
    while (pq >= 0) { flush(pq, any); any = false; pq--; }
  }
}

// ---- event compaction: run_bounds (rl:745-802) and find_events (rl:1760-1901) -------------------------------------
// `runs` is a 0/1 field (a mask, or the output of runs_with_holes).  Per (period, cell) the k-th run in time order
// writes row k of the (P, maxev, C) outputs; rows past the number of runs stay NaN (the reference pads with NaN):
//   start : index of the run's first step, relative to the period start       end : index of the first step after
//   the run (diff == -1 in run_bounds; NaN when the run reaches the period end)    len : run length (rle)
//   eff   : steps of the run where `eff` is true (event_effective_length; = len when eff is NULL)
//   sum   : _cumsum_reset_xr(data.where(runs == 1), index="first", reset_on_zero=False) at the run start, restated
//           with the reference's arithmetic (fp32 cumsum of the reversed series, NaN adds 0, minus its value at the
//           latest NaN): the sum of `data` from the run start up to the first NaN of data inside the run.
// Pass 1 counts the runs (the backward march numbers them from the end), pass 2 marches backward.
__global__ void __launch_bounds__(XH_BLOCK)
k_run_events(const float* __restrict__ runs, const float* __restrict__ eff, const float* __restrict__ data, int64_t C,
             int64_t st, const int64_t* __restrict__ seg_off, int P, int maxev, float* __restrict__ o_start,
             float* __restrict__ o_end, float* __restrict__ o_len, float* __restrict__ o_eff, float* __restrict__ o_sum) {
  int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    const int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    int ne = 0;
    bool prev = false;
    for (int64_t t = t0; t < t1; ++t) {
      const bool on = runs[t * st + c] == 1.0f;
      ne += (on && !prev) ? 1 : 0;
      prev = on;
    }
    const int64_t base = (int64_t)p * maxev * C + c;
    for (int k = (ne < maxev ? ne : maxev); k < maxev; ++k) {
      const int64_t o = base + (int64_t)k * C;
      o_start[o] = xh_nan32();
      if (o_end) o_end[o] = xh_nan32();
      if (o_len) o_len[o] = xh_nan32();
      if (o_eff) o_eff[o] = xh_nan32();
      if (o_sum) o_sum[o] = xh_nan32();
    }
    int k = ne, run = 0, effc = 0;
    float cs = 0.0f, csr = 0.0f, endv = xh_nan32();
    bool on = t1 > t0 ? (runs[(t1 - 1) * st + c] == 1.0f) : false;
    for (int64_t t = t1 - 1; t >= t0; --t) {
      const bool before = t > t0 ? (runs[(t - 1) * st + c] == 1.0f) : false;
      float dv = xh_nan32();
      if (data && on) dv = data[t * st + c];
      cs = cs + ((dv == dv) ? dv : 0.0f);
      if (!(dv == dv)) csr = cs;
      if (on) {
        if (run == 0) endv = (t + 1 < t1) ? (float)(t + 1 - t0) : xh_nan32();
        run++;
        if (eff) { const float e = eff[t * st + c]; effc += (e == e && e != 0.0f) ? 1 : 0; } else effc++;
        if (!before) {  // first step of the run
          k--;
          if (k < maxev) {
            const int64_t o = base + (int64_t)k * C;
            o_start[o] = (float)(t - t0);
            if (o_end) o_end[o] = endv;
            if (o_len) o_len[o] = (float)run;
            if (o_eff) o_eff[o] = (float)effc;
            if (o_sum) o_sum[o] = cs - csr;
          }
          run = 0; effc = 0;
        }
      }
      on = before;
    }
  }
}

// ---- suspicious_run (rl:1668-1757): steps that belong to a run of >= window IDENTICAL values (optionally only runs
// whose value satisfies `op thresh`).  rle_1d compares with != , so every NaN is a run of its own.  Forward march;
// when a run reaches `window` its first window-1 steps are flagged retroactively.
__global__ void __launch_bounds__(XH_BLOCK)
k_suspicious_run(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, int window, int op, float thresh,
                 uint8_t* __restrict__ out, int64_t out_st) {
  int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  float cur = 0.0f;
  int cnt = 0;
  for (int64_t t = 0; t < T; ++t) {
    const float v = x[t * st + c];
    cnt = (t > 0 && v == cur) ? cnt + 1 : 1;  // NaN == NaN is false: NaNs never extend a run
    cur = v;
    const bool okv = op < 0 ? true : xh_cmp_f32(v, op, thresh);
    const bool flag = okv && cnt >= window;
    out[t * out_st + c] = flag ? 1 : 0;
    if (flag && cnt == window)
      for (int k = 1; k < window; ++k) out[(t - k) * out_st + c] = 1;
  }
}

static int chk(const char* fn, xh_ctx* ctx, const void* x, int64_t T, int64_t C, int64_t st, int64_t sc) {
  XH_REQUIRE(ctx && x, XH_ERR_ARG, "%s: NULL argument", fn);
  XH_REQUIRE(T >= 0 && C >= 0, XH_ERR_ARG, "%s: negative shape", fn);
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT, "%s: needs a time-major view (sc == 1, st >= C)", fn);
  return XH_OK;
}

static int upload_seg(xh_ctx* ctx, size_t* cur, const int64_t* seg_off, int P, int64_t T, const char* fn,
                      const int64_t** d_seg) {
  XH_REQUIRE(seg_off && P >= 1, XH_ERR_ARG, "%s: seg_off NULL or P < 1", fn);
  for (int p = 0; p < P; ++p)
    XH_REQUIRE(seg_off[p] <= seg_off[p + 1] && seg_off[p] >= 0 && seg_off[p + 1] <= T, XH_ERR_ARG,
               "%s: seg_off must be non-decreasing within [0, T]", fn);
  void* d = nullptr;
  int rc = xh_scratch_upload(ctx, cur, seg_off, sizeof(int64_t) * (size_t)(P + 1), &d);
  if (rc) return rc;
  *d_seg = (const int64_t*)d;
  return XH_OK;
}

extern "C" {

static int spell_mask_impl(xh_ctx* ctx, const char* fn, const float* const* xs, int nvar, const double* thrs, int combine,
                           int64_t T, int64_t C, int64_t st, int64_t sc, int window, int win_reducer, int op,
                           const float* weights, float* out, int64_t out_st) {
  XH_REQUIRE(ctx && xs && thrs, XH_ERR_ARG, "%s: NULL argument", fn);
  XH_REQUIRE(nvar >= 1 && nvar <= 8, XH_ERR_LIMIT, "%s: 1 to 8 variables are supported (got %d)", fn, nvar);
  XH_REQUIRE(combine == 1 || combine == 2, XH_ERR_ARG, "%s: combine must be 1 (all) or 2 (any)", fn);
  for (int i = 0; i < nvar; ++i) {
    int rc = chk(fn, ctx, xs[i], T, C, st, sc);
    if (rc) return rc;
  }
  XH_REQUIRE(out && out_st >= C, XH_ERR_ARG, "%s: out NULL or out_st < C", fn);
  XH_REQUIRE(window >= 1, XH_ERR_ARG, "%s: window must be >= 1", fn);
  XH_REQUIRE(win_reducer >= 0 && win_reducer <= 4, XH_ERR_OP, "%s: win_reducer %d not recognized", fn, win_reducer);
  XH_REQUIRE(op >= XH_OP_GT && op <= XH_OP_NE, XH_ERR_OP, "Operation `%d` not recognized.", op);
  XH_REQUIRE(win_reducer != 4 || weights, XH_ERR_ARG, "%s: weights required for the weighted mean", fn);
  if (T == 0 || C == 0) return XH_OK;
  const float* d_w = nullptr;
  if (win_reducer == 4) {
    size_t cur = 0;
    void* d = nullptr;
    int rc = xh_scratch_upload(ctx, &cur, weights, sizeof(float) * (size_t)window, &d);
    if (rc) return rc;
    d_w = (const float*)d;
  }
  if (nvar == 1) {  // one variable, window <= 8: register ring (window.hip)
    int rr = xh_launch_spell_ring(ctx, xs[0], T, C, st, window, win_reducer, op, (float)thrs[0], d_w, out, out_st);
    if (rr != XH_ERR_NOTIMPL) return rr;
  }
  SpellVars vars;
  for (int i = 0; i < 8; ++i) {
    vars.x[i] = i < nvar ? xs[i] : nullptr;
    vars.thr[i] = i < nvar ? (float)thrs[i] : 0.0f;
  }
  hipLaunchKernelGGL(k_spell_mask, dim3((unsigned)cdiv64(C, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, vars, nvar, combine, T, C,
                     st, window, win_reducer, op, d_w, out, out_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_spell_mask(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window, int win_reducer,
                  int op, double thr, const float* weights, float* out, int64_t out_st) {
  return spell_mask_impl(ctx, "xh_spell_mask", &x, 1, &thr, 1, T, C, st, sc, window, win_reducer, op, weights, out, out_st);
}

int xh_spell_mask_multi(xh_ctx* ctx, const float* const* xs, int nvar, const double* thrs, int combine, int64_t T, int64_t C,
                        int64_t st, int64_t sc, int window, int win_reducer, int op, const float* weights, float* out,
                        int64_t out_st) {
  return spell_mask_impl(ctx, "xh_spell_mask_multi", xs, nvar, thrs, combine, T, C, st, sc, window, win_reducer, op, weights,
                         out, out_st);
}

int xh_spell_run_stats(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window, int win_reducer,
                       int op, double thr, const float* weights, int stat, const int64_t* seg_off, int P, float* out,
                       int32_t* valid_out) {
  int rc = chk("xh_spell_run_stats", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(out, XH_ERR_ARG, "xh_spell_run_stats: out is NULL");
  XH_REQUIRE(window >= 1, XH_ERR_ARG, "xh_spell_run_stats: window must be >= 1");
  XH_REQUIRE(win_reducer >= 0 && win_reducer <= 4, XH_ERR_OP, "xh_spell_run_stats: win_reducer %d not recognized", win_reducer);
  XH_REQUIRE(op >= XH_OP_GT && op <= XH_OP_NE, XH_ERR_OP, "Operation `%d` not recognized.", op);
  XH_REQUIRE(win_reducer != 4 || weights, XH_ERR_ARG, "xh_spell_run_stats: weights required for the weighted mean");
  XH_REQUIRE(stat >= XH_RUN_MAX && stat <= XH_RUN_STD, XH_ERR_OP, "xh_spell_run_stats: statistic %d not supported", stat);
  size_t cur = 0;
  const int64_t* d_seg = nullptr;
  rc = upload_seg(ctx, &cur, seg_off, P, T, "xh_spell_run_stats", &d_seg);
  if (rc) return rc;
  const float* d_w = nullptr;
  if (win_reducer == 4) {
    void* d = nullptr;
    rc = xh_scratch_upload(ctx, &cur, weights, sizeof(float) * (size_t)window, &d);
    if (rc) return rc;
    d_w = (const float*)d;
  }
  if (C == 0) return XH_OK;
  return xh_launch_spell_runs(ctx, x, T, C, st, window, win_reducer, op, (float)thr, d_w, stat, d_seg, P, out, valid_out);
}

int xh_runs_with_holes(xh_ctx* ctx, const float* start, const float* stop, int64_t T, int64_t C, int64_t st, int64_t sc,
                       int window_start, int window_stop, float* out, int64_t out_st) {
  int rc = chk("xh_runs_with_holes", ctx, start, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(out && out_st >= C, XH_ERR_ARG, "xh_runs_with_holes: out NULL or out_st < C");
  XH_REQUIRE(window_start >= 1 && window_stop >= 1, XH_ERR_ARG, "xh_runs_with_holes: windows must be >= 1");
  if (T == 0 || C == 0) return XH_OK;
  if (window_start <= 64 && window_stop <= 64) {
    // VEC = 4 only when that still leaves a few workgroups per CU (no time chunking is possible here)
    const bool v4 = xh_pick_vec(start, C, st) == 4 && (!stop || xh_pick_vec(stop, C, st) == 4) && xh_pick_vec(out, C, out_st) == 4 &&
                    cdiv64(cdiv64(C, 4), XH_BLOCK) >= 4 * (int64_t)ctx->num_cu;
    const dim3 grid((unsigned)cdiv64(cdiv64(C, v4 ? 4 : 1), XH_BLOCK));
#define XH_RWH(V, TW)                                                                                                   \
  hipLaunchKernelGGL((k_runs_with_holes_fwd<V, TW>), grid, dim3(XH_BLOCK), 0, ctx->stream, start, stop ? stop : start, T, C, st, \
                     st, window_start, window_stop, out, out_st)
    if (v4) { if (stop) XH_RWH(4, true); else XH_RWH(4, false); }
    else { if (stop) XH_RWH(1, true); else XH_RWH(1, false); }
#undef XH_RWH
    XH_LAUNCH_CHECK();
    return XH_OK;
  }
  hipLaunchKernelGGL(k_runs_with_holes, dim3((unsigned)cdiv64(C, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, start,
                     stop ? stop : start, T, C, st, st, window_start, window_stop, stop ? 0 : 1, out, out_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_keep_longest_run(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const int64_t* seg_off,
                        int P, float* out, int64_t out_st) {
  int rc = chk("xh_keep_longest_run", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(out && out_st >= C, XH_ERR_ARG, "xh_keep_longest_run: out NULL or out_st < C");
  size_t cur = 0;
  const int64_t* d_seg = nullptr;
  rc = upload_seg(ctx, &cur, seg_off, P, T, "xh_keep_longest_run", &d_seg);
  if (rc) return rc;
  if (C == 0) return XH_OK;
  XH_REQUIRE(seg_off[0] == 0 && seg_off[P] == T, XH_ERR_ARG, "xh_keep_longest_run: segments must cover [0, T)");
  if (xh_pick_vec(x, C, st) == 4 && xh_pick_vec(out, C, out_st) == 4 && cdiv64(cdiv64(C, 4), XH_BLOCK) >= 2 * (int64_t)ctx->num_cu)
    hipLaunchKernelGGL((k_keep_longest_run<4>), dim3((unsigned)cdiv64(cdiv64(C, 4), XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, x, C,
                       st, d_seg, P, out, out_st);
  else
    hipLaunchKernelGGL((k_keep_longest_run<1>), dim3((unsigned)cdiv64(C, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, x, C, st,
                       d_seg, P, out, out_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_season(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window,
              const int64_t* seg_off, const int32_t* mid_idx, int P, float* start_out, float* end_out, float* len_out) {
  int rc = chk("xh_season", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(start_out && end_out && len_out, XH_ERR_ARG, "xh_season: NULL output");
  XH_REQUIRE(window >= 1, XH_ERR_ARG, "xh_season: window must be >= 1");
  size_t cur = 0;
  const int64_t* d_seg = nullptr;
  rc = upload_seg(ctx, &cur, seg_off, P, T, "xh_season", &d_seg);
  if (rc) return rc;
  void* d_mid = nullptr;
  if (mid_idx) {
    rc = xh_scratch_upload(ctx, &cur, mid_idx, sizeof(int32_t) * (size_t)P, &d_mid);
    if (rc) return rc;
  }
  if (C == 0) return XH_OK;
  if (window >= 2 && xh_pick_vec(x, C, st) == 4)
    hipLaunchKernelGGL((k_season_w<4>), dim3((unsigned)cdiv64(cdiv64(C, 4), XH_BLOCK), (unsigned)(P > 4096 ? 4096 : P)),
                       dim3(XH_BLOCK), 0, ctx->stream, x, C, st, window, d_seg, (const int32_t*)d_mid, mid_idx ? 1 : 0, P,
                       start_out, end_out, len_out);
  else if (window >= 2)
    hipLaunchKernelGGL((k_season_w<1>), dim3((unsigned)cdiv64(C, XH_BLOCK), (unsigned)(P > 4096 ? 4096 : P)), dim3(XH_BLOCK), 0,
                       ctx->stream, x, C, st, window, d_seg, (const int32_t*)d_mid, mid_idx ? 1 : 0, P, start_out, end_out,
                       len_out);
  else if (xh_pick_vec(x, C, st) == 4)
    hipLaunchKernelGGL((k_season<4>), dim3((unsigned)cdiv64(cdiv64(C, 4), XH_BLOCK), (unsigned)(P > 4096 ? 4096 : P)),
                       dim3(XH_BLOCK), 0, ctx->stream, x, C, st, window, d_seg, (const int32_t*)d_mid, mid_idx ? 1 : 0, P,
                       start_out, end_out, len_out);
  else
    hipLaunchKernelGGL((k_season<1>), dim3((unsigned)cdiv64(C, XH_BLOCK), (unsigned)(P > 4096 ? 4096 : P)), dim3(XH_BLOCK), 0,
                       ctx->stream, x, C, st, window, d_seg, (const int32_t*)d_mid, mid_idx ? 1 : 0, P, start_out, end_out,
                       len_out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_max_run_sum(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window,
                   const int64_t* seg_off, int P, int cut_at_segments, float* out) {
  int rc = chk("xh_max_run_sum", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(out, XH_ERR_ARG, "xh_max_run_sum: out is NULL");
  XH_REQUIRE(window >= 1, XH_ERR_ARG, "xh_max_run_sum: window must be >= 1");
  size_t cur = 0;
  const int64_t* d_seg = nullptr;
  rc = upload_seg(ctx, &cur, seg_off, P, T, "xh_max_run_sum", &d_seg);
  if (rc) return rc;
  if (C == 0) return XH_OK;
  const unsigned py = cut_at_segments ? (unsigned)(P > 4096 ? 4096 : P) : 1u;
  if (!cut_at_segments)
    XH_REQUIRE(seg_off[0] == 0 && seg_off[P] == T, XH_ERR_ARG, "xh_max_run_sum: resample-after mode needs segments covering [0, T)");
  const bool v4 = xh_pick_vec(x, C, st) == 4 && (cut_at_segments || cdiv64(cdiv64(C, 4), XH_BLOCK) >= 2 * (int64_t)ctx->num_cu);
  dim3 grid((unsigned)cdiv64(cdiv64(C, v4 ? 4 : 1), XH_BLOCK), py);
  if (v4) {
    if (cut_at_segments) hipLaunchKernelGGL((k_max_run_sum<4, true>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, window, d_seg, P, out);
    else hipLaunchKernelGGL((k_max_run_sum<4, false>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, window, d_seg, P, out);
  } else {
    if (cut_at_segments) hipLaunchKernelGGL((k_max_run_sum<1, true>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, window, d_seg, P, out);
    else hipLaunchKernelGGL((k_max_run_sum<1, false>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, window, d_seg, P, out);
  }
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_run_events(xh_ctx* ctx, const float* runs, const float* eff, const float* data, int64_t T, int64_t C, int64_t st,
                  int64_t sc, const int64_t* seg_off, int P, int maxev, float* start_out, float* end_out, float* len_out,
                  float* eff_out, float* sum_out) {
  int rc = chk("xh_run_events", ctx, runs, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(start_out, XH_ERR_ARG, "xh_run_events: start_out is NULL");
  XH_REQUIRE(maxev >= 0, XH_ERR_ARG, "xh_run_events: maxev must be >= 0");
  XH_REQUIRE(!sum_out || data, XH_ERR_ARG, "xh_run_events: sum_out needs data");
  size_t cur = 0;
  const int64_t* d_seg = nullptr;
  rc = upload_seg(ctx, &cur, seg_off, P, T, "xh_run_events", &d_seg);
  if (rc) return rc;
  if (C == 0 || maxev == 0) return XH_OK;
  hipLaunchKernelGGL(k_run_events, dim3((unsigned)cdiv64(C, XH_BLOCK), (unsigned)(P > 4096 ? 4096 : P)), dim3(XH_BLOCK), 0,
                     ctx->stream, runs, eff, data, C, st, d_seg, P, maxev, start_out, end_out, len_out, eff_out, sum_out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_suspicious_run(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window, int op,
                      double thresh, uint8_t* out, int64_t out_st) {
  int rc = chk("xh_suspicious_run", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(out && out_st >= C, XH_ERR_ARG, "xh_suspicious_run: out NULL or out_st < C");
  XH_REQUIRE(window >= 1, XH_ERR_ARG, "xh_suspicious_run: window must be >= 1");
  XH_REQUIRE(op >= -1 && op <= XH_OP_NE, XH_ERR_OP, "Operation `%d` not recognized.", op);
  if (C == 0 || T == 0) return XH_OK;
  hipLaunchKernelGGL(k_suspicious_run, dim3((unsigned)cdiv64(C, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, window,
                     op, (float)thresh, out, out_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // extern "C"
