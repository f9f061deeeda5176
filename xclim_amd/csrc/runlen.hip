// runlen.hip — run-length family: one marching state machine per cell along time.
//
// Reference semantics (indices/run_length.py):
//   _cumsum_reset_np  rl:143-151   c[t] = b[t] * (c[t-1] + 1)           (NaN -> 0 first, rl:208)
//   rle               rl:223-272   length kept at the first/last element of a run, NaN inside, 0 outside;
//                                  QUIRK restated on purpose: the kept element is masked by
//                                  `da.shift(-1, fill_value=0) == 0` on the NaN-carrying input, so a run whose
//                                  outer neighbour (before its first element for index="first", after its last
//                                  for index="last") is NaN loses its length ("invisible" run).
//   rle_statistics    rl:275-335   reducer over runs >= window, 0 when none
//   windowed_run_*    rl:381-488
//   _boundary_run     rl:543-640   first_run / last_run incl. the argmax == argmin "no run" test
// Layout: time-major (T, C); a lane owns VEC consecutive cells; state lives in registers; HBM traffic is the
// compulsory 4 B per cell-timestep read plus the (P, C) outputs.
#include "common.h"

#include "runacc.h"

// per-cell run state
struct RunState {
  int run;       // current run length (0 = not in a run)
  bool vis;      // index="first": outer neighbour before the run start was not NaN
  bool prevnan;  // previous element was NaN (mask mode only)
  int startp;    // period of the run's first element (resample-after mode)
};

template <int VEC, bool CUT, int SG = 0>
__global__ void __launch_bounds__(XH_BLOCK)
k_run_stats(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, int fused_op, float thr, int window, int stat,
            int index_first, const int64_t* __restrict__ seg_off, int P, float* __restrict__ out,
            int32_t* __restrict__ valid_out) {
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const bool fused = fused_op >= 0;

  if (CUT) {
    for (int p = blockIdx.y; p < P; p += gridDim.y) {
      int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
      RunAcc acc[VEC];
      RunState s[VEC];
      int nvalid[VEC], plainsum[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        acc_reset(acc[i]);
        s[i].run = 0; s[i].vis = true; s[i].prevnan = false; s[i].startp = 0;
        nvalid[i] = 0; plainsum[i] = 0;
      }
      xh_march_rows<VEC, 8>(x + c, st, t0, t1, [&](int64_t, const VecF<VEC>& xv) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          float v = xv.v[i];
          bool isn = v != v;
          bool on = fused ? xh_cmp_f32(v, fused_op, thr) : (v > 0.0f);
          bool masknan = (!fused) && isn;
          nvalid[i] += isn ? 0 : 1;
          plainsum[i] += on ? 1 : 0;
          // branch-free state machine (the serial march is VALU-issue bound otherwise)
          bool visible = index_first >= 2 ? true : (index_first ? s[i].vis : !masknan);
          bool ended = !on && s[i].run > 0;
          int len = (ended && visible && s[i].run >= window) ? s[i].run : 0;
          acc_add_if<SG>(acc[i], len);
          s[i].vis = (on && s[i].run == 0) ? !s[i].prevnan : s[i].vis;
          s[i].run = on ? s[i].run + 1 : 0;
          s[i].prevnan = masknan;
        }
      });
      int64_t o = (int64_t)p * C + c;
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        if (s[i].run > 0) {
          bool visible = index_first == 1 ? s[i].vis : true;  // beyond the segment end: shift fill_value 0
          if (visible && s[i].run >= window) acc_add(acc[i], s[i].run);
        }
        float r = acc_result(acc[i], stat, plainsum[i]);
        // index mode 3 = statistics_run_1d (rl:1408-1437): with a NaN step in the series and no qualifying run the
        // early `return 0` is skipped (NaN * length < window is False) and the nan-reducer of an all-NaN list gives NaN
        if (index_first == 3 && acc[i].cnt == 0 && nvalid[i] < (int)(t1 - t0) && stat != XH_RUN_COUNT && stat != XH_RUN_SUM)
          r = xh_nan32();  // nanmax / nanmin / nanmean / nanstd of nothing; nansum of nothing is 0
        out[o + i] = r;
        if (valid_out) valid_out[o + i] = nvalid[i];
      }
    }
  } else {
    // resample AFTER run length: runs cross period edges, attributed to the period of the indexed element
    RunAcc acc[VEC];
    RunState s[VEC];
    int accp[VEC], nvalid[VEC], plainsum[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      acc_reset(acc[i]);
      s[i].run = 0; s[i].vis = true; s[i].prevnan = false; s[i].startp = 0;
      accp[i] = 0; nvalid[i] = 0; plainsum[i] = 0;
    }
    int pt = 0;  // period of the current time step (uniform across lanes)
    int64_t tend = seg_off[P];
    int pprev = 0;                   // period of the previous time step
    int64_t next_edge = seg_off[1];  // kept in a scalar register: one s_load per period, not per step
    xh_march_rows<VEC, 8>(x + c, st, seg_off[0], tend, [&](int64_t t, const VecF<VEC>& xv) {
      while (t >= next_edge) {
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          if (valid_out) valid_out[(int64_t)pt * C + c + i] = nvalid[i];
          if (stat == XH_RUN_PLAINSUM) out[(int64_t)pt * C + c + i] = (float)plainsum[i];
          nvalid[i] = 0; plainsum[i] = 0;
        }
        pt++;
        next_edge = seg_off[pt + 1];
        // early flush, all lanes together: a cell that is not inside a run has every period before pt final (any later
        // run is attributed to pt or later), so the lazy per-run flush below only fires for runs that span an edge
        if (stat != XH_RUN_PLAINSUM) {
#pragma unroll
          for (int i = 0; i < VEC; ++i) {
            if (s[i].run == 0) {
              while (accp[i] < pt) {
                out[(int64_t)accp[i] * C + c + i] = acc_result(acc[i], stat, 0);
                acc_reset(acc[i]);
                accp[i]++;
              }
            }
          }
        }
      }
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float v = xv.v[i];
        bool isn = v != v;
        bool on = fused ? xh_cmp_f32(v, fused_op, thr) : (v > 0.0f);
        bool masknan = (!fused) && isn;
        nvalid[i] += isn ? 0 : 1;
        plainsum[i] += on ? 1 : 0;
        // branch-free step; only a qualifying run that ENDS here takes the (rare, divergent) attribution path
        const bool visible = index_first >= 2 ? true : (index_first ? s[i].vis : !masknan);
        const bool qual = !on && visible && s[i].run >= window && stat != XH_RUN_PLAINSUM;  // window >= 1: run > 0
        // the run ended at t-1: index="last" attributes it to the period of step t-1 (pprev), "first" to its start's
        const int pa = index_first ? s[i].startp : pprev;
        if (qual && accp[i] < pa) {
          do {
            out[(int64_t)accp[i] * C + c + i] = acc_result(acc[i], stat, 0);
            acc_reset(acc[i]);
            accp[i]++;
          } while (accp[i] < pa);
        }
        acc_add_if<SG>(acc[i], qual ? s[i].run : 0);
        const bool starts = on && s[i].run == 0;
        s[i].vis = starts ? !s[i].prevnan : s[i].vis;
        s[i].startp = starts ? pt : s[i].startp;
        s[i].run = on ? s[i].run + 1 : 0;
        s[i].prevnan = masknan;
      }
      pprev = pt;
    });
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      if (s[i].run > 0 && stat != XH_RUN_PLAINSUM) {
        bool visible = index_first == 1 ? s[i].vis : true;
        if (visible && s[i].run >= window) {
          int pa = index_first ? s[i].startp : pt;
          while (accp[i] < pa) {
            out[(int64_t)accp[i] * C + c + i] = acc_result(acc[i], stat, 0);
            acc_reset(acc[i]);
            accp[i]++;
          }
          acc_add_if<SG>(acc[i], s[i].run);
        }
      }
      if (stat != XH_RUN_PLAINSUM) {
        while (accp[i] < P) {
          out[(int64_t)accp[i] * C + c + i] = acc_result(acc[i], stat, 0);
          acc_reset(acc[i]);
          accp[i]++;
        }
      }
      // trailing periods (pt .. P-1): the last one holds the live counters, later ones are empty
      for (int p = pt; p < P; ++p) {
        if (valid_out) valid_out[(int64_t)p * C + c + i] = (p == pt) ? nvalid[i] : 0;
        if (stat == XH_RUN_PLAINSUM) out[(int64_t)p * C + c + i] = (p == pt) ? (float)plainsum[i] : 0.0f;
      }
    }
  }
}

// Run statistics of a 1 / 0 / NaN MASK with the runs cut at the period edges — the common case behind rle_statistics /
// windowed_run_count / windowed_run_events on a precomputed condition (rl:275-488).  The generic kernel above resolves
// the index mode, the fused compare and the statistic per element (PMC: 122 VALU + 115 SALU wave-instructions per row of
// 4 cells, 0.50 ms at 365 x 1440 x 720); here they are template parameters and the state machine is written for the
// minimum of selects: ~13 VALU per cell-step for the maximum.
//   IDX 0: index="last" (a run whose NEXT step is NaN is dropped), 1: index="first" (a run that STARTS right after a
//   NaN step is dropped), 2 / 3: the 1-D ufunc paths (NaN steps only break runs; 3 = statistics_run_1d: NaN result when
//   the series has NaN steps and no qualifying run) — rl:223-272, 1334-1437 and DESIGN.md "two reference paths".
//   SG 1: max, 2: sum / count / mean, 0: every field (min, std).
//   FUSED: the condition is `sgn * x > thr` on the data itself (xh_one_cmp form of > < >= <=: spell_length, the
//   *_spell_* indices with window 1, gen:543-585, 1204-1252); a NaN step is then just a False step (IDX 2).
template <int VEC, int IDX, int SG, bool FUSED = false>
__global__ void __launch_bounds__(XH_BLOCK)
k_run_stats_mask(const float* __restrict__ x, int64_t C, int64_t st, int window, int stat, const int64_t* __restrict__ seg_off,
                 int P, float* __restrict__ out, int32_t* __restrict__ valid_out, float sgn = 1.0f, float thr = 0.0f) {
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    const int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    RunAcc acc[VEC];
    int run[VEC], nnan[VEC];
    bool vis[VEC], prevnan[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      acc_reset(acc[i]);
      run[i] = 0; nnan[i] = 0; vis[i] = true; prevnan[i] = false;
    }
    xh_march_rows<VEC, 8>(x + c, st, t0, t1, [&](int64_t, const VecF<VEC>& xv) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float v = xv.v[i];
        const bool isn = v != v, on = FUSED ? (v * sgn > thr) : (v > 0.0f);
        nnan[i] += isn ? 1 : 0;
        int len = on ? 0 : run[i];                    // length of the run that ended with the previous step (0: none)
        if (IDX == 0) len = isn ? 0 : len;            // hidden by the NaN that follows it
        if (IDX == 1) len = vis[i] ? len : 0;         // hidden by the NaN that preceded it
        len = len >= window ? len : 0;
        if (SG == 1) acc[i].mx = len > acc[i].mx ? len : acc[i].mx;
        else acc_add_if<SG>(acc[i], len);
        if (IDX == 1) {
          vis[i] = (on && run[i] == 0) ? !prevnan[i] : vis[i];
          prevnan[i] = isn;
        }
        run[i] = on ? run[i] + 1 : 0;
      }
    });
    const int64_t o = (int64_t)p * C + c;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      if (run[i] > 0 && run[i] >= window && (IDX == 1 ? vis[i] : true)) {  // beyond the segment end: shift fill_value 0
        if (SG == 1) acc[i].mx = run[i] > acc[i].mx ? run[i] : acc[i].mx;
        else acc_add(acc[i], run[i]);
      }
      if (SG == 1) acc[i].cnt = acc[i].mx > 0 ? 1 : 0;
      float r = acc_result(acc[i], stat, 0);
      if (IDX == 3 && acc[i].cnt == 0 && nnan[i] > 0 && stat != XH_RUN_COUNT && stat != XH_RUN_SUM) r = xh_nan32();
      out[o + i] = r;
      if (valid_out) valid_out[o + i] = (int)(t1 - t0) - nnan[i];
    }
  }
}

// Fast path of maximum_consecutive_{dry,wet}_days & friends (gen:543-585 with window == 1, reducer "max", resample
// before run length): the mask comes from a compare, so it has no NaN and every run is visible; the longest run is
// max over t of the running length, no run-end bookkeeping at all.  ~6 VALU ops per cell-step -> HBM bound.
template <int VEC, int OP>
__global__ void __launch_bounds__(XH_BLOCK)
k_run_max_fused(const float* __restrict__ x, int64_t C, int64_t st, float thr, int window,
                const int64_t* __restrict__ seg_off, int P, float* __restrict__ out, int32_t* __restrict__ valid_out) {
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    int run[VEC], mx[VEC], nvalid[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) run[i] = 0, mx[i] = 0, nvalid[i] = 0;
    xh_march_rows<VEC, 8>(x + c, st, t0, t1, [&](int64_t, const VecF<VEC>& xv) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        float v = xv.v[i];
        run[i] = xh_cmp_t<OP>(v, thr) ? run[i] + 1 : 0;
        mx[i] = run[i] > mx[i] ? run[i] : mx[i];
        nvalid[i] += (v == v) ? 1 : 0;
      }
    });
    int64_t o = (int64_t)p * C + c;
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      out[o + i] = mx[i] >= window ? (float)mx[i] : 0.0f;
      if (valid_out) valid_out[o + i] = nvalid[i];
    }
  }
}

// first_run / last_run (rl:543-740).  d[t] = (c[t] >= window) with c the reset-cumsum run towards the far end of
// the run (index="first": remaining length from t on -> march backwards; "last": length ending at t -> forwards);
// window == 1: d = (mask > 0) [fillna(0)].  Result per period: first (last) t with d == 1, relative to the
// period start; NaN when d is constant over the period (argmax == argmin, rl:603-605: also the all-True case).
template <int VEC, bool FIRST, bool CUT>
__global__ void __launch_bounds__(XH_BLOCK)
k_boundary_run(const float* __restrict__ x, int64_t C, int64_t st, int fused_op, float thr, int window,
               const int64_t* __restrict__ seg_off, int P, float* __restrict__ out, int32_t* __restrict__ valid_out) {
  // a lane owns VEC cells; rows in double-buffered batches of 8 (xh_march_rows / _rev)
  int64_t c = ((int64_t)blockIdx.x * XH_BLOCK + threadIdx.x) * VEC;
  if (c >= C) return;
  const bool fused = fused_op >= 0;
  int run[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) run[i] = 0;
  int pstep = CUT ? (int)gridDim.y : 1;
  int pbeg = CUT ? (int)blockIdx.y : 0;
  for (int pp = pbeg; pp < P; pp += pstep) {
    int p = FIRST ? (P - 1 - pp) : pp;  // FIRST marches backwards through periods
    int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    int ones[VEC], nvalid[VEC], hit[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      if (CUT) run[i] = 0;
      ones[i] = 0; nvalid[i] = 0; hit[i] = -1;
    }
    auto step = [&](int64_t t, const VecF<VEC>& xv) {
      const int rel = (int)(t - t0);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float v = xv.v[i];
        const bool on = fused ? xh_cmp_f32(v, fused_op, thr) : (v > 0.0f);
        nvalid[i] += (v == v) ? 1 : 0;
        run[i] = on ? run[i] + 1 : 0;
        const bool h = run[i] >= window;
        hit[i] = h ? rel : hit[i];
        ones[i] += h ? 1 : 0;
      }
    };
    if (FIRST) xh_march_rows_rev<VEC, 8>(x + c, st, t0, t1, step);
    else xh_march_rows<VEC, 8>(x + c, st, t0, t1, step);
    const int len = (int)(t1 - t0);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      out[(int64_t)p * C + c + i] = (ones[i] == 0 || ones[i] == len) ? xh_nan32() : (float)hit[i];
      if (valid_out) valid_out[(int64_t)p * C + c + i] = nvalid[i];
    }
  }
}

// _cumsum_reset (full-shape output).  MODE 0: cumsum_reset; MODE 1: rle.
template <int MODE>
__global__ void __launch_bounds__(XH_BLOCK)
k_cumsum_rle(const float* __restrict__ x, int64_t T, int64_t C, int64_t st, int index_first, float* __restrict__ out,
             int64_t out_st) {
  int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  if (MODE == 0) {
    float run = 0.f;
    if (index_first) {
      for (int64_t t = T - 1; t >= 0; --t) {
        float v = x[t * st + c];
        float b = (v == v) ? v : 0.f;
        run = b * (run + 1.0f);  // rl:150 — exact for binary input
        out[t * out_st + c] = run;
      }
    } else {
      for (int64_t t = 0; t < T; ++t) {
        float v = x[t * st + c];
        float b = (v == v) ? v : 0.f;
        run = b * (run + 1.0f);
        out[t * out_st + c] = run;
      }
    }
  } else {
    // rle: march in the direction that finishes at the indexed element.
    // index="first": reversed marching (t = T-1 .. 0), the kept element is the last one visited of each run,
    // whose outer neighbour is x[t-1]; index="last": forward marching, outer neighbour x[t+1].
    float run = 0.f;
    const float nanv = xh_nan32();
    if (index_first) {
      float cur = T > 0 ? x[(T - 1) * st + c] : 0.f;
      for (int64_t t = T - 1; t >= 0; --t) {
        float nxt = t > 0 ? x[(t - 1) * st + c] : 0.f;  // shift(-1, fill_value=0) in reversed order
        float b = (cur == cur) ? cur : 0.f;
        run = b * (run + 1.0f);
        float o = (nxt == 0.0f) ? run : nanv;  // rl:264
        o = (cur > 0.0f) ? o : 0.0f;           // rl:265
        out[t * out_st + c] = o;
        cur = nxt;
      }
    } else {
      float cur = T > 0 ? x[c] : 0.f;
      for (int64_t t = 0; t < T; ++t) {
        float nxt = t + 1 < T ? x[(t + 1) * st + c] : 0.f;
        float b = (cur == cur) ? cur : 0.f;
        run = b * (run + 1.0f);
        float o = (nxt == 0.0f) ? run : nanv;
        o = (cur > 0.0f) ? o : 0.0f;
        out[t * out_st + c] = o;
        cur = nxt;
      }
    }
  }
}

static int check_tc2(const char* fn, xh_ctx* ctx, const void* x, int64_t T, int64_t C, int64_t st, int64_t sc) {
  XH_REQUIRE(ctx && x, XH_ERR_ARG, "%s: NULL argument", fn);
  XH_REQUIRE(T >= 0 && C >= 0, XH_ERR_ARG, "%s: negative shape", fn);
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT,
             "%s: streaming kernels need a time-major view (sc == 1, st >= C); got st=%lld sc=%lld — transpose first",
             fn, (long long)st, (long long)sc);
  return XH_OK;
}

// Percentile-spell indices (warm / cold_spell_duration_index, indices/_multivariate.py:66-152, 1693-1793) with
// resample_before_rl: the daily condition x[t] op table[tidx[t]] (fp64 compare against the per-doy percentile, i.e.
// compare(da, op, resample_doy(per, da))) feeds the run state machine directly — neither the (T, C) fp64 threshold
// field of the reference nor a mask is written.  A compare never yields NaN, so every run is visible (rl:223-272).
// One lane per cell; x rows and table rows are loaded in batches of 8 before any use.
__global__ void __launch_bounds__(XH_BLOCK)
k_run_stats_doy(const float* __restrict__ x, int64_t C, int64_t st, int op, const double* __restrict__ table,
                const int32_t* __restrict__ tidx, int window, int stat, const int64_t* __restrict__ seg_off, int P,
                float* __restrict__ out, int32_t* __restrict__ valid_out) {
  const int64_t c = (int64_t)blockIdx.x * XH_BLOCK + threadIdx.x;
  if (c >= C) return;
  for (int p = blockIdx.y; p < P; p += gridDim.y) {
    const int64_t t0 = seg_off[p], t1 = seg_off[p + 1];
    RunAcc acc;
    acc_reset(acc);
    int run = 0, nvalid = 0, plainsum = 0;
    auto step = [&](float xv, double tv) {
      const bool on = xh_cmp_f64((double)xv, op, tv);
      nvalid += (xv == xv) ? 1 : 0;
      plainsum += on ? 1 : 0;
      const int len = (!on && run >= window) ? run : 0;  // a run of at least `window` (>= 1) steps ended at t - 1
      acc_add_if<0>(acc, len);
      run = on ? run + 1 : 0;
    };
    int64_t t = t0;
    for (; t + 8 <= t1; t += 8) {
      int r[8];
      float xv[8];
      double tv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) r[u] = tidx[t + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = x[(t + u) * st + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) tv[u] = table[(int64_t)r[u] * C + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) step(xv[u], tv[u]);
    }
    for (; t < t1; ++t) step(x[t * st + c], table[(int64_t)tidx[t] * C + c]);
    if (run >= window && run > 0) acc_add(acc, run);  // run cut by the period end
    const int64_t o = (int64_t)p * C + c;
    out[o] = acc_result(acc, stat, plainsum);
    if (valid_out) valid_out[o] = nvalid;
  }
}

extern "C" {

int xh_cumsum_reset(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int index_first,
                    float* out, int64_t out_st) {
  int rc = check_tc2("xh_cumsum_reset", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(out && out_st >= C, XH_ERR_ARG, "xh_cumsum_reset: out NULL or out_st < C");
  if (T == 0 || C == 0) return XH_OK;
  hipLaunchKernelGGL((k_cumsum_rle<0>), dim3((unsigned)cdiv64(C, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st,
                     index_first, out, out_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_rle(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int index_first, float* out,
           int64_t out_st) {
  int rc = check_tc2("xh_rle", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(out && out_st >= C, XH_ERR_ARG, "xh_rle: out NULL or out_st < C");
  if (T == 0 || C == 0) return XH_OK;
  hipLaunchKernelGGL((k_cumsum_rle<1>), dim3((unsigned)cdiv64(C, XH_BLOCK)), dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st,
                     index_first, out, out_st);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_run_stats(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int fused_op, double thr,
                 int window, int stat, int index_first, const int64_t* seg_off, int P, int cut_at_segments, float* out,
                 int32_t* valid_out) {
  int rc = check_tc2("xh_run_stats", ctx, x, T, C, st, sc);
  if (rc) return rc;
  XH_REQUIRE(fused_op <= XH_OP_NE, XH_ERR_OP, "Operation `%d` not recognized.", fused_op);
  XH_REQUIRE(window >= 1, XH_ERR_ARG, "xh_run_stats: window must be >= 1");
  XH_REQUIRE(stat >= XH_RUN_MAX && stat <= XH_RUN_PLAINSUM, XH_ERR_OP, "xh_run_stats: stat %d not recognized", stat);
  XH_REQUIRE(out && seg_off && P >= 1, XH_ERR_ARG, "xh_run_stats: NULL out/seg_off or P < 1");
  for (int p = 0; p < P; ++p)
    XH_REQUIRE(seg_off[p] <= seg_off[p + 1] && seg_off[p] >= 0 && seg_off[p + 1] <= T, XH_ERR_ARG,
               "xh_run_stats: seg_off must be non-decreasing within [0, T]");
  size_t cur = 0;
  void* d = nullptr;
  rc = xh_scratch_upload(ctx, &cur, seg_off, sizeof(int64_t) * (size_t)(P + 1), &d);
  if (rc) return rc;
  const int64_t* d_seg = (const int64_t*)d;
  if (C == 0) return XH_OK;
  unsigned py = (unsigned)(P > 4096 ? 4096 : P);
  if (stat == XH_RUN_FIRST || stat == XH_RUN_LAST) {
    const int bvec = (xh_pick_vec(x, C, st) == 4 && cdiv64(cdiv64(C, 4), XH_BLOCK) * (cut_at_segments ? py : 1u) >= 2u * (unsigned)ctx->num_cu) ? 4 : 1;
    dim3 grid((unsigned)cdiv64(cdiv64(C, bvec), XH_BLOCK), cut_at_segments ? py : 1u);
#define XH_BR(V, F, CU)                                                                                                 \
  hipLaunchKernelGGL((k_boundary_run<V, F, CU>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, fused_op, (float)thr, window, \
                     d_seg, P, out, valid_out)
    const bool first = stat == XH_RUN_FIRST;
    if (bvec == 4) {
      if (first) { if (cut_at_segments) XH_BR(4, true, true); else XH_BR(4, true, false); }
      else { if (cut_at_segments) XH_BR(4, false, true); else XH_BR(4, false, false); }
    } else {
      if (first) { if (cut_at_segments) XH_BR(1, true, true); else XH_BR(1, true, false); }
      else { if (cut_at_segments) XH_BR(1, false, true); else XH_BR(1, false, false); }
    }
#undef XH_BR
    XH_LAUNCH_CHECK();
    return XH_OK;
  }
  int vec = xh_pick_vec(x, C, st);
  if (cut_at_segments && fused_op >= 0 && stat == XH_RUN_MAX) {
    dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), py);
#define XH_RMF(V, O)                                                                                                   \
  case O:                                                                                                              \
    hipLaunchKernelGGL((k_run_max_fused<V, O>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, (float)thr, window, d_seg, \
                       P, out, valid_out);                                                                             \
    break;
    if (vec == 4) {
      switch (fused_op) { XH_RMF(4, XH_OP_GT) XH_RMF(4, XH_OP_LT) XH_RMF(4, XH_OP_GE) XH_RMF(4, XH_OP_LE) XH_RMF(4, XH_OP_EQ)
        XH_RMF(4, XH_OP_NE) }
    } else {
      switch (fused_op) { XH_RMF(1, XH_OP_GT) XH_RMF(1, XH_OP_LT) XH_RMF(1, XH_OP_GE) XH_RMF(1, XH_OP_LE) XH_RMF(1, XH_OP_EQ)
        XH_RMF(1, XH_OP_NE) }
    }
#undef XH_RMF
    XH_LAUNCH_CHECK();
    return XH_OK;
  }
  const XhOneCmp oc = xh_one_cmp(fused_op, (float)thr);
  if (cut_at_segments && (fused_op < 0 || oc.ok) && stat != XH_RUN_PLAINSUM && index_first >= 0 && index_first <= 3) {
    // a mask (or a one-compare condition on the data), runs cut at the period edges: specialised state machine (index
    // mode and statistic group at compile time)
    dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), py);
    const int sg = stat == XH_RUN_MAX ? 1 : ((stat == XH_RUN_SUM || stat == XH_RUN_COUNT || stat == XH_RUN_MEAN) ? 2 : 0);
#define XH_RSM(V, I, G)                                                                                                \
  hipLaunchKernelGGL((k_run_stats_mask<V, I, G>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, window, stat, d_seg, P, out, \
                     valid_out, 1.0f, 0.0f)
#define XH_RSF(V, I, G)                                                                                                \
  hipLaunchKernelGGL((k_run_stats_mask<V, I, G, true>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, window, stat, d_seg, P, \
                     out, valid_out, oc.sgn, oc.thr)
#define XH_RSM_G(V, I) { if (sg == 1) XH_RSM(V, I, 1); else if (sg == 2) XH_RSM(V, I, 2); else XH_RSM(V, I, 0); }
#define XH_RSF_G(V, I) { if (sg == 1) XH_RSF(V, I, 1); else if (sg == 2) XH_RSF(V, I, 2); else XH_RSF(V, I, 0); }
#define XH_RSM_I(V)                                                                         \
  {                                                                                         \
    if (fused_op >= 0) { if (index_first == 3) XH_RSF_G(V, 3) else XH_RSF_G(V, 2) }         \
    else if (index_first == 0) XH_RSM_G(V, 0) else if (index_first == 1) XH_RSM_G(V, 1)     \
    else if (index_first == 2) XH_RSM_G(V, 2) else XH_RSM_G(V, 3)                           \
  }
    if (vec == 4) XH_RSM_I(4) else XH_RSM_I(1)
#undef XH_RSM_I
#undef XH_RSM_G
#undef XH_RSF_G
#undef XH_RSM
#undef XH_RSF
    XH_LAUNCH_CHECK();
    return XH_OK;
  }
  if (cut_at_segments) {
    dim3 grid((unsigned)cdiv64(cdiv64(C, vec), XH_BLOCK), py);
    // stat group (compile time): the accumulator fields the statistic does not read are not maintained
    // (measured at 365 x 1440 x 720: max 0.55 -> 0.50 ms; the sum / count group came out SLOWER than the generic body,
    //  0.68 vs 0.54 ms, so only max is specialised)
    const int sg = stat == XH_RUN_MAX ? 1 : 0;
#define XH_RS(V, G)                                                                                                    \
  hipLaunchKernelGGL((k_run_stats<V, true, G>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, fused_op, (float)thr, \
                     window, stat, index_first, d_seg, P, out, valid_out)
    if (vec == 4) {
      if (sg == 1) XH_RS(4, 1); else XH_RS(4, 0);
    } else {
      if (sg == 1) XH_RS(1, 1); else XH_RS(1, 0);
    }
#undef XH_RS
  } else {
    XH_REQUIRE(seg_off[0] == 0 && seg_off[P] == T, XH_ERR_ARG,
               "xh_run_stats: resample-after mode needs segments covering [0, T)");
    // one cell per lane: four per lane are slower with the early flush as well (0.89 vs 0.50 ms; round 1, lazy flush only:
    // 1.11 vs 0.57) — one serial march per cell and a quarter of the waves; XH_RUNSTATS_NOCUT_VEC=4 (diagnostics) selects it
    const char* ev = xh_diag_env("XH_RUNSTATS_NOCUT_VEC");
    const int nvec = (ev && atoi(ev) == 4 && vec == 4) ? 4 : 1;
    dim3 grid((unsigned)cdiv64(cdiv64(C, nvec), XH_BLOCK), 1);
    // stat group: 1 max, 2 sum / count / mean, 3 min, 0 all (std)
    const int sg = stat == XH_RUN_MAX ? 1 : (stat == XH_RUN_SUM || stat == XH_RUN_COUNT || stat == XH_RUN_MEAN || stat == XH_RUN_PLAINSUM) ? 2
                   : stat == XH_RUN_MIN ? 3 : 0;
#define XH_RSN(V, G)                                                                                                          \
  hipLaunchKernelGGL((k_run_stats<V, false, G>), grid, dim3(XH_BLOCK), 0, ctx->stream, x, T, C, st, fused_op, (float)thr, \
                     window, stat, index_first, d_seg, P, out, valid_out)
    if (nvec == 4) {
      if (sg == 1) XH_RSN(4, 1); else if (sg == 2) XH_RSN(4, 2); else if (sg == 3) XH_RSN(4, 3); else XH_RSN(4, 0);
    } else {
      if (sg == 1) XH_RSN(1, 1); else if (sg == 2) XH_RSN(1, 2); else if (sg == 3) XH_RSN(1, 3); else XH_RSN(1, 0);
    }
#undef XH_RSN
  }
  XH_LAUNCH_CHECK();
  return XH_OK;
}

int xh_run_stats_doy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op, const double* table,
                     int D, const int32_t* tidx, int window, int stat, const int64_t* seg_off, int P, float* out,
                     int32_t* valid_out) {
  XH_REQUIRE(ctx && x && table && out, XH_ERR_ARG, "xh_run_stats_doy: NULL argument");
  XH_REQUIRE(T >= 0 && C >= 0 && D >= 1, XH_ERR_ARG, "xh_run_stats_doy: bad shape");
  XH_REQUIRE(sc == 1 && st >= C, XH_ERR_LAYOUT, "xh_run_stats_doy: needs a time-major view (cell stride 1)");
  XH_REQUIRE(op >= XH_OP_GT && op <= XH_OP_NE, XH_ERR_OP, "xh_run_stats_doy: operator %d not recognized", op);
  XH_REQUIRE(window >= 1, XH_ERR_ARG, "xh_run_stats_doy: window must be >= 1");
  XH_REQUIRE((stat >= XH_RUN_MAX && stat <= XH_RUN_STD) || stat == XH_RUN_PLAINSUM, XH_ERR_OP,
             "xh_run_stats_doy: statistic %d not supported (run-length reducers only)", stat);
  XH_REQUIRE(seg_off && P >= 1 && tidx, XH_ERR_ARG, "xh_run_stats_doy: seg_off / tidx NULL or P < 1");
  for (int p = 0; p < P; ++p)
    XH_REQUIRE(seg_off[p] <= seg_off[p + 1] && seg_off[p] >= 0 && seg_off[p + 1] <= T, XH_ERR_ARG,
               "xh_run_stats_doy: seg_off must be non-decreasing within [0, T]");
  for (int64_t t = 0; t < T; ++t)
    XH_REQUIRE(tidx[t] >= 0 && tidx[t] < D, XH_ERR_ARG, "xh_run_stats_doy: tidx[%lld] = %d outside the table (D = %d)",
               (long long)t, tidx[t], D);
  size_t cur = 0;
  void *d_tidx = nullptr, *d_seg = nullptr;
  int rc = xh_scratch_upload(ctx, &cur, tidx, sizeof(int32_t) * (size_t)T, &d_tidx);
  if (rc) return rc;
  rc = xh_scratch_upload(ctx, &cur, seg_off, sizeof(int64_t) * (size_t)(P + 1), &d_seg);
  if (rc) return rc;
  if (C == 0) return XH_OK;
  dim3 grid((unsigned)cdiv64(C, XH_BLOCK), (unsigned)(P > 4096 ? 4096 : P));
  hipLaunchKernelGGL(k_run_stats_doy, grid, dim3(XH_BLOCK), 0, ctx->stream, x, C, st, op, table, (const int32_t*)d_tidx, window,
                     stat, (const int64_t*)d_seg, P, out, valid_out);
  XH_LAUNCH_CHECK();
  return XH_OK;
}

}  // extern "C"
