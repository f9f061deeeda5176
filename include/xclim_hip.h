/*
 * xclim_hip.h — C ABI of libxclimhip.so, the MI355X (gfx950) backend for xclim's
 * index / run-length / percentile / quantile-mapping hot path.
 *
 * The reference (Ouranosinc/xclim, /root/reference) has NO FFI: its "operator
 * boundary" is a set of Python module attributes plus the numpy callees handed to
 * xr.apply_ufunc.  Every entry point below names the reference
 * function(s) (file:line under /root/reference/src/xclim) it replaces.
 *
 * Conventions
 *   - every function returns int: 0 = XH_OK, <0 = error (message: xh_last_error()).
 *   - `x`, `out`, ... are DEVICE pointers (xh_malloc / any hipMalloc'd memory);
 *     host<->device staging is explicit (xh_memcpy_h2d / xh_memcpy_d2h).
 *   - arrays are 2-D views (T, C): T time steps, C grid cells (lat*lon flattened).
 *     `st`, `sc` are ELEMENT strides of the time and cell axes.  Streaming kernels
 *     need sc == 1 ("time-major", xarray's native (time, lat, lon) C order);
 *     column kernels (full-series quantiles) take st == 1 ("time-minor", what
 *     apply_ufunc hands its callee) or transpose internally.
 *   - periods of `resample(time=freq)` are contiguous time segments described by
 *     seg_off[P+1] (host computes them from the calendar, xclim_amd/timeaxis.py).
 *   - work is enqueued on the context's HIP stream; xh_sync / xh_memcpy_d2h wait.
 *   - a context is not re-entrant (one per device per caller thread).
 */
#ifndef XCLIM_HIP_H
#define XCLIM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XH_ABI_VERSION 1

/* error codes */
#define XH_OK 0
#define XH_ERR_HIP (-1)      /* a HIP runtime call failed */
#define XH_ERR_ARG (-2)      /* invalid argument value */
#define XH_ERR_LAYOUT (-3)   /* unsupported stride / alignment combination */
#define XH_ERR_OP (-4)       /* operator / reducer not recognized (gen:285) */
#define XH_ERR_NOTIMPL (-5)
#define XH_ERR_NODEVICE (-6) /* no gfx950 device visible */
#define XH_ERR_LIMIT (-7)    /* problem exceeds an on-chip capacity limit */

/* comparison operators — xclim.indices.generic.binary_ops (indices/generic.py:40-75, 255-298) */
#define XH_OP_GT 0
#define XH_OP_LT 1
#define XH_OP_GE 2
#define XH_OP_LE 3
#define XH_OP_EQ 4
#define XH_OP_NE 5

/* threshold kinds for xh_threshold_count */
#define XH_THR_SCALAR_F32 0 /* python-float threshold: compare in fp32 (NumPy weak scalar) */
#define XH_THR_SCALAR_F64 1 /* np.float64 threshold: data widened to fp64 */
#define XH_THR_DOY_F64 2    /* per-day-of-year table (D, C) fp64 + tidx[T] (resample_doy, core/calendar.py:763-790) */
#define XH_THR_DOY_F32 3
#define XH_THR_FULL_F64 4   /* full (T, C) threshold array */
#define XH_THR_FULL_F32 5

/* segmented reducers — select_resample_op (indices/generic.py:83-125) */
#define XH_RED_SUM 0
#define XH_RED_MEAN 1
#define XH_RED_MIN 2
#define XH_RED_MAX 3
#define XH_RED_STD 4
#define XH_RED_VAR 5
#define XH_RED_COUNT 6
#define XH_RED_ARGMIN 7
#define XH_RED_ARGMAX 8

/* run statistics — rle_statistics reducers and friends (indices/run_length.py:275-540) */
#define XH_RUN_MAX 0    /* rle_statistics(reducer="max") / longest_run */
#define XH_RUN_MIN 1
#define XH_RUN_SUM 2    /* rle_statistics("sum") == windowed_run_count (window>1 or freq) */
#define XH_RUN_COUNT 3  /* rle_statistics("count") == windowed_run_events */
#define XH_RUN_MEAN 4
#define XH_RUN_STD 5
#define XH_RUN_FIRST 6  /* first_run: index of first element of first run >= window, NaN if none */
#define XH_RUN_LAST 7   /* last_run:  index of last element of last run >= window, NaN if none  */
#define XH_RUN_PLAINSUM 8 /* windowed_run_count(window==1, freq=None): da.sum(dim) (rl:478-479) */

typedef struct xh_ctx xh_ctx;

/* ---- context, memory, timing ------------------------------------------------------------ */
int xh_abi_version(void);
const char* xh_last_error(void);
int xh_device_count(int* n);
int xh_create(int device, xh_ctx** out);
int xh_destroy(xh_ctx* ctx);
int xh_sync(xh_ctx* ctx);
int xh_device_name(xh_ctx* ctx, char* buf, size_t buflen);
int xh_mem_info(xh_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
int xh_malloc(xh_ctx* ctx, size_t bytes, void** dptr);
int xh_free(xh_ctx* ctx, void* dptr);
int xh_memset(xh_ctx* ctx, void* dptr, int value, size_t bytes);
int xh_memcpy_h2d(xh_ctx* ctx, void* dst, const void* src, size_t bytes);
int xh_memcpy_d2h(xh_ctx* ctx, void* dst, const void* src, size_t bytes); /* synchronises */
int xh_memcpy_d2d(xh_ctx* ctx, void* dst, const void* src, size_t bytes);
/* HIP-event timer on the context's stream (bench.py roofline: kernel time without host overhead) */
int xh_timer_start(xh_ctx* ctx);
int xh_timer_stop(xh_ctx* ctx, float* elapsed_ms); /* synchronises */
/* raw hipStream_t of the context, for interop (e.g. ordering against RCCL collectives) */
int xh_stream(xh_ctx* ctx, void** stream);

/* ---- multi-GPU exchange ------------------------------------------------------------------------------
 * The path shards over (lat, lon) with no halo; its ONE exchange step is the all-gather of the reduced outputs ((P, C/N)
 * counts / statistics, (nq, C/N) quantile nodes; scen stays sharded).  RCCL over xGMI, one process per GPU; librccl.so is
 * dlopen'ed at the first xh_comm_* call (single-GPU users never load it).  The reference has no collectives:
 * no reference line to match — this is north_star's "RCCL over xGMI only for the final gather".
 *   rendezvous  rank 0: xh_comm_unique_id -> hands the XH_COMM_ID_BYTES to every rank (file / socket / environment:
 *               xclim_amd/shard.py uses a node-local file); every rank: xh_comm_init (collective, blocks).
 *   all-gather  recv holds nranks * bytes_per_rank bytes, rank r's block at r * bytes_per_rank.  slot < 0: enqueued on
 *               the context's stream, in line with the kernels.  slot in [0, 4): runs on the communicator's own stream
 *               after everything queued on the context's stream so far, overlapping later kernels; xh_comm_fence(slot)
 *               makes the context's stream wait for that collective (before its send buffer is overwritten),
 *               xh_comm_sync waits on the host for both streams.
 *   scalars     xh_comm_allreduce_f64: 1 or 2 host doubles reduced over the ranks in place (timings, checksums);
 *               xh_comm_barrier = drain this rank's streams + a one-word all-reduce. */
typedef struct xh_comm xh_comm;
#define XH_COMM_ID_BYTES 128
#define XH_COMM_SUM 0
#define XH_COMM_MAX 2
#define XH_COMM_MIN 3
int xh_comm_unique_id(void* id /* host, XH_COMM_ID_BYTES */);
int xh_comm_init(xh_ctx* ctx, int nranks, int rank, const void* id /* host */, xh_comm** out);
int xh_comm_destroy(xh_comm* comm);
int xh_comm_size(xh_comm* comm, int* nranks, int* rank);
int xh_comm_allgather(xh_comm* comm, const void* send, void* recv, size_t bytes_per_rank, int slot);
int xh_comm_fence(xh_comm* comm, int slot);
int xh_comm_sync(xh_comm* comm);
int xh_comm_allreduce_f64(xh_comm* comm, double* host_values, int count, int op);
int xh_comm_barrier(xh_comm* comm);

/* ---- block adapter support (chunked host inputs streamed through the device; the reference's
 * dask / map_blocks layer, indices/helpers.py:898-974, core/indicator.py:865-944) --------------------------------
 * Pinned host memory (xh_host_alloc, or xh_host_register on caller memory) makes the strided copies below asynchronous
 * and full PCIe rate.  Lanes: 0 = the compute stream every xh_* op runs on, 1 = copy-in stream, 2 = copy-out stream.
 * xh_memcpy2d copies `height` rows of `width` bytes (pitches in bytes; a cell slab of a (T, C) host array is
 * width = slab * 4, spitch = C * 4); kind 0 = host -> device, 1 = device -> host, 2 = device -> device; blocking != 0 waits for the copy
 * (required for pageable host memory).  xh_lane_fence(a, b): work queued on lane b from now on waits for everything
 * queued on lane a so far.  xh_lane_sync: the host waits for the lane. */
int xh_host_alloc(xh_ctx* ctx, size_t bytes, void** hptr);
int xh_host_free(xh_ctx* ctx, void* hptr);
int xh_host_register(xh_ctx* ctx, void* hptr, size_t bytes);
int xh_host_unregister(xh_ctx* ctx, void* hptr);
int xh_memcpy2d(xh_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, int kind,
                int lane, int blocking);
int xh_lane_fence(xh_ctx* ctx, int from_lane, int to_lane);
int xh_lane_sync(xh_ctx* ctx, int lane);

/* ---- synthetic inputs: counter-based generator, restated in oracle/synth.py -- */
/* out[t, c] = base[t] + amp * z(seed, t, cell0 + c),  z = sum of 4 hashed uniforms - 2 (var 1/3);
 * kind 0: temperature-like (as above); kind 1: precipitation-like (wet w.p. p_wet, amount = amp*u^3,
 * else 0); nan_per_million: independent NaN probability.  Layout (T, C) with row stride st. */
int xh_fill_synthetic(xh_ctx* ctx, float* out, int64_t T, int64_t C, int64_t st, int kind, uint64_t seed,
                      int64_t cell0, const float* base /* device, T */, float amp, float p_wet,
                      uint32_t nan_per_million);

/* tiled transpose (T, C) -> (C, T) (row strides in elements); used by column kernels */
int xh_transpose_f32(xh_ctx* ctx, const float* in, int64_t rows, int64_t cols, int64_t in_stride, float* out,
                     int64_t out_stride);

/* ---- threshold / count family (G1-G3, M1) -------------------------------------------------- */
/* threshold_count (indices/generic.py:329-361) = (compare(da, op, thr) * 1).resample(time=freq).sum("time")
 * fused with MissingAny's valid count (core/missing.py:201-220, 318-322).
 *   thr_kind scalar: `thr_scalar`; DOY: thr_table (D, C) row stride thr_stride, tidx[T] row index per step;
 *   FULL: thr_table (T, C) row stride thr_stride.
 *   count_out (P, C) int32; valid_out (P, C) int32 = #non-NaN of x per period (may be NULL). */
int xh_threshold_count(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op,
                       int thr_kind, double thr_scalar, const void* thr_table, int64_t thr_stride,
                       const int32_t* tidx, const int64_t* seg_off, int P, int32_t* count_out,
                       int32_t* valid_out);

/* The DOY_F64 form with the number of table rows stated (ndoy = D): same results as xh_threshold_count(thr_kind =
 * XH_THR_DOY_F64).  On multi-year series (tx90p over a 30-year period: threshold_count(tasmax, ">", resample_doy(per)),
 * indices/_threshold.py + core/calendar.py resample_doy) one workgroup keeps its columns' table slice on chip instead of
 * re-reading the (D, C) table once per year; tidx[t] outside [0, D) compares as a NaN threshold. */
int xh_threshold_count_doy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op,
                           const double* thr_table, int64_t thr_stride, int ndoy, const int32_t* tidx,
                           const int64_t* seg_off, int P, int32_t* count_out, int32_t* valid_out);

/* domain_count / count_occurrences-style two-sided scalar conditions (indices/generic.py:364-392):
 *   cond = (x op1 thr1) AND|OR (x op2 thr2);  combine: 1 = and, 2 = or.  fp32 compares. */
int xh_domain_count(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op1,
                    double thr1, int op2, double thr2, int combine, const int64_t* seg_off, int P,
                    int32_t* count_out, int32_t* valid_out);

/* Two-variable counts: count_level_crossings (indices/generic.py:913-957) and bivariate_count_occurrences
 * (generic.py:1002-1073): cond = (x1 op1 thr1) AND(1)|OR(2) (x2 op2 thr2), summed per period.  valid_out counts
 * the days on which both variables are non-NaN. */
int xh_bivariate_count(xh_ctx* ctx, const float* x1, const float* x2, int64_t T, int64_t C, int64_t st1, int64_t st2,
                       int op1, double thr1, int op2, double thr2, int combine, const int64_t* seg_off, int P,
                       int32_t* count_out, int32_t* valid_out);

/* Two-variable range reductions per period, out (P, C) float32, all NaN-skipping (xarray default for floats):
 *   mode 0 diurnal_temperature_range          (generic.py:1076-1105): reducer (XH_RED_SUM|MEAN|MIN|MAX) of (high - low)
 *   mode 1 interday_diurnal_temperature_range (generic.py:1360-1385): mean of |d(high - low)/dt| (diff drops day 0)
 *   mode 2 extreme_temperature_range          (generic.py:1388-1414): max(high) - min(low)
 * valid_out counts the days on which both variables are non-NaN. */
int xh_range_reduce(xh_ctx* ctx, const float* low, const float* high, int64_t T, int64_t C, int64_t st_low,
                    int64_t st_high, int mode, int reducer, const int64_t* seg_off, int P, float* out,
                    int32_t* valid_out);

/* select_time (core/calendar.py:1259-1378), the `**indexer` of select_resample_op & co.: out row i = x row idx[i] (host
 * int64[n]), NaN when idx[i] < 0.  `da.where(mask)`: n = T, idx[t] = mask[t] ? t : -1; `drop=True`: the selected rows. */
int xh_select_rows(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const int64_t* idx, int64_t n,
                   float* out, int64_t st_out);

/* A bool / uint8 mask (n elements, non-zero = True) as the float32 1 / 0 mask the run-length entry points read
 * (`da` of indices/run_length.py is boolean): host masks cross PCIe as bytes. */
int xh_mask_u8_to_f32(xh_ctx* ctx, const uint8_t* mask, int64_t n, float* out);

/* compare (generic.py:301-326) / get_daily_events (generic.py:395-431) as an elementwise map of a (T, C) field against a
 * scalar (fp32 compare, or fp64 when thr_is_f64) or a second field b (NULL for the scalar form):
 *   out_kind 0: uint8 mask    1: float32 1/0 with NaN where a is NaN    2: float32 a.where(cond) (NaN elsewhere)
 *            3: float32 1/0 (the boolean mask as float, NaN compares False)
 *            4: float32 (a - thr).clip(0), NaN where a is NaN (`op` unused; hot_spell_max_magnitude, _threshold.py:2056-2057) */
int xh_compare_map(xh_ctx* ctx, const float* a, int64_t T, int64_t C, int64_t st, int op, double thr, int thr_is_f64,
                   const float* b, int64_t st_b, int out_kind, void* out, int64_t st_out);

/* Thresholded reductions per period, out (P, C) float32:
 *   mode 0 thresholded_statistics (generic.py:1278-1320): reducer (XH_RED_SUM|MEAN|MIN|MAX) of data.where(cond)
 *   mode 1 temperature_sum        (generic.py:1323-1357): direction * sum((data - thr).where(cond))
 *   mode 2 cumulative_difference  (generic.py:1514-1552): sum(clip(data - thr, 0)) or sum(clip(thr - data, 0)) */
int xh_thresholded_reduce(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op,
                          double thr, int mode, int reducer, const int64_t* seg_off, int P, float* out,
                          int32_t* valid_out);

/* da.where(lo[p] <= t - seg_off[p] < hi[p]).fillna(0) as a 0/1 mask (invert != 0: of NOT da): the masking step of
 * first_run_after_date / last_run_before_date / first_run_before_date / run_end_after_date (run_length.py:1148-1331). */
int xh_mask_rows(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const int64_t* seg_off, int P,
                 const int32_t* lo, const int32_t* hi, int invert, float* out, int64_t out_st);

/* climatological_mean_doy (core/calendar.py:907-931): per-doy nanmean / nanstd (ddof 0) over all years and the
 * centred window; same tbase table as xh_percentile_doy.  mean_out, std_out (ndoy, C) float32. */
int xh_doy_mean_std(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const int32_t* tbase,
                    int nyears, int ndoy, int window, float* mean_out, float* std_out);

/* select_resample_op (indices/generic.py:83-125): segmented reduction over periods.
 *   float reducers write float32 `out` (P, C) (fp64 accumulation); COUNT/ARGMIN/ARGMAX write int32.
 *   skipna != 0: NaN ignored (all-NaN -> NaN, SUM -> 0) as xarray's default for floats.
 *   valid_out (P, C) int32 may be NULL. */
int xh_resample_reduce(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int reducer,
                       int skipna, const int64_t* seg_off, int P, void* out, int32_t* valid_out);

/* MissingAny (core/missing.py:318-322): out64[p, c] = valid[p, c] != expected[p] ? NaN : value.
 * value_kind 0: int32 input, 1: float32 input. */
int xh_apply_missing_mask(xh_ctx* ctx, const void* value, int value_kind, const int32_t* valid,
                          const int32_t* expected, int P, int64_t C, double* out64);

/* rolling(time=window, center).{sum,mean,min,max,std,var,count}() (indices/generic.py:128-174);
 * NaN anywhere in the window -> NaN (min_periods == window), incomplete windows -> NaN. */
int xh_rolling_reduce(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window,
                      int center, int reducer, float* out, int64_t out_st);

/* ---- run-length family (R1-R5, S1 fast path) ------------------------------------------------ */
/* _cumsum_reset (indices/run_length.py:143-219): c[t] = b[t] * (c[t-1] + 1), NaN -> 0;
 * index_first != 0 runs the recurrence from the end. */
int xh_cumsum_reset(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int index_first,
                    float* out, int64_t out_st);
/* rle (indices/run_length.py:223-272): run length at the first (last) element of each run, NaN inside,
 * 0 outside. */
int xh_rle(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int index_first, float* out,
           int64_t out_st);

/* rle_statistics / longest_run / windowed_run_count / windowed_run_events / first_run / last_run
 * (indices/run_length.py:275-488, 543-740), optionally fused with the spell condition of
 * spell_length_statistics (indices/generic.py:499-502, 543-585):
 *   fused_op < 0 : x IS the mask (float32: >0 True, 0 False, NaN = NaN as select_time leaves it)
 *   fused_op >= 0: mask = compare(x, fused_op, thr) in fp32 (NaN -> False); valid_out counts non-NaN x.
 *   cut_at_segments != 0: resample BEFORE run length (runs cut at period edges, rl:122-129);
 *   else resample AFTER (run attributed to the period of its first/last element, rl:330-334).
 *   index_first: 0 = runs indexed by their last step, 1 = by their first step — both as the N-D path of the reference
 *   (rle, rl:223-272: a run whose outer neighbour is NaN loses its length) — 2 = first step with the semantics of the
 *   reference's 1-D ufunc path (rle_1d / statistics_run_1d, rl:1334-1618, taken for grids under 9000 cells when no
 *   resampling follows, rl:70-78): NaN steps only break runs, a run next to a NaN keeps its length; 3 = the same plus
 *   statistics_run_1d's result for a series WITH NaN steps and WITHOUT a qualifying run: NaN instead of 0 for MAX / MIN / MEAN / STD (numpy's nan-reducers of an empty selection; nansum and the count stay 0).
 *   out (P, C) float32 (integer valued; NaN for FIRST/LAST when no run); valid_out may be NULL. */
int xh_run_stats(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int fused_op,
                 double thr, int window, int stat, int index_first, const int64_t* seg_off, int P,
                 int cut_at_segments, float* out, int32_t* valid_out);

/* spell_mask (indices/generic.py:434-540): out[t] = 1 iff day t belongs to any window of `window` consecutive
 * days whose statistic (win_reducer 0 sum, 1 mean, 2 min, 3 max, 4 weighted mean with host `weights[window]`)
 * satisfies `stat op thr`; NaN in a window -> not satisfied.  out (T, C) float32 0/1. */
int xh_spell_mask(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window,
                  int win_reducer, int op, double thr, const float* weights /* host, may be NULL */, float* out,
                  int64_t out_st);

/* spell_length_statistics with a window > 1, resample_before_rl (generic.py:543-686) in one pass: the spell mask of
 * xh_spell_mask is not materialised, its run statistics (XH_RUN_MAX .. XH_RUN_STD, runs cut at the period edges) are
 * accumulated directly.  One variable, window <= 8 (XH_ERR_NOTIMPL beyond: use xh_spell_mask + xh_run_stats).
 * valid_out: non-NaN steps of x per period (may be NULL). */
int xh_spell_run_stats(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window, int win_reducer,
                       int op, double thr, const float* weights /* host, may be NULL */, int stat, const int64_t* seg_off,
                       int P, float* out, int32_t* valid_out);

/* spell_mask on a list of variables (generic.py:434-540, `data` a sequence): xs[nvar] device pointers to (T, C) fields of
 * the same layout, thrs[nvar] their thresholds (host arrays); the per-variable window conditions are combined with
 * all (combine = 1) or any (2) before the spell is propagated.  nvar <= 8. */
int xh_spell_mask_multi(xh_ctx* ctx, const float* const* xs, int nvar, const double* thrs, int combine, int64_t T,
                        int64_t C, int64_t st, int64_t sc, int window, int win_reducer, int op, const float* weights,
                        float* out, int64_t out_st);
/* runs_with_holes (indices/run_length.py:844-888): hysteresis mask — on after window_start consecutive `start`,
 * off after window_stop consecutive `stop` (stop == NULL means stop = NOT start); out (T, C) float32 0/1. */
int xh_runs_with_holes(xh_ctx* ctx, const float* start, const float* stop, int64_t T, int64_t C, int64_t st,
                       int64_t sc, int window_start, int window_stop, float* out, int64_t out_st);
/* keep_longest_run (indices/run_length.py:805-841): keep only the first longest run of each period. */
int xh_keep_longest_run(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc,
                        const int64_t* seg_off, int P, float* out, int64_t out_st);
/* season (indices/run_length.py:891-1145) per period: start / end indices relative to the period start (NaN when
 * undefined) and length.  mid_idx[P]: index of `mid_date` inside each period, < 0 when absent; NULL = no date. */
int xh_season(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window,
              const int64_t* seg_off, const int32_t* mid_idx, int P, float* start_out, float* end_out,
              float* len_out);
/* windowed_max_run_sum (indices/run_length.py:491-540): max over the runs (x > 0) of at least `window` steps of the
 * run's sum; out (P, C) float32.  cut_at_segments != 0: runs cut at the period edges (rl.resample_and_rl with
 * resample_before_rl, the default of hot_spell_max_magnitude); 0: the reference's own `freq` semantics — cumsum and run
 * lengths over the whole series, a run's sum attributed to the period of its first step (segments must cover [0, T)). */
int xh_max_run_sum(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window,
                   const int64_t* seg_off, int P, int cut_at_segments, float* out);

/* Event compaction for run_bounds (run_length.py:745-802) and find_events / _find_events (run_length.py:1760-1901).
 * `runs` is a 0/1 float field (a mask, or xh_runs_with_holes output).  The k-th run (time order) of (period p, cell c)
 * writes element [(p * maxev + k) * C + c] of every non-NULL output, rows past the last run are NaN:
 *   start_out  first step of the run, relative to the period start      end_out  first step after the run (NaN if none)
 *   len_out    run length       eff_out  steps of the run where `eff` != 0 (NULL: = length)
 *   sum_out    sum of `data` from the run start to the first NaN of data inside the run (the reference's
 *              _cumsum_reset_xr(..., reset_on_zero=False) arithmetic in fp32) */
int xh_run_events(xh_ctx* ctx, const float* runs, const float* eff, const float* data, int64_t T, int64_t C, int64_t st,
                  int64_t sc, const int64_t* seg_off, int P, int maxev, float* start_out, float* end_out, float* len_out,
                  float* eff_out, float* sum_out);

/* suspicious_run / suspicious_run_1d (run_length.py:1668-1757): out (T, C) uint8 = 1 on steps that belong to a run of at
 * least `window` identical values whose value satisfies `op thresh` (op = -1: no threshold). */
int xh_suspicious_run(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window, int op,
                      double thresh, uint8_t* out, int64_t out_st);

/* ---- quantile / percentile family (Q1-Q3) --------------------------------------------------- */
/* calc_perc / _nan_quantile (core/utils.py:279-557): NaN-aware Hyndman-Fan quantiles of N samples per
 * cell.  x is (N, C) with sample stride sn and cell stride sc (either may be 1).  q[nq] in [0, 1].
 * out (nq, C) float64 (row stride C). */
int xh_nan_quantile(xh_ctx* ctx, const float* x, int64_t N, int64_t C, int64_t sn, int64_t sc, const double* q,
                    int nq, double alpha, double beta, double* out);

/* ---- float64 FIELDS ------------------------------------------------------------------------------------------
 * The reference computes in the dtype of its input: compare() of a float64 DataArray is a float64 compare
 * (indices/generic.py:301-326, 360), resample(...).<op>() returns float64 (gen:83-125), _nan_quantile takes `diff` in
 * float64 (core/utils.py:486).  These three entries take float64 fields as they are (no rounding to float32); same
 * layouts, segment tables and outputs as their float32 twins, except:
 *   xh_threshold_count_f64: thr_kind in {XH_THR_SCALAR_F64, XH_THR_DOY_F64, XH_THR_FULL_F64} (float64 tables);
 *   xh_resample_reduce_f64: float reducers write FLOAT64 `out` (P, C);
 *   xh_nan_quantile_f64:    N <= 4096 samples per cell, out (nq, C) float64. */
int xh_threshold_count_f64(xh_ctx* ctx, const double* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op, int thr_kind,
                           double thr_scalar, const double* thr_table, int64_t thr_stride, const int32_t* tidx,
                           const int64_t* seg_off, int P, int32_t* count_out, int32_t* valid_out);
int xh_resample_reduce_f64(xh_ctx* ctx, const double* x, int64_t T, int64_t C, int64_t st, int64_t sc, int reducer, int skipna,
                           const int64_t* seg_off, int P, void* out, int32_t* valid_out);
int xh_nan_quantile_f64(xh_ctx* ctx, const double* x, int64_t N, int64_t C, int64_t sn, int64_t sc, const double* q, int nq,
                        double alpha, double beta, double* out);

/* Weighted quantiles over the first axis (ensemble_percentiles with `weights`, ensembles/_base.py:346-356, which calls
 * xarray's DataArrayWeighted.quantile: Kish effective sample size + type-7 weighted estimator, NaN samples and zero
 * weights dropped).  x (N, C) member-major (sc == 1), weights[N] / q[nq] on the host, out (nq, C) float64; N <= 128.
 * PARITY UNPINNED (the arithmetic is xarray's, restated from its published form; equal weights reproduce
 * xh_nan_quantile with alpha = beta = 1). */
int xh_weighted_quantile(xh_ctx* ctx, const float* x, int64_t N, int64_t C, int64_t sn, int64_t sc,
                         const double* weights /* host */, const double* q /* host */, int nq, double* out);

/* percentile_doy (core/calendar.py:395-494) before the 366-day adjustment: for each day-of-year row d
 * and year y, tbase[y * ndoy + d] is the time index of that calendar day (or -1 if the year lacks it);
 * the sample set is x[tbase - window/2 .. tbase + window - 1 - window/2] over all years (NaN outside
 * [0, T)), reduced with _nan_quantile.  out (nper, ndoy, C) float64. */
int xh_percentile_doy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc,
                      const int32_t* tbase, int nyears, int ndoy, int window, const double* per /* host */,
                      int nper, double alpha, double beta, double* out);

/* percentile_doy on a VIRTUAL time axis: tbase indexes virtual days 0..Tv-1 and vmap[Tv] (host) maps each virtual
 * day to a physical row of x (-1 = absent -> NaN).  Lets percentile_bootstrap (core/bootstrapping.py:81-282) swap a
 * year of the base period for another one by changing the index table instead of deep-copying the data. */
int xh_percentile_doy_mapped(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc,
                             const int32_t* tbase, int nyears, int ndoy, int window, const double* per /* host */,
                             int nper, double alpha, double beta, const int32_t* vmap /* host */, int64_t Tv,
                             double* out);

/* Fused percentile_doy + threshold_count (the tx90p family, indices/_multivariate.py:1534-1650, when the percentile
 * base period IS the analysed series): count_out[p, c] = #{t in period p : x[t, c] op percentile_doy(x)[doy(t), c]}
 * (the fp64 compare of the reference), valid_out[p, c] = non-NaN days (may be NULL), doy_period[nyears * ndoy] = period
 * of every (year, doy) day (< 0 where tbase is -1).  The (D, C) float64 table is never materialised.  Covered: one
 * contiguous year (sliding-window kernel), and multi-year base periods whose doys are all regular (no calendar gaps) with
 * a percentile that selects within the 16 largest / smallest samples (register top-16 kernel; ops > >= < <=).  Returns
 * XH_ERR_NOTIMPL otherwise (windows other than 3 / 5 / 7, gaps, central percentiles): use the two-step chain there. */
int xh_percentile_doy_count(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const int32_t* tbase,
                            int nyears, int ndoy, int window, double per, double alpha, double beta, int op,
                            const int32_t* doy_period, int P, int32_t* count_out, int32_t* valid_out);

/* _interpolate_doy_calendar (core/calendar.py:690-726): interpolate_na along doy then linear re-grid
 * D_in -> D_out with host-computed tables (scipy interp1d form): slope = (in[i1[j]] - in[i0[j]]) / dxs[j];
 * out[j] = slope * dxn[j] + in[i0[j]],  dxn = x_new - x_lo, dxs = x_hi - x_lo.  in (D_in, C), out (D_out, C).
 * xsrc[D_in] (host, may be NULL = row index): the dayofyear coordinate of the source rows, used by the interpolate_na
 * step (xarray fills NaN gaps linearly IN THE COORDINATE, which is not uniform when a doy never occurs in the series). */
int xh_doy_interp(xh_ctx* ctx, const double* in, int D_in, int64_t C, const int32_t* i0, const int32_t* i1,
                  const double* dxn, const double* dxs, int D_out, double* out, const double* xsrc);

/* resample_doy (core/calendar.py:763-790): out (T, C) float64 = table[tidx[t]] for a (D, C) per-doy table; tidx is the
 * host array of table rows per time step.  (xh_threshold_count fuses this gather; this entry materialises the field.) */
int xh_doy_broadcast(xh_ctx* ctx, const double* table, int D, int64_t C, const int32_t* tidx, int64_t T, double* out);

/* within_bnds_doy (core/calendar.py:934-954): out (T, C) uint8 = (low[tidx[t]] < x[t]) && (x[t] < high[tidx[t]]),
 * fp64 compares, low / high (D, C) float64 per-doy tables. */
int xh_within_bnds_doy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const double* low,
                       const double* high, int D, const int32_t* tidx, uint8_t* out);

/* select_time(da, doy_bounds=(start, end)) with PER-CELL bounds (mask_between_doys, core/calendar.py:1166-1257, bounds
 * without a time dimension): out (T, C) = x where doy[t] lies in the cell's [start[c], end[c]] (wrapping over the new
 * year when start > end), NaN elsewhere.  doy[T] on the host; start / end (C,) float32 on the device, already shifted
 * by +-1 for exclusive bounds, NaN = open (1 / 366). */
int xh_mask_doy_cells(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const int32_t* doy /* host */,
                      const float* start, const float* end, float* out, int64_t out_st);

/* ... with bounds that carry a TIME dimension (cal:1211-1246: one (start, end) pair per period of the bounds' own
 * frequency and per cell): lo / hi (P, C) float32 on the device = the bounds as days since the first step of period p
 * (doy_to_days_since on the host, NaN -> 0 / 366, lo = +inf for a period the bounds do not cover); seg_off[P + 1] on the
 * host, covering [0, T).  out (T, C) = x where lo[p, c] <= t - seg_off[p] <= hi[p, c], NaN elsewhere. */
int xh_mask_days_cells(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const int64_t* seg_off /* host */,
                       int P, const float* lo, const float* hi, float* out, int64_t out_st);

/* The weighted "spell value" of spell_mask (indices/generic.py:523-524: rolling(time=window).construct("window").dot(
 * weights)) as a field: out[t] = sum_k weights[k] * x[t - window + 1 + k] (float64 sum, one rounding to float32), NaN while
 * the window is incomplete or holds a NaN.  weights[window] on the host.  Used when the threshold differs per cell. */
int xh_rolling_dot(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window, const double* weights /* host */,
                   float* out, int64_t out_st);

/* compare(da, op, resample_doy(per, da)) as a float32 1/0 mask (fp64 compare against the (D, C) per-doy table): the
 * first step of warm_spell_duration_index / cold_spell_duration_index (indices/_multivariate.py:66-152, 1693-1793);
 * xh_run_stats on the mask gives the index. */
int xh_compare_doy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op, const double* table,
                   int D, const int32_t* tidx, float* out, int64_t st_out);

/* warm / cold_spell_duration_index with resample_before_rl (indices/_multivariate.py:66-152, 1693-1793) in one pass:
 * xh_run_stats(cut_at_segments = 1) on the condition x[t] op table[tidx[t]] (fp64 compare, (D, C) float64 per-doy
 * table), without materialising the mask.  stat: XH_RUN_MAX..XH_RUN_STD or XH_RUN_PLAINSUM. */
int xh_run_stats_doy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op, const double* table,
                     int D, const int32_t* tidx, int window, int stat, const int64_t* seg_off, int P, float* out,
                     int32_t* valid_out);

/* days_over_precip_thresh / fraction_over_precip_thresh (indices/_multivariate.py:1174-1232, 1236-1296) in one pass.
 * tp = max(table[tidx[t]], thr) in float64 (a NaN percentile gives thr, `pr_per.where(pr_per > thresh, thresh)`); the table
 * is the (D, C) float64 per-doy percentile, or D = 1 with tidx all 0 for a per-cell percentile.  op is > or >=.
 *   n_over (P, C) int32 : days with x op tp (float64 compare)                      [may be NULL]
 *   frac   (P, C) f32   : sum(x where x op tp) / sum(x where x op (float)thr)      [may be NULL]; 0/0 = NaN
 *   valid_out           : per-period count of non-NaN x                            [may be NULL] */
int xh_precip_over_doy(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int op, double thr,
                       const double* table, int D, const int32_t* tidx, const int64_t* seg_off, int P, float* frac,
                       int32_t* n_over, int32_t* valid_out);

/* ---- sdba empirical quantile mapping (E1-E4; xsdba >= 0.4.0, not in the reference tree) ------ */
/* nbutils.quantile: per-cell NaN-aware type-7 quantiles of the whole series at nq nodes. out (nq, C) f32 */
int xh_quantile_series(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc,
                       const double* q /* host */, int nq, float* out);
/* eqm_train: ref_q, hist_q = quantile(ref), quantile(hist); af = ref_q - hist_q (kind 0 "+") or
 * ref_q / hist_q (kind 1 "*").  af, hist_q (nq, C) float32. */
int xh_eqm_train(xh_ctx* ctx, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st, int64_t sc,
                 const double* q /* host */, int nq, int kind, float* af, float* hist_q);
/* eqm_train over a SLIDING row sample: the G groups of Grouper("time.dayofyear", window=W) on a series where every year holds
 * every day (xsdba >= 0.4 base.Grouper + _adjustment.eqm_train per windowed block; documented standard configuration:
 * /root/reference/docs/sdba.rst:64-65, re-exported by /root/reference/src/xclim/sdba.py:10).  rows0 [n0 <= 1024] (host): the
 * time steps of the first group's sample, -1 = beyond the series; leave / enter [G - 1][per <= 64] (host): the time steps that
 * leave / enter the sample from group g to g + 1 (-1 = none).  Every cell keeps its window sorted and updates it per step
 * (winsel.hip) — results bit-identical to xh_eqm_train on each group's gathered sample.  af, hist_q (G, nq <= 32, C) float32.
 * XH_ERR_NOTIMPL (no error text) for other shapes: gather each group's sample and call xh_eqm_train. */
int xh_eqm_train_window(xh_ctx* ctx, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st,
                        const int32_t* rows0 /* host */, int n0, const int32_t* enter /* host */,
                        const int32_t* leave /* host */, int G, int per, const double* q /* host */, int nq, int kind,
                        float* af, float* hist_q);
/* dqm_train over the same sliding row sample (xsdba._adjustment.dqm_train per day-of-year group with a window; not in the
 * reference tree — /root/reference/src/xclim/sdba.py:10 re-exports xsdba): ref and hist are normalised by the mean of the
 * group's sample (x - mean for kind 0, x / mean for kind 1: fp64 operation, fp32 result as xh_trend_apply), af / hist_q are
 * the corrections / quantiles of the normalised samples.  scaling (G, C) float64 = mean(ref) - mean(hist) resp. mean(ref) /
 * mean(hist); mu_hist (G, C) float64 = the means of hist.  The means are fp64 sums over the sorted window: equal to
 * xh_poly_trend(degree 0) on the gathered sample up to the summation order.  XH_ERR_NOTIMPL as xh_eqm_train_window. */
int xh_dqm_train_window(xh_ctx* ctx, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st,
                        const int32_t* rows0 /* host */, int n0, const int32_t* enter /* host */,
                        const int32_t* leave /* host */, int G, int per, const double* q /* host */, int nq, int kind,
                        float* af, float* hist_q, double* scaling, double* mu_hist);
/* eqm_train / dqm_train for ALL groups of a sub-grouping with small groups (a day-of-year grouping WITHOUT a window: one row per
 * year) in one launch per field.  rows (host, offs[G] entries): the row numbers of group 0, then of group 1, ... (the order a
 * group's mean is summed in); offs (host, G + 1).  af, hist_q (G, nq, C) float32; scaling, mu_hist (G, C) float64 as
 * xh_dqm_train_window.  Bit-identical to xh_eqm_train (resp. xh_poly_trend degree 0 + xh_trend_apply + xh_eqm_train) on each
 * group's gathered rows.  XH_ERR_NOTIMPL (no error text): a group of more than 64 rows. */
int xh_eqm_train_groups(xh_ctx* ctx, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st,
                        const int32_t* rows /* host */, const int64_t* offs /* host */, int G, const double* q /* host */, int nq,
                        int kind, float* af, float* hist_q);
int xh_dqm_train_groups(xh_ctx* ctx, const float* ref, const float* hist, int64_t T, int64_t C, int64_t st,
                        const int32_t* rows /* host */, const int64_t* offs /* host */, int G, const double* q /* host */, int nq,
                        int kind, float* af, float* hist_q, double* scaling, double* mu_hist);
/* qm_adjust: af_t = interp_on_quantiles(sim, hist_q, af) (interp 0 nearest, 1 linear, 2 cubic [not-a-knot spline as
 * scipy interp1d(kind="cubic"), nq <= 32, >= 4 valid nodes per cell else NaN]; extrap 0 constant,
 * 1 nan); scen = sim + af_t (kind 0) or sim * af_t (kind 1); kind 2: scen = af_t, the interpolated factor itself
 * (what xsdba.utils.interp_on_quantiles returns for group="time").  scen (T, C) row stride scen_st. */
int xh_eqm_adjust(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, int64_t sc, const float* af,
                  const float* hist_q, int nq, int kind, int interp, int extrap, float* scen, int64_t scen_st);
/* The same for the steps of ONE group of a month / day-of-year grouping with interp "nearest" the way
 * xsdba.utils.interp_on_quantiles does it with a sub-grouping (upstream xsdba, re-exported by
 * /root/reference/src/xclim/sdba.py:10): _interp_on_quantiles_2D = scipy griddata(method="nearest") over the nodes of ALL
 * groups in the (hist_q, group coordinate) plane (cyclic copies of the last / first group at coordinates 0 / G + 1), then
 * _extrapolate_on_quantiles with the step's own group (constant | nan).  sim / scen (n, C): the group's rows; af_all /
 * hq_all (G, nq, C); gcoord in 1 .. G; kind 0 (+) | 1 (*) | 2 (factor only).  Parity unpinned. */
int xh_eqm_adjust_g2d(xh_ctx* ctx, const float* sim, int64_t n, int64_t C, int64_t st, const float* af_all, const float* hq_all,
                      int G, int nq, int gcoord, int kind, int extrap, float* scen, int64_t scen_st);
/* xsdba.utils.apply_correction(base, fac, kind) on two fields of one shape (upstream xsdba, re-exported by
 * /root/reference/src/xclim/sdba.py:10): out = base + fac (kind 0) | base * fac (kind 1); (T, C) views with row strides st,
 * fst, out_st.  Used where the factor was interpolated on another abscissa than the field itself (QDM "cubic"). */
int xh_apply_factor(xh_ctx* ctx, const float* base, const float* fac, int64_t T, int64_t C, int64_t st, int64_t fst, int kind,
                    float* out, int64_t out_st);
/* interp = "linear" with a month / day-of-year Grouper: xsdba.utils.interp_on_quantiles' 2-D branch (upstream xsdba,
 * re-exported by /root/reference/src/xclim/sdba.py:10; the documented standard use, /root/reference/docs/sdba.rst:64-65,
 * /root/reference/CHANGELOG.rst:338): scipy griddata(method="linear") = barycentric interpolation on the Delaunay
 * triangulation of the nodes of ALL groups in the (abscissa, group coordinate) plane (cyclic copies of the last / first
 * group at coordinates 0 / G + 1), then _extrapolate_on_quantiles "constant" (first / last factor, np.interp'ed between the
 * two neighbouring groups for a fractional coordinate).  xnew (T, C) row stride st: the abscissa of every step (EQM / DQM:
 * sim itself; QDM: its percentage rank); base (T, C) or NULL: the values the factor is applied to (NULL: xnew); gnew (T)
 * DEVICE float64: the group coordinate of every step (Grouper.get_index(interp=True): month - 0.5 + day / days_in_month, or
 * the day of year); node abscissae xq_all (G, nq, C) device float32 OR xq_common (nq) HOST float64 (QDM: the quantiles);
 * factors yq_all (G, nq, C); kind 0 (+) | 1 (*) | 2 (factor only); scen (T, C) row stride scen_st.  The Delaunay
 * triangle of a query is found by a dual-simplex walk from a starting triangle (plane.hip), no triangulation is stored;
 * cocircular node sets (regular grids) have no unique answer — see plane.hip.  Parity unpinned. */
int xh_plane_linear(xh_ctx* ctx, const float* xnew, const float* base, int64_t T, int64_t C, int64_t st, const double* gnew,
                    const float* xq_all, const double* xq_common, const float* yq_all, int G, int nq, int kind, float* scen,
                    int64_t scen_st);
/* interp = "nearest" with a month / day-of-year Grouper over the WHOLE series in one call (the per-group form is
 * xh_eqm_adjust_g2d above; same rule: upstream xsdba's _interp_on_quantiles_2D = scipy griddata(method="nearest") over the
 * nodes of all groups, then _extrapolate_on_quantiles with the step's own group; re-exported by
 * /root/reference/src/xclim/sdba.py:10).  Arguments as xh_plane_linear; gnew (T) DEVICE float64 must hold INTEGER group
 * coordinates 1 .. G (upstream passes the integer group index for "nearest"); nq <= 32; extrap 0 constant | 1 nan.  A row
 * kernel keeps a group's nodes in registers for all of its steps: the own row's nearest node stands when it is at most one
 * group step away, the rest is listed and searched over the neighbouring rows.  Parity unpinned. */
int xh_plane_nearest(xh_ctx* ctx, const float* xnew, const float* base, int64_t T, int64_t C, int64_t st, const double* gnew,
                     const float* xq_all, const double* xq_common, const float* yq_all, int G, int nq, int kind, int extrap, float* scen,
                     int64_t scen_st);
/* QuantileDeltaMapping.adjust (xsdba._adjustment.qdm_adjust, group "time"): sim_q = rank(sim, pct=True) along time
 * (average ranks of the valid samples r / n, rescaled mx (r/n - mn) / (mx - mn) as xsdba.utils.rank does);
 * af_t = interp_on_quantiles(sim_q, q, af) with the nq quantile nodes q (host, strictly increasing) as abscissa
 * (interp 0 nearest, 1 linear; extrap 0 constant, 1 nan; NaN factors dropped per cell); scen = sim + af_t (kind 0) or
 * sim * af_t (kind 1); kind 2: scen = af_t alone.  af (nq, C) float32 as trained by xh_eqm_train.  scen has the layout of sim (st, sc);
 * 1 <= T < 2^27: up to 32768 steps a column's keys stay in one workgroup (qdm.hip, qdm2.hip), longer series (1950-2100
 * daily = 55 152) are ranked through a global sort in column batches (qdm3.hip).  Parity unpinned (xsdba is not in the
 * reference tree). */
int xh_qdm_adjust(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, int64_t sc, const float* af,
                  const double* q /* host */, int nq, int kind, int interp, int extrap, float* scen);
/* qdm_adjust for ALL groups of a sub-grouping with small groups in one launch (a day-of-year grouping: one row per year; xsdba
 * ranks inside each group: group.apply(rank, sim, main_only=True)).  rows (host, offs[G] entries): the row numbers of group 0,
 * then of group 1, ...; offs (host, G + 1); af (G, nq, C) float32: every group's factors.  scen is written at the listed rows only
 * (not in place).  Bit-identical to xh_qdm_adjust on each group's gathered rows.  XH_ERR_NOTIMPL (no error text): a group of
 * more than 64 rows or nq > 32 — gather each group and call xh_qdm_adjust. */
int xh_qdm_adjust_groups(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, const int32_t* rows /* host */,
                         const int64_t* offs /* host */, int G, const float* af, const double* q /* host */, int nq, int kind,
                         int interp, int extrap, float* scen, int64_t scen_st);
/* xsdba.nbutils.vecquantiles(da, rnk, dim) (upstream xsdba, re-exported by /root/reference/src/xclim/sdba.py:10): ONE
 * quantile per cell at that cell's own probability q_cell[c] (DEVICE float64, NaN -> NaN), Hyndman-Fan type 7 over the
 * valid samples (= /root/reference/src/xclim/core/utils.py:370-491 with alpha = beta = 1).  x (T, C) with element strides
 * (st, sc), one of them 1; out (C) float32.  Any T < 2^31 (radix select per column).  Parity unpinned. */
int xh_quantile_cells(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const double* q_cell, float* out);
/* The value-replacement step of xsdba.processing.adapt_freq (upstream _processing._adapt_freq; precipitation
 * pre-processing before a multiplicative quantile mapping):
 *   sim_ad = sim.where(dP0 < 0, sim.where((rank < P0_ref) | (rank > P0_sim) | isnull(sim), (pth - thresh) * U + thresh))
 * rank = sim.rank(dim, pct=True): average ranks of the valid samples / their count (exact, through a sort of (key, time
 * index) pairs); per cell p0_ref / p0_sim / dp0 (DEVICE float64) and pth (DEVICE float32) come from the caller
 * (xh_threshold_count with "<=", xh_quantile_cells).  U in [0, 1) is a counter-based uniform keyed by (seed, tindex[t] or
 * t, cell0 + c) — upstream draws from numpy's global generator; oracle/sdba.py restates this one bit for bit.  tindex:
 * DEVICE int64 (T) global time index of every row, or NULL.  sim / scen (T, C) strides (st, sc), one of them 1.
 * Parity unpinned. */
int xh_adapt_freq(xh_ctx* ctx, const float* sim, int64_t T, int64_t C, int64_t st, int64_t sc, const double* p0_ref,
                  const double* p0_sim, const double* dp0, const float* pth, double thresh, uint64_t seed, const int64_t* tindex,
                  int64_t cell0, float* scen);

/* DetrendedQuantileMapping pieces (xsdba._adjustment.dqm_train / dqm_adjust, xsdba.detrending.PolyDetrend; parity
 * unpinned).  xh_poly_trend: per-cell least-squares polynomial of degree 0 (mean) or 1 over the valid samples of the
 * time-major series, p0[C] + p1[C] * (t - (T - 1) / 2) in float64 (p1 may be NULL for degree 0; nvalid[C] optional).
 * xh_trend_apply: out = x OP (p0[c] + p1[c] * (t - (T - 1) / 2)), mode 0 "+", 1 "-", 2 "*", 3 "/"; p1 NULL = a per-cell
 * constant (the scaling / normalisation steps); float64 arithmetic, rounded once to float32. */
int xh_poly_trend(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int degree, double* p0,
                  double* p1, int32_t* nvalid);
int xh_trend_apply(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const double* p0,
                   const double* p1, int mode, float* out, int64_t out_st);
/* The same two with the rows' own time coordinate u[t] (DEVICE float64, T) in place of the centred row number: the
 * per-group fit / trend of PolyDetrend with a sub-grouping (xsdba.detrending.PolyDetrend(group=...): DataArray.polyfit over
 * the time coordinate of the group's steps) on a gathered block of rows.  Parity unpinned. */
int xh_poly_trend_u(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int degree, const double* u,
                    double* p0, double* p1, int32_t* nvalid);
int xh_trend_apply_u(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, const double* u, const double* p0,
                     const double* p1, int mode, float* out, int64_t out_st);
/* out[t, c] = mean of the valid samples among x[t - window / 2 .. t + window / 2, c] (rows outside the series do not exist),
 * NaN when there is none: the series PolyDetrend fits when the Grouper has a window (xsdba.detrending
 * _polydetrend_get_trend: rolling(center=True).construct("window") then da.mean over the window dimension — upstream
 * xsdba, re-exported by /root/reference/src/xclim/sdba.py:10; parity unpinned).  float64 running sum, float32 result. */
int xh_window_nanmean(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, int64_t sc, int window, float* out,
                      int64_t out_st);
/* PolyDetrend / apply_correction over GROUPS of rows in one launch (DetrendedQuantileMapping.adjust with a sub-grouping; xsdba
 * detrending.PolyDetrend(group=...) + u.broadcast — not in the reference tree, /root/reference/src/xclim/sdba.py:10).  rows (host,
 * offs[G] entries): the row numbers of group 0, then of group 1, ...; offs (host, G + 1): where every group's rows start; u (DEVICE
 * float64, T): the coordinate of every row of x (e.g. days since the mean date of the row's group).  p0, p1: (G, C) float64.
 * xh_poly_trend_groups: the least-squares fit over each group's valid samples (p1 NULL for degree 0) — bit-identical to
 * xh_poly_trend_u on the gathered rows.  xh_trend_apply_groups: out[t, c] = x[t, c] OP (p0[g, c] + p1[g, c] u[t]) for the rows t of
 * every group g (mode as xh_trend_apply; p1 NULL: a per-group constant, u may then be NULL; rows in no group are not written; x ==
 * out allowed) — bit-identical to xh_trend_apply_u. */
int xh_poly_trend_groups(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* rows /* host */,
                         const int64_t* offs /* host */, int G, const double* u, int degree, double* p0, double* p1);
int xh_trend_apply_groups(xh_ctx* ctx, const float* x, int64_t T, int64_t C, int64_t st, const int32_t* rows /* host */,
                          const int64_t* offs /* host */, int G, const double* u, const double* p0, const double* p1, int mode,
                          float* out, int64_t out_st);

#ifdef __cplusplus
}
#endif
#endif /* XCLIM_HIP_H */
