"""HIP-event timing of the secondary kernels on 365 x 1440 x 720 fp32 (one MI355X).  Run on the GPU box:
python tools/bench_kernels.py  ->  one JSON line per kernel: ms, algorithmic GB/s, fraction of the 8 TB/s peak."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from xclim_amd import kernels as K  # noqa: E402
from xclim_amd._capi import Device  # noqa: E402
from xclim_amd.timeaxis import TimeAxis  # noqa: E402

T, Y, X = 365, 1440, 720
C = Y * X
E = float(T) * C
dev = Device(0)
ta = TimeAxis.daily("2001-01-01", T, "noleap")
seg_y, _ = ta.segments("YS")
seg_m, _ = ta.segments("MS")
base = bench.seasonal_base(T)
tas = K.fill_synthetic(dev, T, C, 0, 2, base, 3.0)
tas2 = K.fill_synthetic(dev, T, C, 0, 7, base + np.float32(6.0), 3.0)
pr = K.fill_synthetic(dev, T, C, 1, 3, np.zeros(T, np.float32), 40.0 / 86400.0, 0.3)
mask = K.compare_map(dev, pr, "<", 1.0 / 86400.0, "maskf")


def run(name, fn, nbytes, reps=5):
    ms = bench.event_time(dev, fn, reps)
    print(json.dumps({"kernel": name, "ms": round(ms, 4), "GB/s": round(nbytes / ms / 1e6, 1),
                      "frac": round(nbytes / ms / 1e6 / bench.HBM_PEAK_GBS, 3)}), flush=True)


P, PM = len(seg_y) - 1, len(seg_m) - 1
run("threshold_count scalar YS", lambda: K.threshold_count(dev, tas, ">", seg_y, scalar=290.0), 4 * E)
run("threshold_count scalar MS", lambda: K.threshold_count(dev, tas, ">", seg_m, scalar=290.0), 4 * E)
run("domain_count", lambda: K.domain_count(dev, tas, ">", 280.0, "<=", 295.0, "and", seg_y), 4 * E)
for red in ("mean", "max", "std", "argmax"):
    run(f"resample_reduce {red} YS", lambda red=red: K.resample_reduce(dev, tas, red, seg_y), (8 if red == "std" else 4) * E)
run("resample_reduce mean MS", lambda: K.resample_reduce(dev, tas, "mean", seg_m), 4 * E)
run("rolling_reduce mean w5", lambda: K.rolling_reduce(dev, tas, 5, "mean", True), 8 * E)
run("rolling_reduce sum w5", lambda: K.rolling_reduce(dev, tas, 5, "sum", True), 8 * E)
run("rolling_reduce max w5", lambda: K.rolling_reduce(dev, tas, 5, "max", True), 8 * E)
run("rolling_reduce mean w3", lambda: K.rolling_reduce(dev, tas, 3, "mean", True), 8 * E)
run("bivariate_count", lambda: K.bivariate_count(dev, tas, tas2, "<", 285.0, ">", 295.0, "all", seg_y), 8 * E)
run("thresholded_reduce sum", lambda: K.thresholded_reduce(dev, tas, ">", 290.0, 0, "sum", seg_y), 4 * E)
run("range_reduce mean (dtr)", lambda: K.range_reduce(dev, tas, tas2, "range", "mean", seg_y), 8 * E)
run("range_reduce interday", lambda: K.range_reduce(dev, tas, tas2, "interday", "mean", seg_y), 8 * E)
run("compare_map maskf", lambda: K.compare_map(dev, tas, ">", 290.0, "maskf"), 8 * E)
run("cumsum_reset", lambda: K.cumsum_reset(dev, mask, "last"), 8 * E)
run("rle", lambda: K.rle(dev, mask, "first"), 8 * E)
for stat, w in (("max", 1), ("sum", 6), ("count", 3), ("first", 3)):
    run(f"run_stats {stat} w{w} (mask, cut)", lambda stat=stat, w=w: K.run_stats(dev, mask, stat, w, seg_y, cut=True), 4 * E)
run("run_stats max w1 fused compare", lambda: K.run_stats(dev, pr, "max", 1, seg_y, cut=True, fused_op="<", thresh=1.0 / 86400.0), 4 * E)
run("run_stats sum w6 MS no-cut", lambda: K.run_stats(dev, mask, "sum", 6, seg_m, cut=False), 4 * E)
run("spell_mask w3 mean", lambda: K.spell_mask(dev, pr, 3, "mean", ">=", 1.0 / 86400.0), 8 * E)
run("spell_run_stats w3 mean max (fused)", lambda: K.spell_run_stats(dev, pr, 3, "mean", ">=", 1.0 / 86400.0, "max", seg_y), 4 * E)
run("runs_with_holes", lambda: K.runs_with_holes(dev, mask, 3, None, 2), 8 * E)
run("keep_longest_run", lambda: K.keep_longest_run(dev, mask, seg_y), 8 * E)
run("season w5", lambda: K.season(dev, mask, 5, seg_y, None), 4 * E)
run("max_run_sum w3", lambda: K.max_run_sum(dev, pr, 3, seg_y), 4 * E)
tb, years, doys = ta.doy_table()
run("doy_mean_std w5", lambda: K.doy_mean_std(dev, tas, tb, 5), 4 * E + 8 * len(doys) * C)
# a climatology is taken over decades: 30 years x (1440 x 72) cells, the same 4.5 GB of samples per year-block
T30, C30 = 365 * 30, 1440 * 72
ta30 = TimeAxis.daily("1971-01-01", T30, "noleap")
tb30, _, doys30 = ta30.doy_table()
tas30 = K.fill_synthetic(dev, T30, C30, 0, 5, bench.seasonal_base(T30), 3.0)
run("doy_mean_std w5 30yr", lambda: K.doy_mean_std(dev, tas30, tb30, 5), 4.0 * T30 * C30 + 8 * len(doys30) * C30)
del tas30
p = K.percentile_doy(dev, tas, tb, 5, [90.0])
tidx = np.searchsorted(doys, ta.doy).astype(np.int32)
run("compare_doy", lambda: K.compare_doy(dev, tas, ">", p.reshape(len(doys), C), tidx), 8 * E + 8 * len(doys) * C)
run("run_stats_doy sum w6 (WSDI fused)", lambda: K.run_stats_doy(dev, tas, ">", p.reshape(len(doys), C), tidx, "sum", 6, seg_y),
    4 * E + 8 * len(doys) * C)
run("precip_over_doy (count+frac)", lambda: K.precip_over_doy(dev, tas, ">", 280.0, p.reshape(len(doys), C), tidx, seg_y,
                                                              want=("count", "frac")), 4 * E + 8 * len(doys) * C)
run("doy_broadcast", lambda: K.doy_broadcast(dev, p.reshape(len(doys), C), tidx), 16 * E)
run("transpose", lambda: K.transpose(dev, tas), 8 * E)
# the form the time-major pipelines use: output pitch padded to 64 floats (256-byte aligned column segments, 16-byte stores)
from xclim_amd._capi import _vp  # noqa: E402
tpad = dev.empty((C, 384), np.float32)
run("transpose (pitch 384)", lambda: dev.call("xh_transpose_f32", _vp(tas.ptr), T, C, C, _vp(tpad.ptr), 384), 8 * E)
del tpad
# quantile delta mapping: exact per-column ranks + factor lookup (transposed scratch both ways inside)
qn = (np.arange(20) + 0.5) / 20
af_q, _ = K.eqm_train(dev, tas, tas2, qn, "+")
run("qdm_adjust nearest", lambda: K.qdm_adjust(dev, tas, af_q, qn, "+", "nearest"), 8 * E)
run("qdm_adjust linear", lambda: K.qdm_adjust(dev, tas, af_q, qn, "+", "linear"), 8 * E)
run("qdm_adjust nearest (precipitation)", lambda: K.qdm_adjust(dev, pr, af_q, qn, "*", "nearest"), 8 * E)
