"""Fixtures for the sdba (xsdba) half of the hot path, produced BY xsdba ITSELF — to be run where xsdba >= 0.4.0 and
xarray are installed (they cannot be installed in the build container, where this script exits with a message):

    python tests/golden/make_sdba_golden.py          # writes tests/golden/sdba_vectors.npz

It calls the upstream entry points that oracle/sdba.py restates (table in its module docstring) on seeded inputs and
stores inputs + outputs; tests/test_gpu_sdba_golden.py compares the HIP path (and tests/test_oracle_golden.py the oracle)
with the stored numbers, rtol 1e-6 (the north star's tolerance for float quantiles).  Until the file exists, the sdba
rows of the parity table stay "unpinned" (DESIGN.md §5).
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def main() -> int:
    try:
        import xarray as xr
        import xsdba
        from xsdba import DetrendedQuantileMapping, EmpiricalQuantileMapping, QuantileDeltaMapping
        from xsdba import nbutils, utils
    except ImportError as exc:
        print(f"make_sdba_golden: xsdba / xarray are not importable here ({exc}); nothing written")
        return 0

    rng = np.random.default_rng(20260926)
    T, Y, X = 365 * 4, 3, 4
    time = xr.date_range("2001-01-01", periods=T, freq="D", calendar="noleap", use_cftime=True)
    t = np.arange(T)[:, None, None]

    def field(mean, amp, sigma, units="K", nan_frac=0.0):
        x = (mean + amp * np.sin(2 * np.pi * (t - 100) / 365) + rng.normal(0, sigma, (T, Y, X))).astype(np.float32)
        if nan_frac:
            x[rng.random(x.shape) < nan_frac] = np.nan
        return x, units

    def da(x, units):
        return xr.DataArray(x, dims=("time", "lat", "lon"), coords={"time": time, "lat": np.arange(Y), "lon": np.arange(X)},
                            attrs={"units": units})

    out = {"xsdba_version": np.array(xsdba.__version__), "T": np.array(T)}
    ref, u = field(288, 12, 3.0, nan_frac=0.002)
    hist, _ = field(289.5, 12, 3.3)
    sim, _ = field(291.5, 12, 3.3, nan_frac=0.002)
    pr_ref = np.where(rng.random((T, Y, X)) < 0.3, rng.gamma(0.8, 8.0, (T, Y, X)), 0.0).astype(np.float32) + np.float32(1e-3)
    pr_hist = (pr_ref[::-1] * 1.2 + np.float32(1e-3)).astype(np.float32)
    pr_sim = (pr_ref * 1.1).astype(np.float32)
    out.update(ref=ref, hist=hist, sim=sim, pr_ref=pr_ref, pr_hist=pr_hist, pr_sim=pr_sim)

    out["nodes_20"] = np.asarray(utils.equally_spaced_nodes(20))
    out["nodes_eps"] = np.asarray(utils.equally_spaced_nodes(15, eps=1e-6))
    q = out["nodes_20"]
    out["quantile"] = np.asarray(nbutils.quantile(da(ref, "K"), q, "time").transpose("quantiles", "lat", "lon").values)

    for kind, (r, h, s, un) in {"+": (ref, hist, sim, "K"), "*": (pr_ref, pr_hist, pr_sim, "mm/d")}.items():
        k = "add" if kind == "+" else "mul"
        eqm = EmpiricalQuantileMapping.train(da(r, un), da(h, un), nquantiles=20, kind=kind, group="time")
        out[f"eqm_{k}_af"] = eqm.ds.af.transpose("quantiles", "lat", "lon").values
        out[f"eqm_{k}_hist_q"] = eqm.ds.hist_q.transpose("quantiles", "lat", "lon").values
        for interp in ("nearest", "linear", "cubic"):
            for extrap in ("constant", "nan"):
                scen = eqm.adjust(da(s, un), interp=interp, extrapolation=extrap)
                out[f"eqm_{k}_{interp}_{extrap}"] = scen.transpose("time", "lat", "lon").values
        qdm = QuantileDeltaMapping.train(da(r, un), da(h, un), nquantiles=20, kind=kind, group="time")
        out[f"qdm_{k}_af"] = qdm.ds.af.transpose("quantiles", "lat", "lon").values
        for interp in ("nearest", "linear", "cubic"):   # (cubic: round 5)
            out[f"qdm_{k}_{interp}"] = qdm.adjust(da(s, un), interp=interp).transpose("time", "lat", "lon").values
        dqm = DetrendedQuantileMapping.train(da(r, un), da(h, un), nquantiles=20, kind=kind, group="time")
        out[f"dqm_{k}_af"] = dqm.ds.af.transpose("quantiles", "lat", "lon").values
        out[f"dqm_{k}_hist_q"] = dqm.ds.hist_q.transpose("quantiles", "lat", "lon").values
        out[f"dqm_{k}_scaling"] = np.asarray(dqm.ds.scaling.transpose("lat", "lon").values)
        for deg in (0, 1):
            scen = dqm.adjust(da(s, un), interp="nearest", detrend=deg)
            out[f"dqm_{k}_scen_d{deg}"] = scen.transpose("time", "lat", "lon").values

    for group, window in (("time.month", 1), ("time.dayofyear", 31)):
        g = xsdba.Grouper(group, window=window)
        tag = group.split(".")[1]
        eqm = EmpiricalQuantileMapping.train(da(ref, "K"), da(hist, "K"), nquantiles=15, kind="+", group=g)
        gdim = [d for d in eqm.ds.af.dims if d not in ("quantiles", "lat", "lon")][0]
        out[f"eqmg_{tag}_labels"] = np.asarray(eqm.ds.af[gdim].values)
        out[f"eqmg_{tag}_af"] = eqm.ds.af.transpose(gdim, "quantiles", "lat", "lon").values
        out[f"eqmg_{tag}_hist_q"] = eqm.ds.hist_q.transpose(gdim, "quantiles", "lat", "lon").values
        out[f"eqmg_{tag}_scen"] = eqm.adjust(da(sim, "K"), interp="nearest").transpose("time", "lat", "lon").values
        # round 5: interp="linear" = griddata over the (quantile, group) plane (utils.interp_on_quantiles, 2-D branch)
        out[f"eqmg_{tag}_scen_linear"] = eqm.adjust(da(sim, "K"), interp="linear").transpose("time", "lat", "lon").values
        qdm = QuantileDeltaMapping.train(da(ref, "K"), da(hist, "K"), nquantiles=15, kind="+", group=g)
        out[f"qdmg_{tag}_scen_linear"] = qdm.adjust(da(sim, "K"), interp="linear").transpose("time", "lat", "lon").values
        # round 5: DetrendedQuantileMapping with the same (windowed) Grouper — PolyDetrend fits on the window mean
        dqm = DetrendedQuantileMapping.train(da(ref, "K"), da(hist, "K"), nquantiles=15, kind="+", group=g)
        out[f"dqmg_{tag}_af"] = dqm.ds.af.transpose(gdim, "quantiles", "lat", "lon").values
        out[f"dqmg_{tag}_scaling"] = dqm.ds.scaling.transpose(gdim, "lat", "lon").values
        for interp in ("nearest", "linear"):
            out[f"dqmg_{tag}_scen_{interp}"] = dqm.adjust(da(sim, "K"), interp=interp, detrend=1).transpose("time", "lat", "lon").values
    # precipitation in mm/d with a month grouping: node spacings of many group steps (triangles spanning several months)
    g = xsdba.Grouper("time.month")
    eqm = EmpiricalQuantileMapping.train(da(pr_ref, "mm/d"), da(pr_hist, "mm/d"), nquantiles=15, kind="*", group=g)
    out["eqmg_month_pr_scen_linear"] = eqm.adjust(da(pr_sim, "mm/d"), interp="linear").transpose("time", "lat", "lon").values

    # adapt_freq (xsdba.processing.adapt_freq): pth and dP0 are deterministic; sim_ad draws its fill values from numpy's
    # global generator, so only the UNCHANGED samples of sim_ad (and which samples changed) can be pinned
    from xsdba.processing import adapt_freq

    dry = rng.random((T, Y, X))
    ad_ref = np.where(dry < 0.4, rng.random((T, Y, X)) * 0.3, rng.gamma(0.7, 5.0, (T, Y, X))).astype(np.float32)
    ad_sim = np.where(dry[::-1] < 0.7, rng.random((T, Y, X)) * 0.3, rng.gamma(0.7, 4.0, (T, Y, X))).astype(np.float32)
    out.update(adapt_ref=ad_ref, adapt_sim=ad_sim, adapt_thresh=np.array(0.5))
    for group, window in (("time", 1), ("time.month", 1)):
        tag = group.split(".")[-1]
        sim_ad, pth, dP0 = adapt_freq(da(ad_ref, "mm/d"), da(ad_sim, "mm/d"), thresh="0.5 mm/d", group=xsdba.Grouper(group, window=window))
        lead = [d for d in pth.dims if d not in ("lat", "lon")]
        out[f"adapt_{tag}_pth"] = pth.transpose(*lead, "lat", "lon").values
        out[f"adapt_{tag}_dP0"] = dP0.transpose(*lead, "lat", "lon").values
        out[f"adapt_{tag}_sim_ad"] = sim_ad.transpose("time", "lat", "lon").values

    path = os.path.join(HERE, "sdba_vectors.npz")
    out["format"] = np.int64(6)   # the generator's round: tests/test_gpu_sdba_golden.py REQUIRES every key of that round and earlier
    np.savez_compressed(path, **out)
    print(f"make_sdba_golden: wrote {path} ({len(out)} arrays, xsdba {xsdba.__version__})")
    return 0


if __name__ == "__main__":
    sys.exit(main())
