"""A ~150-line stand-in for the part of ``xarray.DataArray`` that xclim_amd/xr_adapter.py touches (its module docstring
lists the protocol), so that the xarray-facing wrappers are executed on the GPU box, where xarray cannot be installed.
TEST INFRASTRUCTURE ONLY — nothing in the product imports it.

Also: stand-in MODULES with the reference's import structure (``make_reference_like_modules``): the index bodies are
restated from the reference in a few lines each (file:line given) and import ``threshold_count`` / ``compare`` /
``resample_doy`` BY NAME like the reference does, so that ``patch.install(env, modules)`` is exercised with the same
resolution rules as on a real installation.
"""
import types

import numpy as np


class _Dt:
    def __init__(self, coord):
        self._c = coord

    def __getattr__(self, name):
        if name == "calendar":
            return self._c._cal
        return DataArray(self._c._fields[name], dims=self._c.dims)


class _Resampled:
    def __init__(self, da, freq):
        self._da, self._freq = da, freq

    def first(self, **kw):
        from xclim_amd.timeaxis import TimeAxis

        f = self._da._fields
        seg, _ = TimeAxis(f["year"], f["month"], f["day"], self._da._cal).segments(self._freq)
        first = np.asarray(seg[:-1])
        lab = time_coord_like(self._da, first)
        return DataArray(self._da.values[first], coords={"time": lab}, dims=("time",))


class ChunkedArray:
    """Stand-in for a dask array behind a DataArray (``da.data``): ``.chunks`` / ``.dask`` / ``.shape`` / ``.dtype``, lazy
    ``transpose`` and slicing; turning it into numpy (``np.asarray``: what ``DataArray.values`` does) is RECORDED in
    ``loads`` (number of elements per materialisation), so a test can assert that a wrapper never pulled more than one
    block of a chunked field into host memory (xr_adapter.reduce_blocks; the reference: core/calendar.py:460-479)."""

    def __init__(self, array, chunks, loads=None, index=None, perm=None):
        self._a = array                      # the backing numpy array, original axis order
        self._index = index or tuple(slice(0, n) for n in array.shape)   # one slice per ORIGINAL axis
        self._perm = tuple(perm) if perm is not None else tuple(range(array.ndim))
        self._chunks0 = tuple(tuple(int(c) for c in ch) for ch in chunks)  # per ORIGINAL axis, of the full array
        self.loads = loads if loads is not None else []
        self.dask = {"stand-in": True}

    dtype = property(lambda self: self._a.dtype)
    ndim = property(lambda self: self._a.ndim)

    @property
    def shape(self):
        return tuple(self._index[ax].stop - self._index[ax].start for ax in self._perm)

    @property
    def chunks(self):
        out = []
        for ax in self._perm:
            sl, edges = self._index[ax], np.concatenate([[0], np.cumsum(self._chunks0[ax])])
            sizes = [min(e1, sl.stop) - max(e0, sl.start) for e0, e1 in zip(edges[:-1], edges[1:])]
            out.append(tuple(int(n) for n in sizes if n > 0))
        return tuple(out)

    def transpose(self, *perm):
        return ChunkedArray(self._a, self._chunks0, self.loads, self._index, tuple(self._perm[i] for i in perm))

    def __getitem__(self, key):
        key = key if isinstance(key, tuple) else (key,)
        key = key + (slice(None),) * (self.ndim - len(key))
        idx = list(self._index)
        for pos, k in enumerate(key):
            ax = self._perm[pos]
            start, stop, step = k.indices(self._index[ax].stop - self._index[ax].start)
            assert step == 1
            idx[ax] = slice(self._index[ax].start + start, self._index[ax].start + stop)
        return ChunkedArray(self._a, self._chunks0, self.loads, tuple(idx), self._perm)

    def __array__(self, dtype=None, copy=None):
        v = self._a[self._index].transpose(self._perm)
        self.loads.append(int(v.size))
        return v.astype(dtype) if dtype is not None else v


class DataArray:
    def __init__(self, data, coords=None, dims=None, name=None, attrs=None):
        self._data = data if isinstance(data, ChunkedArray) else np.asarray(data)
        self.dims = tuple(dims) if dims is not None else tuple(f"dim_{i}" for i in range(self._data.ndim))
        assert len(self.dims) == self._data.ndim, (self.dims, self._data.shape)
        self.name, self.attrs = name, dict(attrs or {})
        self.coords = {}
        self._fields, self._cal = None, None
        for k, v in (coords or {}).items():
            if not isinstance(v, DataArray):
                v = DataArray(np.asarray(v), dims=(k,))
            self.coords[k] = v

    dtype = property(lambda self: self._data.dtype)
    shape = property(lambda self: tuple(self._data.shape))
    data = property(lambda self: self._data)

    @property
    def values(self):
        return np.asarray(self._data)

    @values.setter
    def values(self, v):
        self._data = np.asarray(v)

    def isel(self, indexers=None, **kw):
        """Positional slices per dimension (slices only: what xr_adapter.block_values asks for)."""
        sel = dict(indexers or {}, **kw)
        key = tuple(sel.get(d, slice(None)) for d in self.dims)
        coords = {}
        for k, c in self.coords.items():
            if c.dims == (k,) and k in sel:
                cc = DataArray(c.values[sel[k]], dims=(k,))
                cc._fields = None if c._fields is None else {f: v[sel[k]] for f, v in c._fields.items()}
                cc._cal = c._cal
                coords[k] = cc
            else:
                coords[k] = c
        out = DataArray(self._data[key], coords=coords, dims=self.dims, name=self.name, attrs=self.attrs)
        out._fields, out._cal = self._fields, self._cal
        return out

    def __getitem__(self, key):
        return self.coords[key]

    @property
    def dt(self):
        return _Dt(self)

    def transpose(self, *dims):
        if Ellipsis in dims:
            i = dims.index(Ellipsis)
            rest = tuple(d for d in self.dims if d not in dims)
            dims = dims[:i] + rest + dims[i + 1:]
        perm = [self.dims.index(d) for d in dims]
        out = DataArray(self._data.transpose(*perm), coords=self.coords, dims=dims, name=self.name, attrs=self.attrs)
        out._fields, out._cal = self._fields, self._cal
        return out

    def resample(self, time):
        return _Resampled(self, time)

    def assign_attrs(self, **kw):
        out = self.copy()
        out.attrs.update(kw)
        return out

    def copy(self, data=None):
        out = DataArray(self.values.copy() if data is None else data, coords=self.coords, dims=self.dims, name=self.name,
                        attrs=self.attrs)
        out._fields, out._cal = self._fields, self._cal
        return out

    def sel(self, **kw):
        out = self
        for k, v in kw.items():
            ax = out.dims.index(k)
            j = int(np.nonzero(out.coords[k].values == v)[0][0])
            out = DataArray(np.take(out.values, j, axis=ax), coords={c: x for c, x in out.coords.items() if c != k},
                            dims=tuple(d for d in out.dims if d != k), name=out.name, attrs=out.attrs)
        return out

    def _bin(self, other, fn):
        o = other.transpose(*self.dims).values if isinstance(other, DataArray) else np.asarray(other)
        out = DataArray(fn(self.values, o), coords=self.coords, dims=self.dims)
        out._fields, out._cal = self._fields, self._cal
        return out

    def where(self, cond):
        c = cond.transpose(*self.dims).values if isinstance(cond, DataArray) else np.asarray(cond)
        out = self.copy(np.where(c, self.values.astype(np.result_type(self.values.dtype, np.float32)), np.nan))
        return out

    def __invert__(self):
        return self.copy(~self.values)

    def __gt__(self, o): return self._bin(o, np.greater)
    def __lt__(self, o): return self._bin(o, np.less)
    def __ge__(self, o): return self._bin(o, np.greater_equal)
    def __le__(self, o): return self._bin(o, np.less_equal)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __add__(self, o): return self._bin(o, np.add)
    def __sub__(self, o): return self._bin(o, np.subtract)
    __hash__ = object.__hash__


def time_coord(ta) -> DataArray:
    """The ``time`` coordinate of a daily TimeAxis (values = ordinals; ``.dt`` serves year / month / day / dayofyear)."""
    c = DataArray(np.asarray(ta.ordinal()), dims=("time",))
    c._fields = {"year": np.asarray(ta.year), "month": np.asarray(ta.month), "day": np.asarray(ta.day), "dayofyear": np.asarray(ta.doy)}
    c._cal = ta.calendar
    return c


def time_coord_like(c, idx) -> DataArray:
    out = DataArray(c.values[idx], dims=("time",))
    out._fields = {k: v[idx] for k, v in c._fields.items()}
    out._cal = c._cal
    return out


def field(x, ta, dims=("time", "lat", "lon"), attrs=None, name=None, chunks=None) -> DataArray:
    """A (time, lat, lon)-like DataArray (any dim order) on the TimeAxis ``ta``.  ``chunks``: {dim: chunk length} makes it
    "dask-backed" (a :class:`ChunkedArray` behind ``.data``; dimensions not named are one chunk)."""
    coords = {d: np.arange(n) for d, n in zip(dims, np.shape(x)) if d != "time"}
    coords["time"] = time_coord(ta)
    if chunks is not None:
        x = np.asarray(x)
        cks = []
        for d, n in zip(dims, x.shape):
            c = int(chunks.get(d, n))
            cks.append(tuple([c] * (n // c) + ([n % c] if n % c else [])))
        x = ChunkedArray(x, cks)
    return DataArray(x, coords=coords, dims=dims, attrs=attrs or {"units": "K"}, name=name)


def make_env():
    """xr_adapter.Env of the stand-in: thresholds are plain floats / DataArrays already in the data's units."""
    from xclim_amd.xr_adapter import Env

    def convert_units_to(thr, data, context=None):
        return thr

    def to_agg_units(out, orig, op, dim="time", **kw):
        out.attrs["units"] = {"count": "days", "integral": f"{orig.attrs.get('units', '')} days"}.get(op, orig.attrs.get("units", ""))
        return out

    return Env(DataArray, convert_units_to, to_agg_units)


def make_reference_like_modules(env):
    """name -> module, wired like the reference: ``generic`` / ``calendar`` / ``run_length`` / ``utils`` define the functions
    (here: stubs that fail loudly — after ``install`` nothing may reach them), ``_multivariate`` and ``_threshold`` import
    them BY NAME and hold ``rl`` as a module object (indices/_multivariate.py:13, 22-24; indices/_threshold.py:25-36)."""
    def stub(name):
        def f(*a, **k):
            raise AssertionError(f"the reference's {name} was reached: the wrapper did not replace it")
        f.__name__ = name
        return f

    mods = {}
    for modname, names in {
        "xclim.indices.generic": ("threshold_count", "count_occurrences", "domain_count", "select_resample_op",
                                  "spell_length_statistics", "cumulative_difference", "compare", "season",
                                  "first_day_threshold_reached", "bivariate_count_occurrences"),
        "xclim.core.calendar": ("percentile_doy", "resample_doy"),
        "xclim.indices.run_length": ("rle", "rle_statistics", "longest_run", "windowed_run_events", "windowed_run_count",
                                     "first_run", "last_run", "season_length", "resample_and_rl", "_cumsum_reset_np"),
        "xclim.core.utils": ("calc_perc",),
    }.items():
        m = types.ModuleType(modname)
        for n in names:
            setattr(m, n, stub(n))
        mods[modname] = m
    gen, cal, rl = mods["xclim.indices.generic"], mods["xclim.core.calendar"], mods["xclim.indices.run_length"]

    mv = types.ModuleType("xclim.indices._multivariate")
    mv.resample_doy, mv.compare, mv.threshold_count, mv.select_resample_op = cal.resample_doy, gen.compare, gen.threshold_count, gen.select_resample_op
    mv.rl = rl

    def tx90p(tasmax, tasmax_per, freq="YS", op=">"):  # indices/_multivariate.py:1584-1592
        tasmax_per = env.convert_units_to(tasmax_per, tasmax)
        thresh = mv.resample_doy(tasmax_per, tasmax)
        out = mv.threshold_count(tasmax, op, thresh, freq, constrain=(">", ">="))
        return env.to_agg_units(out, tasmax, "count", deffreq="D")

    def warm_spell_duration_index(tasmax, tasmax_per, window=6, freq="YS", resample_before_rl=True, op=">"):  # :1779-1793
        thresh = env.convert_units_to(tasmax_per, tasmax)
        thresh = mv.resample_doy(thresh, tasmax)
        above = mv.compare(tasmax, op, thresh, constrain=(">", ">="))
        out = mv.rl.resample_and_rl(above, resample_before_rl, mv.rl.windowed_run_count, window=window, freq=freq)
        return env.to_agg_units(out, tasmax, "count", deffreq="D")

    mv.tx90p, mv.warm_spell_duration_index = tx90p, warm_spell_duration_index
    mods[mv.__name__] = mv

    th = types.ModuleType("xclim.indices._threshold")
    th.spell_length_statistics, th.threshold_count, th.count_occurrences, th.domain_count = (
        gen.spell_length_statistics, gen.threshold_count, gen.count_occurrences, gen.domain_count)
    th.cumulative_difference, th.compare, th.rl = gen.cumulative_difference, gen.compare, rl

    def maximum_consecutive_dry_days(pr, thresh=1.0 / 86400.0, op="<", freq="YS", resample_before_rl=True):  # _threshold.py:2925-2937
        return th.spell_length_statistics(pr, thresh, 1, win_reducer=None, op=op, spell_reducer="max", freq=freq,
                                          resample_before_rl=resample_before_rl)

    def frost_days(tasmin, thresh=273.15, freq="YS"):  # _threshold.py: threshold_count(tasmin, "<", frz, freq) + to_agg_units
        out = th.threshold_count(tasmin, "<", env.convert_units_to(thresh, tasmin), freq)
        return env.to_agg_units(out, tasmin, "count")

    def hot_spell_frequency(tasmax, thresh=303.15, window=3, freq="YS", op=">", resample_before_rl=True):  # _threshold.py (hot_spell_frequency)
        cond = th.compare(tasmax, op, thresh, constrain=(">", ">="))
        return th.rl.resample_and_rl(cond, resample_before_rl, th.rl.windowed_run_events, window=window, freq=freq)

    def growing_degree_days(tas, thresh=277.15, freq="YS"):  # _threshold.py: cumulative_difference(tas, thresh, ">", freq)
        return th.cumulative_difference(tas, threshold=thresh, op=">", freq=freq)

    th.season, th.first_day_threshold_reached = gen.season, gen.first_day_threshold_reached
    th.bivariate_count_occurrences = gen.bivariate_count_occurrences

    def tx_tn_days_above(tasmin, tasmax, thresh_tasmin=295.15, thresh_tasmax=303.15, freq="YS", op=">"):  # _threshold.py (tx_tn_days_above)
        return th.bivariate_count_occurrences(data_var1=tasmin, data_var2=tasmax, threshold_var1=thresh_tasmin, threshold_var2=thresh_tasmax,
                                              freq=freq, op_var1=op, op_var2=op, var_reducer="all", constrain_var1=(">", ">="),
                                              constrain_var2=(">", ">="))

    th.tx_tn_days_above = tx_tn_days_above

    def growing_season_length(tas, thresh=278.15, window=6, mid_date="07-01", freq="YS", op=">="):  # _threshold.py (growing_season_length)
        return th.season(tas, thresh=thresh, window=window, op=op, stat="length", freq=freq, mid_date=mid_date, constrain=(">=", ">"))

    def growing_season_start(tas, thresh=278.15, mid_date="07-01", window=5, freq="YS", op=">="):
        return th.season(tas, thresh=thresh, window=window, op=op, stat="start", freq=freq, mid_date=mid_date, constrain=(">=", ">"))

    def first_day_temperature_above(tas, thresh=273.15, op=">", after_date="01-01", window=1, freq="YS"):  # _threshold.py
        return th.first_day_threshold_reached(tas, threshold=thresh, op=op, after_date=after_date, window=window, freq=freq,
                                              constrain=(">", ">="))

    th.growing_season_length, th.growing_season_start, th.first_day_temperature_above = (
        growing_season_length, growing_season_start, first_day_temperature_above)
    th.maximum_consecutive_dry_days, th.frost_days, th.hot_spell_frequency, th.growing_degree_days = (
        maximum_consecutive_dry_days, frost_days, hot_spell_frequency, growing_degree_days)
    mods[th.__name__] = th

    ms = types.ModuleType("xclim.core.missing")

    class MissingAny:  # core/missing.py:311-322 (+ MissingBase.__call__ :253-298)
        def __call__(self, da, freq=None, src_timestep=None, **indexer):
            raise AssertionError("the reference's MissingAny.__call__ was reached: the wrapper did not replace it")

    ms.MissingAny = MissingAny
    mods[ms.__name__] = ms

    sp = types.ModuleType("xclim.indices._simple")
    sp.select_resample_op = gen.select_resample_op

    def tg_mean(tas, freq="YS"):  # indices/_simple.py:113
        return sp.select_resample_op(tas, op="mean", freq=freq)

    sp.tg_mean = tg_mean
    mods[sp.__name__] = sp

    su = types.ModuleType("xsdba.utils")
    su.interp_on_quantiles = stub("interp_on_quantiles")
    mods[su.__name__] = su
    sa = types.ModuleType("xsdba._adjustment")
    sa.u = su

    def qm_adjust(sim, af, hist_q, kind="+", interp="nearest", extrapolation="constant"):  # xsdba._adjustment.qm_adjust, group="time"
        af_t = sa.u.interp_on_quantiles(sim, hist_q, af, group="time", method=interp, extrapolation=extrapolation)
        return sim._bin(af_t, np.add if kind == "+" else np.multiply)              # utils.apply_correction

    sa.qm_adjust = qm_adjust
    mods[sa.__name__] = sa
    return mods
