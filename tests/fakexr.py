"""A small stand-in for the part of ``xarray.DataArray`` that xclim_amd/xr_adapter.py touches (its module docstring
lists the protocol), so that the xarray-facing wrappers are executed on the GPU box, where xarray cannot be installed.
TEST INFRASTRUCTURE ONLY — nothing in the product imports it.

Also: stand-in MODULES with the reference's import structure (``make_reference_like_modules``).  Their index functions are
NOT re-typed: each one replays the call program that tests/golden/make_call_programs.py recorded by executing the
reference's own body (tests/callprog.py), inside the stand-in module's namespace, where ``threshold_count`` / ``compare``
/ ``resample_doy`` are held BY NAME and ``rl`` as a module object like in the reference — so ``patch.install(env,
modules)`` is exercised with the reference's resolution rules AND the reference's call sequences.  What
the DataArray stand-in offers beyond the adapter's protocol (``where(cond, other)``, ``&``, ``resample().sum()``,
``attrs[...] = ``) is there because a recorded reference line uses it.
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

    def _reduce(self, fn, dim="time"):
        """resample(time=freq).<sum|max|min|mean>(dim="time") of a data array (host numpy: not a replaced function)."""
        from xclim_amd.timeaxis import TimeAxis

        da = self._da
        c = da.coords["time"]
        seg, _ = TimeAxis(c._fields["year"], c._fields["month"], c._fields["day"], c._cal).segments(self._freq)
        ax = da.dims.index("time")
        v = np.moveaxis(da.values, ax, 0)
        out = np.stack([fn(v[a:b], axis=0) for a, b in zip(seg[:-1], seg[1:])])
        coords = dict(da.coords)
        coords["time"] = time_coord_like(c, np.asarray(seg[:-1]))
        return DataArray(np.moveaxis(out, 0, ax), coords=coords, dims=da.dims, attrs=da.attrs)

    def sum(self, dim="time"):
        return self._reduce(np.sum, dim)

    def max(self, dim="time"):
        return self._reduce(np.max, dim)

    def min(self, dim="time"):
        return self._reduce(np.min, dim)


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
            # (a VIEW, like xarray's own basic indexing for a scalar label)
            out = DataArray(out.values[(slice(None),) * ax + (j,)], coords={c: x for c, x in out.coords.items() if c != k},
                            dims=tuple(d for d in out.dims if d != k), name=out.name, attrs=out.attrs)
        return out

    def _bin(self, other, fn):
        o = other.transpose(*self.dims).values if isinstance(other, DataArray) else np.asarray(other)
        out = DataArray(fn(self.values, o), coords=self.coords, dims=self.dims)
        out._fields, out._cal = self._fields, self._cal
        return out

    def where(self, cond, other=None):
        c = cond.transpose(*self.dims).values if isinstance(cond, DataArray) else np.asarray(cond)
        if other is None:
            return self.copy(np.where(c, self.values.astype(np.result_type(self.values.dtype, np.float32)), np.nan))
        o = other.transpose(*self.dims).values if isinstance(other, DataArray) else other
        return self.copy(np.where(c, self.values, o))

    def __and__(self, o): return self._bin(o, np.logical_and)
    def __or__(self, o): return self._bin(o, np.logical_or)
    def __truediv__(self, o): return self._bin(o, np.divide)

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
        """core/units.py:334-420 as far as the recorded index bodies need it: threshold STRINGS ("1 mm/day", "30 degC") in
        the units of ``data`` (tests/fakeunits.py, the hydro context is the bodies' ``units.context("hydro")``); numbers and
        DataArrays are taken to be in the data's units already."""
        import fakeunits

        if isinstance(thr, str):
            return fakeunits.convert_units_to(thr, data.attrs["units"], context="hydro")
        if isinstance(thr, DataArray) and isinstance(data, str):   # a FIELD into other units: convert_units_to(pr, "mm/d")
            out = thr * np.float32(fakeunits.convert_units_to("1 " + thr.attrs["units"], data, context="hydro"))
            out.attrs = dict(thr.attrs, units=data)
            return out
        return thr

    def to_agg_units(out, orig, op, dim="time", **kw):
        out.attrs["units"] = {"count": "days", "integral": f"{orig.attrs.get('units', '')} days"}.get(op, orig.attrs.get("units", ""))
        return out

    return Env(DataArray, convert_units_to, to_agg_units)


def make_reference_like_modules(env):
    """name -> module, wired like the reference: ``generic`` / ``calendar`` / ``run_length`` / ``utils`` define the functions
    (here: stubs that fail loudly — after ``install`` nothing may reach them), ``_multivariate`` / ``_threshold`` / ``_simple``
    import them BY NAME and hold ``rl`` as a module object (indices/_multivariate.py:13, 22-24; indices/_threshold.py:25-36;
    indices/_simple.py:10).  Their index functions replay the reference's recorded call programs (tests/callprog.py)."""
    import contextlib

    import callprog

    def stub(name):
        def f(*a, **k):
            raise AssertionError(f"the reference's {name} was reached: the wrapper did not replace it")
        f.__name__ = name
        return f

    generic_names = ("threshold_count", "count_occurrences", "domain_count", "select_resample_op", "spell_length_statistics",
                     "cumulative_difference", "compare", "season", "first_day_threshold_reached", "bivariate_count_occurrences")
    mods = {}
    for modname, names in {
        "xclim.indices.generic": generic_names,
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

    # what the index modules import (by name) or hold besides the replaced functions: the unit helpers of core/units.py —
    # here the Env's stand-ins — and the ``units`` registry whose "hydro" context some bodies enter
    units = types.SimpleNamespace(context=lambda name: contextlib.nullcontext())
    for modname in ("xclim.indices._multivariate", "xclim.indices._threshold", "xclim.indices._simple"):
        m = types.ModuleType(modname)
        for n in generic_names:
            setattr(m, n, getattr(gen, n))
        m.percentile_doy, m.resample_doy, m.rl = cal.percentile_doy, cal.resample_doy, rl
        m.convert_units_to, m.to_agg_units, m.units = env.convert_units_to, env.to_agg_units, units

        def rate2amount(da, out_units=None):
            """core/units.py rate2amount for DAILY fields: a rate per day times one day (the magnitude stays, the unit
            loses its time part)"""
            assert da.attrs["units"] in ("mm/d", "mm/day") and out_units == "mm"
            out = da.copy()
            out.attrs = dict(da.attrs, units="mm")
            return out

        m.rate2amount = rate2amount
        mods[modname] = m
    for name, prog in callprog.load_programs().items():
        m = mods[prog["module"]]
        setattr(m, name, callprog.make_index(name, prog, m.__dict__))

    ms = types.ModuleType("xclim.core.missing")

    class MissingAny:  # core/missing.py:311-322 (+ MissingBase.__call__ :253-298)
        def __call__(self, da, freq=None, src_timestep=None, **indexer):
            raise AssertionError("the reference's MissingAny.__call__ was reached: the wrapper did not replace it")

    ms.MissingAny = MissingAny

    class _MissingBase:  # core/missing.py:162-177, 253-298: the options live in self.options; __call__ is INHERITED by the methods
        def __init__(self, **options):
            self.options = options

        def __call__(self, da, freq=None, src_timestep=None, **indexer):
            raise AssertionError(f"the reference's {type(self).__name__}.__call__ was reached: the wrapper did not replace it")

    class MissingSomeButNotAll(_MissingBase):
        pass

    class MissingWMO(_MissingBase):
        def __init__(self, nm=11, nc=5):
            super().__init__(nm=nm, nc=nc, subfreq="MS")

    class MissingPct(_MissingBase):
        def __init__(self, tolerance=0.1, subfreq=None):
            super().__init__(tolerance=tolerance, subfreq=subfreq)

    class AtLeastNValid(_MissingBase):
        def __init__(self, n=20, subfreq=None):
            super().__init__(n=n, subfreq=subfreq)

    ms.MissingSomeButNotAll, ms.MissingWMO, ms.MissingPct, ms.AtLeastNValid = MissingSomeButNotAll, MissingWMO, MissingPct, AtLeastNValid
    mods[ms.__name__] = ms

    im = types.ModuleType("xclim.core.indicator")

    class Indicator:  # the data flow of core/indicator.py:865-944 + :1522-1549: compute, then the missing-value mask
        def __init__(self, compute, src_freq="D"):
            self.compute, self.src_freq = compute, src_freq

        def __call__(self, da, *args, freq="YS", **kw):
            out = self.compute(da, *args, freq=freq, **kw)
            miss = ms.MissingAny()(da, freq, self.src_freq)
            return out.where(~miss)

    im.Indicator = Indicator
    mods[im.__name__] = im

    # xsdba is not in the reference tree (src/xclim/sdba.py:10 re-exports the installed package): nothing to record —
    # qm_adjust below restates xsdba._adjustment.qm_adjust for group="time" (interp_on_quantiles + apply_correction)
    su = types.ModuleType("xsdba.utils")
    su.interp_on_quantiles = stub("interp_on_quantiles")
    mods[su.__name__] = su
    sa = types.ModuleType("xsdba._adjustment")
    sa.u = su

    def qm_adjust(sim, af, hist_q, kind="+", interp="nearest", extrapolation="constant"):
        af_t = sa.u.interp_on_quantiles(sim, hist_q, af, group="time", method=interp, extrapolation=extrapolation)
        return sim._bin(af_t, np.add if kind == "+" else np.multiply)              # utils.apply_correction

    sa.qm_adjust = qm_adjust
    mods[sa.__name__] = sa
    return mods
