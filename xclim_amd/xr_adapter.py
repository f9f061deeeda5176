"""Tier 1 of the drop-in boundary: same-signature replacements of the reference's hot-path functions
that take and return ``xarray.DataArray`` — unwrap (time first), call the host mirror (-> HIP kernels), re-wrap with the
coordinates and unit attributes the reference would give.

Nothing here imports xarray or xclim: the few things the wrappers need from them come in through an :class:`Env`
(``patch.install()`` builds it from the real packages; tests/fakexr.py provides a ~150-line stand-in so that every
wrapper is EXECUTED on the GPU box, where xarray is not installable).  What a DataArray has to offer (real or fake):

    da.dims, da.dtype, da.attrs, da.name, da.values, da.coords, da[name], da.transpose("time", ...),
    da["time"].dt.{year, month, day, dayofyear, calendar}, da["time"].resample(time=freq).first()  (period labels),
    Env.DataArray(data, coords=..., dims=..., name=..., attrs=...), isinstance(x, Env.DataArray)

Reference functions replaced (signatures identical, file:line of the original):

    indices/generic.py      threshold_count :329, count_occurrences :960, domain_count :364, select_resample_op :83,
                            spell_length_statistics :588, cumulative_difference :1514, compare :301, season :770, bivariate_count_occurrences :1003,
                            first_day_threshold_reached :1556
    core/calendar.py        percentile_doy :395, resample_doy :763
    indices/run_length.py   rle :223, rle_statistics :275, longest_run :338, windowed_run_events :381,
                            windowed_run_count :437, first_run :643, last_run :693, season_length :1113,
                            resample_and_rl :87, _cumsum_reset_np :143
    core/utils.py           calc_perc :279
    core/missing.py         MissingAny.__call__ (MissingBase.__call__ :253 + is_valid :201 + is_missing :318)
    xsdba (when importable) nbutils.quantile

The chain of an index stays on the fused kernels the benchmark measures: ``resample_doy`` returns a :class:`DoyThreshold`
(the per-doy table on the device + the target time axis — NOT the (time, lat, lon) float64 field of the reference),
``threshold_count`` takes it through ``XH_THR_DOY_F64``; ``compare`` of a DataArray with a DoyThreshold returns a
:class:`LazyCompare` that ``resample_and_rl`` runs as ONE fused launch (``xh_run_stats_doy``); ``spell_length_statistics``
with window 1 (maximum_consecutive_dry_days ...) is the fused compare + run-length kernel.  Both lazy objects turn into
real DataArrays the moment anything else touches them (attribute access / arithmetic), so code that was not replaced
keeps working.  Indicator-level fusion: every reducer's kernel also returns the per-period count of valid steps; the
wrappers remember it per input buffer and the replaced ``MissingAny.__call__`` (what ``Indicator._postprocess`` runs on
the same DataArray right after the compute, core/indicator.py:1522-1549) answers from it — compute and missing-value
mask in ONE pass over the data; unit conversion costs no pass either, because the reference converts the THRESHOLD to
the data's units (``convert_units_to(thresh, data)``), not the data.  A call a wrapper cannot serve (callable ``op``, thresholds with unexpected dims, ``dim != "time"``) is
forwarded to the ORIGINAL function (``orig``), never approximated.  The run-length functions accept any ``dim`` of an
in-memory DataArray when no resampling is asked for (the DataArray is transposed so that ``dim`` comes first).
"""

from __future__ import annotations

import itertools
import operator

import numpy as np

from . import calendar as hcal
from . import generic as hgen
from . import run_length as hrl
from . import utils as hutl
from ._capi import Float64FieldError, get_device
from .timeaxis import TimeAxis

__all__ = ["Env", "DoyThreshold", "LazyCompare", "make_wrappers", "time_axis_of"]


class Env:
    """What the wrappers need from xarray / xclim.  ``convert_units_to(threshold, data, context=None)`` -> float or
    DataArray in the units of ``data``; ``to_agg_units(out, orig, op, dim="time")`` -> ``out`` with the units of the
    aggregation; ``finish_select_resample_op(out, da, op, out_units)`` = the tail of gen:118-125."""

    def __init__(self, DataArray, convert_units_to, to_agg_units, finish_select_resample_op=None,
                 build_climatology_bounds=None, difference_attrs=None):
        self.difference_attrs = difference_attrs  # units string -> CF attrs of a DIFFERENCE in these units (gen:1550)
        self.DataArray = DataArray
        self.convert_units_to = convert_units_to
        self.to_agg_units = to_agg_units
        self.finish_select_resample_op = finish_select_resample_op or (lambda out, da, op, out_units: out)
        self.build_climatology_bounds = build_climatology_bounds


def time_axis_of(da) -> TimeAxis:
    t = da["time"].dt
    return TimeAxis(np.asarray(t.year.values), np.asarray(t.month.values), np.asarray(t.day.values), str(t.calendar))


def is_chunked(da) -> bool:
    """dask-backed (the reference's chunked path: core/calendar.py:460-479 ``dask="parallelized"``, indices/helpers.py:898-974,
    core/indicator.py:865-944)?  Such a field is NEVER pulled into host memory in one piece (``.values``): the wrappers
    walk its chunk grid over the cell dimensions, one block with the whole time axis at a time."""
    d = getattr(da, "data", None)
    return d is not None and hasattr(d, "dask") and hasattr(d, "chunks")
    # NOTE on laziness: the wrappers EVALUATE a chunked input when they are called — one ``.values`` per cell block (each
    # re-runs the upstream dask graph of that block) — and return numpy-backed results, where the reference returns lazy
    # dask arrays that are computed later.  The numbers are the same; the evaluation time moves to the call.


def _tfirst(da):
    """The DataArray with time first and its values as a C-contiguous array in the array's OWN dtype — ``None`` for a
    chunked (dask-backed) field, whose values only ever exist block by block (:func:`chunk_index`, :func:`block_values`)."""
    a = da.transpose("time", ...)
    if is_chunked(a):
        return a, None
    return a, np.ascontiguousarray(a.values)


def chunk_index(a):
    """Index tuples (one slice per cell dimension) of the dask chunk grid of a time-first DataArray.  The time axis is
    taken whole whatever its chunking (the reference re-chunks to ``time: -1`` itself, cal:463-467)."""
    spans = []
    for dim, c in zip(a.dims, a.data.chunks):
        if dim == "time":
            continue
        edges = np.concatenate([[0], np.cumsum(c)])
        spans.append([slice(int(e0), int(e1)) for e0, e1 in zip(edges[:-1], edges[1:])])
    return list(itertools.product(*spans))


def block_values(a, idx):
    """The values of one cell block (whole time axis) of a time-first DataArray, C-contiguous, own dtype: the only part of
    a chunked field that is on the host at any time."""
    sub = a.isel({d: sl for d, sl in zip(_cell_dims(a), idx)})
    return np.ascontiguousarray(sub.values)


def reduce_blocks(a, x, compute):
    """``compute(x_block, idx) -> array (R, *block cells)`` or a tuple of such.  In-memory field: one call with ``idx =
    None``; chunked field: one call per block of :func:`chunk_index`, results stitched into ``(R, *cells)`` arrays (the
    role ``xr.map_blocks`` / ``apply_ufunc(dask="parallelized")`` play in the reference; every op on the path is
    independent per cell)."""
    if x is not None:
        return compute(x, None)
    cells = tuple(a.shape[1:])
    outs, was_tuple = None, False
    grid = chunk_index(a)
    if not grid or 0 in cells:
        # a cell dimension of length zero: nothing to walk — the shapes come from one call on an empty block
        r = compute(np.empty((a.shape[0],) + cells, dtype=a.dtype), None)
        return r
    for idx in grid:
        r = compute(block_values(a, idx), idx)
        was_tuple = isinstance(r, tuple)
        rt = r if was_tuple else (r,)
        if outs is None:
            outs = [None if o is None else np.empty((np.shape(o)[0],) + cells, np.asarray(o).dtype) for o in rt]
        for o, dst in zip(rt, outs):
            if dst is not None:
                dst[(slice(None),) + tuple(idx)] = np.asarray(o)
    return tuple(outs) if was_tuple else outs[0]


class ChunkedThreshold:
    """A full-shape (time, *cells) threshold that is itself dask-backed: like the field, it only ever reaches the host block
    by block (:func:`threshold_block`)."""

    def __init__(self, da):
        self.da = da  # time first, dims of the field


def threshold_block(thr, idx, a):
    """The part of a threshold that belongs to a cell block: scalars as they are, per-cell and full arrays sliced,
    per-doy tables as the block's own slab (``DoyPercentile.block``)."""
    if isinstance(thr, ChunkedThreshold):
        return np.ascontiguousarray(thr.da.values) if idx is None else block_values(thr.da, idx)
    if idx is None or isinstance(thr, (int, float)) or (np.ndim(thr) == 0 and not isinstance(thr, hcal.DoyPercentile)):
        return thr
    if isinstance(thr, hcal.DoyPercentile):
        return thr.block(idx)
    thr = np.asarray(thr)
    if thr.ndim == len(_cell_dims(a)):
        return np.ascontiguousarray(thr[tuple(idx)])
    return np.ascontiguousarray(thr[(slice(None),) + tuple(idx)])


def _cell_dims(a):
    return tuple(d for d in a.dims if d != "time")


def _cell_coords(a):
    return {k: v for k, v in a.coords.items() if "time" not in getattr(v, "dims", ())}


class _LazyBase:
    """Proxy: anything but the wrappers sees the materialised DataArray."""

    _da = None

    def materialize(self):
        raise NotImplementedError

    def _get(self):
        if self._da is None:
            self._da = self.materialize()
        return self._da

    def __getattr__(self, name):  # only reached for attributes the proxy does not define itself
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return getattr(self._get(), name)

    def __getitem__(self, key):
        return self._get()[key]

    def __array__(self, dtype=None, copy=None):
        v = np.asarray(self._get().values)
        return v.astype(dtype) if dtype is not None else v


def _delegate(opname):
    fn = getattr(operator, opname)

    def method(self, other):
        return fn(self._get(), other)

    def rmethod(self, other):
        return fn(other, self._get())

    return method, rmethod


for _op, _dunder in (("add", "add"), ("sub", "sub"), ("mul", "mul"), ("truediv", "truediv"), ("gt", "gt"), ("lt", "lt"),
                     ("ge", "ge"), ("le", "le"), ("eq", "eq"), ("ne", "ne"), ("and_", "and"), ("or_", "or")):
    _m, _r = _delegate(_op)
    setattr(_LazyBase, f"__{_dunder}__", _m)
    if _dunder in ("add", "sub", "mul", "truediv", "and", "or"):
        setattr(_LazyBase, f"__r{_dunder}__", _r)
_LazyBase.__hash__ = object.__hash__


class DoyThreshold(_LazyBase):
    """Result of the patched ``resample_doy(doy, arr)`` (cal:763-790): the per-doy table (device, float64) + the time
    axis and coordinates of ``arr``.  ``threshold_count`` / ``compare`` consume it without the (T, Y, X) float64 field."""

    def __init__(self, env, doy: hcal.DoyPercentile, like, time: TimeAxis, device):
        self._env, self.doy, self.like, self.time, self._dev = env, doy, like, time, device

    def materialize(self):
        a = self.like.transpose("time", ...)
        data = hcal.resample_doy(self.doy, self.time, device=self._dev)
        coords = dict(_cell_coords(a))
        coords["time"] = a["time"]
        return self._env.DataArray(data, coords=coords, dims=a.dims, attrs=dict(self.doy.attrs))


class LazyCompare(_LazyBase):
    """``compare(da, op, DoyThreshold)``: the mask is only formed when something other than ``resample_and_rl`` /
    ``threshold_count`` asks for it."""

    def __init__(self, env, da, op, thr: DoyThreshold, constrain, device):
        self._env, self.da, self.op, self.thr, self.constrain, self._dev = env, da, op, thr, constrain, device

    def materialize(self):
        from . import kernels as K

        a, x = _tfirst(self.da)
        dev = self._dev

        def one(xb, idx):
            doy = hcal.adjust_doy_calendar(threshold_block(self.thr.doy, idx, a), self.thr.time, dev)
            xd, _ = hcal._flatten(xb, dev)
            table = doy.data.reshape(doy.data.shape[1], doy.data.shape[2])
            # x[t] OP table[dayofyear(t)] compared in float64 like numpy does for float32 data against a float64 threshold
            m = K.compare_doy(dev, xd, hgen.get_op(self.op, self.constrain), table, hcal.resample_doy_index(doy, self.thr.time))
            return m.get().reshape(xb.shape) != 0

        # (a chunked field: the mask is assembled block by block — whoever touches a LazyCompare asked for the full mask)
        mask = reduce_blocks(a, x, one)
        coords = dict(_cell_coords(a))
        coords["time"] = a["time"]
        return self._env.DataArray(mask, coords=coords, dims=a.dims)


def make_wrappers(env: Env, orig: dict | None = None, device=None) -> dict:
    """name -> callable.  ``orig``: the reference's own functions by the same names (fallback for calls the HIP path does
    not serve; absent in the tests, where such calls raise NotImplementedError)."""
    orig = {} if orig is None else orig  # (the SAME dict: install() adds entries after the wrappers exist)
    DA = env.DataArray

    def dev():
        return device or get_device()

    def fallback(name, *args, **kwargs):
        if name in orig:
            return orig[name](*args, **kwargs)
        raise NotImplementedError(f"{name}: this call is not served by the HIP path and no original function was supplied")

    # ---- Indicator-level fusion (core/indicator.py:1522-1549 -> core/missing.py:253-298) -------------
    # Every reducer returns the per-period count of valid (non-NaN) steps as a side output of the SAME kernel launch.  It
    # is remembered per input buffer, so that the missing-value check the Indicator runs right after the compute
    # (MissingAny()(da, freq, src_timestep, **indexer) on the same DataArray) costs no second pass over the data.
    import weakref

    valid_cache = {}
    valid_hooked = []  # devices whose forget_inputs() also clears this cache

    def _buffer_key(x, freq, indexer):
        ai = x.__array_interface__
        idx = tuple(sorted((k, repr(v)) for k, v in indexer.items() if v is not None)) if indexer else ()
        return (ai["data"][0], x.shape, x.strides, x.dtype.str, freq, idx)

    def _fingerprint(x):
        """A guard against a buffer that was modified in place between the reducer and the missing-value check
        (``da.values[t] = nan``): the bit patterns of evenly spaced samples over ALL rows (NaN payloads included; every
        element of fields up to 2^16 elements — ``Device._host_fingerprint``) plus the NaN count of the first, a middle and
        the last time step.  (Round 4 compared the three NaN counts only: an edit of any other step went unnoticed.)"""
        from ._capi import Device

        T = x.shape[0]
        if not T or x.dtype.kind != "f" or x.dtype.itemsize not in (4, 8) or not x.flags.c_contiguous:
            return ()
        return Device._host_fingerprint(x.reshape(-1)) + tuple(int(np.isnan(x[t]).sum()) for t in sorted({0, T // 2, T - 1}))

    def plain_indexer(indexer):
        """Indexers whose values are plain python / numpy values.  DataArray-valued ``doy_bounds`` (per cell, or with a time
        dimension: cal:1199-1246, missing.py:140-146) carry their own dimension order: such calls go to the reference."""
        return not any(isinstance(v, (DA, _LazyBase)) or (isinstance(v, (tuple, list)) and any(isinstance(e, (DA, _LazyBase)) for e in v))
                       for v in (indexer or {}).values())

    def remember_valid(da, x, freq, valid, indexer=None):
        """`x`: the C-contiguous time-first values the kernel read.  Only buffers that ARE the DataArray's own storage can be
        recognised again (a transposed / non-contiguous input was copied by _tfirst: nothing to remember)."""
        d = dev()
        if not d._inputs_active():  # only inside a keep_inputs scope (Indicator.__call__ opens one): same contract as the input copies
            return
        if d not in valid_hooked:
            valid_hooked.append(d)
            d._forget_hooks.append(valid_cache.clear)
        base = da.values if isinstance(da, DA) else None
        if base is None or not isinstance(base, np.ndarray) or base.__array_interface__["data"][0] != x.__array_interface__["data"][0]:
            return
        owner = base if base.base is None else base.base
        try:
            ref = weakref.ref(owner)
        except TypeError:
            return
        if len(valid_cache) > 64:
            valid_cache.clear()
        valid_cache[_buffer_key(x, freq, indexer)] = (ref, np.asarray(valid), _fingerprint(x))

    def recall_valid(x, freq, indexer):
        """ONE-SHOT: the Indicator runs the missing-value check right after the compute (core/indicator.py:1522-1549); an
        entry that was used, or whose buffer no longer looks the same, is gone."""
        hit = valid_cache.pop(_buffer_key(x, freq, indexer), None)
        if hit is None or hit[0]() is None or hit[2] != _fingerprint(x):
            return None
        return hit[1]

    # ---- per-doy tables stay on the device between percentile_doy and the index that consumes them ------------------------
    # percentile_doy must hand back a real DataArray (anything may touch it), so its table is downloaded once — but the
    # device copy is remembered under the host array the DataArray holds.  ``resample_doy(per.sel(percentiles=90), tasmax)``
    # (what tx90p / tn10p / the spell-duration indices do with it, indices/_multivariate.py) receives a VIEW of that array:
    # same owner object, the offset and strides of one percentile -> the table is taken from the device instead of being
    # transposed and uploaded again (3 GB for a 1440 x 720 grid).  Guarded like the input cache: weak reference to the
    # owner + the bit patterns of three day-of-year rows.
    table_cache = {}
    hooked = []

    def _owner_of(v):
        while isinstance(getattr(v, "base", None), np.ndarray):
            v = v.base
        return v

    def _rows_fingerprint(tv):
        """Bit patterns of up to 32 evenly spaced day-of-year rows (three rows let an edit of any other row
        through), each sampled at up to 2^14 evenly spaced cells: ~1 ms on the 3 GB table of a 1440 x 720 grid."""
        if tv.dtype != np.float64 or not tv.size:
            return None
        n = tv.shape[0]
        rows = tv[np.unique(np.linspace(0, n - 1, min(n, 32)).astype(np.intp))].reshape(-1, int(np.prod(tv.shape[1:], dtype=np.int64)))
        rows = np.ascontiguousarray(rows[:, ::max(1, rows.shape[1] >> 14)])
        return int(rows.view(np.uint64).sum(dtype=np.uint64)), int(np.isnan(rows).sum())

    def remember_table(vals, p):
        """vals: the host array (ndoy, *cells, nper) the returned DataArray is built on; p: the device DoyPercentile.
        Like the input copies (Device.resident) the device table is only remembered inside a ``keep_inputs`` scope and dies
        with it — or with the host array, whichever comes first (the entry holds NO reference to the host data: address,
        shape and strides of each percentile's slice only)."""
        d = dev()
        if not d._inputs_active():
            return
        owner = _owner_of(vals)
        try:
            ref = weakref.ref(owner)
        except TypeError:
            return
        if d not in hooked:
            hooked.append(d)
            d._forget_hooks.append(table_cache.clear)
        if len(table_cache) > 8:
            table_cache.clear()
        slices = []
        for j in range(vals.shape[-1]):
            v = vals[..., j]
            slices.append((v.ctypes.data, v.shape, v.strides, _rows_fingerprint(v)))
        table_cache[id(owner)] = (ref, slices, p)
        weakref.finalize(owner, table_cache.pop, id(owner), None)

    def recall_table(tv):
        """tv: (ndoy, *cells) float64 view.  -> device DoyPercentile of that one percentile, or None."""
        owner = _owner_of(tv)
        hit = table_cache.get(id(owner))
        if hit is None or hit[0]() is not owner or not dev()._inputs_active():
            return None
        _, slices, p = hit
        for j, (addr, shape, strides, fp) in enumerate(slices):
            if shape == tv.shape and strides == tv.strides and addr == tv.ctypes.data and fp is not None and fp == _rows_fingerprint(tv):
                return p.sel(p.percentiles[j])
        return None

    def wrap_periods(a, data, freq, attrs=None, name=None):
        """(P, *cells) -> DataArray(time = period labels of the reference's own resample, *cell dims)."""
        labels = a["time"].resample(time=freq).first()["time"]
        coords = dict(_cell_coords(a))
        coords["time"] = labels
        data = np.asarray(data)
        return DA(data.reshape((len(labels.values),) + data.shape[1:]), coords=coords, dims=("time",) + _cell_dims(a),
                  attrs=dict(attrs or {}), name=name)

    def wrap_cells(a, data, attrs=None, name=None):
        return DA(np.asarray(data), coords=_cell_coords(a), dims=_cell_dims(a), attrs=dict(attrs or {}), name=name)

    def wrap_full(a, data, attrs=None, name=None):
        coords = dict(_cell_coords(a))
        coords["time"] = a["time"]
        return DA(np.asarray(data), coords=coords, dims=a.dims, attrs=dict(attrs or {}), name=name)

    def as_threshold(thr, a):
        """float | per-cell array (cell dims of `a`) | full array (dims of `a`) | DoyPercentile; None = not servable."""
        if isinstance(thr, DoyThreshold):
            return thr.doy
        if isinstance(thr, _LazyBase):
            thr = thr._get()
        if isinstance(thr, DA):
            if set(thr.dims) == set(a.dims):
                if is_chunked(thr):  # a chunked (time, *cells) threshold is sliced per block, never materialised whole
                    return ChunkedThreshold(thr.transpose(*a.dims))
                return np.ascontiguousarray(thr.transpose(*a.dims).values)
            if set(thr.dims) == set(_cell_dims(a)):
                return np.ascontiguousarray(thr.transpose(*_cell_dims(a)).values)
            if len(thr.dims) == 0:
                return thr.values[()]
            return None
        if np.ndim(thr) == 0:
            return thr
        return None

    def plain_scalar(thr, x):
        """True when `thr` compares with the field `x` in x's own dtype under NumPy's promotion rules: a python number (weak
        scalar) or a numpy scalar that casts safely to x.dtype.  A 0-d float64 DataArray against a float32 field is a
        float64 compare in the reference (0-d arrays are not weak): the kernels that only compare in the field's dtype leave
        such calls to the original function."""
        if np.ndim(thr) != 0:
            return False
        if isinstance(thr, np.generic):
            return np.can_cast(thr.dtype, x.dtype, "safe")  # (x: the values or the DataArray — only its dtype is used)
        return isinstance(thr, (int, float))

    # ---- indices/generic.py ---------------------------------------------------------------------------------------
    def threshold_count(da, op, threshold, freq, constrain=None):  # gen:329-361
        if isinstance(threshold, LazyCompare) or not isinstance(da, DA):
            return fallback("threshold_count", da, op, threshold, freq, constrain)
        a, x = _tfirst(da)
        thr = as_threshold(threshold, a)
        if thr is None:
            return fallback("threshold_count", da, op, threshold, freq, constrain)
        t = time_axis_of(a)
        out, valid = reduce_blocks(a, x, lambda xb, idx: hgen.threshold_count(xb, op, threshold_block(thr, idx, a), t, freq, constrain,
                                                                              device=dev(), with_valid=True))
        if x is not None:
            remember_valid(a, x, freq, valid)
        return wrap_periods(a, np.asarray(out).astype(np.int64), freq)  # (bool * 1).resample.sum: int64, no attrs

    def count_occurrences(data, threshold, freq, op, constrain=None):  # gen:960-999
        a, x = _tfirst(data)
        thr = as_threshold(env.convert_units_to(threshold, data), a)
        if thr is None:
            return fallback("count_occurrences", data, threshold, freq, op, constrain)
        t = time_axis_of(a)
        out, valid = reduce_blocks(a, x, lambda xb, idx: hgen.count_occurrences(xb, threshold_block(thr, idx, a), op, t, freq, constrain,
                                                                                device=dev(), with_valid=True))
        if x is not None:
            remember_valid(a, x, freq, valid)
        return env.to_agg_units(wrap_periods(a, np.asarray(out).astype(np.int64), freq, data.attrs), data, "count", dim="time")

    def domain_count(da, low, high, freq):  # gen:364-392
        if np.ndim(low) != 0 or np.ndim(high) != 0 or isinstance(low, DA) or isinstance(high, DA):
            return fallback("domain_count", da, low, high, freq)
        a, x = _tfirst(da)
        if not plain_scalar(low, a) or not plain_scalar(high, a):
            return fallback("domain_count", da, low, high, freq)
        t = time_axis_of(a)
        out = reduce_blocks(a, x, lambda xb, idx: hgen.domain_count(xb, low, high, t, freq, device=dev()))
        return wrap_periods(a, np.asarray(out).astype(np.int64), freq)

    def select_resample_op(da, op, freq="YS", out_units=None, **indexer):  # gen:83-125
        # callables and the names gen:221 maps to callables (doymin / doymax: resample_map of a DataArray function) stay with
        # the reference
        if (not isinstance(op, str) or op not in ("min", "max", "mean", "std", "var", "count", "sum", "integral", "argmax", "argmin")
                or not plain_indexer(indexer)):
            return fallback("select_resample_op", da, op, freq, out_units, **indexer)
        a, x = _tfirst(da)
        t = time_axis_of(a)
        out, valid = reduce_blocks(a, x, lambda xb, idx: hgen.select_resample_op(xb, op, t, freq, device=dev(), with_valid=True, **indexer))
        if x is not None:
            remember_valid(a, x, freq, valid, indexer)
        o = wrap_periods(a, out, freq, da.attrs, da.name)
        return env.finish_select_resample_op(o, da, op, out_units)

    def spell_length_statistics(data, threshold, window, win_reducer, op, spell_reducer, freq, min_gap=1,
                                resample_before_rl=True, **indexer):  # gen:588-686 -> 543-585
        if not isinstance(data, DA):
            return fallback("spell_length_statistics", data, threshold, window, win_reducer, op, spell_reducer, freq,
                            min_gap=min_gap, resample_before_rl=resample_before_rl, **indexer)
        a, x = _tfirst(data)
        thr = as_threshold(env.convert_units_to(threshold, data, context="infer"), a)
        if (thr is None or isinstance(thr, hcal.DoyPercentile) or (np.ndim(thr) == 0 and not plain_scalar(thr, a))
                or not plain_indexer(indexer)):
            return fallback("spell_length_statistics", data, threshold, window, win_reducer, op, spell_reducer, freq,
                            min_gap=min_gap, resample_before_rl=resample_before_rl, **indexer)
        reducers = [spell_reducer] if isinstance(spell_reducer, str) else list(spell_reducer)
        outs = []
        t = time_axis_of(a)
        for sr in reducers:
            o, valid = reduce_blocks(a, x, lambda xb, idx: hgen.spell_length_statistics(
                xb, threshold_block(thr, idx, a), window, win_reducer, op, sr, t, freq, min_gap, resample_before_rl, device=dev(),
                with_valid=True, **indexer))
            if not indexer and x is not None:  # (with an indexer the count is of the unselected data: not what MissingAny wants)
                remember_valid(a, x, freq, valid)
            w = wrap_periods(a, o, freq, data.attrs)
            if sr == "count":
                w.attrs["units"] = ""
                outs.append(w)
            else:
                outs.append(env.to_agg_units(w, data, "count"))
        return outs[0] if len(outs) == 1 else tuple(outs)

    def cumulative_difference(data, threshold, op, freq=None):  # gen:1514-1552
        if freq is None:
            return fallback("cumulative_difference", data, threshold, op, freq)
        a, x = _tfirst(data)
        thr = as_threshold(env.convert_units_to(threshold, data), a)
        if thr is None or not plain_scalar(thr, a):
            return fallback("cumulative_difference", data, threshold, op, freq)
        t = time_axis_of(a)
        out = wrap_periods(a, reduce_blocks(a, x, lambda xb, idx: hgen.cumulative_difference(xb, float(thr), op, t, freq, device=dev())),
                           freq, data.attrs)
        if env.difference_attrs is not None and "units" in data.attrs:  # a sum of differences: delta units (gen:1550)
            out.attrs.update(env.difference_attrs(data.attrs["units"]))
        return env.to_agg_units(out, data, op="integral")

    def bivariate_count_occurrences(*, data_var1, data_var2, threshold_var1, threshold_var2, freq, op_var1, op_var2,
                                    var_reducer, constrain_var1=None, constrain_var2=None):  # gen:1003-1073
        kw = dict(data_var1=data_var1, data_var2=data_var2, threshold_var1=threshold_var1, threshold_var2=threshold_var2,
                  freq=freq, op_var1=op_var1, op_var2=op_var2, var_reducer=var_reducer, constrain_var1=constrain_var1,
                  constrain_var2=constrain_var2)
        if not (isinstance(data_var1, DA) and isinstance(data_var2, DA)) or tuple(data_var1.dims) != tuple(data_var2.dims):
            return fallback("bivariate_count_occurrences", **kw)
        a, x1 = _tfirst(data_var1)
        b, x2 = _tfirst(data_var2)
        t1 = as_threshold(env.convert_units_to(threshold_var1, data_var1), a)
        t2 = as_threshold(env.convert_units_to(threshold_var2, data_var2), b)
        if (t1 is None or t2 is None or not plain_scalar(t1, a) or not plain_scalar(t2, b) or tuple(a.shape) != tuple(b.shape)
                or (x1 is None) != (x2 is None)):
            return fallback("bivariate_count_occurrences", **kw)
        t = time_axis_of(a)
        out = reduce_blocks(a, x1, lambda xb, idx: hgen.bivariate_count_occurrences(
            data_var1=xb, data_var2=x2 if idx is None else block_values(b, idx), threshold_var1=float(t1), threshold_var2=float(t2),
            time=t, freq=freq, op_var1=op_var1, op_var2=op_var2, var_reducer=var_reducer, constrain_var1=constrain_var1,
            constrain_var2=constrain_var2, device=dev()))
        return env.to_agg_units(wrap_periods(a, np.asarray(out).astype(np.int64), freq, data_var1.attrs), data_var1, "count", dim="time")

    def season(data, thresh, window, op, stat, freq, mid_date=None, constrain=None):  # gen:770-853
        a, x = _tfirst(data)
        thr = as_threshold(env.convert_units_to(thresh, data, context="infer"), a)
        if thr is None or not plain_scalar(thr, a) or stat not in ("start", "end", "length"):
            return fallback("season", data, thresh, window, op, stat, freq, mid_date=mid_date, constrain=constrain)
        hgen.get_op(op, constrain)
        t = time_axis_of(a)
        res = reduce_blocks(a, x, lambda xb, idx: hgen.season(xb, float(thr), window, op, t, freq, mid_date, device=dev())[stat])
        out = wrap_periods(a, res, freq, data.attrs)
        if stat == "length":
            return env.to_agg_units(out, data, "count")
        out.attrs.update(units="", is_dayofyear=np.int32(1), calendar=str(a["time"].dt.calendar))
        return out

    def first_day_threshold_reached(data, *, threshold, op, after_date, window=1, freq="YS", constrain=None):  # gen:1556-1608
        a, x = _tfirst(data)
        thr = as_threshold(env.convert_units_to(threshold, data), a)
        if thr is None or not plain_scalar(thr, a):
            return fallback("first_day_threshold_reached", data, threshold=threshold, op=op, after_date=after_date, window=window,
                            freq=freq, constrain=constrain)
        t = time_axis_of(a)
        res = reduce_blocks(a, x, lambda xb, idx: hgen.first_day_threshold_reached(
            xb, threshold=float(thr), op=op, after_date=after_date, time=t, window=window, freq=freq, constrain=constrain, device=dev()))
        out = wrap_periods(a, res, freq, data.attrs)
        out.attrs.update(units="", is_dayofyear=np.int32(1), calendar=str(a["time"].dt.calendar))
        return out

    def compare(left, op, right, constrain=None):  # gen:301-326
        if isinstance(right, DoyThreshold) and isinstance(left, DA):
            hgen.get_op(op, constrain)  # the reference's ValueError for an unknown / constrained operator comes first
            if left.dtype != np.float32:
                # the fused kernels behind a LazyCompare read float32 fields; a float64 (or integer) field is compared by
                # the reference itself against the materialised (time, ...) threshold — forwarded, never refused later
                return fallback("compare", left, op, right._get(), constrain)
            return LazyCompare(env, left, op, right, constrain, dev())
        if isinstance(left, _LazyBase):
            left = left._get()
        if isinstance(right, _LazyBase):
            right = right._get()
        if not isinstance(left, DA) or is_chunked(left):  # (a chunked full-shape mask is the reference's own dask business)
            return fallback("compare", left, op, right, constrain)
        a, x = _tfirst(left) if "time" in left.dims else (left, np.ascontiguousarray(left.values))
        thr = as_threshold(right, a) if "time" in left.dims else (right if np.ndim(right) == 0 and not isinstance(right, DA) else None)
        if thr is None:
            return fallback("compare", left, op, right, constrain)
        if isinstance(thr, ChunkedThreshold):  # in-memory field against a chunked full-shape threshold: the mask is full-shape anyway
            thr = threshold_block(thr, None, a)
        if np.ndim(thr) == x.ndim - 1 and np.ndim(thr) > 0:
            thr = np.broadcast_to(thr, x.shape)
        data = hgen.compare(x if x.ndim > 1 else x[:, None], op, thr if np.ndim(thr) == 0 or x.ndim > 1 else thr[:, None],
                            constrain, device=dev())
        data = np.asarray(data).astype(bool).reshape(x.shape)
        return DA(data, coords=dict(a.coords), dims=a.dims)

    # ---- core/calendar.py -----------------------------------------------------------------------------------------
    def percentile_doy(arr, window=5, per=10.0, alpha=1.0 / 3.0, beta=1.0 / 3.0, copy=True):  # cal:395-494
        pers = [per] if np.isscalar(per) else list(per)
        a, x = _tfirst(arr)
        t = time_axis_of(a)
        meta = {}

        def one(xb, idx):
            p = hcal.percentile_doy(xb, t, window=window, per=pers, alpha=alpha, beta=beta, device=dev())
            meta.setdefault("p", p)
            return p.values()  # (ndoy, *cells, nper)

        if x is not None:
            vals = one(x, None)
            remember_table(vals, meta["p"])
        else:  # chunked: block by block (the reference: apply_ufunc(dask="parallelized") after time: -1, cal:460-479)
            vals = None
            for idx in chunk_index(a):
                v = one(block_values(a, idx), idx)
                if vals is None:
                    vals = np.empty((v.shape[0],) + tuple(a.shape[1:]) + (v.shape[-1],), v.dtype)
                vals[(slice(None),) + tuple(idx)] = v
        p = meta["p"]
        coords = dict(_cell_coords(a))
        coords.update(dayofyear=np.asarray(p.dayofyear), percentiles=np.asarray(pers, dtype=np.float64))
        out = DA(vals, coords=coords, dims=("dayofyear",) + _cell_dims(a) + ("percentiles",), attrs=dict(arr.attrs),
                 name="per")
        # the reference's order: unstack("time") appends (year, dayofyear) after the cell dimensions, apply_ufunc appends
        # the percentile dimension (cal:448-480) -> (*cells, dayofyear, percentiles)
        out = out.transpose(*(_cell_dims(a) + ("dayofyear", "percentiles")))
        bounds = env.build_climatology_bounds(arr) if env.build_climatology_bounds else p.attrs["climatology_bounds"]
        out.attrs.update(climatology_bounds=bounds, window=window, alpha=alpha, beta=beta)
        return out


    def resample_doy(doy, arr):  # cal:763-790
        if not isinstance(doy, DA) or "dayofyear" not in doy.dims or "percentiles" in doy.dims or "time" not in arr.dims:
            return fallback("resample_doy", doy, arr)
        a = arr.transpose("time", ...)
        cd = _cell_dims(a)
        if set(doy.dims) != {"dayofyear", *cd}:
            return fallback("resample_doy", doy, arr)
        tv = doy.transpose("dayofyear", *cd).values
        d = dev()
        known = recall_table(tv) if isinstance(tv, np.ndarray) and tv.dtype == np.float64 else None
        if known is not None:
            # the table of an earlier percentile_doy call, still on the device; its host form (a copy when the view is not
            # contiguous) is only made if a chunked field asks for slabs
            dp = hcal.DoyPercentile(known.data, np.asarray(doy["dayofyear"].values), [np.nan], tv.shape[1:], dict(doy.attrs),
                                    host=lambda: np.ascontiguousarray(tv, dtype=np.float64)[None], device=d)
            return DoyThreshold(env, dp, arr, time_axis_of(a), d)
        table = np.ascontiguousarray(tv, dtype=np.float64)
        # the table stays on the host until a kernel wants it: whole (in-memory field) or slab by slab (chunked field)
        dp = hcal.DoyPercentile(None, np.asarray(doy["dayofyear"].values), [np.nan], table.shape[1:], dict(doy.attrs),
                                host=table[None], device=d)
        return DoyThreshold(env, dp, arr, time_axis_of(a), d)

    # ---- indices/run_length.py ------------------------------------------------------------------------------------
    def _mask_values(da):
        """(a, values, time axis) of a mask-like DataArray / LazyCompare; LazyCompare stays lazy for resample_and_rl only."""
        if isinstance(da, _LazyBase):
            da = da._get()
        a, x = _tfirst(da)
        if x is not None and x.dtype == bool:
            x = x.astype(np.float32)
        return a, x, time_axis_of(a)

    def mask_block(xb):
        return xb.astype(np.float32) if xb.dtype == bool else xb

    def rl_blocks(a, x, fn):
        """fn(mask values) -> (P, *cells) or (*cells) [freq=None]; chunked masks block by block"""
        if x is not None:
            return fn(x)
        out = None
        for idx in chunk_index(a):
            r = np.asarray(fn(mask_block(block_values(a, idx))))
            lead = r.ndim - len(idx)  # 1 with a period axis, 0 for one value per cell
            if out is None:
                out = np.empty(r.shape[:lead] + tuple(a.shape[1:]), r.dtype)
            out[(slice(None),) * lead + tuple(idx)] = r
        return out

    def _rl_out(a, out, freq, name=None):
        return wrap_cells(a, out, name=name) if freq is None else wrap_periods(a, out, freq, name=name)

    # ---- run lengths along another dimension (rl:223, 275, 338: ``dim`` is any dimension of the DataArray) --------------------
    # Without a resampling frequency nothing in these functions is about TIME: the DataArray is transposed so that `dim`
    # comes first and the same kernels march along it (the numpy mirrors take the axis as ``dim=0``).  What stays with the
    # reference: a resampling frequency on another dimension, chunked inputs, ``coord=<datetime accessor name>``.
    def _along(da, dim):
        """(DataArray with `dim` first, its values as a float mask) — or None when the call belongs to the reference"""
        if isinstance(da, _LazyBase):
            da = da._get()
        if not isinstance(da, DA) or dim not in da.dims or is_chunked(da):
            return None
        a = da.transpose(dim, ...)
        x = np.ascontiguousarray(a.values)
        return a, (x.astype(np.float32) if x.dtype == bool else x)

    def _wrap_others(a, dim, data, name=None):
        """(*other dims) result: every coordinate that does not run along `dim` stays (xarray's reductions keep them)"""
        coords = {k: v for k, v in a.coords.items() if dim not in getattr(v, "dims", ())}
        return DA(np.asarray(data), coords=coords, dims=tuple(d for d in a.dims if d != dim), name=name)

    def _rl_other_dim(name, host_call, da, dim, freq, *fb_args):
        got = _along(da, dim) if freq is None else None
        if got is None:
            return fallback(name, *fb_args)
        a, x = got
        return _wrap_others(a, dim, host_call(x))

    def rle(da, dim="time", index="first"):  # rl:223-272
        if isinstance(da, DA) and is_chunked(da):  # (a chunked full-shape result: the reference's dask path)
            return fallback("rle", da, dim, index)
        if dim != "time":
            got = _along(da, dim)
            if got is None:
                return fallback("rle", da, dim, index)
            a, x = got
            return DA(hrl.rle(x, 0, index, device=dev()), coords=dict(a.coords), dims=a.dims, attrs=dict(da.attrs))
        a, x, _ = _mask_values(da)
        return wrap_full(a, hrl.rle(x, dim, index, device=dev()), da.attrs if isinstance(da, DA) else None)

    def rle_statistics(da, reducer, window, dim="time", freq=None, ufunc_1dim="from_context", index="first"):  # rl:275-335
        if dim != "time":
            return _rl_other_dim("rle_statistics", lambda x: hrl.rle_statistics(x, reducer, window, 0, None, ufunc_1dim, index, device=dev()),
                                 da, dim, freq, da, reducer, window, dim, freq, ufunc_1dim, index)
        a, x, t = _mask_values(da)
        return _rl_out(a, rl_blocks(a, x, lambda xb: hrl.rle_statistics(xb, reducer, window, dim, freq, ufunc_1dim, index, time=t, device=dev())), freq)

    def longest_run(da, dim="time", freq=None, ufunc_1dim="from_context", index="first"):  # rl:338-378
        if dim != "time":
            return _rl_other_dim("longest_run", lambda x: hrl.longest_run(x, 0, None, ufunc_1dim, index, device=dev()),
                                 da, dim, freq, da, dim, freq, ufunc_1dim, index)
        a, x, t = _mask_values(da)
        return _rl_out(a, rl_blocks(a, x, lambda xb: hrl.longest_run(xb, dim, freq, ufunc_1dim, index, time=t, device=dev())), freq)

    def windowed_run_events(da, window, dim="time", freq=None, ufunc_1dim="from_context", index="first"):  # rl:381-434
        if dim != "time":
            return _rl_other_dim("windowed_run_events", lambda x: hrl.windowed_run_events(x, window, 0, None, ufunc_1dim, index, device=dev()),
                                 da, dim, freq, da, window, dim, freq, ufunc_1dim, index)
        a, x, t = _mask_values(da)
        return _rl_out(a, rl_blocks(a, x, lambda xb: hrl.windowed_run_events(xb, window, dim, freq, ufunc_1dim, index, time=t, device=dev())), freq)

    def windowed_run_count(da, window, dim="time", freq=None, ufunc_1dim="from_context", index="first"):  # rl:437-488
        if dim != "time":
            return _rl_other_dim("windowed_run_count", lambda x: hrl.windowed_run_count(x, window, 0, None, ufunc_1dim, index, device=dev()),
                                 da, dim, freq, da, window, dim, freq, ufunc_1dim, index)
        a, x, t = _mask_values(da)
        return _rl_out(a, rl_blocks(a, x, lambda xb: hrl.windowed_run_count(xb, window, dim, freq, ufunc_1dim, index, time=t, device=dev())), freq)

    def _boundary(name, host, da, window, dim, freq, coord, ufunc_1dim):
        if dim != "time":
            # the index along `dim`, or (coord=True) the coordinate value there — rl:586-596: ``lazy_indexing(da[dim], out)``
            got = _along(da, dim) if freq is None and (not coord or coord is True) else None
            if got is None or (coord is True and np.asarray(got[0][dim].values).dtype.kind not in "iuf"):
                return fallback(name, da, window, dim, freq, coord, ufunc_1dim)
            a, x = got
            idx = np.asarray(host(x, window, 0, None, None, ufunc_1dim, device=dev()))
            if coord is True:
                crd = np.asarray(a[dim].values, dtype=np.float64)
                idx = np.where(np.isnan(idx), np.nan, crd[np.nan_to_num(idx).astype(np.int64)])
            return _wrap_others(a, dim, idx)
        if coord not in (None, False, "dayofyear", "year", "month", "day"):
            return fallback(name, da, window, dim, freq, coord, ufunc_1dim)  # coord=True: datetime labels (reference path)
        a, x, t = _mask_values(da)
        return _rl_out(a, rl_blocks(a, x, lambda xb: host(xb, window, dim, freq, coord or None, ufunc_1dim, time=t, device=dev())), freq)

    def first_run(da, window, dim="time", freq=None, coord=None, ufunc_1dim="from_context"):  # rl:643-690
        return _boundary("first_run", hrl.first_run, da, window, dim, freq, coord, ufunc_1dim)

    def last_run(da, window, dim="time", freq=None, coord=None, ufunc_1dim="from_context"):  # rl:693-740
        return _boundary("last_run", hrl.last_run, da, window, dim, freq, coord, ufunc_1dim)

    def season_length(da, window, mid_date=None, dim="time"):  # rl:1113-1145
        if dim != "time":
            return fallback("season_length", da, window, mid_date, dim)
        a, x, t = _mask_values(da)
        return wrap_cells(a, rl_blocks(a, x, lambda xb: hrl.season_length(xb, window, mid_date, dim, time=t, device=dev())))

    rl_host = {}  # wrapper -> host mirror, for resample_and_rl (which receives the function OBJECT rl.<name>)
    lazy_ok = set()  # the run statistics that xh_run_stats_doy fuses with the per-doy compare

    def resample_and_rl(da, resample_before_rl, compute, *args, freq, dim="time", **kwargs):  # rl:87-132
        host = rl_host.get(compute)
        if host is None or dim != "time":
            return fallback("resample_and_rl", da, resample_before_rl, compute, *args, freq=freq, dim=dim, **kwargs)
        if isinstance(da, LazyCompare) and compute in lazy_ok and da._da is None and kwargs.get("index", "first") == "first":
            # compare(da, op, resample_doy(per, da)) -> run statistic: ONE fused launch on the per-doy table
            # (xh_run_stats_doy; the host mirror is the percentile-spell path of indices.py)
            from . import indices as hind

            a, x = _tfirst(da.da)
            t = time_axis_of(a)
            params = dict(zip({"rle_statistics": ("reducer", "window"), "longest_run": (), "windowed_run_events": ("window",),
                               "windowed_run_count": ("window",)}[host.__name__], args))
            params.update(kwargs)
            stat = {"rle_statistics": params.get("reducer"), "longest_run": "max", "windowed_run_events": "count",
                    "windowed_run_count": "sum"}[host.__name__]
            out = reduce_blocks(a, x, lambda xb, idx: hind.percentile_run_stat(
                xb, threshold_block(da.thr.doy, idx, a), da.op, stat, int(params.get("window", 1)), t, freq, resample_before_rl,
                constrain=da.constrain, device=dev()))
            return wrap_periods(a, out, freq)
        a, x, t = _mask_values(da)
        out = rl_blocks(a, x, lambda xb: hrl.resample_and_rl(xb, resample_before_rl, host, *args, freq=freq, time=t, dim=dim,
                                                              device=dev(), **kwargs))
        return wrap_periods(a, out, freq)

    lazy_ok.update((windowed_run_count, windowed_run_events, rle_statistics, longest_run))
    rl_host.update({rle_statistics: hrl.rle_statistics, longest_run: hrl.longest_run,
                    windowed_run_events: hrl.windowed_run_events, windowed_run_count: hrl.windowed_run_count,
                    first_run: hrl.first_run, last_run: hrl.last_run})

    # ---- core/missing.py --------------------------------------------------------------------------------------------------
    def missing_any_call(self, da, freq=None, src_timestep=None, **indexer):  # MissingBase.__call__ for MissingAny, :253-298, 318-322
        if freq is None or not isinstance(da, DA) or "time" not in da.dims or (src_timestep not in (None, "D", "1D")):
            return fallback("MissingAny.__call__", self, da, freq, src_timestep, **indexer)
        if not plain_indexer(indexer):
            return fallback("MissingAny.__call__", self, da, freq, src_timestep, **indexer)
        a, x = _tfirst(da)
        t = time_axis_of(a)
        if len(t) > 1 and not np.all(np.diff(t.ordinal()) == 1):  # xr.infer_freq(da.time) must be daily
            return fallback("MissingAny.__call__", self, da, freq, src_timestep, **indexer)
        idx = {k: v for k, v in indexer.items() if v is not None}
        valid = recall_valid(x, freq, idx) if x is not None else None
        if valid is None:
            try:
                valid = reduce_blocks(a, x, lambda xb, bi: hgen.select_resample_op(xb, "count", t, freq, device=dev(), **idx))
            except Float64FieldError:  # count of a float64 field with a time selection: the reference's business
                return fallback("MissingAny.__call__", self, da, freq, src_timestep, **indexer)
        expected = t.expected_count(freq, **idx)
        valid = np.asarray(valid)
        miss = valid != expected.reshape((-1,) + (1,) * (valid.ndim - 1))
        return wrap_periods(a, miss, freq)

    # the other registered methods (core/missing.py:325-512): same inputs, the host mirrors of xclim_amd/missing.py.  The options
    # live in ``self.options`` (MissingBase.__init__, :174-177).
    def _missing_call(cls_name, host, option_names):
        def call(self, da, freq=None, src_timestep=None, **indexer):
            name = f"{cls_name}.__call__"
            if (not isinstance(da, DA) or "time" not in da.dims or src_timestep not in (None, "D", "1D") or is_chunked(da)
                    or not plain_indexer(indexer) or da.dtype.kind != "f"):
                return fallback(name, self, da, freq, src_timestep, **indexer)
            a, x = _tfirst(da)
            t = time_axis_of(a)
            if len(t) < 2 or not np.all(np.diff(t.ordinal()) == 1):  # xr.infer_freq(da.time) must be daily (WMO: :425-426)
                return fallback(name, self, da, freq, src_timestep, **indexer)
            idx = {k: v for k, v in indexer.items() if v is not None}
            opts = {k: self.options[k] for k in option_names if k in self.options}
            try:
                miss = host(x, freq, t, device=dev(), **opts, **idx)
            except (NotImplementedError, Float64FieldError):   # sub-frequencies other than months, day selections of the two-step merge
                return fallback(name, self, da, freq, src_timestep, **indexer)
            if freq is None:   # the time dimension is collapsed
                return wrap_cells(a, miss[0])
            return wrap_periods(a, miss, freq)

        call.__name__ = "__call__"
        return call

    from . import missing as hmiss

    missing_calls = {
        "MissingSomeButNotAll.__call__": _missing_call("MissingSomeButNotAll", hmiss.missing_some_but_not_all, ()),
        "MissingWMO.__call__": _missing_call("MissingWMO", hmiss.missing_wmo, ("nm", "nc")),
        "MissingPct.__call__": _missing_call("MissingPct", hmiss.missing_pct, ("tolerance", "subfreq")),
        "AtLeastNValid.__call__": _missing_call("AtLeastNValid", hmiss.at_least_n_valid, ("n", "subfreq")),
    }

    # ---- apply_ufunc callees (tier 2) and xsdba ----------------------------------------------------------------------
    def calc_perc(arr, percentiles=None, alpha=1.0, beta=1.0, copy=True):  # utl:279-323
        return hutl.calc_perc(arr, percentiles, alpha, beta, copy, device=dev())

    def sdba_quantile(da, q, dim):  # xsdba.nbutils.quantile(da, q, dim): quantiles along `dim` (a name or a list of names)
        dims = [dim] if isinstance(dim, str) else list(dim)
        if dims != ["time"] or not isinstance(da, DA):
            return fallback("sdba_quantile", da, q, dim)
        from . import sdba as hsdba

        a, x = _tfirst(da)
        qs = np.atleast_1d(np.asarray(q, dtype=np.float64))
        out = reduce_blocks(a, x, lambda xb, idx: np.asarray(hsdba.quantile(xb, qs, device=dev())))  # (nq, *cells)
        coords = dict(_cell_coords(a))
        coords["quantiles"] = qs
        return DA(np.moveaxis(np.asarray(out), 0, -1), coords=coords, dims=_cell_dims(a) + ("quantiles",), attrs=dict(da.attrs),
                  name=da.name)

    def sdba_interp_on_quantiles(newx, xq, yq, *, group="time", method="linear", extrapolation="constant"):
        """xsdba.utils.interp_on_quantiles for group="time" (no sub-grouping): the 1-D interpolation of the factors `yq`
        given at the quantile values `xq`, evaluated at `newx`, per cell (scipy interp1d semantics on the non-NaN nodes,
        constant or NaN extrapolation) -> xh_eqm_adjust with kind "factor".  Sub-groupings (the 2-D interpolation over
        quantile and group) and anything unexpected go to the original."""
        kw = dict(group=group, method=method, extrapolation=extrapolation)
        prop = "group"
        gdim = "time"
        if isinstance(group, str):
            if "." in group:
                gdim, prop = group.split(".", 1)
            else:
                gdim = group
        else:
            gdim, prop = getattr(group, "dim", None), getattr(group, "prop", None)
        if (prop != "group" or gdim != "time" or method not in ("nearest", "linear", "cubic")
                or extrapolation not in ("constant", "nan") or not all(isinstance(v, DA) for v in (newx, xq, yq))):
            return fallback("sdba_interp_on_quantiles", newx, xq, yq, **kw)
        if "time" not in newx.dims or "quantiles" not in xq.dims or set(xq.dims) != set(yq.dims):
            return fallback("sdba_interp_on_quantiles", newx, xq, yq, **kw)
        if is_chunked(newx) or not (newx.dtype == xq.dtype == yq.dtype == np.float32):
            # a chunked full-shape result is the reference's dask path; float64 nodes or factors are never rounded here
            return fallback("sdba_interp_on_quantiles", newx, xq, yq, **kw)
        a, x = _tfirst(newx)
        cd = _cell_dims(a)
        if set(xq.dims) != {"quantiles", *cd}:
            return fallback("sdba_interp_on_quantiles", newx, xq, yq, **kw)
        from . import kernels as K_

        d = dev()
        hq = np.ascontiguousarray(xq.transpose("quantiles", *cd).values, dtype=np.float32)
        af = np.ascontiguousarray(yq.transpose("quantiles", *cd).values, dtype=np.float32)
        nq = hq.shape[0]
        if nq > (32 if method == "cubic" else 64):
            return fallback("sdba_interp_on_quantiles", newx, xq, yq, **kw)
        T = x.shape[0]
        out = K_.eqm_adjust(d, d.to_device(x.reshape(T, -1)), d.to_device(af.reshape(nq, -1)), d.to_device(hq.reshape(nq, -1)),
                            "factor", method, extrapolation).get().reshape(x.shape)
        coords = dict(_cell_coords(a))
        coords["time"] = a["time"]
        # xsdba: apply_ufunc(..., output_core_dims=[[dim]]) puts the core dimension LAST
        res = DA(out, coords=coords, dims=a.dims, attrs=dict(newx.attrs), name=newx.name)
        return res.transpose(*(cd + ("time",)))

    def forwarding(name, fn):
        """A float64 field on a float32-only kernel (Float64FieldError) goes to the reference's own function: the
        wrappers never round an input behind the caller's back."""
        import functools

        def restore_order(out, args, kwargs):
            """xarray's resample / groupby reductions and element-wise results keep the dimension ORDER of their input
            (GroupBy._restore_dim_order); the wrappers work time-first.  A result with the same dimensions as the first
            DataArray argument that has a time dimension is put back into that argument's order."""
            src = next((v for v in list(args) + list(kwargs.values()) if isinstance(v, DA) and "time" in v.dims), None)
            if src is None:
                return out

            def fix(o):
                if isinstance(o, DA) and set(o.dims) == set(src.dims) and tuple(o.dims) != tuple(src.dims):
                    return o.transpose(*src.dims)
                return o

            return tuple(fix(o) for o in out) if isinstance(out, tuple) else fix(out)

        keep_order = name in ("sdba_interp_on_quantiles",)  # (apply_ufunc results: core dimension last, not the input's order)

        @functools.wraps(fn)
        def wrapper(*args, **kwargs):
            try:
                out = fn(*args, **kwargs)
                return out if keep_order else restore_order(out, args, kwargs)
            except Float64FieldError:
                return fallback(name, *args, **kwargs)

        # functools.wraps leaves wrapper.__wrapped__ = fn: what bootstrap_func calls as percentile_doy.__wrapped__
        # (bootstrapping.py:195) — a plain function with the reference's signature, no loop for inspect.unwrap
        return wrapper

    table = {
        "threshold_count": threshold_count, "count_occurrences": count_occurrences, "domain_count": domain_count,
        "select_resample_op": select_resample_op, "spell_length_statistics": spell_length_statistics,
        "cumulative_difference": cumulative_difference, "compare": compare, "season": season,
        "bivariate_count_occurrences": bivariate_count_occurrences,
        "first_day_threshold_reached": first_day_threshold_reached,
        "percentile_doy": percentile_doy, "resample_doy": resample_doy,
        "rle": rle, "rle_statistics": rle_statistics, "longest_run": longest_run, "windowed_run_events": windowed_run_events,
        "windowed_run_count": windowed_run_count, "first_run": first_run, "last_run": last_run, "season_length": season_length,
        "resample_and_rl": resample_and_rl, "calc_perc": calc_perc, "sdba_quantile": sdba_quantile,
        "sdba_interp_on_quantiles": sdba_interp_on_quantiles,
        "MissingAny.__call__": missing_any_call,
        **missing_calls,
    }
    out = {name: forwarding(name, fn) for name, fn in table.items()}
    out["_clear_valid_cache"] = valid_cache.clear  # (patch.uninstall)
    # resample_and_rl receives the PATCHED rl.<name> objects (the forwarding wrappers): map those to the host mirrors too
    for inner, host in list(rl_host.items()):
        for name, fn in table.items():
            if fn is inner:
                rl_host[out[name]] = host
    lazy_ok.update(out[n] for n in ("windowed_run_count", "windowed_run_events", "rle_statistics", "longest_run"))
    return out
