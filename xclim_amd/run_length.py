"""Host mirror of ``xclim.indices.run_length`` (reference: src/xclim/indices/run_length.py) over the HIP kernels.

Inputs are masks with TIME ON AXIS 0 (bool, or float32 with NaN where ``select_time`` masked values); results are
identical for the reference's N-D and 1-D (``ufunc_1dim``) code paths, so that option is accepted and ignored
except for the one error the reference raises (rl:67-68).
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from ._capi import DeviceArray, get_device
from .calendar import _flatten
from .timeaxis import TimeAxis

npts_opt = 9000  # rl:26 — kept for API parity; no dispatch depends on it here


def use_ufunc(ufunc_1dim, da=None, dim="time", freq=None, index="first") -> bool:
    """rl:33-78: only the argument validation matters on this backend."""
    if ufunc_1dim is True and freq is not None:
        raise ValueError("Resampling after run length operations is not implemented for 1d method")
    return False


def _mask(da, dev):
    if isinstance(da, DeviceArray):
        return da.reshape(da.shape[0], -1), da.shape[1:]
    a = np.asarray(da)
    if a.dtype != np.float32:
        a = a.astype(np.float32)
    return _flatten(a, dev)


def _whole(T):
    return np.array([0, T], dtype=np.int64)


def _run(da, stat, window, time, freq, index, device, keep, cut=False):
    dev = device or get_device()
    m, cell_shape = _mask(da, dev)
    if freq is None:
        seg = _whole(m.shape[0])
        cut = True
    else:
        seg, _ = time.segments(freq)
    out, _ = K.run_stats(dev, m, stat, window, seg, cut=cut, index=index, want_valid=False)
    if keep:
        return out
    o = out.get().reshape((out.shape[0],) + tuple(cell_shape))
    return o[0] if freq is None else o


def _cumsum_reset(da, dim="time", index="last", *, device=None, keep=False):
    """rl:172-219."""
    dev = device or get_device()
    m, cell_shape = _mask(da, dev)
    out = K.cumsum_reset(dev, m, index)
    return out if keep else out.get().reshape((m.shape[0],) + tuple(cell_shape))


def rle(da, dim="time", index="first", *, device=None, keep=False):
    """rl:223-272."""
    dev = device or get_device()
    m, cell_shape = _mask(da, dev)
    out = K.rle(dev, m, index)
    return out if keep else out.get().reshape((m.shape[0],) + tuple(cell_shape))


def rle_statistics(da, reducer: str, window: int, dim="time", freq=None, ufunc_1dim="from_context", index="first", *,
                   time: TimeAxis | None = None, device=None, keep=False):
    """rl:275-335.  ``freq`` given -> resample AFTER the run-length encoding."""
    use_ufunc(ufunc_1dim, freq=freq, index=index)
    if reducer.startswith("q"):
        raise NotImplementedError("quantile run statistics are not wired to the HIP path yet")
    return _run(da, reducer, window, time, freq, index, device, keep)


def longest_run(da, dim="time", freq=None, ufunc_1dim="from_context", index="first", *, time=None, device=None,
                keep=False):
    """rl:338-378."""
    return rle_statistics(da, "max", 1, dim, freq, ufunc_1dim, index, time=time, device=device, keep=keep)


def windowed_run_events(da, window: int, dim="time", freq=None, ufunc_1dim="from_context", index="first", *, time=None,
                        device=None, keep=False):
    """rl:381-434."""
    use_ufunc(ufunc_1dim, freq=freq, index=index)
    return _run(da, "count", window, time, freq, index, device, keep)


def windowed_run_count(da, window: int, dim="time", freq=None, ufunc_1dim="from_context", index="first", *, time=None,
                       device=None, keep=False):
    """rl:437-488 (window == 1 and freq None: plain sum, rl:478-479)."""
    use_ufunc(ufunc_1dim, freq=freq, index=index)
    stat = "plainsum" if (window == 1 and freq is None) else "sum"
    return _run(da, stat, window, time, freq, index, device, keep)


def first_run(da, window: int, dim="time", freq=None, coord=None, ufunc_1dim="from_context", *, time=None, device=None,
              keep=False):
    """rl:643-690 (index output; coord lookups are host work on the 1-D time axis)."""
    return _run(da, "first", window, time, freq, "first", device, keep, cut=(window == 1))


def last_run(da, window: int, dim="time", freq=None, coord=None, ufunc_1dim="from_context", *, time=None, device=None,
             keep=False):
    """rl:693-740."""
    return _run(da, "last", window, time, freq, "first", device, keep, cut=(window == 1))


_STATS = {"rle_statistics", "longest_run", "windowed_run_events", "windowed_run_count", "first_run", "last_run"}


def resample_and_rl(da, resample_before_rl: bool, compute, *args, freq: str, time: TimeAxis, dim="time", device=None,
                    keep=False, **kwargs):
    """rl:87-132: cut the series per period first (default of every index) or resample after."""
    name = compute.__name__
    if name not in _STATS:
        raise NotImplementedError(f"resample_and_rl: {name} is not available on the HIP path")
    if not resample_before_rl:
        return compute(da, *args, freq=freq, time=time, device=device, keep=keep, **kwargs)
    # resample before: same kernel with cut_at_segments = 1
    params = dict(kwargs)
    names = {"rle_statistics": ("reducer", "window"), "longest_run": (), "windowed_run_events": ("window",),
             "windowed_run_count": ("window",), "first_run": ("window",), "last_run": ("window",)}[name]
    params.update(dict(zip(names, args)))
    index = params.get("index", "first")
    window = params.get("window", 1)
    if name == "rle_statistics":
        stat = params["reducer"]
    elif name == "longest_run":
        stat = "max"
    elif name == "windowed_run_events":
        stat = "count"
    elif name == "windowed_run_count":
        stat = "plainsum" if window == 1 else "sum"
    else:
        stat = "first" if name == "first_run" else "last"
    dev = device or get_device()
    m, cell_shape = _mask(da, dev)
    seg, _ = time.segments(freq)
    out, _ = K.run_stats(dev, m, stat, window, seg, cut=True, index=index, want_valid=False)
    return out if keep else out.get().reshape((out.shape[0],) + tuple(cell_shape))
