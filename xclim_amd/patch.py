"""Boundary adapter: the callables that slot into xclim in place of the reference's hot-path functions.

Two tiers, as the reference resolves its own hot path:

* **tier 2 — ``xr.apply_ufunc`` callees** (numpy in, numpy out; no xarray needed, tested on the GPU in
  tests/test_gpu_patch.py):

  - :func:`cumsum_reset_np` replaces ``xclim.indices.run_length._cumsum_reset_np`` (run_length.py:143-151, called at
    :209-216): the core dim arrives LAST, possibly as a non-contiguous view of the ``(time, lat, lon)`` array; the callee
    MUTATES ``arr`` and returns it.
  - :func:`calc_perc` replaces ``xclim.core.utils.calc_perc`` (core/utils.py:279-323, imported at call time by
    ``percentile_doy``, core/calendar.py:441, 469-479): ``(…, stack_dim)`` strided view -> ``(…, nper)``.

* **tier 1 — module attributes** (same-signature functions on ``xr.DataArray``): :func:`install` patches them when xarray
  and xclim are importable.  xarray cannot be installed in the build environment, so this part is exercised only where
  it is (``python -c "import xclim_amd.patch as p; p.install()"`` then the reference's own test-suite); the wrappers
  are deliberately thin: unwrap ``.values`` with time first, call the host mirror, re-wrap with the coordinates of the
  reference's own resample template.

``percentile_doy`` keeps a ``__wrapped__`` attribute because ``bootstrap_func`` calls ``percentile_doy.__wrapped__``
(core/bootstrapping.py:195).
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from . import utils as _hutl
from ._capi import get_device

__all__ = ["cumsum_reset_np", "calc_perc", "install", "uninstall"]


# ---- tier 2 --------------------------------------------------------------------------------------------------------
def cumsum_reset_np(arr: np.ndarray, index: str, one=None, *, device=None) -> np.ndarray:
    """Drop-in for ``run_length._cumsum_reset_np(arr, index, one)`` (run_length.py:143-151).

    ``arr``: binary (0 / 1) values, integer or float dtype, core dim on the LAST axis, leading dims arbitrary, possibly a
    non-contiguous view.  ``index``: "last" (forward) or "first" (backward).  ``one`` is only the dtype-carrying
    constant of the numba version and is ignored.  The result is written into ``arr`` (its dtype) and ``arr`` is
    returned, like the reference.  The transposed view xarray hands over (time moved last on a ``(time, lat, lon)``
    array) is uploaded without a host-side copy: moving the axis back gives the C-contiguous ``(T, C)`` layout the
    kernel wants.
    """
    if index not in ("first", "last"):
        raise ValueError(f"index must be 'first' or 'last', got {index!r}")
    if arr.ndim == 0 or arr.shape[-1] == 0 or arr.size == 0:
        return arr
    dev = device or get_device()
    tfirst = np.moveaxis(arr, -1, 0)  # (T, ...) view
    T = tfirst.shape[0]
    flat = np.ascontiguousarray(tfirst).reshape(T, -1)  # no copy when `arr` is the transposed view of a C-ordered array
    if flat.dtype in (np.bool_, np.uint8):
        m = K.mask_to_f32(dev, dev.to_device(flat.view(np.uint8)))
    else:
        m = dev.to_device(flat, dtype=np.float32)
    out = K.cumsum_reset(dev, m, index).get().reshape(tfirst.shape)
    tfirst[...] = out.astype(arr.dtype, copy=False)  # writes through the view into `arr`
    return arr


def calc_perc(arr: np.ndarray, percentiles=None, alpha: float = 1.0, beta: float = 1.0, copy: bool = True, *, device=None):
    """Drop-in for ``core.utils.calc_perc`` (core/utils.py:279-323): percentiles along the last axis of a possibly strided
    ``(…, N)`` view, percentile axis LAST in the result, float64.  ``copy`` is accepted and irrelevant: the input is never
    modified (the reference sorts a copy unless told otherwise)."""
    return _hutl.calc_perc(arr, percentiles, alpha, beta, copy, device=device)


# ---- tier 1 (needs xarray + xclim) ----------------------------------------------------------------------------------
_saved: dict = {}


def _time_axis(da):
    from .timeaxis import TimeAxis

    t = da["time"].dt
    return TimeAxis(t.year.values, t.month.values, t.day.values, str(t.calendar))


def _tfirst_f32(da):
    return np.ascontiguousarray(da.transpose("time", ...).values, dtype=np.float32)


def _make_wrappers():
    import xarray as xr

    from . import calendar as hcal
    from . import generic as hgen

    def threshold_count(da, op, threshold, freq, constrain=None):  # indices/generic.py:329-361
        thr = threshold.transpose("time", ...).values if isinstance(threshold, xr.DataArray) else threshold
        out = hgen.threshold_count(_tfirst_f32(da), op, thr, _time_axis(da), freq, constrain)
        tmpl = da.transpose("time", ...).resample(time=freq).first(skipna=False)  # period labels / coordinates only
        return tmpl.copy(data=np.asarray(out).astype("int64").reshape(tmpl.shape))

    def percentile_doy(arr, window=5, per=10.0, alpha=1.0 / 3.0, beta=1.0 / 3.0, copy=True):  # core/calendar.py:395-494
        from xclim.core.calendar import build_climatology_bounds

        pers = [per] if np.isscalar(per) else list(per)
        a = arr.transpose("time", ...)
        p = hcal.percentile_doy(_tfirst_f32(a), _time_axis(a), window=window, per=pers, alpha=alpha, beta=beta)
        data = p.values()  # (ndoy, *cells, nper): the reference's dim order
        dims = ("dayofyear",) + tuple(d for d in a.dims if d != "time") + ("percentiles",)
        coords = {d: a[d] for d in a.dims if d != "time" and d in a.coords}
        coords.update(dayofyear=np.asarray(p.dayofyear), percentiles=pers)
        out = xr.DataArray(data, dims=dims, coords=coords, attrs=dict(arr.attrs), name="per")
        out.attrs.update(climatology_bounds=build_climatology_bounds(arr), window=window, alpha=alpha, beta=beta)
        return out

    percentile_doy.__wrapped__ = percentile_doy  # bootstrap_func calls percentile_doy.__wrapped__ (bootstrapping.py:195)
    return {"threshold_count": threshold_count, "percentile_doy": percentile_doy}


def install() -> list[str]:
    """Patch the reference's module attributes (SURVEY.md §8b resolution rules); returns the patched names.

    ``rl.*`` is always reached through the module object, ``threshold_count`` / ``percentile_doy`` are imported BY NAME
    into the index modules and must be replaced in each of them, ``calc_perc`` is imported at call time."""
    import importlib

    wrappers = _make_wrappers()
    done = []

    def patch(modname, attr, fn):
        try:
            mod = importlib.import_module(modname)
        except ImportError:
            return
        if hasattr(mod, attr):
            _saved.setdefault((modname, attr), getattr(mod, attr))
            setattr(mod, attr, fn)
            done.append(f"{modname}.{attr}")

    patch("xclim.indices.run_length", "_cumsum_reset_np", cumsum_reset_np)
    patch("xclim.core.utils", "calc_perc", calc_perc)
    for modname in ("xclim.indices.generic", "xclim.indices._threshold", "xclim.indices._multivariate",
                    "xclim.indices._simple", "xclim.indices._hydrology", "xclim.indices._anuclim"):
        patch(modname, "threshold_count", wrappers["threshold_count"])
    for modname in ("xclim.core.calendar", "xclim.indices._multivariate", "xclim.core.bootstrapping", "xclim.indices.stats"):
        patch(modname, "percentile_doy", wrappers["percentile_doy"])
    return done


def uninstall() -> None:
    import importlib

    for (modname, attr), fn in _saved.items():
        setattr(importlib.import_module(modname), attr, fn)
    _saved.clear()
