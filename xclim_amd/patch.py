"""Boundary adapter: the callables that slot into xclim in place of the reference's hot-path functions.

Two tiers, as the reference resolves its own hot path:

* **tier 2 — ``xr.apply_ufunc`` callees** (numpy in, numpy out; no xarray needed, tested on the GPU in
  tests/test_gpu_patch.py):

  - :func:`cumsum_reset_np` replaces ``xclim.indices.run_length._cumsum_reset_np`` (run_length.py:143-151, called at
    :209-216): the core dim arrives LAST, possibly as a non-contiguous view of the ``(time, lat, lon)`` array; the callee
    MUTATES ``arr`` and returns it.
  - :func:`calc_perc` replaces ``xclim.core.utils.calc_perc`` (core/utils.py:279-323, imported at call time by
    ``percentile_doy``, core/calendar.py:441, 469-479): ``(…, stack_dim)`` strided view -> ``(…, nper)``.

* **tier 1 — module attributes** (same-signature functions on ``xr.DataArray``, xclim_amd/xr_adapter.py):
  :func:`install` replaces them in every module that holds them (``rl.*`` through the module object, the generic /
  calendar functions also in the index modules that imported them by name, ``calc_perc`` at its definition).  xarray is
  not installable in the build environment, so the wrappers take what they need from xarray / xclim through an ``Env``
  and are EXECUTED in tests/test_gpu_adapter.py against a small DataArray stand-in (tests/fakexr.py) installed into
  stand-in modules with the reference's import structure; with the real packages:
  ``python -c "import xclim_amd.patch as p; print(p.install())"``.

``percentile_doy`` keeps a ``__wrapped__`` attribute because ``bootstrap_func`` calls ``percentile_doy.__wrapped__``
(core/bootstrapping.py:195).
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from . import utils as _hutl
from ._capi import get_device

__all__ = ["cumsum_reset_np", "calc_perc", "install", "uninstall", "real_env"]


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

# reference module -> names imported BY NAME there (/root/reference/src/xclim/indices/_threshold.py:25-36,
# _multivariate.py:13, 22-24, _simple.py:10, _hydrology.py:14, _anuclim.py:26, core/bootstrapping.py:17, indices/stats.py:23);
# a name is only replaced where the module really holds it (hasattr), so the table may list more than a version imports
_GENERIC_NAMES = ("threshold_count", "count_occurrences", "domain_count", "select_resample_op", "spell_length_statistics",
                  "cumulative_difference", "compare", "season", "first_day_threshold_reached", "bivariate_count_occurrences")
_BY_NAME = {
    "xclim.indices.generic": _GENERIC_NAMES,
    "xclim.indices._threshold": _GENERIC_NAMES,
    "xclim.indices._multivariate": _GENERIC_NAMES + ("percentile_doy", "resample_doy"),
    "xclim.indices._simple": _GENERIC_NAMES,
    "xclim.indices._hydrology": _GENERIC_NAMES,
    "xclim.indices._anuclim": _GENERIC_NAMES,
    "xclim.indices._agro": _GENERIC_NAMES + ("percentile_doy", "resample_doy"),
    "xclim.indices._conversion": _GENERIC_NAMES,
    "xclim.indicators.generic._stats": ("select_resample_op",),   # indicators/generic/_stats.py:6
    "xclim.ensembles._robustness": ("compare",),                  # ensembles/_robustness.py:24
    "xclim.core.calendar": ("percentile_doy", "resample_doy"),
    "xclim.core.bootstrapping": ("percentile_doy",),
    "xclim.indices.stats": ("percentile_doy",),
    # rl.* is always reached through the module object: patching the module attributes is enough (gen:37, _threshold.py:25)
    "xclim.indices.run_length": ("rle", "rle_statistics", "longest_run", "windowed_run_events", "windowed_run_count",
                                 "first_run", "last_run", "season_length", "resample_and_rl"),
    "xclim.core.utils": ("calc_perc",),
}


def real_env():
    """The :class:`xr_adapter.Env` of an installation that has xarray and xclim."""
    import xarray as xr
    from xclim.core.calendar import build_climatology_bounds
    from xclim.core.units import convert_units_to, pint2cfattrs, to_agg_units, units2pint

    from .xr_adapter import Env

    def finish_select_resample_op(out, da, op, out_units):  # the tail of indices/generic.py:118-125
        if out_units is not None:
            return out.assign_attrs(units=out_units)
        if op in ("std", "var"):
            out.attrs.update(pint2cfattrs(units2pint(out.attrs["units"]), is_difference=True))
        return to_agg_units(out, da, op)

    return Env(xr.DataArray, convert_units_to, to_agg_units, finish_select_resample_op, build_climatology_bounds,
               difference_attrs=lambda u: pint2cfattrs(units2pint(u), is_difference=True))


def install(env=None, modules=None) -> list[str]:
    """Replace the reference's hot-path functions (SURVEY.md §8b resolution rules); returns the patched names.

    ``env`` / ``modules``: dependency injection for the tests (tests/test_gpu_adapter.py installs the wrappers into
    stand-in modules with the reference's import structure, because xarray / xclim are not installable there);
    by default the real packages are used.  Functions the HIP path does not serve are forwarded to the saved originals."""
    import importlib

    from .xr_adapter import make_wrappers

    env = env or real_env()

    def resolve(modname):
        if modules is not None:
            return modules.get(modname)
        try:
            return importlib.import_module(modname)
        except ImportError:
            return None

    orig = {}  # the reference's own functions, for the calls the HIP path forwards
    for modname in ("xclim.indices.generic", "xclim.indices.run_length", "xclim.core.calendar", "xclim.core.utils"):
        mod = resolve(modname)
        for name in _BY_NAME[modname]:
            if mod is not None and hasattr(mod, name):
                orig.setdefault(name, _saved.get((modname, name), getattr(mod, name)))
    nbu = resolve("xsdba.nbutils")
    if nbu is not None and hasattr(nbu, "quantile"):  # the original behind the wrapper's forwards (other dims, float64 fields)
        orig["sdba_quantile"] = _saved.get(("xsdba.nbutils", "quantile"), nbu.quantile)
    sut = resolve("xsdba.utils")
    if sut is not None and hasattr(sut, "interp_on_quantiles"):
        orig["sdba_interp_on_quantiles"] = _saved.get(("xsdba.utils", "interp_on_quantiles"), sut.interp_on_quantiles)
    wrappers = make_wrappers(env, orig)
    _cleanups.append(wrappers.pop("_clear_valid_cache"))
    wrappers["_cumsum_reset_np"] = cumsum_reset_np
    done = []

    def patch(modname, attr, fn):
        mod = resolve(modname)
        if mod is not None and hasattr(mod, attr):
            _saved.setdefault((modname, attr), getattr(mod, attr))
            setattr(mod, attr, fn)
            done.append(f"{modname}.{attr}")

    for modname, names in _BY_NAME.items():
        for name in names:
            patch(modname, name, wrappers[name])
    patch("xclim.indices.run_length", "_cumsum_reset_np", cumsum_reset_np)
    # the missing-value check of Indicator._postprocess (core/indicator.py:1522-1549): a METHOD of MissingAny
    miss_mod = resolve("xclim.core.missing")
    # ... and of the other registered methods (:325-512; MissingTwoSteps.__call__ :352-393 for wmo / pct / at_least_n).  Each
    # class is patched where it DEFINES __call__ semantics of its own; the base classes stay untouched (custom subclasses
    # registered by users keep the reference's code)
    for cname in ("MissingAny", "MissingSomeButNotAll", "MissingWMO", "MissingPct", "AtLeastNValid"):
        cls = getattr(miss_mod, cname, None) if miss_mod is not None else None
        key = f"{cname}.__call__"
        if cls is not None and key in wrappers:
            if ("xclim.core.missing", key) not in _saved:
                # (the function found on the class — possibly inherited; uninstall() removes the override again)
                _saved[("xclim.core.missing", key)] = cls.__dict__.get("__call__", _INHERITED)
            own = _saved[("xclim.core.missing", key)]
            # what the wrapper forwards to: the class's own function, or the one it inherits (MissingBase / MissingTwoSteps)
            orig[key] = own if own is not _INHERITED else next(b.__dict__["__call__"] for b in cls.__mro__[1:] if "__call__" in b.__dict__)
            cls.__call__ = wrappers[key]
            done.append(f"xclim.core.missing.{key}")
    # Indicator.__call__ (core/indicator.py:865-944): parse -> compute (the index, its percentile / resample helpers) ->
    # missing-value check (:1522-1549), all on the same DataArrays and without user code in between: ONE scope of
    # device-resident inputs (Device.keep_inputs) per call, dropped when the call returns
    ind_mod = resolve("xclim.core.indicator")
    icls = getattr(ind_mod, "Indicator", None) if ind_mod is not None else None
    if icls is not None:
        if ("xclim.core.indicator", "Indicator.__call__") not in _saved:
            _saved[("xclim.core.indicator", "Indicator.__call__")] = icls.__call__
        icall = _saved[("xclim.core.indicator", "Indicator.__call__")]

        def _indicator_call(self, *args, **kwds):
            with get_device().keep_inputs():
                return icall(self, *args, **kwds)

        _indicator_call.__wrapped__ = icall
        _indicator_call.__doc__ = icall.__doc__
        icls.__call__ = _indicator_call
        done.append("xclim.core.indicator.Indicator.__call__")
    # xsdba (third party, re-exported by src/xclim/sdba.py:10): the per-cell multi-quantile entry point; xsdba's own
    # modules reach it through the module object (``nbu.quantile``), so the one attribute is enough
    patch("xsdba.nbutils", "quantile", wrappers["sdba_quantile"])
    # qm_adjust / qdm_adjust reach the factor interpolation as ``u.interp_on_quantiles`` (module object): group="time" -> xh_eqm_adjust
    patch("xsdba.utils", "interp_on_quantiles", wrappers["sdba_interp_on_quantiles"])
    _saved_modules.update({} if modules is None else modules)
    return done


_INHERITED = object()  # marker in _saved: the class did not define the attribute itself (uninstall deletes the override)
_saved_modules: dict = {}
_cleanups: list = []  # per install(): drops the valid-count cache of its wrappers


def uninstall() -> None:
    import importlib

    for (modname, attr), fn in _saved.items():
        mod = _saved_modules.get(modname) or importlib.import_module(modname)
        if "." in attr:  # a method: "Class.name"
            cname, meth = attr.split(".")
            if fn is _INHERITED:
                delattr(getattr(mod, cname), meth)
            else:
                setattr(getattr(mod, cname), meth, fn)
        else:
            setattr(mod, attr, fn)
    _saved.clear()
    _saved_modules.clear()
    for fn in _cleanups:
        fn()
    _cleanups.clear()
