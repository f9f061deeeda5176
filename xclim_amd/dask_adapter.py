"""dask-facing wrapper: index functions as ``map_blocks`` callees.

The reference parallelises chunked inputs with ``xr.map_blocks`` / ``apply_ufunc(dask="parallelized")``
(indices/helpers.py:898-974 ``resample_map``, core/indicator.py:865-944, core/calendar.py:469-479): dask calls a numpy
callee once per chunk and stitches the results.  Here the callee is an index of :mod:`xclim_amd.indices` bound to its
time axis: it takes the numpy block(s) of one chunk — the WHOLE time axis, any spatial extent, exactly what the reference
requires of its own chunked paths (``percentile_doy`` re-chunks to ``time: -1``, core/calendar.py:463-467) — and returns
the ``(periods, *chunk cells)`` numpy block.

* :func:`block_function` builds the callee; hand it to ``dask.array.map_blocks`` / ``xr.map_blocks`` yourself, or
* :func:`map_blocks` does it: with dask arrays it builds the graph (``drop_axis`` / ``new_axis`` for the time -> period
  axis); with numpy arrays and ``chunks=`` it walks the chunk grid itself (the same callee, the same stitching; this is
  also how the adapter is tested where dask is not installed).

Each block call uploads its chunk, runs the kernels and downloads the result.  Inside one big host array the slab
pipeline of :mod:`xclim_amd.blocks` (upload / compute / download overlapped on three streams) is the faster engine;
dask's own scheduler may call the callee from several threads: a block call holds the device's lock (``Device.lock``, an
RLock) from its first upload to its last download, so the kernels of two blocks never interleave on the one stream.
"""

from __future__ import annotations

import itertools

import numpy as np

from . import indices as _indices
from ._capi import get_device
from .timeaxis import TimeAxis

__all__ = ["block_function", "map_blocks", "chunk_grid"]


def block_function(index, time: TimeAxis, *args, nvars: int = 1, **kwargs):
    """``f(*blocks) -> numpy`` for ``index`` (a function of :mod:`xclim_amd.indices` or its name): the first `nvars`
    positional arguments of the index are the data blocks, `args` / `kwargs` follow (thresholds, ``freq=``, ...); the
    ``time=`` axis is passed the way the index expects it (positionally after the data and thresholds for most indices
    — use keyword arguments for everything but the data to stay independent of the order)."""
    fn = getattr(_indices, index) if isinstance(index, str) else index

    def callee(*blocks):
        if len(blocks) != nvars:
            raise ValueError(f"expected {nvars} data block(s), got {len(blocks)}")
        arrs = []
        for b in blocks:
            # the block in its OWN dtype: a float64 chunk takes the float64 kernels where they exist and raises
            # Float64FieldError elsewhere (_capi.handle_float64) — never rounded to float32 behind the caller's back
            a = np.ascontiguousarray(b)
            if a.shape[0] != len(time):
                raise ValueError("every block must hold the whole time axis (rechunk with time: -1, cal:463-467)")
            arrs.append(a)
        dev = kwargs.get("device") or get_device()
        with dev.lock:  # the whole multi-kernel index, not just each C call
            return np.asarray(fn(*arrs, *args, time=time, **kwargs))

    callee.__name__ = f"xclim_amd_{getattr(fn, '__name__', 'index')}"
    return callee


def chunk_grid(shape, chunks):
    """Slices of a dask-style chunk grid over the non-time axes: `chunks` = one chunk length (or tuple of lengths) per
    spatial axis."""
    spans = []
    for n, c in zip(shape, chunks):
        sizes = list(c) if isinstance(c, (tuple, list)) else [c] * (-(-n // c))
        edges = np.minimum(np.cumsum([0] + sizes), n)
        spans.append([slice(int(a), int(b)) for a, b in zip(edges[:-1], edges[1:]) if b > a])
    return list(itertools.product(*spans))


def map_blocks(index, arrays, time: TimeAxis, *args, chunks=None, **kwargs):
    """Apply an index chunk by chunk.  `arrays`: one array or a sequence (all chunked alike); dask arrays -> a lazy dask
    array (``dask.array.map_blocks``), numpy arrays + ``chunks=(cy, cx, ...)`` -> the stitched numpy result."""
    arrs = list(arrays) if isinstance(arrays, (list, tuple)) else [arrays]
    f = block_function(index, time, *args, nvars=len(arrs), **kwargs)
    if hasattr(arrs[0], "dask"):
        import dask.array as dsa

        if len(arrs[0].chunks[0]) != 1:
            raise ValueError("the time axis must be in one chunk (rechunk({0: -1}))")
        probe = f(*[np.zeros((len(time),) + (1,) * (a.ndim - 1), a.dtype) for a in arrs])  # (a float64 input may raise HERE)
        nper = probe.shape[0]  # (one single-cell call at graph-construction time: the period count and the result dtype)
        return dsa.map_blocks(f, *arrs, dtype=probe.dtype, chunks=((nper,),) + tuple(arrs[0].chunks[1:]))
    if chunks is None:
        return f(*arrs)
    a0 = np.asarray(arrs[0])
    out = None
    for sl in chunk_grid(a0.shape[1:], chunks):
        blk = f(*[np.asarray(a)[(slice(None),) + sl] for a in arrs])
        if out is None:
            out = np.empty((blk.shape[0],) + a0.shape[1:], blk.dtype)
        out[(slice(None),) + sl] = blk
    return out
