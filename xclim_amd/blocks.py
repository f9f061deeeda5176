"""Block adapter: stream host arrays through the device in slabs of grid cells.

The reference leaves chunked inputs to dask: ``xr.map_blocks`` / ``apply_ufunc(dask="parallelized")`` call the numpy
callee once per chunk (indices/helpers.py:898-974, core/indicator.py:865-944; core/calendar.py:469-479 for
percentile_doy).  Every op on the hot path is independent per grid cell, so a chunk here is a contiguous
slab ``[c0, c1)`` of the flattened cell axis with the WHOLE time axis — the same cut as the multi-GPU shards (shard.py).

``map_cell_blocks(func, inputs)`` uploads slab k + 1 on the copy-in lane while ``func`` computes slab k on the compute
stream and the results of slab k - 1 travel back on the copy-out lane (three HIP streams, fenced by events; device
inputs are double buffered).  Copies are strided 2-D DMA straight out of / into the caller's arrays; they are
asynchronous only for page-locked arrays (``Device.pinned_empty`` / ``Device.register``) — pageable arrays work, but
every copy then blocks the host.  ``func`` must stay on the device (``keep=True`` style calls, no ``.get()``).
"""

from __future__ import annotations

import numpy as np

from ._capi import DeviceArray, get_device

COMPUTE, COPY_IN, COPY_OUT = 0, 1, 2


def cell_blocks(ncells: int, block_cells: int, align: int = 4) -> list[tuple[int, int]]:
    """Contiguous slabs covering [0, ncells); every slab start is a multiple of `align` cells (16-byte row loads)."""
    if ncells < 0 or block_cells < 1:
        raise ValueError("ncells must be >= 0 and block_cells >= 1")
    step = max(align, block_cells // align * align)
    return [(c0, min(c0 + step, ncells)) for c0 in range(0, ncells, step)]


def default_block_cells(row_bytes: int, ncells: int, target_bytes: int = 256 << 20, align: int = 4) -> int:
    """Slab width such that one slab of all inputs is about `target_bytes` (row_bytes = sum of T_i * itemsize)."""
    cb = max(align, int(target_bytes // max(row_bytes, 1)) // align * align)
    return min(cb, max(ncells, align))


def map_cell_blocks(func, inputs, *, block_cells: int | None = None, device=None, pinned_out: bool = True, out=None):
    """Apply ``func(dev, *slabs) -> DeviceArray | tuple`` to cell slabs of the host arrays `inputs` and assemble the results.

    inputs : sequence of C-contiguous numpy arrays, time on axis 0, identical trailing (cell) shape; the time lengths
             and dtypes may differ (e.g. ref / hist / sim).
    func   : receives one dense ``(T_i, slab)`` DeviceArray per input and returns device arrays of shape ``(R_j, slab)``.
    out    : optional preallocated host array(s) ``(R_j, *cells)`` to receive the results (page-locked ones make the
             copy-out asynchronous); by default page-locked arrays are allocated (`pinned_out`), which costs about
             0.15 s per GB — pass `out` when a large output is produced repeatedly.
    Returns a numpy array ``(R_j, *cells)`` per output (a tuple when func returns a tuple).
    """
    dev = device or get_device()
    arrs = [np.asarray(a) for a in inputs]
    if not arrs:
        raise ValueError("at least one input is required")
    cell_shape = arrs[0].shape[1:]
    for a in arrs:
        if a.shape[1:] != cell_shape or not a.flags.c_contiguous or a.ndim < 1:
            raise ValueError("inputs must be C-contiguous with time on axis 0 and the same cell shape")
    C_ = int(np.prod(cell_shape, dtype=np.int64))
    flat = [a.reshape(a.shape[0], C_) for a in arrs]
    if block_cells is None:
        block_cells = default_block_cells(sum(a.shape[0] * a.itemsize for a in flat), C_)
    blocks = cell_blocks(C_, block_cells)
    if not blocks:
        raise ValueError("inputs have no grid cells")
    cbmax = blocks[0][1] - blocks[0][0]
    async_in = [dev.is_pinned(a) for a in flat]
    bufs = [[dev.empty((a.shape[0], cbmax), a.dtype) for a in flat] for _ in range(min(2, len(blocks)))]

    def upload(k):
        c0, c1 = blocks[k]
        views = []
        for a, b, pinned in zip(flat, bufs[k % 2], async_in):
            w = (c1 - c0) * a.itemsize
            dev.copy2d(b.ptr, w, a.ctypes.data + c0 * a.itemsize, C_ * a.itemsize, w, a.shape[0], "h2d", COPY_IN, not pinned)
            v = dev.wrap(b.ptr, (a.shape[0], c1 - c0), a.dtype)  # dense (T, slab) matrix inside the buffer
            v._owner = b
            views.append(v)
        return views

    outs_host, single, pending = None, False, None
    nxt = upload(0)
    for k, (c0, c1) in enumerate(blocks):
        cur = nxt
        dev.lane_fence(COPY_IN, COMPUTE)          # slab k is on the device before func touches it
        if k + 1 < len(blocks):
            dev.lane_fence(COMPUTE, COPY_IN)      # buffer set (k + 1) % 2 was last read by func on slab k - 1
            nxt = upload(k + 1)
        res = func(dev, *cur)
        single = isinstance(res, DeviceArray)
        res = (res,) if single else tuple(res)
        for r in res:
            if r.shape[-1] != c1 - c0:
                raise ValueError(f"func must return (R, slab) arrays; got {r.shape} for a slab of {c1 - c0} cells")
        if outs_host is None:
            out_shapes = [tuple(r.shape[:-1]) for r in res]
            if out is not None:
                given = [out] if isinstance(out, np.ndarray) else list(out)
                if len(given) != len(res):
                    raise ValueError(f"func returns {len(res)} arrays but {len(given)} output arrays were given")
                for g, r, shp in zip(given, res, out_shapes):
                    if g.shape != shp + tuple(cell_shape) or g.dtype != r.dtype or not g.flags.c_contiguous:
                        raise ValueError(f"out must be C-contiguous {r.dtype} of shape {shp + tuple(cell_shape)}")
                outs_host = [g.reshape(-1, C_) for g in given]
            else:
                alloc = dev.pinned_empty if pinned_out else (lambda shape, dtype: np.empty(shape, dtype))
                outs_host = [alloc((int(np.prod(shp, dtype=np.int64)), C_), r.dtype) for r, shp in zip(res, out_shapes)]
            async_out = [dev.is_pinned(h) for h in outs_host]
        if pending is not None:
            dev.lane_sync(COPY_OUT)               # slab k - 1 has arrived; its device buffers may return to the pool
        dev.lane_fence(COMPUTE, COPY_OUT)
        for r, h, pinned in zip(res, outs_host, async_out):
            w = (c1 - c0) * h.itemsize
            dev.copy2d(h.ctypes.data + c0 * h.itemsize, C_ * h.itemsize, r.ptr, w, w, h.shape[0], "d2h", COPY_OUT, not pinned)
        pending = res
    dev.lane_sync(COPY_OUT)
    dev.lane_sync(COMPUTE)
    del pending
    outs = [h.reshape(shp + tuple(cell_shape)) for h, shp in zip(outs_host, out_shapes)]
    return outs[0] if single else tuple(outs)
