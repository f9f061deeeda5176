"""Multi-GPU layout: the (lat, lon) grid shards over ranks, one process per GPU, no halo.

Every op on the path is independent per grid cell (only time-axis windows exist), so the flattened cell axis is
cut into contiguous slabs, each rank runs the single-GPU kernels on its slab, and the ONLY exchange is one
all-gather of the reduced ``(P, cells)`` outputs — RCCL over xGMI when the tensors are on the GPU (torch.distributed
backend "nccl" is RCCL on ROCm), gloo for the CPU tests.  ``scen``-sized outputs (EQM adjust) stay sharded.

torch is imported lazily: the single-GPU product path never needs it.
"""

from __future__ import annotations

import numpy as np


def shard_bounds(ncells: int, world: int, rank: int, align: int = 4) -> tuple[int, int]:
    """Contiguous slab [c0, c1) of rank `rank`; slab starts are multiples of `align` cells so that 16-byte row
    loads stay aligned on every rank.  Slabs differ by at most `align` cells."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    units = -(-ncells // align)  # ceil
    base, extra = divmod(units, world)
    u0 = rank * base + min(rank, extra)
    u1 = u0 + base + (1 if rank < extra else 0)
    return min(u0 * align, ncells), min(u1 * align, ncells)


def all_bounds(ncells: int, world: int, align: int = 4):
    return [shard_bounds(ncells, world, r, align) for r in range(world)]


def gather_cells(local, ncells: int, group=None, align: int = 4):
    """All-gather per-rank ``(P, c_local)`` results into the full ``(P, ncells)`` array on every rank.

    `local`: numpy array (-> gloo/CPU tensors) or a torch tensor (CUDA tensors go through RCCL).  Shards are padded
    to the largest slab so that one fixed-size collective suffices.
    """
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    bounds = all_bounds(ncells, world, align)
    cmax = max(b - a for a, b in bounds)
    as_numpy = isinstance(local, np.ndarray)
    t = torch.from_numpy(np.ascontiguousarray(local)) if as_numpy else local
    P = t.shape[0]
    pad = torch.zeros((P, cmax), dtype=t.dtype, device=t.device)
    pad[:, : t.shape[1]] = t
    out = torch.empty((world, P, cmax), dtype=t.dtype, device=t.device)
    dist.all_gather_into_tensor(out.view(world * P, cmax), pad, group=group)
    full = torch.cat([out[r, :, : b - a] for r, (a, b) in enumerate(bounds)], dim=1)
    return full.numpy() if as_numpy else full
