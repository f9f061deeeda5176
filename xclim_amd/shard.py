"""Multi-GPU layout: the (lat, lon) grid shards over ranks, one process per GPU, no halo.

Every op on the path is independent per grid cell (only time-axis windows exist), so the flattened cell axis is
cut into contiguous slabs, each rank runs the single-GPU kernels on its slab, and the ONLY exchange is one
all-gather of the reduced ``(P, cells)`` outputs.  ``scen``-sized outputs (EQM adjust) stay sharded.

The exchange goes through the C ABI (``xh_comm_*`` in include/xclim_hip.h: RCCL over xGMI, librccl dlopen'ed by
libxclimhip.so) — :class:`Comm` below.  Nothing in this package imports torch: the gloo all-gather that checks the slab
arithmetic on CPU-only machines lives with its test (tests/test_shard_gloo.py).
"""

from __future__ import annotations

import ctypes as C
import os
import time

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


# ---- RCCL through the C ABI (no torch) ---------------------------------------------------------------------------------
def _rendezvous_path() -> str:
    """Node-local file through which rank 0 hands the RCCL unique id to the other ranks of ONE launch.  The name is built
    from what every rank of a launch shares and consecutive launches do not: the rendezvous endpoint and the PID of
    the common parent (the torchrun agent / the shell that started the ranks)."""
    d = os.environ.get("XH_RENDEZVOUS_DIR") or os.environ.get("TMPDIR") or "/tmp"
    key = os.environ.get("XH_RENDEZVOUS_KEY") or "{}_{}_{}".format(
        os.environ.get("MASTER_ADDR", "local"), os.environ.get("MASTER_PORT", "0"), os.getppid())
    return os.path.join(d, f"xclim_amd_rccl_{key}.id")


def exchange_unique_id(path: str, rank: int, make_id, nbytes: int, timeout_s: float = 300.0) -> bytes:
    """The file rendezvous of a launch: rank 0 creates the id (``make_id()``) and publishes it atomically at `path`, every
    other rank polls until the complete id is there.  Pure host logic (covered on the CPU by tests/test_shard_gloo.py)."""
    if rank == 0:
        uid = make_id()
        if len(uid) != nbytes:
            raise ValueError(f"unique id must be {nbytes} bytes")
        tmp = f"{path}.{os.getpid()}.tmp"
        with open(tmp, "wb") as f:
            f.write(uid)
        os.replace(tmp, path)  # atomic: readers see the whole id or nothing
        return uid
    t0 = time.time()
    while True:
        try:
            with open(path, "rb") as f:
                uid = f.read()
            if len(uid) == nbytes:
                return uid
        except OSError:
            pass
        if time.time() - t0 > timeout_s:
            raise TimeoutError(f"rank {rank}: no RCCL unique id at {path} after {timeout_s:.0f} s")
        time.sleep(0.02)


class FileComm:
    """Last-resort stand-in for :class:`Comm` when RCCL cannot be brought up on a node (no librccl, IPC refused ...):
    barriers and scalar reductions through files in the rendezvous directory, NO data exchange — ``all_gather`` only
    copies the rank's own block into its place.  ``bench.py`` uses it so that a multi-GPU launch still reports the
    sharded throughput (flagged ``"exchange": "none (RCCL unavailable: ...)"``) instead of dying; library users get the
    original exception from :meth:`Comm.from_env`."""

    kind = "file"

    def __init__(self, dev, world: int, rank: int, base: str, reason: str = ""):
        self.dev, self.world, self.rank, self.reason = dev, int(world), int(rank), reason
        self._base, self._n = base, 0

    def _exchange(self, value: float, timeout_s: float = 300.0):
        tag = f"{self._base}.b{self._n}"
        self._n += 1
        mine = f"{tag}.r{self.rank}"
        with open(mine + ".tmp", "w") as f:
            f.write(repr(float(value)))
        os.replace(mine + ".tmp", mine)
        vals, t0 = [], time.time()
        for r in range(self.world):
            while True:
                try:
                    with open(f"{tag}.r{r}") as f:
                        vals.append(float(f.read()))
                    break
                except (OSError, ValueError):
                    if time.time() - t0 > timeout_s:
                        raise TimeoutError(f"rank {self.rank}: rank {r} did not reach barrier {self._n - 1}")
                    time.sleep(0.002)
        return vals

    def all_gather(self, send, recv, slot: int = -1) -> None:
        isz = send.nbytes
        self.dev.copy2d(recv.ptr + self.rank * isz, isz, send.ptr, isz, isz, 1, "d2d", blocking=False)

    def fence(self, slot: int) -> None:
        pass

    def sync(self) -> None:
        self.dev.sync()

    def barrier(self) -> None:
        self.dev.sync()
        self._exchange(0.0)

    def allreduce(self, values, op: str = "max"):
        v = np.atleast_1d(np.asarray(values, dtype=np.float64)).copy()
        red = {"sum": np.sum, "max": np.max, "min": np.min}[op]
        for i in range(len(v)):
            v[i] = red(self._exchange(v[i]))
        return v

    def close(self) -> None:
        """Remove this rank's barrier files (after a last barrier: nobody is still polling for them)."""
        import glob

        try:
            self._exchange(0.0, timeout_s=30.0)
        except TimeoutError:
            pass
        for f in glob.glob(f"{self._base}.b*.r{self.rank}"):
            if not f.endswith(f".b{self._n - 1}.r{self.rank}"):  # the last one may still be read by a slower rank
                try:
                    os.unlink(f)
                except OSError:
                    pass


class Comm:
    """One RCCL communicator over the ranks of a launch (one process per GPU), on a Device's context.

    ``Comm.from_env(dev)`` reads RANK / WORLD_SIZE (as set by torch.distributed.run, mpirun wrappers, ...) and does the
    file rendezvous; ``all_gather`` / ``fence`` / ``sync`` / ``allreduce`` / ``barrier`` map one to one onto xh_comm_*."""

    ID_BYTES = 128

    def __init__(self, dev, world: int, rank: int, unique_id: bytes):
        if len(unique_id) != self.ID_BYTES:
            raise ValueError("unique_id must be 128 bytes")
        self.dev, self.world, self.rank = dev, int(world), int(rank)
        buf = C.create_string_buffer(unique_id, self.ID_BYTES)
        h = C.c_void_p()
        dev.call("xh_comm_init", self.world, self.rank, C.cast(buf, C.c_void_p), C.byref(h))
        self.handle = h

    @staticmethod
    def new_unique_id(dev) -> bytes:
        from ._capi import _check

        buf = C.create_string_buffer(Comm.ID_BYTES)
        _check(dev.lib, dev.lib.xh_comm_unique_id(C.cast(buf, C.c_void_p)))
        return buf.raw

    @classmethod
    def from_env(cls, dev, timeout_s: float = 300.0) -> "Comm":
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
        local_world = os.environ.get("LOCAL_WORLD_SIZE")
        if local_world is not None and int(local_world) != world:
            # the unique id travels through a node-LOCAL file: a launch over several nodes would wait for it until the
            # timeout — say so at once instead
            raise RuntimeError(f"Comm.from_env: WORLD_SIZE={world} but LOCAL_WORLD_SIZE={local_world}: the file rendezvous "
                               "of the RCCL id only works inside one node (hand the id over yourself and use Comm(...))")
        # one node, one process per GPU: RCCL's socket bootstrap runs over the loopback interface unless the caller chose
        # another one (a box without an outside interface has nothing else; the data path is xGMI / shared memory anyway)
        os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
        path = _rendezvous_path()
        uid = exchange_unique_id(path, rank, lambda: cls.new_unique_id(dev), cls.ID_BYTES, timeout_s)
        comm = cls(dev, world, rank, uid)  # collective: returns once every rank has joined (and so has read the file)
        if rank == 0:
            try:
                os.unlink(path)
            except OSError:
                pass
        return comm

    def _call(self, name, *args):
        from ._capi import _check

        with self.dev.lock:
            _check(self.dev.lib, getattr(self.dev.lib, name)(self.handle, *args))

    def all_gather(self, send, recv, slot: int = -1) -> None:
        """recv (device, world * send.nbytes) <- every rank's `send` (device); slot >= 0: overlapped (see the header)."""
        if recv.nbytes != self.world * send.nbytes:
            raise ValueError("all_gather: recv must hold world * send.nbytes bytes")
        self._call("xh_comm_allgather", C.c_void_p(send.ptr), C.c_void_p(recv.ptr), send.nbytes, int(slot))

    def fence(self, slot: int) -> None:
        self._call("xh_comm_fence", int(slot))

    def sync(self) -> None:
        self._call("xh_comm_sync")

    def barrier(self) -> None:
        self._call("xh_comm_barrier")

    def allreduce(self, values, op: str = "max"):
        v = np.ascontiguousarray(np.atleast_1d(values), dtype=np.float64)
        self._call("xh_comm_allreduce_f64", v.ctypes.data_as(C.POINTER(C.c_double)), len(v), {"sum": 0, "max": 2, "min": 3}[op])
        return v

    def gather_cells(self, local, ncells: int, align: int = 4, out=None):
        """All-gather per-rank (P, c_local) device arrays (slabs of `shard_bounds`) into (world, P, cmax) on every rank:
        rank r's slab is out[r, :, : c1 - c0].  Slabs are padded to the largest one so that one fixed-size collective
        suffices; returns (out, bounds)."""
        bounds = all_bounds(ncells, self.world, align)
        cmax = max(b - a for a, b in bounds)
        P, cl = local.shape
        send = local
        if cl != cmax:
            send = self.dev.zeros((P, cmax), local.dtype)
            isz = np.dtype(local.dtype).itemsize
            self.dev.copy2d(send.ptr, cmax * isz, local.ptr, cl * isz, cl * isz, P, "d2d", blocking=False)
        if out is None:
            out = self.dev.empty((self.world, P, cmax), local.dtype)
        self.all_gather(send, out)
        return out, bounds

    def close(self) -> None:
        if self.handle:
            self.dev.lib.xh_comm_destroy(self.handle)
            self.handle = None
