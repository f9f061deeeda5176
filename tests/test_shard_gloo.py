"""N > 1 path on CPU: world_size-2 gloo processes shard the cell axis, compute their slab (the oracle stands in for
the per-rank kernels here — there is no GPU), all-gather the reduced outputs and must reproduce the unsharded
result exactly."""

import os
import subprocess
import sys

import numpy as np
import pytest

from xclim_amd.shard import all_bounds, shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
import numpy as np
sys.path.insert(0, {root!r})
import torch.distributed as dist
import torch
from oracle import indices as oidx, synth
from oracle.timeutil import OTime
from xclim_amd.shard import all_bounds, shard_bounds


def gather_cells(local, ncells, align=4):
    # the exchange of Comm.gather_cells (slabs padded to the largest, one fixed-size all-gather) over gloo: torch lives
    # here, with the test, not in the product package
    world = dist.get_world_size()
    bounds = all_bounds(ncells, world, align)
    cmax = max(b - a for a, b in bounds)
    t = torch.from_numpy(np.ascontiguousarray(local))
    pad = torch.zeros((t.shape[0], cmax), dtype=t.dtype)
    pad[:, : t.shape[1]] = t
    out = torch.empty((world, t.shape[0], cmax), dtype=t.dtype)
    dist.all_gather_into_tensor(out.view(world * t.shape[0], cmax), pad)
    return torch.cat([out[r, :, : b - a] for r, (a, b) in enumerate(bounds)], dim=1).numpy()


dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
T, C = 730, 203
ot = OTime.noleap(2001, T)
zero = np.zeros(T, np.float32)
c0, c1 = shard_bounds(C, world, rank)
pr = synth.fill_synthetic(T, np.arange(c0, c1), 1, 3, zero, 40.0 / 86400.0, 0.3)  # shard of the global field
local = oidx.maximum_consecutive_dry_days(pr, 1.0 / 86400.0, ot, "YS").astype(np.float32)
full = gather_cells(local, C)
if rank == 0:
    ref = oidx.maximum_consecutive_dry_days(synth.fill_synthetic(T, np.arange(C), 1, 3, zero, 40.0 / 86400.0, 0.3),
                                            1.0 / 86400.0, ot, "YS")
    np.testing.assert_array_equal(full, ref)
    print("SHARD-OK", full.shape)
dist.barrier()
dist.destroy_process_group()
"""


def test_shard_bounds_cover_and_align():
    for C in (1, 3, 4, 17, 203, 1036800):
        for world in (1, 2, 3, 8):
            b = all_bounds(C, world)
            assert b[0][0] == 0 and b[-1][1] == C
            for (a0, a1), (b0, b1) in zip(b[:-1], b[1:]):
                assert a1 == b0 and a0 <= a1
            assert all(a % 4 == 0 for a, _ in b if a < C)
    assert shard_bounds(1036800, 8, 3) == (388800, 518400)
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def test_two_rank_gloo_gather(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT))
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29517", str(script)]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    assert "SHARD-OK (2, 203)" in res.stdout


_RDV_WORKER = r"""
import os, sys, time
sys.path.insert(0, sys.argv[1])
from xclim_amd.shard import _rendezvous_path, exchange_unique_id
rank = int(os.environ["RANK"])
if rank == 0:
    time.sleep(0.3)            # the other ranks are already polling when the id appears
uid = exchange_unique_id(_rendezvous_path(), rank, lambda: bytes(range(128)), 128, timeout_s=20.0)
sys.stdout.write(uid.hex())
"""


def test_unique_id_file_rendezvous_three_ranks(tmp_path):
    """The only host-side logic of a multi-GPU launch that is not RCCL: rank 0 publishes the 128-byte id atomically, the
    other ranks (started earlier, polling) read the complete id; the file name is shared inside one launch (same
    MASTER_ADDR / MASTER_PORT / parent process) and differs between launches."""
    script = tmp_path / "rdv.py"
    script.write_text(_RDV_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, XH_RENDEZVOUS_DIR=str(tmp_path), MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="3")
    procs = [subprocess.Popen([sys.executable, str(script), root], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE)
             for r in (1, 2, 0)]
    outs = [p.communicate(timeout=60)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs)
    assert outs[0] == outs[1] == outs[2] == bytes(range(128)).hex()
    # a different port (another launch) uses another file
    from xclim_amd.shard import _rendezvous_path

    a = _rendezvous_path()
    os.environ["MASTER_PORT"], old = "29534", os.environ.get("MASTER_PORT")
    try:
        os.environ["MASTER_PORT"] = "1"
        b = _rendezvous_path()
        os.environ["MASTER_PORT"] = "2"
        c = _rendezvous_path()
    finally:
        if old is None:
            os.environ.pop("MASTER_PORT", None)
        else:
            os.environ["MASTER_PORT"] = old
    assert b != c and a is not None
    # a rank that never gets an id gives up with a clear error
    with pytest.raises(TimeoutError):
        from xclim_amd.shard import exchange_unique_id

        exchange_unique_id(str(tmp_path / "never.id"), 1, None, 128, timeout_s=0.2)


_FC_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1])
from xclim_amd.shard import FileComm
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
fc = FileComm(None, world, rank, sys.argv[2])
vals = []
for step in range(3):
    vals.append(float(fc.allreduce([10.0 * step + rank], "max")[0]))
    vals.append(float(fc.allreduce([1.0 + rank], "sum")[0]))
sys.stdout.write(repr(vals))
"""


def test_file_comm_scalar_reductions(tmp_path):
    """FileComm (bench.py's stand-in when RCCL cannot be initialised): max / sum over three ranks through files."""
    script = tmp_path / "fc.py"
    script.write_text(_FC_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="3")
    procs = [subprocess.Popen([sys.executable, str(script), root, str(tmp_path / "fc")], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE) for r in range(3)]
    outs = [p.communicate(timeout=60)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs)
    expect = repr([2.0, 6.0, 12.0, 6.0, 22.0, 6.0])
    assert outs == [expect, expect, expect]


@pytest.mark.parametrize("workload", ["c2", "c5"])
def test_bench_two_ranks_plumbing_without_gpus(tmp_path, workload):
    """``bench.py --gpus 2`` exactly as the driver launches it (torch.distributed.run, one rank per GPU), on a machine
    without GPUs: a host-memory mock device (tools/mock_device.py, no-op kernels) and XH_BENCH_NO_RCCL=1 (file barriers, no
    exchange) — environment handling, rendezvous fallback, barriers, max-over-ranks timing and the ONE JSON line on
    rank 0 must work before the first 8-GPU run.  The numbers are meaningless and labelled so."""
    import json

    env = dict(os.environ, OMP_NUM_THREADS="1", XH_BENCH_MOCK_DEVICE="1", XH_BENCH_NO_RCCL="1", XH_RENDEZVOUS_DIR=str(tmp_path))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29541" if workload == "c2" else "29543", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--grid", "365x4x8" if workload == "c2" else "1095x4x8", "--no-cpu", "--no-extra", "--workload", workload]
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["data"].startswith("mock") and "NO exchange" in rec["config"]["sharding"]
    assert rec["value"] > 0 and rec["roofline"]["bound"] == "hbm" and rec["unit"] == "cell-timesteps/s"
    if workload == "c5":  # config 5's per-GPU slab flow: tx90p + EQM train + adjust, one gather buffer per step
        assert rec["config"]["grid_per_gpu"] == [1095, 4, 8] and "configs[4]" in rec["config"]["workload"]
