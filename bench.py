#!/usr/bin/env python
"""bench.py — the reference's headline workload on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--no-cpu] [--no-extra] [--no-full]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ... [--workload c5]

Workload (BASELINE.json configs[1]): tx90p = percentile_doy(window 5, per 90) + threshold_count(">", per-doy fp64
threshold, freq "YS") + MissingAny mask on a synthetic 365 x 1440 x 720 fp32 tasmax grid (noleap, time-major), all
resident in HBM.  One "step" = one full pass of that chain through the C ABI.  With N > 1 every rank owns one
1440x720 slab of an N-times larger grid (lat split, weak scaling) and the reduced (P, C) outputs are all-gathered
over RCCL at the end of each step (the only exchange on this path) — through the C ABI (xh_comm_*), no torch.
--workload c5 runs BASELINE configs[4] instead (tx90p + EQM on the 30-year 360x1440 slab one of 8 GPUs owns).

Prints ONE JSON line (rank 0): metric = grid-cells x timesteps / s (whole job), plus
  roofline     — dominant kernel (percentile_doy), algorithmic bytes / HIP-event time on the kernel's own stream
  cpu_baseline — the numpy oracle (a port: the reference stack is not installable) on a bounded lat-band sample
  extra        — the two other north-star workloads at the same grid (cdd run-length, EQM train+adjust)
Nothing here imports torch: with N > 1 the launcher (torch.distributed.run, or anything that sets RANK / LOCAL_RANK /
WORLD_SIZE) only starts the ranks; rendezvous and the collective go through xclim_amd.shard.Comm -> libxclimhip.so -> RCCL.
"""

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s best measured copy)


# untimed steps before the W warm-up steps of the contract: the first steps after a process start (or after the box sat
# idle) ran up to 40 % slower on one of ~10 boxes of the pool; they are not part of the measurement either way
WAKEUP_STEPS = 30

PMC_TABLES = ("profiles/r06/pmc_hbm_traffic.json", "profiles/r05/pmc_hbm_traffic.json", "profiles/r04/pmc_hbm_traffic.json", "profiles/r03/pmc_hbm_traffic.json", "profiles/r02/pmc_hbm_traffic.json")


def pmc_traffic(kernel, grid):
    """(HBM bytes per launch of `kernel`, source file) from the COMMITTED rocprofv3 PMC passes of this command
    (FETCH_SIZE / WRITE_SIZE collected in separate runs by tools/profile_bench.sh, corrected per MI355X_MICROARCH.md by
    tools/summarize_pmc.py) — counters cannot be collected inside a timed run, so this number is replayed from the file
    named in ``traffic_source``, not measured now.  Only valid for the grid the profile was taken on (365x1440x720)."""
    if tuple(grid) != (365, 1440, 720):
        return None, None
    for rel in PMC_TABLES:
        try:
            table = json.load(open(os.path.join(ROOT, rel)))
        except (OSError, ValueError):
            continue
        for name in (kernel, kernel[:-1] + ", false>"):  # the kernel gained a trailing COUNT template flag (false = this path)
            if name in table:
                return table[name]["hbm_bytes_per_launch"], rel
    return None, None


def train365_traffic(grid):
    """HBM bytes of one xh_eqm_train at 365 steps: two register-sort launches + the correction kernel (round 5: FETCH_SIZE
    is calibrated on the register sort's load pattern, tools/regsort_ubench.hip)."""
    a, _ = pmc_traffic("k_select_regsort<183, 360>", (grid[0], 1440, grid[1] // 1440) if grid[1] == 1440 * 720 else (0, 0, 0))
    b, _ = pmc_traffic("k_correction_fix", (grid[0], 1440, grid[1] // 1440) if grid[1] == 1440 * 720 else (0, 0, 0))
    return None if a is None or b is None else 2 * a + b


def hbm_roofline(nbytes, ms, kernel=None, **more):
    """The roofline object of the JSON contract for an HBM-bound launch (or chain): algorithmic bytes / HIP-event time."""
    gbs = nbytes / (ms * 1e-3) / 1e9
    r = {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
         "algorithmic_bytes": nbytes, "ms": ms}
    if kernel:
        r["kernel"] = kernel
    r.update(more)
    return r


def valu_bound(kernel):
    """The VALU-issue view of a kernel from the committed SQ + GRBM PMC passes (tools/gpu_r03_prof.sh ->
    tools/summarize_sq.py -> profiles/r03/valu_busy.json): VALU instructions per SIMD x the measured issue cost of a
    wave64 instruction (2.5 cycles of the ~2.4 GHz clock for add / xor / mov, 4.3 for min / max / cmp:
    profiles/r03/valu_ubench.txt, bank_ubench.txt) over the kernel's cycles.  busy_hi near 1 = no faster without issuing fewer instructions.  None when not profiled."""
    for rel in ("profiles/r06/valu_busy.json", "profiles/r05/valu_busy.json", "profiles/r04/valu_busy.json", "profiles/r03/valu_busy.json"):
        try:
            table = json.load(open(os.path.join(ROOT, rel)))
        except (OSError, ValueError):
            continue
        for name, v in table.items():
            if name.startswith(kernel):
                return {"bound": "valu", "kernel": name, "valu_insts_per_simd": v["valu_insts_per_simd"], "cycles": v["cycles"],
                        "busy_at_2.5_cycles": v["busy_lo"], "busy_at_4.3_cycles": v["busy_hi"], "source": rel}
    return None


PMC_30YR = "profiles/r06/pmc_hbm_traffic_30yr.json"


def pmc_traffic_30yr(*kernels):
    """Sum of the HBM bytes per launch of the 30-year kernels from the committed PMC passes of the full configurations
    (PMC_30YR; only kernels that run at one size there), or None (a kernel that is absent or not calibrated)."""
    try:
        table = json.load(open(os.path.join(ROOT, PMC_30YR)))
    except (OSError, ValueError):
        return None
    tot = 0.0
    for k in kernels:
        if k not in table or table[k].get("hbm_bytes_per_launch") is None:
            return None
        tot += table[k]["hbm_bytes_per_launch"]
    return tot


def seasonal_base(T, mean=288.0, amp=12.0, phase=100.0, period=365.0):
    """Seasonal cycle table handed to xh_fill_synthetic (same formula as oracle/synth.py, which only the CPU baseline
    leg imports: the measured path does not touch oracle/)."""
    t = np.arange(T, dtype=np.float64)
    return (mean + amp * np.sin(2.0 * np.pi * (t - phase) / period)).astype(np.float32)


def event_time(dev, fn, reps):
    """Average duration (ms) of `fn` over `reps` launches, HIP events on the kernel's own stream."""
    fn()
    dev.sync()
    dev.timer_start()
    for _ in range(reps):
        fn()
    return dev.timer_stop() / reps


def chain_event_times(dev, fns, reps):
    """Average duration (ms) of every kernel of a chain, each launch bracketed by HIP events, the chain run `reps` times."""
    for f in fns:
        f()
    dev.sync()
    tot = [0.0] * len(fns)
    for _ in range(reps):
        for i, f in enumerate(fns):
            dev.timer_start()
            f()
            tot[i] += dev.timer_stop()
    return [t / reps for t in tot]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--grid", type=str, default=None,
                    help="TxYxX per GPU; default 365x1440x720 (c2) / 10950x360x1440 (c5, the slab config 5 gives one of 8 GPUs)")
    ap.add_argument("--workload", choices=["c2", "c5"], default="c2",
                    help="c2 (default): BASELINE configs[1], tx90p on 365x1440x720 per GPU; c5: configs[4], tx90p + EQM on a "
                         "30-year 360x1440 slab per GPU")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-full", action="store_true", help="skip the 3650-step / 30-year extras (they allocate up to 182 GB)")
    ap.add_argument("--no-long", action="store_true", help="skip the 55 152-step extras (PMC passes: per-kernel means must not mix sizes)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        sys.exit("--gpus N > 1 must be launched with one rank per GPU (python -m torch.distributed.run ... bench.py --gpus N)")
    # XH_BENCH_FORCE_DIST=1: go through the RCCL path even with one rank (single-GPU validation of the exchange code)
    use_dist = world > 1 or bool(os.environ.get("XH_BENCH_FORCE_DIST"))

    from xclim_amd import kernels as K
    from xclim_amd._capi import Device
    from xclim_amd.timeaxis import TimeAxis

    mock = bool(os.environ.get("XH_BENCH_MOCK_DEVICE"))
    if mock:  # plumbing test of a multi-rank launch on a box without GPUs (tests/test_shard_gloo.py): no-op kernels
        from tools.mock_device import MockDevice as Device  # noqa: F811
    dev = Device(local_rank)
    comm = None
    if use_dist:
        # RCCL through the C ABI (xh_comm_*, include/xclim_hip.h): no torch anywhere in this process.  The launcher only
        # provides RANK / LOCAL_RANK / WORLD_SIZE; the unique id travels through a node-local file (xclim_amd/shard.py).
        from xclim_amd.shard import Comm, FileComm, _rendezvous_path

        try:
            if os.environ.get("XH_BENCH_NO_RCCL"):  # exercise the fallback below
                raise RuntimeError("XH_BENCH_NO_RCCL is set")
            comm = Comm.from_env(dev, timeout_s=120.0)
        except Exception as exc:  # noqa: BLE001 — RCCL could not be brought up: report the sharded throughput without the exchange
            sys.stderr.write(f"[bench] rank {rank}: RCCL unavailable ({exc}); falling back to file barriers, NO gather\n")
            comm = FileComm(dev, world, rank, _rendezvous_path() + ".fc", reason=str(exc)[:200])
    if args.workload == "c5":
        return bench_config5(args, dev, K, comm, world, rank)
    T, Y, X = (int(v) for v in (args.grid or "365x1440x720").split("x"))
    C = Y * X
    ta = TimeAxis.daily("2001-01-01", T, "noleap")
    tb, years, doys = ta.doy_table()
    seg, _ = ta.segments("YS")
    P = len(seg) - 1
    expected = ta.expected_count("YS")
    tidx_h = np.searchsorted(doys, ta.doy).astype(np.int32)
    base = seasonal_base(T)
    tasmax = K.fill_synthetic(dev, T, C, 0, 2, base, 3.0, cell0=rank * C)
    per = dev.empty((1, len(doys), C), np.float64)
    cnt, val = dev.empty((P, C), np.int32), dev.empty((P, C), np.int32)
    res = dev.empty((P, C), np.float64)
    tidx = dev.to_device(tidx_h)
    table = per.reshape(len(doys), C)
    overlap = use_dist and not os.environ.get("XH_BENCH_SYNC_GATHER")
    if use_dist:
        # Two result buffers: the all-gather of step k runs on the communicator's stream while step k + 1 computes (the
        # only exchange of the path; nothing downstream of it inside a step): xh_comm_allgather(slot) orders
        # itself after the kernels queued so far, xh_comm_fence(slot) protects the buffer before it is rewritten; the
        # host never waits inside a step.
        res_bufs = [dev.empty((P, C), np.float64) for _ in range(2)]
        gathered = [dev.empty((world, P, C), np.float64) for _ in range(2)]
        res = res_bufs[0]

    def k_pdoy():
        K.percentile_doy(dev, tasmax, tb, 5, [90.0], out=per)

    def k_count():
        K.threshold_count(dev, tasmax, ">", seg, doy_table=table, tidx=tidx, out=(cnt, val))

    def k_mask(out=None):
        K.apply_missing_mask(dev, cnt, val, expected, out=res if out is None else out)

    nstep = [0]

    def step():
        b = nstep[0] % 2
        nstep[0] += 1
        if not use_dist:
            k_pdoy(); k_count(); k_mask()
            return
        if overlap and nstep[0] > 2:
            comm.fence(b)  # the gather of two steps ago has read res_bufs[b]
        k_pdoy(); k_count(); k_mask(res_bufs[b])
        comm.all_gather(res_bufs[b], gathered[b], slot=b if overlap else -1)

    def fence():
        dev.sync()
        if use_dist:
            comm.sync()
            comm.barrier()

    for _ in range(WAKEUP_STEPS + args.warmup):  # (clocks / caches / pools settle; all untimed)
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if use_dist:
        lastb = (nstep[0] - 1) % 2  # the slab this rank contributed must have arrived in the gathered field
        mine = dev.wrap(gathered[lastb].ptr + rank * res_bufs[lastb].nbytes, (P, C), np.float64).get()
        if not np.array_equal(mine, res_bufs[lastb].get(), equal_nan=True):
            sys.exit("bench: the all-gathered field does not hold this rank's result")
        dt = float(comm.allreduce([dt], "max")[0])
    units_step = float(T) * C * world
    value = units_step * args.steps / dt

    # ---- roofline of the dominant kernel (per launch, this rank) ----
    # Each kernel is timed with HIP events on the kernel's own stream INSIDE the chain (pdoy -> count -> mask repeated
    # like the timed steps): the write-heavy percentile kernel runs ~10 % slower when it is launched back-to-back with
    # itself (sustained 3 GB writes) than in its place in the chain, and the chain is what `value` measures.
    reps = max(5, min(args.steps, 20))
    ms_pdoy, ms_count, ms_mask = chain_event_times(dev, [k_pdoy, k_count, k_mask], reps)
    E = float(T) * C
    D = float(len(doys))
    bytes_pdoy = 4 * E + 8 * D * C  # read x once, write (D, C) fp64
    bytes_count = 4 * E + 8 * D * C + 8 * P * C  # read x, read per-doy fp64 table, write count+valid int32
    roofline = {
        "bound": "hbm",
        "kernel": "k_pdoy_slide<5, 4> (xh_percentile_doy)",
        "achieved": bytes_pdoy / (ms_pdoy * 1e-3) / 1e9,
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": bytes_pdoy / (ms_pdoy * 1e-3) / 1e9 / HBM_PEAK_GBS,
        "traffic": pmc_traffic("k_pdoy_slide<5, 4>", (T, Y, X))[0],
        "traffic_source": pmc_traffic("k_pdoy_slide<5, 4>", (T, Y, X))[1],
        "algorithmic_bytes": bytes_pdoy,
        "ms": ms_pdoy,
        "chain": {
            "percentile_doy": {"ms": ms_pdoy, "GB/s": bytes_pdoy / ms_pdoy / 1e6},
            "threshold_count_doy": {"ms": ms_count, "GB/s": bytes_count / ms_count / 1e6},
            "missing_mask": {"ms": ms_mask},
            "tx90p_unfused_total": {"ms": ms_pdoy + ms_count + ms_mask,
                                    "GB/s": (bytes_pdoy + bytes_count) / (ms_pdoy + ms_count + ms_mask) / 1e6},
        },
    }

    extra = {}
    if not args.no_extra and rank == 0 and world == 1:
        extra = bench_extra(dev, K, ta, T, C, seg, P, tasmax, tb, per, len(doys), full_configs=not args.no_full, long_series=not args.no_long)

    cpu = None
    if not args.no_cpu and rank == 0 and world == 1:
        cpu = cpu_baseline(T, Y, X)

    if rank == 0:
        line = {
            "metric": "grid-cells x timesteps / s (tx90p = percentile_doy + threshold_count + missing mask)",
            "value": value,
            "unit": "cell-timesteps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "mock (no GPU: plumbing test, the numbers mean nothing)" if mock else "synthetic",
            "config": {"workload": f"tx90p (percentile_doy window 5 per 90 + threshold_count > + MissingAny) on {T}x{Y}x{X} fp32 "
                                   f"per GPU, noleap, freq YS, time-major, resident in HBM",
                       "grid_per_gpu": [T, Y, X], "untimed_wakeup_steps": WAKEUP_STEPS, "sharding": ("lat slabs, one per rank; RCCL (xh_comm_allgather, C ABI) all_gather of (P,C) fp64" + (", overlapped with the next step" if overlap else ""))
                       if getattr(comm, "kind", "rccl") != "file" else f"lat slabs, one per rank; NO exchange (RCCL unavailable: {comm.reason})"},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "extra": extra,
        }
        print(json.dumps(line))
    if use_dist:
        comm.barrier()
        comm.close()


def bench_config5(args, dev, K, comm, world, rank):
    """--workload c5: BASELINE configs[4] — tx90p AND EQM train + adjust on 30 noleap years of a 2880 x 1440 grid cut into
    8 lat slabs of 360 x 1440 cells.  Every rank owns ONE such slab (cells [rank * 518400, ...) of the global counter-based
    field; with 8 ranks this is exactly config 5, with fewer it is the same per-GPU work on a smaller grid: weak scaling).
    One step = percentile_doy(window 5, per 90) -> threshold_count(">") -> missing mask, eqm_train(ref, hist, 20 nodes,
    "+") -> eqm_adjust(sim, nearest, constant), then the ONE exchange: all-gather of the (30, C) masked counts and of the
    (20, C) af / hist_q nodes (scen stays sharded)."""
    from xclim_amd.timeaxis import TimeAxis

    T, Y, X = (int(v) for v in (args.grid or "10950x360x1440").split("x"))  # (--grid: plumbing tests on small slabs)
    C = Y * X
    cell0 = rank * C
    ta = TimeAxis.daily("1981-01-01", T, "noleap")
    tb, years, doys = ta.doy_table()
    seg, _ = ta.segments("YS")
    P, D = len(seg) - 1, len(doys)
    expected = ta.expected_count("YS")
    base = seasonal_base(T)
    tas = K.fill_synthetic(dev, T, C, 0, 2, base, 3.0, cell0=cell0)
    ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0, cell0=cell0)
    hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3, cell0=cell0)
    sim = K.fill_synthetic(dev, T, C, 0, 6, base + np.float32(3.5), 3.3, cell0=cell0)
    scen = dev.empty((T, C), np.float32)
    per = dev.empty((1, D, C), np.float64)
    cnt, val = dev.empty((P, C), np.int32), dev.empty((P, C), np.int32)
    tidx = dev.to_device(np.searchsorted(doys, ta.doy).astype(np.int32))
    q = (np.arange(20) + 0.5) / 20
    # one send buffer: (P, C) float64 counts followed by (2, 20, C) float32 nodes -> a single collective per step
    nb_res, nb_nodes = P * C * 8, 2 * 20 * C * 4
    send = dev.empty((nb_res + nb_nodes,), np.uint8)
    res = dev.wrap(send.ptr, (P, C), np.float64)
    af = dev.wrap(send.ptr + nb_res, (20, C), np.float32)
    hq = dev.wrap(send.ptr + nb_res + 20 * C * 4, (20, C), np.float32)
    recv = dev.empty((world, nb_res + nb_nodes), np.uint8) if comm else None

    def step():
        K.percentile_doy(dev, tas, tb, 5, [90.0], out=per)
        K.threshold_count(dev, tas, ">", seg, doy_table=per.reshape(D, C), tidx=tidx, out=(cnt, val))
        K.apply_missing_mask(dev, cnt, val, expected, out=res)
        K.eqm_train(dev, ref, hist, q, "+", out=(af, hq))
        K.eqm_adjust(dev, sim, af, hq, "+", "nearest", "constant", out=scen)
        if comm:
            comm.all_gather(send, recv, slot=-1)

    def fence():
        dev.sync()
        if comm:
            comm.sync()
            comm.barrier()

    for _ in range(WAKEUP_STEPS + args.warmup):  # (clocks / caches / pools settle; all untimed)
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if comm:
        mine = dev.wrap(recv.ptr + rank * send.nbytes, (send.nbytes,), np.uint8).get()
        if not np.array_equal(mine, send.get()):
            sys.exit("bench: the all-gathered field does not hold this rank's result")
        dt = float(comm.allreduce([dt], "max")[0])
    E = float(T) * C
    bytes_step = (2 * (4 * E + 8 * D * C) + 8 * P * C) + 16 * E  # tx90p unfused + EQM train + adjust
    if rank == 0:
        print(json.dumps({
            "metric": "grid-cells x timesteps / s (tx90p + EQM train + adjust, 30-year daily series)",
            "value": E * world * args.steps / dt, "unit": "cell-timesteps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32",
            "data": "mock (no GPU: plumbing test, the numbers mean nothing)" if os.environ.get("XH_BENCH_MOCK_DEVICE") else "synthetic",
            "config": {"workload": f"BASELINE configs[4]: tx90p + EmpiricalQuantileMapping train+adjust on {T} x {Y} x {X} fp32 "
                                   "per GPU (one of the 8 lat slabs of the 2880 x 1440 grid), noleap, resident in HBM",
                       "grid_per_gpu": [T, Y, X],
                       "sharding": "lat slabs, one per rank; one RCCL all_gather of (P,C) fp64 counts + (2,20,C) fp32 nodes per step"
                       if getattr(comm, "kind", "rccl") != "file" else f"lat slabs, one per rank; NO exchange (RCCL unavailable: {comm.reason})"},
            "roofline": {"bound": "hbm", "kernel": "whole step (5 kernels chains)", "achieved": bytes_step / (dt / args.steps) / 1e9,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bytes_step / (dt / args.steps) / 1e9 / HBM_PEAK_GBS,
                         "traffic": None, "algorithmic_bytes": bytes_step},
            "cpu_baseline": None}))
    if comm:
        comm.barrier()
        comm.close()


def bench_extra(dev, K, ta, T, C, seg, P, tasmax, tb, per, D, full_configs=True, long_series=True):
    """The other two north-star workloads on the same 365 x 1440 x 720 grid (HIP-event times, one GPU)."""
    out = {}
    E = float(T) * C
    # --- percentile_doy at a percentile that needs the full sort + lerp (per = 90 with 5 samples clips to the sample
    #     maximum, utl:443-447, and takes the no-sort fast path; per = 50 does not)
    ms50 = event_time(dev, lambda: K.percentile_doy(dev, tasmax, tb, 5, [50.0], out=per), 10)
    b50 = 4 * E + 8 * D * C
    out["percentile_doy_per50"] = {"ms": ms50, "GB/s": b50 / ms50 / 1e6, "frac": b50 / ms50 / 1e6 / HBM_PEAK_GBS}
    # --- tx90p with the percentile table fused away (xh_percentile_doy_count): same counts as the timed chain, the
    #     (D, C) fp64 table is never written / re-read.  Reported next to the API-faithful two-step chain, not as `value`.
    cntf, valf = dev.empty((P, C), np.int32), dev.empty((P, C), np.int32)
    period = (np.searchsorted(seg, tb[0], side="right") - 1).astype(np.int32)
    if K.percentile_doy_count(dev, tasmax, tb, 5, 90.0, ">", period, P, out=(cntf, valf)) is not None:
        msf = event_time(dev, lambda: K.percentile_doy_count(dev, tasmax, tb, 5, 90.0, ">", period, P, out=(cntf, valf)), 10)
        bf = 4 * E + 8 * P * C
        out["tx90p_fused"] = {"ms": msf, "GB/s": bf / msf / 1e6, "frac": bf / msf / 1e6 / HBM_PEAK_GBS,
                              "cell-timesteps/s": E / (msf * 1e-3), "algorithmic_bytes": bf,
                              "note": "percentile_doy + threshold_count in one kernel (k_pdoy_slide<5,4,COUNT>)"}
    # --- maximum_consecutive_dry_days: fused compare + run-length max + valid count ---
    pr = K.fill_synthetic(dev, T, C, 1, 3, np.zeros(T, np.float32), 40.0 / 86400.0, 0.3)
    o, v = dev.empty((P, C), np.float32), dev.empty((P, C), np.int32)
    ms = event_time(dev, lambda: K.run_stats(dev, pr, "max", 1, seg, cut=True, fused_op="<", thresh=1.0 / 86400.0,
                                             out=(o, v)), 10)
    b = 4 * E + 8 * P * C
    out["cdd_rle_365"] = {"ms": ms, "GB/s": b / ms / 1e6, "frac": b / ms / 1e6 / HBM_PEAK_GBS,
                          "cell-timesteps/s": E / (ms * 1e-3), "algorithmic_bytes": b,
                          "roofline": hbm_roofline(b, ms, "k_run_max_fused<4, LT> (xh_run_stats)")}
    del pr
    # --- EQM train + adjust (nquantiles 20, "+", nearest, constant) ---
    base = seasonal_base(T)
    ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0)
    hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
    sim = K.fill_synthetic(dev, T, C, 0, 6, base + np.float32(3.5), 3.3)
    q = (np.arange(20) + 0.5) / 20
    af, hq = dev.empty((20, C), np.float32), dev.empty((20, C), np.float32)
    scen = dev.empty((T, C), np.float32)
    ms_tr = event_time(dev, lambda: K.eqm_train(dev, ref, hist, q, "+", out=(af, hq)), 3)
    ms_ad = event_time(dev, lambda: K.eqm_adjust(dev, sim, af, hq, "+", "nearest", "constant", out=scen), 10)
    b_tr = 8 * E + 8 * 20 * C
    b_ad = 8 * E + 8 * 20 * C
    out["eqm_train_365"] = {"ms": ms_tr, "GB/s": b_tr / ms_tr / 1e6, "frac": b_tr / ms_tr / 1e6 / HBM_PEAK_GBS,
                            "algorithmic_bytes": b_tr, "note": "time-major input, read in place (register sorting network)",
                            "roofline": hbm_roofline(b_tr, ms_tr, "2 x k_select_regsort<183, 360> + k_correction (xh_eqm_train)",
                                                     traffic=train365_traffic((T, C)), traffic_source=PMC_TABLES[0]),
                            "roofline_valu": dict(valu_bound("k_select_regsort") or {}, note=(
                                "VERDICT r4 #1d: the 50 % target of this leg is RETIRED with the instruction-count bound — 275 K VALU per "
                                "SIMD and launch, almost all v_min_u32 / v_max_u32 at 4.3 cycles: >= 0.46 ms per array at 100 % issue, with "
                                "the adjust kernel <= 0.48 of the HBM peak for train + adjust (DESIGN.md 7 item 3); a v_min3 / v_med3 / "
                                "v_max3 network was priced at <= 1.25x fewer sort instructions (-> <= 0.40) and not built")) or None}
    out["eqm_adjust_365"] = {"ms": ms_ad, "GB/s": b_ad / ms_ad / 1e6, "frac": b_ad / ms_ad / 1e6 / HBM_PEAK_GBS,
                             "algorithmic_bytes": b_ad, "roofline": hbm_roofline(b_ad, ms_ad, "k_eqm_adjust<20, 0> (xh_eqm_adjust)")}
    out["eqm_train_adjust_365"] = {"ms": ms_tr + ms_ad, "GB/s": (b_tr + b_ad) / (ms_tr + ms_ad) / 1e6,
                                   "frac": (b_tr + b_ad) / (ms_tr + ms_ad) / 1e6 / HBM_PEAK_GBS,
                                   "cell-timesteps/s": E / ((ms_tr + ms_ad) * 1e-3),
                                   "roofline": hbm_roofline(b_tr + b_ad, ms_tr + ms_ad, "xh_eqm_train + xh_eqm_adjust")}
    # --- QuantileDeltaMapping.adjust (nearest, constant; the factors of the EQM training above) ---
    ms_qd = event_time(dev, lambda: K.qdm_adjust(dev, sim, af, q, "+", "nearest", "constant", out=scen), 5)
    out["qdm_adjust_365"] = {"ms": ms_qd, "GB/s": 8 * E / ms_qd / 1e6, "frac": 8 * E / ms_qd / 1e6 / HBM_PEAK_GBS,
                             "algorithmic_bytes": 8 * E,
                             "roofline": hbm_roofline(8 * E, ms_qd, "k_qdm_regsort<183, 360> + k_cut_classify (xh_qdm_adjust) + k_qdm_columns for columns with ties"),
                             "roofline_valu": valu_bound("k_qdm_regsort")}
    for a in (ref, hist, sim, scen, af, hq):
        a.free()
    out["adapter_e2e"] = bench_adapter_e2e(dev, K, ta, T, C, tasmax)
    if full_configs:
        out.update(bench_full_configs(dev, K, C, long_series))
    return out


def bench_adapter_e2e(dev, K, ta, T, C, tasmax):
    """What the drop-in layer costs END TO END when the field lives in host memory (reported beside, never
    instead of, the HBM roofline): numpy in -> numpy out through the host mirrors the xarray wrappers call after
    unwrapping (xr_adapter._tfirst -> xclim_amd.calendar / indices), i.e. upload + kernels + download, wall clock, best of
    3; once from pageable memory (what ``DataArray.values`` is) and once from page-locked memory."""
    from xclim_amd import indices as xi
    from xclim_amd.calendar import percentile_doy

    host = tasmax.get()                                   # pageable
    pinned = dev.pinned_empty(host.shape, np.float32)
    pinned[...] = host
    pr_host = K.fill_synthetic(dev, T, C, 1, 3, np.zeros(T, np.float32), 40.0 / 86400.0, 0.3).get()
    E = float(T) * C

    def best(fn, n=3):
        ts = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            dev.sync()
            ts.append(time.perf_counter() - t0)
        return min(ts) * 1e3

    def tx90p(x):
        # one keep_inputs scope per timed call (what patch.install() opens around an Indicator call, or a user around the
        # two calls): the field crosses PCIe once per call, never zero times — the scope's exit drops the device copy
        with dev.keep_inputs():
            per = percentile_doy(x, ta, window=5, per=90.0, device=dev)
            return xi.tx90p(x, per, ta, freq="YS", device=dev)

    def uploads_of(fn, nbytes):
        trace = dev.start_trace()
        fn()
        dev.stop_trace()
        return sum(1 for n, a in trace if n == "h2d" and a[0] == nbytes)

    res = {"note": "host-resident float32 field -> numpy result through the host mirrors (H2D + kernels + D2H); PCIe-inclusive, "
                   "NOT the headline value; percentile_doy and the count read the same host buffer inside one keep_inputs scope: ONE transfer "
                   "(Device.resident; outside a scope every call uploads its inputs)",
           "field_GB": host.nbytes / 1e9}
    for label, x in (("pageable", host), ("pinned", pinned)):
        ms = best(lambda: tx90p(x))
        n_up = uploads_of(lambda: tx90p(x), host.nbytes)
        res[f"tx90p_{label}"] = {"ms": ms, "cell-timesteps/s": E / (ms * 1e-3), "uploads": n_up,
                                 "host_link_GB/s": n_up * host.nbytes / ms / 1e6}
    dev.forget_inputs()
    ms = best(lambda: (dev.forget_inputs(), xi.maximum_consecutive_dry_days(pr_host, 1.0 / 86400.0, ta, freq="YS", device=dev)))
    res["cdd_pageable"] = {"ms": ms, "cell-timesteps/s": E / (ms * 1e-3), "uploads": 1, "host_link_GB/s": pr_host.nbytes / ms / 1e6}
    return res


def bench_full_configs(dev, K, C, long_series=True):
    """BASELINE configs[2] (cdd on 3650 steps), the 30-year tx90p of configs[4] and configs[3] (EQM on 30 years) at
    their own size on this GPU's grid (same byte formulas as SURVEY 8d / tools/bench_configs.py), inputs generated on the
    device.  Up to four 45.4 GB arrays are resident at once."""
    from xclim_amd.timeaxis import TimeAxis

    out = {}
    # ---- configs[2]: maximum_consecutive_dry_days, 3650 steps
    T = 3650
    ta = TimeAxis.daily("2001-01-01", T, "noleap")
    seg, _ = ta.segments("YS")
    P = len(seg) - 1
    pr = K.fill_synthetic(dev, T, C, 1, 3, np.zeros(T, np.float32), 40.0 / 86400.0, 0.3)
    o, v = dev.empty((P, C), np.float32), dev.empty((P, C), np.int32)
    ms = event_time(dev, lambda: K.run_stats(dev, pr, "max", 1, seg, cut=True, fused_op="<", thresh=1.0 / 86400.0,
                                             out=(o, v)), 5)
    E = float(T) * C
    b = 4 * E + 8 * P * C
    out["cdd_3650"] = {"ms": ms, "GB/s": b / ms / 1e6, "frac": b / ms / 1e6 / HBM_PEAK_GBS, "cell-timesteps/s": E / ms * 1e3,
                       "algorithmic_bytes": b, "config": "BASELINE configs[2]",
                       "roofline": hbm_roofline(b, ms, "k_run_max_fused<4, LT> (xh_run_stats)")}
    for a in (pr, o, v):
        a.free()
    # ---- tx90p on 30 years (150 samples per day of year)
    T = 10950
    ta = TimeAxis.daily("1981-01-01", T, "noleap")
    tb, years, doys = ta.doy_table()
    seg, _ = ta.segments("YS")
    P, D = len(seg) - 1, len(doys)
    base = seasonal_base(T)
    tas = K.fill_synthetic(dev, T, C, 0, 2, base, 3.0)
    per = dev.empty((1, D, C), np.float64)
    cnt, val = dev.empty((P, C), np.int32), dev.empty((P, C), np.int32)
    tidx = dev.to_device(np.searchsorted(doys, ta.doy).astype(np.int32))
    ms_p = event_time(dev, lambda: K.percentile_doy(dev, tas, tb, 5, [90.0], out=per), 2)
    ms_c = event_time(dev, lambda: K.threshold_count(dev, tas, ">", seg, doy_table=per.reshape(D, C), tidx=tidx,
                                                     out=(cnt, val)), 3)
    E = float(T) * C
    bp, bc = 4 * E + 8 * D * C, 4 * E + 8 * D * C + 8 * P * C
    out["tx90p_30yr"] = {"percentile_doy_ms": ms_p, "percentile_doy_GB/s": bp / ms_p / 1e6, "threshold_count_ms": ms_c,
                         "threshold_count_GB/s": bc / ms_c / 1e6, "ms": ms_p + ms_c, "GB/s": (bp + bc) / (ms_p + ms_c) / 1e6,
                         "frac": (bp + bc) / (ms_p + ms_c) / 1e6 / HBM_PEAK_GBS, "cell-timesteps/s": E / (ms_p + ms_c) * 1e3,
                         "algorithmic_bytes": bp + bc, "config": "BASELINE configs[4], the tx90p half on one GPU's 1440x720 grid",
                         "roofline": hbm_roofline(bp + bc, ms_p + ms_c, "k_pdoy_quad<32, false> (xh_percentile_doy) + k_tc_doy<0, false> (xh_threshold_count_doy)",
                                                  traffic=pmc_traffic_30yr("k_pdoy_quad<32, false, 0>", "k_tc_doy<0, false>"), traffic_source=PMC_30YR),
                         "roofline_percentile_doy": hbm_roofline(bp, ms_p, "k_pdoy_quad<32, false>", traffic=pmc_traffic_30yr("k_pdoy_quad<32, false, 0>"),
                                                                 traffic_source=PMC_30YR),
                         "roofline_threshold_count": hbm_roofline(bc, ms_c, "k_tc_doy<0, false>", traffic=pmc_traffic_30yr("k_tc_doy<0, false>"),
                                                                  traffic_source=PMC_30YR),
                         "roofline_valu": valu_bound("k_pdoy_quad")}
    # a percentile in the MIDDLE of the distribution on the same field (the register top-16 kernels do not apply)
    ms50 = event_time(dev, lambda: K.percentile_doy(dev, tas, tb, 5, [50.0], out=per), 2)
    out["percentile_doy_30yr_median"] = {"ms": ms50, "GB/s": bp / ms50 / 1e6, "frac": bp / ms50 / 1e6 / HBM_PEAK_GBS,
                                          "cell-timesteps/s": E / ms50 * 1e3, "algorithmic_bytes": bp,
                                          "roofline": hbm_roofline(bp, ms50, "k_pdoy_walk<5> (xh_percentile_doy, per = 50)"),
                                          "note": "sorted day-set lists in LDS, a split that walks from day to day; latency-bound (one wave per SIMD)"}
    period = (np.searchsorted(seg, tb, side="right") - 1).astype(np.int32)
    period[tb < 0] = -1
    fused = K.percentile_doy_count(dev, tas, tb, 5, 90.0, ">", period, P, out=(cnt, val))
    if fused is not None:
        msf = event_time(dev, lambda: K.percentile_doy_count(dev, tas, tb, 5, 90.0, ">", period, P, out=(cnt, val)), 2)
        bf = 4 * E + 8 * P * C
        out["tx90p_30yr_one_call"] = {"ms": msf, "GB/s": bf / msf / 1e6, "frac": bf / msf / 1e6 / HBM_PEAK_GBS,
                                   "cell-timesteps/s": E / msf * 1e3, "algorithmic_bytes": bf,
                                   "note": "xh_percentile_doy_count on a multi-year base period is ONE CALL, not one fused kernel (the 1-year "
                                           "form, extra.tx90p_fused, is): table kernel into scratch, then the count kernel (key renamed in round 5: "
                                           "was tx90p_30yr_fused)",
                                   "roofline": hbm_roofline(bf, msf, "xh_percentile_doy_count = k_pdoy_quad<32, false> into scratch + k_tc_doy<0, false>",
                                                            passes="the samples cross HBM twice (table kernel, count kernel) + the (D, C) fp64 scratch table once each way: 8E + 16DC bytes for 4E algorithmic")}
    for a in (per, cnt, val):
        a.free()
    # ---- configs[3]: EQM train + adjust on 30 years
    ref = tas
    hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
    q = (np.arange(20) + 0.5) / 20
    af, hq = dev.empty((20, C), np.float32), dev.empty((20, C), np.float32)
    ms_tr = event_time(dev, lambda: K.eqm_train(dev, ref, hist, q, "+", out=(af, hq)), 2)
    ref.free()
    sim = K.fill_synthetic(dev, T, C, 0, 6, base + np.float32(3.5), 3.3)
    scen = dev.empty((T, C), np.float32)
    ms_ad = event_time(dev, lambda: K.eqm_adjust(dev, sim, af, hq, "+", "nearest", "constant", out=scen), 3)
    out["eqm_c4"] = {"train_ms": ms_tr, "train_GB/s": 8 * E / ms_tr / 1e6, "train_frac": 8 * E / ms_tr / 1e6 / HBM_PEAK_GBS,
                     "adjust_ms": ms_ad, "adjust_GB/s": 8 * E / ms_ad / 1e6, "ms": ms_tr + ms_ad,
                     "GB/s": 16 * E / (ms_tr + ms_ad) / 1e6, "frac": 16 * E / (ms_tr + ms_ad) / 1e6 / HBM_PEAK_GBS,
                     "cell-timesteps/s": E / (ms_tr + ms_ad) * 1e3, "algorithmic_bytes": 16 * E,
                     "config": "BASELINE configs[3]",
                     "roofline": hbm_roofline(16 * E, ms_tr + ms_ad, "xh_eqm_train + xh_eqm_adjust"),
                     "roofline_train": hbm_roofline(8 * E, ms_tr, "2 x (k_hs_sample + k_hs_hist + k_hs_collect) (select4.hip)",
                                                    passes="two streaming passes per array: 16E + 0.25E bytes cross HBM for 8E algorithmic",
                                                    traffic=(lambda t: None if t is None else 2 * t)(pmc_traffic_30yr(
                                                        "k_hs_sample<4>", "k_hs_hist<8, 5, false>", "k_hs_collect<8, 5, false>")),
                                                    traffic_source=PMC_30YR),
                     "roofline_adjust": hbm_roofline(8 * E, ms_ad, "k_eqm_adjust<20, 0>")}
    # ---- QDM adjust at the realistic size (VERDICT r4 #3c): 30 years, full grid, "nearest".  Round 5: three streaming passes
    #      (xh_qdm_hist: histogram, collect + cut values, classification) instead of the exact-rank kernel that keeps a column
    #      in one workgroup (k_qdm_columns) behind 128 x 128 transposes: 228 -> 47.5 ms
    ms_q4 = event_time(dev, lambda: K.qdm_adjust(dev, sim, af, q, "+", "nearest", "constant", out=scen), 1)
    out["qdm_c4"] = {"ms": ms_q4, "GB/s": 8 * E / ms_q4 / 1e6, "frac": 8 * E / ms_q4 / 1e6 / HBM_PEAK_GBS, "algorithmic_bytes": 8 * E,
                     "cell-timesteps/s": E / ms_q4 * 1e3, "config": "QuantileDeltaMapping.adjust on the grid of BASELINE configs[3]",
                     "roofline": hbm_roofline(8 * E, ms_q4, "k_hs_sample + k_hs_hist<8, 5, true> + k_hs_collect<8, 5, true> (select4.hip, QDM mode) + k_cut_classify (qdm2.hip)",
                                              passes="three streaming passes over sim + one write of scen: 16E + 0.25E bytes cross HBM for 8E algorithmic "
                                                     "(floor 0.5 of the two-pass-select + classify design)")}
    for a in (hist, sim, scen, af, hq):
        a.free()
    out["eqm_doy_linear"] = bench_plane_linear(dev, K, C // 8)
    out["eqm_month_linear"] = bench_plane_month(dev, K, C // 8)
    out["eqm_doy_window31"] = bench_doy_window(dev, K, C // 8)
    out["tx90p_bootstrap_band"] = bench_bootstrap(dev, K, C // 8)
    out["eqm_930_celsius"] = bench_zero_straddle(dev, K, C // 8)
    out["c5_slab"] = bench_c5_slab(dev, K)
    if not long_series:
        return out
    # ---- 1950-2100 daily (55 152 steps: beyond the 32768-step column kernels) on a 1440 x 90 band of the grid: EQM train
    #      (select4.hip streams any T <= 65535) and QDM adjust (qdm3.hip: ranks through a global sort) — round 3 refused both
    T, Cb = 55152, C // 8
    base = seasonal_base(T)
    ref = K.fill_synthetic(dev, T, Cb, 0, 4, base, 3.0)
    hist = K.fill_synthetic(dev, T, Cb, 0, 5, base + np.float32(1.5), 3.3)
    af, hq = dev.empty((20, Cb), np.float32), dev.empty((20, Cb), np.float32)
    ms_tr = event_time(dev, lambda: K.eqm_train(dev, ref, hist, q, "+", out=(af, hq)), 2)
    E = float(T) * Cb
    out["eqm_55k"] = {"train_ms": ms_tr, "GB/s": 8 * E / ms_tr / 1e6, "frac": 8 * E / ms_tr / 1e6 / HBM_PEAK_GBS,
                      "grid": [T, 1440, 90], "algorithmic_bytes": 8 * E,
                      "roofline": hbm_roofline(8 * E, ms_tr, "2 x (k_hs_sample + k_hs_hist + 4 x k_hs_collect) (select4.hip, T = 55152)",
                                               passes="the candidates of a 64-column tile (~81 K keys) fill the 32768-key LDS pool 4 times: 5 reads per array; "
                                                      "round 5: lists of 2049 .. 8192 keys are sorted in LDS instead of sending their columns (4 %) through the "
                                                      "radix select: 85 -> 67 ms")}
    ref.free()
    scen = dev.empty((T, Cb), np.float32)
    ms_qd = event_time(dev, lambda: K.qdm_adjust(dev, hist, af, q, "+", "nearest", "constant", out=scen), 1)
    out["qdm_55k"] = {"ms": ms_qd, "GB/s": 8 * E / ms_qd / 1e6, "frac": 8 * E / ms_qd / 1e6 / HBM_PEAK_GBS, "grid": [T, 1440, 90],
                      "algorithmic_bytes": 8 * E,
                      "roofline": hbm_roofline(8 * E, ms_qd, "k_hs_hist / k_hs_collect (QDM mode, 4 collect rounds; lists beyond 2048 keys sorted in LDS) + k_cut_classify",
                                               passes="round 5: the streaming path of qdm_c4 (histogram + 4 collect rounds + classification = 6 reads, 1 write); "
                                                      "round 4 ranked every column through a global sort: 779 ms")}
    for a in (hist, scen, af, hq):
        a.free()
    return out


def bench_plane_linear(dev, K, Cb):
    """EmpiricalQuantileMapping(group="time.dayofyear").adjust(interp="linear") — the documented standard configuration
    (docs/sdba.rst:64-65): factors interpolated over the (quantile, day-of-year) plane, xh_plane_linear, on a 1440 x 90 band,
    30 years.  Synthetic node tables (365 groups x 20 nodes per cell, a seasonal cycle + a per-cell offset).  Integer group
    coordinates take the row kernel (a row's nodes in registers, serving its 30 steps; gaps under 2 group steps are decided
    there) + the Delaunay walk for the listed rest: 301 ms with the walk for everything -> 42 ms."""
    from xclim_amd.timeaxis import TimeAxis

    T, G, nq = 10950, 365, 20
    ta = TimeAxis.daily("1981-01-01", T, "noleap")
    g = np.asarray(ta.doy, dtype=np.float64)
    rng = np.random.default_rng(5)
    node = (288.0 + 12.0 * np.sin(2 * np.pi * (np.arange(G) - 100) / 365))[:, None] + 3.0 * np.sort(rng.normal(0, 1, (G, nq)), axis=1)
    hq = (node[:, :, None] + rng.normal(0, 0.2, Cb)[None, None, :]).astype(np.float32)
    af = (1.5 + 0.3 * rng.normal(0, 1, (G, nq)))[:, :, None].astype(np.float32) + np.zeros((1, 1, Cb), np.float32)
    d_hq, d_af = dev.to_device(hq), dev.to_device(af)
    del hq, af
    sim = K.fill_synthetic(dev, T, Cb, 0, 6, seasonal_base(T), 3.3)
    scen = dev.empty((T, Cb), np.float32)
    gd = dev.to_device(g)
    ms = event_time(dev, lambda: K.plane_linear(dev, sim, gd, d_af, xq_all=d_hq, kind="+", out=scen), 1)
    E = float(T) * Cb
    b = 8 * E + 2 * 4.0 * G * nq * Cb
    res = {"ms": ms, "GB/s": b / ms / 1e6, "frac": b / ms / 1e6 / HBM_PEAK_GBS, "cell-timesteps/s": E / ms * 1e3, "grid": [T, 1440, 90],
           "algorithmic_bytes": b, "groups": G, "nodes": nq,
           "roofline": hbm_roofline(b, ms, "k_plane_pack + k_plane_rows<20> + k_plane_work (plane.hip)",
                                    passes="row kernel 12 ms (streams sim / scen + the node tables once), appends 2 ms, the Delaunay walk of the "
                                           "listed queries (node gaps >= 2 group steps) 28 ms: gathers + fp64, not a streaming kernel")}
    for a in (d_hq, d_af, sim, scen, gd):
        a.free()
    return res


def bench_plane_month(dev, K, Cb):
    """EmpiricalQuantileMapping(group="time.month").adjust(interp="linear"): FRACTIONAL group coordinates (month - 0.5 + day /
    days_in_month), 12 groups x 20 nodes, 30 years on a 1440 x 90 band.  k_plane_pair (round 5): the two rows around a step
    in registers + lane-private LDS, the triangle of the two-row tiling by a count over the cuts, accepted when its
    circumcircle reaches no third row; the rest (a seasonal cycle of 12 K shifts neighbouring months by several kelvin = several
    units of the plane: ~30 % of the queries) takes the Delaunay walk from the work list."""
    from xclim_amd.timeaxis import TimeAxis

    T, G, nq = 10950, 12, 20
    ta = TimeAxis.daily("1981-01-01", T, "noleap")
    g = np.asarray(ta.month - 0.5 + ta.day / ta.days_in_month(), dtype=np.float64)
    from statistics import NormalDist

    rng = np.random.default_rng(6)
    mid = (np.arange(G) + 0.5) * 365.0 / 12.0
    # the nodes a trained model has: the quantiles (i + 1/2) / nq of a normal distribution around the month's mean (+ a little
    # noise per month), not 20 random draws (whose tail gaps of 2-3 K send more queries to the walk)
    z = np.array([NormalDist().inv_cdf((i + 0.5) / nq) for i in range(nq)])
    node = (288.0 + 12.0 * np.sin(2 * np.pi * (mid - 100) / 365))[:, None] + 3.3 * z[None, :] + rng.normal(0, 0.05, (G, nq))
    node = np.sort(node, axis=1)
    hq = (node[:, :, None] + rng.normal(0, 0.2, Cb)[None, None, :]).astype(np.float32)
    af = (1.5 + 0.3 * rng.normal(0, 1, (G, nq)))[:, :, None].astype(np.float32) + np.zeros((1, 1, Cb), np.float32)
    d_hq, d_af = dev.to_device(hq), dev.to_device(af)
    del hq, af
    sim = K.fill_synthetic(dev, T, Cb, 0, 6, seasonal_base(T), 3.3)
    scen = dev.empty((T, Cb), np.float32)
    gd = dev.to_device(g)
    ms = event_time(dev, lambda: K.plane_linear(dev, sim, gd, d_af, xq_all=d_hq, kind="+", out=scen), 1)
    E = float(T) * Cb
    b = 8 * E + 2 * 4.0 * G * nq * Cb
    res = {"ms": ms, "GB/s": b / ms / 1e6, "frac": b / ms / 1e6 / HBM_PEAK_GBS, "cell-timesteps/s": E / ms * 1e3, "grid": [T, 1440, 90],
           "algorithmic_bytes": b, "groups": G, "nodes": nq,
           "roofline": hbm_roofline(b, ms, "k_plane_pack + k_plane_pair<20> + k_plane_work (plane.hip)",
                                    passes="pair kernel ~35 ms (fp64 issue-bound: ~1 000 VALU instructions per query incl. the apex search per "
                                           "pair of rows) + the Delaunay walk of the listed queries (gathers + fp64): ~28 ms on these nodes, 85 ms "
                                           "on the noisier nodes of a model trained on the synthetic field (tools/experiments/r05/gpu_r05_p.sh)")}
    for a in (d_hq, d_af, sim, scen, gd):
        a.free()
    return res


def bench_doy_window(dev, K, Cb):
    """The documented standard configuration END TO END through the host mirror (docs/sdba.rst:64-65:
    ``EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, group=Grouper("time.dayofyear", window=31))`` then
    ``.adjust(sim, interp=...)``), 30 years on a 1440 x 90 band, device-resident series, wall clock of the Python calls
    (365 x 2 quantile problems of 930 samples per cell in training; round 6: xh_eqm_train_window keeps every cell's window
    sorted from one day of the year to the next — winsel.hip — instead of selecting each group from its gathered sample:
    508 -> 150 ms).  The adjust legs run on the TRAINED node tables (smooth from one day of the year to the next), not on the
    random ones of extra.eqm_doy_linear."""
    import time

    from xclim_amd import sdba
    from xclim_amd.timeaxis import TimeAxis

    T = 10950
    ta = TimeAxis.daily("1981-01-01", T, "noleap")
    base = seasonal_base(T)
    ref = K.fill_synthetic(dev, T, Cb, 0, 4, base, 3.0)
    hist = K.fill_synthetic(dev, T, Cb, 0, 5, base + np.float32(1.5), 3.3)
    sim = K.fill_synthetic(dev, T, Cb, 0, 6, base + np.float32(3.5), 3.3)

    def timed(fn, n):
        r = fn()
        dev.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        dev.sync()
        return (time.perf_counter() - t0) / n * 1e3, r

    ms_tr, eqm = timed(lambda: sdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, kind="+", group="time.dayofyear", window=31,
                                                                  time=ta, device=dev), 1)
    ms_near, _ = timed(lambda: eqm.adjust(sim, interp="nearest", time=ta, keep=True), 2)
    ms_lin, _ = timed(lambda: eqm.adjust(sim, interp="linear", time=ta, keep=True), 2)
    # the other two mappings on the same configuration (round 6: QDM ranks all 365 groups in one launch — xh_qdm_adjust_groups;
    # DQM trains through the normalising instance of the sliding window — xh_dqm_train_window — and detrends all groups in one
    # launch per stage — xh_poly_trend_groups / xh_trend_apply_groups)
    qdm = sdba.QuantileDeltaMapping(dev, eqm._af, eqm._hist_q, eqm.quantiles, eqm.kind, eqm.cell_shape, eqm.group, eqm.group_labels)
    ms_qdm, _ = timed(lambda: qdm.adjust(sim, interp="nearest", time=ta, keep=True), 2)
    ms_dtr, dqm = timed(lambda: sdba.DetrendedQuantileMapping.train(ref, hist, nquantiles=20, kind="+", group="time.dayofyear", window=31,
                                                                    time=ta, device=dev), 1)
    ms_dad, _ = timed(lambda: dqm.adjust(sim, interp="nearest", time=ta, keep=True), 2)
    # ... and the grouping WITHOUT a window (365 groups of one row per year: xh_eqm_train_groups / xh_dqm_train_groups, one launch per field)
    ms_tr1, m1 = timed(lambda: sdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, kind="+", group="time.dayofyear", time=ta, device=dev), 2)
    ms_dtr1, m2 = timed(lambda: sdba.DetrendedQuantileMapping.train(ref, hist, nquantiles=20, kind="+", group="time.dayofyear", time=ta, device=dev), 2)
    del qdm, dqm, m1, m2
    E = float(T) * Cb
    res = {"train_ms": ms_tr, "adjust_nearest_ms": ms_near, "adjust_linear_ms": ms_lin, "qdm_adjust_nearest_ms": ms_qdm,
           "dqm_train_ms": ms_dtr, "dqm_adjust_nearest_ms": ms_dad, "train_nowindow_ms": ms_tr1, "dqm_train_nowindow_ms": ms_dtr1, "grid": [T, 1440, 90], "groups": 365, "window": 31,
           "nodes": 20, "train_samples_GB": 2 * 365 * 930 * 4.0 * Cb / 1e9,
           "train_GB/s": 2 * 365 * 930 * 4.0 * Cb / ms_tr / 1e6, "adjust_linear_GB/s": 8 * E / ms_lin / 1e6,
           "adjust_linear_frac": 8 * E / ms_lin / 1e6 / HBM_PEAK_GBS,
           "kernel": "k_window_quantiles x 2 + k_correction (winsel.hip) | xh_plane_nearest | xh_plane_linear",
           "note": "wall clock of the host mirror calls (device-resident inputs); training = 730 quantile problems of 930 samples per "
                   "cell as ONE sliding sorted window per cell and field (one wave per cell, 30 samples leave and 30 enter per step); "
                   "train_GB/s counts the bytes of the samples the 730 problems select from, not HBM traffic (each sample is read "
                   "twice: entering and leaving)"}
    for a in (ref, hist, sim):
        a.free()
    return res


def bench_zero_straddle(dev, K, Cb):
    """eqm_train on a 930-row sample (a month group of 30 years; k_select_quantile: histogram selection in LDS) of the same
    synthetic field in kelvin and in degrees Celsius.  The histogram bins were linear in the order-preserving KEY (log-like in
    the value): fine for 288 +- 15, but a field that straddles zero spreads over a few binade-wide bins and the exact search
    inside a bin turned quadratic — 1.36 against 19.7 ms (round 6, tools/experiments/r06/zero_straddle_time.py).  Keys on both
    sides of zero now take bins linear in the value (common.h xh_value_bins: k_select_quantile, k_select_grp, k_qdm_columns)."""
    T = 930
    q = np.array([(i + 0.5) / 20 for i in range(20)])
    res = {"rows": T, "grid": [T, 1440, 90], "kernel": "k_select_quantile<64, 16, 512> x 2 (select.hip)"}
    for name, mean in (("kelvin", 288.0), ("celsius", 15.0), ("anomaly", 0.0)):
        base = seasonal_base(T, mean=mean)
        ref = K.fill_synthetic(dev, T, Cb, 0, 4, base, 3.0)
        hist = K.fill_synthetic(dev, T, Cb, 0, 5, base + np.float32(1.5), 3.3)
        af, hq = dev.empty((20, Cb), np.float32), dev.empty((20, Cb), np.float32)
        res[name + "_ms"] = event_time(dev, lambda: K.eqm_train(dev, ref, hist, q, "+", out=(af, hq)), 3)
        for a in (ref, hist, af, hq):
            a.free()
    return res


def bench_bootstrap(dev, K, Cb):
    """tx90p with the percentile bootstrap of Zhang et al. 2005 (core/bootstrapping.py:81-282: every in-base year is
    evaluated against the n - 1 percentile tables in which it is replaced by another base year): 31 years on a 1440 x 90
    band, base period = the first 30 -> 30 x 29 = 870 thirty-year percentile tables.  The reference deep-copies the base
    period per replica; here a replica is an index table (`vmap`) over the resident series.  Wall clock of the host
    mirror, device-resident input, result on the device."""
    import time as _time

    from xclim_amd import indices as xi
    from xclim_amd.calendar import percentile_doy
    from xclim_amd.timeaxis import TimeAxis

    nyears, nbase = 31, 30
    T = 365 * nyears
    ta = TimeAxis.daily("1961-01-01", T, "noleap")
    tas = K.fill_synthetic(dev, T, Cb, 0, 2, seasonal_base(T), 3.0)
    nb = 365 * nbase
    p90 = percentile_doy(dev.wrap(tas.ptr, (nb, Cb), np.float32), ta.subset(slice(0, nb)), 5, 90.0, device=dev)
    run = lambda: xi.tx90p(tas, p90, ta, freq="YS", device=dev, bootstrap=True)  # noqa: E731
    run()
    dev.sync()
    t0 = _time.perf_counter()
    run()
    dev.sync()
    ms = (_time.perf_counter() - t0) * 1e3
    nrep = nbase * (nbase - 1)
    E = float(T) * Cb
    res = {"ms": ms, "replicas": nrep, "ms_per_replica": ms / nrep, "grid": [T, 1440, 90], "cell-timesteps/s": E / ms * 1e3,
           "replica-cell-timesteps/s": float(nb) * Cb * nrep / ms * 1e3,
           "note": "870 percentile tables over 30 years (k_pdoy_quad through a virtual time map) + the year's exceedance count "
                   "against each; wall clock incl. the host loop; the reference copies the base period once per replica"}
    tas.free()
    return res


def bench_c5_slab(dev, K):
    """The per-GPU work of BASELINE configs[4] (`--workload c5 --gpus 1`: tx90p + EQM train + adjust on ONE 360 x 1440 slab of
    the 2880 x 1440 grid, 30 years) inside the default line — the N = 1 anchor of a future 8-GPU SCALE run lives in the
    same file.  No exchange (one rank)."""
    from xclim_amd.timeaxis import TimeAxis

    T, C = 10950, 360 * 1440
    ta = TimeAxis.daily("1981-01-01", T, "noleap")
    tb, years, doys = ta.doy_table()
    seg, _ = ta.segments("YS")
    P, D = len(seg) - 1, len(doys)
    expected = ta.expected_count("YS")
    base = seasonal_base(T)
    tas = K.fill_synthetic(dev, T, C, 0, 2, base, 3.0)
    ref = K.fill_synthetic(dev, T, C, 0, 4, base, 3.0)
    hist = K.fill_synthetic(dev, T, C, 0, 5, base + np.float32(1.5), 3.3)
    sim = K.fill_synthetic(dev, T, C, 0, 6, base + np.float32(3.5), 3.3)
    scen = dev.empty((T, C), np.float32)
    per = dev.empty((1, D, C), np.float64)
    cnt, val = dev.empty((P, C), np.int32), dev.empty((P, C), np.int32)
    res = dev.empty((P, C), np.float64)
    af, hq = dev.empty((20, C), np.float32), dev.empty((20, C), np.float32)
    tidx = dev.to_device(np.searchsorted(doys, ta.doy).astype(np.int32))
    q = (np.arange(20) + 0.5) / 20

    def step():
        K.percentile_doy(dev, tas, tb, 5, [90.0], out=per)
        K.threshold_count(dev, tas, ">", seg, doy_table=per.reshape(D, C), tidx=tidx, out=(cnt, val))
        K.apply_missing_mask(dev, cnt, val, expected, out=res)
        K.eqm_train(dev, ref, hist, q, "+", out=(af, hq))
        K.eqm_adjust(dev, sim, af, hq, "+", "nearest", "constant", out=scen)

    ms = event_time(dev, step, 3)
    E = float(T) * C
    b = (2 * (4 * E + 8 * D * C) + 8 * P * C) + 16 * E
    out = {"ms": ms, "GB/s": b / ms / 1e6, "frac": b / ms / 1e6 / HBM_PEAK_GBS, "cell-timesteps/s": E / ms * 1e3, "grid_per_gpu": [T, 360, 1440],
           "algorithmic_bytes": b, "config": "BASELINE configs[4], one of the 8 slabs (= `bench.py --workload c5 --gpus 1`), no exchange",
           "roofline": hbm_roofline(b, ms, "percentile_doy + threshold_count + missing mask + eqm_train + eqm_adjust")}
    for a in (tas, ref, hist, sim, scen, per, cnt, val, res, af, hq, tidx):
        a.free()
    return out


def _cpu_worker(job):
    """One CPU worker (spawned process, never touches the GPU): the tx90p chain of the oracle on 512-cell blocks of the
    synthetic field, every `stride`-th block starting at `first`, until `budget_s` seconds of compute have elapsed."""
    T, ncells_total, first, stride, budget_s = job
    os.environ.setdefault("OMP_NUM_THREADS", "1")
    sys.path.insert(0, ROOT)
    from oracle import calendar as ocal
    from oracle import indices as oidx
    from oracle import synth
    from oracle.timeutil import OTime

    ot = OTime.noleap(2001, T)
    base = synth.seasonal_base(T)
    chunk = 512

    def one(c0):
        xs = synth.fill_synthetic(T, np.arange(c0, c0 + chunk), 0, 2, base, 3.0)
        t0 = time.perf_counter()
        p, doys = ocal.percentile_doy(xs, ot, 5, 90.0)
        cnt = oidx.tx90p(xs, p[..., 0], doys, ot, "YS")
        oidx.apply_missing(cnt, xs, ot, "YS")
        return time.perf_counter() - t0

    nblocks = ncells_total // chunk
    one((first % max(nblocks, 1)) * chunk)  # untimed warm-up block
    spent, ncells, blk = 0.0, 0, first + stride
    while spent < budget_s and nblocks > 0:
        spent += one((blk % nblocks) * chunk)  # (wraps around the grid when the sample is exhausted: same work again)
        ncells += chunk
        blk += stride
    return ncells, spent


def cpu_baseline(T, Y, X, budget_s=10.0):
    """1-core AND N-core figures: ``value`` / ``cores`` are the N-core measurement (the faster one: what the
    host can do), ``one_core`` the single worker measured first with the same sampling."""
    one = _cpu_baseline(T, Y, X, budget_s, want=1)
    many = _cpu_baseline(T, Y, X, budget_s, want=0)
    many["one_core"] = {k: one[k] for k in ("value", "unit", "cores", "sample")}
    return many


def _cpu_baseline(T, Y, X, budget_s=10.0, want=0):
    """Oracle (numpy restatement of the reference) on a sample of the SAME synthetic field, on the host cores of this box.

    One worker PROCESS per core (up to 16: on the MI355X boxes 64 workers gave LESS aggregate throughput, 1.38e8 vs
    1.57e8 cell-timesteps/s — the oracle is memory bound; `XH_BENCH_CPU_CORES` overrides), started as `python bench.py --cpu-worker ...`
    (they never touch the GPU), each working through its own 512-cell blocks (cache-resident temporaries) for ~budget_s
    seconds; value = cells x steps done by all workers / the longest worker time.  A worker that fails or overruns its
    deadline is killed (by PID) and ignored; with no usable worker the measurement falls back to one in-process worker.
    """
    import subprocess

    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        avail = os.cpu_count() or 1
    want = want or int(os.environ.get("XH_BENCH_CPU_CORES", "0")) or min(avail, 16)
    res = []
    if want > 1:
        env = dict(os.environ, OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1", MKL_NUM_THREADS="1")
        procs = []
        for w in range(want):
            job = json.dumps([T, Y * X, w, want, budget_s])
            try:
                procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", job],
                                              stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True))
            except OSError:
                break
        deadline = time.time() + budget_s + 90.0
        for pr in procs:
            try:
                out, _ = pr.communicate(timeout=max(1.0, deadline - time.time()))
                r = json.loads(out.strip().splitlines()[-1])
                if r[0] > 0:
                    res.append((int(r[0]), float(r[1])))
            except Exception:
                pr.kill()
    cores = len(res)
    if not res:
        cores = 1
        res = [_cpu_worker((T, Y * X, 0, 1, budget_s))]
    ncells = sum(r[0] for r in res)
    spent = max(r[1] for r in res)
    return {"value": T * ncells / spent, "unit": "cell-timesteps/s", "cores": cores, "kind": "port",
            "sample": f"tx90p oracle (numpy) on {ncells} cells x {T} steps of the same synthetic field, "
                      f"{spent:.1f} s timed per worker, 512-cell blocks, {cores} single-threaded worker process(es)"}


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-worker":
        print(json.dumps(_cpu_worker(tuple(json.loads(sys.argv[2])))))
        sys.exit(0)
    main()
