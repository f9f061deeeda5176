"""CPU-side checks that need no GPU: calendar tables vs pandas, C-ABI symbol export, host-side validation, and the
synthetic generator's determinism.  (`-m "not gpu"` suite.)"""

import ctypes
import os
import re
import sys

import numpy as np
import pandas as pd
import pytest

from oracle import synth
from oracle.timeutil import OTime, days_in_period, groups
from xclim_amd import _capi, generic
from xclim_amd import run_length as xrl
from xclim_amd.calendar import doy_interp_tables
from xclim_amd.timeaxis import TimeAxis, parse_freq

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ---- calendar tables -------------------------------------------------------------------------------------------
@pytest.mark.parametrize("start,n", [("2000-07-01", 365), ("1999-12-15", 800), ("2001-01-01", 1461), ("2000-02-28", 3)])
@pytest.mark.parametrize("freq", ["YS", "MS", "QS-DEC", "YS-JUL", "QS", "ME", "YE", "QE-NOV", "AS-JUL"])
def test_segments_match_pandas_resample(start, n, freq):
    ta = TimeAxis.daily(start, n)
    seg, starts = ta.segments(freq)
    idx = pd.date_range(start, periods=n, freq="D")
    pfreq = freq.replace("AS", "YS")
    sizes = pd.Series(np.arange(n), index=idx).resample(pfreq).count().values
    np.testing.assert_array_equal(np.diff(seg), sizes)
    assert seg[0] == 0 and seg[-1] == n
    # same as the oracle's independent grouping
    og = groups(OTime.standard(start, n), pfreq)
    assert [len(g[1]) for g in og] == list(np.diff(seg))


@pytest.mark.parametrize("freq", ["YS", "MS", "QS-DEC", "YS-JUL"])
def test_expected_count_matches_oracle(freq):
    ta = TimeAxis.daily("1999-11-20", 900)
    np.testing.assert_array_equal(ta.expected_count(freq), days_in_period(OTime.standard("1999-11-20", 900), freq))


@pytest.mark.parametrize("calendar,ndoy", [("noleap", 365), ("360_day", 360)])
def test_nonstandard_calendars(calendar, ndoy):
    ta = TimeAxis.daily("2001-01-01", ndoy * 3, calendar)
    ot = OTime.noleap(2001, ndoy * 3, calendar)
    np.testing.assert_array_equal(ta.doy, ot.doy)
    np.testing.assert_array_equal(ta.month, ot.month)
    seg, _ = ta.segments("YS")
    np.testing.assert_array_equal(seg, [0, ndoy, 2 * ndoy, 3 * ndoy])
    np.testing.assert_array_equal(ta.expected_count("YS"), [ndoy] * 3)
    tb, years, doys = ta.doy_table()
    assert tb.shape == (3, ndoy) and (tb >= 0).all() and tb[1, 0] == ndoy


def test_doy_table_leap():
    ta = TimeAxis.daily("1999-01-01", 365 + 366)
    tb, years, doys = ta.doy_table()
    assert tb.shape == (2, 366) and tb[0, 365] == -1 and tb[1, 365] == 365 + 365
    np.testing.assert_array_equal(ta.doy, pd.date_range("1999-01-01", periods=731).dayofyear.values)


def test_parse_freq_errors():
    assert parse_freq("YS-JUL") == ("Y", 7) and parse_freq("QE-NOV") == ("Q", 12) and parse_freq("YE") == ("Y", 1)
    assert parse_freq("7D") == ("D", 7) and parse_freq("D") == ("D", 1) and parse_freq("W") == ("W", 6) and parse_freq("W-WED") == ("W", 2)
    with pytest.raises(NotImplementedError):
        parse_freq("2h")
    with pytest.raises(ValueError):
        parse_freq("YS-FOO")
    with pytest.raises(ValueError):
        parse_freq("W-FOO")


def test_doy_interp_tables():
    i0, i1, dxn, dxs = doy_interp_tables(365, 366, 1)
    assert len(i0) == 366 and i0[0] == 0 and i1[-1] == 364 and (i1 == i0 + 1).all()
    x = np.linspace(1, 366, 365)
    np.testing.assert_allclose(x[i0] + dxn, np.arange(1, 367))
    np.testing.assert_allclose(dxs, x[i1] - x[i0])


# ---- C ABI -------------------------------------------------------------------------------------------------------
def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "xclim_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(xh_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_header_symbol():
    """The library must load on a GPU-less host and export exactly what include/xclim_hip.h declares."""
    lib = _capi.load_library()
    names = _header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in the header but not exported"
        assert n in _capi.SIGNATURES, f"{n} has no ctypes prototype"
    assert set(_capi.SIGNATURES) == set(names)
    assert lib.xh_abi_version() == 1
    n = ctypes.c_int(-1)
    assert lib.xh_device_count(ctypes.byref(n)) == 0 and n.value >= 0


def test_comm_entry_points_validate_arguments(monkeypatch, tmp_path):
    """xh_comm_* (RCCL behind the C ABI, multi-GPU exchange): argument validation happens before librccl is touched, so
    it can be driven on a GPU-less host; the rendezvous file name is unique per launch and shared by its ranks."""
    lib = _capi.load_library()
    null = ctypes.c_void_p(0)
    assert lib.xh_comm_unique_id(null) == _capi.XH_ERR_ARG
    h = ctypes.c_void_p()
    buf = ctypes.create_string_buffer(128)
    assert lib.xh_comm_init(null, 2, 0, ctypes.cast(buf, ctypes.c_void_p), ctypes.byref(h)) == _capi.XH_ERR_ARG
    assert b"NULL" in lib.xh_last_error()
    assert lib.xh_comm_allgather(null, null, null, 16, -1) == _capi.XH_ERR_ARG
    assert lib.xh_comm_fence(null, 0) == _capi.XH_ERR_ARG
    assert lib.xh_comm_sync(null) == _capi.XH_ERR_ARG
    assert lib.xh_comm_barrier(null) == _capi.XH_ERR_ARG
    one = ctypes.c_double(1.0)
    assert lib.xh_comm_allreduce_f64(null, ctypes.byref(one), 1, 2) == _capi.XH_ERR_ARG
    assert lib.xh_comm_destroy(null) == 0  # destroying nothing is fine (like free(NULL))
    from xclim_amd import shard

    monkeypatch.setenv("XH_RENDEZVOUS_DIR", str(tmp_path))
    monkeypatch.setenv("MASTER_ADDR", "127.0.0.1")
    monkeypatch.setenv("MASTER_PORT", "29511")
    a = shard._rendezvous_path()
    assert a.startswith(str(tmp_path)) and str(os.getppid()) in a and "29511" in a
    monkeypatch.setenv("MASTER_PORT", "29512")
    assert shard._rendezvous_path() != a
    monkeypatch.setenv("XH_RENDEZVOUS_KEY", "job42")
    assert shard._rendezvous_path().endswith("xclim_amd_rccl_job42.id")


def test_no_cpu_fallback_without_device():
    """The product path fails loudly when no GPU is visible (no silent CPU fallback)."""
    if _capi.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_capi.BackendUnavailable):
        _capi.Device(0)
    with pytest.raises(_capi.BackendUnavailable):
        generic.threshold_count(np.zeros((3, 2), np.float32), ">", 1.0, TimeAxis.daily("2000-01-01", 3), "YS")


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "xclim_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


# ---- host-side validation (mirrors the reference's error behaviour) ---------------------------------------------
def test_operator_validation():
    assert generic.get_op("gt") == ">" and generic.get_op(">=") == ">=" and generic.get_op("ne") == "!="
    with pytest.raises(ValueError, match="not recognized"):
        generic.get_op("=>")
    with pytest.raises(ValueError, match="not permitted"):
        generic.get_op("==", constrain=(">", "<"))
    with pytest.raises(ValueError, match="not implemented for 1d method"):
        xrl.use_ufunc(True, freq="YS")


# ---- synthetic generator ---------------------------------------------------------------------------------------------
def test_equally_spaced_nodes():
    """xsdba.utils.equally_spaced_nodes: centres of n equal bins, optionally with the end points eps / 1 - eps."""
    from xclim_amd.sdba import equally_spaced_nodes

    np.testing.assert_allclose(equally_spaced_nodes(4), [0.125, 0.375, 0.625, 0.875])
    q = equally_spaced_nodes(20, eps=1e-6)
    assert len(q) == 22 and q[0] == 1e-6 and q[-1] == 1 - 1e-6
    np.testing.assert_allclose(q[1:-1], (np.arange(20) + 0.5) / 20)


def test_get_op_accepts_a_string_constrain():
    """gen:289-290: `constrain` may be a single operator string; '>=' must not be read as the two operators '>' and '='."""
    assert generic.get_op(">=", constrain=">=") == ">="
    with pytest.raises(ValueError, match="not permitted"):
        generic.get_op(">", constrain=">=")


def test_synthetic_generator_is_counter_based():
    base = synth.seasonal_base(50)
    a = synth.fill_synthetic(50, np.arange(100, 140), 0, 7, base, 3.0, nan_per_million=20000)
    b = synth.fill_synthetic(50, np.arange(120, 130), 0, 7, base, 3.0, nan_per_million=20000)
    np.testing.assert_array_equal(a[:, 20:30], b)  # a shard sees the same global field
    assert a.dtype == np.float32 and np.isnan(a).any() and abs(np.nanstd(a - base[:, None]) - 3.0 / np.sqrt(3)) < 0.1
    pr = synth.fill_synthetic(400, np.arange(64), 1, 3, np.zeros(400, np.float32), 40 / 86400.0, 0.3)
    assert 0.25 < (pr > 0).mean() < 0.35 and pr.min() == 0


def test_cell_blocks_cover_and_align():
    """blocks.py: slabs tile [0, C) in order, start on multiples of 4 cells, and the default width follows the byte target."""
    from xclim_amd.blocks import cell_blocks, default_block_cells

    for C, b in [(10, 4), (10, 5), (1003, 256), (1036800, 183856), (3, 1000), (8, 1)]:
        bl = cell_blocks(C, b)
        assert bl[0][0] == 0 and bl[-1][1] == C
        assert all(a1 == b0 for (_, a1), (b0, _) in zip(bl[:-1], bl[1:]))
        assert all(c0 % 4 == 0 and c1 > c0 for c0, c1 in bl)
    assert cell_blocks(0, 4) == []
    assert default_block_cells(365 * 4, 1036800) == 183856
    assert default_block_cells(365 * 4, 100) == 100
    with pytest.raises(ValueError):
        cell_blocks(10, 0)


def test_bench_finds_pmc_traffic_of_the_dominant_kernel():
    """bench.py reports roofline.traffic from the committed PMC passes: the lookup must follow the kernel's name."""
    import importlib.util

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    t, source = bench.pmc_traffic("k_pdoy_slide<5, 4>", (365, 1440, 720))
    assert t is not None and 4.4e9 < t < 5.5e9   # algorithmic 4.54 GB
    assert source.startswith("profiles/r0") and os.path.exists(os.path.join(root, source))   # named in roofline.traffic_source
    assert bench.pmc_traffic("k_pdoy_slide<5, 4>", (365, 10, 10)) == (None, None)


def test_dim_other_than_time_is_refused():
    """The host mirrors keep the reference's `dim` argument: "time" = axis 0, or (round 6) an INTEGER axis of the plain array;
    a dimension NAME other than "time" means nothing for an array without names and must fail loudly (before any device
    work), not be ignored — and resampling stays with the time axis."""
    from xclim_amd import sdba

    x = np.zeros((4, 3), np.float32)
    for f, args in ((xrl.rle, ()), (xrl.rle_statistics, ("max", 1)), (xrl.first_run, (2,)), (xrl.longest_run, ()),
                    (xrl.windowed_run_count, (2,)), (xrl.keep_longest_run, ()), (xrl.season, (2,))):
        with pytest.raises(NotImplementedError, match="integer axis"):
            f(x, *args, dim="lat")
    with pytest.raises(ValueError, match="resample the time axis"):
        xrl.longest_run(x, 1, "YS")
    with pytest.raises(NotImplementedError):
        sdba.quantile(x, [0.5], dim="lat")
    assert xrl.rle_statistics.__name__ == "rle_statistics"  # resample_and_rl dispatches on the name


def test_threshold_units():
    """tests/fakeunits.convert_units_to (the stand-in's unit conversion): the threshold strings of the hot-path indicators (core/units.py:334-420; the
    reference's defaults "25.0 degC", "1 mm/day" with the hydro context -> 1/86400 kg m-2 s-1)."""
    from fakeunits import convert_units_to as cvt

    assert cvt("25 degC", "K") == pytest.approx(298.15)
    assert cvt("25.0 degC", "degC") == 25.0 and cvt("-10 C", "K") == pytest.approx(263.15)
    assert cvt("86 degF", "degC") == pytest.approx(30.0)
    assert cvt("1 mm/day", "kg m-2 s-1", "hydro") == pytest.approx(1.0 / 86400.0)
    assert cvt("1 mm/d", "kg/m2/s", context="hydro") == pytest.approx(1.0 / 86400.0)
    assert cvt("2 kg/m**2/s", "kg m-2 s-1") == 2.0 and cvt("0.5 kg m-2 s-1", "mm/day", "hydro") == pytest.approx(43200.0)
    assert cvt("1 mm d-1", "mm/day") == 1.0 and cvt("1 cm/day", "mm/day") == pytest.approx(10.0)
    assert cvt("10 cm", "m") == pytest.approx(0.1) and cvt(3.5, "K") == 3.5
    for bad in (("1 mm/day", "kg m-2 s-1", None), ("1 mm", "K", None), ("1 furlong", "m", None), ("abc", "K", None)):
        with pytest.raises(ValueError):
            cvt(*bad)


def test_float64_inputs_are_flagged():
    """float64 fields / array thresholds are rounded to float32 — not silently: PrecisionWarning."""
    import warnings

    from xclim_amd._capi import PrecisionWarning, warn_downcast

    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        warn_downcast(np.zeros((3, 2), dtype=np.float64), "field")
        warn_downcast(np.zeros((3, 2), dtype=np.float32), "field")
        warn_downcast(1.5, "scalar")
        warn_downcast(np.float64(1.5), "scalar")
    assert len(w) == 1 and issubclass(w[0].category, PrecisionWarning)


@pytest.mark.parametrize("freq", ["W", "W-SUN", "W-WED", "W-MON", "7D", "10D", "D", "30D"])
def test_weekly_and_day_frequencies_match_pandas(freq):
    """resample(time="W" / "W-XXX" / "nD"): the segments of TimeAxis equal pandas' bins (weekly: closed and labelled on
    the right; n days: from the first day on), missing days inside the span included; expected counts are 7 / n."""
    import pandas as pd

    from xclim_amd.timeaxis import TimeAxis

    idx = pd.date_range("2003-12-27", periods=400, freq="D")
    keep = np.ones(len(idx), dtype=bool)
    keep[40:62] = False                       # a gap: empty bins stay
    idx = idx[keep]
    ta = TimeAxis.from_pandas(idx)
    seg, starts = ta.segments(freq)
    ref = pd.Series(1.0, index=idx).resample(freq).count()
    np.testing.assert_array_equal(np.diff(seg), ref.values)
    n = 7 if freq.startswith("W") else int(freq[:-1] or 1)
    np.testing.assert_array_equal(ta.expected_count(freq), np.full(len(ref), n))
    assert sum(s is None for s in starts) == int((ref.values == 0).sum())


def test_weekly_frequency_needs_weekdays():
    from xclim_amd.timeaxis import TimeAxis

    with pytest.raises(NotImplementedError):
        TimeAxis.daily("2001-01-01", 30, "360_day").segments("W")
    seg, _ = TimeAxis.daily("2001-01-01", 65, "360_day").segments("30D")
    np.testing.assert_array_equal(seg, [0, 30, 60, 65])


def test_adapter_wrappers_have_the_reference_signatures():
    """Every DataArray-level wrapper of xr_adapter.py takes exactly the parameters of the reference function it replaces
    (names, order, keyword-only and variadic parts) — read from the reference sources with ast; skipped where the reference
    tree is not present (the GPU box)."""
    import ast
    import inspect
    import os
    import sys

    root = "/root/reference/src/xclim"
    if not os.path.isdir(root):
        pytest.skip("reference tree not present")
    sys.path.insert(0, os.path.join(os.path.dirname(__file__)))
    import fakexr
    from xclim_amd.xr_adapter import make_wrappers

    wrappers = make_wrappers(fakexr.make_env(), orig={}, device=None)
    where = {"indices/generic.py": ("threshold_count", "count_occurrences", "domain_count", "select_resample_op",
                                    "spell_length_statistics", "cumulative_difference", "compare", "season",
                                    "first_day_threshold_reached", "bivariate_count_occurrences"),
             "core/calendar.py": ("percentile_doy", "resample_doy"),
             "indices/run_length.py": ("rle", "rle_statistics", "longest_run", "windowed_run_events", "windowed_run_count",
                                       "first_run", "last_run", "season_length", "resample_and_rl"),
             "core/utils.py": ("calc_perc",)}
    P = inspect.Parameter
    checked = 0
    for rel, names in where.items():
        tree = ast.parse(open(os.path.join(root, rel)).read())
        defs = {n.name: n.args for n in tree.body if isinstance(n, ast.FunctionDef)}
        for name in names:
            a = defs[name]
            ref = ([x.arg for x in a.posonlyargs + a.args], [x.arg for x in a.kwonlyargs], a.vararg.arg if a.vararg else None,
                   a.kwarg.arg if a.kwarg else None)
            sig = inspect.signature(wrappers[name])  # (follows __wrapped__ to the wrapper's own function)
            ps = list(sig.parameters.values())
            ours = ([p.name for p in ps if p.kind in (P.POSITIONAL_ONLY, P.POSITIONAL_OR_KEYWORD)],
                    [p.name for p in ps if p.kind == P.KEYWORD_ONLY],
                    next((p.name for p in ps if p.kind == P.VAR_POSITIONAL), None),
                    next((p.name for p in ps if p.kind == P.VAR_KEYWORD), None))
            assert ours == ref, (name, ours, ref)
            checked += 1
    assert checked == 22


def test_call_programs_are_what_the_reference_bodies_do():
    """tests/golden/call_programs.json (replayed by the adapter tests on the GPU box) equals a fresh recording from the
    reference's index bodies (AST-extracted and executed with symbolic arguments, tests/golden/make_call_programs.py)."""
    import json

    if not os.path.isdir("/root/reference/src/xclim"):
        pytest.skip("the reference tree only exists in the build container")
    sys.path.insert(0, os.path.join(os.path.dirname(__file__), "golden"))
    import make_call_programs as mcp

    fresh = json.loads(json.dumps(mcp.build(), sort_keys=True))
    with open(os.path.join(os.path.dirname(__file__), "golden", "call_programs.json")) as f:
        stored = json.load(f)
    assert fresh == stored
    assert len(stored) >= 25 and all(p["source"].startswith("src/xclim/indices/") for p in stored.values())


def _shape_key(name, npos, kws):
    return f"{name}({npos}; {', '.join(sorted(k for k in kws if k != '**'))})" + (" + **indexer" if "**" in kws else "")


def test_every_call_shape_of_a_replaced_function_is_recorded():
    """which distinct (function, positional count, keyword set) shapes do the reference's index modules use
    when they call a function ``patch.install`` replaces — and is each one exercised by a recorded call program (replayed
    through the wrappers on the GPU box, tests/test_gpu_adapter.py) or explicitly listed as forwarded?  Walks
    /root/reference/src/xclim/indices/{_threshold,_multivariate,_simple}.py with ``ast`` (128 call sites, 19 shapes)."""
    import ast
    import json

    from xclim_amd.patch import _BY_NAME

    ref = "/root/reference/src/xclim/indices"
    if not os.path.isdir(ref):
        pytest.skip("the reference tree only exists in the build container")
    generic, rlnames = set(_BY_NAME["xclim.indices._multivariate"]), set(_BY_NAME["xclim.indices.run_length"])
    sites = {}
    for f in ("_threshold.py", "_multivariate.py", "_simple.py"):
        with open(os.path.join(ref, f)) as fh:
            tree = ast.parse(fh.read())
        for fn in (n for n in tree.body if isinstance(n, ast.FunctionDef)):
            for node in (n for n in ast.walk(fn) if isinstance(n, ast.Call)):
                name = None
                if isinstance(node.func, ast.Name) and node.func.id in generic:
                    name = node.func.id
                elif (isinstance(node.func, ast.Attribute) and isinstance(node.func.value, ast.Name) and node.func.value.id == "rl"
                      and node.func.attr in rlnames):
                    name = "rl." + node.func.attr
                if name:
                    key = _shape_key(name, len(node.args), [k.arg or "**" for k in node.keywords])
                    sites.setdefault(key.replace(" + **indexer", ""), []).append(f"{f}:{fn.name}:{node.lineno}")
    with open(os.path.join(os.path.dirname(__file__), "golden", "call_programs.json")) as fh:
        progs = json.load(fh)
    recorded = {}
    for idx, prog in progs.items():
        vals = []
        for op in prog["ops"]:
            res = None
            if op["op"] == "getattr" and op["obj"].get("g") == "rl":
                res = "rl." + op["name"]
            elif op["op"] == "call":
                fn = op["fn"]
                name = fn["g"] if fn.get("g") in generic else (vals[fn["v"]] if "v" in fn and isinstance(vals[fn["v"]], str) else None)
                if name and (name in generic or name[3:] in rlnames):
                    recorded.setdefault(_shape_key(name, len(op["args"]), list(op["kwargs"])), []).append(idx)
            vals.append(res)
    # shapes no recorded program exercises and the wrappers hand to the original instead (none at present)
    forwarded = set()
    missing = sorted(k for k in sites if k not in recorded and k not in forwarded)
    assert not missing, {k: sites[k][:3] for k in missing}
    assert len(sites) == 19 and sum(len(v) for v in sites.values()) == 128      # (the walk itself: a new reference version moves these)
    assert not [k for k in recorded if k not in sites]


@pytest.mark.parametrize("calendar,start,T,window", [("noleap", "2001-01-01", 365 * 4, 31), ("noleap", "2001-01-01", 365 * 3, 5),
                                                     ("360_day", "2001-01-01", 360 * 3, 15), ("standard", "2001-01-01", 1461, 7),
                                                     ("noleap", "2001-03-01", 365 * 3, 7)])
def test_group_samples_ring_holds_the_rows_of_sample_rows(monkeypatch, calendar, start, T, window):
    """Grouper.group_samples keeps the windowed sample of day-of-year groups as a ring (one row per year replaced from one
    day to the next) when every year holds every day; whatever the route, the matrix of a group must hold exactly the rows
    that sample_rows lists (as a multiset: the order of a sample does not matter) — here with numpy standing in for the
    device (rows carry their own index as value)."""
    from xclim_amd import kernels as K
    from xclim_amd import sdba

    class FakeArr:
        def __init__(self, a):
            self.a, self.shape = a, a.shape

    class FakeDev:
        def empty(self, shape, dtype):
            return FakeArr(np.full(shape, -7.0, dtype=np.float32))

    calls = {"rows": 0}

    def fake_select_rows(dev, x, idx, out=None, out_row=0, out_stride_rows=1):
        idx = np.asarray(idx)
        vals = np.where(idx[:, None] < 0, np.nan, x.a[np.clip(idx, 0, None)])
        calls["rows"] += len(idx)
        if out is None:
            return FakeArr(vals.astype(np.float32))
        out.a[out_row + np.arange(len(idx)) * out_stride_rows] = vals
        return out

    monkeypatch.setattr(K, "select_rows", fake_select_rows)
    ta = TimeAxis.daily(start, T, calendar) if calendar != "standard" else TimeAxis.daily(start, T)
    grp = sdba.Grouper("time.dayofyear", window)
    field = FakeArr(np.arange(T, dtype=np.float32)[:, None])
    expected = grp.sample_rows(ta)
    ngroups = 0
    for g, (smp,) in grp.group_samples(FakeDev(), (field,), ta):
        got = np.sort(np.where(np.isnan(smp.a[:, 0]), -1, smp.a[:, 0]).astype(np.int64))
        np.testing.assert_array_equal(got, np.sort(expected[g]), err_msg=f"group {g}")
        ngroups += 1
    assert ngroups == len(expected)
    full = sum(len(e) for e in expected)
    regular = calendar != "standard" and start.endswith("01-01")
    assert (calls["rows"] < full / 3) == regular      # the ring gathers a fraction; irregular calendars / partial years gather all
