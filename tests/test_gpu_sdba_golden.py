"""The sdba half of the hot path against numbers produced by xsdba itself (tests/golden/make_sdba_golden.py, to be run
where xsdba is installed).  Skipped while tests/golden/sdba_vectors.npz does not exist — the sdba rows stay "parity
unpinned" until then; with the file, every entry of the table in oracle/sdba.py is held to rtol 1e-6."""
import os

import numpy as np
import pytest

from xclim_amd import sdba as xsdba
from xclim_amd.timeaxis import TimeAxis

pytestmark = pytest.mark.gpu
PATH = os.path.join(os.path.dirname(__file__), "golden", "sdba_vectors.npz")
needs_fixture = pytest.mark.skipif(not os.path.exists(PATH), reason="tests/golden/sdba_vectors.npz absent (needs xsdba to generate)")
RTOL = 1e-6


@pytest.fixture(scope="module")
def G():
    return np.load(PATH)


def has(G, key, since):
    """A key the generator writes since its round `since`: a fixture of that format or a later one MUST hold it (a
    missing key used to pass silently); only a fixture older than the key may lack it."""
    if key in G.files:
        return True
    fmt = int(G["format"]) if "format" in G.files else 4
    assert fmt < since, f"{key} is missing from a format-{fmt} fixture: regenerate tests/golden/sdba_vectors.npz"
    return False


@needs_fixture
def test_nodes_and_quantile(dev, G):
    np.testing.assert_allclose(xsdba.equally_spaced_nodes(20), G["nodes_20"], rtol=1e-15)
    np.testing.assert_allclose(xsdba.equally_spaced_nodes(15, eps=1e-6), G["nodes_eps"], rtol=1e-15)
    np.testing.assert_allclose(xsdba.quantile(G["ref"], G["nodes_20"], device=dev), G["quantile"], rtol=RTOL, equal_nan=True)


@needs_fixture
@pytest.mark.parametrize("kind", ["+", "*"])
def test_eqm_qdm_dqm_against_xsdba(dev, G, kind):
    k = "add" if kind == "+" else "mul"
    r, h, s = (G["ref"], G["hist"], G["sim"]) if kind == "+" else (G["pr_ref"], G["pr_hist"], G["pr_sim"])
    eqm = xsdba.EmpiricalQuantileMapping.train(r, h, nquantiles=20, kind=kind, device=dev)
    np.testing.assert_allclose(eqm.hist_q, G[f"eqm_{k}_hist_q"], rtol=RTOL, equal_nan=True)
    np.testing.assert_allclose(eqm.af, G[f"eqm_{k}_af"], rtol=RTOL, atol=1e-6, equal_nan=True)
    for interp in ("nearest", "linear", "cubic"):
        for extrap in ("constant", "nan"):
            got = eqm.adjust(s, interp=interp, extrapolation=extrap)
            np.testing.assert_allclose(got, G[f"eqm_{k}_{interp}_{extrap}"], rtol=2e-6 if interp == "cubic" else RTOL, equal_nan=True,
                                       err_msg=f"{interp} {extrap}")
    qdm = xsdba.QuantileDeltaMapping.train(r, h, nquantiles=20, kind=kind, device=dev)
    for interp in ("nearest", "linear", "cubic"):
        if interp != "cubic" or has(G, f"qdm_{k}_{interp}", 5):   # (cubic: fixtures generated since round 5)
            np.testing.assert_allclose(qdm.adjust(s, interp=interp), G[f"qdm_{k}_{interp}"], rtol=2e-6 if interp == "cubic" else RTOL,
                                       equal_nan=True, err_msg=interp)
    dqm = xsdba.DetrendedQuantileMapping.train(r, h, nquantiles=20, kind=kind, device=dev)
    np.testing.assert_allclose(dqm.scaling, G[f"dqm_{k}_scaling"], rtol=RTOL)
    np.testing.assert_allclose(dqm.af, G[f"dqm_{k}_af"], rtol=RTOL, atol=1e-6, equal_nan=True)
    for deg in (0, 1):
        np.testing.assert_allclose(dqm.adjust(s, interp="nearest", detrend=deg), G[f"dqm_{k}_scen_d{deg}"], rtol=2e-6, equal_nan=True)


@needs_fixture
@pytest.mark.parametrize("group,window", [("time.month", 1), ("time.dayofyear", 31)])
def test_grouped_eqm_against_xsdba(dev, G, group, window):
    tag = group.split(".")[1]
    ta = TimeAxis.daily("2001-01-01", int(G["T"]), "noleap")
    eqm = xsdba.EmpiricalQuantileMapping.train(G["ref"], G["hist"], nquantiles=15, kind="+", group=group, window=window, time=ta, device=dev)
    np.testing.assert_array_equal(eqm.group_labels, G[f"eqmg_{tag}_labels"])
    np.testing.assert_allclose(eqm.hist_q, G[f"eqmg_{tag}_hist_q"], rtol=RTOL, equal_nan=True)
    np.testing.assert_allclose(eqm.af, G[f"eqmg_{tag}_af"], rtol=RTOL, atol=1e-6, equal_nan=True)
    np.testing.assert_allclose(eqm.adjust(G["sim"], interp="nearest", time=ta), G[f"eqmg_{tag}_scen"], rtol=RTOL, equal_nan=True)
    if has(G, f"eqmg_{tag}_scen_linear", 5):   # (fixtures generated before round 5 lack the linear keys)
        np.testing.assert_allclose(eqm.adjust(G["sim"], interp="linear", time=ta), G[f"eqmg_{tag}_scen_linear"], rtol=RTOL, equal_nan=True)
        qdm = xsdba.QuantileDeltaMapping.train(G["ref"], G["hist"], nquantiles=15, kind="+", group=group, window=window, time=ta, device=dev)
        # (month: the regular (quantile, group) grid has no unique Delaunay triangulation — see test_qdm_grouped_matches_oracle)
        tol = dict(rtol=RTOL) if tag == "dayofyear" else dict(rtol=1e-3, atol=0.05)
        np.testing.assert_allclose(qdm.adjust(G["sim"], interp="linear", time=ta), G[f"qdmg_{tag}_scen_linear"], equal_nan=True, **tol)
        if has(G, f"dqmg_{tag}_af", 5):   # DQM with the (windowed) Grouper: round 5
            dqm = xsdba.DetrendedQuantileMapping.train(G["ref"], G["hist"], nquantiles=15, kind="+", group=group, window=window, time=ta,
                                                       device=dev)
            np.testing.assert_allclose(dqm.scaling, G[f"dqmg_{tag}_scaling"], rtol=RTOL)
            np.testing.assert_allclose(dqm.af, G[f"dqmg_{tag}_af"], rtol=RTOL, atol=1e-6, equal_nan=True)
            for interp in ("nearest", "linear"):
                np.testing.assert_allclose(dqm.adjust(G["sim"], interp=interp, detrend=1, time=ta), G[f"dqmg_{tag}_scen_{interp}"],
                                           rtol=2e-6, equal_nan=True, err_msg=interp)
        if tag == "month":
            eqp = xsdba.EmpiricalQuantileMapping.train(G["pr_ref"], G["pr_hist"], nquantiles=15, kind="*", group=group, time=ta, device=dev)
            np.testing.assert_allclose(eqp.adjust(G["pr_sim"], interp="linear", time=ta), G["eqmg_month_pr_scen_linear"], rtol=RTOL,
                                       equal_nan=True)


@needs_fixture
@pytest.mark.parametrize("group", ["time", "time.month"])
def test_adapt_freq_against_xsdba(dev, G, group):
    """xsdba.processing.adapt_freq: pth and dP0 exactly, sim_ad where upstream left the value alone, and the SAME samples
    replaced (the fill values themselves come from different random generators)."""
    from xclim_amd import sdba as xsdba
    from xclim_amd.timeaxis import TimeAxis

    tag = group.split(".")[-1]
    ref, sim, thresh = G["adapt_ref"], G["adapt_sim"], float(G["adapt_thresh"])
    ta = TimeAxis.daily("2001-01-01", ref.shape[0], "noleap")
    got, pth, dp0 = xsdba.adapt_freq(ref, sim, thresh, group=group, time=ta, device=dev)
    np.testing.assert_allclose(dp0, G[f"adapt_{tag}_dP0"].reshape(dp0.shape), rtol=1e-6, equal_nan=True)
    np.testing.assert_allclose(pth, G[f"adapt_{tag}_pth"].reshape(pth.shape), rtol=1e-6, equal_nan=True)
    up = G[f"adapt_{tag}_sim_ad"]
    same_up = (up == sim) | (np.isnan(up) & np.isnan(sim))
    same_me = (got == sim) | (np.isnan(got) & np.isnan(sim))
    np.testing.assert_array_equal(same_me, same_up)
