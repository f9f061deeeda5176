"""debug of test_grouper_add_dims_pools_the_members[time.dayofyear-15-+]: where does the one element differ?"""
import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
from oracle import sdba as osdba
from oracle.timeutil import OTime
from xclim_amd import sdba as xsdba
from xclim_amd._capi import get_device
from xclim_amd.timeaxis import TimeAxis
dev = get_device()
rng = np.random.default_rng(20240925)
kind, group, window = "+", "time.dayofyear", 15
T, R = 365 * 3, 3
ta, ot = TimeAxis.daily("2001-01-01", T, "noleap"), OTime.noleap(2001, T, "noleap")
shape = (T, R, 2, 3)
t = np.arange(T)[:, None, None, None]
seas = 8 * np.sin(2 * np.pi * (t - 100) / 365)
member = np.arange(R)[None, :, None, None] * 0.7
base = 0.0
ref = (base + 10 + seas + member + rng.normal(0, 3, shape)).astype(np.float32)
hist = (base + 11.5 + 1.2 * seas + 2 * member + rng.normal(0, 4, shape)).astype(np.float32)
sim = (base + 12 + 1.2 * seas + 2 * member + rng.normal(0, 4, shape)).astype(np.float32)
sim[rng.random(shape) < 0.01] = np.nan
hist[:40, 1, 0, 0] = np.nan
grp = xsdba.Grouper(group, window, add_dims=1)
mdl = xsdba.EmpiricalQuantileMapping.train(ref, hist, nquantiles=12, kind=kind, group=grp, time=ta, device=dev)
labels = np.unique(osdba.group_values(ot, "dayofyear"))
got = mdl.adjust(sim, time=ta)
got1 = mdl.adjust(sim[:, 1], time=ta)
got_grp = mdl.adjust(sim, time=ta, grouped_nearest="group")
for r in range(R):
    exp = osdba.eqm_adjust_grouped(sim[:, r], ot, "dayofyear", labels, mdl.af, mdl.hist_q, kind, "nearest", "constant", mode="griddata")
    bad = ~np.isclose(got[:, r], exp, rtol=1e-6, atol=1e-6, equal_nan=True)
    for tt, i, j in np.argwhere(bad):
        x, d = sim[tt, r, i, j], ot.doy[tt]
        print("member", r, "t", tt, "cell", (i, j), "doy", d, "x", x, "got", got[tt, r, i, j], "exp", exp[tt, i, j], "no-member-axis", got1[tt, i, j] if r == 1 else None,
              "own-group rule", got_grp[tt, r, i, j])
        hq, af = mdl.hist_q[:, :, i, j].astype(np.float64), mdl.af[:, :, i, j].astype(np.float64)
        G = len(labels)
        for dr in (-2, -1, 0, 1, 2):
            g = (d - 1 + dr) % G
            d2 = (hq[g] - float(x)) ** 2 + dr * dr
            k = int(np.argmin(d2))
            print("   row", d + dr, "nearest node", k, "x", hq[g, k], "d2", d2[k], "af", af[g, k], "scen", float(x) + af[g, k])
