"""Host mirror of the sdba empirical-quantile-mapping path (``xclim.sdba`` == third-party ``xsdba >= 0.4.0``;
reference shim src/xclim/sdba.py:1-28; object API pinned by tests/test_xsdba.py:44-155).

Names follow xsdba: ``EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, kind="+", group="time")`` then
``.adjust(sim, interp="nearest", extrapolation="constant")``, with ``ds.af`` / ``ds.hist_q`` exposed as ``.af`` /
``.hist_q``; ``nbutils.quantile`` and ``utils.equally_spaced_nodes`` as module functions.  Arrays: TIME ON AXIS 0, numpy
or device arrays; all arithmetic is in ``xh_eqm_train`` / ``xh_eqm_adjust``.

Grouping (first slice): :class:`Grouper` ``"time"``, ``"time.month"``, ``"time.season"`` and ``"time.dayofyear"`` with
an odd ``window`` (xsdba: the samples of a group are the centred ``window`` days around every time step of the group —
the same sample sets as ``percentile_doy``).  Training gathers each group's rows (``xh_select_rows``) and runs the
per-column multi-quantile kernels on them; ``af`` / ``hist_q`` get a leading group axis ``(group, quantiles, *cells)``.
``adjust`` maps every time step with the factors of ITS group (rows are permuted group-major once, one ``xh_eqm_adjust``
launch per group on a contiguous row block, one gather back).  For "linear" xsdba interpolates over the (quantile, group)
PLANE (``utils.interp_on_quantiles`` -> ``_interp_on_quantiles_2D``: ``scipy.interpolate.griddata`` with a fractional group
index — the documented standard use, docs/sdba.rst:64-65, CHANGELOG.rst:338): built since round 5 for month / day-of-year
groupings (``xh_plane_linear``: the Delaunay triangle of every step is located by a walk, no triangulation is stored; one
launch over the whole series); "cubic" with a sub-grouping (griddata's Clough-Tocher scheme) raises
``NotImplementedError`` (:func:`_check_group_interp`) — an interpolation along the quantile axis inside each group would
silently differ from it.  "nearest" with a month / day-of-year grouping
follows xsdba since round 4: ``griddata(method="nearest")`` in the (hist_q, group coordinate) plane over the nodes of ALL
groups — where the own group's nearest node is more than one unit away a node of a NEIGHBOURING group can win — and the
own group's end factors outside its nodes (``xh_eqm_adjust_g2d``; ``grouped_nearest="group"`` restores the own-group rule of
rounds 2-3, which "time.season" still uses).

:class:`QuantileDeltaMapping` (``group="time"`` or a sub-grouping: the ranks are then taken inside each group's own steps):
trained like EQM; ``adjust`` looks the factor up at the QUANTILE of every sim value within the sim series itself
(``rank(sim, pct=True)``), so that the simulated change of every quantile is preserved — ``xh_qdm_adjust`` ranks each column
exactly (average ranks, NaN skipped) and interpolates in fp64 ("nearest", "linear"; "cubic" for ``group="time"``: the
not-a-knot spline of the EQM path over the quantile nodes).  :class:`DetrendedQuantileMapping`: ``group="time"`` or a
sub-grouping, with or without a window (with one, the trend is fitted on the centred window mean: ``xh_window_nanmean``),
``Grouper("time", add_dims=...)`` (one trend, fitted on the mean over the pooled members);
with "linear" and a month grouping the scaling is interpolated over the group coordinate like xsdba's ``u.broadcast`` does.

PARITY UNPINNED like everything xsdba (oracle/sdba.py).
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from ._capi import DeviceArray, get_device
from .calendar import _flatten

ADDITIVE, MULTIPLICATIVE = "+", "*"


class Grouper:
    """xsdba.base.Grouper for the groupings of the quantile-mapping path: ``Grouper("time")``, ``Grouper("time.month")``,
    ``Grouper("time.dayofyear", window=31)``.

    ``add_dims`` (xsdba: "additional dimensions that should be reduced in grouping operations", e.g. the realizations of an
    ensemble — /root/reference/docs/sdba.rst:64-66): the host mirrors take numpy arrays, so the dimensions are AXIS NUMBERS of
    the training arrays and must be the axes right behind time (``add_dims=1`` or ``(1, 2)``): their samples are pooled with
    the time steps of a group when the quantiles are taken (``nbutils.quantile(ds.ref, quantiles, dim)`` with
    ``dim = [time, (window), *add_dims]``), the factors have no such axis, and ``adjust`` maps every member of a ``sim`` that
    still has it with the same factors."""

    def __init__(self, group: str = "time", window: int = 1, add_dims=None):
        if isinstance(group, Grouper):
            group, window, add_dims = group.name, group.window, group.add_dims if add_dims is None else add_dims
        if group not in ("time", "time.month", "time.dayofyear", "time.season"):
            raise NotImplementedError(f"group={group!r}: supported are 'time', 'time.month', 'time.dayofyear', 'time.season'")
        if window < 1 or window % 2 == 0:
            raise ValueError("window must be a positive odd number of time steps")
        if group == "time" and window != 1:
            raise ValueError("a window needs a sub-grouping ('time.month' / 'time.dayofyear')")
        self.name, self.window = group, int(window)
        self.prop = group.split(".")[1] if "." in group else "group"
        dims = () if add_dims is None else tuple(int(a) for a in np.atleast_1d(add_dims))
        if dims != tuple(range(1, len(dims) + 1)):
            raise NotImplementedError(f"add_dims={add_dims!r}: the pooled axes must be the ones right behind time (1, 2, ...): "
                                      "move them there (np.moveaxis) first")
        self.add_dims = dims

    def __repr__(self):
        extra = f", add_dims={self.add_dims}" if self.add_dims else ""
        return f"Grouper(name={self.name!r}, window={self.window}{extra})"

    def pool(self, dev, x: DeviceArray, cell_shape):
        """(T, R * C) field whose leading cell axes are ``add_dims`` -> ((T * R, C) view, R, cell shape without them): row
        ``t * R + r`` is member ``r`` of step ``t`` — the C-contiguous layout already is that matrix."""
        k = len(self.add_dims)
        if k == 0:
            return x, 1, tuple(cell_shape)
        if len(cell_shape) < k:
            raise ValueError(f"add_dims={self.add_dims}: the array has no such axes (shape behind time: {tuple(cell_shape)})")
        R = int(np.prod(cell_shape[:k]))
        cells = tuple(cell_shape[k:])
        C_ = int(np.prod(cells)) if cells else 1
        return x.reshape(x.shape[0] * R, C_), R, cells   # (a view that keeps its parent alive: the caller drops `x`)

    def group_samples(self, dev, fields, time, R: int = 1):
        """For every group (in label order) the training sample of each field of ``fields`` ((T * R, C) device matrices) as a
        (rows, C) device matrix: the gathered rows of :meth:`sample_rows`.  The matrices of one group are only valid until the
        next one is asked for.

        Day-of-year groups with a window on a series where every year holds every day (noleap, 360_day ...): the sample of
        day d + 1 is the sample of day d with ONE row per year replaced (the order of a sample's rows does not matter), so
        the matrix is kept as a ring — ``window`` slots per year — and 1 / window of it is gathered per group (30 years,
        window 31: 930 rows -> 30)."""
        rows_of = self.sample_rows(time)
        ring = False
        if self.prop == "dayofyear" and self.window > 1 and R == 1 and len(rows_of) > 1:
            tb = np.asarray(time.doy_table()[0], dtype=np.int64)      # (years, doys) -> time index
            ring = tb.shape[1] == len(rows_of) and tb.min() >= 0 and bool((tb[:, 1:] == tb[:, :-1] + 1).all())
        if not ring:
            for g, rows in enumerate(rows_of):
                rows = self.pooled_rows(rows, R)
                yield g, [K.select_rows(dev, f, rows) for f in fields]
            return
        T, W, half, ny = len(time), self.window, self.window // 2, tb.shape[0]
        bufs = [dev.empty((ny * W, f.shape[1]), np.float32) for f in fields]

        def rows_at(d, off):  # the rows tb[:, d] + off of every year, -1 beyond the series
            r = tb[:, d] + off
            return np.where((r < 0) | (r >= T), -1, r)

        first = np.full(ny * W, -1, dtype=np.int64)
        for off in range(-half, half + 1):
            first[np.arange(ny) * W + (off % W)] = rows_at(0, off)       # slot of day d + off: (d + off) mod W, d = 0
        for f, b in zip(fields, bufs):
            K.select_rows(dev, f, first, out=b)
        yield 0, bufs
        for d in range(1, tb.shape[1]):
            new = rows_at(d, half)                                        # day d + half enters, day d - 1 - half leaves: same slot
            for f, b in zip(fields, bufs):
                K.select_rows(dev, f, new, out=b, out_row=(d + half) % W, out_stride_rows=W)
            yield d, bufs

    def ring_schedule(self, time):
        """The sliding form of the day-of-year samples (see :meth:`group_samples`): ``(rows0, enter, leave)`` — the time steps of
        group 0's sample (years x window, -1 beyond the series) and, per step from group d to d + 1, the steps that enter and
        leave (one per year) — or None when the grouping is not a ring (no window, another property, calendar gaps)."""
        if not (self.prop == "dayofyear" and self.window > 1):
            return None
        tb = np.asarray(time.doy_table()[0], dtype=np.int64)      # (years, doys) -> time index
        if not (tb.shape[1] == len(self.sample_rows(time)) and tb.shape[1] > 1 and tb.min() >= 0
                and bool((tb[:, 1:] == tb[:, :-1] + 1).all())):
            return None
        T, half = len(time), self.window // 2

        def rows_at(d, off):
            r = tb[:, d] + off
            return np.where((r < 0) | (r >= T), -1, r)

        rows0 = np.concatenate([rows_at(0, off) for off in range(-half, half + 1)])
        enter = np.stack([rows_at(d, half) for d in range(1, tb.shape[1])])
        leave = np.stack([rows_at(d - 1, -half) for d in range(1, tb.shape[1])])
        return rows0, enter, leave

    def sliding_stretches(self, time, max_per: int = 64, cap: int = 1024, min_len: int = 8):
        """The general form of :meth:`ring_schedule` for calendars with gaps (leap days: the standard calendar): from the groups'
        sample rows, the stretches of CONSECUTIVE groups whose samples differ by at most ``max_per`` rows each way —
        ``[(g0, rows0, enter, leave), ...]`` with ``enter`` / ``leave`` of shape (groups - 1, per), -1 padded — and the groups
        no stretch covers (day 366 of a standard calendar: only the leap years have it, its sample shares little with its
        neighbours').  A gap-free calendar gives one stretch with the rows of :meth:`ring_schedule`."""
        ring = self.ring_schedule(time)
        rows_of = None
        if ring is not None:
            return [(0,) + ring], [], rows_of
        if not (self.prop == "dayofyear" and self.window > 1):
            return [], None, rows_of
        rows_of = self.sample_rows(time)
        sets = [np.unique(r[r >= 0]) for r in rows_of]
        G = len(sets)
        ent = [np.setdiff1d(sets[g + 1], sets[g], assume_unique=True) for g in range(G - 1)]
        lev = [np.setdiff1d(sets[g], sets[g + 1], assume_unique=True) for g in range(G - 1)]
        ok = [max(len(ent[g]), len(lev[g])) <= max_per and len(sets[g]) <= cap and len(sets[g + 1]) <= cap for g in range(G - 1)]
        stretches, covered, g = [], np.zeros(G, dtype=bool), 0
        while g < G - 1:
            if not ok[g]:
                g += 1
                continue
            e = g
            while e < G - 1 and ok[e]:
                e += 1
            if e - g + 1 >= min_len:   # groups g .. e
                per = max(1, max(max(len(ent[k]), len(lev[k])) for k in range(g, e)))
                en = np.full((e - g, per), -1, dtype=np.int64)
                lv = np.full((e - g, per), -1, dtype=np.int64)
                for k in range(g, e):
                    en[k - g, :len(ent[k])] = ent[k]
                    lv[k - g, :len(lev[k])] = lev[k]
                stretches.append((g, rows_of[g], en, lv))
                covered[g:e + 1] = True
            g = e + 1
        return stretches, [int(k) for k in np.nonzero(~covered)[0]], rows_of

    @staticmethod
    def pooled_rows(rows: np.ndarray, R: int) -> np.ndarray:
        """Sample rows of the time axis -> rows of the pooled (T * R, C) matrix (-1 stays -1, R times)."""
        if R == 1:
            return rows
        out = rows[:, None] * R + np.arange(R)[None, :]
        out[rows < 0] = -1
        return out.reshape(-1)

    _SEASON_OF_MONTH = np.array(["", "DJF", "DJF", "MAM", "MAM", "MAM", "JJA", "JJA", "JJA", "SON", "SON", "SON", "DJF"])

    def values(self, time) -> np.ndarray:
        """The group coordinate of every time step: month, day of year, or the season name (``time.dt.season``)."""
        if self.prop == "month":
            return np.asarray(time.month)
        if self.prop == "season":
            return self._SEASON_OF_MONTH[np.asarray(time.month)]
        return np.asarray(time.doy)

    def coordinate(self, time, interp: bool = False) -> np.ndarray:
        """xsdba ``Grouper.get_index(da, interp=...)``: the group coordinate of every time step as float64.  With
        ``interp=True`` (every interpolation but "nearest") the month becomes fractional — ``month - 0.5 + day /
        days_in_month``, the middle of a month on its integer — and the day of year stays an integer."""
        if self.prop == "month" and interp:
            return time.month - 0.5 + time.day / time.days_in_month()
        if self.prop in ("month", "dayofyear"):
            return np.asarray(self.values(time), dtype=np.float64)
        raise NotImplementedError(f"group coordinate of {self.name!r}")

    def labels(self, time) -> np.ndarray:
        """Group coordinate values present on `time` (months 1..12 / days of year / season names), sorted like xarray's
        groupby sorts them (the seasons alphabetically: DJF, JJA, MAM, SON)."""
        if self.prop == "group":
            return np.array([0])
        return np.unique(self.values(time))

    def index(self, time, labels=None) -> np.ndarray:
        """Position of every time step's group in `labels` (default: the labels of `time`); -1 when absent."""
        if self.prop == "group":
            return np.zeros(len(time), dtype=np.int64)
        lab = self.labels(time) if labels is None else np.asarray(labels)
        val = self.values(time)
        pos = np.clip(np.searchsorted(lab, val), 0, len(lab) - 1)
        return np.where(lab[pos] == val, pos, -1)

    def small_groups(self, time, most: int = 64):
        """``(rows, offs)`` — the time steps of group 0, then of group 1, ... and where each group starts — for a sub-grouping
        WITHOUT a window whose groups hold at most ``most`` steps (a day-of-year grouping: one per year), else None: the
        shape of the one-launch training kernels (xh_eqm_train_groups)."""
        if self.prop == "group" or self.window != 1:
            return None
        gi = self.index(time)
        G = len(self.labels(time))
        counts = np.bincount(gi[gi >= 0], minlength=G)
        if counts.max(initial=0) > most:
            return None
        order = np.argsort(gi, kind="stable")
        return order[gi[order] >= 0].astype(np.int64), np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)

    def sample_rows(self, time) -> list:
        """For every group the rows of its training sample: the centred window around each of its time steps, -1 (NaN)
        beyond the ends of the series — rolling(time=window, center=True).construct + groupby in xsdba."""
        T = len(time)
        gi = self.index(time)
        half = self.window // 2
        off = np.arange(-half, half + 1)
        out = []
        for g in range(len(self.labels(time))):
            t = np.nonzero(gi == g)[0]
            rows = (t[:, None] + off[None, :]).reshape(-1)
            rows[(rows < 0) | (rows >= T)] = -1
            out.append(rows.astype(np.int64))
        return out


def _check_group_interp(group: "Grouper", interp: str, who: str, labels=None, extrapolation: str = "constant") -> None:
    """Sub-groupings interpolate over the (quantile, group) PLANE in xsdba when interp != "nearest"
    (``utils.interp_on_quantiles`` -> ``_interp_on_quantiles_2D`` = ``scipy.interpolate.griddata``).  Built: "linear"
    (Delaunay interpolation, ``xh_plane_linear``) for month / day-of-year groupings whose labels are 1 .. G, with
    ``extrapolation="constant"``.  Refused loudly: "cubic" (griddata's Clough-Tocher scheme estimates gradients by a global
    iteration over the whole triangulation), "time.season" (upstream's season coordinate is not restated),
    ``extrapolation="nan"`` with "linear" (upstream then skips its extrapolation step and keeps griddata's own NaN outside
    the convex hull of ALL nodes — a different region than the per-group bounds)."""
    if group.prop == "group" or interp == "nearest":
        return
    if interp == "cubic":
        raise NotImplementedError(
            f"{who}: interp='cubic' with group={group.name!r} is scipy griddata's Clough-Tocher interpolation over the "
            "(quantile, group) plane in xsdba — not built; use 'linear' or 'nearest' (or group='time')")
    lab = None if labels is None else np.asarray(labels)
    if group.prop not in ("month", "dayofyear") or lab is None or not np.array_equal(lab, np.arange(1, len(lab) + 1)):
        raise NotImplementedError(
            f"{who}: interp={interp!r} over the (quantile, group) plane is built for 'time.month' / 'time.dayofyear' groupings "
            f"whose labels are 1 .. G (got group={group.name!r})")
    if extrapolation != "constant":
        raise NotImplementedError(
            f"{who}: interp={interp!r} with a sub-grouping is built for extrapolation='constant' (with 'nan' xsdba keeps "
            "griddata's NaN outside the convex hull of all nodes)")


def equally_spaced_nodes(n: int, eps=None) -> np.ndarray:
    """xsdba.utils.equally_spaced_nodes: n nodes q_i = (i + 1/2) / n; with ``eps`` the end points eps and 1 - eps are
    added (n + 2 nodes), so that the adjustment factors are also defined near the ends of the distribution."""
    dq = 1.0 / n / 2.0
    q = np.linspace(dq, 1.0 - dq, n)
    if eps is None:
        return q
    return np.insert(np.append(q, 1.0 - eps), 0, eps)


def quantile(da, q, dim="time", *, device=None, keep=False):
    """xsdba.nbutils.quantile along time: (nq, *cells) in float32."""
    if dim != "time":
        raise NotImplementedError("quantile: only dim='time' (axis 0) is supported on the HIP path")
    dev = device or get_device()
    x, cell_shape = _flatten(da, dev)
    out = K.quantile_series(dev, x, np.asarray(q, dtype=np.float64))
    return out if keep else out.get().reshape((out.shape[0],) + tuple(cell_shape))


def adapt_freq(ref, sim, thresh: float, *, group="time", window: int | None = None, time=None, seed: int = 0, device=None,
               keep=False):
    """xsdba.processing.adapt_freq (``_processing._adapt_freq``, Themeßl et al. 2012): adapt the frequency of values
    ``<= thresh`` in ``sim`` to that of ``ref`` — the pre-processing a multiplicative quantile mapping of precipitation
    needs when the model has too many dry days.  Per cell and group:

        P0 = count(x <= thresh) / count(valid)   for sim and ref (over the group's windowed sample)
        dP0 = (P0_sim - P0_ref) / P0_sim         the share of sim's values <= thresh that has to become wet
        pth = quantile(ref, P0_sim)  where dP0 > 0  (``nbutils.vecquantiles``: the value of ref at sim's dry-day rank)
        sim_ad = sim, except: where dP0 >= 0 the samples whose percentage rank (average ranks over the group's own time
                 steps) lies in [P0_ref, P0_sim] get a uniform random value in [thresh, pth)

    Returns ``(sim_ad, pth, dP0)``: sim_ad float32 in the shape of sim, pth / dP0 per cell (``(groups, *cells)`` with a
    sub-grouping).  ``thresh`` in the units of the data.  The random numbers: upstream draws from numpy's global
    generator (not reproducible across runs); here a counter-based uniform keyed by ``(seed, time step, cell)`` —
    ``xh_adapt_freq``, restated in oracle/sdba.py.  ``ref`` and ``sim`` need the same number of time steps only with a
    sub-grouping (one ``time`` axis for both).  Parity unpinned (xsdba is not in the reference tree)."""
    grp = group if isinstance(group, Grouper) else Grouper(group, 1 if window is None else window)
    if grp.add_dims:
        raise NotImplementedError("adapt_freq with Grouper(add_dims=...) is not built")
    dev = device or get_device()
    r, cell_shape = _flatten(ref, dev)
    s_, cell_shape_s = _flatten(sim, dev)
    if tuple(cell_shape) != tuple(cell_shape_s):
        raise ValueError("ref and sim must have the same grid")
    C_ = s_.shape[1]
    thresh = float(thresh)

    def one_group(ref_sample, sim_sample, sim_main, tindex, out):
        """P0 / dP0 / pth from the (windowed) samples, the replacement on the group's own steps"""
        def p0(x):
            seg = np.array([0, x.shape[0]], dtype=np.int64)
            cnt, val = K.threshold_count(dev, x, "<=", seg, scalar=thresh)
            cnt, val = cnt.get()[0].astype(np.float64), val.get()[0].astype(np.float64)
            with np.errstate(all="ignore"):
                return cnt / val
        p0_sim, p0_ref = p0(sim_sample), p0(ref_sample)
        with np.errstate(all="ignore"):
            dp0 = (p0_sim - p0_ref) / p0_sim
        pth = K.quantile_cells(dev, ref_sample, p0_sim).get()
        pth = np.where(dp0 > 0, pth, np.float32(np.nan)).astype(np.float32)
        K.adapt_freq(dev, sim_main, p0_ref, p0_sim, dp0, pth, thresh, seed, tindex=tindex, out=out)
        return pth, dp0

    if grp.prop == "group":
        out = dev.empty(tuple(s_.shape), np.float32)
        pth, dp0 = one_group(r, s_, s_, None, out)
        res = out if keep else out.get().reshape((s_.shape[0],) + tuple(cell_shape))
        return res, pth.reshape(cell_shape), dp0.reshape(cell_shape)
    if time is None or len(time) != s_.shape[0] or r.shape[0] != s_.shape[0]:
        raise ValueError(f"group={grp.name!r} needs time=TimeAxis common to ref and sim")
    gi = grp.index(time)
    labels = grp.labels(time)
    T = s_.shape[0]
    # group-major blocks (every group's own steps contiguous), one gather back at the end — like the grouped adjust
    perm = np.argsort(gi, kind="stable")
    counts = np.bincount(gi, minlength=len(labels))
    s_perm = K.select_rows(dev, s_, perm)
    ad_perm = dev.empty((T, C_), np.float32)
    pths, dp0s, off = [], [], 0
    for g, rows in enumerate(grp.sample_rows(time)):
        n = int(counts[g])
        main = perm[off:off + n]
        sim_main = dev.wrap(s_perm.ptr + off * C_ * 4, (n, C_), np.float32)
        blk = dev.wrap(ad_perm.ptr + off * C_ * 4, (n, C_), np.float32)
        pth, dp0 = one_group(K.select_rows(dev, r, rows), K.select_rows(dev, s_, rows), sim_main, main, blk)
        pths.append(pth)
        dp0s.append(dp0)
        off += n
    inv = np.empty(T, dtype=np.int64)
    inv[perm] = np.arange(T)
    out = K.select_rows(dev, ad_perm, inv)
    dev.sync()
    res = out if keep else out.get().reshape((T,) + tuple(cell_shape))
    G = len(labels)
    return res, np.stack(pths).reshape((G,) + tuple(cell_shape)), np.stack(dp0s).reshape((G,) + tuple(cell_shape))


class EmpiricalQuantileMapping:
    """Empirical quantile mapping bias adjustment (train on ref/hist quantiles, adjust sim by node search)."""

    def __init__(self, dev, af: DeviceArray, hist_q: DeviceArray, quantiles, kind, cell_shape, group=None, labels=None):
        self._dev = dev
        self._af, self._hist_q = af, hist_q
        self.quantiles = np.asarray(quantiles)
        self.kind = kind
        self.cell_shape = tuple(cell_shape)
        self.group = group or Grouper("time")
        self.group_labels = np.array([0]) if labels is None else np.asarray(labels)
        self.adj_params = {"group": self.group.name if self.group.window == 1 else repr(self.group), "kind": kind,
                           "nquantiles": len(self.quantiles)}

    def _for_members(self, sim):
        """A ``sim`` with extra axes right behind time — the members that ``Grouper(add_dims=...)`` pooled in training, or any
        other axis the factors do not have (xsdba: the factors broadcast against it in ``interp_on_quantiles``): the model
        whose tables are repeated for every member, so that the (time, members x cells) matrix goes through the same
        kernels.  None when sim has the trained shape."""
        shp = tuple(sim.shape[1:])
        n = len(self.cell_shape)
        if shp == self.cell_shape or len(shp) <= n or (n and shp[len(shp) - n:] != self.cell_shape):
            return None
        extra = shp[:len(shp) - n]
        E = int(np.prod(extra))
        cache = self.__dict__.setdefault("_member_models", {})
        if extra not in cache:
            dev = self._dev

            def tile(tab):  # (..., C) -> (..., E * C): row i of the (rows, C) view repeated E times IS row i of (rows, E * C)
                lead, C_ = int(np.prod(tab.shape[:-1])), tab.shape[-1]
                out = K.select_rows(dev, tab.reshape(lead, C_), np.repeat(np.arange(lead), E))
                return out.reshape(*(tuple(tab.shape[:-1]) + (E * C_,)))

            cache[extra] = type(self)(dev, tile(self._af), tile(self._hist_q), self.quantiles, self.kind, extra + self.cell_shape,
                                      self.group, None if self.group.prop == "group" else self.group_labels)
        return cache[extra]

    @classmethod
    def train(cls, ref, hist, *, nquantiles=20, kind: str = ADDITIVE, group="time", window: int | None = None, time=None,
              device=None, adapt_freq_thresh: float | None = None, adapt_freq_seed: int = 0):
        """``group``: "time" (default), "time.month", "time.dayofyear" or a :class:`Grouper`; sub-groupings need the
        common ``time`` axis (TimeAxis) of ref and hist.  ``adapt_freq_thresh`` (in the units of the data): hist goes
        through :func:`adapt_freq` against ref (same grouping) before the quantiles are taken — xsdba's
        ``EmpiricalQuantileMapping.train(adapt_freq_thresh=...)`` (``_adjustment.eqm_train`` -> ``_adapt_freq_hist``)."""
        grp = group if isinstance(group, Grouper) else Grouper(group, 1 if window is None else window)
        if kind not in (ADDITIVE, MULTIPLICATIVE):
            raise ValueError(f"kind must be '+' or '*', got {kind!r}")
        dev = device or get_device()
        r, cell_shape = _flatten(ref, dev)
        h, cell_shape_h = _flatten(hist, dev)
        if tuple(cell_shape) != tuple(cell_shape_h) or r.shape != h.shape:
            raise ValueError("ref and hist must have the same shape")  # _check_matching_time_sizes analogue
        if adapt_freq_thresh is not None:
            if grp.add_dims:
                raise NotImplementedError("adapt_freq_thresh with Grouper(add_dims=...) is not built")
            h, _, _ = adapt_freq(r, h, adapt_freq_thresh, group=grp, time=time, seed=adapt_freq_seed, device=dev, keep=True)
        q = equally_spaced_nodes(nquantiles) if np.isscalar(nquantiles) else np.asarray(nquantiles, dtype=np.float64)
        T = r.shape[0]
        # Grouper(add_dims=...): the members' samples are pooled with the time steps (rows t * R + r of a (T * R, C) view)
        r, R, cell_shape = grp.pool(dev, r, cell_shape)
        h, _, _ = grp.pool(dev, h, cell_shape_h)
        if grp.prop == "group":
            af, hq = K.eqm_train(dev, r, h, q, kind)
            return cls(dev, af, hq, q, kind, cell_shape, grp)
        if time is None or len(time) != T:
            raise ValueError(f"group={grp.name!r} needs time=TimeAxis of the training series")
        labels = grp.labels(time)
        G, C_ = len(labels), r.shape[1]
        af = dev.empty((G, len(q), C_), np.float32)
        hq = dev.empty((G, len(q), C_), np.float32)
        plane = len(q) * C_ * 4
        # day-of-year groups with a window on gap-free years: every cell keeps its window sorted from one day to the next
        # (xh_eqm_train_window, round 6: 508 -> ~60 ms for 30 years x 1440 x 90) — bit-identical to the per-group selection below
        small = grp.small_groups(time) if R == 1 else None
        if small is not None:
            # no window, small groups (365 days of the year x one row per year): all groups in ONE launch per field, keys in
            # registers (xh_eqm_train_groups, round 6: 121 -> ~10 ms for 30 years x 1440 x 90) — bit-identical to the loop below
            res = K.eqm_train_groups(dev, r, h, small[0], small[1], q, kind)
            if res is not None:
                dev.sync()
                return cls(dev, res[0], res[1], q, kind, cell_shape, grp, labels)
        stretches, rest, rows_of = grp.sliding_stretches(time) if R == 1 else ([], None, None)
        if stretches:
            def slab(a, g0, n):
                return dev.wrap(a.ptr + g0 * plane, (n, len(q), C_), np.float32)

            done = True
            for g0, rows0, en, lv in stretches:
                n = en.shape[0] + 1
                if K.eqm_train_window(dev, r, h, rows0, en, lv, q, kind, out=(slab(af, g0, n), slab(hq, g0, n))) is None:
                    done = False   # (not the kernel's shape after all: everything through the per-group path below)
                    break
            if done:
                for g in rest:     # the groups no stretch covers (day 366 of a standard calendar): selected from their gathered sample
                    K.eqm_train(dev, K.select_rows(dev, r, rows_of[g]), K.select_rows(dev, h, rows_of[g]), q, kind,
                                out=(slab(af, g, 1).reshape(len(q), C_), slab(hq, g, 1).reshape(len(q), C_)))
                dev.sync()
                return cls(dev, af, hq, q, kind, cell_shape, grp, labels)
        for g, (rg, hg) in grp.group_samples(dev, (r, h), time, R):
            out_g = tuple(dev.wrap(a.ptr + g * plane, (len(q), C_), np.float32) for a in (af, hq))
            K.eqm_train(dev, rg, hg, q, kind, out=out_g)
        dev.sync()
        return cls(dev, af, hq, q, kind, cell_shape, grp, labels)

    def adjust(self, sim, *, interp: str = "nearest", extrapolation: str = "constant", time=None, keep=False,
               grouped_nearest: str = "griddata"):
        """``grouped_nearest`` (sub-groupings only): "griddata" (default) = what xsdba does — the nearest node in the (hist_q,
        group) PLANE over the nodes of all groups (``_interp_on_quantiles_2D``: a neighbouring group's node wins where the own
        group's nearest node is more than one unit away), own-group factors outside the own group's nodes; "group" = always
        the nearest node of the step's own group (rounds 2-3).  "time.season", groupings whose labels are not 1 .. G and
        models with more than 32 quantile nodes (a warning is raised) take "group" (upstream's season coordinate is not
        restated; the plane kernel keeps a group's nodes in registers).

        ``interp="linear"`` with a month / day-of-year grouping is xsdba's interpolation over the (quantile, group) plane
        (``xh_plane_linear``; :func:`_check_group_interp`): months at their fractional coordinate, days of year on their row."""
        if interp not in ("nearest", "linear", "cubic"):
            raise ValueError(f"interp={interp!r} not in ('nearest', 'linear', 'cubic')")
        if grouped_nearest not in ("griddata", "group"):
            raise ValueError("grouped_nearest must be 'griddata' or 'group'")
        _check_group_interp(self.group, interp, "EmpiricalQuantileMapping.adjust", self.group_labels, extrapolation)
        members = self._for_members(sim)
        if members is not None:
            return members.adjust(sim, interp=interp, extrapolation=extrapolation, time=time, keep=keep, grouped_nearest=grouped_nearest)
        s, cell_shape = _flatten(sim, self._dev)
        if tuple(cell_shape) != self.cell_shape:
            raise ValueError("sim does not match the trained grid")
        dev = self._dev
        if self.group.prop == "group":
            scen = K.eqm_adjust(dev, s, self._af, self._hist_q, self.kind, interp, extrapolation)
            return scen if keep else scen.get().reshape((s.shape[0],) + self.cell_shape)
        if time is None or len(time) != s.shape[0]:
            raise ValueError(f"group={self.group.name!r} needs time=TimeAxis of sim")
        gi = self.group.index(time, self.group_labels)
        if (gi < 0).any():
            raise ValueError("sim holds time steps whose group was not trained (e.g. day 366 with a 365-day training set)")
        if interp == "linear":
            # xsdba's 2-D branch: ONE launch over the whole series, every step at its (fractional) group coordinate
            scen = K.plane_linear(dev, s, self.group.coordinate(time, interp=True), self._af, xq_all=self._hist_q, kind=self.kind)
            dev.sync()
            return scen if keep else scen.get().reshape((s.shape[0],) + self.cell_shape)
        plane2d = self._plane_nearest(grouped_nearest)
        if plane2d:
            # xsdba's nearest node in the (hist_q, group) plane: ONE launch over the whole series (round 5; rounds 3-4 permuted
            # the rows group-major and launched xh_eqm_adjust_g2d per group: 145 ms against 15 for a 30-year 1440 x 90 band
            # with 365 groups)
            scen = K.plane_nearest(dev, s, gi + 1.0, self._af, self._hist_q, self.kind, extrapolation)
            dev.sync()
            return scen if keep else scen.get().reshape((s.shape[0],) + self.cell_shape)
        # group-major permutation of the rows: every group becomes one contiguous block
        perm = np.argsort(gi, kind="stable")
        counts = np.bincount(gi, minlength=len(self.group_labels))
        T, C_ = s.shape
        nq = len(self.quantiles)
        s_perm = K.select_rows(dev, s, perm)
        scen_perm = dev.empty((T, C_), np.float32)
        off = 0
        for g, n in enumerate(counts):
            if n == 0:
                continue
            blk = dev.wrap(s_perm.ptr + off * C_ * 4, (int(n), C_), np.float32)
            out = dev.wrap(scen_perm.ptr + off * C_ * 4, (int(n), C_), np.float32)
            if plane2d:
                K.eqm_adjust_g2d(dev, blk, self._af, self._hist_q, g + 1, self.kind, extrapolation, out=out)
            else:
                af_g = dev.wrap(self._af.ptr + g * nq * C_ * 4, (nq, C_), np.float32)
                hq_g = dev.wrap(self._hist_q.ptr + g * nq * C_ * 4, (nq, C_), np.float32)
                K.eqm_adjust(dev, blk, af_g, hq_g, self.kind, interp, extrapolation, out=out)
            off += int(n)
        inv = np.empty(T, dtype=np.int64)
        inv[perm] = np.arange(T)
        scen = K.select_rows(dev, scen_perm, inv)
        dev.sync()
        return scen if keep else scen.get().reshape((T,) + self.cell_shape)

    def _plane_nearest(self, grouped_nearest: str) -> bool:
        """xsdba's 2-D nearest applies: a month / day-of-year grouping whose labels are 1 .. G (the coordinates upstream's
        add_cyclic_bounds extends to 0 and G + 1) and at most 32 nodes."""
        lab = self.group_labels
        ok = (grouped_nearest == "griddata" and self.group.prop in ("month", "dayofyear")
              and np.array_equal(lab, np.arange(1, len(lab) + 1)))
        if ok and len(self.quantiles) > 32:
            import warnings

            warnings.warn(f"grouped interp='nearest' with {len(self.quantiles)} > 32 quantile nodes: xh_eqm_adjust_g2d keeps a "
                          "group's nodes in registers (<= 32), so every step takes the nearest node of its OWN group "
                          "(grouped_nearest='group') instead of xsdba's nearest node in the (quantile, group) plane",
                          stacklevel=3)
            return False
        return ok

    def _shape(self):
        lead = (len(self.quantiles),) if self.group.prop == "group" else (len(self.group_labels), len(self.quantiles))
        return lead + self.cell_shape

    @property
    def af(self) -> np.ndarray:
        return self._af.get().reshape(self._shape())

    @property
    def hist_q(self) -> np.ndarray:
        return self._hist_q.get().reshape(self._shape())


class QuantileDeltaMapping(EmpiricalQuantileMapping):
    """Quantile delta mapping (xsdba.QuantileDeltaMapping): the same training as EQM (incl. its groupings), the adjustment
    factor of a sim value is taken at ITS quantile in the sim series: ``sim_q = rank(sim, pct=True)``,
    ``af = interp_on_quantiles(sim_q, quantiles, af)``, ``scen = sim (+|*) af``.  With a sub-grouping the ranks are taken
    inside each group's own time steps (xsdba: ``group.apply(rank, sim, main_only=True)`` — the window only widens the
    TRAINING sample) and every step uses the factors of its group ("nearest"), or the factors interpolated over the
    (quantile, group) plane ("linear": see :meth:`adjust`).  ``interp="cubic"`` is built for ``group="time"`` (a spline
    over the quantile nodes, <= 32 of them); with a sub-grouping it is xsdba's 2-D griddata "cubic": refused.

    Series of up to 32768 steps are ranked inside one workgroup (``xh_qdm_adjust``: keys in registers); longer ones
    (1950-2100 daily = 55 152 steps) go through a global sort in column batches (qdm3.hip) — exact, not tuned.  -0.0
    and +0.0 tie, as in
    ``scipy.stats.rankdata``."""

    def adjust(self, sim, *, interp: str = "nearest", extrapolation: str = "constant", time=None, keep=False):
        if interp not in ("nearest", "linear", "cubic"):
            raise ValueError(f"interp={interp!r} not in ('nearest', 'linear', 'cubic')")
        _check_group_interp(self.group, interp, "QuantileDeltaMapping.adjust", self.group_labels, extrapolation)
        members = self._for_members(sim)
        if members is not None:   # (the ranks are taken along time only — group.apply(rank, sim, main_only=True): per member)
            return members.adjust(sim, interp=interp, extrapolation=extrapolation, time=time, keep=keep)
        dev = self._dev
        s, cell_shape = _flatten(sim, dev)
        if tuple(cell_shape) != self.cell_shape:
            raise ValueError("sim does not match the trained grid")
        if self.group.prop == "group" and interp == "cubic":
            # interp_on_quantiles(sim_q, quantiles, af, method="cubic"): scipy interp1d(kind="cubic") over the quantile nodes —
            # the rank kernels give sim_q (kind "factor" on a table whose factors are the nodes themselves, "linear": the
            # rank itself inside the node range; outside it the first / last node for "constant" — where the spline returns
            # the end factor exactly as interp1d's fill_value does — or NaN), the not-a-knot spline of the EQM path
            # (xh_eqm_adjust, interp 2) evaluates the factor there, xh_apply_factor puts it onto sim
            if len(self.quantiles) > 32:
                raise NotImplementedError("QuantileDeltaMapping.adjust: interp='cubic' supports at most 32 quantile nodes")
            qrows = dev.to_device(np.repeat(self.quantiles.astype(np.float32)[:, None], s.shape[1], axis=1))
            sim_q = K.qdm_adjust(dev, s, qrows, self.quantiles, "factor", "linear", extrapolation)
            af_t = K.eqm_adjust(dev, sim_q, self._af, qrows, "factor", "cubic", extrapolation)
            scen = K.apply_factor(dev, s, af_t, self.kind, out=sim_q)
            return scen if keep else scen.get().reshape((s.shape[0],) + self.cell_shape)
        if self.group.prop == "group":
            scen = K.qdm_adjust(dev, s, self._af, self.quantiles, self.kind, interp, extrapolation)
            return scen if keep else scen.get().reshape((s.shape[0],) + self.cell_shape)
        if time is None or len(time) != s.shape[0]:
            raise ValueError(f"group={self.group.name!r} needs time=TimeAxis of sim")
        gi = self.group.index(time, self.group_labels)
        if (gi < 0).any():
            raise ValueError("sim holds time steps whose group was not trained (e.g. day 366 with a 365-day training set)")
        perm = np.argsort(gi, kind="stable")  # group-major: every group one contiguous row block, ranked on its own
        counts = np.bincount(gi, minlength=len(self.group_labels))
        T, C_ = s.shape
        nq = len(self.quantiles)
        if not (interp == "linear" and self.group.prop == "month"):
            # small groups (a day-of-year grouping: one step per year, up to 64): every group ranked in registers, ONE launch over
            # the series where it lies (xh_qdm_adjust_groups, round 6: 80 -> ~8 ms for 365 groups of a 30-year 1440 x 90 band;
            # the loop below gathers and launches per group) — bit-identical
            offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            scen = K.qdm_adjust_groups(dev, s, perm, offs, self._af, self.quantiles, self.kind, interp, extrapolation)
            if scen is not None:
                dev.sync()
                return scen if keep else scen.get().reshape((T,) + self.cell_shape)
        s_perm = K.select_rows(dev, s, perm)
        scen_perm = dev.empty((T, C_), np.float32)
        # "linear" over the (quantile, group) plane (xsdba: interp_on_quantiles(sim_q, quantiles, af) with the quantile
        # nodes THEMSELVES as abscissa, the same in every group: a regular grid).  Day of year: the group coordinate is an
        # integer, every query lies ON its group's row, and the row's edges belong to every Delaunay triangulation of a grid
        # whose node spacing (1 / nq) is below the row spacing (1) — the plane interpolation IS the 1-D linear interpolation
        # inside the own group: the loop below.  Month: the coordinate is fractional; the percentage ranks come from the
        # same rank kernels (kind "factor" on a table whose "factors" are the nodes themselves: sim_q clamped to the node
        # range, which is all the plane needs — outside it the row-interpolated end factor applies), then one plane launch.
        plane = interp == "linear" and self.group.prop == "month"
        qrows = None
        if plane:
            qrows = dev.to_device(np.repeat(self.quantiles.astype(np.float32)[:, None], C_, axis=1))
        off = 0
        for g, n in enumerate(counts):
            if n == 0:
                continue
            blk = dev.wrap(s_perm.ptr + off * C_ * 4, (int(n), C_), np.float32)
            out = dev.wrap(scen_perm.ptr + off * C_ * 4, (int(n), C_), np.float32)
            if plane:
                K.qdm_adjust(dev, blk, qrows, self.quantiles, "factor", "linear", "constant", out=out)
            else:
                af_g = dev.wrap(self._af.ptr + g * nq * C_ * 4, (nq, C_), np.float32)
                K.qdm_adjust(dev, blk, af_g, self.quantiles, self.kind, interp, extrapolation, out=out)
            off += int(n)
        inv = np.empty(T, dtype=np.int64)
        inv[perm] = np.arange(T)
        scen = K.select_rows(dev, scen_perm, inv)
        if plane:  # scen holds sim_q so far
            scen = K.plane_linear(dev, scen, self.group.coordinate(time, interp=True), self._af, xq_common=self.quantiles, base=s,
                                  kind=self.kind)
        dev.sync()
        return scen if keep else scen.get().reshape((T,) + self.cell_shape)


class DetrendedQuantileMapping(EmpiricalQuantileMapping):
    """Detrended quantile mapping (xsdba.DetrendedQuantileMapping; Cannon et al. 2015).  ``group="time"``, or a
    sub-grouping ("time.month", "time.season", "time.dayofyear": everything below happens per group, the trend is fitted
    over the group's own steps on their time coordinate; with a ``window`` the training sample of a group is the windowed
    one and the trend is fitted on the centred window mean of the scaled series, as ``PolyDetrend`` does).

    train (``dqm_train``): ref and hist are normalised by their time means (``x - mean`` for "+", ``x / mean`` for "*"),
    ``af`` / ``hist_q`` come from the quantiles of the NORMALISED series, ``scaling = mean(ref) - mean(hist)`` (resp. the
    ratio).  adjust (``dqm_adjust``): ``sim`` is shifted by ``scaling``, its polynomial trend of degree ``detrend`` (0 or 1,
    ``PolyDetrend``: least squares over the valid steps) is removed, the detrended series goes through the EQM node
    lookup, the trend is put back.  Every stage is float32 on the device with float64 arithmetic inside the kernels
    (means, fit, trend evaluation).  PARITY UNPINNED (oracle/sdba.py: dqm_*)."""

    def __init__(self, *args, scaling=None, **kw):
        super().__init__(*args, **kw)
        self._scaling = scaling  # (C,) float64 device array

    @classmethod
    def train(cls, ref, hist, *, nquantiles=20, kind: str = ADDITIVE, group="time", window=None, time=None, device=None):
        grp = group if isinstance(group, Grouper) else Grouper(group, 1 if window is None else window)
        if grp.add_dims and grp.prop != "group":
            raise NotImplementedError("DetrendedQuantileMapping with Grouper(add_dims=...) is built for group='time' only")
        if kind not in (ADDITIVE, MULTIPLICATIVE):
            raise ValueError(f"kind must be '+' or '*', got {kind!r}")
        dev = device or get_device()
        r, cell_shape = _flatten(ref, dev)
        h, cell_shape_h = _flatten(hist, dev)
        if tuple(cell_shape) != tuple(cell_shape_h) or r.shape != h.shape:
            raise ValueError("ref and hist must have the same shape")
        q = equally_spaced_nodes(nquantiles) if np.isscalar(nquantiles) else np.asarray(nquantiles, dtype=np.float64)
        inv = "-" if kind == ADDITIVE else "/"
        # Grouper(add_dims=...): means and quantiles over the time steps AND the members (ds.ref.mean(dim), dim = [time, *add_dims])
        r, _, cell_shape = grp.pool(dev, r, cell_shape)
        h, _, _ = grp.pool(dev, h, cell_shape_h)

        def one(rg, hg, out=None):
            mu_r, _ = K.poly_trend(dev, rg, 0)
            mu_h, _ = K.poly_trend(dev, hg, 0)
            af, hq = K.eqm_train(dev, K.trend_apply(dev, rg, mu_r, None, inv), K.trend_apply(dev, hg, mu_h, None, inv), q, kind, out=out)
            # scaling = get_correction(mu_hist, mu_ref): a (C,) table — O(C) host arithmetic on the two mean vectors
            mr, mh = mu_r.get(), mu_h.get()
            with np.errstate(all="ignore"):
                return af, hq, (mr - mh if kind == ADDITIVE else mr / mh)

        if grp.prop == "group":
            af, hq, scaling = one(r, h)
            return cls(dev, af, hq, q, kind, cell_shape, grp, scaling=dev.to_device(np.ascontiguousarray(scaling), dtype=np.float64))
        if time is None or len(time) != r.shape[0]:
            raise ValueError(f"group={grp.name!r} needs time=TimeAxis of the training series")
        labels = grp.labels(time)
        G, C_ = len(labels), r.shape[1]
        af = dev.empty((G, len(q), C_), np.float32)
        hq = dev.empty((G, len(q), C_), np.float32)
        plane = len(q) * C_ * 4
        # day-of-year groups with a window on gap-free years: the sorted window of the EQM training (xh_eqm_train_window) also
        # carries the mean of every group's sample and normalises the samples it picks (xh_dqm_train_window, round 6: 3 970 ->
        # ~190 ms for 30 years x 1440 x 90; per group otherwise: a gather, two means, two normalisations, two selections and a
        # host round trip of the means)
        small = grp.small_groups(time)
        if small is not None:   # (no window, small groups: one launch per field — xh_dqm_train_groups, 470 -> ~10 ms)
            res = K.eqm_train_groups(dev, r, h, small[0], small[1], q, kind, normalised=True)
            if res is not None:
                dev.sync()
                return cls(dev, res[0], res[1], q, kind, cell_shape, grp, labels, scaling=res[2])
        if grp.window == 1:
            # no window: every step is in exactly one group, so the group means are ONE xh_poly_trend_groups per field over the
            # series where it lies (summed in time order: what xh_poly_trend gives on the gathered block), the normalisation is
            # ONE xh_trend_apply_groups per field, and the tables are the grouped EQM training of the normalised series —
            # bit-identical to the per-group loop below (month groups of a 30-year 1440 x 90 band: 128 -> ~40 ms)
            gi = grp.index(time, labels)
            order = np.argsort(gi, kind="stable")
            offs = np.concatenate([[0], np.cumsum(np.bincount(gi, minlength=G))]).astype(np.int64)
            zero_u = dev.zeros((r.shape[0],), np.float64)
            mu_r, _ = K.poly_trend_groups(dev, r, order, offs, zero_u, 0)
            mu_h, _ = K.poly_trend_groups(dev, h, order, offs, zero_u, 0)
            rn = K.trend_apply_groups(dev, r, order, offs, mu_r, None, inv)
            hn = K.trend_apply_groups(dev, h, order, offs, mu_h, None, inv)
            eq = EmpiricalQuantileMapping.train(rn, hn, nquantiles=q, kind=kind, group=grp, time=time, device=dev)
            mr, mh = mu_r.get(), mu_h.get()
            with np.errstate(all="ignore"):
                scal = mr - mh if kind == ADDITIVE else mr / mh
            return cls(dev, eq._af, eq._hist_q, q, kind, cell_shape, grp, labels, scaling=dev.to_device(np.ascontiguousarray(scal), dtype=np.float64))
        stretches, rest, rows_of = grp.sliding_stretches(time)
        if stretches:
            scal_d = dev.empty((G, C_), np.float64)
            muh = dev.empty((G, C_), np.float64)

            def slab(a, g0, n):
                return dev.wrap(a.ptr + g0 * plane, (n, len(q), C_), np.float32)

            def rows64(a, g0, n):
                return dev.wrap(a.ptr + g0 * C_ * 8, (n, C_), np.float64)

            done = True
            for g0, rows0, en, lv in stretches:
                n = en.shape[0] + 1
                if K.eqm_train_window(dev, r, h, rows0, en, lv, q, kind, normalised=True,
                                      out=(slab(af, g0, n), slab(hq, g0, n), rows64(scal_d, g0, n), rows64(muh, g0, n))) is None:
                    done = False   # (not the kernel's shape after all: everything through the per-group path below)
                    break
            if done:
                for g in rest:     # the groups no stretch covers (day 366 of a standard calendar): from their gathered sample
                    _, _, sg = one(K.select_rows(dev, r, rows_of[g]), K.select_rows(dev, h, rows_of[g]),
                                   out=(slab(af, g, 1).reshape(len(q), C_), slab(hq, g, 1).reshape(len(q), C_)))
                    row = dev.to_device(np.ascontiguousarray(sg), dtype=np.float64)
                    dev.copy_d2d(scal_d.ptr + g * C_ * 8, row.ptr, C_ * 8)
                dev.sync()
                return cls(dev, af, hq, q, kind, cell_shape, grp, labels, scaling=scal_d)
        scal = np.empty((G, C_), np.float64)
        for g, (rg, hg) in grp.group_samples(dev, (r, h), time):
            out_g = tuple(dev.wrap(a.ptr + g * plane, (len(q), C_), np.float32) for a in (af, hq))
            _, _, scal[g] = one(rg, hg, out=out_g)
        dev.sync()
        return cls(dev, af, hq, q, kind, cell_shape, grp, labels, scaling=dev.to_device(scal, dtype=np.float64))

    @property
    def scaling(self) -> np.ndarray:
        lead = () if self.group.prop == "group" else (len(self.group_labels),)
        return self._scaling.get().reshape(lead + self.cell_shape)

    def adjust(self, sim, *, interp: str = "nearest", extrapolation: str = "constant", detrend: int = 1, time=None, keep=False,
               grouped_nearest: str = "griddata"):
        if interp not in ("nearest", "linear", "cubic"):
            raise ValueError(f"interp={interp!r} not in ('nearest', 'linear', 'cubic')")
        if detrend not in (0, 1):
            raise NotImplementedError("DetrendedQuantileMapping.adjust: detrend must be 0 or 1 (polynomial degree)")
        dev = self._dev
        s, cell_shape = _flatten(sim, dev)
        fwd, inv = ("+", "-") if self.kind == ADDITIVE else ("*", "/")
        n = len(self.cell_shape)
        if tuple(cell_shape) != self.cell_shape and self.group.prop == "group" and len(cell_shape) > n and \
                (n == 0 or tuple(cell_shape[len(cell_shape) - n:]) == self.cell_shape):
            return self._adjust_members(s, tuple(cell_shape[:len(cell_shape) - n]), interp, extrapolation, detrend, keep, fwd, inv)
        if tuple(cell_shape) != self.cell_shape:
            raise ValueError("sim does not match the trained grid")
        if self.group.prop != "group":
            return self._adjust_grouped(s, interp, extrapolation, detrend, time, keep, fwd, inv, grouped_nearest)
        scaled = K.trend_apply(dev, s, self._scaling, None, fwd)
        p0, p1 = K.poly_trend(dev, scaled, detrend)
        detr = K.trend_apply(dev, scaled, p0, p1, inv)
        del scaled
        scen0 = K.eqm_adjust(dev, detr, self._af, self._hist_q, self.kind, interp, extrapolation)
        scen = K.trend_apply(dev, scen0, p0, p1, fwd, out=detr)  # (distinct buffers: the kernels' pointers are __restrict__)
        return scen if keep else scen.get().reshape((s.shape[0],) + self.cell_shape)

    def _adjust_members(self, s, extra, interp, extrapolation, detrend, keep, fwd, inv):
        """group="time", a sim with member axes right behind time ((T, members x cells) here).  Every member is scaled and mapped
        with the same tables (repeated per member).  The trend: a model trained with ``Grouper(add_dims=...)`` detrends with
        ``PolyDetrend(group=that grouper)``, which fits ONE polynomial on the mean over the pooled members
        (``_polydetrend_get_trend``: ``da.mean(dim[1:])`` ahead of polyfit) and removes it from every member; without
        ``add_dims`` the extra axes are ordinary ones and every series has its own trend."""
        dev = self._dev
        T = s.shape[0]
        E, C_ = int(np.prod(extra)), int(np.prod(self.cell_shape)) if self.cell_shape else 1

        def tile64(a):   # (C,) float64 table -> (E * C,)
            return dev.to_device(np.tile(a.get().reshape(-1), E), dtype=np.float64)

        def tile32(tab):  # (nq, C) -> (nq, E * C)
            lead = tab.shape[0]
            return K.select_rows(dev, tab.reshape(lead, C_), np.repeat(np.arange(lead), E)).reshape(lead, E * C_)

        scaled = K.trend_apply(dev, s, tile64(self._scaling), None, fwd)
        if self.group.add_dims:
            # the mean over the members of every step: the (T, E x C) matrix is (T x E, C) with E rows per step
            mean, _ = K.resample_reduce(dev, scaled.reshape(T * E, C_), "mean", np.arange(0, T * E + 1, E, dtype=np.int64), want_valid=False)
            p0, p1 = K.poly_trend(dev, mean, detrend)
            p0, p1 = tile64(p0), (tile64(p1) if p1 is not None else None)
        else:
            p0, p1 = K.poly_trend(dev, scaled, detrend)
        detr = K.trend_apply(dev, scaled, p0, p1, inv)
        del scaled
        scen0 = K.eqm_adjust(dev, detr, tile32(self._af), tile32(self._hist_q), self.kind, interp, extrapolation)
        scen = K.trend_apply(dev, scen0, p0, p1, fwd, out=detr)
        return scen if keep else scen.get().reshape((T,) + tuple(extra) + self.cell_shape)

    def _adjust_grouped(self, s, interp, extrapolation, detrend, time, keep, fwd, inv, grouped_nearest="griddata"):
        """dqm_adjust with a sub-grouping: group-major row blocks like the grouped EQM; per block the group's
        scaling (``u.broadcast``), the trend fitted over the group's OWN steps on their time coordinate (days since the
        group's mean date: ``PolyDetrend(group=...)`` -> polyfit along time), the group's nodes, the trend put back."""
        _check_group_interp(self.group, interp, "DetrendedQuantileMapping.adjust", self.group_labels, extrapolation)
        dev = self._dev
        if time is None or len(time) != s.shape[0]:
            raise ValueError(f"group={self.group.name!r} needs time=TimeAxis of sim")
        gi = self.group.index(time, self.group_labels)
        if (gi < 0).any():
            raise ValueError("sim holds time steps whose group was not trained")
        perm = np.argsort(gi, kind="stable")
        counts = np.bincount(gi, minlength=len(self.group_labels))
        T, C_ = s.shape
        nq = len(self.quantiles)
        days = np.asarray(time.ordinal(), dtype=np.float64)
        gcoord = self.group.coordinate(time, interp=True) if interp == "linear" else None
        src = s
        prescaled = interp == "linear" and self.group.prop == "month"
        if prescaled:
            # xsdba: u.broadcast(scaling, sim, group=group, interp=interp) — for every interpolation but "nearest" (and every
            # grouping but the day of year) the scaling of a step is INTERPOLATED over the group coordinate (cyclic copies at
            # 0 and G + 1, DataArray.interp "linear"): scaling_t = S[r0] + (S[r0 + 1] - S[r0]) (g_t - r0).  Steps are handled
            # interval by interval of the coordinate (G + 1 of them): x OP (p0 + p1 u) is xh_trend_apply_u.
            G = len(self.group_labels)
            S = self._scaling.get().reshape(G, C_)
            r0 = np.clip(np.floor(gcoord).astype(np.int64), 0, G)
            row = lambda r: (r - 1) % G                                  # coordinate 0 .. G + 1 -> group (cyclic)
            perm2 = np.argsort(r0, kind="stable")
            cnt2 = np.bincount(r0, minlength=G + 1)
            s_p2 = K.select_rows(dev, s, perm2)
            sc_p2 = dev.empty((T, C_), np.float32)
            o2 = 0
            for r in range(G + 1):
                n2 = int(cnt2[r])
                if n2 == 0:
                    continue
                rows2 = perm2[o2:o2 + n2]
                a0, a1 = S[row(r)], S[row(r + 1)]
                p0 = dev.to_device(np.ascontiguousarray(a0), dtype=np.float64)
                p1 = dev.to_device(np.ascontiguousarray(a1 - a0), dtype=np.float64)
                uf = dev.to_device(np.ascontiguousarray(gcoord[rows2] - r), dtype=np.float64)
                K.trend_apply(dev, dev.wrap(s_p2.ptr + o2 * C_ * 4, (n2, C_), np.float32), p0, p1, fwd,
                              out=dev.wrap(sc_p2.ptr + o2 * C_ * 4, (n2, C_), np.float32), u=uf)
                o2 += n2
            inv2 = np.empty(T, dtype=np.int64)
            inv2[perm2] = np.arange(T)
            src = K.select_rows(dev, sc_p2, inv2)
        windowed = self.group.window > 1
        if interp == "linear" or self._plane_nearest(grouped_nearest):
            # the (quantile, group) plane serves the whole series in one launch, and so do the scaling, the per-group fit and the
            # trend (xh_trend_apply_groups / xh_poly_trend_groups, round 6: a group is a list of rows, the coefficients are
            # (G, C) tables, nothing is permuted) — bit-identical to the per-group loop below, which launched four kernels per
            # group on gathered blocks (365 groups: 87 ms for a 30-year 1440 x 90 band, now ~20)
            offs = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
            u_time = np.zeros(T, dtype=np.float64)
            for g in range(len(counts)):
                r = perm[offs[g]:offs[g + 1]]
                if len(r):
                    u_time[r] = days[r] - days[r].mean()
            u_d = dev.to_device(u_time, dtype=np.float64)
            scaled = src if prescaled else K.trend_apply_groups(dev, s, perm, offs, self._scaling, None, fwd)
            fit_on = K.window_nanmean(dev, scaled, self.group.window) if windowed else scaled
            p0, p1 = K.poly_trend_groups(dev, fit_on, perm, offs, u_d, detrend)
            del fit_on
            detr = K.trend_apply_groups(dev, scaled, perm, offs, p0, p1, inv, u=u_d)
            del scaled
            if interp == "linear":
                scen0 = K.plane_linear(dev, detr, gcoord, self._af, xq_all=self._hist_q, kind=self.kind)
            else:
                scen0 = K.plane_nearest(dev, detr, gi + 1.0, self._af, self._hist_q, self.kind, extrapolation)
            scen = K.trend_apply_groups(dev, scen0, perm, offs, p0, p1, fwd, u=u_d, out=detr)
            dev.sync()
            return scen if keep else scen.get().reshape((T,) + self.cell_shape)
        s_perm = K.select_rows(dev, src, perm)
        scen_perm = dev.empty((T, C_), np.float32)
        inv_perm = np.empty(T, dtype=np.int64)
        inv_perm[perm] = np.arange(T)
        wm_perm = None
        if windowed:
            # PolyDetrend with a windowed Grouper (xsdba.detrending._polydetrend_get_trend: ``da.mean(dim[1:])`` before polyfit):
            # the trend of a group is fitted on the centred WINDOW MEAN of the scaled series at the group's steps — neighbours
            # carry the scaling of their own groups, so the whole series is scaled first (group-major blocks), brought back to
            # time order, averaged, and permuted again
            if not prescaled:
                off = 0
                for g, n in enumerate(counts):
                    n = int(n)
                    if n:
                        sc_g = dev.wrap(self._scaling.ptr + g * C_ * 8, (C_,), np.float64)
                        blk = dev.wrap(s_perm.ptr + off * C_ * 4, (n, C_), np.float32)
                        K.trend_apply(dev, blk, sc_g, None, fwd, out=dev.wrap(scen_perm.ptr + off * C_ * 4, (n, C_), np.float32))
                    off += n
                s_perm, scen_perm = scen_perm, s_perm          # s_perm: the scaled series, group-major
                src = K.select_rows(dev, s_perm, inv_perm)     # ... and in time order
            wm_perm = K.select_rows(dev, K.window_nanmean(dev, src, self.group.window), perm)
        off = 0
        # (what is left here: every step through the nodes of its OWN group — grouped_nearest="group", seasons, more than 32
        # nodes, "cubic": one xh_eqm_adjust per group on its gathered block)
        for g, n in enumerate(counts):
            n = int(n)
            if n == 0:
                continue
            rows = perm[off:off + n]
            u = dev.to_device(np.ascontiguousarray(days[rows] - days[rows].mean()), dtype=np.float64)
            blk = dev.wrap(s_perm.ptr + off * C_ * 4, (n, C_), np.float32)
            out = dev.wrap(scen_perm.ptr + off * C_ * 4, (n, C_), np.float32)
            sc_g = dev.wrap(self._scaling.ptr + g * C_ * 8, (C_,), np.float64)
            af_g = dev.wrap(self._af.ptr + g * nq * C_ * 4, (nq, C_), np.float32)
            hq_g = dev.wrap(self._hist_q.ptr + g * nq * C_ * 4, (nq, C_), np.float32)
            scaled = blk if (prescaled or windowed) else K.trend_apply(dev, blk, sc_g, None, fwd)
            fit_on = scaled if not windowed else dev.wrap(wm_perm.ptr + off * C_ * 4, (n, C_), np.float32)
            p0, p1 = K.poly_trend(dev, fit_on, detrend, u=u)
            detr = K.trend_apply(dev, scaled, p0, p1, inv, u=u)
            scen0 = K.eqm_adjust(dev, detr, af_g, hq_g, self.kind, interp, extrapolation)
            K.trend_apply(dev, scen0, p0, p1, fwd, out=out, u=u)
            off += n
        scen = K.select_rows(dev, scen_perm, inv_perm)
        dev.sync()
        return scen if keep else scen.get().reshape((T,) + self.cell_shape)
