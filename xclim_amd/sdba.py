"""Host mirror of the sdba empirical-quantile-mapping path (``xclim.sdba`` == third-party ``xsdba >= 0.4.0``;
reference shim src/xclim/sdba.py:1-28; object API pinned by tests/test_xsdba.py:44-155).

Names follow xsdba: ``EmpiricalQuantileMapping.train(ref, hist, nquantiles=20, kind="+", group="time")`` then
``.adjust(sim, interp="nearest", extrapolation="constant")``, with ``ds.af`` / ``ds.hist_q`` exposed as ``.af`` /
``.hist_q``; ``nbutils.quantile`` and ``utils.equally_spaced_nodes`` as module functions.  Only ``group="time"``
(no sub-grouping) is supported.  Arrays: TIME ON AXIS 0, numpy or device arrays; all arithmetic is in
``xh_eqm_train`` / ``xh_eqm_adjust``.
"""

from __future__ import annotations

import numpy as np

from . import kernels as K
from ._capi import DeviceArray, get_device
from .calendar import _flatten

ADDITIVE, MULTIPLICATIVE = "+", "*"


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


class EmpiricalQuantileMapping:
    """Empirical quantile mapping bias adjustment (train on ref/hist quantiles, adjust sim by node search)."""

    def __init__(self, dev, af: DeviceArray, hist_q: DeviceArray, quantiles, kind, cell_shape):
        self._dev = dev
        self._af, self._hist_q = af, hist_q
        self.quantiles = np.asarray(quantiles)
        self.kind = kind
        self.cell_shape = tuple(cell_shape)
        self.adj_params = {"group": "time", "kind": kind, "nquantiles": len(self.quantiles)}

    @classmethod
    def train(cls, ref, hist, *, nquantiles=20, kind: str = ADDITIVE, group: str = "time", device=None):
        if group != "time":
            raise NotImplementedError("only group='time' is supported by the HIP backend")
        if kind not in (ADDITIVE, MULTIPLICATIVE):
            raise ValueError(f"kind must be '+' or '*', got {kind!r}")
        dev = device or get_device()
        r, cell_shape = _flatten(ref, dev)
        h, cell_shape_h = _flatten(hist, dev)
        if tuple(cell_shape) != tuple(cell_shape_h) or r.shape != h.shape:
            raise ValueError("ref and hist must have the same shape")  # _check_matching_time_sizes analogue
        q = equally_spaced_nodes(nquantiles) if np.isscalar(nquantiles) else np.asarray(nquantiles, dtype=np.float64)
        af, hq = K.eqm_train(dev, r, h, q, kind)
        return cls(dev, af, hq, q, kind, cell_shape)

    def adjust(self, sim, *, interp: str = "nearest", extrapolation: str = "constant", keep=False):
        if interp not in ("nearest", "linear", "cubic"):
            raise ValueError(f"interp={interp!r} not in ('nearest', 'linear', 'cubic')")
        s, cell_shape = _flatten(sim, self._dev)
        if tuple(cell_shape) != self.cell_shape:
            raise ValueError("sim does not match the trained grid")
        scen = K.eqm_adjust(self._dev, s, self._af, self._hist_q, self.kind, interp, extrapolation)
        return scen if keep else scen.get().reshape((s.shape[0],) + self.cell_shape)

    @property
    def af(self) -> np.ndarray:
        return self._af.get().reshape((len(self.quantiles),) + self.cell_shape)

    @property
    def hist_q(self) -> np.ndarray:
        return self._hist_q.get().reshape((len(self.quantiles),) + self.cell_shape)
