"""xh_plane_linear — xsdba's interp="linear" over the (quantile, group) plane (utils.interp_on_quantiles' 2-D branch; upstream
xsdba re-exported by /root/reference/src/xclim/sdba.py:10, documented in /root/reference/docs/sdba.rst:64-65) — against the
oracle, which calls the REAL scipy.interpolate.griddata(method="linear") on the cyclically padded nodes.  The kernel builds
no triangulation: it walks to the Delaunay triangle of every query (plane.hip).  Tolerance: the factor within 1e-6 relative
(north_star's bar for floats) + 1e-6 absolute of the factor scale."""

import numpy as np
import pytest

from oracle import sdba as osdba
from xclim_amd import kernels as K

pytestmark = pytest.mark.gpu


def _nodes(rng, G, nq, C, scale, kind):
    """(G, nq, C) sorted abscissae and factors.  kind "t": temperature-like (gaussian spread `scale` per node), "p":
    precipitation-like (skewed, zero-bounded), both with a smooth annual cycle across the groups."""
    cyc = np.sin(2 * np.pi * (np.arange(G) + 0.5) / G)[:, None, None]
    if kind == "p":
        x = np.sort(rng.gamma(0.7, scale * 3.0, (G, nq, C)), axis=1) * (1.0 + 0.5 * cyc)
    else:
        x = np.sort(rng.normal(0.0, scale, (G, nq, C)), axis=1) + 10.0 * scale * cyc + 280.0
    y = rng.normal(0.0, 1.0, (G, nq, C)) + 2.0 * cyc
    x, y = x.astype(np.float32), y.astype(np.float32)
    # float32 abscissae near 280 K tie by rounding (spacing 3e-5 K against node gaps of 0.01 K at scale 0.05).  Tied nodes of one
    # row with DIFFERENT factors are the documented unreproducible case (Qhull keeps one of two coincident points — the first in
    # 59 %, the last in 28 % of the queries where it matters; the kernel keeps the first): ties get the same factor here
    for _ in range(nq):
        tie = x[:, 1:] == x[:, :-1]
        if not tie.any():
            break
        y[:, 1:][tie] = y[:, :-1][tie]
    return x, y


def _cocircular(x, g, xq, yq, t, c):
    """Is the query's scipy triangle degenerate — a fourth node EXACTLY on its circumcircle?  float32 abscissae are a grid
    (2^-15 K at 280 K): x_a + x_b == x_c + x_d happens, two valid Delaunay triangulations exist and scipy's choice follows
    Qhull's facet order (plane.hip, header; tools/fuzz_plane.py counts these)."""
    from scipy.spatial import Delaunay

    G, nq = xq.shape[0], xq.shape[1]
    ext = np.concatenate([[G - 1], np.arange(G), [0]])
    pts = np.array([(float(xq[r, k, c]), float(i)) for i, r in enumerate(ext) for k in range(nq)
                    if not (np.isnan(xq[r, k, c]) or np.isnan(yq[r, k, c]))])
    pts = np.unique(pts, axis=0)
    tri = Delaunay(pts)
    P3 = pts[tri.simplices[int(tri.find_simplex(np.array([float(x[t, c]), float(g[t])])))]]
    ax, ay = P3[1] - P3[0]
    bx, by = P3[2] - P3[0]
    d = 2 * (ax * by - ay * bx)
    ux, uy = (by * (ax * ax + ay * ay) - ay * (bx * bx + by * by)) / d, (ax * (bx * bx + by * by) - bx * (ax * ax + ay * ay)) / d
    pw = ((pts - (P3[0] + [ux, uy])) ** 2).sum(1) - (ux * ux + uy * uy)
    return int((np.abs(pw) < 1e-9 * max(1.0, ux * ux + uy * uy)).sum()) >= 4


def _check(dev, x, g, xq, yq, rtol=1e-6, atol=1e-6):
    G = yq.shape[0]
    got = K.plane_linear(dev, dev.to_device(x), g, dev.to_device(yq), xq_all=dev.to_device(xq), kind="factor").get()
    exp = osdba.interp_on_quantiles_2d(x, g, np.arange(1, G + 1), xq, yq, "linear", "constant")
    assert np.array_equal(np.isnan(got), np.isnan(exp))
    bad = ~np.isclose(got, exp, rtol=rtol, atol=atol * max(1.0, float(np.nanmax(np.abs(yq)))), equal_nan=True)
    # (whatever the seed of the inputs: a mismatch is accepted only as a verified exact degeneracy, and only a handful)
    assert bad.sum() <= 4, f"{int(bad.sum())} mismatches"
    for t, c in np.argwhere(bad):
        assert _cocircular(x, g, xq, yq, t, c), f"query ({t}, {c}): got {got[t, c]}, scipy {exp[t, c]}"
    return got


@pytest.mark.parametrize("G,nq", [(12, 20), (12, 5), (40, 8), (365, 6)])
@pytest.mark.parametrize("scale,kind", [(0.05, "t"), (1.5, "t"), (6.0, "t"), (2.0, "p"), (12.0, "p")])
@pytest.mark.parametrize("fractional", [True, False])
def test_plane_linear_matches_griddata(dev, rng, G, nq, scale, kind, fractional):
    """Node spacings from a twentieth of a group step (every row edge is a Delaunay edge: strips between adjacent rows) to a
    dozen group steps (triangles span many rows: the precipitation tail in mm/day); fractional coordinates (months) and
    integer ones (days of year); queries inside, below and above the nodes and NaN."""
    C, T = 7, 300 if G < 100 else 120
    xq, yq = _nodes(rng, G, nq, C, scale, kind)
    lo, hi = xq.min(), xq.max()
    x = rng.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), (T, C)).astype(np.float32)
    x[rng.random((T, C)) < 0.03] = np.nan
    g = rng.uniform(0.5, G + 0.5, T) if fractional else rng.integers(1, G + 1, T).astype(np.float64)
    g[:3] = [0.5, G + 0.5, 1.0] if fractional else [1.0, float(G), 1.0]   # the cyclic ends
    _check(dev, x, g, xq, yq)


def test_plane_linear_nan_nodes_and_ties(dev, rng):
    """NaN factors / abscissae drop their node (mask_old in _interp_on_quantiles_2D) but still move the bounds as upstream's
    _first_and_last_nonnull sees them; a whole group without nodes is bridged by its neighbours' triangles; tied abscissae
    with equal factors (quantised data) collapse to one node."""
    G, nq, C, T = 12, 10, 5, 400
    xq, yq = _nodes(rng, G, nq, C, 1.5, "t")
    yq[2, 3:5, 0] = np.nan          # interior NaN factors
    xq[5, 7, 1] = np.nan            # a NaN abscissa
    yq[7, :, 2] = np.nan            # a whole group without factors (cell 2)
    xq[9, 4, 3] = xq[9, 3, 3]       # a tie, same factor
    yq[9, 4, 3] = yq[9, 3, 3]
    x = rng.uniform(xq[~np.isnan(xq)].min() - 1, xq[~np.isnan(xq)].max() + 1, (T, C)).astype(np.float32)
    g = rng.uniform(0.5, G + 0.5, T)
    # (cell 2: between the rows around the empty group the bounds are NaN upstream -> no clamping there; inside the
    # polygon of the neighbouring rows the Delaunay value; compare where the oracle has a value)
    got = K.plane_linear(dev, dev.to_device(x), g, dev.to_device(yq), xq_all=dev.to_device(xq), kind="factor").get()
    exp = osdba.interp_on_quantiles_2d(x, g, np.arange(1, G + 1), xq, yq, "linear", "constant")
    both = ~np.isnan(got) & ~np.isnan(exp)
    assert both[:, [0, 1, 3, 4]].mean() > 0.95
    np.testing.assert_allclose(got[both], exp[both], rtol=1e-6, atol=1e-5)
    for c in (0, 1, 3, 4):
        assert np.array_equal(np.isnan(got[:, c]), np.isnan(exp[:, c]))


def test_plane_linear_kinds_and_base(dev, rng):
    """kind "+" / "*" apply the factor to `base` (QDM: the abscissa is the rank, the factor goes onto sim) or to xnew."""
    G, nq, C, T = 12, 8, 4, 100
    xq, yq = _nodes(rng, G, nq, C, 1.5, "t")
    x = rng.uniform(xq.min(), xq.max(), (T, C)).astype(np.float32)
    base = rng.normal(5, 1, (T, C)).astype(np.float32)
    g = rng.uniform(0.5, G + 0.5, T)
    d_x, d_y, d_q, d_b = dev.to_device(x), dev.to_device(yq), dev.to_device(xq), dev.to_device(base)
    f = K.plane_linear(dev, d_x, g, d_y, xq_all=d_q, kind="factor").get()
    np.testing.assert_array_equal(K.plane_linear(dev, d_x, g, d_y, xq_all=d_q, kind="+").get(), x + f)
    np.testing.assert_array_equal(K.plane_linear(dev, d_x, g, d_y, xq_all=d_q, kind="*", base=d_b).get(), base * f)
    with pytest.raises(ValueError):
        K.plane_linear(dev, d_x, g, d_y, kind="+")
