"""The Delaunay point location of xh_plane_linear (xclim_amd/csrc/plane.hip), restated in numpy (tools/experiments/r05/proto_plane.py:
the same starting triangle from two rows, the same dual-simplex walk with the deepest site inside the circumcircle entering
and the ratio test choosing the leaving vertex), against scipy.interpolate.griddata(method="linear") — what xsdba's
interp_on_quantiles calls for a month / day-of-year Grouper (upstream xsdba, re-exported by /root/reference/src/xclim/sdba.py:10).
No GPU: this pins the ALGORITHM; tests/test_gpu_plane.py pins the kernel."""
import importlib.util
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("proto_plane", os.path.join(HERE, "..", "tools", "experiments", "r05", "proto_plane.py"))
proto = importlib.util.module_from_spec(spec)
spec.loader.exec_module(proto)


@pytest.mark.parametrize("G,nq,scale,skewed,on_rows", [(12, 13, 0.05, False, False), (12, 8, 5.0, False, False), (40, 6, 30.0, False, False),
                                                       (12, 10, 2.0, True, False), (40, 7, 1.0, False, True), (12, 5, 8.0, True, True)])
def test_walk_finds_the_delaunay_triangle(G, nq, scale, skewed, on_rows):
    """Node spacings from a twentieth of a group step (every row edge is a Delaunay edge) to thirty (triangles span a dozen
    rows), gaussian and skewed node sets, fractional and integer group coordinates."""
    from scipy.interpolate import griddata

    rng = np.random.default_rng(7)
    xs = np.sort(rng.gamma(0.7, scale * 3, (G + 2, nq)), axis=1) if skewed else np.sort(
        rng.normal(0, scale, (G + 2, nq)) + rng.normal(0, scale, (G + 2, 1)) * 0.3, axis=1)
    vs = rng.normal(0, 1, (G + 2, nq))
    gg = np.repeat(np.arange(G + 2.0)[:, None], nq, 1)
    n = 300
    yq = rng.integers(1, G + 1, n).astype(float) if on_rows else rng.uniform(0.5, G + 0.5, n)
    xq = rng.uniform(xs.min(), xs.max(), n)
    ref = griddata((xs.ravel(), gg.ravel()), vs.ravel(), (xq, yq), method="linear")
    rows_x, rows_v = [xs[r] for r in range(G + 2)], [vs[r] for r in range(G + 2)]
    pivots = []
    got = np.array([np.nan if (v := proto.locate(rows_x, rows_v, a, b, pivots)) is None else v for a, b in zip(xq, yq)])
    both = ~np.isnan(ref) & ~np.isnan(got)
    assert both.sum() > n // 4
    assert not (np.isnan(ref) & ~np.isnan(got)).any()        # outside the hull -> outside the strip polygon of the two rows
    np.testing.assert_allclose(got[both], ref[both], rtol=0, atol=1e-12)
    assert max(pivots) <= 40
