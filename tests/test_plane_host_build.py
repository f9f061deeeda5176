"""The DEVICE code of the plane kernels — plane_locate (the Delaunay walk of xh_plane_linear) and plane_nearest (xh_plane_nearest),
xclim_amd/csrc/plane.hip between its "host-testable" markers — compiled for the host with g++ and checked against
scipy.interpolate.griddata, which is what xsdba's interp_on_quantiles calls for a month / day-of-year Grouper (upstream xsdba,
re-exported by /root/reference/src/xclim/sdba.py:10).  No GPU: tests/test_plane_walk_cpu.py pins the ALGORITHM (a numpy
restatement), this test pins the C++ the GPU runs; tests/test_gpu_plane.py and tools/fuzz_plane.py pin the kernels."""
import ctypes
import os
import shutil
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "xclim_amd", "csrc", "plane.hip")

SHIM = r"""
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#define __device__
#define __forceinline__ inline
#define __restrict__
static inline double xh_nan64() { return NAN; }
static inline float xh_nan32() { return NAN; }
static inline double __longlong_as_double(long long v) { double d; __builtin_memcpy(&d, &v, 8); return d; }
struct PlaneTabs {
  const float* px; const float* py; const uint8_t* cnt; const float* fx; const float* lx; const float* fy; const float* ly;
  int G, nq; int64_t C;
};
#include "body.inc"
extern "C" double locate(const float* px, const float* py, const uint8_t* cnt, int G, int nq, int64_t C, int64_t c, double qx, double qy) {
  PlaneTabs t{px, py, cnt, nullptr, nullptr, nullptr, nullptr, G, nq, C};
  PlaneCell P{t, c};
  return plane_locate(P, qx, qy);
}
extern "C" double nearest(const float* px, const float* py, const uint8_t* cnt, int G, int nq, int64_t C, int64_t c, double x, int r) {
  PlaneTabs t{px, py, cnt, nullptr, nullptr, nullptr, nullptr, G, nq, C};
  PlaneCell P{t, c};
  return plane_nearest(P, x, r);
}
"""


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    d = tmp_path_factory.mktemp("plane_host")
    src = open(SRC).read()
    a, b = src.index("// [host-testable: begin]"), src.index("// [host-testable: end]")
    body = "\n".join(line for line in src[a:b].splitlines() if not line.lstrip().startswith("#pragma unroll"))
    (d / "body.inc").write_text(body)
    (d / "shim.cpp").write_text(SHIM)
    subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-o", str(d / "libplane_host.so"), str(d / "shim.cpp")], check=True, cwd=d)
    lib = ctypes.CDLL(str(d / "libplane_host.so"))
    common = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_double]
    lib.locate.restype = lib.nearest.restype = ctypes.c_double
    lib.locate.argtypes = common + [ctypes.c_double]
    lib.nearest.argtypes = common + [ctypes.c_int]
    return lib


def _pack(xq, yq):
    """k_plane_pack on the host: NaN nodes dropped, strictly increasing abscissa (the first of tied nodes stays)."""
    G, nq, C = xq.shape
    px, py, cnt = np.zeros((G, nq, C), np.float32), np.zeros((G, nq, C), np.float32), np.zeros((G, C), np.uint8)
    for g in range(G):
        for c in range(C):
            m, last = 0, 0.0
            for k in range(nq):
                x, y = xq[g, k, c], yq[g, k, c]
                if x == x and y == y and (m == 0 or x > last):
                    px[g, m, c], py[g, m, c], last, m = x, y, x, m + 1
            cnt[g, c] = m
    return px, py, cnt


@pytest.mark.parametrize("G,nq,scale,skewed,fractional", [(12, 13, 0.05, False, True), (12, 8, 5.0, False, True), (40, 6, 30.0, False, False),
                                                          (12, 10, 2.0, True, True), (3, 7, 1.0, False, True), (12, 5, 8.0, True, False)])
def test_plane_locate_as_compiled_matches_griddata(lib, G, nq, scale, skewed, fractional):
    from scipy.interpolate import griddata

    rng = np.random.default_rng(11)
    C, n = 3, 250
    xq = (np.sort(rng.gamma(0.7, scale * 3, (G, nq, C)), axis=1) if skewed else
          np.sort(rng.normal(0, scale, (G, nq, C)) + rng.normal(0, scale, (G, 1, C)) * 0.3, axis=1)).astype(np.float32)
    yq = rng.normal(0, 1, (G, nq, C)).astype(np.float32)
    yq[1, 2, 0] = np.nan   # a dropped node
    px, py, cnt = _pack(xq, yq)
    ext = np.concatenate([[G - 1], np.arange(G), [0]])   # the cyclic copies at rows 0 and G + 1
    checked = 0
    for c in range(C):
        ok = ~np.isnan(yq[ext, :, c])
        gg = np.repeat(np.arange(G + 2.0)[:, None], nq, 1)
        pts = (xq[ext, :, c].astype(np.float64)[ok], gg[ok])
        qy = rng.uniform(0.5, G + 0.5, n) if fractional else rng.integers(1, G + 1, n).astype(float)
        qx = rng.uniform(xq[:, :, c].min(), xq[:, :, c].max(), n)
        ref = griddata(pts, yq[ext, :, c].astype(np.float64)[ok], (qx, qy), method="linear")
        got = np.array([lib.locate(px.ctypes.data, py.ctypes.data, cnt.ctypes.data, G, nq, C, c, a, b) for a, b in zip(qx, qy)])
        both = ~np.isnan(ref) & ~np.isnan(got)
        assert both.sum() > n // 5
        # (outside the strip polygon of its two rows the device function answers NaN — the kernel's bounds test decides those —
        #  while the convex hull of ALL nodes may still cover the query: only the converse must never happen)
        assert not (np.isnan(ref) & ~np.isnan(got)).any()
        np.testing.assert_allclose(got[both], ref[both], rtol=0, atol=1e-9)
        checked += int(both.sum())
    assert checked > n


def test_plane_nearest_as_compiled_matches_griddata(lib):
    from scipy.interpolate import griddata

    rng = np.random.default_rng(12)
    G, nq, C, n = 12, 9, 2, 300
    xq = np.sort(rng.gamma(0.7, 12.0, (G, nq, C)), axis=1).astype(np.float32)   # gaps of many group steps: neighbours win
    yq = rng.normal(0, 1, (G, nq, C)).astype(np.float32)
    px, py, cnt = _pack(xq, yq)
    ext = np.concatenate([[G - 1], np.arange(G), [0]])
    for c in range(C):
        gg = np.repeat(np.arange(G + 2.0)[:, None], nq, 1)
        qr = rng.integers(1, G + 1, n)
        qx = rng.uniform(0, xq[:, :, c].max(), n)
        ref = griddata((xq[ext, :, c].astype(np.float64).ravel(), gg.ravel()), yq[ext, :, c].astype(np.float64).ravel(), (qx, qr.astype(float)),
                       method="nearest")
        got = np.array([lib.nearest(px.ctypes.data, py.ctypes.data, cnt.ctypes.data, G, nq, C, c, a, int(r)) for a, r in zip(qx, qr)])
        assert (got != ref).sum() <= 2          # (an exact tie between two rows may go either way in cKDTree)
        assert (got != ref).mean() < 0.01
