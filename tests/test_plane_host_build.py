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
    subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-o", str(d / "libplane_host.so"), str(d / "shim.cpp")], check=False, cwd=d).returncode == 0 or pytest.skip("g++ did not build the host stand-in here")
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


# ---- the generic kernels themselves (k_plane_pack, k_plane_linear: no LDS, no wave intrinsics), run thread by thread on the host ----
KSHIM = r"""
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#define __device__
#define __global__
#define __forceinline__ inline
#define __restrict__
#define __launch_bounds__(x)
#define XH_BLOCK 256
struct Dim3 { unsigned x, y, z; };
static Dim3 blockIdx, threadIdx, gridDim;
static inline double xh_nan64() { return NAN; }
static inline float xh_nan32() { return NAN; }
static inline double __longlong_as_double(long long v) { double d; __builtin_memcpy(&d, &v, 8); return d; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }
#include "kbody.inc"
// xh_plane_linear's generic route: pack, then one "thread" per cell and row chunk
extern "C" void plane_linear_host(const float* xnew, int64_t T, int64_t C, const double* gnew, const float* xq_all, const float* yq_all, int G,
                                  int nq, int kind, float* scen, float* px, float* py, uint8_t* cnt, float* fx, float* lx, float* fy, float* ly) {
  gridDim = {(unsigned)cdiv64(C, XH_BLOCK), (unsigned)G, 1};
  for (blockIdx.y = 0; blockIdx.y < gridDim.y; ++blockIdx.y)
    for (blockIdx.x = 0; blockIdx.x < gridDim.x; ++blockIdx.x)
      for (threadIdx.x = 0; threadIdx.x < XH_BLOCK; ++threadIdx.x)
        k_plane_pack(xq_all, nullptr, yq_all, G, nq, C, px, py, cnt, fx, lx, fy, ly);
  PlaneTabs tabs{px, py, cnt, fx, lx, fy, ly, G, nq, C};
  gridDim = {(unsigned)cdiv64(C, XH_BLOCK), 3, 1};
  for (blockIdx.y = 0; blockIdx.y < gridDim.y; ++blockIdx.y)
    for (blockIdx.x = 0; blockIdx.x < gridDim.x; ++blockIdx.x)
      for (threadIdx.x = 0; threadIdx.x < XH_BLOCK; ++threadIdx.x)
        k_plane_linear(xnew, nullptr, T, C, gnew, tabs, kind, scen, C);
}
"""


@pytest.fixture(scope="module")
def klib(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    d = tmp_path_factory.mktemp("plane_kernels_host")
    src = open(SRC).read()

    def grab(start, end):
        a = src.index(start)
        return src[a:src.index(end, a) + len(end)]

    parts = [grab("struct PlaneTabs {", "};\n"), grab("__global__ void __launch_bounds__(XH_BLOCK)\nk_plane_pack(", "\n}\n"),
             src[src.index("// [host-testable: begin]"):src.index("// [host-testable: end]")],
             grab("__global__ void __launch_bounds__(XH_BLOCK)\nk_plane_linear(", "\n}\n")]
    body = "\n".join(line for line in "\n".join(parts).splitlines() if not line.lstrip().startswith("#pragma unroll"))
    (d / "kbody.inc").write_text(body)
    (d / "kshim.cpp").write_text(KSHIM)
    subprocess.run(["g++", "-O1", "-shared", "-fPIC", "-o", str(d / "libplane_kernels_host.so"), str(d / "kshim.cpp")], check=False, cwd=d).returncode == 0 or pytest.skip("g++ did not build the host stand-in here")
    lib = ctypes.CDLL(str(d / "libplane_kernels_host.so"))
    vp = ctypes.c_void_p
    lib.plane_linear_host.argtypes = [vp, ctypes.c_int64, ctypes.c_int64, vp, vp, vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, vp] + [vp] * 7
    return lib


@pytest.mark.parametrize("G,nq,scale,kind,fractional", [(12, 10, 1.5, "t", True), (12, 6, 8.0, "p", True), (30, 7, 0.4, "t", False), (4, 12, 3.0, "p", False)])
def test_plane_linear_kernels_on_the_host_match_the_oracle(klib, G, nq, scale, kind, fractional):
    """xh_plane_linear end to end without a GPU: k_plane_pack + k_plane_linear executed thread by thread, against the oracle's
    interp_on_quantiles_2d (scipy.griddata + upstream's bounds / constant extrapolation), incl. NaN queries, queries outside the
    nodes, NaN factors and tied abscissae with equal factors."""
    from oracle import sdba as osdba

    rng = np.random.default_rng(21)
    C, T = 5, 160
    cyc = np.sin(2 * np.pi * (np.arange(G) + 0.5) / G)[:, None, None]
    if kind == "p":
        xq = np.sort(rng.gamma(0.7, scale * 3.0, (G, nq, C)), axis=1) * (1.0 + 0.5 * cyc)
    else:
        xq = np.sort(rng.normal(0.0, scale, (G, nq, C)), axis=1) + 3.0 * scale * cyc
    yq = (rng.normal(0.0, 1.0, (G, nq, C)) + 2.0 * cyc).astype(np.float32)
    xq = xq.astype(np.float32)
    yq[2, 1, 0] = np.nan
    xq[1, 3, 1], yq[1, 3, 1] = xq[1, 2, 1], yq[1, 2, 1]
    lo, hi = float(xq.min()), float(xq.max())
    x = rng.uniform(lo - 0.1 * (hi - lo), hi + 0.1 * (hi - lo), (T, C)).astype(np.float32)
    x[rng.random((T, C)) < 0.03] = np.nan
    g = rng.uniform(0.5, G + 0.5, T) if fractional else rng.integers(1, G + 1, T).astype(np.float64)
    g[:2] = [0.5, G + 0.5] if fractional else [1.0, float(G)]
    got = np.empty((T, C), np.float32)
    scratch = [np.zeros((G, nq, C), np.float32), np.zeros((G, nq, C), np.float32), np.zeros((G, C), np.uint8)] + [np.zeros((G, C), np.float32) for _ in range(4)]
    klib.plane_linear_host(x.ctypes.data, T, C, g.ctypes.data, xq.ctypes.data, yq.ctypes.data, G, nq, 2, got.ctypes.data,
                           *[a.ctypes.data for a in scratch])
    exp = osdba.interp_on_quantiles_2d(x, g, np.arange(1, G + 1), xq, yq, "linear", "constant")
    cells = [c for c in range(C) if c != 0]   # (cell 0 has a NaN factor: the bounds next to it differ by design, see the GPU test)
    assert np.array_equal(np.isnan(got[:, cells]), np.isnan(exp[:, cells]))
    np.testing.assert_allclose(got[:, cells], exp[:, cells], rtol=1e-6, atol=1e-6, equal_nan=True)
    both = ~np.isnan(got[:, 0]) & ~np.isnan(exp[:, 0])
    assert both.mean() > 0.8
    np.testing.assert_allclose(got[both, 0], exp[both, 0], rtol=1e-6, atol=1e-5)
