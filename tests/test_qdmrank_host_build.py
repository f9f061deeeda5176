"""qdmrank.h — the class boundaries of QuantileDeltaMapping "nearest" as doubled average RANKS (qdm_min_r2, shared by the one-year
register kernel of qdm2.hip and the streaming kernels of select4.hip) — compiled for the host with g++ and checked against the
definition evaluated by brute force in numpy: R = min { r2 in [1, 2n] : test(pct(r2)) } with
pct = mx (r2 / 2n - mn) / (mx - mn) in the operation order of xsdba's rank(pct=True) (upstream xsdba, re-exported by
/root/reference/src/xclim/sdba.py:10; oracle/sdba.py rank_pct).  No GPU."""
import ctypes
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "xclim_amd", "csrc")


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    d = tmp_path_factory.mktemp("qdmrank_host")
    common = open(os.path.join(CSRC, "common.h")).read()
    m = re.search(r"__device__ __forceinline__ double xh_div_int\(.*?\n}\n", common, re.S)
    body = open(os.path.join(CSRC, "qdmrank.h")).read().replace('#include "common.h"', "").replace("#pragma once", "")
    (d / "shim.cpp").write_text("#include <math.h>\n#include <stdint.h>\n#define __device__\n#define __forceinline__ inline\n" + m.group(0) + body +
                                '\nextern "C" uint32_t min_r2(int first, double thr, uint32_t nn, uint32_t c0, uint32_t cmax) '
                                "{ return qdm_min_r2(first != 0, thr, nn, c0, cmax); }\n"
                                'extern "C" uint32_t pos_of_r2(uint32_t R) { return qdm_pos_of_r2(R); }\n')
    # (-ffp-contract=off like the device build: the fma calls are explicit)
    subprocess.run(["g++", "-O1", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(d / "libqdmrank_host.so"), str(d / "shim.cpp")], check=False, cwd=d).returncode == 0 or pytest.skip("g++ did not build the host stand-in here")
    lib = ctypes.CDLL(str(d / "libqdmrank_host.so"))
    lib.min_r2.restype = lib.pos_of_r2.restype = ctypes.c_uint32
    lib.min_r2.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32]
    lib.pos_of_r2.argtypes = [ctypes.c_uint32]
    return lib


def _brute(first, thr, n, c0, cmax):
    r2 = np.arange(1, 2 * n + 1, dtype=np.float64)
    mn = ((c0 + 1) / 2.0) / n
    mx = ((2 * n - cmax + 1) / 2.0) / n
    with np.errstate(all="ignore"):
        pct = mx * ((r2 * 0.5) / n - mn) / (mx - mn)
        ok = ~(pct < thr) if first else pct > thr
    hit = np.flatnonzero(ok)
    return int(hit[0]) + 1 if len(hit) else 2 * n + 1


def test_min_r2_is_the_first_rank_that_passes(lib):
    rng = np.random.default_rng(3)
    checked = 0
    for n in [2, 3, 7, 30, 365, 366, 930, 10950]:
        for _ in range(60):
            c0 = int(rng.integers(1, max(2, n // 2)))
            cmax = int(rng.integers(1, max(2, min(4, n - c0))))
            if c0 + cmax > n:
                continue
            for first in (True, False):
                # thresholds: random, the nodes of equally spaced quantiles and their midpoints, and values that ARE a pct exactly
                r2 = rng.integers(1, 2 * n + 1)
                mn, mx = ((c0 + 1) / 2.0) / n, ((2 * n - cmax + 1) / 2.0) / n
                exact = mx * ((r2 * 0.5) / n - mn) / (mx - mn)
                for thr in (float(rng.random()), (int(rng.integers(0, 20)) + 0.5) / 20, int(rng.integers(1, 20)) / 20, float(exact),
                            float(np.nextafter(exact, 2)), float(np.nextafter(exact, -2)), 0.0, 1.0):
                    got = lib.min_r2(int(first), thr, n, c0, cmax)
                    assert got == _brute(first, thr, n, c0, cmax), (n, c0, cmax, first, thr)
                    checked += 1
    assert checked > 5000
    # position whose run decides boundary R: ceil((R - 2) / 2), 0 for R <= 2
    for R in range(1, 50):
        assert lib.pos_of_r2(R) == max(0, -(-(R - 2) // 2))
