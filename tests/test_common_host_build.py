"""The scalar helpers every kernel leans on (xclim_amd/csrc/common.h) compiled for the host with g++ and checked against numpy —
no GPU:
  xh_f2key / xh_key2f   the order-preserving float <-> uint32 key of the selection kernels (NaN sorts last, like numpy.sort)
  xh_cmp_f32            the run-time operator of `compare` (indices/generic.py:83-126: numpy semantics, NaN compares False but !=)
  xh_one_cmp            its one-compare form  x OP t  <=>  sgn * x > t'  (threshold_count's inner loop)
  xh_div_int            the 3-FMA quotient used where a division runs once per cell-timestep: must equal IEEE division."""
import ctypes
import os
import re
import shutil
import subprocess

import numpy as np
import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _grab(src, start, end="\n}\n"):
    a = src.index(start)
    return src[a:src.index(end, a) + len(end)]


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    d = tmp_path_factory.mktemp("common_host")
    common = open(os.path.join(ROOT, "xclim_amd", "csrc", "common.h")).read()
    header = open(os.path.join(ROOT, "include", "xclim_hip.h")).read()
    ops = "\n".join(re.findall(r"#define XH_OP_\w+ \d+", header))
    parts = [_grab(common, "__device__ __forceinline__ uint32_t xh_f2key"), _grab(common, "__device__ __forceinline__ float xh_key2f"),
             _grab(common, "__device__ __forceinline__ bool xh_cmp_f32"), _grab(common, "struct XhOneCmp", "};\n"),
             _grab(common, "static inline XhOneCmp xh_one_cmp"), _grab(common, "__device__ __forceinline__ double xh_div_int")]
    shim = ("#include <math.h>\n#include <stdint.h>\n#include <string.h>\n#define __device__\n#define __forceinline__ inline\n" + ops + "\n"
            "static inline uint32_t __float_as_uint(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }\n"
            "static inline float __uint_as_float(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }\n" + "\n".join(parts) +
            '\nextern "C" void f2key(const float* x, uint32_t* k, int n) { for (int i = 0; i < n; ++i) k[i] = xh_f2key(x[i]); }\n'
            'extern "C" void key2f(const uint32_t* k, float* x, int n) { for (int i = 0; i < n; ++i) x[i] = xh_key2f(k[i]); }\n'
            'extern "C" void cmp(const float* a, int op, float t, uint8_t* full, uint8_t* one, int* ok, int n) {\n'
            "  const XhOneCmp c = xh_one_cmp(op, t); *ok = c.ok;\n"
            "  for (int i = 0; i < n; ++i) { full[i] = xh_cmp_f32(a[i], op, t); one[i] = c.sgn * a[i] > c.thr; } }\n"
            'extern "C" void divint(const double* s, double n, double* q, int cnt) { const double inv = 1.0 / n; '
            "for (int i = 0; i < cnt; ++i) q[i] = xh_div_int(s[i], n, inv); }\n")
    (d / "shim.cpp").write_text(shim)
    subprocess.run(["g++", "-O1", "-ffp-contract=off", "-shared", "-fPIC", "-o", str(d / "libcommon_host.so"), str(d / "shim.cpp")], check=False, cwd=d).returncode == 0 or pytest.skip("g++ did not build the host stand-in here")
    lib = ctypes.CDLL(str(d / "libcommon_host.so"))
    vp = ctypes.c_void_p
    lib.f2key.argtypes = lib.key2f.argtypes = [vp, vp, ctypes.c_int]
    lib.cmp.argtypes = [vp, ctypes.c_int, ctypes.c_float, vp, vp, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
    lib.divint.argtypes = [vp, ctypes.c_double, vp, ctypes.c_int]
    return lib


def _floats(rng, n):
    """random bit patterns (every exponent, subnormals, both zeros, infinities, NaN) + a block of ordinary values"""
    bits = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    x = np.concatenate([bits.view(np.float32), rng.normal(0, 10, n).astype(np.float32),
                        np.array([0.0, -0.0, np.inf, -np.inf, np.nan, 1e-45, -1e-45, 3.4028235e38, -3.4028235e38], np.float32)])
    return x


def test_keys_preserve_the_order_of_floats(lib):
    rng = np.random.default_rng(1)
    x = _floats(rng, 200000)
    k = np.empty(len(x), np.uint32)
    lib.f2key(x.ctypes.data, k.ctypes.data, len(x))
    order = np.argsort(k, kind="stable")
    xs = x[order]
    fin = ~np.isnan(xs)
    assert not fin[np.argmin(fin):].any() if (~fin).any() else True      # NaN after everything else (numpy.sort's order)
    assert np.all(np.diff(xs[fin].astype(np.float64)) >= 0)              # ascending; -0.0 before +0.0 is still "not descending"
    assert (k[np.isnan(x)] == 0xFFFFFFFF).all() and (k[~np.isnan(x)] != 0xFFFFFFFF).all()
    back = np.empty(len(x), np.float32)
    lib.key2f(k.ctypes.data, back.ctypes.data, len(x))
    np.testing.assert_array_equal(back.view(np.uint32)[~np.isnan(x)], x.view(np.uint32)[~np.isnan(x)])   # bit-exact round trip
    assert np.isnan(back[np.isnan(x)]).all()
    # distinct floats get distinct keys, and -0.0 < +0.0 as keys (the callers that must tie them add +0.0f first)
    z = np.array([-0.0, 0.0], np.float32)
    kz = np.empty(2, np.uint32)
    lib.f2key(z.ctypes.data, kz.ctypes.data, 2)
    assert kz[0] + 1 == kz[1]


@pytest.mark.parametrize("op,fn", [(0, np.greater), (1, np.less), (2, np.greater_equal), (3, np.less_equal), (4, np.equal), (5, np.not_equal)])
def test_compare_and_its_one_compare_form(lib, op, fn):
    rng = np.random.default_rng(2 + op)
    a = _floats(rng, 50000)
    full, one, ok = np.empty(len(a), np.uint8), np.empty(len(a), np.uint8), ctypes.c_int(0)
    specials = [0.0, -0.0, 1e-45, -1e-45, 1.0, -1.0, 3.4028235e38, -3.4028235e38, np.inf, -np.inf, np.nan]
    for t in specials + list(a[rng.integers(0, len(a), 40)]) + list(rng.normal(0, 10, 10)):
        t = np.float32(t)
        lib.cmp(a.ctypes.data, op, ctypes.c_float(float(t)), full.ctypes.data, one.ctypes.data, ctypes.byref(ok), len(a))
        with np.errstate(invalid="ignore"):
            exp = fn(a, t)
        np.testing.assert_array_equal(full.astype(bool), exp, err_msg=f"xh_cmp_f32 op {op} t {t!r}")
        if ok.value:   # ordering operators against a finite threshold: one multiply + one compare must decide the same
            assert op < 4 and np.isfinite(t)
            np.testing.assert_array_equal(one.astype(bool), exp, err_msg=f"xh_one_cmp op {op} t {t!r}")
        else:
            assert op >= 4 or not np.isfinite(t)


def test_div_int_is_ieee_division(lib):
    rng = np.random.default_rng(9)
    for n in list(range(1, 70)) + [365, 366, 930, 10950, 55152] + list(rng.integers(2, 2 ** 20, 40)):
        s = np.concatenate([rng.normal(0, 1e3, 4000), rng.integers(0, 2 ** 40, 4000).astype(np.float64) * 0.5,
                            rng.random(2000) * 10.0 ** rng.integers(-300, 300, 2000), [0.0, -0.0, np.inf, -np.inf, np.nan]])
        q = np.empty(len(s))
        lib.divint(s.ctypes.data, ctypes.c_double(float(n)), q.ctypes.data, len(s))
        with np.errstate(all="ignore"):
            exp = s / float(n)
        ok = np.isfinite(exp) & (np.abs(exp) > 1e-290)      # (Markstein's argument needs the residual not to underflow)
        np.testing.assert_array_equal(q[ok], exp[ok], err_msg=f"n = {n}")
        assert np.array_equal(np.isnan(q), np.isnan(exp)) and np.array_equal(np.isinf(q), np.isinf(exp))
