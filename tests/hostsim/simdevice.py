"""A Device whose C ABI is the HOST SIMULATION build of the translation units that need neither LDS nor wave intrinsics
(tests/hostsim/hip/hip_runtime.h: every kernel runs thread by thread on the CPU).  Test infrastructure for the `-m "not gpu"`
tier only — it pins kernel arithmetic and entry-point dispatch against the oracle on machines without a GPU; entry points
that are not simulated raise (never a silent no-op), and nothing of this is reachable from the product package."""
import ctypes as C
import os
import re
import shutil
import subprocess
import sys
import threading

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.join(HERE, "..", "..")
CSRC = os.path.join(ROOT, "xclim_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))

from mock_device import MockDevice, _MockLib  # noqa: E402
from xclim_amd import _capi  # noqa: E402

SIMULATED_UNITS = ("detrend", "window", "runlen", "reduce", "spell", "elemwise", "eqm", "plane", "wquantile")
# compiled, but their kernels (or the selection kernels behind them) speak to the wave: refused
WAVE_ENTRY_POINTS = ()
# eqm.hip votes `__all(m == nq)` only to pick between two forms that are each right for the lane that takes them
UNIT_DEFINES = {"eqm": ["-D__all(x)=((x)!=0)"],
                # plane.hip appends to its work lists wave by wave and keeps lane-private LDS columns: see wave_of_one.h
                "plane": ["-include", os.path.join(HERE, "wave_of_one.h")],
                # wquantile.hip's LDS arrays are lane-private columns ([i * 64 + lane]): static arrays do
                "wquantile": ["-D__shared__=static"]}


# units whose kernels talk through LDS / the wave in WAVE-UNIFORM control flow: every workgroup as a set of fibers (simt.h)
FIBER_UNITS = ("f64", "select", "select5", "tcount", "qdm", "quantile", "doystats", "core", "reduce2", "pdoy_top", "pdoy_quad", "pdoy_walk", "select3", "qdm2", "select2", "select4", "winsel")
# topnet.h (the comparator networks of the register percentile kernels) issues v_min_f32 / v_max_f32 and a NaN-replace-and-count
# triple as inline ISA: four statements, rewritten to the C++ they stand for (NaN never enters the min / max: the callers replace
# it first), in a copy of the header that the fiber units include instead
HEADER_REWRITES = {"topnet.h": [
    ('asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));', "r = a < b ? a : b;"),
    ('asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));', "r = a > b ? a : b;"),
    ('''  asm("v_cmp_u_f32 vcc, %2, %2\\n\\tv_cndmask_b32 %0, %2, %3, vcc\\n\\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "=&v"(key), "+v"(nn)
      : "v"(raw), "v"(sentinel)
      : "vcc");''', "  { const bool n_ = raw != raw; key = n_ ? sentinel : raw; nn += n_ ? 1 : 0; }"),
    ('''  asm("v_cmp_u_f32 vcc, %0, %0\\n\\tv_cndmask_b32 %0, %0, %2, vcc\\n\\tv_addc_co_u32 %1, vcc, 0, %1, vcc"
      : "+v"(v), "+v"(nn)
      : "v"(sentinel)
      : "vcc");''', "  { const bool n_ = v != v; v = n_ ? sentinel : v; nn += n_ ? 1 : 0; }"),
]}
# The register sorting networks (select3.hip, qdm2.hip) split a sorted column across the lane pair with DPP moves written as ISA:
# xor with the lane's sign mask, take the partner's complement (v_not_b32_dpp quad_perm [1,0,3,2]), keep the larger — the same in C++
# with a shuffle; qdm2.hip's key conversion counts NaN with a compare / select / add-with-carry triple.
_SPLIT_PAIR = (r"#define XH_SP\(i, j\).*?\n  \}\n",
               "#define XH_SP(i, j) { k##i ^= mA3; k##j ^= mA3; const uint32_t t0_ = ~__shfl_xor(k##j, 1), t1_ = ~__shfl_xor(k##i, 1); "
               "k##i = k##i > t0_ ? k##i : t0_; k##j = k##j > t1_ ? k##j : t1_; }\n")
_SPLIT_ONE = (r"#define XH_SM\(m\).*?\n  \}\n",
              "#define XH_SM(m) { k##m ^= mA3; const uint32_t t0_ = ~__shfl_xor(k##m, 1); k##m = k##m > t0_ ? k##m : t0_; }\n")
UNIT_REWRITES = {
    # select2.hip counts through the wave on VCC (v_cmp + s_bcnt1_i32_b64): a vote + a population count
    "select2": [(r'asm volatile\("v_cmp_o_f32 vcc, %3, %3\\n\\ts_bcnt1_i32_b64 %1, vcc\\n\\tv_cndmask_b32 %0, -1, %2, vcc"\s*: "=v"\(o\), "=s"\(c\)\s*: "v"\(kk\), "v"\(u\)\s*: "vcc", "scc"\);',
                 "{ const bool o_ = __uint_as_float(u) == __uint_as_float(u); c = (uint32_t)__popcll(__ballot(o_)); o = o_ ? kk : 0xFFFFFFFFu; }"),
                (r'asm volatile\("v_cmp_eq_u32 vcc, %1, %2\\n\\ts_bcnt1_i32_b64 %0, vcc" : "=s"\(c\) : "v"\(key\[k\]\), "v"\(kmin\) : "vcc", "scc"\);',
                 "c = (uint32_t)__popcll(__ballot(key[k] == kmin));")],
    "select3": [_SPLIT_PAIR, _SPLIT_ONE],
    # select4.hip (the streaming two-pass selection): the lane-xor exchanges of its wave-wide bitonic sort are DPP moves (quad_perm,
    # row_shl / row_shr / row_ror) = the value of lane ^ m; the sign of a float difference is one v_med3_i32 on its bits; column
    # extremes through v_min / v_max (which return the other operand for a NaN one, like fminf / fmaxf); the candidate appends are
    # sixteen LDS writes under the execution mask (v_cmpx) at an address kept as a 32-bit LDS offset
    # winsel.hip (the sliding sorted window of the day-of-year training): the same DPP lane exchanges as select4's wave sort
    "winsel": [(r"(__device__ __forceinline__ uint32_t ws_lane_xor\(uint32_t v, int m\) \{).*?\n\}\n", r"\1 return (uint32_t)__shfl_xor((int)v, m); }\n")],
    "select4": [
        (r"(__device__ __forceinline__ uint32_t hs_lane_xor\(uint32_t v, int m\) \{).*?\n\}\n", r"\1 return (uint32_t)__shfl_xor((int)v, m); }\n"),
        (r'asm\("v_med3_i32 %0, %1, -1, 1" : "=v"\(r\) : "v"\(z\)\);', "{ int zi_; memcpy(&zi_, &z, 4); r = zi_ < -1 ? -1 : (zi_ > 1 ? 1 : zi_); }"),
        (r'asm\("v_min_f32 %0, %0, %1" : "\+v"\((\w+)\) : "v"\(([^;]+?)\)\);', r"\1 = fminf(\1, \2);"),
        (r'asm\("v_max_f32 %0, %0, %1" : "\+v"\((\w+)\) : "v"\(([^;]+?)\)\);', r"\1 = fmaxf(\1, \2);"),
        (r'asm\("v_min3_f32 %0, %0, %1, %2" : "\+v"\((\w+)\) : "v"\(([^;]+?)\), "v"\(([^;]+?)\)\);', r"\1 = fminf(fminf(\1, \2), \3);"),
        (r'asm\("v_max3_f32 %0, %0, %1, %2" : "\+v"\((\w+)\) : "v"\(([^;]+?)\), "v"\(([^;]+?)\)\);', r"\1 = fmaxf(fmaxf(\1, \2), \3);"),
        # (round 6: hs_qdm_pick scans the column's records in LDS — no v_readlane from divergent code is left to emulate)
        (r"typedef __attribute__\(\(address_space\(3\)\)\) uint32_t lds_u32;", ""),
        (r"uint32_t addr = \(uint32_t\)\(uintptr_t\)\(lds_u32\*\)\(cand \+ lbase\[colo\]\) \+ pos \* 4u;", "uint32_t* addr = (uint32_t*)(cand + lbase[colo]) + pos;"),
        (r'asm volatile\(\s*"s_mov_b64 %\[sv\], exec\\n\\t".*?: "vcc", "memory"\);', "(void)sv; if (bit) { uint32_t b_; memcpy(&b_, &v[u], 4); *addr++ = b_; }"),
    ],
    "qdm2": [_SPLIT_PAIR, _SPLIT_ONE,
             (r'asm volatile\("v_cmp_u_f32 vcc, %2, %2\\n\\tv_cndmask_b32_e64 %0, %0, -1, vcc\\n\\tv_addc_co_u32_e32 %1, vcc, 0, %1, vcc" \\\n\s*: "\+v"\(kk_\), "\+v"\(nanc\) : "v"\(f_\) : "vcc"\);',
              "{ const bool n_ = f_ != f_; kk_ = n_ ? 0xFFFFFFFFu : kk_; nanc += n_ ? 1u : 0u; }")],
}
_DYN_LDS = re.compile(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([\w ]+?)\s+(\w+)\[\];")


def build(workdir: str) -> str:
    """g++ the simulated translation units (sources unchanged, except that an `extern __shared__ T name[];` of a fiber unit
    becomes a pointer to the workgroup's LDS buffer) + sim_runtime.cpp into workdir/libxclimhip_hostsim.so."""
    if shutil.which("g++") is None:
        raise RuntimeError("no g++")
    for header, rules in HEADER_REWRITES.items():
        text = open(os.path.join(CSRC, header)).read()
        for old, new in rules:
            if old not in text:
                raise RuntimeError(f"{header}: the statement the simulation rewrites has changed: {old[:60]!r}")
            text = text.replace(old, new)
        open(os.path.join(workdir, header), "w").write(text)
    flags = ["-std=c++17", "-O1", "-fPIC", "-ffp-contract=off", "-I", workdir, "-I", HERE, "-I", CSRC]
    # HOSTSIM_SANITIZE=undefined (or address,undefined with LD_PRELOAD=libasan.so): an audit build of the kernels under the
    # compiler's sanitizers — out-of-bounds LDS / scratch accesses, shifts, signed overflow
    san = os.environ.get("HOSTSIM_SANITIZE")
    if san:
        flags += ["-g", f"-fsanitize={san}", "-fno-sanitize-recover=all" if os.environ.get("HOSTSIM_SANITIZE_FATAL") else "-fsanitize-recover=all"]
    def compile_unit(unit):
        obj = os.path.join(workdir, unit + ".o")
        src = os.path.join(CSRC, unit + ".hip")
        extra = list(UNIT_DEFINES.get(unit, []))
        if unit in FIBER_UNITS:
            text = open(src).read()
            if unit == "core":   # the runtime half of core.hip is sim_runtime.cpp's job: only its kernels (synthetic fields, transposes)
                text = '#include "common.h"\n' + text[text.index("// ---- synthetic generator"):]
            text = _DYN_LDS.sub(lambda m: f"{m.group(1)}* {m.group(2)} = ({m.group(1)}*)sim_dynamic_lds();", text)
            # the LDS-only workgroup barrier (s_waitcnt lgkmcnt(0); s_barrier) is a workgroup barrier; empty asm statements are
            # compiler fences whose operand class "v" / "s" (a VGPR / SGPR) becomes "r"
            text = text.replace('asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ::: "memory")', "__syncthreads()")
            text = re.sub(r'asm volatile\(""\s*:\s*"\+[vs]"', 'asm volatile("" : "+r"', text)
            for pat, rep in UNIT_REWRITES.get(unit, []):
                text, nsub = re.subn(pat, rep, text, flags=re.S)
                if nsub == 0:
                    raise RuntimeError(f"{unit}.hip: the statement the simulation rewrites has changed: {pat[:50]!r}")
            src = os.path.join(workdir, unit + ".sim.cpp")
            open(src, "w").write(text)
            extra += ["-DSIM_FIBERS=1", "-D__shared__=static"]
            if unit == "select4" and san and "undefined" in san and not os.environ.get("HOSTSIM_SANITIZE_BOUNDS_ONLY"):
                # g++ 11: with ALL of -fsanitize=undefined the index check of `v[u]` on the ring's register set (a reference to
                # an array handed to a lambda) reads a wrong temporary and the access after it faults — with the loop variable
                # verified intact by an explicit check in front of it.  Either half alone is clean: this build carries every
                # check but `bounds`; HOSTSIM_SANITIZE=bounds HOSTSIM_SANITIZE_BOUNDS_ONLY=1 is the other half.
                extra += ["-fno-sanitize=bounds"]
        subprocess.run(["g++", "-x", "c++", *flags, *extra, "-c", src, "-o", obj], check=True)
        return obj

    from concurrent.futures import ThreadPoolExecutor

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:   # (g++ runs outside the GIL)
        objs = list(pool.map(compile_unit, SIMULATED_UNITS + FIBER_UNITS))
    obj = os.path.join(workdir, "sim_runtime.o")
    subprocess.run(["g++", *flags, "-c", os.path.join(HERE, "sim_runtime.cpp"), "-o", obj], check=True)
    out = os.path.join(workdir, "libxclimhip_hostsim.so")
    subprocess.run(["g++", "-shared", *([f"-fsanitize={san}"] if san else []), "-o", out, *objs, obj], check=True)
    return out


class _SimLib:
    """The simulated entry points with the prototypes of xclim_amd._capi; memory entry points on host buffers (MockDevice's);
    anything else raises."""

    def __init__(self, path: str):
        self._dll = C.CDLL(path)
        self._mock = _MockLib()

    def __getattr__(self, name):
        if not name.startswith("xh_"):
            raise AttributeError(name)
        try:
            if name in WAVE_ENTRY_POINTS:
                raise AttributeError(name)
            fn = getattr(self._dll, name)
        except AttributeError:
            # memory entry points on host buffers; fences between the copy lanes and page-locking are no-ops here (one thread,
            # everything in order, host memory)
            if name in _MockLib._SPECIAL or name in ("xh_malloc", "xh_free", "xh_timer_start", "xh_lane_fence", "xh_lane_sync",
                                                      "xh_host_register", "xh_host_unregister"):
                return getattr(self._mock, name)
            raise NotImplementedError(f"{name}: not simulated on the host (kernels written at ISA level, or a runtime service the "
                                      "simulation does not provide)") from None
        fn.argtypes = _capi.SIGNATURES.get(name)
        fn.restype = _capi._RESTYPES.get(name, C.c_int)
        setattr(self, name, fn)
        return fn


class SimDevice(MockDevice):
    def __init__(self, path: str):
        super().__init__(0)
        self.path = path
        self.lib = _SimLib(path)
        ctx = C.c_void_p()
        self.lib._dll.xh_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        assert self.lib._dll.xh_create(0, C.byref(ctx)) == 0
        self.ctx = ctx
        self.lock = threading.RLock()

    def sync(self):
        return None
