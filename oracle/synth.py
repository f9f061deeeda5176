"""Oracle: the counter-based synthetic input generator of xh_fill_synthetic (xclim_amd/csrc/core.hip), restated in
numpy so that CPU and GPU see bit-identical inputs without shipping them.  TEST INFRASTRUCTURE ONLY.

value(seed, t, cell): key = seed*K0 + t*K1 + cell*K2 (mod 2^64); z = mix64(key); w = mix64(z + K1)
  kind 0 (temperature-like): base[t] + amp * (((u0 + u1) + (u2 + u3)) - 2),  u_i = 16-bit chunks of z / 65536
  kind 1 (precipitation-like): wet iff (w & 0xFFFFFF)/2^24 < p_wet; amount = ((ua*ua)*ua)*amp, ua = (z >> 40)/2^24
  NaN iff (((w >> 32) * 1e6) >> 32) < nan_per_million.   All float math is fp32 without FMA contraction.
"""

from __future__ import annotations

import numpy as np

K0 = np.uint64(0xD1342543DE82EF95)
K1 = np.uint64(0x9E3779B97F4A7C15)
K2 = np.uint64(0xC2B2AE3D27D4EB4F)


def _mix64(z):
    z = z ^ (z >> np.uint64(30))
    z = z * np.uint64(0xBF58476D1CE4E5B9)
    z = z ^ (z >> np.uint64(27))
    z = z * np.uint64(0x94D049BB133111EB)
    z = z ^ (z >> np.uint64(31))
    return z


def fill_synthetic(T, cells, kind, seed, base, amp, p_wet=0.3, nan_per_million=0):
    """`cells`: 1-D array of global cell ids.  Returns float32 (T, len(cells))."""
    cells = np.asarray(cells, dtype=np.uint64)
    base = np.asarray(base, dtype=np.float32)
    t = np.arange(T, dtype=np.uint64)[:, None]
    with np.errstate(over="ignore"):
        key = np.uint64(seed) * K0 + t * K1 + cells[None, :] * K2
        z = _mix64(key)
        w = _mix64(z + K1)
    f32 = np.float32
    if kind == 0:
        u = [((z >> np.uint64(s)) & np.uint64(0xFFFF)).astype(f32) * f32(1.0 / 65536.0) for s in (0, 16, 32, 48)]
        z4 = ((u[0] + u[1]) + (u[2] + u[3])) - f32(2.0)
        val = base[:, None] + f32(amp) * z4
    else:
        uw = (w & np.uint64(0xFFFFFF)).astype(f32) * f32(1.0 / 16777216.0)
        ua = (z >> np.uint64(40)).astype(f32) * f32(1.0 / 16777216.0)
        amount = ((ua * ua) * ua) * f32(amp)
        val = np.where(uw < f32(p_wet), base[:, None] + amount, f32(0.0)).astype(f32)
    r = (((w >> np.uint64(32)) & np.uint64(0xFFFFFFFF)) * np.uint64(1000000)) >> np.uint64(32)
    val = np.where(r < np.uint64(nan_per_million), f32(np.nan), val).astype(f32)
    return val


def seasonal_base(T, mean=288.0, amp=12.0, phase=100.0, period=365.0):
    """Host-side seasonal cycle table (float32), passed to both generators so no device trig is involved."""
    t = np.arange(T, dtype=np.float64)
    return (mean + amp * np.sin(2.0 * np.pi * (t - phase) / period)).astype(np.float32)
