"""TEST INFRASTRUCTURE (the stand-in's ``convert_units_to``; units are OUT OF SCOPE for the product, SURVEY.md §2 — round 3
shipped this file as xclim_amd/units.py).  Threshold units on the hot path (reference: core/units.py:334-420
``convert_units_to`` with the "hydro" context).

The reference parses quantities with pint.  The index functions of this backend take thresholds as floats in the units
of the data; this module is the small, dependency-free piece of ``convert_units_to`` an adapter needs for the threshold
strings the hot-path indicators use ("25 degC", "1 mm/day", "0.5 kg m-2 s-1", "10 cm"): temperatures, precipitation
rates (the hydro context equates 1 kg m-2 of water with 1 mm) and lengths.  Anything else raises ``ValueError``.
"""

from __future__ import annotations

import re

_NUM = re.compile(r"^\s*([-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?)\s*(.*?)\s*$")
_TEMP = {"k": "K", "degk": "K", "kelvin": "K", "degc": "C", "°c": "C", "c": "C", "celsius": "C", "degf": "F", "°f": "F", "f": "F",
         "fahrenheit": "F", "deg_c": "C", "degree_celsius": "C", "degrees_celsius": "C"}
_LENGTH = {"mm": 1e-3, "cm": 1e-2, "m": 1.0, "km": 1e3, "in": 0.0254, "inch": 0.0254}
_TIME = {"s": 1.0, "sec": 1.0, "second": 1.0, "h": 3600.0, "hr": 3600.0, "hour": 3600.0, "d": 86400.0, "day": 86400.0}
_WATER = 1000.0  # kg m-3: the hydro context (core/units.py: 1 kg m-2 of water == 1 mm)


def _split(q):
    m = _NUM.match(q)
    if not m:
        raise ValueError(f"cannot parse the quantity {q!r}")
    return float(m.group(1)), m.group(2)


def _classify(unit: str):
    """('temp', scale-name) | ('length', metres per unit) | ('rate', metres of water per second per unit, is_mass)."""
    u = unit.strip()
    key = u.lower().replace(" ", "")
    if key in _TEMP:
        return ("temp", _TEMP[key])
    if key in _LENGTH:
        return ("length", _LENGTH[key])
    norm = u.replace("**", "^").replace("^", "").replace("·", " ")
    norm = re.sub(r"\s*/\s*", "/", norm.strip())
    # mass flux: kg m-2 s-1 | kg/m2/s | kg m-2/s ...
    if norm.replace(" ", "") in ("kgm-2s-1", "kg/m2/s", "kgm-2/s", "kg/m2s-1"):
        return ("rate", 1.0 / _WATER, True)
    m = re.fullmatch(r"([a-zA-Z]+)(?:/|\s+)([a-zA-Z]+)(-1)?", norm)
    if m and m.group(1).lower() in _LENGTH and m.group(2).lower() in _TIME and (("/" in norm) != bool(m.group(3))):
        return ("rate", _LENGTH[m.group(1).lower()] / _TIME[m.group(2).lower()], False)
    raise ValueError(f"unit {unit!r} is not one of the threshold units supported here (temperature, length, precipitation rate)")


def convert_units_to(source, target: str, context: str | None = None) -> float:
    """Magnitude of `source` ("25 degC", "1 mm/day", or a float already in `target`) in the units `target`.
    ``context="hydro"`` allows mass flux <-> depth rate (kg m-2 s-1 <-> mm/day) as in the reference."""
    if not isinstance(source, str):
        return float(source)
    val, unit = _split(source)
    src, dst = _classify(unit), _classify(target)
    if src[0] != dst[0]:
        raise ValueError(f"cannot convert {unit!r} to {target!r}")
    if src[0] == "temp":
        kelvin = {"K": val, "C": val + 273.15, "F": (val - 32.0) * 5.0 / 9.0 + 273.15}[src[1]]
        return {"K": kelvin, "C": kelvin - 273.15, "F": (kelvin - 273.15) * 9.0 / 5.0 + 32.0}[dst[1]]
    if src[0] == "length":
        return val * src[1] / dst[1]
    if src[2] != dst[2] and context != "hydro":
        raise ValueError(f"converting {unit!r} to {target!r} needs context='hydro' (water mass <-> depth)")
    return val * src[1] / dst[1]
