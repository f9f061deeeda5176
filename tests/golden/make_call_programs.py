"""Record, by EXECUTING the reference's own index bodies, the calls they make: "call programs" for the adapter tests.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_call_programs.py

Why: the adapter tests (tests/test_gpu_adapter.py) drive the wrappers of xclim_amd/xr_adapter.py through index functions
that live in stand-in modules (tests/fakexr.py).  Up to round 3 those index bodies were RE-TYPED from the reference by
hand — a test like that cannot find a place where the real body calls a replaced function differently (another keyword,
another argument order, an attribute it reads from the result).  The reference cannot be imported here (xarray, pint,
numba ... are not installed) and its sources must not be copied, but an index body is a few calls on its arguments: it is
AST-extracted (decorators and annotations stripped, exactly like make_golden.py does for the numpy bodies), executed
ONCE with symbolic arguments, and every operation it performs on them is recorded:

    call (of a global function, a method, a module attribute), getattr, setitem, binary operators, ``with`` enter / exit,
    ``in`` (answered from the scenario's assumptions, which are stored with the program)

The result — tests/golden/call_programs.json — holds, per index, its parameter list with the literal defaults of the
signature, the scenario (the scalar arguments of the recorded run), and the operation list with symbolic references
(parameters, earlier results, module globals by NAME).  tests/callprog.py replays a program inside a stand-in module's
namespace, so the by-name resolution rules of ``patch.install`` are exercised with the reference's own call sequences.
tests/test_host_cpu.py re-records the programs when /root/reference is present and compares them with the committed file.
"""

import ast
import builtins
import json
import os
import sys

REF = "/root/reference/src/xclim"
HERE = os.path.dirname(os.path.abspath(__file__))

# module file (under REF/indices) -> {index: scenario}; a scenario gives the scalar arguments of the recorded run
# ("*" marks the DataArray parameters) and the answers to ``in`` tests on symbolic values
INDICES = {
    "_multivariate.py": {
        "tx90p": {"args": {"tasmax": "*", "tasmax_per": "*"}},
        "tn10p": {"args": {"tasmin": "*", "tasmin_per": "*"}},
        "warm_spell_duration_index": {"args": {"tasmax": "*", "tasmax_per": "*"}},
        "cold_spell_duration_index": {"args": {"tasmin": "*", "tasmin_per": "*"}},
        "heat_wave_frequency": {"args": {"tasmin": "*", "tasmax": "*"}},
        "tx_tn_days_above": {"args": {"tasmin": "*", "tasmax": "*"}},
        "days_over_precip_thresh": {"args": {"pr": "*", "pr_per": "*"}, "assume": {"contains": True}},
        "heat_wave_max_length": {"args": {"tasmin": "*", "tasmax": "*"}},     # rl.resample_and_rl(..., window, reducer)
        # round 6
        "heat_wave_total_length": {"args": {"tasmin": "*", "tasmax": "*"}},
        "tg90p": {"args": {"tas": "*", "tas_per": "*"}},
        "tg10p": {"args": {"tas": "*", "tas_per": "*"}},
        "tn90p": {"args": {"tasmin": "*", "tasmin_per": "*"}},
        "tx10p": {"args": {"tasmax": "*", "tasmax_per": "*"}},
    },
    "_threshold.py": {
        "maximum_consecutive_dry_days": {"args": {"pr": "*"}},
        "maximum_consecutive_wet_days": {"args": {"pr": "*"}},
        "hot_spell_frequency": {"args": {"tasmax": "*"}},
        "hot_spell_max_length": {"args": {"tasmax": "*"}},
        "cold_spell_days": {"args": {"tas": "*"}},
        "growing_degree_days": {"args": {"tas": "*"}},
        "cooling_degree_days": {"args": {"tas": "*"}},
        "growing_season_length": {"args": {"tas": "*"}},
        "growing_season_start": {"args": {"tas": "*"}},
        "first_day_temperature_above": {"args": {"tas": "*"}},
        "tx_days_above": {"args": {"tasmax": "*"}},
        "tn_days_below": {"args": {"tasmin": "*"}},
        "dry_days": {"args": {"pr": "*"}},
        "wetdays": {"args": {"pr": "*"}},
        # round 5: the call shapes of replaced functions no recorded body had used yet (tests/test_host_cpu.py walks the
        # three index modules and lists every shape)
        "holiday_snow_and_snowfall_days": {"args": {"snd": "*", "prsn": "*"}},  # bivariate_count_occurrences(keywords only)
        "rprctot": {"args": {"pr": "*", "prc": "*"}},                            # compare without constrain
        "holiday_snow_days": {"args": {"snd": "*"}},                             # count_occurrences
        "days_with_snow": {"args": {"prsn": "*"}},                               # domain_count
        "snd_season_end": {"args": {"snd": "*"}},                                # season(2 positional)
        "dry_spell_frequency": {"args": {"pr": "*"}},                            # spell_length_statistics(**indexer)
        # round 6: every other body of the module that only calls replaced functions and the unit helpers (recorded with
        # the default scenario: DataArray parameters symbolic, scalars at their defaults)
        **{n: {"args": {"sfcWind": "*"}} for n in ("calm_days", "windy_days")},
        **{n: {"args": {"tas": "*"}} for n in ("cold_spell_frequency", "cold_spell_max_length", "cold_spell_total_length", "tg_days_above",
                                               "tg_days_below", "heating_degree_days", "growing_season_end", "first_day_temperature_below")},
        **{n: {"args": {"tasmin": "*"}} for n in ("frost_season_length", "frost_free_season_start", "frost_free_season_end",
                                                  "frost_free_season_length", "frost_free_spell_max_length", "tn_days_above",
                                                  "warm_night_frequency", "maximum_consecutive_frost_days",
                                                  "maximum_consecutive_frost_free_days")},
        **{n: {"args": {"tasmax": "*"}} for n in ("heat_wave_index", "hot_spell_total_length", "tx_days_below", "warm_day_frequency",
                                                  "maximum_consecutive_tx_days")},
        **{n: {"args": {"pr": "*"}} for n in ("dry_spell_total_length", "dry_spell_max_length", "wet_spell_frequency",
                                              "wet_spell_total_length", "wet_spell_max_length")},
    },
    "_simple.py": {
        "tg_mean": {"args": {"tas": "*"}},
        "tx_max": {"args": {"tasmax": "*"}},
        "tn_min": {"args": {"tasmin": "*"}},
        "frost_days": {"args": {"tasmin": "*"}},
        # round 6
        "tg_min": {"args": {"tas": "*"}}, "tn_max": {"args": {"tasmin": "*"}}, "tn_mean": {"args": {"tasmin": "*"}},
        "tx_mean": {"args": {"tasmax": "*"}}, "tx_min": {"args": {"tasmax": "*"}}, "hot_days": {"args": {"tasmax": "*"}},
        "ice_days": {"args": {"tasmax": "*"}}, "max_1day_precipitation_amount": {"args": {"pr": "*"}},
    },
}


class PStr(str):
    """A scalar argument that remembers which parameter it came from: passed on unchanged it is recorded as that
    parameter, not as the constant of this run (so the replay may use other values)."""


class PFloat(float):
    pass


class PInt(int):  # (also stands for bool parameters: True -> PInt(1))
    pass


def tag(name, v):
    cls = {str: PStr, float: PFloat, int: PInt, bool: PInt}.get(type(v))
    if cls is None:
        return v
    t = cls(v)
    t._p = name
    return t


class Recorder:
    def __init__(self, assume):
        self.ops, self.assume = [], dict(assume or {})

    def enc(self, x):
        if isinstance(x, Sym):
            return x._ref
        if getattr(x, "_p", None) is not None:
            return {"p": x._p}
        if isinstance(x, (str, int, float, bool)) or x is None:
            return {"c": x}
        if isinstance(x, (tuple, list)):
            return {"t": [self.enc(e) for e in x]}
        if isinstance(x, dict):
            return {"d": {k: self.enc(v) for k, v in x.items()}}
        raise TypeError(f"cannot record a value of type {type(x).__name__}")

    def op(self, kind, **fields):
        if len(self.ops) > 500:  # (a body that loops over a symbolic value would record forever)
            raise RuntimeError("more than 500 operations: the body is not a straight-line call program")
        self.ops.append(dict(op=kind, **fields))
        return Sym(self, {"v": len(self.ops) - 1})


def _binop(name, swap=False):
    def method(self, other):
        a, b = (other, self) if swap else (self, other)
        return self._r.op("binop", name=name, args=[self._r.enc(a), self._r.enc(b)])

    return method


class Sym:
    """A symbolic value: everything done to it is written to the recorder."""

    def __init__(self, rec, ref):
        object.__setattr__(self, "_r", rec)
        object.__setattr__(self, "_ref", ref)

    def __getattr__(self, name):
        if name.startswith("__") and name.endswith("__"):
            raise AttributeError(name)
        return self._r.op("getattr", obj=self._ref, name=name)

    def __setattr__(self, name, value):
        self._r.op("setattr", obj=self._ref, name=name, value=self._r.enc(value))

    def __call__(self, *args, **kwargs):
        return self._r.op("call", fn=self._ref, args=[self._r.enc(a) for a in args], kwargs={k: self._r.enc(v) for k, v in kwargs.items()})

    def __getitem__(self, key):
        return self._r.op("getitem", obj=self._ref, key=self._r.enc(key))

    def __setitem__(self, key, value):
        self._r.op("setitem", obj=self._ref, key=self._r.enc(key), value=self._r.enc(value))

    def __contains__(self, item):
        ans = bool(self._r.assume.get("contains", False))
        self._r.op("contains", obj=self._ref, item=self._r.enc(item), answer=ans)
        return ans

    def __enter__(self):
        self._r.op("enter", obj=self._ref)
        return self

    def __exit__(self, *exc):
        self._r.op("exit", obj=self._ref)
        return False

    def __bool__(self):
        raise TypeError("the index body branches on a symbolic value: not recordable")

    __hash__ = object.__hash__


for _n in ("add", "sub", "mul", "truediv", "and", "or", "gt", "lt", "ge", "le", "eq", "ne"):
    setattr(Sym, f"__{_n}__", _binop(_n))
    if _n in ("add", "sub", "mul", "truediv", "and", "or"):
        setattr(Sym, f"__r{_n}__", _binop(_n, swap=True))
Sym.__invert__ = lambda self: self._r.op("unop", name="invert", args=[self._ref])
Sym.__neg__ = lambda self: self._r.op("unop", name="neg", args=[self._ref])


class Globals(dict):
    """Module globals of the executed body: every free name is a symbolic global (recorded by NAME)."""

    def __init__(self, rec):
        super().__init__()
        self._rec = rec

    def __missing__(self, name):
        if hasattr(builtins, name):
            raise KeyError(name)
        s = Sym(self._rec, {"g": name})
        self[name] = s
        return s


def literal(node):
    try:
        return ast.literal_eval(node)
    except Exception:  # a non-literal default (none on the recorded indices)
        raise RuntimeError(f"non-literal default: {ast.dump(node)}")


def record(path, name, scenario):
    tree = ast.parse(open(path).read())
    fn = next((n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name), None)
    if fn is None:
        raise RuntimeError(f"{name} not found in {path}")
    line0, line1 = fn.lineno, fn.end_lineno
    decorators = [ast.unparse(d).split("(")[0] for d in fn.decorator_list]  # names only: what was stripped
    fn.decorator_list, fn.returns = [], None
    a = fn.args
    for arg in a.posonlyargs + a.args + a.kwonlyargs:
        arg.annotation = None
    for sub in ast.walk(fn):  # annotated assignments inside the body ("x: DataArray = f(...)")
        if isinstance(sub, ast.AnnAssign):
            sub.annotation = ast.Constant(value=None)
    # parameters and their literal defaults
    pos = [p.arg for p in a.posonlyargs + a.args]
    defaults = dict(zip(pos[len(pos) - len(a.defaults):], [literal(d) for d in a.defaults]))
    kwonly = [p.arg for p in a.kwonlyargs]
    defaults.update({p: literal(d) for p, d in zip(kwonly, a.kw_defaults) if d is not None})
    has_varkw = a.kwarg is not None
    rec = Recorder(scenario.get("assume"))
    ns = Globals(rec)
    mod = ast.Module(body=[fn], type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, "exec"), ns)
    call = {}
    for p in pos + kwonly:
        v = scenario["args"].get(p, defaults.get(p, "__missing__"))
        if v == "*":
            call[p] = Sym(rec, {"p": p})
        elif v == "__missing__":
            raise RuntimeError(f"{name}: no value for parameter {p}")
        else:
            call[p] = tag(p, v)
    del rec.ops[:]  # (nothing is recorded at definition time, but be sure)
    ret = dict.__getitem__(ns, name)(**call)
    rel = os.path.relpath(path, "/root/reference")
    return {
        "source": f"{rel}:{line0}-{line1}",
        "decorators": decorators,
        "params": pos + kwonly,
        "kwonly": kwonly,
        "varkw": has_varkw,
        "defaults": defaults,
        "arrays": [p for p, v in scenario["args"].items() if v == "*"],
        "scenario": {p: (bool(v) if isinstance(defaults.get(p, scenario["args"].get(p)), bool) else
                         {PStr: str, PFloat: float, PInt: int}.get(type(v), lambda z: z)(v))
                     for p, v in call.items() if not isinstance(v, Sym)},
        "ops": rec.ops,
        "ret": rec.enc(ret),
    }


_ALT = {"op": {">": ">=", "<": "<=", ">=": ">", "<=": "<"}}


def alternative(p, v):
    """Another value of the same kind for the scalar parameter `p` (None: leave it)."""
    if isinstance(v, bool):
        return not v
    if isinstance(v, int):
        return v + 1
    if isinstance(v, float):
        return v + 1.0
    if isinstance(v, str):
        if p == "freq":
            return "MS" if v != "MS" else "YS"
        if p in _ALT:
            return _ALT[p].get(v)
        if len(v) == 5 and v[2] == "-" and v.replace("-", "").isdigit():  # "MM-DD"
            return "02-15" if v != "02-15" else "03-15"
        parts = v.split(" ", 1)
        try:
            return f"{float(parts[0]) + 1.0:g} {parts[1]}" if len(parts) == 2 else None
        except ValueError:
            return None
    return None


def build():
    """Every index is recorded with its scenario and once more per scalar parameter with another value: a parameter whose
    change leaves the program the same (it is only passed on) is FREE — the replay may set it; the others are BOUND to
    the scenario's value (the body computed with them or branched on them)."""
    out = {}
    for fname, idxs in INDICES.items():
        modname = "xclim.indices." + fname[:-3]
        for name, scen in idxs.items():
            path = os.path.join(REF, "indices", fname)
            prog = record(path, name, scen)
            prog["module"] = modname
            free = []
            for p, v in prog["scenario"].items():
                alt = alternative(p, v)
                if alt is None:
                    continue
                try:
                    other = record(path, name, {"args": dict(scen["args"], **{p: alt}), "assume": scen.get("assume")})
                except Exception:
                    continue
                if other["ops"] == prog["ops"] and other["ret"] == prog["ret"]:
                    free.append(p)
            if "percentile_bootstrap" in prog["decorators"] and "bootstrap" in free:
                free.remove("bootstrap")  # consumed by the stripped decorator (core/bootstrapping.py:20-78): the body is its False branch
            prog["free"] = sorted(free)
            out[name] = prog
    return out


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; call programs can only be recorded in the build container")
    progs = build()
    with open(os.path.join(HERE, "call_programs.json"), "w") as f:
        json.dump(progs, f, indent=1, sort_keys=True)
    for k, p in progs.items():
        print(f"{k:32s} {p['source']:50s} {len(p['ops']):2d} ops, free: {', '.join(p['free'])}")


if __name__ == "__main__":
    main()
