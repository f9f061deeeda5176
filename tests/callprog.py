"""Replay of the call programs that tests/golden/make_call_programs.py recorded from the reference's own index bodies
(tests/golden/call_programs.json).  TEST INFRASTRUCTURE ONLY.

``make_index(name, prog, namespace)`` builds a function with the reference's parameter list and defaults that performs the
recorded operations on real objects; module globals ({"g": name}) are looked up BY NAME in ``namespace`` at call time —
hand it the ``__dict__`` of a stand-in module and the replay sees exactly what ``patch.install`` put there.  A global the
namespace lacks is a KeyError: the stand-in offers only what some recorded reference line asks for.  Scalar parameters
the body merely passes on are free; the others are bound to the recorded scenario (the body computed with them or
branched on them) and the replay refuses other values.
"""
import json
import operator
import os

_BIN = {"add": operator.add, "sub": operator.sub, "mul": operator.mul, "truediv": operator.truediv, "and": operator.and_,
        "or": operator.or_, "gt": operator.gt, "lt": operator.lt, "ge": operator.ge, "le": operator.le, "eq": operator.eq,
        "ne": operator.ne}
_UN = {"invert": operator.invert, "neg": operator.neg}


def load_programs(path=None):
    path = path or os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "call_programs.json")
    with open(path) as f:
        return json.load(f)


def run_program(prog, namespace, args):
    vals = []

    def dec(e):
        if "p" in e:
            return args[e["p"]]
        if "v" in e:
            return vals[e["v"]]
        if "g" in e:
            return namespace[e["g"]]
        if "c" in e:
            return e["c"]
        if "t" in e:
            return tuple(dec(x) for x in e["t"])
        return {k: dec(v) for k, v in e["d"].items()}

    for op in prog["ops"]:
        kind, r = op["op"], None
        if kind == "call":
            r = dec(op["fn"])(*[dec(a) for a in op["args"]], **{k: dec(v) for k, v in op["kwargs"].items()})
        elif kind == "getattr":
            r = getattr(dec(op["obj"]), op["name"])
        elif kind == "setattr":
            setattr(dec(op["obj"]), op["name"], dec(op["value"]))
        elif kind == "getitem":
            r = dec(op["obj"])[dec(op["key"])]
        elif kind == "setitem":
            dec(op["obj"])[dec(op["key"])] = dec(op["value"])
        elif kind == "binop":
            a, b = (dec(x) for x in op["args"])
            r = _BIN[op["name"]](a, b)
        elif kind == "unop":
            r = _UN[op["name"]](dec(op["args"][0]))
        elif kind == "enter":
            r = dec(op["obj"]).__enter__()
        elif kind == "exit":
            dec(op["obj"]).__exit__(None, None, None)
        elif kind == "contains":
            r = dec(op["item"]) in dec(op["obj"])
            if r != op["answer"]:
                raise AssertionError(f"the recorded run assumed {op['item']} in ... == {op['answer']}")
        else:
            raise ValueError(f"unknown recorded operation {kind!r}")
        vals.append(r)
    return dec(prog["ret"])


def make_index(name, prog, namespace):
    params, kwonly, defaults = prog["params"], set(prog["kwonly"]), prog["defaults"]
    positional = [p for p in params if p not in kwonly]
    bound = {p: v for p, v in prog["scenario"].items() if p not in prog["free"]}

    def index(*args, **kwargs):
        if len(args) > len(positional):
            raise TypeError(f"{name}() takes {len(positional)} positional arguments but {len(args)} were given")
        call = dict(defaults)
        call.update(zip(positional, args))
        for k, v in kwargs.items():
            if k not in params:
                raise TypeError(f"{name}() got an unexpected keyword argument {k!r}")
            call[k] = v
        missing = [p for p in params if p not in call]
        if missing:
            raise TypeError(f"{name}() missing arguments: {missing}")
        for p, v in bound.items():
            if call[p] != v:
                raise NotImplementedError(f"{name}: the recorded program holds for {p}={v!r} only (the body computes with it)")
        return run_program(prog, namespace, call)

    index.__name__ = name
    index.__doc__ = f"replay of the calls recorded from /root/reference/{prog['source']}"
    return index
