"""Generate golden vectors by EXECUTING the reference's own pure-numpy / njit function bodies.

Run in the build container only (needs /root/reference, which does not exist on the GPU box):

    python tests/golden/make_golden.py

The reference cannot be imported (xarray, numba, ... are not installed), but these functions depend on numpy only:
  * src/xclim/core/utils.py      calc_perc, nan_calc_percentiles, _compute_virtual_index, _get_gamma,
                                 _get_indexes, _linear_interpolation, _nan_quantile          (utl:279-557)
  * src/xclim/indices/run_length.py  _cumsum_reset_np (rl:143-151), _rle_1d (rl:1334-1340)   (@njit stripped)
They are AST-extracted (nothing is copied into this repository) and executed on seeded inputs; inputs and outputs
are stored in tests/golden/reference_vectors.npz and checked by tests/test_oracle_golden.py against oracle/.
"""

import ast
import os
import sys

import numpy as np

REF = "/root/reference/src/xclim"
HERE = os.path.dirname(os.path.abspath(__file__))


def extract(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"np": np, "Sequence": object}
    chunks = []
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            node.decorator_list = []  # strip @njit
            node.returns = None
            for a in node.args.args + node.args.kwonlyargs:
                a.annotation = None
            chunks.append(node)
    mod = ast.Module(body=chunks, type_ignores=[])
    ast.fix_missing_locations(mod)
    exec(compile(mod, path, "exec"), ns)
    missing = set(names) - set(ns)
    if missing:
        raise RuntimeError(f"not found in {path}: {missing}")
    return ns


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; golden vectors can only be regenerated in the build container")
    u = extract(os.path.join(REF, "core/utils.py"),
                ["calc_perc", "nan_calc_percentiles", "_compute_virtual_index", "_get_gamma", "_get_indexes",
                 "_linear_interpolation", "_nan_quantile"])
    r = extract(os.path.join(REF, "indices/run_length.py"), ["_cumsum_reset_np", "_rle_1d"])
    rng = np.random.default_rng(1234)
    out = {}

    # --- quantiles: (cells, N) fp32 with NaNs, several N, type 7 and type 8, several percentiles
    pers = [0.0, 1.0, 10.0, 40.0, 50.0, 90.0, 99.0, 100.0]
    for k, N in enumerate([1, 2, 3, 5, 8, 30, 150]):
        x = rng.normal(285, 8, (40, N)).astype(np.float32)
        x[rng.random(x.shape) < 0.15] = np.nan
        x[0] = np.nan
        if N > 1:
            x[1, 1:] = np.nan
        x[2] = 7.25
        out[f"q_in_{k}"] = x
        import warnings

        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out[f"q_t7_{k}"] = u["calc_perc"](x.copy(), percentiles=pers, alpha=1.0, beta=1.0)
            out[f"q_t8_{k}"] = u["calc_perc"](x.copy(), percentiles=pers, alpha=1.0 / 3.0, beta=1.0 / 3.0)
    out["q_pers"] = np.array(pers)
    # float64 input
    x64 = rng.normal(0, 1, (16, 11))
    x64[rng.random(x64.shape) < 0.1] = np.nan
    out["q64_in"] = x64
    out["q64_t8"] = u["calc_perc"](x64.copy(), percentiles=pers, alpha=1.0 / 3.0, beta=1.0 / 3.0)
    # the reference's own known answers (tests/test_utils.py:28-73)
    arr = np.asarray([15.0, 20.0, 35.0, 40.0, 50.0])
    out["ka_type7"] = u["nan_calc_percentiles"](arr, percentiles=[40.0], alpha=1, beta=1)
    out["ka_type8"] = u["nan_calc_percentiles"](np.stack([arr, arr]), percentiles=[40.0], alpha=1 / 3.0, beta=1 / 3.0)
    out["ka_partial_nan"] = u["nan_calc_percentiles"](np.asarray([np.nan, 41.0, 41.0, 43.0, 43.0]), percentiles=[50.0],
                                                      alpha=1 / 3.0, beta=1 / 3.0)

    # --- reset-cumsum: core dim LAST as apply_ufunc hands it (rl:209-216)
    b = (rng.random((6, 7, 50)) < 0.6).astype(np.uint8)
    out["cs_in"] = b
    out["cs_last"] = r["_cumsum_reset_np"](b.copy(), "last", np.uint8(1))
    out["cs_first"] = r["_cumsum_reset_np"](b.copy(), "first", np.uint8(1))
    bf = b.astype(np.float32)
    out["cs_last_f32"] = r["_cumsum_reset_np"](bf.copy(), "last", np.uint8(1))
    # --- true RLE of a 1-D series
    ia = rng.random(200) < 0.5
    v, l, p = r["_rle_1d"](ia)
    out["rle1d_in"], out["rle1d_v"], out["rle1d_l"], out["rle1d_p"] = ia, v, l, p

    path = os.path.join(HERE, "reference_vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "with", len(out), "arrays")


if __name__ == "__main__":
    main()
