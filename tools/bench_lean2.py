"""k_select_lean alone on a padded, 16-byte aligned time-minor copy (the layout of the transposed scratch)."""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from xclim_amd import kernels as K
from xclim_amd._capi import Device, _vp, np_ptr
T, C = 10950, int(sys.argv[1]) if len(sys.argv) > 1 else 12096 * 4
Tp = 11072
dev = Device(0)
x = K.fill_synthetic(dev, T, C, 0, 4, bench.seasonal_base(T), 3.0)
xT = dev.empty((C, Tp), np.float32)
dev.call("xh_transpose_f32", _vp(x.ptr), T, C, C, _vp(xT.ptr), Tp)
q = np.ascontiguousarray((np.arange(20) + 0.5) / 20)
out = dev.empty((20, C), np.float32)
fn = lambda: dev.call("xh_quantile_series", _vp(xT.ptr), T, C, 1, Tp, np_ptr(q), 20, _vp(out.ptr))
ms = bench.event_time(dev, fn, 3)
print(json.dumps({"T": T, "C": C, "abl": os.environ.get("XH_SELECT_ABL"), "noglds": os.environ.get("XH_LEAN_NOGLDS"),
                  "select_only_ms_per_12096_cols": ms * 12096 / C}))
