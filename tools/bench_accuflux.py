"""accuflux on the router's block plan against a router call and against one launch per level, device-resident
engine-order vectors: python tools/bench_accuflux.py family size"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
from lisflood_amd import _lib, synthetic as syn          # noqa: E402
from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave  # noqa: E402

fam, size = sys.argv[1], int(sys.argv[2])
H = W = size
N = H * W
codes = syn.make_ldd(fam, H, W, {"shallow": 1, "deep": 2, "river": 7}[fam])
p = syn.router_params(N)
kw = kinematicWave(None, None, p["alpha"], p["beta"], p["dx"], p["dt"], graph=Graph(ldd_raster=codes))
Q = kw.to_engine_order(_lib.DeviceArray.from_host(p["Q0"]))
q = kw.to_engine_order(_lib.DeviceArray.from_host(syn.lateral_inflow(N, 0)))
acc = _lib.DeviceArray(N)
L = _lib.lib()


def timed(f, reps=5):
    f()
    _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    _lib.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


ms_route = timed(lambda: kw.route_ordered(Q, q))
l_route = kw.last_launches()["launches"]
ms_acc = timed(lambda: _lib.check(L.lf_accuflux_ordered_device(kw._h, q.ptr, acc.ptr)))
l_acc = kw.last_launches()["launches"]
a1 = acc.download()
os.environ["LF_ROUTE_CONES"] = "0"
ms_lvl = timed(lambda: _lib.check(L.lf_accuflux_ordered_device(kw._h, q.ptr, acc.ptr)), reps=2)
l_lvl = kw.last_launches()["launches"]
same = np.array_equal(a1, acc.download())
print("%s %d^2: router call %.2f ms (%d launches) | accuflux on level blocks %.2f ms (%d launches) = %.2fx a router call | "
      "accuflux one launch per level %.2f ms (%d launches) | identical=%s" % (fam, size, ms_route, l_route, ms_acc, l_acc,
                                                                             ms_acc / ms_route, ms_lvl, l_lvl, same), flush=True)
