"""Sensitivity of the level sweep to its bytes: per-pixel dx (52 B requested per cell) against a scalar dx (44 B).
python tools/bench_dx_scalar.py [size]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lisflood-code_amd"))
from lisflood_amd import _lib, synthetic as syn          # noqa: E402
from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
N = size * size
codes = syn.make_ldd("shallow", size, size, 1)
p = syn.router_params(N)
g = Graph(ldd_raster=codes)
perm = g.layout()[0].astype(np.int64)
for name, dx in (("per-pixel dx", p["dx"]), ("scalar dx", float(np.mean(p["dx"])))):
    kw = kinematicWave(None, None, p["alpha"], p["beta"], dx, p["dt"], graph=g)
    Q = _lib.DeviceArray.from_host(np.ascontiguousarray(p["Q0"][perm]))
    lat = _lib.DeviceArray.from_host(np.ascontiguousarray(syn.lateral_inflow(N, 0)[perm]))
    for _ in range(3):
        kw.route_ordered(Q, lat)
    _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        kw.route_ordered(Q, lat)
    _lib.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 20
    print("%s: %.4f ms per call, %.1f Gcell-steps/s" % (name, ms, N / ms / 1e6), flush=True)
    Q.free(); lat.free(); kw.close()
