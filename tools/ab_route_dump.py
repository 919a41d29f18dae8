"""A/B helper: 12 router calls on a raster with the library LISFLOOD_AMD_LIBRARY names; prints ms per call and saves the
final discharge (engine order) -- python tools/ab_route_dump.py family size out.npy"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
import bench  # noqa: E402
from lisflood_amd import _lib, synthetic as syn  # noqa: E402

fam, size, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
kw, p, g = bench.build_case(fam, size, size)
N = kw.num_pixels
Q = _lib.DeviceArray.from_host(p["Q0"])
qs = [_lib.DeviceArray.from_host(syn.lateral_inflow(N, s)) for s in range(3)]
tmp = _lib.DeviceArray(N)
for d in [Q] + qs:
    kw.to_engine_order(d, tmp)
    d.copy_from(tmp)
for s in range(2):
    kw.route_ordered(Q, qs[s % 3])
_lib.synchronize()
t0 = time.perf_counter()
for s in range(10):
    kw.route_ordered(Q, qs[s % 3])
_lib.synchronize()
ms = (time.perf_counter() - t0) * 100
print("%s %s %d: %.4f ms per call" % (os.path.basename(os.environ.get("LISFLOOD_AMD_LIBRARY", "default")), fam, size, ms),
      flush=True)
np.save(out, Q.download())
