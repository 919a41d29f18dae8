"""Same-process A/B of the cone sweep: one wavefront per cone (LF_ROUTE_SPLIT=0, the round-3 kernel) against the
supply/chain pair of wavefronts (default) -- python tools/ab_cone_split.py [size] [families...]
Prints ms per router call for both and whether the discharge after 6 calls is bit-identical."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
import bench  # noqa: E402
from lisflood_amd import _lib, synthetic as syn  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
fams = sys.argv[2:] or ["deep", "river"]
for fam in fams:
    kw, p, g = bench.build_case(fam, size, size)
    N = kw.num_pixels
    qs = [_lib.DeviceArray.from_host(syn.lateral_inflow(N, s)) for s in range(3)]
    tmp = _lib.DeviceArray(N)
    for d in qs:
        kw.to_engine_order(d, tmp)
        d.copy_from(tmp)
    res = {}
    for mode in ("0", "1", "0", "1"):
        os.environ["LF_ROUTE_SPLIT"] = mode
        Q = _lib.DeviceArray.from_host(p["Q0"])
        kw.to_engine_order(Q, tmp)
        Q.copy_from(tmp)
        for s in range(6):
            kw.route_ordered(Q, qs[s % 3])
        out = Q.download()
        _lib.synchronize()
        t0 = time.perf_counter()
        for s in range(10):
            kw.route_ordered(Q, qs[s % 3])
        _lib.synchronize()
        ms = (time.perf_counter() - t0) * 100
        print("%s %d split=%s: %.4f ms per call (%s)" % (fam, size, mode, ms, kw.last_launches()), flush=True)
        res.setdefault(mode, out)
        Q.free()
    same = np.array_equal(res["0"], res["1"])
    print("%s %d: bit-identical = %s  max |d| = %.3e  finite = %s" % (
        fam, size, same, float(np.nanmax(np.abs(res["0"] - res["1"]))), bool(np.isfinite(res["1"]).all())), flush=True)
    for d in qs:
        d.free()
    tmp.free()
