"""Same-process A/B of the two shapes of k_sweep_cones_split (LF_ROUTE_SPLIT_SHAPE=few|many) against the automatic
choice and the one-wavefront kernel -- python tools/ab_cone_shape.py size family [family ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
import bench  # noqa: E402
from lisflood_amd import _lib, synthetic as syn  # noqa: E402

size = int(sys.argv[1])
for fam in sys.argv[2:]:
    kw, p, g = bench.build_case(fam, size, size)
    N = kw.num_pixels
    print(fam, size, kw.route_plan_stats(), flush=True)
    qs = [_lib.DeviceArray.from_host(syn.lateral_inflow(N, s)) for s in range(3)]
    tmp = _lib.DeviceArray(N)
    for d in qs:
        kw.to_engine_order(d, tmp)
        d.copy_from(tmp)
    ref = None
    for mode in ("old", "few", "many", "auto"):
        os.environ.pop("LF_ROUTE_SPLIT_SHAPE", None)
        os.environ["LF_ROUTE_SPLIT"] = "0" if mode == "old" else "1"
        if mode in ("few", "many"):
            os.environ["LF_ROUTE_SPLIT_SHAPE"] = mode
        Q = _lib.DeviceArray.from_host(p["Q0"])
        kw.to_engine_order(Q, tmp)
        Q.copy_from(tmp)
        for s in range(4):
            kw.route_ordered(Q, qs[s % 3])
        out = Q.download()
        ref = out if ref is None else ref
        _lib.synchronize()
        t0 = time.perf_counter()
        for s in range(10):
            kw.route_ordered(Q, qs[s % 3])
        _lib.synchronize()
        print("  %-5s %.4f ms per call, bit-identical to the one-wavefront kernel: %s" % (
            mode, (time.perf_counter() - t0) * 100, np.array_equal(out, ref)), flush=True)
        Q.free()
    for d in qs + [tmp]:
        d.free()
    kw.close()
