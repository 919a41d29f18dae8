"""A/B of the level sweep against the component layout on one raster: python tools/bench_components.py family H W
[cap,bin ...].  Prints ms per engine-order kinematicWaveRouting call, same call, same box."""
import sys
import time

import numpy as np

sys.path.insert(0, "lisflood-code_amd")
from lisflood_amd import _lib, synthetic as syn          # noqa: E402
from lisflood_amd._lib import DeviceArray                # noqa: E402
from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave  # noqa: E402


def run(codes, comp, N, reps=5):
    t0 = time.time()
    g = Graph(ldd_raster=codes, components=comp)
    t_graph = time.time() - t0
    p = syn.router_params(N, seed=3)
    kw = kinematicWave(None, None, p["alpha"], p["beta"], p["dx"], p["dt"], graph=g)
    perm = g.layout()[0].astype(np.int64)
    dq = DeviceArray.from_host(np.ascontiguousarray(p["Q0"][perm]))
    dl = DeviceArray.from_host(np.ascontiguousarray(syn.lateral_inflow(N, 0)[perm]))
    for _ in range(2):
        kw.route_ordered(dq, dl)
    _lib.synchronize()
    _lib.timer_start()
    for _ in range(reps):
        kw.route_ordered(dq, dl)
    ms = _lib.timer_stop() / reps
    q = np.empty(N); q[perm] = dq.download()
    st = g.components
    launches = kw.last_launches()["launches"]
    dq.free(); dl.free(); kw.close()
    return ms, q, st, launches, t_graph


if __name__ == "__main__":
    fam, H, W = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    configs = [None] + [tuple(int(x) for x in a.split(",")) for a in sys.argv[4:]]
    codes = syn.make_ldd(fam, H, W, 1 if fam == "shallow" else 2)
    N = H * W
    ref = None
    for comp in configs:
        ms, q, st, launches, tg = run(codes, comp, N)
        same = "-" if ref is None else str(bool(np.array_equal(q, ref)))
        if ref is None:
            ref = q
        print("%s %dx%d comp=%s: %.3f ms/call  %.1f Gcell-steps/s  launches=%d identical=%s graph %.1fs %s"
              % (fam, H, W, comp, ms, N / ms / 1e6, launches, same, tg, st or ""), flush=True)
