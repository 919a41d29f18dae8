"""A/B of the fused model step with the derived statics recomputed (default) against streamed (LF_NO_RECOMPUTE=1), same
process: python tools/bench_recompute.py family size"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
sys.path.insert(0, ROOT)
from lisflood_amd import _lib, synthetic as syn          # noqa: E402
from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave  # noqa: E402
from bench_support import RoutingStepDevice  # noqa: E402

fam, size = sys.argv[1], int(sys.argv[2])
H = W = size
N = H * W
nsteps = 24
codes = syn.make_ldd(fam, H, W, {"shallow": 1, "deep": 2, "river": 7}[fam])
p = syn.router_params(N)
vals, dt = syn.model_step_values(N, p)
kw = kinematicWave(None, None, p["alpha"], p["beta"], p["dx"], dt, alpha_floodplains=vals["ChannelAlpha2"],
                   graph=Graph(ldd_raster=codes))
ref = None
for rep in range(2):
    for mode in ("1", "0"):
        os.environ["LF_NO_RECOMPUTE"] = mode
        st = RoutingStepDevice(kw, vals, True, p["beta"], 1 / dt, dt * nsteps)
        st.run_fused(nsteps)
        q = {k: st.download(k) for k in ("ChanQ", "Chan2QKin", "sumDisDay", "ChanM3Kin", "CrossSection2Area")}
        _lib.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            st.run_fused(nsteps)
        _lib.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / 3
        same = "-" if ref is None else str(all(np.array_equal(q[k], ref[k]) for k in q))
        if ref is None:
            ref = q
        print("%s %d^2 LF_NO_RECOMPUTE=%s: %.2f ms per model step  %.1f Gcell-steps/s identical=%s" % (
            fam, size, mode, ms, 2 * nsteps * N / ms / 1e6, same), flush=True)
        st.free()
