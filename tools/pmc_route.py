"""Lean driver for PMC passes: K engine-order router calls (lf_router_route_ordered) on one synthetic family, or -- with
`fused` -- model steps of 24 split-routing sub-steps (lf_routing_substeps_fused).  Nothing else is launched, so every
kernel of a rocprofv3 pass belongs to the workload.

    python tools/pmc_route.py route <family> <size> <calls>
    python tools/pmc_route.py fused <family> <size> <model steps>
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lisflood-code_amd"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lisflood_amd import _lib, synthetic as syn          # noqa: E402
from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave  # noqa: E402

mode, fam, size, reps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
H = W = size
N = H * W
codes = syn.make_ldd(fam, H, W, {"shallow": 1, "deep": 2, "river": 7}[fam])
p = syn.router_params(N)
g = Graph(ldd_raster=codes)
if mode == "route":
    kw = kinematicWave(None, None, p["alpha"], p["beta"], p["dx"], p["dt"], graph=g)
    Q = _lib.DeviceArray.from_host(p["Q0"])
    q = _lib.DeviceArray.from_host(syn.lateral_inflow(N, 0))
    tmp = _lib.DeviceArray(N)
    for d in (Q, q):
        kw.to_engine_order(d, tmp)
        d.copy_from(tmp)
    _lib.synchronize()
    for _ in range(reps):
        kw.route_ordered(Q, q)
    _lib.synchronize()
    print("route", fam, size, "calls", reps, "launches/call", kw.last_launches()["launches"], "NL", g.num_levels, flush=True)
else:
    from bench_support import RoutingStepDevice
    nsteps = 24
    vals, dtr = syn.model_step_values(N, p)
    kw = kinematicWave(None, None, p["alpha"], p["beta"], p["dx"], dtr, alpha_floodplains=vals["ChannelAlpha2"], graph=g)
    st = RoutingStepDevice(kw, vals, True, p["beta"], 1.0 / dtr, dtr * nsteps)
    for _ in range(reps):
        st.run_fused(nsteps)
    _lib.synchronize()
    print("fused", fam, size, "model steps", reps, "launches/step", kw.last_launches()["launches"], "NL", g.num_levels,
          flush=True)
