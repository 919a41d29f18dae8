"""A/B of a model step (24 split-routing sub-steps) on the level wavefront against the component layout:
python tools/bench_fused_components.py family size [cap,bin ...]"""
import sys
import time

import numpy as np

sys.path.insert(0, "lisflood-code_amd")
sys.path.insert(0, ".")
from lisflood_amd import _lib, synthetic as syn          # noqa: E402
from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave  # noqa: E402
from bench_support import RoutingStepDevice  # noqa: E402

fam, size = sys.argv[1], int(sys.argv[2])
configs = [None] + [tuple(int(x) for x in a.split(",")) if "," in a else True for a in sys.argv[3:]]
H = W = size
N = H * W
nsteps = 24
codes = syn.make_ldd(fam, H, W, {"shallow": 1, "deep": 2}.get(fam, 7))
p = syn.router_params(N)
rng = np.random.default_rng(17)
beta, dt = p["beta"], 3600.0
alpha, length = p["alpha"], p["dx"]
alpha2 = alpha * rng.uniform(1.2, 2.0, N)
qlimit = 2.0 * p["Q0"] * rng.uniform(0.3, 1.2, N)
vals = dict(ChanLength=length, InvChanLength=1 / length, ChannelAlpha=alpha, InvChannelAlpha=1 / alpha,
            ChannelAlpha2=alpha2, InvChannelAlpha2=1 / alpha2, QLimit=qlimit, M3Limit=alpha * length * qlimit ** beta,
            Chan2M3Start=alpha2 * length * qlimit ** beta, Chan2QStart=qlimit * 0.1, PixelArea=np.full(N, 2.5e7),
            IsChannelKinematic=np.ones(N, bool), SideflowChanM3=syn.lateral_inflow(N, 0) * length * dt)
vals["Chan2M3Kin"] = vals["Chan2M3Start"].copy()
vals["ChanM3Kin"] = alpha * length * p["Q0"] ** beta
vals["ChanQKin"] = p["Q0"].copy()
vals["Chan2QKin"] = (vals["Chan2M3Kin"] / length / alpha2) ** (1 / beta)
ref = None
for comp in configs:
    g = Graph(ldd_raster=codes, components=comp)
    kw = kinematicWave(None, None, alpha, beta, length, dt, alpha_floodplains=alpha2, graph=g)
    st = RoutingStepDevice(kw, vals, True, beta, 1.0 / dt, dt * nsteps)
    st.run_fused(nsteps)
    q = st.download("ChanQ")
    _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        st.run_fused(nsteps)
    _lib.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 3
    same = "-" if ref is None else str(bool(np.array_equal(q, ref)))
    if ref is None:
        ref = q
    print("%s %d^2 comp=%s: %.2f ms per model step  %.1f Gcell-steps/s  launches=%d identical=%s %s"
          % (fam, size, comp, ms, 2 * nsteps * N / ms / 1e6, kw.last_launches()["launches"], same, g.components or ""),
          flush=True)
    st.free(); kw.close()
