"""Same-process A/B of a model step (24 split-routing sub-steps, lf_routing_substeps_fused) with the cone kernel in its
chain / supply form (default) against the one-wavefront-per-cone kernel (LF_FUSED_SPLIT=0); every state vector compared
bit by bit; `bits` is a hash of the state vectors for comparisons ACROSS libraries (LISFLOOD_AMD_LIBRARY=another build).
    python tools/ab_fused_split.py family size [nsteps] [split 0|1]"""
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
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 24
split = (sys.argv[4] != "0") if len(sys.argv) > 4 else True
H = W = size
N = H * W
codes = syn.make_ldd(fam, H, W, {"shallow": 1, "deep": 2, "river": 7}[fam])
p = syn.router_params(N)
vals, dt = syn.model_step_values(N, p)
kw = kinematicWave(None, None, p["alpha"], p["beta"], p["dx"], dt, alpha_floodplains=vals["ChannelAlpha2"] if split else None,
                   graph=Graph(ldd_raster=codes))
names = ("ChanQ", "ChanQKin", "ChanM3Kin", "sumDisDay", "FlowVelocity", "TravelDistance") + (
    ("Chan2QKin", "Chan2M3Kin", "CrossSection2Area", "Sideflow1Chan") if split else ())
ref = None
for rep in range(2):
    for mode in ("0", "1", "auto"):
        os.environ.pop("LF_FUSED_SPLIT", None)
        if mode != "auto":
            os.environ["LF_FUSED_SPLIT"] = mode
        st = RoutingStepDevice(kw, vals, split, p["beta"], 1 / dt, dt * nsteps)
        st.run_fused(nsteps)
        st.run_fused(nsteps)
        q = {k: st.download(k) for k in names}
        _lib.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            st.run_fused(nsteps)
        _lib.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / 3
        same = "-" if ref is None else ",".join(k for k in names if not np.array_equal(q[k], ref[k], equal_nan=True)) or "all identical"
        if ref is None:
            ref = q
        chk = sum(int(q[k].view(np.uint64).sum(dtype=np.uint64)) for k in names) & 0xffffffffffffffff  # across libraries
        print("%s %d^2 %d sub-steps split=%d LF_FUSED_SPLIT=%s: %.2f ms per model step  %.1f Gcell-steps/s  launches=%d  differing: %s  bits %016x" % (
            fam, size, nsteps, split, mode, ms, (2 if split else 1) * nsteps * N / ms / 1e6, kw.last_launches()["launches"], same, chk),
            flush=True)
        st.free()
