"""Randomised comparison of the fused sub-step wavefront on level blocks (k_fused_cones) with the per-level wavefront
(LF_FUSED_LEVELS=1): families x sizes x sub-step counts x split / single x block lengths, isolated pixels and zero
states included.  Every state vector must be bit-identical.  python tools/stress_fused_cones.py [ncases] [seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lisflood-code_amd"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lisflood_amd import synthetic as syn          # noqa: E402
from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave  # noqa: E402
from bench_support import RoutingStepDevice  # noqa: E402

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
keys = ("ChanQ", "ChanQKin", "ChanM3Kin", "sumDisDay", "FlowVelocity", "TravelDistance")
keys2 = ("Chan2QKin", "Chan2M3Kin", "CrossSection2Area", "Sideflow1Chan")
bad = 0
for case in range(ncases):
    fam = ["shallow", "deep", "river", "saddle"][case % 4]
    H, W = int(rng.integers(40, 420)), int(rng.integers(40, 420))
    N = H * W
    nsteps = int(rng.choice([1, 2, 5, 24]))
    split = bool(rng.integers(0, 2))
    lmax = int(rng.choice([2, 3, 7, 16, 64]))
    codes = syn.make_ldd(fam, H, W, int(rng.integers(1, 1000)))
    if rng.uniform() < 0.5:                      # knock out some cells: isolated pixels, broken-up catchments
        flat = codes.reshape(-1)
        flat[rng.uniform(size=N) < 0.05] = 5
    p = syn.router_params(N, seed=int(rng.integers(1, 1000)))
    vals, dt = syn.model_step_values(N, p, seed=int(rng.integers(1, 1000)))
    if rng.uniform() < 0.5:                      # non-channel pixels with an all-zero state (the inert skip)
        dry = rng.uniform(size=N) < 0.3
        vals["IsChannelKinematic"] = ~dry
        for k in ("ChanQKin", "ChanM3Kin", "Chan2QKin", "Chan2M3Kin", "QLimit", "Chan2QStart", "Chan2M3Start", "M3Limit"):
            vals[k] = np.where(dry, 0.0, vals[k])
    g = Graph(ldd_raster=codes)
    out = []
    for L in (1, lmax):
        os.environ["LF_FUSED_LEVELS"] = str(L)
        kw = kinematicWave(None, None, p["alpha"], p["beta"], p["dx"], dt, alpha_floodplains=vals["ChannelAlpha2"], graph=g)
        st = RoutingStepDevice(kw, vals, split, p["beta"], 1.0 / dt, dt * nsteps)
        for rep in range(2):
            st.run_fused(nsteps)
        out.append({k: st.download(k) for k in keys + (keys2 if split else ())})
        launches = kw.last_launches()["launches"]
        st.free(); kw.close()
    same = all(np.array_equal(out[0][k], out[1][k], equal_nan=True) for k in out[0])
    bad += not same
    print("%-8s %3dx%-3d NL=%-4d nsteps=%-2d split=%d lmax=%-2d launches=%-4d %s" % (fam, H, W, g.num_levels, nsteps, split, lmax, launches,
                                                                                "identical" if same else "DIFFERENT"), flush=True)
    g.close()
print("cases %d, different %d" % (ncases, bad))
sys.exit(1 if bad else 0)
