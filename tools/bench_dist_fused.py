"""A model step (24 split-routing sub-steps) on the row-block partition, all blocks on ONE GPU (halo by device copy):
lf_dist_fused_phase per block and phase against the single-domain wavefront and, optionally, the sub-step-by-sub-step
partition path.  python tools/bench_dist_fused.py family size nblocks [seq]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
sys.path.insert(0, ROOT)
from lisflood_amd import _lib, dist as D, synthetic as syn          # noqa: E402
from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave  # noqa: E402
from bench_support import RoutingStepDevice  # noqa: E402

fam, size, nblocks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
H = W = size
N = H * W
nsteps = 24
codes = syn.make_ldd(fam, H, W, {"shallow": 1, "deep": 2, "river": 7}[fam])
p = syn.router_params(N)
vals, dt = syn.model_step_values(N, p)
kw = kinematicWave(None, None, p["alpha"], p["beta"], p["dx"], dt, alpha_floodplains=vals["ChannelAlpha2"],
                   graph=Graph(ldd_raster=codes))
ref = RoutingStepDevice(kw, vals, True, p["beta"], 1 / dt, dt * nsteps)
ref.run_fused(nsteps)
_lib.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    ref.run_fused(nsteps)
_lib.synchronize()
ms_ref = (time.perf_counter() - t0) * 1e3 / 3
print("%s %d^2 single domain: %.2f ms per model step, %d launches" % (fam, size, ms_ref, kw.last_launches()["launches"]), flush=True)
blocks = D.row_blocks(H, nblocks)
graphs = [D.DistGraph(codes[r0:r1], None, codes[r0 - 1] if r0 > 0 else None, None, codes[r1] if r1 < H else None, None)
          for (r0, r1) in blocks]
nph = D.settle_phases_local(graphs)
sl = [slice(r0 * W, r1 * W) for (r0, r1) in blocks]
routers = [D.DistRouter(g, p["alpha"][s], p["beta"], p["dx"][s], dt, alpha_floodplains=vals["ChannelAlpha2"][s])
           for g, s in zip(graphs, sl)]
steps = [D.DistRoutingStep(r, {k: (a[s] if isinstance(a, np.ndarray) else a) for k, a in vals.items()}, True, p["beta"],
                           1 / dt, dt * nsteps) for r, s in zip(routers, sl)]
for g in graphs:
    per_phase = [g.phase_range(j) for j in range(nph)]
    print("  block: cells %d units %d phases %d cells per phase %s non-contiguous %d" % (
        g.num_pixels, g.num_launch_units, nph, [b - a for a, b in per_phase], g.num_noncontiguous), flush=True)
for st in steps:      # same start as the reference's 4 model steps
    pass
for rep in range(1):
    D.loopback_substeps_fused(steps, nsteps)
_lib.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    D.loopback_substeps_fused(steps, nsteps)
_lib.synchronize()
ms = (time.perf_counter() - t0) * 1e3 / 3
same = all(np.array_equal(np.concatenate([st.download(k) for st in steps]), ref.download(k)) for k in ("ChanQ", "sumDisDay", "Chan2QKin"))
print("%s %d^2 in %d row blocks on one GPU, fused per phase: %.2f ms per model step (%.2fx the single domain), launches %s, "
      "identical=%s" % (fam, size, nblocks, ms, ms / ms_ref, [r.last_launches() for r in routers], same), flush=True)
if len(sys.argv) > 4:
    t0 = time.perf_counter()
    for _ in range(nsteps):
        D.loopback_substep(steps)
    _lib.synchronize()
    print("  sub-step by sub-step on the partition: %.2f ms per model step" % ((time.perf_counter() - t0) * 1e3), flush=True)
