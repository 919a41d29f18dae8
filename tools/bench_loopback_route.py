"""Router calls on the row-block partition with all blocks on ONE GPU (halo by device copy, tests' loopback driver):
python tools/bench_loopback_route.py family size nblocks [calls].  Wall time per call includes the Python driver; run
under rocprofv3 --kernel-trace --stats for the kernel time by kernel (what the tail phases cost)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
from lisflood_amd import _lib, dist as D, synthetic as syn  # noqa: E402

fam, S, R = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
calls = int(sys.argv[4]) if len(sys.argv) > 4 else 5
seed = {"shallow": 1, "deep": 2, "river": 7}[fam]
codes = syn.make_ldd(fam, S, S, seed)
N = S * S
p = syn.router_params(N, seed=6)
blocks = D.row_blocks(S, R)
graphs = [D.DistGraph(codes[r0:r1], None, codes[r0 - 1] if r0 > 0 else None, None, codes[r1] if r1 < S else None, None)
          for (r0, r1) in blocks]
nph = D.settle_phases_local(graphs)
sl = [slice(r0 * S, r1 * S) for (r0, r1) in blocks]
routers = [D.DistRouter(g, p["alpha"][s], p["beta"], p["dx"][s], p["dt"]) for g, s in zip(graphs, sl)]
Qs = [r.new_state(p["Q0"][s]) for r, s in zip(routers, sl)]
lats = [r.new_state(syn.lateral_inflow(N, 0)[s]) for r, s in zip(routers, sl)]
for g in graphs:
    pass
print("phases", nph, "cells per phase/part of the middle block:",
      [[graphs[R // 2].part_range(j, pt)[1] - graphs[R // 2].part_range(j, pt)[0] for pt in (0, 1)] for j in range(nph)],
      "launch units", [g.num_launch_units for g in graphs])
D.loopback_route(routers, Qs, lats)
_lib.synchronize()
t0 = time.perf_counter()
for c in range(calls):
    D.loopback_route(routers, Qs, lats)
_lib.synchronize()
print("%s %d^2, %d blocks on one GPU: %.3f ms per call (wall, Python-driven), launches per call %s" % (
    fam, S, R, (time.perf_counter() - t0) * 1e3 / calls, [int(r.last_launches()) for r in routers]))
