"""Loopback timing of several model steps per call on the row-block partition (runs on the GPU box):
`deep` / `shallow` S x S on B blocks of one GPU, 24 split-routing sub-steps per model step: ms per model step for the
single domain, for the partition model step by model step (loopback_substeps_fused) and with M model steps per call
(loopback_model_steps_fused)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "lisflood-code_amd"), ROOT, os.path.join(ROOT, "tests")]
from lisflood_amd import _lib, dist as D                    # noqa: E402
from lisflood_amd._lib import DeviceArray                    # noqa: E402
import test_dist_fused_gpu as T                              # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
nblocks = int(sys.argv[2]) if len(sys.argv) > 2 else 4
nsteps = 24
for family, seed in (("deep", 2), ("shallow", 1)):
    codes, p, vals, dt = T._case(family, size, size, seed, channel_frac=1.0)
    kw, ref = T._whole(codes, p, vals, dt, True, nsteps)
    ref.run_fused(nsteps); _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        ref.run_fused(nsteps)
    _lib.synchronize()
    single = (time.perf_counter() - t0) * 1e3 / 2
    steps, nph = T._blocks(codes, p, vals, dt, True, nsteps, nblocks)
    D.loopback_substeps_fused(steps, nsteps); _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        D.loopback_substeps_fused(steps, nsteps)
    _lib.synchronize()
    one = (time.perf_counter() - t0) * 1e3 / 2
    D.loopback_substeps_fused(steps, nsteps, lanes=True); _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        D.loopback_substeps_fused(steps, nsteps, lanes=True)
    _lib.synchronize()
    one_l = (time.perf_counter() - t0) * 1e3 / 2
    out = {"family": family, "size": size, "blocks": nblocks, "phases": nph, "single_domain_ms": round(single, 2),
           "partition_step_by_step_ms": round(one, 2), "partition_step_by_step_blocks_on_lanes_ms": round(one_l, 2)}
    for M in (2, 4, 5):
        sums = [DeviceArray((M, st.N)).zero() for st in steps]
        D.loopback_model_steps_fused(steps, nsteps, M, sums); _lib.synchronize()
        t0 = time.perf_counter()
        D.loopback_model_steps_fused(steps, nsteps, M, sums)
        _lib.synchronize()
        out["partition_%d_model_steps_per_call_ms_per_step" % M] = round((time.perf_counter() - t0) * 1e3 / M, 2)
        D.loopback_model_steps_fused(steps, nsteps, M, sums, lanes=False); _lib.synchronize()
        t0 = time.perf_counter()
        D.loopback_model_steps_fused(steps, nsteps, M, sums, lanes=False)
        _lib.synchronize()
        out["the_same_on_one_stream"] = out.get("the_same_on_one_stream", {})
        out["the_same_on_one_stream"][M] = round((time.perf_counter() - t0) * 1e3 / M, 2)
        for a in sums:
            a.free()
    print(out, flush=True)
    for st in steps:
        st.free()
    ref.free(); kw.close()
