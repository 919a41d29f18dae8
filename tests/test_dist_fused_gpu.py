"""-m gpu: a whole model step of routing (NoRoutSteps x routing.dynamic, routing.py:512-603) on the row-block partition
as lf_dist_routing_substeps_fused runs it -- phase by phase, every sub-step of a phase as one wavefront, one halo block
per phase -- against lf_routing_substeps_fused on the whole raster: bit-identical.  The blocks live on ONE GPU and the
halo travels by device copy (the test boxes have one GPU); kernels and plan are those of the RCCL path."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    from lisflood_amd import _lib
    if _lib.device_count() == 0:
        pytest.fail("no HIP device: the gpu tests must run on an MI355X box")
    return _lib


def _case(family, H, W, seed, channel_frac=0.9):
    from lisflood_amd import synthetic as syn
    N = H * W
    codes = syn.make_ldd(family, H, W, seed)
    p = syn.router_params(N, seed=6)
    vals, dt = syn.model_step_values(N, p, seed=19)
    if channel_frac < 1:
        vals["IsChannelKinematic"] = np.random.default_rng(3).random(N) < channel_frac
    return codes, p, vals, dt


def _whole(codes, p, vals, dt, split, nsteps):
    from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave
    from bench_support import RoutingStepDevice
    kw = kinematicWave(None, None, p["alpha"], p["beta"], p["dx"], dt,
                       alpha_floodplains=vals["ChannelAlpha2"] if split else None, graph=Graph(ldd_raster=codes))
    ref = RoutingStepDevice(kw, vals, split, p["beta"], 1 / dt, dt * nsteps)
    return kw, ref


def _blocks(codes, p, vals, dt, split, nsteps, nblocks):
    from lisflood_amd import dist as D
    H, W = codes.shape
    blocks = D.row_blocks(H, nblocks)
    graphs = [D.DistGraph(codes[r0:r1], None, codes[r0 - 1] if r0 > 0 else None, None,
                          codes[r1] if r1 < H else None, None) for (r0, r1) in blocks]
    nph = D.settle_phases_local(graphs)
    sl = [slice(r0 * W, r1 * W) for (r0, r1) in blocks]
    routers = [D.DistRouter(g, p["alpha"][s], p["beta"], p["dx"][s], dt,
                            alpha_floodplains=vals["ChannelAlpha2"][s] if split else None) for g, s in zip(graphs, sl)]
    steps = [D.DistRoutingStep(r, {k: (a[s] if isinstance(a, np.ndarray) else a) for k, a in vals.items()}, split,
                               p["beta"], 1 / dt, dt * nsteps) for r, s in zip(routers, sl)]
    return steps, nph


def _same(steps, ref, split, what):
    from lisflood_amd.routing import _OUT, _STATE
    for k in _STATE + _OUT:
        if not split and k in ("Chan2QKin", "Chan2M3Kin", "CrossSection2Area", "Sideflow1Chan"):
            continue
        got = np.concatenate([st.download(k) for st in steps])
        assert np.array_equal(got, ref.download(k), equal_nan=True), (what, k)


@pytest.mark.parametrize("split", [True, False])
@pytest.mark.parametrize("family,seed,nblocks", [("deep", 2, 3), ("shallow", 1, 4), ("saddle", 6, 3), ("river", 7, 5)])
def test_fused_model_step_on_row_blocks_small(amd, family, seed, nblocks, split):
    """150 x 130, 7 sub-steps, two model steps in a row (the second starts from the first one's state); non-channel
    pixels in the domain; flow crossing the cuts both ways (saddle)"""
    from lisflood_amd import dist as D
    nsteps = 7
    codes, p, vals, dt = _case(family, 150, 130, seed)
    kw, ref = _whole(codes, p, vals, dt, split, nsteps)
    steps, nph = _blocks(codes, p, vals, dt, split, nsteps, nblocks)
    for rep in range(2):
        ref.run_fused(nsteps)
        D.loopback_substeps_fused(steps, nsteps, lanes=(rep == 1))      # (second model step: the blocks of a phase on lanes)
        _same(steps, ref, split, (family, rep))
    assert nph >= 2
    # one block holding everything: the composite entry point, no communicator needed
    one, _ = _blocks(codes, p, vals, dt, split, nsteps, 1)
    one[0].substeps_fused(nsteps)
    one[0].substeps_fused(nsteps)
    _same(one, ref, split, (family, "one block"))
    for st in steps + one:
        st.free()
    ref.free()
    kw.close()


@pytest.mark.parametrize("family,seed", [("deep", 2), ("river", 7)])
def test_fused_model_step_on_row_blocks_config4_shape(amd, family, seed):
    """BASELINE.json configs[4]'s shape on one GPU: 2000 x 2000, 8 row blocks, NoRoutSteps = 24, split routing -- state
    after the model step bit-identical to the single-domain wavefront; the launch count shows the fused path ran
    (per phase: levels + 23 launches, not 24 x 2 x levels)"""
    from lisflood_amd import dist as D
    nsteps, nblocks = 24, 8
    codes, p, vals, dt = _case(family, 2000, 2000, seed, channel_frac=1.0)
    kw, ref = _whole(codes, p, vals, dt, True, nsteps)
    steps, nph = _blocks(codes, p, vals, dt, True, nsteps, nblocks)
    ref.run_fused(nsteps)
    D.loopback_substeps_fused(steps, nsteps)
    _same(steps, ref, True, family)
    for st in steps:
        g = st.router.graph
        assert st.router.last_launches() <= g.num_launch_units + nph * (nsteps - 1), (st.router.last_launches(),
                                                                                      g.num_launch_units, nph)
    for st in steps:
        st.free()
    ref.free()
    kw.close()


@pytest.mark.parametrize("family,seed,nblocks", [("deep", 2, 3), ("shallow", 1, 4), ("saddle", 6, 3), ("river", 7, 4)])
def test_router_call_in_the_order_the_two_streams_allow(amd, oracle, family, seed, nblocks):
    """lf_dist_router_route runs a phase's boundary-critical part, hands the round's halo to a second stream and sweeps
    the bulk part beside it: the same order on one GPU (packs taken right after part 0, ghost slots filled after part 1)
    gives the single-domain oracle's discharge, three calls"""
    from lisflood_amd import dist as D
    from lisflood_amd import synthetic as syn
    H, W = 300, 260
    N = H * W
    codes = syn.make_ldd(family, H, W, seed)
    mask = np.ones((H, W), bool)
    p = syn.router_params(N, seed=6)
    cpu = oracle.kinematicWave(codes.reshape(-1).astype(np.float64), mask, p["alpha"], p["beta"], p["dx"], p["dt"])
    blocks = D.row_blocks(H, nblocks)
    graphs = [D.DistGraph(codes[r0:r1], None, codes[r0 - 1] if r0 > 0 else None, None,
                          codes[r1] if r1 < H else None, None) for (r0, r1) in blocks]
    D.settle_phases_local(graphs)
    sl = [slice(r0 * W, r1 * W) for (r0, r1) in blocks]
    routers = [D.DistRouter(g, p["alpha"][s], p["beta"], p["dx"][s], p["dt"]) for g, s in zip(graphs, sl)]
    Qs = [r.new_state(p["Q0"][s]) for r, s in zip(routers, sl)]
    Qs2 = [r.new_state(p["Q0"][s]) for r, s in zip(routers, sl)]
    Qc = p["Q0"].copy()
    for step in range(3):
        q = syn.lateral_inflow(N, step)
        lats = [r.new_state(q[s]) for r, s in zip(routers, sl)]
        D.loopback_route(routers, Qs, lats, overlap_order=True)
        D.loopback_route(routers, Qs2, lats)
        cpu.kinematicWaveRouting(Qc, q)
        got = np.concatenate([r.download_pix(Q) for r, Q in zip(routers, Qs)])
        got2 = np.concatenate([r.download_pix(Q) for r, Q in zip(routers, Qs2)])
        assert np.array_equal(got, got2)
        np.testing.assert_allclose(got, Qc, rtol=1e-9, atol=1e-12)
        for d in lats:
            d.free()


@pytest.mark.parametrize("late", [False, True])
@pytest.mark.parametrize("family,seed,nblocks,ncalls", [("shallow", 1, 4, 5), ("saddle", 6, 3, 4), ("river", 7, 4, 3),
                                                        ("deep", 2, 3, 2)])
def test_pipelined_router_calls_on_alternating_state_vectors(amd, family, seed, nblocks, ncalls, late):
    """lf_dist_router_route_many starts phase 0 of call s + 1 before the later phases of call s by alternating between two
    state vectors.  Its kernel order on one GPU, with every halo round landing at the earliest or at the latest moment the
    real exchange may (late), gives bit for bit what call-by-call routing gives; with one block the C function itself
    (no communicator: call by call) agrees too."""
    from lisflood_amd import dist as D
    from lisflood_amd import synthetic as syn
    if os.environ.get("LF_GENERAL_POW") == "1":
        pytest.skip("the pipelined form reads the old discharge from a second vector: the beta = 3/5 path only "
                    "(lf_dist_router_route_many routes call by call otherwise)")
    H, W = 300, 260
    N = H * W
    codes = syn.make_ldd(family, H, W, seed)
    p = syn.router_params(N, seed=6)
    blocks = D.row_blocks(H, nblocks)
    graphs = [D.DistGraph(codes[r0:r1], None, codes[r0 - 1] if r0 > 0 else None, None,
                          codes[r1] if r1 < H else None, None) for (r0, r1) in blocks]
    D.settle_phases_local(graphs)
    sl = [slice(r0 * W, r1 * W) for (r0, r1) in blocks]
    routers = [D.DistRouter(g, p["alpha"][s], p["beta"], p["dx"][s], p["dt"]) for g, s in zip(graphs, sl)]
    Qa = [r.new_state(p["Q0"][s]) for r, s in zip(routers, sl)]
    Qb = [r.new_state(p["Q0"][s]) for r, s in zip(routers, sl)]
    lats = [[r.new_state(syn.lateral_inflow(N, c)[s]) for c in range(ncalls)] for r, s in zip(routers, sl)]
    for c in range(ncalls):
        D.loopback_route(routers, Qa, [l[c] for l in lats])
    D.loopback_route_many(routers, Qb, lats, late_halo=late)
    for r, a, b in zip(routers, Qa, Qb):
        assert np.array_equal(r.download_pix(a), r.download_pix(b))
    for l in lats:
        for d in l:
            d.free()


@pytest.mark.parametrize("family,seed,nblocks", [("deep", 2, 3), ("saddle", 6, 4), ("shallow", 1, 2)])
def test_several_model_steps_per_call_on_row_blocks(amd, family, seed, nblocks):
    """lf_dist_fused_phase_model_steps / lf_dist_routing_model_steps_fused: three model steps of 8 split-routing sub-steps,
    each with its own sideflow vector, in ONE pass over the phases (every phase runs the sub-steps of all three as one
    wavefront, its halo block carries the slabs of all three) against lf_routing_model_steps_fused on the whole raster --
    which the single-domain tests hold bit-identical to model step after model step: every state vector and every model
    step's discharge sum bit for bit."""
    from lisflood_amd import dist as D
    from lisflood_amd import synthetic as syn
    from lisflood_amd._lib import DeviceArray
    nsteps, M = 8, 3
    H, W = 150, 130
    N = H * W
    codes, p, vals, dt = _case(family, H, W, seed)
    kw, ref = _whole(codes, p, vals, dt, True, nsteps)
    steps, nph = _blocks(codes, p, vals, dt, True, nsteps, nblocks)
    sides = [syn.lateral_inflow(N, 60 + m) * p["dx"] * dt for m in range(M)]
    want = ref.run_model_steps(nsteps, sides)
    blocks = D.row_blocks(H, nblocks)
    sums, sfl = [], []
    for st, (r0, r1) in zip(steps, blocks):
        sl = slice(r0 * W, r1 * W)
        sums.append(DeviceArray((M, st.N)).zero())
        sfl.append(DeviceArray.from_host(np.ascontiguousarray(np.stack([s[sl][st.perm] for s in sides]))))
    D.loopback_model_steps_fused(steps, nsteps, M, sums, sfl)
    from lisflood_amd.routing import _OUT, _STATE
    for k in [x for x in _STATE + _OUT if x != "sumDisDay"]:
        got = np.concatenate([st.download(k) for st in steps])
        assert np.array_equal(got, ref.download(k), equal_nan=True), (family, k)
    for m in range(M):
        got = []
        for st, sm in zip(steps, sums):
            row = np.empty(st.N)
            row[st.perm] = sm.download()[m]
            got.append(row)
        assert np.array_equal(np.concatenate(got), want[m]), (family, m)
    # one block holding everything: the composite entry point (no communicator needed)
    one, _ = _blocks(codes, p, vals, dt, True, nsteps, 1)
    kw2, ref2 = _whole(codes, p, vals, dt, True, nsteps)
    want2 = ref2.run_model_steps(nsteps, sides)
    s1 = DeviceArray((M, N)).zero()
    f1 = DeviceArray.from_host(np.ascontiguousarray(np.stack([s[one[0].perm] for s in sides])))
    one[0].model_steps_fused(nsteps, M, s1, f1)
    got = np.empty((M, N))
    got[:, one[0].perm] = s1.download()
    assert np.array_equal(got, want2)
    for a in sums + sfl + [s1, f1]:
        a.free()
    for st in steps + one:
        st.free()
    ref.free(); ref2.free(); kw.close(); kw2.close()
