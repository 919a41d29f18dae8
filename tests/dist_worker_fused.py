"""Worker of tests/test_dist_multirank_gpu.py: one process per rank, a model step of NoRoutSteps split-routing sub-steps on
the rank's row block (lf_dist_routing_substeps_fused: one halo block per phase, neighbour and section through the
communicator), twice, then three more model steps in ONE call (lf_dist_routing_model_steps_fused); rank 0 runs the same
model steps on the whole raster (lf_routing_substeps_fused, lf_routing_model_steps_fused) and compares bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lisflood-code_amd"), ROOT):
    sys.path.insert(0, p)

from lisflood_amd import _lib, dist as D, synthetic as syn  # noqa: E402


def main():
    T = D.SocketTransport.from_env(timeout=180.0)
    rank, world = T.rank, T.nranks
    device = rank % max(_lib.device_count(), 1)
    family = os.environ.get("LF_TEST_FAMILY", "saddle")
    H, W = (int(x) for x in os.environ.get("LF_TEST_SHAPE", "240x200").split("x"))
    split = os.environ.get("LF_TEST_SPLIT", "1") == "1"
    nsteps = int(os.environ.get("LF_TEST_NSTEPS", "7"))
    seed = {"shallow": 1, "deep": 2, "saddle": 6, "river": 7}[family]
    N = H * W
    codes = syn.make_ldd(family, H, W, seed)
    p = syn.router_params(N, seed=6)
    vals, dt = syn.model_step_values(N, p, seed=19)
    vals["IsChannelKinematic"] = np.random.default_rng(3).random(N) < 0.9
    r0, r1 = D.row_blocks(H, world)[rank]
    g = D.DistGraph(codes[r0:r1], None, codes[r0 - 1] if r0 > 0 else None, None, codes[r1] if r1 < H else None, None)
    D.settle_phases(g, T)
    comm = D.Comm(T.broadcast(D.Comm.unique_id() if rank == 0 else None), world, rank, device)
    s = slice(r0 * W, r1 * W)
    router = D.DistRouter(g, p["alpha"][s], p["beta"], p["dx"][s], dt, alpha_floodplains=vals["ChannelAlpha2"][s] if split else None,
                          device=device, comm=comm, rank_top=rank - 1 if rank > 0 else -1,
                          rank_bottom=rank + 1 if rank + 1 < world else -1)
    st = D.DistRoutingStep(router, {k: (a[s] if isinstance(a, np.ndarray) else a) for k, a in vals.items()}, split, p["beta"],
                           1 / dt, dt * nsteps)
    from lisflood_amd.routing import _OUT, _STATE
    names = [k for k in _STATE + _OUT
             if split or k not in ("Chan2QKin", "Chan2M3Kin", "CrossSection2Area", "Sideflow1Chan")]
    outs = []
    for rep in range(2):
        st.substeps_fused(nsteps)
        _lib.synchronize(device)
        outs.append([st.download(k) for k in names])          # (the transport carries lists, not dicts)
    # several model steps per call (lf_dist_routing_model_steps_fused): the halo block of a phase carries the slabs of all of
    # them; the resident sideflow vector in every model step, the per-model-step discharge sums in [M, N]
    M = int(os.environ.get("LF_TEST_MODEL_STEPS", "3"))
    sums = _lib.DeviceArray((M, max(st.N, 1)), device=device).zero()
    st.model_steps_fused(nsteps, M, sums)
    _lib.synchronize(device)
    per_step = np.empty((M, st.N))
    per_step[:, st.perm] = sums.download()[:, :st.N]
    outs.append([st.download(k) for k in names] + [per_step])
    sums.free()
    gathered = T.allgather(outs)
    if rank == 0:
        from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave
        from bench_support import RoutingStepDevice
        kw = kinematicWave(None, None, p["alpha"], p["beta"], p["dx"], dt,
                           alpha_floodplains=vals["ChannelAlpha2"] if split else None, graph=Graph(ldd_raster=codes))
        ref = RoutingStepDevice(kw, vals, split, p["beta"], 1 / dt, dt * nsteps)
        for rep in range(2):
            ref.run_fused(nsteps)
            for i, k in enumerate(names):
                got = np.concatenate([gathered[r][rep][i] for r in range(world)])
                assert np.array_equal(got, ref.download(k), equal_nan=True), (family, rep, k)
        want = ref.run_model_steps(nsteps, nmodel=M)
        for i, k in enumerate(names):
            if k == "sumDisDay":
                continue
            got = np.concatenate([gathered[r][2][i] for r in range(world)])
            assert np.array_equal(got, ref.download(k), equal_nan=True), (family, "model steps", k)
        got = np.concatenate([gathered[r][2][len(names)] for r in range(world)], axis=1)
        assert np.array_equal(got, want), (family, "model step sums")
        ref.free()
        kw.close()
        print("DIST_FUSED_OK phases=%d ranks=%d" % (g.num_phases, world))
    T.barrier()
    st.free()
    router.close()
    comm.close()
    T.close()


if __name__ == "__main__":
    main()
