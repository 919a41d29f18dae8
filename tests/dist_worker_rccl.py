"""Worker of tests/test_gpu_parity.py::test_two_rank_rccl_halo_exchange: one process per GPU, RCCL Send/Recv of the
boundary discharge (lf_dist_router_route), SocketTransport for set-up; rank 0 compares with the oracle."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lisflood-code_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import oracle  # noqa: E402
from lisflood_amd import _lib, dist as D, synthetic as syn  # noqa: E402


def main():
    T = D.SocketTransport.from_env(timeout=180.0)
    rank, world = T.rank, T.nranks
    device = rank % max(_lib.device_count(), 1)
    H, W = 400, 300
    codes = syn.make_ldd(os.environ.get("LF_TEST_FAMILY", "saddle"), H, W, 6)
    N = H * W
    p = syn.router_params(N, seed=9)
    r0, r1 = D.row_blocks(H, world)[rank]
    g = D.DistGraph(codes[r0:r1], None, codes[r0 - 1] if r0 > 0 else None, None, codes[r1] if r1 < H else None, None)
    D.settle_phases(g, T)
    comm = D.Comm(T.broadcast(D.Comm.unique_id() if rank == 0 else None), world, rank, device)
    sel = slice(r0 * W, r1 * W)
    router = D.DistRouter(g, p["alpha"][sel], p["beta"], p["dx"][sel], p["dt"], device=device, comm=comm,
                          rank_top=rank - 1 if rank > 0 else -1, rank_bottom=rank + 1 if rank + 1 < world else -1)
    Q = router.new_state(p["Q0"][sel])
    # four calls one by one and four as one pipelined sequence (lf_dist_router_route_many: alternating state vectors,
    # halo rounds on the second stream beside the next call's phase 0); LF_TEST_MANY_FIRST=1: the pipelined four first
    # (the halo stream and its events are then created by route_many, and the plain calls must find them)
    many_first = os.environ.get("LF_TEST_MANY_FIRST", "0") == "1"
    outs = []

    def plain(steps):
        for s in steps:
            lat = router.new_state(syn.lateral_inflow(N, s)[sel])
            router.route(Q, lat)
            _lib.synchronize(device)
            lat.free()
            outs.append((s, router.download_pix(Q)))

    def pipelined(steps):
        lats = [router.new_state(syn.lateral_inflow(N, s)[sel]) for s in steps]
        router.route_many(Q, lats)
        _lib.synchronize(device)
        outs.append((steps[-1], router.download_pix(Q)))
        for d in lats:
            d.free()

    if many_first:
        pipelined(range(0, 4))
        plain(range(4, 8))
    else:
        plain(range(0, 4))
        pipelined(range(4, 8))
    gathered = T.allgather(outs)
    if rank == 0:
        kw = oracle.kinematicWave(codes.reshape(-1).astype(np.float64), np.ones((H, W), bool), p["alpha"], p["beta"], p["dx"],
                                  p["dt"])
        Qo = p["Q0"].copy()
        want = {}
        for s in range(8):
            kw.kinematicWaveRouting(Qo, syn.lateral_inflow(N, s))
            want[s] = Qo.copy()
        for i, (s, _) in enumerate(outs):
            full = np.concatenate([gathered[k][i][1] for k in range(world)])
            np.testing.assert_allclose(full, want[s], rtol=1e-9, atol=1e-12, err_msg="after call %d" % s)
        print("DIST_RCCL_OK phases=%d ranks=%d" % (g.num_phases, world))
    T.barrier()
    comm.close()
    T.close()


if __name__ == "__main__":
    main()
