"""Worker of tests/test_dist_plan_cpu.py::test_two_process_gloo_run (launched by torch.distributed.run)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lisflood-code_amd"), os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

import oracle  # noqa: E402
from lisflood_amd import dist as D  # noqa: E402
from lisflood_amd import synthetic as syn  # noqa: E402
import dist_plan_exec as X  # noqa: E402


class TorchTransport:
    """torch.distributed transport for the set-up fixpoint (vertical neighbours = rank -/+ 1): the two methods
    lisflood_amd.dist.settle_phases needs.  Test infrastructure -- the product's own transport is SocketTransport."""

    def __init__(self, dist):
        self.dist = dist
        self.rank, self.nranks = dist.get_rank(), dist.get_world_size()

    def exchange_int32(self, top_send, bottom_send, n_top_recv, n_bottom_recv):
        dist = self.dist
        top_recv = torch.zeros(n_top_recv, dtype=torch.int32)
        bot_recv = torch.zeros(n_bottom_recv, dtype=torch.int32)
        ops = []
        up, dn = self.rank - 1, self.rank + 1
        if up >= 0:
            if len(top_send):
                ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(top_send)), up))
            if n_top_recv:
                ops.append(dist.P2POp(dist.irecv, top_recv, up))
        if dn < self.nranks:
            if len(bottom_send):
                ops.append(dist.P2POp(dist.isend, torch.from_numpy(np.ascontiguousarray(bottom_send)), dn))
            if n_bottom_recv:
                ops.append(dist.P2POp(dist.irecv, bot_recv, dn))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return top_recv.numpy(), bot_recv.numpy()

    def allreduce_max(self, value):
        t = torch.tensor([int(value)], dtype=torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.item())


def swap(dist, rank, world, j, rk):
    """halo values of round j with the vertical neighbours (gloo send/recv of float64)."""
    ops, bufs = [], {}
    for side, peer in ((0, rank - 1), (1, rank + 1)):
        if peer < 0 or peer >= world:
            continue
        c = rk.g.round_counts(j)
        if c["send"][side]:
            ops.append(dist.P2POp(dist.isend, torch.from_numpy(rk.send_values(j, side)), peer))
        if c["recv"][side]:
            bufs[side] = torch.zeros(c["recv"][side], dtype=torch.float64)
            ops.append(dist.P2POp(dist.irecv, bufs[side], peer))
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    for side, b in bufs.items():
        rk.recv_values(j, side, b.numpy())


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    H, W = 70, 40
    codes = syn.make_ldd("saddle", H, W, 6)      # flow crosses the row cut in both directions
    mask = np.ones((H, W), bool)
    N = H * W
    p = syn.router_params(N, seed=9)
    r0, r1 = D.row_blocks(H, world)[rank]
    g = D.DistGraph(codes[r0:r1], mask[r0:r1], codes[r0 - 1] if r0 > 0 else None, None,
                    codes[r1] if r1 < H else None, None)
    D.settle_phases(g, TorchTransport(dist))
    sel = np.arange(r0 * W, r1 * W)
    rk = X.RankState(g, p["alpha"][sel], p["dx"][sel], p["dt"], p["beta"], p["Q0"][sel])
    outs = []
    for s in range(3):
        q = syn.lateral_inflow(N, s)
        rk.begin_call(q[sel])
        for j in range(g.num_phases):
            rk.compute_phase(j)
            if j + 1 < g.num_phases:
                swap(dist, rank, world, j, rk)
        outs.append(rk.pixel_values())
    gathered = [None] * world
    dist.all_gather_object(gathered, outs)
    if rank == 0:
        kw = oracle.kinematicWave(codes.reshape(-1).astype(np.float64), mask, p["alpha"], p["beta"], p["dx"], p["dt"])
        Q = p["Q0"].copy()
        for s in range(3):
            kw.kinematicWaveRouting(Q, syn.lateral_inflow(N, s))
            full = np.concatenate([gathered[k][s] for k in range(world)])
            assert np.array_equal(full, Q), "step %d differs" % s
        print("DIST_GLOO_OK phases=%d" % g.num_phases)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
