"""Test-only executor of the row-block partition plan (lisflood_amd.dist / csrc/lf_dist.hip) on the CPU:
the plan (phases, sweep order, index lists, send positions, ghost slots) comes from the product's host
code, the per-cell arithmetic from the oracle.  Used by the single-process and the gloo tests."""
import math

import numpy as np

import oracle
from lisflood_amd import dist as D


def build_blocks(codes, mask, nranks):
    """codes/mask [H, W]; returns (blocks [(r0, r1)], graphs [DistGraph]) -- phases not settled yet."""
    H, W = codes.shape
    blocks = D.row_blocks(H, nranks)
    graphs = []
    for (r0, r1) in blocks:
        top = (codes[r0 - 1], mask[r0 - 1]) if r0 > 0 else (None, None)
        bot = (codes[r1], mask[r1]) if r1 < H else (None, None)
        graphs.append(D.DistGraph(codes[r0:r1], mask[r0:r1], top[0], top[1], bot[0], bot[1]))
    return blocks, graphs


class RankState:
    """Host mirror of lf_dist_router: parameters in engine order, state vector with ghost slots."""

    def __init__(self, graph, alpha, dx, dt, beta, Q0):
        self.g = graph
        self.beta = beta
        self.perm, _ = graph.layout()
        self.ups_ptr, self.ups_idx = graph.csr()
        self.a = (alpha * dx / dt)[self.perm]
        self.ba = beta * self.a
        self.dx = np.broadcast_to(dx, alpha.shape)[self.perm]
        self.state = np.zeros(graph.state_size)
        self.state[:graph.num_pixels] = Q0[self.perm]
        self.constant = np.zeros(graph.num_pixels)

    def begin_call(self, q_lat_pix):
        n = self.g.num_pixels
        lateral = q_lat_pix[self.perm] * self.dx
        # libm pow element by element: numpy's SIMD pow differs from libm (and from the oracle) in the last ulp
        qb = np.array([math.pow(x, self.beta) for x in self.state[:n]]) if n else np.zeros(0)
        self.constant = self.a * qb + lateral                             # kinematic_wave_parallel.py:175

    def compute_phase(self, j):
        b, e = self.g.phase_range(j)
        oracle.sweep_positions(self.state, self.constant, self.ups_ptr, self.ups_idx, self.a, self.ba, self.beta, b, e)

    def compute_part(self, j, part):
        b, e = self.g.part_range(j, part)
        oracle.sweep_positions(self.state, self.constant, self.ups_ptr, self.ups_idx, self.a, self.ba, self.beta, b, e)

    def compute_stage_cones(self, st, plan, reverse=False, max_cone=64):
        """Stage `st` (= 2 * phase + part) as k_sweep_cones_dist runs it on the block plan `plan` (DistGraph.route_plan) --
        or phase `st` on the fused path's plan (DistGraph.fused_plan, max_cone = 256: what k_fused_cones<DIST> may read
        of a sub-step's router outputs) --:
        block after block; the cones of a block in any order; inside a cone unit by unit, where ONLY the unit just solved
        is visible as new (the kernel's LDS row) -- every other cell of the block still shows its old value until the
        whole block is done (what the kernel may read from the state vector must be final before the launch).  A plan
        whose cones did not hold everything a unit needs from the unit above would read stale values here."""
        ls = plan["level_start"]
        for b in range(plan["stage_block"][st], plan["stage_block"][st + 1]):
            k0 = int(plan["level"][b])
            nl = int(plan["level"][b + 1]) - k0
            ncones = int(plan["row"][b + 1] - plan["row"][b]) - 1
            rows = plan["cone"][plan["off"][b]:plan["off"][b] + (ncones + 1) * nl].reshape(ncones + 1, nl)
            for j in range(nl):                                  # the cones tile every unit of the block
                assert rows[0, j] == ls[k0 + j] and rows[ncones, j] == ls[k0 + j + 1] and (np.diff(rows[:, j]) >= 0).all()
            if nl == 1:                                          # a single (wide) unit: the level kernel
                self._sweep(int(ls[k0]), int(ls[k0 + 1]))
                continue
            done = []
            for c in (range(ncones - 1, -1, -1) if reverse else range(ncones)):
                prev = None
                for j in range(nl):
                    lo, hi = int(rows[c, j]), int(rows[c + 1, j])
                    assert hi - lo <= max_cone
                    old = self.state[lo:hi].copy()
                    self._sweep(lo, hi)
                    done.append((lo, hi, self.state[lo:hi].copy()))
                    if prev is not None:                         # the unit above is not visible any longer
                        self.state[prev[0]:prev[1]] = prev[2]
                    prev = (lo, hi, old)
                self.state[prev[0]:prev[1]] = prev[2]
            for lo, hi, new in done:                             # the block's discharges: visible to the next launch
                self.state[lo:hi] = new

    def _sweep(self, b, e):
        oracle.sweep_positions(self.state, self.constant, self.ups_ptr, self.ups_idx, self.a, self.ba, self.beta, b, e)

    def send_values(self, j, side):
        return self.state[self.g.round_send_positions(j, side)].copy()

    def recv_values(self, j, side, values):
        slot = self.g.round_recv_slot(j, side)
        n = self.g.round_counts(j)["recv"][side]
        assert len(values) == n
        self.state[slot:slot + n] = values

    def pixel_values(self):
        out = np.empty(self.g.num_pixels)
        out[self.perm] = self.state[:self.g.num_pixels]
        return out
