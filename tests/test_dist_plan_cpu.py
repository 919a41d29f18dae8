"""Row-block partition (multi-GPU path), CPU side: the partition plan produced by the product's host code is
executed with the oracle's per-cell arithmetic and must reproduce the single-domain oracle BIT FOR BIT --
same solves, same upstream summation order -- for any number of ranks, on synthetic and real LDDs.
Includes a true 2-process run over torch.distributed (gloo)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, golden

from lisflood_amd import dist as D
from lisflood_amd import synthetic as syn

import dist_plan_exec as X


def global_reference(oracle, codes, mask, alpha, dx, dt, beta, Q0, qs):
    kw = oracle.kinematicWave(codes[mask].astype(np.float64), mask, alpha, beta, dx, dt)
    Q = Q0.copy()
    out = []
    for q in qs:
        kw.kinematicWaveRouting(Q, q)
        out.append(Q.copy())
    return out


def run_partitioned(codes, mask, nranks, alpha, dx, dt, beta, Q0, qs):
    H, W = codes.shape
    blocks, graphs = X.build_blocks(codes, mask, nranks)
    nph = D.settle_phases_local(graphs)
    ids = np.full((H, W), -1, np.int64)
    ids[mask] = np.arange(int(mask.sum()))
    sel = [ids[r0:r1][mask[r0:r1]] for (r0, r1) in blocks]     # global pixel ids of each rank's local pixels
    ranks = [X.RankState(g, alpha[s], dx[s] if np.ndim(dx) else dx, dt, beta, Q0[s]) for g, s in zip(graphs, sel)]
    results = []
    for q in qs:
        for rk, s in zip(ranks, sel):
            rk.begin_call(q[s])
        for j in range(nph):
            for rk in ranks:
                rk.compute_phase(j)
            if j + 1 < nph:
                for k, rk in enumerate(ranks):
                    if k > 0:
                        rk.recv_values(j, 0, ranks[k - 1].send_values(j, 1))
                    if k + 1 < nranks:
                        rk.recv_values(j, 1, ranks[k + 1].send_values(j, 0))
        full = np.empty(int(mask.sum()))
        for rk, s in zip(ranks, sel):
            full[s] = rk.pixel_values()
        results.append(full)
    return results, nph, graphs


CASES = [("shallow", 1, 61, 47), ("deep", 2, 64, 50), ("saddle", 6, 50, 44)]


@pytest.mark.parametrize("family,seed,H,W", CASES)
@pytest.mark.parametrize("nranks", [1, 2, 3, 8])
def test_partition_is_bit_identical_to_single_domain(oracle, family, seed, H, W, nranks):
    codes = syn.make_ldd(family, H, W, seed)
    mask = np.ones((H, W), bool)
    N = H * W
    p = syn.router_params(N, seed=5)
    qs = [syn.lateral_inflow(N, s) for s in range(3)]
    ref = global_reference(oracle, codes, mask, p["alpha"], p["dx"], p["dt"], p["beta"], p["Q0"], qs)
    got, nph, graphs = run_partitioned(codes, mask, nranks, p["alpha"], p["dx"], p["dt"], p["beta"], p["Q0"], qs)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)
    assert nph >= 1 and (nranks == 1) == (nph == 1 and all(sum(g.n_ghost) == 0 for g in graphs))
    if family == "deep" and nranks > 1:
        assert nph >= nranks        # sheet flow crosses every boundary on its way down


def test_partition_real_ldd_with_mask(oracle):
    g = golden("route_etrs89")
    mask = g["mask"]
    codes = np.zeros(mask.shape, np.uint8)
    codes[mask] = g["codes"].astype(np.uint8)
    qs = [g["q"][s] for s in range(4)]
    ref = [g["Q"][s] for s in range(4)]      # reference-captured vectors (main channel)
    for nranks in (2, 5):
        got, nph, _ = run_partitioned(codes, mask, nranks, g["alpha"], g["dx"], float(g["dt"]), float(g["beta"]),
                                      g["Q0"], qs)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b)


def test_plan_invariants():
    codes = syn.make_ldd("shallow", 40, 33, 3)
    mask = np.ones((40, 33), bool); mask[10:14, 5:20] = False; mask[19:21, :] &= (np.arange(33) % 3 != 0)
    codes[~mask] = 0
    blocks, graphs = X.build_blocks(codes, mask, 4)
    nph = D.settle_phases_local(graphs)
    for k, g in enumerate(graphs):
        perm, ph = g.layout()
        ups_ptr, ups_idx = g.csr()
        n = g.num_pixels
        assert sorted(perm.tolist()) == list(range(n)) and (np.diff(ph) >= 0).all()
        for p in range(n):                       # every dependency is earlier in the sweep, or a ghost of an earlier round
            for e in ups_idx[ups_ptr[p]:ups_ptr[p + 1]]:
                assert e < p or e >= n
        # messages match pairwise
        for j in range(nph):
            c = g.round_counts(j)
            if k > 0:
                assert c["send"][0] == graphs[k - 1].round_counts(j)["recv"][1]
                assert c["recv"][0] == graphs[k - 1].round_counts(j)["send"][1]
            else:
                assert c["send"][0] == 0 and c["recv"][0] == 0
        assert g.round_counts(nph - 1)["send"] == (0, 0) or nph == 1


def test_two_process_gloo_run():
    """2 ranks over torch.distributed/gloo: set-up fixpoint and halo values really cross a process boundary."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29531", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29531", os.path.join(ROOT, "tests", "dist_worker_gloo.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "DIST_GLOO_OK" in r.stdout


def test_three_process_socket_run():
    """3 ranks, no PyTorch: rendezvous, fixpoint and halo values over lisflood_amd.dist.SocketTransport (what bench.py's
    N > 1 leg uses for everything but the RCCL data path); result bit-identical to the single-domain oracle."""
    procs = []
    for rank in range(3):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="3", LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29547", TORCHELASTIC_RUN_ID="pytest%d" % os.getpid(), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker_socket.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, o[-2000:] + e[-2000:]
    assert "DIST_SOCKET_OK" in outs[0][0] and "ranks=3" in outs[0][0]


def test_bench_dist_leg_is_torch_free():
    """the N > 1 bench leg and the transport it uses import no PyTorch (north_star: ctypes host code, no PyTorch)"""
    import ast
    for name in ("dist_bench.py", "dist.py"):
        tree = ast.parse(open(os.path.join(ROOT, "lisflood-code_amd", "lisflood_amd", name)).read())
        top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
        mods = [a.name for n in top if isinstance(n, ast.Import) for a in n.names] + \
               [n.module or "" for n in top if isinstance(n, ast.ImportFrom)]
        assert not any(m.split(".")[0] == "torch" for m in mods), (name, mods)
    src = open(os.path.join(ROOT, "lisflood-code_amd", "lisflood_amd", "dist_bench.py")).read()
    assert "import torch" not in src
