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


def run_partitioned(codes, mask, nranks, alpha, dx, dt, beta, Q0, qs, cones=False):
    """cones: every stage through the block plan of single router calls (RankState.compute_stage_cones), the cones of a
    block in alternating order"""
    H, W = codes.shape
    blocks, graphs = X.build_blocks(codes, mask, nranks)
    nph = D.settle_phases_local(graphs)
    plans = [g.route_plan() if cones else None for g in graphs]
    ids = np.full((H, W), -1, np.int64)
    ids[mask] = np.arange(int(mask.sum()))
    sel = [ids[r0:r1][mask[r0:r1]] for (r0, r1) in blocks]     # global pixel ids of each rank's local pixels
    ranks = [X.RankState(g, alpha[s], dx[s] if np.ndim(dx) else dx, dt, beta, Q0[s]) for g, s in zip(graphs, sel)]
    results = []
    for q in qs:
        for rk, s in zip(ranks, sel):
            rk.begin_call(q[s])
        for j in range(nph):
            for rk, pl in zip(ranks, plans):
                if pl is None:
                    rk.compute_phase(j)
                else:
                    for part in (0, 1):
                        rk.compute_stage_cones(2 * j + part, pl, reverse=bool((j + part + len(results)) % 2))
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
        for cones in (False, True):          # (True: the block plan of single router calls on the real, masked LDD)
            got, nph, graphs = run_partitioned(codes, mask, nranks, g["alpha"], g["dx"], float(g["dt"]), float(g["beta"]),
                                               g["Q0"], qs, cones=cones)
            for a, b in zip(got, ref):
                assert np.array_equal(a, b)
            assert not cones or any(gr.route_plan() is not None for gr in graphs)


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_block_plan_on_random_masked_rasters(oracle, seed):
    """random family / shape / rank count / land mask with holes and missing rows at the cuts: the block plan of single
    router calls executed under the kernel's visibility rules equals the single domain bit for bit"""
    rng = np.random.default_rng(100 + seed)
    family = ["shallow", "deep", "saddle", "river"][seed % 4]
    H, W = int(rng.integers(40, 110)), int(rng.integers(30, 90))
    nranks = int(rng.integers(2, 7))
    codes = syn.make_ldd(family, H, W, seed)
    mask = rng.random((H, W)) < 0.9
    r0 = int(rng.integers(3, H - 6))
    mask[r0:r0 + 2, int(rng.integers(0, W // 2)):int(rng.integers(W // 2, W))] = False       # a gap that may sit on a cut
    n = int(mask.sum())
    p = syn.router_params(n, seed=seed)
    qs = [syn.lateral_inflow(n, s) for s in range(2)]
    ref = global_reference(oracle, codes, mask, p["alpha"], p["dx"], p["dt"], p["beta"], p["Q0"], qs)
    got, nph, graphs = run_partitioned(codes, mask, nranks, p["alpha"], p["dx"], p["dt"], p["beta"], p["Q0"], qs, cones=True)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b), (family, H, W, nranks)


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
    for name in (os.path.join(ROOT, "bench_dist.py"), os.path.join(ROOT, "lisflood-code_amd", "lisflood_amd", "dist.py")):
        tree = ast.parse(open(name).read())
        top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom))]
        mods = [a.name for n in top if isinstance(n, ast.Import) for a in n.names] + \
               [n.module or "" for n in top if isinstance(n, ast.ImportFrom)]
        assert not any(m.split(".")[0] == "torch" for m in mods), (name, mods)
    src = open(os.path.join(ROOT, "bench_dist.py")).read()
    assert "import torch" not in src


def test_socket_transport_wire_format_is_typed_not_pickled():
    """messages are a fixed typed encoding: round trip of everything the ranks send, and refusal of anything else
    (object arrays, unknown tags, truncated input) -- nothing a peer sends is ever unpickled"""
    from lisflood_amd import dist as D
    msg = [None, True, 7, -3.5, b"\x00id", "txt", (np.arange(5, dtype=np.int32), np.zeros((2, 3))), [np.float64(2.0)]]
    parts = []
    D._encode(msg, parts)
    back, end = D._decode(b"".join(parts))
    assert end == len(b"".join(parts))
    assert back[:6] == [None, True, 7, -3.5, b"\x00id", "txt"]
    assert np.array_equal(back[6][0], msg[6][0]) and back[6][0].dtype == np.int32 and back[6][1].shape == (2, 3)
    assert back[7] == [2.0]
    with pytest.raises(TypeError):
        D._encode(np.array([object()]), [])
    with pytest.raises(TypeError):
        D._encode({"a": 1}, [])
    import pickle
    for bad in (pickle.dumps([1, 2]), b"a\x03|O8\x01" + b"\x01" + b"\x00" * 7, b"b" + b"\xff" * 8, b"l" + b"\xff" * 8):
        with pytest.raises((ValueError, struct_error())):
            D._decode(bad)
    assert "pickle" not in open(D.__file__).read().replace("no pickle", "").replace("unpickled", "")


def struct_error():
    import struct
    return struct.error


def test_socket_transport_rejects_strangers_and_survives_a_stale_file(tmp_path):
    """rank 0 drops a connection that does not present the run's token (and still completes the rendezvous with the real
    rank); a rendezvous file left behind by a crashed run (dead port) does not break the next run"""
    import socket
    import threading
    from lisflood_amd import dist as D
    rdv = str(tmp_path / "rdv")
    dead = socket.socket()
    dead.bind(("127.0.0.1", 0))
    port = dead.getsockname()[1]
    dead.close()
    gone = os.fork()                                          # a pid that no longer exists: the crashed run's rank 0
    if gone == 0:
        os._exit(0)
    os.waitpid(gone, 0)
    open(rdv, "w").write("%d %s %d" % (port, "0" * 32, gone))   # stale: nobody listens there, its owner is gone
    res = {}

    def rank0():
        res[0] = D.SocketTransport(0, 2, rdv, timeout=30.0)

    t0 = threading.Thread(target=rank0)
    t0.start()
    import time
    deadline = time.time() + 20
    live = None
    while time.time() < deadline:                            # the file rank 0 publishes replaces the stale one
        try:
            ps, tok = open(rdv).read().split()[:2]
            if int(ps) != port:
                live = (int(ps), tok)
                break
        except (OSError, ValueError):
            pass
        time.sleep(0.02)
    assert live is not None
    s = socket.create_connection(("127.0.0.1", live[0]))     # a stranger: right port, wrong token
    s.sendall(D.SocketTransport._MAGIC + b"f" * 32 + (1).to_bytes(4, "little") + (2).to_bytes(4, "little"))
    s.settimeout(10)
    assert s.recv(2) == b""                                  # dropped without an answer
    s.close()
    s = socket.create_connection(("127.0.0.1", live[0]))     # a rank of a job of another size: right token, wrong world
    s.sendall(D.SocketTransport._MAGIC + live[1].encode("ascii") + (1).to_bytes(4, "little") + (3).to_bytes(4, "little"))
    s.settimeout(10)
    assert s.recv(2) == b""
    s.close()
    idle = socket.create_connection(("127.0.0.1", live[0]))  # says nothing: costs the others one second, not the run
    # a second job's rank 0 on the same file while the first lives
    probe = ("import sys; sys.path.insert(0, %r); from lisflood_amd import dist as D\n"
             "try:\n    D.SocketTransport(0, 2, %r, timeout=5.0)\nexcept RuntimeError as e:\n    print('REFUSED', e)\n"
             % (os.path.join(ROOT, "lisflood-code_amd"), rdv))
    out = subprocess.run([sys.executable, "-c", probe], capture_output=True, text=True, timeout=60)
    assert "REFUSED" in out.stdout, out.stdout + out.stderr
    assert os.path.exists(rdv)                               # (and it did not clobber the live file)
    t1 = D.SocketTransport(1, 2, rdv, timeout=30.0)
    t0.join(30)
    assert 0 in res
    out = {}
    th = threading.Thread(target=lambda: out.setdefault(0, res[0].allgather(np.arange(3))))
    th.start()
    got = t1.allgather(np.arange(3) + 10)
    th.join(30)
    assert np.array_equal(got[0], np.arange(3)) and np.array_equal(out[0][1], np.arange(3) + 10)
    t1.close()
    res[0].close()
    idle.close()
    assert not os.path.exists(rdv)


def run_partitioned_phase_major(codes, mask, nranks, alpha, dx, dt, beta, Q0, qs):
    """The schedule of lf_dist_routing_substeps_fused on the CPU, with a plain router call as the "sub-step": every rank
    sweeps ONE phase for ALL calls before the next phase starts; what crosses a phase or a rank boundary travels through
    the slabs [slot][call] (tables of lf_dist_graph_get_fused_tables), one halo block per phase and neighbour."""
    import math
    H, W = codes.shape
    S = len(qs)
    blocks, graphs = X.build_blocks(codes, mask, nranks)
    nph = D.settle_phases_local(graphs)
    ids = np.full((H, W), -1, np.int64)
    ids[mask] = np.arange(int(mask.sum()))
    sel = [ids[r0:r1][mask[r0:r1]] for (r0, r1) in blocks]
    ranks = []
    for g, s in zip(graphs, sel):
        n = g.num_pixels
        perm, _ = g.layout()
        ups_ptr, _idx = g.csr()
        out_slot, idx_f = g.fused_tables()
        lay = g.slab_layout()
        a = (alpha[s] * dx[s] / dt)[perm]
        rk = dict(g=g, n=n, perm=perm, ups_ptr=ups_ptr, out_slot=out_slot, lay=lay, a=a, ba=beta * a, dx=dx[s][perm],
                  idx=np.where(idx_f >= 0, idx_f, n + (-idx_f - 1)).astype(np.int32),
                  # per call: [router outputs of the n local cells | slab column of that call]
                  state=[np.zeros(n + lay["slots"]) for _ in range(S)], Q0=Q0[s][perm],
                  lat=[q[s][perm] for q in qs])
        assert (out_slot < lay["slots"]).all() and ((idx_f >= 0) | (-idx_f - 1 < lay["slots"])).all()
        ranks.append(rk)
    for j in range(nph):
        for rk in ranks:
            b, e = rk["g"].phase_range(j)
            for s in range(S):
                prev = rk["Q0"] if s == 0 else rk["state"][s - 1][:rk["n"]]
                const = np.zeros(rk["n"])
                const[b:e] = rk["a"][b:e] * np.array([math.pow(x, beta) for x in prev[b:e]]) + rk["lat"][s][b:e] * rk["dx"][b:e]
                st = rk["state"][s]
                oracle_mod().sweep_positions(st, const, rk["ups_ptr"], rk["idx"], rk["a"], rk["ba"], beta, b, e)
                has = np.nonzero(rk["out_slot"][b:e] >= 0)[0] + b
                st[rk["n"] + rk["out_slot"][has]] = st[has]
        if j + 1 == nph:
            break
        for k, rk in enumerate(ranks):          # halo of round j: exports of the neighbour -> my ghost slots, every call
            for side, src, src_side in ((0, k - 1, 1), (1, k + 1, 0)):
                if not (0 <= src < nranks):
                    continue
                cnt = rk["g"].round_counts(j)["recv"][side]
                if cnt == 0:
                    continue
                other = ranks[src]
                assert other["g"].round_counts(j)["send"][src_side] == cnt
                r_off = rk["lay"]["ghost"][side] + sum(rk["g"].round_counts(i)["recv"][side] for i in range(j))
                s_off = other["lay"]["export"][src_side] + sum(other["g"].round_counts(i)["send"][src_side] for i in range(j))
                for s in range(S):
                    rk["state"][s][rk["n"] + r_off:rk["n"] + r_off + cnt] = \
                        other["state"][s][other["n"] + s_off:other["n"] + s_off + cnt]
    results = []
    for s in range(S):
        full = np.empty(int(mask.sum()))
        for rk, ss in zip(ranks, sel):
            loc = np.empty(rk["n"])
            loc[rk["perm"]] = rk["state"][s][:rk["n"]]
            full[ss] = loc
        results.append(full)
    return results, nph


def oracle_mod():
    import oracle
    return oracle


@pytest.mark.parametrize("family,seed,H,W", CASES + [("river", 7, 60, 56)])
@pytest.mark.parametrize("nranks", [1, 3, 8])
def test_phase_major_slab_plan_is_bit_identical_to_single_domain(oracle, family, seed, H, W, nranks):
    """the fused sub-step schedule (phase-major, slabs, one halo block per phase) reproduces the call-by-call oracle"""
    codes = syn.make_ldd(family, H, W, seed)
    mask = np.ones((H, W), bool)
    if family == "shallow":
        mask[20:23, 10:30] = False
        codes[~mask] = 0
    N = int(mask.sum())
    p = syn.router_params(N, seed=5)
    qs = [syn.lateral_inflow(N, s) for s in range(4)]
    ref = global_reference(oracle, codes, mask, p["alpha"], p["dx"], p["dt"], p["beta"], p["Q0"], qs)
    got, nph = run_partitioned_phase_major(codes, mask, nranks, p["alpha"], p["dx"], p["dt"], p["beta"], p["Q0"], qs)
    for a, b in zip(got, ref):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("family,seed,H,W", CASES + [("river", 7, 60, 56)])
def test_boundary_critical_part_is_independent_of_the_bulk(oracle, family, seed, H, W):
    """inside a phase the exports and what drains into them (part 0) and the rest (part 1) do not depend on each other:
    the halo of a round is final after part 0, and the bulk part may run before, beside or after it -- what lets
    lf_dist_router_route exchange on a second stream beside the bulk of the phase"""
    nranks = 4
    codes = syn.make_ldd(family, H, W, seed)
    mask = np.ones((H, W), bool)
    N = H * W
    p = syn.router_params(N, seed=5)
    qs = [syn.lateral_inflow(N, s) for s in range(2)]
    ref = global_reference(oracle, codes, mask, p["alpha"], p["dx"], p["dt"], p["beta"], p["Q0"], qs)
    blocks, graphs = X.build_blocks(codes, mask, nranks)
    nph = D.settle_phases_local(graphs)
    sel = [np.arange(r0 * W, r1 * W) for (r0, r1) in blocks]
    for g in graphs:
        perm, _ph = g.layout()
        for j in range(nph):
            a0, a1 = g.part_range(j, 0)
            b0, b1 = g.part_range(j, 1)
            assert (a0, b1) == g.phase_range(j) and a1 == b0
            for side in (0, 1):                                  # every export of the phase is boundary-critical
                sp = g.round_send_positions(j, side)
                assert ((sp >= a0) & (sp < a1)).all()
    ranks = [X.RankState(g, p["alpha"][s], p["dx"][s], p["dt"], p["beta"], p["Q0"][s]) for g, s in zip(graphs, sel)]
    for step, q in enumerate(qs):
        for rk, s in zip(ranks, sel):
            rk.begin_call(q[s])
        for j in range(nph):
            order = (1, 0) if (step + j) % 2 else (0, 1)         # bulk first on odd turns
            sends = {}
            for rk_i, rk in enumerate(ranks):
                for part in order:
                    rk.compute_part(j, part)
                    if part == 0 and j + 1 < nph:                # the halo as it is right after part 0
                        sends[rk_i] = [rk.send_values(j, side) for side in (0, 1)]
            if j + 1 < nph:
                for k, rk in enumerate(ranks):
                    assert all(np.array_equal(sends[k][side], rk.send_values(j, side)) for side in (0, 1))
                    if k > 0:
                        rk.recv_values(j, 0, sends[k - 1][1])
                    if k + 1 < nranks:
                        rk.recv_values(j, 1, sends[k + 1][0])
        full = np.empty(N)
        for rk, s in zip(ranks, sel):
            full[s] = rk.pixel_values()
        assert np.array_equal(full, ref[step])


@pytest.mark.parametrize("family,seed,H,W,nranks", [("deep", 2, 90, 70, 3), ("saddle", 6, 80, 64, 4), ("river", 7, 120, 90, 3),
                                                    ("shallow", 1, 64, 48, 8)])
def test_block_plan_of_single_router_calls(oracle, family, seed, H, W, nranks):
    """lf_dist_graph's plan for k_sweep_cones<DIST> (one plan per stage: blocks of launch units, cones of <= 64 cells per
    unit), executed on the CPU the way the kernel may see memory -- inside a cone only the unit just solved is new, the
    rest of the block shows old values until the block is complete, the cones of a block run in either order --: the
    single-domain oracle's discharge bit for bit, two calls.  LF_ROUTE_LEVELS=4 in a second pass cuts the blocks short."""
    codes = syn.make_ldd(family, H, W, seed)
    mask = np.ones((H, W), bool)
    N = H * W
    p = syn.router_params(N, seed=5)
    qs = [syn.lateral_inflow(N, s) for s in range(2)]
    ref = global_reference(oracle, codes, mask, p["alpha"], p["dx"], p["dt"], p["beta"], p["Q0"], qs)
    for lmax in (None, "4"):
        if lmax:
            os.environ["LF_ROUTE_LEVELS"] = lmax
        try:
            blocks, graphs = X.build_blocks(codes, mask, nranks)
            nph = D.settle_phases_local(graphs)
        finally:
            os.environ.pop("LF_ROUTE_LEVELS", None)
        plans = [g.route_plan() for g in graphs]
        assert any(pl is not None for pl in plans)
        sel = [np.arange(r0 * W, r1 * W) for (r0, r1) in blocks]
        ranks = [X.RankState(g, p["alpha"][s], p["dx"][s], p["dt"], p["beta"], p["Q0"][s]) for g, s in zip(graphs, sel)]
        multi = 0
        for step, q in enumerate(qs):
            for rk, s in zip(ranks, sel):
                rk.begin_call(q[s])
            for j in range(nph):
                for rk, pl in zip(ranks, plans):
                    for part in (0, 1):
                        if pl is None:
                            rk.compute_part(j, part)
                        else:
                            assert len(pl["stage_block"]) == 2 * nph + 1
                            rk.compute_stage_cones(2 * j + part, pl, reverse=bool((step + j + part) % 2))
                            multi += int((np.diff(pl["level"]) > 1).sum())
                if j + 1 < nph:
                    sends = [[rk.send_values(j, side) for side in (0, 1)] for rk in ranks]
                    for k, rk in enumerate(ranks):
                        if k > 0:
                            rk.recv_values(j, 0, sends[k - 1][1])
                        if k + 1 < nranks:
                            rk.recv_values(j, 1, sends[k + 1][0])
            full = np.empty(N)
            for rk, s in zip(ranks, sel):
                full[s] = rk.pixel_values()
            assert np.array_equal(full, ref[step]), (family, lmax, step)
        assert multi > 0                                          # blocks of several units did occur


def test_block_plan_switched_off(monkeypatch):
    """LF_ROUTE_LEVELS=1: no block holds more than one launch unit -> no plan, the per-unit schedule is what runs"""
    monkeypatch.setenv("LF_ROUTE_LEVELS", "1")
    codes = syn.make_ldd("river", 60, 50, 7)
    blocks, graphs = X.build_blocks(codes, np.ones((60, 50), bool), 3)
    D.settle_phases_local(graphs)
    assert all(g.route_plan() is None for g in graphs)


@pytest.mark.parametrize("family,seed,H,W,nranks", [("deep", 2, 90, 70, 3), ("river", 7, 130, 100, 4), ("saddle", 6, 80, 64, 4)])
def test_block_plan_of_the_fused_path(oracle, family, seed, H, W, nranks):
    """the fused sub-step path's plan (one per phase, cones of <= 256 cells) under the same visibility rules: what a cone
    of k_fused_cones<DIST> reads of the router outputs of a sub-step is either its own unit above or final"""
    codes = syn.make_ldd(family, H, W, seed)
    mask = np.ones((H, W), bool)
    N = H * W
    p = syn.router_params(N, seed=5)
    qs = [syn.lateral_inflow(N, s) for s in range(2)]
    ref = global_reference(oracle, codes, mask, p["alpha"], p["dx"], p["dt"], p["beta"], p["Q0"], qs)
    os.environ["LF_FUSED_LEVELS"] = "6"
    try:
        blocks, graphs = X.build_blocks(codes, mask, nranks)
        nph = D.settle_phases_local(graphs)
    finally:
        os.environ.pop("LF_FUSED_LEVELS", None)
    plans = [g.fused_plan() for g in graphs]
    assert any(pl is not None and len(pl["stage_block"]) == nph + 1 for pl in plans)
    sel = [np.arange(r0 * W, r1 * W) for (r0, r1) in blocks]
    ranks = [X.RankState(g, p["alpha"][s], p["dx"][s], p["dt"], p["beta"], p["Q0"][s]) for g, s in zip(graphs, sel)]
    for step, q in enumerate(qs):
        for rk, s in zip(ranks, sel):
            rk.begin_call(q[s])
        for j in range(nph):
            for rk, pl in zip(ranks, plans):
                if pl is None:
                    rk.compute_phase(j)
                else:
                    rk.compute_stage_cones(j, pl, reverse=bool((step + j) % 2), max_cone=256)
            if j + 1 < nph:
                sends = [[rk.send_values(j, side) for side in (0, 1)] for rk in ranks]
                for k, rk in enumerate(ranks):
                    if k > 0:
                        rk.recv_values(j, 0, sends[k - 1][1])
                    if k + 1 < nranks:
                        rk.recv_values(j, 1, sends[k + 1][0])
        full = np.empty(N)
        for rk, s in zip(ranks, sel):
            full[s] = rk.pixel_values()
        assert np.array_equal(full, ref[step]), (family, step)
