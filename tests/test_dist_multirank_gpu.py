"""-m gpu: the row-block partition as an N-GPU job runs it -- one PROCESS per rank, socket rendezvous, the product's own
sequence of group calls / sends / receives / streams / events -- on the ONE GPU a test box has.

RCCL refuses two ranks on one device, so the workers find tests/fake_rccl/librccl.so.1 first on LD_LIBRARY_PATH: a test
stand-in for the eight RCCL entry points csrc/lf_dist.hip binds, moving the messages between processes through hipIpc
mailboxes with RCCL's semantics (asynchronous on the caller's stream, grouped operations progress together, messages
matched per (source, destination) in order, byte counts checked, every wait bounded).  What these tests prove is the
product's side of the protocol between real processes: who sends what to whom in which order, that every receive has
its send, that kernels wait for the halo they read.  The wire itself (RCCL over xGMI) is covered by
test_gpu_parity.py::test_two_rank_rccl_halo_exchange on boxes with two devices."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAKE = os.path.join(ROOT, "tests", "fake_rccl")
_PORT = [29700]


@pytest.fixture(scope="module")
def fake_rccl():
    from lisflood_amd import _lib
    if _lib.device_count() == 0:
        pytest.fail("no HIP device: the gpu tests must run on an MI355X box")
    src, so = os.path.join(FAKE, "fake_rccl.hip"), os.path.join(FAKE, "librccl.so.1")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["hipcc", "-O2", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so, src, "-lrt"], check=True,
                       timeout=300)
    return FAKE


def check_issue_order(log_dir, world):
    """RCCL pairs the operations of two ranks in host issue order: for every pair (a, b) the sizes of a's sends to b and of
    b's receives from a, each in the order its rank issued them, must be the same sequence -- whatever streams, groups
    and events lie between (every rank walks the same sequence of (call, round) pairs).  -> the
    number of operations checked; raises AssertionError naming the first pair and position that differ."""
    ops = {}
    for r in range(world):
        path = os.path.join(log_dir, "rank%d.log" % r)
        rows = [l.split() for l in open(path)] if os.path.exists(path) else []
        ops[r] = [(int(g), kind, int(peer), int(nbytes)) for g, kind, peer, nbytes in rows]
    n = 0
    for a in range(world):
        for b in range(world):
            if a == b:
                continue
            sends = [x[3] for x in ops[a] if x[1] == "send" and x[2] == b]
            recvs = [x[3] for x in ops[b] if x[1] == "recv" and x[2] == a]
            assert sends == recvs, "ranks %d -> %d: %d sends %s..., %d receives %s..." % (
                a, b, len(sends), sends[:8], len(recvs), recvs[:8])
            n += len(sends)
    return n


def run_ranks(fake, world, script, extra_env, args=(), timeout=600, expect_failure=False):
    """`world` processes of `script`, all on device 0, rendezvous on a fresh port; -> [(stdout, stderr)] per rank.  Every
    run also checks the issue order the stand-in recorded (check_issue_order)."""
    import tempfile
    _PORT[0] += 1
    procs = []
    log_dir = tempfile.mkdtemp(prefix="fake_rccl_order_")
    extra_env = dict(extra_env, FAKE_RCCL_LOG_DIR=log_dir)
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(_PORT[0]), TORCHELASTIC_RUN_ID="fake%d_%d" % (os.getpid(), _PORT[0]),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", FAKE_RCCL_TIMEOUT_S="60",
                   HIP_VISIBLE_DEVICES=os.environ.get("HIP_VISIBLE_DEVICES", "0").split(",")[0],  # ONE device, whatever the box
                   LD_LIBRARY_PATH=fake + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
        env.update(extra_env)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, script)] + list(args), env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=timeout))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    if expect_failure:
        return outs, log_dir, [p.returncode for p in procs]
    if any(p.returncode != 0 for p in procs):
        pytest.fail("\n".join("--- rank %d rc=%s\n%s\n%s" % (rank, p.returncode, o[-1500:], e[-2500:])
                              for rank, (p, (o, e)) in enumerate(zip(procs, outs))))
    check_issue_order(log_dir, world)
    return outs


@pytest.mark.parametrize("family,world", [("saddle", 2), ("saddle", 3), ("shallow", 4), ("deep", 3), ("river", 4)])
def test_router_calls_between_processes(fake_rccl, family, world):
    """lf_dist_router_route (boundary-critical part, halo on the second stream beside the bulk) and
    lf_dist_router_route_many (calls pipelined on alternating state vectors) with `world` ranks: the single-domain
    oracle's discharge after 4 + 4 calls"""
    outs = run_ranks(fake_rccl, world, "tests/dist_worker_rccl.py", {"LF_TEST_FAMILY": family})
    assert "DIST_RCCL_OK" in outs[0][0] and "ranks=%d" % world in outs[0][0]


def test_a_misordered_rank_is_caught(fake_rccl):
    """the check of the check: three ranks exchange their halo rounds through the product's own lf_dist_router_exchange,
    rank 1 in REVERSE round order.  The stand-in's issue log must show the mismatch (and the run itself must not pass: the
    byte counts of the rounds differ, or a receive waits for a send that comes later)"""
    (outs, log_dir, rcs) = run_ranks(fake_rccl, 3, "tests/dist_worker_order.py",
                                     {"LF_TEST_FAMILY": "shallow", "LF_TEST_REVERSED_RANK": "1", "FAKE_RCCL_TIMEOUT_S": "5"},
                                     timeout=300, expect_failure=True)
    diag = "\n".join("--- rank %d rc=%s\n%s\n%s" % (r, rcs[r], o[-800:], e[-1500:]) for r, (o, e) in enumerate(outs))
    try:      # rank 1's neighbours see its sizes in the wrong order (the middle rank talks to both)
        check_issue_order(log_dir, 3)
    except AssertionError:
        pass
    else:
        pytest.fail("the reversed rank went unnoticed\n" + diag)
    assert any(rc != 0 for rc in rcs) or not all("ORDER_WORKER_OK" in o for o, _ in outs), [o for o, _ in outs]
    # ... and the same worker with every rank in round order is clean
    outs = run_ranks(fake_rccl, 3, "tests/dist_worker_order.py", {"LF_TEST_FAMILY": "shallow"}, timeout=300)
    assert all("ORDER_WORKER_OK" in o for o, _ in outs)


def test_plain_router_call_after_a_pipelined_sequence(fake_rccl):
    """the order nothing else runs: lf_dist_router_route_many first (it creates the halo stream), plain
    lf_dist_router_route calls -- which order the halo stream by two events -- after it"""
    outs = run_ranks(fake_rccl, 2, "tests/dist_worker_rccl.py", {"LF_TEST_FAMILY": "saddle", "LF_TEST_MANY_FIRST": "1"})
    assert "DIST_RCCL_OK" in outs[0][0]


@pytest.mark.parametrize("family,world,split", [("saddle", 2, True), ("saddle", 3, False), ("shallow", 4, True),
                                                ("deep", 3, True), ("river", 4, True)])
def test_fused_model_step_between_processes(fake_rccl, family, world, split):
    """lf_dist_routing_substeps_fused with `world` ranks, two model steps of 7 sub-steps: bit-identical to the whole
    raster's lf_routing_substeps_fused"""
    outs = run_ranks(fake_rccl, world, "tests/dist_worker_fused.py",
                     {"LF_TEST_FAMILY": family, "LF_TEST_SPLIT": "1" if split else "0"})
    assert "DIST_FUSED_OK" in outs[0][0]


@pytest.mark.parametrize("world", [2, 8])
def test_bench_line_of_a_multi_rank_job(fake_rccl, world):
    """`bench.py --gpus N` as the driver launches it (one process per rank, RANK / WORLD_SIZE / MASTER_* from the
    environment) at a reduced raster: ONE JSON line on rank 0's stdout, finite, and the row-block partition's discharge
    sum equal to the catchment partition's (both bit-identical to the single domain) to summation order"""
    import json
    outs = run_ranks(fake_rccl, world, "bench.py", {},
                     args=["--gpus", str(world), "--size", "1600", "--steps", "4", "--warmup", "1"], timeout=900)
    lines = [l for l in outs[0][0].splitlines() if l.strip()]
    assert len(lines) == 1, outs[0][0][-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == world and d["finite"] and d["value"] > 0
    assert d["row_block_vs_catchment_partition_sumQ_rel_diff"] < 1e-12
    step = d["model_step_24_substeps_split_row_blocks"]
    assert step.get("finite") is True, step
    for rank in range(1, world):
        assert outs[rank][0].strip() == ""


@pytest.mark.parametrize("how", ["raises", "hangs"])
def test_bench_line_when_the_row_block_path_fails(fake_rccl, how):
    """the N > 1 line must exist even if the row-block RCCL path fails or hangs on the first real links it meets: the
    catchment partition (measured first, no exchange on the data path) becomes the headline and the line says why"""
    import json
    env = {"LF_BENCH_FAIL_ROW_BLOCKS": "1"} if how == "raises" else {"LF_BENCH_HANG_ROW_BLOCKS": "1", "LF_BENCH_RCCL_TIMEOUT_S": "15"}
    outs = run_ranks(fake_rccl, 2, "bench.py", env,
                     args=["--gpus", "2", "--size", "1200", "--steps", "4", "--warmup", "1"], timeout=600)
    lines = [l for l in outs[0][0].splitlines() if l.strip()]
    assert len(lines) == 1, outs[0][0][-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["finite"] and d["value"] > 0 and d["metric"].startswith("Mcell-steps/s")
    assert "row_block_error" in d and "catchment partition" in d["config"]["parallelism"]
    assert outs[1][0].strip() == ""
