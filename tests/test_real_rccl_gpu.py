"""The N > 1 code path of bench.py against the REAL librccl, every round: `bench.py --force-dist` with one rank builds an
RCCL communicator of size 1 through the library under /opt/rocm (not tests/fake_rccl), runs the row-block router calls,
the row-block model step and the catchment partition, and prints the one JSON line.  With one rank the catchment
partition IS the single-domain engine, so the checksum comparison of the two partitions pins the row-block path to it.
(RCCL with more than one rank needs a second GPU: tests/test_dist_multirank_gpu.py runs that where there is one.)"""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_force_dist_one_rank_on_the_real_rccl(tmp_path):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "LF_RCCL_LIBRARY"):
        env.pop(k, None)
    env["MASTER_ADDR"] = "127.0.0.1"
    env["LD_LIBRARY_PATH"] = ":".join(p for p in env.get("LD_LIBRARY_PATH", "").split(":") if p and "fake_rccl" not in p)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--force-dist", "--size", "3000", "--steps", "4",
                        "--warmup", "1"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = r.stdout.strip().splitlines()
    assert len(lines) == 1, lines[:5]                       # stdout = the ONE JSON line (RCCL's banner goes to stderr)
    d = json.loads(lines[0])
    lib = d.get("rccl_library") or ""
    assert lib.startswith("/opt/rocm") and "fake_rccl" not in lib, lib
    assert d["n_gpus"] == 1 and d["finite"] and d["scaling"] == "strong" and d["value"] > 0
    assert d["model_step_24_substeps_split_row_blocks"]["finite"], d["model_step_24_substeps_split_row_blocks"]
    assert d["catchment_partition"]["finite"], d["catchment_partition"]
    assert d["row_block_vs_catchment_partition_sumQ_rel_diff"] < 1e-12


def test_two_ranks_under_the_driver_launcher_give_a_line_either_way(tmp_path):
    """`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`, the driver's own command, against the real
    librccl.  Where there are two devices the row-block path runs over RCCL and is the headline; on a one-GPU box RCCL
    refuses two ranks on one device (ncclCommInitRank: invalid usage) -- the first real failure of that path anyone has
    seen -- and the line must still go out, with the catchment partition as the headline and the reason in it."""
    from lisflood_amd import _lib
    try:
        import torch.distributed.run  # noqa: F401  (only the launcher; the ranks import no PyTorch)
    except Exception:
        pytest.skip("no torch.distributed.run launcher here")
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LF_RCCL_LIBRARY"):
        env.pop(k, None)
    env["LD_LIBRARY_PATH"] = ":".join(p for p in env.get("LD_LIBRARY_PATH", "").split(":") if p and "fake_rccl" not in p)
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env["LF_BENCH_RCCL_TIMEOUT_S"] = "120"
    port = 29600 + os.getpid() % 300
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size", "1600",
                        "--steps", "4", "--warmup", "1"], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["finite"] and d["value"] > 0 and d["metric"].startswith("Mcell-steps/s")
    assert (d.get("rccl_library") or "").startswith("/opt/rocm")
    if _lib.device_count() < 2:
        assert "ncclCommInitRank" in d["row_block_error"] and "catchment partition" in d["config"]["parallelism"], d
    else:
        assert "row_block_error" not in d and d["row_block_vs_catchment_partition_sumQ_rel_diff"] < 1e-12, d
