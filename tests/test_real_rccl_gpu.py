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
