"""N ranks of `bench.py --gpus N ...` on ONE GPU through tests/fake_rccl (see tests/test_dist_multirank_gpu.py): a
functional run of the multi-rank job at full size -- memory, phases, halo protocol, checksums -- NOT a scaling
measurement (the ranks share one device's CUs and HBM).  usage: python tools/ranks_on_one_gpu.py N [bench.py arguments]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))


def main():
    import subprocess
    import test_dist_multirank_gpu as M
    world = int(sys.argv[1])
    fake = M.FAKE
    src, so = os.path.join(fake, "fake_rccl.hip"), os.path.join(fake, "librccl.so.1")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.run(["hipcc", "-O2", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", so, src, "-lrt"], check=True)
    outs = M.run_ranks(fake, world, "bench.py", {"FAKE_RCCL_TIMEOUT_S": "300"},
                       args=["--gpus", str(world)] + sys.argv[2:], timeout=3000)
    sys.stderr.write(outs[0][1][-6000:])
    sys.stdout.write(outs[0][0])


if __name__ == "__main__":
    main()
