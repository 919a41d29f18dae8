"""N ranks of `bench.py --gpus N --no-extra ...` on one GPU through tests/fake_rccl, rank 0 under rocprofv3 --kernel-trace:
where a rank's time goes between its kernels when the halo rounds are real cross-process exchanges.
usage: python tools/trace_rank0_of_shared_gpu_job.py N outdir [bench.py arguments]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))


def main():
    world, out = int(sys.argv[1]), os.path.abspath(sys.argv[2])
    args = ["--gpus", str(world), "--no-extra"] + sys.argv[3:]
    fake = os.path.join(ROOT, "tests", "fake_rccl")
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29911", TORCHELASTIC_RUN_ID="trace%d" % os.getpid(), HSA_ENABLE_IPC_MODE_LEGACY="0",
                   FAKE_RCCL_TIMEOUT_S="120", HIP_VISIBLE_DEVICES="0", TMPDIR="/tmp",
                   LD_LIBRARY_PATH=fake + os.pathsep + os.environ.get("LD_LIBRARY_PATH", ""))
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
        if rank == 0:
            cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "-o", "r0", "--"] + cmd
        procs.append(subprocess.Popen(cmd, env=env, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1200) for p in procs]
    for r, (p, (o, e)) in enumerate(zip(procs, outs)):
        print("rank", r, "rc", p.returncode, (o.strip().splitlines() or [""])[-1][:300])
        if p.returncode != 0:
            print(e[-1500:])


if __name__ == "__main__":
    main()
