"""Worker of tests/test_gpu_parity.py::test_handles_created_after_a_fork: the reference runs Monte-Carlo / EnKF members as
forked processes (main.py:104-106), each building its own model.  The library is loaded BEFORE the fork (no HIP call
happens at load time), every child then creates its own handles, routes and checks against the oracle; the parent does
the same after its children are gone."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lisflood-code_amd"), os.path.join(ROOT, "oracle")):
    sys.path.insert(0, p)

import oracle  # noqa: E402
from lisflood_amd import _lib, synthetic as syn  # noqa: E402
from lisflood_amd.kinematic_wave_parallel import kinematicWave  # noqa: E402


def member(seed):
    H, W = 60, 50
    codes = syn.make_ldd("shallow", H, W, seed)
    mask = np.ones((H, W), bool)
    N = H * W
    p = syn.router_params(N, seed=seed)
    c = codes.reshape(-1).astype(np.float64)
    gpu = kinematicWave(c, mask, p["alpha"], p["beta"], p["dx"], p["dt"])
    cpu = oracle.kinematicWave(c, mask, p["alpha"], p["beta"], p["dx"], p["dt"])
    Qg, Qc = p["Q0"].copy(), p["Q0"].copy()
    for s in range(3):
        q = syn.lateral_inflow(N, s)
        gpu.kinematicWaveRouting(Qg, q)
        cpu.kinematicWaveRouting(Qc, q)
    np.testing.assert_allclose(Qg, Qc, rtol=1e-9, atol=1e-12)
    gpu.close()


def main():
    oracle.build()
    _lib.lib()                      # shared object loaded, symbols resolved -- and no device touched yet
    pids = []
    for k in range(3):
        pid = os.fork()
        if pid == 0:
            try:
                member(10 + k)
                os._exit(0)
            except BaseException as e:  # noqa: BLE001
                print("child %d failed: %r" % (k, e), file=sys.stderr, flush=True)
                os._exit(1)
        pids.append(pid)
    bad = [pid for pid in pids if os.waitpid(pid, 0)[1] != 0]
    assert not bad, bad
    member(99)                      # the parent creates its handles after the fork
    print("FORK_OK members=%d" % len(pids))


if __name__ == "__main__":
    main()
