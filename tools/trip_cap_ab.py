"""Same-process A/B of the soil trip cap on the resident model step: python tools/trip_cap_ab.py [size] [caps]
(LF_SOIL_TRIP_CAP is read at every call; blocks of 6 steps per cap, alternating, so the drift of the soil state is shared)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
from lisflood_amd import _lib  # noqa: E402
from lisflood_amd import synthetic as syn  # noqa: E402
from lisflood_amd.hotpath import HotPathDevice  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
caps = sys.argv[2].split(",") if len(sys.argv) > 2 else ["16", "10", "8"]
H = W = size
N = H * W
values, sc, mask, ldd_to_chan, ldd_kin = syn.hotpath_scenario(H, W, family="deep", block=1_000_000)
hp = HotPathDevice(values, sc, mask, ldd_to_chan, ldd_kin, split=True)
del values
for s in range(2):
    f = {k: (a if hp.pixel_of_position is None else a[hp.pixel_of_position]) for k, a in syn.hotpath_forcing(N, s).items()}
    hp.step(f, s + 1, ordered=True)
_lib.synchronize()
n = 3
for rnd in range(4):
    for cap in caps:
        os.environ["LF_SOIL_TRIP_CAP"] = cap
        hp.step(None, n); n += 1
        _lib.synchronize()
        t0 = time.perf_counter()
        for s in range(6):
            hp.step(None, n); n += 1
        _lib.synchronize()
        print("round %d  cap %-3s %.3f ms per model step" % (rnd, cap, (time.perf_counter() - t0) * 1e3 / 6), flush=True)
hp.free()
