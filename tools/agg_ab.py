"""Same-process A/B of the per-pixel aggregates kernel: block by block (LF_AGG_ALL=0) against all loads up front
python tools/agg_ab.py [size]"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
from lisflood_amd import _lib  # noqa: E402
from lisflood_amd import synthetic as syn  # noqa: E402
from lisflood_amd.hotpath import HotPathDevice  # noqa: E402
from lisflood_amd._lib import check, lib  # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
H = W = size
N = H * W
values, sc, mask, ldd_to_chan, ldd_kin = syn.hotpath_scenario(H, W, family="deep", block=1_000_000)
hp = HotPathDevice(values, sc, mask, ldd_to_chan, ldd_kin, split=True)
del values
for s in range(2):
    f = {k: (a if hp.pixel_of_position is None else a[hp.pixel_of_position]) for k, a in syn.hotpath_forcing(N, s).items()}
    hp.step(f, s + 1, ordered=True)
_lib.synchronize()
L = lib()
dev = C.c_int(hp.device)
nb = hp.stage_bytes()["pixel_aggregates"]
for x in ("0", "1", "0", "1", "0", "1"):
    os.environ["LF_AGG_ALL"] = x
    for _ in range(3):
        check(L.lf_pixel_aggregates_device(dev, C.byref(hp.pixel)))
    _lib.synchronize()
    t0 = time.perf_counter()
    reps = 20
    for _ in range(reps):
        check(L.lf_pixel_aggregates_device(dev, C.byref(hp.pixel)))
    _lib.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / reps
    print("LF_AGG_ALL=%s  %.3f ms  %.2f TB/s algorithmic" % (x, ms, nb / ms / 1e9), flush=True)
hp.free()
