"""Bandwidth ceiling of k read streams + 1 write stream of fp64 (8 bytes per lane), the byte mix of the level sweep
without arithmetic or gather: python tools/bench_streams.py [n]   (lf_calibration_streams)"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lisflood-code_amd"))
from lisflood_amd import _lib  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
L = _lib.lib()
srcs = [_lib.DeviceArray(n).zero() for _ in range(8)]
dst = _lib.DeviceArray(n).zero()
ptrs = (C.c_void_p * 8)(*[s.ptr.value for s in srcs])
for k in range(1, 9):
    for rep in range(3):
        _lib.check(L.lf_calibration_streams(C.c_int(0), C.c_int(k), ptrs, dst.ptr, C.c_int64(n)))
    _lib.synchronize()
    _lib.timer_start()
    reps = 20
    for rep in range(reps):
        _lib.check(L.lf_calibration_streams(C.c_int(0), C.c_int(k), ptrs, dst.ptr, C.c_int64(n)))
    ms = _lib.timer_stop() / reps
    print("%d read + 1 write streams of %d doubles: %.1f us  %.2f TB/s" % (k, n, ms * 1e3, (k + 1) * 8 * n / (ms * 1e-3) / 1e12),
          flush=True)
