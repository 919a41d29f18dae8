"""Micro-benchmark of the raster-space upstream reduction (LDS-staged 3x3 LDD neighbourhoods) at 10000^2."""
import sys, ctypes as C, numpy as np
sys.path.insert(0, "lisflood-code_amd")
from lisflood_amd._lib import DeviceArray, check, lib, timer_start, timer_stop
H = W = 10000
rng = np.random.default_rng(0)
ldd = rng.integers(1, 10, (H, W)).astype(np.uint8)
w = rng.random((H, W))
dl, dw = DeviceArray.from_host(ldd), DeviceArray.from_host(w)
do = DeviceArray((H, W), np.float64)
L = lib()
for _ in range(3):
    check(L.lf_upstream_sum_raster_device(0, dl.ptr, dw.ptr, do.ptr, H, W))
timer_start(0)
for _ in range(20):
    check(L.lf_upstream_sum_raster_device(0, dl.ptr, dw.ptr, do.ptr, H, W))
ms = timer_stop(0) / 20
print("raster upstream 10000^2: %.3f ms, %.2f TB/s algorithmic (17 B/cell)" % (ms, H * W * 17 / ms / 1e9))
