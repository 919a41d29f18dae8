"""Sub-step histogram of the deferred columns of bench.py's `wet` soil, step by step (runs on the GPU box)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lisflood-code_amd"))
from lisflood_amd import _lib, synthetic as syn          # noqa: E402
from lisflood_amd.soilloop import SoilColumnsDevice      # noqa: E402

dev = SoilColumnsDevice(syn.soil_params(4_000_000, seed=3))
k = np.arange(128)
for s in range(12):
    dev.step()
    _lib.synchronize()
    h = dev.substep_histogram()
    c = np.cumsum(h)
    print(s, "deferred", int(h.sum()), "mean sub-steps %.2f" % ((h * k).sum() / max(h.sum(), 1)),
          "median/p90/p99/max", [int(np.searchsorted(c, q * h.sum())) for q in (0.5, 0.9, 0.99)] + [int(np.nonzero(h)[0].max())])
    if s in (2, 6, 11):     # the tail: share of the deferred columns and of their sub-steps above a threshold
        tot, work = h.sum(), (h * k).sum()
        print("   ", ["T=%d: %.3f of deferred, %.3f of their sub-steps" % (T, h[T + 1:].sum() / tot, (h * k)[T + 1:].sum() / work)
                      for T in (4, 8, 12, 16, 24, 32, 48)])
