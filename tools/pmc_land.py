"""Lean driver for counter passes over the land-surface stages of the resident model step (round 6):

    python tools/pmc_land.py <fused|separate|lean> <size> <model steps>

builds HotPathDevice on the size x size hot-path scenario (bench.py's, `deep` LDD) and runs <model steps> steps on forcing
resident in HBM.  fused: canopy + ESMax + soil columns in one pass (lf_land_columns_device, the default); separate: the
three launches of rounds 1-5 (land_fused=False); lean: fused with every optional map left out (report=()).
Under `rocprofv3 --pmc FETCH_SIZE` / `WRITE_SIZE` (tools/pmc_r06.sh) the per-kernel totals give HBM bytes per pixel-step."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "lisflood-code_amd"))
from lisflood_amd import _lib, synthetic as syn          # noqa: E402
from lisflood_amd.hotpath import HotPathDevice           # noqa: E402

mode, size, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
H = W = size
N = H * W
values, sc, mask, ldd_to_chan, ldd_kin = syn.hotpath_scenario(H, W, family="deep", block=1_000_000)
hp = HotPathDevice(values, sc, mask, ldd_to_chan, ldd_kin, split=True, land_fused=(mode != "separate"),
                   report=(() if mode == "lean" else None))
del values
for s in range(2):
    hp.step(syn.hotpath_forcing(N, s), s + 1)
_lib.synchronize()
for s in range(steps):
    hp.step(None, s + 3)
_lib.synchronize()
print("land", mode, size, "pixels", N, "model steps", steps + 2, "stage bytes", hp.stage_bytes(), flush=True)
