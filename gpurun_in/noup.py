import sys, time, numpy as np
sys.path[:0]=['lisflood-code_amd','.']
from lisflood_amd import _lib, synthetic as syn
from lisflood_amd.hotpath import HotPathDevice
import ctypes as C
H=W=5000; N=H*W
values, sc, mask, l2c, lk = syn.hotpath_scenario(H, W, block=1_000_000)
hp = HotPathDevice(values, sc, mask, l2c, lk, split=True); del values
forc=[]
for s in range(2):
    f=hp.pinned_forcing()
    for k,a in syn.hotpath_forcing(N,s).items(): f[k][:]=a[hp.pixel_of_position]
    forc.append(f)
for w in range(3): hp.step(forc[w%2], w+1, ordered=True)
_lib.synchronize()
for mode in ("upload","no_upload"):
    for overlap in (True, False):
        hp.overlap_channel = overlap
        hp.step(forc[0], 10, ordered=True); _lib.synchronize()
        t0=time.perf_counter()
        for s in range(6):
            if mode=="upload":
                hp.step(forc[s%2], 11+s, ordered=True); hp.prefetch(forc[(s+1)%2], ordered=True)
            else:
                L=_lib.lib(); b=hp.steps_done%2
                hp._use_set(b); hp._enqueue(11+s)
        _lib.synchronize()
        print(mode, "overlap" if overlap else "one_stream", round((time.perf_counter()-t0)*1e3/6,3), "ms")
