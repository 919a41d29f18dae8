import sys, ctypes as C, numpy as np
sys.path[:0]=['lisflood-code_amd','.']
from lisflood_amd import _lib, synthetic as syn
from lisflood_amd.hotpath import HotPathDevice
H=W=2000; N=H*W
values, sc, mask, l2c, lk = syn.hotpath_scenario(H, W, block=1_000_000)
hp = HotPathDevice(values, sc, mask, l2c, lk, split=True)
for s in range(6):
    hp.step(syn.hotpath_forcing(N, s % 2), s+1); _lib.synchronize()
    nd=C.c_int64(0); _lib.check(_lib.lib().lf_soil_last_deferred(C.c_int(0), C.byref(nd)))
    h=np.zeros(128,np.int64); _lib.check(_lib.lib().lf_soil_substep_histogram(C.c_int(0), h.ctypes.data_as(C.c_void_p), C.c_int(128)))
    k=np.arange(128); c=np.cumsum(h)
    print(s, "multi frac %.4f" % (nd.value/(3*N)), "mean %.1f" % ((h*k).sum()/max(h.sum(),1)), "p50/p90/p99", [int(np.searchsorted(c, q*h.sum())) for q in (0.5,0.9,0.99)], ">16: %.3f of multi" % (h[17:].sum()/max(h.sum(),1)))
