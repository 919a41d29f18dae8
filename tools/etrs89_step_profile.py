"""Where a model step of the LF_ETRS89 chain spends its time (run on the GPU box): host enqueue rate against the wall
clock, with and without a synchronisation per step, and a cProfile of the host side."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.join(os.getcwd(), "lisflood-code_amd"))
import numpy as np
from lisflood_amd import _lib
from lisflood_amd.hotpath import HotPathDevice
g = np.load("tests/golden/etrs89_chain.npz")
cp = lambda d: {k: (np.array(a, copy=True) if isinstance(a, np.ndarray) else a) for k, a in d.items()}
values = {k[4:]: g[k] for k in g.files if k.startswith("val_")}
sc = {k[3:]: float(g[k]) for k in g.files if k.startswith("sc_")}
st = {k[3:]: (g[k] if g[k].ndim else float(g[k])) for k in g.files if k.startswith("st_")}
qin = np.array(g["QInM3"])
forcing = [{k[5:]: np.ascontiguousarray(g[k][s]) for k in g.files if k.startswith("forc_")} for s in range(12)]
hp = HotPathDevice(cp(values), sc, g["mask"], g["ldd_to_chan"], g["ldd_cut"], split=True, structures=cp(st))
for s, f in enumerate(forcing): hp.step(f, s + 1, QInM3=qin[s])
_lib.synchronize()
def loop(n, sync_each):
    t0 = time.perf_counter()
    for k in range(n):
        s = k % 12
        hp.step(forcing[s], s + 1, QInM3=qin[s])
        if sync_each: _lib.synchronize()
    t1 = time.perf_counter()
    _lib.synchronize()
    t2 = time.perf_counter()
    return (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3
print("enqueue-only ms/step, total ms/step:", loop(60, False))
print("sync each step:", loop(60, True))
t0=time.perf_counter()
for k in range(60): hp.step(None, k+1)
_lib.synchronize(); print("step(None) no uploads:", (time.perf_counter()-t0)/60*1e3)
pr = cProfile.Profile(); pr.enable()
for k in range(30):
    s = k % 12; hp.step(forcing[s], s + 1, QInM3=qin[s])
_lib.synchronize(); pr.disable()
o = io.StringIO(); pstats.Stats(pr, stream=o).sort_stats("cumulative").print_stats(18); print(o.getvalue()[:3500])
