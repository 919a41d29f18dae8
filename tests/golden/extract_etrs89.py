"""Extract the LF_ETRS89_UseCase static maps the routing hot path needs into one small .npz.

THIS CONTAINER ONLY. Run with the conda interpreter (the only one that has h5py):
    /opt/conda/bin/python3.9 tests/golden/extract_etrs89.py
Reads netCDF-4/HDF5 test maps from /root/reference/tests/data/LF_ETRS89_UseCase/maps (data files held
by the reference's own tests -- fixtures, not source) and writes tests/golden/etrs89_static.npz.
"""
import os
import numpy as np
import h5py

SRC = "/root/reference/tests/data/LF_ETRS89_UseCase/maps"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "etrs89_static.npz")

FILES = {
    "ldd": "ec_ldd.nc", "chan": "chan.nc", "chanlength": "chanlength.nc", "changrad": "changrad.nc",
    "chanman": "ec_chanman.nc", "chanbw": "ec_chanbw.nc", "chanbnkf": "ec_chanbnkf.nc", "chans": "chans.nc",
    "pixarea": "pixarea.nc", "pixleng": "pixleng.nc", "gradient": "gradient.nc",
    "lakes": "ec_lakes.nc", "res": "ec_res.nc", "outlets": "ec_outlets.nc", "uparea": "ec_upArea.nc",
    "calchanman1": "parameters/params_CalChanMan1.nc", "calchanman2": "parameters/params_CalChanMan2.nc",
    "lakemultiplier": "parameters/params_LakeMultiplier.nc", "adjust_normal_flood": "parameters/params_adjust_Normal_Flood.nc",
    "reservoirrnormqmult": "parameters/params_ReservoirRnormqMult.nc", "avgdis": "safe_init/avgdis.nc",
}
# look-up tables of lakes.initial / reservoir.initial (lakes.py:96-131, reservoir.py:84-118): two columns, id value
TABLES = ("lakearea", "lakea", "lakeavinflow", "rtstor", "rclim", "rnlim", "rflim", "rndq", "rnormq", "rminq")

def main_var(f):
    best = None
    for k, v in f.items():
        if isinstance(v, h5py.Dataset) and v.ndim == 2:
            best = k
    return best

out = {}
for key, fn in FILES.items():
    p = os.path.join(SRC, fn)
    if not os.path.exists(p):
        print("missing", p); continue
    with h5py.File(p, "r") as f:
        name = main_var(f)
        ds = f[name]
        arr = ds[...]
        fill = ds.attrs.get("_FillValue", None)
        print(key, fn, name, arr.dtype, arr.shape, "fill", fill)
        out[key] = arr
        if fill is not None:
            out[key + "_fill"] = np.asarray(fill).reshape(-1)[:1]
for t in TABLES:
    rows = [ln.split() for ln in open(os.path.join(SRC, "tables", t + ".txt")) if ln.strip()]
    out["table_" + t] = np.array([[float(a), float(b)] for a, b in rows])
    print("table", t, out["table_" + t].shape)

# ---- PCRaster .map catchment masks of the use case (CSF 2 files: 256-byte header, then the cells; UINT1 boolean maps
# with 255 = missing value): mask.map is the model domain of cold.xml, subcatchment_mask.map / intercatchment_mask.map
# the sub-domains of the reference's tests/test_subcatchments.py.  Each is a PCRaster-made catchment of the LDD: the only
# PCRaster `catchment` outputs in the checkout, placed here on the 57 x 80 grid of the netCDF maps by their origin.
import struct


def read_csf_mask(path):
    b = open(path, "rb").read()
    assert b[:27] == b"RUU CROSS SYSTEM MAP FORMAT"
    cell_repr, = struct.unpack_from("<H", b, 66)
    xul, yul = struct.unpack_from("<dd", b, 84)
    nr, nc = struct.unpack_from("<II", b, 100)
    cs, = struct.unpack_from("<d", b, 108)
    assert cell_repr == 0                                   # CR_UINT1
    a = np.frombuffer(b, np.uint8, nr * nc, 256).reshape(nr, nc)
    return a == 1, int(round((2615000.0 - yul) / cs)), int(round((xul - 4050000.0) / cs))


shape = out["ldd"].shape
for key, fn in (("mask_map", "mask.map"), ("subcatchment_mask", "subcatchment_mask.map"),
                ("intercatchment_mask", "intercatchment_mask.map")):
    m, r0, c0 = read_csf_mask(os.path.join(SRC, fn))
    full = np.zeros(shape, bool)
    full[r0:r0 + m.shape[0], c0:c0 + m.shape[1]] = m
    out[key] = full
    print(key, fn, m.shape, "origin row/col", r0, c0, "cells", int(full.sum()))
np.savez_compressed(OUT, **out)
print("wrote", OUT, os.path.getsize(OUT))

# ---- meteorological forcing of the same use case (meteo_1950/, 6-hourly fields): the first NT fields of
# precipitation and the three reference evaporation rates, as the forcing of the model-step chain fixture
# (make_golden.py chain).  Data files of the reference's own tests; float32 as stored.
NT = 16
MET = os.path.join(os.path.dirname(SRC), "meteo_1950")
met = {}
for key, fn in (("pr", "pr.nc"), ("e0", "e0.nc"), ("es", "es.nc"), ("et", "et.nc"), ("ta", "ta.nc")):
    with h5py.File(os.path.join(MET, fn), "r") as f:
        name = [k for k, v in f.items() if isinstance(v, h5py.Dataset) and v.ndim == 3][0]
        met[key] = f[name][:NT].astype(np.float32)
        print(key, fn, name, met[key].shape, float(np.nanmin(met[key])), float(np.nanmax(met[key])))
OUT_MET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "etrs89_meteo.npz")
np.savez_compressed(OUT_MET, **met)
print("wrote", OUT_MET, os.path.getsize(OUT_MET))
