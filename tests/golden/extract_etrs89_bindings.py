"""Extract what cold.xml BINDS for the land-surface initialisation of LF_ETRS89 into two small .npz files.

THIS CONTAINER ONLY.  Run with the conda interpreter (the only one that has h5py):
    /opt/conda/bin/python3.9 tests/golden/extract_etrs89_bindings.py

The reference's modules ask for their inputs by binding name (`loadmap('MapKSat1')`, add1.py): the settings file maps the
name to a number or to a netCDF map under maps/.  This script resolves the names exactly as the settings file does --
it parses settings/cold.xml (lfuser + lfbinding, $(Name) substitution) -- and stores, per binding the hot-path
initialisations read (soil.py:71-469, groundwater.py:44-120, landusechange.py:55-100, leafarea.py:44-78,
surface_routing.py:43-113, opensealed), either the scalar or the map as stored (float32 / float64, whole 57 x 80 grid).
Data files held by the reference's own tests -- fixtures, not source.

  tests/golden/etrs89_bindings.npz     binding name -> scalar | [57, 80] map | [36, 57, 80] LAI stack
  tests/golden/etrs89_meteo_long.npz   pr / e0 / es / et / ta, the first NT 6-hourly fields of meteo_1950 on the land
                                       pixels of mask.map ([NT, 2847] float32 as stored)
make_golden.py `long` turns them into etrs89_long.npz by running the reference's own initial() and dynamic() methods.
"""
import os
import re
import xml.etree.ElementTree as ET

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CASE = "/root/reference/tests/data/LF_ETRS89_UseCase"
NT = 72

NAMES = """
SoilDepth1 SoilDepth1Forest SoilDepth2 SoilDepth2Forest SoilDepth3 SoilDepth3Forest
CourantCrit LeafDrainageTimeConstant AvWaterRateThreshold kdf SMaxSealed CumIntSealedInitValue
MapCropCoef MapForestCropCoef MapIrrigationCropCoef MapCropGroupNumber MapForestCropGroupNumber MapIrrigationCropGroupNumber
MapN MapForestN
MapKSat1 MapKSat1Forest MapKSat2 MapKSat2Forest MapKSat3
MapLambda1 MapLambda1Forest MapLambda2 MapLambda2Forest MapLambda3
MapGenuAlpha1 MapGenuAlpha1Forest MapGenuAlpha2 MapGenuAlpha2Forest MapGenuAlpha3
MapThetaSat1 MapThetaSat1Forest MapThetaSat2 MapThetaSat2Forest MapThetaSat3
MapThetaRes1 MapThetaRes1Forest MapThetaRes2 MapThetaRes2Forest MapThetaRes3
ThetaInit1Value ThetaInit2Value ThetaInit3Value ThetaForestInit1Value ThetaForestInit2Value ThetaForestInit3Value
ThetaIrrigationInit1Value ThetaIrrigationInit2Value ThetaIrrigationInit3Value
b_Xinanjiang PowerPrefFlow
DSLRInitValue DSLRForestInitValue DSLRIrrigationInitValue CumIntInitValue CumIntForestInitValue CumIntIrrigationInitValue
UpperZoneTimeConstant LowerZoneTimeConstant GwPercValue GwLoss LZThreshold LZInitValue UZInitValue UZForestInitValue
UZIrrigationInitValue LZAvInflowMap
ForestFraction DirectRunoffFraction WaterFraction IrrigationFraction RiceFraction OtherFraction
LAIOtherMaps LAIForestMaps LAIIrrigationMaps
OFOtherInitValue OFForestInitValue OFDirectInitValue Grad GradMin OFDepRef
""".split()


def bindings(path):
    """name -> fully substituted value string, as the reference's settings parser does (settings.py: user variables
    are replaced inside the binding values, repeatedly, until no $(...) is left)"""
    root = ET.parse(path).getroot()
    user, bind = {}, {}
    for group, dst in (("lfuser", user), ("lfbinding", bind)):
        for g in root.iter(group):
            for tv in g.iter("textvar"):
                dst[tv.get("name")] = tv.get("value")
    user["ProjectDir"] = user["ProjectPath"] = CASE
    user["SettingsPath"] = user["SettingsDir"] = os.path.dirname(path)        # built-in variables of the settings parser
    pat = re.compile(r"\$\(([^)]+)\)")

    def expand(v, depth=0):
        assert depth < 20, v
        return pat.sub(lambda m: expand(user.get(m.group(1), bind.get(m.group(1), m.group(0))), depth + 1)
                       if m.group(1) in user or m.group(1) in bind else m.group(0), v)
    return {k: expand(v) for k, v in bind.items()}


def read_map(path):
    if not path.endswith(".nc"):
        path = path + ".nc"
    with h5py.File(path, "r") as f:
        names = [k for k, v in f.items() if isinstance(v, h5py.Dataset) and v.ndim >= 2]
        ds = f[names[-1]]
        return ds[...]


def main():
    b = bindings(os.path.join(CASE, "settings", "cold.xml"))
    out, missing = {}, []
    for n in NAMES:
        v = b.get(n)
        if v is None:
            missing.append(n)
            continue
        try:
            out[n] = np.float64(float(v))
            continue
        except ValueError:
            pass
        p = os.path.normpath(v if os.path.isabs(v) else os.path.join(CASE, "settings", v))
        try:
            a = read_map(p)
        except OSError as e:
            missing.append("%s -> %s (%s)" % (n, v, str(e)[:40]))
            continue
        out[n] = a
        print("%-32s %-60s %s %s" % (n, os.path.relpath(p, CASE), a.dtype, a.shape))
    print("scalars:", {k: float(x) for k, x in out.items() if np.ndim(x) == 0})
    print("missing:", missing)
    np.savez_compressed(os.path.join(HERE, "etrs89_bindings.npz"), **out)
    print("etrs89_bindings.npz", os.path.getsize(os.path.join(HERE, "etrs89_bindings.npz")))
    mask = np.load(os.path.join(HERE, "etrs89_static.npz"))["mask_map"]
    met = {}
    for key in ("pr", "e0", "es", "et", "ta"):
        with h5py.File(os.path.join(CASE, "meteo_1950", key + ".nc"), "r") as f:
            name = [k for k, v in f.items() if isinstance(v, h5py.Dataset) and v.ndim == 3][0]
            a = f[name][:NT]
        met[key] = np.ascontiguousarray(a[:, mask])
        assert met[key].dtype == np.float32 and np.isfinite(met[key]).all(), key
    np.savez_compressed(os.path.join(HERE, "etrs89_meteo_long.npz"), **met)
    print("etrs89_meteo_long.npz", os.path.getsize(os.path.join(HERE, "etrs89_meteo_long.npz")), met["pr"].shape)


if __name__ == "__main__":
    main()
