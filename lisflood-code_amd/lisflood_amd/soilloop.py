"""Soil / vegetation column kernels -- drop-ins for the numba functions of the reference's
src/lisflood/hydrological_modules/soilloop.py, executed by csrc/lf_soil.hip on MI355X:

    interception_water_balance(...)   soilloop.py:27-70     same 8 positional arguments
    potentialTranspiration(...)       soilloop.py:73-75
    soilColumnsWaterBalance(...)      soilloop.py:78-355    same 73 positional arguments

All array arguments are caller-owned numpy buffers in the reference's layout ([V,N] / [L,N] C-order
fp64, bool masks); the written ones are updated IN PLACE and the functions return None, as the numba
kernels do.  `SoilColumnsDevice` is the device-resident variant (state stays in HBM between steps).
"""
import ctypes as C

import numpy as np

from ._lib import DeviceArray, check, f64, lib, ptr, u8

_L_FIELDS = ("PoreSpaceNotZero1a PoreSpaceNotZero1b PoreSpaceNotZero2 KSat1a KSat1b KSat2 GenuInvM1a GenuInvM1b "
             "GenuInvM2 GenuM1a GenuM1b GenuM2 WRes1a WRes1b WRes1 WRes2 WWP1a WWP1b WWP1 WWP2 WFC1a WFC1b WFC1 WFC2 "
             "SoilDepth1a SoilDepth1b SoilDepth2 WS1a WS1b WS1 WS2 StoreMaxPervious").split()
_N_FIELDS = "Rain SnowMelt b_Xinanjiang PowerInfPot PowerPrefFlow UpperZoneK GwPercStep isFrozenSoil".split()
_V_IN = "LeafDrainage Interception ESMax".split()
_V_IO = ("AvailableWaterForInfiltration DSLR ESAct PrefFlow Infiltration W1a W1b W1 W2 Theta1a Theta1b Theta2 "
         "Sat1a Sat1b Sat1 Sat2 SeepTopToSubA SeepTopToSubB SeepSubToGW UZOutflow UZ GwPercUZLZ").split()
_SMALL = "index_landuse_all is_irrigated is_paddy_irrig paddy_inactive paddy_any".split()
_BOOL = {"PoreSpaceNotZero1a", "PoreSpaceNotZero1b", "PoreSpaceNotZero2", "isFrozenSoil"}

# positional order of the reference signature (soilloop.py:79-99)
ARG_ORDER = (
    "index_landuse_all is_irrigated is_paddy_irrig paddy_inactive DtDay AvailableWaterForInfiltration Rain "
    "SnowMelt LeafDrainage Interception DSLR AvWaterThreshold ESAct ESMax isFrozenSoil b_Xinanjiang "
    "StoreMaxPervious PowerInfPot PrefFlow PowerPrefFlow Infiltration CourantCrit PoreSpaceNotZero1a "
    "PoreSpaceNotZero1b PoreSpaceNotZero2 KSat1a KSat1b KSat2 GenuInvM1a GenuInvM1b GenuInvM2 GenuM1a GenuM1b "
    "GenuM2 W1a W1b W1 W2 Theta1a Theta1b Theta2 Sat1a Sat1b Sat1 Sat2 SeepTopToSubA SeepTopToSubB SeepSubToGW "
    "WRes1a WRes1b WRes1 WRes2 WWP1a WWP1b WWP1 WWP2 WFC1a WFC1b WFC1 WFC2 SoilDepth1a SoilDepth1b SoilDepth2 "
    "WS1a WS1b WS1 WS2 UpperZoneK DrainedFraction GwPercStep UZOutflow UZ GwPercUZLZ").split()


class _SoilArgs(C.Structure):  # lf_soil_args, include/lisflood_amd.h
    _fields_ = ([(k, C.c_void_p) for k in _L_FIELDS + _N_FIELDS + _V_IN + _V_IO + _SMALL] +
                [("DtDay", C.c_double), ("AvWaterThreshold", C.c_double), ("CourantCrit", C.c_double),
                 ("DrainedFraction", C.c_double), ("V", C.c_int64), ("L", C.c_int64), ("N", C.c_int64)])


class _InterceptionArgs(C.Structure):  # lf_interception_args
    _fields_ = [(k, C.c_void_p) for k in ("Interception", "TaInterception", "LeafDrainage", "CumInterception", "LAI",
                                          "Rain", "TaInterceptionMax")] + [
        ("drainageK", C.c_double), ("V", C.c_int64), ("N", C.c_int64)]


def _inplace(a, name):
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags.c_contiguous and a.flags.writeable):
        raise ValueError("%s must be a writeable C-contiguous float64 array (it is updated in place)" % name)
    return a


def interception_water_balance(Interception, TaInterception, LeafDrainage, CumInterception, LAI, Rain,
                               TaInterceptionMax, drainageK, device=0):
    a = _InterceptionArgs()
    V, N = Interception.shape
    keep = [_inplace(Interception, "Interception"), _inplace(TaInterception, "TaInterception"),
            _inplace(LeafDrainage, "LeafDrainage"), _inplace(CumInterception, "CumInterception"),
            f64(LAI), f64(Rain), f64(TaInterceptionMax)]
    for k, arr in zip(("Interception", "TaInterception", "LeafDrainage", "CumInterception", "LAI", "Rain",
                       "TaInterceptionMax"), keep):
        setattr(a, k, arr.ctypes.data)
    a.drainageK, a.V, a.N = float(drainageK), V, N
    check(lib().lf_interception_host(C.c_int(device), C.byref(a)))


def potentialTranspiration(TranspirMax, TaInterception):
    """soilloop.py:73-75 -- one element-wise maximum; kept on the host (it is not a kernel worth a launch)."""
    return np.maximum(TranspirMax - TaInterception, 0)


def _fill_small(a, keep, d, V):
    idx = np.ascontiguousarray(d["index_landuse_all"], dtype=np.int64)
    irr = u8(d["is_irrigated"])
    pad = u8(d["is_paddy_irrig"])
    keep += [idx, irr, pad]
    a.index_landuse_all, a.is_irrigated, a.is_paddy_irrig = idx.ctypes.data, irr.ctypes.data, pad.ctypes.data
    return int(pad.sum())


def soilColumnsWaterBalance(*args, device=0):
    if len(args) != len(ARG_ORDER):
        raise TypeError("soilColumnsWaterBalance takes %d positional arguments (%d given)" % (len(ARG_ORDER), len(args)))
    d = dict(zip(ARG_ORDER, args))
    a = _SoilArgs()
    keep = []
    V, N = d["Interception"].shape
    L = np.asarray(d["WS1a"]).shape[0]
    for k in _V_IO:
        arr = _inplace(d[k], k)
        setattr(a, k, arr.ctypes.data)
    for k in _L_FIELDS + _N_FIELDS + _V_IN:
        arr = u8(d[k]) if k in _BOOL else f64(d[k])
        keep.append(arr)
        setattr(a, k, arr.ctypes.data)
    n_paddy = _fill_small(a, keep, d, V)
    if n_paddy:
        pi = u8(d["paddy_inactive"])
        keep.append(pi)
        a.paddy_inactive = pi.ctypes.data
    a.DtDay, a.AvWaterThreshold = float(d["DtDay"]), float(d["AvWaterThreshold"])
    a.CourantCrit, a.DrainedFraction = float(d["CourantCrit"]), float(d["DrainedFraction"])
    a.V, a.L, a.N = V, L, N
    check(lib().lf_soil_columns_host(C.c_int(device), C.byref(a)))


class SoilColumnsDevice:
    """Device-resident soil columns: every array of the reference call lives in HBM; `step()` is one
    soilColumnsWaterBalance pass, `interception()` one interception_water_balance pass."""

    def __init__(self, d, device=0):
        self.device = device
        self.V, self.N = d["Interception"].shape
        self.L = np.asarray(d["WS1a"]).shape[0]
        self.dev = {}
        for k in _L_FIELDS + _N_FIELDS + _V_IN + _V_IO:
            self.dev[k] = DeviceArray.from_host(u8(d[k]) if k in _BOOL else f64(d[k]), device)
        self._keep = []
        self.args = _SoilArgs()
        for k, v in self.dev.items():
            setattr(self.args, k, v.ptr.value)
        n_paddy = _fill_small(self.args, self._keep, d, self.V)
        if n_paddy:
            pi = u8(d["paddy_inactive"])
            self.dev["paddy_inactive"] = DeviceArray.from_host(pi, device)
            self.args.paddy_inactive = self.dev["paddy_inactive"].ptr.value
            any_ = np.ascontiguousarray(pi.reshape(-1, self.N).any(axis=1).astype(np.uint8))
            self._keep.append(any_)
            self.args.paddy_any = any_.ctypes.data
        a = self.args
        a.DtDay, a.AvWaterThreshold = float(d["DtDay"]), float(d["AvWaterThreshold"])
        a.CourantCrit, a.DrainedFraction = float(d["CourantCrit"]), float(d["DrainedFraction"])
        a.V, a.L, a.N = self.V, self.L, self.N

    def step(self):
        check(lib().lf_soil_columns_device(C.c_int(self.device), C.byref(self.args)))

    def get(self, name):
        out = self.dev[name].download()
        return out

    def set(self, name, value):
        self.dev[name].upload(u8(value) if name in _BOOL else f64(value))
