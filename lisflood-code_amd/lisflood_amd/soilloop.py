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

from ._lib import BufferCache, DeviceArray, check, f64, lib, ptr, u8

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


def derived_parameters_hold(d):
    """True when the ten derived parameter arrays of soilColumnsWaterBalance are, bit for bit, what soil.py:180-228 makes
    them from the others (GenuInvM = 1 / GenuM; WS1 = WS1a + WS1b, likewise WRes1, WFC1, WWP1; PoreSpaceNotZero = SoilDepth
    != 0 and WS != 0) -- then lf_soil_columns_device_derived recomputes them instead of reading them (59 B per column).
    One pass over host arrays at set-up time; any array that is missing or different -> False (the streamed form)."""
    try:
        same = lambda a, b: np.array_equal(np.asarray(a, np.float64), np.asarray(b, np.float64))
        with np.errstate(all="ignore"):
            ok = all(same(d["GenuInvM" + l], 1 / np.asarray(d["GenuM" + l], np.float64)) for l in ("1a", "1b", "2"))
            ok = ok and all(same(d[k + "1"], np.asarray(d[k + "1a"], np.float64) + np.asarray(d[k + "1b"], np.float64))
                            for k in ("WS", "WRes", "WFC", "WWP"))
            ok = ok and all(np.array_equal(np.asarray(d["PoreSpaceNotZero" + l]) != 0,
                                           (np.asarray(d["SoilDepth" + l]) != 0) & (np.asarray(d["WS" + l]) != 0))
                            for l in ("1a", "1b", "2"))
        return bool(ok)
    except KeyError:
        return False


class SoilColumnsDevice:
    """Device-resident soil columns: every array of the reference call lives in HBM; `step()` is one
    soilColumnsWaterBalance pass, `interception()` one interception_water_balance pass."""

    def __init__(self, d, device=0):
        self.device = device
        self.V, self.N = d["Interception"].shape
        self.L = np.asarray(d["WS1a"]).shape[0]
        self.derived = derived_parameters_hold(d)      # checked once: the parameter arrays are static (set() refuses them)
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
        fn = lib().lf_soil_columns_device_derived if self.derived else lib().lf_soil_columns_device
        check(fn(C.c_int(self.device), C.byref(self.args)))

    def substep_histogram(self, nbins=128):
        """hist[k] = columns of the last step that needed k Courant sub-steps, k >= 2 (columns with one sub-step are not
        listed; the engine keeps trip counts up to 127, the last bin holds the rest) -- lf_soil_substep_histogram"""
        hist = (C.c_int64 * nbins)()
        check(lib().lf_soil_substep_histogram(C.c_int(self.device), hist, C.c_int(nbins)))
        return np.array(hist[:], dtype=np.int64)

    def get(self, name):
        out = self.dev[name].download()
        return out

    def set(self, name, value):
        if name in _L_FIELDS:
            self.derived = False        # a parameter array changed: back to the streamed form (nothing re-checks it)
        self.dev[name].upload(u8(value) if name in _BOOL else f64(value))


# =================================================================================================
# Module class: soilloop(HydroModule) -- same shape as the reference class (soilloop.py:438-722)
# =================================================================================================
from .hydro_module import HydroModule  # noqa: E402

_CANOPY_IO = "Interception TaInterception LeafDrainage CumInterception potential_transpiration RWS Ta W1a W1b W1".split()
_CANOPY_V_IN = "LAI LAITerm".split()
_CANOPY_L_IN = "CropCoef CropGroupNumber WFC1 WFC1a WFC1b WWP1 WWP1a WWP1b".split()
_CANOPY_N_IN = "Rain EWRef ETRef isFrozenSoil".split()


class _CanopyArgs(C.Structure):  # lf_canopy_args
    _fields_ = ([(k, C.c_void_p) for k in _CANOPY_IO + _CANOPY_V_IN + _CANOPY_L_IN + _CANOPY_N_IN] +
                [("index_landuse", C.c_void_p), ("LeafDrainageK", C.c_double), ("DtDay", C.c_double),
                 ("InvDtDay", C.c_double), ("V", C.c_int64), ("L", C.c_int64), ("N", C.c_int64),
                 ("SoilMoistureStressDays", C.c_void_p), ("WFilla", C.c_void_p), ("WFillb", C.c_void_p),
                 ("WPF3a", C.c_void_p), ("WPF3b", C.c_void_p), ("irrigated_veg", C.c_int64)])


def _values(x):
    """`.values` of the reference's NumpyModified / xarray containers, or the array itself."""
    return np.asarray(getattr(x, "values", x))


class soilloop(HydroModule):
    """Soil/vegetation loop for the three prescribed fractions.  `dynamic_canopy()` and `dynamic_soil()` read and
    write the same `var` attributes as the reference methods; each is one device pass.
    Option branches: options={"repStressDays": True} writes `SoilMoistureStressDays` (soilloop.py:597-598),
    {"wateruse": True} sets `WFilla` / `WFillb` of the irrigated fraction (:582-587), {"simulatePF": True} produces
    `pF0..2`.  Not produced: the `cropsEPIC` rows (the EPIC module is not part of the reference checkout)."""
    input_files_keys = {'wateruse': []}
    module_name = 'SoilLoop'

    def __init__(self, soilloop_variable, device=0, options=None):
        """options: the reference's option switches this module reads (`simulatePF`, `wateruse`, `repStressDays`)"""
        self.var = soilloop_variable
        self.device = device
        self.options = dict(options or {})
        self._cache = BufferCache(device)      # device buffers live as long as the module (no hipMalloc per call)

    def initial(self):
        v = self.var
        # soilloop.py:462-472
        self.index_landuse_all = np.array([v.SOIL_USES.index(v.VEGETATION_LANDUSE[x]) for x in v.vegetation], np.int64)
        self.index_landuse_prescr = np.array([v.SOIL_USES.index(v.VEGETATION_LANDUSE[x])
                                              for x in v.PRESCRIBED_VEGETATION], np.int64)
        self.is_irrigated = np.array([v.VEGETATION_LANDUSE[x] == 'Irrigated' for x in v.vegetation])
        self.is_paddy_irrig = np.zeros(len(v.vegetation), bool)
        if list(v.vegetation) != list(v.prescribed_vegetation):
            raise NotImplementedError("only the prescribed vegetation fractions are supported (no EPIC crops)")

    def _inplace_rows(self, name):
        a = _values(getattr(self.var, name))
        return _inplace(a, name)

    def dynamic_canopy(self):
        v = self.var
        a = _CanopyArgs()
        dev, host = {}, {}
        put = self._cache.put
        V, N = _values(v.Interception).shape
        for k in _CANOPY_IO:
            host[k] = self._inplace_rows(k)
            dev[k] = put("canopy." + k, host[k])
        for k in _CANOPY_V_IN:
            dev[k] = put("canopy." + k, f64(_values(getattr(v, k))))
        for k in _CANOPY_L_IN:          # crop / soil parameter maps: uploaded once (fingerprint check, BufferCache.put_static)
            dev[k] = self._cache.put_static("canopy." + k, f64(_values(getattr(v, k))))
        for k in _CANOPY_N_IN:
            x = _values(getattr(v, k))
            dev[k] = put("canopy." + k, u8(x) if k == "isFrozenSoil" else f64(np.broadcast_to(x, (N,))))
        for k, d in dev.items():
            setattr(a, k, d.ptr.value)
        idx = np.ascontiguousarray(self.index_landuse_prescr, dtype=np.int64)
        a.index_landuse = idx.ctypes.data
        a.LeafDrainageK, a.DtDay, a.InvDtDay = float(v.LeafDrainageK), float(v.DtDay), float(v.InvDtDay)
        a.V, a.L, a.N = V, _values(v.WFC1).shape[0], N
        a.irrigated_veg = -1
        stress = fill = None
        if self.options.get("repStressDays"):                  # soilloop.py:597-598
            stress = self._inplace_rows("SoilMoistureStressDays")
            a.SoilMoistureStressDays = self._cache.get("canopy.SoilMoistureStressDays", (V, N)).ptr.value
        if self.options.get("wateruse"):                       # soilloop.py:582-587: the "Irrigated" fraction
            irrigated = [i for i, veg in enumerate(v.prescribed_vegetation) if v.VEGETATION_LANDUSE[veg] == "Irrigated"]
            if irrigated:
                a.irrigated_veg = irrigated[-1]
                fill = (self._cache.get("canopy.WFilla", (N,)), self._cache.get("canopy.WFillb", (N,)))
                a.WFilla, a.WFillb = fill[0].ptr.value, fill[1].ptr.value
                a.WPF3a = self._cache.put_static("canopy.WPF3a", f64(_values(v.WPF3a))).ptr.value
                a.WPF3b = self._cache.put_static("canopy.WPF3b", f64(_values(v.WPF3b))).ptr.value
        check(lib().lf_canopy_device(C.c_int(self.device), C.byref(a)))
        for k in _CANOPY_IO:
            dev[k].download(host[k])
        if stress is not None:
            self._cache.buf["canopy.SoilMoistureStressDays"].download(stress)
        if fill is not None:
            v.WFilla, v.WFillb = fill[0].download(), fill[1].download()

    def dynamic_soil(self):
        v = self.var
        N = _values(v.Interception).shape[1]
        # ESMax = ESRef * LAITerm (soilloop.py:638)
        es = self._cache.put("soil.ESRef", f64(np.broadcast_to(_values(v.ESRef), (N,))))
        lt = self._cache.put("soil.LAITerm", f64(_values(v.LAITerm)))
        V = _values(v.LAITerm).shape[0]
        out = self._cache.get("soil.ESMax", (V, N))
        check(lib().lf_scale_rows_device(C.c_int(self.device), es.ptr, lt.ptr, out.ptr, C.c_int64(V), C.c_int64(N)))
        # soilColumnsWaterBalance (soilloop.py:645-665) on cached device buffers: the static parameter maps ([L,N] soil
        # hydraulic parameters and the five [N] constants) are uploaded once, the state and forcing every call -- other
        # modules may have changed them on `var` --, the 22 written arrays come back in place
        g = lambda k: _values(getattr(v, k))
        a = _SoilArgs()
        cache = self._cache
        written = {}
        for k in _V_IO:
            written[k] = _inplace(g(k), k)
            setattr(a, k, cache.put("soil." + k, written[k]).ptr.value)
        static_n = ("b_Xinanjiang", "PowerInfPot", "PowerPrefFlow", "UpperZoneK", "GwPercStep")
        for k in _L_FIELDS + _N_FIELDS:
            x = g(k)
            if k in _N_FIELDS:
                x = np.broadcast_to(x, (N,))
            x = u8(x) if k in _BOOL else f64(x)
            d = cache.put_static("soil." + k, x) if (k in _L_FIELDS or k in static_n) else cache.put("soil." + k, x)
            setattr(a, k, d.ptr.value)
        a.LeafDrainage = cache.put("soil.LeafDrainage", f64(g("LeafDrainage"))).ptr.value
        a.Interception = cache.put("soil.Interception", f64(g("Interception"))).ptr.value
        a.ESMax = out.ptr.value                              # already on the device
        keep = []
        _fill_small(a, keep, dict(index_landuse_all=self.index_landuse_all, is_irrigated=self.is_irrigated,
                                  is_paddy_irrig=self.is_paddy_irrig), V)      # no paddy rows: paddy_inactive unused (:644)
        a.DtDay, a.AvWaterThreshold = float(v.DtDay), float(v.AvWaterThreshold)
        a.CourantCrit, a.DrainedFraction = float(v.CourantCrit), float(v.DrainedFraction)
        a.V, a.L, a.N = V, g("WS1a").shape[0], N
        check(lib().lf_soil_columns_device(C.c_int(self.device), C.byref(a)))
        for k, host in written.items():
            cache.buf["soil." + k].download(host)
        if self.options.get("simulatePF"):
            self.soil_pf()

    _PF_V = "W1a W1b W2".split()
    _PF_L = ("WRes1a WRes1b WRes2 WS1a WS1b WS2 PoreSpaceNotZero1a PoreSpaceNotZero1b PoreSpaceNotZero2 GenuInvAlpha1a "
             "GenuInvAlpha1b GenuInvAlpha2 GenuInvM1a GenuInvM1b GenuInvM2 GenuInvN1a GenuInvN1b GenuInvN2").split()

    def soil_pf(self):
        """suctionUnsaturatedSoilPF (soilloop.py:673-704, option simulatePF): var.pF0 / pF1 / pF2 from the soil moisture"""
        v = self.var
        a = _SoilPfArgs()
        V, N = _values(v.W1a).shape
        outs = {}
        for k in ("pF0", "pF1", "pF2"):
            cur = getattr(v, k, None)
            if cur is None:
                cur = np.zeros((V, N))
                setattr(v, k, cur)
            outs[k] = _inplace(_values(cur), k)
            setattr(a, k, self._cache.get("pf." + k, (V, N)).ptr.value)
        for k in self._PF_V + self._PF_L:
            x = _values(getattr(v, k))
            setattr(a, k, self._cache.put("pf." + k, u8(x) if k.startswith("Pore") else f64(x)).ptr.value)
        idx = np.ascontiguousarray(self.index_landuse_all, dtype=np.int64)
        a.index_landuse_all = idx.ctypes.data
        a.HeadMax = float(v.HeadMax)
        a.V, a.L, a.N = V, _values(v.WS1a).shape[0], N
        check(lib().lf_soil_pf_device(C.c_int(self.device), C.byref(a)))
        for k in outs:
            self._cache.buf["pf." + k].download(outs[k])


class _SoilPfArgs(C.Structure):  # lf_soil_pf_args
    _fields_ = ([(k, C.c_void_p) for k in ["pF0", "pF1", "pF2"] + soilloop._PF_V + soilloop._PF_L] +
                [("index_landuse_all", C.c_void_p), ("HeadMax", C.c_double), ("V", C.c_int64), ("L", C.c_int64),
                 ("N", C.c_int64)])
