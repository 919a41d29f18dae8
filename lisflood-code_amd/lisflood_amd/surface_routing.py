"""surface_routing -- overland-flow routing module in the reference's HydroModule shape
(src/lisflood/hydrological_modules/surface_routing.py): three kinematic-wave routers (Direct / Other / Forest)
on the LddToChan graph, one call each per model step, plus the runoff-component and storage arithmetic of
`dynamic()` (surface_routing.py:115-212) -- two element-wise device passes around the three router calls.

The three routers share ONE graph object (the reference rebuilds the same graph three times,
surface_routing.py:108-113)."""
import ctypes as C

import numpy as np

from ._lib import BufferCache, DeviceArray, check, f64, lib, u8
from .hydro_module import HydroModule
from .kinematic_wave_parallel import Graph, kinematicWave

_V_IN = "SoilFraction AvailableWaterForInfiltration Infiltration OFAlpha".split()
_N_IN = "DirectRunoff UZOutflowPixel LZOutflowToChannelPixel IsChannel".split()
_STATE = "OFQDirect OFQOther OFQForest".split()
_OUT = ("OFM3Direct OFM3Other OFM3Forest SurfaceRunoff TotalRunoff OFToChanM3 WaterDepth ToChanM3Runoff "
        "ToChanM3RunoffDt").split()


class _SurfaceArgs(C.Structure):  # lf_surface_args
    _fields_ = ([(k, C.c_void_p) for k in _V_IN + _N_IN + _STATE + _OUT + ["SurfaceRunSoil", "scratch"]] +
                [(k, C.c_double) for k in ("Beta", "MMtoM3", "M3toMM", "PixelLength", "InvPixelLength", "DtSec",
                                           "InvDtSec", "InvNoRoutSteps")] + [("N", C.c_int64)])


def _values(x):
    return np.asarray(getattr(x, "values", x))


def _scalar(x, name):
    if np.ndim(x) != 0:
        x = np.asarray(x)
        if not (x == x.flat[0]).all():
            raise NotImplementedError("%s must be uniform over the catchment in this version" % name)
        return float(x.flat[0])
    return float(x)


class surface_routing(HydroModule):
    input_files_keys = {'all': ['OFOtherInitValue', 'OFForestInitValue', 'OFDirectInitValue', 'Grad', 'GradMin',
                                'OFDepRef']}    # surface_routing.py:34-35
    module_name = 'SurfaceRouting'

    def __init__(self, surface_routing_variable, device=0):
        self.var = surface_routing_variable
        self.device = device
        self.direct_surface_router = self.other_surface_router = self.forest_surface_router = None

    def initial(self, NManning=None, Grad=None, OFDepRef=None):
        """OFAlpha and the initial overland discharges (surface_routing.py:44-95) from arrays the caller loaded
        (map loading is outside this engine): NManning [3,N] rows Other/Forest/Direct, Grad [N], OFDepRef."""
        v = self.var
        if NManning is not None:
            perim = v.PixelLength + 2 * 0.001 * OFDepRef
            v.OFAlpha = ((np.asarray(NManning) / np.sqrt(Grad)) ** v.Beta) * (perim ** (2.0 / 3.0 * v.Beta))   # :77-83
        alpha = _values(v.OFAlpha)
        runoff = list(v.dim_runoff[1])
        v.InvOFAlpha = 1 / alpha
        for name in ("Direct", "Other", "Forest"):                                                             # :93-95
            m3 = np.asarray(getattr(v, "OFM3" + name), dtype=np.float64)
            setattr(v, "OFQ" + name, (m3 * v.InvPixelLength * (1 / alpha[runoff.index(name)])) ** v.InvBeta)

    def initialSecond(self, compressed_ldd_to_chan, land_mask, flagnancheck=False):
        """Three routers on LddToChan (surface_routing.py:97-113), sharing one graph."""
        v = self.var
        dt = v.DtSec / getattr(v, "NoSubStepsOF", 1)
        alpha = _values(v.OFAlpha)
        runoff = list(v.dim_runoff[1])
        g = Graph(compressed_ldd_to_chan, land_mask)
        mk = lambda name: kinematicWave(None, None, alpha[runoff.index(name)], v.Beta, v.PixelLength, dt,
                                        flagnancheck=flagnancheck, device=self.device, graph=g)
        self.direct_surface_router, self.other_surface_router, self.forest_surface_router = (
            mk("Direct"), mk("Other"), mk("Forest"))

    def dynamic(self):
        v = self.var
        if self.direct_surface_router is None:
            raise RuntimeError("surface_routing.initialSecond() must be called first")
        if list(v.SOIL_USES) != ["Rainfed", "Forest", "Irrigated"] or list(v.dim_runoff[1]) != ["Other", "Forest", "Direct"]:
            raise NotImplementedError("only the prescribed fractions Rainfed/Forest/Irrigated are supported")
        N = self.direct_surface_router.num_pixels
        a = _SurfaceArgs()
        dev = {}
        if getattr(self, "_cache", None) is None:        # device buffers live as long as the module
            self._cache = BufferCache(self.device)
        put, get = self._cache.put, self._cache.get
        static = ("OFAlpha", "IsChannel")   # parameter maps: uploaded once (BufferCache.put_static); SoilFraction is not
        #                                     one (landusechange.py:107-139 rewrites it during a run)
        for k in _V_IN:
            dev[k] = (self._cache.put_static if k in static else put)(k, f64(_values(getattr(v, k))))
        for k in _N_IN:
            x = _values(getattr(v, k))
            dev[k] = (self._cache.put_static if k in static else put)(k, u8(x) if k == "IsChannel" else f64(np.broadcast_to(x, (N,))))
        host_state = {}
        for k in _STATE:
            host_state[k] = np.ascontiguousarray(getattr(v, k), dtype=np.float64)
            dev[k] = put(k, host_state[k])
        for k in _OUT:
            dev[k] = get(k, N)
        dev["SurfaceRunSoil"] = get("SurfaceRunSoil", (3, N))
        dev["scratch"] = get("scratch", (3, N))
        for k, d in dev.items():
            setattr(a, k, d.ptr.value)
        a.Beta = float(v.Beta)
        a.MMtoM3, a.M3toMM = _scalar(v.MMtoM3, "MMtoM3"), _scalar(v.M3toMM, "M3toMM")
        a.PixelLength, a.InvPixelLength = _scalar(v.PixelLength, "PixelLength"), _scalar(v.InvPixelLength, "InvPixelLength")
        a.DtSec, a.InvDtSec, a.InvNoRoutSteps = float(v.DtSec), float(v.InvDtSec), float(v.InvNoRoutSteps)
        a.N = N
        check(lib().lf_surface_step(self.direct_surface_router._h, self.other_surface_router._h,
                                    self.forest_surface_router._h, C.byref(a)))
        for k in _STATE:            # OFQ* are updated in place by the routers in the reference
            cur = getattr(v, k)
            if isinstance(cur, np.ndarray) and cur.dtype == np.float64 and cur.flags.c_contiguous and cur.flags.writeable:
                dev[k].download(cur)
            else:
                setattr(v, k, dev[k].download())
        for k in _OUT:
            setattr(v, k, dev[k].download())
        srs = dev["SurfaceRunSoil"].download()
        alloc = getattr(v, "allocateDataArray", None)
        if alloc is not None:
            v.SurfaceRunSoil = alloc([v.dim_landuse, v.dim_pixel])
            _values(v.SurfaceRunSoil)[...] = srs
        else:
            v.SurfaceRunSoil = srs
        v.Qall = v.OFQDirect + v.OFQOther + v.OFQForest                 # surface_routing.py:195-196
        v.M3all = v.OFM3Direct + v.OFM3Other + v.OFM3Forest
