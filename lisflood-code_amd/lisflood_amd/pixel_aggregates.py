"""The per-pixel aggregates between the soil columns and surface routing -- opensealed.dynamic
(opensealed.py:40-71), soil.dynamic_perpixel (soil.py:471-514) and groundwater.dynamic
(groundwater.py:134-180) -- as one device pass over `var` (same attribute names as the reference)."""
import ctypes as C

import numpy as np

from ._lib import BufferCache, check, f64, lib

_V_IN = ("SoilFraction TaInterception Ta ESAct PrefFlow Infiltration SeepTopToSubA SeepTopToSubB SeepSubToGW Theta1a "
         "Theta1b Theta2 W1a W1b W2 UZOutflow GwPercUZLZ SoilDepthTotal").split()
_N_IN = "Rain SnowMelt EWRef SMaxSealed DirectRunoffFraction WaterFraction LowerZoneK LZThreshold GwLossStep".split()
_STATE = "CumInterSealed LZ LZInflowCUM TaInterceptionCUM TaCUM ESActCUM GwLossCUM".split()
_OUT = ("RainSnowmelt EWaterAct InterSealed TASealed DirectRunoff TaInterceptionAll TaPixel ESActPixel PrefFlowPixel "
        "InfiltrationPixel ThetaAll SeepTopToSubPixelA SeepTopToSubPixelB SeepSubToGWPixel Theta1aPixel Theta1bPixel "
        "Theta2Pixel LZOutflow UZOutflowPixel GwPercUZLZPixel GwLossLZ LZAvInflow LZOutflowToChannelPixel").split()


class _PixelArgs(C.Structure):  # lf_pixel_args, include/lisflood_amd.h
    _fields_ = ([(k, C.c_void_p) for k in _V_IN + _N_IN + _STATE + _OUT + ["Theta"]] +
                [("InvDtDay", C.c_double), ("TimeSinceStart", C.c_double), ("N", C.c_int64)])


def _values(x):
    return np.asarray(getattr(x, "values", x))


def dynamic(var, device=0):
    """opensealed.dynamic(); soil.dynamic_perpixel(); groundwater.dynamic() -- in that order, in place on `var`."""
    v = var
    N = _values(v.SoilFraction).shape[1]
    if _values(v.SoilFraction).shape[0] != 3:
        raise NotImplementedError("only the three prescribed fractions are supported")
    a = _PixelArgs()
    dev = {}
    cache = getattr(v, "_lf_pixel_buffers", None)       # device buffers live as long as `var` does
    if cache is None or cache.device != device:
        cache = v._lf_pixel_buffers = BufferCache(device)
    # parameter maps: uploaded once (BufferCache.put_static, content checksum).  NOT the land-use fractions
    # (SoilFraction, DirectRunoffFraction, WaterFraction): the reference rewrites them during a run
    # (landusechange.py:107-139, evapowater.py:108-119), they are staged on every call
    static = ("SoilDepthTotal", "SMaxSealed", "LowerZoneK", "LZThreshold", "GwLossStep")
    for k in _V_IN:
        dev[k] = (cache.put_static if k in static else cache.put)(k, f64(_values(getattr(v, k))))
    for k in _N_IN + _STATE:
        x = f64(np.broadcast_to(_values(getattr(v, k)), (N,)))
        dev[k] = (cache.put_static if (k in static and np.ndim(_values(getattr(v, k))) == 1) else cache.put)(k, x)
    for k in _OUT:
        dev[k] = cache.get(k, N)
    dev["Theta"] = cache.get("Theta", (3, N))
    for k, d in dev.items():
        setattr(a, k, d.ptr.value)
    a.InvDtDay, a.TimeSinceStart, a.N = float(v.InvDtDay), float(v.TimeSinceStart), N
    check(lib().lf_pixel_aggregates_device(C.c_int(device), C.byref(a)))
    for k in _STATE + _OUT:
        setattr(v, k, dev[k].download())
    th = dev["Theta"].download()
    cur = getattr(v, "Theta", None)
    if cur is not None and _values(cur).shape == th.shape:
        _values(cur)[...] = th
    else:
        v.Theta = th
    v.TaInterceptionWB, v.TaWB, v.ESActWB = v.TaInterceptionAll, v.TaPixel, v.ESActPixel      # soil.py:477,484,489
    v.LZOutflowToChannel, v.GwLossPixel, v.GwLossWB = v.LZOutflow, v.GwLossLZ, v.GwLossLZ       # groundwater.py:141,170,172
