"""Reference bootstrap -- TEST INFRASTRUCTURE, THIS CONTAINER ONLY.

Loads the reference's own hot-path modules *in place* from /root/reference (read-only) so that golden
vectors can be captured from the reference's code itself.  It holds
no reference code: it only fakes the third-party modules the image lacks (numba -> identity decorators,
numexpr -> eval over numpy with numexpr's caller-frame name resolution, nine, pcraster) and the parent
packages.  Nothing here travels to the GPU box (there is no /root/reference there) and nothing in the
product (lisflood-code_amd/) imports it.  Only tests/golden/make_golden.py and the reference-pinning
tests (skipped when /root/reference is absent) use it.
"""
import importlib
import os
import sys
import types
from collections import defaultdict

import numpy as np

REF = "/root/reference/src/lisflood/"


def reference_available():
    return os.path.isdir(REF)


def _catch_all(name):
    m = types.ModuleType(name)

    def __getattr__(attr):
        if attr.startswith("__"):
            raise AttributeError(attr)

        def dummy(*a, **k):
            raise RuntimeError("stub called: %s.%s" % (name, attr))
        return dummy
    m.__getattr__ = __getattr__
    sys.modules[name] = m
    return m


class _Settings:
    """Stand-in for LisSettings / EPICSettings singletons (settings.py:349, 235)."""
    options = defaultdict(bool)
    flags = {"nancheck": False}
    binding = {}
    soil_uses = ["Rainfed", "Forest", "Irrigated"]
    vegetation_landuse = {}

    @classmethod
    def instance(cls):
        return cls


class _MaskInfo:
    n = 0

    @classmethod
    def instance(cls):
        return cls

    @classmethod
    def in_zero(cls):
        return np.zeros(cls.n)


_loaded = {}


def load():
    """Returns dict(kwp=..., kwpt=..., soilloop=..., routing=..., surface=..., LisSettings=..., MaskInfo=...)."""
    if _loaded:
        return _loaded
    if not reference_available():
        raise RuntimeError("reference checkout not present at " + REF)
    import pandas  # noqa: F401  must be imported BEFORE the fake numexpr is registered

    nb = types.ModuleType("numba")

    def njit(*a, **k):
        if len(a) == 1 and callable(a[0]) and not k:
            return a[0]
        return lambda f: f
    nb.njit, nb.prange = njit, range
    nb.vectorize = lambda *a, **k: (lambda f: np.vectorize(f))
    sys.modules["numba"] = nb

    nx = types.ModuleType("numexpr")

    def evaluate(expr, local_dict=None, global_dict=None):
        f = sys._getframe(1)
        d = dict(f.f_globals if global_dict is None else global_dict)
        d.update(f.f_locals if local_dict is None else local_dict)
        return eval(expr, {"__builtins__": {}}, d)
    nx.evaluate = evaluate
    sys.modules["numexpr"] = nx

    nine = _catch_all("nine")
    nine.range = range
    nine.iteritems = lambda d: d.items()
    for n in ("pcraster", "pcraster.framework", "pcraster.operations", "lisflood.global_modules.add1"):
        _catch_all(n)

    def _pkg(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m
    _pkg("lisflood", REF)
    _pkg("lisflood.global_modules", REF + "global_modules")
    hm = _pkg("lisflood.hydrological_modules", REF + "hydrological_modules")
    hm.HydroModule = type("HydroModule", (object,), {})

    st = types.ModuleType("lisflood.global_modules.settings")
    st.LisSettings, st.MaskInfo, st.EPICSettings = _Settings, _MaskInfo, _Settings
    sys.modules["lisflood.global_modules.settings"] = st

    # numpy >= 1.24 dropped np.bool8 / np.int etc. that the 2024 reference still uses in routing.py
    for alias, target in (("bool8", np.bool_),):
        if not hasattr(np, alias):
            setattr(np, alias, target)

    kwp = importlib.import_module("lisflood.hydrological_modules.kinematic_wave_parallel")
    soil = importlib.import_module("lisflood.hydrological_modules.soilloop")
    rout = importlib.import_module("lisflood.hydrological_modules.routing")
    surf = importlib.import_module("lisflood.hydrological_modules.surface_routing")
    extra = {}
    sys.modules.setdefault("xarray", types.ModuleType("xarray"))    # soil.py imports it, the methods used here do not
    for name in ("lakes", "reservoir", "inflow", "transmission", "soil", "groundwater", "opensealed"):
        extra[name] = importlib.import_module("lisflood.hydrological_modules." + name)
    _loaded.update(kwp=kwp, kwpt=kwp.kwpt, soilloop=soil, routing=rout, surface=surf,
                   LisSettings=_Settings, MaskInfo=_MaskInfo, **extra)
    return _loaded


if __name__ == "__main__":
    m = load()
    print("ok", m["kwp"].kinematicWave, m["soilloop"].soilColumnsWaterBalance, m["routing"].routing,
          m["surface"].surface_routing)
