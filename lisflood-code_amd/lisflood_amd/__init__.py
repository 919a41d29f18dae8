"""lisflood_amd -- MI355X (gfx950) engine for LISFLOOD's per-timestep routing / soil hot path.

Host side is plain Python + numpy calling hand-written HIP kernels through the C ABI declared in
include/lisflood_amd.h (ctypes, no PyTorch).  The modules mirror the reference's
src/lisflood/hydrological_modules layout for this path:

    kinematic_wave_parallel.kinematicWave      <- kinematic_wave_parallel.py:114-184
    soilloop.interception_water_balance / soilColumnsWaterBalance / soilloop  <- soilloop.py
    routing.routing, surface_routing.surface_routing  (HydroModule-shaped)    <- routing.py, surface_routing.py

There is no CPU fallback: importing a compute module without the built HIP library raises.
"""
from ._lib import LisfloodAmdError, lib, library_path  # noqa: F401

__version__ = "0.1.0"
