"""Bench / test helper (not part of the product package): device-resident routing state for a whole model step
(engine order).

`RoutingStepDevice` keeps every vector of routing.dynamic (routing.py:435-706) in HBM in the router's sweep
order and runs the NoRoutSteps sub-steps either one by one (`run_sequential`, lf_routing_substep) or as one
skewed wavefront (`run_fused`, lf_routing_substeps_fused).  Host arrays are permuted once at construction and
once in `download()`."""
import ctypes as C

import numpy as np

import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "lisflood-code_amd"))
from lisflood_amd._lib import DeviceArray, check, f64, lib, u8  # noqa: E402
from lisflood_amd.routing import _OUT, _STATE, _STATIC, _SubstepArgs  # noqa: E402


class RoutingStepDevice:
    def __init__(self, router, values, split, Beta, InvDtRouting, DtSec, device=0):
        """values: dict name -> pixel-order host array for the names of lf_substep_args (missing ones = 0)."""
        self.router, self.device, self.split = router, device, bool(split)
        N = router.num_pixels
        self.N = N
        self.perm = router.graph.layout()[0].astype(np.int64)
        self.dev = {}
        zeros = np.zeros(N)
        for k in _STATIC + _STATE:
            x = values.get(k)
            if x is None:
                x = np.ones(N, bool) if k == "IsChannelKinematic" else zeros
            x = np.broadcast_to(x, (N,))[self.perm]
            self.dev[k] = DeviceArray.from_host(u8(x) if k == "IsChannelKinematic" else f64(x), device)
        for k in _OUT + ["scratch0", "scratch1"]:
            self.dev[k] = DeviceArray(N, np.float64, device).zero()
        side = np.broadcast_to(values.get("SideflowChanM3", zeros), (N,))[self.perm]
        self.dev["SideflowChanM3"] = DeviceArray.from_host(f64(side), device)
        a = self.args = _SubstepArgs()
        for k, d in self.dev.items():
            setattr(a, k, d.ptr.value)
        a.Beta, a.InvBeta, a.InvDtRouting, a.DtSec = float(Beta), 1.0 / float(Beta), float(InvDtRouting), float(DtSec)
        a.split = 1 if self.split else 0
        a.engine_order = 1

    def run_sequential(self, nsteps):
        for _ in range(nsteps):
            check(lib().lf_routing_substep(self.router._h, C.byref(self.args)))

    def run_single_sweep(self, nsteps):
        """sub-step by sub-step (structures can run in between), each as ONE level sweep over both routers"""
        for _ in range(nsteps):
            check(lib().lf_routing_substeps_fused(self.router._h, C.byref(self.args), C.c_int(1), C.c_int64(0)))

    def run_fused(self, nsteps):
        check(lib().lf_routing_substeps_fused(self.router._h, C.byref(self.args), C.c_int(nsteps), C.c_int64(0)))

    def download(self, name):
        out = np.empty(self.N)
        out[self.perm] = self.dev[name].download()
        return out

    def free(self):
        for d in self.dev.values():
            d.free()
        self.dev = {}
