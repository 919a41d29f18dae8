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

    def run_fused_sideflow_per_substep(self, sideflows):
        """one model step whose sub-steps each have their own sideflow vector (sideflow_stride = N): sideflows = list of
        pixel-order vectors, one per sub-step"""
        N = self.N
        side = DeviceArray.from_host(f64(np.stack([np.broadcast_to(x, (N,))[self.perm] for x in sideflows])), self.device)
        a = _SubstepArgs.from_buffer_copy(self.args)
        a.SideflowChanM3 = side.ptr.value
        check(lib().lf_routing_substeps_fused(self.router._h, C.byref(a), C.c_int(len(sideflows)), C.c_int64(N)))
        check(lib().lf_device_synchronize(C.c_int(self.device)))
        side.free()

    def run_model_steps_resident(self, nsteps, nmodel, sums):
        """timing form of run_model_steps: the resident sideflow vector for every model step, the sums into the
        caller's [nmodel, N] device array (not zeroed, not downloaded)"""
        a = _SubstepArgs.from_buffer_copy(self.args)
        a.sumDisDay = sums.ptr.value
        check(lib().lf_routing_model_steps_fused(self.router._h, C.byref(a), C.c_int(nsteps), C.c_int(nmodel), C.c_int64(0)))

    def run_model_steps(self, nsteps, sideflows=None, nmodel=None, sums0=None):
        """`nmodel` model steps of nsteps sub-steps as ONE wavefront (lf_routing_model_steps_fused).  sideflows: list of
        pixel-order sideflow vectors, one per model step (None: the resident vector for `nmodel` steps).  Returns the
        [nmodel, N] discharge sums (pixel order); the state vectors hold what the last model step leaves."""
        M = len(sideflows) if sideflows is not None else int(nmodel)
        N = self.N
        sums = DeviceArray((M, N), np.float64, self.device).zero()
        if sums0 is not None:                  # [M, N] pixel order: what the sums hold when the call starts (default +0.0)
            sums.upload(f64(np.asarray(sums0)[:, self.perm]))
        a = _SubstepArgs.from_buffer_copy(self.args)
        a.sumDisDay = sums.ptr.value
        side, stride = None, 0
        if sideflows is not None:
            side = DeviceArray.from_host(f64(np.stack([np.broadcast_to(x, (N,))[self.perm] for x in sideflows])), self.device)
            a.SideflowChanM3 = side.ptr.value
            stride = N
        check(lib().lf_routing_model_steps_fused(self.router._h, C.byref(a), C.c_int(nsteps), C.c_int(M), C.c_int64(stride)))
        out = np.empty((M, N))
        out[:, self.perm] = sums.download()
        sums.free()
        if side is not None:
            side.free()
        return out

    def download(self, name):
        out = np.empty(self.N)
        out[self.perm] = self.dev[name].download()
        return out

    def free(self):
        for d in self.dev.values():
            d.free()
        self.dev = {}
