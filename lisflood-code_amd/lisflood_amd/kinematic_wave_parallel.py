"""kinematicWave -- drop-in for the reference class of the same name
(src/lisflood/hydrological_modules/kinematic_wave_parallel.py:114-184), running on MI355X.

Same constructor arguments, same `kinematicWaveRouting(discharge, specific_lateral_inflow, section)`
contract (discharge is mutated in place, returns None, a bad section raises Exception, optional
one-shot NaN/Inf warning), same public attributes (downstream_lookup, upstream_lookup,
num_upstream_pixels, pixels_ordered, order_start_stop).  The solve itself is the hand-written HIP
level sweep of csrc/lf_router.hip, called through the C ABI.
"""
import ctypes as C
import warnings

import numpy as np

from . import _lib
from ._lib import DeviceArray, check, f64, lib, ptr, u8

try:  # inside the reference tree the warning class lives here (global_modules/errors.py:42-53)
    from lisflood.global_modules.errors import LisfloodWarning  # type: ignore
except Exception:  # stand-alone
    class LisfloodWarning(Warning):
        pass


class Graph:
    """LDD -> adjacency -> routing orders (lf_graph); host memory only."""

    def __init__(self, compressed_encoded_ldd=None, land_mask=None, ldd_raster=None, virtual_down=None):
        """virtual_down ([N] int, -1 = none; compressed form only): for the pits that structures.py:44-61 cut just
        upstream of a lake / reservoir, the pixel they drain into in the uncut LDD -- they are put on that pixel's
        level (lf_graph_create_ex), which the fused sub-step wavefront with structures needs."""
        self._h = C.c_void_p()
        self.has_links = virtual_down is not None
        if ldd_raster is not None:
            r = np.ascontiguousarray(ldd_raster, dtype=np.uint8)
            H, W = r.shape
            m = None if land_mask is None else u8(land_mask)
            check(lib().lf_graph_create_raster(ptr(r), ptr(m), C.c_int(H), C.c_int(W), C.byref(self._h)))
        else:
            land_mask = np.asarray(land_mask)
            H, W = land_mask.shape
            codes = f64(compressed_encoded_ldd)
            m = u8(land_mask)
            if codes.size != int(m.sum()):
                raise ValueError("compressed LDD has %d values but the land mask has %d land cells"
                                 % (codes.size, int(m.sum())))
            vd = None if virtual_down is None else np.ascontiguousarray(virtual_down, dtype=np.int64)
            if vd is not None and vd.size != codes.size:
                raise ValueError("virtual_down needs one entry per land pixel")
            check(lib().lf_graph_create_ex(ptr(codes), ptr(m), C.c_int(H), C.c_int(W), ptr(vd), C.byref(self._h)))
        self.shape = (H, W)
        self.num_pixels = int(lib().lf_graph_num_pixels(self._h))
        self.num_levels = int(lib().lf_graph_num_levels(self._h))
        self.max_upstream = int(lib().lf_graph_max_upstream(self._h))

    def lookups(self):
        N, K = self.num_pixels, self.max_upstream
        down = np.empty(N, np.float64)
        ups = np.empty((N, K), np.int64)
        nups = np.empty(N, np.int64)
        check(lib().lf_graph_get_lookups(self._h, ptr(down), ptr(ups), ptr(nups)))
        return down, ups, nups

    def orders(self):
        po = np.empty(self.num_pixels, np.int64)
        ss = np.empty((self.num_levels, 2), np.int64)
        check(lib().lf_graph_get_orders(self._h, ptr(po), ptr(ss)))
        return po, ss

    def links(self):
        """[N] bool by engine position: zero-length structure links (Graph(virtual_down=...))."""
        out = np.zeros(self.num_pixels, np.uint8)
        check(lib().lf_graph_get_links(self._h, ptr(out)))
        return out.astype(bool)

    def layout(self):
        perm = np.empty(self.num_pixels, np.int32)
        ups_ptr = np.empty(self.num_pixels + 1, np.int32)
        level_start = np.empty(self.num_levels + 1, np.int64)
        check(lib().lf_graph_get_layout(self._h, ptr(perm), ptr(ups_ptr), ptr(level_start)))
        return perm, ups_ptr, level_start

    def close(self):
        if self._h:
            lib().lf_graph_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class kinematicWave:
    """See module docstring.  Extra keyword `device` selects the GPU (default 0)."""

    def __init__(self, compressed_encoded_ldd, land_mask, alpha_channel, beta, space_delta, time_delta,
                 alpha_floodplains=None, flagnancheck=False, device=0, graph=None):
        self.kinematic_wave_warning_printed = False
        self.flagnancheck = flagnancheck
        self.device = device
        self.space_delta = space_delta
        self.beta = beta
        self.inv_beta = 1 / beta
        self.b_minus_1 = beta - 1
        self.graph = graph if graph is not None else Graph(compressed_encoded_ldd, land_mask)
        N = self.graph.num_pixels
        self.num_pixels = N
        alpha = f64(np.broadcast_to(alpha_channel, (N,)))
        if np.ndim(space_delta) == 0:
            dx, dxs = None, float(space_delta)
        else:
            dx, dxs = f64(space_delta), 0.0
        a2 = None if alpha_floodplains is None else f64(np.broadcast_to(alpha_floodplains, (N,)))
        self._h = C.c_void_p()
        check(lib().lf_router_create(self.graph._h, ptr(alpha), C.c_double(beta), ptr(dx), C.c_double(dxs),
                                     C.c_double(time_delta), ptr(a2), C.c_int(device), C.byref(self._h)))
        self._lookups = None
        self._orders = None

    # --- the reference's public attributes, materialised lazily (init-time only data) ---------------
    def _lk(self):
        if self._lookups is None:
            self._lookups = self.graph.lookups()
        return self._lookups

    @property
    def downstream_lookup(self):
        return self._lk()[0]

    @property
    def upstream_lookup(self):
        return self._lk()[1]

    @property
    def num_upstream_pixels(self):
        return self._lk()[2]

    @property
    def pixels_ordered(self):
        if self._orders is None:
            self._orders = self.graph.orders()
        return self._orders[0]

    @property
    def order_start_stop(self):
        if self._orders is None:
            self._orders = self.graph.orders()
        return self._orders[1]

    # --- routing -----------------------------------------------------------------------------------
    @staticmethod
    def _section(section):
        if section not in _lib.SECTION:
            raise Exception("The section parameter must be either 'main_channel' or 'floodplain'!")
        return _lib.SECTION[section]

    def kinematicWaveRouting(self, discharge, specific_lateral_inflow, section="main_channel"):
        """Host-vector form, exactly the reference call: `discharge` (fp64, C-contiguous) is updated in place."""
        sec = self._section(section)
        if not (isinstance(discharge, np.ndarray) and discharge.dtype == np.float64 and discharge.flags.c_contiguous
                and discharge.size == self.num_pixels):
            raise ValueError("discharge must be a C-contiguous float64 vector of %d land pixels" % self.num_pixels)
        q = f64(np.broadcast_to(specific_lateral_inflow, (self.num_pixels,)))
        if self.num_pixels == 0:
            return
        check(lib().lf_router_route_host(self._h, ptr(discharge), ptr(q), C.c_int(sec)))
        if self.flagnancheck and not self.kinematic_wave_warning_printed:
            if not np.all(np.isfinite(discharge)):
                self._warn()

    def route_device(self, discharge_dev, lateral_dev, section="main_channel"):
        """Device-resident form: DeviceArray vectors in pixel order; asynchronous."""
        sec = self._section(section)
        check(lib().lf_router_route_device(self._h, discharge_dev.ptr, lateral_dev.ptr, C.c_int(sec)))
        if self.flagnancheck and not self.kinematic_wave_warning_printed:
            n = C.c_int64(0)
            check(lib().lf_count_nonfinite(C.c_int(self.device), discharge_dev.ptr, C.c_int64(self.num_pixels),
                                           C.byref(n)))
            if n.value:
                self._warn()

    def to_engine_order(self, src_pix_dev, dst_ord_dev=None):
        """pixel order -> engine (sweep) order on the device; returns the DeviceArray in engine order."""
        if dst_ord_dev is None:
            dst_ord_dev = DeviceArray(self.num_pixels, np.float64, self.device)
        check(lib().lf_router_to_engine_order(self._h, src_pix_dev.ptr, dst_ord_dev.ptr))
        return dst_ord_dev

    def from_engine_order(self, src_ord_dev, dst_pix_dev=None):
        if dst_pix_dev is None:
            dst_pix_dev = DeviceArray(self.num_pixels, np.float64, self.device)
        check(lib().lf_router_from_engine_order(self._h, src_ord_dev.ptr, dst_pix_dev.ptr))
        return dst_pix_dev

    def route_ordered(self, discharge_ord_dev, lateral_ord_dev, section="main_channel"):
        """Engine-order resident form: both DeviceArrays are in sweep order, discharge is updated in place."""
        sec = self._section(section)
        check(lib().lf_router_route_ordered(self._h, discharge_ord_dev.ptr, lateral_ord_dev.ptr, C.c_int(sec)))

    @staticmethod
    def route_together(routers, discharge_devs, lateral_devs, section="main_channel", engine_order=False):
        """Several routers built on the same graph (the direct / other / forest overland routers of
        surface_routing.py:108-113) swept with one launch per level for all of them -- lf_router_route_device_multi.
        DeviceArray vectors, pixel order unless engine_order."""
        n = len(routers)
        sec = kinematicWave._section(section)
        hs = (C.c_void_p * n)(*[r._h for r in routers])
        qs = (C.c_void_p * n)(*[q.ptr for q in discharge_devs])
        ls = (C.c_void_p * n)(*[x.ptr for x in lateral_devs])
        check(lib().lf_router_route_device_multi(C.c_int(n), hs, qs, ls, C.c_int(sec), C.c_int(1 if engine_order else 0)))

    def _warn(self):
        self.kinematic_wave_warning_printed = True
        warnings.warn(LisfloodWarning("Warning: NaN or Inf values after kinematicRouting module. Suggestion: please "
                                      "check the input maps (e.g. channel geometry and ldd)"))

    # --- LDD reductions on the same graph ------------------------------------------------------------
    def upstream_sum(self, weights):
        """np.bincount(downstruct, weights)[:N] / PCRaster upstream(ldd, x) (routing.py:159-164, 387)."""
        w = f64(weights)
        out = np.empty(self.num_pixels)
        if self.num_pixels == 0:
            return out
        check(lib().lf_upstream_sum_host(self._h, ptr(w), ptr(out)))
        return out

    def accuflux(self, x):
        """PCRaster accuflux(ldd, x): x accumulated over all upstream cells incl. the cell (routing.py:98)."""
        xv = f64(np.broadcast_to(x, (self.num_pixels,)))
        out = np.empty(self.num_pixels)
        if self.num_pixels == 0:
            return out
        check(lib().lf_accuflux_host(self._h, ptr(xv), ptr(out)))
        return out

    # --- instrumentation -------------------------------------------------------------------------
    def last_launches(self):
        s = (C.c_int64 * 4)()
        check(lib().lf_router_last_launches(self._h, s))
        return dict(launches=s[0], wide=s[1], narrow=s[2], levels=s[3])

    def route_plan_stats(self):
        """shape of the block plan of single router calls; lane_use = cells per 64-lane cone level"""
        s = (C.c_int64 * 6)()
        check(lib().lf_router_route_plan_stats(self._h, s))
        return dict(blocks=s[0], cone_blocks=s[1], cones=s[2], cone_levels=s[3], cells=s[4], max_cones_per_launch=s[5],
                    lane_use=(s[4] / (64.0 * s[3]) if s[3] else 0.0))

    def profile(self, on):
        check(lib().lf_router_profile_enable(self._h, C.c_int(1 if on else 0)))

    def profile_read(self, reset=True):
        o = (C.c_double * 9)()
        check(lib().lf_router_profile_read(self._h, o, C.c_int(1 if reset else 0)))
        names = ("prep", "wide_level", "narrow_run")
        return {names[i]: dict(launches=o[3 * i], ms=o[3 * i + 1], cells=o[3 * i + 2]) for i in range(3)}

    def close(self):
        if getattr(self, "_h", None):
            lib().lf_router_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
