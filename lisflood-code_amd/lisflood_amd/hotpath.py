"""Device-resident hot path of one LISFLOOD model step (Lisflood_dynamic.py:114-229, the stages SURVEY.md section 8
covers), with every vector kept in HBM between steps:

    soilloop.dynamic_canopy -> soilloop.dynamic_soil -> opensealed / soil.dynamic_perpixel / groundwater
    -> surface_routing.dynamic (3 routers on LddToChan) -> NoRoutSteps x routing.dynamic (1-2 routers, fused
    wavefront) -> ChanQAvg

Per step only the meteorological forcing (Rain, SnowMelt, EWRef, ETRef, ESRef -- five [N] vectors) crosses PCIe.
The stages are exactly the kernels the module classes (`soilloop`, `pixel_aggregates`, `surface_routing`,
`routing`) call -- those stage their `var` arrays through the host on every call, this class wires the same
device buffers from one stage to the next.  Channel vectors live in the channel router's engine order; the one
vector that crosses from the surface graph to the channel graph (ToChanM3RunoffDt) is permuted on the device.
"""
import ctypes as C
import types

import numpy as np

from . import pixel_aggregates as PA
from . import routing as RT
from . import soilloop as SL
from . import surface_routing as SR
from ._lib import DeviceArray, check, f64, lib, u8
from .kinematic_wave_parallel import Graph, kinematicWave

FORCING = ("Rain", "SnowMelt", "EWRef", "ETRef", "ESRef")
_CHANNEL_NAMES = set(RT._STATIC + RT._STATE + RT._OUT)


# Maps of the land-surface stages that nothing else on the hot path reads (HotPathDevice(report=...)): the soil kernel's
# diagnostics (soilloop.py:330-336), the per-pixel diagnostics of opensealed / soil.dynamic_perpixel / groundwater and the
# cumulative sums of the mass-balance report.  Everything else -- states, `dis`, what a later stage reads -- always exists.
OPTIONAL_MAPS = ("Theta1a Theta1b Theta2 Sat1a Sat1b Sat1 Sat2 "
                 "RainSnowmelt EWaterAct InterSealed TASealed TaInterceptionAll TaPixel ESActPixel PrefFlowPixel InfiltrationPixel "
                 "Theta ThetaAll SeepTopToSubPixelA SeepTopToSubPixelB SeepSubToGWPixel Theta1aPixel Theta1bPixel Theta2Pixel "
                 "LZOutflow GwPercUZLZPixel GwLossLZ LZAvInflow LZInflowCUM TaInterceptionCUM TaCUM ESActCUM GwLossCUM").split()
# [3,N] inputs of the per-pixel aggregates -> the optional outputs that read them (none reported: the input is not streamed)
PA_READERS = dict(TaInterception=("TaInterceptionAll", "TaInterceptionCUM"), Ta=("TaPixel", "TaCUM"), ESAct=("ESActPixel", "ESActCUM"),
                  PrefFlow=("PrefFlowPixel",), Infiltration=("InfiltrationPixel",), SeepTopToSubA=("SeepTopToSubPixelA",),
                  SeepTopToSubB=("SeepTopToSubPixelB",), SeepSubToGW=("SeepSubToGWPixel",), Theta1a=("Theta1aPixel",),
                  Theta1b=("Theta1bPixel",), Theta2=("Theta2Pixel",), W1a=("Theta", "ThetaAll"), W1b=("Theta", "ThetaAll"),
                  W2=("Theta", "ThetaAll"), SoilDepthTotal=("Theta", "ThetaAll"))


class HotPathDevice:
    def __init__(self, values, scalars, land_mask, ldd_to_chan, ldd_kinematic, split=True, device=0, structures=None,
                 compact=True, overlap_channel=True, surface_order=True, land_fused=None, report=None):
        """values: name -> host array in pixel order ([N], [3,N]) for every vector of the stages (reference
        attribute names); scalars: Beta, DtSec, DtRouting, NoRoutSteps, DtDay, PixelLength, MMtoM3, M3toMM,
        LeafDrainageK, AvWaterThreshold, CourantCrit, DrainedFraction, InvDtDay.  ldd_to_chan / ldd_kinematic:
        compressed LDD codes of the overland and the channel graph.
        structures: optional dict of the reference's lake / reservoir / inflow / transmission attributes (the names
        routing.attach_structures reads, incl. `downstruct` of the uncut LDD); ldd_kinematic is then the CUT LDD
        (structures.py:44-61) and the sub-step loop runs with the structures inside the wavefront.
        compact: leave inert pixels out of the channel router's domain (see inert_pixels): on real domains most land
        pixels are not channel pixels, and the channel wavefront then touches only the ones that are."""
        self.device, self.split = device, bool(split)
        # surface_order: every per-pixel vector that is not a channel vector lives in the sweep order of the OVERLAND
        # routers' graph (the canopy / soil / aggregate kernels are element-wise and do not care; the three overland routers
        # then stream their vectors instead of going through the position -> pixel table: lf_surface_step_ordered).
        # Vectors come in and go out in pixel order all the same (constructor, step(), download(), state files); a caller
        # whose forcing is already in that order (`pixel_of_position`) says step(..., ordered=True) and saves the gather.
        # overlap_channel: the channel wavefront of a step on a second HIP stream, beside the canopy / soil / overland kernels
        # of the step after it (same kernels, same results; False: one stream, as rounds 1-3)
        self.overlap_channel = bool(overlap_channel)
        self.rmod = None
        # the ten derived soil parameter arrays recomputed instead of read where they are what soil.py:180-228 makes them
        self.soil_derived = SL.derived_parameters_hold(values)
        # land_fused: canopy + ESMax + soil columns as ONE pass over the columns (lf_land_columns_device; the three prescribed
        # fractions map to their own land-use rows, which is what it needs).  None: on, unless LF_LAND_FUSED=0 (A/B switch);
        # False: the three separate launches of rounds 1-5 -- same bits either way
        import os
        self.land_fused = (os.environ.get("LF_LAND_FUSED", "1") != "0") if land_fused is None else bool(land_fused)
        # report: which of the OPTIONAL maps (OPTIONAL_MAPS below: diagnostics and cumulative sums nothing else on the hot
        # path reads) are wanted.  None: all of them, as the reference computes them every step; an iterable of names: only
        # those -- the others get no device vector, are not computed, and the [3,N] vectors only they read are not streamed
        # (the reference writes the maps its rep* options name; what is not reported need not exist).  `dis`, the state
        # maps and everything another stage reads are always there.
        self.report = None if report is None else set(report)
        if self.report is not None and not self.report <= set(OPTIONAL_MAPS):
            raise ValueError("report: not optional maps: %s" % sorted(self.report - set(OPTIONAL_MAPS)))
        self.sc = dict(scalars)
        land_mask = np.asarray(land_mask, bool)
        self.N = N = int(land_mask.sum())
        sc = self.sc
        # ---- channel domain: all land pixels, or only the ones whose routing sub-step is not the identity ----------
        values = dict(values)
        ldd_kinematic = np.asarray(ldd_kinematic, np.float64)
        drop = inert_pixels(values, ldd_kinematic, land_mask, self.split, structures) if compact else np.zeros(N, bool)
        if drop.sum() < 0.1 * N:
            drop[:] = False
        self.ids = np.nonzero(~drop)[0]                       # channel-domain pixel -> land pixel
        Nk = self.Nk = self.ids.size
        chan_mask = land_mask
        if Nk < N:
            chan_mask = np.zeros(land_mask.shape, bool)
            chan_mask[land_mask] = ~drop
            ldd_kinematic = ldd_kinematic[self.ids]
            for k in set(RT._STATIC + RT._STATE) & set(values):
                values[k] = np.broadcast_to(np.asarray(values[k]), (N,))[self.ids]
            if structures is not None:
                structures = _structures_on_subdomain(structures, self.ids, N)
        # ---- routers -------------------------------------------------------------------------------
        g_surf = Graph(ldd_to_chan, land_mask)
        alpha_of = np.asarray(values["OFAlpha"], float)
        mk = lambda row: kinematicWave(None, None, alpha_of[row], sc["Beta"], sc["PixelLength"], sc["DtSec"],
                                       device=device, graph=g_surf)
        self.r_other, self.r_forest, self.r_direct = mk(0), mk(1), mk(2)
        chan_names = set(RT._STATIC + RT._STATE)
        self.d = {}
        if structures is None:
            self.river = kinematicWave(ldd_kinematic, chan_mask, values["ChannelAlpha"], sc["Beta"], values["ChanLength"],
                                       sc["DtRouting"], alpha_floodplains=values["ChannelAlpha2"] if split else None,
                                       device=device)
        else:       # the channel part is a resident, engine-order routing module with its structures attached
            v = types.SimpleNamespace(**{k: np.array(values[k], copy=True) for k in chan_names if k in values})
            v.Beta, v.InvBeta, v.DtRouting, v.InvDtRouting = sc["Beta"], 1 / sc["Beta"], sc["DtRouting"], 1 / sc["DtRouting"]
            v.DtSec, v.NoRoutSteps, v.InvNoRoutSteps = sc["DtSec"], int(sc["NoRoutSteps"]), 1 / sc["NoRoutSteps"]
            v.ToChanM3RunoffDt = np.zeros(Nk)
            for k, a in structures.items():
                setattr(v, k, a)
            opts = dict(SplitRouting=self.split, InitLisflood=False, simulateLakes="LakeIndex" in structures,
                        simulateReservoirs="ReservoirIndex" in structures, inflow="QInM3Old" in structures,
                        TransLoss="UpTrans" in structures)
            m = self.rmod = RT.routing(v, options=opts, device=device, engine_order=True)
            m.attach_router(ldd_kinematic, chan_mask)
            m.attach_structures()
            m.begin_step()                              # uploads the channel state once; it stays resident
            m._structures_substep(0, launch=False)      # site state from the dense maps, once
            m._args.split = 1 if self.split else 0
            self.river = m.river_router
            if "QInM3Old" in structures:       # inflow hydrographs: QInM3 of the previous model step, channel domain
                self._qin_old = f64(np.broadcast_to(structures["QInM3Old"], (Nk,))).copy()
            for k in chan_names | set(RT._OUT) | {"SideflowChanM3"}:
                self.d[k] = m._dev[k]
        self.perm = self.river.graph.layout()[0].astype(np.int64)        # engine position -> channel-domain pixel
        self.gpix = self.ids[self.perm]                                   # engine position -> land pixel
        # position -> pixel of the overland graph's sweep order (None: the non-channel vectors stay in pixel order)
        self.pixel_of_position = g_surf.layout()[0].astype(np.int64) if surface_order else None
        if self.pixel_of_position is not None:
            inv = np.empty(N, np.int64)
            inv[self.pixel_of_position] = np.arange(N)
            src = inv[self.gpix]                                          # where a channel cell's land pixel sits in those vectors
        else:
            src = self.gpix
        self._gidx = DeviceArray.from_host(src.astype(np.int32), device)
        # ---- device vectors ------------------------------------------------------------------------
        bool_names = SL._BOOL | {"IsChannel", "IsChannelKinematic"}
        for k, a in values.items():
            a = np.asarray(a)
            if self.report is not None and k in OPTIONAL_MAPS and k not in self._kept():
                continue                              # an optional map nobody asked for: no device vector at all
            if k in chan_names:                       # channel vectors: engine order of the river router
                if self.rmod is not None:
                    continue
                a = np.broadcast_to(a, (Nk,))[self.perm]
            elif self.pixel_of_position is not None and a.ndim >= 1 and a.shape[-1] == N and k not in bool_names:
                self.d[k] = self._upload_ordered(f64(a))      # permuted on the device (a host gather of ~100 fields is slow)
                continue
            else:
                a = self._ordered(a)
            self.d[k] = DeviceArray.from_host(u8(a) if k in bool_names else f64(a), device)
        if getattr(self, "_perm_tmp", None) is not None:
            self._perm_tmp.free(); self._perm_idx.free()
            self._perm_tmp = self._perm_idx = None

        def zeros(name, shape):
            if name not in self.d:
                self.d[name] = DeviceArray(shape, np.float64, device).zero()
            return self.d[name]
        # forcing: two buffer sets, so that the upload of the next step's vectors (second HIP stream) overlaps the
        # kernels of the current step -- see prefetch()
        self.force = [{k: DeviceArray(N, np.float64, device).zero() for k in FORCING} for _ in range(2)]
        self._prefetched, self._keep = [None, None], [None, None]
        for k in FORCING:
            self.d[k] = self.force[0][k]
        # ---- argument blocks (pointers into self.d) -----------------------------------------------------
        def fill(args, names, shape_of):
            for k in names:
                setattr(args, k, zeros(k, shape_of(k)).ptr.value)
        vn = lambda k: (3, N)
        n1 = lambda k: N
        a = self.canopy = SL._CanopyArgs()
        fill(a, SL._CANOPY_IO + SL._CANOPY_V_IN + SL._CANOPY_L_IN, vn)
        fill(a, SL._CANOPY_N_IN, n1)
        self._idx = np.arange(3, dtype=np.int64)
        a.index_landuse = self._idx.ctypes.data
        a.LeafDrainageK, a.DtDay, a.InvDtDay = sc["LeafDrainageK"], sc["DtDay"], sc["InvDtDay"]
        a.V, a.L, a.N = 3, 3, N
        zeros("LAITerm", (3, N)); zeros("ESMax", (3, N))
        s = self.soil = SL._SoilArgs()
        wanted = lambda names: [k for k in names if self.report is None or k not in OPTIONAL_MAPS or k in self._kept()]
        fill(s, wanted(SL._L_FIELDS + SL._V_IN + SL._V_IO), vn)
        fill(s, SL._N_FIELDS, n1)
        self._irr = np.array([0, 0, 1], np.uint8)
        self._pad = np.zeros(3, np.uint8)
        s.index_landuse_all, s.is_irrigated, s.is_paddy_irrig = (self._idx.ctypes.data, self._irr.ctypes.data,
                                                                 self._pad.ctypes.data)
        s.DtDay, s.AvWaterThreshold, s.CourantCrit, s.DrainedFraction = (sc["DtDay"], sc["AvWaterThreshold"],
                                                                         sc["CourantCrit"], sc["DrainedFraction"])
        s.V, s.L, s.N = 3, 3, N
        p = self.pixel = PA._PixelArgs()
        fill(p, wanted(PA._V_IN + ["Theta"]), vn)
        fill(p, wanted(PA._N_IN + PA._STATE + PA._OUT), n1)
        if self.report is not None:        # a [3,N] input that only unreported maps read is not handed to the kernel
            for src, outs in PA_READERS.items():
                if not any(o in self._kept() for o in outs):
                    setattr(p, src, None)
        p.InvDtDay, p.N = sc["InvDtDay"], N
        f = self.surface = SR._SurfaceArgs()
        fill(f, SR._V_IN + ["SurfaceRunSoil", "scratch"], vn)
        fill(f, SR._N_IN + SR._STATE + SR._OUT, n1)
        f.Beta, f.MMtoM3, f.M3toMM = sc["Beta"], sc["MMtoM3"], sc["M3toMM"]
        f.PixelLength, f.InvPixelLength = sc["PixelLength"], 1 / sc["PixelLength"]
        f.DtSec, f.InvDtSec, f.InvNoRoutSteps, f.N = sc["DtSec"], 1 / sc["DtSec"], 1 / sc["NoRoutSteps"], N
        r = self.rout = RT._SubstepArgs()
        fill(r, RT._STATIC + ["SideflowChanM3"] + RT._STATE + RT._OUT + ["scratch0", "scratch1"], lambda k: max(Nk, 1))
        r.Beta, r.InvBeta, r.InvDtRouting, r.DtSec = sc["Beta"], 1 / sc["Beta"], 1 / sc["DtRouting"], sc["DtSec"]
        r.split, r.engine_order = (1 if self.split else 0), 1
        self.steps_done = 0

    def _kept(self):
        """the optional maps that exist in this object: the reported ones plus what they are made of (the per-pixel Theta
        averages read the soil kernel's Theta1a / Theta1b / Theta2)"""
        if self.report is None:
            return set(OPTIONAL_MAPS)
        keep = set(self.report)
        for pix, col in (("Theta1aPixel", "Theta1a"), ("Theta1bPixel", "Theta1b"), ("Theta2Pixel", "Theta2"),
                         ("LZAvInflow", "LZInflowCUM")):
            if pix in keep:
                keep.add(col)
        return keep

    def _upload_ordered(self, a):
        """fp64 [N] or [R, N] host array in pixel order -> device array in the order of pixel_of_position: every row is
        uploaded as it is and gathered on the device (lf_gather_device)"""
        L, dev, N = lib(), self.device, self.N
        if getattr(self, "_perm_tmp", None) is None:
            self._perm_tmp = DeviceArray(N, np.float64, dev)
            self._perm_idx = DeviceArray.from_host(self.pixel_of_position.astype(np.int32), dev)
        dst = DeviceArray(a.shape, np.float64, dev)
        rows = a.reshape(-1, N)
        for r in range(rows.shape[0]):
            self._perm_tmp.upload(np.ascontiguousarray(rows[r]))
            check(L.lf_gather_device(C.c_int(dev), C.c_int64(N), self._perm_idx.ptr, self._perm_tmp.ptr,
                                     C.c_void_p(dst.ptr.value + r * N * 8)))
        return dst

    def _ordered(self, a):
        """a per-pixel array ([N] or [..., N], pixel order) in the order the non-channel device vectors are kept in"""
        a = np.asarray(a)
        if self.pixel_of_position is None or a.ndim == 0 or a.shape[-1] != self.N:
            return a
        return np.ascontiguousarray(a[..., self.pixel_of_position])

    def _pixel_order(self, a):
        """inverse of _ordered"""
        if self.pixel_of_position is None or a.ndim == 0 or a.shape[-1] != self.N:
            return a
        out = np.empty_like(a)
        out[..., self.pixel_of_position] = a
        return out

    def _upload(self, b, forcing, ordered=False):
        """float32 vectors (meteo as the netCDF files store it) go up as float32 and are widened on the device -- exactly
        what widening them on the host first would give, at half the PCIe traffic"""
        L, dev = lib(), self.device
        as_is = lambda a: np.ascontiguousarray(a) if np.asarray(a).dtype == np.float32 else f64(a)
        host = [as_is(forcing[k]) if ordered else as_is(self._ordered(forcing[k])) for k in FORCING]
        check(L.lf_upload_begin(C.c_int(dev), C.c_int(b)))
        for k, a in zip(FORCING, host):
            if a.size != self.N:
                raise ValueError("forcing vector %s must have %d entries" % (k, self.N))
            if a.dtype == np.float32:
                check(L.lf_upload_copy_f32(C.c_int(dev), self.force[b][k].ptr, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.size)))
                continue
            check(L.lf_upload_copy(C.c_int(dev), self.force[b][k].ptr, a.ctypes.data_as(C.c_void_p), C.c_size_t(a.nbytes)))
        check(L.lf_upload_end(C.c_int(dev), C.c_int(b)))
        self._keep[b] = host                 # the host vectors stay alive until the set is uploaded again

    def pinned_forcing(self, dtype=np.float64):
        """A forcing dict (Rain, SnowMelt, EWRef, ETRef, ESRef) of [N] arrays in page-locked host memory, to be filled in
        place (e.g. by the netCDF reader) and passed to prefetch() / step(): their upload is an asynchronous DMA at the
        PCIe rate instead of a staged, blocking copy.  The arrays belong to this object (freed by free()).
        The DMA reads the arrays AFTER prefetch() / step() has returned: call upload_wait() before writing the next
        time step into a set that was handed over (or alternate between two pinned sets and wait before reuse)."""
        from ._lib import PinnedArray
        bufs = {k: PinnedArray(self.N, dtype, self.device) for k in FORCING}
        self.__dict__.setdefault("_pinned", []).append(bufs)
        return {k: b.a for k, b in bufs.items()}

    def upload_wait(self, buffer_set=None):
        """Block until the forcing uploads started by prefetch() / step() have left the host arrays (lf_upload_wait):
        after it the arrays of pinned_forcing() may be refilled in place.  buffer_set: 0 / 1, default both."""
        for b in ((0, 1) if buffer_set is None else (int(buffer_set),)):
            check(lib().lf_upload_wait(C.c_int(self.device), C.c_int(b)))

    def prefetch(self, forcing, ordered=False):
        """Start uploading the forcing of the NEXT step() call now: the copies run on a second stream while the kernels
        of the step just enqueued are still executing.  Pass the same dict object to the next step().
        ordered: the vectors are already in the order of `pixel_of_position` (e.g. compressed from the netCDF raster
        with the composed index) -- otherwise they are in pixel order and are permuted on the host first."""
        b = self.steps_done % 2
        self._upload(b, forcing, ordered)
        self._prefetched[b] = forcing

    def _use_set(self, b):
        for k in FORCING:
            self.d[k] = self.force[b][k]
            for st in (self.canopy, self.soil, self.pixel, self.surface):
                if hasattr(type(st), k):
                    setattr(st, k, self.force[b][k].ptr.value)

    def set_lai(self, LAI, LAITerm):
        """The prescribed leaf area index of the coming steps: what leafarea.dynamic sets once per ten-day interval
        (leafarea.py:80-91: LAI of the interval, LAITerm = exp(-kgb * LAI)), [3, N] each, pixel order.  Waits for the
        step in flight (its canopy kernel reads the vectors)."""
        check(lib().lf_device_synchronize(C.c_int(self.device)))
        for k, a in (("LAI", LAI), ("LAITerm", LAITerm)):
            a = np.asarray(a, np.float64)
            if a.shape != (3, self.N):
                raise ValueError("%s must be [3, %d]" % (k, self.N))
            self.d[k].upload(f64(self._ordered(a)))

    def set_inflow(self, QInM3):
        """inflow.dynamic + dynamic_init (inflow.py:108-125) for the coming step: QInM3 [N] is the hydrograph volume of
        the model step [m3]; QDelta = (QInM3 - QInM3Old) * InvNoRoutSteps goes to the device next to QInM3Old, and
        QInM3 becomes QInM3Old once the step is enqueued (Lisflood_dynamic.py:185)."""
        if self.rmod is None or getattr(self, "_qin_old", None) is None:
            raise RuntimeError("inflow hydrographs need structures= with QInM3Old / QDelta (the `inflow` option)")
        q = f64(np.broadcast_to(np.asarray(QInM3, np.float64), (self.N,)))
        if self.Nk < self.N:
            rest = np.ones(self.N, bool); rest[self.ids] = False
            if np.any(q[rest] != 0):
                raise ValueError("inflow at a pixel this object left out of the channel domain; flag the inflow pixels in "
                                 "structures['InflowPoints'] or build it with compact=False")
        q = np.ascontiguousarray(q[self.ids])
        m, dev = self.rmod, self.rmod._st["dev"]
        delta = (q - self._qin_old) * m.var.InvNoRoutSteps
        # Only the channel wavefront reads the two vectors: they go up on ITS stream (the side stream when the wavefront
        # runs there), behind the wavefront of the step before and ahead of the next one, without a host wait and without
        # making the main stream wait for the side stream (lf_memcpy_h2d would do both)
        L, d = lib(), C.c_int(self.device)
        side = self.overlap_channel
        if side:
            check(L.lf_side_stream_begin(d))
        try:
            dev["QInM3Old"].upload_staged(f64(m._up(self._qin_old)))
            dev["QDelta"].upload_staged(f64(m._up(delta)))
        finally:
            if side:
                check(L.lf_side_stream_end(d))
        self._qin_old = q

    def step(self, forcing, time_since_start=None, QInM3=None, ordered=False):
        d, dev = self.d, self.device
        L = lib()
        if QInM3 is not None:
            self.set_inflow(QInM3)
        b = self.steps_done % 2
        if forcing is None:          # the vectors this buffer set already holds (uploaded for an earlier step): no PCIe traffic
            if self._keep[b] is None:
                raise ValueError("step(None): buffer set %d has never been uploaded" % b)
        elif self._prefetched[b] is not forcing:
            self._upload(b, forcing, ordered)
        self._prefetched[b] = None
        check(L.lf_compute_acquire(C.c_int(dev), C.c_int(b)))
        self._use_set(b)
        try:
            self._enqueue(time_since_start)
        finally:
            check(L.lf_compute_release(C.c_int(dev), C.c_int(b)))

    def _enqueue(self, time_since_start, stage_ms=None):
        """stage_ms: a dict -> every stage is bracketed by the device stopwatch (which synchronises) and its
        milliseconds are added under its name; the channel wavefront then runs on the main stream (step_profile)."""
        d, dev = self.d, self.device
        L = lib()
        from . import _lib as LB

        class stage:                       # `with stage("soil"):` -- a no-op unless stage_ms was passed
            def __init__(self, name):
                self.name = name

            def __enter__(self):
                if stage_ms is not None:
                    LB.timer_start(dev)

            def __exit__(self, *exc):
                if stage_ms is not None and exc[0] is None:
                    stage_ms[self.name] = stage_ms.get(self.name, 0.0) + LB.timer_stop(dev)
                return False
        if self.land_fused:
            # canopy, ESMax and the soil columns in ONE pass (k_soil_fused<.., CANOPY>): the lane that runs a column's canopy
            # carries LeafDrainage / Interception / W1a / W1b / W1 / ESMax into its soil water balance in registers
            with stage("land_surface"):
                check(L.lf_land_columns_device(C.c_int(dev), C.byref(self.canopy), C.byref(self.soil), d["ESRef"].ptr,
                                               C.c_int(1 if self.soil_derived else 0)))                 # dyn.py:114-123
        else:
            with stage("canopy"):
                check(L.lf_canopy_device(C.c_int(dev), C.byref(self.canopy)))                           # dyn.py:114
                check(L.lf_scale_rows_device(C.c_int(dev), d["ESRef"].ptr, d["LAITerm"].ptr, d["ESMax"].ptr,
                                             C.c_int64(3), C.c_int64(self.N)))                         # soilloop.py:638
            with stage("soil_columns"):
                soil_fn = L.lf_soil_columns_device_derived if self.soil_derived else L.lf_soil_columns_device
                check(soil_fn(C.c_int(dev), C.byref(self.soil)))                                        # dyn.py:123
        self.steps_done += 1
        self.pixel.TimeSinceStart = float(time_since_start if time_since_start else self.steps_done)
        with stage("pixel_aggregates"):
            check(L.lf_pixel_aggregates_device(C.c_int(dev), C.byref(self.pixel)))                      # dyn.py:129-149
        with stage("overland"):
            step_fn = L.lf_surface_step_ordered if self.pixel_of_position is not None else L.lf_surface_step
            check(step_fn(self.r_direct._h, self.r_other._h, self.r_forest._h, C.byref(self.surface)))   # dyn.py:165
        # The channel wavefront reads nothing but its own vectors and the sideflow gathered below, and nothing of the NEXT
        # step's canopy / soil / aggregate / overland kernels reads a channel vector: it runs on the side stream, beside
        # them (a latency-bound chain of small launches beside bandwidth-bound streaming kernels).  Before the gather
        # overwrites the sideflow the main stream waits for the wavefront of the step before; downloads join by themselves.
        check(L.lf_side_stream_join(C.c_int(dev)))
        with stage("sideflow_gather"):
            if self.rmod is not None:       # lakes / reservoirs / inflow / transmission loss inside the wavefront
                m = self.rmod
                check(L.lf_gather_device(C.c_int(dev), C.c_int64(self.Nk), self._gidx.ptr, d["ToChanM3RunoffDt"].ptr,
                                         m._st["dev"]["ToChanM3RunoffDt"].ptr))
            else:
                check(L.lf_gather_device(C.c_int(dev), C.c_int64(self.Nk), self._gidx.ptr, d["ToChanM3RunoffDt"].ptr,
                                         d["SideflowChanM3"].ptr))
        side = self.overlap_channel and stage_ms is None
        if side:
            check(L.lf_side_stream_begin(C.c_int(dev)))
        try:
            with stage("channel_wavefront"):
                d["sumDisDay"].zero()                                                                   # dyn.py:177
                if self.rmod is not None:
                    m = self.rmod
                    check(L.lf_routing_substeps_fused_structures(self.river._h, C.byref(m._args), C.byref(m._inloop),
                                                                 C.c_int(int(self.sc["NoRoutSteps"]))))  # dyn.py:179-180
                else:
                    check(L.lf_routing_substeps_fused(self.river._h, C.byref(self.rout), C.c_int(int(self.sc["NoRoutSteps"])),
                                                      C.c_int64(0)))                                    # dyn.py:179-180
        finally:
            if side:
                check(L.lf_side_stream_end(C.c_int(dev)))

    def step_profile(self, forcing, time_since_start=None, ordered=False):
        """step() with every stage timed on its own (synchronising; no overlap between the stages) -> {stage: ms}"""
        L, dev = lib(), self.device
        b = self.steps_done % 2
        if self._prefetched[b] is not forcing:
            self._upload(b, forcing, ordered)
        self._prefetched[b] = None
        check(L.lf_compute_acquire(C.c_int(dev), C.c_int(b)))
        self._use_set(b)
        ms = {}
        try:
            self._enqueue(time_since_start, ms)
        finally:
            check(L.lf_compute_release(C.c_int(dev), C.c_int(b)))
        return ms

    def stage_bytes(self):
        """algorithmic HBM bytes of one model step, stage by stage: every vector a stage reads or writes counted once
        (8 B per fp64, 1 B per flag; in/out vectors twice), the routers at 48 B per cell and call"""
        N, Nk, n_sub = self.N, self.Nk, int(self.sc["NoRoutSteps"])
        v8 = lambda names: 3 * 8 * len(names)
        canopy = N * (2 * v8(SL._CANOPY_IO) + v8(SL._CANOPY_V_IN) + v8(SL._CANOPY_L_IN) + 8 * len(SL._CANOPY_N_IN) + 3 * 24)
        soil = 3 * N * 504
        kept = self._kept()
        have = lambda names: [k for k in names if k not in OPTIONAL_MAPS or k in kept]
        v_in = [k for k in PA._V_IN if k not in PA_READERS or any(o in kept for o in PA_READERS[k])]
        io = set(have(PA._STATE))
        pixel = N * (v8(v_in) + (24 if "Theta" in kept else 0) + 8 * len(PA._N_IN) + 16 * len(io) +
                     8 * len([k for k in have(PA._OUT) if k not in io]))
        soil -= 3 * N * 8 * len([k for k in ("Theta1a", "Theta1b", "Theta2", "Sat1a", "Sat1b", "Sat1", "Sat2") if k not in kept])
        sio = set(SR._STATE)
        overland = N * (v8(SR._V_IN) + 2 * 24 + 8 * len(SR._N_IN) + 16 * len(sio) + 8 * len([k for k in SR._OUT if k not in sio])
                        + 3 * 48)
        channel = Nk * n_sub * (96 if self.split else 48)
        # the fused land surface: the soil's 504 B per column without the three streams that now stay in registers
        # (LeafDrainage, Interception, ESMax), the canopy's own streams -- 5 read (LAI, LAITerm, CumInterception, CropCoef,
        # CropGroupNumber; its WWP / WFC / W1 reads are the soil's), 7 written -- and the three [N] vectors EWRef, ETRef, ESRef
        land = soil + 3 * N * (-24 + 8 * (5 + 7)) + 24 * N
        out = dict(canopy=canopy, soil_columns=soil, land_surface=land, pixel_aggregates=pixel, overland=overland,
                   sideflow_gather=Nk * 20, channel_wavefront=channel)
        if self.land_fused:
            out.pop("canopy"); out.pop("soil_columns")
        else:
            out.pop("land_surface")
        return out

    def download(self, name):
        a = self.d[name].download()
        if name in set(RT._STATIC + RT._STATE + RT._OUT + ["SideflowChanM3"]):
            out = np.zeros(self.N, a.dtype)          # pixels outside the channel domain: their state is identically 0
            out[self.gpix] = a[:self.Nk]
            return out
        return self._pixel_order(a)

    def download_site(self, name):
        """lake / reservoir site vectors (LakeStorageM3CC, ReservoirStorageM3CC, ...) and the dense in-loop outputs"""
        a = self.rmod._st["dev"][name].download()
        if a.size == self.Nk and name not in RT._LAKE_STATE + RT._RES_STATE:
            out = np.zeros(self.N)
            out[self.ids] = self.rmod._down(a)
            return out
        return a

    # ---- warm start (the reference writes its state maps as end / state files, default_options.py:131-160) --------
    # everything a stage reads back from the previous step: the in/out vectors of the canopy and soil kernels, the
    # accumulators of the per-pixel aggregates, the overland and channel router states
    STATE = tuple(dict.fromkeys(SL._CANOPY_IO + list(SL._V_IO) + PA._STATE + SR._STATE +
                                ["OFM3Direct", "OFM3Other", "OFM3Forest"] + RT._STATE))

    def state_names(self):
        return [k for k in self.STATE if k in self.d]

    def save_state(self, path):
        """every state vector of the chain, pixel order, as one .npz (+ the site vectors of the structures)"""
        out = {k: self.download(k) for k in self.state_names()}
        if self.rmod is not None:
            for k in RT._LAKE_STATE + RT._RES_STATE + ["TransCum"]:
                if k in self.rmod._st["dev"]:
                    out["site_" + k] = self.download_site(k)
        out["steps_done"] = np.int64(self.steps_done)
        np.savez(path, **out)

    def load_state(self, path):
        z = np.load(path)
        chan = set(RT._STATIC + RT._STATE + RT._OUT)
        for k in self.state_names():
            if k not in z.files and k in OPTIONAL_MAPS:
                continue                              # a state file written by an object that did not report this map
            a = z[k]
            if k in chan:
                if self.Nk < self.N:
                    rest = np.ones(self.N, bool); rest[self.ids] = False
                    if np.any(a[rest] != 0):
                        raise ValueError("state file holds water in %s on pixels this object left out of the channel "
                                         "domain; build it with compact=False" % k)
                a = a[self.gpix]
            else:
                a = self._ordered(a)
            self.d[k].upload(f64(a))
        if self.rmod is not None:
            for k in RT._LAKE_STATE + RT._RES_STATE + ["TransCum"]:
                if "site_" + k in z.files:
                    a = z["site_" + k]
                    self.rmod._st["dev"][k].upload(f64(self.rmod._up(a[self.ids]) if k == "TransCum" else a))
        self.steps_done = int(z["steps_done"])

    # ---- state maps under the reference's names (default_options.py: the 'repStateMaps' / 'repEndMaps' entries) --------
    # name of the state map -> (attribute, row of a [3,N] array or None).  What the reference writes at a state step and
    # reads back through the *InitValue bindings of a warm run (settings/warm.xml); -9999 in a map read back means "cold
    # start value" (routing.py:203-218, surface_routing.py:49-63, soil.py:239-251, 393-406, groundwater.py:93-107).
    STATE_MAPS = dict(
        ChanQState=("ChanQ", None), ChanCrossSectionState=("TotalCrossSectionArea", None),
        CrossSection2State=("CrossSection2Area", None), ChSideState=("Sideflow1Chan", None),
        OFDirectState=("OFM3Direct", None), OFOtherState=("OFM3Other", None), OFForestState=("OFM3Forest", None),
        CumIntSealedState=("CumInterSealed", None), LZState=("LZ", None),
        CumInterceptionState=("CumInterception", 0), CumInterceptionForestState=("CumInterception", 1),
        CumInterceptionIrrigationState=("CumInterception", 2),
        DSLRState=("DSLR", 0), DSLRForestState=("DSLR", 1), DSLRIrrigationState=("DSLR", 2),
        UZState=("UZ", 0), UZForestState=("UZ", 1), UZIrrigationState=("UZ", 2),
        Theta1State=("Theta1a", 0), Theta1ForestState=("Theta1a", 1), Theta1IrrigationState=("Theta1a", 2),
        Theta2State=("Theta1b", 0), Theta2ForestState=("Theta1b", 1), Theta2IrrigationState=("Theta1b", 2),
        Theta3State=("Theta2", 0), Theta3ForestState=("Theta2", 1), Theta3IrrigationState=("Theta2", 2),
        LakeLevelState=("LakeLevel", None), LakePrevInflowState=("LakeInflowOld", None),
        LakePrevOutflowState=("LakeOutflow", None), ReservoirFillState=("ReservoirFill", None))

    def state_maps(self):
        """dict: reference state-map name -> [N] vector in pixel order -- what `repStateMaps` writes at a state step"""
        out = {}
        chan = {k: self.download(k) for k in ("ChanQ", "ChanM3Kin", "Chan2M3Kin", "Chan2M3Start", "InvChanLength",
                                              "CrossSection2Area", "Sideflow1Chan")}
        m3 = chan["ChanM3Kin"] + chan["Chan2M3Kin"] - chan["Chan2M3Start"] if self.split else chan["ChanM3Kin"]
        chan["TotalCrossSectionArea"] = m3 * chan["InvChanLength"]                  # Lisflood_dynamic.py:194-205
        depth = {k: self.download(k) for k in ("SoilDepth1a", "SoilDepth1b", "SoilDepth2")}
        pore = {k: self.download(k) for k in ("PoreSpaceNotZero1a", "PoreSpaceNotZero1b", "PoreSpaceNotZero2")}
        for name, (attr, row) in self.STATE_MAPS.items():
            if attr in chan:
                out[name] = chan[attr]
            elif attr.startswith("Theta"):          # thetaFun: W / SoilDepth, 0 without pore space (soilloop.py:386-387)
                lay = attr[5:]
                w, d, p = self.download("W" + lay)[row], depth["SoilDepth" + lay][row], pore["PoreSpaceNotZero" + lay][row]
                with np.errstate(divide="ignore", invalid="ignore"):
                    out[name] = np.where(p != 0, w / d, 0.0)
            elif attr in ("LakeLevel", "LakeInflowOld", "LakeOutflow", "ReservoirFill"):
                if self.rmod is None:
                    continue
                cc = {"LakeLevel": "LakeLevelCC", "LakeInflowOld": "LakeInflowOldCC", "LakeOutflow": "LakeOutflowCC",
                      "ReservoirFill": "ReservoirFillCC"}[attr]
                if cc not in self.rmod._st["dev"]:
                    continue
                idx = np.asarray(self.rmod.var.LakeIndex if attr.startswith("Lake") else self.rmod.var.ReservoirIndex)
                dense = np.zeros(self.N)
                dense[self.ids[idx]] = self.rmod._st["dev"][cc].download()          # lakes.py:283-292, reservoir.py:311-315
                out[name] = dense
            elif attr in self.d:
                a = self.download(attr)
                out[name] = a if row is None else a[row]
        return out

    def load_state_maps(self, maps):
        """Warm start from state maps under the reference's names (what a warm run reads through its *InitValue
        bindings); a map that is missing, or -9999, keeps the value this object was constructed with -- the reference's
        cold-start rule.  The channel state is rebuilt from TotalCrossSectionArea / CrossSection2Area exactly as
        routing.initial + initialSecond do (routing.py:203-218, 391-397); like the reference's, such a warm start
        continues to rounding, not to the bit (use save_state / load_state for that)."""
        g = lambda name: (None if name not in maps else np.asarray(maps[name], np.float64))
        up = lambda attr, a: self.d[attr].upload(f64(a[self.gpix] if attr in _CHANNEL_NAMES else self._ordered(a)))
        cur = lambda attr: self.download(attr)
        beta = self.sc["Beta"]
        # channel (routing.py:203-218, 243-248, 391-397)
        area = g("ChanCrossSectionState")
        if area is not None:
            length, alpha = cur("ChanLength"), cur("ChannelAlpha")
            with np.errstate(divide="ignore", invalid="ignore"):
                m3_cur = cur("ChanM3Kin") + (cur("Chan2M3Kin") - cur("Chan2M3Start") if self.split else 0.0)
                chan_m3 = np.where(area == -9999, m3_cur, area * length)
                if self.split:
                    c2 = g("CrossSection2State")
                    c2 = cur("CrossSection2Area") if c2 is None else np.where(c2 == -9999, 0.0, c2)
                    start, alpha2 = cur("Chan2M3Start"), cur("ChannelAlpha2")
                    m3_2 = c2 * length + start
                    m3_1 = chan_m3 - m3_2 + start
                    m3_1 = np.where((m3_1 < 0.0) & (m3_1 > -0.0000001), 0.0, m3_1)
                    up("CrossSection2Area", c2); up("Chan2M3Kin", m3_2); up("ChanM3Kin", m3_1)
                    up("Chan2QKin", (m3_2 * (1 / length) * (1 / alpha2)) ** (1 / beta))
                    up("ChanQKin", (m3_1 * (1 / length) * (1 / alpha)) ** (1 / beta))
                else:
                    up("ChanM3Kin", chan_m3)
                    up("ChanQKin", np.where(alpha > 0, (chan_m3 / length / alpha) ** (1 / beta), 0.0))
        q = g("ChanQState")
        if q is not None:
            up("ChanQ", np.where(q == -9999, cur("ChanQKin"), q))                   # routing.py:332-334
        side = g("ChSideState")
        if side is not None and self.split:
            up("Sideflow1Chan", np.where(side == -9999, 0.0, side))
        # overland flow (surface_routing.py:49-63, 93-95)
        ofa = self.download("OFAlpha")
        for name, row in (("Other", 0), ("Forest", 1), ("Direct", 2)):
            m3 = g("OF%sState" % name)
            if m3 is None:
                continue
            m3 = np.where(m3 == -9999, 0.0, m3)
            up("OFM3" + name, m3)
            up("OFQ" + name, (m3 * (1 / self.sc["PixelLength"]) * (1 / ofa[row])) ** (1 / beta))
        # soil, groundwater, interception
        for attr in ("CumInterception", "DSLR", "UZ"):
            a = cur(attr)
            hit = False
            for name, (at, row) in self.STATE_MAPS.items():
                if at == attr and g(name) is not None:
                    x = g(name)
                    a[row] = np.where(x == -9999, a[row], x)
                    hit = True
            if hit:
                up(attr, np.maximum(a, 1) if attr == "DSLR" else a)                  # soil.py:396-398
        for lay, thetas in (("1a", "Theta1"), ("1b", "Theta2"), ("2", "Theta3")):
            w, depth, pore = cur("W" + lay), cur("SoilDepth" + lay), cur("PoreSpaceNotZero" + lay)
            hit = False
            for row, suffix in enumerate(("State", "ForestState", "IrrigationState")):
                x = g(thetas + suffix)
                if x is not None:
                    w[row] = np.where(pore[row] != 0, np.where(x == -9999, w[row], x * depth[row]), 0.0)   # soil.py:261-267
                    hit = True
            if hit:
                up("W" + lay, w)
        if any(g(n) is not None for n in ("Theta1State", "Theta2State", "Theta1ForestState", "Theta2ForestState",
                                          "Theta1IrrigationState", "Theta2IrrigationState")):
            up("W1", cur("W1a") + cur("W1b"))                                       # soil.py:268
        for name, attr in (("LZState", "LZ"), ("CumIntSealedState", "CumInterSealed")):
            x = g(name)
            if x is not None:
                up(attr, np.where(x == -9999, cur(attr), x))
        if self.rmod is not None:
            dev = self.rmod._st["dev"]
            for name, cc, idx_name in (("LakeLevelState", "LakeLevelCC", "LakeIndex"),
                                       ("LakePrevInflowState", "LakeInflowOldCC", "LakeIndex"),
                                       ("LakePrevOutflowState", "LakeOutflowCC", "LakeIndex"),
                                       ("ReservoirFillState", "ReservoirFillCC", "ReservoirIndex")):
                x = g(name)
                if x is None or cc not in dev:
                    continue
                sites = self.ids[np.asarray(getattr(self.rmod.var, idx_name))]
                val = np.where(x[sites] == -9999, dev[cc].download(), x[sites])
                dev[cc].upload(f64(val))
                if cc == "LakeLevelCC":          # lakes.py:119-121: storage from level x area
                    area = dev["LakeAreaCC"].download()
                    dev["LakeStorageM3CC"].upload(f64(val * area))
                    dev["LakeStorageM3BalanceCC"].upload(f64(val * area))
                if cc == "ReservoirFillCC":      # reservoir.py:152-157
                    dev["ReservoirStorageM3CC"].upload(f64(val * dev["TotalReservoirStorageM3CC"].download()))

    def mass_balance(self):
        """Option repMBTs after a step() with structures: downloads the channel state into the routing module's `var`
        and runs its bookkeeping (routing.mbts_after_fused) -> (MBErrorSplitRoutingM3, OutletDischargeErrorSplitRouting)
        over the channel domain's pixels.  A diagnostic: it synchronises."""
        m = self.rmod
        if m is None:
            raise RuntimeError("mass_balance() needs structures=")
        m.options["repMBTs"] = True
        m._download_state()
        m._structures_download(int(self.sc["NoRoutSteps"]) - 1)
        m.var.ToChanM3RunoffDt = m._down(m._st["dev"]["ToChanM3RunoffDt"].download())
        m.mbts_after_fused()
        return m.var.MBErrorSplitRoutingM3, m.var.OutletDischargeErrorSplitRouting

    def chan_q_avg(self):
        """ChanQAvg = sumDisDay / NoRoutSteps (Lisflood_dynamic.py:209): the `dis` output of the reference."""
        return self.download("sumDisDay") / self.sc["NoRoutSteps"]

    def free(self):
        arrays = {id(a): a for a in list(self.d.values()) + [self._gidx]}
        arrays.update({id(a): a for fs in self.force for a in fs.values()})
        if self.rmod is not None:
            arrays.update({id(a): a for a in list(self.rmod._st["dev"].values()) + list(self.rmod._dev.values())})
        for a in arrays.values():
            a.free()
        for r in (self.r_other, self.r_forest, self.r_direct, self.river):
            r.close()
        _lib_sync = lib().lf_device_synchronize
        check(_lib_sync(C.c_int(self.device)))       # no upload may still be reading the page-locked buffers
        for bufs in self.__dict__.pop("_pinned", []):
            for b in bufs.values():
                b.free()


def inert_pixels(values, ldd_kinematic, land_mask, split, structures=None):
    """Land pixels whose routing sub-step is the identity for the whole run: no upstream and no downstream pixel in the
    kinematic LDD, not a channel pixel (so no sideflow, routing.py:512), regular parameters (0 * inf would turn the
    zero state into NaN, as it does in the reference), zero split-routing thresholds, an all-zero state, and no part
    in a structure.  Host mirror of k_inert_flags + the state test of the cell kernel (csrc/lf_router.hip)."""
    from . import ldd as L
    N = int(np.asarray(land_mask, bool).sum())
    get = lambda k, default=0.0: np.broadcast_to(np.asarray(values.get(k, default), dtype=np.float64), (N,))
    plus0 = lambda a: np.ascontiguousarray(a).view(np.int64) == 0
    down = L.downstream_index(ldd_kinematic, land_mask)
    has_up = np.bincount(down[down >= 0], minlength=N) > 0
    ok = (down < 0) & ~has_up & ~np.broadcast_to(np.asarray(values["IsChannelKinematic"], bool), (N,))
    with np.errstate(all="ignore"):
        t = get("InvChanLength") * get("ChanLength") * get("ChannelAlpha") * get("InvChannelAlpha")
        ok &= np.isfinite(t) & (get("InvChanLength") >= 0)
        names = ["ChanQKin", "ChanM3Kin", "ChanQ"]
        if split:
            ok &= np.isfinite(get("ChannelAlpha2") * get("InvChannelAlpha2"))
            names += ["QLimit", "Chan2QStart", "Chan2M3Start", "Chan2QKin", "Chan2M3Kin", "CrossSection2Area",
                      "Sideflow1Chan"]
    for k in names:
        ok &= plus0(get(k))
    if structures is not None:
        for k in ("LakeIndex", "ReservoirIndex"):
            if k in structures:
                ok[np.asarray(structures[k]).astype(np.int64)] = False
        for k in ("QInM3Old", "QDelta", "TransCum"):
            if k in structures:
                ok &= np.broadcast_to(np.asarray(structures[k], np.float64), (N,)) == 0
        if "InflowPoints" in structures:      # pixels that receive a hydrograph in later steps
            ok &= ~np.broadcast_to(np.asarray(structures["InflowPoints"], bool), (N,))
    return ok


def _structures_on_subdomain(st, ids, N):
    """The structure attributes (routing.attach_structures) renumbered for the channel domain `ids` of an N-pixel one."""
    Nk = ids.size
    new_id = np.full(N + 1, Nk, np.int64)
    new_id[ids] = np.arange(Nk)
    out = {}
    for k, a in st.items():
        if k in ("LakeIndex", "ReservoirIndex"):
            out[k] = new_id[np.asarray(a).astype(np.int64)]
            if (out[k] == Nk).any():
                raise ValueError("a structure sits on a pixel outside the channel domain")
        elif k == "downstruct":
            out[k] = new_id[np.minimum(np.asarray(a).astype(np.int64), N)][ids].astype(np.int32)
        elif isinstance(a, np.ndarray) and a.shape == (N,):
            out[k] = a[ids]
        else:
            out[k] = a
    return out
