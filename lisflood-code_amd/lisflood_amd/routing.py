"""routing -- channel routing module in the reference's HydroModule shape
(src/lisflood/hydrological_modules/routing.py), with the per-sub-step work of `dynamic()` executed on
MI355X: sideflow assembly, 1 (single) or 2 (split) kinematic-wave router calls, the volume <-> discharge
fix-ups, sumDisDay, FlowVelocity/TravelDistance (routing.py:435-706).

All state lives on `var` under the reference's attribute names.  Two ways to run:
  * drop-in (default): every `dynamic(s)` uploads the state vectors it reads from `var`, runs the fused
    device sub-step and writes the results back into `var` -- other (CPU) modules may touch `var`
    between sub-steps exactly as in the reference;
  * resident: `begin_step()` uploads once, `dynamic(s)` only uploads the sideflow, `end_step()`
    downloads -- the NoRoutSteps x (1..2) router calls of a model step never leave the GPU.

What is NOT here (out of this round's scope): `initial()`'s PCRaster LDD preprocessing
and map loading (routing.py:61-339), lakes/reservoirs/polder/inflow/transmission in-loop calls
(routing.py:441-450) -- pass them as `inloop_modules` callables and they are invoked at the same place.
"""
import ctypes as C
import types

import numpy as np

from ._lib import DeviceArray, check, f64, lib, u8
from .hydro_module import HydroModule
from .kinematic_wave_parallel import kinematicWave

_STATIC = ("ChanLength InvChanLength ChannelAlpha InvChannelAlpha ChannelAlpha2 InvChannelAlpha2 Chan2M3Start "
           "Chan2QStart M3Limit QLimit PixelArea IsChannelKinematic").split()
_STATE = "ChanQKin ChanM3Kin Chan2QKin Chan2M3Kin CrossSection2Area Sideflow1Chan ChanQ sumDisDay".split()
_OUT = "FlowVelocity TravelDistance".split()


class _SubstepArgs(C.Structure):  # lf_substep_args, include/lisflood_amd.h
    _fields_ = ([(k, C.c_void_p) for k in _STATIC] + [("SideflowChanM3", C.c_void_p)] +
                [(k, C.c_void_p) for k in _STATE + _OUT] + [("scratch0", C.c_void_p), ("scratch1", C.c_void_p)] +
                [("Beta", C.c_double), ("InvBeta", C.c_double), ("InvDtRouting", C.c_double), ("DtSec", C.c_double),
                 ("split", C.c_int32), ("engine_order", C.c_int32)])


_INLOOP_PTRS = (
    "ChanQ n_lakes lake_cell lake_ups_ptr lake_ups_idx LakeFactor LakeFactorSqr LakeAreaCC LakeStorageM3CC "
    "LakeInflowOldCC LakeOutflowCC LakeStorageM3BalanceCC LakeLevelCC LakeInflowCC QLakeOutM3Dt n_res res_cell "
    "res_ups_ptr res_ups_idx TotalReservoirStorageM3CC MinReservoirOutflowCC NormalReservoirOutflowCC "
    "NonDamagingReservoirOutflowCC ConservativeStorageLimitCC NormalStorageLimitCC FloodStorageLimitCC "
    "Normal_FloodStorageLimitCC DeltaO DeltaLN DeltaNFL ReservoirStorageM3CC ReservoirFillCC ReservoirInflowCC "
    "QResOutM3Dt QInM3Old QDelta QInDt QinADDEDM3 UpTrans TransLossM3Dt TransCum").split()


class _InloopArgs(C.Structure):  # lf_inloop_args, include/lisflood_amd.h
    _fields_ = ([(k, C.c_int64 if k in ("n_lakes", "n_res") else C.c_void_p) for k in _INLOOP_PTRS] +
                [("TransPower1", C.c_double), ("TransPower2", C.c_double), ("TransSub", C.c_double)] +
                [(k, C.c_void_p) for k in ("ToChanM3RunoffDt", "EvaAddM3Dt", "WUseAddM3Dt", "ChannelToPolderM3Dt",
                                           "SideflowChanM3")] +
                [("DtRouting", C.c_double), ("InvNoRoutSteps", C.c_double), ("N", C.c_int64), ("step", C.c_int32)])


_LAKE_PARAM = "LakeFactor LakeFactorSqr LakeAreaCC".split()
_LAKE_STATE = "LakeStorageM3CC LakeInflowOldCC LakeOutflowCC LakeStorageM3BalanceCC LakeLevelCC LakeInflowCC".split()
_RES_PARAM = ("TotalReservoirStorageM3CC MinReservoirOutflowCC NormalReservoirOutflowCC NonDamagingReservoirOutflowCC "
              "ConservativeStorageLimitCC NormalStorageLimitCC FloodStorageLimitCC Normal_FloodStorageLimitCC DeltaO "
              "DeltaLN DeltaNFL").split()
_RES_STATE = "ReservoirStorageM3CC ReservoirFillCC ReservoirInflowCC".split()


class routing(HydroModule):
    input_files_keys = {'all': ['beta', 'ChanLength', 'Ldd', 'Channels', 'ChanGrad', 'ChanGradMin', 'CalChanMan',
                                'ChanMan', 'ChanBottomWidth', 'ChanDepthThreshold', 'ChanSdXdY',
                                'TotalCrossSectionAreaInitValue', 'PrevDischarge'],
                        'SplitRouting': ['CrossSection2AreaInitValue', 'PrevSideflowInitValue', 'CalChanMan2'],
                        'dynamicWave': ['ChannelsDynamic']}   # routing.py:44-50
    module_name = 'Routing'

    def __init__(self, routing_variable, split_routing=False, init_lisflood=False, options=None, device=0,
                 inloop_modules=(), engine_order=True, compact=False):
        """engine_order=True (the default since round 4) keeps the module's device vectors in the router's sweep order
        (permuted on upload and download -- `var.*` is pixel order as always --, site lists of the structures mapped to
        positions) and runs each sub-step as ONE level sweep that updates both routers of a cell
        (lf_routing_substeps_fused with one sub-step): half the launches of the pixel-order path under split routing and
        contiguous upstream reads; results are bit-identical.  engine_order=False: device vectors in pixel order, two
        router calls per sub-step that gather / scatter through the permutation (2.3 x slower per call at 10000^2)."""
        self.engine_order = bool(engine_order)
        # compact (engine order only): pixels whose sub-step is the identity for the whole run (hotpath.inert_pixels:
        # isolated, not a channel pixel, regular parameters, zero thresholds and state, no structure) are left out of
        # the router's domain; their vectors stay zero.  Decided in attach_router from the state `var` holds then.
        self.compact = bool(compact) and self.engine_order
        self._nfull = None
        self.var = routing_variable
        self.options = dict(options or {})
        self.options.setdefault("SplitRouting", split_routing)
        self.options.setdefault("InitLisflood", init_lisflood)
        self.device = device
        self.inloop_modules = tuple(inloop_modules)
        self.river_router = None
        self._dev = {}
        self._resident = False

    # ------------------------------------------------------------------------------------------
    def initial(self, maps, land_mask):
        """routing.initial (routing.py:61-339) from arrays the caller has loaded (map / netCDF loading and the
        settings layer are outside this engine).  `maps`: dict of compressed vectors / scalars keyed by the
        reference's binding names: beta, ChanLength, Ldd, Channels, ChanGrad, ChanGradMin, CalChanMan, ChanMan,
        ChanBottomWidth, ChanDepthThreshold, ChanSdXdY, PixelArea and optionally TotalCrossSectionAreaInitValue,
        PrevDischarge, CrossSection2AreaInitValue, PrevSideflowInitValue (-9999 / missing = cold start).
        PCRaster's LDD operations run on the device (lisflood_amd.ldd: lddrepair_device, lddmask_device, LddDevice)."""
        from . import ldd as L
        v, o = self.var, self.options
        g = lambda k, d=None: maps[k] if k in maps else d
        self._maps = maps
        N = int(np.asarray(land_mask, bool).sum())
        zero = np.zeros(N)
        v.avgdis = zero.copy()
        v.Beta = float(g('beta'))                                               # :66
        v.InvBeta = 1 / v.Beta
        v.ChanLength = np.broadcast_to(np.asarray(g('ChanLength'), float), (N,)).copy()
        v.InvChanLength = 1 / v.ChanLength
        v.NoRoutSteps = int(np.maximum(1, round(v.DtSec / v.DtSecChannel, 0)))  # :73
        if o.get('InitLisflood'):
            v.NoRoutSteps = 1
        v.DtRouting = v.DtSec / v.NoRoutSteps
        v.InvDtRouting = 1 / v.DtRouting
        v.InvNoRoutSteps = 1 / float(v.NoRoutSteps)
        codes = np.asarray(g('Ldd'), float)
        v.PixelArea = np.broadcast_to(np.asarray(g('PixelArea'), float), (N,)).copy()
        v.Ldd = L.lddrepair_device(codes, land_mask, self.device)              # lddmask(Ldd, MaskMap), :90
        kw_all = L.LddDevice(v.Ldd, land_mask, self.device)
        v.UpArea = kw_all.accuflux(v.PixelArea)                                 # :98
        v.InvUpArea = 1 / v.UpArea
        v.IsChannel = np.asarray(g('Channels')).astype(bool)                    # :107-108
        v.IsChannelKinematic = v.IsChannel.copy()
        v.IsStructureKinematic = np.zeros(N, bool)
        ldd_chan_codes, chan_mask = L.lddmask_device(v.Ldd, land_mask, v.IsChannel, self.device)   # :118
        v.LddKinematic = np.zeros(N)                                            # non-channel cells: code 0 (no flow)
        v.LddKinematic[v.IsChannel] = ldd_chan_codes
        v.LddToChan = L.lddrepair_device(np.where(v.IsChannel, L.PIT, v.Ldd), land_mask, self.device)  # :125
        v.AtLastPointC = v.Ldd == L.PIT                                         # boolean(pit(Ldd)), :127,155-156
        v.downstruct = L.downstruct(v.LddKinematic, land_mask)                  # :159-164 (an index vector: host)
        v.Catchments = kw_all.catchment(L.uniqueid(v.AtLastPointC)).astype(np.int32)   # :168-171
        CatchArea = np.bincount(v.Catchments, weights=v.PixelArea)[v.Catchments]
        v.InvCatchArea = 1 / CatchArea
        # channel geometry, :184-199
        v.ChanGrad = np.maximum(g('ChanGrad'), g('ChanGradMin'))
        v.CalChanMan = np.asarray(g('CalChanMan'), float)
        v.ChanMan = v.CalChanMan * g('ChanMan')
        v.ChanBottomWidth = np.asarray(g('ChanBottomWidth'), float)
        depth, sdxdy = np.asarray(g('ChanDepthThreshold'), float), np.asarray(g('ChanSdXdY'), float)
        v.ChanUpperWidth = v.ChanBottomWidth + 2 * sdxdy * depth
        v.TotalCrossSectionAreaBankFull = 0.5 * depth * (v.ChanUpperWidth + v.ChanBottomWidth)
        half = 0.5 * v.TotalCrossSectionAreaBankFull
        init = np.broadcast_to(np.asarray(g('TotalCrossSectionAreaInitValue', -9999.0), float), (N,))
        v.TotalCrossSectionArea = np.where(init == -9999, half, init)           # :203-204
        if o.get('SplitRouting'):
            c2 = np.broadcast_to(np.asarray(g('CrossSection2AreaInitValue', -9999.0), float), (N,))
            v.CrossSection2Area = np.where(c2 == -9999, zero, c2)               # :210-212
            ps = np.broadcast_to(np.asarray(g('PrevSideflowInitValue', -9999.0), float), (N,))
            v.Sideflow1Chan = np.where(ps == -9999, zero, ps)                   # :216-218
        # channel alpha, :227-236
        d_alpha = np.where(v.IsChannel, 0.5 * depth, 0.0)
        v.ChanWettedPerimeterAlpha = v.ChanBottomWidth + 2 * np.sqrt(np.square(d_alpha) + np.square(d_alpha * sdxdy))
        AlpTermChan = (v.ChanMan / (np.sqrt(v.ChanGrad))) ** v.Beta
        v.AlpPow = 2.0 / 3.0 * v.Beta
        v.ChannelAlpha = (AlpTermChan * (v.ChanWettedPerimeterAlpha ** v.AlpPow)).astype(float)
        with np.errstate(divide="ignore"):
            v.InvChannelAlpha = 1 / v.ChannelAlpha
        # initial volume and discharge, :243-248, 326-327
        v.ChanM3 = v.TotalCrossSectionArea * v.ChanLength
        v.ChanIniM3 = v.ChanM3.copy()
        v.ChanM3Kin = v.ChanIniM3.copy().astype(float)
        with np.errstate(divide="ignore", invalid="ignore"):
            v.ChanQKin = np.where(v.ChannelAlpha > 0, (v.TotalCrossSectionArea / v.ChannelAlpha) ** v.InvBeta, 0).astype(float)
        v.CumQ = zero.copy()
        prev = np.broadcast_to(np.asarray(g('PrevDischarge', -9999.0), float), (N,))
        v.ChanQ = np.where(prev == -9999, v.ChanQKin, prev)
        v.DischargeM3Out, v.TotalQInM3, v.sumDis, v.sumInWB = zero.copy(), zero.copy(), zero.copy(), zero.copy()
        self._land_mask = np.asarray(land_mask, bool)
        self._ldd_all = kw_all      # full-LDD operations stay available (repMBTs catchment totals)

    def step_end(self, time_since_start=None):
        """What Lisflood_dynamic.py:194-229 does after the sub-step loop: ChanM3, TotalCrossSectionArea, sumDis,
        ChanQAvg (= `dis` of the reference's outputs), avgdis, DischargeM3Out."""
        v, o = self.var, self.options
        if o.get('InitLisflood') or not o.get('SplitRouting'):
            v.ChanM3 = v.ChanM3Kin.copy()
        else:
            v.ChanM3 = v.ChanM3Kin + v.Chan2M3Kin - v.Chan2M3Start
        v.TotalCrossSectionArea = v.ChanM3 * v.InvChanLength
        v.sumDis = getattr(v, "sumDis", 0.0) + v.sumDisDay
        v.ChanQAvg = v.sumDisDay / v.NoRoutSteps
        if (o.get('InitLisflood') or o.get('repAverageDis')) and time_since_start:
            v.CumQ = getattr(v, "CumQ", 0.0) + v.ChanQ
            v.avgdis = v.CumQ / time_since_start
        if hasattr(v, "AtLastPointC"):
            v.DischargeM3Out = getattr(v, "DischargeM3Out", 0.0) + np.where(v.AtLastPointC, v.ChanQ * v.DtSec, 0)

    def structure_links(self, compressed_ldd_kinematic):
        """[N] int64, -1 = none: for every pit of the (cut) kinematic LDD that drains into a lake or reservoir cell
        in the uncut LDD (`downstruct`, routing.py:159-164), that cell."""
        v, o = self.var, self.options
        N = np.asarray(compressed_ldd_kinematic).size
        site = np.zeros(N + 1, bool)
        if o.get("simulateLakes") and not o.get("InitLisflood"):
            site[np.asarray(v.LakeIndex).astype(np.int64)] = True
        if o.get("simulateReservoirs") and not o.get("InitLisflood"):
            site[np.asarray(v.ReservoirIndex).astype(np.int64)] = True
        ds = np.minimum(np.asarray(v.downstruct).astype(np.int64), N)
        feeds = site[ds] & (ds < N)
        return np.where(feeds, ds, -1)

    def attach_router(self, compressed_ldd_kinematic, land_mask, flagnancheck=False):
        """The router construction of initialSecond (routing.py:401-403).  In engine order with lakes or reservoirs
        switched on, the graph also carries their uncut links (Graph(virtual_down=...)), so that dynamic_fused()
        can run the structures inside the wavefront."""
        v, o = self.var, self.options
        land_mask = np.asarray(land_mask, bool)
        self._land_mask = land_mask
        codes = np.asarray(compressed_ldd_kinematic, np.float64)
        N = self._nfull = int(land_mask.sum())
        structures = not o.get("InitLisflood") and (o.get("simulateLakes") or o.get("simulateReservoirs")) and \
            hasattr(v, "downstruct")
        ids = np.arange(N)
        if self.compact and all(hasattr(v, k) for k in _STATE[:-1]):
            from .hotpath import inert_pixels
            vals = {k: getattr(v, k) for k in _STATIC + _STATE if hasattr(v, k)}
            st = {k: getattr(v, k) for k, opt in (("LakeIndex", "simulateLakes"), ("ReservoirIndex", "simulateReservoirs"),
                                                  ("QInM3Old", "inflow"), ("QDelta", "inflow"), ("TransCum", "TransLoss"))
                  if o.get(opt) and hasattr(v, k)}
            drop = inert_pixels(vals, codes, land_mask, self._split(), st or None)
            if drop.sum() >= 0.1 * N:
                ids = np.nonzero(~drop)[0]
        self._ids = ids
        sub = lambda a: np.broadcast_to(np.asarray(a), (N,))[ids]
        mask = land_mask
        if ids.size < N:
            keep = np.zeros(N, bool)
            keep[ids] = True
            mask = np.zeros(land_mask.shape, bool)
            mask[land_mask] = keep
        graph = None
        if self.engine_order and structures:
            from .kinematic_wave_parallel import Graph
            vd = self.structure_links(codes)
            new_id = np.full(N, -1, np.int64); new_id[ids] = np.arange(ids.size)
            vd = np.where(vd >= 0, new_id[np.maximum(vd, 0)], -1)[ids]
            graph = Graph(codes[ids], mask, virtual_down=vd)
        a2 = getattr(v, "ChannelAlpha2", None)
        self.river_router = kinematicWave(codes[ids], mask, sub(v.ChannelAlpha), v.Beta, sub(v.ChanLength), v.DtRouting,
                                          alpha_floodplains=None if a2 is None else sub(a2),
                                          flagnancheck=flagnancheck, device=self.device, graph=graph)
        if self.engine_order:
            self._perm = ids[self.river_router.graph.layout()[0].astype(np.int64)]     # position -> pixel
            self._pos = np.full(N, -1, np.int64)
            self._pos[self._perm] = np.arange(self._perm.size)                        # pixel -> position (-1: left out)
        return self.river_router

    def _up(self, x):
        """host [N] vector in pixel order -> the order (and domain) the device vectors are kept in"""
        return x[self._perm] if self.engine_order else x

    def _down(self, a, out=None):
        """device-order vector -> pixel order (into `out` when given); pixels outside a compact domain get 0"""
        if not self.engine_order:
            if out is None:
                return a
            out[...] = a
            return out
        if out is None:
            out = np.zeros(self._nfull, a.dtype)
        elif self._perm.size < self._nfull:
            out[...] = 0
        out[self._perm] = a
        return out

    def initialSecond(self, compressed_ldd_kinematic=None, land_mask=None, flagnancheck=False):
        """Split-routing start values (routing.py:355-397) + router (401-403).  The one-hop upstream sum of
        QLimit (PCRaster `upstream`, routing.py:387) runs on the device graph."""
        v = self.var
        split = self.options["SplitRouting"]
        if compressed_ldd_kinematic is None:
            compressed_ldd_kinematic, land_mask = v.LddKinematic, self._land_mask
        if split:
            if getattr(v, "ChannelAlpha2", None) is None:
                cal2 = getattr(v, "CalChanMan2", None)
                if cal2 is None:
                    cal2 = getattr(self, "_maps", {}).get("CalChanMan2")         # loadmap('CalChanMan2'), :355
                if cal2 is None:
                    raise ValueError("SplitRouting needs var.ChannelAlpha2 or CalChanMan2 (routing.py:355-358)")
                ChanMan2 = (v.ChanMan / v.CalChanMan) * np.asarray(cal2, np.float64)                    # :355
                v.ChannelAlpha2 = ((ChanMan2 / np.sqrt(v.ChanGrad)) ** v.Beta) * (v.ChanWettedPerimeterAlpha ** v.AlpPow)
            with np.errstate(divide="ignore"):
                v.InvChannelAlpha2 = 1 / v.ChannelAlpha2
        self.attach_router(compressed_ldd_kinematic, land_mask, flagnancheck)
        if split and not self.options["InitLisflood"]:
            maps = getattr(self, "_maps", {})
            if "AvgDis" in maps:                                                                    # :364
                v.QLimit = np.asarray(maps["AvgDis"], np.float64) * np.asarray(maps.get("QSplitMult", 2.0), np.float64)
            v.M3Limit = v.ChannelAlpha * v.ChanLength * (v.QLimit ** v.Beta)                       # :371
            v.Chan2M3Start = v.ChannelAlpha2 * v.ChanLength * (v.QLimit ** v.Beta)                 # :384
            ups = np.zeros(self._nfull)
            ups[self._ids] = self.river_router.upstream_sum(np.broadcast_to(v.QLimit, (self._nfull,))[self._ids])
            v.Chan2QStart = v.QLimit - ups                                                         # :387
            v.Chan2M3Kin = v.CrossSection2Area * v.ChanLength + v.Chan2M3Start                     # :391
            v.ChanM3Kin = v.ChanM3 - v.Chan2M3Kin + v.Chan2M3Start                                 # :392
            v.ChanM3Kin = np.where((v.ChanM3Kin < 0.0) & (v.ChanM3Kin > -0.0000001), 0.0, v.ChanM3Kin)  # :394
            v.Chan2QKin = (v.Chan2M3Kin * v.InvChanLength * v.InvChannelAlpha2) ** v.InvBeta      # :396
            v.ChanQKin = (v.ChanM3Kin * v.InvChanLength * v.InvChannelAlpha) ** v.InvBeta          # :397
        if self.options.get("repMBTs"):
            self._mbts_initial()

    def _ldd_device(self):
        d = getattr(self, "_ldd_all", None)
        if d is None:
            from . import ldd as L
            d = self._ldd_all = L.LddDevice(self.var.Ldd, self._land_mask, self.device)
        return d

    def _catchment_totals(self, w):
        """np.take(np.bincount(Catchments, weights=w), Catchments): totals over the trees of the full LDD, on the device"""
        return self._ldd_device().catchment_totals(w)

    def _catchment_totals_multi(self, ws):
        """several totals in one device pass (lf_catchment_totals_multi_host)"""
        return self._ldd_device().catchment_totals_multi(ws)

    def _mbts_added(self, s, nsub=1, qin=None):
        """AddedTRUN (routing.py:483-499): water added to the channels, summed per catchment and over the sub-steps.
        nsub > 1: the sub-steps of a fused call at once (the terms are constant over the model step; `qin` = QinADDEDM3,
        the inflow of all sub-steps)."""
        v, o = self.var, self.options
        w = np.array(v.ToChanM3RunoffDt, dtype=np.float64) * nsub
        if o.get('inflow'):
            w = w + (np.asarray(v.QInDt, np.float64) if qin is None else np.asarray(qin, np.float64))
        if o.get('openwaterevapo'):
            w = w - np.asarray(v.EvaAddM3Dt, np.float64) * nsub
        if o.get('wateruse'):
            w = w - np.asarray(v.WUseAddM3Dt, np.float64) * nsub
        tot = self._catchment_totals(w)
        v.AddedTRUN = tot if s < 1 else v.AddedTRUN + tot

    def _mbts_last(self):
        """mass-balance error of the split-routing module at the last sub-step (routing.py:645-691)"""
        v, o = self.var, self.options
        if o.get('InitLisflood') or not o.get('SplitRouting'):
            return                                  # the reference's code for these cases is commented out (:608-643)
        N = self._nfull
        at_last = np.asarray(v.AtLastPointC).astype(bool)
        avg = v.sumDisDay / v.NoRoutSteps
        # the up to four catchment totals of this block (:649-683) in ONE device pass
        terms = [np.where(at_last, avg, 0.0) * v.DtSec]                                                   # :649-651
        storage = v.ChanM3Kin + v.Chan2M3Kin - v.Chan2M3Start                                             # :654
        ups = np.asarray(getattr(v, "IsUpsOfStructureKinematicC", np.zeros(N))) > 0
        res_on, lakes_on = bool(o.get('simulateReservoirs')), bool(o.get('simulateLakes'))
        if res_on:
            storage = storage + v.ReservoirStorageM3                                                      # :664
        if lakes_on:
            storage = storage + v.LakeStorageM3Balance                                                    # :672
        terms.append(storage)
        if res_on or lakes_on:
            terms.append(np.where(ups, v.ChanQ * v.DtRouting, 0.0))                                       # :666 / :674
        if lakes_on:
            lake = np.zeros(N)
            lake[np.asarray(v.LakeIndex)] = 0.5 * np.asarray(v.LakeInflowCC) * v.DtRouting                # :676
            terms.append(lake)
        tot = self._catchment_totals_multi(terms)
        out_step, storage1 = tot[0], tot[1]                                                               # :651, :683
        r = np.zeros(N)
        if lakes_on:                    # (the lake branch overwrites the reservoir one, as in the reference)
            r = tot[2] + tot[3] - v.DischargeM3StructuresIni                                              # :674-679
        elif res_on:
            r = tot[2] - v.DischargeM3StructuresIni                                                       # :666-668
        v.MBErrorSplitRoutingM3 = -storage1 + v.StorageStepINIT - out_step - r + v.AddedTRUN              # :685
        corr = np.where(at_last, v.MBErrorSplitRoutingM3 / v.DtRouting, 0.0)                              # :687-688
        v.OutletDischargeErrorSplitRouting = self._catchment_totals(corr)
        v.StorageStepINIT = storage1 + r                                                                  # :691

    def _mbts_initial(self):
        """mass-balance start values of option repMBTs (routing.py:405-431; with split routing the reference takes
        DischargeM3StructuresIni from waterbalance.initial, waterbalance.py:91-109 -- the same formula)"""
        v, o = self.var, self.options
        N = self._nfull
        zero = np.zeros(N)
        lakes_on, res_on = bool(o.get("simulateLakes")), bool(o.get("simulateReservoirs"))
        init, split = bool(o.get("InitLisflood")), bool(o.get("SplitRouting"))
        store = np.array(v.ChanM3Kin, dtype=np.float64)
        if not init and split:
            store = v.ChanM3Kin + v.Chan2M3Kin - v.Chan2M3Start                                  # :426
        if res_on:
            store = store + v.ReservoirStorageIniM3
        if lakes_on:
            store = store + v.LakeStorageIniM3
        if init:
            v.DischargeM3StructuresIni = zero.copy()                                             # :407
            v.StorageStepINIT = self._catchment_totals(store)                                    # :412
            return
        dis = np.where(np.asarray(getattr(v, "IsUpsOfStructureKinematicC", zero)) > 0, v.ChanQ * v.DtRouting, 0.0)   # :415
        if lakes_on:
            dis = dis + np.where(np.asarray(getattr(v, "IsUpsOfStructureLake", zero)) > 0, 0.5 * v.ChanQ * v.DtRouting, 0.0)
        v.DischargeM3StructuresIni = self._catchment_totals(dis)                                 # :424 / waterbalance.py:109
        # the reference totals StorageStepINIT per catchment only in the split-routing branch (:431 vs :417-423)
        v.StorageStepINIT = self._catchment_totals(store) if split else store

    # ------------------------------------------------------------------------------------------
    def _split(self):
        return bool(self.options["SplitRouting"]) and not self.options["InitLisflood"]   # routing.py:518

    def _ensure_device(self):
        v = self.var
        N, Nk = self._nfull, self.river_router.num_pixels      # host vectors / device vectors (compact domain)
        if "scratch0" in self._dev:
            return
        zeros = np.zeros(N)
        for k in _STATIC:
            a = getattr(v, k, None)
            if a is None:
                a = np.ones(N, bool) if k == "IsChannelKinematic" else zeros
            a = self._up(np.broadcast_to(a, (N,)))
            a = u8(a) if k == "IsChannelKinematic" else f64(a)
            self._dev[k] = DeviceArray.from_host(a, self.device)
        for k in _STATE + _OUT + ["SideflowChanM3", "scratch0", "scratch1"]:
            self._dev[k] = DeviceArray(max(Nk, 1), np.float64, self.device).zero()
        a = self._args = _SubstepArgs()
        for k, d in self._dev.items():
            setattr(a, k, d.ptr.value)
        a.Beta, a.InvBeta, a.InvDtRouting, a.DtSec = float(v.Beta), float(v.InvBeta), float(v.InvDtRouting), float(v.DtSec)
        a.engine_order = 1 if self.engine_order else 0

    def _upload_state(self):
        v = self.var
        N = self._nfull
        for k in _STATE:
            a = getattr(v, k, None)
            if a is None:
                a = np.zeros(N)
                setattr(v, k, a)
            self._dev[k].upload(f64(self._up(np.broadcast_to(a, (N,)))))

    def _download_state(self):
        v = self.var
        split = self._split()
        names = _STATE + _OUT if split else ["ChanQKin", "ChanM3Kin", "ChanQ", "sumDisDay"] + _OUT
        for k in names:
            cur = getattr(v, k, None)
            inplace = isinstance(cur, np.ndarray) and cur.dtype == np.float64 and cur.flags.c_contiguous and \
                cur.size == self._nfull and cur.flags.writeable
            if self.engine_order:
                a = self._dev[k].download()
                if inplace:
                    self._down(a, cur)
                else:
                    setattr(v, k, self._down(a))
            elif inplace:
                self._dev[k].download(cur)          # in place, like the numba kernels
            else:
                setattr(v, k, self._dev[k].download())

    def begin_step(self):
        """resident mode: upload the routing state once per model step."""
        self._ensure_device()
        self._upload_state()
        self._resident = True

    def end_step(self):
        self._download_state()
        self._resident = False

    def sideflow_m3(self):
        """SideflowChanM3 assembly, routing.py:462-478 (each term option-gated)."""
        v, o = self.var, self.options
        s = np.array(v.ToChanM3RunoffDt, dtype=np.float64, copy=True)
        if o.get('openwaterevapo'):
            s -= v.EvaAddM3Dt
        if o.get('wateruse'):
            v.WUseAddM3Dt = v.withdrawal_CH_actual_M3_routStep - v.returnflow_GwAbs2Channel_M3_routStep
            s -= v.WUseAddM3Dt
        if o.get('inflow'):
            s += v.QInDt
        if o.get('TransLoss'):
            s -= v.TransLossM3Dt
        if not o.get('InitLisflood'):
            if o.get('simulateLakes'):
                s += v.QLakeOutM3Dt
            if o.get('simulateReservoirs'):
                s += v.QResOutM3Dt
            if o.get('simulatePolders'):
                s -= v.ChannelToPolderM3Dt
        return s

    # ------------------------------------------------------------------------------------------
    def attach_structures(self):
        """Lakes, reservoirs, inflow hydrographs and transmission loss inside the sub-step loop (routing.py:441-450),
        on the device.  Reads the reference's attribute names from `var` according to the options
        simulateLakes / simulateReservoirs / inflow / TransLoss: LakeIndex, LakeFactor, LakeFactorSqr, LakeAreaCC,
        LakeStorageM3, LakeInflowOldCC, LakeOutflowCC, LakeStorageM3BalanceCC, LakeLevelCC (lakes.py);
        ReservoirIndex, the eleven *CC parameter vectors, ReservoirStorageM3 (reservoir.py); QInM3Old, QDelta
        (inflow.py); UpTrans, TransPower1/2, TransSub, TransCum (transmission.py); and `downstruct` of the UNCUT
        kinematic LDD (routing.py:159-164) for the structures' inflow."""
        v, o = self.var, self.options
        N, Nk = self._nfull, self.river_router.num_pixels        # host vectors / device vectors (compact domain)
        self._ensure_device()
        check(lib().lf_router_reset_site_cache(self.river_router._h))    # new site lists: their levels are re-validated
        st = self._st = dict(dev={}, lakes=0, res=0)
        a = self._inloop = _InloopArgs()
        ds = np.asarray(v.downstruct).astype(np.int64)
        order = np.argsort(ds, kind="stable")               # sources grouped by target, ascending source id
        starts = np.searchsorted(ds[order], np.arange(N + 1))

        def site_csr(cells):
            ptr = np.zeros(len(cells) + 1, np.int32)
            idx = []
            for i, c in enumerate(cells):
                u = order[starts[c]:starts[c + 1]]
                idx.append(u)
                ptr[i + 1] = ptr[i] + u.size
            idx = np.concatenate(idx).astype(np.int64) if idx else np.zeros(0, np.int64)
            if self.engine_order:                      # same summation order (ascending pixel id), device positions
                idx = self._pos[idx]
                if (idx < 0).any():
                    raise ValueError("a pixel draining into a structure was left out of the compact domain")
            idx = idx.astype(np.int32)
            return ptr, (idx if idx.size else np.zeros(1, np.int32))

        def cell_ids(cells):
            cells = np.asarray(cells).astype(np.int64)
            return (self._pos[cells] if self.engine_order else cells).astype(np.int32)

        def put(name, arr):
            st["dev"][name] = DeviceArray.from_host(np.ascontiguousarray(arr), self.device)
            setattr(a, name, st["dev"][name].ptr.value)

        if o.get("simulateLakes") and not o.get("InitLisflood"):
            cells = np.asarray(v.LakeIndex).astype(np.int32)
            st["lakes"] = a.n_lakes = cells.size
            ptr, idx = site_csr(cells)
            put("lake_cell", cell_ids(cells)); put("lake_ups_ptr", ptr); put("lake_ups_idx", idx)
            for k in _LAKE_PARAM:
                put(k, f64(np.broadcast_to(getattr(v, k), (cells.size,))))
            for k in _LAKE_STATE:
                put(k, f64(np.broadcast_to(getattr(v, k, 0.0), (cells.size,))))
            put("QLakeOutM3Dt", np.zeros(Nk))
        if o.get("simulateReservoirs") and not o.get("InitLisflood"):
            cells = np.asarray(v.ReservoirIndex).astype(np.int32)
            st["res"] = a.n_res = cells.size
            ptr, idx = site_csr(cells)
            put("res_cell", cell_ids(cells)); put("res_ups_ptr", ptr); put("res_ups_idx", idx)
            for k in _RES_PARAM:
                put(k, f64(np.broadcast_to(getattr(v, k), (cells.size,))))
            for k in _RES_STATE:
                put(k, f64(np.broadcast_to(getattr(v, k, 0.0), (cells.size,))))
            put("QResOutM3Dt", np.zeros(Nk))
        if o.get("inflow"):
            put("QInM3Old", f64(self._up(np.asarray(v.QInM3Old)))); put("QDelta", f64(self._up(np.asarray(v.QDelta))))
            put("QInDt", np.zeros(Nk)); put("QinADDEDM3", np.zeros(Nk))
        if o.get("TransLoss"):
            put("UpTrans", u8(self._up(np.asarray(v.UpTrans)))); put("TransLossM3Dt", np.zeros(Nk))
            put("TransCum", f64(self._up(np.broadcast_to(getattr(v, "TransCum", 0.0), (N,)))))
            a.TransPower1, a.TransPower2, a.TransSub = float(v.TransPower1), float(v.TransPower2), float(v.TransSub)
        for k in ("ToChanM3RunoffDt", "EvaAddM3Dt", "WUseAddM3Dt", "ChannelToPolderM3Dt"):
            st["dev"][k] = DeviceArray(max(Nk, 1), np.float64, self.device).zero()
        a.ToChanM3RunoffDt = st["dev"]["ToChanM3RunoffDt"].ptr.value
        a.ChanQ = self._dev["ChanQ"].ptr.value
        a.SideflowChanM3 = self._dev["SideflowChanM3"].ptr.value
        a.DtRouting, a.InvNoRoutSteps, a.N = float(v.DtRouting), float(v.InvNoRoutSteps), Nk

    def _structures_substep(self, s, launch=True):
        v, o, st, a = self.var, self.options, self._st, self._inloop
        N = self._nfull
        if s == 0:      # lakes.py:211-212, reservoir.py:195-196: site state from the dense state maps
            if st["lakes"]:
                st["dev"]["LakeStorageM3CC"].upload(f64(np.asarray(v.LakeStorageM3)[np.asarray(v.LakeIndex)]))
            if st["res"]:
                st["dev"]["ReservoirStorageM3CC"].upload(f64(np.asarray(v.ReservoirStorageM3)[np.asarray(v.ReservoirIndex)]))
            if "QInM3Old" in st["dev"]:      # inflow.dynamic_init recomputes QDelta and the driver moves QInM3Old on
                for k in ("QInM3Old", "QDelta"):    # every model step (inflow.py:108, Lisflood_dynamic.py:185)
                    st["dev"][k].upload(f64(self._up(np.broadcast_to(np.asarray(getattr(v, k), np.float64), (N,)))))
        st["dev"]["ToChanM3RunoffDt"].upload(f64(self._up(np.broadcast_to(v.ToChanM3RunoffDt, (N,)))))
        for k, opt in (("EvaAddM3Dt", "openwaterevapo"), ("WUseAddM3Dt", "wateruse"), ("ChannelToPolderM3Dt", "simulatePolders")):
            if o.get(opt):
                if k == "WUseAddM3Dt":
                    v.WUseAddM3Dt = v.withdrawal_CH_actual_M3_routStep - v.returnflow_GwAbs2Channel_M3_routStep
                st["dev"][k].upload(f64(self._up(np.broadcast_to(getattr(v, k), (N,)))))
                setattr(a, k, st["dev"][k].ptr.value)
        a.step = int(s)
        if launch:
            check(lib().lf_inloop_structures(C.c_int(self.device), C.byref(a)))

    def _structures_download(self, s):
        v, st = self.var, self._st
        names = []
        if st["lakes"]:
            names += _LAKE_STATE + ["QLakeOutM3Dt"]
        if st["res"]:
            names += _RES_STATE + ["QResOutM3Dt"]
        names += [k for k in ("QInDt", "QinADDEDM3", "TransLossM3Dt", "TransCum") if k in st["dev"]]
        dense = ("QLakeOutM3Dt", "QResOutM3Dt", "QInDt", "QinADDEDM3", "TransLossM3Dt", "TransCum")
        for k in names:
            x = st["dev"][k].download()
            setattr(v, k, self._down(x) if k in dense else x)
        if s == v.NoRoutSteps - 1:      # lakes.py:283-292, reservoir.py:311-315: expand to the dense state maps
            N = self._nfull
            if st["lakes"]:
                for dense, cc in (("LakeStorageM3", "LakeStorageM3CC"), ("LakeStorageM3Balance", "LakeStorageM3BalanceCC"),
                                  ("LakeLevel", "LakeLevelCC"), ("LakeInflowOld", "LakeInflowOldCC"),
                                  ("LakeOutflow", "LakeOutflowCC")):
                    d = np.zeros(N); d[np.asarray(v.LakeIndex)] = getattr(v, cc); setattr(v, dense, d)
            if st["res"]:
                for dense, cc in (("ReservoirStorageM3", "ReservoirStorageM3CC"), ("ReservoirFill", "ReservoirFillCC")):
                    d = np.zeros(N); d[np.asarray(v.ReservoirIndex)] = getattr(v, cc); setattr(v, dense, d)

    def dynamic_fused(self, sideflows=None):
        """All NoRoutSteps sub-steps of a model step in one call (the loop of Lisflood_dynamic.py:179-180), run as a
        skewed wavefront over (level, sub-step) on the device -- NL + NoRoutSteps - 1 launches instead of
        NoRoutSteps x (1..2) x NL.  `sideflows`: SideflowChanM3 of the step -- one [N] vector (the model's case
        when nothing in the loop changes it) or [NoRoutSteps, N]; default: assembled from `var` as in dynamic().
        Bit-identical to calling dynamic(0..NoRoutSteps-1).  In-loop modules (lakes, reservoirs...) cannot run
        inside the wavefront; use dynamic() when they are active."""
        if self.river_router is None:
            raise RuntimeError("routing.initialSecond()/attach_router() must be called first")
        if self.inloop_modules:
            raise RuntimeError("dynamic_fused cannot run host in-loop modules; use dynamic() or attach_structures()")
        if getattr(self, "_inloop", None) is not None:
            if not self.engine_order or sideflows is not None:
                raise RuntimeError("structures inside the wavefront need routing(..., engine_order=True) and the "
                                   "sideflow assembled from `var`")
            return self._fused_with_structures()
        _fused(self, self.sideflow_m3() if sideflows is None else sideflows)

    def _fused_with_structures(self):
        """Lakes, reservoirs, inflow and transmission loss inside the wavefront
        (lf_routing_substeps_fused_structures): the loop of Lisflood_dynamic.py:179-180 in one call."""
        v = self.var
        nsteps = int(v.NoRoutSteps)
        self._ensure_device()
        if not self._resident:
            self._upload_state()
        self._structures_substep(0, launch=False)      # site state + the dense terms of the sideflow, once
        self._args.split = 1 if self._split() else 0
        check(lib().lf_routing_substeps_fused_structures(self.river_router._h, C.byref(self._args),
                                                         C.byref(self._inloop), C.c_int(nsteps)))
        if not self._resident:
            self._download_state()
        self._structures_download(nsteps - 1)
        v.SideflowChanM3 = self._down(self._dev["SideflowChanM3"].download())
        if self.options.get("repMBTs"):
            self.mbts_after_fused()

    def mbts_after_fused(self):
        """The mass-balance bookkeeping of a whole model step after a fused call (state already on `var`): AddedTRUN of
        all sub-steps at once, then the last-sub-step block.  Equal to the sub-step-by-sub-step sums to rounding."""
        v = self.var
        nsteps = int(v.NoRoutSteps)
        self._mbts_added(0, nsub=nsteps, qin=getattr(v, "QinADDEDM3", None))
        self._mbts_last()

    def dynamic(self, NoRoutingExecuted):
        """One routing sub-step (routing.py:435-706)."""
        if self.river_router is None:
            raise RuntimeError("routing.initialSecond()/attach_router() must be called first")
        for m in self.inloop_modules:          # lakes / reservoirs / polder / inflow / transmission, :441-450
            m(NoRoutingExecuted)
        self._ensure_device()
        if not self._resident:
            self._upload_state()
        if getattr(self, "_inloop", None) is not None:
            self._structures_substep(NoRoutingExecuted)      # structures + sideflow assembly on the device
        else:
            self._dev["SideflowChanM3"].upload(f64(self._up(self.sideflow_m3())))
        self._args.split = 1 if self._split() else 0
        if self.engine_order:       # one level sweep updating both routers of a cell (one sub-step of the wavefront)
            check(lib().lf_routing_substeps_fused(self.river_router._h, C.byref(self._args), C.c_int(1), C.c_int64(0)))
        else:
            check(lib().lf_routing_substep(self.river_router._h, C.byref(self._args)))
        if not self._resident:
            self._download_state()
        if getattr(self, "_inloop", None) is not None:
            self._structures_download(NoRoutingExecuted)
        if self.options.get("repMBTs"):                  # routing.py:483-499, 645-691
            if self._resident:
                raise RuntimeError("repMBTs reads the routing state on `var` every sub-step: use the drop-in mode or "
                                   "dynamic_fused()")
            self._mbts_added(NoRoutingExecuted)
            if NoRoutingExecuted == int(self.var.NoRoutSteps) - 1:
                self._mbts_last()


def _fused(self, sideflows):
    """See routing.dynamic_fused."""
    v = self.var
    r = self.river_router
    N, Nk = self._nfull, r.num_pixels                         # host vectors / device vectors (compact domain)
    perm = self._perm if self.engine_order else self._ids[r.graph.layout()[0].astype(np.int64)]
    sideflows = np.ascontiguousarray(np.atleast_2d(np.asarray(sideflows, dtype=np.float64)))
    nsteps_in, stride = (sideflows.shape[0], Nk) if sideflows.shape[0] > 1 else (1, 0)
    nsteps = int(v.NoRoutSteps)
    if stride and nsteps_in != nsteps:
        raise ValueError("need one sideflow vector, or NoRoutSteps of them")
    a = _SubstepArgs()
    dev = {}
    zeros = np.zeros(N)
    for k in _STATIC:
        x = getattr(v, k, None)
        if x is None:
            x = np.ones(N, bool) if k == "IsChannelKinematic" else zeros
        x = np.broadcast_to(x, (N,))[perm]
        dev[k] = DeviceArray.from_host(u8(x) if k == "IsChannelKinematic" else f64(x), self.device)
    for k in _STATE:
        x = getattr(v, k, None)
        dev[k] = DeviceArray.from_host(f64(np.broadcast_to(zeros if x is None else x, (N,))[perm]), self.device)
    for k in _OUT + ["scratch0", "scratch1"]:
        dev[k] = DeviceArray(max(Nk, 1), np.float64, self.device).zero()
    dev["SideflowChanM3"] = DeviceArray.from_host(np.ascontiguousarray(sideflows[:, perm]), self.device)
    for k, d in dev.items():
        setattr(a, k, d.ptr.value)
    a.Beta, a.InvBeta, a.InvDtRouting, a.DtSec = float(v.Beta), float(v.InvBeta), float(v.InvDtRouting), float(v.DtSec)
    a.split = 1 if self._split() else 0
    a.engine_order = 1
    check(lib().lf_routing_substeps_fused(r._h, C.byref(a), C.c_int(nsteps), C.c_int64(stride)))
    names = _STATE + _OUT if self._split() else ["ChanQKin", "ChanM3Kin", "ChanQ", "sumDisDay"] + _OUT
    for k in names:
        out = np.zeros(N)                # pixels outside a compact domain keep their zero state
        out[perm] = dev[k].download()[:Nk]
        setattr(v, k, out)
    for d in dev.values():
        d.free()


def var_from_fixture(g):
    """Build the `var` namespace routing.dynamic needs from a tests/golden/substep_*.npz fixture."""
    v = types.SimpleNamespace()
    for k in ("ChannelAlpha", "ChannelAlpha2", "ChanLength", "PixelArea", "IsChannelKinematic", "QLimit", "M3Limit",
              "Chan2M3Start", "Chan2QStart"):
        setattr(v, k, g[k])
    v.Beta = float(g["Beta"]); v.InvBeta = 1 / v.Beta
    v.DtRouting = float(g["DtRouting"]); v.InvDtRouting = 1 / v.DtRouting
    v.NoRoutSteps = int(g["NoRoutSteps"]); v.DtSec = v.DtRouting * v.NoRoutSteps
    v.InvChanLength, v.InvChannelAlpha, v.InvChannelAlpha2 = 1 / v.ChanLength, 1 / v.ChannelAlpha, 1 / v.ChannelAlpha2
    for k in ("ChanQKin", "ChanM3Kin", "Chan2QKin", "Chan2M3Kin", "CrossSection2Area", "Sideflow1Chan"):
        setattr(v, k, g["init_" + k].copy())
    v.sumDisDay = np.zeros(v.ChanQKin.size)
    v.ChanQ = v.ChanQKin.copy()
    return v
