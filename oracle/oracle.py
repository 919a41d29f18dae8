"""ctypes front-end of the CPU oracle (oracle/lf_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module -- as the
checker / the reported CPU baseline, never as the path that is measured or shipped.  The product
(lisflood-code_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "liblf_oracle.so")

_lib = None
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")


def build(force=False):
    src = os.path.join(HERE, "lf_oracle.c")
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "liblf_oracle.so"], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB):
            build()
        # GPU boxes expose hundreds of hardware threads but only a cgroup share of them: an unbounded
        # OpenMP team is slower than one thread there.  Default to a small team unless the caller chose.
        os.environ.setdefault("OMP_NUM_THREADS", str(min(8, os.cpu_count() or 1)))
        _lib = C.CDLL(LIB)
        _lib.lfo_lookups.restype = C.c_int
        _lib.lfo_orders.restype = C.c_int64
    return _lib


def set_threads(n):
    """OpenMP team size of the oracle's parallel loops (numba's set_num_threads, Lisflood_initial.py:101-104)."""
    lib()
    import ctypes.util
    C.CDLL(ctypes.util.find_library("gomp") or "libgomp.so.1").omp_set_num_threads(int(n))


def first_touch(a):
    """A copy of `a` (any 8-byte dtype, C-contiguous) whose pages are first written by the OpenMP team with the static
    schedule of the oracle's loops (lfo_parallel_copy): on a multi-socket host every thread's share of the vector then
    lies in its own NUMA node.  numpy's own allocation is touched by the calling thread alone.  bench baseline only."""
    a = np.ascontiguousarray(a)
    assert a.dtype.itemsize == 8
    out = np.empty(a.shape, a.dtype)                      # large allocation: mmap'ed, pages untouched until written
    lib().lfo_parallel_copy(_ptr(out.view(np.float64)), _ptr(a.view(np.float64)), C.c_int64(a.size))
    return out


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _u8(a):
    return np.ascontiguousarray(np.asarray(a).astype(np.uint8))


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def lookups(codes, mask):
    """-> downstream_lookup[N] float64, upstream_lookup[N,K] int64 (kinematic_wave_parallel.py:73-90)."""
    mask = np.asarray(mask, bool)
    H, W = mask.shape
    N = int(mask.sum())
    codes = _f(codes)
    m8 = _u8(mask)
    down = np.empty(N, np.int64)
    ups = np.empty((N, 8), np.int64)
    nups = np.empty(N, np.int64)
    K = lib().lfo_lookups(_ptr(codes), _ptr(m8), C.c_int(H), C.c_int(W), _ptr(down), _ptr(ups), _ptr(nups))
    return down.astype(np.float64), np.ascontiguousarray(ups[:, :K]), nups


def orders(downstream_lookup, upstream_lookup8, num_ups):
    N = downstream_lookup.size
    down = np.ascontiguousarray(downstream_lookup, dtype=np.int64)
    ups = np.full((N, 8), -1, np.int64)
    ups[:, :upstream_lookup8.shape[1]] = upstream_lookup8
    po = np.empty(N, np.int64)
    ss = np.empty(2 * max(N, 1), np.int64)
    NL = lib().lfo_orders(_ptr(down), _ptr(ups), _ptr(np.ascontiguousarray(num_ups, dtype=np.int64)),
                          C.c_int64(N), _ptr(po), _ptr(ss))
    if NL < 0:
        raise ValueError("cyclic LDD")
    return po, ss[:2 * NL].reshape(NL, 2).copy()


class kinematicWave:
    """CPU oracle with the reference's interface (kinematic_wave_parallel.py:114-184)."""

    def __init__(self, compressed_encoded_ldd, land_mask, alpha_channel, beta, space_delta, time_delta,
                 alpha_floodplains=None, flagnancheck=False):
        self.space_delta = space_delta
        self.beta = beta
        self.a_dx_div_dt_channel = _f(alpha_channel * space_delta / time_delta)
        self.b_a_dx_div_dt_channel = _f(beta * self.a_dx_div_dt_channel)
        if alpha_floodplains is not None:
            self.a_dx_div_dt_floodplains = _f(alpha_floodplains * space_delta / time_delta)
            self.b_a_dx_div_dt_floodplains = _f(beta * self.a_dx_div_dt_floodplains)
        self.downstream_lookup, self.upstream_lookup, self.num_upstream_pixels = lookups(compressed_encoded_ldd,
                                                                                         land_mask)
        ups8 = np.full((self.num_upstream_pixels.size, 8), -1, np.int64)
        ups8[:, :self.upstream_lookup.shape[1]] = self.upstream_lookup
        self.pixels_ordered, self.order_start_stop = orders(self.downstream_lookup, ups8, self.num_upstream_pixels)
        self._scratch = np.empty(self.num_upstream_pixels.size)
        self.last_iters = (0, 0)

    def first_touch(self):
        """re-home the router's own vectors (see first_touch above); call after set_threads()"""
        for k in ("a_dx_div_dt_channel", "b_a_dx_div_dt_channel", "a_dx_div_dt_floodplains", "b_a_dx_div_dt_floodplains",
                  "upstream_lookup", "num_upstream_pixels", "pixels_ordered", "_scratch"):
            if hasattr(self, k):
                setattr(self, k, first_touch(getattr(self, k)))
        if np.ndim(self.space_delta) != 0:
            self.space_delta = first_touch(_f(self.space_delta))

    def kinematicWaveRouting(self, discharge, specific_lateral_inflow, section="main_channel"):
        if section == "main_channel":
            a, ba = self.a_dx_div_dt_channel, self.b_a_dx_div_dt_channel
        elif section == "floodplains":
            a, ba = self.a_dx_div_dt_floodplains, self.b_a_dx_div_dt_floodplains
        else:
            raise Exception("The section parameter must be either 'main_channel' or 'floodplain'!")
        assert discharge.dtype == np.float64 and discharge.flags.c_contiguous
        q = _f(specific_lateral_inflow)
        N = discharge.size
        if np.ndim(self.space_delta) == 0:
            dxp, dxs = None, float(self.space_delta)
        else:
            dxa = _f(self.space_delta)
            dxp, dxs = _ptr(dxa), 0.0
        it = np.zeros(2, np.int64)
        K = self.upstream_lookup.shape[1]
        lib().lfo_route(_ptr(discharge), _ptr(q), dxp, C.c_double(dxs), _ptr(a), _ptr(ba), C.c_double(self.beta),
                        _ptr(self.upstream_lookup), C.c_int(K), _ptr(self.num_upstream_pixels),
                        _ptr(self.pixels_ordered), _ptr(self.order_start_stop), C.c_int64(self.order_start_stop.shape[0]),
                        C.c_int64(N), _ptr(self._scratch), _ptr(it))
        self.last_iters = (int(it[0]), int(it[1]))


def interception(Interception, TaInterception, LeafDrainage, CumInterception, LAI, Rain, TaInterceptionMax, drainageK):
    V, N = Interception.shape
    lib().lfo_interception(_ptr(Interception), _ptr(TaInterception), _ptr(LeafDrainage), _ptr(CumInterception),
                           _ptr(_f(LAI)), _ptr(_f(Rain)), _ptr(_f(TaInterceptionMax)), C.c_double(drainageK),
                           C.c_int64(V), C.c_int64(N))


_SOIL_FIELDS = (
    # order of lfo_soil_args in lf_oracle.c
    "PoreSpaceNotZero1a PoreSpaceNotZero1b PoreSpaceNotZero2 "
    "KSat1a KSat1b KSat2 GenuInvM1a GenuInvM1b GenuInvM2 GenuM1a GenuM1b GenuM2 "
    "WRes1a WRes1b WRes1 WRes2 WWP1a WWP1b WWP1 WWP2 WFC1a WFC1b WFC1 WFC2 "
    "SoilDepth1a SoilDepth1b SoilDepth2 WS1a WS1b WS1 WS2 StoreMaxPervious "
    "Rain SnowMelt b_Xinanjiang PowerInfPot PowerPrefFlow UpperZoneK GwPercStep isFrozenSoil "
    "LeafDrainage Interception ESMax "
    "AvailableWaterForInfiltration DSLR ESAct PrefFlow Infiltration W1a W1b W1 W2 "
    "Theta1a Theta1b Theta2 Sat1a Sat1b Sat1 Sat2 SeepTopToSubA SeepTopToSubB SeepSubToGW "
    "UZOutflow UZ GwPercUZLZ "
    "index_landuse_all is_irrigated is_paddy_irrig paddy_inactive").split()
_SOIL_WRITTEN = set(
    "AvailableWaterForInfiltration DSLR ESAct PrefFlow Infiltration W1a W1b W1 W2 Theta1a Theta1b Theta2 Sat1a "
    "Sat1b Sat1 Sat2 SeepTopToSubA SeepTopToSubB SeepSubToGW UZOutflow UZ GwPercUZLZ".split())


class _SoilArgs(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in _SOIL_FIELDS] + [
        ("DtDay", C.c_double), ("AvWaterThreshold", C.c_double), ("CourantCrit", C.c_double),
        ("DrainedFraction", C.c_double), ("V", C.c_int64), ("L", C.c_int64), ("N", C.c_int64)]


def soil_columns(d):
    """d: dict keyed by the reference's argument names (soilloop.py:79-99); written arrays updated in place."""
    a = _SoilArgs()
    keep = []
    for k in _SOIL_FIELDS:
        v = d[k]
        if k in _SOIL_WRITTEN:
            assert v.dtype == np.float64 and v.flags.c_contiguous, k
            arr = v
        elif k == "index_landuse_all":
            arr = np.ascontiguousarray(v, dtype=np.int64)
        elif np.asarray(v).dtype == np.bool_ or np.asarray(v).dtype == np.uint8:
            arr = _u8(v)
        else:
            arr = _f(v)
        keep.append(arr)
        setattr(a, k, arr.ctypes.data)
    a.DtDay, a.AvWaterThreshold = float(d["DtDay"]), float(d["AvWaterThreshold"])
    a.CourantCrit, a.DrainedFraction = float(d["CourantCrit"]), float(d["DrainedFraction"])
    a.V, a.N = d["W1a"].shape
    a.L = d["WS1a"].shape[0]
    lib().lfo_soil_columns(C.byref(a))


def upstream_sum(downstruct, w):
    out = np.empty(w.size)
    lib().lfo_upstream_sum(_ptr(np.ascontiguousarray(downstruct, dtype=np.int32)), _ptr(_f(w)), C.c_int64(w.size),
                           _ptr(out))
    return out


class RoutingSubstep:
    """Element-wise arithmetic of routing.dynamic (routing.py:512-603, 693-703) around an oracle router."""

    def __init__(self, router, v):
        self.r, self.v = router, v   # v: namespace with the reference's var names

    def dynamic(self, split, sideflow_m3=None):
        """sideflow_m3: SideflowChanM3 of the sub-step (default: the runoff alone, no in-loop modules)"""
        v, L = self.v, lib()
        N = v.ChanQKin.size
        n = C.c_int64(N)
        side = np.empty(N)
        L.lfo_sideflow(_ptr(_f(v.ToChanM3RunoffDt if sideflow_m3 is None else sideflow_m3)), _ptr(_u8(v.IsChannelKinematic)), _ptr(_f(v.InvChanLength)),
                       C.c_double(v.InvDtRouting), C.c_int(0 if split else 1), n, _ptr(side))
        fix = lambda: L.lfo_main_fixup(_ptr(v.ChanQKin), _ptr(v.ChanM3Kin), _ptr(_f(v.ChanLength)),
                                       _ptr(_f(v.ChannelAlpha)), _ptr(_f(v.InvChanLength)),
                                       _ptr(_f(v.InvChannelAlpha)), C.c_double(v.Beta), C.c_double(v.InvBeta), n)
        if not split:
            self.r.kinematicWaveRouting(v.ChanQKin, side, "main_channel")
            fix()
            v.ChanQ = v.ChanQKin.copy()
            v.sumDisDay += v.ChanQ
        else:
            s1, s2 = np.empty(N), np.empty(N)
            L.lfo_split_sideflow(_ptr(side), _ptr(v.ChanM3Kin), _ptr(v.Chan2M3Kin), _ptr(_f(v.Chan2M3Start)),
                                 _ptr(_f(v.M3Limit)), _ptr(_f(v.Chan2QStart)), _ptr(_f(v.InvChanLength)), n,
                                 _ptr(s1), _ptr(s2))
            v.Sideflow1Chan = s1
            self.r.kinematicWaveRouting(v.ChanQKin, s1, "main_channel")
            fix()
            self.r.kinematicWaveRouting(v.Chan2QKin, s2, "floodplains")
            v.ChanQ = np.empty(N)
            L.lfo_floodplain_fixup(_ptr(v.Chan2QKin), _ptr(v.Chan2M3Kin), _ptr(v.CrossSection2Area), _ptr(v.ChanQ),
                                   _ptr(v.ChanQKin), _ptr(_f(v.ChanLength)), _ptr(_f(v.ChannelAlpha2)),
                                   _ptr(_f(v.InvChanLength)), _ptr(_f(v.InvChannelAlpha2)), _ptr(_f(v.Chan2M3Start)),
                                   _ptr(_f(v.QLimit)), C.c_double(v.Beta), C.c_double(v.InvBeta), n)
            v.sumDisDay += v.ChanQ
        v.FlowVelocity, v.TravelDistance = np.empty(N), np.empty(N)
        L.lfo_velocity(_ptr(v.ChanM3Kin), _ptr(v.ChanQKin), _ptr(_f(v.InvChanLength)), _ptr(_f(v.PixelArea)),
                       C.c_double(v.DtSec), n, _ptr(v.FlowVelocity), _ptr(v.TravelDistance))


_INLOOP_PTRS = (
    "ChanQ n_lakes lake_cell lake_ups_ptr lake_ups_idx LakeFactor LakeFactorSqr LakeAreaCC LakeStorageM3CC "
    "LakeInflowOldCC LakeOutflowCC LakeStorageM3BalanceCC LakeLevelCC LakeInflowCC QLakeOutM3Dt n_res res_cell "
    "res_ups_ptr res_ups_idx TotalReservoirStorageM3CC MinReservoirOutflowCC NormalReservoirOutflowCC "
    "NonDamagingReservoirOutflowCC ConservativeStorageLimitCC NormalStorageLimitCC FloodStorageLimitCC "
    "Normal_FloodStorageLimitCC DeltaO DeltaLN DeltaNFL ReservoirStorageM3CC ReservoirFillCC ReservoirInflowCC "
    "QResOutM3Dt QInM3Old QDelta QInDt QinADDEDM3 UpTrans TransLossM3Dt TransCum").split()


class _InloopArgs(C.Structure):  # lfo_inloop_args (lf_oracle.c)
    _fields_ = ([(k, C.c_int64 if k in ("n_lakes", "n_res") else C.c_void_p) for k in _INLOOP_PTRS] +
                [("TransPower1", C.c_double), ("TransPower2", C.c_double), ("TransSub", C.c_double)] +
                [(k, C.c_void_p) for k in ("ToChanM3RunoffDt", "EvaAddM3Dt", "WUseAddM3Dt", "ChannelToPolderM3Dt",
                                           "SideflowChanM3")] +
                [("DtRouting", C.c_double), ("InvNoRoutSteps", C.c_double), ("N", C.c_int64), ("step", C.c_int32)])


class InloopStructures:
    """lakes / reservoir / inflow / transmission .dynamic_inloop + the SideflowChanM3 assembly (routing.py:441-478)
    on host arrays of a `var` namespace with the reference's attribute names (all four options on)."""

    def __init__(self, v):
        self.v = v
        N = self.N = np.asarray(v.ChanQ).size
        ds = np.asarray(v.downstruct).astype(np.int64)
        order = np.argsort(ds, kind="stable")
        starts = np.searchsorted(ds[order], np.arange(N + 1))

        def csr(cells):
            ptr, idx = np.zeros(len(cells) + 1, np.int32), []
            for i, c in enumerate(cells):
                u = order[starts[c]:starts[c + 1]]
                idx.append(u)
                ptr[i + 1] = ptr[i] + u.size
            idx = np.concatenate(idx).astype(np.int32) if idx else np.zeros(0, np.int32)
            return ptr, (idx if idx.size else np.zeros(1, np.int32))
        self.lake_cell = np.asarray(v.LakeIndex).astype(np.int32)
        self.res_cell = np.asarray(v.ReservoirIndex).astype(np.int32)
        self.lake_csr, self.res_csr = csr(self.lake_cell), csr(self.res_cell)
        nl, nr = self.lake_cell.size, self.res_cell.size
        for k in ("LakeInflowCC",):
            setattr(v, k, np.zeros(nl))
        for k in ("ReservoirInflowCC", "ReservoirFillCC"):
            setattr(v, k, np.zeros(nr))
        for k in ("QLakeOutM3Dt", "QResOutM3Dt", "QInDt", "QinADDEDM3", "TransLossM3Dt", "SideflowChanM3"):
            setattr(v, k, np.zeros(N))

    def dynamic_inloop(self, step):
        v = self.v
        if step == 0:      # lakes.py:212-213, reservoir.py:195-196
            v.LakeStorageM3CC = _f(np.asarray(v.LakeStorageM3)[self.lake_cell]).copy()
            v.ReservoirStorageM3CC = _f(np.asarray(v.ReservoirStorageM3)[self.res_cell]).copy()
        a, keep = _InloopArgs(), []

        def put(name, arr):
            keep.append(arr)
            setattr(a, name, arr.ctypes.data)
        put("ChanQ", _f(v.ChanQ))
        a.n_lakes, a.n_res = self.lake_cell.size, self.res_cell.size
        put("lake_cell", self.lake_cell); put("lake_ups_ptr", self.lake_csr[0]); put("lake_ups_idx", self.lake_csr[1])
        put("res_cell", self.res_cell); put("res_ups_ptr", self.res_csr[0]); put("res_ups_idx", self.res_csr[1])
        for k in ("LakeFactor", "LakeFactorSqr", "LakeAreaCC", "TotalReservoirStorageM3CC", "MinReservoirOutflowCC",
                  "NormalReservoirOutflowCC", "NonDamagingReservoirOutflowCC", "ConservativeStorageLimitCC",
                  "NormalStorageLimitCC", "FloodStorageLimitCC", "Normal_FloodStorageLimitCC", "DeltaO", "DeltaLN",
                  "DeltaNFL", "QInM3Old", "QDelta", "ToChanM3RunoffDt"):
            put(k, _f(np.broadcast_to(getattr(v, k), (self.N,) if k in ("QInM3Old", "QDelta", "ToChanM3RunoffDt") else np.shape(getattr(v, k)))))
        put("UpTrans", _u8(v.UpTrans))
        for k in ("LakeStorageM3CC", "LakeInflowOldCC", "LakeOutflowCC", "LakeStorageM3BalanceCC", "LakeLevelCC",
                  "LakeInflowCC", "QLakeOutM3Dt", "ReservoirStorageM3CC", "ReservoirFillCC", "ReservoirInflowCC",
                  "QResOutM3Dt", "QInDt", "QinADDEDM3", "TransLossM3Dt", "TransCum", "SideflowChanM3"):
            x = getattr(v, k)
            assert x.dtype == np.float64 and x.flags.c_contiguous, k
            put(k, x)
        a.TransPower1, a.TransPower2, a.TransSub = float(v.TransPower1), float(v.TransPower2), float(v.TransSub)
        a.DtRouting, a.InvNoRoutSteps, a.N, a.step = float(v.DtRouting), float(v.InvNoRoutSteps), self.N, int(step)
        lib().lfo_inloop_structures(C.byref(a))
        if step == int(v.NoRoutSteps) - 1:      # lakes.py:283-292, reservoir.py:311-315: back to the dense state maps
            v.LakeStorageM3 = np.zeros(self.N)
            v.LakeStorageM3[self.lake_cell] = v.LakeStorageM3CC
            v.ReservoirStorageM3 = np.zeros(self.N)
            v.ReservoirStorageM3[self.res_cell] = v.ReservoirStorageM3CC


_PIX_V_IN = ("SoilFraction TaInterception Ta ESAct PrefFlow Infiltration SeepTopToSubA SeepTopToSubB SeepSubToGW Theta1a "
             "Theta1b Theta2 W1a W1b W2 UZOutflow GwPercUZLZ SoilDepthTotal").split()
_PIX_N_IN = "Rain SnowMelt EWRef SMaxSealed DirectRunoffFraction WaterFraction LowerZoneK LZThreshold GwLossStep".split()
_PIX_STATE = "CumInterSealed LZ LZInflowCUM TaInterceptionCUM TaCUM ESActCUM GwLossCUM".split()
_PIX_OUT = ("RainSnowmelt EWaterAct InterSealed TASealed DirectRunoff TaInterceptionAll TaPixel ESActPixel PrefFlowPixel "
            "InfiltrationPixel ThetaAll SeepTopToSubPixelA SeepTopToSubPixelB SeepSubToGWPixel Theta1aPixel Theta1bPixel "
            "Theta2Pixel LZOutflow UZOutflowPixel GwPercUZLZPixel GwLossLZ LZAvInflow LZOutflowToChannelPixel").split()


class _PixelArgs(C.Structure):  # lfo_pixel_args (lf_oracle.c)
    _fields_ = ([(k, C.c_void_p) for k in _PIX_V_IN + _PIX_N_IN + _PIX_STATE + _PIX_OUT + ["Theta"]] +
                [("InvDtDay", C.c_double), ("TimeSinceStart", C.c_double), ("N", C.c_int64)])


def pixel_aggregates(v):
    """opensealed.dynamic(); soil.dynamic_perpixel(); groundwater.dynamic() on a `var` namespace (in place)."""
    N = np.asarray(v.SoilFraction).shape[1]
    a, keep = _PixelArgs(), []

    def put(name, arr):
        keep.append(arr)
        setattr(a, name, arr.ctypes.data)
    for k in _PIX_V_IN:
        put(k, _f(np.asarray(getattr(v, k))))
    for k in _PIX_N_IN:
        put(k, _f(np.broadcast_to(np.asarray(getattr(v, k)), (N,))))
    for k in _PIX_STATE:
        x = np.array(np.broadcast_to(np.asarray(getattr(v, k), dtype=np.float64), (N,)))
        setattr(v, k, x)
        put(k, x)
    for k in _PIX_OUT:
        x = np.empty(N)
        setattr(v, k, x)
        put(k, x)
    v.Theta = np.empty((3, N))
    put("Theta", v.Theta)
    a.InvDtDay, a.TimeSinceStart, a.N = float(v.InvDtDay), float(v.TimeSinceStart), N
    lib().lfo_pixel_aggregates(C.byref(a))


def canopy(v, index_landuse):
    """soilloop.dynamic_canopy() on a `var` namespace (in place): [V,N] / [L,N] float64 C-contiguous arrays."""
    V, N = np.asarray(v.Interception).shape
    io = ("Interception", "TaInterception", "LeafDrainage", "CumInterception", "potential_transpiration", "RWS", "Ta",
          "W1a", "W1b", "W1")
    for k in io:
        x = getattr(v, k)
        assert isinstance(x, np.ndarray) and x.dtype == np.float64 and x.flags.c_contiguous, k
    ins = [_f(np.asarray(getattr(v, k))) for k in ("LAI", "LAITerm", "CropCoef", "CropGroupNumber", "WFC1", "WFC1a", "WFC1b",
                                                    "WWP1", "WWP1a", "WWP1b")]
    pix = [_f(np.broadcast_to(np.asarray(getattr(v, k)), (N,))) for k in ("Rain", "EWRef", "ETRef")]
    frozen = _u8(v.isFrozenSoil)
    idx = np.ascontiguousarray(index_landuse, dtype=np.int64)
    lib().lfo_canopy(*[_ptr(getattr(v, k)) for k in io], *[_ptr(a) for a in ins], *[_ptr(a) for a in pix], _ptr(frozen),
                     _ptr(idx), C.c_double(v.LeafDrainageK), C.c_double(v.InvDtDay), C.c_int64(V), C.c_int64(N))


class SurfaceRouting:
    """surface_routing.dynamic (surface_routing.py:115-212) on a `var` namespace: three oracle routers on LddToChan
    (initialSecond, :103-113) and the arithmetic around them."""

    def __init__(self, v, ldd_to_chan, land_mask):
        self.v = v
        a = np.asarray(v.OFAlpha, dtype=np.float64)
        mk = lambda row: kinematicWave(ldd_to_chan, land_mask, a[row], v.Beta, v.PixelLength, v.DtSec)
        self.other, self.forest, self.direct = mk(0), mk(1), mk(2)

    def dynamic(self):
        v, L = self.v, lib()
        N = np.asarray(v.DirectRunoff).size
        n = C.c_int64(N)
        v.SurfaceRunSoil, v.SurfaceRunoff, v.TotalRunoff = np.empty((3, N)), np.empty(N), np.empty(N)
        side = np.empty((3, N))
        uz, lz = _f(v.UZOutflowPixel), _f(v.LZOutflowToChannelPixel)
        L.lfo_surface_pre(_ptr(_f(v.SoilFraction)), _ptr(_f(v.AvailableWaterForInfiltration)), _ptr(_f(v.Infiltration)),
                          _ptr(_f(v.DirectRunoff)), _ptr(uz), _ptr(lz), C.c_double(v.MMtoM3),
                          C.c_double(v.InvPixelLength), C.c_double(v.InvDtSec), n, _ptr(v.SurfaceRunSoil),
                          _ptr(v.SurfaceRunoff), _ptr(v.TotalRunoff), _ptr(side))
        self.direct.kinematicWaveRouting(v.OFQDirect, side[0])          # :151-153
        self.other.kinematicWaveRouting(v.OFQOther, side[1])
        self.forest.kinematicWaveRouting(v.OFQForest, side[2])
        for k in ("OFM3Direct", "OFM3Other", "OFM3Forest", "OFToChanM3", "WaterDepth", "ToChanM3Runoff", "ToChanM3RunoffDt"):
            setattr(v, k, np.empty(N))
        L.lfo_surface_post(_ptr(v.OFQDirect), _ptr(v.OFQOther), _ptr(v.OFQForest), _ptr(_f(v.OFAlpha)),
                           _ptr(_u8(v.IsChannel)), _ptr(uz), _ptr(lz), C.c_double(v.Beta), C.c_double(v.PixelLength),
                           C.c_double(v.DtSec), C.c_double(v.MMtoM3), C.c_double(v.M3toMM), C.c_double(v.InvNoRoutSteps),
                           n, _ptr(v.OFM3Direct), _ptr(v.OFM3Other), _ptr(v.OFM3Forest), _ptr(v.OFToChanM3),
                           _ptr(v.WaterDepth), _ptr(v.ToChanM3Runoff), _ptr(v.ToChanM3RunoffDt))


def sweep_positions(state, constant, ups_ptr, ups_idx, a, ba, beta, begin, end):
    """solve1Pixel over positions [begin, end) of an indexed state vector (row-block plan tests)."""
    assert state.dtype == np.float64 and state.flags.c_contiguous
    lib().lfo_sweep_positions(_ptr(state), _ptr(_f(constant)), _ptr(np.ascontiguousarray(ups_ptr, dtype=np.int32)),
                              _ptr(np.ascontiguousarray(ups_idx, dtype=np.int32)), _ptr(_f(a)), _ptr(_f(ba)),
                              C.c_double(beta), C.c_int64(begin), C.c_int64(end))


PF_L = ("WRes1a WRes1b WRes2 WS1a WS1b WS2 PoreSpaceNotZero1a PoreSpaceNotZero1b PoreSpaceNotZero2 GenuInvAlpha1a "
        "GenuInvAlpha1b GenuInvAlpha2 GenuInvM1a GenuInvM1b GenuInvM2 GenuInvN1a GenuInvN1b GenuInvN2").split()


def soil_pf(d, index_landuse, HeadMax):
    """suctionUnsaturatedSoilPF (soilloop.py:673-695): d holds W1a, W1b, W2 [V,N] and the [L,N] arrays of PF_L -> pF0, pF1, pF2"""
    V, N = np.asarray(d["W1a"]).shape
    out = [np.empty((V, N)) for _ in range(3)]
    args = [_f(d[k]) for k in ("W1a", "W1b", "W2")] + [(_u8(d[k]) if k.startswith("Pore") else _f(d[k])) for k in PF_L]
    idx = np.ascontiguousarray(index_landuse, dtype=np.int64)
    lib().lfo_soil_pf(*[_ptr(a) for a in out], *[_ptr(a) for a in args], _ptr(idx), C.c_double(HeadMax), C.c_int64(V),
                      C.c_int64(N))
    return out
