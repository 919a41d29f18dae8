"""Row N1: the whole model step, chained, against the reference itself.

tests/golden/etrs89_chain.npz holds 12 model steps on the LF_ETRS89 domain (real LDD, channel geometry, lake and
reservoir sites, real meteorological fields) produced by the reference's OWN module methods called in the order of
Lisflood_dynamic.py:114-229 (tests/golden/make_golden.py gen_chain): per step ChanQAvg -- the `dis` output of the
reference (Lisflood_dynamic.py:208) -- ChanQ and the runoff into the channels, plus snapshots of every state vector.

  * not gpu: the C oracle's chain must reproduce it (pins the checker itself at whole-step level);
  * gpu: HotPathDevice (resident chain) and the drop-in module classes must reproduce `dis` to <= 1e-6 relative
    (north_star's tolerance; the test asserts 1e-9) on every step, and the series survives the .tss / netCDF writers.
"""
import os
import sys
import types

import numpy as np
import pytest

from conftest import golden

RTOL_DIS = 1e-9          # north_star asks for 1e-6 relative on dis; the engine holds 1e-9
SNAP_RTOL = 1e-8


def fixture():
    g = golden("etrs89_chain")
    values = {k[4:]: g[k] for k in g.files if k.startswith("val_")}
    sc = {k[3:]: float(g[k]) for k in g.files if k.startswith("sc_")}
    st = {k[3:]: (g[k] if g[k].ndim else float(g[k])) for k in g.files if k.startswith("st_")}
    forcing = [{k[5:]: np.ascontiguousarray(g[k][s]) for k in g.files if k.startswith("forc_")}
               for s in range(g["QInM3"].shape[0])]
    return g, values, sc, st, forcing


def cp(d):
    return {k: (np.array(a, copy=True) if isinstance(a, np.ndarray) else a) for k, a in d.items()}


def close(got, want, rtol, msg):
    want = np.asarray(want, dtype=np.float64)
    scale = max(1.0, float(np.abs(want).max())) if want.size else 1.0
    np.testing.assert_allclose(got, want, rtol=rtol, atol=rtol * 1e-3 * scale, err_msg=str(msg))


def close_mb(got, want, g, msg):
    """mass-balance terms are differences of catchment storages (~2e9 m3): held to 1e-12 of that scale"""
    scale = float(np.abs(g["st_StorageStepINIT"]).max())
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-12 * scale, err_msg=str(msg))


def namespace(values, sc, st):
    v = types.SimpleNamespace()
    for k, a in list(cp(values).items()) + list(sc.items()) + list(cp(st).items()):
        setattr(v, k, np.ascontiguousarray(a, dtype=np.float64) if isinstance(a, np.ndarray) and a.dtype.kind == "f" else a)
    v.InvBeta, v.InvPixelLength, v.InvDtSec = 1 / v.Beta, 1 / v.PixelLength, 1 / v.DtSec
    v.NoRoutSteps = int(v.NoRoutSteps)
    v.InvDtRouting, v.InvNoRoutSteps = 1 / v.DtRouting, 1 / v.NoRoutSteps
    return v


def post_loop(v):
    """Lisflood_dynamic.py:185-208"""
    v.QInM3Old = v.QInM3
    v.ChanM3 = v.ChanM3Kin + v.Chan2M3Kin - v.Chan2M3Start
    v.TotalCrossSectionArea = v.ChanM3 * v.InvChanLength
    v.sumDis = getattr(v, "sumDis", 0.0) + v.sumDisDay
    v.ChanQAvg = v.sumDisDay / v.NoRoutSteps


def check_snapshots(g, i, get, names, rtol=SNAP_RTOL):
    for k in names:
        close(get(k), g["snap_" + k][i], rtol, ("snapshot", i, k))


def test_oracle_chain_reproduces_the_reference_chain():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import oracle
    g, values, sc, st, forcing = fixture()
    mask = g["mask"]
    N = int(mask.sum())
    v = namespace(values, sc, st)
    idx = np.arange(3)
    surf = oracle.SurfaceRouting(v, g["ldd_to_chan"], mask)
    kw = oracle.kinematicWave(g["ldd_cut"], mask, v.ChannelAlpha, v.Beta, v.ChanLength, v.DtRouting,
                              alpha_floodplains=v.ChannelAlpha2)
    stru, sub = oracle.InloopStructures(v), oracle.RoutingSubstep(kw, v)
    sampled = list(g["sampled"])
    for step, f in enumerate(forcing):
        for k, a in f.items():
            setattr(v, k, a)
        oracle.canopy(v, idx)
        d = dict(vars(v))
        d["ESMax"] = np.ascontiguousarray(v.ESRef * v.LAITerm)
        d.update(index_landuse_all=idx, is_irrigated=np.array([False, False, True]), is_paddy_irrig=np.zeros(3, bool),
                 paddy_inactive=np.zeros((1, N), bool))
        oracle.soil_columns(d)
        v.TimeSinceStart = float(step + 1)
        oracle.pixel_aggregates(v)
        surf.dynamic()
        v.QInM3 = g["QInM3"][step]
        v.QDelta = (v.QInM3 - v.QInM3Old) * v.InvNoRoutSteps                    # inflow.py:108
        v.sumDisDay = np.zeros(N)
        for s in range(v.NoRoutSteps):
            stru.dynamic_inloop(s)
            sub.dynamic(split=True, sideflow_m3=v.SideflowChanM3)
        post_loop(v)
        close(v.ToChanM3RunoffDt, g["out_ToChanM3RunoffDt"][step], 1e-10, (step, "ToChanM3RunoffDt"))
        close(v.ChanQ, g["out_ChanQ"][step], 1e-10, (step, "ChanQ"))
        close(v.ChanQAvg, g["out_ChanQAvg"][step], 1e-10, (step, "dis"))
        if step in sampled:
            check_snapshots(g, sampled.index(step), lambda k: getattr(v, k),
                            ("W1a", "W1b", "W2", "UZ", "LZ", "CumInterception", "DSLR", "OFQOther", "ChanQKin", "Chan2QKin",
                             "ChanM3", "sumDis", "LakeStorageM3CC", "LakeOutflowCC", "ReservoirStorageM3CC", "TransCum",
                             "QinADDEDM3", "TotalCrossSectionArea"), rtol=1e-10)


def test_dis_series_round_trips_through_the_writers(tmp_path):
    """ChanQAvg of the chain as the reference would report it: dis.tss at a few gauges and the dis map stack."""
    from lisflood_amd import output as out
    g = golden("etrs89_chain")
    mask = g["mask"]
    dis = g["out_ChanQAvg"]
    gauges = np.array([10, 999, 2500, 2846])
    path = str(tmp_path / "dis.tss")
    out.write_tss(path, list(gauges + 1), 1, dis[:, gauges])
    first, ids, step0, data = out.read_tss(path)
    assert ids == list(gauges + 1) and step0 == 1 and data.shape == (dis.shape[0], gauges.size)
    np.testing.assert_allclose(data, dis[:, gauges], rtol=5e-6)      # " %14g": six significant digits, as the reference
    nc = str(tmp_path / "dis.nc")
    stack = np.stack([out.decompress(d, mask, fill=np.nan) for d in dis])
    H, W = mask.shape
    out.write_netcdf_classic(nc, "dis", stack, x=np.arange(W) * 5000.0, y=np.arange(H)[::-1] * 5000.0,
                             time_values=np.arange(dis.shape[0], dtype=float), units="m3/s")
    back, x, y, t = out.read_netcdf_classic(nc, "dis")
    assert np.array_equal(back[:, mask], dis) and np.isnan(back[:, ~mask]).all() and t.size == dis.shape[0]
    # ... and as the reference writes dis.nc: netCDF-4, zlib, chunks (1, H, W), _FillValue -9999 (netcdf.py:432-583)
    nc4 = str(tmp_path / "dis4.nc")
    out.write_netcdf4(nc4, "dis", stack, x=np.arange(W) * 5000.0, y=np.arange(H)[::-1] * 5000.0,
                      time_values=np.arange(dis.shape[0], dtype=float), time_units="days since 2016-01-02 06:00:00.0",
                      standard_name="DischargeMaps", long_name="ChanQAvg", units="m3/s")
    back, x, y, t = out.read_netcdf4(nc4, "dis")
    assert np.array_equal(back[:, mask], dis) and np.isnan(back[:, ~mask]).all() and t.size == dis.shape[0]
    import os
    assert os.path.getsize(nc4) < 0.7 * os.path.getsize(nc)            # deflate + shuffle at work


# ---------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def amd():
    import lisflood_amd
    from lisflood_amd import _lib
    n = _lib.device_count()
    if n < 1:
        pytest.fail("no HIP device visible: the -m gpu tests need an MI355X")
    return lisflood_amd


@pytest.mark.gpu
def test_hot_path_reproduces_the_reference_dis(amd):
    """HotPathDevice (everything resident, structures inside the channel wavefront) on the reference's chain."""
    from lisflood_amd.hotpath import HotPathDevice
    g, values, sc, st, forcing = fixture()
    hp = HotPathDevice(cp(values), sc, g["mask"], g["ldd_to_chan"], g["ldd_cut"], split=True, structures=cp(st))
    sampled = list(g["sampled"])
    worst = 0.0
    sum_dis = np.zeros(int(g["mask"].sum()))
    for step, f in enumerate(forcing):
        hp.step(f, time_since_start=step + 1, QInM3=g["QInM3"][step])
        dis = hp.chan_q_avg()
        want = g["out_ChanQAvg"][step]
        close(dis, want, RTOL_DIS, (step, "dis"))
        worst = max(worst, float(np.max(np.abs(dis - want) / np.maximum(np.abs(want), 1e-3))))
        close(hp.download("ChanQ"), g["out_ChanQ"][step], RTOL_DIS, (step, "ChanQ"))
        close(hp.download("ToChanM3RunoffDt"), g["out_ToChanM3RunoffDt"][step], RTOL_DIS, (step, "ToChanM3RunoffDt"))
        sum_dis += hp.download("sumDisDay")
        mb, qerr = hp.mass_balance()                       # option repMBTs (routing.py:483-499, 645-691)
        close_mb(mb, g["out_MBErrorSplitRoutingM3"][step], g, (step, "MBErrorSplitRoutingM3"))
        close_mb(qerr * 3600.0, g["out_OutletDischargeErrorSplitRouting"][step] * 3600.0, g, (step, "OutletDischargeError"))
        close_mb(hp.rmod.var.StorageStepINIT, g["out_StorageStepINIT"][step], g, (step, "StorageStepINIT"))
        if step in sampled:
            i = sampled.index(step)
            check_snapshots(g, i, hp.download, ("W1a", "W1b", "W2", "UZ", "LZ", "CumInterception", "DSLR", "Infiltration",
                                               "DirectRunoff", "OFQDirect", "OFQOther", "OFQForest", "ChanQKin", "Chan2QKin",
                                               "ChanM3Kin", "Chan2M3Kin", "CrossSection2Area", "Sideflow1Chan",
                                               "UZOutflowPixel", "LZOutflowToChannelPixel"))
            check_snapshots(g, i, hp.download_site, ("LakeStorageM3CC", "LakeOutflowCC", "LakeLevelCC", "ReservoirStorageM3CC",
                                                    "ReservoirFillCC", "TransCum", "QinADDEDM3"))
            close(sum_dis, g["snap_sumDis"][i], SNAP_RTOL, (step, "sumDis"))
    print("max relative deviation of dis over %d steps: %.3e" % (len(forcing), worst))
    assert worst < 1e-6
    hp.free()


@pytest.mark.gpu
@pytest.mark.parametrize("engine_order", [False, True])
def test_module_classes_reproduce_the_reference_dis(amd, engine_order):
    """The drop-in module classes (soilloop, pixel aggregates, surface_routing, routing with its structures), stepping
    the sub-step loop one routing.dynamic(s) at a time on host `var` arrays, exactly as the reference's driver would."""
    from test_gpu_parity import _model_var
    from lisflood_amd import pixel_aggregates as PA
    from lisflood_amd.soilloop import soilloop
    from lisflood_amd.surface_routing import surface_routing
    g, values, sc, st, forcing = fixture()
    mask = g["mask"]
    N = int(mask.sum())
    v = _model_var(N)
    for k, a in list(cp(values).items()) + list(sc.items()) + list(cp(st).items()):
        setattr(v, k, a)
    v.InvBeta, v.InvPixelLength, v.InvDtSec = 1 / v.Beta, 1 / v.PixelLength, 1 / v.DtSec
    v.NoRoutSteps = int(v.NoRoutSteps)
    v.InvDtRouting, v.InvNoRoutSteps = 1 / v.DtRouting, 1 / v.NoRoutSteps
    m_soil = soilloop(v); m_soil.initial()
    m_surf = surface_routing(v); m_surf.initialSecond(g["ldd_to_chan"], mask)
    m_rout = amd.routing.routing(v, options=dict(SplitRouting=True, InitLisflood=False, simulateLakes=True,
                                                 simulateReservoirs=True, inflow=True, TransLoss=True, repMBTs=True),
                                 engine_order=engine_order)
    m_rout.attach_router(g["ldd_cut"], mask)
    m_rout.attach_structures()
    nsteps = 6            # the first half of the chain incl. the storm (each step is 24 host round trips here)
    for step, f in enumerate(forcing[:nsteps]):
        for k, a in f.items():
            setattr(v, k, a.copy())
        v.TimeSinceStart = float(step + 1)
        m_soil.dynamic_canopy(); m_soil.dynamic_soil()
        PA.dynamic(v)
        m_surf.dynamic()
        v.QInM3 = g["QInM3"][step].copy()
        v.QDelta = (v.QInM3 - v.QInM3Old) * v.InvNoRoutSteps                    # inflow.dynamic_init, inflow.py:108
        v.sumDisDay = np.zeros(N)
        for s in range(v.NoRoutSteps):
            m_rout.dynamic(s)
        v.QInM3Old = v.QInM3                                                    # Lisflood_dynamic.py:185
        m_rout.step_end()
        close(v.ChanQAvg, g["out_ChanQAvg"][step], RTOL_DIS, (step, "dis"))
        close(v.ChanQ, g["out_ChanQ"][step], RTOL_DIS, (step, "ChanQ"))
        close_mb(v.AddedTRUN, g["out_AddedTRUN"][step], g, (step, "AddedTRUN"))
        close_mb(v.MBErrorSplitRoutingM3, g["out_MBErrorSplitRoutingM3"][step], g, (step, "MBErrorSplitRoutingM3"))
        close_mb(v.StorageStepINIT, g["out_StorageStepINIT"][step], g, (step, "StorageStepINIT"))


@pytest.mark.gpu
def test_subcatchment_run_equals_the_whole_domain(amd):
    """The reference's own test strategy for this path (tests/test_subcatchments.py): a run on subcatchment_mask.map
    must give, on that mask, exactly the arrays of the run on the whole domain (NetCDFComparator(array_equal=True)).
    Here: the resident chain on the 1 023-pixel sub-catchment against the 2 847-pixel domain, six model steps, `dis`
    and the state bit for bit."""
    from lisflood_amd.hotpath import HotPathDevice, _structures_on_subdomain
    g, values, sc, st, forcing = fixture()
    mask = g["mask"]
    N = int(mask.sum())
    sub2d = golden("etrs89_static")["subcatchment_mask"] & mask
    sel = sub2d[mask]
    ids = np.nonzero(sel)[0]
    assert ids.size == 1023
    for k in ("Catchments", "AtLastPointC", "IsUpsOfStructureKinematicC", "StorageStepINIT", "DischargeM3StructuresIni", "Ldd"):
        st.pop(k)                                           # mass-balance bookkeeping is not part of this comparison
    keep_lake = sel[st["LakeIndex"]]
    keep_res = sel[st["ReservoirIndex"]]
    st_sub = dict(st)
    for k, a in st.items():                                # site vectors of the structures inside the sub-catchment
        if isinstance(a, np.ndarray) and a.shape == st["LakeIndex"].shape and k.startswith("Lake") and k != "LakeStorageM3":
            st_sub[k] = a[keep_lake]
        elif isinstance(a, np.ndarray) and a.shape == st["ReservoirIndex"].shape and k != "ReservoirStorageM3" and (
                k.startswith("Reservoir") or k.endswith("CC") or k.startswith("Delta")):
            st_sub[k] = a[keep_res]
    st_sub = _structures_on_subdomain(st_sub, ids, N)
    assert st_sub["LakeIndex"].size + st_sub["ReservoirIndex"].size > 0
    vals_sub = {k: np.ascontiguousarray(np.asarray(a)[..., ids]) for k, a in values.items()}
    whole = HotPathDevice(cp(values), sc, mask, g["ldd_to_chan"], g["ldd_cut"], split=True, structures=cp(st))
    part = HotPathDevice(vals_sub, sc, sub2d, g["ldd_to_chan"][ids], g["ldd_cut"][ids], split=True, structures=st_sub)
    for step, f in enumerate(forcing[:6]):
        whole.step(f, step + 1, QInM3=g["QInM3"][step])
        part.step({k: np.ascontiguousarray(a[ids]) for k, a in f.items()}, step + 1, QInM3=g["QInM3"][step][ids])
        assert np.array_equal(part.chan_q_avg(), whole.chan_q_avg()[ids]), step
        for k in ("ChanQ", "ChanQKin", "Chan2QKin", "W1a", "UZ", "LZ", "OFQOther"):
            assert np.array_equal(part.download(k), whole.download(k)[..., ids]), (step, k)
    whole.free(); part.free()


@pytest.mark.gpu
def test_warm_start_from_reference_named_state_maps(amd, tmp_path):
    """f4: the state of the resident chain written as the reference's state maps (ChanQState, ChanCrossSectionState,
    Theta1ForestState, ... default_options.py 'repStateMaps'), read back, and a FRESH engine warm-started from them
    (-9999 = cold value, channel state rebuilt as routing.initial / initialSecond do): the run continues to rounding
    -- like the reference's own warm start, which goes through the same maps -- and dis stays within 1e-9."""
    from lisflood_amd import output as out
    from lisflood_amd.hotpath import HotPathDevice
    g, values, sc, st, forcing = fixture()
    mask = g["mask"]
    mk = lambda: HotPathDevice(cp(values), sc, mask, g["ldd_to_chan"], g["ldd_cut"], split=True, structures=cp(st))
    a = mk()
    for step in range(4):
        a.step(forcing[step], step + 1, QInM3=g["QInM3"][step])
    maps = a.state_maps()
    assert set(maps) == set(HotPathDevice.STATE_MAPS)
    np.testing.assert_allclose(maps["ChanQState"], g["out_ChanQ"][3], rtol=1e-9)
    out.write_state_maps(str(tmp_path / "state"), maps, mask, time_value=4)
    back = out.read_state_maps(str(tmp_path / "state"), mask)
    assert set(back) == set(maps) and all(np.array_equal(back[k], maps[k]) for k in maps)
    b = mk()
    b.load_state_maps(back)
    b.set_inflow(g["QInM3"][3])                    # the previous step's hydrograph (QInM3Old) is not a state map
    for step in range(4, 8):
        a.step(forcing[step], step + 1, QInM3=g["QInM3"][step])
        b.step(forcing[step], step + 1, QInM3=g["QInM3"][step])
        close(b.chan_q_avg(), a.chan_q_avg(), 1e-9, (step, "dis after the warm start"))
        close(b.chan_q_avg(), g["out_ChanQAvg"][step], 1e-9, (step, "dis vs the reference chain"))
        close(b.download("W1a"), a.download("W1a"), 1e-12, (step, "W1a"))
    # -9999 everywhere = the cold-start values the engine was built with
    c = mk()
    before = {k: c.download(k) for k in ("ChanQ", "ChanQKin", "Chan2QKin", "W1a", "W2", "UZ", "LZ", "OFQOther")}
    c.load_state_maps({k: np.full(int(mask.sum()), -9999.0) for k in maps})
    for k, want in before.items():
        if k == "OFQOther":
            continue                                # -9999 means "no water on the surface" (surface_routing.py:57-63)
        close(c.download(k), want, 1e-12, ("cold", k))
    assert (c.download("OFQOther") == 0).all()
    a.free(); b.free(); c.free()


# ---------------------------------------------------------------------------------------------------------------------
# Row N1, long form (tests/golden/etrs89_long.npz, make_golden.py `long`): 72 model steps, the land surface initialised
# by the reference's OWN soil / groundwater / landusechange / surface_routing initial() from what cold.xml binds (no
# seeded soil parameter left), LAI from the use case's ten-day stack, two storms: 10 of the 31 reservoirs move between
# storage regimes inside the window.  `dis` of every step, the lake / reservoir series of every step, state snapshots
# every tenth step.
# ---------------------------------------------------------------------------------------------------------------------
def long_fixture():
    g = golden("etrs89_long")
    values = {k[4:]: g[k] for k in g.files if k.startswith("val_")}
    sc = {k[3:]: float(g[k]) for k in g.files if k.startswith("sc_")}
    st = {k[3:]: (g[k] if g[k].ndim else float(g[k])) for k in g.files if k.startswith("st_")}
    nsteps, N = g["out_dis"].shape
    forcing = [{k[5:]: g[k][s] for k in g.files if k.startswith("forc_")} for s in range(nsteps)]    # float32, as stored
    qin = np.zeros((nsteps, N))
    qin[:, g["QInM3_points"]] = g["QInM3_values"]
    lai = [g["LAI"][i] for i in g["lai_interval_of_step"]]
    laiterm = [np.exp(-float(g["kgb"]) * x) for x in lai]                        # leafarea.py:91
    values["LAI"], values["LAITerm"] = lai[0].copy(), laiterm[0].copy()
    return g, values, sc, st, forcing, qin, lai, laiterm


def test_oracle_chain_reproduces_the_long_reference_chain():
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    from oracle_chain import OracleChain
    g, values, sc, st, forcing, qin, lai, laiterm = long_fixture()
    ch = OracleChain(values, sc, g["mask"], g["ldd_to_chan"], g["ldd_cut"], structures=st, split=True)
    snap = list(g["snap_steps"])
    worst = 0.0
    for step, f in enumerate(forcing):
        ch.v.LAI, ch.v.LAITerm = lai[step].copy(), laiterm[step].copy()
        dis = ch.step({k: a.astype(np.float64) for k, a in f.items()}, qin[step])
        want = g["out_dis"][step]
        worst = max(worst, float(np.max(np.abs(dis - want) / np.maximum(np.abs(want), 1e-3))))
        close(dis, want, 1e-9, (step, "dis"))
        for k in ("LakeStorageM3CC", "ReservoirStorageM3CC", "ReservoirFillCC"):
            close(getattr(ch.v, k), g["site_" + k][step], 1e-9, (step, k))
        if step in snap:
            for k in ("LZ", "ChanQKin", "Chan2QKin", "ChanM3", "sumDis", "OFQOther", "TransCum", "CumInterSealed"):
                close(getattr(ch.v, k), g["snap_" + k][snap.index(step)], 1e-8, (step, k))
    print("oracle chain, %d steps: max relative deviation of dis %.3e" % (len(forcing), worst))
    assert worst < 1e-8
    fill = g["site_ReservoirFillCC"]
    assert ((fill.max(0) - fill.min(0)) > 0.02).sum() >= 5          # the reservoirs do move in the window


@pytest.mark.gpu
def test_hot_path_reproduces_the_long_reference_dis(amd):
    """72 model steps of HotPathDevice on the reference-initialised LF_ETRS89 land surface: `dis` within 1e-6 relative
    (north_star's bar; asserted at 1e-8) on every step, the lake / reservoir series on every step, the state snapshots."""
    from lisflood_amd.hotpath import HotPathDevice
    g, values, sc, st, forcing, qin, lai, laiterm = long_fixture()
    hp = HotPathDevice(cp(values), sc, g["mask"], g["ldd_to_chan"], g["ldd_cut"], split=True, structures=cp(st))
    snap, snapv = list(g["snap_steps"]), list(g["snapv_steps"])
    idx = g["lai_interval_of_step"]
    worst = 0.0
    for step, f in enumerate(forcing):
        if step == 0 or idx[step] != idx[step - 1]:
            hp.set_lai(lai[step], laiterm[step])                                 # leafarea.dynamic, once per interval
        hp.step(f, time_since_start=step + 1, QInM3=qin[step])                  # float32 forcing: widened on the device
        dis = hp.chan_q_avg()
        want = g["out_dis"][step]
        worst = max(worst, float(np.max(np.abs(dis - want) / np.maximum(np.abs(want), 1e-3))))
        close(dis, want, 1e-8, (step, "dis"))
        close(dis[g["gauges"]], want[g["gauges"]], 1e-8, (step, "dis at the gauges"))
        for k in ("LakeStorageM3CC", "LakeOutflowCC", "LakeLevelCC", "ReservoirStorageM3CC", "ReservoirFillCC"):
            close(hp.download_site(k), g["site_" + k][step], 1e-8, (step, k))
        if step in snap:
            i = snap.index(step)
            for k in ("LZ", "ChanQKin", "Chan2QKin", "OFQOther", "CumInterSealed", "UZOutflowPixel"):
                close(hp.download(k), g["snap_" + k][i], SNAP_RTOL, (step, k))
            close(hp.download_site("TransCum"), g["snap_TransCum"][i], SNAP_RTOL, (step, "TransCum"))
        if step in snapv:
            i = snapv.index(step)
            for k in ("W1a", "W1b", "W2", "UZ", "DSLR", "CumInterception"):
                close(hp.download(k), g["snapv_" + k][i], SNAP_RTOL, (step, k))
    print("max relative deviation of dis over %d steps: %.3e" % (len(forcing), worst))
    assert worst < 1e-6
    hp.free()
