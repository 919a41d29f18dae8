"""LDD operations on the device (csrc/lf_ldd.hip through the C ABI) against brute-force walks, against the host
helpers, and against the reference's own PCRaster-made maps of LF_ETRS89: ec_upArea.nc (accuflux) and the catchment
masks mask.map / subcatchment_mask.map (catchment)."""
import numpy as np
import pytest

from conftest import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def amd():
    import lisflood_amd
    from lisflood_amd import _lib, routing  # noqa: F401  (the tests reach the module class through the package)
    if _lib.device_count() < 1:
        pytest.fail("no HIP device visible: the -m gpu tests need an MI355X")
    return lisflood_amd


def walk_down(down, p):
    while down[p] >= 0:
        p = down[p]
    return p


def case(H=61, W=47):
    from lisflood_amd import synthetic as syn
    mask = np.ones((H, W), bool); mask[:4, :6] = False; mask[22, 10:14] = False
    codes = syn.make_ldd("deep", H, W, 21, land_mask=mask)[mask].astype(np.float64)
    return codes, mask


def test_downstream_catchment_totals_vs_walks(amd):
    from lisflood_amd import ldd as L
    codes, mask = case()
    N = codes.size
    down = L.downstream_index(codes, mask)
    d = L.LddDevice(codes, mask)
    x = np.random.default_rng(3).uniform(0, 9, N)
    assert np.array_equal(d.downstream(x), np.where(down >= 0, x[np.maximum(down, 0)], x))
    assert np.array_equal(d.downstream(x), L.downstream(codes, mask, x))
    outlets = L.uniqueid(down < 0)
    lab = d.catchment(outlets)
    root = np.array([walk_down(down, p) for p in range(N)])
    assert np.array_equal(lab, outlets[root]) and (lab > 0).all()
    pts = np.zeros(N, np.int64); pts[[50, 300, 400, 1700]] = [7, 8, 9, 7]          # interior points stop the walk
    lab2 = d.subcatchment(pts)
    assert np.array_equal(lab2, L.subcatchment(codes, mask, pts))
    assert np.array_equal(d.catchment(pts), L.catchment(codes, mask, pts))      # PCRaster catchment: the enclosing point wins
    for p in range(0, N, 11):
        q, hit = p, 0
        while True:
            if pts[q]:
                hit = pts[q]; break
            if down[q] < 0:
                break
            q = down[q]
        assert lab2[p] == hit
    w = np.random.default_rng(4).uniform(0, 5, N)
    tot = d.catchment_totals(w)
    want = np.bincount(root, weights=w, minlength=N)[root]                          # routing.py:483-499
    np.testing.assert_allclose(tot, want, rtol=1e-13)
    # several weight vectors in one sweep (the mass-balance terms of routing.py:645-691): each equals its own call
    ws = [np.random.default_rng(40 + i).uniform(0, 5, N) for i in range(6)]
    for a, w1 in zip(d.catchment_totals_multi(ws), ws):
        assert np.array_equal(a, d.catchment_totals(w1))
    d.close()


@pytest.mark.parametrize("family,seed", [("deep", 2), ("river", 7), ("shallow", 1)])
def test_accuflux_on_level_blocks_equals_the_level_schedule(amd, family, seed, monkeypatch):
    """accuflux runs on the router's block plan (blocks of up to 64 levels cone by cone through LDS, k_accu_cones): the
    same additions in the same order as one launch per level (LF_ROUTE_CONES=0) -- bit-identical, and equal to the
    brute-force sum over every cell's upstream tree"""
    from lisflood_amd import synthetic as syn
    from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave
    H, W = 700, 600
    N = H * W
    codes = syn.make_ldd(family, H, W, seed)
    kw = kinematicWave(None, None, np.ones(N), 0.6, 1.0, 1.0, graph=Graph(ldd_raster=codes))
    x = np.random.default_rng(8).uniform(0, 3, N)
    got = kw.accuflux(x)
    assert kw.last_launches()["launches"] <= kw.graph.num_levels // 32 + 8      # blocks of up to 64 levels, not levels
    monkeypatch.setenv("LF_ROUTE_CONES", "0")
    ref = kw.accuflux(x)
    assert np.array_equal(got, ref)
    down = kw.graph.lookups()[0].astype(np.int64)
    po, ss = kw.graph.orders()
    acc = x.copy()                                             # level by level from the headwaters (float order differs)
    for k in range(ss.shape[0]):
        cells = po[ss[k, 0]:ss[k, 1]]
        d = down[cells]
        np.add.at(acc, d[d >= 0], acc[cells[d >= 0]])
    np.testing.assert_allclose(got, acc, rtol=1e-12)
    kw.close()


def test_lddrepair_and_lddmask_device_equal_the_host_helpers(amd):
    from lisflood_amd import ldd as L
    codes, mask = case()
    N = codes.size
    bad = codes.copy(); bad[3] = 0; bad[5] = 77; bad[9] = 2.5
    assert np.array_equal(L.lddrepair_device(bad, mask), L.lddrepair(bad, mask))
    down = L.downstream_index(codes, mask)
    target = int(down[down >= 0][7])                      # an unknown code on a pixel that HAS inflow: its feeders become pits
    bad2 = codes.copy(); bad2[target] = 77
    assert np.array_equal(L.lddrepair_device(bad2, mask), L.lddrepair(bad2, mask))
    assert (L.lddrepair_device(bad2, mask)[down == target] == L.PIT).all()
    a2, _ = L.lddmask_device(bad2, mask, np.ones(N, bool))
    b2, _ = L.lddmask(bad2, mask, np.ones(N, bool))
    assert np.array_equal(np.where(np.isin(a2, range(1, 10)), a2, 0), np.where(np.isin(b2, range(1, 10)), b2, 0))
    keep = np.random.default_rng(5).random(N) < 0.6
    a, am = L.lddmask_device(codes, mask, keep)
    b, bm = L.lddmask(codes, mask, keep)
    assert np.array_equal(a, b) and np.array_equal(am, bm)
    # the channel / overland LDDs of routing.initial (routing.py:118, 125)
    assert np.array_equal(L.lddrepair_device(np.where(keep, L.PIT, codes), mask), L.lddrepair(np.where(keep, L.PIT, codes), mask))


def test_catchment_reproduces_the_pcraster_masks_of_the_use_case(amd):
    """mask.map (the model domain of cold.xml) and subcatchment_mask.map (tests/test_subcatchments.py) are PCRaster
    catchments of LF_ETRS89's LDD: catchment(ldd, outlet of the mask) must give exactly the mask."""
    from lisflood_amd import ldd as L
    z = golden("etrs89_static")
    ldd = z["ldd"]
    land = ldd != -1
    codes = ldd[land].astype(np.float64)
    down = L.downstream_index(codes, land)
    d = L.LddDevice(codes, land)
    for key, cells in (("mask_map", 2847), ("subcatchment_mask", 1023)):
        m = z[key][land]
        assert m.sum() == cells
        inside_down = (down >= 0) & m[np.maximum(down, 0)]
        outlet = np.nonzero(m & ~inside_down)[0]
        assert outlet.size == 1                                   # one outlet: the mask is one catchment
        pts = np.zeros(codes.size, np.int64); pts[outlet[0]] = 1
        assert np.array_equal(d.catchment(pts) == 1, m), key
    # accuflux against ec_upArea.nc on the upstream-closed model domain
    m = z["mask_map"][land]
    up = d.accuflux(z["pixarea"][land].astype(np.float64))
    np.testing.assert_allclose(up[m], z["uparea"][land][m].astype(np.float64), rtol=1e-6)
    d.close()


def test_catchments_large_deep_raster(amd):
    """pointer jumping on a 10^3-level network: every cell's label is its outlet's"""
    from lisflood_amd import ldd as L
    from lisflood_amd import synthetic as syn
    H, W = 1500, 400
    codes = syn.make_ldd("deep", H, W, 8).reshape(-1).astype(np.float64)
    mask = np.ones((H, W), bool)
    down = L.downstream_index(codes, mask)
    d = L.LddDevice(codes, mask)
    lab = d.catchment(L.uniqueid(down < 0))
    assert (lab > 0).all()
    ok = down >= 0
    assert np.array_equal(lab[ok], lab[down[ok]])                  # a label never changes along a flow path
    assert np.array_equal(lab[~ok], L.uniqueid(down < 0)[~ok])
    d.close()


@pytest.mark.parametrize("engine_order", [True, False])
def test_channel_initialisation_reproduces_the_reference_on_cold_xml_inputs(amd, engine_order):
    """routing.initial -> lakes.initial -> reservoir.initial -> structures.initial -> routing.initialSecond on the real
    inputs of settings/cold.xml (model domain mask.map, 5 lakes, 31 reservoirs, avgdis of the use case's pre-run),
    against tests/golden/etrs89_initial.npz -- produced by the reference's OWN methods in that order (PCRaster emulated:
    tests/golden/make_golden.py gen_initial).  Pins a10, a11 (split branch + mass-balance start values), the lake and
    reservoir parameter derivation and the cut LDD."""
    import types
    from lisflood_amd import structures as ST
    g = golden("etrs89_initial")
    mask = g["mask"]
    maps = {k[4:]: (g[k] if g[k].ndim else float(g[k])) for k in g.files if k.startswith("map_")}
    tab = {k[6:]: g[k] for k in g.files if k.startswith("table_")}
    tables = dict(TabLakeArea=tab["lakearea"], TabLakeA=tab["lakea"], TabLakeAvNetInflowEstimate=tab["lakeavinflow"],
                  TabTotStorage=tab["rtstor"], TabConservativeStorageLimit=tab["rclim"], TabNormalStorageLimit=tab["rnlim"],
                  TabFloodStorageLimit=tab["rflim"], TabNonDamagingOutflowQ=tab["rndq"], TabNormalOutflowQ=tab["rnormq"],
                  TabMinOutflowQ=tab["rminq"])
    opts = dict(InitLisflood=False, SplitRouting=True, simulateLakes=True, simulateReservoirs=True, repMBTs=True)
    v = types.SimpleNamespace(DtSec=float(g["DtSec"]), DtSecChannel=float(g["DtSecChannel"]))
    m = amd.routing.routing(v, options=opts, engine_order=engine_order)
    m.initial(maps, mask)
    ST.lakes(v, opts, maps, tables).initial(mask)
    ST.reservoir(v, opts, maps, tables).initial()
    ST.structures(v, opts).initial(mask)
    m.initialSecond()
    exact = ("Ldd IsChannel LddKinematic LddToChan AtLastPointC downstruct Catchments LddStructuresKinematic "
             "IsStructureKinematic IsUpsOfStructureKinematicC IsUpsOfStructureLake LakeIndex ReservoirIndex").split()
    for k in exact:
        assert np.array_equal(np.asarray(getattr(v, k)).astype(np.float64), g["out_" + k].astype(np.float64)), k
    skip = set(exact) | {"UpArea", "StorageStepINIT", "DischargeM3StructuresIni", "InvCatchArea"}
    for k in [f[4:] for f in g.files if f.startswith("out_")]:
        if k in skip:
            continue
        np.testing.assert_allclose(np.asarray(getattr(v, k), np.float64).ravel(), g["out_" + k].astype(np.float64).ravel(),
                                   rtol=1e-14, atol=0, err_msg=k)          # host arithmetic in the same order: to the bit
    for k in ("UpArea", "StorageStepINIT", "DischargeM3StructuresIni", "InvCatchArea"):   # tree totals: summation order
        np.testing.assert_allclose(getattr(v, k), g["out_" + k], rtol=1e-12, err_msg=k)
    if not engine_order:
        # the router initialSecond built sweeps the cut LDD exactly like the reference's
        assert np.array_equal(m.river_router.pixels_ordered, g["router_pixels_ordered"])
        assert np.array_equal(m.river_router.order_start_stop, g["router_order_start_stop"])
    else:
        # engine order (the default) with structures: the graph also holds the structures' uncut links as zero-length
        # links, which puts the cells draining into a lake or reservoir on the structure's own level -- the same pixels,
        # still every cell after all of its upstream cells, but not the reference's level table
        po, ss = m.river_router.pixels_ordered, m.river_router.order_start_stop
        assert np.array_equal(np.sort(po), np.sort(g["router_pixels_ordered"]))
        level_of = np.empty(po.size, np.int64)
        for k, (a, b) in enumerate(ss):
            level_of[po[a:b]] = k
        down = np.asarray(m.river_router.downstream_lookup).astype(np.int64)
        has = down >= 0
        assert (level_of[down[has]] > level_of[np.nonzero(has)[0]]).all()


def test_initlisflood_prerun_reproduces_the_reference(amd):
    """The pre-run that produces avgdis (option InitLisflood): routing.initial forces NoRoutSteps = 1 (routing.py:78-79),
    initialSecond builds no split branch, routing.dynamic(0) takes the single branch even with SplitRouting on
    (:518-538), and the post-loop lines give ChanM3 / ChanQAvg / CumQ / avgdis (Lisflood_dynamic.py:194-226) -- against
    tests/golden/initlisflood_prerun.npz, six model steps driven by the reference's OWN routing.initial /
    initialSecond / dynamic on cold.xml's channel maps (make_golden.py gen_prerun)."""
    import types
    from lisflood_amd import structures as ST
    g0 = golden("etrs89_initial")
    g = golden("initlisflood_prerun")
    mask = g["mask"]
    maps = {k[4:]: (g0[k] if g0[k].ndim else float(g0[k])) for k in g0.files if k.startswith("map_")}
    opts = dict(InitLisflood=True, SplitRouting=True, simulateLakes=True, simulateReservoirs=True)
    v = types.SimpleNamespace(DtSec=float(g0["DtSec"]), DtSecChannel=float(g0["DtSecChannel"]))
    m = amd.routing.routing(v, options=opts)
    m.initial(maps, mask)
    assert v.NoRoutSteps == 1 == int(g["NoRoutSteps"]) and v.DtRouting == v.DtSec == float(g["DtRouting"])
    ST.structures(v, opts).initial(mask)          # (lakes / reservoirs are switched off by InitLisflood)
    m.initialSecond()
    assert np.array_equal(np.asarray(v.LddKinematic, np.float64), g["LddKinematic"])
    for k in ("ChanQKin", "ChanM3Kin", "ChanQ"):
        np.testing.assert_allclose(getattr(v, k), g["init_" + k], rtol=1e-14, err_msg=k)
    assert not hasattr(v, "Chan2QKin") or not m._split()
    for step in range(g["ToChanM3RunoffDt"].shape[0]):
        v.ToChanM3RunoffDt = g["ToChanM3RunoffDt"][step].copy()
        v.sumDisDay = np.zeros(v.ChanQKin.size)
        for s in range(v.NoRoutSteps):
            m.dynamic(s)
        m.step_end(time_since_start=float(step + 1))
        # discharges: rtol 1e-9 + the reference's own Newton tolerance (1e-12 m3/s, kinematic_wave_parallel_tools.py:26);
        # volumes V = L alpha Q^0.6 carry that absolute tolerance of Q relatively (dV/V = 0.6 dQ/Q): 1e-8 on nearly dry cells
        for k in ("ChanQ", "ChanQKin", "sumDisDay", "ChanQAvg", "avgdis"):
            np.testing.assert_allclose(getattr(v, k), g["out_" + k][step], rtol=1e-9, atol=1e-12, err_msg=(step, k))
        for k in ("ChanM3Kin", "ChanM3"):
            np.testing.assert_allclose(getattr(v, k), g["out_" + k][step], rtol=1e-8, atol=1e-9, err_msg=(step, k))


def test_total_time_series_and_water_use_sum_on_the_device_sweep(amd, tmp_path):
    """the two per-step consumers of accuflux the reference has outside routing: the 'total' time-series operation
    (global_modules/output.py:573: catchmenttotal(x * PixelArea, Ldd) * InvUpArea at the gauges) and WUseSumM3
    (waterabstraction.py:533: accuflux(Ldd, withdrawal * InvDtSec)), both on LddDevice's sweep, against a host
    accumulation in downstream order"""
    from lisflood_amd import ldd as L
    from lisflood_amd import output as O
    from lisflood_amd import synthetic as syn
    H, W = 120, 90
    mask = np.random.default_rng(2).random((H, W)) > 0.15
    codes = syn.make_ldd("river", H, W, 5, land_mask=mask)[mask].astype(np.float64)
    N = codes.size
    down = L.downstream_index(codes, mask)
    d = L.LddDevice(codes, mask)

    def host_accuflux(x):          # upstream cells first: process in order of decreasing distance to the outlet
        from lisflood_amd.kinematic_wave_parallel import Graph
        po, ss = Graph(codes, mask).orders()
        out = np.array(x, dtype=np.float64)
        for k in range(ss.shape[0]):
            for p in po[ss[k, 0]:ss[k, 1]]:
                if down[p] >= 0:
                    out[down[p]] += out[p]
        return out
    rng = np.random.default_rng(3)
    area = rng.uniform(2e7, 3e7, N)
    inv_up = 1.0 / host_accuflux(area)
    gauges = rng.choice(N, 12, replace=False)
    w = O.TssWriter(str(tmp_path / "tot.tss"), list(range(1, 13)), gauges, how="total", router=d, pixel_area=area,
                    inv_up_area=inv_up, date="D")
    xs = [rng.uniform(0, 5, N) for _ in range(3)]
    for x in xs:
        w.sample(x)
    w.close()
    vals = O.read_tss(str(tmp_path / "tot.tss"))[3]
    for row, x in zip(vals, xs):
        np.testing.assert_allclose(row, (host_accuflux(x * area) * inv_up)[gauges], rtol=5e-6)   # (%14g keeps 6 digits)
    withdrawal = rng.uniform(0, 1e4, N)
    np.testing.assert_allclose(d.water_use_sum(withdrawal, 1 / 86400.0), host_accuflux(withdrawal / 86400.0), rtol=1e-12)
    d.close()


# ---------------------------------------------------------------------------------------------------------------------
# a21, independent pin: the device forms against tests/golden/ldd_ops.npz, which tests/golden/pcr_naive.py makes by
# cell-by-cell walks restated from the PCRaster manual (numpy only; no code shared with lisflood_amd or oracle/).
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["etrs89", "syn48_holes"])
def test_device_ldd_operations_equal_the_independent_fixture(amd, name):
    """lddmask / lddrepair / pit / downstream / uniqueid / catchment / subcatchment / upstream / accuflux and the structures
    cut, in the order routing.initial (routing.py:90-171, 387) and structures.initial (structures.py:51-59) call them, on
    LF_ETRS89's LDD and on a raster with MV holes, non-keypad codes (0, 77, 2.5, NaN) and off-grid pointers"""
    from lisflood_amd import ldd as L
    z = golden("ldd_ops")
    g = {k[len(name) + 2:]: z[k] for k in z.files if k.startswith(name + "__")}
    c, land = g["codes"], g["land_mask"]
    N = c.size
    known = lambda a: np.where(np.isin(a, range(1, 10)), a, 0)
    mv = g["Ldd"] == 0
    sub, sub_mask = L.lddmask_device(c, land, g["domain"])                                            # routing.py:90
    assert np.array_equal(known(sub), g["lddmask_domain"][g["domain"]]) and sub_mask.sum() == g["domain"].sum()
    Ldd = L.lddrepair_device(c, land)                   # MV pixels become pits in the compressed form (ldd.lddrepair)
    assert np.array_equal(Ldd[~mv], g["Ldd"][~mv]) and (Ldd[mv] == L.PIT).all()
    d = ~mv
    dm = land.copy(); dm[land] = d
    ldd, chan = g["Ldd"][d], g["is_channel"][d]
    kin, kin_mask = L.lddmask_device(ldd, dm, chan)                                                   # routing.py:118
    assert np.array_equal(kin, g["LddChan"][d][chan])
    assert np.array_equal(L.lddrepair_device(np.where(chan, L.PIT, ldd), dm), g["LddToChan"][d])      # routing.py:125
    pits = L.pit(ldd)
    assert np.array_equal(pits, g["pit"][d])                                                          # routing.py:127
    dev = L.LddDevice(ldd, dm)
    assert np.array_equal(dev.downstream((pits != 0).astype(np.float64)), g["downstream_AtOutflow"][d])   # routing.py:141
    assert np.array_equal(dev.catchment(L.uniqueid(g["AtLastPoint"][d])), g["Catchments"][d])         # routing.py:168-170
    assert np.array_equal(dev.catchment(pits), g["catchment_of_pits"][d])
    assert np.array_equal(dev.catchment(g["points_nested"][d]), g["catchment_nested"][d])
    assert np.array_equal(dev.subcatchment(g["points_nested"][d]), g["subcatchment_nested"][d])
    w = g["w"][d]
    assert np.array_equal(dev.upstream(w), g["upstream_w_Ldd"][d])             # ascending source index: bit-exact
    np.testing.assert_allclose(dev.accuflux(w), g["accuflux_w"][d], rtol=1e-12)                       # routing.py:98
    dev.close()
    kdev = L.LddDevice(kin, kin_mask)
    ids = np.arange(N, dtype=np.float64)[d][chan]
    assert np.array_equal(kdev.downstream(ids), g["downstruct_ids"][d][chan])                         # routing.py:159-162
    assert np.array_equal(kdev.upstream(w[chan]), g["upstream_w"][d][chan])                           # routing.py:387
    st = g["is_structure"][d][chan]
    ups = kdev.downstream(st.astype(np.float64)) > 0                                                  # structures.py:51-53
    assert np.array_equal(ups, g["IsUpsOfStructure"][d][chan])
    cut = L.lddrepair_device(np.where(ups, L.PIT, kin), kin_mask)                                     # structures.py:59
    assert np.array_equal(cut, g["LddKinematic_cut"][d][chan])
    kdev.close()
    # the whole-raster one-hop reduction (LDS-staged 3 x 3 neighbourhoods) on the same LDD
    ras = np.zeros(land.shape, np.uint8); ras[dm] = ldd.astype(np.uint8)
    wr = np.zeros(land.shape); wr[dm] = w
    assert np.array_equal(L.upstream_raster(ras, wr)[dm], g["upstream_w_Ldd"][d])
