"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol the
header declares, struct mirrors have the right size, the host graph builder reproduces the reference's
lookups/orders on the golden vectors, and compute entry points fail loudly without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden

from lisflood_amd import _lib
from lisflood_amd import synthetic as syn
from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave

HEADER = os.path.join(ROOT, "include", "lisflood_amd.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    syms = declared_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(L, s), "missing export " + s


def test_struct_mirrors_have_the_c_size():
    from lisflood_amd import soilloop, routing, surface_routing
    out = (C.c_int64 * 7)()
    assert _lib.lib().lf_struct_sizes(out) == 0
    assert C.sizeof(routing._SubstepArgs) == out[0]
    assert C.sizeof(soilloop._InterceptionArgs) == out[1]
    assert C.sizeof(soilloop._SoilArgs) == out[2]
    assert C.sizeof(soilloop._CanopyArgs) == out[3]
    assert C.sizeof(surface_routing._SurfaceArgs) == out[4]
    assert C.sizeof(routing._InloopArgs) == out[5]
    from lisflood_amd import pixel_aggregates
    assert C.sizeof(pixel_aggregates._PixelArgs) == out[6]


@pytest.mark.parametrize("name", ["syn64_shallow", "syn64_deep", "syn48_masked", "etrs89"])
def test_graph_matches_reference_golden(name):
    g = golden("graph_" + name)
    G = Graph(g["codes"], g["mask"])
    down, ups, nups = G.lookups()
    po, ss = G.orders()
    assert np.array_equal(down, g["downstream_lookup"])
    assert np.array_equal(ups, g["upstream_lookup"])
    assert np.array_equal(nups, g["num_upstream_pixels"])
    assert np.array_equal(po, g["pixels_ordered"])
    assert np.array_equal(ss, g["order_start_stop"])


@pytest.mark.parametrize("name", ["syn64_shallow", "syn48_masked", "etrs89"])
def test_sweep_layout_invariants(name):
    """upstream cells of position p are the contiguous positions [ups_ptr[p], ups_ptr[p+1]) of the
    previous level, in ascending pixel id (the summation order of kinematic_wave_parallel_tools.py:57-58)."""
    g = golden("graph_" + name)
    G = Graph(g["codes"], g["mask"])
    perm, ups_ptr, level_start = G.layout()
    N = G.num_pixels
    assert sorted(perm.tolist()) == list(range(N))
    assert level_start[0] == 0 and level_start[-1] == N and ups_ptr[0] == 0
    ups = g["upstream_lookup"]
    level_of = np.empty(N, np.int64)
    for k in range(G.num_levels):
        level_of[level_start[k]:level_start[k + 1]] = k
    for p in range(N):
        want = [u for u in ups[perm[p]] if u >= 0]
        got = perm[ups_ptr[p]:ups_ptr[p + 1]].tolist()
        assert got == want
        if want:
            assert (level_of[ups_ptr[p]:ups_ptr[p + 1]] == level_of[p] - 1).all()


def test_raster_form_equals_compressed_form():
    codes = syn.make_ldd("shallow", 40, 50, 1)
    mask = np.ones((40, 50), bool)
    a = Graph(codes[mask].astype(float), mask)
    b = Graph(ldd_raster=codes)
    for x, y in zip(a.layout(), b.layout()):
        assert np.array_equal(x, y)
    m2 = mask.copy(); m2[:5, :7] = False
    c = Graph(codes[m2].astype(float), m2)
    d = Graph(ldd_raster=codes, land_mask=m2)
    for x, y in zip(c.layout(), d.layout()):
        assert np.array_equal(x, y)


def test_edge_graphs():
    one = Graph(np.array([5.0]), np.ones((1, 1), bool))
    assert (one.num_pixels, one.num_levels, one.max_upstream) == (1, 1, 1)
    with pytest.raises(_lib.LisfloodAmdError) as e:
        Graph(np.array([6.0, 4.0]), np.ones((1, 2), bool))      # 2-cycle: the reference hangs (kwp.py:99)
    assert e.value.code == _lib.LF_E_CYCLE
    with pytest.raises(ValueError):
        Graph(np.array([5.0]), np.ones((1, 2), bool))           # ragged: codes do not match the mask
    # unknown codes are "no flow" (documented deviation from the reference's uninitialised np.empty)
    odd = Graph(np.array([6.0, np.nan, 11.0, 5.0]), np.ones((1, 4), bool))
    assert odd.lookups()[0].tolist() == [1.0, -1.0, -1.0, -1.0]


def test_synthetic_generator_is_chunk_consistent_and_acyclic():
    for fam, seed in (("shallow", 1), ("deep", 2)):
        full = syn.make_ldd(fam, 300, 70, seed)
        part = syn.make_ldd(fam, 300, 70, seed, r0=250, r1=290)
        assert np.array_equal(full[250:290], part)
        G = Graph(ldd_raster=full)           # raises on a cycle
        assert G.num_pixels == 300 * 70
    assert Graph(ldd_raster=syn.make_ldd("deep", 300, 70, 2)).num_levels >= 300


@pytest.mark.skipif(_lib.device_count() > 0, reason="only meaningful on a box without a GPU")
def test_compute_fails_loudly_without_gpu():
    g = golden("graph_syn64_shallow")
    with pytest.raises(_lib.LisfloodAmdError) as e:
        kinematicWave(g["codes"], g["mask"], np.ones(4096), 0.6, 1000.0, 3600.0)
    assert e.value.code == _lib.LF_E_NO_DEVICE
    from lisflood_amd.soilloop import soilColumnsWaterBalance
    d = syn.soil_params(8)
    with pytest.raises(_lib.LisfloodAmdError):
        soilColumnsWaterBalance(*[d[k] for k in syn.SOIL_ARG_ORDER])
    # the entry points added in round 5 fail the same way (page-locked host memory needs a HIP device too)
    import ctypes as C
    L = _lib.lib()
    p = C.c_void_p()
    assert L.lf_host_alloc(C.c_int(0), C.c_size_t(64), C.byref(p)) == _lib.LF_E_NO_DEVICE and not p.value
    from lisflood_amd.soilloop import _SoilArgs
    a = _SoilArgs()
    idx = np.zeros(3, np.int64)
    a.index_landuse_all = idx.ctypes.data
    a.V, a.L, a.N = 3, 3, 8
    assert L.lf_soil_columns_device_derived(C.c_int(0), C.byref(a)) == _lib.LF_E_NO_DEVICE
    buf = np.zeros(4, np.float32)
    assert L.lf_upload_copy_f32(C.c_int(0), C.c_void_p(8), buf.ctypes.data_as(C.c_void_p), C.c_size_t(4)) == _lib.LF_E_NO_DEVICE
    # round 6: the land surface in one pass checks its two argument blocks against each other BEFORE it needs a device
    from lisflood_amd.soilloop import _CanopyArgs
    cn = _CanopyArgs()
    cn.index_landuse = idx.ctypes.data
    cn.V, cn.L, cn.N = 3, 3, 8
    es = C.c_void_p(8)
    assert L.lf_land_columns_device(C.c_int(0), C.byref(cn), C.byref(a), es, C.c_int(1)) == _lib.LF_E_INVALID   # rows 0,0,0
    assert b"index_landuse" in L.lf_last_error()
    idx[:] = [0, 1, 2]
    cn.W1a = 64                                                     # canopy and soil name different W1a vectors
    assert L.lf_land_columns_device(C.c_int(0), C.byref(cn), C.byref(a), es, C.c_int(1)) == _lib.LF_E_INVALID
    assert b"different vectors" in L.lf_last_error()
    cn.W1a = None
    assert L.lf_land_columns_device(C.c_int(0), C.byref(cn), C.byref(a), es, C.c_int(1)) == _lib.LF_E_NO_DEVICE
    assert L.lf_upload_wait(C.c_int(0), C.c_int(0)) == _lib.LF_E_NO_DEVICE
    assert L.lf_device_trim(C.c_int(0)) == _lib.LF_E_NO_DEVICE


def test_graph_with_structure_links():
    """lf_graph_create_ex: the pits cut just upstream of a structure are put on the structure's level (zero-length
    links) at the end of that level, outside every upstream range; everything else keeps the sweep-order invariants."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
    from lisflood_amd import ldd as L
    from lisflood_amd.kinematic_wave_parallel import Graph
    z = np.load(os.path.join(ROOT, "tests", "golden", "inloop_structures.npz"))
    mask, cut = z["mask"], z["codes_cut"]
    N = cut.size
    ds = z["downstruct"].astype(np.int64)
    sites = np.concatenate([z["LakeIndex"], z["ReservoirIndex"]]).astype(np.int64)
    is_site = np.zeros(N + 1, bool); is_site[sites] = True
    vd = np.where(is_site[np.minimum(ds, N)] & (ds < N), ds, -1)
    assert (vd >= 0).sum() > 50
    g0, g = Graph(cut, mask), Graph(cut, mask, virtual_down=vd)
    perm, ups_ptr, level_start = g.layout()
    linked = g.links()
    pos = np.empty(N, np.int64); pos[perm] = np.arange(N)
    level = np.searchsorted(level_start, np.arange(N), side="right") - 1          # by position
    assert np.array_equal(np.sort(perm), np.arange(N))
    assert linked.sum() == (vd >= 0).sum() and np.array_equal(np.nonzero(linked)[0], np.sort(pos[vd >= 0]))
    # links: same level as their structure cell
    u = np.nonzero(vd >= 0)[0]
    assert np.array_equal(level[pos[u]], level[pos[vd[u]]])
    # real edges: strictly increasing level, and the upstream ranges (minus links) are the real upstream cells
    down = g.lookups()[0].astype(np.int64)
    assert np.array_equal(down, g0.lookups()[0].astype(np.int64))
    has = down >= 0
    assert (level[pos[np.nonzero(has)[0]]] + 1 == level[pos[down[has]]]).all()
    for p in range(N):
        rng = np.arange(ups_ptr[p], ups_ptr[p + 1])
        rng = rng[~linked[rng]]
        want = np.nonzero(down == perm[p])[0]
        assert np.array_equal(perm[rng], want), p
    assert g.num_levels >= g0.num_levels
    # a link whose source is not a pit is rejected
    bad = np.full(N, -1, np.int64)
    src = int(np.nonzero(has)[0][0])
    bad[src] = int(down[src])
    with pytest.raises(Exception):
        Graph(cut, mask, virtual_down=bad)


def test_tss_writer_reproduces_the_reference_file(tmp_path):
    """The reference's own dis.tss (LF_ETRS89_UseCase/reference/output_reference_daily) parsed and written back must
    be byte-identical: header layout, ` %8g` step column, ` %14g` values (zusatz.py:201-290)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
    from lisflood_amd import output as O
    src = os.path.join(ROOT, "tests", "golden", "etrs89_dis_reference.tss")
    first, ids, t0, vals = O.read_tss(src)
    assert len(ids) == 30 and t0 == 9497 and vals.shape == (216 - 33, 30) and np.isfinite(vals).all()
    dst = tmp_path / "dis.tss"
    O.write_tss(str(dst), ids, t0, vals, first_line=first)
    assert open(src).read() == open(dst).read()
    # missing values and the writer object
    w = O.TssWriter(str(tmp_path / "x.tss"), [7, 9], [2, 0], first_timestep=5, date="D")
    w.sample(np.array([1.5, 2.0, np.nan])); w.sample(np.array([1e-7, 2.0, 123456789.0]))
    w.close()
    txt = open(tmp_path / "x.tss").read().splitlines()
    assert txt[0] == "timeseries valuescale.scalar settingsfile:  date: D" and txt[1:5] == ["3", "timestep", "7", "9"]
    assert txt[5] == "        5           1e31            1.5" and txt[6] == "        6    1.23457e+08          1e-07"


def test_netcdf_classic_maps_round_trip(tmp_path):
    """dis-style [T, H, W] stack and a single state map written in the reference's structure (netcdf.py:432-583:
    time / y / x, _FillValue -9999 outside the land mask) and read back."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
    from lisflood_amd import output as O
    rng = np.random.default_rng(0)
    mask = rng.random((7, 9)) > 0.3
    N = int(mask.sum())
    x, y = 4e6 + 5000.0 * np.arange(9), 3e6 - 5000.0 * np.arange(7)
    stack = np.stack([O.decompress(rng.random(N), mask) for _ in range(4)])
    O.write_netcdf_classic(str(tmp_path / "dis.nc"), "dis", np.where(stack == O.FILL, np.nan, stack), x, y,
                           time_values=[1, 2, 3, 4], standard_name="dis", long_name="discharge", units="m3/s")
    a, rx, ry, t = O.read_netcdf_classic(str(tmp_path / "dis.nc"), "dis")
    assert np.array_equal(rx, x) and np.array_equal(ry, y) and np.array_equal(t, [1, 2, 3, 4])
    assert np.array_equal(np.isnan(a), np.broadcast_to(~mask, a.shape)) and np.array_equal(a[:, mask], stack[:, mask])
    O.write_netcdf_classic(str(tmp_path / "ch.nc"), "chanq", O.decompress(np.arange(N), mask, np.nan), x, y)
    b, _, _, t = O.read_netcdf_classic(str(tmp_path / "ch.nc"), "chanq")
    assert t is None and np.array_equal(b[mask], np.arange(N)) and np.isnan(b[~mask]).all()
    assert open(tmp_path / "dis.nc", "rb").read(4) == b"CDF\x02"


@pytest.mark.parametrize("family", ["shallow", "deep"])
def test_catchment_partition_is_closed_and_balanced(family):
    """Every pixel's outlet label equals the outlet reached by walking the LDD; a rank's pixels form a self-contained
    domain (no link leaves it) and the ranks are balanced up to the largest catchment."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
    from lisflood_amd import partition as P, synthetic as syn
    from lisflood_amd.kinematic_wave_parallel import Graph
    H, W = 90, 120
    mask = np.ones((H, W), bool); mask[:7, :9] = False
    raster = syn.make_ldd(family, H, W, 3, land_mask=mask)
    codes = raster[mask].astype(np.float64)
    g = Graph(codes, mask)
    down = g.lookups()[0].astype(np.int64)
    roots = P.catchment_roots(g)
    walk = np.arange(down.size)
    for _ in range(H * W):
        nxt = np.where(down[walk] >= 0, down[walk], walk)
        if np.array_equal(nxt, walk):
            break
        walk = nxt
    assert np.array_equal(roots, walk)
    rank, sizes = P.split_catchments(roots, 4)
    cnt = np.bincount(rank, minlength=4)
    assert cnt.sum() == down.size and cnt.max() - cnt.min() <= 2 * sizes.max()
    has = down >= 0
    assert np.array_equal(rank[has], rank[down[has]])                       # no link crosses ranks
    parts, counts = P.catchment_partition(codes, mask, 4)
    assert np.array_equal(counts, cnt)
    for c, m, ids in parts:
        gs = Graph(c, m)                                                     # builds: a closed sub-domain
        assert gs.num_pixels == ids.size and (gs.lookups()[0] >= -1).all()


def test_inert_pixels_and_subdomain_renumbering():
    """Host logic of the compact channel domain (lisflood_amd.hotpath): a pixel may be left out only if EVERY condition
    holds -- isolated in the kinematic LDD, not a channel pixel, regular parameters, zero thresholds, +0.0 state, no
    structure on it -- and the structure attributes are renumbered consistently."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))
    from lisflood_amd import hotpath as HP, synthetic as syn
    H, W = 30, 40
    values, sc, mask, l2c, kin = syn.hotpath_scenario(H, W)
    N = H * W
    base = HP.inert_pixels(values, kin, mask, True)
    non = np.nonzero(~values["IsChannelKinematic"])[0]
    assert base.sum() == non.size and not base[values["IsChannelKinematic"]].any()
    def with_(k, i, x):
        v = {a: (np.array(b, copy=True) if isinstance(b, np.ndarray) else b) for a, b in values.items()}
        v[k][i] = x
        return HP.inert_pixels(v, kin, mask, True)
    p = int(non[3])
    for k, x in (("ChanQKin", 1e-300), ("ChanM3Kin", 2.0), ("ChanQ", -0.0), ("Chan2QKin", 1.0), ("Chan2M3Kin", 1.0),
                 ("CrossSection2Area", 1.0), ("Sideflow1Chan", 1.0), ("QLimit", 3.0), ("Chan2QStart", 1.0),
                 ("Chan2M3Start", 1.0), ("InvChannelAlpha", np.inf), ("InvChanLength", np.nan), ("InvChannelAlpha2", np.inf)):
        r = with_(k, p, x)
        assert not r[p] and r.sum() == base.sum() - 1, k
    assert with_("IsChannelKinematic", p, True)[p] == False
    # a pixel that receives flow, or sends it, is never inert even if it is not a channel pixel
    kin2 = kin.copy()
    kin2[p] = 6.0 if (p % W) < W - 1 else 4.0
    r = HP.inert_pixels(values, kin2, mask, True)
    tgt = p + 1 if (p % W) < W - 1 else p - 1
    assert not r[p] and not r[tgt]
    # without split routing the floodplain vectors do not matter
    v = dict(values); v["QLimit"] = values["QLimit"] + 1.0
    assert HP.inert_pixels(v, kin, mask, False).sum() == base.sum()
    # structures: a site pixel stays; indices and downstruct are renumbered onto the kept pixels
    st = {"LakeIndex": np.array([int(non[5])]), "ReservoirIndex": np.array([int(np.nonzero(~base)[0][7])]),
          "downstruct": np.arange(1, N + 1).astype(np.int32), "QInM3Old": np.zeros(N), "QDelta": np.zeros(N)}
    st["QInM3Old"][non[9]] = 5.0
    r = HP.inert_pixels(values, kin, mask, True, st)
    assert not r[non[5]] and not r[non[9]] and r.sum() == base.sum() - 2
    ids = np.nonzero(~r)[0]
    sub = HP._structures_on_subdomain(st, ids, N)
    assert ids[sub["LakeIndex"][0]] == non[5] and ids[sub["ReservoirIndex"][0]] == st["ReservoirIndex"][0]
    ds = sub["downstruct"]
    assert ds.size == ids.size and ((ds == ids.size) | (ids[np.minimum(ds, ids.size - 1)] == st["downstruct"][ids])).all()
    assert sub["QInM3Old"].size == ids.size and sub["QInM3Old"].sum() == 5.0


def test_root_floor_constant_is_the_exact_threshold_of_the_fifth_power():
    """lf_math.h decides `root^5 > 1e-12` (kinematic_wave_parallel_tools.py:77,81-82) on the root itself:
    LF_ROOT_FLOOR must be the smallest double whose fifth power -- rounded product by product as the kernel forms it --
    exceeds 1e-12, so that the two tests agree for every double (the product is monotone in r)."""
    import re
    src = open(os.path.join(ROOT, "lisflood-code_amd", "csrc", "lf_math.h")).read()
    floor = float.fromhex(re.search(r"#define LF_ROOT_FLOOR (\S+)", src).group(1))

    def q(r):
        r = np.float64(r)
        r2 = np.float64(r * r)
        return np.float64(np.float64(r2 * r2) * r)
    assert q(floor) > 1e-12 and not q(np.nextafter(floor, 0.0)) > 1e-12
    r = np.sort(np.concatenate([np.random.default_rng(1).uniform(0.5 * floor, 2 * floor, 200000),
                                floor * (1 + np.arange(-2000, 2000) * 2.0 ** -52)]))
    r2 = r * r
    np.testing.assert_array_equal(r >= floor, (r2 * r2) * r > 1e-12)


def test_static_upload_fingerprint_sees_every_element():
    """BufferCache.put_static skips an upload while the content checksum of the host array is unchanged: an in-place edit
    of ANY element must change it (rounds 2-3 fingerprinted a strided sample of 4096 elements and missed edits between
    the sample points), also in the odd bytes of a buffer that is not a multiple of 8 bytes long"""
    fp = _lib.BufferCache._fingerprint
    rng = np.random.default_rng(5)
    x = rng.random(2_000_003)
    f0 = fp(x)
    assert fp(x.copy()) == f0                       # content, not address
    for i in rng.integers(0, x.size, 50):
        old = x[i]
        x[i] = np.nextafter(old, 2.0)
        assert fp(x) != f0, i
        x[i] = old
    assert fp(x) == f0
    b = np.ones(1003, np.uint8)
    f1 = fp(b)
    b[1002] = 0
    assert fp(b) != f1
    assert fp(x.reshape(-1, 1)) != f0               # shape is part of it


def test_tss_writer_operations_total_and_mapmaximum(tmp_path):
    """the `operation` of the reference's time-series definitions (global_modules/output.py:568-575): 'total' samples
    catchmenttotal(x * PixelArea, Ldd) * InvUpArea, 'mapmaximum' the map's maximum.  CPU: the accumulation is a plain
    walk over a 7-pixel chain with one tributary (the device sweep is tied to it in tests/test_ldd_gpu.py)"""
    from lisflood_amd import output as O
    #   0 -> 1 -> 2 -> 3 (outlet), 4 -> 5 -> 2, 6 isolated
    down = np.array([1, 2, 3, -1, 5, 2, -1])

    class Walk:
        @staticmethod
        def accuflux(x):
            out = np.array(x, dtype=np.float64)
            for start in range(down.size):
                p = down[start]
                while p >= 0:
                    out[p] += x[start]
                    p = down[p]
            return out
    area = np.array([1.0, 2.0, 1.0, 4.0, 1.0, 1.0, 3.0])
    up_area = Walk.accuflux(area)
    x1, x2 = np.arange(7.0), np.array([5.0, 0, 0, 1, 2, 2, 9])
    w = O.TssWriter(str(tmp_path / "t.tss"), [11, 12, 13], [3, 2, 6], how="total", router=Walk, pixel_area=area,
                    inv_up_area=1 / up_area, date="D")
    w.sample(x1); w.sample(x2); w.close()
    vals = O.read_tss(str(tmp_path / "t.tss"))[3]
    for row, x in zip(vals, (x1, x2)):
        want = [(x * area).sum() - x[6] * area[6], (x * area)[[0, 1, 2, 4, 5]].sum(), x[6] * area[6]]
        np.testing.assert_allclose(row, np.array(want) / up_area[[3, 2, 6]], rtol=5e-6)      # (%14g keeps 6 digits)
    m = O.TssWriter(str(tmp_path / "m.tss"), [1, 2], [0, 4], how="mapmaximum", date="D")
    m.sample(x2); m.close()
    assert (O.read_tss(str(tmp_path / "m.tss"))[3] == 9.0).all()
    with pytest.raises(ValueError):
        O.TssWriter(str(tmp_path / "e.tss"), [1], [0], how="total")


def test_block_plan_statistics_of_a_graph():
    """lf_graph_block_plan_stats (host only): the plan a router would sweep the graph with -- every cell of a multi-level
    block is counted once, the cones tile the levels (cone levels x 64 lanes hold them), shorter blocks are fuller"""
    codes = syn.make_ldd("river", 300, 260, 4)
    g = Graph(ldd_raster=codes)
    lib = _lib.lib()

    def stats(lmax, max_cone):
        o = (C.c_int64 * 6)()
        _lib.check(lib.lf_graph_block_plan_stats(g._h, lmax, 262144, max_cone, o))
        return list(o)
    perm, ups_ptr, level_start = g.layout()
    a = stats(16, 64)
    b = stats(256, 64)
    for blocks, multi, cones, cone_levels, cells, most in (a, b):
        assert 0 < multi <= blocks and cones >= multi and most <= cones
        assert cells <= 300 * 260 and cells <= 64 * cone_levels
    assert a[0] > b[0]                                   # more, shorter blocks
    assert a[4] / (64.0 * a[3]) > b[4] / (64.0 * b[3])   # ... with fuller wavefronts
    one = stats(1, 64)                                   # one level per block: nothing is swept cone by cone
    assert one[1] == 0 and one[0] == level_start.size - 1
    g.close()


@pytest.mark.parametrize("family,H,W", [("deep", 220, 180), ("river", 300, 260), ("shallow", 400, 400), ("saddle", 200, 150)])
def test_cone_plan_invariants_the_kernels_rely_on(family, H, W):
    """k_sweep_cones_split / k_fused_cones_split read a cell's upstream discharges from the LDS row of the level above at
    slot (upstream position - start of the cone's range above): every upstream range must lie inside that range, the slot
    must fit the row (64), the count the eight address slots -- for the router plan (blocks of 256 levels) and the fused
    plans (16 and 32 levels), also on a masked raster and on a graph with structure links parked at the end of levels"""
    lib = _lib.lib()

    def check(g, lmax):
        o = (C.c_int64 * 4)()
        _lib.check(lib.lf_graph_block_plan_check(g._h, lmax, 262144, 64, o))
        bad, slots, cnt, cells = list(o)
        assert bad == 0 and slots <= 64 and cnt <= 8, (family, lmax, list(o))
        return cells
    codes = syn.make_ldd(family, H, W, 9)
    g = Graph(ldd_raster=codes)
    assert sum(check(g, lmax) for lmax in (256, 32, 16)) > 0 or family == "shallow"
    g.close()
    mask = np.random.default_rng(1).random((H, W)) > 0.25
    gm = Graph(ldd_raster=syn.make_ldd(family, H, W, 9, land_mask=mask), land_mask=mask)
    for lmax in (256, 16):
        check(gm, lmax)
    gm.close()
    # structure links: cut the LDD at a few cells and hand their inflow cells back as zero-length links
    from lisflood_amd import ldd as L
    flat = codes.reshape(-1).astype(np.float64)
    allm = np.ones((H, W), bool)
    down = L.downstream_index(flat, allm)
    rng = np.random.default_rng(2)
    sites = rng.choice(np.nonzero(down >= 0)[0], 12, replace=False)
    is_site = np.zeros(flat.size, bool); is_site[sites] = True
    feeds = (down >= 0) & is_site[np.maximum(down, 0)]
    cut = flat.copy(); cut[feeds] = L.PIT
    vd = np.where(feeds, down, -1)
    gl = Graph(cut, allm, virtual_down=vd)
    for lmax in (32, 16):
        check(gl, lmax)
    gl.close()


@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 63, 85, 107, 2040])
def test_cones_of_a_launch_are_dealt_out_xcd_contiguously(n):
    """lf_xcd_contiguous (lf_blocks.h; the function k_fused_cones calls, here through its host entry): whatever the linear
    id of a (block, sub-step)'s first workgroup, the launch positions map ONTO the cones 0..n-1, and the positions of one
    XCD (linear id mod 8) get one run of consecutive cones"""
    lib = _lib.lib()
    for first in (0, 1, 5, 13, 4095, 1000003, 0xfffffff9):
        out = (C.c_int32 * n)()
        _lib.check(lib.lf_xcd_contiguous_order(C.c_int(n), C.c_uint(first), out))
        order = np.array(out[:], dtype=np.int64)
        assert np.array_equal(np.sort(order), np.arange(n)), (n, first)
        xcd = (first + np.arange(n, dtype=np.uint64)) % 2**32 % 8
        for c in range(8):
            mine = np.sort(order[xcd == c])
            assert mine.size == 0 or np.array_equal(mine, np.arange(mine[0], mine[0] + mine.size)), (n, first, c)
            # ... in launch order: a workgroup's successor on the same XCD takes the next cone
            assert np.array_equal(order[xcd == c], mine), (n, first, c)


def test_derived_soil_parameters_check_is_exact():
    """soilloop.derived_parameters_hold: True exactly when the ten derived parameter arrays are what soil.py:180-228 makes of
    the others, bit for bit (then the device recomputes them instead of reading them)"""
    from lisflood_amd.soilloop import derived_parameters_hold
    d = syn.soil_params(257, seed=5)
    assert derived_parameters_hold(d)
    for name, bump in (("WS1", lambda a: a * (1 + 2e-16)), ("GenuInvM2", lambda a: np.nextafter(a, np.inf)),
                       ("WWP1", lambda a: a + 1e-13), ("PoreSpaceNotZero1b", lambda a: ~a)):
        e = dict(d)
        e[name] = bump(np.array(d[name], copy=True))
        assert not derived_parameters_hold(e), name
    e = dict(d)
    del e["WFC1"]
    assert not derived_parameters_hold(e)


def test_hotpath_scenario_drawn_for_a_block_and_repeated():
    """synthetic.hotpath_scenario(block=...): per-pixel fields drawn for `block` pixels and repeated, the LDD and the channel
    mask at full size; without `block` nothing changes (the fixtures and GPU tests rely on that)"""
    H, W = 30, 40
    N = H * W
    full, sc, mask, l2c, lk = syn.hotpath_scenario(H, W)
    tiled, sc2, mask2, l2c2, lk2 = syn.hotpath_scenario(H, W, block=500)
    assert sc == sc2 and np.array_equal(l2c, l2c2) and np.array_equal(lk, lk2) and np.array_equal(mask, mask2)
    assert np.array_equal(full["IsChannel"], tiled["IsChannel"])
    for k, a in tiled.items():
        assert a.shape[-1] == N and a.flags.c_contiguous, k
    assert np.array_equal(tiled["KSat1a"][:, :500], tiled["KSat1a"][:, 500:1000])          # repeated block
    chan = tiled["IsChannel"]
    assert (tiled["ChanQKin"][~chan] == 0).all() and (tiled["ChannelAlpha"][~chan] == 1).all()
    assert np.allclose(tiled["InvChannelAlpha2"], 1 / tiled["ChannelAlpha2"])
    river = syn.hotpath_scenario(H, W, family="river", block=500)[3]
    assert river.shape == (N,) and not np.array_equal(river, l2c)
