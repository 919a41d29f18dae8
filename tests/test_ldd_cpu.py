"""LDD operations restated from PCRaster's documented semantics (lisflood_amd/ldd.py), checked against brute
force walks on small catchments.  CPU only (host-side, init-time code)."""
import numpy as np
import pytest

from lisflood_amd import ldd as L
from lisflood_amd import synthetic as syn


def walk_down(down, p):
    path = [p]
    while down[p] >= 0:
        p = down[p]
        path.append(p)
    return path


def case():
    H, W = 23, 31
    mask = np.ones((H, W), bool); mask[:4, :6] = False; mask[12, 10:14] = False
    codes = syn.make_ldd("deep", H, W, 21, land_mask=mask)[mask]
    return codes, mask


def test_downstream_and_downstruct():
    codes, mask = case()
    down = L.downstream_index(codes, mask)
    x = np.arange(codes.size, dtype=float) * 1.5
    d = L.downstream(codes, mask, x)
    for p in range(codes.size):
        assert d[p] == (x[down[p]] if down[p] >= 0 else x[p])
    ds = L.downstruct(codes, mask)
    assert ((ds == codes.size) == (down < 0)).all() and (ds[down >= 0] == down[down >= 0]).all()


def test_catchment_labels_by_walking():
    codes, mask = case()
    down = L.downstream_index(codes, mask)
    outlets = L.uniqueid(down < 0)
    lab = L.catchment(codes, mask, outlets)
    assert (lab > 0).all()
    for p in range(0, codes.size, 7):
        assert lab[p] == outlets[walk_down(down, p)[-1]]
    # subcatchment: interior points override what lies downstream of them; catchment: the enclosing point wins
    pts = np.zeros(codes.size, np.int64); pts[[50, 300, 400]] = [7, 8, 9]
    pts[walk_down(down, 50)[3]] = 11                                  # a point downstream of point 7
    lab2 = L.subcatchment(codes, mask, pts)
    lab3 = L.catchment(codes, mask, pts)
    for p in range(codes.size):
        hit = [pts[q] for q in walk_down(down, p) if pts[q]]
        assert lab2[p] == (hit[0] if hit else 0)
        assert lab3[p] == (hit[-1] if hit else 0)
    assert lab2[50] == 7 and lab3[50] == 11


def test_lddmask_and_repair_make_pits_at_the_cut():
    codes, mask = case()
    N = codes.size
    keep = np.ones(N, bool); keep[N // 2:] = False
    sub_codes, sub_mask = L.lddmask(codes, mask, keep)
    assert sub_mask.sum() == keep.sum()
    down_full = L.downstream_index(codes, mask)
    cut = (down_full >= 0) & keep & ~keep[np.maximum(down_full, 0)]
    assert (sub_codes[cut[keep]] == L.PIT).all()
    down_sub = L.downstream_index(sub_codes, sub_mask)            # acyclic and closed inside the sub-mask
    assert (down_sub < sub_codes.size).all()
    bad = codes.copy(); bad[3] = 0; bad[5] = 77
    rep = L.lddrepair(bad, mask)
    assert rep[3] == L.PIT and rep[5] == L.PIT
    # a land pixel with an unknown code is a missing value of the ldd map: what drains INTO it becomes a pit as well
    target = int(down_full[down_full >= 0][7])
    feeders = np.nonzero(down_full == target)[0]
    bad2 = codes.copy(); bad2[target] = 77
    rep2 = L.lddrepair(bad2, mask)
    assert rep2[target] == L.PIT and (rep2[feeders] == L.PIT).all()
    others = np.setdiff1d(np.arange(N), np.append(feeders, target))
    assert np.array_equal(rep2[others], L.lddrepair(codes, mask)[others])
    m2, _ = L.lddmask(bad2, mask, np.ones(N, bool))
    assert (m2[feeders] == L.PIT).all()
    assert (L.pit(rep) > 0).sum() == (rep == L.PIT).sum() and L.pit(rep).max() == (rep == L.PIT).sum()


def test_cut_at_structures_matches_the_reference_driven_fixture():
    """The LDD cut of the in-loop fixture was produced by the generator with the reference's rule
    (structures.py:51-59) on LF_ETRS89's real lake / reservoir sites."""
    from conftest import golden
    g = golden("inloop_structures")
    z = golden("etrs89_static")
    mask = g["mask"]
    codes = z["ldd"][mask].astype(float)
    is_struct = np.zeros(codes.size, bool)
    is_struct[g["LakeIndex"]] = True
    is_struct[g["ReservoirIndex"]] = True
    cut, ups = L.cut_at_structures(codes, mask, is_struct)
    want = g["codes_cut"].copy()
    off_mask = (L.downstream_index(want, mask) < 0)            # lddrepair also pits cells leaving the mask
    assert np.array_equal(cut[~off_mask], want[~off_mask]) and (cut[off_mask] == L.PIT).all()
    assert ups.sum() > 0 and (cut[ups] == L.PIT).all()


def test_host_catchment_reproduces_the_pcraster_masks_of_the_use_case():
    """mask.map and subcatchment_mask.map of LF_ETRS89 are PCRaster-made catchments of ec_ldd: the only `catchment`
    outputs of PCRaster in the reference's checkout.  catchment(ldd, the mask's outlet) must give exactly the mask."""
    from conftest import golden
    z = golden("etrs89_static")
    land = z["ldd"] != -1
    codes = z["ldd"][land].astype(float)
    down = L.downstream_index(codes, land)
    for key in ("mask_map", "subcatchment_mask"):
        m = z[key][land]
        outlet = np.nonzero(m & ~((down >= 0) & m[np.maximum(down, 0)]))[0]
        assert outlet.size == 1
        pts = np.zeros(codes.size, np.int64); pts[outlet[0]] = 1
        assert np.array_equal(L.catchment(codes, land, pts) == 1, m), key


def test_cyclic_ldd_is_an_error_by_default_and_can_be_broken_explicitly():
    """routing.py:125 relies on PCRaster's lddrepair to make a cyclic ldd sound.  Here: the graph builder refuses the
    cycle with LF_E_CYCLE and a message that names the repair (the documented deviation); ldd.break_cycles /
    lddrepair(..., break_cycles_too=True) turn the FIRST cell of every cycle in row-major order into a pit, after which
    the graph builds, every other cell keeps its direction and all of them drain to the new pit"""
    from lisflood_amd import _lib
    from lisflood_amd import ldd as L
    from lisflood_amd.kinematic_wave_parallel import Graph
    # 4 x 5 raster: a 4-cycle (cells 6 -> 7 -> 12 -> 11 -> 6), a 2-cycle (3 <-> 4), two trees hanging on the 4-cycle,
    # the rest draining to the pit in the lower right corner
    codes = np.array([[6, 2, 4, 6, 4],
                      [6, 6, 2, 2, 2],
                      [6, 8, 4, 6, 2],
                      [6, 6, 6, 6, 5]], np.float64)
    mask = np.ones(codes.shape, bool)
    flat = codes.reshape(-1)
    with pytest.raises(_lib.LisfloodAmdError) as e:
        Graph(flat, mask)
    assert e.value.code == _lib.LF_E_CYCLE
    fixed, ncycles = L.break_cycles(flat, mask)
    assert ncycles == 2
    assert np.array_equal(np.nonzero(fixed != flat)[0], [3, 6]) and (fixed[[3, 6]] == L.PIT).all()
    np.testing.assert_array_equal(L.lddrepair(flat, mask, break_cycles_too=True), fixed)
    g = Graph(fixed, mask)
    roots_of = L.catchment(fixed, mask, L.pit(fixed))
    pits = L.pit(fixed)
    assert roots_of[4] == pits[3] and roots_of[3] == pits[3]                          # the 2-cycle drains to cell 3
    for cell in (0, 1, 5, 6, 7, 10, 11, 12):                                          # the 4-cycle and its trees: to cell 6
        assert roots_of[cell] == pits[6], cell
    assert roots_of[15] == pits[19] and roots_of[2] == pits[6]
    g.close()
    assert L.break_cycles(fixed, mask)[1] == 0                                        # sound now: nothing to break


# ---------------------------------------------------------------------------------------------------------------------
# the independent pin of a21: tests/golden/ldd_ops.npz is made by tests/golden/pcr_naive.py -- cell-by-cell walks
# restated from the PCRaster manual, numpy only, no code shared with lisflood_amd -- on LF_ETRS89's real LDD and on a
# seeded raster with MV holes, non-keypad codes (0, 77, 2.5, NaN) and cells pointing off the grid / into holes.
# ---------------------------------------------------------------------------------------------------------------------
def ldd_ops_case(name):
    from conftest import golden
    z = golden("ldd_ops")
    return {k[len(name) + 2:]: z[k] for k in z.files if k.startswith(name + "__")}


def known_only(a):
    """MV (0) where PCRaster has a missing value; the product's compressed forms put a pit there (documented in
    ldd.lddrepair: a compressed vector has no missing values)"""
    return np.where(np.isin(a, range(1, 10)), a, 0)


@pytest.mark.parametrize("name", ["etrs89", "syn48_holes"])
def test_host_ldd_operations_equal_the_independent_fixture(name):
    g = ldd_ops_case(name)
    c, land = g["codes"], g["land_mask"]
    N = c.size
    mv = g["Ldd"] == 0                                            # land pixels whose code is a missing value
    # lddmask(ldd, domain): routing.py:90
    sub, sub_mask = L.lddmask(c, land, g["domain"])
    want = g["lddmask_domain"]
    assert np.array_equal(known_only(sub), want[g["domain"]]) and sub_mask.sum() == g["domain"].sum()
    # lddrepair over the whole land mask: MV pixels become pits in the compressed form, every other cell is PCRaster's
    Ldd = L.lddrepair(c, land)
    assert np.array_equal(Ldd[~mv], g["Ldd"][~mv]) and (Ldd[mv] == L.PIT).all()
    # from here on the sound Ldd of the fixture restricted to its defined cells is the input (as routing.initial sees it)
    d = ~mv
    dm = land.copy(); dm[land] = d
    ldd = g["Ldd"][d]
    chan = g["is_channel"][d]
    kin, kin_mask = L.lddmask(ldd, dm, chan)                                                         # routing.py:118
    assert np.array_equal(kin, g["LddChan"][d][chan])
    assert np.array_equal(L.lddrepair(np.where(chan, L.PIT, ldd), dm), g["LddToChan"][d])             # routing.py:125
    assert np.array_equal(L.pit(ldd), g["pit"][d])                                                    # routing.py:127
    at_out = (L.pit(ldd) != 0).astype(np.float64)
    assert np.array_equal(L.downstream(ldd, dm, at_out), g["downstream_AtOutflow"][d])                # routing.py:141
    assert np.array_equal(L.uniqueid(g["AtLastPoint"][d]), g["OutflowPoints"][d])                     # routing.py:168
    assert np.array_equal(L.catchment(ldd, dm, g["OutflowPoints"][d]), g["Catchments"][d])            # routing.py:170
    assert np.array_equal(L.catchment(ldd, dm, g["pit"][d]), g["catchment_of_pits"][d])
    assert np.array_equal(L.catchment(ldd, dm, g["points_nested"][d]), g["catchment_nested"][d])
    assert np.array_equal(L.subcatchment(ldd, dm, g["points_nested"][d]), g["subcatchment_nested"][d])
    # the kinematic (channel) LDD on the channel pixels: downstruct, one-hop upstream sum, structures cut
    ids = np.arange(N, dtype=np.float64)[d][chan]
    assert np.array_equal(L.downstream(kin, kin_mask, ids), g["downstruct_ids"][d][chan])             # routing.py:159-162
    w = g["w"][d]
    down_kin = L.downstream_index(kin, kin_mask)
    ups = np.bincount(np.where(down_kin >= 0, down_kin, kin.size), weights=w[chan], minlength=kin.size + 1)[:kin.size]
    assert np.array_equal(ups, g["upstream_w"][d][chan])                                              # routing.py:387
    cut, is_ups = L.cut_at_structures(kin, kin_mask, g["is_structure"][d][chan])                      # structures.py:51-59
    assert np.array_equal(is_ups, g["IsUpsOfStructure"][d][chan])
    assert np.array_equal(cut, g["LddKinematic_cut"][d][chan])
