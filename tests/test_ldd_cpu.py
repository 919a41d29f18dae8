"""LDD operations restated from PCRaster's documented semantics (lisflood_amd/ldd.py), checked against brute
force walks on small catchments.  CPU only (host-side, init-time code)."""
import numpy as np

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
    # interior points override what lies downstream of them
    pts = np.zeros(codes.size, np.int64); pts[[50, 300, 400]] = [7, 8, 9]
    lab2 = L.catchment(codes, mask, pts)
    for p in range(codes.size):
        hit = [pts[q] for q in walk_down(down, p) if pts[q]]
        assert lab2[p] == (hit[0] if hit else 0)


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
