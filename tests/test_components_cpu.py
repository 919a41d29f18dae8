"""Component layout (lf_graph_build_components): host-side invariants of the plan -- no GPU needed.

Every cell's upstream positions as the plan presents them (contiguous same-tier range, or the index list of a tier >= 1
cell) must be exactly its upstream pixels in ascending pixel id (the reference's summation order,
kinematic_wave_parallel_tools.py:57-58, 127-128), and every one of them must be finished before the cell is swept: an
earlier tier, or the previous local level of the same bin."""
import numpy as np
import pytest

from conftest import golden
from lisflood_amd import synthetic as syn
from lisflood_amd.kinematic_wave_parallel import Graph


def check_plan(g):
    N = g.num_pixels
    perm, ups_ptr, _ = g.layout()
    t = g.component_tables()
    assert np.array_equal(np.sort(perm), np.arange(N))
    down, ups, nups = g.lookups()
    pos = np.empty(N, np.int64); pos[perm] = np.arange(N)
    tbs, off, nl, lvl = t["tier_bin_start"], t["bin_lvl_off"], t["bin_nl"], t["lvl"]
    tier_of = np.empty(N, np.int64); bin_of = np.empty(N, np.int64); level_of = np.empty(N, np.int64)
    ups_end = np.empty(N, np.int64)
    at = 0
    for tier in range(tbs.size - 1):
        for b in range(tbs[tier], tbs[tier + 1]):
            l = lvl[off[b]:off[b] + nl[b] + 1]
            assert l[0] == at and (np.diff(l) > 0).all()          # bins tile the positions, no empty level
            at = l[-1]
            tier_of[l[0]:l[-1]] = tier; bin_of[l[0]:l[-1]] = b
            for k in range(nl[b]):
                level_of[l[k]:l[k + 1]] = k
            ups_end[l[0]:l[-1]] = np.minimum(ups_ptr[l[0] + 1:l[-1] + 1], l[nl[b] - 1])
    assert at == N
    assert t["trunk_first"] == (np.nonzero(tier_of > 0)[0][0] if (tier_of > 0).any() else N)
    for p in range(N):
        want = ups[perm[p], :nups[perm[p]]]                          # ascending pixel id
        if p < t["trunk_first"]:
            got = perm[ups_ptr[p]:ups_end[p]]
            u = np.arange(ups_ptr[p], ups_end[p])
        else:
            q = p - t["trunk_first"]
            u = t["t_idx"][t["t_ptr"][q]:t["t_ptr"][q + 1]].astype(np.int64)
            got = perm[u]
        assert np.array_equal(got, want), p
        for e in u:
            assert tier_of[e] < tier_of[p] or (bin_of[e] == bin_of[p] and level_of[e] == level_of[p] - 1), (p, e)
    return tier_of


@pytest.mark.parametrize("family,shape,cap", [("deep", (60, 40), 64), ("shallow", (50, 70), 32), ("deep", (120, 9), 16),
                                              ("saddle", (40, 40), 100000)])
def test_component_plan_synthetic(family, shape, cap):
    H, W = shape
    g = Graph(ldd_raster=syn.make_ldd(family, H, W, 5), components=(cap, max(cap, 48)))
    tiers = check_plan(g)
    st = g.component_stats()
    assert st["tiers"] == tiers.max() + 1 and st["cap"] == cap
    if cap >= H * W:
        assert st["tiers"] == 1 and st["trunk_cells"] == 0


def test_component_plan_etrs89_and_masked():
    z = golden("graph_etrs89")
    g = Graph(z["codes"], z["mask"], components=(256, 256))
    tiers = check_plan(g)
    assert tiers.max() >= 1                                         # the main stems drain more than 256 cells
    z = golden("graph_syn48_masked")
    check_plan(Graph(z["codes"], z["mask"], components=(40, 100)))


def test_component_layout_keeps_the_reference_attributes():
    """lookups and routing orders (the reference's attributes) do not depend on the layout"""
    z = golden("graph_etrs89")
    a, b = Graph(z["codes"], z["mask"]), Graph(z["codes"], z["mask"], components=True)
    for x, y in zip(a.lookups() + a.orders(), b.lookups() + b.orders()):
        assert np.array_equal(x, y)
    assert np.array_equal(b.orders()[0], z["pixels_ordered"])
