"""Catchment partition: trees of the LDD rooted at its outlets are mutually independent (the reference relies on it
for sub-catchment runs, tests/test_subcatchments.py:110-112; `Catchments` labels, routing.py:168-171), so giving
every GPU a set of WHOLE catchments needs no exchange at all during routing -- each rank runs the single-GPU engine
on its own compressed sub-domain.  The row-block partition (lisflood_amd.dist) is the answer when one catchment is
bigger than a GPU's share; this one is the answer when there are many.
"""
import numpy as np

from .kinematic_wave_parallel import Graph


def catchment_roots(graph):
    """[N] int32, pixel order: pixel id of the outlet every pixel drains to (host, O(N), level by level from the
    outlets upstream over the engine layout)."""
    perm, ups_ptr, level_start = graph.layout()
    N = perm.size
    counts = np.diff(ups_ptr)
    # children ranges tile [0, first outlet position) in parent order -> parent position of every non-outlet position
    parent = np.repeat(np.arange(N, dtype=np.int32), counts)
    root_pos = np.arange(N, dtype=np.int32)                      # outlets (and isolated pixels) are their own root
    NL = level_start.size - 1
    for k in range(NL - 2, -1, -1):                              # level NL-1 holds the outlets
        a, b = int(level_start[k]), int(level_start[k + 1])
        root_pos[a:b] = root_pos[parent[a:b]]
    out = np.empty(N, np.int32)
    out[perm] = perm[root_pos]
    return out


def catchment_roots_of_raster(ldd_raster, land_mask=None):
    """catchment_roots without building a graph: the outlet of every land pixel (compressed pixel ids, as catchment_roots
    returns them) straight from the uint8 LDD raster by pointer jumping -- every pixel's pointer doubles its reach per pass,
    log2(longest flow path) passes over one int32 vector (the device form is lf_ldd.hip's k_jump_step).  Peak host memory
    ~3 int32 vectors of the raster's size: what a rank of a many-rank job can afford where a whole-raster Graph (20 GB at
    10000^2) is not.  Codes follow kinematic_wave_parallel.py:49-51; a pixel pointing off the raster, at a missing value
    or at non-land is a pit, as lddrepair leaves it (routing.py:125)."""
    from .synthetic import FLOW_CODE, IX_ADDS
    ldd = np.asarray(ldd_raster)
    H, W = ldd.shape
    land = (ldd != 0) if land_mask is None else (np.asarray(land_mask, bool) & (ldd != 0))
    if land_mask is not None and (np.asarray(land_mask, bool) & (ldd == 0)).any():
        raise ValueError("LDD code 0 on a land pixel")
    n = H * W
    parent = np.arange(n, dtype=np.int32).reshape(H, W)           # raster index of the downstream pixel (pits: itself)
    rows = np.arange(H, dtype=np.int32)[:, None]
    cols = np.arange(W, dtype=np.int32)[None, :]
    for k, (dr, dc) in enumerate(IX_ADDS):
        sel = ldd == FLOW_CODE[k]
        r2, c2 = rows + dr, cols + dc
        ok = sel & (r2 >= 0) & (r2 < H) & (c2 >= 0) & (c2 < W)
        tgt = (np.clip(r2, 0, H - 1) * W + np.clip(c2, 0, W - 1)).astype(np.int32)
        ok &= land.reshape(-1)[tgt]                               # a link into non-land is cut
        parent[ok] = tgt[ok]
        del sel, ok, tgt
    parent = parent.reshape(-1)
    parent0 = parent.copy()                                       # one hop: a root is a fixed point of it
    # parent <- parent o parent until nothing moves: a path of length L settles in ceil(log2 L) passes, so more than
    # log2(n) + 1 passes means a cycle (odd cycles never settle, even ones would split into several false roots)
    for _ in range(int(np.ceil(np.log2(max(n, 2)))) + 2):
        nxt = parent[parent]
        if np.array_equal(nxt, parent):
            break
        parent = nxt
    if not np.array_equal(parent[parent], parent) or not np.array_equal(parent0[parent], parent):
        raise ValueError("the LDD has a cycle (the Graph path reports LF_E_CYCLE for the same raster)")
    landf = land.reshape(-1)
    comp = np.cumsum(landf, dtype=np.int64).astype(np.int32) - 1  # raster index -> compressed pixel id
    return comp[parent[landf]]


def split_catchments(roots, nparts):
    """rank of every pixel: catchments ordered by outlet pixel id, cut where the running cell count passes k*N/nparts
    (whole catchments only; the imbalance is bounded by the largest catchment)."""
    N = roots.size
    ids, inv, sizes = np.unique(roots, return_inverse=True, return_counts=True)
    cum = np.cumsum(sizes)
    part_of_catchment = np.minimum((cum - sizes) * nparts // max(N, 1), nparts - 1).astype(np.int32)
    return part_of_catchment[inv], sizes


def sub_domain(codes, land_mask, pixel_rank, rank):
    """(compressed LDD codes, land mask, global pixel ids) of the pixels of `rank`: a self-contained domain, every
    downstream link stays inside it."""
    land_mask = np.asarray(land_mask, bool)
    sel = np.asarray(pixel_rank) == rank
    sub_mask = np.zeros(land_mask.shape, bool)
    sub_mask[land_mask] = sel
    return np.asarray(codes)[sel], sub_mask, np.nonzero(sel)[0]


def catchment_partition(codes, land_mask, nparts, ldd_raster=None):
    """-> list of (codes_r, mask_r, pixel_ids_r) for r in range(nparts), plus the per-rank cell counts"""
    g = Graph(ldd_raster=ldd_raster, land_mask=land_mask) if ldd_raster is not None else Graph(codes, land_mask)
    roots = catchment_roots(g)
    g.close()
    rank, _ = split_catchments(roots, nparts)
    if codes is None:
        lm = np.ones(ldd_raster.shape, bool) if land_mask is None else np.asarray(land_mask, bool)
        codes, land_mask = np.asarray(ldd_raster)[lm].astype(np.float64), lm
    parts = [sub_domain(codes, land_mask, rank, r) for r in range(nparts)]
    return parts, np.bincount(rank, minlength=nparts)
