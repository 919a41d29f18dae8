"""PCRaster-free LDD operations on compressed vectors.

The reference calls PCRaster 4.3.3 (un-vendored C++) for these at initialisation: `lddmask`, `lddrepair`, `pit`,
`downstream`, `upstream`, `accuflux`, `catchment`, `uniqueid` (routing.py:90-171, 387; structures.py:51-59).
They are restated from PCRaster's documented semantics.  What pins them is the reference's own DATA: `accuflux`
reproduces `ec_upArea.nc` where the test mask is upstream-closed, `catchment` reproduces the PCRaster-made catchment
masks of the use case (`mask.map`, `subcatchment_mask.map`), the one-hop `upstream` the np.bincount fixture; the rest
is checked against brute-force walks.

Two forms of every operation:
  * `<name>_device` -- the product path (routing.initial uses it): csrc/lf_ldd.hip through the C ABI (`lf_lddrepair_raster`,
    `lf_lddmask_raster`, `lf_downstream`, `lf_catchments`, `lf_catchment_totals`); catchment labels by pointer jumping;
  * `<name>` -- a host helper over the host-side graph (scenario generators, tests; needs no GPU).

All functions take / return 1-D vectors over the land pixels of `land_mask` (row-major, add1.py:268-305).
"""
import ctypes as C
import numpy as np

from .kinematic_wave_parallel import Graph

PIT = 5
_ROW = {2: 1, 3: 1, 6: 0, 9: -1, 8: -1, 7: -1, 4: 0, 1: 1}
_COL = {2: 0, 3: 1, 6: 1, 9: 1, 8: 0, 7: -1, 4: -1, 1: -1}


def _raster(codes, land_mask, fill=0):
    r = np.full(land_mask.shape, fill, dtype=np.asarray(codes).dtype)
    r[land_mask] = codes
    return r


def downstream_index(codes, land_mask):
    """pixel id of the downstream neighbour, -1 if the cell is a pit or drains off the map / out of the mask."""
    return Graph(np.asarray(codes, np.float64), land_mask).lookups()[0].astype(np.int64)


def lddmask(codes, land_mask, keep):
    """PCRaster lddmask(ldd, mask): the LDD cut to `keep` (bool per land pixel).  Returns (codes, land_mask) of the
    sub-domain; a cell whose downstream neighbour is outside `keep` becomes a pit (routing.py:90, 118)."""
    keep = np.asarray(keep, bool)
    down = downstream_index(codes, land_mask)
    out = np.asarray(codes).copy()
    known = np.isin(out, (1, 2, 3, 4, 5, 6, 7, 8, 9))   # any other value is a missing value of the ldd map
    # drains off the map / into a missing value (incl. a land pixel with an unknown code) / out of `keep`
    lost = (down < 0) | ~keep[np.maximum(down, 0)] | ~known[np.maximum(down, 0)]
    out[lost & np.isin(out, (1, 2, 3, 4, 6, 7, 8, 9))] = PIT
    sub = land_mask.copy()
    sub[land_mask] = keep
    return out[keep], sub


def _down_no_graph(codes, land_mask):
    """downstream_index without building a graph (a graph refuses a cyclic LDD): -1 for pits, unknown codes and links
    that leave the map or the mask"""
    land_mask = np.asarray(land_mask, bool)
    c = np.asarray(codes)
    H, W = land_mask.shape
    ids = np.full((H, W), -1, np.int64)
    ids[land_mask] = np.arange(c.size)
    rr, cc = np.nonzero(land_mask)
    down = np.full(c.size, -1, np.int64)
    for code in _ROW:
        sel = c == code
        r2, c2 = rr[sel] + _ROW[code], cc[sel] + _COL[code]
        ok = (r2 >= 0) & (r2 < H) & (c2 >= 0) & (c2 < W)
        t = np.full(int(sel.sum()), -1, np.int64)
        t[ok] = ids[r2[ok], c2[ok]]
        down[sel] = t
    return down


def break_cycles(codes, land_mask):
    """-> (codes with every cycle of the LDD broken, number of cycles).  PCRaster's lddrepair makes a cyclic ldd sound by
    turning a cell of the cycle into a pit; its manual does not say which one, so this is a CHOICE, not a restatement:
    the cell of the cycle that comes first in row-major order (lowest pixel id) becomes the pit.  Every other cell keeps
    its direction, so the trees hanging on the cycle and the rest of the cycle drain to the new pit.  Host, init-time:
    pointer doubling finds the cells that lie on cycles (the image of the N-fold downstream map minus the pits)."""
    c = np.asarray(codes).copy()
    down = _down_no_graph(c, land_mask)
    n = c.size
    if n == 0:
        return c, 0
    p = np.where(down >= 0, down, np.arange(n))            # pits point at themselves
    hop = p.copy()
    reach = 1
    while reach < n:                                       # hop = p^(2^k), 2^k >= n: every cell lands on a cycle or a pit
        hop = hop[hop]
        reach *= 2
    on_cycle = np.zeros(n, bool)
    on_cycle[np.unique(hop)] = True
    on_cycle &= down >= 0                                  # (a pit is a fixed point, not a cycle)
    cyc = np.nonzero(on_cycle)[0]
    if cyc.size == 0:
        return c, 0
    lowest = np.arange(n)
    while True:                                            # the lowest id along each cycle, passed round until it settles
        nxt = np.minimum(lowest[cyc], lowest[p[cyc]])
        if np.array_equal(nxt, lowest[cyc]):
            break
        lowest[cyc] = nxt
    heads = cyc[lowest[cyc] == cyc]
    c[heads] = PIT
    return c, int(heads.size)


def lddrepair(codes, land_mask, break_cycles_too=False):
    """PCRaster lddrepair: cells draining to a missing value or off the map become pits (routing.py:125).  A land pixel
    whose code is not a keypad code is a missing value of the ldd map: cells draining into it become pits, and -- a
    compressed vector has no missing values -- so does the pixel itself (the device form does the same).  PCRaster's
    lddrepair also breaks cycles.  By default a cyclic LDD is NOT repaired here: building the graph on it raises
    LF_E_CYCLE (the reference's own graph builder would hang on it, kinematic_wave_parallel.py:99, were it not for
    PCRaster's repair) -- which cell of a cycle PCRaster turns into a pit is not documented, and a silent choice would
    move an outlet.  break_cycles_too=True makes that choice explicit (break_cycles: the first cell in row-major order)."""
    c = np.asarray(codes).copy()
    if break_cycles_too:
        c, _ = break_cycles(c, land_mask)
    down = _down_no_graph(c, land_mask) if break_cycles_too else downstream_index(c, land_mask)
    valid = np.isin(c, (1, 2, 3, 4, 6, 7, 8, 9))
    known = valid | (c == PIT)
    c[(down < 0) | ~valid | ~known[np.maximum(down, 0)]] = PIT
    return c


def pit(codes):
    """PCRaster pit(ldd): unique id 1..n (row-major) at pit cells, 0 elsewhere (routing.py:127)."""
    c = np.asarray(codes)
    out = np.zeros(c.shape, np.int64)
    is_pit = c == PIT
    out[is_pit] = np.arange(1, int(is_pit.sum()) + 1)
    return out


def uniqueid(flags):
    """PCRaster uniqueid: 1..n (row-major) where `flags` is true, 0 elsewhere (routing.py:166)."""
    f = np.asarray(flags, bool)
    out = np.zeros(f.shape, np.int64)
    out[f] = np.arange(1, int(f.sum()) + 1)
    return out


def downstream(codes, land_mask, x):
    """PCRaster downstream(ldd, x): value of x at the downstream neighbour; pits keep their own value
    (routing.py:141, 162; structures.py:51)."""
    x = np.asarray(x)
    down = downstream_index(codes, land_mask)
    return np.where(down >= 0, x[np.maximum(down, 0)], x)


def downstruct(codes, land_mask):
    """routing.py:159-164: id of the downstream pixel, pits get N (the "drop" bin of np.bincount)."""
    down = downstream_index(codes, land_mask)
    N = down.size
    out = np.where(down >= 0, down, N).astype(np.int32)
    out[np.asarray(codes) == PIT] = N
    return out


def subcatchment(codes, land_mask, points):
    """PCRaster subcatchment(ldd, points): every cell gets the id of the FIRST non-zero point met going downstream
    (a point cell belongs to its own sub-catchment); 0 if none."""
    g = Graph(np.asarray(codes, np.float64), land_mask)
    down = g.lookups()[0].astype(np.int64)
    po, ss = g.orders()
    pts = np.asarray(points).astype(np.int64)
    lab = np.zeros(pts.shape, np.int64)
    for k in range(ss.shape[0] - 1, -1, -1):            # outlets first, then upstream level by level
        cells = po[ss[k, 0]:ss[k, 1]]
        d = down[cells]
        inherited = np.where(d >= 0, lab[np.maximum(d, 0)], 0)
        lab[cells] = np.where(pts[cells] != 0, pts[cells], inherited)
    return lab


def _enclosing_points(points, first_hit, down):
    """the points that have no other point strictly downstream of them: the only ones PCRaster's catchment() labels with
    (`first_hit` = subcatchment labels of `points`)"""
    pts = np.asarray(points).astype(np.int64)
    below = np.where(down >= 0, first_hit[np.maximum(down, 0)], 0)
    return np.where(below == 0, pts, 0)


def catchment(codes, land_mask, points):
    """PCRaster catchment(ldd, points): every cell upstream of a non-zero point (the point included) gets the point's
    id, 0 if no point lies downstream; sub-catchments are NOT identified -- where one point lies in the catchment of
    another, the enclosing (most downstream) point wins (routing.py:168-171, whose outflow points are never nested:
    each sits next to a pit).  = subcatchment() over the points that have no point downstream of them."""
    down = downstream_index(codes, land_mask)
    first = subcatchment(codes, land_mask, points)
    return subcatchment(codes, land_mask, _enclosing_points(points, first, down))


def cut_at_structures(codes, land_mask, is_structure):
    """structures.initial (structures.py:44-61): the cells just upstream of a lake / reservoir become pits of the
    kinematic LDD (their outflow reaches the structure through its own inflow term instead).  Returns
    (cut codes, IsUpsOfStructureKinematicC)."""
    down = downstream_index(codes, land_mask)
    st = np.asarray(is_structure, bool)
    ups = np.where(down >= 0, st[np.maximum(down, 0)], st)   # downstream(LddKinematic, IsStructureKinematic): a pit reads itself
    out = np.asarray(codes).copy()
    out[ups] = PIT
    return lddrepair(out, land_mask), ups


def upstream_raster(ldd_raster, w_raster, device=0):
    """PCRaster upstream(ldd, w) on whole H x W rasters, on the device (LDS-staged 3 x 3 neighbourhoods):
    out[cell] = sum of w over the neighbours draining into the cell, ascending source index.  Cells whose code
    is 0 send nothing."""
    import ctypes as C
    from ._lib import DeviceArray, check, lib
    ldd_raster = np.ascontiguousarray(ldd_raster, dtype=np.uint8)
    w_raster = np.ascontiguousarray(w_raster, dtype=np.float64)
    H, W = ldd_raster.shape
    d_l, d_w = DeviceArray.from_host(ldd_raster, device), DeviceArray.from_host(w_raster, device)
    d_o = DeviceArray((H, W), np.float64, device)
    check(lib().lf_upstream_sum_raster_device(C.c_int(device), d_l.ptr, d_w.ptr, d_o.ptr, C.c_int(H), C.c_int(W)))
    out = d_o.download()
    for d in (d_l, d_w, d_o):
        d.free()
    return out


# ---------------------------------------------------------------------------------------------------------------------
# device forms (csrc/lf_ldd.hip)
# ---------------------------------------------------------------------------------------------------------------------
def _codes_raster(codes, land_mask):
    """compressed codes -> H x W uint8 raster, 0 = missing value (outside the mask or not a keypad code)"""
    c = np.asarray(codes, np.float64)
    ok = (c >= 1) & (c <= 9) & (c == np.floor(c))
    r = np.zeros(np.asarray(land_mask).shape, np.uint8)
    r[np.asarray(land_mask, bool)] = np.where(ok, c, 0).astype(np.uint8)
    return r


def _raster_op(ldd_raster, keep_raster, device):
    from ._lib import check, lib, ptr
    ldd_raster = np.ascontiguousarray(ldd_raster, np.uint8)
    H, W = ldd_raster.shape
    out = np.empty((H, W), np.uint8)
    keep = None if keep_raster is None else np.ascontiguousarray(keep_raster, np.uint8)
    check(lib().lf_ldd_raster_host(C.c_int(device), ptr(ldd_raster), ptr(keep), ptr(out), C.c_int(H), C.c_int(W)))
    return out


def lddrepair_device(codes, land_mask, device=0):
    """lddrepair on the device; same contract as lddrepair (cells with an unknown code become pits too)"""
    land_mask = np.asarray(land_mask, bool)
    c = np.asarray(codes, np.float64)
    r = _codes_raster(c, land_mask)
    valid = r[land_mask] > 0
    out = _raster_op(r, None, device)[land_mask].astype(np.float64)
    return np.where(valid, out, PIT)


def lddmask_device(codes, land_mask, keep, device=0):
    """lddmask on the device -> (codes of the kept pixels, their land mask)"""
    land_mask = np.asarray(land_mask, bool)
    keep = np.asarray(keep, bool)
    k2 = np.zeros(land_mask.shape, np.uint8)
    k2[land_mask] = keep
    out = _raster_op(_codes_raster(codes, land_mask), k2, device)
    sub = land_mask.copy()
    sub[land_mask] = keep
    return out[sub].astype(np.float64), sub


class LddDevice:
    """The downstream / catchment operations of one LDD on the device: a router on the LDD's graph with unit parameters
    (kinematicWave also carries upstream_sum and accuflux)."""

    def __init__(self, codes, land_mask, device=0):
        from .kinematic_wave_parallel import kinematicWave
        self.N = int(np.asarray(land_mask, bool).sum())
        self.kw = kinematicWave(np.asarray(codes, np.float64), land_mask, np.ones(self.N), 0.6, 1.0, 1.0, device=device)

    def downstream(self, x):
        from ._lib import check, f64, lib, ptr
        x = f64(np.broadcast_to(x, (self.N,)))
        out = np.empty(self.N)
        if self.N:
            check(lib().lf_downstream_host(self.kw._h, ptr(x), ptr(out)))
        return out

    def subcatchment(self, points):
        """PCRaster subcatchment: the first non-zero point met going downstream (pointer jumping, lf_catchments)"""
        from ._lib import check, lib, ptr
        pts = np.ascontiguousarray(np.broadcast_to(points, (self.N,)), dtype=np.int64)
        out = np.empty(self.N, np.int64)
        if self.N:
            check(lib().lf_catchments(self.kw._h, ptr(pts), ptr(out)))
        return out

    def catchment(self, points):
        """PCRaster catchment: the most downstream point wins (see catchment() above) -- two labelling passes and one
        downstream() on the device"""
        pts = np.ascontiguousarray(np.broadcast_to(points, (self.N,)), dtype=np.int64)
        first = self.subcatchment(pts)
        if self.N == 0:
            return first
        at_pit = self.kw.graph.lookups()[0] < 0
        below = np.where(at_pit, 0, self.downstream(first.astype(np.float64)).astype(np.int64))
        return self.subcatchment(np.where(below == 0, pts, 0))

    def catchment_totals(self, w):
        """np.take(np.bincount(Catchments, weights=w), Catchments) for Catchments = catchment(ldd, pit(ldd))"""
        return self.catchment_totals_multi([w])[0]

    def catchment_totals_multi(self, ws):
        """the same for several weight vectors at once: one upload, accuflux sweeps of up to four vectors, one download"""
        from ._lib import check, f64, lib, ptr
        ws = [np.broadcast_to(np.asarray(w, np.float64), (self.N,)) for w in ws]
        out = []
        for i in range(0, len(ws), 4):
            chunk = f64(np.stack(ws[i:i + 4])) if self.N else np.zeros((len(ws[i:i + 4]), 0))
            res = np.empty_like(chunk)
            if self.N:
                check(lib().lf_catchment_totals_multi_host(self.kw._h, C.c_int(chunk.shape[0]), ptr(chunk), ptr(res)))
            out.extend(res[k] for k in range(chunk.shape[0]))
        return out

    def upstream(self, w):
        return self.kw.upstream_sum(w)

    def accuflux(self, x):
        return self.kw.accuflux(x)

    def water_use_sum(self, withdrawal_m3, inv_dt_sec):
        """self.var.WUseSumM3 of waterabstraction.py:533: accuflux(Ldd, withdrawal_CH_actual_M3 * InvDtSec), the channel
        withdrawal of a model step accumulated downstream as a flow [m3/s] -- one device sweep on the LDD's block plan"""
        return self.kw.accuflux(np.asarray(withdrawal_m3, np.float64) * inv_dt_sec)

    def close(self):
        self.kw.close()
