"""PCRaster's LDD operations, restated naively from the PCRaster manual -- FIXTURE GENERATOR ONLY.

The reference calls PCRaster 4.3.3 (C++, not vendored under /root/reference, not installed here) for the map algebra
of its initialisation: routing.py:90-171, 387; structures.py:51-59; lakes.py:87-90; reservoir.py:90.  This file is the
stand-in the golden generator drives the reference's modules through (make_golden.py: PcrEmu) and the source of
tests/golden/ldd_ops.npz.  It is deliberately INDEPENDENT of the code under test: it imports numpy only -- nothing from
lisflood_amd, nothing from oracle/ -- and works cell by cell on the 2-D raster, following every cell's downstream path
with plain Python loops.  It is meant to be read next to the manual, not to be fast (57 x 80 cells).

Semantics restated (PCRaster manual, "Operators on local drain direction maps"):
  ldd codes    numeric keypad: 7 8 9 = row-1, 4 5 6 = same row, 1 2 3 = row+1; 1 4 7 = col-1, 3 6 9 = col+1; 5 = pit.
               Anything else is a missing value (MV); here MV is 0 on ldd / nominal / boolean maps, NaN on scalar maps.
  lddmask      ldd where the mask is TRUE, MV elsewhere; the result is made sound: a cell whose downstream neighbour is
               off the map or MV becomes a pit.
  lddrepair    a cell whose downstream neighbour is off the map or MV becomes a pit (cycles: not restated -- the manual
               does not say which cell of a cycle becomes the pit; the generator asserts its inputs have none).
  pit          1, 2, 3 ... at the pits in row-major order, 0 at every other defined cell.
  uniqueid     1, 2, 3 ... at the TRUE cells in row-major order, 0 at FALSE cells.
  downstream   the value of the downstream neighbour; a pit keeps its own value.
  upstream     the sum of the values of the cells whose downstream neighbour is this cell (0 if none).
  accuflux     the cell's own material plus the material of every cell upstream of it.
  catchment    every cell upstream of (and including) a non-zero point gets the point's value; where the catchment of
               one point lies inside the catchment of another, the ENCLOSING (most downstream) point wins; 0 if no
               point lies downstream.
  subcatchment as catchment, but sub-catchments ARE identified: the first non-zero point met going downstream wins.
"""
import numpy as np

D_ROW = {7: -1, 8: -1, 9: -1, 4: 0, 6: 0, 1: 1, 2: 1, 3: 1}
D_COL = {7: -1, 4: -1, 1: -1, 8: 0, 2: 0, 9: 1, 6: 1, 3: 1}


def _code(v):
    """keypad code of a raster value, 0 (MV) for anything that is not one of 1..9"""
    try:
        f = float(v)
    except (TypeError, ValueError):
        return 0
    if f != f or f != int(f) or not 1 <= int(f) <= 9:
        return 0
    return int(f)


class NaivePcr:
    """the operations over one land mask; every argument and result is a 1-D vector over the mask's TRUE cells in
    row-major order (the reference's compressed form, add1.py:268-305)"""

    def __init__(self, land_mask):
        self.mask = np.asarray(land_mask, bool)
        self.H, self.W = self.mask.shape
        self.cells = [(int(r), int(c)) for r, c in zip(*np.nonzero(self.mask))]      # row-major
        self.N = len(self.cells)
        self.index = {rc: i for i, rc in enumerate(self.cells)}

    # -- raster <-> vector ------------------------------------------------------------------------------------------
    def _ldd_raster(self, ldd):
        ldd = np.broadcast_to(np.asarray(ldd), (self.N,))
        ras = [[0] * self.W for _ in range(self.H)]
        for (r, c), v in zip(self.cells, ldd):
            ras[r][c] = _code(v)
        return ras

    def _next(self, ras, r, c):
        """(row, col) of the downstream neighbour of a non-pit cell, None if it is off the map or MV"""
        k = ras[r][c]
        r2, c2 = r + D_ROW[k], c + D_COL[k]
        if not (0 <= r2 < self.H and 0 <= c2 < self.W) or ras[r2][c2] == 0:
            return None
        return r2, c2

    def _sound(self, ras):
        """pits where the downstream neighbour is off the map or MV; -> vector (MV = 0)"""
        out = np.zeros(self.N)
        for i, (r, c) in enumerate(self.cells):
            k = ras[r][c]
            if k == 0:
                continue
            out[i] = 5 if (k == 5 or self._next(ras, r, c) is None) else k
        return out

    def _down_ids(self, ldd):
        """per land cell: vector index of the downstream neighbour; -1 for pits, MV cells and links that leave the map,
        reach an MV cell or reach a cell outside the land mask"""
        ras = self._ldd_raster(ldd)
        down = [-1] * self.N
        for i, (r, c) in enumerate(self.cells):
            if ras[r][c] in (0, 5):
                continue
            nxt = self._next(ras, r, c)
            if nxt is not None:
                down[i] = self.index.get(nxt, -1)
        return down, ras

    def _assert_no_cycle(self, down):
        for i in range(self.N):
            j, hops = i, 0
            while down[j] >= 0:
                j = down[j]
                hops += 1
                assert hops <= self.N, "cyclic ldd: outside what this stand-in restates"

    # -- the operations ---------------------------------------------------------------------------------------------
    def lddmask(self, ldd, keep):
        keep = np.broadcast_to(np.asarray(keep), (self.N,))
        ras = self._ldd_raster(ldd)
        for (r, c), k in zip(self.cells, keep):
            if not (k == k and k != 0):                      # FALSE or MV
                ras[r][c] = 0
        return self._sound(ras)

    def lddrepair(self, ldd):
        out = self._sound(self._ldd_raster(ldd))
        self._assert_no_cycle(self._down_ids(out)[0])
        return out

    def pit(self, ldd):
        ras = self._ldd_raster(ldd)
        out = np.zeros(self.N, np.int64)
        n = 0
        for i, (r, c) in enumerate(self.cells):
            if ras[r][c] == 5:
                n += 1
                out[i] = n
        return out

    def uniqueid(self, flags):
        flags = np.broadcast_to(np.asarray(flags), (self.N,))
        out = np.zeros(self.N, np.int64)
        n = 0
        for i in range(self.N):
            if flags[i] == flags[i] and flags[i] != 0:
                n += 1
                out[i] = n
        return out

    def downstream(self, ldd, x):
        x = np.array(np.broadcast_to(np.asarray(x), (self.N,)))
        down, _ = self._down_ids(ldd)
        out = x.copy()
        for i in range(self.N):
            if down[i] >= 0:
                out[i] = x[down[i]]
        return out

    def upstream(self, ldd, x):
        x = np.broadcast_to(np.asarray(x, float), (self.N,))
        down, _ = self._down_ids(ldd)
        out = np.zeros(self.N)
        for i in range(self.N):                              # ascending source index = np.bincount's order
            if down[i] >= 0:
                out[down[i]] += x[i]
        return out

    def accuflux(self, ldd, x):
        x = np.broadcast_to(np.asarray(x, float), (self.N,))
        down, _ = self._down_ids(ldd)
        self._assert_no_cycle(down)
        out = np.zeros(self.N)
        for i in range(self.N):                              # every cell hands its material to its whole downstream path
            j = i
            while j >= 0:
                out[j] += x[i]
                j = down[j]
        return out

    def catchment(self, ldd, points):
        pts = np.broadcast_to(np.asarray(points), (self.N,))
        down, _ = self._down_ids(ldd)
        self._assert_no_cycle(down)
        out = np.zeros(self.N, np.int64)
        for i in range(self.N):
            j, lab = i, 0
            while j >= 0:
                if pts[j] == pts[j] and pts[j] != 0:
                    lab = int(pts[j])                        # keep walking: an enclosing catchment wins
                j = down[j]
            out[i] = lab
        return out

    def subcatchment(self, ldd, points):
        pts = np.broadcast_to(np.asarray(points), (self.N,))
        down, _ = self._down_ids(ldd)
        self._assert_no_cycle(down)
        out = np.zeros(self.N, np.int64)
        for i in range(self.N):
            j = i
            while j >= 0:
                if pts[j] == pts[j] and pts[j] != 0:
                    out[i] = int(pts[j])                     # stop at the first point
                    break
                j = down[j]
        return out
