"""Initialisation of the structures of the routing loop, in the reference's module shape:

    lakes.initial       lakes.py:48-197        site list, table look-ups, Modified-Puls start values
    reservoir.initial   reservoir.py:52-171    site list, table look-ups, outflow rule parameters, initial fill
    structures.initial  structures.py:44-61    pits just upstream of every structure (the cut kinematic LDD)

The sub-step work of these modules (`dynamic_inloop`) lives in the routing loop on the device
(`routing.attach_structures`, csrc/lf_modules.hip: lf_inloop_structures and the fused wavefront).  The reference reads
its inputs through `loadmap(binding)` and PCRaster's `lookupscalar(table, id map)`; here `maps` is a dict keyed by the
same binding names (compressed vectors over the land pixels, or scalars) and `tables` a dict of two-column arrays
(id, value) keyed by the binding of the table (TabLakeArea, ...).  Everything is init-time work on a handful of sites:
host numpy, as in the reference -- except the LDD operations, which run on the device (lisflood_amd.ldd)."""
import warnings

import numpy as np

from . import ldd as L
from .hydro_module import HydroModule

try:
    from lisflood.global_modules.errors import LisfloodWarning  # type: ignore
except Exception:
    class LisfloodWarning(Warning):
        pass


def lookupscalar(table, ids):
    """PCRaster lookupscalar(table, nominal map) for the two-column tables of the reference (key, value): NaN where no
    row matches (missing value)."""
    ids = np.asarray(ids)
    out = np.full(ids.shape, np.nan)
    for key, val in np.asarray(table, dtype=np.float64).reshape(-1, 2):
        out[ids == key] = val
    return out


class _SiteModule(HydroModule):
    def __init__(self, variable, options=None, maps=None, tables=None, device=0):
        self.var, self.options, self.device = variable, options if options is not None else {}, device
        self.maps, self.tables = dict(maps or {}), dict(tables or {})

    def _map(self, name, N=None):
        x = self.maps[name]
        x = np.array(x, dtype=np.float64, copy=True) if isinstance(x, np.ndarray) else float(x)
        return x if N is None else np.broadcast_to(x, (N,)).copy()


class lakes(_SiteModule):
    input_files_keys = {'simulateLakes': ['LakeSites', 'TabLakeArea', 'TabLakeA', 'LakeMultiplier', 'LakeInitialLevelValue',
                                          'TabLakeAvNetInflowEstimate', 'PrevDischarge', 'LakePrevInflowValue',
                                          'LakePrevOutflowValue']}   # lakes.py:37-40
    module_name = 'Lakes'

    def initial(self, land_mask):
        v, o = self.var, self.options
        if not o.get('simulateLakes') or o.get('InitLisflood'):
            return
        N = np.asarray(v.IsChannel).size
        sites = self._map('LakeSites', N)                                            # lakes.py:60-63
        sites[~(sites >= 1)] = 0
        sites[np.asarray(v.IsChannel) == 0] = 0
        v.LakeSitesCC = sites[sites > 0]
        v.LakeIndex = np.nonzero(sites)[0]
        if v.LakeSitesCC.size == 0:                                                  # :66-73
            warnings.warn(LisfloodWarning('There are no lakes. Lakes simulation won\'t run'))
            o['simulateLakes'] = False
            o['repsimulateLakes'] = False
            return
        on = sites > 0
        v.IsStructureKinematic = np.where(on, True, np.asarray(v.IsStructureKinematic, bool))   # :76
        # the cells draining into a lake (downstream(LddKinematic, IsStructureLake), :85): their outflow reaches the lake
        d = L.LddDevice(v.LddKinematic, land_mask, self.device)
        v.IsUpsOfStructureLake = d.downstream(on.astype(np.float64)) > 0     # (a pit reads its own value, as in PCRaster)
        d.close()
        inflow = np.bincount(np.asarray(v.downstruct), weights=np.asarray(v.ChanQ, np.float64), minlength=N + 1)
        v.LakeInflowOldCC = inflow[v.LakeIndex]                                      # :91-94
        v.LakeAreaCC = lookupscalar(self.tables['TabLakeArea'], sites)[on]           # :96-98
        v.LakeSitesC2 = sites
        v.LakeACC = (lookupscalar(self.tables['TabLakeA'], sites) * self._map('LakeMultiplier', N))[on]   # :104-106
        level0 = self._map('LakeInitialLevelValue', N)
        cold = np.max(level0) == -9999                                               # :112
        if cold:
            v.LakeAvNetCC = lookupscalar(self.tables['TabLakeAvNetInflowEstimate'], sites)[on]
            storage = v.LakeAreaCC * np.sqrt(v.LakeAvNetCC / v.LakeACC)              # :116
            v.LakeLevelCC = storage / v.LakeAreaCC
        else:
            v.LakeLevelCC = level0[on]
            storage = v.LakeAreaCC * v.LakeLevelCC
            v.LakeAvNetCC = self._map('PrevDischarge', N)[on]                        # :124
        if not cold:
            v.LakeInflowOldCC = self._map('LakePrevInflowValue', N)[on]              # :127-133
        v.LakeFactor = v.LakeAreaCC / (v.DtRouting * np.sqrt(v.LakeACC))             # :140
        v.LakeFactorSqr = np.square(v.LakeFactor)
        indicator = storage / v.DtRouting + v.LakeAvNetCC / 2                        # :146
        out0 = self._map('LakePrevOutflowValue', N)
        if np.max(out0) == -9999:                                                    # :151-157
            v.LakeOutflowCC = np.square(-v.LakeFactor + np.sqrt(v.LakeFactorSqr + 2 * indicator))
        else:
            v.LakeOutflowCC = out0[on]
        v.LakeStorageM3CC = storage.copy()                                           # :160-161
        v.LakeStorageM3BalanceCC = storage.copy()
        for name, cc in (("LakeStorageIniM3", storage), ("LakeLevel", v.LakeLevelCC), ("LakeInflowOld", v.LakeInflowOldCC),
                         ("LakeOutflow", v.LakeOutflowCC)):                          # :165-177
            dense = np.zeros(N)
            dense[v.LakeIndex] = cc
            setattr(v, name, dense)
        v.LakeStorageM3 = v.LakeStorageIniM3.copy()
        v.EWLakeCUMM3, v.EWLakeWBM3 = np.zeros(N), np.zeros(N)


class reservoir(_SiteModule):
    input_files_keys = {'simulateReservoirs': ['ReservoirSites', 'TabTotStorage', 'TabConservativeStorageLimit',
                                               'TabNormalStorageLimit', 'TabFloodStorageLimit', 'TabNonDamagingOutflowQ',
                                               'TabNormalOutflowQ', 'TabMinOutflowQ', 'adjust_Normal_Flood',
                                               'ReservoirRnormqMult', 'ReservoirInitialFillValue']}   # reservoir.py:40-45
    module_name = 'Reservoir'

    def initial(self):
        v, o = self.var, self.options
        if not o.get('simulateReservoirs') or o.get('InitLisflood'):
            return
        N = np.asarray(v.IsChannel).size
        sites = self._map('ReservoirSites', N)                                       # reservoir.py:64-67
        sites[~(sites >= 1)] = 0
        sites[np.asarray(v.IsChannel) == 0] = 0
        v.ReservoirSitesC = sites
        v.ReservoirSitesCC = sites[sites > 0]
        if v.ReservoirSitesCC.size == 0:                                             # :70-76
            warnings.warn(LisfloodWarning('There are no reservoirs. Reservoirs simulation won\'t run'))
            o['simulateReservoirs'] = False
            o['repsimulateReservoirs'] = False
            return
        on = sites > 0
        v.ReservoirIndex = np.nonzero(sites)[0]
        v.IsStructureKinematic = np.where(on, True, np.asarray(v.IsStructureKinematic, bool))   # :81
        look = lambda tab: lookupscalar(self.tables[tab], sites)
        total = look('TabTotStorage')                                                # :91-95
        v.TotalReservoirStorageM3C = np.where(np.isnan(total), 0, total)
        v.TotalReservoirStorageM3CC = v.TotalReservoirStorageM3C[on]
        v.ConservativeStorageLimitCC = look('TabConservativeStorageLimit')[on]       # :98-115
        v.NormalStorageLimitCC = look('TabNormalStorageLimit')[on]
        v.FloodStorageLimitCC = look('TabFloodStorageLimit')[on]
        v.NonDamagingReservoirOutflowCC = look('TabNonDamagingOutflowQ')[on]
        v.NormalReservoirOutflowCC = look('TabNormalOutflowQ')[on]
        v.MinReservoirOutflowCC = look('TabMinOutflowQ')[on]
        adjust = self._map('adjust_Normal_Flood', N)[on]                             # :121-125
        v.Normal_FloodStorageLimitCC = v.NormalStorageLimitCC + adjust * (v.FloodStorageLimitCC - v.NormalStorageLimitCC)
        mult = self._map('ReservoirRnormqMult', N)[on]                               # :128-135
        q = v.NormalReservoirOutflowCC * mult
        q = np.where(q > v.MinReservoirOutflowCC, q, v.MinReservoirOutflowCC + 0.01)
        v.NormalReservoirOutflowCC = np.where(q < v.NonDamagingReservoirOutflowCC, q, v.NonDamagingReservoirOutflowCC - 0.01)
        v.DeltaO = v.NormalReservoirOutflowCC - v.MinReservoirOutflowCC               # :139-142
        v.DeltaLN = v.NormalStorageLimitCC - 2 * v.ConservativeStorageLimitCC
        v.DeltaLF = v.FloodStorageLimitCC - v.NormalStorageLimitCC
        v.DeltaNFL = v.FloodStorageLimitCC - v.Normal_FloodStorageLimitCC
        fill0 = self._map('ReservoirInitialFillValue', N)
        fill = v.NormalStorageLimitCC.copy() if np.max(fill0) == -9999 else fill0[on]   # :148-152
        v.ReservoirFillCC = fill
        storage = fill * v.TotalReservoirStorageM3CC
        v.ReservoirStorageM3CC = storage.copy()
        v.ReservoirFill = np.zeros(N)
        v.ReservoirStorageIniM3 = np.zeros(N)
        v.ReservoirStorageIniM3[v.ReservoirIndex] = storage                          # :163
        v.ReservoirStorageM3 = v.ReservoirStorageIniM3


class structures(_SiteModule):
    module_name = 'Structures'

    def initial(self, land_mask):
        """structures.py:44-61: LddStructuresKinematic keeps the uncut LDD, the cells just upstream of a lake or reservoir
        become pits of LddKinematic (device: lf_downstream + lf_lddrepair)."""
        v, o = self.var, self.options
        v.LddStructuresKinematic = v.LddKinematic
        if o.get('InitLisflood'):
            return
        is_structure = np.asarray(v.IsStructureKinematic, bool)
        d = L.LddDevice(v.LddKinematic, land_mask, self.device)
        # downstream(LddKinematic, IsStructureKinematic); a pit reads its own value, as in PCRaster: a structure that sits
        # on a pit counts as upstream of itself (it is a pit already)
        v.IsUpsOfStructureKinematicC = d.downstream(is_structure.astype(np.float64)) > 0
        d.close()
        v.LddKinematic = L.lddrepair_device(np.where(v.IsUpsOfStructureKinematicC, L.PIT, v.LddKinematic), land_mask,
                                            self.device)                             # :59
