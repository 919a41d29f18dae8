"""The whole model step assembled from the C oracle -- TEST / BENCH INFRASTRUCTURE, not part of the product.

One `step()` = Lisflood_dynamic.py:114-229 restricted to the hot path, in the reference's order:
    soilloop.dynamic_canopy -> soilColumnsWaterBalance -> opensealed / soil.dynamic_perpixel / groundwater (per-pixel
    aggregates) -> surface_routing.dynamic -> NoRoutSteps x (lakes / reservoirs / inflow / transmission in-loop,
    routing.dynamic) -> the post-loop bookkeeping (:185-208)
every piece the oracle function pinned at 0-2 ulp against the reference's own methods (tests/test_oracle_golden.py), the
chain as a whole against the reference-driven LF_ETRS89 chain (tests/test_chain.py).  Used by the `-m gpu` chain tests as
the checker and by bench_cpu.py as the CPU baseline of the model step."""
import types

import numpy as np

import oracle


def namespace(values, sc, st=None):
    """the `var` object of the chain from a scenario: values (per-pixel arrays), sc (scalars), st (structures or None)"""
    cp = lambda d: {k: (np.array(a, copy=True) if isinstance(a, np.ndarray) else a) for k, a in d.items()}
    v = types.SimpleNamespace()
    for k, a in list(cp(values).items()) + list(sc.items()) + list(cp(st or {}).items()):
        setattr(v, k, np.ascontiguousarray(a, dtype=np.float64) if isinstance(a, np.ndarray) and a.dtype.kind == "f" else a)
    v.InvBeta, v.InvPixelLength, v.InvDtSec = 1 / v.Beta, 1 / v.PixelLength, 1 / v.DtSec
    v.NoRoutSteps = int(v.NoRoutSteps)
    v.InvDtRouting, v.InvNoRoutSteps = 1 / v.DtRouting, 1 / v.NoRoutSteps
    return v


class OracleChain:
    def __init__(self, values, sc, mask, ldd_to_chan, ldd_kin, structures=None, split=True):
        self.v = v = namespace(values, sc, structures)
        self.mask, self.N, self.split = mask, int(np.asarray(mask).sum()), bool(split)
        self.idx = np.arange(3)
        self.surf = oracle.SurfaceRouting(v, ldd_to_chan, mask)
        self.kw = oracle.kinematicWave(ldd_kin, mask, v.ChannelAlpha, v.Beta, v.ChanLength, v.DtRouting,
                                       alpha_floodplains=v.ChannelAlpha2 if split else None)
        self.stru = oracle.InloopStructures(v) if structures is not None else None
        self.sub = oracle.RoutingSubstep(self.kw, v)
        self.steps_done = 0

    def step(self, forcing, qin_m3=None):
        v, N = self.v, self.N
        for k, a in forcing.items():
            setattr(v, k, np.ascontiguousarray(a, dtype=np.float64))
        oracle.canopy(v, self.idx)                                              # soilloop.py:519-627
        d = dict(vars(v))
        d["ESMax"] = np.ascontiguousarray(v.ESRef * v.LAITerm)                  # soilloop.py:638
        d.update(index_landuse_all=self.idx, is_irrigated=np.array([False, False, True]), is_paddy_irrig=np.zeros(3, bool),
                 paddy_inactive=np.zeros((1, N), bool))
        oracle.soil_columns(d)                                                  # soilloop.py:78-355
        v.TimeSinceStart = float(self.steps_done + 1)
        oracle.pixel_aggregates(v)                                              # opensealed / soil per pixel / groundwater
        self.surf.dynamic()                                                     # surface_routing.py:115-212
        if self.stru is not None and qin_m3 is not None:
            v.QInM3 = qin_m3
            v.QDelta = (v.QInM3 - v.QInM3Old) * v.InvNoRoutSteps                # inflow.py:108
        v.sumDisDay = np.zeros(N)
        for s in range(v.NoRoutSteps):                                          # Lisflood_dynamic.py:179-180
            if self.stru is not None:
                self.stru.dynamic_inloop(s)
                self.sub.dynamic(split=self.split, sideflow_m3=v.SideflowChanM3)
            else:
                self.sub.dynamic(split=self.split, sideflow_m3=v.ToChanM3RunoffDt)
        if self.stru is not None and qin_m3 is not None:
            v.QInM3Old = v.QInM3                                                # Lisflood_dynamic.py:185
        v.ChanM3 = v.ChanM3Kin + (v.Chan2M3Kin - v.Chan2M3Start if self.split else 0.0)
        v.TotalCrossSectionArea = v.ChanM3 * v.InvChanLength
        v.sumDis = getattr(v, "sumDis", 0.0) + v.sumDisDay
        v.ChanQAvg = v.sumDisDay / v.NoRoutSteps                                # the reference's `dis`
        self.steps_done += 1
        return v.ChanQAvg
