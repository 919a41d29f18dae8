#!/usr/bin/env python
"""bench.py -- kinematic-wave routing throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W           # single GPU
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   # row-block split

A "step" is one kinematicWaveRouting call (one pass of the hot path over the whole raster): every land
cell goes through one implicit Newton-Raphson solve = one cell-step.  Inputs (discharge, lateral
inflow, parameters, graph) are resident in HBM when the timed region starts.  One JSON line is printed
by rank 0.

Workload at N=1: the 10000 x 10000 fp64 synthetic raster BASELINE.json quotes the metric on
("random LDD" = `shallow` family, seed 1).  `--family deep` gives the level-sequential worst case
(NL = H+2); a short run of it is reported under "other_workloads".
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "lisflood-code_amd"))

B_ALG = 48.0          # algorithmic bytes per cell-step
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=10000, help="raster is size x size cells")
    ap.add_argument("--family", default="shallow", choices=["shallow", "deep", "river"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the secondary workloads / soil kernel")
    ap.add_argument("--cpu-sample", type=int, default=2000, help="CPU baseline raster is sample x sample")
    ap.add_argument("--only", choices=["soil", "model_step", "hotpath", "structures", "overland", "etrs89"], default=None, help="run only the named secondary benchmark")
    ap.add_argument("--force-dist", action="store_true",
                    help="run the row-block/RCCL path even with a single rank (smoke test of that path)")
    ap.add_argument("--calibrate", action="store_true",
                    help="run the known-traffic stream copies first (PMC calibration under rocprofv3)")
    return ap.parse_args()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def level_sizes(g):
    """Level-size histogram of the input (throughput is a function of NL): the full list when short,
    else a summary."""
    import numpy as np
    ls = getattr(g, "level_widths", None)
    if ls is None:
        ls = np.diff(g.layout()[2])
    if len(ls) <= 32:
        return [int(x) for x in ls]
    q = np.percentile(ls, [0, 25, 50, 75, 100])
    return {"levels": int(len(ls)), "min": int(q[0]), "p25": int(q[1]), "median": int(q[2]), "p75": int(q[3]),
            "max": int(q[4]), "narrow_levels_le_1024": int((ls <= 1024).sum())}


SEEDS = {"shallow": 1, "deep": 2, "river": 7}


def build_case(family, H, W):
    """the synthetic raster of `family` with a router on it: runs of narrow levels are swept in blocks of up to 256 levels,
    cone by cone (k_sweep_cones, one wavefront per cone), wide levels one launch each (k_level)"""
    from lisflood_amd import synthetic as syn
    from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave
    t = time.time()
    codes = syn.make_ldd(family, H, W, SEEDS[family])
    N = H * W
    p = syn.router_params(N)
    g = Graph(ldd_raster=codes)
    g.level_widths = np.diff(g.layout()[2])
    kw = kinematicWave(None, None, p["alpha"], p["beta"], p["dx"], p["dt"], graph=g)
    log("[bench] %s %dx%d: N=%d NL=%d K=%d built in %.1f s" % (family, H, W, N, g.num_levels, g.max_upstream, time.time() - t))
    return kw, p, g


def run_routing(kw, p, steps, warmup, nq=3, profile_steps=2, ordered=True):
    """-> dict(ms_per_step, event_ms_per_step, prof) with inputs resident in HBM.

    ordered=True : discharge / lateral inflow resident in the engine's sweep-order layout (lf_router_route_ordered)
    ordered=False: resident in the reference's pixel order, gathered/scattered inside the sweep (lf_router_route_device)
    """
    from lisflood_amd import _lib
    from lisflood_amd import synthetic as syn
    N = kw.num_pixels
    Q = _lib.DeviceArray.from_host(p["Q0"])
    qs = [_lib.DeviceArray.from_host(syn.lateral_inflow(N, s)) for s in range(nq)]
    if ordered:
        tmp = _lib.DeviceArray(N)
        for d in [Q] + qs:
            kw.to_engine_order(d, tmp)
            d.copy_from(tmp)
        _lib.synchronize()
        tmp.free()
        route = kw.route_ordered
    else:
        route = kw.route_device
    for s in range(warmup):
        route(Q, qs[s % nq])
    _lib.synchronize()
    t0 = time.perf_counter()
    _lib.timer_start()
    for s in range(steps):
        route(Q, qs[s % nq])
    ev_ms = _lib.timer_stop()
    _lib.synchronize()
    t1 = time.perf_counter()
    stats = kw.last_launches()
    # per-kernel durations with hipEvents on the launch stream (separate short pass: the event pairs
    # add host work per launch, so they are kept out of the timed region)
    kw.profile(True)
    kw.profile_read(reset=True)
    for s in range(profile_steps):
        route(Q, qs[s % nq])
    _lib.synchronize()
    prof = kw.profile_read(reset=True)
    kw.profile(False)
    Qh = Q.download()
    ok = bool(np.isfinite(Qh).all() and (Qh >= 0).all())
    import zlib
    crc = zlib.crc32(Qh.tobytes())          # of the final discharge: equal between two runs <=> bit-identical (A/B tools)
    for d in qs + [Q]:
        d.free()
    return dict(ms_per_step=(t1 - t0) * 1e3 / steps, event_ms_per_step=ev_ms / steps, prof=prof, stats=stats,
                finite=ok, profile_steps=profile_steps, cells=N, checksum=crc)


def pmc_traffic(kernel_key, cells_per_launch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC digest (separate --pmc
    passes of this same command, FETCH_SIZE x2 / WRITE_SIZE x1 after calibration on a known-traffic copy;
    tools/gpu_profile.sh -> profiles/*_digest.json).  None when no digest matches this workload."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_digest.json"))):
        try:
            d = json.load(open(f))["kernels"].get(kernel_key)
        except Exception:
            continue
        if d and "hbm_read_bytes_per_launch" in d and abs(d.get("threads_per_launch", 0) / cells_per_launch - 1) < 0.01:
            best = (f, d)
    if not best:
        return None, None
    f, d = best
    return d["hbm_read_bytes_per_launch"] + d["hbm_write_bytes_per_launch"], os.path.relpath(f, ROOT)


def pmc_traffic_r03(workload, kernel_substr):
    """HBM bytes per launch of `kernel_substr` in the newest committed profiles/*_pmc_digest.json (tools/pmc_r03.sh:
    separate --pmc passes of tools/pmc_route.py on this workload, FETCH_SIZE x 2 + WRITE_SIZE) -> (bytes, source, extra)"""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_digest.json")), reverse=True):
        try:
            w = json.load(open(f))["workloads"].get(workload)
        except Exception:
            continue
        if not w:
            continue
        for name, k in w["kernels"].items():
            if kernel_substr in name and "hbm_read_bytes_per_launch" in k:
                extra = {x: round(k[x], 1) for x in ("waves_per_launch", "sq_wave_cycles_per_wave", "sq_wait_any_per_wave",
                                                     "sq_wait_inst_any_per_wave", "sq_active_inst_valu_per_wave",
                                                     "sq_insts_valu_per_wave") if x in k}
                return k["hbm_read_bytes_per_launch"] + k["hbm_write_bytes_per_launch"], w["source"], extra
    return None, None, None


def committed_rocprof_mean(cells_per_launch):
    """Mean duration of the headline kernel in the newest committed rocprofv3 --kernel-trace --stats summary of this
    same workload (profiles/*head_kernel_stats.csv, tools/gpu_profile_r02.sh): another box, another day -- reported
    next to the live hipEvent figure so that the two can be compared (the boxes differ by up to ~8 % on this kernel)."""
    import csv
    import glob
    # newest first by its tag (r02c < r03 < r03d: the text before "head")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*head_kernel_stats.csv")),
                    key=lambda f: os.path.basename(f).split("head")[0], reverse=True):
        try:
            for r in csv.DictReader(open(f)):
                if "k_level<true, true, false" in r["Name"]:       # (round 6: k_level<true, true, false, 2>, the static-record form)
                    us = float(r["AverageNs"]) / 1e3
                    return dict(file=os.path.relpath(f, ROOT), mean_launch_us=round(us, 3), calls=int(r["Calls"]),
                                frac=round(B_ALG * cells_per_launch / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 6))
        except Exception:
            continue
    return None


def roofline_of(res, kernel_key=None, workload=None):
    """Dominant sweep kernel: achieved algorithmic GB/s = 48 B x cells per launch / mean launch duration.
    workload: key of the committed PMC digest (e.g. "route_deep_10000") for roofline.traffic."""
    prof = res["prof"]
    wide, narrow = prof["wide_level"], prof["narrow_run"]
    dom = wide if wide["ms"] >= narrow["ms"] else narrow
    name = "k_level" if dom is wide else "k_sweep_cones_split / k_sweep_cones (blocks of narrow levels, cone by cone)"
    if dom["launches"] == 0 or dom["ms"] == 0:
        return None
    cells_per_launch = dom["cells"] / dom["launches"]
    ms_per_launch = dom["ms"] / dom["launches"]
    achieved = B_ALG * cells_per_launch / (ms_per_launch * 1e-3) / 1e9
    traffic, src = pmc_traffic(kernel_key, cells_per_launch) if kernel_key else (None, None)
    counters = None
    if workload:        # round-3 digests: per workload, the dominant kernel by name
        t3, s3, counters = pmc_traffic_r03(workload, "k_level<true, true, false" if dom is wide else "k_sweep_cones")
        if t3 is not None:
            traffic, src = t3, s3
    check = committed_rocprof_mean(cells_per_launch) if (kernel_key and dom is wide) else None
    return dict(bound="hbm", kernel=name, achieved=round(achieved, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                committed_rocprof=check, pmc_per_wavefront=counters,
                frac=round(achieved / HBM_PEAK_GBS, 6), traffic=traffic, traffic_unit="bytes per launch",
                traffic_source=src, alg_bytes_per_launch=B_ALG * cells_per_launch,
                launches_per_step=int(round(dom["launches"] / max(res.get("profile_steps", 1), 1))),
                mean_launch_us=round(ms_per_launch * 1e3, 3), cells_per_launch=round(cells_per_launch, 1),
                alg_bytes_per_cell_step=B_ALG,
                prep_ms_per_step=round(prof["prep"]["ms"] / max(prof["prep"]["launches"], 1), 4))


def cpu_baseline(family, sample, steps=2, full=10000, budget_s=25.0):
    """The CPU baseline lives in bench_cpu.py and runs as a separate process: the OpenMP runtime must START with thread
    binding (OMP_PROC_BIND=close OMP_PLACES=cores).  It times the oracle (C restatement of the reference algorithm,
    OpenMP where numba uses prange; parallel first touch of every vector) on bounded samples of the GPU legs' workloads:
    routing (team-size sweep {1, 16, 32, 64, physical cores, all threads}, then the best team on the largest raster up to
    the bench's size that fits the budget), soil columns, the whole model step and the LF_ETRS89 chain.  Returns the
    routing figure in the contract's shape with the other three as sub-objects."""
    import subprocess
    cmd = [sys.executable, os.path.join(ROOT, "bench_cpu.py"), "--family", family, "--sample", str(sample), "--full", str(full),
           "--steps", str(steps), "--budget", str(budget_s)]
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, text=True)
        res = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:
        return dict(value=None, unit="Mcell-steps/s", cores=0, kind="port", sample="bench_cpu.py failed: %r" % (e,))
    out = dict(res.get("routing") or dict(value=None, unit="Mcell-steps/s", cores=0, kind="port", sample="routing leg missing"))
    out.update(cpu_model=res.get("cpu_model"), host_cpus=res.get("host_cpus"), usable_cpus=res.get("usable_cpus"),
               physical_cores=res.get("physical_cores"), wall_s=res.get("wall_s"), cgroup_cpu_quota=res.get("cgroup_cpu_quota"),
               loadavg=res.get("loadavg"))
    for k in ("soil", "model_step", "etrs89", "soil_error", "model_step_error", "etrs89_error"):
        if k in res:
            out[k] = res[k]
    return out


def soil_bench(N=4_000_000, steps=10):
    """Secondary metric: soil column-steps/s (lf_soil.hip: k_soil_fused + k_soil_stragglers), 504 B algorithmic per
    (veg,pixel)-step.  Two regimes of the same synthetic soil: `wet` (KSat 5-500 mm/d, 30 mm/d rain: 17-30 % of the
    columns need several Courant sub-steps, mean 2.6-4.9 sub-steps per column, up to 92) and `single_substep` (KSat / 50:
    nearly every column needs one sub-step, the memory-bound regime of the kernel)."""
    import ctypes as C
    from lisflood_amd import _lib
    from lisflood_amd import synthetic as syn
    from lisflood_amd.soilloop import SoilColumnsDevice
    out = {}
    only = os.environ.get("LF_BENCH_SOIL_REGIME")          # (tools/pmc_soil_r04.sh: one regime per counter pass)
    for regime in ("wet", "single_substep"):
        if only and regime != only:
            continue
        d = syn.soil_params(N, seed=3)
        if regime == "single_substep":
            for k in ("KSat1a", "KSat1b", "KSat2"):
                d[k] = d[k] / 50.0
        dev = SoilColumnsDevice(d)
        for _ in range(2):
            dev.step()
        _lib.synchronize()
        _lib.timer_start()
        for _ in range(steps):
            dev.step()
        ms = _lib.timer_stop() / steps
        cols = 3 * N
        gbs = 504.0 * cols / (ms * 1e-3) / 1e9
        nd = C.c_int64(0)
        _lib.check(_lib.lib().lf_soil_last_deferred(C.c_int(0), C.byref(nd)))
        out[regime] = dict(value=round(cols / ms / 1e3, 2), unit="Mcolumn-steps/s", ms_per_step=round(ms, 4),
                           achieved_GBs=round(gbs, 1), frac_hbm=round(gbs / HBM_PEAK_GBS, 4),
                           multi_substep_columns_frac=round(nd.value / cols, 4), derived_parameters_recomputed=bool(dev.derived))
        # committed counter passes of this very command (tools/pmc_soil_r05.sh): HBM bytes per call of the regime's kernels
        tr = {}
        for key, sub in (("columns", "k_soil_fused"), ("stragglers", "k_soil_stragglers")):
            t3, s3, counters = pmc_traffic_r03("soil_%s_%d" % (regime, N), sub)
            if t3 is not None:
                tr[key] = dict(kernel=sub, traffic=round(t3, 1), traffic_unit="bytes per launch", traffic_source=s3,
                               bytes_per_column=round(t3 / cols, 1), pmc_per_wavefront=counters)
        if tr:
            out[regime]["traffic"] = tr
            out[regime]["traffic_bytes_per_column_step"] = round(sum(v["traffic"] for v in tr.values()) / cols, 1)
        for a in dev.dev.values():
            a.free()
    out.update(metric="soil Mcolumn-steps/s", columns=3 * N, alg_bytes_per_column_step=504)
    return out


def model_step_bench(size=5000, nsteps=24, family="deep"):
    """One LISFLOOD model step of channel routing on the `deep` raster: NoRoutSteps = 24 sub-steps, split routing
    (2 router calls per sub-step) = 48 cell-steps per cell, all vectors resident in engine order.
    `fused` = lf_routing_substeps_fused (one skewed wavefront), `sequential` = 24 x lf_routing_substep."""
    from lisflood_amd import _lib
    from lisflood_amd import synthetic as syn
    from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave
    from bench_support import RoutingStepDevice
    H = W = size
    N = H * W
    codes = syn.make_ldd(family, H, W, SEEDS[family])
    p = syn.router_params(N)
    rng = np.random.default_rng(17)
    beta, dt = p["beta"], 3600.0
    alpha, length = p["alpha"], p["dx"]
    alpha2 = alpha * rng.uniform(1.2, 2.0, N)
    g = Graph(ldd_raster=codes)
    kw = kinematicWave(None, None, alpha, beta, length, dt, alpha_floodplains=alpha2, graph=g)
    qlimit = 2.0 * p["Q0"] * rng.uniform(0.3, 1.2, N)
    vals = dict(ChanLength=length, InvChanLength=1 / length, ChannelAlpha=alpha, InvChannelAlpha=1 / alpha,
                ChannelAlpha2=alpha2, InvChannelAlpha2=1 / alpha2, QLimit=qlimit,
                M3Limit=alpha * length * qlimit ** beta, Chan2M3Start=alpha2 * length * qlimit ** beta,
                Chan2QStart=qlimit * 0.1, PixelArea=np.full(N, 2.5e7), IsChannelKinematic=np.ones(N, bool),
                SideflowChanM3=syn.lateral_inflow(N, 0) * length * dt)
    m3 = alpha * length * p["Q0"] ** beta
    vals["Chan2M3Kin"] = vals["Chan2M3Start"].copy()
    vals["ChanM3Kin"] = m3
    vals["ChanQKin"] = p["Q0"].copy()
    vals["Chan2QKin"] = (vals["Chan2M3Kin"] / length / alpha2) ** (1 / beta)
    out = {}
    ref = None
    os.environ["LF_FUSED_LEVELS"] = "1"      # A/B: the wavefront one level per launch (the round-1 kernel)
    kw1 = kinematicWave(None, None, alpha, beta, length, dt, alpha_floodplains=alpha2, graph=g)
    del os.environ["LF_FUSED_LEVELS"]
    for mode in ("fused", "fused_level_by_level", "sequential", "sequential_single_sweep"):
        st = RoutingStepDevice(kw1 if mode == "fused_level_by_level" else kw, vals, True, beta, 1.0 / dt, dt * nsteps)
        run = {"fused": st.run_fused, "fused_level_by_level": st.run_fused, "sequential": st.run_sequential,
               "sequential_single_sweep": st.run_single_sweep}[mode]
        reps = 3 if mode.startswith("fused") else 1
        if mode.startswith("fused"):
            run(nsteps)                  # warm-up
        _lib.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            run(nsteps)
        _lib.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / reps
        out[mode] = dict(ms_per_model_step=round(ms, 3), value=round(2 * nsteps * N / ms / 1e3, 2),
                         unit="Mcell-steps/s",
                         launches_per_model_step=(kw1 if mode == "fused_level_by_level" else kw).last_launches()["launches"] *
                         {"fused": 1, "fused_level_by_level": 1, "sequential": 2 * nsteps, "sequential_single_sweep": nsteps}[mode])
        st.free()
    # several model steps in flight: the skew runs on across the model-step boundaries (lf_routing_model_steps_fused), so
    # the pipeline fill of NB launches is paid once per call instead of once per model step
    try:
        M = 5
        st = RoutingStepDevice(kw, vals, True, beta, 1.0 / dt, dt * nsteps)
        sums = _lib.DeviceArray((M, N)).zero()
        st.run_model_steps_resident(nsteps, M, sums)
        _lib.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            st.run_model_steps_resident(nsteps, M, sums)
        _lib.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / 2 / M
        out["fused_5_model_steps_in_flight"] = dict(ms_per_model_step=round(ms, 3), value=round(2 * nsteps * N / ms / 1e3, 2),
                                                    unit="Mcell-steps/s", model_steps_per_call=M,
                                                    launches_per_model_step=round(kw.last_launches()["launches"] / M, 1),
                                                    frac_hbm=round(2 * B_ALG * N * nsteps / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
        sums.free()
        st.free()
    except Exception as e:
        out["fused_5_model_steps_in_flight"] = {"error": repr(e)}
    out["config"] = "%dx%d %s LDD (NL=%d), NoRoutSteps=%d, split routing: %d cell-steps per cell per model step" % (
        H, W, family, g.num_levels, nsteps, 2 * nsteps)
    t3, s3, counters = pmc_traffic_r03("fused_%s_%d" % (family, size), "k_fused_cones")
    if t3 is not None:      # committed counter passes of this workload: HBM bytes of the cone launches per (cell, sub-step)
        launches = out["fused"]["launches_per_model_step"]
        per = t3 * launches / (N * nsteps)
        ms = out["fused"]["ms_per_model_step"]
        out["fused"]["roofline"] = dict(bound="hbm", kernel="k_fused_cones", traffic=round(t3, 1), traffic_unit="bytes per launch",
                                        traffic_source=s3, hbm_bytes_per_cell_substep=round(per, 1),
                                        achieved=round(2 * B_ALG * N * nsteps / (ms * 1e-3) / 1e9, 1), peak=HBM_PEAK_GBS,
                                        unit="GB/s", frac=round(2 * B_ALG * N * nsteps / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                        moved_GBs=round(per * N * nsteps / (ms * 1e-3) / 1e9, 1), pmc_per_wavefront=counters,
                                        note="frac by the 48 B per router call measure (2 calls per sub-step); moved_GBs = "
                                             "counter traffic / time: the step is bound by its own ~25 state vectors")
    out["note"] = ("fused = lf_routing_substeps_fused: one wavefront over blocks of up to 16 levels, each block swept cone by "
                   "cone through LDS (k_fused_cones); fused_level_by_level = the same call with LF_FUSED_LEVELS=1")
    kw1.close()
    kw.close()
    return out


def structures_step_bench(size=3000, nsteps=24):
    """A model step of routing WITH lakes, reservoirs, inflow hydrographs and transmission loss in the loop
    (routing.py:441-478), device-resident: the whole loop as one wavefront (lf_routing_substeps_fused_structures)
    against sub-step by sub-step (lf_inloop_structures + one level sweep over both routers)."""
    import ctypes as C
    import types
    from lisflood_amd import _lib
    from lisflood_amd import synthetic as syn
    from lisflood_amd.routing import routing
    H = W = size
    N = H * W
    codes = syn.make_ldd("deep", H, W, 2).reshape(-1).astype(np.float64)
    mask = np.ones((H, W), bool)
    p = syn.router_params(N)
    rng = np.random.default_rng(17)
    beta, dt = p["beta"], 3600.0
    alpha, length = p["alpha"], p["dx"]
    alpha2 = alpha * rng.uniform(1.2, 2.0, N)
    qlimit = 2.0 * p["Q0"] * rng.uniform(0.3, 1.2, N)
    v = types.SimpleNamespace(
        ChanLength=length, InvChanLength=1 / length, ChannelAlpha=alpha, InvChannelAlpha=1 / alpha, ChannelAlpha2=alpha2,
        InvChannelAlpha2=1 / alpha2, QLimit=qlimit, M3Limit=alpha * length * qlimit ** beta,
        Chan2M3Start=alpha2 * length * qlimit ** beta, Chan2QStart=qlimit * 0.1, PixelArea=np.full(N, 2.5e7),
        IsChannelKinematic=np.ones(N, bool), Beta=beta, InvBeta=1 / beta, DtRouting=dt, InvDtRouting=1 / dt,
        NoRoutSteps=nsteps, InvNoRoutSteps=1 / nsteps, DtSec=dt * nsteps,
        ToChanM3RunoffDt=syn.lateral_inflow(N, 0) * length * dt)
    v.Chan2M3Kin = v.Chan2M3Start.copy()
    v.ChanM3Kin = alpha * length * p["Q0"] ** beta
    v.ChanQKin = p["Q0"].copy()
    v.Chan2QKin = (v.Chan2M3Kin / length / alpha2) ** (1 / beta)
    v.ChanQ = v.ChanQKin.copy()
    v.CrossSection2Area, v.Sideflow1Chan, v.sumDisDay = np.zeros(N), np.zeros(N), np.zeros(N)
    d, cut = syn.structures_scenario(codes, (H, W), v.ChanQ, dt)
    if os.environ.get("LF_BENCH_NO_TRANS") == "1":        # A/B: no transmission-loss reach (the scenario flags 30 % of the cells)
        d["UpTrans"] = np.zeros(N, bool)
    for k, x in d.items():
        setattr(v, k, x)
    m = routing(v, options=dict(SplitRouting=True, InitLisflood=False, simulateLakes=True, simulateReservoirs=True,
                                inflow=True, TransLoss=True), engine_order=True)
    m.attach_router(cut, mask)
    m.attach_structures()
    m.begin_step()
    m.dynamic_fused()                                   # warm-up; leaves every argument block wired
    L, r = _lib.lib(), m.river_router
    out = {}
    _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        _lib.check(L.lf_routing_substeps_fused_structures(r._h, C.byref(m._args), C.byref(m._inloop), C.c_int(nsteps)))
    _lib.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / 3
    out["fused"] = dict(ms_per_model_step=round(ms, 3), value=round(2 * nsteps * N / ms / 1e3, 2), unit="Mcell-steps/s",
                        launches_per_model_step=r.last_launches()["launches"])
    try:    # committed counter passes of this workload (tools/pmc_r06.sh, profiles/r06_pmc_digest.json)
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_digest.json")), reverse=True):
            w = json.load(open(f)).get("workloads", {}).get("fused_structures_deep_%d" % size)
            if w and "hbm_bytes_per_cell_substep" in w:
                per = w["hbm_bytes_per_cell_substep"]
                out["fused"]["roofline"] = dict(bound="launch-latency", kernel="k_fused_cones (structures in the wavefront)",
                                                hbm_bytes_per_cell_substep=per, traffic_source=w["source"],
                                                frac=round(2 * B_ALG * N * nsteps / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                moved_GBs=round(per * N * nsteps / (ms * 1e-3) / 1e9, 1),
                                                note="counter bytes per (cell, sub-step) against 2 x 48 B algorithmic: the cone "
                                                     "kernel also streams the in-loop vectors (runoff, inflow, transmission loss, "
                                                     "sideflow) and re-reads the statics at each of a cell's 24 visits")
                break
    except Exception:
        pass
    _lib.synchronize()
    t0 = time.perf_counter()
    for s in range(nsteps):
        m._inloop.step = s
        _lib.check(L.lf_inloop_structures(C.c_int(0), C.byref(m._inloop)))
        _lib.check(L.lf_routing_substeps_fused(r._h, C.byref(m._args), C.c_int(1), C.c_int64(0)))
    _lib.synchronize()
    ms = (time.perf_counter() - t0) * 1e3
    out["sequential"] = dict(ms_per_model_step=round(ms, 3), value=round(2 * nsteps * N / ms / 1e3, 2), unit="Mcell-steps/s",
                             launches_per_model_step=(r.last_launches()["launches"] + 2) * nsteps)
    out["config"] = ("%dx%d deep LDD cut at %d lakes + %d reservoirs (NL=%d with their links), 32 inflow points, "
                     "transmission loss on 30 %% of the reaches, NoRoutSteps=%d, split routing"
                     % (H, W, d["LakeIndex"].size, d["ReservoirIndex"].size, r.graph.num_levels, nsteps))
    finite = bool(np.isfinite(m._dev["ChanQ"].download()).all())
    out["finite"] = finite
    # (last: reaches that run dry under the default powers would be NaN under the scenario's own, whose inner term must stay
    # positive -- nothing above may see that state)
    # the same step with the transmission-loss powers of the reference's settings (TransPower1 = 2, TransPower2 = 1 / 2,
    # TransSub = 0.3: cold.xml:1323-1329, transmission.py:57-59) instead of the scenario's 1 / 0.95 and 0.95: numpy evaluates
    # those two as a square and a square root and so does the engine (lf_pow_scalar_exponent) -- one and ~15 instructions
    # instead of two ~75-instruction powers per flagged cell and sub-step
    saved = (m._inloop.TransPower1, m._inloop.TransPower2, m._inloop.TransSub)
    m._inloop.TransPower1, m._inloop.TransPower2, m._inloop.TransSub = 2.0, 0.5, 0.3
    _lib.check(L.lf_routing_substeps_fused_structures(r._h, C.byref(m._args), C.byref(m._inloop), C.c_int(nsteps)))
    _lib.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        _lib.check(L.lf_routing_substeps_fused_structures(r._h, C.byref(m._args), C.byref(m._inloop), C.c_int(nsteps)))
    _lib.synchronize()
    ms_d = (time.perf_counter() - t0) * 1e3 / 3
    out["fused_with_the_settings_default_transmission_powers"] = dict(
        ms_per_model_step=round(ms_d, 3), value=round(2 * nsteps * N / ms_d / 1e3, 2), unit="Mcell-steps/s",
        finite=bool(np.isfinite(m._dev["ChanQ"].download()).all()))
    m._inloop.TransPower1, m._inloop.TransPower2, m._inloop.TransSub = saved
    return out


def overland_bench(size=4000, channel_frac=0.04, steps=6):
    """The three overland routers (surface_routing.py:104-113, 151-153) on a domain where overland flow routes: a `deep`
    land LDD with `channel_frac` channel pixels, LddToChan cut at the channels -- a graph hundreds of levels deep and
    millions of cells wide --, the routers swept together in engine order (three routers per cone / per wide level)."""
    from lisflood_amd import _lib
    from lisflood_amd import synthetic as syn
    from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave
    H = W = size
    N = H * W
    rng = np.random.default_rng(43)
    codes = syn.make_ldd("deep", H, W, SEEDS["deep"])
    is_chan = rng.random((H, W)) < channel_frac
    raster = np.where(is_chan, np.uint8(5), codes)        # ifthenelse(IsChannel, 5, Ldd): channel pixels are pits
    g = Graph(ldd_raster=raster)
    p = syn.router_params(N, seed=12)
    kws = [kinematicWave(None, None, p["alpha"] * f, p["beta"], 5000.0, 86400.0, graph=g) for f in (1.0, 2.5, 0.4)]
    tmp = _lib.DeviceArray(N)
    qs, lats = [], []
    for i, kw in enumerate(kws):
        for host, dst in ((np.minimum(p["Q0"], 50.0) * rng.uniform(0, 1, N), qs), (syn.lateral_inflow(N, i, hi=2e-5), lats)):
            d = _lib.DeviceArray.from_host(host)
            kw.to_engine_order(d, tmp)
            d.copy_from(tmp)
            dst.append(d)
    out = {}
    for mode in ("together", "one_wavefront_per_cone", "one_by_one"):
        if mode == "one_wavefront_per_cone":
            os.environ["LF_ROUTE_SPLIT"] = "0"
        run = (lambda: kinematicWave.route_together(kws, qs, lats, engine_order=True)) if mode != "one_by_one" else (
            lambda: [kw.route_ordered(q, x) for kw, q, x in zip(kws, qs, lats)])
        run()
        _lib.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            run()
        _lib.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / steps
        os.environ.pop("LF_ROUTE_SPLIT", None)
        out[mode] = dict(ms_per_overland_step=round(ms, 3), value=round(3 * N / ms / 1e3, 2), unit="Mcell-steps/s",
                         launches=kws[0].last_launches()["launches"],
                         frac_hbm_whole_step=round(3 * B_ALG * N / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
    out["finite"] = bool(all(np.isfinite(q.download()).all() for q in qs))
    out["config"] = ("%dx%d deep land LDD, %.0f %% channel pixels (pits of LddToChan): NL=%d, K=%d; three routers (other / "
                     "forest / direct) on one graph, engine order, one call each per overland step"
                     % (H, W, 100 * channel_frac, g.num_levels, g.max_upstream if hasattr(g, "max_upstream") else -1))
    out["cone_plan"] = kws[0].route_plan_stats()
    for d in qs + lats + [tmp]:
        d.free()
    for kw in kws:
        kw.close()
    return out


def hotpath_bench(size=2000, steps=6, family="deep", block=1_000_000, lean_too=False):
    """The whole device-resident hot path of a model step (canopy -> soil -> per-pixel aggregates -> 3 overland
    routers -> 24 split-routing channel sub-steps), lisflood_amd.hotpath.HotPathDevice; only the five forcing
    vectors cross PCIe per step.  `stages`: every stage timed on its own (one model step, synchronising after each)
    with its algorithmic bytes (HotPathDevice.stage_bytes) and the fraction of the HBM roofline that makes."""
    from lisflood_amd import _lib
    from lisflood_amd import synthetic as syn
    from lisflood_amd.hotpath import HotPathDevice
    H = W = size
    N = H * W
    t = time.time()
    values, sc, mask, ldd_to_chan, ldd_kin = syn.hotpath_scenario(H, W, family=family, block=block)
    t_scen = time.time() - t
    hp = HotPathDevice(values, sc, mask, ldd_to_chan, ldd_kin, split=True)
    del values
    forc = []
    for s in range(2):      # page-locked forcing buffers, filled in place in the engine's own pixel order (what a netCDF
        f = hp.pinned_forcing()     # reader does when it compresses the raster with the composed index)
        for k, a in syn.hotpath_forcing(N, s).items():
            f[k][:] = a if hp.pixel_of_position is None else a[hp.pixel_of_position]
        forc.append(f)
    log("[bench] hot-path scenario %s %dx%d: fields %.1f s, device set-up %.1f s" % (family, H, W, t_scen, time.time() - t - t_scen))
    for w in range(3):
        hp.step(forc[w % 2], w + 1, ordered=True)
        _lib.synchronize()
    # (a) forcing resident in HBM when the timed region starts (the two buffer sets hold the two forcing sets)
    t0 = time.perf_counter()
    for s in range(steps):
        hp.step(None, s + 4)
    _lib.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    # (b) PCIe-inclusive: every step's five forcing vectors come up from page-locked host memory while the step before runs
    t0 = time.perf_counter()
    hp.prefetch(forc[0], ordered=True)
    for s in range(steps):
        hp.step(forc[s % 2], s + 4 + steps, ordered=True)
        hp.prefetch(forc[(s + 1) % 2], ordered=True)          # next step's forcing goes up while this step's kernels run
    _lib.synchronize()
    ms_up = (time.perf_counter() - t0) * 1e3 / steps
    # (c) the same with the forcing held as float32 (as the reference's meteo files store it) and widened on the device
    forc32 = []
    for f in forc:
        g = hp.pinned_forcing(np.float32)
        for k in f:
            g[k][:] = f[k]
        forc32.append(g)
    hp.step(forc32[0], 3 * steps + 4, ordered=True)
    _lib.synchronize()
    t0 = time.perf_counter()
    hp.prefetch(forc32[1], ordered=True)
    for s in range(steps):
        hp.step(forc32[(s + 1) % 2], 3 * steps + 5 + s, ordered=True)
        hp.prefetch(forc32[s % 2], ordered=True)
    _lib.synchronize()
    ms_up32 = (time.perf_counter() - t0) * 1e3 / steps
    for b_ in range(2):         # the fp64 sets back into the buffers (the legs below step on resident forcing)
        hp.step(forc[b_], 4 * steps + 6 + b_, ordered=True)
    _lib.synchronize()
    q = hp.chan_q_avg()
    out = dict(ms_per_model_step=round(ms, 3), model_steps_per_s=round(1e3 / ms, 2), pixels=N, channel_pixels=int(hp.Nk),
               Mpixel_steps_per_s=round(N / ms / 1e3, 2), finite=bool(np.isfinite(q).all()),
               ms_per_model_step_with_forcing_upload=round(ms_up, 3),
               ms_per_model_step_with_float32_forcing_upload=round(ms_up32, 3),
               forcing_upload_GBs=round(5 * 8 * N / 1e9 / (ms_up * 1e-3), 1),
               levels=dict(channel=int(hp.river.graph.num_levels), overland=int(hp.r_other.graph.num_levels)),
               config="%dx%d %s LDD, 30 %% channel pixels, V=3 fractions, NoRoutSteps=24 split routing; ms_per_model_step: the "
                      "forcing of the step resident in HBM; ..._with_forcing_upload: its five vectors (40 B per pixel) uploaded "
                      "from page-locked host memory every step, overlapped with the step before; the channel wavefront on a "
                      "second stream beside the next step's canopy / soil / overland kernels; parameter fields drawn for %d "
                      "pixels and repeated" % (H, W, family, min(block, N)))
    # A/B: everything on one stream (rounds 1-3)
    hp.overlap_channel = False
    hp.step(None, 2 * steps + 4)
    _lib.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        hp.step(None, 2 * steps + 5 + s)
    _lib.synchronize()
    out["one_stream_ms_per_model_step"] = round((time.perf_counter() - t0) * 1e3 / steps, 3)
    # ... and the two-stream plan again right behind it: the soil drains from step to step and its sub-step counts with it, so
    # only figures taken on the same stretch of steps compare (the headline figure above is taken ~20 steps earlier)
    hp.overlap_channel = True
    hp.step(None, 3 * steps + 5)
    _lib.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        hp.step(None, 3 * steps + 6 + s)
    _lib.synchronize()
    out["two_streams_right_after_ms_per_model_step"] = round((time.perf_counter() - t0) * 1e3 / steps, 3)
    # stage by stage (two profiled steps, the mean)
    acc = {}
    nprof = 2
    for s in range(nprof):
        for k, x in hp.step_profile(forc[s % 2], 2 * steps + 6 + s, ordered=True).items():
            acc[k] = acc.get(k, 0.0) + x / nprof
    nbytes = hp.stage_bytes()
    out["stages"] = {k: dict(ms=round(x, 3), alg_GB=round(nbytes[k] / 1e9, 3),
                             frac_hbm=round(nbytes[k] / (x * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)) for k, x in acc.items()}
    out["stages_sum_ms"] = round(sum(acc.values()), 3)
    out["upload_GB_per_step"] = round(5 * 8 * N / 1e9, 3)
    hp.free()
    if lean_too or os.environ.get("LF_BENCH_LEAN") == "1":
        # the same step with the OPTIONAL maps left out (HotPathDevice(report=())): the soil diagnostics, the per-pixel
        # diagnostics and the cumulative sums of the mass-balance report are not computed and their inputs not streamed --
        # a run that reports `dis` and writes state maps needs none of them.  Never the leg's headline figure.
        values, sc, mask, ldd_to_chan, ldd_kin = syn.hotpath_scenario(H, W, family=family, block=block)
        hp = HotPathDevice(values, sc, mask, ldd_to_chan, ldd_kin, split=True, report=())
        del values
        for s in range(2):
            f = {k: (a if hp.pixel_of_position is None else a[hp.pixel_of_position]) for k, a in syn.hotpath_forcing(N, s).items()}
            hp.step(f, s + 1, ordered=True)
        _lib.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            hp.step(None, s + 3)
        _lib.synchronize()
        lean_ms = (time.perf_counter() - t0) * 1e3 / steps
        acc = hp.step_profile(f, steps + 3, ordered=True)
        nb = hp.stage_bytes()
        out["unreported_maps_left_out"] = dict(
            ms_per_model_step=round(lean_ms, 3), finite=bool(np.isfinite(hp.chan_q_avg()).all()),
            stages={k: dict(ms=round(x, 3), alg_GB=round(nb[k] / 1e9, 3), frac_hbm=round(nb[k] / (x * 1e-3) / 1e9 / HBM_PEAK_GBS, 4))
                    for k, x in acc.items()},
            note="HotPathDevice(report=()): no optional map computed; dis and every state vector identical to the full step "
                 "(tests/test_gpu_parity.py::test_hot_path_with_unreported_maps_left_out)")
        hp.free()
    return out


def etrs89_bench(passes=3):
    """The one real-data configuration (BASELINE.json configs[1]): the LF_ETRS89 domain of tests/golden/etrs89_chain.npz
    (2 847 pixels, 113 levels cut at 5 lakes + 31 reservoirs, 24 split-routing sub-steps, real meteo fields) through
    HotPathDevice -- the fixed cost of a model step (Python, ctypes, ~100+ launches) on a domain far too small to fill the
    GPU, which is the size the reference's users test on.  ms per model step with the forcing uploaded every step (40 kB),
    the launches of the channel wavefront, and `dis` checked against the reference-driven fixture on the first pass."""
    from lisflood_amd import _lib
    from lisflood_amd.hotpath import HotPathDevice
    g = np.load(os.path.join(ROOT, "tests", "golden", "etrs89_chain.npz"))
    cp = lambda d: {k: (np.array(a, copy=True) if isinstance(a, np.ndarray) else a) for k, a in d.items()}
    values = {k[4:]: g[k] for k in g.files if k.startswith("val_")}
    sc = {k[3:]: float(g[k]) for k in g.files if k.startswith("sc_")}
    st = {k[3:]: (g[k] if g[k].ndim else float(g[k])) for k in g.files if k.startswith("st_")}
    qin = np.array(g["QInM3"])                  # (an .npz member is decompressed on every access: not inside the timed loop)
    want_dis = np.array(g["out_ChanQAvg"])
    forcing = [{k[5:]: np.ascontiguousarray(g[k][s]) for k in g.files if k.startswith("forc_")} for s in range(qin.shape[0])]
    t0 = time.perf_counter()
    hp = HotPathDevice(cp(values), sc, g["mask"], g["ldd_to_chan"], g["ldd_cut"], split=True, structures=cp(st))
    setup_s = time.perf_counter() - t0
    worst = 0.0
    for step, f in enumerate(forcing):                     # first pass: parity with the reference's dis, and the warm-up
        hp.step(f, time_since_start=step + 1, QInM3=qin[step])
        want = want_dis[step]
        worst = max(worst, float(np.max(np.abs(hp.chan_q_avg() - want) / np.maximum(np.abs(want), 1e-3))))
    launches = hp.river.last_launches()["launches"]
    _lib.synchronize()
    n = 0
    t0 = time.perf_counter()
    for p in range(passes):
        for step, f in enumerate(forcing):
            hp.step(f, time_since_start=step + 1, QInM3=qin[step])
            n += 1
    _lib.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / n
    N, Nk = hp.N, hp.Nk
    hp.free()
    return dict(ms_per_model_step=round(ms, 3), model_steps_per_s=round(1e3 / ms, 1), pixels=int(N), channel_pixels=int(Nk),
                channel_wavefront_launches=int(launches), setup_s=round(setup_s, 2), dis_max_rel_dev_first_pass=float("%.3e" % worst),
                bound="launch-latency", note="%d model steps timed (wall clock, forcing and inflow uploaded every step); a domain of "
                "%d pixels cannot fill 256 CUs: the step is the sum of its launches and host calls, not of its bytes" % (n, N))


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus > 1 or world > 1 or a.force_dist:
        import bench_dist
        return bench_dist.main(a)
    if a.only == "soil":
        print(json.dumps(soil_bench()), flush=True)
        return
    if a.only == "structures":
        explicit = any(x.startswith("--size") for x in sys.argv)
        print(json.dumps(structures_step_bench(a.size if explicit else 3000)))
        return
    if a.only == "etrs89":
        print(json.dumps(etrs89_bench()), flush=True)
        return
    if a.only == "overland":
        print(json.dumps(overland_bench()), flush=True)
        return
    if a.only == "hotpath":
        explicit = any(x.startswith("--size") or x.startswith("--family") for x in sys.argv)
        print(json.dumps(hotpath_bench(a.size if explicit else 5000, family=a.family if explicit and a.family != "shallow" else "deep")), flush=True)
        return
    if a.only == "model_step":
        # default: the 5000^2 deep raster; `--family shallow --size 10000` with --only model_step gives the wide-level case
        explicit = any(x.startswith("--size") or x.startswith("--family") for x in sys.argv)
        print(json.dumps(model_step_bench(a.size if explicit else 5000, family=a.family if explicit else "deep")), flush=True)
        return
    H = W = a.size
    if a.calibrate:
        import ctypes as C
        from lisflood_amd import _lib
        n = 100_000_000
        src = _lib.DeviceArray(n).zero()
        dst = _lib.DeviceArray(n).zero()
        for width in (8, 16):
            for _ in range(3):
                _lib.check(_lib.lib().lf_calibration_copy(C.c_int(0), src.ptr, dst.ptr, C.c_int64(n), C.c_int(width)))
        _lib.synchronize()
        src.free(); dst.free()
    kw, p, g = build_case(a.family, H, W)
    res = run_routing(kw, p, a.steps, a.warmup)
    N = kw.num_pixels
    value = N / res["ms_per_step"] / 1e3          # Mcell-steps/s
    out = {
        "metric": "Mcell-steps/s kinematic routing", "value": round(value, 2), "unit": "Mcell-steps/s",
        "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(res["ms_per_step"], 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%dx%d fp64 raster, %s LDD (seed %d), all land, beta=0.6, 1 router call per step"
                               % (H, W, {"shallow": "random ('shallow')", "deep": "sheet-flow ('deep')",
                                         "river": "dendritic ('river')"}[a.family], SEEDS[a.family]),
                   "engine_layout": "levels (blocks of levels cone by cone + single wide levels)",
                   "cells": N, "levels": g.num_levels, "level_sizes": level_sizes(g),
                   "launches_per_step": res["stats"]["launches"],
                   "layout": "discharge and lateral inflow resident in HBM in the engine's sweep order "
                             "(lf_router_route_ordered); see pixel_order_call for the reference-order device call",
                   "parallelism": "1 GPU"},
        "event_ms_per_step": round(res["event_ms_per_step"], 4),
        "hbm_frac_whole_step": round(B_ALG * N / (res["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
        "finite": res["finite"],
    }
    out["roofline"] = roofline_of(res, "k_level[fused+ordered]", workload="route_%s_%d" % (a.family, a.size))
    try:
        rp = run_routing(kw, p, max(3, a.steps // 3), 1, nq=2, profile_steps=1, ordered=False)
        out["pixel_order_call"] = dict(value=round(N / rp["ms_per_step"] / 1e3, 2), unit="Mcell-steps/s",
                                       ms_per_step=round(rp["ms_per_step"], 4),
                                       note="same call with vectors resident in the reference's pixel order "
                                            "(lf_router_route_device: gather/scatter through perm inside the sweep)")
        # the reference's own call signature: host numpy vectors in, discharge updated in place (PCIe both ways,
        # pageable memory) -- never the headline value, reported so the cost of staying on the host is visible
        import numpy as _np
        from lisflood_amd import synthetic as _syn
        Qh, qh = _np.array(p["Q0"]), _syn.lateral_inflow(N, 0)
        kw.kinematicWaveRouting(Qh, qh)
        t0 = time.perf_counter()
        for _ in range(2):
            kw.kinematicWaveRouting(Qh, qh)
        host_ms = (time.perf_counter() - t0) / 2 * 1e3
        out["pixel_order_call"]["host_vectors_call"] = dict(
            ms_per_step=round(host_ms, 2), value=round(N / host_ms / 1e3, 2), unit="Mcell-steps/s",
            note="kinematicWaveRouting(discharge, lateral) on host numpy arrays: 16 B/cell up + 8 B/cell down over PCIe")
    except Exception as e:
        out["pixel_order_call"] = {"error": repr(e)}
    kw.close()
    if not a.no_extra:
        extra = {}
        for other in [f for f in ("deep", "river", "shallow") if f != a.family]:
            try:    # the other LDD families
                entry = {}

                def leg(kw_, p_, wl=None):
                    r_ = run_routing(kw_, p_, max(2, a.steps // 5), 1, nq=1, profile_steps=1)
                    return dict(value=round(kw_.num_pixels / r_["ms_per_step"] / 1e3, 2), unit="Mcell-steps/s",
                                ms_per_step=round(r_["ms_per_step"], 3), launches_per_step=r_["stats"]["launches"],
                                roofline=roofline_of(r_, workload=wl))
                kw2, p2, g2 = build_case(other, H, W)
                entry.update(leg(kw2, p2, "route_%s_%d" % (other, H)), levels=g2.num_levels, level_sizes=level_sizes(g2))
                if g2.num_levels > 64:      # deep networks: A/B against one launch per level (same router) ...
                    os.environ["LF_ROUTE_CONES"] = "0"
                    try:
                        entry["level_sweep"] = leg(kw2, p2)
                    finally:
                        del os.environ["LF_ROUTE_CONES"]
                    # ... and against the round-3 cone kernel (one wavefront per cone does everything; same solver)
                    os.environ["LF_ROUTE_SPLIT"] = "0"
                    try:
                        entry["one_wavefront_per_cone"] = leg(kw2, p2)
                    finally:
                        del os.environ["LF_ROUTE_SPLIT"]
                    entry["cone_plan"] = kw2.route_plan_stats()
                kw2.close()
                extra[other] = entry
            except Exception as e:  # secondary numbers must never break the headline line
                extra[other + "_error"] = repr(e)
        try:
            extra["soil"] = soil_bench()
        except Exception as e:
            extra["soil_error"] = repr(e)
        try:
            extra["model_step_24_substeps_split"] = model_step_bench()
        except Exception as e:
            extra["model_step_error"] = repr(e)
        try:
            extra["model_step_with_structures"] = structures_step_bench()
        except Exception as e:
            extra["structures_error"] = repr(e)
        try:
            extra["overland_sparse_channels"] = overland_bench()
        except Exception as e:
            extra["overland_sparse_channels_error"] = repr(e)
        try:
            extra["etrs89_chain"] = etrs89_bench()
        except Exception as e:
            extra["etrs89_chain_error"] = repr(e)
        for fam in ("deep", "river"):    # the whole resident model step at a BASELINE size, stage by stage
            try:
                extra["resident_hot_path_step_%s_5000" % fam] = hotpath_bench(5000, family=fam, lean_too=(fam == "deep"))
            except Exception as e:
                extra["resident_hot_path_%s_error" % fam] = repr(e)
        out["other_workloads"] = extra
    if not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.family, a.cpu_sample, full=a.size)
    # stdout carries ONE compact line (the driver keeps only the tail of it); everything else goes to the sidecar
    detail_path = os.path.join(ROOT, "bench_detail.json")
    try:
        with open(detail_path, "w") as f:
            json.dump(out, f, indent=1)
    except OSError as e:
        detail_path = "not written: %r" % (e,)
    print(json.dumps(compact_line(out, os.path.relpath(detail_path, ROOT) if os.path.isabs(detail_path) else detail_path)),
          flush=True)


def compact_line(out, detail):
    """The bench line the driver sees: the contract's keys, `roofline` and `cpu_baseline` in short form, and per
    secondary leg only ms / value / frac / traffic ratio / launches (<= 4 KB in all; the rest is in `detail`)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "finite")
    line = {k: out[k] for k in keep if k in out}
    c = out["config"]
    line["config"] = dict(workload=c["workload"], cells=c["cells"], levels=c["levels"], launches_per_step=c["launches_per_step"],
                          parallelism=c["parallelism"])
    r = out.get("roofline")
    if r:
        line["roofline"] = dict(bound=r["bound"], kernel=r["kernel"].split(" (")[0], achieved=r["achieved"], peak=r["peak"], unit=r["unit"],
                                frac=r["frac"], traffic=r["traffic"], alg_bytes_per_launch=r["alg_bytes_per_launch"],
                                mean_launch_us=r["mean_launch_us"], launches_per_step=r["launches_per_step"])
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = dict(value=cb["value"], unit=cb["unit"], cores=cb["cores"], kind=cb["kind"],
                                    sample=cb["sample"].split(";")[0].split(" (")[0] + "; C restatement (oracle/lf_oracle.c), "
                                           "OpenMP bound to cores, parallel first touch",
                                    physical_cores=cb.get("physical_cores"), team_rates=cb.get("team_rates"), cpu=cb.get("cpu_model"))
        for k in ("soil", "model_step", "etrs89"):          # the other CPU figures of SURVEY 8(d), next to their GPU legs below
            if isinstance(cb.get(k), dict):
                line["cpu_baseline"][k] = {x: cb[k][x] for x in ("value", "unit", "ms_per_model_step", "cores") if x in cb[k]}
    legs = {}

    def ratio(rf):
        if rf and rf.get("traffic") and rf.get("alg_bytes_per_launch"):
            return round(rf["traffic"] / rf["alg_bytes_per_launch"], 3)
        return None
    ow = out.get("other_workloads", {})
    for fam in ("deep", "river", "shallow"):
        e = ow.get(fam)
        if e:
            rf = e.get("roofline") or {}
            legs["route_" + fam] = dict(ms=e["ms_per_step"], value=e["value"], frac=rf.get("frac"), traffic_ratio=ratio(rf),
                                        launches=e["launches_per_step"])
    po = out.get("pixel_order_call") or {}
    if "ms_per_step" in po:
        legs["pixel_order_call"] = dict(ms=po["ms_per_step"], value=po["value"])
    so = ow.get("soil") or {}
    for regime in ("wet", "single_substep"):
        e = so.get(regime)
        if e:
            tb = e.get("traffic_bytes_per_column_step")
            legs["soil_" + regime] = dict(ms=e["ms_per_step"], value=e["value"], frac=e["frac_hbm"],
                                          traffic_ratio=round(tb / 504.0, 3) if tb else None, launches=2)
    for key, short in (("model_step_24_substeps_split", "model_step_deep_5000"), ("model_step_with_structures", "model_step_structures_3000")):
        e = ow.get(key) or {}
        f = e.get("fused") or {}
        if f:
            rf = f.get("roofline") or {}
            per = rf.get("hbm_bytes_per_cell_substep")
            cells = {"model_step_deep_5000": 25e6, "model_step_structures_3000": 9e6}[short]
            if short == "model_step_deep_5000" and "ms_per_model_step" in (e.get("fused_5_model_steps_in_flight") or {}):
                x = e["fused_5_model_steps_in_flight"]
                legs["model_step_deep_5000_5_in_flight"] = dict(ms=x["ms_per_model_step"], value=x["value"], frac=x["frac_hbm"],
                                                                launches=x["launches_per_model_step"])
            legs[short] = dict(ms=f.get("ms_per_model_step"), value=f.get("value"),
                               frac=rf.get("frac", round(48 * B_ALG * cells / (f["ms_per_model_step"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)),
                               traffic_ratio=round(per / (2 * B_ALG), 3) if per else None, launches=f.get("launches_per_model_step"))
    e = (ow.get("overland_sparse_channels") or {}).get("together")
    if e:
        legs["overland_sparse_4000"] = dict(ms=e["ms_per_overland_step"], value=e["value"], frac=e["frac_hbm_whole_step"], launches=e["launches"])
    for fam in ("deep", "river"):
        e = ow.get("resident_hot_path_step_%s_5000" % fam)
        if e:
            legs["hot_path_%s_5000" % fam] = dict(ms=e["ms_per_model_step"], value=e["Mpixel_steps_per_s"], unit="Mpixel-steps/s",
                                                   ms_with_forcing_upload=e.get("ms_per_model_step_with_forcing_upload"),
                                                   stages={k: [x["ms"], x["frac_hbm"]] for k, x in e["stages"].items()})
            if "unreported_maps_left_out" in e:
                legs["hot_path_%s_5000" % fam]["ms_unreported_maps_left_out"] = e["unreported_maps_left_out"]["ms_per_model_step"]
    e = ow.get("etrs89_chain")
    if e:
        legs["etrs89_chain"] = dict(ms=e["ms_per_model_step"], pixels=e["pixels"], launches=e["channel_wavefront_launches"],
                                    dis_dev=e["dis_max_rel_dev_first_pass"])
        cpu = (cb or {}).get("etrs89") or {}
        if "ms_per_model_step" in cpu:
            legs["etrs89_chain"]["cpu_ms"] = cpu["ms_per_model_step"]
    # what bounds each leg, with the live number that says so (the analysis behind the labels: DESIGN.md section 5):
    #   hbm            bytes actually moved (frac x traffic_ratio) are at the rate this memory system gives many streams
    #   level-latency  a chain of dependent levels: ms / levels is the per-level latency, bytes per level are tiny
    #   launch-latency ms / launches is a kernel launch + drain, not a transfer time
    #   valu           fp64 issue (pow chains of the soil sub-steps, or 64-lane wavefronts a fraction full)
    def bound(leg, kind, **why):
        if leg in legs:
            legs[leg]["bound"] = kind
            legs[leg].update({k: x for k, x in why.items() if x is not None})
    for fam in ("deep", "river", "shallow"):
        leg = legs.get("route_" + fam)
        if leg:
            lv = (ow.get(fam) or {}).get("levels")
            moved = round(leg["frac"] * (leg.get("traffic_ratio") or 1.0), 3) if leg.get("frac") else None
            if lv and lv >= 1000 and (moved or 0.0) < 0.5:
                bound("route_" + fam, "level-latency", us_per_level=round(leg["ms"] * 1e3 / lv, 3), levels=lv)
            else:     # the bytes actually moved are at half the peak or more: the memory system, not the chain, sets the pace
                bound("route_" + fam, "hbm", moved_frac=moved, levels=lv)
    bound("soil_wet", "valu")
    bound("soil_single_substep", "hbm")
    for k in ("model_step_deep_5000", "model_step_deep_5000_5_in_flight"):
        if k in legs:
            tr = legs["model_step_deep_5000"].get("traffic_ratio")
            # (instruction issue of the cone kernel -- ~730 instructions per level of a cone, DESIGN_NOTEBOOK 4.3b -- with the
            # moved bytes at about half the peak)
            bound(k, "valu", moved_frac=round(legs[k]["frac"] * tr, 3) if tr and legs[k].get("frac") else None)
    if "model_step_structures_3000" in legs:
        x = legs["model_step_structures_3000"]
        # (not the launches: 422 / 235 / 141 launches with 16 / 32 / 64 levels per block all take 21.3-22.4 ms -- the ~6 us a lone
        # wavefront needs per level of a cone, times the levels, over the ~1100 cones a launch has in flight)
        bound("model_step_structures_3000", "level-latency", us_per_launch=round(x["ms"] * 1e3 / x["launches"], 1) if x.get("launches") else None)
    bound("overland_sparse_4000", "valu")
    bound("etrs89_chain", "launch-latency")
    if any(k.startswith("hot_path_") for k in legs):
        line["stage_bound"] = dict(land_surface="valu", pixel_aggregates="hbm", overland="hbm", channel_wavefront="hbm")
    if isinstance((cb or {}).get("soil"), dict) and "soil_wet" in legs:
        legs["soil_wet"]["cpu_value"] = cb["soil"]["value"]
    if isinstance((cb or {}).get("model_step"), dict):
        for fam in ("deep", "river"):
            if "hot_path_%s_5000" % fam in legs:
                legs["hot_path_%s_5000" % fam]["cpu_value"] = cb["model_step"]["value"]
    errs = {k: v for k, v in ow.items() if k.endswith("_error")}
    if errs:
        legs["errors"] = {k: str(v)[:80] for k, v in errs.items()}
    line["legs"] = legs
    line["legs_keys"] = ("ms; value (Mcell-steps/s unless unit given); frac = algorithmic bytes / time / 8 TB/s; traffic_ratio = counter / "
                         "algorithmic bytes; launches per step; bound with its live number (moved_frac = frac x traffic_ratio); "
                         "cpu_value / cpu_ms = oracle on the host; stages: [ms, frac]")
    line["detail"] = detail
    return line


if __name__ == "__main__":
    main()
