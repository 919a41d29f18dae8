"""bench.py's N > 1 leg: the raster is split into row blocks, one rank (process, GPU) per block, launched by
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
(or any launcher that exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).  No PyTorch in the ranks:
rendezvous, the set-up fixpoint, the barrier and the max-over-ranks clock go through lisflood_amd.dist.SocketTransport
(plain TCP on the node); the data path (halo exchange of boundary discharge every call) is RCCL Send/Recv over xGMI
from csrc/lf_dist.hip.  Strong scaling: the raster (BASELINE.json: 10000 x 10000) is fixed, each rank owns H/N rows.
"""
import json
import os
import sys
import time

import numpy as np

B_ALG = 48.0
HBM_PEAK_GBS = 8000.0


def _flush_c_stdio():
    import ctypes
    try:
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def log(rank, *a):
    print("[bench rank %d]" % rank, *a, file=sys.stderr, flush=True)


def loaded_rccl():
    """path of the librccl this process has mapped (the communicator dlopen()s it): the real one lives under /opt/rocm,
    the tests' stand-in under tests/fake_rccl"""
    try:
        for line in open("/proc/self/maps"):
            if "librccl" in line:
                return line.split()[-1]
    except OSError:
        pass
    return None


def peak_rss_gb():
    import resource
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6


def catchment_leg(a, T, rank, world, device, model_step=True):
    """Same raster, same calls, ranks own whole catchments (lisflood_amd.partition): no halo, no collective on the data
    path.  Every rank derives the partition from the full LDD itself (set-up, untimed).  model_step=False: the router
    calls only (what main() measures FIRST, as the line's fallback should the RCCL path fail)."""
    from lisflood_amd import _lib
    from lisflood_amd import partition as P
    from lisflood_amd import synthetic as syn
    from lisflood_amd.kinematic_wave_parallel import Graph, kinematicWave
    H = W = a.size
    N = H * W
    seed = {"shallow": 1, "deep": 2, "river": 7}[a.family]
    t0 = time.time()
    nq = 3
    err = None
    try:
        raster = syn.make_ldd(a.family, H, W, seed)
        # the outlet of every pixel straight from the 1-byte LDD (pointer jumping on one int32 vector): no whole-raster
        # graph in any rank -- round 3 built one per rank, ~20 GB of host memory each at 10000^2
        roots = P.catchment_roots_of_raster(raster)
        pix_rank, sizes = P.split_catchments(roots, world)
        del roots
        ids = np.nonzero(pix_rank == rank)[0]
        mask = np.zeros(N, bool)
        mask[ids] = True
        codes = raster.reshape(-1)[ids].astype(np.float64)
        del raster, pix_rank
        p = syn.router_params(N)
        kw = kinematicWave(codes, mask.reshape(H, W), p["alpha"][ids], p["beta"], p["dx"][ids], p["dt"], device=device)
        n = kw.num_pixels
        Q = _lib.DeviceArray.from_host(np.ascontiguousarray(p["Q0"][ids]), device)
        del p
        qs = [_lib.DeviceArray.from_host(np.ascontiguousarray(syn.lateral_inflow(N, s)[ids]), device) for s in range(nq)]
        tmp = _lib.DeviceArray(max(n, 1), np.float64, device)
        for d in [Q] + qs:
            kw.to_engine_order(d, tmp)
            d.copy_from(tmp)
        _lib.synchronize(device)
        tmp.free()
        log(rank, "catchment partition: %d cells in %d levels, set-up %.1f s, peak host memory so far %.1f GB"
            % (n, kw.graph.num_levels, time.time() - t0, peak_rss_gb()))
    except Exception as e:
        err = repr(e)
    # all ranks take the same branch: a rank that failed its set-up must not leave the others in a barrier
    if int(T.allreduce(0 if err else 1, "min")) == 0:
        return {"error": err or "set-up failed on another rank"}
    for s in range(a.warmup):
        kw.route_ordered(Q, qs[s % nq])
    _lib.synchronize(device)
    T.barrier()
    t0 = time.perf_counter()
    for s in range(a.steps):
        kw.route_ordered(Q, qs[s % nq])
    _lib.synchronize(device)
    dt_local = time.perf_counter() - t0
    T.barrier()
    dt_max = float(T.allreduce(dt_local, "max"))
    Qh = Q.download()
    ok = T.allreduce(np.array([float(np.isfinite(Qh).all() and (Qh >= 0).all()), float(n), float(Qh.sum())]), "sum")
    cells = T.allgather(int(n))
    for d in qs + [Q]:
        d.free()
    kw.close()
    ms = dt_max * 1e3 / a.steps
    out = {"value": round(N / ms / 1e3, 2), "unit": "Mcell-steps/s", "ms_per_step": round(ms, 4),
           "cells_per_rank": [int(x) for x in cells], "catchments": int(sizes.size),
           "largest_catchment": int(sizes.max()), "finite": bool(ok[0] == world and int(ok[1]) == N),
           "checksum_sumQ": float(ok[2]),
           "note": "ranks own whole catchments (no exchange on the data path); engine-order vectors as in the N = 1 run"}
    ctx = {"codes": codes, "mask": mask, "ids": ids, "H": H, "W": W, "N": N}
    if not model_step:
        out["_ctx"] = ctx      # (popped by main(): what catchment_model_step needs later)
        return out
    return catchment_model_step(a, T, rank, world, device, ctx, out)


def catchment_model_step(a, T, rank, world, device, ctx, out):
    """configs[4]'s workload shape on the catchment partition: a model step of 24 split-routing sub-steps as ONE fused
    wavefront per rank (level blocks + cones), no exchange.  Adds its object to `out`."""
    from lisflood_amd import _lib
    from lisflood_amd import synthetic as syn
    from lisflood_amd.kinematic_wave_parallel import kinematicWave
    codes, mask, ids, H, W, N = ctx["codes"], ctx["mask"], ctx["ids"], ctx["H"], ctx["W"], ctx["N"]
    err = None
    nsteps = 24
    try:
        from bench_support import RoutingStepDevice
        p = syn.router_params(N)
        vals, dtr = syn.model_step_values(N, p, ids=ids)
        kw2 = kinematicWave(codes, mask.reshape(H, W), p["alpha"][ids], p["beta"], p["dx"][ids], dtr,
                            alpha_floodplains=vals["ChannelAlpha2"], device=device)
        st = RoutingStepDevice(kw2, vals, True, p["beta"], 1.0 / dtr, dtr * nsteps, device=device)
        del p, vals
        st.run_fused(nsteps)
        _lib.synchronize(device)
    except Exception as e:
        err = repr(e)
    if int(T.allreduce(0 if err else 1, "min")) == 0:
        out["model_step_24_substeps_split"] = {"error": err or "set-up failed on another rank"}
        return out
    T.barrier()
    t0 = time.perf_counter()
    reps = 2
    for _ in range(reps):
        st.run_fused(nsteps)
    _lib.synchronize(device)
    dt_local = time.perf_counter() - t0
    T.barrier()
    ms2 = float(T.allreduce(dt_local, "max")) * 1e3 / reps
    fin = st.download("ChanQ")
    okq = T.allreduce(float(np.isfinite(fin).all() and (fin >= 0).all()), "sum")
    st.free()
    kw2.close()
    out["model_step_24_substeps_split"] = {
        "ms_per_model_step": round(ms2, 3), "value": round(2 * nsteps * N / ms2 / 1e3, 2), "unit": "Mcell-steps/s",
        "finite": bool(okq == world),
        "note": "lf_routing_substeps_fused on every rank's catchments: 48 cell-steps per cell per model step, no exchange"}
    return out


def row_block_model_step(a, T, rank, world, device, graph, comm, N, i0, i1, nsteps=24, reps=2):
    """lf_dist_routing_substeps_fused on this rank's row block: NoRoutSteps = 24, split routing (2 router calls per
    sub-step = 48 cell-steps per cell and model step)."""
    from lisflood_amd import _lib
    from lisflood_amd import dist as D
    from lisflood_amd import synthetic as syn
    err = None
    try:
        ps = syn.router_params_slice(N, i0, i1)
        vals, dtr = syn.model_step_values_slice(N, i0, i1, ps)
        router = D.DistRouter(graph, ps["alpha"], ps["beta"], ps["dx"], dtr, alpha_floodplains=vals["ChannelAlpha2"],
                              device=device, comm=comm, rank_top=rank - 1 if rank > 0 else -1,
                              rank_bottom=rank + 1 if rank + 1 < world else -1)
        st = D.DistRoutingStep(router, vals, True, ps["beta"], 1.0 / dtr, dtr * nsteps)
        del vals, ps
    except Exception as e:
        err = repr(e)
    if int(T.allreduce(0 if err else 1, "min")) == 0:       # all ranks take the same branch (RCCL calls must pair up)
        return {"error": err or "set-up failed on another rank"}
    st.substeps_fused(nsteps)                                 # warm-up (allocates the slabs)
    _lib.synchronize(device)
    T.barrier()
    t0 = time.perf_counter()
    for _ in range(reps):
        st.substeps_fused(nsteps)
    _lib.synchronize(device)
    dt_local = time.perf_counter() - t0
    T.barrier()
    ms = float(T.allreduce(dt_local, "max")) * 1e3 / reps
    q = st.download("ChanQ")
    chk = T.allreduce(np.array([float(q.sum()), float(np.isfinite(q).all() and (q >= 0).all())]), "sum")
    launches = int(T.allreduce(int(router.last_launches()), "max"))
    # several model steps per call: every phase runs the sub-steps of all of them as one wavefront, ONE halo block per phase
    many = None
    M = 4
    sums = None
    try:
        sums = _lib.DeviceArray((M, max(st.N, 1)), device=device).zero()
        st.fused_prepare(nsteps * M)                          # grows the slabs: the allocation that can fail
    except Exception as e:
        many = {"error": repr(e)}
    if int(T.allreduce(0 if many else 1, "min")) == 0:      # all ranks take the same branch (RCCL calls must pair up)
        many = many or {"error": "set-up failed on another rank"}
    else:
        st.model_steps_fused(nsteps, M, sums)                 # warm-up
        _lib.synchronize(device)
        T.barrier()
        t0 = time.perf_counter()
        st.model_steps_fused(nsteps, M, sums)
        _lib.synchronize(device)
        dt_m = time.perf_counter() - t0
        T.barrier()
        ms_m = float(T.allreduce(dt_m, "max")) * 1e3 / M
        many = {"model_steps_per_call": M, "ms_per_model_step": round(ms_m, 3),
                "value": round(2 * nsteps * N / ms_m / 1e3, 2), "unit": "Mcell-steps/s",
                "max_launches_per_model_step": round(int(T.allreduce(int(router.last_launches()), "max")) / M, 1),
                "halo_exchanges_per_model_step": round((graph.num_phases - 1) / M, 2)}
    if sums is not None:
        sums.free()
    st.free()
    router.close()
    return {"ms_per_model_step": round(ms, 3), "value": round(2 * nsteps * N / ms / 1e3, 2), "unit": "Mcell-steps/s",
            "max_launches_per_model_step": launches, "halo_exchanges_per_model_step": graph.num_phases - 1,
            "checksum_sumChanQ": float(chk[0]), "finite": bool(chk[1] == world),
            "several_model_steps_per_call": many,
            "note": "lf_dist_routing_substeps_fused: per phase one wavefront over (level block, sub-step), slabs "
                    "[slot][sub-step] for what crosses a phase or rank boundary, one RCCL Send/Recv block per phase, "
                    "neighbour and section"}


def headline_line(a, world, N, ms, workload, extra_config, roof_kernel):
    """the contract's keys for the N > 1 line (value = the whole raster's cells per second over all ranks)"""
    value = N / ms / 1e3
    cfg = {"workload": workload, "cells": N}
    cfg.update(extra_config)
    return {
        "metric": "Mcell-steps/s kinematic routing", "value": round(value, 2), "unit": "Mcell-steps/s",
        "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms, 4),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": cfg,
        "hbm_frac_whole_step": round(B_ALG * N / (ms * 1e-3) / 1e9 / (HBM_PEAK_GBS * world), 6),
        # per-GPU algorithmic bandwidth over the WHOLE step (sweeps + packs + RCCL halo rounds), not a
        # per-kernel hipEvent figure: the per-kernel roofline is reported by the N = 1 run
        "roofline": {"bound": "hbm", "kernel": roof_kernel,
                     "achieved": round(B_ALG * N / world / (ms * 1e-3) / 1e9, 3), "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": round(B_ALG * N / world / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 6),
                     "traffic": None},
    }


class Watchdog:
    """Bounds a section that may hang where nothing can be cancelled (a collective whose peer never arrives): after
    `seconds` it runs `on_expiry` (rank 0: print the fallback line) and ends THIS process with os._exit -- every rank
    arms its own, so the launcher sees all of them end.  disarm() before the section's normal end."""

    def __init__(self, seconds, on_expiry, what):
        import threading
        self._t = threading.Timer(seconds, self._fire)
        self._t.daemon = True
        self._on_expiry, self._what, self._seconds = on_expiry, what, seconds
        self._t.start()

    def _fire(self):
        try:
            print("[bench] watchdog: %s did not finish within %.0f s" % (self._what, self._seconds), file=sys.stderr, flush=True)
            self._on_expiry("%s did not finish within %.0f s" % (self._what, self._seconds))
        finally:
            os._exit(0)     # (every rank: the launcher must not turn the line that went out into a failed run)

    def disarm(self):
        self._t.cancel()


def main(a):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on these hosts (before HIP starts)
    from lisflood_amd import _lib
    from lisflood_amd import dist as D
    from lisflood_amd import synthetic as syn

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", str(a.gpus)))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    T = D.SocketTransport.from_env(timeout=300.0)     # a rank that dies must not leave the others waiting for ever
    ndev = _lib.device_count()
    device = local_rank % max(ndev, 1)
    H = W = a.size
    N = H * W
    seed = {"shallow": 1, "deep": 2, "river": 7}[a.family]
    family_name = {"shallow": "random ('shallow')", "deep": "sheet-flow ('deep')", "river": "dendritic ('river')"}[a.family]
    workload = "%dx%d fp64 raster, %s LDD (seed %d), all land, beta=0.6, 1 router call per step" % (H, W, family_name, seed)

    # (1) FIRST the partition that needs no exchange: the same raster, the same calls, ranks own whole catchments and run
    # the single-GPU engine.  It is the line's secondary object when the row-block path below works -- and the line's
    # headline should that path fail or hang on the first real links it meets (nothing before the driver's N > 1 run has
    # ever put RCCL on more than one device: DESIGN.md section 6), so that a scaling point exists either way.
    catch, catch_ctx = None, None
    if os.environ.get("LF_BENCH_CATCHMENT_FIRST", "1") == "1":
        try:
            catch = catchment_leg(a, T, rank, world, device, model_step=False)
            catch_ctx = catch.pop("_ctx", None)
        except Exception as e:  # must never cost the headline line
            catch = {"error": repr(e)}

    out = None
    printed = [False]

    def fallback_line(why):
        """the catchment partition as the headline (rank 0; None if that leg has no number either)"""
        if rank != 0 or not catch or "value" not in catch:
            return None
        ms = catch["ms_per_step"]
        line = headline_line(a, world, N, ms, workload,
                             {"cells_per_rank": catch["cells_per_rank"], "catchments": catch["catchments"],
                              "layout": "engine sweep order per rank",
                              "parallelism": "catchment partition x%d: ranks own whole catchments, no exchange on the data "
                                             "path (fallback headline: the row-block RCCL path did not complete)" % world},
                             "k_level[fused+ordered] (whole step, per GPU)")
        line["checksum_sumQ"] = catch["checksum_sumQ"]
        line["finite"] = catch["finite"]
        line["row_block_error"] = why
        line["rccl_library"] = loaded_rccl()
        return line

    def emit(line=None):    # the ONE JSON line (rank 0), whatever happened after the headline was measured
        line = line if line is not None else out
        if rank == 0 and line is not None and not printed[0]:
            printed[0] = True
            line["peak_host_memory_gb_rank0"] = round(peak_rss_gb(), 2)
            _flush_c_stdio()
            print(json.dumps(line), flush=True)

    def on_hang(why):       # (watchdog thread) the row-block path hangs: what has been measured goes out, the process ends
        emit(out if out is not None else fallback_line(why))

    if rank == 0:           # an exception nobody caught, SIGTERM from a launcher that lost another rank: the line still goes out
        import atexit
        import signal
        atexit.register(lambda: emit(out if out is not None else fallback_line("the row-block path raised before its headline")))
        try:
            signal.signal(signal.SIGTERM, lambda *_: (emit(out if out is not None else fallback_line("SIGTERM before the row-block headline")), os._exit(1)))
        except (ValueError, OSError):
            pass

    # (2) the row-block partition: set-up, RCCL communicator, K pipelined router calls.  Every step that can fail is inside
    # the try; the ranks agree on the outcome over the sockets afterwards (a rank that is stuck in a collective is ended by
    # its watchdog, which makes the others' socket calls fail: they end up in the same branch).
    limit = float(os.environ.get("LF_BENCH_RCCL_TIMEOUT_S", "300"))
    dog = Watchdog(limit, on_hang, "the row-block RCCL path (set-up + %d + %d router calls)" % (a.warmup, a.steps))
    err, comm, router, graph = None, None, None, None
    chk, launches, dt_max = None, 0, None
    try:
        if os.environ.get("LF_BENCH_FAIL_ROW_BLOCKS") == "1":     # (tests: the fallback line)
            raise RuntimeError("LF_BENCH_FAIL_ROW_BLOCKS=1")
        if os.environ.get("LF_BENCH_HANG_ROW_BLOCKS") == "1":     # (tests: the watchdog)
            time.sleep(1e6)
        t_setup = time.time()
        r0, r1 = D.row_blocks(H, world)[rank]
        g0, g1 = max(0, r0 - 1), min(H, r1 + 1)
        codes = syn.make_ldd(a.family, H, W, seed, r0=g0, r1=g1)
        top = codes[0] if r0 > 0 else None
        bot = codes[-1] if r1 < H else None
        local = codes[(r0 - g0):(r0 - g0) + (r1 - r0)]
        graph = D.DistGraph(local, None, top, None, bot, None)
        D.settle_phases(graph, T)
        uid = T.broadcast(D.Comm.unique_id() if rank == 0 else None, src=0)
        # RCCL announces its version / library path on stdout (C stdio) while the communicator is built: send that to
        # stderr so that stdout carries the ONE JSON line and nothing else
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            comm = D.Comm(uid, world, rank, device)
            _flush_c_stdio()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
        i0, i1 = r0 * W, r1 * W
        p = syn.router_params_slice(N, i0, i1)
        router = D.DistRouter(graph, p["alpha"], p["beta"], p["dx"], p["dt"], device=device, comm=comm,
                              rank_top=rank - 1 if rank > 0 else -1, rank_bottom=rank + 1 if rank + 1 < world else -1)
        Q = router.new_state(p["Q0"])
        nq = 3
        qs = [router.new_state(syn.lateral_inflow_slice(N, s, i0, i1)) for s in range(nq)]
        _lib.synchronize(device)
        log(rank, "rows [%d,%d) cells=%d phases=%d launch_units=%d ghosts=%s exports=%s non-contiguous inflow: %d cells; "
            "setup %.1f s, peak host memory %.1f GB" % (r0, r1, graph.num_pixels, graph.num_phases, graph.num_launch_units,
                                                         graph.n_ghost, graph.n_export, graph.num_noncontiguous,
                                                         time.time() - t_setup, peak_rss_gb()))
        # the K calls as ONE pipelined sequence (lf_dist_router_route_many: phase 0 of call s + 1 beside the later halo
        # rounds of call s, alternating state vectors); LF_DIST_OVERLAP=0 or a single phase: call by call
        if a.warmup:
            router.route_many(Q, [qs[s % nq] for s in range(a.warmup)])
        _lib.synchronize(device)
        T.barrier()
        t0 = time.perf_counter()
        router.route_many(Q, [qs[s % nq] for s in range(a.steps)])
        _lib.synchronize(device)
        dt_local = time.perf_counter() - t0
        T.barrier()
        dt_max = float(T.allreduce(dt_local, "max"))
        Qh = router.download_pix(Q)
        chk = T.allreduce(np.array([float(Qh.sum()), float(np.isfinite(Qh).all() and (Qh >= 0).all())]), "sum")
        launches = int(T.allreduce(int(router.last_launches()), "max"))
    except Exception as e:
        err = repr(e)
        log(rank, "row-block path failed: %s" % err)
    try:                    # the ranks agree (a peer that is gone makes this raise: same branch)
        all_ok = int(T.allreduce(0 if err else 1, "min")) == 1
    except Exception as e:
        all_ok, err = False, err or ("a peer left the row-block path: %r" % (e,))
    dog.disarm()
    if not all_ok:
        emit(fallback_line(err or "the row-block path failed on another rank"))
        try:
            T.close()
        except Exception:
            pass
        return
    if rank == 0:           # the headline first: nothing a secondary leg does below can cost it
        ms = dt_max * 1e3 / a.steps
        out = headline_line(a, world, N, ms, workload,
                            {"phases": graph.num_phases, "max_launches_per_step": launches,
                             "layout": "engine sweep order per rank, ghost slots appended",
                             "parallelism": "row-block x%d, RCCL Send/Recv halo per phase on a second stream, calls pipelined "
                                            "on alternating state vectors; rendezvous over TCP sockets (no PyTorch in the "
                                            "ranks)" % world},
                            "k_level[fused+ordered+indexed] (whole step, per GPU)")
        out["checksum_sumQ"] = float(chk[0])
        out["finite"] = bool(chk[1] == world)
        out["rccl_library"] = loaded_rccl()
    # from here on a hang costs the secondary objects only: the watchdog prints the line as it stands
    dog = Watchdog(float(os.environ.get("LF_BENCH_EXTRA_TIMEOUT_S", "600")), on_hang, "a secondary leg of the N > 1 bench")
    # configs[4]'s workload shape on the SAME row blocks: a model step of 24 split-routing sub-steps, every sub-step of a
    # phase as one wavefront (level blocks + cones), ONE RCCL halo block per phase and model step
    row_step = None
    if not getattr(a, "no_extra", False):
        try:
            row_step = row_block_model_step(a, T, rank, world, device, graph, comm, N, i0, i1)
        except Exception as e:
            row_step = {"error": repr(e)}
    # secondary: the model step on the catchment partition (its router calls were measured first, above)
    if catch is not None and catch_ctx is not None and "value" in catch and not getattr(a, "no_extra", False):
        try:
            catchment_model_step(a, T, rank, world, device, catch_ctx, catch)
        except Exception as e:  # must never cost the headline line
            catch["model_step_24_substeps_split"] = {"error": repr(e)}
    if rank == 0:
        if row_step is not None:
            out["model_step_24_substeps_split_row_blocks"] = row_step
        if catch is not None:
            out["catchment_partition"] = catch
            if "checksum_sumQ" in catch and catch["checksum_sumQ"]:
                # the two partitions route the same raster through the same calls and are both bit-identical to the single
                # domain: their discharge sums agree to summation order -- a halo exchange that lost or misplaced a value
                # would show here
                out["row_block_vs_catchment_partition_sumQ_rel_diff"] = abs(float(chk[0]) - catch["checksum_sumQ"]) / abs(
                    catch["checksum_sumQ"])
    try:                    # (a secondary leg that lost a rank leaves the transport unusable: the line below still goes out)
        T.barrier()
        comm.close()
        T.close()
    except Exception as e:
        log(rank, "shutdown: %r" % (e,))
    dog.disarm()
    emit()                  # the ONE JSON line, last on stdout
