/*
 * lisflood_amd.h -- C ABI of the MI355X (gfx950) engine for the LISFLOOD per-timestep hot path:
 * kinematic-wave routing, soil-column water balance and LDD-directed reductions.
 *
 * The reference (ec-jrc/lisflood-code v4.3.1) is pure Python and has no FFI of its own; the seams this
 * library sits behind are (paths relative to src/lisflood/hydrological_modules/):
 *   - class kinematicWave                      kinematic_wave_parallel.py:114-184
 *   - numba kernels kinematicRouting/solve1Pixel  kinematic_wave_parallel_tools.py:34-92
 *   - interception_water_balance / soilColumnsWaterBalance   soilloop.py:27-70 / 78-355
 *   - routing.dynamic sub-step arithmetic      routing.py:512-603, 693-703
 *   - np.bincount one-hop upstream sum         routing.py:159-164, lakes.py:215
 * Host code (Python, lisflood-code_amd/lisflood_amd/) binds these entry points with ctypes; the binding
 * a maintainer of the reference would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: opaque handles, raw pointers and sizes, no exceptions across the boundary;
 *   - every function returns LF_OK (0) or a negative LF_E_* code; lf_last_error() gives the message
 *     (thread-local);
 *   - "pixel order" = the reference's compressed 1-D land-pixel vector, row-major over the land mask
 *     (global_modules/add1.py:268-305); all reals are IEEE fp64;
 *   - pointers named *_host are caller-owned host memory, *_dev are device memory obtained from
 *     lf_device_alloc (or any hipMalloc'ed pointer of the same device);
 *   - there is NO CPU fallback: every compute entry point fails with LF_E_NO_DEVICE when no gfx950
 *     device is usable;
 *   - execution model: one HIP stream per device, every entry point enqueues on it and returns (only the *_host
 *     forms, lf_memcpy_* and lf_device_synchronize wait); per-device scratch (soil work lists, router buffers) is
 *     not locked, so drive a device from ONE host thread -- use one process per GPU for multi-GPU work, as
 *     bench.py does.
 */
#ifndef LISFLOOD_AMD_H
#define LISFLOOD_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LF_OK 0
#define LF_E_INVALID (-1)   /* bad argument */
#define LF_E_CYCLE (-2)     /* LDD has a cycle (the reference loops forever: kinematic_wave_parallel.py:99) */
#define LF_E_NO_DEVICE (-3) /* no usable HIP device */
#define LF_E_HIP (-4)       /* HIP runtime error, see lf_last_error() */
#define LF_E_SECTION (-5)   /* section not in {main_channel, floodplains} / floodplains not configured
                               (kinematic_wave_parallel.py:172) */
#define LF_E_COMM (-6)      /* RCCL error */

#define LF_SECTION_MAIN 0        /* "main_channel" */
#define LF_SECTION_FLOODPLAINS 1 /* "floodplains"  */

typedef struct lf_graph lf_graph;   /* host: LDD -> adjacency -> routing orders            */
typedef struct lf_router lf_router; /* device: one kinematicWave instance                   */
typedef struct lf_dist_graph lf_dist_graph;   /* host: one rank's row block of the raster, with halos */
typedef struct lf_dist_router lf_dist_router; /* device: one rank's share of a partitioned kinematicWave */
typedef struct lf_comm lf_comm;               /* RCCL communicator (one rank per GPU)            */

const char *lf_last_error(void);
int lf_version(void);
/* sizeof(lf_substep_args), sizeof(lf_interception_args), sizeof(lf_soil_args), sizeof(lf_canopy_args),
 * sizeof(lf_surface_args), sizeof(lf_inloop_args), sizeof(lf_pixel_args): lets a binding verify its struct mirrors. */
int lf_struct_sizes(int64_t out[7]);

/* ---------------------------------------------------------------------------------------------
 * device plumbing
 * ------------------------------------------------------------------------------------------- */
int lf_device_count(int *count);
int lf_device_name(int device, char *buf, size_t buflen); /* gcnArchName, e.g. "gfx950:sramecc+:xnack-" */
int lf_device_alloc(int device, size_t bytes, void **ptr_dev);
int lf_device_free(int device, void *ptr_dev);
int lf_memcpy_h2d(int device, void *dst_dev, const void *src_host, size_t bytes);
/* the same without waiting: the bytes are copied to a page-locked staging slot owned by the library (the source is free
 * again on return) and go up by DMA on the stream the library calls currently go to, in order with the kernels around it.
 * For small per-step vectors (inflow hydrographs, LAI): lf_memcpy_h2d's wait for the stream would stall the host. */
int lf_memcpy_h2d_staged(int device, void *dst_dev, const void *src_host, size_t bytes);
/* Page-locked host memory for the vectors that cross PCIe every model step (the meteorological forcing, which the
 * reference reads from netCDF into fresh arrays each step, Lisflood_dynamic.py:84-112): a copy from such a buffer is a
 * true asynchronous DMA at the link's rate; from pageable memory the runtime stages it and blocks the caller (measured
 * at 5000^2: 1 GB of forcing per step, 29 GB/s pageable -- longer than the step's kernels).  lf_host_alloc fails
 * without a HIP device like every other entry point. */
int lf_host_alloc(int device, size_t bytes, void **ptr_host);
int lf_host_free(int device, void *ptr_host);
/* Double-buffered uploads on a second HIP stream, so that the copies of the NEXT step's inputs overlap the kernels of
 * the current step.  For buffer set b in {0, 1}:
 *     lf_upload_begin(b); lf_upload_copy(...) ...; lf_upload_end(b)            -- at any time, e.g. right after a step
 *     lf_compute_acquire(b); <entry points whose kernels read set b>; lf_compute_release(b)
 * The copies wait for the kernels that last read set b, those kernels wait for the copies; the other set is free. */
int lf_upload_begin(int device, int set);
int lf_upload_copy(int device, void *dst_dev, const void *src_host, size_t bytes);
/* the same for a vector the caller holds as float32 (the reference's meteo files are float32; readnetcdf widens them on the
 * host): half the bytes over PCIe, widened to fp64 on the device behind the copy -- the same exact conversion */
int lf_upload_copy_f32(int device, double *dst_dev, const float *src_host, size_t count);
int lf_upload_end(int device, int set);
/* host side of the protocol: blocks until the copies of the last lf_upload_begin(set) .. lf_upload_end(set) have finished.
 * Call it before refilling, in place, a page-locked source buffer that was handed to lf_upload_copy for `set` (the copy is
 * a true asynchronous DMA out of that buffer); returns at once if the set was never uploaded. */
int lf_upload_wait(int device, int set);
int lf_compute_acquire(int device, int set);
int lf_compute_release(int device, int set);
/* A second compute stream for a part of a step that the first kernels of the NEXT step do not depend on (the channel
 * wavefront of a model step beside the canopy / soil / overland kernels of the step after it): between _begin and _end every
 * library call of this device goes to the side stream, behind what the main stream holds so far; after _end the side work
 * stays in flight beside the main stream.  _join: the main stream waits for the side work issued so far; the copy entry
 * points (lf_memcpy_*) join by themselves, lf_device_synchronize waits for both streams. */
int lf_side_stream_begin(int device);
int lf_side_stream_end(int device);
int lf_side_stream_join(int device);
/* Lanes: several streams of one device for work items that do not depend on one another -- the blocks of a row-block
 * partition that live on ONE GPU are independent inside a phase (on real hardware each has its own GPU):
 *     lf_lane_fork(); for k: { lf_lane_select(k + 1); <block k's calls> }  lf_lane_select(0); lf_lane_join();
 * a lane first waits for what the main stream held at the fork; the join makes the main stream wait for every lane used. */
int lf_lane_fork(int device);
int lf_lane_select(int device, int lane);
int lf_lane_join(int device);
/* releases what the context keeps for reuse (lane streams, fp32 staging buffers, the staging arena of the *_host forms);
 * waits for the device first, refuses between lf_lane_fork and lf_lane_join or inside a side section */
int lf_device_trim(int device);
int lf_memcpy_d2h(int device, void *dst_host, const void *src_dev, size_t bytes);
int lf_memcpy_d2d(int device, void *dst_dev, const void *src_dev, size_t bytes);
int lf_memset(int device, void *dst_dev, int value, size_t bytes);
int lf_device_synchronize(int device);
/* hipEvent stopwatch on the library's compute stream of `device` (the stream every kernel of this
 * library is launched on): start records an event, stop records another, synchronises and returns the
 * elapsed milliseconds. */
int lf_timer_start(int device);
int lf_timer_stop(int device, double *elapsed_ms);

/* profiling aid: stream copy of n doubles with bytes_per_lane in {8, 16}; its HBM traffic is exactly
 * n*8 B read + n*8 B written, which calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950. */
int lf_calibration_copy(int device, const double *src_dev, double *dst_dev, int64_t n, int bytes_per_lane);
/* nread (1..8) read streams + one write stream of n doubles each (dst[i] = src_0[i] + ... ), 8 bytes per lane: the byte
 * mix of the level sweep without its arithmetic -- the bandwidth ceiling of a kernel with that many streams. */
int lf_calibration_streams(int device, int nread, const double *const *src_dev, double *dst_dev, int64_t n);

/* ---------------------------------------------------------------------------------------------
 * graph: replaces rebuildFlowMatrix/decodeFlowMatrix/streamLookups/topoDistFromSea/_setRoutingOrders
 * (kinematic_wave_parallel.py:59-106, 140-158; kinematic_wave_parallel_tools.py:111-130)
 * ------------------------------------------------------------------------------------------- */
/* ldd_codes: N compressed LISFLOOD keypad codes (doubles, as the reference passes them; 0 = sea,
 * 5 = pit; any value outside {1,2,3,4,6,7,8,9} is "no flow").  land_mask: H*W bytes, non-zero = land. */
int lf_graph_create(const double *ldd_codes, const uint8_t *land_mask, int H, int W, lf_graph **out);
/* same, with zero-length links for the structures of the routing loop: virtual_down[N] (may be NULL; -1 = none)
 * names, for a pit u of THIS (cut) LDD, the pixel v it drains into in the uncut LDD (structures.py:44-61 cuts the
 * LDD just upstream of lakes and reservoirs; routing.py:159-164 keeps the uncut link in `downstruct`).  u gets the
 * same level as v, which is what lets lf_routing_substeps_fused_structures run lakes.py / reservoir.py between two
 * launches of the wavefront.  Levels then differ from the reference's routing orders; router results do not. */
int lf_graph_create_ex(const double *ldd_codes, const uint8_t *land_mask, int H, int W, const int64_t *virtual_down,
                       lf_graph **out);
/* linked[N], by engine position: 1 = the cell is such a zero-length link (it sits at the end of its level, outside
 * every upstream range) */
int lf_graph_get_links(const lf_graph *g, uint8_t *linked);
/* raster form for large domains: H*W uint8 codes; land_mask may be NULL (= all land). */
int lf_graph_create_raster(const uint8_t *ldd_raster, const uint8_t *land_mask, int H, int W, lf_graph **out);
void lf_graph_destroy(lf_graph *g);
int64_t lf_graph_num_pixels(const lf_graph *g);   /* N  */
int64_t lf_graph_num_levels(const lf_graph *g);   /* NL = order_start_stop.shape[0] */
int lf_graph_max_upstream(const lf_graph *g);     /* K  = upstream_lookup.shape[1]  */
/* the reference's attributes, for parity tests: downstream_lookup[N] (float64, -1 = none),
 * upstream_lookup[N*K] (int64, -1 filled, ascending source id), num_upstream_pixels[N] */
int lf_graph_get_lookups(const lf_graph *g, double *downstream, int64_t *upstream, int64_t *num_upstream);
/* pixels_ordered[N], order_start_stop[NL*2] (kinematic_wave_parallel.py:150-158) */
int lf_graph_get_orders(const lf_graph *g, int64_t *pixels_ordered, int64_t *order_start_stop);
/* engine layout: perm[N] = pixel at sweep position p (levels ascending; inside a level, breadth-first
 * from the outlets so that the upstream cells of position p are the contiguous positions
 * [ups_ptr[p], ups_ptr[p+1])); level_start[NL+1]. */
int lf_graph_get_layout(const lf_graph *g, int32_t *perm, int32_t *ups_ptr, int64_t *level_start);

/* ---------------------------------------------------------------------------------------------
 * router: replaces class kinematicWave (kinematic_wave_parallel.py:114-184)
 * ------------------------------------------------------------------------------------------- */
/* alpha[N]; dx[N] or NULL + dx_scalar; alpha_floodplains[N] or NULL (no split routing). */
int lf_router_create(const lf_graph *g, const double *alpha, double beta, const double *dx, double dx_scalar,
                     double dt, const double *alpha_floodplains, int device, lf_router **out);
void lf_router_destroy(lf_router *r);
int lf_router_device(const lf_router *r);
int64_t lf_router_num_pixels(const lf_router *r);
/* kinematicWaveRouting(discharge, specific_lateral_inflow, section): discharge[N] is updated in place.
 * Host-buffer form (PCIe-inclusive): H2D, route, D2H. */
int lf_router_route_host(lf_router *r, double *discharge_host, const double *lateral_host, int section);
/* Device-resident form: discharge_dev[N] (in/out) and lateral_dev[N] in pixel order, asynchronous on the
 * library stream. */
int lf_router_route_device(lf_router *r, double *discharge_dev, const double *lateral_dev, int section);
/* `count` (<= 4) routers built on ONE graph -- surface_routing.py:108-113 builds the direct / other / forest overland
 * routers on the same LDD, and calls them one after the other (:151-153) -- swept together: one launch per level for
 * all of them instead of one per router.  engine_order != 0: vectors in sweep order (lf_router_route_ordered).  Routers
 * with different level schedules are simply swept one after the other. */
int lf_router_route_device_multi(int count, lf_router **routers, double **discharge_dev, const double **lateral_dev,
                                 int section, int engine_order);
/* Engine-order form.  The engine's HBM layout of a per-pixel vector is "sweep order" (levels ascending,
 * breadth-first inside a level; lf_graph_get_layout's perm): in that layout every access of the sweep is
 * a coalesced stream and discharge is updated in place.  Element-wise work (routing.dynamic's fix-ups,
 * sideflow assembly) is order-agnostic, so a model step can keep all its vectors in engine order and
 * convert only at its boundary with these two permutations (dst[p] = src[perm[p]] and its inverse). */
/* dst[i] = src[index[i]], i < n: permutation between two domains (e.g. full-raster pixel order -> engine order of a
 * router that covers a subset of the pixels) */
int lf_gather_device(int device, int64_t n, const int32_t *index_dev, const double *src_dev, double *dst_dev);
int lf_router_to_engine_order(lf_router *r, const double *src_pix_dev, double *dst_ord_dev);
int lf_router_from_engine_order(lf_router *r, const double *src_ord_dev, double *dst_pix_dev);
/* (The first such call of a router with per-pixel channel lengths builds, per section, a 16-byte (alpha*dx/dt, dx) record per
 * cell on the device for its wide levels -- 16 N bytes that stay with the router; LF_LEVEL_STATICS=0 keeps the two vectors.) */
int lf_router_route_ordered(lf_router *r, double *discharge_ord_dev, const double *lateral_ord_dev, int section);
/* nancheck (kinematic_wave_parallel.py:180-184): number of non-finite entries of a device vector. */
int lf_count_nonfinite(int device, const double *x_dev, int64_t n, int64_t *count);
/* launch statistics of the last route call: [0] kernel launches, [1] wide-level launches,
 * [2] narrow-run launches, [3] levels */
int lf_router_last_launches(const lf_router *r, int64_t stats[4]);
/* shape of the block plan of single router calls: [0] blocks, [1] blocks of more than one level (swept cone by cone),
 * and over those: [2] cones, [3] cone levels (one wavefront-level each: 64 lanes), [4] cells, [5] most cones in one launch.
 * [4] / (64 * [3]) = lane use of the cone sweep. */
int lf_router_route_plan_stats(const lf_router *r, int64_t out[6]);
/* the same figures for the plan a router WOULD build on a graph with blocks of up to lmax levels of at most `wide` cells
 * and cones of at most max_cone cells per level (host only: plan shapes can be studied without a device) */
int lf_graph_block_plan_stats(const lf_graph *g, int lmax, int64_t wide, int max_cone, int64_t out[6]);
/* the invariants the cone kernels rely on, checked cell by cell on the host: out[0] cells whose upstream range is not
 * inside the cone's range of the level above (must be 0), out[1] largest LDS slot + 1 a cell reads (<= max_cone),
 * out[2] largest upstream count (<= 8), out[3] cells checked */
int lf_graph_block_plan_check(const lf_graph *g, int lmax, int64_t wide, int max_cone, int64_t out[4]);
/* the order in which the fused cone kernel hands the n cones of one (block, sub-step) to its workgroups: out[i] = cone of
 * launch position i when position 0 is the workgroup with linear id first_linear_id -- the workgroups of one XCD (linear
 * id mod 8) get consecutive cones; a permutation of 0..n-1 (host only: the function the kernel calls, for the tests) */
int lf_xcd_contiguous_order(int n, unsigned int first_linear_id, int32_t *out);
/* per-kernel hipEvent profiling: when enabled every sweep launch is bracketed by an event pair.
 * lf_router_profile_read returns accumulated {launches, milliseconds, cells} per kernel class
 * (0 = prep, 1 = wide level, 2 = narrow run) since the last reset. */
int lf_router_profile_enable(lf_router *r, int on);
int lf_router_profile_read(lf_router *r, double out[9], int reset);

/* ---------------------------------------------------------------------------------------------
 * routing sub-step arithmetic (routing.py:512-603, 693-703), device vectors in pixel order
 * ------------------------------------------------------------------------------------------- */
typedef struct lf_substep_args {
    /* static [N] */
    const double *ChanLength, *InvChanLength, *ChannelAlpha, *InvChannelAlpha, *ChannelAlpha2, *InvChannelAlpha2;
    const double *Chan2M3Start, *Chan2QStart, *M3Limit, *QLimit, *PixelArea;
    const uint8_t *IsChannelKinematic;
    /* per sub-step input [N] */
    const double *SideflowChanM3;
    /* state [N], updated in place */
    double *ChanQKin, *ChanM3Kin, *Chan2QKin, *Chan2M3Kin, *CrossSection2Area, *Sideflow1Chan, *ChanQ, *sumDisDay;
    /* outputs [N] */
    double *FlowVelocity, *TravelDistance;
    /* scratch [N] x 2 */
    double *scratch0, *scratch1;
    double Beta, InvBeta, InvDtRouting, DtSec;
    int32_t split;        /* 0: single routing branch (routing.py:518-538), 1: split routing (543-604) */
    int32_t engine_order; /* 0: all vectors in pixel order, 1: all vectors in engine (sweep) order */
} lf_substep_args;
/* One routing.dynamic() sub-step: sideflow assembly, 1 or 2 router calls, volume/discharge fix-ups,
 * sumDisDay, FlowVelocity/TravelDistance.  All pointers are device memory, all in the same order
 * (pixel order, or engine order when engine_order = 1). */
int lf_routing_substep(lf_router *r, const lf_substep_args *a);
/* its three element-wise stages on their own (0: sideflow assembly, 1: main-channel fix-up + sums, 2: floodplain
 * fix-up) over n cells, for callers that run the router calls in between (lf_dist_routing_substep) */
int lf_substep_stage(int device, int stage, int64_t n, const lf_substep_args *a);
/* nsteps consecutive sub-steps as one skewed wavefront over (level, sub-step): NL + nsteps - 1 launches
 * instead of nsteps x (1..2) x NL.  Needs engine_order = 1.  a->SideflowChanM3 holds the sideflow of every
 * sub-step: sideflow_stride = 0 (one vector used by all sub-steps) or N (nsteps vectors back to back).
 * Bit-identical to nsteps calls of lf_routing_substep. */
int lf_routing_substeps_fused(lf_router *r, const lf_substep_args *a, int nsteps, int64_t sideflow_stride);
/* n_model_steps MODEL steps of steps_per_model_step sub-steps each (the loop of Lisflood_dynamic.py:179-180, model step
 * after model step) as ONE wavefront: a->SideflowChanM3 holds one sideflow vector per model step, sideflow_model_stride
 * elements apart (0: the same vector for all), a->sumDisDay is [n_model_steps][N] and zeroed by the caller (the reference
 * zeroes it at the start of every model step, Lisflood_dynamic.py:177); every other vector of `a` is the state after the
 * last model step, exactly as after n_model_steps calls of lf_routing_substeps_fused.  For callers whose channel routing
 * does not feed back into the sideflow of later steps (no structures in the loop): the land-surface part of several steps
 * first, then their channel routing in one call.  Bit-identical to the step-by-step calls. */
int lf_routing_model_steps_fused(lf_router *r, const lf_substep_args *a, int steps_per_model_step, int n_model_steps,
                                 int64_t sideflow_model_stride);

/* ---------------------------------------------------------------------------------------------
 * LDD one-hop upstream reduction == np.bincount(downstruct, weights)[:N] (routing.py:159-164,
 * lakes.py:215, reservoir.py:190) and upstream(ldd, x) (routing.py:387)
 * ------------------------------------------------------------------------------------------- */
int lf_upstream_sum_device(lf_router *r, const double *w_dev, double *out_dev);
int lf_upstream_sum_host(lf_router *r, const double *w_host, double *out_host);
/* the same reduction directly on H x W rasters (uint8 LDD codes, 0 = sea / missing; fp64 weights, device memory):
 * LDS-staged 3 x 3 neighbourhoods, coalesced rows, neighbours added in ascending source index. */
int lf_upstream_sum_raster_device(int device, const uint8_t *ldd_raster_dev, const double *w_raster_dev,
                                  double *out_raster_dev, int H, int W);
/* accuflux(ldd, x): sum of x over all upstream cells including the cell itself (routing.py:98) */
int lf_accuflux_host(lf_router *r, const double *x_host, double *out_host);

/* accuflux on engine-order device vectors (acc_ord[p] = x_ord[p] + sum over the upstream cells) */
int lf_accuflux_ordered_device(lf_router *r, const double *x_ord_dev, double *acc_ord_dev);

/* ---------------------------------------------------------------------------------------------
 * LDD operations of routing.initial / structures.initial (routing.py:90-171, structures.py:44-61; the reference calls
 * PCRaster 4.3.3: lddrepair, lddmask, downstream, catchment) and the per-catchment totals of routing.dynamic's
 * mass-balance bookkeeping (routing.py:483-499, 645-691).  Rasters are H x W uint8 keypad codes, 0 = missing value.
 * ------------------------------------------------------------------------------------------- */
/* lddrepair: cells draining off the map or into a missing value become pits (routing.py:125) */
int lf_lddrepair_raster_device(int device, const uint8_t *ldd_dev, uint8_t *out_dev, int H, int W);
/* lddmask(ldd, keep): cells outside keep become missing values, cells draining out of keep become pits (:90, :118) */
int lf_lddmask_raster_device(int device, const uint8_t *ldd_dev, const uint8_t *keep_dev, uint8_t *out_dev, int H, int W);
/* host-buffer form of both: keep_host = NULL -> lddrepair, else lddmask */
int lf_ldd_raster_host(int device, const uint8_t *ldd_host, const uint8_t *keep_host, uint8_t *out_host, int H, int W);
/* downstream(ldd, x): value of x at the downstream cell, pits keep their own (routing.py:141, 162; structures.py:51) */
int lf_downstream_device(lf_router *r, const double *x_pix_dev, double *out_pix_dev);
int lf_downstream_host(lf_router *r, const double *x_host, double *out_host);
/* catchment(ldd, points): id of the first non-zero point met going downstream (a point cell belongs to its own
 * catchment), 0 if none (routing.py:168-171).  Pointer jumping over the downstream links: O(log depth) passes. */
int lf_catchments_device(lf_router *r, const int32_t *points_pix_dev, int32_t *labels_pix_dev);
int lf_catchments(lf_router *r, const int64_t *points_host, int64_t *labels_host);
/* np.take(np.bincount(Catchments, weights=w), Catchments) for Catchments = catchment(ldd, pit(ldd)): every cell gets
 * the total of w over its whole tree (routing.py:483-499, 645-691; equal to rounding, the summation order differs) */
/* (synchronous: the totals are complete when it returns; the asynchronous form is the multi-vector one below) */
int lf_catchment_totals_device(lf_router *r, const double *w_pix_dev, double *out_pix_dev);
int lf_catchment_totals_host(lf_router *r, const double *w_host, double *out_host);
/* nv (<= 4) weight vectors in ONE sweep (the mass-balance terms of routing.py:645-691 come several at a time): the
 * device form is asynchronous on the library stream and, after a router's first call (which finds every cell's outlet
 * once, by pointer jumping), neither allocates nor synchronises; the host form takes nv vectors of N doubles back to
 * back.  accuflux runs on the router's block plan (blocks of levels cone by cone), like a router call. */
int lf_catchment_totals_multi_device(lf_router *r, int nv, const double *const *w_pix_dev, double *const *out_pix_dev);
int lf_catchment_totals_multi_host(lf_router *r, int nv, const double *w_host, double *out_host);
int lf_accuflux_ordered_multi_device(lf_router *r, int nv, const double *const *x_ord_dev, double *const *acc_ord_dev);

/* ---------------------------------------------------------------------------------------------
 * soil: replaces interception_water_balance (soilloop.py:27-70) and soilColumnsWaterBalance
 * (soilloop.py:78-355).  Layouts as in the reference: [V,N] / [L,N] C-order fp64, bool arrays 1 byte.
 * ------------------------------------------------------------------------------------------- */
typedef struct lf_interception_args {
    double *Interception, *TaInterception, *LeafDrainage, *CumInterception; /* [V,N] in/out */
    const double *LAI;               /* [V,N] */
    const double *Rain;              /* [N]   */
    const double *TaInterceptionMax; /* [V,N] */
    double drainageK;
    int64_t V, N;
} lf_interception_args;

/* The seven diagnostics Theta1a, Theta1b, Theta2, Sat1a, Sat1b, Sat1, Sat2 (soilloop.py:330-336) are OPTIONAL in the
 * device forms: a NULL pointer = not reported, not computed (nothing else in the column's water balance reads them). */
typedef struct lf_soil_args {
    /* [L,N] statics */
    const uint8_t *PoreSpaceNotZero1a, *PoreSpaceNotZero1b, *PoreSpaceNotZero2;
    const double *KSat1a, *KSat1b, *KSat2, *GenuInvM1a, *GenuInvM1b, *GenuInvM2, *GenuM1a, *GenuM1b, *GenuM2;
    const double *WRes1a, *WRes1b, *WRes1, *WRes2, *WWP1a, *WWP1b, *WWP1, *WWP2, *WFC1a, *WFC1b, *WFC1, *WFC2;
    const double *SoilDepth1a, *SoilDepth1b, *SoilDepth2, *WS1a, *WS1b, *WS1, *WS2, *StoreMaxPervious;
    /* [N] */
    const double *Rain, *SnowMelt, *b_Xinanjiang, *PowerInfPot, *PowerPrefFlow, *UpperZoneK, *GwPercStep;
    const uint8_t *isFrozenSoil;
    /* [V,N] in */
    const double *LeafDrainage, *Interception, *ESMax;
    /* [V,N] in/out and out */
    double *AvailableWaterForInfiltration, *DSLR, *ESAct, *PrefFlow, *Infiltration, *W1a, *W1b, *W1, *W2;
    double *Theta1a, *Theta1b, *Theta2, *Sat1a, *Sat1b, *Sat1, *Sat2, *SeepTopToSubA, *SeepTopToSubB, *SeepSubToGW;
    double *UZOutflow, *UZ, *GwPercUZLZ;
    /* small, ALWAYS host memory */
    const int64_t *index_landuse_all; /* [V] */
    const uint8_t *is_irrigated;      /* [V] */
    const uint8_t *is_paddy_irrig;    /* [V] */
    /* [n_paddy, N]; host memory in lf_soil_columns_host, device memory in lf_soil_columns_device;
     * paddy_any[n_paddy] (host) tells the device form which rows have any inactive pixel */
    const uint8_t *paddy_inactive;
    const uint8_t *paddy_any;
    double DtDay, AvWaterThreshold, CourantCrit, DrainedFraction;
    int64_t V, L, N;
} lf_soil_args;

/* dynamic_canopy (soilloop.py:519-627) for the three prescribed vegetation fractions: TaInterceptionMax,
 * interception kernel, potential transpiration, water-stress reduction, abstraction of transpiration from
 * layers 1a/1b.  [V,N] / [L,N] C-order; index_landuse[V] (host) maps each vegetation row to its land-use row
 * (the reference indexes W1/W1a/W1b by the land-use row there, soilloop.py:592-627 -- reproduced). */
typedef struct lf_canopy_args {
    /* [V,N] in/out */
    double *Interception, *TaInterception, *LeafDrainage, *CumInterception;
    double *potential_transpiration, *RWS, *Ta;
    double *W1a, *W1b, *W1;
    /* [V,N] in */
    const double *LAI, *LAITerm;
    /* [L,N] in */
    const double *CropCoef, *CropGroupNumber, *WFC1, *WFC1a, *WFC1b, *WWP1, *WWP1a, *WWP1b;
    /* [N] in */
    const double *Rain, *EWRef, *ETRef;
    const uint8_t *isFrozenSoil;
    const int64_t *index_landuse; /* [V], host */
    double LeafDrainageK, DtDay, InvDtDay;
    int64_t V, L, N;
    /* option branches of the method, each switched on by a non-NULL output:
     *   repStressDays (soilloop.py:597-598): SoilMoistureStressDays[V,N] = DtDay where RWS < 1, else 0
     *   wateruse      (soilloop.py:582-587): WFilla[N] / WFillb[N] = min(WCrit1a / WCrit1b, WPF3a / WPF3b) of the land-use
     *                 row of vegetation row `irrigated_veg` (the "Irrigated" fraction; < 0: none); WPF3a / WPF3b [L,N] */
    double *SoilMoistureStressDays;
    double *WFilla, *WFillb;
    const double *WPF3a, *WPF3b;
    int64_t irrigated_veg;
} lf_canopy_args;
int lf_canopy_device(int device, const lf_canopy_args *a);
/* The land surface of a model step in ONE pass over the columns: dynamic_canopy (soilloop.py:519-627), ESMax = ESRef *
 * LAITerm (:638) and soilColumnsWaterBalance (:78-355) -- the lane that runs a column's canopy carries LeafDrainage,
 * Interception, W1a / W1b / W1 and ESMax into the column's soil water balance in registers (Lisflood_dynamic.py:114-123
 * calls the two methods back to back; nothing reads the vectors in between).  Bit-identical to lf_canopy_device +
 * lf_scale_rows_device + lf_soil_columns_device (derived != 0: _derived).  `canopy` and `soil` must describe the same
 * columns: V = L, same N, index_landuse[v] == v in both, no paddy fraction, and the vectors both structs name (W1a, W1b,
 * W1, Interception, LeafDrainage, Rain, isFrozenSoil, WWP1a/b, WFC1a/b) must be the same device vectors -- LF_E_INVALID
 * otherwise (use the separate entry points then).  soil->ESMax is not read; ESRef_dev is the [N] vector. */
int lf_land_columns_device(int device, const lf_canopy_args *canopy, const lf_soil_args *soil, const double *ESRef_dev,
                           int derived);

/* suctionUnsaturatedSoilPF + pressureHead (soilloop.py:427-432, 673-695; option simulatePF): pF = log10 of the capillary
 * head of the three soil layers, -1 where the head is not positive.  [V,N] by vegetation row, [L,N] by land-use row. */
typedef struct lf_soil_pf_args {
    double *pF0, *pF1, *pF2;                 /* [V,N] out */
    const double *W1a, *W1b, *W2;            /* [V,N] */
    const double *WRes1a, *WRes1b, *WRes2, *WS1a, *WS1b, *WS2; /* [L,N] */
    const uint8_t *PoreSpaceNotZero1a, *PoreSpaceNotZero1b, *PoreSpaceNotZero2;
    const double *GenuInvAlpha1a, *GenuInvAlpha1b, *GenuInvAlpha2, *GenuInvM1a, *GenuInvM1b, *GenuInvM2, *GenuInvN1a,
        *GenuInvN1b, *GenuInvN2;             /* [L,N] */
    const int64_t *index_landuse_all;        /* [V], host */
    double HeadMax;
    int64_t V, L, N;
} lf_soil_pf_args;
int lf_soil_pf_device(int device, const lf_soil_pf_args *a);
/* out[v,p] = row[p] * m[v,p]  (ESMax = ESRef * LAITerm, soilloop.py:638) */
int lf_scale_rows_device(int device, const double *row_dev, const double *m_dev, double *out_dev, int64_t V, int64_t N);

/* surface_routing.dynamic (surface_routing.py:115-212), prescribed fractions (V = L = 3, rows Rainfed /
 * Forest / Irrigated), all vectors device memory in pixel order.  Three routers share one graph. */
typedef struct lf_surface_args {
    /* [3,N] in */
    const double *SoilFraction, *AvailableWaterForInfiltration, *Infiltration, *OFAlpha; /* OFAlpha rows: Other, Forest, Direct */
    /* [N] in */
    const double *DirectRunoff, *UZOutflowPixel, *LZOutflowToChannelPixel;
    const uint8_t *IsChannel;
    /* [N] state, in/out */
    double *OFQDirect, *OFQOther, *OFQForest;
    /* [N] out */
    double *OFM3Direct, *OFM3Other, *OFM3Forest, *SurfaceRunoff, *TotalRunoff, *OFToChanM3, *WaterDepth, *ToChanM3Runoff,
        *ToChanM3RunoffDt;
    double *SurfaceRunSoil; /* [3,N] out */
    double *scratch;        /* [3,N] */
    double Beta, MMtoM3, M3toMM, PixelLength, InvPixelLength, DtSec, InvDtSec, InvNoRoutSteps;
    int64_t N;
} lf_surface_args;
int lf_surface_step(lf_router *direct_router, lf_router *other_router, lf_router *forest_router,
                    const lf_surface_args *a);
/* The same with EVERY vector of `a` in the sweep (engine) order of the three routers' shared graph instead of pixel order:
 * the element-wise parts do not care, and the routers then stream their vectors (lf_router_route_ordered: ~54 B per cell
 * instead of ~135 through the position -> pixel table).  A caller that keeps a whole model step resident permutes its
 * per-pixel fields once (lf_graph_layout gives the table) -- lisflood_amd.hotpath.HotPathDevice does. */
int lf_surface_step_ordered(lf_router *direct_router, lf_router *other_router, lf_router *forest_router,
                            const lf_surface_args *a);

/* In-loop structures of routing.dynamic (routing.py:441-478): lakes (lakes.py:199-297, Modified Puls), reservoirs
 * (reservoir.py:173-322), inflow hydrographs (inflow.py:129-147), transmission loss (transmission.py:67-89) and
 * the SideflowChanM3 assembly (routing.py:462-478), all on device vectors in ONE order (pixel or engine).
 * Sites are few (LF_ETRS89: 5 lakes, 64 reservoirs): one lane per site; per-site inflow = sum of ChanQ over
 * site_ups_idx[site_ups_ptr[i] .. site_ups_ptr[i+1]) in ascending pixel id (np.bincount(downstruct, ChanQ)).
 * Any pointer group may be NULL / its count 0 (option switched off). */
typedef struct lf_inloop_args {
    const double *ChanQ; /* [N] discharge at the start of the sub-step */
    /* lakes [n_lakes] */
    int64_t n_lakes;
    const int32_t *lake_cell, *lake_ups_ptr, *lake_ups_idx;
    const double *LakeFactor, *LakeFactorSqr, *LakeAreaCC;
    double *LakeStorageM3CC, *LakeInflowOldCC, *LakeOutflowCC, *LakeStorageM3BalanceCC, *LakeLevelCC, *LakeInflowCC;
    double *QLakeOutM3Dt; /* [N], written at the lake cells only */
    /* reservoirs [n_res] */
    int64_t n_res;
    const int32_t *res_cell, *res_ups_ptr, *res_ups_idx;
    const double *TotalReservoirStorageM3CC, *MinReservoirOutflowCC, *NormalReservoirOutflowCC,
        *NonDamagingReservoirOutflowCC, *ConservativeStorageLimitCC, *NormalStorageLimitCC, *FloodStorageLimitCC,
        *Normal_FloodStorageLimitCC, *DeltaO, *DeltaLN, *DeltaNFL;
    double *ReservoirStorageM3CC, *ReservoirFillCC, *ReservoirInflowCC;
    double *QResOutM3Dt; /* [N], written at the reservoir cells only */
    /* inflow hydrographs [N] (NULL = option off) */
    const double *QInM3Old, *QDelta;
    double *QInDt, *QinADDEDM3;
    /* transmission loss [N] (NULL = option off) */
    const uint8_t *UpTrans;
    double *TransLossM3Dt, *TransCum;
    double TransPower1, TransPower2, TransSub;
    /* sideflow assembly [N]: out = ToChanM3RunoffDt - EvaAddM3Dt - WUseAddM3Dt + QInDt - TransLossM3Dt
     *                              + QLakeOutM3Dt + QResOutM3Dt - ChannelToPolderM3Dt   (NULL terms skipped) */
    const double *ToChanM3RunoffDt, *EvaAddM3Dt, *WUseAddM3Dt, *ChannelToPolderM3Dt;
    double *SideflowChanM3;
    double DtRouting, InvNoRoutSteps;
    int64_t N;
    int32_t step; /* NoRoutingExecuted */
} lf_inloop_args;
int lf_inloop_structures(int device, const lf_inloop_args *a);
/* The whole loop `for s in range(NoRoutSteps): lakes/reservoir/inflow/transmission.dynamic_inloop(s);
 * routing.dynamic(s)` (Lisflood_dynamic.py:179-180 with routing.py:441-478) as ONE skewed wavefront: the cell kernel
 * of lf_routing_substeps_fused assembles SideflowChanM3 itself (inflow hydrographs, transmission loss, the
 * structures' outflow) and a one-lane-per-site kernel runs each lake / reservoir for sub-step s = t - level(site)
 * right before launch t.  Needs: engine_order = 1 everywhere (`in` holds engine-order vectors and site lists in
 * engine positions, `in->step` is ignored), and a router whose graph was built by lf_graph_create_ex with the
 * uncut links of the structures (every cell feeding a site on the site's level; checked once per set of device
 * site lists, whose contents must not change afterwards).  Bit-identical to the
 * sub-step-by-sub-step sequence lf_inloop_structures + lf_routing_substep. */
int lf_routing_substeps_fused_structures(lf_router *r, const lf_substep_args *a, const lf_inloop_args *in, int nsteps);
/* drops the cached validation of the site lists (call after rebuilding lake_cell / lake_ups_idx / res_cell /
 * res_ups_idx, even if the new lists live at the old addresses) */
int lf_router_reset_site_cache(lf_router *r);

/* The per-pixel aggregates between the soil columns and surface routing, one pass:
 * opensealed.dynamic (opensealed.py:40-71), soil.dynamic_perpixel (soil.py:471-514; deffraction =
 * sum over the fractions of SoilFraction * X, Lisflood_initial.py:69-71,393-396) and groundwater.dynamic
 * (groundwater.py:134-180).  Prescribed fractions (V = L = 3, fraction v uses land-use row v). */
/* Every output but DirectRunoff, UZOutflowPixel, LZOutflowToChannelPixel and the states CumInterSealed / LZ is OPTIONAL:
 * a NULL pointer means the map is not reported -- it is not computed, and a [3,N] input only it reads may be NULL too
 * (the reference computes all of them every step and writes the ones its rep* options name). */
typedef struct lf_pixel_args {
    /* [3,N] in */
    const double *SoilFraction, *TaInterception, *Ta, *ESAct, *PrefFlow, *Infiltration, *SeepTopToSubA, *SeepTopToSubB,
        *SeepSubToGW, *Theta1a, *Theta1b, *Theta2, *W1a, *W1b, *W2, *UZOutflow, *GwPercUZLZ, *SoilDepthTotal;
    /* [N] in */
    const double *Rain, *SnowMelt, *EWRef, *SMaxSealed, *DirectRunoffFraction, *WaterFraction, *LowerZoneK, *LZThreshold,
        *GwLossStep;
    /* [N] state in/out */
    double *CumInterSealed, *LZ, *LZInflowCUM, *TaInterceptionCUM, *TaCUM, *ESActCUM, *GwLossCUM;
    /* [N] out */
    double *RainSnowmelt, *EWaterAct, *InterSealed, *TASealed, *DirectRunoff, *TaInterceptionAll, *TaPixel, *ESActPixel,
        *PrefFlowPixel, *InfiltrationPixel, *ThetaAll, *SeepTopToSubPixelA, *SeepTopToSubPixelB, *SeepSubToGWPixel,
        *Theta1aPixel, *Theta1bPixel, *Theta2Pixel, *LZOutflow, *UZOutflowPixel, *GwPercUZLZPixel, *GwLossLZ, *LZAvInflow,
        *LZOutflowToChannelPixel;
    double *Theta; /* [3,N] out */
    double InvDtDay, TimeSinceStart;
    int64_t N;
} lf_pixel_args;
int lf_pixel_aggregates_device(int device, const lf_pixel_args *a);

/* host-buffer forms (drop-in for the numba kernels; PCIe-inclusive) */
int lf_interception_host(int device, const lf_interception_args *a);
int lf_soil_columns_host(int device, const lf_soil_args *a);
/* device-resident forms: every array pointer is device memory (except the small per-vegetation ones).  The soil call keeps a
 * per-device workspace for the columns that leave their tile (above LF_SOIL_TRIP_CAP = 6 Courant sub-steps, up to 48 per
 * tile of 256 columns): ~15.4 KB per tile = ~60 bytes per column, grow-only, released by lf_device_trim. */
int lf_interception_device(int device, const lf_interception_args *a);
int lf_soil_columns_device(int device, const lf_soil_args *a);
/* The same for a caller that vouches for the relations soil.py:180-228 establishes between its parameter arrays (GenuInvM =
 * 1 / GenuM; WS1 = WS1a + WS1b, and so WRes1, WFC1, WWP1; PoreSpaceNotZero = SoilDepth != 0 and WS != 0): those ten arrays
 * are recomputed (one IEEE operation each: the same bits) instead of read and may be NULL -- 59 bytes less per column. */
int lf_soil_columns_device_derived(int device, const lf_soil_args *a);
/* instrumentation: columns of the last lf_soil_columns_device call that needed > 1 Courant sub-step */
int lf_soil_last_deferred(int device, int64_t *count);
/* ... and their histogram by trip count (hist[k] = columns with k sub-steps, last bin = nbins-1 or more; the engine
 * keeps counts up to 127) */
int lf_soil_substep_histogram(int device, int64_t *hist, int nbins);

/* ---------------------------------------------------------------------------------------------
 * multi-GPU: the raster is split into contiguous row blocks, one rank (process, GPU) per block; boundary
 * discharge is exchanged with RCCL Send/Recv between vertical neighbours.  The reference has no
 * distributed code; the result is bit-identical to the single-domain call (see csrc/lf_dist.hip).
 * Setup protocol (host side, any transport for the tiny phase vectors):
 *   1. lf_dist_graph_create with the rank's own rows and the one halo row above / below (NULL at the
 *      raster's edge);
 *   2. repeat { get_export_phases -> swap with the neighbours -> set_ghost_phases } until no rank changes;
 *   3. lf_dist_graph_finalize(max over ranks of lf_dist_graph_local_num_phases).
 * ------------------------------------------------------------------------------------------- */
int lf_dist_graph_create(const uint8_t *ldd_local, const uint8_t *mask_local, int H_local, int W,
                         const uint8_t *ldd_top, const uint8_t *mask_top, const uint8_t *ldd_bottom,
                         const uint8_t *mask_bottom, lf_dist_graph **out);
void lf_dist_graph_destroy(lf_dist_graph *g);
int64_t lf_dist_graph_num_pixels(const lf_dist_graph *g);
int lf_dist_graph_counts(const lf_dist_graph *g, int64_t out[4]); /* exports top,bottom; ghosts top,bottom */
int lf_dist_graph_get_export_phases(const lf_dist_graph *g, int32_t *top, int32_t *bottom);
int lf_dist_graph_set_ghost_phases(lf_dist_graph *g, const int32_t *top, const int32_t *bottom, int *changed);
int lf_dist_graph_local_num_phases(const lf_dist_graph *g);
int lf_dist_graph_finalize(lf_dist_graph *g, int nphases);
int64_t lf_dist_graph_state_size(const lf_dist_graph *g); /* N local cells + ghost slots */
int lf_dist_graph_num_phases(const lf_dist_graph *g);
int64_t lf_dist_graph_num_launch_units(const lf_dist_graph *g);
/* cells whose upstream positions are not consecutive in the sweep order (they read through the index list) */
int64_t lf_dist_graph_num_noncontiguous(const lf_dist_graph *g);
int lf_dist_graph_get_layout(const lf_dist_graph *g, int32_t *perm, int32_t *phase_of_position);
int lf_dist_graph_get_csr(const lf_dist_graph *g, int32_t *ups_ptr, int32_t *ups_idx, int64_t *n_edges);
int lf_dist_graph_phase_range(const lf_dist_graph *g, int phase, int64_t out[2]);
int lf_dist_graph_round_counts(const lf_dist_graph *g, int round, int64_t out[4]); /* send t,b; recv t,b */
int lf_dist_graph_round_send_positions(const lf_dist_graph *g, int round, int side, int32_t *positions);
int64_t lf_dist_graph_round_recv_slot(const lf_dist_graph *g, int round, int side);

int lf_comm_unique_id(char id[128]);                 /* rank 0; broadcast the 128 bytes to the other ranks */
int lf_comm_create(const char id[128], int nranks, int rank, int device, lf_comm **out);
void lf_comm_destroy(lf_comm *c);
/* the same, returning what the communicator's teardown reports (an asynchronous error of an earlier Send / Recv) */
int lf_comm_close(lf_comm *c);

/* alpha / dx / alpha_floodplains: per LOCAL pixel (row-major over the rank's own rows) */
int lf_dist_router_create(const lf_dist_graph *g, const double *alpha, double beta, const double *dx,
                          double dx_scalar, double dt, const double *alpha_floodplains, int device,
                          lf_dist_router **out);
void lf_dist_router_destroy(lf_dist_router *r);
int64_t lf_dist_router_state_size(const lf_dist_router *r);
int64_t lf_dist_router_last_launches(const lf_dist_router *r);
int lf_dist_router_to_engine_order(lf_dist_router *r, const double *src_pix_dev, double *dst_ord_dev);
int lf_dist_router_from_engine_order(lf_dist_router *r, const double *src_ord_dev, double *dst_pix_dev);
/* one kinematicWaveRouting call; q_ord_dev has lf_dist_router_state_size entries (local cells in engine
 * order followed by the ghost slots), lat_ord_dev the N local cells in engine order.  rank_top /
 * rank_bottom = -1 at the raster's edge.  Asynchronous on the library stream. */
int lf_dist_router_route(lf_dist_router *r, lf_comm *comm, double *q_ord_dev, const double *lat_ord_dev, int section,
                         int rank_top, int rank_bottom);
/* ncalls calls in a row (lat_ord_dev[s]: lateral inflow of call s), software-pipelined across calls: successive calls
 * alternate between the caller's state vector and a second one of the router, so phase 0 of call s + 1 runs beside the
 * later halo rounds of call s (second stream).  Same result as ncalls calls of lf_dist_router_route, in q_ord_dev.
 * lf_dist_router_compute_part_io is its building block: one part of a phase with the old discharge read from q_in and
 * everything else (new discharge, upstream values, ghost slots) in q_out. */
int lf_dist_router_route_many(lf_dist_router *r, lf_comm *comm, double *q_ord_dev, const double *const *lat_ord_dev, int ncalls,
                              int section, int rank_top, int rank_bottom);
int lf_dist_router_compute_part_io(lf_dist_router *r, const double *q_in_dev, double *q_out_dev, const double *lat_ord_dev,
                                   int section, int phase, int part);
/* One routing.dynamic() sub-step on the partition (= lf_routing_substep on the whole raster): element-wise stages on
 * the rank's own N cells, each router call with its halo exchanges.  engine_order = 1 (the rank's engine order);
 * a->ChanQKin and a->Chan2QKin are state vectors (lf_dist_router_state_size entries), all others have N entries. */
int lf_dist_routing_substep(lf_dist_router *r, lf_comm *comm, const lf_substep_args *a, int rank_top, int rank_bottom);
/* nsteps x routing.dynamic() on the partition (= lf_routing_substeps_fused on the whole raster, bit for bit): a rank
 * sweeps ONE PHASE for ALL sub-steps as a skewed wavefront over (level, sub-step); the router outputs that cross a
 * phase or a rank boundary are kept per sub-step in slabs [slot][sub-step], so there is ONE halo exchange per phase and
 * model step (RCCL Send/Recv of one contiguous block per neighbour and section) instead of one per phase, router call
 * and sub-step.  Vectors as lf_dist_routing_substep; sideflow_stride = 0 or N (one SideflowChanM3 vector per sub-step). */
int lf_dist_routing_substeps_fused(lf_dist_router *r, lf_comm *comm, const lf_substep_args *a, int nsteps,
                                   int64_t sideflow_stride, int rank_top, int rank_bottom);
/* its pieces (other transports, in-process loopback): allocate for nsteps; one phase; where a round's halo sits in the
 * slab of a section -- out = {send offset, send count, recv offset, recv count} in doubles; the RCCL exchange */
int lf_dist_fused_prepare(lf_dist_router *r, const lf_substep_args *a, int nsteps);
/* Several MODEL steps per call on the partition (= lf_routing_model_steps_fused on the whole raster, bit for bit): every phase
 * runs the sub-steps of all n_model_steps model steps as one wavefront and hands over the slabs of all of them in ONE halo
 * block, so the pipeline fill of a phase and the exchange round are paid once per call instead of once per model step.
 * a->SideflowChanM3: one vector per model step, sideflow_model_stride elements apart (0: one for all); a->sumDisDay:
 * [n_model_steps][N], zeroed by the caller.  _phase_model_steps: one phase of it (after lf_dist_fused_prepare for
 * steps_per_model_step * n_model_steps sub-steps), for callers that move the halo themselves. */
int lf_dist_routing_model_steps_fused(lf_dist_router *r, lf_comm *comm, const lf_substep_args *a, int steps_per_model_step,
                                      int n_model_steps, int64_t sideflow_model_stride, int rank_top, int rank_bottom);
int lf_dist_fused_phase_model_steps(lf_dist_router *r, const lf_substep_args *a, int steps_per_model_step, int n_model_steps,
                                    int64_t sideflow_model_stride, int phase);
int lf_dist_fused_phase(lf_dist_router *r, const lf_substep_args *a, int nsteps, int64_t sideflow_stride, int phase);
int lf_dist_fused_halo_block(const lf_dist_router *r, int round, int side, int64_t out[4]);
int lf_dist_fused_slab(const lf_dist_router *r, int section, void **slab_dev);
int lf_dist_fused_exchange(lf_dist_router *r, lf_comm *comm, int round, int split, int rank_top, int rank_bottom);
/* slab slots of the fused path: out[0] = slots, [1..2] first export slot top / bottom, [3..4] first ghost slot top /
 * bottom, [5] first slot of the local cells that feed a later phase; and the tables by position (tests) */
int lf_dist_graph_slab_layout(const lf_dist_graph *g, int64_t out[6]);
int lf_dist_graph_get_fused_tables(const lf_dist_graph *g, int32_t *out_slot, int32_t *ups_idx_f);
/* level blocks of the fused path (per phase, cone by cone): out = {blocks, blocks of more than one level, cones, entries
 * of the cone table}; all 0 when no block holds more than one level (then: one launch per level and sub-step wave) */
int lf_dist_graph_block_stats(const lf_dist_graph *g, int64_t out[4]);
/* the block plan of single router calls on the partition (one plan per stage = phase x {boundary-critical part, bulk};
 * blocks of launch units, each cut into cones of at most 64 cells per unit; see lf_blocks.h for the table layout):
 * sizes = {stages + 1, blocks + 1, rows, entries of the cone table, launch units + 1}; arrays may be NULL (sizes only) */
int lf_dist_graph_get_route_plan(const lf_dist_graph *g, int64_t sizes[5], int32_t *stage_block, int32_t *level,
                                 int32_t *row, int32_t *off, int32_t *cone, int64_t *level_start);
/* the same for the block plan of the fused sub-step path (one plan per phase): sizes = {phases + 1, blocks + 1, rows,
 * entries of the cone table} */
int lf_dist_graph_get_fused_plan(const lf_dist_graph *g, int64_t sizes[4], int32_t *phase_block, int32_t *level,
                                 int32_t *row, int32_t *off, int32_t *cone);
/* the pieces of a call, for transports other than RCCL and for tests */
int lf_dist_router_compute_phase(lf_dist_router *r, double *q_ord_dev, const double *lat_ord_dev, int section,
                                 int phase);
/* one part of a phase: 0 = its boundary-critical cells (the exports of the phase and what drains into them inside the
 * phase), 1 = the rest; the parts are independent, so lf_dist_router_route runs round j's halo exchange on a second
 * stream beside part 1 of phase j (LF_DIST_OVERLAP=0: one stream).  part_range: positions [begin, end) of a part. */
int lf_dist_router_compute_part(lf_dist_router *r, double *q_ord_dev, const double *lat_ord_dev, int section, int phase,
                                int part);
int lf_dist_graph_part_range(const lf_dist_graph *g, int phase, int part, int64_t out[2]);
int lf_dist_router_pack(lf_dist_router *r, const double *q_ord_dev, int round, void *send_ptr[2], int64_t send_count[2]);
int lf_dist_router_recv_slots(const lf_dist_router *r, int round, int64_t slot[2], int64_t count[2]);
int lf_dist_router_exchange(lf_dist_router *r, lf_comm *comm, double *q_ord_dev, int round, int rank_top,
                            int rank_bottom);

#ifdef __cplusplus
}
#endif
#endif /* LISFLOOD_AMD_H */
