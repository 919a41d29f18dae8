// lf_router.hip -- kinematic-wave routing on gfx950: level-ordered implicit sweep with a per-cell
// Newton-Raphson solve.  Replaces kinematicWave.kinematicWaveRouting + kinematicRouting + solve1Pixel
// (kinematic_wave_parallel.py:160-184, kinematic_wave_parallel_tools.py:34-92).
//
// Data layout (HBM, all fp64 unless noted), N = land pixels, positions = sweep order (lf_graph):
//   perm[N]      int32  position -> pixel (gather/scatter to the caller's pixel-order vectors)
//   ups_ptr[N+1] int32  upstream cells of position p are the CONTIGUOUS positions [ups_ptr[p], ups_ptr[p+1])
//   a1[N], a2[N]        alpha*dx/dt for main channel / floodplains, sweep order
//   dx[N]               space delta (only when it is per-pixel), sweep order
//   constant[N]         a*Qold^beta + q*dx, sweep order (written by the prep kernel)
//   qord[N]             new discharge, sweep order (read by the downstream level)
// Because a level is a contiguous range of positions and the upstream cells of consecutive positions
// are consecutive too, every global access of the sweep except the perm-indexed gather/scatter of the
// caller's vectors is a coalesced stream.
//
// Launch structure per call: 1 prep launch over all cells, then one launch per "wide" level
// (> kNarrowMax cells, one cell per lane) and one single-workgroup launch per run of consecutive
// "narrow" levels (the workgroup walks the levels with a barrier between them).  A dependent kernel
// boundary costs ~1.5 us on MI355X, less than any software grid barrier (4-7 us), so wide levels
// are separated by launches, not by in-kernel synchronisation.
#include <algorithm>
#include <cmath>

#include <cstdlib>
#include <cstring>

#include "lf_blocks.h"
#include "lf_structures.h"
#include "lf_sweep.h"

namespace {

__global__ void __launch_bounds__(kBlock) k_gather(int n, const int *__restrict__ perm, const double *__restrict__ src_pix,
                                                   double *__restrict__ dst_ord)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p < n) dst_ord[p] = src_pix[perm[p]];
}

// out[pixel(p)] = sum of w over the upstream cells of p, ascending pixel id (np.bincount order)
// (`linked`: zero-length structure links of lf_graph_create_ex sit behind the last range of a level -- skipped)
__global__ void __launch_bounds__(kBlock) k_upstream_sum(int n, const int *__restrict__ perm,
                                                         const int *__restrict__ ups_ptr, const double *__restrict__ w_pix,
                                                         double *__restrict__ out_pix, const uint8_t *__restrict__ linked)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    double s = 0.0;
    for (int e = ups_ptr[p]; e < ups_ptr[p + 1]; ++e)
        if (!linked || !linked[e]) s += w_pix[perm[e]];
    out_pix[perm[p]] = s;
}

// accuflux: acc[p] = x[p] + sum over upstream acc (upstream first, ascending pixel id, then the cell itself); up to
// kMaxAccu vectors share a sweep (the catchment totals of routing.py:645-691 come four at a time)
constexpr int kMaxAccu = 4;
struct accu_multi {
    const double *x[kMaxAccu];
    double *acc[kMaxAccu];
};

template <int NV>
__global__ void __launch_bounds__(kBlock) k_accu_level(int first, int count, const int *__restrict__ ups_ptr, accu_multi M)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    const int p = first + i;
    const int u0 = ups_ptr[p], u1 = ups_ptr[p + 1];
#pragma unroll
    for (int v = 0; v < NV; ++v) {
        double s = 0.0;
        for (int e = u0; e < u1; ++e) s += M.acc[v][e];
        M.acc[v][p] = s + M.x[v][p];
    }
}

template <int NV>
__global__ void __launch_bounds__(kNarrowBlock) k_accu_narrow(int k0, int k1, const long long *__restrict__ level_start,
                                                              const int *__restrict__ ups_ptr, accu_multi M)
{
    for (int k = k0; k < k1; ++k) {
        const int first = (int)level_start[k], last = (int)level_start[k + 1];
        for (int p = first + (int)threadIdx.x; p < last; p += kNarrowBlock) {
            const int u0 = ups_ptr[p], u1 = ups_ptr[p + 1];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                double s = 0.0;
                for (int e = u0; e < u1; ++e) s += M.acc[v][e];
                M.acc[v][p] = s + M.x[v][p];
            }
        }
        __threadfence_block();
        __syncthreads();
    }
}

// One block of consecutive levels of the router's plan (build_level_blocks, the plan of k_sweep_cones), cone by cone.
// An accumulation level is a handful of additions, far shorter than a trip to memory, so the cone is worked through in
// CHUNKS of LC levels: all operands of a chunk (upstream ranges, x) are requested at once and parked in LDS -- one memory
// latency per chunk instead of one per level --, the levels of the chunk are then added up from LDS alone (the sums
// replace x in place and are the next level's inflow), and the chunk's sums leave with one burst of stores that nothing
// waits for.  CW = cells per level of a cone = threads (64: one wavefront, whose LDS operations complete in order -- no
// barrier).  Same additions in the same order as k_accu_level.
template <int NV, int CW>
__global__ void __launch_bounds__(CW) k_accu_cones(cone_plan_args C, const int *__restrict__ ups_ptr, accu_multi M)
{
    constexpr int LC0 = 49152 / (CW * (8 + 8 * NV)), LC = LC0 < 2 ? 2 : (LC0 > 64 ? 64 : LC0);
    __shared__ int U0[LC][CW], U1[LC][CW];
    __shared__ double X[NV][LC][CW], PREV[NV][CW];
    const int tid = threadIdx.x, nl = C.nl;
    const int *c0 = C.cone + (size_t)blockIdx.x * nl, *c1 = c0 + nl;
    int first_up = 0;
    for (int j0 = 0; j0 < nl; j0 += LC) {
        const int L = nl - j0 < LC ? nl - j0 : LC;
        // ---- every operand of the chunk: independent loads, all in flight together ----
#pragma unroll 8
        for (int jj = 0; jj < L; ++jj) {
            const int p = ld_table(c0, j0 + jj) + tid;
            const bool act = p < ld_table(c1, j0 + jj);
            const int pc = act ? p : 0;
            const int u0 = ups_ptr[pc], u1 = ups_ptr[pc + 1];
            U0[jj][tid] = u0;
            U1[jj][tid] = act ? u1 : u0; // a lane beyond the cone's range: no upstream cells, its sum is never stored
#pragma unroll
            for (int v = 0; v < NV; ++v) X[v][jj][tid] = M.x[v][pc];
        }
        cone_sync<CW>();
        // ---- the levels of the chunk, from LDS ----
        for (int jj = 0; jj < L; ++jj) {
            const int first = ld_table(c0, j0 + jj);
            const int u0 = U0[jj][tid], u1 = U1[jj][tid];
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                double t[8];
                if (j0 + jj == 0) { // from the block before (previous launch)
#pragma unroll
                    for (int k = 0; k < 8; ++k) t[k] = (u0 + k < u1) ? M.acc[v][u0 + k] : 0.0;
                } else {
                    const double *z = jj == 0 ? &PREV[v][0] : &X[v][jj - 1][0];
                    const int base = u0 - first_up;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const bool have = u0 + k < u1;
                        const double w = z[have ? base + k : 0];
                        t[k] = have ? w : 0.0;
                    }
                }
                double sum = 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) sum += t[k];
                const double a = sum + X[v][jj][tid];
                X[v][jj][tid] = a;
            }
            cone_sync<CW>();
            first_up = first;
        }
        // ---- the chunk's sums: one burst of stores; the last level stays behind for the next chunk ----
        for (int jj = 0; jj < L; ++jj) {
            const int p = ld_table(c0, j0 + jj) + tid;
            if (p < ld_table(c1, j0 + jj)) {
#pragma unroll
                for (int v = 0; v < NV; ++v) M.acc[v][p] = X[v][jj][tid];
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) PREV[v][tid] = X[v][L - 1][tid];
        cone_sync<CW>();
    }
}

__global__ void __launch_bounds__(kBlock) k_scatter(int n, const int *__restrict__ perm, const double *__restrict__ src_ord,
                                                    double *__restrict__ dst_pix)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p < n) dst_pix[perm[p]] = src_ord[p];
}

__global__ void __launch_bounds__(kBlock) k_count_nonfinite(long long n, const double *__restrict__ x,
                                                            unsigned long long *count)
{
    long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    unsigned long long local = 0;
    for (; i < n; i += (long long)gridDim.x * kBlock) local += !isfinite(x[i]);
    for (int off = 32; off > 0; off >>= 1) local += __shfl_down(local, off, 64);
    if ((threadIdx.x & 63) == 0 && local) atomicAdd(count, local);
}

struct segment {
    int k0, k1; // levels [k0, k1); wide segments have k1 == k0 + 1
    bool wide;
};

} // namespace

struct lf_router {
    int device = 0;
    lf_device_ctx *ctx = nullptr;
    int64_t N = 0, NL = 0;
    double beta = 0, inv_beta = 0, b_minus_1 = 0, dx_scalar = 0, dt = 0;
    bool has_floodplains = false, dx_per_pixel = false;
    lf_dbuf<unsigned int> derived_ok; // fused sub-steps: flag of k_check_derived (fused_args::recompute)
    int kmax = 8;
    bool fused = false; // beta == 3/5: prep fused into the sweep, polynomial closure solve (lf_math.h)
    lf_dbuf<int32_t> perm, ups_ptr;

    lf_dbuf<long long> level_start;
    lf_dbuf<double> a1, a2, dx, constant, qord, io_q, io_lat, tmp_ord, fused_qr1, fused_qr2;
    lf_dbuf<double> fused_hist1, fused_hist2; // [nsteps][N] router outputs of every sub-step (k_fused_level_steps)
    size_t fused_hist_refused = SIZE_MAX;     // smallest history size that did not fit its budget (lf_history_ensure)
    lf_dbuf<unsigned long long> counter;
    lf_dbuf<uint8_t> linked; // zero-length structure links (lf_graph_create_ex); null without them
    lf_dbuf<int> level_nlinked; // ... and how many of them are parked at the end of every level (k_fused_cones_split)
    lf_dbuf<int32_t> parent; // [N] downstream position of every position, -1 = outlet (lf_ldd.hip builds it on demand)
    lf_dbuf<int32_t> root;   // [N] position of the outlet every position drains to (lf_ldd.hip, on demand: catchment totals)
    lf_dbuf<double> totals_scratch; // catchment totals: engine-order copies of the weights and their accuflux
    lf_dbuf<uint8_t> isolated; // [N] by position: 1 = no upstream and no downstream cell (e.g. non-channel land pixels)
    lf_dbuf<uint8_t> inert;    // [N] per fused call: isolated, not a channel, zero split-routing thresholds
    int64_t n_isolated = 0;
    lf_dbuf<int> site_level; // levels of the lake and reservoir cells of the last fused-with-structures call
    std::vector<int> site_level_sorted;
    const void *site_key[4] = {nullptr, nullptr, nullptr, nullptr}; // the site lists those levels were checked for
    int64_t site_cnt[2] = {-1, -1};
    std::vector<int64_t> h_level_start;
    std::vector<segment> schedule;
    // level blocks of the fused sub-step wavefront (build_level_blocks): block b = levels [fb_level[b], fb_level[b+1]),
    // cut into cones = the upstream ranges of chunks of its last level, no range wider than a workgroup; block b has
    // fb_row[b+1] - fb_row[b] - 1 cones and one more row (the end of every level) in fb_cone, from entry fb_off[b] on,
    // one start per level and row
    std::vector<int> fb_level, fb_row, fb_off;
    uint64_t graph_serial = 0; // lf_graph::serial of the graph the router was built on: same object <=> same plan
    lf_dbuf<int> fb_level_dev, fb_row_dev, fb_cone;
    lf_dbuf<int> fb_off_dev, fb_lvl2blk_dev; // fb_lvl2blk: block of every level (the sites of the structures variant)
    std::vector<int> fb_lvl2blk;
    int fb_lmax = 0, fb_cw = kBlock;
    // the same plan with longer blocks for plain router calls (k_sweep_cones: no sub-step dimension to fill the machine
    // with, so fewer, longer launches pay): host tables + the cone starts on the device
    std::vector<int> rb_level, rb_row, rb_off;
    // the static vectors of a cell as one record per section, for the wide levels of ordered beta = 3/5 calls (k_level<.., STATICS>):
    // built on the first such call; statics_refused: the allocation failed once, the separate streams stay
    lf_dbuf<double2> adx1, adx2;
    bool statics_refused = false;
    lf_dbuf<int> rb_cone;
    int rb_lmax = 0, rb_cw = kBlock; // levels per block, cells per level of a cone (LF_ROUTE_CONE_WIDTH: 64 or 256)
    int64_t last_stats[4] = {0, 0, 0, 0};
    // profiling
    bool profile = false;
    std::vector<hipEvent_t> ev_pool;
    struct rec {
        int cls;
        size_t e0, e1;
        int64_t cells;
    };
    std::vector<rec> recs;
    size_t ev_used = 0;
    double prof_acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    ~lf_router()
    {
        for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    }
    int ev_get(size_t *idx)
    {
        if (ev_used == ev_pool.size()) {
            hipEvent_t e;
            LF_HIP(hipEventCreate(&e));
            ev_pool.push_back(e);
        }
        *idx = ev_used++;
        return LF_OK;
    }
    int prof_begin(int cls, int64_t cells)
    {
        if (!profile) return LF_OK;
        rec r{cls, 0, 0, cells};
        LF_TRY(ev_get(&r.e0));
        LF_TRY(ev_get(&r.e1));
        LF_HIP(hipEventRecord(ev_pool[r.e0], ctx->stream));
        recs.push_back(r);
        return LF_OK;
    }
    int prof_end()
    {
        if (!profile) return LF_OK;
        LF_HIP(hipEventRecord(ev_pool[recs.back().e1], ctx->stream));
        return LF_OK;
    }
    int prof_collect()
    {
        if (recs.empty()) return LF_OK;
        LF_HIP(hipStreamSynchronize(ctx->stream));
        for (const rec &r : recs) {
            float ms = 0.f;
            LF_HIP(hipEventElapsedTime(&ms, ev_pool[r.e0], ev_pool[r.e1]));
            prof_acc[3 * r.cls + 0] += 1.0;
            prof_acc[3 * r.cls + 1] += (double)ms;
            prof_acc[3 * r.cls + 2] += (double)r.cells;
        }
        recs.clear();
        ev_used = 0;
        return LF_OK;
    }
};

namespace {

sweep_args make_sweep_args(lf_router *r, double *q_dev, const double *lat_dev, int section, bool ordered)
{
    sweep_args A;
    A.ups_ptr = r->ups_ptr.p;
    A.ups_idx = nullptr;
    A.ups_base = nullptr;
    A.perm = r->perm.p;
    A.a = (section == LF_SECTION_MAIN) ? r->a1.p : r->a2.p;
    A.constant = r->constant.p;
    A.lat = lat_dev;
    A.dx = r->dx_per_pixel ? r->dx.p : nullptr;
    A.dx_scalar = r->dx_scalar;
    A.beta = r->beta;
    A.inv_beta = r->inv_beta;
    A.b_minus_1 = r->b_minus_1;
    A.kmax = r->kmax;
    A.qord = ordered ? q_dev : r->qord.p;
    A.q_pix = ordered ? nullptr : q_dev;
    A.qold_src = nullptr;
    A.adx = nullptr;
    A.rec24 = nullptr;
    return A;
}

// The (a, dx) records of A's section for the wide levels of an ordered beta = 3/5 call, or nullptr: dx is a scalar (the
// sweep then has one static load anyway), LF_LEVEL_STATICS=0 (A/B switch, read at every call) or no memory for them
// (16 bytes per cell and section).
const double2 *level_statics(lf_router *r, const sweep_args &A)
{
    if (!r->fused || !r->dx_per_pixel || r->statics_refused || r->N <= 0) return nullptr;
    const char *e = std::getenv("LF_LEVEL_STATICS");
    if (e && e[0] == '0') return nullptr;
    lf_dbuf<double2> &buf = (A.a == r->a1.p) ? r->adx1 : r->adx2;
    if (!buf.p) {
        if (buf.alloc((size_t)r->N) != LF_OK) {
            r->statics_refused = true;
            (void)hipGetLastError();
            return nullptr;
        }
        hipLaunchKernelGGL(k_static_records, dim3((unsigned)((r->N + kLevelBlock - 1) / kLevelBlock)), dim3(kLevelBlock), 0, r->ctx->stream,
                           (long long)r->N, A.a, (const double *)r->dx.p, buf.p);
    }
    return buf.p;
}

// one wide level of an ORDERED beta = 3/5 call
void launch_level_ordered_fused(lf_router *r, hipStream_t s, int first, int cells, sweep_args A)
{
    const dim3 grid(level_blocks_for(cells)), block(kLevelBlock);
    A.adx = level_statics(r, A);
    if (A.adx)
        hipLaunchKernelGGL((k_level<true, true, false, 1>), grid, block, 0, s, first, cells, A);
    else
        hipLaunchKernelGGL((k_level<true, true>), grid, block, 0, s, first, cells, A);
}

// A router call block by block (build_level_blocks): blocks of several levels cone by cone (k_sweep_cones), single wide
// levels by the level kernel.  `count` routers of ONE graph share the launches.  LF_ROUTE_CONES=0: level by level.
bool cones_enabled() // (read at every call: bench.py switches it for its A/B legs)
{
    const char *e = std::getenv("LF_ROUTE_CONES");
    return !(e && e[0] == '0');
}

template <int NR, int CW>
void launch_sweep_cones_cw(bool fused, bool ordered, dim3 grid, hipStream_t s, const cone_plan_args &C,
                           const sweep_args_multi &M)
{
    if (fused && ordered)
        hipLaunchKernelGGL((k_sweep_cones<true, true, NR, CW>), grid, dim3(CW), 0, s, C, M);
    else if (fused)
        hipLaunchKernelGGL((k_sweep_cones<true, false, NR, CW>), grid, dim3(CW), 0, s, C, M);
    else if (ordered)
        hipLaunchKernelGGL((k_sweep_cones<false, true, NR, CW>), grid, dim3(CW), 0, s, C, M);
    else
        hipLaunchKernelGGL((k_sweep_cones<false, false, NR, CW>), grid, dim3(CW), 0, s, C, M);
}
bool cone_split_enabled() // LF_ROUTE_SPLIT=0: one wavefront per cone does everything (the round-3 kernel; A/B switch)
{
    const char *e = std::getenv("LF_ROUTE_SPLIT");
    return !(e && e[0] == '0');
}
// k_sweep_cones_split comes in two shapes for a single router.  FEW cones per launch (at most about one per CU: the
// chain-bound networks) -> chunks of 8 levels and four supply wavefronts, 72 KB of LDS per cone; MANY cones (several per
// CU: their wavefronts share the SIMDs, the launch is throughput-bound) -> chunks of 4 levels and two supply wavefronts,
// 37 KB.  Several routers on one graph: chunks of 4 levels, two supply wavefronts.  LF_ROUTE_SPLIT_SHAPE=few|many forces
// one shape (A/B switch); LF_ROUTE_SPLIT_FEW is the threshold in cones per launch.
template <bool FUSED, bool ORDERED, int NR, int KC, int NS>
void launch_split(dim3 grid, hipStream_t s, const cone_plan_args &C, const sweep_args_multi &M)
{
    hipLaunchKernelGGL((k_sweep_cones_split<FUSED, ORDERED, NR, KC, NS>), grid, dim3(64 * (1 + NS)), 0, s, C, M);
}
template <bool FUSED, bool ORDERED, int NR>
void launch_split_shape(dim3 grid, hipStream_t s, const cone_plan_args &C, const sweep_args_multi &M)
{
    if (NR > 1) {
        launch_split<FUSED, ORDERED, NR, 4, 2>(grid, s, C, M);
        return;
    }
    static const int few_limit = [] {
        const char *e = std::getenv("LF_ROUTE_SPLIT_FEW");
        return e ? std::atoi(e) : 256;
    }();
    const char *shape = std::getenv("LF_ROUTE_SPLIT_SHAPE"); // (read at every call: A/B legs switch it)
    const bool few = shape ? shape[0] == 'f' : (int)grid.x <= few_limit;
    if (few)
        launch_split<FUSED, ORDERED, 1, LF_CONE_KC, LF_CONE_NS>(grid, s, C, M);
    else
        launch_split<FUSED, ORDERED, 1, 4, 2>(grid, s, C, M);
}
template <int NR>
void launch_sweep_cones(int cw, bool fused, bool ordered, dim3 grid, hipStream_t s, const cone_plan_args &C,
                        const sweep_args_multi &M)
{
    if (cw == 64 && cone_split_enabled() && C.n_cells < (1 << 29)) { // (byte offsets of the buffer stores: 32 bits)
        if (fused && ordered)
            launch_split_shape<true, true, NR>(grid, s, C, M);
        else if (fused)
            launch_split_shape<true, false, NR>(grid, s, C, M);
        else if (ordered)
            launch_split_shape<false, true, NR>(grid, s, C, M);
        else
            launch_split_shape<false, false, NR>(grid, s, C, M);
    } else if (cw == 64)
        launch_sweep_cones_cw<NR, 64>(fused, ordered, grid, s, C, M);
    else
        launch_sweep_cones_cw<NR, kBlock>(fused, ordered, grid, s, C, M);
}

int enqueue_blocks(int count, lf_router **rs, const sweep_args_multi &M, bool ordered, int64_t *launches, int64_t *wide,
                   int64_t *narrow)
{
    lf_router *r = rs[0];
    hipStream_t s = r->ctx->stream;
    const int NB = (int)r->rb_level.size() - 1;
    for (int b = 0; b < NB; ++b) {
        const int k0 = r->rb_level[b], nl = r->rb_level[b + 1] - k0;
        if (nl > 1) {
            cone_plan_args C;
            C.cone = r->rb_cone.p + r->rb_off[b];
            C.nl = nl;
            C.n_cells = (int)r->N;
            const dim3 grid((unsigned)(r->rb_row[b + 1] - r->rb_row[b] - 1));
            LF_TRY(r->prof_begin(2, r->h_level_start[k0 + nl] - r->h_level_start[k0]));
            if (count == 1)
                launch_sweep_cones<1>(r->rb_cw, r->fused, ordered, grid, s, C, M);
            else if (count == 2)
                launch_sweep_cones<2>(r->rb_cw, r->fused, ordered, grid, s, C, M);
            else if (count == 3)
                launch_sweep_cones<3>(r->rb_cw, r->fused, ordered, grid, s, C, M);
            else
                launch_sweep_cones<4>(r->rb_cw, r->fused, ordered, grid, s, C, M);
            LF_TRY(r->prof_end());
            ++*narrow;
        } else {
            const int first = (int)r->h_level_start[k0];
            const int cells = (int)(r->h_level_start[k0 + 1] - r->h_level_start[k0]);
            const dim3 grid(blocks_for(cells), count), block(kBlock);
            const dim3 grid1(level_blocks_for(cells)), block1(kLevelBlock);
            LF_TRY(r->prof_begin(1, cells));
            if (count == 1) {
                if (r->fused && ordered)
                    launch_level_ordered_fused(r, s, first, cells, M.r[0]);
                else if (r->fused)
                    hipLaunchKernelGGL((k_level<true, false>), grid1, block1, 0, s, first, cells, M.r[0]);
                else if (ordered)
                    hipLaunchKernelGGL((k_level<false, true>), grid1, block1, 0, s, first, cells, M.r[0]);
                else
                    hipLaunchKernelGGL((k_level<false, false>), grid1, block1, 0, s, first, cells, M.r[0]);
            } else if (r->fused && ordered)
                hipLaunchKernelGGL((k_level_multi<true, true>), grid, block, 0, s, first, cells, M);
            else if (r->fused)
                hipLaunchKernelGGL((k_level_multi<true, false>), grid, block, 0, s, first, cells, M);
            else if (ordered)
                hipLaunchKernelGGL((k_level_multi<false, true>), grid, block, 0, s, first, cells, M);
            else
                hipLaunchKernelGGL((k_level_multi<false, false>), grid, block, 0, s, first, cells, M);
            LF_TRY(r->prof_end());
            ++*wide;
        }
        ++*launches;
    }
    return LF_OK;
}

// `count` routers built on the same graph (same level schedule), swept level by level with ONE launch per level
int enqueue_route_multi(int count, lf_router **rs, double **q_dev, const double **lat_dev, int section, bool ordered)
{
    lf_router *r = rs[0];
    hipStream_t s = r->ctx->stream;
    const int n = (int)r->N;
    sweep_args_multi M;
    for (int i = 0; i < kMaxMulti; ++i) M.r[i] = make_sweep_args(rs[i < count ? i : 0], q_dev[i < count ? i : 0], lat_dev[i < count ? i : 0], section, ordered);
    int64_t launches = 0, wide = 0, narrow = 0;
    if (n > 0 && !r->fused)
        for (int i = 0; i < count; ++i) {
            hipLaunchKernelGGL(k_prep, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, ordered ? nullptr : rs[i]->perm.p, q_dev[i],
                               lat_dev[i], M.r[i].a, M.r[i].dx, rs[i]->dx_scalar, rs[i]->beta, rs[i]->constant.p);
            ++launches;
        }
    bool same_graph = r->rb_lmax > 1 && cones_enabled();
    for (int i = 1; i < count; ++i) same_graph = same_graph && rs[i]->graph_serial == r->graph_serial && r->graph_serial != 0 && rs[i]->rb_lmax == r->rb_lmax && rs[i]->rb_cw == r->rb_cw;
    if (same_graph) {
        LF_TRY(enqueue_blocks(count, rs, M, ordered, &launches, &wide, &narrow));
        for (int i = 0; i < count; ++i) {
            rs[i]->last_stats[0] = launches;
            rs[i]->last_stats[1] = wide;
            rs[i]->last_stats[2] = narrow;
            rs[i]->last_stats[3] = rs[i]->NL;
        }
        return LF_OK;
    }
    for (const segment &g : r->schedule) {
        if (g.wide) {
            const int first = (int)r->h_level_start[g.k0];
            const int cells = (int)(r->h_level_start[g.k1] - r->h_level_start[g.k0]);
            const dim3 grid(blocks_for(cells), count), block(kBlock);
            if (r->fused && ordered)
                hipLaunchKernelGGL((k_level_multi<true, true>), grid, block, 0, s, first, cells, M);
            else if (r->fused)
                hipLaunchKernelGGL((k_level_multi<true, false>), grid, block, 0, s, first, cells, M);
            else if (ordered)
                hipLaunchKernelGGL((k_level_multi<false, true>), grid, block, 0, s, first, cells, M);
            else
                hipLaunchKernelGGL((k_level_multi<false, false>), grid, block, 0, s, first, cells, M);
            ++wide;
        } else {
            const dim3 grid(count), block(kNarrowBlock);
            if (r->fused && ordered)
                hipLaunchKernelGGL((k_levels_narrow_multi<true, true>), grid, block, 0, s, g.k0, g.k1, r->level_start.p, M);
            else if (r->fused)
                hipLaunchKernelGGL((k_levels_narrow_multi<true, false>), grid, block, 0, s, g.k0, g.k1, r->level_start.p, M);
            else if (ordered)
                hipLaunchKernelGGL((k_levels_narrow_multi<false, true>), grid, block, 0, s, g.k0, g.k1, r->level_start.p, M);
            else
                hipLaunchKernelGGL((k_levels_narrow_multi<false, false>), grid, block, 0, s, g.k0, g.k1, r->level_start.p, M);
            ++narrow;
        }
        ++launches;
    }
    for (int i = 0; i < count; ++i) {
        rs[i]->last_stats[0] = launches;
        rs[i]->last_stats[1] = wide;
        rs[i]->last_stats[2] = narrow;
        rs[i]->last_stats[3] = rs[i]->NL;
    }
    return LF_OK;
}

int enqueue_route(lf_router *r, double *q_dev, const double *lat_dev, int section, bool ordered)
{
    hipStream_t s = r->ctx->stream;
    const int n = (int)r->N;
    int64_t launches = 0, wide = 0, narrow = 0;
    sweep_args A = make_sweep_args(r, q_dev, lat_dev, section, ordered);
    const double *a = A.a;
    if (n > 0 && !r->fused) {
        LF_TRY(r->prof_begin(0, n));
        hipLaunchKernelGGL(k_prep, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, ordered ? nullptr : r->perm.p, q_dev,
                           lat_dev, a, A.dx, r->dx_scalar, r->beta, r->constant.p);
        LF_TRY(r->prof_end());
        ++launches;
    }
    if (r->rb_lmax > 1 && cones_enabled()) {
        sweep_args_multi M;
        for (int i = 0; i < kMaxMulti; ++i) M.r[i] = A;
        lf_router *one[1] = {r};
        LF_TRY(enqueue_blocks(1, one, M, ordered, &launches, &wide, &narrow));
        r->last_stats[0] = launches;
        r->last_stats[1] = wide;
        r->last_stats[2] = narrow;
        r->last_stats[3] = r->NL;
        return LF_OK;
    }
    for (const segment &g : r->schedule) {
        if (g.wide) {
            const int first = (int)r->h_level_start[g.k0];
            const int count = (int)(r->h_level_start[g.k1] - r->h_level_start[g.k0]);
            LF_TRY(r->prof_begin(1, count));
            const dim3 grid(level_blocks_for(count)), block(kLevelBlock);
            if (r->fused && ordered)
                launch_level_ordered_fused(r, s, first, count, A);
            else if (r->fused)
                hipLaunchKernelGGL((k_level<true, false>), grid, block, 0, s, first, count, A);
            else if (ordered)
                hipLaunchKernelGGL((k_level<false, true>), grid, block, 0, s, first, count, A);
            else
                hipLaunchKernelGGL((k_level<false, false>), grid, block, 0, s, first, count, A);
            LF_TRY(r->prof_end());
            ++wide;
        } else {
            LF_TRY(r->prof_begin(2, r->h_level_start[g.k1] - r->h_level_start[g.k0]));
            const dim3 grid(1), block(kNarrowBlock);
            if (r->fused && ordered)
                hipLaunchKernelGGL((k_levels_narrow<true, true>), grid, block, 0, s, g.k0, g.k1, r->level_start.p, A);
            else if (r->fused)
                hipLaunchKernelGGL((k_levels_narrow<true, false>), grid, block, 0, s, g.k0, g.k1, r->level_start.p, A);
            else if (ordered)
                hipLaunchKernelGGL((k_levels_narrow<false, true>), grid, block, 0, s, g.k0, g.k1, r->level_start.p, A);
            else
                hipLaunchKernelGGL((k_levels_narrow<false, false>), grid, block, 0, s, g.k0, g.k1, r->level_start.p,
                                   A);
            LF_TRY(r->prof_end());
            ++narrow;
        }
        ++launches;
    }
    r->last_stats[0] = launches;
    r->last_stats[1] = wide;
    r->last_stats[2] = narrow;
    r->last_stats[3] = r->NL;
    return LF_OK;
}

int route_device(lf_router *r, double *q_dev, const double *lat_dev, int section, bool ordered = false)
{
    if (section != LF_SECTION_MAIN && section != LF_SECTION_FLOODPLAINS)
        return lf_set_error(LF_E_SECTION, "The section parameter must be either 'main_channel' or 'floodplain'!");
    if (section == LF_SECTION_FLOODPLAINS && !r->has_floodplains)
        return lf_set_error(LF_E_SECTION, "floodplains routing requested but alpha_floodplains was not given");
    if (r->linked.p)
        return lf_set_error(LF_E_INVALID, "a router on a graph with structure links (lf_graph_create_ex) runs only the "
                            "fused sub-step path (lf_routing_substeps_fused*)");
    LF_HIP(hipSetDevice(r->device));
    LF_TRY(enqueue_route(r, q_dev, lat_dev, section, ordered));
    LF_HIP(hipGetLastError());
    if (r->profile) LF_TRY(r->prof_collect());
    return LF_OK;
}

} // namespace

// Level blocks for the fused sub-step wavefront (k_fused_cones): runs of consecutive levels of at most `wide` cells are
// cut into blocks of up to lmax levels (fused wavefront: LF_FUSED_LEVELS, default 16; plain router calls, k_sweep_cones:
// LF_ROUTE_LEVELS, default 64; 1 = off); a wider level is a block of its own.  A
// block is cut into cones: chunks of its last level, as long as possible with no level of the cone wider than kBlock
// cells; a block whose thinnest possible cone (one cell of the last level) is still too wide somewhere loses levels
// until it fits (one level always does).  Nothing is built when no block holds more than one level.
static int build_level_blocks(lf_router *r, const lf_graph *g, bool for_route, int lmax_override = 0)
{
    int lmax = for_route ? 256 : 16; // LF_ROUTE_LEVELS / LF_FUSED_LEVELS (measured: §4.1c / §4.3b of DESIGN.md)
    if (!for_route && !g->has_links) {
        // graphs whose launches all stay chain-bound (the chain / supply cone kernel, k_fused_cones_split, runs them): blocks
        // of 32 levels halve the pipeline fill of that kernel (deep 2000^2: 8.5 -> 7.7 ms per model step); k_fused_cones,
        // which takes the launches of wider graphs, is slower on them (44 vs 31 ms at 5000^2) and keeps 16
        int64_t widest = 0;
        for (int64_t k = 0; k < g->NL; ++k) widest = std::max(widest, g->level_start[k + 1] - g->level_start[k]);
        if (widest <= 3000) lmax = 32;
    }
    // a catchment of a few thousand cells (LF_ETRS89: 2 847 cells, 113 levels): every launch is its own latency, a cone's chain
    // of levels is what it costs, and 8 levels per block balance chain against launch count (model step with structures, ms:
    // 2.29 / 1.96 / 1.96 / 2.20 / 2.96 for 2 / 4 / 8 / 16 / 32 levels per block)
    if (!for_route && g->N <= 65536) lmax = 8;
    if (const char *e = std::getenv(for_route ? "LF_ROUTE_LEVELS" : "LF_FUSED_LEVELS")) lmax = std::atoi(e);
    if (lmax_override > 0) lmax = lmax_override;
    lmax = lmax < 1 ? 1 : (lmax > (for_route ? 512 : 64) ? (for_route ? 512 : 64) : lmax);
    int64_t wide = 262144;
    if (const char *e = std::getenv("LF_FUSED_WIDE")) wide = std::atoll(e);
    const int64_t NL = g->NL;
    // (the fused cone kernels address a cell by a 32-bit byte offset into its arrays: below 2^29 cells; a larger graph keeps
    // the level-by-level wavefront)
    if (lmax <= 1 || NL < 2 || g->N >= ((int64_t)1 << (for_route ? 31 : 29))) return LF_OK;
    int cw = kBlock;
    if (for_route) { // one wavefront per cone: no barrier between the levels (deep 10 000^2: 11.3 -> 10.1 ms per call)
        cw = 64;
        if (const char *e = std::getenv("LF_ROUTE_CONE_WIDTH")) cw = std::atoi(e) == 64 ? 64 : kBlock;
    } else {
        // fused wavefront: one wavefront per cone (the chain / supply kernel needs it; k_fused_cones itself measured the
        // same at 64 and 256: DESIGN.md section 8b)
        cw = 64;
        if (const char *e = std::getenv("LF_FUSED_CONE_WIDTH")) cw = std::atoi(e) == 64 ? 64 : kBlock;
    }
    lf_block_plan plan;
    try {
        // every cell below the last level drains into the next level, so the upstream ranges tile the level before:
        // the first position draining at or behind `pos` is the first upstream position of `pos`
        lf_build_level_blocks(g->level_start, 0, NL, lmax, wide, cw, [&](int64_t pos) { return (int64_t)g->ups_ptr[pos]; },
                              plan);
    } catch (const std::bad_alloc &) {
        return lf_set_error(LF_E_INVALID, "out of host memory while building the level blocks");
    }
    const bool any = plan.any_multi;
    std::vector<int> &level = plan.level, &row = plan.row, &cone = plan.cone, &off = plan.off;
    if (!any || cone.size() >= ((size_t)1 << 31)) return LF_OK;
    level.push_back((int)NL);
    if (for_route) {
        LF_TRY(r->rb_cone.upload(cone.data(), cone.size(), r->ctx->stream));
        r->rb_level = level;
        r->rb_row = row;
        r->rb_off = off;
        r->rb_lmax = lmax;
        r->rb_cw = cw;
        return LF_OK;
    }
    LF_TRY(r->fb_level_dev.upload(level.data(), level.size(), r->ctx->stream));
    LF_TRY(r->fb_row_dev.upload(row.data(), row.size(), r->ctx->stream));
    LF_TRY(r->fb_off_dev.upload(off.data(), off.size(), r->ctx->stream));
    LF_TRY(r->fb_cone.upload(cone.data(), cone.size(), r->ctx->stream));
    std::vector<int> lvl2blk((size_t)NL, 0);
    for (size_t b = 0; b + 1 < level.size(); ++b)
        for (int k = level[b]; k < level[b + 1]; ++k) lvl2blk[k] = (int)b;
    LF_TRY(r->fb_lvl2blk_dev.upload(lvl2blk.data(), lvl2blk.size(), r->ctx->stream));
    r->fb_lvl2blk = lvl2blk;
    r->fb_level = level;
    r->fb_row = row;
    r->fb_off = off;
    r->fb_lmax = lmax;
    r->fb_cw = cw;
    return LF_OK;
}

// ---- levels per block of plain router calls, chosen per graph ------------------------------------------------------------
// Blocks of 256 levels are right for graphs whose cone launches are CHAIN-bound (deep 10 000^2: 157 cones per launch, every
// launch costs its fill + 256 level times whatever the block length, so fewer launches win).  A graph of many short trees --
// the overland graph of a domain with few channel pixels: 4 223 cones of ~100 levels whose ranges are 64 cells wide at one
// level and ~10 on average -- is THROUGHPUT-bound on lanes that idle: a cone must fit its widest level, so the longer the
// block, the thinner the rest of the cone.  Shorter blocks re-cut the cones where the graph narrows (overland 4000^2, 4 %
// channel pixels: 1.43 ms with 256 levels per block, 1.08 / 0.90 / 0.86 / 0.90 / 1.06 with 128 / 64 / 32 / 16 / 8).  Which
// length is best depends on the width profile, so it is MEASURED: when the default plan fills less than 40 % of its lanes
// and has more cones in a launch than the chip holds at once, the candidates are built and timed on scratch vectors (three
// router calls each, hipEvents) and the fastest stays.  Routers of one graph share the result (they must: swept together
// they use one plan).  The plan does not change a single bit of the results (every routing test runs on whatever it picks).
// LF_ROUTE_LEVELS=n fixes the length, LF_ROUTE_TUNE=0 keeps the default.
__global__ void __launch_bounds__(kBlock) k_fill_f64(long long n, double *x, double v)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) x[i] = v;
}

static int tune_route_blocks(lf_router *r, const lf_graph *g)
{
    static std::map<uint64_t, int> tuned; // graph serial -> levels per block (one thread per device context drives this)
    if (std::getenv("LF_ROUTE_LEVELS")) return LF_OK;
    if (const char *e = std::getenv("LF_ROUTE_TUNE"))
        if (e[0] == '0') return LF_OK;
    if (r->rb_lmax <= 1 || r->N < 1000000) return LF_OK;
    int64_t st[6];
    LF_TRY(lf_router_route_plan_stats(r, st));
    const double lane_use = st[3] > 0 ? (double)st[4] / ((double)r->rb_cw * (double)st[3]) : 1.0;
    if (lane_use >= 0.4 || st[5] <= 1024) return LF_OK; // lanes busy, or few enough cones per launch to be chain-bound
    auto it = tuned.find(g->serial);
    if (g->serial != 0 && it != tuned.end()) return it->second == r->rb_lmax ? LF_OK : build_level_blocks(r, g, true, it->second);
    lf_dbuf<double> Q, q;
    LF_TRY(Q.alloc((size_t)r->N));
    LF_TRY(q.alloc((size_t)r->N));
    hipStream_t s = r->ctx->stream;
    hipEvent_t e0, e1;
    LF_HIP(hipEventCreate(&e0));
    LF_HIP(hipEventCreate(&e1));
    int best = r->rb_lmax, rc = LF_OK;
    float best_ms = 1e30f;
    for (int lmax : {256, 128, 64, 32, 16}) {
        if (lmax != r->rb_lmax) rc = build_level_blocks(r, g, true, lmax);
        if (rc != LF_OK || r->rb_lmax != lmax) break; // (no multi-level block at this length: nothing shorter will have one)
        hipLaunchKernelGGL(k_fill_f64, dim3(blocks_for(r->N)), dim3(kBlock), 0, s, (long long)r->N, Q.p, 1.0);
        hipLaunchKernelGGL(k_fill_f64, dim3(blocks_for(r->N)), dim3(kBlock), 0, s, (long long)r->N, q.p, 1.0e-4);
        rc = route_device(r, Q.p, q.p, 0, true); // warm
        if (rc != LF_OK) break;
        (void)hipEventRecord(e0, s);
        for (int k = 0; k < 3 && rc == LF_OK; ++k) rc = route_device(r, Q.p, q.p, 0, true);
        (void)hipEventRecord(e1, s);
        if (rc != LF_OK || hipEventSynchronize(e1) != hipSuccess) break;
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        if (ms < best_ms) {
            best_ms = ms;
            best = lmax;
        } else if (ms > 1.15f * best_ms)
            break; // past the minimum
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc == LF_OK && r->rb_lmax != best) rc = build_level_blocks(r, g, true, best);
    if (rc == LF_OK && g->serial != 0) tuned[g->serial] = best;
    return rc;
}

extern "C" {

int lf_router_create(const lf_graph *g, const double *alpha, double beta, const double *dx, double dx_scalar, double dt,
                     const double *alpha_floodplains, int device, lf_router **out)
{
    if (!g || !alpha || !out) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *ctx;
    LF_TRY(lf_ctx(device, &ctx));
    lf_router *r = new lf_router();
    r->device = device;
    r->ctx = ctx;
    r->N = g->N;
    r->NL = g->NL;
    r->graph_serial = g->serial; // routers swept together (cone plan and upstream ranges of router 0) must share the graph
    r->kmax = g->K;
    r->beta = beta;
    r->inv_beta = 1 / beta;      // kinematic_wave_parallel.py:125
    r->b_minus_1 = beta - 1;     // :126
    r->dx_scalar = dx_scalar;
    r->dt = dt;
    r->dx_per_pixel = dx != nullptr;
    r->has_floodplains = alpha_floodplains != nullptr;
    // beta == 3/5 (every LISFLOOD setting): fused prep + polynomial solve.  LF_GENERAL_POW=1 forces the
    // general path (the reference's own Newton iteration with pow) for A/B parity and timing.
    const char *force_general = std::getenv("LF_GENERAL_POW");
    r->fused = (beta == 0.6) && !(force_general && force_general[0] == '1');
    const int64_t n = g->N;
    int rc = LF_OK;
    const std::vector<int32_t> &g_perm = g->perm;
    const std::vector<int32_t> &g_ups_ptr = g->ups_ptr;
    {
        // a_dx_div_dt = alpha * dx / dt, evaluated left to right (:127), permuted into sweep order
        std::vector<double> h(n);
        auto fill = [&](const double *al) {
            for (int64_t p = 0; p < n; ++p) {
                const int32_t pix = g_perm[p];
                h[p] = al[pix] * (dx ? dx[pix] : dx_scalar) / dt;
            }
        };
        fill(alpha);
        rc = r->a1.upload(h.data(), n, ctx->stream);
        if (rc == LF_OK && alpha_floodplains) {
            fill(alpha_floodplains);
            rc = r->a2.upload(h.data(), n, ctx->stream);
        }
        if (rc == LF_OK && dx) {
            for (int64_t p = 0; p < n; ++p) h[p] = dx[g_perm[p]];
            rc = r->dx.upload(h.data(), n, ctx->stream);
        }
    }
    if (rc == LF_OK) rc = r->perm.upload(g_perm.data(), n, ctx->stream);
    if (rc == LF_OK) rc = r->ups_ptr.upload(g_ups_ptr.data(), n + 1, ctx->stream);
    if (rc == LF_OK && g->has_links) rc = r->linked.upload(g->linked.data(), n, ctx->stream);
    if (rc == LF_OK && g->has_links) {
        std::vector<int> parked((size_t)g->NL, 0);
        for (int64_t k = 0; k < g->NL; ++k)
            for (int64_t p = g->level_start[k]; p < g->level_start[k + 1]; ++p) parked[k] += g->linked[p] ? 1 : 0;
        rc = r->level_nlinked.upload(parked.data(), parked.size(), ctx->stream);
    }
    if (rc == LF_OK && n > 0) {
        std::vector<uint8_t> has_up(n, 0), iso(n, 0);
        for (int64_t p = 0; p < n; ++p)
            if (g->down[p] >= 0) has_up[g->down[p]] = 1;
        for (int64_t p = 0; p < n; ++p) {
            const int32_t pix = g_perm[p];
            iso[p] = (g->down[pix] < 0 && !has_up[pix] && !(g->has_links && g->linked[p])) ? 1 : 0;
            r->n_isolated += iso[p];
        }
        if (r->n_isolated > 0) rc = r->isolated.upload(iso.data(), n, ctx->stream);
    }
    if (rc == LF_OK) {
        std::vector<long long> ls(g->level_start.begin(), g->level_start.end());
        rc = r->level_start.upload(ls.data(), ls.size(), ctx->stream);
    }
    if (rc == LF_OK && !r->fused) rc = r->constant.alloc(n);
    if (rc == LF_OK) rc = r->qord.alloc(n);
    if (rc == LF_OK) rc = r->counter.alloc(1);
    if (rc != LF_OK) {
        delete r;
        return rc;
    }
    r->h_level_start = g->level_start;
    // launch schedule: runs of narrow levels -> one single-workgroup launch each; wide levels -> one launch per level
    {
        const int64_t NL = g->NL;
        auto level_size = [&](int64_t k) { return g->level_start[k + 1] - g->level_start[k]; };
        for (int64_t k = 0; k < NL;) {
            if (level_size(k) <= kNarrowMax) {
                int64_t e = k + 1;
                while (e < NL && level_size(e) <= kNarrowMax) ++e;
                r->schedule.push_back({(int)k, (int)e, false});
                k = e;
            } else {
                r->schedule.push_back({(int)k, (int)k + 1, true});
                ++k;
            }
        }
    }
    // (zero-length structure links need nothing special: such cells sit at the end of their level, inside the upstream
    // range of the LAST cell of the next level -- which adds their 0.0 -- and so inside the last cone of a block)
    rc = build_level_blocks(r, g, false);
    if (rc == LF_OK && !g->has_links) rc = build_level_blocks(r, g, true);
    if (rc == LF_OK && !g->has_links) rc = tune_route_blocks(r, g);
    if (rc != LF_OK) {
        delete r;
        return rc;
    }
    *out = r;
    return LF_OK;
}

void lf_router_destroy(lf_router *r)
{
    if (!r) return;
    (void)hipSetDevice(r->device);
    (void)hipStreamSynchronize(r->ctx->stream);
    delete r;
}

int lf_router_route_device(lf_router *r, double *discharge_dev, const double *lateral_dev, int section)
{
    if (!r || !discharge_dev || !lateral_dev) return lf_set_error(LF_E_INVALID, "null argument");
    return route_device(r, discharge_dev, lateral_dev, section);
}

// Several routers with the same level schedule (built on one lf_graph: e.g. the three overland routers of
// surface_routing.py:108-113, which differ in alpha only) swept together, one launch per level for all of them.  Routers
// whose schedules differ are swept one after the other.  engine_order != 0: the vectors
// are resident in the routers' sweep order (lf_router_route_ordered), else in pixel order (lf_router_route_device).
int lf_router_route_device_multi(int count, lf_router **routers, double **discharge_dev, const double **lateral_dev,
                                 int section, int engine_order)
{
    if (count < 1 || !routers || !discharge_dev || !lateral_dev) return lf_set_error(LF_E_INVALID, "null argument");
    for (int i = 0; i < count; ++i)
        if (!routers[i] || !discharge_dev[i] || !lateral_dev[i]) return lf_set_error(LF_E_INVALID, "null argument");
    if (section != LF_SECTION_MAIN && section != LF_SECTION_FLOODPLAINS)
        return lf_set_error(LF_E_SECTION, "The section parameter must be either 'main_channel' or 'floodplain'!");
    bool together = count > 1 && count <= kMaxMulti;
    for (int i = 0; i < count && together; ++i) {
        const lf_router *r = routers[i], *r0 = routers[0];
        together = !r->linked.p && r->device == r0->device && r->ctx == r0->ctx && r->N == r0->N &&
                   r->fused == r0->fused && r->h_level_start == r0->h_level_start && !r->profile &&
                   (section == LF_SECTION_MAIN || r->has_floodplains);
    }
    if (!together) {
        for (int i = 0; i < count; ++i)
            LF_TRY(route_device(routers[i], discharge_dev[i], lateral_dev[i], section, engine_order != 0));
        return LF_OK;
    }
    LF_HIP(hipSetDevice(routers[0]->device));
    LF_TRY(enqueue_route_multi(count, routers, discharge_dev, lateral_dev, section, engine_order != 0));
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int lf_router_route_ordered(lf_router *r, double *discharge_ord_dev, const double *lateral_ord_dev, int section)
{
    if (!r || !discharge_ord_dev || !lateral_ord_dev) return lf_set_error(LF_E_INVALID, "null argument");
    return route_device(r, discharge_ord_dev, lateral_ord_dev, section, true);
}

// dst[i] = src[index[i]] for i < n (device vectors; index is int32): the permutation between two domains, e.g. a
// full-raster pixel vector into the engine order of a router that covers only a subset of the pixels
int lf_gather_device(int device, int64_t n, const int32_t *index_dev, const double *src_dev, double *dst_dev)
{
    if (n < 0 || (n > 0 && (!index_dev || !src_dev || !dst_dev))) return lf_set_error(LF_E_INVALID, "bad argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (n > 0)
        hipLaunchKernelGGL(k_gather, dim3(blocks_for(n)), dim3(kBlock), 0, c->stream, (int)n, (const int *)index_dev, src_dev,
                           dst_dev);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int lf_router_to_engine_order(lf_router *r, const double *src_pix_dev, double *dst_ord_dev)
{
    if (!r || !src_pix_dev || !dst_ord_dev) return lf_set_error(LF_E_INVALID, "null argument");
    LF_HIP(hipSetDevice(r->device));
    const int n = (int)r->N;
    if (n > 0)
        hipLaunchKernelGGL(k_gather, dim3(blocks_for(n)), dim3(kBlock), 0, r->ctx->stream, n, r->perm.p, src_pix_dev,
                           dst_ord_dev);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int lf_router_from_engine_order(lf_router *r, const double *src_ord_dev, double *dst_pix_dev)
{
    if (!r || !src_ord_dev || !dst_pix_dev) return lf_set_error(LF_E_INVALID, "null argument");
    LF_HIP(hipSetDevice(r->device));
    const int n = (int)r->N;
    if (n > 0)
        hipLaunchKernelGGL(k_scatter, dim3(blocks_for(n)), dim3(kBlock), 0, r->ctx->stream, n, r->perm.p, src_ord_dev,
                           dst_pix_dev);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int lf_router_route_host(lf_router *r, double *discharge_host, const double *lateral_host, int section)
{
    if (!r || !discharge_host || !lateral_host) return lf_set_error(LF_E_INVALID, "null argument");
    if (section != LF_SECTION_MAIN && section != LF_SECTION_FLOODPLAINS)
        return lf_set_error(LF_E_SECTION, "The section parameter must be either 'main_channel' or 'floodplain'!");
    LF_HIP(hipSetDevice(r->device));
    const size_t bytes = sizeof(double) * (size_t)r->N;
    if (!r->io_q.p) LF_TRY(r->io_q.alloc(r->N));
    if (!r->io_lat.p) LF_TRY(r->io_lat.alloc(r->N));
    hipStream_t s = r->ctx->stream;
    if (bytes) {
        LF_HIP(hipMemcpyAsync(r->io_q.p, discharge_host, bytes, hipMemcpyHostToDevice, s));
        LF_HIP(hipMemcpyAsync(r->io_lat.p, lateral_host, bytes, hipMemcpyHostToDevice, s));
    }
    LF_TRY(route_device(r, r->io_q.p, r->io_lat.p, section));
    if (bytes) LF_HIP(hipMemcpyAsync(discharge_host, r->io_q.p, bytes, hipMemcpyDeviceToHost, s));
    LF_HIP(hipStreamSynchronize(s));
    return LF_OK;
}

int lf_count_nonfinite(int device, const double *x_dev, int64_t n, int64_t *count)
{
    if (!x_dev || !count) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    lf_dbuf<unsigned long long> ctr;
    LF_TRY(ctr.alloc(1));
    LF_HIP(hipMemsetAsync(ctr.p, 0, sizeof(unsigned long long), c->stream));
    if (n > 0) {
        const int grid = (int)std::min<int64_t>((n + kBlock - 1) / kBlock, 2048);
        hipLaunchKernelGGL(k_count_nonfinite, dim3(grid), dim3(kBlock), 0, c->stream, (long long)n, x_dev, ctr.p);
    }
    unsigned long long h = 0;
    LF_HIP(hipMemcpyAsync(&h, ctr.p, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    LF_HIP(hipStreamSynchronize(c->stream));
    *count = (int64_t)h;
    return LF_OK;
}

int lf_router_device(const lf_router *r) { return r ? r->device : -1; }
int64_t lf_router_num_pixels(const lf_router *r) { return r ? r->N : -1; }

int lf_router_last_launches(const lf_router *r, int64_t stats[4])
{
    if (!r || !stats) return lf_set_error(LF_E_INVALID, "null argument");
    for (int i = 0; i < 4; ++i) stats[i] = r->last_stats[i];
    return LF_OK;
}

// Shape of the block plan of single router calls (build_level_blocks): how full the cone wavefronts are.
int lf_router_route_plan_stats(const lf_router *r, int64_t out[6])
{
    if (!r || !out) return lf_set_error(LF_E_INVALID, "null argument");
    for (int i = 0; i < 6; ++i) out[i] = 0;
    const int NB = r->rb_level.empty() ? 0 : (int)r->rb_level.size() - 1;
    out[0] = NB;
    for (int b = 0; b < NB; ++b) {
        const int k0 = r->rb_level[b], nl = r->rb_level[b + 1] - k0;
        if (nl < 2) continue;
        const int64_t cones = r->rb_row[b + 1] - r->rb_row[b] - 1;
        out[1] += 1;
        out[2] += cones;
        out[3] += cones * nl;
        out[4] += r->h_level_start[k0 + nl] - r->h_level_start[k0];
        out[5] = cones > out[5] ? cones : out[5];
    }
    return LF_OK;
}

// The fused-with-structures call validates the levels of the site lists once per set of device pointers; a caller
// that rebuilds its site lists (possibly at the same addresses) drops that cache here.
int lf_router_reset_site_cache(lf_router *r)
{
    if (!r) return lf_set_error(LF_E_INVALID, "null argument");
    r->site_cnt[0] = r->site_cnt[1] = -1;
    for (const void *&k : r->site_key) k = nullptr;
    r->site_level_sorted.clear();
    return LF_OK;
}

int lf_router_profile_enable(lf_router *r, int on)
{
    if (!r) return lf_set_error(LF_E_INVALID, "null argument");
    r->profile = on != 0;
    return LF_OK;
}

int lf_router_profile_read(lf_router *r, double out[9], int reset)
{
    if (!r || !out) return lf_set_error(LF_E_INVALID, "null argument");
    for (int i = 0; i < 9; ++i) out[i] = r->prof_acc[i];
    if (reset)
        for (int i = 0; i < 9; ++i) r->prof_acc[i] = 0;
    return LF_OK;
}

int lf_upstream_sum_device(lf_router *r, const double *w_dev, double *out_dev)
{
    if (!r || !w_dev || !out_dev) return lf_set_error(LF_E_INVALID, "null argument");
    LF_HIP(hipSetDevice(r->device));
    const int n = (int)r->N;
    if (n > 0)
        hipLaunchKernelGGL(k_upstream_sum, dim3(blocks_for(n)), dim3(kBlock), 0, r->ctx->stream, n, r->perm.p,
                           r->ups_ptr.p, w_dev, out_dev, (const uint8_t *)r->linked.p);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int lf_upstream_sum_host(lf_router *r, const double *w_host, double *out_host)
{
    if (!r || !w_host || !out_host) return lf_set_error(LF_E_INVALID, "null argument");
    LF_HIP(hipSetDevice(r->device));
    const size_t bytes = sizeof(double) * (size_t)r->N;
    if (!r->io_q.p) LF_TRY(r->io_q.alloc(r->N));
    if (!r->io_lat.p) LF_TRY(r->io_lat.alloc(r->N));
    hipStream_t s = r->ctx->stream;
    if (bytes) LF_HIP(hipMemcpyAsync(r->io_lat.p, w_host, bytes, hipMemcpyHostToDevice, s));
    LF_TRY(lf_upstream_sum_device(r, r->io_lat.p, r->io_q.p));
    if (bytes) LF_HIP(hipMemcpyAsync(out_host, r->io_q.p, bytes, hipMemcpyDeviceToHost, s));
    LF_HIP(hipStreamSynchronize(s));
    return LF_OK;
}

// accuflux over engine-order device vectors: acc[p] = x[p] + sum of acc over the upstream cells; nv (<= 4) vectors in
// one sweep.  On the level layout it runs on the router's block plan (blocks of up to 64 levels cone by cone through
// LDS, single wide levels one launch each), like a router call.
int lf_accuflux_ordered_multi_device(lf_router *r, int nv, const double *const *x_ord_dev, double *const *acc_ord_dev)
{
    if (!r || !x_ord_dev || !acc_ord_dev || nv < 1 || nv > kMaxAccu) return lf_set_error(LF_E_INVALID, "bad argument");
    for (int v = 0; v < nv; ++v)
        if (!x_ord_dev[v] || !acc_ord_dev[v]) return lf_set_error(LF_E_INVALID, "null argument");
    if (r->linked.p) return lf_set_error(LF_E_INVALID, "accuflux is not defined on a graph with structure links");
    LF_HIP(hipSetDevice(r->device));
    hipStream_t s = r->ctx->stream;
    if (r->N == 0) return LF_OK;
    accu_multi M;
    for (int v = 0; v < kMaxAccu; ++v) {
        M.x[v] = x_ord_dev[v < nv ? v : 0];
        M.acc[v] = acc_ord_dev[v < nv ? v : 0];
    }
    int64_t launches = 0;
#define LF_ACCU(KERNEL, GRID, BLOCK, ...)                                                      \
    do {                                                                                       \
        if (nv == 1)                                                                           \
            hipLaunchKernelGGL(KERNEL<1>, GRID, BLOCK, 0, s, __VA_ARGS__);                     \
        else if (nv == 2)                                                                      \
            hipLaunchKernelGGL(KERNEL<2>, GRID, BLOCK, 0, s, __VA_ARGS__);                     \
        else if (nv == 3)                                                                      \
            hipLaunchKernelGGL(KERNEL<3>, GRID, BLOCK, 0, s, __VA_ARGS__);                     \
        else                                                                                   \
            hipLaunchKernelGGL(KERNEL<4>, GRID, BLOCK, 0, s, __VA_ARGS__);                     \
        ++launches;                                                                            \
    } while (0)
#define LF_ACCU_CW(NVV, GRID, ...)                                                            \
    do {                                                                                       \
        if (r->rb_cw == 64)                                                                    \
            hipLaunchKernelGGL((k_accu_cones<NVV, 64>), GRID, dim3(64), 0, s, __VA_ARGS__);    \
        else                                                                                   \
            hipLaunchKernelGGL((k_accu_cones<NVV, kBlock>), GRID, dim3(kBlock), 0, s, __VA_ARGS__); \
    } while (0)
#define LF_ACCU_CONES(GRID, ...)                                                               \
    do {                                                                                       \
        if (nv == 1)                                                                           \
            LF_ACCU_CW(1, GRID, __VA_ARGS__);                                                  \
        else if (nv == 2)                                                                      \
            LF_ACCU_CW(2, GRID, __VA_ARGS__);                                                  \
        else if (nv == 3)                                                                      \
            LF_ACCU_CW(3, GRID, __VA_ARGS__);                                                  \
        else                                                                                   \
            LF_ACCU_CW(4, GRID, __VA_ARGS__);                                                  \
        ++launches;                                                                            \
    } while (0)
    if (r->rb_lmax > 1 && cones_enabled()) {
        const int NB = (int)r->rb_level.size() - 1;
        for (int b = 0; b < NB; ++b) {
            const int k0 = r->rb_level[b], nl = r->rb_level[b + 1] - k0;
            if (nl > 1) {
                cone_plan_args C;
                C.cone = r->rb_cone.p + r->rb_off[b];
                C.nl = nl;
                C.n_cells = (int)r->N;
                const dim3 grid((unsigned)(r->rb_row[b + 1] - r->rb_row[b] - 1));
                LF_ACCU_CONES(grid, C, r->ups_ptr.p, M);
            } else {
                const int first = (int)r->h_level_start[k0];
                const int count = (int)(r->h_level_start[k0 + 1] - r->h_level_start[k0]);
                LF_ACCU(k_accu_level, dim3(blocks_for(count)), dim3(kBlock), first, count, r->ups_ptr.p, M);
            }
        }
    } else {
        for (const segment &g : r->schedule) {
            if (g.wide) {
                const int first = (int)r->h_level_start[g.k0];
                const int count = (int)(r->h_level_start[g.k0 + 1] - r->h_level_start[g.k0]);
                LF_ACCU(k_accu_level, dim3(blocks_for(count)), dim3(kBlock), first, count, r->ups_ptr.p, M);
            } else {
                LF_ACCU(k_accu_narrow, dim3(1), dim3(kNarrowBlock), g.k0, g.k1, r->level_start.p, r->ups_ptr.p, M);
            }
        }
    }
#undef LF_ACCU
#undef LF_ACCU_CONES
#undef LF_ACCU_CW
    r->last_stats[0] = launches;
    r->last_stats[1] = r->last_stats[2] = 0;
    r->last_stats[3] = r->NL;
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int lf_accuflux_ordered_device(lf_router *r, const double *x_ord_dev, double *acc_ord_dev)
{
    const double *x[1] = {x_ord_dev};
    double *acc[1] = {acc_ord_dev};
    return lf_accuflux_ordered_multi_device(r, 1, x, acc);
}

int lf_accuflux_host(lf_router *r, const double *x_host, double *out_host)
{
    if (!r || !x_host || !out_host) return lf_set_error(LF_E_INVALID, "null argument");
    if (r->linked.p) return lf_set_error(LF_E_INVALID, "accuflux is not defined on a graph with structure links");
    LF_HIP(hipSetDevice(r->device));
    const int n = (int)r->N;
    const size_t bytes = sizeof(double) * (size_t)r->N;
    if (!r->io_q.p) LF_TRY(r->io_q.alloc(r->N));
    if (!r->io_lat.p) LF_TRY(r->io_lat.alloc(r->N));
    if (!r->tmp_ord.p) LF_TRY(r->tmp_ord.alloc(r->N));
    hipStream_t s = r->ctx->stream;
    if (n == 0) return LF_OK;
    LF_HIP(hipMemcpyAsync(r->io_lat.p, x_host, bytes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_gather, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, r->perm.p, r->io_lat.p, r->tmp_ord.p);
    LF_TRY(lf_accuflux_ordered_device(r, r->tmp_ord.p, r->qord.p));
    hipLaunchKernelGGL(k_scatter, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, r->perm.p, r->qord.p, r->io_q.p);
    LF_HIP(hipGetLastError());
    LF_HIP(hipMemcpyAsync(out_host, r->io_q.p, bytes, hipMemcpyDeviceToHost, s));
    LF_HIP(hipStreamSynchronize(s));
    return LF_OK;
}

} // extern "C"

// accessors for lf_ldd.hip (the router struct is private to this file)
struct lf_router_view {
    int device;
    lf_device_ctx *ctx;
    int64_t N;
    const int32_t *perm, *ups_ptr;
    const uint8_t *linked;
    int32_t **parent_slot;
    int32_t **root_slot;
};

int lf_router_view_of(lf_router *r, lf_router_view *v)
{
    if (!r || !v) return lf_set_error(LF_E_INVALID, "null argument");
    LF_HIP(hipSetDevice(r->device));
    v->device = r->device;
    v->ctx = r->ctx;
    v->N = r->N;
    v->perm = r->perm.p;
    v->ups_ptr = r->ups_ptr.p;
    v->linked = r->linked.p;
    v->parent_slot = &r->parent.p;
    v->root_slot = &r->root.p;
    return LF_OK;
}

int lf_router_alloc_root(lf_router *r)
{
    if (!r) return lf_set_error(LF_E_INVALID, "null argument");
    return r->root.p ? LF_OK : r->root.alloc((size_t)r->N);
}

// (a root table whose contents never arrived must not be trusted by later calls)
int lf_router_drop_root(lf_router *r)
{
    if (!r) return lf_set_error(LF_E_INVALID, "null argument");
    r->root.release();
    return LF_OK;
}

// grow-only scratch of the catchment totals (count doubles)
int lf_router_totals_scratch(lf_router *r, size_t count, double **p)
{
    if (!r || !p) return lf_set_error(LF_E_INVALID, "null argument");
    if (r->totals_scratch.n < count) {
        LF_HIP(hipStreamSynchronize(r->ctx->stream));
        LF_TRY(r->totals_scratch.alloc(count));
    }
    *p = r->totals_scratch.p;
    return LF_OK;
}

int lf_router_alloc_parent(lf_router *r)
{
    if (!r) return lf_set_error(LF_E_INVALID, "null argument");
    return r->parent.p ? LF_OK : r->parent.alloc((size_t)r->N);
}

// ================================================================================================
// routing.dynamic() sub-step: element-wise arithmetic around the router calls (routing.py:512-603,
// 693-703), fused into three kernels so that a model step's NoRoutSteps x (1..2) router calls never
// leave the device.
// ================================================================================================
namespace {

// routing.py:512 (+524 in the single branch) and, for split routing, 549-567
__global__ void __launch_bounds__(kBlock) k_substep_sideflow(int n, lf_substep_args A)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    double side = A.IsChannelKinematic[p] ? A.SideflowChanM3[p] * A.InvChanLength[p] * A.InvDtRouting : 0.0;
    if (!A.split) {
        if (isnan(side)) side = 0.0; // :524
        A.scratch0[p] = side;
        return;
    }
    const double m3 = A.ChanM3Kin[p], m3_2 = A.Chan2M3Kin[p];
    const double tot = m3 + m3_2;
    const double ratio = (tot > 0) ? m3 / tot : 0.0;                                  // :549
    double s1 = ((tot - A.Chan2M3Start[p]) > A.M3Limit[p]) ? ratio * side : side;     // :557-558
    if (fabs(side) < 1e-7) s1 = side;                                                 // :563
    A.Sideflow1Chan[p] = s1;
    A.scratch0[p] = s1;
    A.scratch1[p] = (side - s1) + A.Chan2QStart[p] * A.InvChanLength[p];              // :565-567
}

__device__ __forceinline__ void velocity(const lf_substep_args &A, int p, double m3, double q)
{
    double area = m3 * A.InvChanLength[p]; // :693
    if (area < 0.01) area = 0.01;
    const double v1 = q / area, v2 = 0.36 * pow(q, 0.24);
    double v = (v2 < v1) ? v2 : v1; // np.minimum, NaN propagates
    if (isnan(v2)) v = v2;
    double sinu = sqrt(A.PixelArea[p]) * A.InvChanLength[p];
    if (sinu > 1) sinu = 1;
    v *= sinu;
    A.FlowVelocity[p] = v;
    A.TravelDistance[p] = v * A.DtSec;
}

// routing.py:527-538 (single) / 574-578 (split, main channel)
__global__ void __launch_bounds__(kBlock) k_substep_main(int n, lf_substep_args A)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const bool b35 = A.Beta == 0.6;
    double v = A.ChanLength[p] * A.ChannelAlpha[p] * (b35 ? lf_pow_3_5(A.ChanQKin[p]) : pow(A.ChanQKin[p], A.Beta));
    if (v < 0.0) v = 0.0;
    const double x = v * A.InvChanLength[p] * A.InvChannelAlpha[p];
    const double q = b35 ? lf_pow_5_3(x) : pow(x, A.InvBeta);
    A.ChanM3Kin[p] = v;
    A.ChanQKin[p] = q;
    if (!A.split) {
        A.ChanQ[p] = q;
        A.sumDisDay[p] += q;
        velocity(A, p, v, q);
    }
}

// routing.py:584-603 + 693-703
__global__ void __launch_bounds__(kBlock) k_substep_floodplain(int n, lf_substep_args A)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const double start = A.Chan2M3Start[p];
    const bool b35 = A.Beta == 0.6;
    double v = A.ChanLength[p] * A.ChannelAlpha2[p] * (b35 ? lf_pow_3_5(A.Chan2QKin[p]) : pow(A.Chan2QKin[p], A.Beta));
    if ((v - start) < 0.0) v = start;
    A.Chan2M3Kin[p] = v;
    A.CrossSection2Area[p] = (v - start) * A.InvChanLength[p];
    const double x2 = v * A.InvChanLength[p] * A.InvChannelAlpha2[p];
    const double q2 = b35 ? lf_pow_5_3(x2) : pow(x2, A.InvBeta);
    A.Chan2QKin[p] = q2;
    const double q1 = A.ChanQKin[p];
    double q = q1 + q2 - A.QLimit[p];
    if (q < 0.0) q = 0.0; // np.maximum(q, 0.0), NaN propagates
    A.ChanQ[p] = q;
    A.sumDisDay[p] += q;
    velocity(A, p, A.ChanM3Kin[p], q1);
}

} // namespace

// The three element-wise stages of a sub-step on their own (stage 0: sideflow assembly, 1: main-channel fix-up and
// discharge sums, 2: floodplain fix-up), for callers that run the router calls in between themselves -- the row-block
// partition (lf_dist_routing_substep) does, with halo exchanges inside each router call.
extern "C" int lf_substep_stage(int device, int stage, int64_t n, const lf_substep_args *a)
{
    if (!a || stage < 0 || stage > 2 || n < 0) return lf_set_error(LF_E_INVALID, "bad argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (n == 0) return LF_OK;
    const dim3 grid(blocks_for(n)), block(kBlock);
    if (stage == 0)
        hipLaunchKernelGGL(k_substep_sideflow, grid, block, 0, c->stream, (int)n, *a);
    else if (stage == 1)
        hipLaunchKernelGGL(k_substep_main, grid, block, 0, c->stream, (int)n, *a);
    else
        hipLaunchKernelGGL(k_substep_floodplain, grid, block, 0, c->stream, (int)n, *a);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

extern "C" int lf_routing_substep(lf_router *r, const lf_substep_args *a)
{
    if (!r || !a) return lf_set_error(LF_E_INVALID, "null argument");
    if (a->split && !r->has_floodplains)
        return lf_set_error(LF_E_SECTION, "split routing requested but the router has no floodplain alpha");
    if (r->linked.p)
        return lf_set_error(LF_E_INVALID, "a router on a graph with structure links runs only lf_routing_substeps_fused*");
    LF_HIP(hipSetDevice(r->device));
    hipStream_t s = r->ctx->stream;
    const int n = (int)r->N;
    if (n == 0) return LF_OK;
    const dim3 grid(blocks_for(n)), block(kBlock);
    hipLaunchKernelGGL(k_substep_sideflow, grid, block, 0, s, n, *a);
    const bool ordered = a->engine_order != 0;
    LF_TRY(route_device(r, a->ChanQKin, a->scratch0, LF_SECTION_MAIN, ordered));
    hipLaunchKernelGGL(k_substep_main, grid, block, 0, s, n, *a);
    if (a->split) {
        LF_TRY(route_device(r, a->Chan2QKin, a->scratch1, LF_SECTION_FLOODPLAINS, ordered));
        hipLaunchKernelGGL(k_substep_floodplain, grid, block, 0, s, n, *a);
    }
    LF_HIP(hipGetLastError());
    return LF_OK;
}

// ================================================================================================
// Fused multi-sub-step routing: the NoRoutSteps sub-steps of a model step (Lisflood_dynamic.py:179-180) as ONE
// skewed wavefront.  Sub-step s+1 of a cell needs only the cell's own state after sub-step s and the
// sub-step-(s+1) router output of its upstream cells (one level up), so launch t processes every (level k,
// sub-step s) with k + s = t: NL + S - 1 launches instead of S x (1..2) x NL, each S times wider.  Router
// outputs live in two parity buffers per section (sub-step s writes buffer s&1, which sub-step s+2 may only
// overwrite one launch after its last reader).  Arithmetic per cell is exactly that of lf_routing_substep, so
// the result is bit-identical to S sequential sub-steps.  Valid when the sideflow of every sub-step is known
// up front (stride 0: the same vector for all sub-steps, as in the model when no lake / reservoir sits in the
// loop; stride N: one vector per sub-step).
// ================================================================================================
#include "lf_fused.h"

namespace {

int fused_impl(lf_router *r, const lf_substep_args *a, int nsteps, int64_t sideflow_stride, const lf_inloop_args *in,
               int msteps = 0, int64_t side_mstride = 0)
{
    if (!r || !a || nsteps < 1) return lf_set_error(LF_E_INVALID, "bad argument");
    if (!a->engine_order) return lf_set_error(LF_E_INVALID, "the fused sub-step wavefront needs engine-order vectors");
    if (a->split && !r->has_floodplains)
        return lf_set_error(LF_E_SECTION, "split routing requested but the router has no floodplain alpha");
    if (sideflow_stride != 0 && sideflow_stride != r->N) return lf_set_error(LF_E_INVALID, "sideflow_stride must be 0 or N");
    if (msteps <= 0) msteps = nsteps; // one model step
    if (nsteps % msteps != 0) return lf_set_error(LF_E_INVALID, "the sub-step count must be a multiple of the sub-steps per model step");
    if (msteps != nsteps && (in || sideflow_stride != 0 || (side_mstride != 0 && side_mstride < r->N)))
        return lf_set_error(LF_E_INVALID, "several model steps per call: no structures, one sideflow vector per model step "
                                          "(stride 0 or >= N)");
    LF_HIP(hipSetDevice(r->device));
    const int64_t n = r->N;
    if (n == 0) return LF_OK;
    if (!r->fused_qr1.p) LF_TRY(r->fused_qr1.alloc(2 * n));
    if (a->split && !r->fused_qr2.p) LF_TRY(r->fused_qr2.alloc(2 * n));
    fused_args F;
    F.S = *a;
    F.ups_ptr = r->ups_ptr.p;
    F.a1 = r->a1.p;
    F.a2 = r->a2.p;
    F.dx = r->dx_per_pixel ? r->dx.p : nullptr;
    F.level_start = r->level_start.p;
    F.qr1 = r->fused_qr1.p;
    F.qr2 = r->fused_qr2.p;
    F.hist1 = F.hist2 = nullptr;
    F.root1 = F.root2 = nullptr;
    F.nroots = 0;
    F.root_ss = 1;
    F.root_st = 0;
    F.n = n;
    F.side_stride = sideflow_stride;
    F.msteps = msteps;
    F.side_mstride = side_mstride;
    F.dx_scalar = r->dx_scalar;
    F.beta = r->beta;
    F.inv_beta = r->inv_beta;
    F.b_minus_1 = r->b_minus_1;
    F.kmax = r->kmax;
    F.nlevels = (int)r->NL;
    F.nsteps = nsteps;
    F.solve35 = r->fused ? 1 : 0;
    F.linked = r->linked.p;
    F.level_nlinked = r->level_nlinked.p;
    F.inert = nullptr;
    F.site_level = nullptr;
    F.fb_level = F.fb_row = F.fb_cone = nullptr;
    F.fb_off = nullptr;
    F.fb_lvl2blk = nullptr;
    F.fb_nblocks = 0;
    F.fb_block0 = 0;
    F.d_ups_base = F.d_ups_idx = F.d_out_slot = nullptr;
    F.recompute = nullptr;
    F.dt = r->dt;
    F.use_lvl = 0;
    std::memset(&F.I, 0, sizeof(F.I));
    hipStream_t s = r->ctx->stream;
    const int NL = (int)r->NL;
    // ---- structures: levels of the site cells; every cell feeding a site must sit on the site's own level ----------
    std::vector<int> lv_sorted;
    int64_t nsites = 0;
    if (in) {
        F.I = *in;
        lf_inloop_args &I = F.I;
        if (I.n_lakes <= 0) {
            I.n_lakes = 0;
            I.QLakeOutM3Dt = nullptr;
        }
        if (I.n_res <= 0) {
            I.n_res = 0;
            I.QResOutM3Dt = nullptr;
        }
        if (!I.ToChanM3RunoffDt || !I.SideflowChanM3 || I.N != n)
            return lf_set_error(LF_E_INVALID, "lf_inloop_args: ToChanM3RunoffDt, SideflowChanM3 and N = num_pixels are needed");
        I.ChanQ = a->ChanQ;
        nsites = I.n_lakes + I.n_res;
        const void *key[4] = {I.lake_cell, I.lake_ups_idx, I.res_cell, I.res_ups_idx};
        const bool checked = nsites > 0 && r->site_level.p && r->site_cnt[0] == I.n_lakes && r->site_cnt[1] == I.n_res &&
                             std::memcmp(key, r->site_key, sizeof(key)) == 0;
        if (checked) { // same device lists as the last call: levels already validated and resident
            F.site_level = r->site_level.p;
            lv_sorted = r->site_level_sorted;
        } else if (nsites > 0) {
            auto level_of = [&](int pos) {
                return (int)(std::upper_bound(r->h_level_start.begin(), r->h_level_start.end(), (int64_t)pos) -
                             r->h_level_start.begin()) - 1;
            };
            std::vector<int> lv(nsites);
            auto check_sites = [&](int64_t cnt, const int32_t *cell_dev, const int32_t *ptr_dev, const int32_t *idx_dev,
                                   int64_t off, const char *what) -> int {
                if (cnt == 0) return LF_OK;
                if (!cell_dev || !ptr_dev || !idx_dev) return lf_set_error(LF_E_INVALID, "%s site lists missing", what);
                std::vector<int32_t> cell(cnt), ptr(cnt + 1);
                LF_HIP(hipMemcpy(cell.data(), cell_dev, sizeof(int32_t) * cnt, hipMemcpyDeviceToHost));
                LF_HIP(hipMemcpy(ptr.data(), ptr_dev, sizeof(int32_t) * (cnt + 1), hipMemcpyDeviceToHost));
                std::vector<int32_t> idx(std::max<int32_t>(ptr[cnt], 1));
                if (ptr[cnt] > 0)
                    LF_HIP(hipMemcpy(idx.data(), idx_dev, sizeof(int32_t) * ptr[cnt], hipMemcpyDeviceToHost));
                for (int64_t i = 0; i < cnt; ++i) {
                    if (cell[i] < 0 || cell[i] >= n) return lf_set_error(LF_E_INVALID, "%s cell out of range", what);
                    lv[off + i] = level_of(cell[i]);
                    for (int32_t e = ptr[i]; e < ptr[i + 1]; ++e)
                        if (idx[e] < 0 || idx[e] >= n || level_of(idx[e]) != lv[off + i])
                            return lf_set_error(LF_E_INVALID, "%s %lld: a cell draining into it is not on its level -- build "
                                                "the router's graph with lf_graph_create_ex(virtual_down)", what, (long long)i);
                }
                return LF_OK;
            };
            LF_TRY(check_sites(I.n_lakes, I.lake_cell, I.lake_ups_ptr, I.lake_ups_idx, 0, "lake"));
            LF_TRY(check_sites(I.n_res, I.res_cell, I.res_ups_ptr, I.res_ups_idx, I.n_lakes, "reservoir"));
            LF_TRY(r->site_level.upload(lv.data(), (size_t)nsites, r->ctx->stream));
            F.site_level = r->site_level.p;
            lv_sorted = lv;
            std::sort(lv_sorted.begin(), lv_sorted.end());
            r->site_level_sorted = lv_sorted;
            std::memcpy(r->site_key, key, sizeof(key));
            r->site_cnt[0] = I.n_lakes;
            r->site_cnt[1] = I.n_res;
        }
    }
    int64_t launches = 0;
    bool grid2d = false;
    {
        const char *e2 = std::getenv("LF_FUSED_2D_GRID"); // A/B switch: one grid row per sub-step, sized by the widest
        grid2d = e2 && e2[0] == '1';
        const char *e = std::getenv("LF_NO_INERT_SKIP"); // A/B switch
        if (!in && r->n_isolated > 0 && nsteps > 1 && !(e && e[0] == '1')) {
            if (!r->inert.p) LF_TRY(r->inert.alloc(n));
            hipLaunchKernelGGL(k_inert_flags, dim3(blocks_for(n)), dim3(kBlock), 0, s, (long long)n, *a, r->isolated.p,
                               r->a1.p, r->a2.p, F.dx, r->inert.p);
            F.inert = r->inert.p;
            ++launches;
        }
    }
    {
        const char *e = std::getenv("LF_NO_RECOMPUTE"); // A/B switch: stream the derived statics as given
        if (!(e && e[0] == '1') && nsteps > 1) {
            if (!r->derived_ok.p) LF_TRY(r->derived_ok.alloc(1));
            LF_HIP(hipMemsetD32Async((hipDeviceptr_t)r->derived_ok.p, 3, 1, s));
            hipLaunchKernelGGL(k_check_derived, dim3(blocks_for(n)), dim3(kBlock), 0, s, (long long)n, *a, r->a1.p, r->a2.p,
                               F.dx, r->dx_scalar, r->dt, r->derived_ok.p);
            F.recompute = r->derived_ok.p;
            ++launches;
        }
    }
    // ---- few, wide levels: level after level, every level through all its sub-steps (k_fused_level_steps) ----------------
    // LF_FUSED_TIME_MAJOR=0 / 1: never / whenever it applies (A/B switch); LF_FUSED_TIME_MAJOR_LEVELS: the level count up to
    // which it is the default -- each launch carries a dependent chain of nsteps solves (~10 us), so NL launches of that
    // kind must stay small beside what the saved traffic (~3.8 kB per cell and model step) is worth
    {
        static const int tm_levels = [] {
            const char *e = std::getenv("LF_FUSED_TIME_MAJOR_LEVELS");
            return e ? std::atoi(e) : 192;
        }();
        const char *e = std::getenv("LF_FUSED_TIME_MAJOR");
        const bool applies = !in && !F.linked && nsteps > 1;
        const bool want = e ? e[0] != '0' : (NL <= tm_levels && n >= 20000 * (int64_t)NL);
        if (applies && want) {
            const bool ok = lf_history_ensure(r->fused_hist1, r->fused_hist2, r->fused_hist_refused, (size_t)nsteps * n, a->split);
            if (ok) {
                F.hist1 = r->fused_hist1.p;
                F.hist2 = r->fused_hist2.p;
                for (int k = 0; k < NL; ++k) {
                    const int64_t w = r->h_level_start[k + 1] - r->h_level_start[k];
                    if (w <= 0) continue;
                    const bool all35 = r->fused && a->Beta == 0.6;
                    if (a->split && all35)
                        hipLaunchKernelGGL((k_fused_level_steps<true, true>), dim3(blocks_for(w)), dim3(kBlock), 0, s, F, k);
                    else if (a->split)
                        hipLaunchKernelGGL((k_fused_level_steps<true, false>), dim3(blocks_for(w)), dim3(kBlock), 0, s, F, k);
                    else if (all35)
                        hipLaunchKernelGGL((k_fused_level_steps<false, true>), dim3(blocks_for(w)), dim3(kBlock), 0, s, F, k);
                    else
                        hipLaunchKernelGGL((k_fused_level_steps<false, false>), dim3(blocks_for(w)), dim3(kBlock), 0, s, F, k);
                    ++launches;
                }
                LF_HIP(hipGetLastError());
                r->last_stats[0] = launches;
                r->last_stats[1] = launches;
                r->last_stats[2] = 0;
                r->last_stats[3] = r->NL;
                return LF_OK;
            }
            // no room for the history inside its budget (remembered in fused_hist_refused): the skewed wavefront below
        }
    }
    if (r->fb_lmax > 1 && nsteps <= kMaxPackedSteps) { // several levels per launch (k_fused_cones)
        const int NB = (int)r->fb_level.size() - 1;
        F.fb_level = r->fb_level_dev.p;
        F.fb_row = r->fb_row_dev.p;
        F.fb_cone = r->fb_cone.p;
        F.fb_off = r->fb_off_dev.p;
        F.fb_lvl2blk = r->fb_lvl2blk_dev.p;
        F.fb_nblocks = NB;
        std::vector<int> site_blocks; // sorted blocks of the lakes and reservoirs
        for (int lv : lv_sorted) site_blocks.push_back(r->fb_lvl2blk[lv]);
        auto cones = [&](int b) { return (int64_t)(r->fb_row[b + 1] - r->fb_row[b] - 1); };
        auto multi = [&](int b) { return r->fb_level[b + 1] - r->fb_level[b] > 1; };
        const bool all35 = r->fused && a->Beta == 0.6; // otherwise: run-time flags and inlined OCML pow, as fused_cell
        for (int t = 0; t < NB + nsteps - 1; ++t) {
            // (block t - q, sub-step q), q = 0 .. nsteps-1, are independent of each other: the blocks of several levels go
            // to the cone kernel, the single (wide) levels to the level kernel, which streams them at full occupancy
            F.t = t;
            if (nsites > 0) { // sites of the blocks [t - nsteps + 1, t]: sub-step t - block, before the cells of that block
                const int b_lo = std::max(0, t - nsteps + 1), b_hi = std::min(NB - 1, t);
                auto it = std::lower_bound(site_blocks.begin(), site_blocks.end(), b_lo);
                if (it != site_blocks.end() && *it <= b_hi) {
                    hipLaunchKernelGGL(k_sites_blocks, dim3(blocks_for(nsites)), dim3(kBlock), 0, s, F);
                    ++launches;
                }
            }
            int64_t acc = 0;
            for (int q = 0; q < nsteps; ++q) {
                F.blk_start[q] = (int)acc;
                const int b = t - q;
                if (b >= 0 && b < NB && multi(b)) acc += cones(b);
            }
            F.blk_start[nsteps] = (int)acc;
            if (acc >= ((int64_t)1 << 31)) return lf_set_error(LF_E_INVALID, "fused sub-steps: grid too large");
            if (acc > 0) {
                F.packed = 1;
                F.use_lvl = 0;
                const dim3 grid((unsigned)acc);
#define LF_CONES_CW(ST, CW)                                                                                  \
    do {                                                                                                     \
        if (a->split && all35)                                                                               \
            hipLaunchKernelGGL((k_fused_cones<true, true, ST, false, CW>), grid, dim3(CW), 0, s, F);         \
        else if (a->split)                                                                                   \
            hipLaunchKernelGGL((k_fused_cones<true, false, ST, false, CW>), grid, dim3(CW), 0, s, F);        \
        else if (all35)                                                                                      \
            hipLaunchKernelGGL((k_fused_cones<false, true, ST, false, CW>), grid, dim3(CW), 0, s, F);        \
        else                                                                                                 \
            hipLaunchKernelGGL((k_fused_cones<false, false, ST, false, CW>), grid, dim3(CW), 0, s, F);       \
    } while (0)
#define LF_CONES(ST)                                                                                         \
    do {                                                                                                     \
        if (r->fb_cw == 64)                                                                                  \
            LF_CONES_CW(ST, 64);                                                                             \
        else                                                                                                 \
            LF_CONES_CW(ST, kBlock);                                                                         \
    } while (0)
                // The chain / supply form (lf_fused.h: k_fused_cones_split) where it applies and where the launch is
                // chain-bound: up to ~1100 cones in flight (24 sub-steps x 2900 cells per level) it is 1.1 - 1.7 x faster,
                // beyond that the launch is bound by throughput (three wavefronts and 31 KB of LDS per cone) and the
                // one-wavefront kernel wins (deep 2000^2, ~860 cones: 7.7 vs 8.7 ms per model step,
                // 3000^2, ~1220: 15.9 vs 14.5).  LF_FUSED_SPLIT=0 / 1: never / always (A/B switch).
                static const int64_t split_max = [] {
                    const char *e = std::getenv("LF_FUSED_SPLIT_MAX");
                    return e ? std::atoll(e) : (long long)1100;
                }();
                // (with structures in the loop the supply wavefronts also carry the sideflow assembly and, on reaches with
                // transmission loss, two OCML pow calls per cell: the crossover is lower -- 3000^2 with 256 sites: 5.5 vs 8.2 ms
                // at 1000^2, 16.1 vs 15.2 at 2000^2)
                static const int64_t split_max_struct = [] {
                    const char *e = std::getenv("LF_FUSED_SPLIT_MAX_STRUCT");
                    return e ? std::atoll(e) : (long long)600;
                }();
                const char *es = std::getenv("LF_FUSED_SPLIT");
                // (with structures in the loop: on a graph with their links; without them: on a graph without links)
                const bool split_form = all35 && !F.inert && (in ? F.linked != nullptr : F.linked == nullptr) && r->fb_cw == 64 &&
                                        n < ((int64_t)1 << 29) && (es ? es[0] != '0' : acc <= (in ? split_max_struct : split_max));
                if (split_form && in && a->split)
                    hipLaunchKernelGGL((k_fused_cones_split<true, true>), grid, dim3(64 * (1 + kFusedKC)), 0, s, F);
                else if (split_form && in)
                    hipLaunchKernelGGL((k_fused_cones_split<false, true>), grid, dim3(64 * (1 + kFusedKC)), 0, s, F);
                else if (split_form && a->split)
                    hipLaunchKernelGGL((k_fused_cones_split<true>), grid, dim3(64 * (1 + kFusedKC)), 0, s, F);
                else if (split_form)
                    hipLaunchKernelGGL((k_fused_cones_split<false>), grid, dim3(64 * (1 + kFusedKC)), 0, s, F);
                else if (in)
                    LF_CONES(true);
                else
                    LF_CONES(false);
#undef LF_CONES_CW
#undef LF_CONES
                ++launches;
            }
            int64_t acc1 = 0, widest = 0;
            for (int q = 0; q < nsteps; ++q) {
                F.blk_start[q] = (int)acc1;
                F.lvl[q] = -1;
                const int b = t - q;
                if (b >= 0 && b < NB && !multi(b)) {
                    const int k = r->fb_level[b];
                    const int64_t w = r->h_level_start[k + 1] - r->h_level_start[k];
                    F.lvl[q] = k;
                    acc1 += blocks_for(w);
                    widest = std::max(widest, w);
                }
            }
            F.blk_start[nsteps] = (int)acc1;
            if (acc1 > 0) {
                F.use_lvl = 1;
                F.packed = 0;
                dim3 grid(blocks_for(widest), nsteps);
                if (!grid2d && 2 * acc1 <= (int64_t)blocks_for(widest) * nsteps && acc1 < ((int64_t)1 << 31)) {
                    F.packed = 1; // as below: packed where it halves the grid
                    grid = dim3((unsigned)acc1, 1);
                }
                if (in && a->split)
                    hipLaunchKernelGGL((k_fused_substeps<true, true>), grid, dim3(kBlock), 0, s, F);
                else if (in)
                    hipLaunchKernelGGL((k_fused_substeps<false, true>), grid, dim3(kBlock), 0, s, F);
                else if (a->split)
                    hipLaunchKernelGGL((k_fused_substeps<true, false>), grid, dim3(kBlock), 0, s, F);
                else
                    hipLaunchKernelGGL((k_fused_substeps<false, false>), grid, dim3(kBlock), 0, s, F);
                ++launches;
            }
        }
        LF_HIP(hipGetLastError());
        r->last_stats[0] = launches;
        r->last_stats[1] = launches;
        r->last_stats[2] = 0;
        r->last_stats[3] = r->NL;
        return LF_OK;
    }
    for (int t = 0; t < NL + nsteps - 1; ++t) {
        // widest level inside the window [t - nsteps + 1, t]
        const int k_lo = std::max(0, t - nsteps + 1), k_hi = std::min(NL - 1, t);
        int64_t widest = 0;
        for (int k = k_lo; k <= k_hi; ++k) widest = std::max(widest, r->h_level_start[k + 1] - r->h_level_start[k]);
        F.t = t;
        if (nsites > 0) { // any site with a level in [k_lo, k_hi]?
            auto it = std::lower_bound(lv_sorted.begin(), lv_sorted.end(), k_lo);
            if (it != lv_sorted.end() && *it <= k_hi) {
                hipLaunchKernelGGL(k_sites_wave, dim3(blocks_for(nsites)), dim3(kBlock), 0, s, F);
                ++launches;
            }
        }
        dim3 grid(blocks_for(widest), nsteps);
        const dim3 block(kBlock);
        // packed 1-D grid only where it saves at least half of the blocks: finding its sub-step costs a block ~1.7 us
        // (24 scalar kernarg loads), which shows on latency-bound launches (deep 5000^2: 65.2 vs 56.7 ms per model
        // step, same-call A/B) but is nothing against 360 000 empty blocks (2000^2 hot path: 14.5 vs 16.1 ms)
        F.packed = 0;
        if (nsteps <= kMaxPackedSteps && !grid2d) {
            int64_t acc = 0;
            for (int q = 0; q < nsteps; ++q) {
                F.blk_start[q] = (int)acc;
                const int k = t - q;
                if (k >= 0 && k < NL) acc += blocks_for(r->h_level_start[k + 1] - r->h_level_start[k]);
            }
            F.blk_start[nsteps] = (int)acc;
            if (2 * acc <= (int64_t)blocks_for(widest) * nsteps && acc < ((int64_t)1 << 31)) {
                F.packed = 1;
                grid = dim3((unsigned)std::max<int64_t>(acc, 1), 1);
            }
        }
        if (in && a->split)
            hipLaunchKernelGGL((k_fused_substeps<true, true>), grid, block, 0, s, F);
        else if (in)
            hipLaunchKernelGGL((k_fused_substeps<false, true>), grid, block, 0, s, F);
        else if (a->split)
            hipLaunchKernelGGL((k_fused_substeps<true, false>), grid, block, 0, s, F);
        else
            hipLaunchKernelGGL((k_fused_substeps<false, false>), grid, block, 0, s, F);
        ++launches;
    }
    LF_HIP(hipGetLastError());
    r->last_stats[0] = launches;
    r->last_stats[1] = launches;
    r->last_stats[2] = 0;
    r->last_stats[3] = r->NL;
    return LF_OK;
}

} // namespace

extern "C" int lf_routing_substeps_fused(lf_router *r, const lf_substep_args *a, int nsteps, int64_t sideflow_stride)
{
    return fused_impl(r, a, nsteps, sideflow_stride, nullptr);
}

// Several MODEL steps as one wavefront: the skew runs on across the model-step boundary, so the pipeline fill (one launch per
// level block) is paid once per call instead of once per model step -- what a deep network needs (deep 5000^2: 313 blocks +
// 23 launches per model step on its own, 24 per model step in steady state).
extern "C" int lf_routing_model_steps_fused(lf_router *r, const lf_substep_args *a, int steps_per_model_step, int n_model_steps,
                                            int64_t sideflow_model_stride)
{
    if (steps_per_model_step < 1 || n_model_steps < 1) return lf_set_error(LF_E_INVALID, "bad argument");
    return fused_impl(r, a, steps_per_model_step * n_model_steps, 0, nullptr, steps_per_model_step, sideflow_model_stride);
}

extern "C" int lf_routing_substeps_fused_structures(lf_router *r, const lf_substep_args *a, const lf_inloop_args *in,
                                                    int nsteps)
{
    if (!in) return lf_set_error(LF_E_INVALID, "null argument");
    return fused_impl(r, a, nsteps, 0, in);
}

// ================================================================================================
// PMC calibration: a stream copy with exactly known HBM traffic (n*8 B read + n*8 B written) in this
// engine's access widths, so that rocprofv3's FETCH_SIZE / WRITE_SIZE can be calibrated on gfx950
// (MI355X_MICROARCH.md, HBM section) before they are compared with byte counts.
// ================================================================================================
namespace {
__global__ void __launch_bounds__(kBlock) k_calib_copy8(long long n, const double *__restrict__ src, double *__restrict__ dst)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ void __launch_bounds__(kBlock) k_calib_copy16(long long n2, const double2 *__restrict__ src,
                                                         double2 *__restrict__ dst)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i < n2) dst[i] = src[i];
}
} // namespace

namespace {
struct calib_streams {
    const double *src[8];
};
template <int NS>
__global__ void __launch_bounds__(kBlock) k_calib_streams(long long n, calib_streams S, double *__restrict__ dst)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    double v[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) v[k] = S.src[k][i];
    double acc = v[0];
#pragma unroll
    for (int k = 1; k < NS; ++k) acc += v[k];
    dst[i] = acc;
}
} // namespace

// nread (1..8) read streams and one write stream of n doubles, one element per lane: the byte mix of the level sweep
// without its arithmetic and without its gather -- what the memory system gives a kernel of that many 8-byte streams
extern "C" int lf_calibration_streams(int device, int nread, const double *const *src_dev, double *dst_dev, int64_t n)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (!src_dev || !dst_dev || n <= 0 || nread < 1 || nread > 8) return lf_set_error(LF_E_INVALID, "bad argument");
    calib_streams S;
    for (int k = 0; k < 8; ++k) S.src[k] = src_dev[k < nread ? k : 0];
    const dim3 grid((unsigned)((n + kBlock - 1) / kBlock)), block(kBlock);
    switch (nread) {
    case 1: hipLaunchKernelGGL(k_calib_streams<1>, grid, block, 0, c->stream, (long long)n, S, dst_dev); break;
    case 2: hipLaunchKernelGGL(k_calib_streams<2>, grid, block, 0, c->stream, (long long)n, S, dst_dev); break;
    case 3: hipLaunchKernelGGL(k_calib_streams<3>, grid, block, 0, c->stream, (long long)n, S, dst_dev); break;
    case 4: hipLaunchKernelGGL(k_calib_streams<4>, grid, block, 0, c->stream, (long long)n, S, dst_dev); break;
    case 5: hipLaunchKernelGGL(k_calib_streams<5>, grid, block, 0, c->stream, (long long)n, S, dst_dev); break;
    case 6: hipLaunchKernelGGL(k_calib_streams<6>, grid, block, 0, c->stream, (long long)n, S, dst_dev); break;
    case 7: hipLaunchKernelGGL(k_calib_streams<7>, grid, block, 0, c->stream, (long long)n, S, dst_dev); break;
    default: hipLaunchKernelGGL(k_calib_streams<8>, grid, block, 0, c->stream, (long long)n, S, dst_dev); break;
    }
    LF_HIP(hipGetLastError());
    return LF_OK;
}

extern "C" int lf_calibration_copy(int device, const double *src_dev, double *dst_dev, int64_t n, int bytes_per_lane)
{
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (!src_dev || !dst_dev || n <= 0) return lf_set_error(LF_E_INVALID, "bad argument");
    if (bytes_per_lane == 16) {
        const long long n2 = n / 2;
        hipLaunchKernelGGL(k_calib_copy16, dim3((unsigned)((n2 + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream, n2,
                           (const double2 *)src_dev, (double2 *)dst_dev);
    } else {
        hipLaunchKernelGGL(k_calib_copy8, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, c->stream,
                           (long long)n, src_dev, dst_dev);
    }
    LF_HIP(hipGetLastError());
    return LF_OK;
}
