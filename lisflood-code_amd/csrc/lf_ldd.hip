// lf_ldd.hip -- the LDD operations routing.initial / structures.initial delegate to PCRaster (routing.py:90-171,
// structures.py:44-61): lddrepair, lddmask, downstream, catchment, and the per-catchment totals of the mass-balance
// bookkeeping (routing.py:483-499, 645-691: np.take(np.bincount(Catchments, w), Catchments)).
//
// PCRaster 4.3.3 is an un-vendored dependency of the reference; these are restatements of its documented semantics.
// What pins them: the reference's own data -- ec_upArea.nc (accuflux) and the catchment masks mask.map /
// subcatchment_mask.map of LF_ETRS89 (catchment) -- and brute-force walks in the tests.
//
//   lddrepair / lddmask : element-wise on the H x W uint8 raster (code 0 = missing value)
//   downstream          : out[p] = x[downstream cell of p], pits keep their own value; parent position per cell
//   catchment           : pointer jumping over the parent array -- O(log depth) passes instead of one launch per level
//   catchment totals    : accuflux reaches the outlet with the whole tree's sum; every cell then reads its outlet
#include <cstring>

#include "lf_sweep.h"

// accessors implemented in lf_router.hip (the router struct is private to it)
struct lf_router_view {
    int device;
    lf_device_ctx *ctx;
    int64_t N;
    const int32_t *perm, *ups_ptr;
    const uint8_t *linked;
    int32_t **parent_slot;                   // lazily built parent array, owned by the router
    int32_t **root_slot;                     // lazily built outlet-of-every-cell array, owned by the router
};
int lf_router_view_of(lf_router *r, lf_router_view *v);
int lf_router_alloc_parent(lf_router *r);    // allocates *parent_slot (N int32)
int lf_router_alloc_root(lf_router *r);      // allocates *root_slot (N int32)
int lf_router_drop_root(lf_router *r);       // frees it again (its contents did not arrive)
int lf_router_totals_scratch(lf_router *r, size_t count, double **p);
int lf_accuflux_ordered_device(lf_router *r, const double *x_ord_dev, double *acc_ord_dev);
int lf_accuflux_ordered_multi_device(lf_router *r, int nv, const double *const *x_ord_dev, double *const *acc_ord_dev);

namespace {

__constant__ int c_row_add[10] = {0, 1, 1, 1, 0, 0, 0, -1, -1, -1}; // by keypad code 1..9 (kinematic_wave_parallel.py:49-51)
__constant__ int c_col_add[10] = {0, -1, 0, 1, -1, 0, 1, -1, 0, 1};

__device__ __forceinline__ bool flows(int c) { return c >= 1 && c <= 9 && c != 5; }

// lddrepair: a cell draining off the map or into a missing value becomes a pit; unknown codes are missing values
__global__ void __launch_bounds__(kBlock) k_lddrepair(int H, int W, const uint8_t *__restrict__ ldd, uint8_t *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (long long)H * W) return;
    const int r = (int)(i / W), c = (int)(i % W);
    int code = ldd[i];
    if (code > 9) code = 0;
    if (flows(code)) {
        const int rr = r + c_row_add[code], cc = c + c_col_add[code];
        const bool inside = rr >= 0 && cc >= 0 && rr < H && cc < W;
        const int target = inside ? ldd[(long long)rr * W + cc] : 0;
        if (!inside || target == 0 || target > 9) code = 5;
    }
    out[i] = (uint8_t)code;
}

// lddmask(ldd, keep): cells outside `keep` become missing values, cells draining out of `keep` become pits
__global__ void __launch_bounds__(kBlock) k_lddmask(int H, int W, const uint8_t *__restrict__ ldd,
                                                    const uint8_t *__restrict__ keep, uint8_t *__restrict__ out)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (long long)H * W) return;
    const int r = (int)(i / W), c = (int)(i % W);
    int code = ldd[i];
    if (code > 9 || !keep[i]) code = 0;
    if (flows(code)) {
        const int rr = r + c_row_add[code], cc = c + c_col_add[code];
        const bool inside = rr >= 0 && cc >= 0 && rr < H && cc < W;
        const long long j = (long long)rr * W + cc;
        if (!inside || !keep[j] || ldd[j] == 0 || ldd[j] > 9) code = 5;
    }
    out[i] = (uint8_t)code;
}

__global__ void __launch_bounds__(kBlock) k_fill_i32(long long n, int *x, int v)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) x[i] = v;
}

// parent[e] = p for every upstream position e of p (structure links excluded); outlets keep -1
__global__ void __launch_bounds__(kBlock) k_parents(lf_router_view V, int *__restrict__ parent)
{
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= V.N) return;
    for (int e = V.ups_ptr[p]; e < V.ups_ptr[p + 1]; ++e)
        if (!V.linked || !V.linked[e]) parent[e] = (int)p;
}

// out[pixel(p)] = x[pixel(parent(p))], pits keep their own value (PCRaster downstream)
__global__ void __launch_bounds__(kBlock) k_downstream(long long n, const int *__restrict__ perm, const int *__restrict__ parent,
                                                       const double *__restrict__ x_pix, double *__restrict__ out_pix)
{
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const int d = parent[p];
    out_pix[perm[p]] = x_pix[perm[d >= 0 ? d : p]];
}

// jump[p] = p where the walk downstream stops (a point cell or an outlet), else parent(p)
__global__ void __launch_bounds__(kBlock) k_jump_init(long long n, const int *__restrict__ perm, const int *__restrict__ parent,
                                                      const int *__restrict__ points_pix, int *__restrict__ jump)
{
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    // both loads unconditional and ONE select: the short-circuit form (`parent < 0 || points[...] != 0`) is miscompiled by
    // hipcc 7.2 for gfx950 -- the lanes that stop at a point keep their parent (seen in the ISA and on the device)
    const int par = parent[p];
    const int pt = points_pix ? points_pix[perm[p]] : 0;
    jump[p] = ((par < 0) | (pt != 0)) ? (int)p : par;
}

__global__ void __launch_bounds__(kBlock) k_jump_step(long long n, const int *__restrict__ in, int *__restrict__ out,
                                                      unsigned long long *changed)
{
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const int a = in[p], b = in[a];
    out[p] = b;
    if (a != b) *changed = 1ull; // benign race: every writer stores the same value
}

__global__ void __launch_bounds__(kBlock) k_labels(long long n, const int *__restrict__ perm, const int *__restrict__ jump,
                                                   const int *__restrict__ points_pix, int *__restrict__ labels_pix)
{
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    labels_pix[perm[p]] = points_pix[perm[jump[p]]];
}

__global__ void __launch_bounds__(kBlock) k_take_root(long long n, const int *__restrict__ perm, const int *__restrict__ jump,
                                                      const double *__restrict__ acc_ord, double *__restrict__ out_pix)
{
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    out_pix[perm[p]] = acc_ord[jump[p]];
}

constexpr int kMaxTotals = 4;
struct totals_multi {
    const double *w_pix[kMaxTotals];
    double *w_ord[kMaxTotals];
    const double *acc_ord[kMaxTotals];
    double *out_pix[kMaxTotals];
};
__global__ void __launch_bounds__(kBlock) k_totals_gather(long long n, int nv, const int *__restrict__ perm, totals_multi T)
{
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const int pix = perm[p];
    for (int v = 0; v < nv; ++v) T.w_ord[v][p] = T.w_pix[v][pix];
}
__global__ void __launch_bounds__(kBlock) k_totals_take(long long n, int nv, const int *__restrict__ perm,
                                                        const int *__restrict__ root, totals_multi T)
{
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const int pix = perm[p], o = root[p];
    for (int v = 0; v < nv; ++v) T.out_pix[v][pix] = T.acc_ord[v][o];
}

int ensure_parent(lf_router *r, lf_router_view *V)
{
    LF_TRY(lf_router_view_of(r, V));
    if (*V->parent_slot || V->N == 0) return LF_OK;
    LF_TRY(lf_router_alloc_parent(r));
    LF_TRY(lf_router_view_of(r, V));
    hipStream_t s = V->ctx->stream;
    hipLaunchKernelGGL(k_fill_i32, dim3(blocks_for(V->N)), dim3(kBlock), 0, s, (long long)V->N, *V->parent_slot, -1);
    hipLaunchKernelGGL(k_parents, dim3(blocks_for(V->N)), dim3(kBlock), 0, s, *V, *V->parent_slot);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

// jump array after pointer jumping: the position every cell's walk downstream stops at
int converge_jumps(const lf_router_view &V, const int *points_pix_dev, lf_dbuf<int> &a, lf_dbuf<int> &b, int **result)
{
    const long long n = V.N;
    hipStream_t s = V.ctx->stream;
    LF_TRY(a.alloc(n));
    LF_TRY(b.alloc(n));
    lf_dbuf<unsigned long long> flag;
    LF_TRY(flag.alloc(1));
    hipLaunchKernelGGL(k_jump_init, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, V.perm, *V.parent_slot, points_pix_dev, a.p);
    int *in = a.p, *out = b.p;
    for (int round = 0; round < 40; ++round) { // the walk length halves per round: 2^40 levels is out of reach
        LF_HIP(hipMemsetAsync(flag.p, 0, sizeof(unsigned long long), s));
        hipLaunchKernelGGL(k_jump_step, dim3(blocks_for(n)), dim3(kBlock), 0, s, n, in, out, flag.p);
        unsigned long long h = 0;
        LF_HIP(hipMemcpyAsync(&h, flag.p, sizeof(h), hipMemcpyDeviceToHost, s));
        LF_HIP(hipStreamSynchronize(s));
        int *t = in;
        in = out;
        out = t;
        if (!h) break;
    }
    *result = in;
    return LF_OK;
}

} // namespace

extern "C" {

int lf_lddrepair_raster_device(int device, const uint8_t *ldd_dev, uint8_t *out_dev, int H, int W)
{
    if (!ldd_dev || !out_dev || H <= 0 || W <= 0 || ldd_dev == out_dev) return lf_set_error(LF_E_INVALID, "bad argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    hipLaunchKernelGGL(k_lddrepair, dim3(blocks_for((int64_t)H * W)), dim3(kBlock), 0, c->stream, H, W, ldd_dev, out_dev);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int lf_lddmask_raster_device(int device, const uint8_t *ldd_dev, const uint8_t *keep_dev, uint8_t *out_dev, int H, int W)
{
    if (!ldd_dev || !keep_dev || !out_dev || H <= 0 || W <= 0 || ldd_dev == out_dev)
        return lf_set_error(LF_E_INVALID, "bad argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    hipLaunchKernelGGL(k_lddmask, dim3(blocks_for((int64_t)H * W)), dim3(kBlock), 0, c->stream, H, W, ldd_dev, keep_dev, out_dev);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

// host-buffer forms: keep (may be NULL = lddrepair only)
int lf_ldd_raster_host(int device, const uint8_t *ldd_host, const uint8_t *keep_host, uint8_t *out_host, int H, int W)
{
    if (!ldd_host || !out_host || H <= 0 || W <= 0) return lf_set_error(LF_E_INVALID, "bad argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    const size_t n = (size_t)H * (size_t)W;
    lf_dbuf<uint8_t> a, k, o;
    LF_TRY(a.upload(ldd_host, n, c->stream));
    LF_TRY(o.alloc(n));
    if (keep_host) {
        LF_TRY(k.upload(keep_host, n, c->stream));
        LF_TRY(lf_lddmask_raster_device(device, a.p, k.p, o.p, H, W));
    } else {
        LF_TRY(lf_lddrepair_raster_device(device, a.p, o.p, H, W));
    }
    LF_HIP(hipMemcpyAsync(out_host, o.p, n, hipMemcpyDeviceToHost, c->stream));
    LF_HIP(hipStreamSynchronize(c->stream));
    return LF_OK;
}

int lf_downstream_device(lf_router *r, const double *x_pix_dev, double *out_pix_dev)
{
    if (!r || !x_pix_dev || !out_pix_dev || x_pix_dev == out_pix_dev) return lf_set_error(LF_E_INVALID, "bad argument");
    lf_router_view V;
    LF_TRY(ensure_parent(r, &V));
    if (V.N == 0) return LF_OK;
    hipLaunchKernelGGL(k_downstream, dim3(blocks_for(V.N)), dim3(kBlock), 0, V.ctx->stream, (long long)V.N, V.perm,
                       *V.parent_slot, x_pix_dev, out_pix_dev);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int lf_downstream_host(lf_router *r, const double *x_host, double *out_host)
{
    if (!r || !x_host || !out_host) return lf_set_error(LF_E_INVALID, "null argument");
    lf_router_view V;
    LF_TRY(lf_router_view_of(r, &V));
    if (V.N == 0) return LF_OK;
    lf_dbuf<double> a, b;
    LF_TRY(a.upload(x_host, (size_t)V.N, V.ctx->stream));
    LF_TRY(b.alloc((size_t)V.N));
    LF_TRY(lf_downstream_device(r, a.p, b.p));
    LF_HIP(hipMemcpyAsync(out_host, b.p, sizeof(double) * (size_t)V.N, hipMemcpyDeviceToHost, V.ctx->stream));
    LF_HIP(hipStreamSynchronize(V.ctx->stream));
    return LF_OK;
}

// catchment(ldd, points): labels_pix[p] = points_pix[first cell with a non-zero point on the way downstream of p,
// p included], 0 if there is none.  Device vectors in pixel order, int32.
int lf_catchments_device(lf_router *r, const int32_t *points_pix_dev, int32_t *labels_pix_dev)
{
    if (!r || !points_pix_dev || !labels_pix_dev) return lf_set_error(LF_E_INVALID, "null argument");
    lf_router_view V;
    LF_TRY(ensure_parent(r, &V));
    if (V.N == 0) return LF_OK;
    lf_dbuf<int> a, b;
    int *jump = nullptr;
    LF_TRY(converge_jumps(V, points_pix_dev, a, b, &jump));
    hipLaunchKernelGGL(k_labels, dim3(blocks_for(V.N)), dim3(kBlock), 0, V.ctx->stream, (long long)V.N, V.perm, jump,
                       points_pix_dev, labels_pix_dev);
    LF_HIP(hipGetLastError());
    LF_HIP(hipStreamSynchronize(V.ctx->stream)); // a, b go out of scope
    return LF_OK;
}

// host form with the reference's integer width (Catchments is compared and used as a bincount index: int64 in, int64 out)
int lf_catchments(lf_router *r, const int64_t *points_host, int64_t *labels_host)
{
    if (!r || !points_host || !labels_host) return lf_set_error(LF_E_INVALID, "null argument");
    lf_router_view V;
    LF_TRY(lf_router_view_of(r, &V));
    const int64_t n = V.N;
    if (n == 0) return LF_OK;
    std::vector<int32_t> h(n);
    for (int64_t i = 0; i < n; ++i) {
        if (points_host[i] < INT32_MIN || points_host[i] > INT32_MAX)
            return lf_set_error(LF_E_INVALID, "point id %lld does not fit 32 bits", (long long)points_host[i]);
        h[i] = (int32_t)points_host[i];
    }
    lf_dbuf<int32_t> pts, lab;
    LF_TRY(pts.upload(h.data(), (size_t)n, V.ctx->stream));
    LF_TRY(lab.alloc((size_t)n));
    LF_TRY(lf_catchments_device(r, pts.p, lab.p));
    LF_HIP(hipMemcpyAsync(h.data(), lab.p, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, V.ctx->stream));
    LF_HIP(hipStreamSynchronize(V.ctx->stream));
    for (int64_t i = 0; i < n; ++i) labels_host[i] = h[i];
    return LF_OK;
}

// np.take(np.bincount(Catchments, weights=w), Catchments) for Catchments = catchment(Ldd, pits) (routing.py:168-171,
// 483-499): every cell gets the total of w over its whole tree.  accuflux carries the tree's total to its outlet
// (upstream first, then the cell: not np.bincount's ascending-pixel order -- equal to rounding, not to the bit), every
// cell then reads the outlet its walk downstream ends in.  Device vectors in pixel order.
int lf_catchment_totals_multi_device(lf_router *r, int nv, const double *const *w_pix_dev, double *const *out_pix_dev)
{
    if (!r || !w_pix_dev || !out_pix_dev || nv < 1 || nv > kMaxTotals) return lf_set_error(LF_E_INVALID, "bad argument");
    for (int v = 0; v < nv; ++v)
        if (!w_pix_dev[v] || !out_pix_dev[v]) return lf_set_error(LF_E_INVALID, "null argument");
    lf_router_view V;
    LF_TRY(ensure_parent(r, &V));
    const int64_t n = V.N;
    if (n == 0) return LF_OK;
    hipStream_t s = V.ctx->stream;
    if (!*V.root_slot) { // once per router: the outlet of every cell, by pointer jumping (the only synchronising part)
        lf_dbuf<int> a, b;
        int *jump = nullptr;
        LF_TRY(converge_jumps(V, nullptr, a, b, &jump));
        LF_TRY(lf_router_alloc_root(r));
        LF_TRY(lf_router_view_of(r, &V));
        hipError_t e = hipMemcpyAsync(*V.root_slot, jump, sizeof(int) * (size_t)n, hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s); // a, b go out of scope
        if (e != hipSuccess) { // the table is published only complete: later calls trust a non-null slot
            (void)lf_router_drop_root(r);
            return lf_set_error(LF_E_HIP, "outlet table of the catchment totals: %s", hipGetErrorString(e));
        }
    }
    double *scratch = nullptr;
    LF_TRY(lf_router_totals_scratch(r, (size_t)(2 * nv) * (size_t)n, &scratch));
    totals_multi T;
    const double *x[kMaxTotals];
    double *acc[kMaxTotals];
    for (int v = 0; v < kMaxTotals; ++v) {
        const int u = v < nv ? v : 0;
        T.w_pix[v] = w_pix_dev[u];
        T.out_pix[v] = out_pix_dev[u];
        T.w_ord[v] = scratch + (size_t)(2 * u) * (size_t)n;
        T.acc_ord[v] = acc[v] = scratch + (size_t)(2 * u + 1) * (size_t)n;
        x[v] = T.w_ord[v];
    }
    hipLaunchKernelGGL(k_totals_gather, dim3(blocks_for(n)), dim3(kBlock), 0, s, (long long)n, nv, V.perm, T);
    LF_TRY(lf_accuflux_ordered_multi_device(r, nv, x, acc));
    hipLaunchKernelGGL(k_totals_take, dim3(blocks_for(n)), dim3(kBlock), 0, s, (long long)n, nv, V.perm, *V.root_slot, T);
    LF_HIP(hipGetLastError());
    return LF_OK; // asynchronous on the library stream: no allocation, no synchronisation after the first call
}

int lf_catchment_totals_device(lf_router *r, const double *w_pix_dev, double *out_pix_dev)
{
    const double *w[1] = {w_pix_dev};
    double *o[1] = {out_pix_dev};
    LF_TRY(lf_catchment_totals_multi_device(r, 1, w, o));
    // synchronous, as it has been since round 1: C callers read out_pix_dev from the host or from another stream right
    // after the call (the asynchronous form is lf_catchment_totals_multi_device with nv = 1)
    lf_router_view V;
    LF_TRY(lf_router_view_of(r, &V));
    LF_HIP(hipStreamSynchronize(V.ctx->stream));
    return LF_OK;
}

// nv vectors of N doubles back to back, host memory
int lf_catchment_totals_multi_host(lf_router *r, int nv, const double *w_host, double *out_host)
{
    if (!r || !w_host || !out_host || nv < 1 || nv > kMaxTotals) return lf_set_error(LF_E_INVALID, "bad argument");
    lf_router_view V;
    LF_TRY(lf_router_view_of(r, &V));
    const size_t n = (size_t)V.N;
    if (n == 0) return LF_OK;
    lf_dbuf<double> a, b;
    LF_TRY(a.upload(w_host, n * (size_t)nv, V.ctx->stream));
    LF_TRY(b.alloc(n * (size_t)nv));
    const double *w[kMaxTotals];
    double *o[kMaxTotals];
    for (int v = 0; v < nv; ++v) {
        w[v] = a.p + (size_t)v * n;
        o[v] = b.p + (size_t)v * n;
    }
    LF_TRY(lf_catchment_totals_multi_device(r, nv, w, o));
    LF_HIP(hipMemcpyAsync(out_host, b.p, sizeof(double) * n * (size_t)nv, hipMemcpyDeviceToHost, V.ctx->stream));
    LF_HIP(hipStreamSynchronize(V.ctx->stream));
    return LF_OK;
}

int lf_catchment_totals_host(lf_router *r, const double *w_host, double *out_host)
{
    return lf_catchment_totals_multi_host(r, 1, w_host, out_host);
}

} // extern "C"
