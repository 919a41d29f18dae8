// lf_modules.hip -- element-wise device code of the module-level methods around the kernels:
//   soilloop.dynamic_canopy  (soilloop.py:519-627)      -> k_canopy
//   soilloop.dynamic_soil's ESMax (soilloop.py:638)     -> k_scale_rows
//   surface_routing.dynamic  (surface_routing.py:115-212) -> k_surface_pre / k_surface_post + 3 router calls
// The reference spends ~40 numpy passes per vegetation fraction here; each method is one pass on the GPU.
#include <cmath>

#include "lf_common.h"
#include "lf_math.h"

namespace {
constexpr int kBlock = 256;
constexpr int kMaxVeg = 16;
inline int blocks_for(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }

__device__ __forceinline__ double dmin(double a, double b) { return (b < a) ? b : a; } // builtins.min(a, b)
__device__ __forceinline__ double dmax(double a, double b) { return (b > a) ? b : a; } // builtins.max(a, b)
// numpy semantics (NaN propagates from either argument)
__device__ __forceinline__ double npmin(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b < a ? b : a)); }
__device__ __forceinline__ double npmax(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b > a ? b : a)); }

struct veg_map {
    int landuse[kMaxVeg];
};

__global__ void __launch_bounds__(kBlock) k_canopy(lf_canopy_args A, veg_map M)
{
    const long long pix = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long N = A.N;
    if (pix >= N) return;
    const double rain = A.Rain[pix], ewref = A.EWRef[pix], etref = A.ETRef[pix];
    const bool frozen = A.isFrozenSoil[pix] != 0;
    for (int veg = 0; veg < (int)A.V; ++veg) {
        const long long i = (long long)veg * N + pix;
        const long long j = (long long)M.landuse[veg] * N + pix;
        // --- interception (soilloop.py:531-544, kernel 27-70) ---
        const double one_minus_lt = 1. - A.LAITerm[i];       // :531
        const double ta_max = ewref * one_minus_lt;           // :532
        const double lai = A.LAI[i];
        double smax;
        if (lai <= .1)
            smax = 0.;
        else if (lai <= 43.3)
            smax = 0.935 + 0.498 * lai - 0.00575 * (lai * lai);
        else
            smax = 11.718;
        double cum = A.CumInterception[i], inter;
        if (smax > 0) {
            double v = smax - cum;
            v = dmin(v, smax * (1. - exp(-0.046 * lai * rain / smax)));
            v = dmin(v, rain);
            inter = v;
            cum += inter;
        } else
            inter = 0.;
        double ta_int, drain;
        if (cum > 0.) {
            ta_int = dmax(dmin(cum, ta_max), 0.);
            cum = dmax(cum - ta_int, 0.);
            drain = A.LeafDrainageK * cum;
            cum = dmax(cum - drain, 0.);
        } else {
            ta_int = 0.;
            drain = 0.;
        }
        A.Interception[i] = inter;
        A.TaInterception[i] = ta_int;
        A.LeafDrainage[i] = drain;
        A.CumInterception[i] = cum;
        // --- potential transpiration (:549-556) ---
        const double transpir_max = A.CropCoef[j] * etref * one_minus_lt;
        const double pot = npmax(transpir_max - ta_int, 0.);
        A.potential_transpiration[i] = pot;
        // --- water stress and abstraction (:564-627) ---
        const double cgn = A.CropGroupNumber[j];
        const double e = npmin(0.1 * etref * A.InvDtDay, 1.0);
        double swdf = 1 / (0.76 + 1.5 * e) - 0.10 * (5 - cgn);
        if (cgn <= 2.5) swdf = swdf + (e - 0.6) / (cgn * (cgn + 3));
        swdf = npmax(npmin(swdf, 1.0), 0.);
        const double wwp1 = A.WWP1[j], wwp1a = A.WWP1a[j], wwp1b = A.WWP1b[j];
        const double wcrit1 = ((1 - swdf) * (A.WFC1[j] - wwp1)) + wwp1;
        const double wcrit1a = ((1 - swdf) * (A.WFC1a[j] - wwp1a)) + wwp1a;
        const double wcrit1b = ((1 - swdf) * (A.WFC1b[j] - wwp1b)) + wwp1b;
        const double w1 = A.W1[j]; // the reference indexes W1 by the land-use row here (:592)
        double rws = ((wcrit1 - wwp1) > 0) ? (w1 - wwp1) / (wcrit1 - wwp1) : 1.;
        rws = npmax(npmin(rws, 1.), 0.);
        A.RWS[i] = rws;
        const double transpirable = npmax(w1 - wwp1, 0.);
        double ta = npmin(rws * pot, transpirable);
        if (frozen) ta = 0.;
        A.Ta[i] = ta;
        double w1a = A.W1a[j], w1b = A.W1b[j];
        const double wc1a = npmax(w1a - wcrit1a, 0.), wc1b = npmax(w1b - wcrit1b, 0.);
        double ta1a = npmin(ta, wc1a);
        double rest = npmax(ta - ta1a, 0.);
        double ta1b = npmin(rest, wc1b);
        rest = npmax(rest - ta1b, 0.);
        const double sa = npmax(w1a - ta1a - wwp1a, 0.), sb = npmax(w1b - ta1b - wwp1b, 0.);
        const double st = sa + sb;
        const bool avail = st > 0;
        const double fa = avail ? sa / st : 0., fb = avail ? sb / st : 0.;
        ta1a += fa * rest;
        ta1b += fb * rest;
        w1a -= ta1a;
        w1b -= ta1b;
        A.W1a[j] = w1a;
        A.W1b[j] = w1b;
        A.W1[i] = w1a + w1b; // row of the vegetation fraction (:627)
    }
}

__global__ void __launch_bounds__(kBlock) k_scale_rows(long long V, long long N, const double *__restrict__ row,
                                                       const double *__restrict__ m, double *__restrict__ out)
{
    const long long pix = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (pix >= N) return;
    const double r = row[pix];
    for (long long v = 0; v < V; ++v) out[v * N + pix] = r * m[v * N + pix];
}

// surface_routing.py:122-149: runoff components and the three sideflows (scratch rows: Direct, Other, Forest)
__global__ void __launch_bounds__(kBlock) k_surface_pre(lf_surface_args A)
{
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long N = A.N;
    if (p >= N) return;
    double srs[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) { // one prescribed fraction per land use: the np.sum over it is that single term
        const long long i = l * N + p;
        srs[l] = A.SoilFraction[i] * npmax(A.AvailableWaterForInfiltration[i] - A.Infiltration[i], 0.);
        A.SurfaceRunSoil[i] = srs[l];
    }
    const double direct = A.DirectRunoff[p];
    const double surf = direct + ((srs[0] + srs[1]) + srs[2]); // np.sum over the landuse axis
    A.SurfaceRunoff[p] = surf;
    A.TotalRunoff[p] = surf + A.UZOutflowPixel[p] + A.LZOutflowToChannelPixel[p];
    A.scratch[p] = direct * A.MMtoM3 * A.InvPixelLength * A.InvDtSec;                       // SideflowDirect
    A.scratch[N + p] = (srs[0] + srs[2]) * A.MMtoM3 * A.InvPixelLength * A.InvDtSec;       // Rainfed + Irrigated
    A.scratch[2 * N + p] = srs[1] * A.MMtoM3 * A.InvPixelLength * A.InvDtSec;              // Forest
}

// surface_routing.py:191-212
__global__ void __launch_bounds__(kBlock) k_surface_post(lf_surface_args A)
{
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long N = A.N;
    if (p >= N) return;
    const bool b35 = A.Beta == 0.6;
    const double qd = A.OFQDirect[p], qo = A.OFQOther[p], qf = A.OFQForest[p];
    // OFAlpha rows follow dim_runoff = [Other, Forest, Direct] (Lisflood_initial.py:288-290)
    const double m3d = A.PixelLength * A.OFAlpha[2 * N + p] * (b35 ? lf_pow_3_5(qd) : pow(qd, A.Beta));
    const double m3o = A.PixelLength * A.OFAlpha[p] * (b35 ? lf_pow_3_5(qo) : pow(qo, A.Beta));
    const double m3f = A.PixelLength * A.OFAlpha[N + p] * (b35 ? lf_pow_3_5(qf) : pow(qf, A.Beta));
    A.OFM3Direct[p] = m3d;
    A.OFM3Other[p] = m3o;
    A.OFM3Forest[p] = m3f;
    const double qall = qd + qo + qf, m3all = m3d + m3o + m3f;
    const double tochan = A.IsChannel[p] ? qall * A.DtSec : 0.;
    A.OFToChanM3[p] = tochan;
    A.WaterDepth[p] = m3all * A.M3toMM;
    const double run = (A.UZOutflowPixel[p] + A.LZOutflowToChannelPixel[p]) * A.MMtoM3 + tochan;
    A.ToChanM3Runoff[p] = run;
    A.ToChanM3RunoffDt[p] = run * A.InvNoRoutSteps;
}
} // namespace

extern "C" {

int lf_canopy_device(int device, const lf_canopy_args *a)
{
    if (!a || !a->index_landuse) return lf_set_error(LF_E_INVALID, "null argument");
    if (a->V > kMaxVeg) return lf_set_error(LF_E_INVALID, "V exceeds %d", kMaxVeg);
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    veg_map M;
    for (int v = 0; v < (int)a->V; ++v) {
        M.landuse[v] = (int)a->index_landuse[v];
        if (M.landuse[v] < 0 || M.landuse[v] >= a->L) return lf_set_error(LF_E_INVALID, "index_landuse out of range");
    }
    if (a->N > 0 && a->V > 0) hipLaunchKernelGGL(k_canopy, dim3(blocks_for(a->N)), dim3(kBlock), 0, c->stream, *a, M);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int lf_scale_rows_device(int device, const double *row_dev, const double *m_dev, double *out_dev, int64_t V, int64_t N)
{
    if (!row_dev || !m_dev || !out_dev) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (N > 0 && V > 0)
        hipLaunchKernelGGL(k_scale_rows, dim3(blocks_for(N)), dim3(kBlock), 0, c->stream, (long long)V, (long long)N,
                           row_dev, m_dev, out_dev);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int lf_surface_step(lf_router *direct_router, lf_router *other_router, lf_router *forest_router, const lf_surface_args *a)
{
    if (!direct_router || !other_router || !forest_router || !a) return lf_set_error(LF_E_INVALID, "null argument");
    const int device = lf_router_device(direct_router);
    if (lf_router_device(other_router) != device || lf_router_device(forest_router) != device)
        return lf_set_error(LF_E_INVALID, "the three surface routers must live on one device");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    const int64_t N = a->N;
    if (N == 0) return LF_OK;
    const dim3 grid(blocks_for(N)), block(kBlock);
    hipLaunchKernelGGL(k_surface_pre, grid, block, 0, c->stream, *a);
    // surface_routing.py:151-153 (section defaults to "main_channel")
    LF_TRY(lf_router_route_device(direct_router, a->OFQDirect, a->scratch, LF_SECTION_MAIN));
    LF_TRY(lf_router_route_device(other_router, a->OFQOther, a->scratch + N, LF_SECTION_MAIN));
    LF_TRY(lf_router_route_device(forest_router, a->OFQForest, a->scratch + 2 * N, LF_SECTION_MAIN));
    hipLaunchKernelGGL(k_surface_post, grid, block, 0, c->stream, *a);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

} // extern "C"
