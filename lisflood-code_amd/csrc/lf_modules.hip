// lf_modules.hip -- element-wise device code of the module-level methods around the kernels:
//   soilloop.dynamic_canopy  (soilloop.py:519-627)      -> k_canopy
//   soilloop.dynamic_soil's ESMax (soilloop.py:638)     -> k_scale_rows
//   surface_routing.dynamic  (surface_routing.py:115-212) -> k_surface_pre / k_surface_post + 3 router calls
// The reference spends ~40 numpy passes per vegetation fraction here; each method is one pass on the GPU.
#include <cmath>
#include <cstdlib>

#include "lf_common.h"
#include "lf_canopy.h"
#include "lf_math.h"
#include "lf_structures.h"

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
    for (int veg = 0; veg < (int)A.V; ++veg) { // the body of a column: lf_canopy.h (shared with the land-surface form of the soil kernel)
        const long long i = (long long)veg * N + pix;
        const long long j = (long long)M.landuse[veg] * N + pix;
        const lf_canopy::column_out o = lf_canopy::column(A, veg, pix, i, j, rain, ewref, etref, frozen);
        A.W1a[j] = o.w1a;
        A.W1b[j] = o.w1b;
        A.W1[i] = o.w1;
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
// opensealed.dynamic + soil.dynamic_perpixel + groundwater.dynamic, one lane per pixel
// Every diagnostic output is optional: a NULL pointer means the map is not reported, it is then not computed and the
// [3,N] vectors only it needs are not read (uniform branches on kernel arguments).  What the rest of the model step needs
// -- DirectRunoff, UZOutflowPixel, LZOutflowToChannelPixel, the states CumInterSealed and LZ -- is always computed.
// ALL: every optional map is reported (what HotPathDevice does by default, as the reference computes them every step).
// The kernel is then straight code and -- the point -- requests EVERY input of the pixel before any arithmetic: written
// block by block (inputs of a map, its arithmetic, its store, the next map) each wavefront walks ~20 dependent round
// trips, because the struct's pointers may alias and no load may pass the store before it; 8 wavefronts x 3 loads in
// flight per SIMD.  With ~70 loads in flight per lane (3 wavefronts per SIMD) the same bytes move at the rate of
// independent streams (tools/micro/stream_count.hip).  Same operations per map, same bits.
template <bool ALL>
__global__ void __launch_bounds__(kBlock) k_pixel_aggregates(lf_pixel_args A)
{
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    const long long N = A.N;
    if (p >= N) return;
#define LF_ST(X, v)                                                                                                      \
    do {                                                                                                                 \
        if (ALL || A.X) A.X[p] = (v);                                                                                    \
    } while (0)
    struct tri {
        double a, b, c;
    };
    auto ld3 = [&](const double *X) { return tri{X[p], X[N + p], X[2 * N + p]}; };
    tri t_taint = {}, t_ta = {}, t_es = {}, t_pref = {}, t_inf = {}, t_w1a = {}, t_w1b = {}, t_w2 = {}, t_sdt = {}, t_sa = {},
        t_sb = {}, t_sg = {}, t_th1a = {}, t_th1b = {}, t_th2 = {}, t_uzo = {}, t_gwp = {};
    double c_taint = 0, c_ta = 0, c_es = 0, c_lzin = 0, c_gwloss = 0;
    if (ALL) {
        t_taint = ld3(A.TaInterception); t_ta = ld3(A.Ta); t_es = ld3(A.ESAct); t_pref = ld3(A.PrefFlow);
        t_inf = ld3(A.Infiltration); t_w1a = ld3(A.W1a); t_w1b = ld3(A.W1b); t_w2 = ld3(A.W2); t_sdt = ld3(A.SoilDepthTotal);
        t_sa = ld3(A.SeepTopToSubA); t_sb = ld3(A.SeepTopToSubB); t_sg = ld3(A.SeepSubToGW); t_th1a = ld3(A.Theta1a);
        t_th1b = ld3(A.Theta1b); t_th2 = ld3(A.Theta2); t_uzo = ld3(A.UZOutflow); t_gwp = ld3(A.GwPercUZLZ);
        c_taint = A.TaInterceptionCUM[p]; c_ta = A.TaCUM[p]; c_es = A.ESActCUM[p]; c_lzin = A.LZInflowCUM[p];
        c_gwloss = A.GwLossCUM[p];
    }
    // ---- opensealed.py:45-70 ----
    const double ewref = A.EWRef[p], drf = A.DirectRunoffFraction[p], wf = A.WaterFraction[p];
    const double in_rain = A.Rain[p], in_snow = A.SnowMelt[p], in_cis = A.CumInterSealed[p], in_smax = A.SMaxSealed[p];
    const double f0 = A.SoilFraction[p], f1 = A.SoilFraction[N + p], f2 = A.SoilFraction[2 * N + p];
    const double in_lz = A.LZ[p], in_lzk = A.LowerZoneK[p], in_lzthr = A.LZThreshold[p], in_gwloss = A.GwLossStep[p];
    const double rsm = npmax(in_rain + in_snow, 0.0);
    double ewact = npmin(ewref, rsm);
    ewact = npmax(ewact * 1.0, 0.0);
    double cis = in_cis;
    double inter = npmax(in_smax - cis, 0.0);
    inter = npmin(inter, rsm);
    cis += inter;
    const double tas = npmax(npmin(cis, ewref), 0.0);
    cis = npmax(cis - tas, 0.0);
    LF_ST(RainSnowmelt, rsm);
    LF_ST(EWaterAct, ewact);
    LF_ST(InterSealed, inter);
    LF_ST(TASealed, tas);
    A.CumInterSealed[p] = cis;
    A.DirectRunoff[p] = drf * (rsm - inter) + wf * (rsm - ewact);
    // ---- soil.py:475-513 : deffraction(X) = (SoilFraction * X).sum(vegetation) = ((f0 x0 + f1 x1) + f2 x2) ----
#define LF_DEF(X, T) (ALL ? ((f0 * T.a + f1 * T.b) + f2 * T.c) : ((f0 * A.X[p] + f1 * A.X[N + p]) + f2 * A.X[2 * N + p]))
    if (ALL || A.TaInterceptionAll || A.TaInterceptionCUM) {
        const double ta_int_all = LF_DEF(TaInterception, t_taint) + drf * tas;
        LF_ST(TaInterceptionAll, ta_int_all);
        if (ALL) A.TaInterceptionCUM[p] = c_taint + ta_int_all;
        else if (A.TaInterceptionCUM) A.TaInterceptionCUM[p] += ta_int_all;
    }
    if (ALL || A.TaPixel || A.TaCUM) {
        const double ta_pix = LF_DEF(Ta, t_ta);
        LF_ST(TaPixel, ta_pix);
        if (ALL) A.TaCUM[p] = c_ta + ta_pix;
        else if (A.TaCUM) A.TaCUM[p] += ta_pix;
    }
    if (ALL || A.ESActPixel || A.ESActCUM) {
        const double es_pix = LF_DEF(ESAct, t_es) + wf * ewact;
        LF_ST(ESActPixel, es_pix);
        if (ALL) A.ESActCUM[p] = c_es + es_pix;
        else if (A.ESActCUM) A.ESActCUM[p] += es_pix;
    }
    if (ALL || A.PrefFlowPixel) A.PrefFlowPixel[p] = LF_DEF(PrefFlow, t_pref);
    if (ALL || A.InfiltrationPixel) A.InfiltrationPixel[p] = LF_DEF(Infiltration, t_inf);
    if (ALL || A.Theta || A.ThetaAll) {
        double th[3];
        const double fv[3] = {f0, f1, f2};
        const double w1a[3] = {t_w1a.a, t_w1a.b, t_w1a.c}, w1b[3] = {t_w1b.a, t_w1b.b, t_w1b.c}, w2[3] = {t_w2.a, t_w2.b, t_w2.c},
                     sdt[3] = {t_sdt.a, t_sdt.b, t_sdt.c};
#pragma unroll
        for (int v = 0; v < 3; ++v) {
            const long long i = v * N + p;
            const double tot_sm = ALL ? w1a[v] + w1b[v] + w2[v] : A.W1a[i] + A.W1b[i] + A.W2[i];
            th[v] = (ALL ? fv[v] : A.SoilFraction[i]) * tot_sm / (ALL ? sdt[v] : A.SoilDepthTotal[i]);
            if (ALL || A.Theta) A.Theta[i] = th[v];
        }
        const double fsum = (f0 + f1) + f2;
        LF_ST(ThetaAll, (fsum > 0) ? ((th[0] + th[1]) + th[2]) / fsum : 0.0);
    }
    if (ALL || A.SeepTopToSubPixelA) A.SeepTopToSubPixelA[p] = LF_DEF(SeepTopToSubA, t_sa);
    if (ALL || A.SeepTopToSubPixelB) A.SeepTopToSubPixelB[p] = LF_DEF(SeepTopToSubB, t_sb);
    if (ALL || A.SeepSubToGWPixel) A.SeepSubToGWPixel[p] = LF_DEF(SeepSubToGW, t_sg);
    if (ALL || A.Theta1aPixel) A.Theta1aPixel[p] = LF_DEF(Theta1a, t_th1a);
    if (ALL || A.Theta1bPixel) A.Theta1bPixel[p] = LF_DEF(Theta1b, t_th1b);
    if (ALL || A.Theta2Pixel) A.Theta2Pixel[p] = LF_DEF(Theta2, t_th2);
    // ---- groundwater.py:137-180 ----
    double lz = in_lz;
    double lzout = npmin(in_lzk * lz, lz - in_lzthr);
    lzout = npmax(lzout, 0.0);
    LF_ST(LZOutflow, lzout);
    lz -= lzout;
    A.UZOutflowPixel[p] = LF_DEF(UZOutflow, t_uzo);
    const double perc = LF_DEF(GwPercUZLZ, t_gwp);
    LF_ST(GwPercUZLZPixel, perc);
    lz += perc;
    const double loss = npmax(npmin(in_gwloss, lz), 0.0);
    lz = lz - loss;
    LF_ST(GwLossLZ, loss);
    A.LZ[p] = lz;
    if (ALL || A.LZInflowCUM) {
        double cum = (ALL ? c_lzin : A.LZInflowCUM[p]) + (perc - loss);
        cum = npmax(cum, 0.0);
        A.LZInflowCUM[p] = cum;
        LF_ST(LZAvInflow, (cum * A.InvDtDay) / A.TimeSinceStart);
    }
    if (ALL) A.GwLossCUM[p] = c_gwloss + loss;
    else if (A.GwLossCUM) A.GwLossCUM[p] += loss;
    A.LZOutflowToChannelPixel[p] = lzout;
#undef LF_DEF
#undef LF_ST
}

// lakes.dynamic_inloop (lakes.py:215-258) and reservoir.dynamic_inloop (reservoir.py:190-296): one lane per site
__global__ void __launch_bounds__(kBlock) k_inloop_sites(lf_inloop_args A)
{
    lf_site_update(A, (long long)blockIdx.x * kBlock + threadIdx.x);
}

// inflow.py:142-144, transmission.py:76-87 and the sideflow assembly routing.py:462-478
__global__ void __launch_bounds__(kBlock) k_inloop_dense(lf_inloop_args A)
{
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= A.N) return;
    double side = A.ToChanM3RunoffDt[p];
    if (A.EvaAddM3Dt) side -= A.EvaAddM3Dt[p];
    if (A.WUseAddM3Dt) side -= A.WUseAddM3Dt[p];
    if (A.QInM3Old) {
        const double qin = (A.QInM3Old[p] + (A.step + 1) * A.QDelta[p]) * A.InvNoRoutSteps;
        A.QInDt[p] = qin;
        A.QinADDEDM3[p] = (A.step < 1 ? 0.0 : A.QinADDEDM3[p]) + qin;
        side += qin;
    }
    if (A.UpTrans) {
        const double q = A.ChanQ[p];
        const double tout = A.UpTrans[p] ? lf_pow_scalar_exponent(lf_pow_scalar_exponent(q, A.TransPower2) - A.TransSub, A.TransPower1) : q;
        const double loss = (q - tout) * A.DtRouting;
        A.TransLossM3Dt[p] = loss;
        A.TransCum[p] += loss;
        side -= loss;
    }
    if (A.QLakeOutM3Dt) side += A.QLakeOutM3Dt[p];
    if (A.QResOutM3Dt) side += A.QResOutM3Dt[p];
    if (A.ChannelToPolderM3Dt) side -= A.ChannelToPolderM3Dt[p];
    A.SideflowChanM3[p] = side;
}
// pressureHead of one layer (soilloop.py:427-432) behind saturationDegree (:378-383), then log10 (:691-695)
__device__ __forceinline__ double pf_layer(double w, bool pore, double wres, double ws, double inv_alpha, double inv_m,
                                           double inv_n, double head_max)
{
    double sat = 0.0;
    if (pore) sat = dmax(dmin((w - wres) / (ws - wres), 1.), 0.); // builtins max / min, as the soil kernel
    double head = head_max;
    if (sat != 0.0) head = dmin(head_max, inv_alpha * pow(pow(1. / sat, inv_m) - 1., inv_n));
    return head > 0 ? log10(head) : -1.;
}

__global__ void __launch_bounds__(kBlock) k_soil_pf(lf_soil_pf_args A, veg_map M)
{
    const long long pix = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (pix >= A.N) return;
    for (int veg = 0; veg < (int)A.V; ++veg) {
        const long long i = (long long)veg * A.N + pix, j = (long long)M.landuse[veg] * A.N + pix;
        A.pF0[i] = pf_layer(A.W1a[i], A.PoreSpaceNotZero1a[j] != 0, A.WRes1a[j], A.WS1a[j], A.GenuInvAlpha1a[j], A.GenuInvM1a[j],
                            A.GenuInvN1a[j], A.HeadMax);
        A.pF1[i] = pf_layer(A.W1b[i], A.PoreSpaceNotZero1b[j] != 0, A.WRes1b[j], A.WS1b[j], A.GenuInvAlpha1b[j], A.GenuInvM1b[j],
                            A.GenuInvN1b[j], A.HeadMax);
        A.pF2[i] = pf_layer(A.W2[i], A.PoreSpaceNotZero2[j] != 0, A.WRes2[j], A.WS2[j], A.GenuInvAlpha2[j], A.GenuInvM2[j],
                            A.GenuInvN2[j], A.HeadMax);
    }
}
} // namespace

extern "C" {

int lf_soil_pf_device(int device, const lf_soil_pf_args *a)
{
    if (!a || !a->index_landuse_all) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (a->V > kMaxVeg || a->V < 0) return lf_set_error(LF_E_INVALID, "too many vegetation rows");
    veg_map M;
    for (int v = 0; v < (int)a->V; ++v) {
        if (a->index_landuse_all[v] < 0 || a->index_landuse_all[v] >= a->L) return lf_set_error(LF_E_INVALID, "bad land-use row");
        M.landuse[v] = (int)a->index_landuse_all[v];
    }
    if (a->N <= 0 || a->V == 0) return LF_OK;
    hipLaunchKernelGGL(k_soil_pf, dim3(blocks_for(a->N)), dim3(kBlock), 0, c->stream, *a, M);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int lf_pixel_aggregates_device(int device, const lf_pixel_args *a)
{
    if (!a) return lf_set_error(LF_E_INVALID, "null argument");
    // always needed: what the rest of the model step reads, and the inputs behind it
    if (!a->SoilFraction || !a->UZOutflow || !a->GwPercUZLZ || !a->Rain || !a->SnowMelt || !a->EWRef || !a->SMaxSealed ||
        !a->DirectRunoffFraction || !a->WaterFraction || !a->LowerZoneK || !a->LZThreshold || !a->GwLossStep ||
        !a->CumInterSealed || !a->LZ || !a->DirectRunoff || !a->UZOutflowPixel || !a->LZOutflowToChannelPixel)
        return lf_set_error(LF_E_INVALID, "pixel aggregates: a required vector is NULL");
    // optional outputs (NULL = the map is not reported): the [3,N] input each one reads must be there
    if (((a->TaInterceptionAll || a->TaInterceptionCUM) && !a->TaInterception) || ((a->TaPixel || a->TaCUM) && !a->Ta) ||
        ((a->ESActPixel || a->ESActCUM) && !a->ESAct) || (a->PrefFlowPixel && !a->PrefFlow) ||
        (a->InfiltrationPixel && !a->Infiltration) ||
        ((a->Theta || a->ThetaAll) && (!a->W1a || !a->W1b || !a->W2 || !a->SoilDepthTotal)) ||
        (a->SeepTopToSubPixelA && !a->SeepTopToSubA) || (a->SeepTopToSubPixelB && !a->SeepTopToSubB) ||
        (a->SeepSubToGWPixel && !a->SeepSubToGW) || (a->Theta1aPixel && !a->Theta1a) || (a->Theta1bPixel && !a->Theta1b) ||
        (a->Theta2Pixel && !a->Theta2) || (a->LZAvInflow && !a->LZInflowCUM))
        return lf_set_error(LF_E_INVALID, "pixel aggregates: an output is requested whose input vector is NULL");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    // every optional map reported: the straight-line form with all loads up front (LF_AGG_ALL=0: the block-by-block form)
    const bool all = a->RainSnowmelt && a->EWaterAct && a->InterSealed && a->TASealed && a->TaInterceptionAll && a->TaInterceptionCUM &&
                     a->TaPixel && a->TaCUM && a->ESActPixel && a->ESActCUM && a->PrefFlowPixel && a->InfiltrationPixel && a->Theta &&
                     a->ThetaAll && a->SeepTopToSubPixelA && a->SeepTopToSubPixelB && a->SeepSubToGWPixel && a->Theta1aPixel &&
                     a->Theta1bPixel && a->Theta2Pixel && a->LZOutflow && a->GwPercUZLZPixel && a->GwLossLZ && a->LZInflowCUM &&
                     a->LZAvInflow && a->GwLossCUM;
    const char *e = std::getenv("LF_AGG_ALL");
    if (a->N > 0) {
        if (all && !(e && e[0] == '0'))
            hipLaunchKernelGGL(k_pixel_aggregates<true>, dim3(blocks_for(a->N)), dim3(kBlock), 0, c->stream, *a);
        else
            hipLaunchKernelGGL(k_pixel_aggregates<false>, dim3(blocks_for(a->N)), dim3(kBlock), 0, c->stream, *a);
    }
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int lf_inloop_structures(int device, const lf_inloop_args *a)
{
    if (!a || !a->ChanQ || !a->ToChanM3RunoffDt || !a->SideflowChanM3) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    lf_inloop_args A = *a;
    if (A.n_lakes <= 0) {
        A.n_lakes = 0;
        A.QLakeOutM3Dt = nullptr;
    }
    if (A.n_res <= 0) {
        A.n_res = 0;
        A.QResOutM3Dt = nullptr;
    }
    const int64_t nsites = A.n_lakes + A.n_res;
    if (nsites > 0) hipLaunchKernelGGL(k_inloop_sites, dim3(blocks_for(nsites)), dim3(kBlock), 0, c->stream, A);
    if (A.N > 0) hipLaunchKernelGGL(k_inloop_dense, dim3(blocks_for(A.N)), dim3(kBlock), 0, c->stream, A);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

int lf_canopy_device(int device, const lf_canopy_args *a)
{
    if (!a || !a->index_landuse) return lf_set_error(LF_E_INVALID, "null argument");
    if (a->V > kMaxVeg) return lf_set_error(LF_E_INVALID, "V exceeds %d", kMaxVeg);
    if ((a->WFilla || a->WFillb) && (!a->WFilla || !a->WFillb || !a->WPF3a || !a->WPF3b))
        return lf_set_error(LF_E_INVALID, "wateruse: WFilla, WFillb, WPF3a and WPF3b are needed together");
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

static int surface_step(lf_router *direct_router, lf_router *other_router, lf_router *forest_router, const lf_surface_args *a,
                        int engine_order);

int lf_surface_step(lf_router *direct_router, lf_router *other_router, lf_router *forest_router, const lf_surface_args *a)
{
    return surface_step(direct_router, other_router, forest_router, a, 0);
}

int lf_surface_step_ordered(lf_router *direct_router, lf_router *other_router, lf_router *forest_router,
                            const lf_surface_args *a)
{
    return surface_step(direct_router, other_router, forest_router, a, 1);
}

static int surface_step(lf_router *direct_router, lf_router *other_router, lf_router *forest_router, const lf_surface_args *a,
                        int engine_order)
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
    {   // the three routers share the LDD (surface_routing.py:108-113): one sweep for all of them
        lf_router *rs[3] = {direct_router, other_router, forest_router};
        double *q[3] = {a->OFQDirect, a->OFQOther, a->OFQForest};
        const double *lat[3] = {a->scratch, a->scratch + N, a->scratch + 2 * N};
        LF_TRY(lf_router_route_device_multi(3, rs, q, lat, LF_SECTION_MAIN, engine_order));
    }
    hipLaunchKernelGGL(k_surface_post, grid, block, 0, c->stream, *a);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

} // extern "C"

// ================================================================================================
// Raster-space one-hop upstream reduction: out[cell] = sum of w over the (up to 8) neighbours whose LDD points at
// the cell -- PCRaster upstream(ldd, w) / np.bincount(downstruct, weights=w) (routing.py:159-164, 387;
// lakes.py:215) on a full H x W raster.  A 64 x 16 tile of LDD codes and weights is staged in LDS with a one-cell
// halo, so every global access is a coalesced row segment and the 3 x 3 neighbourhood tests hit LDS.  Neighbours
// are added in ascending source index (NW, N, NE, W, E, SW, S, SE) = the reference's summation order.
// Cells with code 0 (sea / missing) neither send nor (unless pointed at) matter; flow off the raster is dropped.
// ================================================================================================
namespace {
constexpr int kTX = 64, kTY = 16, kRY = 4;   // tile 64 x 16 cells, 64 x 4 threads, 4 rows per thread

__global__ void __launch_bounds__(kTX *kRY) k_upstream_sum_raster(int H, int W, const uint8_t *__restrict__ ldd,
                                                                  const double *__restrict__ w, double *__restrict__ out)
{
    __shared__ uint8_t s_ldd[kTY + 2][kTX + 2];
    __shared__ double s_w[kTY + 2][kTX + 2];
    const int c0 = blockIdx.x * kTX - 1, r0 = blockIdx.y * kTY - 1;
    const int tx = threadIdx.x, ty = threadIdx.y;
    for (int lr = ty; lr < kTY + 2; lr += kRY) {             // one wavefront per staged row: coalesced 512 B
        const int r = r0 + lr;
        const bool rin = r >= 0 && r < H;
        for (int lc = tx; lc < kTX + 2; lc += kTX) {
            const int c = c0 + lc;
            const bool in = rin && c >= 0 && c < W;
            s_ldd[lr][lc] = in ? ldd[(long long)r * W + c] : (uint8_t)0;
            s_w[lr][lc] = in ? w[(long long)r * W + c] : 0.0;
        }
    }
    __syncthreads();
    const int c = blockIdx.x * kTX + tx;
    if (c >= W) return;
    const int lc = tx + 1;
    // keypad code a neighbour at (dr, dc) must carry to drain into this cell: it points by (-dr, -dc)
    //   NW(-1,-1) -> 3 (SE), N(-1,0) -> 2 (S), NE(-1,+1) -> 1 (SW), W(0,-1) -> 6 (E), E(0,+1) -> 4 (W),
    //   SW(+1,-1) -> 9 (NE), S(+1,0) -> 8 (N), SE(+1,+1) -> 7 (NW)
#pragma unroll
    for (int k = 0; k < kTY / kRY; ++k) {
        const int lr = ty + k * kRY + 1, r = blockIdx.y * kTY + ty + k * kRY;
        if (r >= H) break;
        double s = 0.0;
        if (s_ldd[lr - 1][lc - 1] == 3) s += s_w[lr - 1][lc - 1];
        if (s_ldd[lr - 1][lc] == 2) s += s_w[lr - 1][lc];
        if (s_ldd[lr - 1][lc + 1] == 1) s += s_w[lr - 1][lc + 1];
        if (s_ldd[lr][lc - 1] == 6) s += s_w[lr][lc - 1];
        if (s_ldd[lr][lc + 1] == 4) s += s_w[lr][lc + 1];
        if (s_ldd[lr + 1][lc - 1] == 9) s += s_w[lr + 1][lc - 1];
        if (s_ldd[lr + 1][lc] == 8) s += s_w[lr + 1][lc];
        if (s_ldd[lr + 1][lc + 1] == 7) s += s_w[lr + 1][lc + 1];
        out[(long long)r * W + c] = s;
    }
}
} // namespace

extern "C" int lf_upstream_sum_raster_device(int device, const uint8_t *ldd_raster_dev, const double *w_raster_dev,
                                             double *out_raster_dev, int H, int W)
{
    if (!ldd_raster_dev || !w_raster_dev || !out_raster_dev || H <= 0 || W <= 0)
        return lf_set_error(LF_E_INVALID, "bad argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    const dim3 grid((W + kTX - 1) / kTX, (H + kTY - 1) / kTY), block(kTX, kRY);
    hipLaunchKernelGGL(k_upstream_sum_raster, grid, block, 0, c->stream, H, W, ldd_raster_dev, w_raster_dev,
                       out_raster_dev);
    LF_HIP(hipGetLastError());
    return LF_OK;
}
