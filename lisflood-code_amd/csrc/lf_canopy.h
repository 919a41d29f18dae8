// lf_canopy.h -- soilloop.dynamic_canopy (soilloop.py:519-627) for ONE (vegetation fraction, pixel) column, shared by
// k_canopy (lf_modules.hip: the module method on its own) and by the land-surface form of the soil kernel
// (lf_soil.hip: k_soil_fused<.., CANOPY>, where the canopy of a column runs in the lane that then takes the column through
// its soil water balance).  One body, so both forms give the same bits.
#pragma once
#include "lf_common.h"

namespace lf_canopy {
__device__ __forceinline__ double bmin(double a, double b) { return (b < a) ? b : a; } // builtins.min(a, b)
__device__ __forceinline__ double bmax(double a, double b) { return (b > a) ? b : a; } // builtins.max(a, b)
// numpy semantics (NaN propagates from either argument)
__device__ __forceinline__ double npmin(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b < a ? b : a)); }
__device__ __forceinline__ double npmax(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b > a ? b : a)); }

struct column_out {
    double interception, leaf_drainage; // what soilColumnsWaterBalance reads of the canopy (soilloop.py:643-647)
    double w1a, w1b, w1;                // soil moisture after the abstraction of transpiration (:619-627)
};

// Everything column() reads of the column, as loaded: requested in one go, before any store of the column (the struct's
// pointers may alias, so a load written behind a store stays behind it, and on this hardware the wait for such a load is
// a wait for the store before it too: written input by input between the stores the canopy was a chain of eight dependent
// round trips per column -- in the land-surface kernel, with three tiles resident per compute unit, eight round trips of
// every tile's lifetime).
struct column_in {
    double lai_term, lai, cum, crop_coef, cgn, wwp1, wwp1a, wwp1b, wfc1, wfc1a, wfc1b, w1, w1a, w1b, wpf3a, wpf3b;
};
__device__ __forceinline__ column_in load(const lf_canopy_args &A, int veg, long long i, long long j)
{
    column_in c;
    c.lai_term = A.LAITerm[i];
    c.lai = A.LAI[i];
    c.cum = A.CumInterception[i];
    c.crop_coef = A.CropCoef[j];
    c.cgn = A.CropGroupNumber[j];
    c.wwp1 = A.WWP1[j];
    c.wwp1a = A.WWP1a[j];
    c.wwp1b = A.WWP1b[j];
    c.wfc1 = A.WFC1[j];
    c.wfc1a = A.WFC1a[j];
    c.wfc1b = A.WFC1b[j];
    c.w1 = A.W1[j]; // the reference indexes W1 by the land-use row here (:592)
    c.w1a = A.W1a[j];
    c.w1b = A.W1b[j];
    c.wpf3a = c.wpf3b = 0.;
    if (A.WFilla && veg == (int)A.irrigated_veg) { // (uniform over a tile: a vegetation row)
        c.wpf3a = A.WPF3a[j];
        c.wpf3b = A.WPF3b[j];
    }
    return c;
}

// i = veg * N + pix (row of the vegetation fraction), j = landuse * N + pix (row of its land use).  Stores the seven
// canopy outputs (and the option outputs); the three soil-moisture values are RETURNED -- the caller stores them
// (k_canopy) or carries them into the soil water balance of the same column (k_soil_fused).
__device__ __forceinline__ column_out compute(const lf_canopy_args &A, int veg, long long pix, long long i, long long j,
                                              const column_in &in, double rain, double ewref, double etref, bool frozen)
{
    // --- interception (soilloop.py:531-544, kernel 27-70) ---
    const double one_minus_lt = 1. - in.lai_term;        // :531
    const double ta_max = ewref * one_minus_lt;           // :532
    const double lai = in.lai;
    double smax;
    if (lai <= .1)
        smax = 0.;
    else if (lai <= 43.3)
        smax = 0.935 + 0.498 * lai - 0.00575 * (lai * lai);
    else
        smax = 11.718;
    double cum = in.cum, inter;
    if (smax > 0) {
        double v = smax - cum;
        v = bmin(v, smax * (1. - exp(-0.046 * lai * rain / smax)));
        v = bmin(v, rain);
        inter = v;
        cum += inter;
    } else
        inter = 0.;
    double ta_int, drain;
    if (cum > 0.) {
        ta_int = bmax(bmin(cum, ta_max), 0.);
        cum = bmax(cum - ta_int, 0.);
        drain = A.LeafDrainageK * cum;
        cum = bmax(cum - drain, 0.);
    } else {
        ta_int = 0.;
        drain = 0.;
    }
    A.Interception[i] = inter;
    A.TaInterception[i] = ta_int;
    A.LeafDrainage[i] = drain;
    A.CumInterception[i] = cum;
    // --- potential transpiration (:549-556) ---
    const double transpir_max = in.crop_coef * etref * one_minus_lt;
    const double pot = npmax(transpir_max - ta_int, 0.);
    A.potential_transpiration[i] = pot;
    // --- water stress and abstraction (:564-627) ---
    const double cgn = in.cgn;
    const double e = npmin(0.1 * etref * A.InvDtDay, 1.0);
    double swdf = 1 / (0.76 + 1.5 * e) - 0.10 * (5 - cgn);
    if (cgn <= 2.5) swdf = swdf + (e - 0.6) / (cgn * (cgn + 3));
    swdf = npmax(npmin(swdf, 1.0), 0.);
    const double wwp1 = in.wwp1, wwp1a = in.wwp1a, wwp1b = in.wwp1b;
    const double wcrit1 = ((1 - swdf) * (in.wfc1 - wwp1)) + wwp1;
    const double wcrit1a = ((1 - swdf) * (in.wfc1a - wwp1a)) + wwp1a;
    const double wcrit1b = ((1 - swdf) * (in.wfc1b - wwp1b)) + wwp1b;
    const double w1 = in.w1;
    double rws = ((wcrit1 - wwp1) > 0) ? (w1 - wwp1) / (wcrit1 - wwp1) : 1.;
    rws = npmax(npmin(rws, 1.), 0.);
    A.RWS[i] = rws;
    if (A.SoilMoistureStressDays) A.SoilMoistureStressDays[i] = (rws < 1) ? A.DtDay : 0.; // :597-598 (repStressDays)
    if (A.WFilla && veg == (int)A.irrigated_veg) {                                          // :582-587 (wateruse)
        A.WFilla[pix] = npmin(wcrit1a, in.wpf3a);
        A.WFillb[pix] = npmin(wcrit1b, in.wpf3b);
    }
    const double transpirable = npmax(w1 - wwp1, 0.);
    double ta = npmin(rws * pot, transpirable);
    if (frozen) ta = 0.;
    A.Ta[i] = ta;
    double w1a = in.w1a, w1b = in.w1b;
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
    column_out o;
    o.interception = inter;
    o.leaf_drainage = drain;
    o.w1a = w1a;
    o.w1b = w1b;
    o.w1 = w1a + w1b; // row of the vegetation fraction (:627)
    return o;
}

// the two steps in one (k_canopy: the module method on its own)
__device__ __forceinline__ column_out column(const lf_canopy_args &A, int veg, long long pix, long long i, long long j,
                                             double rain, double ewref, double etref, bool frozen)
{
    const column_in in = load(A, veg, i, j);
    return compute(A, veg, pix, i, j, in, rain, ewref, etref, frozen);
}
} // namespace lf_canopy
