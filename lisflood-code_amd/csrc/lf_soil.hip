// lf_soil.hip -- soil / vegetation column water balance on gfx950.
// Replaces the numba kernels interception_water_balance (soilloop.py:27-70) and
// soilColumnsWaterBalance (soilloop.py:78-355, helpers 360-396).
//
// Columns (vegetation fraction x pixel) are independent: one lane per column, a workgroup per tile of 256 columns of
// one vegetation fraction.  Every array keeps the reference's [V,N] / [L,N] C-order layout, so lanes of a wavefront
// read / write consecutive fp64 of one row: all ~70 streams are coalesced.  The kernel is HBM-bound (504 B per
// column-step) where columns need one Courant sub-step; the columns that need more are worked three
// lanes per column in LDS (k_soil_fused, phase 2) and, above a trip count, in k_soil_stragglers.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "lf_common.h"
#include "lf_canopy.h"
#include "lf_math.h"

namespace {

constexpr double kMaxSoilSubSteps = 1048576.0; // cap of the per-column Courant sub-step count (see k_soil_fused, :249)

constexpr int kBlock = 256;
constexpr int kMaxVeg = 16;

__device__ __forceinline__ double dmin(double a, double b) { return (b < a) ? b : a; } // builtins.min(a, b)
__device__ __forceinline__ double dmax(double a, double b) { return (b > a) ? b : a; } // builtins.max(a, b)

// interception_water_balance, soilloop.py:27-70
__global__ void __launch_bounds__(kBlock) k_interception(lf_interception_args A)
{
    const long long pix = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (pix >= A.N) return;
    const double rain = A.Rain[pix];
    for (long long veg = 0; veg < A.V; ++veg) {
        const long long i = veg * A.N + pix;
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
        double ta, drain;
        if (cum > 0.) {
            ta = dmax(dmin(cum, A.TaInterceptionMax[i]), 0.);
            cum = dmax(cum - ta, 0.);
            drain = A.drainageK * cum;
            cum = dmax(cum - drain, 0.);
        } else {
            ta = 0.;
            drain = 0.;
        }
        A.Interception[i] = inter;
        A.TaInterception[i] = ta;
        A.LeafDrainage[i] = drain;
        A.CumInterception[i] = cum;
    }
}

// saturationDegree (soilloop.py:378-383) + unsaturatedConductivity (360-367)
// x^y with x in [0, 1] and y > 0: lf_pow_pos (lf_math.h) or OCML pow (LF_GENERAL_POW=1)
template <bool FASTPOW>
__device__ __forceinline__ double powxy(double x, double y)
{
    return FASTPOW ? lf_pow_pos(x, y) : pow(x, y);
}

template <bool FASTPOW>
__device__ __forceinline__ double unsat_k(double w, bool pore, double wres, double ws, double ksat, double inv_m,
                                          double m)
{
    // evaluated for every lane and selected (a divergent branch here would split the sub-step loop into basic blocks
    // and serialise the three layers' dependent chains); without pore space the quotient is discarded
    const double sc = dmax(dmin((w - wres) / (ws - wres), 1.), 0.);
    const double s = pore ? sc : 0.;
    const double t = 1. - powxy<FASTPOW>(1. - powxy<FASTPOW>(s, inv_m), m);
    return ksat * sqrt(s) * (t * t);
}

// The same with the reciprocal of the layer's (ws - wres) worked out once per column instead of once per sub-step: the
// quotient below is the hardware's own division sequence (v_rcp_f64, two Newton steps on the reciprocal, product, one
// correction of the quotient) with the denominator's part hoisted out of the sub-step loop -- the same operations, so the
// same bits wherever the hardware sequence does not rescale its operands (it does for denormal or wildly different
// exponents only: water contents in mm are neither; (w - wres) == 0 gives 0 either way).  8 instructions fewer per
// sub-step, one of them a quarter-rate v_rcp_f64.
#ifndef LF_SOIL_HOISTED_RCP
#define LF_SOIL_HOISTED_RCP 1
#endif
__device__ __forceinline__ double soil_rcp_refined(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(r, fma(-d, r, 1.0), r);
    r = fma(r, fma(-d, r, 1.0), r);
    return r;
}
template <bool FASTPOW>
__device__ __forceinline__ double unsat_k_r(double w, bool pore, double wres, double d, double r, double ksat, double inv_m,
                                            double m)
{
    const double n = w - wres;
    const double q0 = n * r;
    const double q = fma(fma(-d, q0, n), r, q0); // n / d
    const double sc = dmax(dmin(q, 1.), 0.);
    const double s = pore ? sc : 0.;
    const double t = 1. - powxy<FASTPOW>(1. - powxy<FASTPOW>(s, inv_m), m);
    return ksat * sqrt(s) * (t * t);
}

// the three layers of a column at once (lf_pow_pos_n: the dependent chains of the layers interleaved)
template <bool FASTPOW>
__device__ __forceinline__ void unsat_k3(const double (&w)[3], const bool (&pore)[3], const double (&wres)[3],
                                         const double (&ws)[3], const double (&ksat)[3], const double (&inv_m)[3],
                                         const double (&m)[3], double (&k)[3])
{
    if (!FASTPOW) {
#pragma unroll
        for (int l = 0; l < 3; ++l) k[l] = unsat_k<false>(w[l], pore[l], wres[l], ws[l], ksat[l], inv_m[l], m[l]);
        return;
    }
    double s[3], a[3], b[3], t[3];
#pragma unroll
    for (int l = 0; l < 3; ++l) {
        const double sc = dmax(dmin((w[l] - wres[l]) / (ws[l] - wres[l]), 1.), 0.);
        s[l] = pore[l] ? sc : 0.;
    }
    lf_pow_pos_n<3>(s, inv_m, a);
#pragma unroll
    for (int l = 0; l < 3; ++l) a[l] = 1. - a[l];
    lf_pow_pos_n<3>(a, m, b);
#pragma unroll
    for (int l = 0; l < 3; ++l) t[l] = 1. - b[l];
#pragma unroll
    for (int l = 0; l < 3; ++l) k[l] = ksat[l] * sqrt(s[l]) * (t[l] * t[l]);
}

struct veg_plan {
    int mode[kMaxVeg];       // 0 skip, 1 all pixels, 2 only pixels with paddy_inactive[row] set
    int landuse[kMaxVeg];    // index_landuse_all
    int drained[kMaxVeg];    // is_drained_irrigation
    int paddy_row[kMaxVeg];
};

// ---- soilColumnsWaterBalance (soilloop.py:78-355) --------------------------------------------------------------
// ONE streaming launch in which every stream of the call is read once and every output written once, in FULL lines
// (a workgroup owns the 256 columns of its tile from the first load to the last store), plus a small second launch for
// the few columns whose Courant sub-step count is far above the rest:
//   phase 1  one lane per column: inputs, evaporation, infiltration, the first conductivities and the Courant
//            number (soilloop.py:123-249).  The four outputs the sub-step loop does not touch are stored here.  A
//            column that needs one sub-step takes it in its lane; the others put what the loop needs (7 values per
//            soil layer) in LDS.
//   phase 2  the tile's multi-sub-step columns, sorted by trip count (heaviest first), THREE lanes per column -- one
//            per soil layer; the only traffic between the layers of a column is the capacity of the layer below and
//            the flux from the layer above (two DPP row shifts per sub-step).  A wavefront carries 20 columns and a
//            layer's lane needs ~60 registers instead of ~170 for the three layers side by side.
//   phase 3  every lane, back in column order: state update, diagnostics, upper zone (soilloop.py:313-354) and 18
//            coalesced stores.
//   stragglers  a tile waits for its slowest column (the trip counts are heavy-tailed: median 4, one in ten above 29,
//            up to 92 on the wet synthetic soil), and a compute unit holds only three tiles at a time, so columns above
//            `trip_cap` sub-steps leave the tile: their state after phase 1 (what the loop and phase 3 need, 39 values)
//            goes to a record in a staging area, the tile stores placeholders for them in its full-line stores, and
//            k_soil_stragglers -- which pools the stragglers of 16 tiles, sorts them by trip count and runs them three
//            lanes per column at high occupancy -- overwrites the placeholders.  (Leaving the stragglers' slots
//            unwritten instead costs more than their share: a line written with a hole is a read-modify-write at the
//            memory, tools/micro/hole_fill.hip -- rounds 1-4 paid that for every deferred column.)
// Arithmetic per column is the reference's, operation by operation (a layer's lane evaluates exactly the terms a
// column's lane would evaluate for that layer), so in-lane, in-tile and straggler columns give the same bits.
constexpr int kClasses = 128;             // trip-count classes of the lists (the count itself, clamped to 127)
constexpr int kLoopFields = 7 * 3 + 1;    // per layer: w, wres, ws, ksat, 1/m, m, k; per column: flags (trip count: s_key)
constexpr int kColsPerWave = 20;          // 3 lanes per column, 5 columns per row of 16 lanes (lane 15 of a row idles)
constexpr int kTile = 256;                // columns per tile
#ifndef LF_STRAG_CAP
#define LF_STRAG_CAP 48 /* (24 until round 6: see kSoilTripCap) */
#endif
constexpr int kStragCap = LF_STRAG_CAP;   // straggler records per tile (more stragglers than that stay in the tile)
constexpr int kStragFields = 40;          // loop record (22) + trip count + lane + 16 values of phase 3
#ifndef LF_STRAG_GROUP
#define LF_STRAG_GROUP 16
#endif
constexpr int kStragGroup = LF_STRAG_GROUP; // tiles pooled by one workgroup of k_soil_stragglers

// value of the lane before / after this one in its row of 16 lanes (DPP row_shr:1 / row_shl:1: a VALU move, no LDS
// round trip -- the exchange sits on the sub-step loop's dependent chain); first / last lane of a row: 0 (bound_ctrl)
__device__ __forceinline__ double row_prev(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x111, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x111, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double row_next(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), 0x101, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), 0x101, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// the sub-step loop (soilloop.py:266-312) of one soil layer of one column, three neighbouring lanes per column (layer 0,
// 1, 2) inside a row of 16 lanes; `trips` is uniform over the wavefront (its heaviest column), a lane's own count is nsub.
// Returns the layer's flux summed over the sub-steps (SeepTopToSubA / SeepTopToSubB / SeepSubToGW).
template <bool FASTPOW>
__device__ __forceinline__ double layer_loop(unsigned int layer, double w, double wres, double ws, double ks, double im,
                                             double m, double k, bool pore, long long nsub, long long trips, double DtDay)
{
    const double dtsub = DtDay / (double)nsub;
    double av = w - wres, wt = w, cap = ws - w, sum = 0.;
#if LF_SOIL_HOISTED_RCP
    const double den = ws - wres, rden = soil_rcp_refined(den);
#endif
    for (long long s = 0; s < trips; ++s) {
#if LF_SOIL_HOISTED_RCP
        if (s > 0) k = unsat_k_r<FASTPOW>(wt, pore, wres, den, rden, ks, im, m);
#else
        if (s > 0) k = unsat_k<FASTPOW>(wt, pore, wres, ws, ks, im, m);
#endif
        const double cap_below = row_next(cap);                // layer + 1 of the same column
        const double limit = (layer == 2u) ? av : cap_below;   // :280-285
        const double flux = dmin(k * dtsub, limit);
        const double flux_above = row_prev(flux);              // layer - 1 of the same column
        const double net = (layer == 0u) ? -flux : flux_above - flux;
        const bool live = s < nsub;
        av = live ? av + net : av;                             // :286-288 (av1a -= fa is av1a + (-fa) exactly)
        wt = av + wres;
        cap = ws - wt;
        sum = live ? sum + flux : sum;
    }
    return sum;
}

// what phase 3 needs of a column besides the three flux sums
struct soil_tail {
    double w1a, w1b, w2, inf, pref, uz, uzout, gwp, ws1a;
    double sd1a, sd1b, sd2, wwp1a, wwp1b, wwp1, wwp2, wfc1a, wfc1b, wfc1, wfc2;
    int flags; // 1 frozen, 2 / 4 / 8 pore space in 1a / 1b / 2
};

// phase 3: soilloop.py:313-354 for one column, 18 stores
__device__ __forceinline__ void soil_finish(const lf_soil_args &A, long long i, bool drained, const soil_tail &T, double sa,
                                            double sb, double sg)
{
    const bool frozen = (T.flags & 1) != 0, pore1a = (T.flags & 2) != 0, pore1b = (T.flags & 4) != 0, pore2 = (T.flags & 8) != 0;
    if (frozen) sa = sb = sg = 0.; // :313-316
    // state update, :319-325 (W1 is taken BEFORE the 1a overflow correction, as the reference does)
    double w1a = T.w1a - sa;
    const double w1b = T.w1b + sa - sb;
    const double w2 = T.w2 + sb - sg;
    const double w1 = w1a + w1b;
    const double inf = T.inf - dmax(w1a - T.ws1a, 0.);
    w1a = dmin(w1a, T.ws1a);
    // upper zone, :342-354 (the outflow before the seepage arrives, :340-341, is phase 1's)
    double uz = T.uz, uzout = T.uzout;
    if (drained) {
        uzout += A.DrainedFraction * sg;
        uz += (1 - A.DrainedFraction) * sg + T.pref;
    } else
        uz += sg + T.pref;
    const double perc = dmin(T.gwp, uz);
    uz = dmax(uz - perc, 0.);
    A.SeepTopToSubA[i] = sa;
    A.SeepTopToSubB[i] = sb;
    A.SeepSubToGW[i] = sg;
    A.Infiltration[i] = inf;
    A.W1a[i] = w1a;
    A.W1b[i] = w1b;
    A.W1[i] = w1;
    A.W2[i] = w2;
    // diagnostics, :330-336.  Nothing on the hot path reads them (the per-pixel Theta averages apart): each is computed
    // and stored only if the caller passed a vector for it (a NULL pointer = the map is not reported; uniform branches)
    if (A.Theta1a) A.Theta1a[i] = pore1a ? w1a / T.sd1a : 0.;
    if (A.Theta1b) A.Theta1b[i] = pore1b ? w1b / T.sd1b : 0.;
    if (A.Theta2) A.Theta2[i] = pore2 ? w2 / T.sd2 : 0.;
    if (A.Sat1a) A.Sat1a[i] = (w1a - T.wwp1a) / (T.wfc1a - T.wwp1a);
    if (A.Sat1b) A.Sat1b[i] = (w1b - T.wwp1b) / (T.wfc1b - T.wwp1b);
    if (A.Sat1) A.Sat1[i] = (w1 - T.wwp1) / (T.wfc1 - T.wwp1);
    if (A.Sat2) A.Sat2[i] = (w2 - T.wwp2) / (T.wfc2 - T.wwp2);
    A.UZOutflow[i] = uzout;
    A.GwPercUZLZ[i] = perc;
    A.UZ[i] = uz;
}

// straggler records: [tile][slot][kStragFields] doubles, a tile's records one contiguous run; per tile a count and per
// record a key (lane | class << 8) for the pool sort
struct soil_strag {
    double *rec;
    unsigned short *key; // [tile][kStragCap]
    unsigned int *count; // [tile]
    int trip_cap;        // columns above this many sub-steps leave their tile (0: none do)
};

// all_list / all_count: every multi-sub-step column of the tile (lane | class << 8), for lf_soil_last_deferred and
// lf_soil_substep_histogram (2 bytes per such column)
// DERIVED: ten of the parameter streams are not read but recomputed from the ones they were made of (soil.py:180-228:
// GenuInvM = 1 / GenuM, WS1 = WS1a + WS1b and the same for WRes1 / WFC1 / WWP1, PoreSpaceNotZero = depth != 0 and WS != 0;
// one IEEE operation each, so the same bits) -- 59 of the 526 bytes a column reads and writes.  The caller vouches for the
// relations (lf_soil_columns_device_derived); the ten pointers are not touched.
// CANOPY (the land-surface form, lf_land_columns_device): the lane first runs soilloop.dynamic_canopy for its column
// (lf_canopy.h -- interception, potential transpiration, water stress, abstraction of transpiration from layers 1a / 1b)
// and carries what the soil water balance reads of it -- LeafDrainage, Interception, W1a, W1b, W1 and ESMax = ESRef *
// LAITerm (soilloop.py:638) -- in registers instead of through HBM: the canopy's 24 streams per column shrink to the 12 the
// soil part does not read anyway, inside a kernel whose sub-step arithmetic leaves the memory system idle part of the time.
// Requires index_landuse[veg] == veg (a column's canopy and soil rows coincide) and every fraction active (mode 1).
template <bool FASTPOW, bool DERIVED, bool CANOPY>
__global__ void __launch_bounds__(kTile) __attribute__((amdgpu_waves_per_eu(3)))
k_soil_fused(lf_soil_args A, veg_plan P, unsigned short *__restrict__ all_list, unsigned int *__restrict__ all_count,
             soil_strag G, unsigned int tile0, unsigned int tiles_per_veg, lf_canopy_args C, const double *__restrict__ ESRef)
{
    constexpr int kLoopCap = kTile / 2; // multi-sub-step columns of a tile handled per round (more -> another round)
    __shared__ unsigned int s_count, s_next, s_all, s_strag;
    __shared__ double s_rec[kLoopFields * kLoopCap];
    __shared__ double s_res[3 * kLoopCap];
    __shared__ unsigned int s_key[kLoopCap];
    __shared__ unsigned short s_order[kLoopCap];
    __shared__ double s_stage[kStragFields * kStragCap]; // slot-major: written out as it lies
    if (threadIdx.x == 0) {
        s_count = 0;
        s_next = 0;
        s_all = 0;
        s_strag = 0;
    }
    __syncthreads();
    const long long N = A.N;
    const double DtDay = A.DtDay;
    const unsigned int tile = tile0 + blockIdx.x; // tiles of a vegetation row one after the other, row after row
    const int veg = (int)(tile / tiles_per_veg);
    const long long pix = (long long)(tile - (unsigned int)veg * tiles_per_veg) * kTile + threadIdx.x;
    const int mode = P.mode[veg];
    bool active = pix < N && mode != 0;
    if (active && mode == 2 && !A.paddy_inactive[(long long)P.paddy_row[veg] * N + pix]) active = false;
    const long long i = (long long)veg * N + pix, j = (long long)P.landuse[veg] * N + pix;

    soil_tail T;
    T.w1a = T.w1b = T.w2 = T.inf = T.pref = T.uz = T.uzout = T.gwp = T.ws1a = 0.;
    T.sd1a = T.sd1b = T.sd2 = 1.;
    T.wwp1a = T.wwp1b = T.wwp1 = T.wwp2 = 0.;
    T.wfc1a = T.wfc1b = T.wfc1 = T.wfc2 = 1.;
    T.flags = 0;
    double sa = 0, sb = 0, sg = 0;
    double k1a = 0, k1b = 0, k2 = 0, nsub_f = 1; // kept for a column whose turn comes in a later round of phase 2
    unsigned int rank = 0xffffffffu;
    if (active) {
        // every input of the column before any arithmetic: ~50 independent loads in flight per lane (the kernel is a
        // stream of ~70 vectors; memory-level parallelism, not ALU, sets the speed of this phase)
        const double in_rain = A.Rain[pix], in_snow = A.SnowMelt[pix];
        double in_leaf, in_int, in_w1a, in_w1b, in_w1, in_esmax;
        // CANOPY: what the canopy of the column reads goes out FIRST and in one go (lf_canopy::load), the soil's own ~45
        // inputs right behind it; the canopy's arithmetic and its stores run while those travel.  (Input by input between
        // its stores, the canopy used to be a chain of eight dependent round trips in front of the soil's requests.)
        lf_canopy::column_in cin = {};
        double c_ewref = 0., c_etref = 0., c_esref = 0.;
        unsigned char c_frozen = 0;
        if (CANOPY) {
            cin = lf_canopy::load(C, veg, i, j);
            c_ewref = C.EWRef[pix];
            c_etref = C.ETRef[pix];
            c_esref = ESRef[pix];
            c_frozen = A.isFrozenSoil[pix];
        } else {
            in_leaf = A.LeafDrainage[i]; in_int = A.Interception[i];
            in_w1a = A.W1a[i]; in_w1b = A.W1b[i]; in_w1 = A.W1[i];
            in_esmax = A.ESMax[i];
        }
        const double in_dslr = A.DSLR[i], in_w2 = A.W2[i];
        const double in_uz = A.UZ[i];
        const double in_store = A.StoreMaxPervious[j];
        const double in_bx = A.b_Xinanjiang[pix], in_pinf = A.PowerInfPot[pix], in_ppref = A.PowerPrefFlow[pix];
        const double in_uzk = A.UpperZoneK[pix];
        T.gwp = A.GwPercStep[pix];
        T.sd1a = A.SoilDepth1a[j]; T.sd1b = A.SoilDepth1b[j]; T.sd2 = A.SoilDepth2[j];
        T.wwp1a = A.WWP1a[j]; T.wwp1b = A.WWP1b[j]; T.wwp2 = A.WWP2[j];
        T.wfc1a = A.WFC1a[j]; T.wfc1b = A.WFC1b[j]; T.wfc2 = A.WFC2[j];
        const double ks1a = A.KSat1a[j], ks1b = A.KSat1b[j], ks2 = A.KSat2[j];
        const double m1a = A.GenuM1a[j], m1b = A.GenuM1b[j], m2 = A.GenuM2[j];
        const double wres1a = A.WRes1a[j], wres1b = A.WRes1b[j], wres2 = A.WRes2[j];
        const double ws1a = A.WS1a[j], ws1b = A.WS1b[j], ws2 = A.WS2[j];
        T.ws1a = ws1a;
        if (CANOPY) {
            const lf_canopy::column_out o = lf_canopy::compute(C, veg, pix, i, j, cin, in_rain, c_ewref, c_etref, c_frozen != 0);
            in_leaf = o.leaf_drainage; in_int = o.interception;
            in_w1a = o.w1a; in_w1b = o.w1b; in_w1 = o.w1;
            in_esmax = c_esref * cin.lai_term;                                   // soilloop.py:638
        }
        const bool is_frozen = A.isFrozenSoil[pix] != 0;
        double im1a, im1b, im2, in_wres1, in_ws1;
        int flags;
        if (DERIVED) {
            im1a = 1 / m1a; im1b = 1 / m1b; im2 = 1 / m2;                      // soil.py:180-182
            in_ws1 = ws1a + ws1b; in_wres1 = wres1a + wres1b;                  // :196, 201
            T.wfc1 = T.wfc1a + T.wfc1b; T.wwp1 = T.wwp1a + T.wwp1b;            // :215, 224
            flags = (is_frozen ? 1 : 0) | ((T.sd1a != 0 && ws1a != 0) ? 2 : 0) | ((T.sd1b != 0 && ws1b != 0) ? 4 : 0) |
                    ((T.sd2 != 0 && ws2 != 0) ? 8 : 0);                        // :226-228
        } else {
            im1a = A.GenuInvM1a[j]; im1b = A.GenuInvM1b[j]; im2 = A.GenuInvM2[j];
            in_ws1 = A.WS1[j]; in_wres1 = A.WRes1[j];
            T.wfc1 = A.WFC1[j]; T.wwp1 = A.WWP1[j];
            flags = (is_frozen ? 1 : 0) | (A.PoreSpaceNotZero1a[j] != 0 ? 2 : 0) | (A.PoreSpaceNotZero1b[j] != 0 ? 4 : 0) |
                    (A.PoreSpaceNotZero2[j] != 0 ? 8 : 0);
        }
        T.flags = flags;
        const bool frozen = (flags & 1) != 0, pore1a = (flags & 2) != 0, pore1b = (flags & 4) != 0, pore2 = (flags & 8) != 0;
        // available water for infiltration, :100,131
        double awi = dmax((in_rain + in_snow) + in_leaf - in_int, 0.);
        // days since last rain, :137-140
        double dslr = in_dslr;
        if (awi > A.AvWaterThreshold)
            dslr = 1;
        else
            dslr += DtDay;
        // bare soil evaporation, :148-163
        double esact, w1a = in_w1a, w1b = in_w1b;
        if (frozen)
            esact = 0.;
        else {
            esact = in_esmax * (sqrt(dslr) - sqrt(dslr - 1));
            esact = dmax(dmin(esact, in_w1 - in_wres1), 0.);
            const double supply1a = w1a - wres1a;
            const double es1a = dmin(esact, supply1a);
            const double es1b = dmax(esact - supply1a, 0.);
            w1a = dmax(w1a - es1a, wres1a);
            w1b = dmax(w1b - es1b, wres1b);
        }
        const double w1_ = w1a + w1b;
        // Xinanjiang infiltration capacity, :168-179
        const double relsat1 = pore1a ? dmin(w1_ / in_ws1, 1.0) : 0.0;
        const double satfrac = 1.0 - powxy<FASTPOW>(1.0 - relsat1, in_bx);
        const double infpot = frozen ? 0.0 : in_store * powxy<FASTPOW>(1. - satfrac, in_pinf) * DtDay;
        // preferential flow, :190-194
        const double pref = powxy<FASTPOW>(relsat1, in_ppref) * awi;
        awi -= pref;
        // infiltration, :201-211
        const double inf = dmax(dmin(awi, infpot), 0.);
        const double test1a = w1a + inf;
        w1a = dmin(ws1a, test1a);
        w1b += dmax(test1a - ws1a, 0.);
        const double w2 = in_w2;
        T.w1a = w1a; T.w1b = w1b; T.w2 = w2; T.inf = inf; T.pref = pref;
        // the outputs the sub-step loop does not touch
        A.DSLR[i] = dslr;
        A.ESAct[i] = esact;
        A.PrefFlow[i] = pref;
        A.AvailableWaterForInfiltration[i] = awi;
        // upper-zone outflow before the seepage arrives, :340-341
        T.uzout = dmin(in_uzk * in_uz, in_uz);
        T.uz = dmax(in_uz - T.uzout, 0.);
        // Van Genuchten conductivities and Courant numbers, :223-249
        {
            const double w_[3] = {w1a, w1b, w2}, wres_[3] = {wres1a, wres1b, wres2}, ws_[3] = {ws1a, ws1b, ws2};
            const double ks_[3] = {ks1a, ks1b, ks2}, im_[3] = {im1a, im1b, im2}, m_[3] = {m1a, m1b, m2};
            const bool pore_[3] = {pore1a, pore1b, pore2};
            double k_[3];
            unsat_k3<FASTPOW>(w_, pore_, wres_, ws_, ks_, im_, m_, k_);
            k1a = k_[0], k1b = k_[1], k2 = k_[2];
        }
        const double av1a = w1a - wres1a, av1b = w1b - wres1b, av2 = w2 - wres2;
        const double ca = (av1a == 0) ? 0. : k1a * DtDay / av1a;
        const double cb = (av1b == 0) ? 0. : k1b * DtDay / av1b;
        const double cg = (av2 == 0) ? 0. : k2 * DtDay / av2;
        const double courant = dmax(dmax(ca, cb), cg);
        // NoSubS = max(1, ceil(Courant / CourantCrit)), :249.  A non-finite or absurd Courant number (zero available
        // water next to a huge conductivity) would make the reference's int conversion overflow and the loop spin for
        // ever: the trip count is capped (documented deviation; CourantCrit > 0 is checked on the host).
        nsub_f = dmin(dmax(1., ceil(courant / A.CourantCrit)), kMaxSoilSubSteps);
#ifdef LF_SOIL_DEBUG_MAXTRIPS /* timing experiments only (tools/build_variant.sh): WRONG results */
        nsub_f = dmin(nsub_f, (double)LF_SOIL_DEBUG_MAXTRIPS);
#endif
        // a frozen column's three seepage sums are set to zero behind the loop (:313-316) and nothing else of the loop is
        // kept: its sub-steps are not run at all (the reference runs them and throws the result away)
        if (frozen) nsub_f = 1.;
        if (nsub_f > 1.) {
            long long c = (long long)nsub_f;
            c = c < kClasses - 1 ? c : kClasses - 1;
            const unsigned short key = (unsigned short)(threadIdx.x | ((int)c << 8));
            all_list[(size_t)tile * kTile + atomicAdd(&s_all, 1u)] = key;
            unsigned int sr = 0xffffffffu;
            if (G.trip_cap > 0 && nsub_f > (double)G.trip_cap) sr = atomicAdd(&s_strag, 1u);
            if (sr < (unsigned int)kStragCap) { // a straggler: its record, then placeholders through phase 3 (sa = sb = sg = 0)
                double *R = s_stage + sr * kStragFields;
                R[0] = w1a; R[1] = wres1a; R[2] = ws1a; R[3] = ks1a; R[4] = im1a; R[5] = m1a; R[6] = k1a;
                R[7] = w1b; R[8] = wres1b; R[9] = ws1b; R[10] = ks1b; R[11] = im1b; R[12] = m1b; R[13] = k1b;
                R[14] = w2; R[15] = wres2; R[16] = ws2; R[17] = ks2; R[18] = im2; R[19] = m2; R[20] = k2;
                R[21] = (double)flags; R[22] = nsub_f; R[23] = (double)threadIdx.x;
                R[24] = inf; R[25] = pref; R[26] = T.uz; R[27] = T.uzout; R[28] = T.gwp;
                R[29] = T.sd1a; R[30] = T.sd1b; R[31] = T.sd2;
                R[32] = T.wwp1a; R[33] = T.wwp1b; R[34] = T.wwp1; R[35] = T.wwp2;
                R[36] = T.wfc1a; R[37] = T.wfc1b; R[38] = T.wfc1; R[39] = T.wfc2;
                G.key[(size_t)tile * kStragCap + sr] = key;
            } else {
                rank = atomicAdd(&s_count, 1u); // LDS: position in the tile's list
                if (rank < (unsigned int)kLoopCap) { // the first round's records (nearly always the only round)
#define REC(f, v) s_rec[(f) * kLoopCap + rank] = (v)
                    REC(0, w1a); REC(1, wres1a); REC(2, ws1a); REC(3, ks1a); REC(4, im1a); REC(5, m1a); REC(6, k1a);
                    REC(7, w1b); REC(8, wres1b); REC(9, ws1b); REC(10, ks1b); REC(11, im1b); REC(12, m1b); REC(13, k1b);
                    REC(14, w2); REC(15, wres2); REC(16, ws2); REC(17, ks2); REC(18, im2); REC(19, m2); REC(20, k2);
                    REC(21, (double)flags);
#undef REC
                    s_key[rank] = (unsigned int)nsub_f;
                }
            }
        } else { // the one sub-step of the column, :266-312 with NoSubS = 1
            sa = dmin(k1a * DtDay, ws1b - w1b);
            sb = dmin(k1b * DtDay, ws2 - w2);
            sg = dmin(k2 * DtDay, av2);
        }
    }
    __syncthreads();
    const unsigned int count = (unsigned int)__builtin_amdgcn_readfirstlane((int)s_count);
    {
        const unsigned int ns = s_strag < (unsigned int)kStragCap ? s_strag : (unsigned int)kStragCap;
        if (threadIdx.x == 0) {
            all_count[tile] = s_all;
            if (G.trip_cap > 0) G.count[tile] = ns;
        }
        double *dst = G.rec + (size_t)tile * kStragCap * kStragFields; // the tile's records as ONE contiguous run
        for (unsigned int idx = threadIdx.x; idx < ns * kStragFields; idx += kTile) dst[idx] = s_stage[idx];
    }
    for (unsigned int base = 0; base < count; base += kLoopCap) { // (uniform over the workgroup; nearly always one round)
        if (base > 0) {
            // a tile with more than kLoopCap multi-sub-step columns: the next kLoopCap of its list.  Their lanes still hold
            // the water contents and the first conductivities; the layer parameters are read again (cache hits).
            __syncthreads(); // the results of the round before have been taken
            if (threadIdx.x == 0) s_next = 0;
            const unsigned int r = rank - base;
            if (r < (unsigned int)kLoopCap) {
#define REC(f, v) s_rec[(f) * kLoopCap + r] = (v)
                REC(0, T.w1a); REC(1, A.WRes1a[j]); REC(2, T.ws1a); REC(3, A.KSat1a[j]); REC(4, DERIVED ? 1 / A.GenuM1a[j] : A.GenuInvM1a[j]); REC(5, A.GenuM1a[j]); REC(6, k1a);
                REC(7, T.w1b); REC(8, A.WRes1b[j]); REC(9, A.WS1b[j]); REC(10, A.KSat1b[j]); REC(11, DERIVED ? 1 / A.GenuM1b[j] : A.GenuInvM1b[j]); REC(12, A.GenuM1b[j]); REC(13, k1b);
                REC(14, T.w2); REC(15, A.WRes2[j]); REC(16, A.WS2[j]); REC(17, A.KSat2[j]); REC(18, DERIVED ? 1 / A.GenuM2[j] : A.GenuInvM2[j]); REC(19, A.GenuM2[j]); REC(20, k2);
                REC(21, (double)T.flags);
#undef REC
                s_key[r] = (unsigned int)nsub_f;
            }
            __syncthreads();
        }
        const unsigned int n = (count - base) < (unsigned int)kLoopCap ? (count - base) : (unsigned int)kLoopCap;
        // descending trip count, ties by list position: entry t goes to place #{entries before it in that order}
        if (threadIdx.x < n) {
            const unsigned int k = s_key[threadIdx.x];
            unsigned int pos = 0;
            for (unsigned int q = 0; q < n; ++q) {
                const unsigned int kq = s_key[q];
                pos += (kq > k || (kq == k && q < threadIdx.x)) ? 1u : 0u;
            }
            s_order[pos] = (unsigned short)threadIdx.x;
        }
        __syncthreads();
        const unsigned int ntasks = (n + kColsPerWave - 1) / kColsPerWave, lane = threadIdx.x & 63u;
        const unsigned int in_row = lane & 15u, col_in_row = in_row / 3u, layer = in_row - 3u * col_in_row;
        const unsigned int col = (lane >> 4) * 5u + col_in_row; // (lane 15 of a row: col_in_row = 5, not a column)
        for (;;) {
            unsigned int t = 0;
            if (lane == 0) t = atomicAdd(&s_next, 1u);
            t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
            if (t >= ntasks) break;
            const unsigned int e = t * kColsPerWave + col;
            const bool valid = col_in_row < 5u && e < n;
            const unsigned int first = s_order[t * kColsPerWave]; // the task's first column is its heaviest
            const unsigned int slot = valid ? s_order[e] : first;
            const double *__restrict__ R = s_rec + (7u * layer) * kLoopCap + slot;
            const double w = R[0], wres = R[kLoopCap], ws = R[2 * kLoopCap], ks = R[3 * kLoopCap], im = R[4 * kLoopCap],
                         m = R[5 * kLoopCap], k0 = R[6 * kLoopCap];
            const int fl = (int)s_rec[21 * kLoopCap + slot];
            const bool pore = ((fl >> (1 + (int)layer)) & 1) != 0;
            const long long trips = (long long)__builtin_amdgcn_readfirstlane((int)s_key[first]);
            const long long nsub = (long long)s_key[slot];
            const double sum = layer_loop<FASTPOW>(layer, w, wres, ws, ks, im, m, k0, pore, valid ? nsub : 0, trips, DtDay);
            if (valid) s_res[layer * kLoopCap + slot] = sum;
        }
        __syncthreads();
        const unsigned int r = rank - base;
        if (r < (unsigned int)kLoopCap) {
            sa = s_res[r];
            sb = s_res[kLoopCap + r];
            sg = s_res[2 * kLoopCap + r];
        }
    }
    if (!active) return;
    soil_finish(A, i, P.drained[veg] != 0, T, sa, sb, sg);
}

// The stragglers of kStragGroup consecutive tiles: sorted by trip count in LDS, 20 columns per wavefront (three lanes per
// column), heaviest first; the lane of layer 0 finishes the column (phase 3 of k_soil_fused) over the placeholders.
template <bool FASTPOW>
__global__ void __launch_bounds__(kTile) k_soil_stragglers(lf_soil_args A, veg_plan P, soil_strag G, unsigned int group0,
                                                           unsigned int ntiles, unsigned int tiles_per_veg)
{
    constexpr int kMaxEntries = kStragGroup * kStragCap;
    static_assert(kClasses <= kTile && kClasses % 64 == 0 && kStragCap < 256 && kStragGroup < 256, "list layout");
    __shared__ unsigned int off[kStragGroup + 1], h[kClasses], base[kClasses], next_task;
    __shared__ unsigned short cls_of[kMaxEntries], rank_of[kMaxEntries], ent_of[kMaxEntries]; // per raw entry
    __shared__ unsigned short entry[kMaxEntries];                                            // (tile in group << 8 | slot), by class
    const unsigned int t0 = (group0 + blockIdx.x) * kStragGroup;
    if (threadIdx.x < kClasses) h[threadIdx.x] = 0;
    if (threadIdx.x == 0) {
        unsigned int t = 0;
        for (int g = 0; g < kStragGroup; ++g) {
            off[g] = t;
            t += (t0 + g < ntiles) ? G.count[t0 + g] : 0;
        }
        off[kStragGroup] = t;
        next_task = 0;
    }
    __syncthreads();
    const unsigned int total = off[kStragGroup];
    if (total == 0) return;
    // class histogram, exclusive scan, scatter (all in LDS)
    for (unsigned int q = threadIdx.x; q < total; q += kTile) {
        unsigned int g = 0;
        while (off[g + 1] <= q) ++g;
        const unsigned int sl = q - off[g];
        const unsigned int cls = G.key[(size_t)(t0 + g) * kStragCap + sl] >> 8;
        cls_of[q] = (unsigned short)cls;
        ent_of[q] = (unsigned short)((g << 8) | sl);
        rank_of[q] = (unsigned short)atomicAdd(&h[cls], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64) { // exclusive scan of the histogram by one wavefront: lane l owns kClasses / 64 consecutive bins
        constexpr int per = kClasses / 64;
        unsigned int mine_sum = 0;
#pragma unroll
        for (int q = 0; q < per; ++q) mine_sum += h[threadIdx.x * per + q];
        unsigned int incl = mine_sum;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int up = __shfl_up(incl, d, 64);
            if ((int)threadIdx.x >= d) incl += up;
        }
        unsigned int acc = incl - mine_sum;
#pragma unroll
        for (int q = 0; q < per; ++q) {
            base[threadIdx.x * per + q] = acc;
            acc += h[threadIdx.x * per + q];
        }
    }
    __syncthreads();
    for (unsigned int q = threadIdx.x; q < total; q += kTile) entry[base[cls_of[q]] + rank_of[q]] = ent_of[q];
    __syncthreads();
    const unsigned int ntasks = (total + kColsPerWave - 1) / kColsPerWave, lane = threadIdx.x & 63u;
    const unsigned int in_row = lane & 15u, col_in_row = in_row / 3u, layer = in_row - 3u * col_in_row;
    const unsigned int col = (lane >> 4) * 5u + col_in_row;
    for (;;) {
        unsigned int t = 0;
        if (lane == 0) t = atomicAdd(&next_task, 1u);
        t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
        if (t >= ntasks) break;
        const unsigned int e = t * kColsPerWave + col;
        const bool valid = col_in_row < 5u && e < total;
        const unsigned int ent = entry[total - 1u - (valid ? e : t * kColsPerWave)]; // heaviest first
        const unsigned int tile = t0 + (ent >> 8);
        const double *__restrict__ R = G.rec + ((size_t)tile * kStragCap + (ent & 0xffu)) * kStragFields;
        const double *__restrict__ L = R + 7u * layer;
        const double w = L[0], wres = L[1], ws = L[2], ks = L[3], im = L[4], m = L[5], k0 = L[6];
        const int fl = (int)R[21];
        const bool pore = ((fl >> (1 + (int)layer)) & 1) != 0;
        const int nsub = valid ? (int)R[22] : 0;
        int trips = nsub; // the wavefront's heaviest column (the classes are clamped, so not simply its first)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int o = __shfl_xor(trips, d, 64);
            trips = o > trips ? o : trips;
        }
        const double sum = layer_loop<FASTPOW>(layer, w, wres, ws, ks, im, m, k0, pore, nsub, trips, A.DtDay);
        const double sb = row_next(sum);
        const double sg = row_next(sb);
        if (valid && layer == 0u) {
            soil_tail T;
            T.w1a = R[0]; T.w1b = R[7]; T.w2 = R[14]; T.ws1a = R[2]; T.flags = fl;
            T.inf = R[24]; T.pref = R[25]; T.uz = R[26]; T.uzout = R[27]; T.gwp = R[28];
            T.sd1a = R[29]; T.sd1b = R[30]; T.sd2 = R[31];
            T.wwp1a = R[32]; T.wwp1b = R[33]; T.wwp1 = R[34]; T.wwp2 = R[35];
            T.wfc1a = R[36]; T.wfc1b = R[37]; T.wfc1 = R[38]; T.wfc2 = R[39];
            const int veg = (int)(tile / tiles_per_veg);
            const long long pix = (long long)(tile - (unsigned int)veg * tiles_per_veg) * kTile + (long long)R[23];
            soil_finish(A, (long long)veg * A.N + pix, P.drained[veg] != 0, T, sum, sb, sg);
        }
    }
}

int make_plan(const lf_soil_args *a, const uint8_t *paddy_any, veg_plan *P)
{
    if (a->V > kMaxVeg) return lf_set_error(LF_E_INVALID, "V = %lld exceeds the supported maximum %d", (long long)a->V, kMaxVeg);
    int count_paddy = 0;
    for (int veg = 0; veg < (int)a->V; ++veg) {
        P->landuse[veg] = (int)a->index_landuse_all[veg];
        if (P->landuse[veg] < 0 || P->landuse[veg] >= a->L)
            return lf_set_error(LF_E_INVALID, "index_landuse_all[%d] = %d out of range", veg, P->landuse[veg]);
        P->paddy_row[veg] = 0;
        if (a->is_paddy_irrig && a->is_paddy_irrig[veg]) { // soilloop.py:107-113
            if (!paddy_any || !paddy_any[count_paddy]) {
                P->mode[veg] = 0; // note: the reference does not advance count_paddy_crop here either
                P->drained[veg] = 0;
                continue;
            }
            P->mode[veg] = 2;
            P->drained[veg] = 0;
            P->paddy_row[veg] = count_paddy++;
        } else {
            P->mode[veg] = 1;
            P->drained[veg] = (a->is_irrigated && a->is_irrigated[veg] && a->DrainedFraction > 0) ? 1 : 0;
        }
    }
    return LF_OK;
}

inline int blocks_for(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }
// default of soil_strag::trip_cap.  Round 6: 6 with 48 straggler records per tile, from 16 with 24.  A tile waits for its
// heaviest in-tile column while three of its four wavefronts idle, so the cap wants to be low -- but with 24 records a tile of
// the wet regime (~35 columns above 8 sub-steps) overflowed, the overflow stayed in the tile at up to 92 sub-steps, and a low
// cap lost there what it won elsewhere (rounds 4-5: "the two regimes want different caps").  With room for 48 the low cap
// wins in both: soil wet 2.25 -> 2.10 ms per 12 M columns, resident step 5000^2 22.2 -> 21.1 ms (early, wet steps) and
// 20.0 -> 18.8 ms (later steps), land-surface stage 13.2 -> 11.8 ms; caps 4 / 5 / 6 / 8 within noise of one another, 64
// records no better than 48.
constexpr int kSoilTripCap = 6;

} // namespace

extern "C" {

int lf_interception_device(int device, const lf_interception_args *a)
{
    if (!a) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    if (a->N > 0 && a->V > 0)
        hipLaunchKernelGGL(k_interception, dim3(blocks_for(a->N)), dim3(kBlock), 0, c->stream, *a);
    LF_HIP(hipGetLastError());
    return LF_OK;
}

// Per-device workspace of the soil call (grow-only): | all_count[ntiles + 4] | all_list[ntiles * 256] | straggler
// count[ntiles + 4] | straggler key[ntiles * kStragCap] | straggler records[ntiles * kStragCap * kStragFields] |
struct soil_ws_layout {
    size_t all_count, all_list, s_count, s_key, s_rec, bytes;
};
static soil_ws_layout soil_layout(size_t ntiles)
{
    auto up = [](size_t x) { return (x + 255) & ~(size_t)255; };
    soil_ws_layout L;
    L.all_count = 0;
    L.all_list = up(sizeof(unsigned int) * (ntiles + 4));
    L.s_count = L.all_list + up(sizeof(unsigned short) * ntiles * kTile);
    L.s_key = L.s_count + up(sizeof(unsigned int) * (ntiles + 4));
    L.s_rec = L.s_key + up(sizeof(unsigned short) * ntiles * kStragCap);
    L.bytes = L.s_rec + sizeof(double) * ntiles * kStragCap * kStragFields;
    return L;
}

// instrumentation: columns of the last lf_soil_columns_device call on `device` that needed more than one Courant sub-step
int lf_soil_last_deferred(int device, int64_t *count)
{
    if (!count) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    *count = 0;
    if (!c->soil_ws || c->soil_ntiles == 0) return LF_OK;
    std::vector<unsigned int> h(c->soil_ntiles);
    LF_HIP(hipMemcpyAsync(h.data(), c->soil_ws, sizeof(unsigned int) * c->soil_ntiles, hipMemcpyDeviceToHost, c->stream));
    LF_HIP(hipStreamSynchronize(c->stream));
    int64_t tot = 0;
    for (unsigned int v : h) tot += v;
    *count = tot;
    return LF_OK;
}

// sub-step histogram of the last call's multi-sub-step columns: hist[k] = columns whose trip count was k (the last bin
// holds k >= nbins - 1; the lists keep the count clamped to kClasses - 1)
int lf_soil_substep_histogram(int device, int64_t *hist, int nbins)
{
    if (!hist || nbins < 2) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    for (int k = 0; k < nbins; ++k) hist[k] = 0;
    if (!c->soil_ws || c->soil_ntiles == 0) return LF_OK;
    const size_t nt = c->soil_ntiles;
    const soil_ws_layout L = soil_layout(nt);
    std::vector<unsigned int> cnt(nt);
    std::vector<unsigned short> lst(nt * kTile);
    LF_HIP(hipMemcpyAsync(cnt.data(), (const char *)c->soil_ws + L.all_count, sizeof(unsigned int) * nt, hipMemcpyDeviceToHost, c->stream));
    LF_HIP(hipMemcpyAsync(lst.data(), (const char *)c->soil_ws + L.all_list, sizeof(unsigned short) * nt * kTile,
                          hipMemcpyDeviceToHost, c->stream));
    LF_HIP(hipStreamSynchronize(c->stream));
    for (size_t t = 0; t < nt; ++t)
        for (unsigned int k = 0; k < cnt[t] && k < (unsigned int)kTile; ++k) {
            const int cls = lst[t * kTile + k] >> 8;
            hist[cls < nbins - 1 ? cls : nbins - 1] += 1;
        }
    return LF_OK;
}

static int soil_columns_device(int device, const lf_soil_args *a, bool derived, const lf_canopy_args *canopy = nullptr,
                               const double *esref = nullptr);

int lf_soil_columns_device(int device, const lf_soil_args *a) { return soil_columns_device(device, a, false); }

// The same for a caller that vouches for the relations of k_soil_fused<.., DERIVED> between its parameter arrays: the ten
// derived ones (GenuInvM1a/1b/2, WS1, WRes1, WFC1, WWP1, PoreSpaceNotZero1a/1b/2) are not read and may be NULL.
int lf_soil_columns_device_derived(int device, const lf_soil_args *a) { return soil_columns_device(device, a, true); }

// The land surface of a model step in one pass: soilloop.dynamic_canopy (soilloop.py:519-627), ESMax = ESRef * LAITerm
// (:638) and soilColumnsWaterBalance (:78-355) -- k_soil_fused<.., CANOPY> + k_soil_stragglers.  `canopy` and `soil` must
// describe the same columns: same V = L, N, identity land-use rows, no paddy fraction, and the vectors both name (W1a, W1b,
// W1, Interception, LeafDrainage, Rain, isFrozenSoil, the WWP / WFC parameters) must be the SAME device vectors; soil->ESMax
// is not read.  Same results, bit for bit, as lf_canopy_device + lf_scale_rows_device + lf_soil_columns_device[_derived].
int lf_land_columns_device(int device, const lf_canopy_args *canopy, const lf_soil_args *soil, const double *ESRef_dev, int derived)
{
    if (!canopy || !soil || !ESRef_dev || !canopy->index_landuse || !soil->index_landuse_all)
        return lf_set_error(LF_E_INVALID, "null argument");
    if (canopy->V != soil->V || canopy->N != soil->N || canopy->L != soil->L || soil->V != soil->L)
        return lf_set_error(LF_E_INVALID, "land columns: canopy and soil must have the same V = L and N");
    for (int v = 0; v < (int)soil->V; ++v)
        if (canopy->index_landuse[v] != v || soil->index_landuse_all[v] != v || (soil->is_paddy_irrig && soil->is_paddy_irrig[v]))
            return lf_set_error(LF_E_INVALID, "land columns: needs index_landuse[v] == v and no paddy fraction (use the "
                                "separate entry points otherwise)");
    if ((canopy->WFilla || canopy->WFillb) && (!canopy->WFilla || !canopy->WFillb || !canopy->WPF3a || !canopy->WPF3b))
        return lf_set_error(LF_E_INVALID, "wateruse: WFilla, WFillb, WPF3a and WPF3b are needed together");
    const bool same = canopy->W1a == soil->W1a && canopy->W1b == soil->W1b && canopy->W1 == soil->W1 &&
                      canopy->Interception == soil->Interception && canopy->LeafDrainage == soil->LeafDrainage &&
                      canopy->Rain == soil->Rain && canopy->isFrozenSoil == soil->isFrozenSoil &&
                      canopy->WWP1a == soil->WWP1a && canopy->WWP1b == soil->WWP1b && canopy->WFC1a == soil->WFC1a &&
                      canopy->WFC1b == soil->WFC1b;
    if (!same) return lf_set_error(LF_E_INVALID, "land columns: canopy and soil arguments name different vectors");
    return soil_columns_device(device, soil, derived != 0, canopy, ESRef_dev);
}

static int soil_columns_device(int device, const lf_soil_args *a, bool derived, const lf_canopy_args *canopy, const double *esref)
{
    if (!a || !a->index_landuse_all) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    veg_plan P;
    LF_TRY(make_plan(a, a->paddy_any, &P));
    if (a->N <= 0 || a->V <= 0) return LF_OK;
    if (!(a->CourantCrit > 0.0) || !(a->DtDay > 0.0))
        return lf_set_error(LF_E_INVALID, "CourantCrit and DtDay must be positive (soilloop.py:249: NoSubS = "
                            "ceil(Courant / CourantCrit))");
    if ((unsigned long long)a->V * (unsigned long long)a->N >= 0xffffffffull)
        return lf_set_error(LF_E_INVALID, "V*N exceeds the 32-bit column id range");
    const unsigned int tiles_per_veg = (unsigned int)((a->N + kTile - 1) / kTile);
    const size_t ntiles = (size_t)tiles_per_veg * (size_t)a->V;
    const soil_ws_layout L = soil_layout(ntiles);
    if (c->soil_ws_bytes < L.bytes) {
        if (c->soil_ws) LF_HIP(hipFree(c->soil_ws));
        c->soil_ws = nullptr;
        c->soil_ws_bytes = 0;
        LF_HIP(hipMalloc(&c->soil_ws, L.bytes));
        c->soil_ws_bytes = L.bytes;
    }
    c->soil_ntiles = ntiles;
    char *ws = (char *)c->soil_ws;
    unsigned int *all_count = (unsigned int *)(ws + L.all_count);
    unsigned short *all_list = (unsigned short *)(ws + L.all_list);
    soil_strag G;
    G.count = (unsigned int *)(ws + L.s_count);
    G.key = (unsigned short *)(ws + L.s_key);
    G.rec = (double *)(ws + L.s_rec);
    // columns above this many Courant sub-steps leave their tile for k_soil_stragglers; LF_SOIL_TRIP_CAP=0: none do
    G.trip_cap = kSoilTripCap;
    if (const char *e = std::getenv("LF_SOIL_TRIP_CAP")) G.trip_cap = (int)std::atol(e) > 0 ? (int)std::atol(e) : 0;
    // LF_GENERAL_POW=1: OCML pow instead of lf_pow_pos (A/B parity and timing)
    const char *force_general = std::getenv("LF_GENERAL_POW");
    const bool fastpow = !(force_general && force_general[0] == '1');
    // (Sending the call out in chunks of tiles with the stragglers of a chunk on a second stream beside the streaming launch
    // of the next chunk was measured and dropped: 2.28 / 2.29 / 2.35 / 2.50 / 2.90 ms for 1 / 2 / 4 / 8 / 16 chunks on the wet
    // synthetic soil -- the streaming launch fills every compute unit, the second stream only gets the chunk tails.  So was
    // a build at four wavefronts per SIMD: the state a lane holds across the sub-step phase spills, 1.72 vs 1.26 ms.)
    const dim3 grid((unsigned)ntiles), block(kTile);
    size_t dyn = 0; // LF_SOIL_DEBUG_LDS=<bytes>: extra LDS per workgroup, to bound the tiles in flight per compute unit (timing experiments)
    if (const char *e = std::getenv("LF_SOIL_DEBUG_LDS")) dyn = (size_t)std::atol(e);
    if (const char *e = std::getenv("LF_SOIL_NO_DERIVED")) derived = derived && e[0] != '1'; // A/B switch
    lf_canopy_args C{};
    if (canopy) C = *canopy;
#define LF_SOIL_LAUNCH(FP, DR, CN)                                                                                        \
    hipLaunchKernelGGL((k_soil_fused<FP, DR, CN>), grid, block, dyn, c->stream, *a, P, all_list, all_count, G, 0u,       \
                       tiles_per_veg, C, esref)
    if (canopy) {
        if (fastpow && derived) LF_SOIL_LAUNCH(true, true, true);
        else if (fastpow) LF_SOIL_LAUNCH(true, false, true);
        else if (derived) LF_SOIL_LAUNCH(false, true, true);
        else LF_SOIL_LAUNCH(false, false, true);
    } else {
        if (fastpow && derived) LF_SOIL_LAUNCH(true, true, false);
        else if (fastpow) LF_SOIL_LAUNCH(true, false, false);
        else if (derived) LF_SOIL_LAUNCH(false, true, false);
        else LF_SOIL_LAUNCH(false, false, false);
    }
#undef LF_SOIL_LAUNCH
    if (G.trip_cap > 0) {
        const dim3 grid2((unsigned)((ntiles + kStragGroup - 1) / kStragGroup));
        if (fastpow)
            hipLaunchKernelGGL(k_soil_stragglers<true>, grid2, block, 0, c->stream, *a, P, G, 0u, (unsigned int)ntiles, tiles_per_veg);
        else
            hipLaunchKernelGGL(k_soil_stragglers<false>, grid2, block, 0, c->stream, *a, P, G, 0u, (unsigned int)ntiles, tiles_per_veg);
    }
    LF_HIP(hipGetLastError());
    return LF_OK;
}

} // extern "C"

// ---- host-buffer forms: stage every array through device memory (PCIe-inclusive) -------------------

namespace {
struct stager {
    hipStream_t s;
    std::vector<void *> bufs; // overflow buffers of this call (freed at its end)
    struct wb {
        void *host;
        void *dev;
        size_t bytes;
    };
    std::vector<wb> writeback;
    lf_device_ctx *ctx = nullptr; // owner of the staging arena; nullptr: every buffer is its own allocation
    size_t used = 0;
    ~stager()
    {
        for (void *p : bufs) (void)hipFree(p);
        if (ctx) ctx->stage_need = used > ctx->stage_need ? used : ctx->stage_need;
    }
    int begin(lf_device_ctx *c)
    {
        ctx = c;
        if (c->stage_need > c->stage_bytes) { // the previous call overflowed: grow now, while nothing is in flight
            LF_HIP(hipStreamSynchronize(s));
            if (c->stage_base) (void)hipFree(c->stage_base);
            c->stage_base = nullptr;
            c->stage_bytes = 0;
            const size_t want = c->stage_need + c->stage_need / 8;
            if (hipMalloc(&c->stage_base, want) == hipSuccess)
                c->stage_bytes = want;
            else
                (void)hipGetLastError(); // no arena: fall back to per-buffer allocations
        }
        return LF_OK;
    }
    int carve(size_t bytes, void **out)
    {
        const size_t need = (bytes + 8 + 255) & ~(size_t)255;
        if (ctx && used + need <= ctx->stage_bytes) {
            *out = (char *)ctx->stage_base + used;
            used += need;
            return LF_OK;
        }
        used += need;
        LF_HIP(hipMalloc(out, need));
        bufs.push_back(*out);
        return LF_OK;
    }
    template <typename T>
    int in(const T *&field, size_t count)
    {
        if (!field) return lf_set_error(LF_E_INVALID, "null array argument");
        void *d = nullptr;
        LF_TRY(carve(count * sizeof(T), &d));
        LF_HIP(hipMemcpyAsync(d, field, count * sizeof(T), hipMemcpyHostToDevice, s));
        field = (const T *)d;
        return LF_OK;
    }
    template <typename T>
    int inout(T *&field, size_t count)
    {
        if (!field) return lf_set_error(LF_E_INVALID, "null array argument");
        void *d = nullptr;
        LF_TRY(carve(count * sizeof(T), &d));
        LF_HIP(hipMemcpyAsync(d, field, count * sizeof(T), hipMemcpyHostToDevice, s));
        writeback.push_back({(void *)field, d, count * sizeof(T)});
        field = (T *)d;
        return LF_OK;
    }
    int finish()
    {
        for (const wb &w : writeback) LF_HIP(hipMemcpyAsync(w.host, w.dev, w.bytes, hipMemcpyDeviceToHost, s));
        LF_HIP(hipStreamSynchronize(s));
        return LF_OK;
    }
};
} // namespace

extern "C" {

int lf_interception_host(int device, const lf_interception_args *a_in)
{
    if (!a_in) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    lf_interception_args a = *a_in;
    stager st{c->stream, {}, {}};
    LF_TRY(st.begin(c));
    const size_t vn = (size_t)(a.V * a.N), n = (size_t)a.N;
    LF_TRY(st.inout(a.Interception, vn));
    LF_TRY(st.inout(a.TaInterception, vn));
    LF_TRY(st.inout(a.LeafDrainage, vn));
    LF_TRY(st.inout(a.CumInterception, vn));
    LF_TRY(st.in(a.LAI, vn));
    LF_TRY(st.in(a.Rain, n));
    LF_TRY(st.in(a.TaInterceptionMax, vn));
    LF_TRY(lf_interception_device(device, &a));
    return st.finish();
}

int lf_soil_columns_host(int device, const lf_soil_args *a_in)
{
    if (!a_in || !a_in->index_landuse_all) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    lf_soil_args a = *a_in;
    stager st{c->stream, {}, {}};
    LF_TRY(st.begin(c));
    const size_t vn = (size_t)(a.V * a.N), ln = (size_t)(a.L * a.N), n = (size_t)a.N;
    // which paddy rows have any inactive pixel (soilloop.py:109)
    std::vector<uint8_t> any;
    int n_paddy = 0;
    if (a.is_paddy_irrig)
        for (int v = 0; v < (int)a.V; ++v) n_paddy += a.is_paddy_irrig[v] != 0;
    if (n_paddy > 0) {
        if (!a.paddy_inactive) return lf_set_error(LF_E_INVALID, "paddy_inactive is required when is_paddy_irrig is set");
        any.assign(n_paddy, 0);
        for (int r = 0; r < n_paddy; ++r)
            for (size_t p = 0; p < n && !any[r]; ++p) any[r] = a.paddy_inactive[(size_t)r * n + p] != 0;
        LF_TRY(st.in(a.paddy_inactive, (size_t)n_paddy * n));
        a.paddy_any = any.data();
    } else {
        a.paddy_inactive = nullptr;
        a.paddy_any = nullptr;
    }
#define LF_IN_L(f) LF_TRY(st.in(a.f, ln))
#define LF_IN_N(f) LF_TRY(st.in(a.f, n))
#define LF_IN_V(f) LF_TRY(st.in(a.f, vn))
#define LF_IO_V(f) LF_TRY(st.inout(a.f, vn))
    LF_IN_L(PoreSpaceNotZero1a); LF_IN_L(PoreSpaceNotZero1b); LF_IN_L(PoreSpaceNotZero2);
    LF_IN_L(KSat1a); LF_IN_L(KSat1b); LF_IN_L(KSat2);
    LF_IN_L(GenuInvM1a); LF_IN_L(GenuInvM1b); LF_IN_L(GenuInvM2);
    LF_IN_L(GenuM1a); LF_IN_L(GenuM1b); LF_IN_L(GenuM2);
    LF_IN_L(WRes1a); LF_IN_L(WRes1b); LF_IN_L(WRes1); LF_IN_L(WRes2);
    LF_IN_L(WWP1a); LF_IN_L(WWP1b); LF_IN_L(WWP1); LF_IN_L(WWP2);
    LF_IN_L(WFC1a); LF_IN_L(WFC1b); LF_IN_L(WFC1); LF_IN_L(WFC2);
    LF_IN_L(SoilDepth1a); LF_IN_L(SoilDepth1b); LF_IN_L(SoilDepth2);
    LF_IN_L(WS1a); LF_IN_L(WS1b); LF_IN_L(WS1); LF_IN_L(WS2); LF_IN_L(StoreMaxPervious);
    LF_IN_N(Rain); LF_IN_N(SnowMelt); LF_IN_N(b_Xinanjiang); LF_IN_N(PowerInfPot); LF_IN_N(PowerPrefFlow);
    LF_IN_N(UpperZoneK); LF_IN_N(GwPercStep); LF_IN_N(isFrozenSoil);
    LF_IN_V(LeafDrainage); LF_IN_V(Interception); LF_IN_V(ESMax);
    LF_IO_V(AvailableWaterForInfiltration); LF_IO_V(DSLR); LF_IO_V(ESAct); LF_IO_V(PrefFlow); LF_IO_V(Infiltration);
    LF_IO_V(W1a); LF_IO_V(W1b); LF_IO_V(W1); LF_IO_V(W2);
    LF_IO_V(Theta1a); LF_IO_V(Theta1b); LF_IO_V(Theta2);
    LF_IO_V(Sat1a); LF_IO_V(Sat1b); LF_IO_V(Sat1); LF_IO_V(Sat2);
    LF_IO_V(SeepTopToSubA); LF_IO_V(SeepTopToSubB); LF_IO_V(SeepSubToGW);
    LF_IO_V(UZOutflow); LF_IO_V(UZ); LF_IO_V(GwPercUZLZ);
#undef LF_IN_L
#undef LF_IN_N
#undef LF_IN_V
#undef LF_IO_V
    LF_TRY(lf_soil_columns_device(device, &a));
    return st.finish();
}

} // extern "C"
