// lf_soil.hip -- soil / vegetation column water balance on gfx950.
// Replaces the numba kernels interception_water_balance (soilloop.py:27-70) and
// soilColumnsWaterBalance (soilloop.py:78-355, helpers 360-396).
//
// Columns (vegetation fraction x pixel) are independent: one lane per pixel, the lane walks the V
// vegetation fractions so that the per-pixel inputs (Rain, SnowMelt, isFrozenSoil, b_Xinanjiang, ...)
// are fetched once.  Every array keeps the reference's [V,N] / [L,N] C-order layout, so lanes of a
// wavefront read/write consecutive fp64 of one row: all ~95 streams are coalesced.  The kernel is
// HBM-bound (~500 B per column-step) unless many Courant sub-steps are needed.
#include <cmath>
#include <cstdlib>

#include "lf_common.h"
#include "lf_math.h"

namespace {

constexpr double kMaxSoilSubSteps = 1048576.0; // cap of the per-column Courant sub-step count (see soil_column)


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

// One soil column (vegetation fraction `veg`, pixel `pix`) of soilColumnsWaterBalance, soilloop.py:123-354.
// Nothing is stored before the end, so a column can be abandoned and recomputed later: with DEFER, a column
// that needs more than one Courant sub-step returns that number without writing anything (0 = column done).
// Deferred columns are handed from pass 1 to pass 2 through a staging area: when pass 1 finds that a column needs
// several sub-steps it has the column's state after evaporation / infiltration and the first conductivities in
// registers, and writes that state, with the parameters the loop and the epilogue read, next to the records of the
// tile's other deferred columns (slot = tile * cap + rank in the tile's list, 42 values per slot).  Pass 2 RESUMES at
// the sub-step loop (no second prologue) and reads six lines per column instead of one 64-byte sector per 8-byte value
// scattered over ~46 vectors (measured: 6 GB fetched for 0.8 GB of inputs).  A tile has room for `cap` columns; the
// ones beyond are recomputed from the vectors.
constexpr int kStageFields = 42;
constexpr unsigned int kStageCap = 96; // slots per tile (of 256 columns); pass 1 collects them in LDS: 42 x 96 x 8 B = 32 KB
struct soil_stage {
    double *buf;      // [nslots][kStageFields]: a column's inputs side by side, a tile's columns one contiguous run
    size_t nslots;    // ntiles * cap
    unsigned int cap; // slots per tile, 0 = staging off
};

// DEFER (pass 1): *lds_count is the tile's list counter; a deferred column takes its rank from it, stages its inputs
// and returns (nsub, rank).  !DEFER (pass 2): `slot` < nslots reads the inputs from the staging area.
// STAGE (pass 1 only): compile the staging stores in.  They cost pass 1 registers (spills around a block that every
// wavefront with a deferred lane executes), so the host uses the variant without them while few columns defer.
template <bool DEFER, bool FASTPOW, bool STAGE = false>
__device__ __forceinline__ long long soil_column(const lf_soil_args &A, const veg_plan &P, int veg, long long pix,
                                                 const soil_stage &S, size_t slot, unsigned int *lds_count,
                                                 unsigned int tile, unsigned int *rank_out, double *lds_stage = nullptr)
{
    const long long N = A.N;
    const double DtDay = A.DtDay;
    const long long i = (long long)veg * N + pix, j = (long long)P.landuse[veg] * N + pix;
    const bool staged = !DEFER && slot < S.nslots;
    // everything the sub-step loop and the epilogue need; filled either from the staging area (pass 2: the state pass 1
    // had reached when it found that the column needs several sub-steps) or by the prologue below
    double w1a, w1b, w2, k1a, k1b, k2, inf, pref, awi, dslr, esact, in_uz, in_uzk, in_gwp;
    double in_sd1a, in_sd1b, in_sd2, wwp1a, wwp1b, wwp1, wwp2, in_wfc1a, in_wfc1b, in_wfc1, in_wfc2;
    double ks1a, ks1b, ks2, im1a, im1b, im2, m1a, m1b, m2, wres1a, wres1b, wres2, ws1a, ws1b, ws2;
    double nsub_f;
    int flags;
    if (staged) {
        const double *__restrict__ R = S.buf + slot * kStageFields;
        int f = 0;
#define LDS_(v) v = R[f++]
        LDS_(w1a); LDS_(w1b); LDS_(w2); LDS_(k1a); LDS_(k1b); LDS_(k2);
        LDS_(inf); LDS_(pref); LDS_(awi); LDS_(dslr); LDS_(esact);
        LDS_(in_uz); LDS_(in_uzk); LDS_(in_gwp);
        LDS_(in_sd1a); LDS_(in_sd1b); LDS_(in_sd2);
        LDS_(wwp1a); LDS_(wwp1b); LDS_(wwp1); LDS_(wwp2);
        LDS_(in_wfc1a); LDS_(in_wfc1b); LDS_(in_wfc1); LDS_(in_wfc2);
        LDS_(ks1a); LDS_(ks1b); LDS_(ks2);
        LDS_(im1a); LDS_(im1b); LDS_(im2);
        LDS_(m1a); LDS_(m1b); LDS_(m2);
        LDS_(wres1a); LDS_(wres1b); LDS_(wres2);
        LDS_(ws1a); LDS_(ws1b); LDS_(ws2);
        double flags_d;
        LDS_(flags_d); LDS_(nsub_f);
#undef LDS_
        flags = (int)flags_d;
    } else {
        // Every input of the column is fetched here, before any arithmetic: ~50 independent loads in flight per lane
        // instead of the handful the compiler keeps when loads sit next to their first use (the kernel is a stream
        // of ~90 vectors; memory-level parallelism, not ALU, sets its speed).
        const double in_rain = A.Rain[pix], in_snow = A.SnowMelt[pix], in_leaf = A.LeafDrainage[i], in_int = A.Interception[i];
        const double in_dslr = A.DSLR[i], in_w1a = A.W1a[i], in_w1b = A.W1b[i], in_w1 = A.W1[i], in_w2 = A.W2[i];
        in_uz = A.UZ[i];
        const double in_esmax = A.ESMax[i], in_wres1 = A.WRes1[j], in_ws1 = A.WS1[j], in_store = A.StoreMaxPervious[j];
        const double in_bx = A.b_Xinanjiang[pix], in_pinf = A.PowerInfPot[pix], in_ppref = A.PowerPrefFlow[pix];
        in_uzk = A.UpperZoneK[pix];
        in_gwp = A.GwPercStep[pix];
        in_sd1a = A.SoilDepth1a[j]; in_sd1b = A.SoilDepth1b[j]; in_sd2 = A.SoilDepth2[j];
        wwp1a = A.WWP1a[j]; wwp1b = A.WWP1b[j]; wwp1 = A.WWP1[j]; wwp2 = A.WWP2[j];
        in_wfc1a = A.WFC1a[j]; in_wfc1b = A.WFC1b[j]; in_wfc1 = A.WFC1[j]; in_wfc2 = A.WFC2[j];
        ks1a = A.KSat1a[j]; ks1b = A.KSat1b[j]; ks2 = A.KSat2[j];
        im1a = A.GenuInvM1a[j]; im1b = A.GenuInvM1b[j]; im2 = A.GenuInvM2[j];
        m1a = A.GenuM1a[j]; m1b = A.GenuM1b[j]; m2 = A.GenuM2[j];
        wres1a = A.WRes1a[j]; wres1b = A.WRes1b[j]; wres2 = A.WRes2[j];
        ws1a = A.WS1a[j]; ws1b = A.WS1b[j]; ws2 = A.WS2[j];
        flags = (A.isFrozenSoil[pix] != 0 ? 1 : 0) | (A.PoreSpaceNotZero1a[j] != 0 ? 2 : 0) |
                (A.PoreSpaceNotZero1b[j] != 0 ? 4 : 0) | (A.PoreSpaceNotZero2[j] != 0 ? 8 : 0);
        const bool frozen = (flags & 1) != 0, pore1a = (flags & 2) != 0, pore1b = (flags & 4) != 0, pore2 = (flags & 8) != 0;
        // available water for infiltration, :100,131
        awi = dmax((in_rain + in_snow) + in_leaf - in_int, 0.);
        // days since last rain, :137-140
        dslr = in_dslr;
        if (awi > A.AvWaterThreshold)
            dslr = 1;
        else
            dslr += DtDay;
        // bare soil evaporation, :148-163
        w1a = in_w1a;
        w1b = in_w1b;
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
        pref = powxy<FASTPOW>(relsat1, in_ppref) * awi;
        awi -= pref;
        // infiltration, :201-211
        inf = dmax(dmin(awi, infpot), 0.);
        const double test1a = w1a + inf;
        w1a = dmin(ws1a, test1a);
        w1b += dmax(test1a - ws1a, 0.);
        w2 = in_w2;
        // Van Genuchten conductivities and Courant numbers, :223-249
        {
            const double w_[3] = {w1a, w1b, w2}, wres_[3] = {wres1a, wres1b, wres2}, ws_[3] = {ws1a, ws1b, ws2};
            const double ks_[3] = {ks1a, ks1b, ks2}, im_[3] = {im1a, im1b, im2}, m_[3] = {m1a, m1b, m2};
            const bool pore_[3] = {pore1a, pore1b, pore2};
            double k_[3];
            unsat_k3<FASTPOW>(w_, pore_, wres_, ws_, ks_, im_, m_, k_);
            k1a = k_[0], k1b = k_[1], k2 = k_[2];
        }
        const double av1a_ = w1a - wres1a, av1b_ = w1b - wres1b, av2_ = w2 - wres2;
        const double ca = (av1a_ == 0) ? 0. : k1a * DtDay / av1a_;
        const double cb = (av1b_ == 0) ? 0. : k1b * DtDay / av1b_;
        const double cg = (av2_ == 0) ? 0. : k2 * DtDay / av2_;
        const double courant = dmax(dmax(ca, cb), cg);
        // NoSubS = max(1, ceil(Courant / CourantCrit)), :249.  A non-finite or absurd Courant number (zero available
        // water next to a huge conductivity) would make the reference's int conversion overflow and this loop spin for
        // ever: the trip count is capped (documented deviation; CourantCrit > 0 is checked on the host).
        nsub_f = dmin(dmax(1., ceil(courant / A.CourantCrit)), kMaxSoilSubSteps);
    }
    const bool frozen = (flags & 1) != 0, pore1a = (flags & 2) != 0, pore1b = (flags & 4) != 0, pore2 = (flags & 8) != 0;
    const long long nsub = (long long)nsub_f;
    if (DEFER && nsub > 1) {
        const unsigned int rank = atomicAdd(lds_count, 1u); // LDS: position in the tile's list
        *rank_out = rank;
        if (STAGE && rank < kStageCap) { // into the block's LDS table; the block writes it out in full lines afterwards
            int f = 0;
#define ST(v) lds_stage[(f++) * kStageCap + rank] = (v)
            // the state reached so far (pass 2 resumes at the sub-step loop) and the parameters the loop and the
            // epilogue read; parameters whose registers are dead by now are read again (cache hits, deferred lanes
            // only) rather than kept alive across the infiltration arithmetic: pass 1 has no registers to spare
            ST(w1a); ST(w1b); ST(w2); ST(k1a); ST(k1b); ST(k2);
            ST(inf); ST(pref); ST(awi); ST(dslr); ST(esact);
            ST(in_uz); ST(in_uzk); ST(in_gwp);
            ST(A.SoilDepth1a[j]); ST(A.SoilDepth1b[j]); ST(A.SoilDepth2[j]);
            ST(A.WWP1a[j]); ST(A.WWP1b[j]); ST(A.WWP1[j]); ST(A.WWP2[j]);
            ST(A.WFC1a[j]); ST(A.WFC1b[j]); ST(A.WFC1[j]); ST(A.WFC2[j]);
            ST(ks1a); ST(ks1b); ST(ks2);
            ST(im1a); ST(im1b); ST(im2);
            ST(m1a); ST(m1b); ST(m2);
            ST(wres1a); ST(wres1b); ST(wres2);
            ST(ws1a); ST(ws1b); ST(ws2);
            ST((double)flags); ST(nsub_f);
#undef ST
        }
        return nsub;
    }
    // sub-step loop, :266-312
    double av1a = w1a - wres1a, av1b = w1b - wres1b, av2 = w2 - wres2;
    double cap1 = ws1b - w1b, cap2 = ws2 - w2;
    double wt1a = w1a, wt1b = w1b, wt2 = w2;
    double sa = 0., sb = 0., sg = 0.;
    const double dtsub = DtDay / (double)nsub;
#ifdef LF_SOIL_DEBUG_MAXTRIPS /* timing experiments only: wrong results */
    const long long trips = DEFER ? 1 : (nsub < LF_SOIL_DEBUG_MAXTRIPS ? nsub : LF_SOIL_DEBUG_MAXTRIPS);
#else
    const long long trips = DEFER ? 1 : nsub; // DEFER: nsub == 1 here, the re-evaluation branch disappears
#endif
    for (long long s = 0; s < trips; ++s) {
        if (s > 0) {
            const double w_[3] = {wt1a, wt1b, wt2}, wres_[3] = {wres1a, wres1b, wres2}, ws_[3] = {ws1a, ws1b, ws2};
            const double ks_[3] = {ks1a, ks1b, ks2}, im_[3] = {im1a, im1b, im2}, m_[3] = {m1a, m1b, m2};
            const bool pore_[3] = {pore1a, pore1b, pore2};
            double k_[3];
            unsat_k3<FASTPOW>(w_, pore_, wres_, ws_, ks_, im_, m_, k_);
            k1a = k_[0], k1b = k_[1], k2 = k_[2];
        }
        const double fa = dmin(k1a * dtsub, cap1);
        const double fb = dmin(k1b * dtsub, cap2);
        const double fg = dmin(k2 * dtsub, av2);
        av1a -= fa;
        av1b += fa - fb;
        av2 += fb - fg;
        wt1a = av1a + wres1a;
        wt1b = av1b + wres1b;
        wt2 = av2 + wres2;
        cap1 = ws1b - wt1b;
        cap2 = ws2 - wt2;
        sa += fa;
        sb += fb;
        sg += fg;
    }
    if (frozen) sa = sb = sg = 0.; // :313-316
    // state update, :319-325 (W1 is taken BEFORE the 1a overflow correction, as the reference does)
    w1a -= sa;
    w1b = w1b + sa - sb;
    w2 = w2 + sb - sg;
    const double w1 = w1a + w1b;
    inf -= dmax(w1a - ws1a, 0.);
    w1a = dmin(w1a, ws1a);
    // upper zone, :340-354
    double uz = in_uz;
    double uzout = dmin(in_uzk * uz, uz);
    uz = dmax(uz - uzout, 0.);
    if (P.drained[veg]) {
        uzout += A.DrainedFraction * sg;
        uz += (1 - A.DrainedFraction) * sg + pref;
    } else
        uz += sg + pref;
    const double perc = dmin(in_gwp, uz);
    uz = dmax(uz - perc, 0.);
    // stores
    A.DSLR[i] = dslr;
    A.ESAct[i] = esact;
    A.PrefFlow[i] = pref;
    A.AvailableWaterForInfiltration[i] = awi;
    A.SeepTopToSubA[i] = sa;
    A.SeepTopToSubB[i] = sb;
    A.SeepSubToGW[i] = sg;
    A.Infiltration[i] = inf;
    A.W1a[i] = w1a;
    A.W1b[i] = w1b;
    A.W1[i] = w1;
    A.W2[i] = w2;
    // diagnostics, :330-336
    A.Theta1a[i] = pore1a ? w1a / in_sd1a : 0.;
    A.Theta1b[i] = pore1b ? w1b / in_sd1b : 0.;
    A.Theta2[i] = pore2 ? w2 / in_sd2 : 0.;
    A.Sat1a[i] = (w1a - wwp1a) / (in_wfc1a - wwp1a);
    A.Sat1b[i] = (w1b - wwp1b) / (in_wfc1b - wwp1b);
    A.Sat1[i] = (w1 - wwp1) / (in_wfc1 - wwp1);
    A.Sat2[i] = (w2 - wwp2) / (in_wfc2 - wwp2);
    A.UZOutflow[i] = uzout;
    A.GwPercUZLZ[i] = perc;
    A.UZ[i] = uz;
    return 0;
}

// Deferred columns are kept per TILE (= the 256 columns of one pass-1 block): pass 1 writes the tile's deferred
// lanes (lane index + sub-step class) to a tile-local list with an LDS counter -- no global atomics -- and pass 2
// gives each workgroup kGroup consecutive tiles, whose deferred columns it sorts by class in LDS (wavefronts then
// run similar trip counts) and finishes.  Keeping pass 2 tile-local keeps its gathers inside a 4096-column window
// of every stream instead of scattering 8-byte reads over the whole vectors (measured 14x over-fetch with a
// global, class-sorted list).
// sort key of a deferred column inside its pool: the trip count itself, clamped to kClasses - 1 (LF_SOIL_LOG2_CLASSES=1:
// floor(log2(nsub)) as before -- A/B switch).  With the inputs staged per column, the order inside the pool no longer
// changes which lines a wavefront touches, so the finer the sort the closer the lanes of a wavefront finish together.
constexpr int kClasses = 128;
#ifndef LF_SOIL_GROUP
#define LF_SOIL_GROUP 16
#endif
constexpr int kGroup = LF_SOIL_GROUP;  // tiles per pass-2 workgroup (a pool of 4096 columns: enough deferred columns to fill
                            // wavefronts with similar trip counts even when the sub-step distribution has a long tail)

#ifndef LF_SOIL_P1_WAVES
#define LF_SOIL_P1_WAVES 4
#endif
// waves_per_eu(4): pass 1 streams ~500 B per column and needs the occupancy; the allocator otherwise wobbles
// between 126 and 133 VGPRs (4 vs 3 waves per SIMD) with unrelated edits to this file; 5 waves spill and are slower
template <bool FASTPOW, bool STAGE>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(LF_SOIL_P1_WAVES))) k_soil_columns(lf_soil_args A, veg_plan P, unsigned short *__restrict__ tile_list,
                                                         unsigned int *__restrict__ tile_count, soil_stage S, int log2_classes)
{
    __shared__ unsigned int count;
    __shared__ double lds_stage[STAGE ? kStageFields * kStageCap : 1];
    if (threadIdx.x == 0) count = 0;
    __syncthreads();
    const long long pix = (long long)blockIdx.x * kBlock + threadIdx.x;
    const int veg = blockIdx.y;
    const unsigned int tile = blockIdx.x + blockIdx.y * gridDim.x;
    const int mode = P.mode[veg];
    bool active = pix < A.N && mode != 0;
    if (active && mode == 2 && !A.paddy_inactive[(long long)P.paddy_row[veg] * A.N + pix]) active = false;
    if (active) {
        unsigned int rank = 0;
        const long long nsub = soil_column<true, FASTPOW, STAGE>(A, P, veg, pix, S, 0, &count, tile, &rank, lds_stage);
        if (nsub > 0) {
            long long c = log2_classes ? 63 - __clzll((unsigned long long)nsub) : nsub; // floor(log2(nsub)) >= 1 / nsub >= 2
            c = c < kClasses - 1 ? c : kClasses - 1;
            tile_list[(size_t)tile * kBlock + rank] = (unsigned short)(threadIdx.x | ((int)c << 8));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) tile_count[tile] = count;
    if (STAGE) { // the tile's staged inputs as ONE contiguous run: slot-major, the 42 fields of a column side by side
        const unsigned int n = count < kStageCap ? count : kStageCap;
        double *dst = S.buf + (size_t)tile * kStageCap * kStageFields;
        for (unsigned int idx = threadIdx.x; idx < (unsigned int)kStageFields * n; idx += kBlock) {
            const unsigned int t = idx / kStageFields, f = idx - t * kStageFields;
            dst[idx] = lds_stage[f * kStageCap + t];
        }
    }
}

// Pass 2: the deferred columns of kGroup consecutive tiles, sorted by sub-step class, one lane each.
#ifndef LF_SOIL_P2_WAVES
#define LF_SOIL_P2_WAVES 2
#endif
template <bool FASTPOW>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(LF_SOIL_P2_WAVES)))
k_soil_columns_deferred(lf_soil_args A, veg_plan P,
                                                                  const unsigned short *__restrict__ tile_list,
                                                                  const unsigned int *__restrict__ tile_count,
                                                                  unsigned int ntiles, unsigned int tiles_per_veg,
                                                                  soil_stage S, unsigned long long *__restrict__ deferred_total)
{
    __shared__ unsigned int cnt[kGroup], h[kClasses], base[kClasses], total, next_chunk;
    __shared__ unsigned int entry[kGroup * kBlock]; // (rank in tile list << 16 | tile-in-group << 8 | lane), by class
    const unsigned int t0 = blockIdx.x * kGroup;
    if (threadIdx.x < kGroup) cnt[threadIdx.x] = (t0 + threadIdx.x < ntiles) ? tile_count[t0 + threadIdx.x] : 0;
    if (threadIdx.x < kClasses) h[threadIdx.x] = 0;
    static_assert(kClasses <= kBlock && kClasses % 64 == 0, "histogram layout");
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int t = 0;
        for (int g = 0; g < kGroup; ++g) t += cnt[g];
        total = t;
        next_chunk = 0;
    }
    __syncthreads();
    if (total == 0) return;
    if (threadIdx.x == 0) atomicAdd(deferred_total, (unsigned long long)total); // one per pool: feeds the staging policy
    // class histogram, exclusive scan, scatter (all in LDS)
    unsigned short mine[kGroup];
    unsigned int rank[kGroup];
#pragma unroll
    for (int g = 0; g < kGroup; ++g) {
        mine[g] = 0xffff;
        if (threadIdx.x < cnt[g]) {
            mine[g] = tile_list[(size_t)(t0 + g) * kBlock + threadIdx.x];
            rank[g] = atomicAdd(&h[mine[g] >> 8], 1u);
        }
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
#pragma unroll
    for (int g = 0; g < kGroup; ++g)
        if (mine[g] != 0xffff)
            entry[base[mine[g] >> 8] + rank[g]] = (threadIdx.x << 16) | ((unsigned int)g << 8) | (mine[g] & 0xff);
    __syncthreads();
    // The sorted entries are handed out in chunks of one wavefront, heaviest first, from a counter in LDS: a wavefront that
    // finishes a short chunk takes the next one, so the four wavefronts of the workgroup end together whatever the
    // distribution of trip counts (measured neutral on the uniform synthetic soil of bench.py, where every pool holds the
    // same mix; it bounds the imbalance when a few columns of a pool need 50x the sub-steps of the rest).
    const unsigned int nchunks = (total + 63u) / 64u, lane = threadIdx.x & 63u;
    for (;;) {
        unsigned int c = 0;
        if (lane == 0) c = atomicAdd(&next_chunk, 1u);
        c = (unsigned int)__builtin_amdgcn_readfirstlane((int)c);
        if (c >= nchunks) break;
        const long long k = (long long)total - 1 - (long long)(c * 64u + lane);
        if (k < 0) continue;
        const unsigned int e = entry[k];
        const unsigned int tile = t0 + ((e >> 8) & 0xff), trank = e >> 16;
        const int veg = (int)(tile / tiles_per_veg);
        const long long pix = (long long)(tile - (unsigned int)veg * tiles_per_veg) * kBlock + (e & 0xff);
        // a staged column resumes at the sub-step loop from the state pass 1 left in its slot; the few beyond a tile's
        // slots are recomputed from the vectors
        const size_t slot = trank < S.cap ? (size_t)tile * S.cap + trank : (size_t)-1;
        soil_column<false, FASTPOW>(A, P, veg, pix, S, slot, nullptr, tile, nullptr);
    }
}

// ---- the one-launch form (round 5) ---------------------------------------------------------------------------
// Every stream of the call is read once and every output written once, in full lines: a workgroup owns the 256
// columns of its tile from the first load to the last store.
//   phase 1  one lane per column: inputs, evaporation, infiltration, the first conductivities and the Courant
//            number (soilloop.py:123-249).  The four outputs the sub-step loop does not touch are stored here.  A
//            column that needs one sub-step takes it in its lane; the others put what the loop needs (7 values per
//            soil layer) in LDS.
//   phase 2  the tile's multi-sub-step columns, sorted by trip count (heaviest first), THREE lanes per column -- one
//            per soil layer; the only traffic between the layers of a column is the capacity of the layer below and
//            the flux from the layer above (two lane shifts per sub-step).  A wavefront carries 21 columns, so the
//            trip count of a wavefront follows the sorted list closely (one lane per column ran 64 columns to the
//            slowest), and a layer's lane needs ~60 registers instead of ~170 for the three layers side by side.
//   phase 3  every lane, back in column order: state update, diagnostics, upper zone (soilloop.py:313-354) and 18
//            coalesced stores.
// Arithmetic per column is the reference's, operation by operation (a layer's lane evaluates exactly the terms
// soil_column evaluates for that layer), so the results are those of the two-pass form bit for bit.
constexpr int kLoopCap = 128;             // multi-sub-step columns of a tile handled per round (more -> another round)
constexpr int kLoopFields = 7 * 3 + 1;    // per layer: w, wres, ws, ksat, 1/m, m, k; per column: flags (trip count: s_key)
constexpr int kColsPerWave = 21;          // 3 lanes per column, lane 63 idles

template <bool FASTPOW>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(LF_SOIL_P1_WAVES)))
k_soil_fused(lf_soil_args A, veg_plan P, unsigned short *__restrict__ tile_list, unsigned int *__restrict__ tile_count,
             int log2_classes)
{
    __shared__ unsigned int s_count, s_next;
    __shared__ double s_rec[kLoopFields * kLoopCap];
    __shared__ double s_res[3 * kLoopCap];
    __shared__ unsigned int s_key[kLoopCap];
    __shared__ unsigned short s_order[kLoopCap];
    if (threadIdx.x == 0) {
        s_count = 0;
        s_next = 0;
    }
    __syncthreads();
    const long long N = A.N;
    const double DtDay = A.DtDay;
    const long long pix = (long long)blockIdx.x * kBlock + threadIdx.x;
    const int veg = blockIdx.y;
    const unsigned int tile = blockIdx.x + blockIdx.y * gridDim.x;
    const int mode = P.mode[veg];
    bool active = pix < N && mode != 0;
    if (active && mode == 2 && !A.paddy_inactive[(long long)P.paddy_row[veg] * N + pix]) active = false;
    const long long i = (long long)veg * N + pix, j = (long long)P.landuse[veg] * N + pix;

    // what phase 3 needs of phase 1
    double w1a = 0, w1b = 0, w2 = 0, inf = 0, pref = 0, uz = 0, uzout = 0, in_gwp = 0, ws1a = 0;
    double in_sd1a = 1, in_sd1b = 1, in_sd2 = 1, wwp1a = 0, wwp1b = 0, wwp1 = 0, wwp2 = 0;
    double in_wfc1a = 1, in_wfc1b = 1, in_wfc1 = 1, in_wfc2 = 1;
    double sa = 0, sb = 0, sg = 0;
    double k1a = 0, k1b = 0, k2 = 0, nsub_f = 1; // kept for a column whose turn comes in a later round of phase 2
    int flags = 0;
    unsigned int rank = 0xffffffffu;
    if (active) {
        // every input of the column before any arithmetic: ~50 independent loads in flight per lane
        const double in_rain = A.Rain[pix], in_snow = A.SnowMelt[pix], in_leaf = A.LeafDrainage[i], in_int = A.Interception[i];
        const double in_dslr = A.DSLR[i], in_w1a = A.W1a[i], in_w1b = A.W1b[i], in_w1 = A.W1[i], in_w2 = A.W2[i];
        const double in_uz = A.UZ[i];
        const double in_esmax = A.ESMax[i], in_wres1 = A.WRes1[j], in_ws1 = A.WS1[j], in_store = A.StoreMaxPervious[j];
        const double in_bx = A.b_Xinanjiang[pix], in_pinf = A.PowerInfPot[pix], in_ppref = A.PowerPrefFlow[pix];
        const double in_uzk = A.UpperZoneK[pix];
        in_gwp = A.GwPercStep[pix];
        in_sd1a = A.SoilDepth1a[j]; in_sd1b = A.SoilDepth1b[j]; in_sd2 = A.SoilDepth2[j];
        wwp1a = A.WWP1a[j]; wwp1b = A.WWP1b[j]; wwp1 = A.WWP1[j]; wwp2 = A.WWP2[j];
        in_wfc1a = A.WFC1a[j]; in_wfc1b = A.WFC1b[j]; in_wfc1 = A.WFC1[j]; in_wfc2 = A.WFC2[j];
        const double ks1a = A.KSat1a[j], ks1b = A.KSat1b[j], ks2 = A.KSat2[j];
        const double im1a = A.GenuInvM1a[j], im1b = A.GenuInvM1b[j], im2 = A.GenuInvM2[j];
        const double m1a = A.GenuM1a[j], m1b = A.GenuM1b[j], m2 = A.GenuM2[j];
        const double wres1a = A.WRes1a[j], wres1b = A.WRes1b[j], wres2 = A.WRes2[j];
        ws1a = A.WS1a[j];
        const double ws1b = A.WS1b[j], ws2 = A.WS2[j];
        flags = (A.isFrozenSoil[pix] != 0 ? 1 : 0) | (A.PoreSpaceNotZero1a[j] != 0 ? 2 : 0) |
                (A.PoreSpaceNotZero1b[j] != 0 ? 4 : 0) | (A.PoreSpaceNotZero2[j] != 0 ? 8 : 0);
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
        double esact;
        w1a = in_w1a;
        w1b = in_w1b;
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
        pref = powxy<FASTPOW>(relsat1, in_ppref) * awi;
        awi -= pref;
        // infiltration, :201-211
        inf = dmax(dmin(awi, infpot), 0.);
        const double test1a = w1a + inf;
        w1a = dmin(ws1a, test1a);
        w1b += dmax(test1a - ws1a, 0.);
        w2 = in_w2;
        // the outputs the sub-step loop does not touch
        A.DSLR[i] = dslr;
        A.ESAct[i] = esact;
        A.PrefFlow[i] = pref;
        A.AvailableWaterForInfiltration[i] = awi;
        // upper-zone outflow before the seepage arrives, :340-341
        uz = in_uz;
        uzout = dmin(in_uzk * uz, uz);
        uz = dmax(uz - uzout, 0.);
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
        // NoSubS = max(1, ceil(Courant / CourantCrit)), :249 (capped: see soil_column)
        nsub_f = dmin(dmax(1., ceil(courant / A.CourantCrit)), kMaxSoilSubSteps);
        if (nsub_f > 1.) {
            rank = atomicAdd(&s_count, 1u); // LDS: position in the tile's list
            long long c = (long long)nsub_f;
            c = log2_classes ? 63 - __clzll((unsigned long long)c) : c;
            c = c < kClasses - 1 ? c : kClasses - 1;
            tile_list[(size_t)tile * kBlock + rank] = (unsigned short)(threadIdx.x | ((int)c << 8));
        } else { // the one sub-step of the column, :266-312 with NoSubS = 1
            sa = dmin(k1a * DtDay, ws1b - w1b);
            sb = dmin(k1b * DtDay, ws2 - w2);
            sg = dmin(k2 * DtDay, av2);
        }
        // the records of the multi-sub-step columns, kLoopCap per round (nearly always one round)
        if (rank < (unsigned int)kLoopCap) {
#define REC(f, v) s_rec[(f) * kLoopCap + rank] = (v)
            REC(0, w1a); REC(1, wres1a); REC(2, ws1a); REC(3, ks1a); REC(4, im1a); REC(5, m1a); REC(6, k1a);
            REC(7, w1b); REC(8, wres1b); REC(9, ws1b); REC(10, ks1b); REC(11, im1b); REC(12, m1b); REC(13, k1b);
            REC(14, w2); REC(15, wres2); REC(16, ws2); REC(17, ks2); REC(18, im2); REC(19, m2); REC(20, k2);
            REC(21, (double)flags);
#undef REC
            s_key[rank] = (unsigned int)nsub_f;
        }
    }
    __syncthreads();
    const unsigned int count = (unsigned int)__builtin_amdgcn_readfirstlane((int)s_count);
    if (threadIdx.x == 0) tile_count[tile] = count;
    for (unsigned int base = 0; base < count; base += kLoopCap) { // (uniform over the workgroup; nearly always one round)
        if (base > 0) {
            // a tile with more than kLoopCap multi-sub-step columns: the next kLoopCap of its list.  Their lanes still hold
            // the water contents and the first conductivities; the layer parameters are read again (cache hits).
            __syncthreads(); // the results of the round before have been taken
            if (threadIdx.x == 0) s_next = 0;
            const unsigned int r = rank - base;
            if (r < (unsigned int)kLoopCap) {
#define REC(f, v) s_rec[(f) * kLoopCap + r] = (v)
                REC(0, w1a); REC(1, A.WRes1a[j]); REC(2, ws1a); REC(3, A.KSat1a[j]); REC(4, A.GenuInvM1a[j]); REC(5, A.GenuM1a[j]); REC(6, k1a);
                REC(7, w1b); REC(8, A.WRes1b[j]); REC(9, A.WS1b[j]); REC(10, A.KSat1b[j]); REC(11, A.GenuInvM1b[j]); REC(12, A.GenuM1b[j]); REC(13, k1b);
                REC(14, w2); REC(15, A.WRes2[j]); REC(16, A.WS2[j]); REC(17, A.KSat2[j]); REC(18, A.GenuInvM2[j]); REC(19, A.GenuM2[j]); REC(20, k2);
                REC(21, (double)flags);
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
        const unsigned int col = lane / 3u, layer = lane - 3u * col;
        for (;;) {
            unsigned int t = 0;
            if (lane == 0) t = atomicAdd(&s_next, 1u);
            t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
            if (t >= ntasks) break;
            const unsigned int e = t * kColsPerWave + col;
            const bool valid = col < (unsigned int)kColsPerWave && e < n;
            const unsigned int first = s_order[t * kColsPerWave]; // the task's first column is its heaviest
            const unsigned int slot = valid ? s_order[e] : first;
            const double *__restrict__ R = s_rec + (7u * layer) * kLoopCap + slot;
            const double w = R[0], wres = R[kLoopCap], ws = R[2 * kLoopCap], ks = R[3 * kLoopCap], im = R[4 * kLoopCap],
                         m = R[5 * kLoopCap];
            double k = R[6 * kLoopCap];
            const long long nsub = valid ? (long long)s_key[slot] : 0;
            const int fl = (int)s_rec[21 * kLoopCap + slot];
            const bool pore = ((fl >> (1 + (int)layer)) & 1) != 0;
            const long long tmax = (long long)__builtin_amdgcn_readfirstlane((int)s_key[first]);
            const double dtsub = DtDay / (double)(long long)s_key[slot];
            double av = w - wres, wt = w, cap = ws - w, sum = 0.;
            for (long long s = 0; s < tmax; ++s) {
                if (s > 0) k = unsat_k<FASTPOW>(wt, pore, wres, ws, ks, im, m);
                const double cap_below = __shfl_down(cap, 1, 64);     // layer + 1 of the same column
                const double limit = (layer == 2u) ? av : cap_below;  // :280-285
                const double flux = dmin(k * dtsub, limit);
                const double flux_above = __shfl_up(flux, 1, 64);     // layer - 1 of the same column
                const double net = (layer == 0u) ? -flux : flux_above - flux;
                const bool live = s < nsub;
                av = live ? av + net : av;                            // :286-288 (av1a -= fa is av1a + (-fa) exactly)
                wt = av + wres;
                cap = ws - wt;
                sum = live ? sum + flux : sum;
            }
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
    const bool frozen = (flags & 1) != 0, pore1a = (flags & 2) != 0, pore1b = (flags & 4) != 0, pore2 = (flags & 8) != 0;
    if (frozen) sa = sb = sg = 0.; // :313-316
    // state update, :319-325 (W1 is taken BEFORE the 1a overflow correction, as the reference does)
    w1a -= sa;
    w1b = w1b + sa - sb;
    w2 = w2 + sb - sg;
    const double w1 = w1a + w1b;
    inf -= dmax(w1a - ws1a, 0.);
    w1a = dmin(w1a, ws1a);
    // upper zone, :342-354
    if (P.drained[veg]) {
        uzout += A.DrainedFraction * sg;
        uz += (1 - A.DrainedFraction) * sg + pref;
    } else
        uz += sg + pref;
    const double perc = dmin(in_gwp, uz);
    uz = dmax(uz - perc, 0.);
    A.SeepTopToSubA[i] = sa;
    A.SeepTopToSubB[i] = sb;
    A.SeepSubToGW[i] = sg;
    A.Infiltration[i] = inf;
    A.W1a[i] = w1a;
    A.W1b[i] = w1b;
    A.W1[i] = w1;
    A.W2[i] = w2;
    // diagnostics, :330-336
    A.Theta1a[i] = pore1a ? w1a / in_sd1a : 0.;
    A.Theta1b[i] = pore1b ? w1b / in_sd1b : 0.;
    A.Theta2[i] = pore2 ? w2 / in_sd2 : 0.;
    A.Sat1a[i] = (w1a - wwp1a) / (in_wfc1a - wwp1a);
    A.Sat1b[i] = (w1b - wwp1b) / (in_wfc1b - wwp1b);
    A.Sat1[i] = (w1 - wwp1) / (in_wfc1 - wwp1);
    A.Sat2[i] = (w2 - wwp2) / (in_wfc2 - wwp2);
    A.UZOutflow[i] = uzout;
    A.GwPercUZLZ[i] = perc;
    A.UZ[i] = uz;
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

// number of columns the last lf_soil_columns_device call on `device` deferred to the second pass
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

// sub-step histogram of the last call's deferred columns: hist[k] = columns whose trip count was k (the last bin holds
// k >= nbins - 1; the lists keep the count clamped to kClasses - 1)
int lf_soil_substep_histogram(int device, int64_t *hist, int nbins)
{
    if (!hist || nbins < 2) return lf_set_error(LF_E_INVALID, "null argument");
    lf_device_ctx *c;
    LF_TRY(lf_ctx(device, &c));
    for (int k = 0; k < nbins; ++k) hist[k] = 0;
    if (!c->soil_ws || c->soil_ntiles == 0) return LF_OK;
    const size_t nt = c->soil_ntiles;
    std::vector<unsigned int> cnt(nt);
    std::vector<unsigned short> lst(nt * kBlock);
    LF_HIP(hipMemcpyAsync(cnt.data(), c->soil_ws, sizeof(unsigned int) * nt, hipMemcpyDeviceToHost, c->stream));
    LF_HIP(hipMemcpyAsync(lst.data(), (const unsigned int *)c->soil_ws + nt + 4, sizeof(unsigned short) * nt * kBlock,
                          hipMemcpyDeviceToHost, c->stream));
    LF_HIP(hipStreamSynchronize(c->stream));
    for (size_t t = 0; t < nt; ++t)
        for (unsigned int k = 0; k < cnt[t] && k < (unsigned int)kBlock; ++k) {
            const int cls = lst[t * kBlock + k] >> 8;
            hist[cls < nbins - 1 ? cls : nbins - 1] += 1;
        }
    return LF_OK;
}

int lf_soil_columns_device(int device, const lf_soil_args *a)
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
    // per-tile lists of the columns that need more than one Courant sub-step (grow-only per-device workspace):
    // tile counts | per-tile lane lists
    const unsigned int tiles_per_veg = (unsigned int)blocks_for(a->N);
    const size_t ntiles = (size_t)tiles_per_veg * (size_t)a->V;
    // LF_SOIL_TWO_PASS=1: the two-launch form of rounds 1-4 (pass 1 + deferred columns from a staging area), kept for A/B
    const char *tp = std::getenv("LF_SOIL_TWO_PASS");
    const bool two_pass = tp && tp[0] == '1';
    // staging area of the deferred columns' inputs: 96 slots per tile of 256 columns (LF_SOIL_STAGE_SLOTS=0: off), used
    // when the previous call deferred at least 4 % of its columns (the count comes back asynchronously)
    unsigned int cap = two_pass ? kStageCap : 0;
    if (const char *e = std::getenv("LF_SOIL_STAGE_SLOTS")) cap = (two_pass && std::atol(e) > 0) ? kStageCap : 0; // 0 = never stage
    const char *force = std::getenv("LF_SOIL_STAGE_ALWAYS"); // A/B switch
    if (!c->soil_deferred_host) {
        LF_HIP(hipHostMalloc((void **)&c->soil_deferred_host, sizeof(unsigned long long), hipHostMallocDefault));
        *c->soil_deferred_host = 0;
        LF_HIP(hipMalloc((void **)&c->soil_deferred_dev, sizeof(unsigned long long)));
        LF_HIP(hipEventCreateWithFlags(&c->soil_deferred_ready, hipEventDisableTiming));
    }
    bool stage = false;
    if (cap > 0) {
        if (force && force[0] == '1')
            stage = true;
        else if (c->soil_deferred_pending && hipEventQuery(c->soil_deferred_ready) == hipSuccess)
            stage = (double)*c->soil_deferred_host >= 0.04 * (double)c->soil_deferred_columns;
        else
            stage = c->soil_stage_last;
    }
    c->soil_stage_last = stage;
    if (!stage) cap = 0;
    const size_t lists = (sizeof(unsigned int) * (ntiles + 4) + sizeof(unsigned short) * ntiles * kBlock + 255) & ~(size_t)255;
    const size_t need = lists + sizeof(double) * (size_t)kStageFields * ntiles * cap;
    if (c->soil_ws_bytes < need) {
        if (c->soil_ws) LF_HIP(hipFree(c->soil_ws));
        c->soil_ws = nullptr;
        c->soil_ws_bytes = 0;
        LF_HIP(hipMalloc(&c->soil_ws, need));
        c->soil_ws_bytes = need;
    }
    unsigned int *tile_count = (unsigned int *)c->soil_ws;
    unsigned short *tile_list = (unsigned short *)(tile_count + ntiles + 4);
    c->soil_ntiles = ntiles;
    soil_stage S;
    S.buf = (double *)((char *)c->soil_ws + lists);
    S.cap = cap;
    S.nslots = ntiles * cap;
    // LF_GENERAL_POW=1: OCML pow instead of lf_pow_pos (A/B parity and timing)
    const char *force_general = std::getenv("LF_GENERAL_POW");
    const bool fastpow = !(force_general && force_general[0] == '1');
    const char *l2 = std::getenv("LF_SOIL_LOG2_CLASSES");
    const int log2_classes = (l2 && l2[0] == '1') ? 1 : 0;
    const dim3 grid1(tiles_per_veg, (unsigned)a->V), block(kBlock);
    const dim3 grid2((unsigned)((ntiles + kGroup - 1) / kGroup));
    if (!two_pass) { // one launch: every stream read once, every output written once (k_soil_fused)
        if (fastpow)
            hipLaunchKernelGGL(k_soil_fused<true>, grid1, block, 0, c->stream, *a, P, tile_list, tile_count, log2_classes);
        else
            hipLaunchKernelGGL(k_soil_fused<false>, grid1, block, 0, c->stream, *a, P, tile_list, tile_count, log2_classes);
        c->soil_deferred_pending = false;
        LF_HIP(hipGetLastError());
        return LF_OK;
    }
    LF_HIP(hipMemsetAsync(c->soil_deferred_dev, 0, sizeof(unsigned long long), c->stream));
    if (fastpow && stage)
        hipLaunchKernelGGL((k_soil_columns<true, true>), grid1, block, 0, c->stream, *a, P, tile_list, tile_count, S, log2_classes);
    else if (fastpow)
        hipLaunchKernelGGL((k_soil_columns<true, false>), grid1, block, 0, c->stream, *a, P, tile_list, tile_count, S, log2_classes);
    else if (stage)
        hipLaunchKernelGGL((k_soil_columns<false, true>), grid1, block, 0, c->stream, *a, P, tile_list, tile_count, S, log2_classes);
    else
        hipLaunchKernelGGL((k_soil_columns<false, false>), grid1, block, 0, c->stream, *a, P, tile_list, tile_count, S, log2_classes);
    if (fastpow)
        hipLaunchKernelGGL(k_soil_columns_deferred<true>, grid2, block, 0, c->stream, *a, P, tile_list, tile_count,
                           (unsigned int)ntiles, tiles_per_veg, S, c->soil_deferred_dev);
    else
        hipLaunchKernelGGL(k_soil_columns_deferred<false>, grid2, block, 0, c->stream, *a, P, tile_list, tile_count,
                           (unsigned int)ntiles, tiles_per_veg, S, c->soil_deferred_dev);
    LF_HIP(hipMemcpyAsync(c->soil_deferred_host, c->soil_deferred_dev, sizeof(unsigned long long), hipMemcpyDeviceToHost,
                          c->stream));
    LF_HIP(hipEventRecord(c->soil_deferred_ready, c->stream));
    c->soil_deferred_pending = true;
    c->soil_deferred_columns = (unsigned long long)a->V * (unsigned long long)a->N;
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
