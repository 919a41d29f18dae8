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
    double s = 0.;
    if (pore) s = dmax(dmin((w - wres) / (ws - wres), 1.), 0.);
    const double t = 1. - powxy<FASTPOW>(1. - powxy<FASTPOW>(s, inv_m), m);
    return ksat * sqrt(s) * (t * t);
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
// Inputs of deferred columns are handed from pass 1 to pass 2 through a staging area: pass 1 has all of them in
// registers when it finds that a column needs several sub-steps, and writes them next to the inputs of the tile's
// other deferred columns (slot = tile * cap + rank in the tile's list, 46 values per slot); pass 2 then reads six
// lines per column instead of one 64-byte sector per 8-byte value scattered over ~46 vectors (measured: 6 GB fetched
// for 0.8 GB of inputs).  A tile has room for `cap` columns; the ones beyond gather from the vectors as before.
constexpr int kStageFields = 46;
constexpr unsigned int kStageCap = 96; // slots per tile (of 256 columns); pass 1 collects them in LDS: 46 x 96 x 8 B = 35 KB
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
    int f_ = 0;
#define LD(expr) (staged ? S.buf[slot * kStageFields + (size_t)(f_++)] : (f_++, (double)(expr)))
    // Every input of the column is fetched here, before any arithmetic: ~50 independent loads in flight per lane
    // instead of the handful the compiler keeps when loads sit next to their first use (the kernel is a stream
    // of ~90 vectors; memory-level parallelism, not ALU, sets its speed).
    const double in_rain = LD(A.Rain[pix]), in_snow = LD(A.SnowMelt[pix]), in_leaf = LD(A.LeafDrainage[i]),
                 in_int = LD(A.Interception[i]);
    const double in_dslr = LD(A.DSLR[i]), in_w1a = LD(A.W1a[i]), in_w1b = LD(A.W1b[i]), in_w1 = LD(A.W1[i]),
                 in_w2 = LD(A.W2[i]), in_uz = LD(A.UZ[i]);
    const double in_esmax = LD(A.ESMax[i]), in_wres1 = LD(A.WRes1[j]), in_ws1 = LD(A.WS1[j]),
                 in_store = LD(A.StoreMaxPervious[j]);
    const double in_bx = LD(A.b_Xinanjiang[pix]), in_pinf = LD(A.PowerInfPot[pix]), in_ppref = LD(A.PowerPrefFlow[pix]);
    const double in_uzk = LD(A.UpperZoneK[pix]), in_gwp = LD(A.GwPercStep[pix]);
    const double in_sd1a = LD(A.SoilDepth1a[j]), in_sd1b = LD(A.SoilDepth1b[j]), in_sd2 = LD(A.SoilDepth2[j]);
    const double wwp1a = LD(A.WWP1a[j]), wwp1b = LD(A.WWP1b[j]), wwp1 = LD(A.WWP1[j]), wwp2 = LD(A.WWP2[j]);
    const double in_wfc1a = LD(A.WFC1a[j]), in_wfc1b = LD(A.WFC1b[j]), in_wfc1 = LD(A.WFC1[j]), in_wfc2 = LD(A.WFC2[j]);
    const double ks1a = LD(A.KSat1a[j]), ks1b = LD(A.KSat1b[j]), ks2 = LD(A.KSat2[j]);
    const double im1a = LD(A.GenuInvM1a[j]), im1b = LD(A.GenuInvM1b[j]), im2 = LD(A.GenuInvM2[j]);
    const double m1a = LD(A.GenuM1a[j]), m1b = LD(A.GenuM1b[j]), m2 = LD(A.GenuM2[j]);
    const double wres1a = LD(A.WRes1a[j]), wres1b = LD(A.WRes1b[j]), wres2 = LD(A.WRes2[j]);
    const double ws1a = LD(A.WS1a[j]), ws1b = LD(A.WS1b[j]), ws2 = LD(A.WS2[j]);
    // the four flags travel as one small integer
    const double flags_d = LD((A.isFrozenSoil[pix] != 0 ? 1 : 0) | (A.PoreSpaceNotZero1a[j] != 0 ? 2 : 0) |
                              (A.PoreSpaceNotZero1b[j] != 0 ? 4 : 0) | (A.PoreSpaceNotZero2[j] != 0 ? 8 : 0));
#undef LD
    const int flags = (int)flags_d;
    const bool frozen = (flags & 1) != 0, pore1a = (flags & 2) != 0, pore1b = (flags & 4) != 0, pore2 = (flags & 8) != 0;
    static_assert(kStageFields == 46, "one staging field per LD() above");
    // available water for infiltration, :100,131
    double awi = dmax((in_rain + in_snow) + in_leaf - in_int, 0.);
    // days since last rain, :137-140
    double dslr = in_dslr;
    if (awi > A.AvWaterThreshold)
        dslr = 1;
    else
        dslr += DtDay;
    // bare soil evaporation, :148-163
    double w1a = in_w1a, w1b = in_w1b, esact;
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
    double w1 = w1a + w1b;
    // Xinanjiang infiltration capacity, :168-179
    const double relsat1 = pore1a ? dmin(w1 / in_ws1, 1.0) : 0.0;
    const double satfrac = 1.0 - powxy<FASTPOW>(1.0 - relsat1, in_bx);
    const double infpot = frozen ? 0.0 : in_store * powxy<FASTPOW>(1. - satfrac, in_pinf) * DtDay;
    // preferential flow, :190-194
    const double pref = powxy<FASTPOW>(relsat1, in_ppref) * awi;
    awi -= pref;
    // infiltration, :201-211
    double inf = dmax(dmin(awi, infpot), 0.);
    const double test1a = w1a + inf;
    w1a = dmin(ws1a, test1a);
    w1b += dmax(test1a - ws1a, 0.);
    double w2 = in_w2;
    // Van Genuchten conductivities and Courant numbers, :223-249
    double k1a = unsat_k<FASTPOW>(w1a, pore1a, wres1a, ws1a, ks1a, im1a, m1a);
    double k1b = unsat_k<FASTPOW>(w1b, pore1b, wres1b, ws1b, ks1b, im1b, m1b);
    double k2 = unsat_k<FASTPOW>(w2, pore2, wres2, ws2, ks2, im2, m2);
    double av1a = w1a - wres1a, av1b = w1b - wres1b, av2 = w2 - wres2;
    double cap1 = ws1b - w1b, cap2 = ws2 - w2;
    const double ca = (av1a == 0) ? 0. : k1a * DtDay / av1a;
    const double cb = (av1b == 0) ? 0. : k1b * DtDay / av1b;
    const double cg = (av2 == 0) ? 0. : k2 * DtDay / av2;
    const double courant = dmax(dmax(ca, cb), cg);
    // NoSubS = max(1, ceil(Courant / CourantCrit)), :249.  A non-finite or absurd Courant number (zero available water
    // next to a huge conductivity) would make the reference's int conversion overflow and this loop spin for ever:
    // the trip count is capped (documented deviation; CourantCrit > 0 is checked on the host).
    const double nsub_f = dmin(dmax(1., ceil(courant / A.CourantCrit)), kMaxSoilSubSteps);
    const long long nsub = (long long)nsub_f;
    if (DEFER && nsub > 1) {
        const unsigned int rank = atomicAdd(lds_count, 1u); // LDS: position in the tile's list
        *rank_out = rank;
        if (STAGE && rank < kStageCap) { // into the block's LDS table; the block writes it out in full lines afterwards
            int f = 0;
#define ST(v) lds_stage[(f++) * kStageCap + rank] = (v)
            // the inputs whose registers are dead by now are read again (cache hits, deferred lanes only) rather than
            // kept alive across the infiltration arithmetic: pass 1 has no registers to spare
            ST(A.Rain[pix]); ST(A.SnowMelt[pix]); ST(A.LeafDrainage[i]); ST(A.Interception[i]);
            ST(A.DSLR[i]); ST(A.W1a[i]); ST(A.W1b[i]); ST(A.W1[i]); ST(A.W2[i]); ST(in_uz);
            ST(A.ESMax[i]); ST(A.WRes1[j]); ST(A.WS1[j]); ST(A.StoreMaxPervious[j]);
            ST(A.b_Xinanjiang[pix]); ST(A.PowerInfPot[pix]); ST(A.PowerPrefFlow[pix]);
            ST(in_uzk); ST(in_gwp);
            ST(in_sd1a); ST(in_sd1b); ST(in_sd2);
            ST(wwp1a); ST(wwp1b); ST(wwp1); ST(wwp2);
            ST(in_wfc1a); ST(in_wfc1b); ST(in_wfc1); ST(in_wfc2);
            ST(ks1a); ST(ks1b); ST(ks2);
            ST(im1a); ST(im1b); ST(im2);
            ST(m1a); ST(m1b); ST(m2);
            ST(wres1a); ST(wres1b); ST(wres2);
            ST(ws1a); ST(ws1b); ST(ws2);
            ST(flags_d);
#undef ST
        }
        return nsub;
    }
    // sub-step loop, :266-312
    double wt1a = w1a, wt1b = w1b, wt2 = w2;
    double sa = 0., sb = 0., sg = 0.;
    const double dtsub = DtDay / (double)nsub;
    const long long trips = DEFER ? 1 : nsub; // DEFER: nsub == 1 here, the re-evaluation branch disappears
    for (long long s = 0; s < trips; ++s) {
        if (s > 0) {
            k1a = unsat_k<FASTPOW>(wt1a, pore1a, wres1a, ws1a, ks1a, im1a, m1a);
            k1b = unsat_k<FASTPOW>(wt1b, pore1b, wres1b, ws1b, ks1b, im1b, m1b);
            k2 = unsat_k<FASTPOW>(wt2, pore2, wres2, ws2, ks2, im2, m2);
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
    w1 = w1a + w1b;
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
constexpr int kClasses = 8; // class = floor(log2(nsub)) clamped to kClasses-1: trip counts inside a class differ < 2x
constexpr int kGroup = 16;  // tiles per pass-2 workgroup (a pool of 4096 columns: enough deferred columns to fill
                            // wavefronts with similar trip counts even when the sub-step distribution has a long tail)

#ifndef LF_SOIL_P1_WAVES
#define LF_SOIL_P1_WAVES 4
#endif
// waves_per_eu(4): pass 1 streams ~500 B per column and needs the occupancy; the allocator otherwise wobbles
// between 126 and 133 VGPRs (4 vs 3 waves per SIMD) with unrelated edits to this file; 5 waves spill and are slower
template <bool FASTPOW, bool STAGE>
__global__ void __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(LF_SOIL_P1_WAVES))) k_soil_columns(lf_soil_args A, veg_plan P, unsigned short *__restrict__ tile_list,
                                                         unsigned int *__restrict__ tile_count, soil_stage S)
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
            int c = 63 - __clzll((unsigned long long)nsub); // floor(log2(nsub)) >= 1
            c = c < kClasses - 1 ? c : kClasses - 1;
            tile_list[(size_t)tile * kBlock + rank] = (unsigned short)(threadIdx.x | (c << 8));
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) tile_count[tile] = count;
    if (STAGE) { // the tile's staged inputs as ONE contiguous run: slot-major, the 46 fields of a column side by side
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
    __shared__ unsigned int cnt[kGroup], h[kClasses], base[kClasses], total;
    __shared__ unsigned int entry[kGroup * kBlock]; // (rank in tile list << 16 | tile-in-group << 8 | lane), by class
    const unsigned int t0 = blockIdx.x * kGroup;
    if (threadIdx.x < kGroup) cnt[threadIdx.x] = (t0 + threadIdx.x < ntiles) ? tile_count[t0 + threadIdx.x] : 0;
    if (threadIdx.x < kClasses) h[threadIdx.x] = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int t = 0;
        for (int g = 0; g < kGroup; ++g) t += cnt[g];
        total = t;
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
    if (threadIdx.x == 0) {
        unsigned int acc = 0;
        for (int c = 0; c < kClasses; ++c) {
            base[c] = acc;
            acc += h[c];
        }
    }
    __syncthreads();
#pragma unroll
    for (int g = 0; g < kGroup; ++g)
        if (mine[g] != 0xffff)
            entry[base[mine[g] >> 8] + rank[g]] = (threadIdx.x << 16) | ((unsigned int)g << 8) | (mine[g] & 0xff);
    __syncthreads();
    for (unsigned int k = threadIdx.x; k < total; k += kBlock) {
        const unsigned int e = entry[k];
        const unsigned int tile = t0 + ((e >> 8) & 0xff), trank = e >> 16;
        const int veg = (int)(tile / tiles_per_veg);
        const long long pix = (long long)(tile - (unsigned int)veg * tiles_per_veg) * kBlock + (e & 0xff);
        const size_t slot = trank < S.cap ? (size_t)tile * S.cap + trank : (size_t)-1;
        soil_column<false, FASTPOW>(A, P, veg, pix, S, slot, nullptr, tile, nullptr);
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
    // staging area of the deferred columns' inputs: 96 slots per tile of 256 columns (LF_SOIL_STAGE_SLOTS=0: off), used
    // when the previous call deferred at least 4 % of its columns (the count comes back asynchronously)
    unsigned int cap = kStageCap;
    if (const char *e = std::getenv("LF_SOIL_STAGE_SLOTS")) cap = std::atol(e) > 0 ? kStageCap : 0; // 0 = never stage
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
    const dim3 grid1(tiles_per_veg, (unsigned)a->V), block(kBlock);
    const dim3 grid2((unsigned)((ntiles + kGroup - 1) / kGroup));
    LF_HIP(hipMemsetAsync(c->soil_deferred_dev, 0, sizeof(unsigned long long), c->stream));
    if (fastpow && stage)
        hipLaunchKernelGGL((k_soil_columns<true, true>), grid1, block, 0, c->stream, *a, P, tile_list, tile_count, S);
    else if (fastpow)
        hipLaunchKernelGGL((k_soil_columns<true, false>), grid1, block, 0, c->stream, *a, P, tile_list, tile_count, S);
    else if (stage)
        hipLaunchKernelGGL((k_soil_columns<false, true>), grid1, block, 0, c->stream, *a, P, tile_list, tile_count, S);
    else
        hipLaunchKernelGGL((k_soil_columns<false, false>), grid1, block, 0, c->stream, *a, P, tile_list, tile_count, S);
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
