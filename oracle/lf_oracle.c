/*
 * lf_oracle.c -- CPU restatement of the LISFLOOD routing / soil hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity oracle and the "port" CPU baseline of bench.py.  It is NOT part of the
 * product: nothing under lisflood-code_amd/ links, loads or calls it.  Only tests/, the smoke() of
 * __graft_entry__.py and bench.py's cpu_baseline leg may use it -- as the checker, never as the path
 * that is measured or shipped.
 *
 * Pinning: every function below is checked bit-for-bit (or to <= 4 ulp where glibc pow and the
 * reference's un-jitted CPython pow could differ -- in this image they are the same libm, so the
 * observed difference is 0) against golden vectors captured from the reference's own Python code,
 * see tests/golden/make_golden.py and tests/test_oracle_golden.py.
 *
 * Build: gcc -O2 -fno-fast-math -ffp-contract=off -fopenmp -shared -fPIC (oracle/Makefile).
 * -ffp-contract=off keeps a*b+c as two roundings, as numpy / numexpr / numba(fastmath=False) do.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference/src/lisflood/hydrological_modules/).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define NEWTON_TOL 1e-12 /* kinematic_wave_parallel_tools.py:26 */
#define MAX_ITERS 3000   /* kinematic_wave_parallel_tools.py:27 */

/*
 * x**2: numba lowers a float raised to the integer literal 2 to x*x (the real reference); the un-jitted
 * CPython import used to capture the golden vectors calls libm pow(x, 2.0), which differs from x*x by
 * 1 ulp in rare near-halfway cases.  Default = numba semantics; lfo_set_cpython_pow2(1) switches to the
 * CPython form so that the golden vectors can be matched bit-for-bit (tests/test_oracle_golden.py).
 */
static int g_cpython_pow2 = 0;
void lfo_set_cpython_pow2(int on) { g_cpython_pow2 = on; }
static volatile double g_two = 2.0; /* volatile: keeps gcc from folding pow(x, 2.0) into x*x */
static inline double sq(double x) { return g_cpython_pow2 ? pow(x, g_two) : x * x; }

static inline double dmin(double a, double b) { return (b < a) ? b : a; } /* Python builtins.min(a,b) */
static inline double dmax(double a, double b) { return (b > a) ? b : a; } /* Python builtins.max(a,b) */

/* ------------------------------------------------------------------------------------------------
 * a3-a5: LDD -> lookups -> routing orders
 * ---------------------------------------------------------------------------------------------- */

/* IX_ADDS / FLOW_CODE, kinematic_wave_parallel.py:49-51 */
static const int IX_ADDS[8][2] = {{1, 0}, {1, 1}, {0, 1}, {-1, 1}, {-1, 0}, {-1, -1}, {0, -1}, {1, -1}};
static const int FLOW_CODE[9] = {2, 3, 6, 9, 8, 7, 4, 1, 5};

/*
 * rebuildFlowMatrix + decodeFlowMatrix + streamLookups + upDownLookups
 * (kinematic_wave_parallel.py:59-90, kinematic_wave_parallel_tools.py:111-130).
 * codes[N]: compressed LDD codes (doubles, as the reference passes them); mask[H*W]: land mask.
 * Out: downstream[N] (-1 = none), upstream[N*8] (-1 filled, ascending source id), num_ups[N].
 * Returns K = max(1, number of non-empty upstream columns).
 * Documented deviation: a code outside {0,1..9} is UB in the reference (np.empty);
 * here it decodes to 8 = no flow.
 */
int lfo_lookups(const double *codes, const uint8_t *mask, int H, int W, int64_t *downstream, int64_t *upstream,
                int64_t *num_ups)
{
    int64_t HW = (int64_t)H * W, n = 0;
    int64_t *land_points = (int64_t *)malloc(sizeof(int64_t) * HW);
    int8_t *dir = (int8_t *)malloc(HW);
    for (int64_t i = 0; i < HW; ++i) {
        land_points[i] = -1;
        dir[i] = 8; /* flow_dir[~land_mask] = 8, kinematic_wave_parallel.py:84 */
        if (mask[i]) {
            double c = codes[n];
            int d = 8; /* SEA_CODE 0 -> FLOW_CODE[8] = pit, :67 */
            for (int k = 0; k < 9; ++k)
                if (c == (double)FLOW_CODE[k]) d = k;
            dir[i] = (int8_t)d;
            land_points[i] = n++;
        }
    }
    for (int64_t p = 0; p < n; ++p) {
        downstream[p] = -1;
        num_ups[p] = 0;
        for (int k = 0; k < 8; ++k) upstream[p * 8 + k] = -1;
    }
    for (int r = 0; r < H; ++r)
        for (int c = 0; c < W; ++c) { /* row-major scan of sources, kwpt.py:119-129 */
            int d = dir[(int64_t)r * W + c];
            if (d < 8) {
                int rr = r + IX_ADDS[d][0], cc = c + IX_ADDS[d][1];
                if (rr != -1 && cc != -1 && rr != H && cc != W && mask[(int64_t)rr * W + cc]) {
                    int64_t up = land_points[(int64_t)r * W + c], dn = land_points[(int64_t)rr * W + cc];
                    downstream[up] = dn;
                    upstream[dn * 8 + num_ups[dn]] = up;
                    num_ups[dn] += 1;
                }
            }
        }
    int K = 0;
    for (int k = 0; k < 8; ++k) {
        int any = 0;
        for (int64_t p = 0; p < n && !any; ++p) any = upstream[p * 8 + k] != -1;
        K += any;
    }
    free(land_points);
    free(dir);
    return K < 1 ? 1 : K;
}

/*
 * topoDistFromSea + _setRoutingOrders (kinematic_wave_parallel.py:92-106, 140-158).
 * order = max(dist) - dist; pixels sorted by (order, pixel id).
 * Out: pixels_ordered[N], start_stop[2*NL] (caller allocates 2*N).  Returns NL, or -1 on a cyclic LDD
 * (the reference loops forever there, :99).
 */
int64_t lfo_orders(const int64_t *downstream, const int64_t *upstream, const int64_t *num_ups, int64_t n,
                   int64_t *pixels_ordered, int64_t *start_stop)
{
    int64_t *dist = (int64_t *)malloc(sizeof(int64_t) * n);
    int64_t *queue = (int64_t *)malloc(sizeof(int64_t) * n);
    int64_t head = 0, tail = 0, maxd = 0;
    for (int64_t p = 0; p < n; ++p) {
        dist[p] = -1;
        if (downstream[p] == -1) {
            dist[p] = 1;
            queue[tail++] = p;
        }
    }
    while (head < tail) {
        int64_t p = queue[head++];
        if (dist[p] > maxd) maxd = dist[p];
        for (int64_t k = 0; k < num_ups[p]; ++k) {
            int64_t u = upstream[p * 8 + k];
            dist[u] = dist[p] + 1;
            queue[tail++] = u;
        }
    }
    if (tail != n) {
        free(dist);
        free(queue);
        return -1;
    }
    /* counting sort by order, ascending pixel id inside an order (pandas sort_values, :152) */
    int64_t NL = maxd;
    int64_t *count = (int64_t *)calloc((size_t)NL + 1, sizeof(int64_t));
    for (int64_t p = 0; p < n; ++p) count[maxd - dist[p] + 1]++;
    for (int64_t k = 0; k < NL; ++k) count[k + 1] += count[k];
    for (int64_t k = 0; k < NL; ++k) {
        start_stop[2 * k] = count[k];
        start_stop[2 * k + 1] = count[k + 1];
    }
    for (int64_t p = 0; p < n; ++p) pixels_ordered[count[maxd - dist[p]]++] = p;
    free(count);
    free(dist);
    free(queue);
    return NL;
}

/* ------------------------------------------------------------------------------------------------
 * a7-a9: one kinematicWaveRouting call
 * ---------------------------------------------------------------------------------------------- */

/* closureError, kinematic_wave_parallel_tools.py:89-92 */
static inline double closure_error(double q, double upper, double a, double beta)
{
    return q + a * pow(q, beta) - upper;
}

/* solve1Pixel, kinematic_wave_parallel_tools.py:48-87.  Returns the Newton iteration count. */
static inline int solve1pixel(int64_t pix, double *discharge, const double *constant, const int64_t *upstream, int K,
                              const int64_t *num_ups, const double *a, const double *ba, double beta, double inv_beta,
                              double b_minus_1)
{
    int count = 0;
    double previous = -1.0, ups = 0.0;
    for (int64_t k = 0; k < num_ups[pix]; ++k) ups += discharge[upstream[pix * K + k]];
    double c = ups + constant[pix];
    if (c <= NEWTON_TOL) {
        discharge[pix] = 0;
        return 0;
    }
    double t = ba[pix] * pow(c, b_minus_1), secant;
    if (t <= 1)
        secant = c / (1 + t);
    else
        secant = c / (1 + pow(t, inv_beta));
    double other = pow((c - secant) / a[pix], inv_beta);
    double q = (secant + other) / 2;
    double err = closure_error(q, c, a[pix], beta);
    while (fabs(err) > NEWTON_TOL && q != previous && count < MAX_ITERS) {
        previous = q;
        q -= err / (1 + ba[pix] * pow(q, b_minus_1));
        q = dmax(q, NEWTON_TOL);
        err = closure_error(q, c, a[pix], beta);
        count += 1;
    }
    if (q == NEWTON_TOL) q = 0;
    discharge[pix] = q;
    return count;
}

/*
 * kinematicWave.kinematicWaveRouting + kinematicRouting
 * (kinematic_wave_parallel.py:160-179, kinematic_wave_parallel_tools.py:34-46).
 * a = alpha*dx/dt, ba = beta*a (precomputed by the caller exactly as kinematicWave.__init__ :127-132).
 * dx: per-pixel array, or NULL with dx_scalar.  discharge is updated in place.
 * scratch: caller-provided double[N] for `constant`.  iters_out (optional): total / max Newton iterations.
 */
void lfo_route(double *discharge, const double *q_lat, const double *dx, double dx_scalar, const double *a,
               const double *ba, double beta, const int64_t *upstream, int K, const int64_t *num_ups,
               const int64_t *pixels_ordered, const int64_t *start_stop, int64_t NL, int64_t n, double *scratch,
               int64_t *iters_out)
{
    double inv_beta = 1 / beta, b_minus_1 = beta - 1;
    double *constant = scratch;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        double lateral = q_lat[p] * (dx ? dx[p] : dx_scalar); /* "q * dx", :163 */
        constant[p] = a[p] * pow(discharge[p], beta) + lateral; /* :175 */
    }
    int64_t tot = 0, mx = 0;
    for (int64_t order = 0; order < NL; ++order) {
        int64_t first = start_stop[2 * order], last = start_stop[2 * order + 1];
#pragma omp parallel for schedule(static) reduction(+ : tot) reduction(max : mx) if (last - first > 256)
        for (int64_t i = first; i < last; ++i) {
            int it = solve1pixel(pixels_ordered[i], discharge, constant, upstream, K, num_ups, a, ba, beta, inv_beta,
                                 b_minus_1);
            tot += it;
            if (it > mx) mx = it;
        }
    }
    if (iters_out) {
        iters_out[0] = tot;
        iters_out[1] = mx;
    }
}

/* ------------------------------------------------------------------------------------------------
 * a12: element-wise arithmetic of one routing sub-step around the router calls (routing.py:512-603, 693-703)
 * ---------------------------------------------------------------------------------------------- */

/* routing.py:512 (+524 NaN fix in the single branch) */
void lfo_sideflow(const double *SideflowChanM3, const uint8_t *IsChannelKinematic, const double *InvChanLength,
                  double InvDtRouting, int nan_to_zero, int64_t n, double *SideflowChan)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        double s = IsChannelKinematic[p] ? SideflowChanM3[p] * InvChanLength[p] * InvDtRouting : 0.0;
        if (nan_to_zero && isnan(s)) s = 0;
        SideflowChan[p] = s;
    }
}

/* routing.py:549-567: split of the sideflow between main channel and floodplain */
void lfo_split_sideflow(const double *SideflowChan, const double *ChanM3Kin, const double *Chan2M3Kin,
                        const double *Chan2M3Start, const double *M3Limit, const double *Chan2QStart,
                        const double *InvChanLength, int64_t n, double *Sideflow1Chan, double *Sideflow2Chan)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        double tot = ChanM3Kin[p] + Chan2M3Kin[p];
        double ratio = (tot > 0) ? ChanM3Kin[p] / tot : 0.0;                                       /* :549 */
        double s1 = ((tot - Chan2M3Start[p]) > M3Limit[p]) ? ratio * SideflowChan[p] : SideflowChan[p]; /* :557 */
        if (fabs(SideflowChan[p]) < 1e-7) s1 = SideflowChan[p];                                    /* :563 */
        Sideflow1Chan[p] = s1;
        Sideflow2Chan[p] = (SideflowChan[p] - s1) + Chan2QStart[p] * InvChanLength[p];             /* :565-567 */
    }
}

/* routing.py:527-532 / 574-578: Q -> volume (clamped) -> Q round trip of the main channel */
void lfo_main_fixup(double *ChanQKin, double *ChanM3Kin, const double *ChanLength, const double *ChannelAlpha,
                    const double *InvChanLength, const double *InvChannelAlpha, double Beta, double InvBeta,
                    int64_t n)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        double v = ChanLength[p] * ChannelAlpha[p] * pow(ChanQKin[p], Beta);
        if (v < 0.0) v = 0.0; /* np.maximum(v, 0.0); NaN propagates */
        ChanM3Kin[p] = v;
        ChanQKin[p] = pow(v * InvChanLength[p] * InvChannelAlpha[p], InvBeta);
    }
}

/* routing.py:584-597: floodplain volume floor, CrossSection2Area, Chan2QKin, superposed ChanQ */
void lfo_floodplain_fixup(double *Chan2QKin, double *Chan2M3Kin, double *CrossSection2Area, double *ChanQ,
                          const double *ChanQKin, const double *ChanLength, const double *ChannelAlpha2,
                          const double *InvChanLength, const double *InvChannelAlpha2, const double *Chan2M3Start,
                          const double *QLimit, double Beta, double InvBeta, int64_t n)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        double v = ChanLength[p] * ChannelAlpha2[p] * pow(Chan2QKin[p], Beta);
        double diff = v - Chan2M3Start[p];
        if (diff < 0.0) v = Chan2M3Start[p];
        Chan2M3Kin[p] = v;
        CrossSection2Area[p] = (v - Chan2M3Start[p]) * InvChanLength[p];
        double q2 = pow(v * InvChanLength[p] * InvChannelAlpha2[p], InvBeta);
        Chan2QKin[p] = q2;
        double q = ChanQKin[p] + q2 - QLimit[p];
        ChanQ[p] = (q > 0.0 || isnan(q)) ? q : 0.0; /* np.maximum(q, 0.0) */
    }
}

/* routing.py:693-703 */
void lfo_velocity(const double *ChanM3Kin, const double *ChanQKin, const double *InvChanLength,
                  const double *PixelArea, double DtSec, int64_t n, double *FlowVelocity, double *TravelDistance)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) {
        double area = ChanM3Kin[p] * InvChanLength[p];
        if (area < 0.01) area = 0.01; /* np.maximum(area, 0.01); NaN propagates */
        double v1 = ChanQKin[p] / area, v2 = 0.36 * pow(ChanQKin[p], 0.24);
        double v = (v2 < v1) ? v2 : v1; /* np.minimum; NaN propagates */
        if (isnan(v2)) v = v2;
        double sinu = sqrt(PixelArea[p]) * InvChanLength[p];
        if (sinu > 1) sinu = 1; /* np.minimum(sinu, 1) */
        v *= sinu;
        FlowVelocity[p] = v;
        TravelDistance[p] = v * DtSec;
    }
}

/* ------------------------------------------------------------------------------------------------
 * a20: one-hop LDD upstream reduction == np.bincount(downstruct, weights)[:N]
 * (routing.py:159-164, lakes.py:215): sum in ascending source index.
 * ---------------------------------------------------------------------------------------------- */
/* parallel first touch: dst (fresh, untouched pages) <- src with the static schedule of every loop above, so that on a
 * multi-socket host each thread's share of a state vector lies in its own NUMA node (bench.py cpu_baseline) */
void lfo_parallel_copy(double *dst, const double *src, int64_t n)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < n; ++p) dst[p] = src[p];
}

void lfo_upstream_sum(const int32_t *downstruct, const double *w, int64_t n, double *out)
{
    for (int64_t p = 0; p < n; ++p) out[p] = 0.0;
    for (int64_t p = 0; p < n; ++p)
        if (downstruct[p] >= 0 && downstruct[p] < n) out[downstruct[p]] += w[p];
}

/* ------------------------------------------------------------------------------------------------
 * a15: interception_water_balance, soilloop.py:27-70.  Arrays [V,N] C-order; Rain [N].
 * ---------------------------------------------------------------------------------------------- */
void lfo_interception(double *Interception, double *TaInterception, double *LeafDrainage, double *CumInterception,
                      const double *LAI, const double *Rain, const double *TaInterceptionMax, double drainageK,
                      int64_t V, int64_t N)
{
    for (int64_t veg = 0; veg < V; ++veg) {
#pragma omp parallel for schedule(static)
        for (int64_t pix = 0; pix < N; ++pix) {
            int64_t i = veg * N + pix;
            double lai = LAI[i], smax;
            if (lai <= .1)
                smax = 0.;
            else if (lai <= 43.3)
                smax = 0.935 + 0.498 * lai - 0.00575 * sq(lai); /* LAI**2 */
            else
                smax = 11.718;
            if (smax > 0) {
                /* min(a, b, c) of Python: sequential */
                double v = smax - CumInterception[i];
                double b = smax * (1. - exp(-0.046 * lai * Rain[pix] / smax));
                v = dmin(v, b);
                v = dmin(v, Rain[pix]);
                Interception[i] = v;
                CumInterception[i] += v;
            } else
                Interception[i] = 0.;
            if (CumInterception[i] > 0.) {
                TaInterception[i] = dmax(dmin(CumInterception[i], TaInterceptionMax[i]), 0.);
                CumInterception[i] = dmax(CumInterception[i] - TaInterception[i], 0.);
                LeafDrainage[i] = drainageK * CumInterception[i];
                CumInterception[i] = dmax(CumInterception[i] - LeafDrainage[i], 0.);
            } else {
                TaInterception[i] = 0.;
                LeafDrainage[i] = 0.;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * a16: soilColumnsWaterBalance, soilloop.py:78-355 (+ helpers 360-396)
 * ---------------------------------------------------------------------------------------------- */

/* saturationDegree, soilloop.py:378-383 */
static inline double saturation_degree(double w, int pore, double wres, double ws)
{
    if (pore) return dmax(dmin((w - wres) / (ws - wres), 1.), 0.);
    return 0.;
}

/* unsaturatedConductivity, soilloop.py:360-367 */
static inline double unsat_k(double w, int pore, double wres, double ws, double ksat, double inv_m, double m)
{
    double s = saturation_degree(w, pore, wres, ws);
    double t = 1. - pow(1. - pow(s, inv_m), m);
    return ksat * sqrt(s) * sq(t); /* (...) ** 2 */
}

typedef struct {
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
    /* small */
    const int64_t *index_landuse_all;
    const uint8_t *is_irrigated, *is_paddy_irrig, *paddy_inactive; /* paddy_inactive [n_paddy, N] */
    double DtDay, AvWaterThreshold, CourantCrit, DrainedFraction;
    int64_t V, L, N;
} lfo_soil_args;

/* statistics of the last lfo_soil_columns call: columns, columns with > 1 sub-step, sum and max of nsub */
static int64_t g_soil_stats[4];
void lfo_soil_stats(int64_t out[4]) { memcpy(out, g_soil_stats, sizeof(g_soil_stats)); }
/* histogram of floor(log2(nsub)) over the columns of the last call (workload characterisation for the bench) */
static int64_t g_soil_hist[32];
void lfo_soil_substep_hist(int64_t out[32]) { memcpy(out, g_soil_hist, sizeof(g_soil_hist)); }

void lfo_soil_columns(const lfo_soil_args *A)
{
    int64_t st_cols = 0, st_multi = 0, st_sum = 0, st_max = 0;
    const int64_t N = A->N;
    memset(g_soil_hist, 0, sizeof(g_soil_hist));
    int64_t count_paddy = 0;
    for (int64_t veg = 0; veg < A->V; ++veg) {
        const uint8_t *inactive = NULL;
        int drained;
        if (A->is_paddy_irrig[veg]) { /* soilloop.py:107-113 */
            inactive = A->paddy_inactive + count_paddy * N;
            int any = 0;
            for (int64_t p = 0; p < N && !any; ++p) any = inactive[p];
            if (!any) continue;
            drained = 0;
            count_paddy += 1;
        } else
            drained = A->is_irrigated[veg] && (A->DrainedFraction > 0);
        const int64_t lo = A->index_landuse_all[veg] * N, vo = veg * N;
#pragma omp parallel for schedule(static) reduction(+ : st_cols, st_multi, st_sum) reduction(max : st_max)
        for (int64_t pix = 0; pix < N; ++pix) {
            if (inactive && !inactive[pix]) continue;
            const int64_t i = vo + pix, j = lo + pix;
            double DtDay = A->DtDay;
            /* available water, :131 */
            double awi = dmax((A->Rain[pix] + A->SnowMelt[pix]) + A->LeafDrainage[i] - A->Interception[i], 0.);
            double dslr = A->DSLR[i];
            if (awi > A->AvWaterThreshold)
                dslr = 1;
            else
                dslr += DtDay;
            A->DSLR[i] = dslr;
            double w1a = A->W1a[i], w1b = A->W1b[i], esact;
            int frozen = A->isFrozenSoil[pix];
            if (frozen)
                esact = 0.;
            else { /* :151-162 */
                esact = A->ESMax[i] * (sqrt(dslr) - sqrt(dslr - 1));
                esact = dmax(dmin(esact, A->W1[i] - A->WRes1[j]), 0.);
                double supply1a = w1a - A->WRes1a[j];
                double es1a = dmin(esact, supply1a);
                double es1b = dmax(esact - supply1a, 0.);
                w1a = dmax(w1a - es1a, A->WRes1a[j]);
                w1b = dmax(w1b - es1b, A->WRes1b[j]);
            }
            A->ESAct[i] = esact;
            double w1 = w1a + w1b; /* :163 */
            /* infiltration capacity, :168-179 */
            double relsat1 = A->PoreSpaceNotZero1a[j] ? dmin(w1 / A->WS1[j], 1.0) : 0.0;
            double satfrac = 1.0 - pow(1.0 - relsat1, A->b_Xinanjiang[pix]);
            double infpot = frozen ? 0.0 : A->StoreMaxPervious[j] * pow(1. - satfrac, A->PowerInfPot[pix]) * DtDay;
            /* preferential flow, :190-194 */
            double pref = pow(relsat1, A->PowerPrefFlow[pix]) * awi;
            A->PrefFlow[i] = pref;
            awi -= pref;
            A->AvailableWaterForInfiltration[i] = awi;
            /* infiltration, :201-211 */
            double inf = dmax(dmin(awi, infpot), 0.);
            double test1a = w1a + inf;
            w1a = dmin(A->WS1a[j], test1a);
            w1b += dmax(test1a - A->WS1a[j], 0.);
            double w2 = A->W2[i];
            /* conductivities & Courant, :223-249 */
            double k1a = unsat_k(w1a, A->PoreSpaceNotZero1a[j], A->WRes1a[j], A->WS1a[j], A->KSat1a[j],
                                 A->GenuInvM1a[j], A->GenuM1a[j]);
            double k1b = unsat_k(w1b, A->PoreSpaceNotZero1b[j], A->WRes1b[j], A->WS1b[j], A->KSat1b[j],
                                 A->GenuInvM1b[j], A->GenuM1b[j]);
            double k2 = unsat_k(w2, A->PoreSpaceNotZero2[j], A->WRes2[j], A->WS2[j], A->KSat2[j], A->GenuInvM2[j],
                                A->GenuM2[j]);
            double av1a = w1a - A->WRes1a[j], av1b = w1b - A->WRes1b[j], av2 = w2 - A->WRes2[j];
            double cap1 = A->WS1b[j] - w1b, cap2 = A->WS2[j] - w2;
            double ca = (av1a == 0) ? 0. : k1a * DtDay / av1a;
            double cb = (av1b == 0) ? 0. : k1b * DtDay / av1b;
            double cg = (av2 == 0) ? 0. : k2 * DtDay / av2;
            double courant = dmax(dmax(ca, cb), cg); /* max(a,b,c) sequential */
            double nsub_f = dmax(1, ceil(courant / A->CourantCrit));
            int64_t nsub = (int64_t)nsub_f;
            st_cols += 1;
            st_multi += nsub > 1;
            st_sum += nsub;
            if (nsub > st_max) st_max = nsub;
            {
                int c = 0;
                for (int64_t t = nsub; t > 1 && c < 31; t >>= 1) ++c;
                g_soil_hist[c] += 1;
            }
            /* sub-step loop, :266-312 */
            double wt1a = w1a, wt1b = w1b, wt2 = w2;
            double sa = 0., sb = 0., sg = 0.;
            double dtsub = DtDay / (double)nsub;
            for (int64_t s = 0; s < nsub; ++s) {
                if (s > 0) {
                    k1a = unsat_k(wt1a, A->PoreSpaceNotZero1a[j], A->WRes1a[j], A->WS1a[j], A->KSat1a[j],
                                  A->GenuInvM1a[j], A->GenuM1a[j]);
                    k1b = unsat_k(wt1b, A->PoreSpaceNotZero1b[j], A->WRes1b[j], A->WS1b[j], A->KSat1b[j],
                                  A->GenuInvM1b[j], A->GenuM1b[j]);
                    k2 = unsat_k(wt2, A->PoreSpaceNotZero2[j], A->WRes2[j], A->WS2[j], A->KSat2[j], A->GenuInvM2[j],
                                 A->GenuM2[j]);
                }
                double fa = dmin(k1a * dtsub, cap1);
                double fb = dmin(k1b * dtsub, cap2);
                double fg = dmin(k2 * dtsub, av2);
                av1a -= fa;
                av1b += fa - fb;
                av2 += fb - fg;
                wt1a = av1a + A->WRes1a[j];
                wt1b = av1b + A->WRes1b[j];
                wt2 = av2 + A->WRes2[j];
                cap1 = A->WS1b[j] - wt1b;
                cap2 = A->WS2[j] - wt2;
                sa += fa;
                sb += fb;
                sg += fg;
            }
            if (frozen) sa = sb = sg = 0.; /* :313-316 */
            A->SeepTopToSubA[i] = sa;
            A->SeepTopToSubB[i] = sb;
            A->SeepSubToGW[i] = sg;
            /* state update, :319-325 */
            w1a -= sa;
            w1b = w1b + sa - sb;
            w2 = w2 + sb - sg;
            w1 = w1a + w1b;
            inf -= dmax(w1a - A->WS1a[j], 0.);
            w1a = dmin(w1a, A->WS1a[j]);
            A->Infiltration[i] = inf;
            A->W1a[i] = w1a;
            A->W1b[i] = w1b;
            A->W1[i] = w1;
            A->W2[i] = w2;
            /* diagnostics, :330-336 (thetaFun :386, satFun :393) */
            A->Theta1a[i] = A->PoreSpaceNotZero1a[j] ? w1a / A->SoilDepth1a[j] : 0.;
            A->Theta1b[i] = A->PoreSpaceNotZero1b[j] ? w1b / A->SoilDepth1b[j] : 0.;
            A->Theta2[i] = A->PoreSpaceNotZero2[j] ? w2 / A->SoilDepth2[j] : 0.;
            A->Sat1a[i] = (w1a - A->WWP1a[j]) / (A->WFC1a[j] - A->WWP1a[j]);
            A->Sat1b[i] = (w1b - A->WWP1b[j]) / (A->WFC1b[j] - A->WWP1b[j]);
            A->Sat1[i] = (w1 - A->WWP1[j]) / (A->WFC1[j] - A->WWP1[j]);
            A->Sat2[i] = (w2 - A->WWP2[j]) / (A->WFC2[j] - A->WWP2[j]);
            /* upper zone, :340-354 */
            double uz = A->UZ[i];
            double uzout = dmin(A->UpperZoneK[pix] * uz, uz);
            uz = dmax(uz - uzout, 0.);
            if (drained) {
                uzout += A->DrainedFraction * sg;
                uz += (1 - A->DrainedFraction) * sg + pref;
            } else
                uz += sg + pref;
            double perc = dmin(A->GwPercStep[pix], uz);
            uz = dmax(uz - perc, 0.);
            A->UZOutflow[i] = uzout;
            A->GwPercUZLZ[i] = perc;
            A->UZ[i] = uz;
        }
    }
    g_soil_stats[0] = st_cols;
    g_soil_stats[1] = st_multi;
    g_soil_stats[2] = st_sum;
    g_soil_stats[3] = st_max;
}

/* ------------------------------------------------------------------------------------------------
 * Test helper for the row-block partition plan (lisflood-code_amd/csrc/lf_dist.hip): the same solve1Pixel
 * applied to positions [begin, end) of a state vector whose upstream cells are given by an index list
 * (local cells + ghost slots).  Positions inside a phase are topologically ordered, so a sequential walk
 * is a valid schedule.  constant[p] = a*Qold^beta + q*dx must be prepared by the caller.
 * ---------------------------------------------------------------------------------------------- */
void lfo_sweep_positions(double *state, const double *constant, const int32_t *ups_ptr, const int32_t *ups_idx,
                         const double *a, const double *ba, double beta, int64_t begin, int64_t end)
{
    double inv_beta = 1 / beta, b_minus_1 = beta - 1;
    for (int64_t p = begin; p < end; ++p) {
        int count = 0;
        double previous = -1.0, ups = 0.0;
        for (int32_t e = ups_ptr[p]; e < ups_ptr[p + 1]; ++e) ups += state[ups_idx[e]];
        double c = ups + constant[p];
        if (c <= NEWTON_TOL) {
            state[p] = 0;
            continue;
        }
        double t = ba[p] * pow(c, b_minus_1), secant;
        if (t <= 1)
            secant = c / (1 + t);
        else
            secant = c / (1 + pow(t, inv_beta));
        double other = pow((c - secant) / a[p], inv_beta);
        double q = (secant + other) / 2;
        double err = closure_error(q, c, a[p], beta);
        while (fabs(err) > NEWTON_TOL && q != previous && count < MAX_ITERS) {
            previous = q;
            q -= err / (1 + ba[p] * pow(q, b_minus_1));
            q = dmax(q, NEWTON_TOL);
            err = closure_error(q, c, a[p], beta);
            count += 1;
        }
        if (q == NEWTON_TOL) q = 0;
        state[p] = q;
    }
}


/* ------------------------------------------------------------------------------------------------
 * Structures inside the routing loop (routing.py:441-478): lakes.dynamic_inloop (lakes.py:199-297),
 * reservoir.dynamic_inloop (reservoir.py:173-322), inflow.dynamic_inloop (inflow.py:129-147),
 * transmission.dynamic_inloop (transmission.py:67-89) and the SideflowChanM3 assembly (routing.py:462-478).
 * Same argument block as lf_inloop_args of include/lisflood_amd.h, host pointers, pixel order.
 * Site inflow = np.bincount(downstruct, weights=ChanQ)[site]: the cells draining into the site, ascending id
 * (lakes.py:215, reservoir.py:190), given as CSR lists.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const double *ChanQ;
    int64_t n_lakes;
    const int32_t *lake_cell, *lake_ups_ptr, *lake_ups_idx;
    const double *LakeFactor, *LakeFactorSqr, *LakeAreaCC;
    double *LakeStorageM3CC, *LakeInflowOldCC, *LakeOutflowCC, *LakeStorageM3BalanceCC, *LakeLevelCC, *LakeInflowCC;
    double *QLakeOutM3Dt;
    int64_t n_res;
    const int32_t *res_cell, *res_ups_ptr, *res_ups_idx;
    const double *TotalReservoirStorageM3CC, *MinReservoirOutflowCC, *NormalReservoirOutflowCC,
        *NonDamagingReservoirOutflowCC, *ConservativeStorageLimitCC, *NormalStorageLimitCC, *FloodStorageLimitCC,
        *Normal_FloodStorageLimitCC, *DeltaO, *DeltaLN, *DeltaNFL;
    double *ReservoirStorageM3CC, *ReservoirFillCC, *ReservoirInflowCC;
    double *QResOutM3Dt;
    const double *QInM3Old, *QDelta;
    double *QInDt, *QinADDEDM3;
    const uint8_t *UpTrans;
    double *TransLossM3Dt, *TransCum;
    double TransPower1, TransPower2, TransSub;
    const double *ToChanM3RunoffDt, *EvaAddM3Dt, *WUseAddM3Dt, *ChannelToPolderM3Dt;
    double *SideflowChanM3;
    double DtRouting, InvNoRoutSteps;
    int64_t N;
    int32_t step;
} lfo_inloop_args;

/* np.minimum / np.maximum: a NaN operand wins, first operand first */
static double np_min(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b < a ? b : a)); }
static double np_max(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b > a ? b : a)); }

void lfo_inloop_structures(const lfo_inloop_args *A)
{
    const double dt = A->DtRouting;
    /* lakes: modified Puls with the outflow as a parabola of the level (lakes.py:215-258) */
    for (int64_t i = 0; i < A->n_lakes; ++i) {
        double qin = 0.0;
        for (int32_t e = A->lake_ups_ptr[i]; e < A->lake_ups_ptr[i + 1]; ++e) qin += A->ChanQ[A->lake_ups_idx[e]];
        A->LakeInflowCC[i] = qin;
        const double mean_in = (qin + A->LakeInflowOldCC[i]) * 0.5; /* lakes.py:218 */
        A->LakeInflowOldCC[i] = qin;
        const double si = A->LakeStorageM3CC[i] / dt - 0.5 * A->LakeOutflowCC[i] + mean_in; /* :224 */
        const double root = -A->LakeFactor[i] + sqrt(A->LakeFactorSqr[i] + 2 * si);           /* :228 */
        const double qout = root * root;
        A->LakeOutflowCC[i] = qout;
        const double vol_out = qout * dt;
        double st = (si - qout * 0.5) * dt; /* :245 */
        if (st < 0 || st != st) st = 0;     /* :250-255 */
        A->LakeStorageM3CC[i] = st;
        A->LakeStorageM3BalanceCC[i] += mean_in * dt - vol_out;
        A->LakeLevelCC[i] = st / A->LakeAreaCC[i];
        A->QLakeOutM3Dt[A->lake_cell[i]] = vol_out;
    }
    /* reservoirs: piecewise outflow rule on the filling fraction (reservoir.py:190-296) */
    for (int64_t r = 0; r < A->n_res; ++r) {
        const double per_day = 1 / 86400.0;
        double qin = 0.0;
        for (int32_t e = A->res_ups_ptr[r]; e < A->res_ups_ptr[r + 1]; ++e) qin += A->ChanQ[A->res_ups_idx[e]];
        A->ReservoirInflowCC[r] = qin;
        const double cap = A->TotalReservoirStorageM3CC[r];
        double st = A->ReservoirStorageM3CC[r] + qin * dt; /* reservoir.py:206 */
        const double fill = st / cap;
        const double qmin = A->MinReservoirOutflowCC[r], qnorm = A->NormalReservoirOutflowCC[r],
                     qnd = A->NonDamagingReservoirOutflowCC[r];
        const double two_lc = 2 * A->ConservativeStorageLimitCC[r], ln = A->NormalStorageLimitCC[r],
                     lf = A->FloodStorageLimitCC[r], lnf = A->Normal_FloodStorageLimitCC[r];
        /* the four candidate rules, :212-229, then the cascade of np.where in the reference's order, :235-245 */
        const double rule1 = np_min(qmin, st * per_day);
        const double rule2 = qmin + A->DeltaO[r] * (fill - two_lc) / A->DeltaLN[r];
        const double rule3 = qnorm + ((fill - lnf) / A->DeltaNFL[r]) * (qnd - qnorm);
        const double rule4 = np_max((fill - lf - 0.01) * cap * per_day, np_min(qnd, np_max(qin * 1.2, qnorm)));
        double q = rule1;
        if (fill > two_lc) q = rule2;
        if (fill > ln) q = qnorm;
        if (fill > lnf) q = rule3;
        if (fill > lf) q = rule4;
        const double damped = np_min(q, np_max(qin, qnorm)); /* :247-251 */
        if ((q > 1.2 * qin) && (q > qnorm) && (fill < lf)) q = damped;
        double vol_out = q * dt;
        vol_out = np_min(vol_out, st);       /* :253-258: not more than is stored, not less than the overflow */
        vol_out = np_max(vol_out, st - cap);
        st -= vol_out;
        double f2 = st / cap;
        if (f2 != f2 || f2 < 0) f2 = 0;
        A->ReservoirStorageM3CC[r] = st;
        A->ReservoirFillCC[r] = f2;
        A->QResOutM3Dt[A->res_cell[r]] = vol_out;
    }
    /* inflow hydrographs, transmission loss, sideflow assembly */
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < A->N; ++p) {
        double side = A->ToChanM3RunoffDt[p];
        if (A->EvaAddM3Dt) side -= A->EvaAddM3Dt[p];
        if (A->WUseAddM3Dt) side -= A->WUseAddM3Dt[p];
        if (A->QInM3Old) { /* inflow.py:142-144 */
            const double qin = (A->QInM3Old[p] + (A->step + 1) * A->QDelta[p]) * A->InvNoRoutSteps;
            A->QInDt[p] = qin;
            A->QinADDEDM3[p] = (A->step < 1 ? 0.0 : A->QinADDEDM3[p]) + qin;
            side += qin;
        }
        if (A->UpTrans) { /* transmission.py:76-87 */
            const double q = A->ChanQ[p];
            const double below = A->UpTrans[p] ? pow(pow(q, A->TransPower2) - A->TransSub, A->TransPower1) : q;
            const double loss = (q - below) * dt;
            A->TransLossM3Dt[p] = loss;
            A->TransCum[p] += loss;
            side -= loss;
        }
        if (A->QLakeOutM3Dt) side += A->QLakeOutM3Dt[p];
        if (A->QResOutM3Dt) side += A->QResOutM3Dt[p];
        if (A->ChannelToPolderM3Dt) side -= A->ChannelToPolderM3Dt[p];
        A->SideflowChanM3[p] = side;
    }
}


/* ------------------------------------------------------------------------------------------------
 * Per-pixel aggregates between the soil columns and surface routing (Lisflood_dynamic.py:129-149):
 * opensealed.dynamic (opensealed.py:40-71), soil.dynamic_perpixel (soil.py:471-514), groundwater.dynamic
 * (groundwater.py:134-180), three prescribed vegetation fractions.  Same argument block as lf_pixel_args.
 * deffraction(X) = (SoilFraction * X).sum("vegetation") = ((f0*x0 + f1*x1) + f2*x2)  (soil.py:460-468).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    const double *SoilFraction, *TaInterception, *Ta, *ESAct, *PrefFlow, *Infiltration, *SeepTopToSubA, *SeepTopToSubB,
        *SeepSubToGW, *Theta1a, *Theta1b, *Theta2, *W1a, *W1b, *W2, *UZOutflow, *GwPercUZLZ, *SoilDepthTotal;
    const double *Rain, *SnowMelt, *EWRef, *SMaxSealed, *DirectRunoffFraction, *WaterFraction, *LowerZoneK, *LZThreshold,
        *GwLossStep;
    double *CumInterSealed, *LZ, *LZInflowCUM, *TaInterceptionCUM, *TaCUM, *ESActCUM, *GwLossCUM;
    double *RainSnowmelt, *EWaterAct, *InterSealed, *TASealed, *DirectRunoff, *TaInterceptionAll, *TaPixel, *ESActPixel,
        *PrefFlowPixel, *InfiltrationPixel, *ThetaAll, *SeepTopToSubPixelA, *SeepTopToSubPixelB, *SeepSubToGWPixel,
        *Theta1aPixel, *Theta1bPixel, *Theta2Pixel, *LZOutflow, *UZOutflowPixel, *GwPercUZLZPixel, *GwLossLZ, *LZAvInflow,
        *LZOutflowToChannelPixel;
    double *Theta;
    double InvDtDay, TimeSinceStart;
    int64_t N;
} lfo_pixel_args;

static double frac_sum(const double *f, const double *x, int64_t N, int64_t p)
{
    return (f[p] * x[p] + f[N + p] * x[N + p]) + f[2 * N + p] * x[2 * N + p];
}

void lfo_pixel_aggregates(const lfo_pixel_args *A)
{
    const int64_t N = A->N;
    const double *f = A->SoilFraction;
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < N; ++p) {
        /* direct runoff from sealed area and open water, opensealed.py:45-70 */
        const double ewref = A->EWRef[p], sealed = A->DirectRunoffFraction[p], water = A->WaterFraction[p];
        const double supply = np_max(A->Rain[p] + A->SnowMelt[p], 0.0);
        const double ewact = np_max(np_min(ewref, supply) * 1.0, 0.0);
        double store = A->CumInterSealed[p];
        const double caught = np_min(np_max(A->SMaxSealed[p] - store, 0.0), supply);
        store += caught;
        const double evap = np_max(np_min(store, ewref), 0.0);
        store = np_max(store - evap, 0.0);
        A->RainSnowmelt[p] = supply;
        A->EWaterAct[p] = ewact;
        A->InterSealed[p] = caught;
        A->TASealed[p] = evap;
        A->CumInterSealed[p] = store;
        A->DirectRunoff[p] = sealed * (supply - caught) + water * (supply - ewact);
        /* fraction-weighted pixel totals, soil.py:475-513 */
        const double ta_int = frac_sum(f, A->TaInterception, N, p) + sealed * evap;
        A->TaInterceptionAll[p] = ta_int;
        A->TaInterceptionCUM[p] += ta_int;
        const double ta = frac_sum(f, A->Ta, N, p);
        A->TaPixel[p] = ta;
        A->TaCUM[p] += ta;
        const double es = frac_sum(f, A->ESAct, N, p) + water * ewact;
        A->ESActPixel[p] = es;
        A->ESActCUM[p] += es;
        A->PrefFlowPixel[p] = frac_sum(f, A->PrefFlow, N, p);
        A->InfiltrationPixel[p] = frac_sum(f, A->Infiltration, N, p);
        double th[3];
        for (int v = 0; v < 3; ++v) {
            const int64_t i = v * N + p;
            th[v] = f[i] * (A->W1a[i] + A->W1b[i] + A->W2[i]) / A->SoilDepthTotal[i];
            A->Theta[i] = th[v];
        }
        const double fsum = (f[p] + f[N + p]) + f[2 * N + p];
        A->ThetaAll[p] = (fsum > 0) ? ((th[0] + th[1]) + th[2]) / fsum : 0.0;
        A->SeepTopToSubPixelA[p] = frac_sum(f, A->SeepTopToSubA, N, p);
        A->SeepTopToSubPixelB[p] = frac_sum(f, A->SeepTopToSubB, N, p);
        A->SeepSubToGWPixel[p] = frac_sum(f, A->SeepSubToGW, N, p);
        A->Theta1aPixel[p] = frac_sum(f, A->Theta1a, N, p);
        A->Theta1bPixel[p] = frac_sum(f, A->Theta1b, N, p);
        A->Theta2Pixel[p] = frac_sum(f, A->Theta2, N, p);
        /* lower groundwater zone, groundwater.py:137-180 */
        double lz = A->LZ[p];
        const double out = np_max(np_min(A->LowerZoneK[p] * lz, lz - A->LZThreshold[p]), 0.0);
        A->LZOutflow[p] = out;
        lz -= out;
        A->UZOutflowPixel[p] = frac_sum(f, A->UZOutflow, N, p);
        const double perc = frac_sum(f, A->GwPercUZLZ, N, p);
        A->GwPercUZLZPixel[p] = perc;
        lz += perc;
        const double loss = np_max(np_min(A->GwLossStep[p], lz), 0.0);
        lz -= loss;
        A->GwLossLZ[p] = loss;
        A->LZ[p] = lz;
        const double cum = np_max(A->LZInflowCUM[p] + (perc - loss), 0.0);
        A->LZInflowCUM[p] = cum;
        A->GwLossCUM[p] += loss;
        A->LZAvInflow[p] = (cum * A->InvDtDay) / A->TimeSinceStart;
        A->LZOutflowToChannelPixel[p] = out;
    }
}


/* ------------------------------------------------------------------------------------------------
 * surface_routing.dynamic (surface_routing.py:115-212) around its three router calls:
 * lfo_surface_pre  -- runoff components and the three lateral inflows (:122-149)
 * lfo_surface_post -- volumes from the routed discharges, flow into the channel, water depth (:191-212)
 * V = 3 prescribed fractions in the order Rainfed, Forest, Irrigated; OFAlpha rows = (Other, Forest, Direct)
 * (Lisflood_initial.py:288-290).  side[3][N] receives (Direct, Other, Forest).
 * ---------------------------------------------------------------------------------------------- */
void lfo_surface_pre(const double *soil_fraction, const double *avail, const double *infiltration,
                     const double *direct_runoff, const double *uz_out, const double *lz_out, double mm_to_m3,
                     double inv_pixel_length, double inv_dt_sec, int64_t N, double *surface_run_soil,
                     double *surface_runoff, double *total_runoff, double *side)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < N; ++p) {
        double part[3];
        for (int l = 0; l < 3; ++l) {
            const int64_t i = l * N + p;
            part[l] = soil_fraction[i] * np_max(avail[i] - infiltration[i], 0.);
            surface_run_soil[i] = part[l];
        }
        const double surf = direct_runoff[p] + ((part[0] + part[1]) + part[2]);
        surface_runoff[p] = surf;
        total_runoff[p] = surf + uz_out[p] + lz_out[p];
        side[p] = direct_runoff[p] * mm_to_m3 * inv_pixel_length * inv_dt_sec;
        side[N + p] = (part[0] + part[2]) * mm_to_m3 * inv_pixel_length * inv_dt_sec;
        side[2 * N + p] = part[1] * mm_to_m3 * inv_pixel_length * inv_dt_sec;
    }
}

void lfo_surface_post(const double *q_direct, const double *q_other, const double *q_forest, const double *of_alpha,
                      const uint8_t *is_channel, const double *uz_out, const double *lz_out, double beta,
                      double pixel_length, double dt_sec, double mm_to_m3, double m3_to_mm, double inv_no_rout_steps,
                      int64_t N, double *m3_direct, double *m3_other, double *m3_forest, double *to_chan_m3,
                      double *water_depth, double *to_chan_runoff, double *to_chan_runoff_dt)
{
#pragma omp parallel for schedule(static)
    for (int64_t p = 0; p < N; ++p) {
        const double vd = pixel_length * of_alpha[2 * N + p] * pow(q_direct[p], beta);
        const double vo = pixel_length * of_alpha[p] * pow(q_other[p], beta);
        const double vf = pixel_length * of_alpha[N + p] * pow(q_forest[p], beta);
        m3_direct[p] = vd;
        m3_other[p] = vo;
        m3_forest[p] = vf;
        const double into_channel = is_channel[p] ? (q_direct[p] + q_other[p] + q_forest[p]) * dt_sec : 0.;
        to_chan_m3[p] = into_channel;
        water_depth[p] = (vd + vo + vf) * m3_to_mm;
        const double run = (uz_out[p] + lz_out[p]) * mm_to_m3 + into_channel;
        to_chan_runoff[p] = run;
        to_chan_runoff_dt[p] = run * inv_no_rout_steps;
    }
}


/* ------------------------------------------------------------------------------------------------
 * soilloop.dynamic_canopy (soilloop.py:519-627) for the prescribed vegetation fractions: interception (the a15
 * kernel inlined, :27-70), potential transpiration (:549-556), water-stress reduction and the abstraction of the
 * transpiration from soil layers 1a / 1b (:564-627).  [V,N] arrays by vegetation row, [L,N] by land-use row
 * (landuse[v] = index_landuse_prescr), Rain/EWRef/ETRef/isFrozenSoil per pixel.
 * ---------------------------------------------------------------------------------------------- */
void lfo_canopy(double *Interception, double *TaInterception, double *LeafDrainage, double *CumInterception,
                double *potential_transpiration, double *RWS, double *Ta, double *W1a, double *W1b, double *W1,
                const double *LAI, const double *LAITerm, const double *CropCoef, const double *CropGroupNumber,
                const double *WFC1, const double *WFC1a, const double *WFC1b, const double *WWP1, const double *WWP1a,
                const double *WWP1b, const double *Rain, const double *EWRef, const double *ETRef,
                const uint8_t *isFrozenSoil, const int64_t *landuse, double LeafDrainageK, double InvDtDay, int64_t V,
                int64_t N)
{
#pragma omp parallel for schedule(static)
    for (int64_t pix = 0; pix < N; ++pix) {
        const double rain = Rain[pix], ewref = EWRef[pix], etref = ETRef[pix];
        for (int64_t veg = 0; veg < V; ++veg) {
            const int64_t i = veg * N + pix, j = landuse[veg] * N + pix;
            /* interception */
            const double bare = 1. - LAITerm[i];
            const double ta_max = ewref * bare;
            const double lai = LAI[i];
            double smax = (lai <= .1) ? 0. : (lai <= 43.3 ? 0.935 + 0.498 * lai - 0.00575 * (lai * lai) : 11.718);
            double cum = CumInterception[i], caught = 0.;
            if (smax > 0) {
                caught = dmin(dmin(smax - cum, smax * (1. - exp(-0.046 * lai * rain / smax))), rain);
                cum += caught;
            }
            double ta_int = 0., drain = 0.;
            if (cum > 0.) {
                ta_int = dmax(dmin(cum, ta_max), 0.);
                cum = dmax(cum - ta_int, 0.);
                drain = LeafDrainageK * cum;
                cum = dmax(cum - drain, 0.);
            }
            Interception[i] = caught;
            TaInterception[i] = ta_int;
            LeafDrainage[i] = drain;
            CumInterception[i] = cum;
            /* potential transpiration */
            const double pot = np_max(CropCoef[j] * etref * bare - ta_int, 0.);
            potential_transpiration[i] = pot;
            /* soil water depletion fraction (crop group number), critical soil moisture, stress factor */
            const double cgn = CropGroupNumber[j];
            const double e = np_min(0.1 * etref * InvDtDay, 1.0);
            double swdf = 1 / (0.76 + 1.5 * e) - 0.10 * (5 - cgn);
            if (cgn <= 2.5) swdf = swdf + (e - 0.6) / (cgn * (cgn + 3));
            swdf = np_max(np_min(swdf, 1.0), 0.);
            const double wcrit1 = ((1 - swdf) * (WFC1[j] - WWP1[j])) + WWP1[j];
            const double wcrit1a = ((1 - swdf) * (WFC1a[j] - WWP1a[j])) + WWP1a[j];
            const double wcrit1b = ((1 - swdf) * (WFC1b[j] - WWP1b[j])) + WWP1b[j];
            const double w1 = W1[j]; /* land-use row, :592 */
            double rws = ((wcrit1 - WWP1[j]) > 0) ? (w1 - WWP1[j]) / (wcrit1 - WWP1[j]) : 1.;
            rws = np_max(np_min(rws, 1.), 0.);
            RWS[i] = rws;
            double ta = np_min(rws * pot, np_max(w1 - WWP1[j], 0.));
            if (isFrozenSoil[pix]) ta = 0.;
            Ta[i] = ta;
            /* abstraction: first what each layer holds above its critical amount, the rest by available water */
            double w1a = W1a[j], w1b = W1b[j];
            double from_a = np_min(ta, np_max(w1a - wcrit1a, 0.));
            double rest = np_max(ta - from_a, 0.);
            double from_b = np_min(rest, np_max(w1b - wcrit1b, 0.));
            rest = np_max(rest - from_b, 0.);
            const double left_a = np_max(w1a - from_a - WWP1a[j], 0.), left_b = np_max(w1b - from_b - WWP1b[j], 0.);
            const double left = left_a + left_b;
            from_a += (left > 0 ? left_a / left : 0.) * rest;
            from_b += (left > 0 ? left_b / left : 0.) * rest;
            w1a -= from_a;
            w1b -= from_b;
            W1a[j] = w1a;
            W1b[j] = w1b;
            W1[i] = w1a + w1b; /* vegetation row, :627 */
        }
    }
}


/* ------------------------------------------------------------------------------------------------
 * suctionUnsaturatedSoilPF + pressureHead (soilloop.py:427-432, 673-695): pF of the three soil layers.
 * ---------------------------------------------------------------------------------------------- */
static double lfo_pf_layer(double w, int pore, double wres, double ws, double inv_alpha, double inv_m, double inv_n,
                           double head_max)
{
    double sat = 0.;
    if (pore) sat = dmax(dmin((w - wres) / (ws - wres), 1.), 0.);      /* saturationDegree, :378-383 */
    double head = head_max;
    if (sat != 0) head = dmin(head_max, inv_alpha * pow(pow(1. / sat, inv_m) - 1., inv_n));   /* :427-432 */
    return head > 0 ? log10(head) : -1.;                                /* :691-695 */
}

void lfo_soil_pf(double *pF0, double *pF1, double *pF2, const double *W1a, const double *W1b, const double *W2,
                 const double *WRes1a, const double *WRes1b, const double *WRes2, const double *WS1a, const double *WS1b,
                 const double *WS2, const uint8_t *P1a, const uint8_t *P1b, const uint8_t *P2, const double *IA1a,
                 const double *IA1b, const double *IA2, const double *IM1a, const double *IM1b, const double *IM2,
                 const double *IN1a, const double *IN1b, const double *IN2, const int64_t *landuse, double HeadMax, int64_t V,
                 int64_t N)
{
    for (int64_t veg = 0; veg < V; ++veg)
#pragma omp parallel for schedule(static)
        for (int64_t pix = 0; pix < N; ++pix) {
            const int64_t i = veg * N + pix, j = landuse[veg] * N + pix;
            pF0[i] = lfo_pf_layer(W1a[i], P1a[j], WRes1a[j], WS1a[j], IA1a[j], IM1a[j], IN1a[j], HeadMax);
            pF1[i] = lfo_pf_layer(W1b[i], P1b[j], WRes1b[j], WS1b[j], IA1b[j], IM1b[j], IN1b[j], HeadMax);
            pF2[i] = lfo_pf_layer(W2[i], P2[j], WRes2[j], WS2[j], IA2[j], IM2[j], IN2[j], HeadMax);
        }
}
