// lf_structures.h -- lakes and reservoirs of the routing loop, one lane per site; shared by the sub-step-by-sub-step
// path (lf_modules.hip: k_inloop_sites) and the fused sub-step wavefront (lf_router.hip: k_sites_wave).
#pragma once
#include "lf_common.h"

// np.minimum / np.maximum: NaN propagates, first argument first
__device__ __forceinline__ double lf_npmin(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b < a ? b : a)); }
__device__ __forceinline__ double lf_npmax(double a, double b) { return (a != a) ? a : ((b != b) ? b : (b > a ? b : a)); }

// lakes.dynamic_inloop (lakes.py:215-258) for lake i (i < n_lakes) or reservoir.dynamic_inloop (reservoir.py:190-296)
// for reservoir i - n_lakes: inflow = np.bincount(downstruct, weights=ChanQ)[site] in ascending source id, then the
// site's storage / outflow update; the outflow volume goes to the dense QLakeOutM3Dt / QResOutM3Dt at the site cell.
__device__ __forceinline__ void lf_site_update(const lf_inloop_args &A, long long i)
{
    if (i < A.n_lakes) {
        double inflow = 0.0; // np.bincount(downstruct, weights=ChanQ)[LakeIndex]: ascending source id
        for (int e = A.lake_ups_ptr[i]; e < A.lake_ups_ptr[i + 1]; ++e) inflow += A.ChanQ[A.lake_ups_idx[e]];
        A.LakeInflowCC[i] = inflow;
        const double lake_in = (inflow + A.LakeInflowOldCC[i]) * 0.5;
        A.LakeInflowOldCC[i] = inflow;
        const double si = A.LakeStorageM3CC[i] / A.DtRouting - 0.5 * A.LakeOutflowCC[i] + lake_in;
        const double y = -A.LakeFactor[i] + sqrt(A.LakeFactorSqr[i] + 2 * si);
        const double out = y * y; // np.square
        A.LakeOutflowCC[i] = out;
        const double out_m3 = out * A.DtRouting;
        double st = (si - out * 0.5) * A.DtRouting;
        if (st < 0 || st != st) st = 0; // lakes.py:250-255
        A.LakeStorageM3CC[i] = st;
        A.LakeStorageM3BalanceCC[i] += lake_in * A.DtRouting - out_m3;
        A.LakeLevelCC[i] = st / A.LakeAreaCC[i];
        A.QLakeOutM3Dt[A.lake_cell[i]] = out_m3;
    }
    const long long r = i - A.n_lakes;
    if (r >= 0 && r < A.n_res) {
        const double inv_day = 1 / 86400.0; // 1 / float(86400)
        double inflow = 0.0;
        for (int e = A.res_ups_ptr[r]; e < A.res_ups_ptr[r + 1]; ++e) inflow += A.ChanQ[A.res_ups_idx[e]];
        A.ReservoirInflowCC[r] = inflow;
        const double tot = A.TotalReservoirStorageM3CC[r];
        double st = A.ReservoirStorageM3CC[r] + inflow * A.DtRouting;
        const double fill = st / tot;
        const double qmin = A.MinReservoirOutflowCC[r], qnorm = A.NormalReservoirOutflowCC[r],
                     qnd = A.NonDamagingReservoirOutflowCC[r];
        const double lc2 = 2 * A.ConservativeStorageLimitCC[r], ln = A.NormalStorageLimitCC[r],
                     lf = A.FloodStorageLimitCC[r], lnf = A.Normal_FloodStorageLimitCC[r];
        const double o1 = lf_npmin(qmin, st * inv_day);
        const double o2 = qmin + A.DeltaO[r] * (fill - lc2) / A.DeltaLN[r];
        const double o3b = qnorm + ((fill - lnf) / A.DeltaNFL[r]) * (qnd - qnorm);
        const double tmp = lf_npmin(qnd, lf_npmax(inflow * 1.2, qnorm));
        const double o4 = lf_npmax((fill - lf - 0.01) * tot * inv_day, tmp);
        double o = o1;
        if (fill > lc2) o = o2;
        if (fill > ln) o = qnorm;
        if (fill > lnf) o = o3b;
        if (fill > lf) o = o4;
        const double tmp2 = lf_npmin(o, lf_npmax(inflow, qnorm));
        if ((o > 1.2 * inflow) && (o > qnorm) && (fill < lf)) o = tmp2;
        double out_m3 = o * A.DtRouting;
        out_m3 = lf_npmin(out_m3, st);
        out_m3 = lf_npmax(out_m3, st - tot);
        st -= out_m3;
        double f2 = st / tot;
        if (f2 != f2 || f2 < 0) f2 = 0;
        A.ReservoirStorageM3CC[r] = st;
        A.ReservoirFillCC[r] = f2;
        A.QResOutM3Dt[A.res_cell[r]] = out_m3;
    }
}
