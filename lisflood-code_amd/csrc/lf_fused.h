// lf_fused.h -- device code of the fused multi-sub-step routing wavefront (routing.py:512-603, 693-703 around
// kinematic_wave_parallel_tools.py:34-92), shared by the single-GPU router (lf_router.hip) and the row-block partition
// (lf_dist.hip).  See the comment block above lf_routing_substeps_fused in lf_router.hip for the scheme.
#pragma once
#include "lf_blocks.h"
#include "lf_structures.h"
#include "lf_sweep.h"

namespace {

constexpr int kMaxPackedSteps = 128;

struct fused_args {
    lf_substep_args S;
    const int *__restrict__ ups_ptr;
    const double *__restrict__ a1, *__restrict__ a2, *__restrict__ dx;
    const long long *__restrict__ level_start;
    double *qr1, *qr2; // [2][N] router outputs by sub-step parity (main channel / floodplains)
    // time-major form (k_fused_level_steps): [nsteps][N] router outputs of EVERY sub-step, read by the level below
    double *hist1, *hist2;
    // slabs of router outputs kept for EVERY sub-step (row-block partition: what crosses a phase or a rank boundary, see
    // lf_dist.hip): the value of (slot, sub-step) sits at slot * root_ss + sub-step * root_st
    double *root1, *root2;
    long long nroots;
    long long root_ss, root_st; // slab index of (slot, sub-step) = slot * root_ss + sub-step * root_st
    long long n, side_stride;
    // several MODEL steps in one wavefront (lf_routing_model_steps_fused): sub-step s belongs to model step s / msteps; it
    // reads the sideflow at SideflowChanM3 + s * side_stride + (s / msteps) * side_mstride and adds to the discharge sum at
    // sumDisDay + (s / msteps) * n; the last sub-step of EVERY model step leaves what fused_cell's `last` leaves.  One model
    // step: msteps = nsteps, side_mstride = 0.
    int msteps;
    long long side_mstride;
    double dx_scalar, beta, inv_beta, b_minus_1;
    int kmax, nlevels, nsteps, t;
    int solve35; // router runs the beta = 3/5 quintic solve (false: general path, e.g. LF_GENERAL_POW=1)
    // *recompute != 0 (device flag, written by k_check_derived before the wavefront): for every cell InvChanLength ==
    // 1 / ChanLength, InvChannelAlpha(2) == 1 / ChannelAlpha(2) and the router's alpha*dx/dt == ChannelAlpha(2) * dx / dt
    // bit for bit -- as routing.initial computes them (routing.py:70, 236, 361; kinematic_wave_parallel.py:127) -- so the
    // five vectors are recomputed (IEEE division) instead of streamed: 40 of ~235 bytes per cell and sub-step
    const unsigned int *__restrict__ recompute;
    double dt; // the router's time step (a = alpha * dx / dt)
    const uint8_t *__restrict__ linked; // zero-length structure links: their router output is stored as 0
    const int *__restrict__ level_nlinked; // number of such links parked at the end of every level (k_fused_cones_split)
    const uint8_t *__restrict__ inert;  // cells whose sub-step is the identity while their state is all +0.0
    lf_inloop_args I;                   // STRUCT: lakes / reservoirs / inflow / transmission loss / sideflow assembly
    const int *__restrict__ site_level; // STRUCT: level of every lake, then every reservoir cell
    // 1-D grid packed by sub-step: blocks [blk_start[s], blk_start[s+1]) work on (level t - s, sub-step s), so no
    // block is launched for the part of a narrow level that a 2-D grid sized by the widest level would cover
    // (packed = 0: 2-D grid, blockIdx.y = sub-step; used when nsteps > kMaxPackedSteps)
    int packed;
    int blk_start[kMaxPackedSteps + 1];
    // level blocks (k_fused_blocked): t counts blocks instead of levels
    const int *__restrict__ fb_level, *__restrict__ fb_row, *__restrict__ fb_cone;
    const int *__restrict__ fb_off; // first entry of block b in fb_cone (rows of fb_level[b+1] - fb_level[b] starts)
    const int *__restrict__ fb_lvl2blk; // block of every level
    int fb_nblocks;
    int fb_block0; // first block of this wavefront (row-block partition: the blocks of one phase), else 0
    // row-block partition (DIST kernels): first upstream position when the upstream cells are a consecutive local run,
    // the upstream lists (same-phase position, or -(slab slot) - 1) behind ups_ptr, the slab slot of a cell's own
    // router outputs (-1 none); the slabs are root1 / root2
    const int *__restrict__ d_ups_base, *__restrict__ d_ups_idx, *__restrict__ d_out_slot;
    // k_fused_substeps beside k_fused_cones: the level of sub-step s at this wave time (-1: none) instead of t - s
    int use_lvl;
    int lvl[kMaxPackedSteps];
};

// sub-step s inside its model step (fused_args::msteps); s is uniform, so these are scalar operations
__device__ __forceinline__ bool fused_last(const fused_args &F, int s) { return (s + 1) % F.msteps == 0; }
__device__ __forceinline__ const double *fused_side(const fused_args &F, int s)
{
    return F.S.SideflowChanM3 + (long long)s * F.side_stride + (long long)(s / F.msteps) * F.side_mstride;
}
__device__ __forceinline__ double *fused_sum(const fused_args &F, int s) { return F.S.sumDisDay + (long long)(s / F.msteps) * F.n; }

__device__ __forceinline__ bool plus_zero(double x) { return __double_as_longlong(x) == 0; }
__device__ __forceinline__ bool same_bits(double a, double b) { return __double_as_longlong(a) == __double_as_longlong(b); }

// see fused_args::recompute; *flag is preset to 1
__global__ void __launch_bounds__(kBlock) k_check_derived(long long n, lf_substep_args A, const double *__restrict__ a1,
                                                          const double *__restrict__ a2, const double *__restrict__ dx,
                                                          double dx_scalar, double dt, unsigned int *flag)
{
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const double dxp = dx ? dx[p] : dx_scalar;
    bool ok = same_bits(A.InvChanLength[p], 1.0 / A.ChanLength[p]) && same_bits(A.InvChannelAlpha[p], 1.0 / A.ChannelAlpha[p]) &&
              same_bits(a1[p], A.ChannelAlpha[p] * dxp / dt);
    if (ok && A.split)
        ok = same_bits(A.InvChannelAlpha2[p], 1.0 / A.ChannelAlpha2[p]) && same_bits(a2[p], A.ChannelAlpha2[p] * dxp / dt);
    if (!ok) atomicAnd(flag, ~1u);
    // bit 1: the router's space step IS the channel length (kinematicWave(..., ChanLength, ...), routing.py:401-403)
    if (!dx || !same_bits(dx[p], A.ChanLength[p])) atomicAnd(flag, ~2u);
}
// bit 0: the five derived vectors are recomputed; bit 1: dx is read from ChanLength
__device__ __forceinline__ unsigned int derived_flags(const fused_args &F)
{
    typedef const unsigned int __attribute__((address_space(4))) *cptr; // scalar load (nothing in the wavefront writes it)
    return F.recompute != nullptr ? ((cptr)(unsigned long long)F.recompute)[0] : 0u;
}
__device__ __forceinline__ bool derived_recomputable(const fused_args &F) { return (derived_flags(F) & 1u) != 0u; }

// Non-channel land pixels sit in the channel router as isolated nodes without sideflow (routing.py:512): once their
// state is zero, a sub-step leaves every vector as it is.  inert[p] marks the candidates (static part of the test);
// the cell kernel then checks the state itself and returns early -- on real domains most land pixels are such cells,
// and being outlets without upstream cells they lie side by side in the last level, so whole lines are skipped.
// The parameters must be finite too: 0 * inf would turn the zero state into NaN, as it does in the reference.
__global__ void __launch_bounds__(kBlock) k_inert_flags(long long n, lf_substep_args A, const uint8_t *__restrict__ isolated,
                                                        const double *__restrict__ a1, const double *__restrict__ a2,
                                                        const double *__restrict__ dx, uint8_t *__restrict__ inert)
{
    const long long p = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    bool f = isolated[p] && !A.IsChannelKinematic[p];
    if (f) {
        const double t = A.InvChanLength[p] * A.ChanLength[p] * A.ChannelAlpha[p] * A.InvChannelAlpha[p] * a1[p] *
                         (dx ? dx[p] : 1.0);
        f = isfinite(t) && A.InvChanLength[p] >= 0.0;
    }
    if (f && A.split) {
        const double t = A.ChannelAlpha2[p] * A.InvChannelAlpha2[p] * a2[p];
        f = isfinite(t) && plus_zero(A.QLimit[p]) && plus_zero(A.Chan2QStart[p]) && plus_zero(A.Chan2M3Start[p]);
    }
    inert[p] = f ? 1 : 0;
}

// Lakes and reservoirs inside the wavefront: site v handles sub-step s = t - level(v) right BEFORE launch t of the
// cell kernel.  The cells that drain into v in the uncut LDD are zero-length links of the graph, i.e. on v's level:
// their ChanQ of sub-step s-1 was stored by launch t-1 and is overwritten only by launch t.
__global__ void __launch_bounds__(kBlock) k_sites_wave(fused_args F)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= F.I.n_lakes + F.I.n_res) return;
    const int s = F.t - F.site_level[i];
    if (s < 0 || s >= F.nsteps) return;
    lf_site_update(F.I, i);
}

// the same on level blocks (k_fused_cones): launch t works on (block t - s, sub-step s)
__global__ void __launch_bounds__(kBlock) k_sites_blocks(fused_args F)
{
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= F.I.n_lakes + F.I.n_res) return;
    const int s = F.t - F.fb_lvl2blk[F.site_level[i]];
    if (s < 0 || s >= F.nsteps) return;
    lf_site_update(F.I, i);
}

__device__ __forceinline__ double solve_any(double c, double ap, bool b35, const fused_args &F)
{
    if (b35 && lf_fast_range(c) && lf_fast_range(ap)) return (c <= LF_NEWTON_TOL) ? 0.0 : lf_solve_3_5(c, ap);
    return lf_solve_cell(c, ap, F.beta * ap, F.beta, F.inv_beta, F.b_minus_1);
}

#ifndef LF_FUSED_PAIRS
#define LF_FUSED_PAIRS 1 /* 0: the contiguous upstream run one value per load everywhere (A/B builds) */
#endif
// The same sum with the run read two values per load (four 16-byte loads instead of eight 8-byte ones, as the level
// sweep of lf_sweep.h does): for runs inside a vector whose positions all precede their reader -- the second value of the
// last pair may lie one position behind the run and is discarded.  NOT for the row-block partition's ghost runs, which may
// end where their buffer ends.
__device__ __forceinline__ double upstream_sum8_pairs(const double *q, int u0, int u1, int kmax)
{
    double v[8];
#if LF_FUSED_PAIRS
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double2 t = make_double2(0.0, 0.0);
        if (2 * j < kmax && u0 + 2 * j < u1) t = *(const double2 *)(q + u0 + 2 * j); // (8-byte aligned)
        v[2 * j] = t.x;
        v[2 * j + 1] = (u0 + 2 * j + 1 < u1) ? t.y : 0.0;
    }
#else
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (k < kmax && u0 + k < u1) ? q[u0 + k] : 0.0;
#endif
    double ups = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) ups += v[k];
    return ups;
}

// The run of BOTH router-output vectors of a cell (main channel, floodplain) requested together: eight 16-byte loads go out
// before any of them is used.  (One sum after the other, the compiler sinks `0.0 + v[0]` into the first load's branch and
// waits there -- for every load in flight, since loads behind branches cannot be counted -- before it requests the rest:
// two to four dependent round trips per sub-step of the time-major kernel instead of one.)
__device__ __forceinline__ void upstream_sum8_pairs2(const double *q1, const double *q2, bool two, int u0, int u1, int kmax,
                                                     double &s1, double &s2)
{
#if LF_FUSED_PAIRS
    double v1[8], v2[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double2 t = make_double2(0.0, 0.0);
        if (2 * j < kmax && u0 + 2 * j < u1) t = *(const double2 *)(q1 + u0 + 2 * j);
        v1[2 * j] = t.x;
        v1[2 * j + 1] = t.y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double2 t = make_double2(0.0, 0.0);
        if (two && 2 * j < kmax && u0 + 2 * j < u1) t = *(const double2 *)(q2 + u0 + 2 * j);
        v2[2 * j] = t.x;
        v2[2 * j + 1] = t.y;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(v1[k]), "+v"(v2[k]));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool have = u0 + 2 * j + 1 < u1;
        v1[2 * j + 1] = have ? v1[2 * j + 1] : 0.0;
        v2[2 * j + 1] = have ? v2[2 * j + 1] : 0.0;
    }
    double a = 0.0, b = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) a += v1[k];
#pragma unroll
    for (int k = 0; k < 8; ++k) b += v2[k];
    s1 = a;
    s2 = two ? b : 0.0;
#else
    s1 = upstream_sum8_pairs(q1, u0, u1, kmax);
    s2 = two ? upstream_sum8_pairs(q2, u0, u1, kmax) : 0.0;
#endif
}

__device__ __forceinline__ double upstream_sum8(const double *q, int u0, int u1, int kmax)
{
    double v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = (k < kmax && u0 + k < u1) ? q[u0 + k] : 0.0;
    double ups = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) ups += v[k];
    return ups;
}

// One (cell, sub-step) of the fused sub-steps: everything but the choice of the cell.  `UPS` sums the router outputs of
// the upstream cells (ascending pixel id) from the parity buffer it is handed: the contiguous range of the level
// layout, or the index lists and slabs of the row-block partition.
template <bool SPLIT, bool STRUCT, class UPS>
__device__ __forceinline__ void fused_cell(const fused_args &F, long long p, int s, const UPS &ups_of,
                                           long long root_slot = -1)
{
    const lf_substep_args &A = F.S;
    if (!STRUCT && F.inert && F.inert[p] && !fused_last(F, s)) { // see k_inert_flags; the last sub-step also writes
        bool zero = plus_zero(A.ChanQKin[p]) && plus_zero(A.ChanM3Kin[p]) && plus_zero(A.ChanQ[p]); // velocities
        if (SPLIT && zero)
            zero = plus_zero(A.Chan2QKin[p]) && plus_zero(A.Chan2M3Kin[p]) && plus_zero(A.CrossSection2Area[p]) &&
                   plus_zero(A.Sideflow1Chan[p]);
        if (zero) return;
    }
    const bool b35 = A.Beta == 0.6;     // fix-up round trips, as k_substep_main / k_substep_floodplain
    const bool s35 = F.solve35 != 0;    // router solve + old-discharge term, as the router itself
    const long long par = (long long)(s & 1) * F.n;
    // ---- every load first: the argument pointers are not restrict-qualified, so a store in the middle of the
    // ---- kernel would pin all later loads behind it (the kernel is a stream of ~30 vectors) -------------------
    const unsigned int dflags = derived_flags(F); // (uniform)
    const bool rc = (dflags & 1u) != 0u;
    const double len = A.ChanLength[p];
    const double dxp = (dflags & 2u) ? len : (F.dx ? F.dx[p] : F.dx_scalar);
    const double inv_len = rc ? 1.0 / len : A.InvChanLength[p];
    const bool is_chan = A.IsChannelKinematic[p] != 0;
    const bool cut = F.linked && F.linked[p];
    double side_m3, qin = 0, qin_added = 0, loss = 0, trans_cum = 0;
    if (!STRUCT)
        side_m3 = fused_side(F, s)[p];
    else { // inflow.py:142-144, transmission.py:76-87, sideflow assembly routing.py:462-478 -- as k_inloop_dense
        const lf_inloop_args &I = F.I;
        side_m3 = I.ToChanM3RunoffDt[p];
        if (I.EvaAddM3Dt) side_m3 -= I.EvaAddM3Dt[p];
        if (I.WUseAddM3Dt) side_m3 -= I.WUseAddM3Dt[p];
        if (I.QInM3Old) {
            qin = (I.QInM3Old[p] + (s + 1) * I.QDelta[p]) * I.InvNoRoutSteps;
            qin_added = (s < 1 ? 0.0 : I.QinADDEDM3[p]) + qin;
            side_m3 += qin;
        }
        if (I.UpTrans) {
            const double qc = A.ChanQ[p];
            const double tout = I.UpTrans[p] ? lf_pow_scalar_exponent(lf_pow_scalar_exponent(qc, I.TransPower2) - I.TransSub, I.TransPower1) : qc;
            loss = (qc - tout) * I.DtRouting;
            trans_cum = I.TransCum[p] + loss;
            side_m3 -= loss;
        }
        if (I.QLakeOutM3Dt) side_m3 += I.QLakeOutM3Dt[p];
        if (I.QResOutM3Dt) side_m3 += I.QResOutM3Dt[p];
        if (I.ChannelToPolderM3Dt) side_m3 -= I.ChannelToPolderM3Dt[p];
    }
    const double qold = A.ChanQKin[p], alpha1 = A.ChannelAlpha[p];
    const double inv_alpha1 = rc ? 1.0 / alpha1 : A.InvChannelAlpha[p], ap1 = rc ? alpha1 * dxp / F.dt : F.a1[p];
    double *const sum_dst = fused_sum(F, s);
    const double sum_old = sum_dst[p];
    const double ups1 = ups_of(F.qr1 + par, 0);
    double m3 = 0, m3_2 = 0, start = 0, m3limit = 0, q2start = 0, ap2 = 0, q2old = 0, alpha2 = 0, inv_alpha2 = 0, qlimit = 0,
           ups2 = 0;
    if (SPLIT) {
        m3 = A.ChanM3Kin[p];
        m3_2 = A.Chan2M3Kin[p];
        start = A.Chan2M3Start[p];
        m3limit = A.M3Limit[p];
        q2start = A.Chan2QStart[p];
        q2old = A.Chan2QKin[p];
        alpha2 = A.ChannelAlpha2[p];
        ap2 = rc ? alpha2 * dxp / F.dt : F.a2[p];
        inv_alpha2 = rc ? 1.0 / alpha2 : A.InvChannelAlpha2[p];
        qlimit = A.QLimit[p];
        ups2 = ups_of(F.qr2 + par, 1);
    }
    const bool last = fused_last(F, s);
    double pix_area = 0;
    if (last) pix_area = A.PixelArea[p];
    // ---- sideflow (routing.py:512, 524 / 549-567) ----
    const double side = is_chan ? side_m3 * inv_len * A.InvDtRouting : 0.0;
    double s1 = side, s2 = 0.0;
    if (!SPLIT) {
        if (isnan(side)) s1 = 0.0;
    } else {
        const double tot = m3 + m3_2;
        const double ratio = (tot > 0) ? m3 / tot : 0.0;
        s1 = ((tot - start) > m3limit) ? ratio * side : side;
        if (fabs(side) < 1e-7) s1 = side;
        s2 = (side - s1) + q2start * inv_len;
    }
    // ---- main channel: router call + fix-up (routing.py:526-532 / 573-578) ----
    const double cst = ap1 * (s35 ? lf_pow_3_5(qold) : pow(qold, F.beta)) + s1 * dxp;
    const double c = ups1 + cst;
    const double qr = solve_any(c, ap1, s35, F);
    double v = len * alpha1 * (b35 ? lf_pow_3_5(qr) : pow(qr, A.Beta));
    if (v < 0.0) v = 0.0;
    const double x = v * inv_len * inv_alpha1;
    const double q = b35 ? lf_pow_5_3(x) : pow(x, A.InvBeta);
    double chanq = q, q2r = 0, v2 = 0, q2 = 0;
    if (SPLIT) { // ---- floodplains (routing.py:583-603) ----
        const double cst2 = ap2 * (s35 ? lf_pow_3_5(q2old) : pow(q2old, F.beta)) + s2 * dxp;
        const double c2 = ups2 + cst2;
        q2r = solve_any(c2, ap2, s35, F);
        v2 = len * alpha2 * (b35 ? lf_pow_3_5(q2r) : pow(q2r, A.Beta));
        if ((v2 - start) < 0.0) v2 = start;
        const double x2 = v2 * inv_len * inv_alpha2;
        q2 = b35 ? lf_pow_5_3(x2) : pow(x2, A.InvBeta);
        chanq = q + q2 - qlimit;
        if (chanq < 0.0) chanq = 0.0;
    }
    // ---- stores ----
    if (STRUCT) {
        const lf_inloop_args &I = F.I;
        if (I.QInM3Old) {
            I.QInDt[p] = qin;
            I.QinADDEDM3[p] = qin_added;
        }
        if (I.UpTrans) {
            I.TransLossM3Dt[p] = loss;
            I.TransCum[p] = trans_cum;
        }
        I.SideflowChanM3[p] = side_m3;
    }
    F.qr1[par + p] = cut ? 0.0 : qr;
    if (root_slot >= 0) F.root1[root_slot * F.root_ss + s * F.root_st] = qr; // kept for the next tier / phase / rank
    // ChanQ, Sideflow1Chan and CrossSection2Area are results of a sub-step that no later sub-step reads (routing.py:557-603
    // recomputes them from the volumes) -- only the structures of the loop read ChanQ (STRUCT): without them the
    // intermediate values are not stored, the last sub-step's are (what the sub-step-by-sub-step sequence leaves behind)
    const bool keep = STRUCT || last;
    A.ChanM3Kin[p] = v;
    A.ChanQKin[p] = q;
    if (keep) A.ChanQ[p] = chanq;
    sum_dst[p] = sum_old + chanq;
    if (SPLIT) {
        if (keep) A.Sideflow1Chan[p] = s1;
        F.qr2[par + p] = cut ? 0.0 : q2r;
        if (root_slot >= 0) F.root2[root_slot * F.root_ss + s * F.root_st] = q2r;
        A.Chan2M3Kin[p] = v2;
        if (keep) A.CrossSection2Area[p] = (v2 - start) * inv_len;
        A.Chan2QKin[p] = q2;
    }
    if (last) { // routing.py:693-703
        double area = v * inv_len;
        if (area < 0.01) area = 0.01;
        const double v1 = q / area, vv2 = 0.36 * pow(q, 0.24);
        double vel = (vv2 < v1) ? vv2 : v1;
        if (isnan(vv2)) vel = vv2;
        double sinu = sqrt(pix_area) * inv_len;
        if (sinu > 1) sinu = 1;
        vel *= sinu;
        A.FlowVelocity[p] = vel;
        A.TravelDistance[p] = vel * A.DtSec;
    }
}

template <bool SPLIT, bool STRUCT>
__global__ void __launch_bounds__(kBlock) k_fused_substeps(fused_args F)
{
    int s, blk;
    if (F.packed) { // the last sub-step whose first block is <= blockIdx.x (starts are non-decreasing; independent
                    // scalar loads and compares -- a binary search would chain its kernarg loads)
        int cnt = 0, start = 0;
        for (int q = 0; q < F.nsteps; ++q) {
            const bool ge = (int)blockIdx.x >= F.blk_start[q];
            cnt += ge ? 1 : 0;
            start = ge ? F.blk_start[q] : start;
        }
        s = cnt - 1;
        blk = (int)blockIdx.x - start;
    } else {
        s = blockIdx.y;
        blk = blockIdx.x;
    }
    const int k = F.use_lvl ? F.lvl[s] : F.t - s; // level handled by this sub-step at wave time t
    if (k < 0 || k >= F.nlevels) return;
    const long long first = F.level_start[k];
    const long long i = (long long)blk * kBlock + threadIdx.x;
    if (i >= F.level_start[k + 1] - first) return;
    const long long p = first + i;
    const int u0 = F.ups_ptr[p], u1 = F.ups_ptr[p + 1];
    const int kmax = F.kmax;
    fused_cell<SPLIT, STRUCT>(F, p, s, [u0, u1, kmax](const double *q, int) { return upstream_sum8_pairs(q, u0, u1, kmax); });
}

// ---- several levels per launch: the wavefront over LEVEL BLOCKS, one workgroup per upstream cone --------------------
// One (level, sub-step) of k_fused_substeps is a dependent chain of ~4 memory round trips (kernel arguments -> level
// table -> upstream ranges -> router outputs and ~25 state vectors) plus the two closure solves: ~9 us of kernel and a
// ~2.5 us boundary on a latency-bound network (deep 5000^2: 5024 launches, 58 ms per model step).  Grouping several
// levels into one launch with a workgroup barrier between them does not help by itself (measured: 58 ms for 1, 2, 4 and
// 8 levels per launch) -- the chain is the cost, not the boundary.  This kernel shortens the chain:
//  * the levels are grouped into blocks of up to fb_lmax consecutive levels and launch t works on (block t - s,
//    sub-step s).  Every cell below the last level has exactly one downstream cell, in the next level, and the upstream
//    cells of a contiguous range of positions are a contiguous range of the level before (lf_common.h, sweep order): a
//    workgroup that owns a chunk of the block's LAST level owns the whole cone above it, level by level one contiguous
//    range (fb_cone: its starts, precomputed; chunks are cut so that no range exceeds the workgroup), and the cones tile
//    every level of the block -- no synchronisation between workgroups inside a block;
//  * inside a cone the router outputs travel through LDS (two buffers by level parity), so the barrier between two
//    levels waits for LDS only -- not for the state stores, which drain behind it -- and only the block's last level
//    writes its router outputs to the parity buffers in HBM (read by the next block in the next launch);
//  * the state of the NEXT level's cell is loaded before the current level is solved (it does not depend on anything
//    computed in this launch), so after a barrier a level costs: 8 LDS reads, the two solves, a few LDS writes.
// The arithmetic of a cell is fused_cell's, operation by operation: results are bit-identical.
struct cone_cell { // what a cell's load phase leaves in registers: loaded values only, nothing computed from them (a
                   // compare on a loaded value would make the wavefront wait for the loads right where they are issued)
    double dxp, inv_len, len, side_m3, ap1, qold, alpha1, inv_alpha1, sum_old;
    double m3, m3_2, start, m3limit, q2start, ap2, q2old, alpha2, inv_alpha2, qlimit, pix_area;
    double chanq_old, csa_old, sf1_old; // the inert test; STRUCT: chanq_old = ChanQ before the sub-step (transmission loss)
    // STRUCT: the terms of the sideflow assembly (routing.py:462-478), as k_inloop_dense / fused_cell read them
    double eva, wuse, qin_old, qdelta, qin_added_old, transcum, lakeout, resout, polder;
    int u0, u1;
    int base, slot; // DIST: consecutive local run of upstream cells (else -1: the list ups_idx[u0 .. u1)); slab slot
    unsigned char chan_raw, inert_raw, uptrans_raw, cut_raw;
    bool active;
};

// The loads and stores of a cone level (round 4).  A wavefront of this kernel is bound by its own instruction stream (one
// instruction at a time: profiles/pmc_r04g_fused_deep_5000.txt counts ~1 300 per level, more scalar than vector), so what
// a level's ~35 loads and ~10 stores cost in instructions AROUND them is what matters:
//  * address = array pointer (scalar registers) + 32-bit byte offset of the cell (one vector register for all arrays of
//    an element size): the `saddr` form of global_load / global_store, no address arithmetic per array (the fused plan is
//    only built for graphs below 2^29 cells: build_level_blocks in lf_router.hip, the phase plans in lf_dist.hip);
//  * ONE divergent region around all loads (lanes beyond the cone) and no uniform branch per optional vector: vectors that a
//    uniform condition switches off are skipped in a few GROUPS (the five derived statics when they are recomputed, the
//    inert test's four, the last sub-step's, the link flags), their fields keep the zeros of the kernel's start-up; the
//    optional vectors of the structures (each with its own null test) are read from a stand-in instead (the channel
//    lengths: any valid array of the size) and the value ignored.  The pointer loads from the kernel-argument segment then
//    come in batches in front of the loads (a handful of scalar-memory waits per level instead of ~25).
struct cone_range { // one level of a cone: cells [first, first + cnt)
    long long first;
    int cnt;
};
struct cone_at { // this lane's cell of the level: byte offsets into arrays of 8-, 4- and 1-byte elements
    unsigned o8, o4, o1;
    const char *standin;
};
__device__ __forceinline__ double cone_ld(const double *base, const cone_at &P) { return *(const double *)((const char *)base + P.o8); }
__device__ __forceinline__ double cone_ld(const double *base, const cone_at &P, bool on)
{
    return *(const double *)(((on && base) ? (const char *)base : P.standin) + P.o8);
}
__device__ __forceinline__ unsigned char cone_ld(const uint8_t *base, const cone_at &P) { return *((const uint8_t *)base + P.o1); }
__device__ __forceinline__ unsigned char cone_ld(const uint8_t *base, const cone_at &P, bool on)
{
    return *(const uint8_t *)(((on && base) ? (const char *)base : P.standin) + P.o1);
}
__device__ __forceinline__ int cone_ld(const int *base, const cone_at &P, unsigned plus = 0)
{
    return *(const int *)((const char *)base + P.o4 + plus);
}
__device__ __forceinline__ void cone_st(double *base, unsigned o8, double v) { *(double *)((char *)base + o8) = v; }

// loads only: nothing here may look at a loaded value (cone_fix does, after the wait)
template <bool SPLIT, bool STRUCT, bool DIST = false>
__device__ __forceinline__ void cone_load(const fused_args &F, const cone_range &L, int tid, int s, cone_cell &R, bool rc = false,
                                          bool dx_is_len = false)
{
    const lf_substep_args &A = F.S;
    R.active = tid < L.cnt;
    if (!R.active) return;
    const unsigned cell = (unsigned)L.first + (unsigned)tid;
    const cone_at P = {cell * 8u, cell * 4u, cell, (const char *)A.ChanLength};
    R.u0 = cone_ld(F.ups_ptr, P);
    R.u1 = cone_ld(F.ups_ptr, P, 4u);
    R.base = R.slot = -1;
    if (DIST) {
        R.base = cone_ld(F.d_ups_base, P);
        R.slot = cone_ld(F.d_out_slot, P);
    }
    // Vectors that a uniform condition switches off are skipped in GROUPS, one uniform branch each (a handful per level --
    // not one per vector): their fields keep what the kernel's start-up left there (zeros, `cone_cell x = {}`), which is
    // what their readers expect (the flags) or is never looked at (values read behind the same condition).
    if (F.dx && !dx_is_len) R.dxp = cone_ld(F.dx, P); // else cone_fix puts the scalar / the length there
    R.len = cone_ld(A.ChanLength, P);
    R.chan_raw = cone_ld(A.IsChannelKinematic, P);
    if (!rc) { // rc: cone_derive fills the five derived values in from len / alpha / dx
        R.inv_len = cone_ld(A.InvChanLength, P);
        R.ap1 = cone_ld(F.a1, P);
        R.inv_alpha1 = cone_ld(A.InvChannelAlpha, P);
        if (SPLIT) {
            R.ap2 = cone_ld(F.a2, P);
            R.inv_alpha2 = cone_ld(A.InvChannelAlpha2, P);
        }
    }
    const bool test_inert = !STRUCT && F.inert && !fused_last(F, s); // see k_inert_flags / fused_cell (uniform)
    if (!STRUCT)
        R.side_m3 = cone_ld(fused_side(F, s), P);
    else { // (absent vectors: a stand-in's value, which cone_compute does not look at -- it tests the same pointers)
        const lf_inloop_args &I = F.I;
        R.side_m3 = cone_ld(I.ToChanM3RunoffDt, P);
        R.eva = cone_ld(I.EvaAddM3Dt, P, true);
        R.wuse = cone_ld(I.WUseAddM3Dt, P, true);
        R.qin_old = cone_ld(I.QInM3Old, P, true);
        R.qdelta = cone_ld(I.QDelta, P, I.QInM3Old != nullptr);
        R.qin_added_old = cone_ld(I.QinADDEDM3, P, I.QInM3Old != nullptr);
        R.uptrans_raw = cone_ld(I.UpTrans, P, true);
        R.transcum = cone_ld(I.TransCum, P, I.UpTrans != nullptr);
        R.lakeout = cone_ld(I.QLakeOutM3Dt, P, true);
        R.resout = cone_ld(I.QResOutM3Dt, P, true);
        R.polder = cone_ld(I.ChannelToPolderM3Dt, P, true);
        R.chanq_old = cone_ld(A.ChanQ, P, I.UpTrans != nullptr); // ChanQ before the sub-step: transmission loss
    }
    if (F.linked) R.cut_raw = cone_ld(F.linked, P); // (also without STRUCT: a sub-step at a time on a graph with structure links)
    R.qold = cone_ld(A.ChanQKin, P);
    R.alpha1 = cone_ld(A.ChannelAlpha, P);
    R.sum_old = cone_ld(fused_sum(F, s), P);
    if (SPLIT) {
        R.m3 = cone_ld(A.ChanM3Kin, P);
        R.m3_2 = cone_ld(A.Chan2M3Kin, P);
        R.start = cone_ld(A.Chan2M3Start, P);
        R.m3limit = cone_ld(A.M3Limit, P);
        R.q2start = cone_ld(A.Chan2QStart, P);
        R.q2old = cone_ld(A.Chan2QKin, P);
        R.alpha2 = cone_ld(A.ChannelAlpha2, P);
        R.qlimit = cone_ld(A.QLimit, P);
    }
    if (test_inert) { // the inert test of cone_skip: the flag and the state it compares with +0.0
        R.inert_raw = cone_ld(F.inert, P);
        R.chanq_old = cone_ld(A.ChanQ, P);
        if (!SPLIT) R.m3 = cone_ld(A.ChanM3Kin, P);
        if (SPLIT) {
            R.csa_old = cone_ld(A.CrossSection2Area, P);
            R.sf1_old = cone_ld(A.Sideflow1Chan, P);
        }
    }
    if (fused_last(F, s)) R.pix_area = cone_ld(A.PixelArea, P);
}

// behind the wait for a level's loads: what depends on uniform conditions AND on loaded values
__device__ __forceinline__ void cone_fix(const fused_args &F, cone_cell &R, bool dx_is_len)
{
    if (dx_is_len)
        R.dxp = R.len; // (same bits as the per-cell dx, see derived_flags)
    else if (!F.dx)
        R.dxp = F.dx_scalar;
}

// fused_args::recompute: the five derived statics from the three loaded ones (same operations as the host's)
template <bool SPLIT>
__device__ __forceinline__ void cone_derive(const fused_args &F, cone_cell &R)
{
    if (!R.active) return;
    R.inv_len = 1.0 / R.len;
    R.inv_alpha1 = 1.0 / R.alpha1;
    R.ap1 = R.alpha1 * R.dxp / F.dt;
    if (SPLIT) {
        R.inv_alpha2 = 1.0 / R.alpha2;
        R.ap2 = R.alpha2 * R.dxp / F.dt;
    }
}

// the early return of fused_cell: an inert cell whose state is all +0.0 stays as it is
template <bool SPLIT>
__device__ __forceinline__ bool cone_skip(const cone_cell &R)
{
    if (!R.active) return true;
    if (!R.inert_raw) return false;
    bool zero = plus_zero(R.qold) && plus_zero(R.m3) && plus_zero(R.chanq_old);
    if (SPLIT && zero) zero = plus_zero(R.q2old) && plus_zero(R.m3_2) && plus_zero(R.csa_old) && plus_zero(R.sf1_old);
    return zero;
}

// The general-exponent paths (OCML pow: ~200 instructions per call site, ~25 call sites inlined into fused_cell) are
// taken by single lanes with extreme arguments or not at all when beta = 3/5.  In the cone kernel's beta = 3/5 variant
// they live out of line, one copy each, so that the loop over the levels of a cone is a few KB of straight code instead
// of ~60 KB of mostly skipped blocks.  Same functions, same results.
__device__ __attribute__((noinline)) double cold_pow(double x, double y) { return pow(x, y); }
__device__ __attribute__((noinline)) double cold_solve(double c, double a, double ba, double beta, double inv_beta,
                                                       double b_minus_1)
{
    return lf_solve_cell(c, a, ba, beta, inv_beta, b_minus_1);
}
// ALL35: beta == 3/5 for the fix-up round trips and for the router (both flags of fused_cell set), known on the host;
// otherwise the flags are tested at run time exactly as fused_cell does
template <bool ALL35>
__device__ __forceinline__ double cone_pow_3_5(double x, double y, bool is35)
{
    if (!ALL35) return is35 ? lf_pow_3_5(x) : pow(x, y);
    if (x == 0.0) return 0.0; // as lf_pow_3_5
    if (lf_fast_range(x)) {
        const double r = lf_root5(x);
        return r * r * r;
    }
    return cold_pow(x, 0.6);
}
template <bool ALL35>
__device__ __forceinline__ double cone_pow_5_3(double x, double y, bool is35)
{
    if (!ALL35) return is35 ? lf_pow_5_3(x) : pow(x, y);
    if (x == 0.0) return 0.0; // as lf_pow_5_3
    if (lf_fast_range(x)) {
        const double r = lf_cbrt(x);
        return x * (r * r);
    }
    return cold_pow(x, 1.0 / 0.6);
}
template <bool ALL35>
__device__ __forceinline__ double cone_solve(double c, double ap, bool is35, const fused_args &F)
{
    if (!ALL35) return solve_any(c, ap, is35, F);
    if (lf_fast_range(c) && lf_fast_range(ap)) return (c <= LF_NEWTON_TOL) ? 0.0 : lf_solve_3_5(c, ap);
    return cold_solve(c, ap, F.beta * ap, F.beta, F.inv_beta, F.b_minus_1);
}

// beta = 3/5, BOTH lines (main channel, floodplain) of a cell stage by stage in straight code: the two independent chains
// interleave in the instruction stream, and the lanes beyond the fast range (extreme values, NaN: rare) are redone in ONE
// divergent region per stage instead of a branch around every power and solve (46 branches per level before).  Value by
// value the operations of cone_pow_3_5 / cone_solve / cone_pow_5_3 <true>.  -DLF_CONE_MATH_TWO=0: the one-by-one forms.
#ifndef LF_CONE_MATH_TWO
#define LF_CONE_MATH_TWO 1
#endif
template <bool TWO>
__device__ __forceinline__ void cone_pow_3_5_two(double x1, double x2, double &y1, double &y2)
{
    const bool f1 = lf_fast_range(x1), f2 = TWO && lf_fast_range(x2);
    const double r1 = lf_root5(f1 ? x1 : 1.0), r2 = TWO ? lf_root5(f2 ? x2 : 1.0) : 0.0;
    y1 = f1 ? r1 * r1 * r1 : 0.0; // (+-0 -> +0 as pow does: dry cells are common and stay on this path)
    y2 = f2 ? r2 * r2 * r2 : 0.0;
    const bool c1 = !f1 && x1 != 0.0, c2 = TWO && !f2 && x2 != 0.0;
    if (c1 || c2) {
        if (c1) y1 = cold_pow(x1, 0.6);
        if (c2) y2 = cold_pow(x2, 0.6);
    }
}
template <bool TWO>
__device__ __forceinline__ void cone_pow_5_3_two(double x1, double x2, double &y1, double &y2)
{
    const bool f1 = lf_fast_range(x1), f2 = TWO && lf_fast_range(x2);
    const double r1 = lf_cbrt(f1 ? x1 : 1.0), r2 = TWO ? lf_cbrt(f2 ? x2 : 1.0) : 0.0;
    y1 = f1 ? x1 * (r1 * r1) : 0.0;
    y2 = f2 ? x2 * (r2 * r2) : 0.0;
    const bool c1 = !f1 && x1 != 0.0, c2 = TWO && !f2 && x2 != 0.0;
    if (c1 || c2) {
        if (c1) y1 = cold_pow(x1, 1.0 / 0.6);
        if (c2) y2 = cold_pow(x2, 1.0 / 0.6);
    }
}
template <bool TWO>
__device__ __forceinline__ void cone_solve_two(double c1, double a1, double c2, double a2, const fused_args &F, double &q1,
                                               double &q2)
{
    const bool g1 = lf_fast_range(c1) && lf_fast_range(a1), g2 = TWO && lf_fast_range(c2) && lf_fast_range(a2);
    const double s1 = lf_solve_3_5(g1 ? c1 : 1.0, g1 ? a1 : 1.0);
    const double s2 = TWO ? lf_solve_3_5(g2 ? c2 : 1.0, g2 ? a2 : 1.0) : 0.0;
    q1 = (c1 <= LF_NEWTON_TOL) ? 0.0 : s1;
    q2 = (c2 <= LF_NEWTON_TOL) ? 0.0 : s2;
    if (!g1 || (TWO && !g2)) {
        if (!g1) q1 = cold_solve(c1, a1, F.beta * a1, F.beta, F.inv_beta, F.b_minus_1);
        if (TWO && !g2) q2 = cold_solve(c2, a2, F.beta * a2, F.beta, F.inv_beta, F.b_minus_1);
    }
}

// everything after the loads of fused_cell<SPLIT, false>, in its order, up to the stores: the new state stays in
// registers (cone_out) and is stored one level later, so that the stores have a whole level's arithmetic to drain
struct cone_out {
    double v, q, chanq, sum, s1, v2, csa, q2, vel, trav;
    double qin, qin_added, loss, trans_cum, side_m3; // STRUCT
    long long p;
    bool valid;
};

template <bool SPLIT, bool ALL35, bool STRUCT>
__device__ __forceinline__ void cone_compute(const fused_args &F, const cone_cell &R, long long p, int s, double ups1,
                                             double ups2, double &qr_out, double &q2r_out, cone_out &O)
{
    const lf_substep_args &A = F.S;
    const bool b35 = A.Beta == 0.6;
    const bool s35 = F.solve35 != 0;
    const bool last = fused_last(F, s);
    double side_m3 = R.side_m3;
    O.qin = O.qin_added = O.loss = O.trans_cum = 0.0;
    if (STRUCT) { // inflow.py:142-144, transmission.py:76-87, sideflow assembly routing.py:462-478 -- as fused_cell
        const lf_inloop_args &I = F.I;
        if (I.EvaAddM3Dt) side_m3 -= R.eva;
        if (I.WUseAddM3Dt) side_m3 -= R.wuse;
        if (I.QInM3Old) {
            O.qin = (R.qin_old + (s + 1) * R.qdelta) * I.InvNoRoutSteps;
            O.qin_added = (s < 1 ? 0.0 : R.qin_added_old) + O.qin;
            side_m3 += O.qin;
        }
        if (I.UpTrans) {
            const double qc = R.chanq_old;
            double tout = qc;
            if (R.uptrans_raw) {
                const double inner = lf_pow_scalar_exponent(qc, I.TransPower2) - I.TransSub;
                tout = lf_pow_scalar_exponent(inner, I.TransPower1);
            }
            O.loss = (qc - tout) * I.DtRouting;
            O.trans_cum = R.transcum + O.loss;
            side_m3 -= O.loss;
        }
        if (I.QLakeOutM3Dt) side_m3 += R.lakeout;
        if (I.QResOutM3Dt) side_m3 += R.resout;
        if (I.ChannelToPolderM3Dt) side_m3 -= R.polder;
    }
    O.side_m3 = side_m3;
    const double side = (R.chan_raw != 0) ? side_m3 * R.inv_len * A.InvDtRouting : 0.0;
    double s1 = side, s2 = 0.0;
    if (!SPLIT) {
        if (isnan(side)) s1 = 0.0;
    } else {
        const double tot = R.m3 + R.m3_2;
        const double ratio = (tot > 0) ? R.m3 / tot : 0.0;
        s1 = ((tot - R.start) > R.m3limit) ? ratio * side : side;
        if (fabs(side) < 1e-7) s1 = side;
        s2 = (side - s1) + R.q2start * R.inv_len;
    }
#ifdef LF_EXP_NO_MATH /* timing experiment only (wrong results): the loads, stores and exchanges without the solves and powers */
    const double cst = R.ap1 * R.qold + s1 * R.dxp;
    const double c = ups1 + cst;
    const double qr = c * 0.25;
    double v = R.len * R.alpha1 * qr;
    if (v < 0.0) v = 0.0;
    const double x = v * R.inv_len * R.inv_alpha1;
    const double q = x;
    double chanq = q, q2r = 0, v2 = 0, q2 = 0;
    if (SPLIT) {
        const double cst2 = R.ap2 * R.q2old + s2 * R.dxp;
        const double c2 = ups2 + cst2;
        q2r = c2 * 0.25;
        v2 = R.len * R.alpha2 * q2r;
        if ((v2 - R.start) < 0.0) v2 = R.start;
        const double x2 = v2 * R.inv_len * R.inv_alpha2;
        q2 = x2;
        chanq = q + q2 - R.qlimit;
        if (chanq < 0.0) chanq = 0.0;
    }
#else
    double qr, q, v, chanq, q2r = 0, v2 = 0, q2 = 0;
    if constexpr (ALL35 && LF_CONE_MATH_TWO) {
        // beta = 3/5: both lines stage by stage in straight code (cone_*_two): the same operations per value as below
        double pw1, pw2, pr1, pr2;
        cone_pow_3_5_two<SPLIT>(R.qold, R.q2old, pw1, pw2);
        const double c = ups1 + (R.ap1 * pw1 + s1 * R.dxp);
        const double c2 = SPLIT ? ups2 + (R.ap2 * pw2 + s2 * R.dxp) : 0.0;
        cone_solve_two<SPLIT>(c, R.ap1, c2, R.ap2, F, qr, q2r);
        cone_pow_3_5_two<SPLIT>(qr, q2r, pr1, pr2);
        v = R.len * R.alpha1 * pr1;
        if (v < 0.0) v = 0.0;
        const double x = v * R.inv_len * R.inv_alpha1;
        double x2 = 0.0;
        if (SPLIT) {
            v2 = R.len * R.alpha2 * pr2;
            if ((v2 - R.start) < 0.0) v2 = R.start;
            x2 = v2 * R.inv_len * R.inv_alpha2;
        }
        cone_pow_5_3_two<SPLIT>(x, x2, q, q2);
        chanq = q;
        if (SPLIT) {
            chanq = q + q2 - R.qlimit;
            if (chanq < 0.0) chanq = 0.0;
        }
    } else {
        const double cst = R.ap1 * cone_pow_3_5<ALL35>(R.qold, F.beta, s35) + s1 * R.dxp;
        const double c = ups1 + cst;
        qr = cone_solve<ALL35>(c, R.ap1, s35, F);
        v = R.len * R.alpha1 * cone_pow_3_5<ALL35>(qr, A.Beta, b35);
        if (v < 0.0) v = 0.0;
        const double x = v * R.inv_len * R.inv_alpha1;
        q = cone_pow_5_3<ALL35>(x, A.InvBeta, b35);
        chanq = q;
        if (SPLIT) {
            const double cst2 = R.ap2 * cone_pow_3_5<ALL35>(R.q2old, F.beta, s35) + s2 * R.dxp;
            const double c2 = ups2 + cst2;
            q2r = cone_solve<ALL35>(c2, R.ap2, s35, F);
            v2 = R.len * R.alpha2 * cone_pow_3_5<ALL35>(q2r, A.Beta, b35);
            if ((v2 - R.start) < 0.0) v2 = R.start;
            const double x2 = v2 * R.inv_len * R.inv_alpha2;
            q2 = cone_pow_5_3<ALL35>(x2, A.InvBeta, b35);
            chanq = q + q2 - R.qlimit;
            if (chanq < 0.0) chanq = 0.0;
        }
    }
#endif
    qr_out = qr;
    q2r_out = q2r;
    O.valid = true;
    O.p = p;
    O.v = v;
    O.q = q;
    O.chanq = chanq;
    O.sum = R.sum_old + chanq;
    O.s1 = s1;
    O.v2 = v2;
    O.csa = (v2 - R.start) * R.inv_len;
    O.q2 = q2;
    O.vel = O.trav = 0.0;
    if (last) { // routing.py:693-703
        double area = v * R.inv_len;
        if (area < 0.01) area = 0.01;
        const double v1 = q / area, vv2 = 0.36 * (ALL35 ? cold_pow(q, 0.24) : pow(q, 0.24));
        double vel = (vv2 < v1) ? vv2 : v1;
        if (isnan(vv2)) vel = vv2;
        double sinu = sqrt(R.pix_area) * R.inv_len;
        if (sinu > 1) sinu = 1;
        vel *= sinu;
        O.vel = vel;
        O.trav = vel * A.DtSec;
    }
}

// ---- time-major form: ALL sub-steps of one level in one launch, the state in registers ------------------------------
// (level k, sub-step s) needs (level k - 1, sub-step s) and (level k, sub-step s - 1): the skewed wavefront above runs
// the pairs with k + s = t together and so streams a cell's ~25 state vectors through HBM once per sub-step (~200 B,
// 24 times per model step).  The other order of the same loop nest -- level after level, each level through all its
// sub-steps -- keeps a cell's state in registers from its first sub-step to its last: the state is read once and written
// once per MODEL step, and what crosses between levels is the router output of every sub-step (hist1 / hist2:
// [nsteps][N], 16 B written and ~16 B read per cell and sub-step).  ~1 kB per cell and model step instead of ~4.8 kB --
// and NL launches instead of NL + nsteps - 1 --, at the price of a dependent chain of nsteps solves per launch: the form
// for graphs of FEW, WIDE levels (the channel network of the hot-path scenario: 27 levels of 280 000 cells; `shallow`
// rasters), not for the deep ones, whose thousands of narrow levels are what the skew and the cones exist for.
// Arithmetic of a (cell, sub-step) is fused_cell's, operation by operation: bit-identical results.
// ALL35: beta == 3/5 for the fix-up round trips and for the router, known on the host -- both lines stage by stage in
// straight code (cone_*_two), the general solve and OCML pow out of line
#ifndef LF_TM_WAVES
#define LF_TM_WAVES 4
#endif
#if LF_TM_WAVES > 0
#define LF_TM_ATTR __attribute__((amdgpu_waves_per_eu(LF_TM_WAVES)))
#else
#define LF_TM_ATTR
#endif
// DIST (row-block partition, a phase's levels): upstream cells are a consecutive local run (hist), or come from the list --
// same-phase positions (hist) and slab slots (earlier phases, other ranks: the slabs hold every sub-step) --, and a cell
// whose router outputs cross a phase or rank boundary also writes them to its slab slot, as k_fused_substeps_dist does.
template <bool SPLIT, bool ALL35, bool DIST = false>
__global__ void __launch_bounds__(kBlock) LF_TM_ATTR k_fused_level_steps(fused_args F, int k)
{
    const lf_substep_args &A = F.S;
    const long long first = F.level_start[k];
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= F.level_start[k + 1] - first) return;
    const long long p = first + i, n = F.n;
    const int nsteps = F.nsteps;
    const int u0 = F.ups_ptr[p], u1 = F.ups_ptr[p + 1], kmax = F.kmax;
    int base = u0;
    long long slot = -1;
    if (DIST) {
        const int base_raw = F.d_ups_base[p]; // (a run of ghost slots is not a local run: see k_fused_substeps_dist)
        base = (base_raw >= 0 && (long long)base_raw + (u1 - u0) <= n) ? base_raw : -1;
        slot = F.d_out_slot[p];
    }
    auto ups_of = [&](const double *hist, const double *slab, int s) {
        if (!DIST) return upstream_sum8_pairs(hist + (long long)s * n, u0, u1, kmax);
        if (base >= 0) return upstream_sum8(hist + (long long)s * n, base, base + (u1 - u0), kmax);
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double x = 0.0;
            if (j < kmax && u0 + j < u1) {
                const int e = F.d_ups_idx[u0 + j];
                x = e >= 0 ? hist[(long long)s * n + e] : slab[(long long)(-(e + 1)) * F.root_ss + (long long)s * F.root_st];
            }
            v[j] = x;
        }
        double ups = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) ups += v[j];
        return ups;
    };
    const bool b35 = A.Beta == 0.6, s35 = F.solve35 != 0;
    const unsigned int dflags = derived_flags(F);
    const bool rc = (dflags & 1u) != 0u;
    const double len = A.ChanLength[p];
    const double dxp = (dflags & 2u) ? len : (F.dx ? F.dx[p] : F.dx_scalar);
    const double inv_len = rc ? 1.0 / len : A.InvChanLength[p];
    const bool is_chan = A.IsChannelKinematic[p] != 0;
    const bool cut = F.linked && F.linked[p];
    const double alpha1 = A.ChannelAlpha[p];
    const double inv_alpha1 = rc ? 1.0 / alpha1 : A.InvChannelAlpha[p], ap1 = rc ? alpha1 * dxp / F.dt : F.a1[p];
    double qold = A.ChanQKin[p], sum = 0.0;
    double m3 = 0, m3_2 = 0, start = 0, m3limit = 0, q2start = 0, ap2 = 0, q2old = 0, alpha2 = 0, inv_alpha2 = 0, qlimit = 0;
    if (SPLIT) {
        m3 = A.ChanM3Kin[p];
        m3_2 = A.Chan2M3Kin[p];
        start = A.Chan2M3Start[p];
        m3limit = A.M3Limit[p];
        q2start = A.Chan2QStart[p];
        q2old = A.Chan2QKin[p];
        alpha2 = A.ChannelAlpha2[p];
        ap2 = rc ? alpha2 * dxp / F.dt : F.a2[p];
        inv_alpha2 = rc ? 1.0 / alpha2 : A.InvChannelAlpha2[p];
        qlimit = A.QLimit[p];
    }
    const double pix_area = A.PixelArea[p];
    int s0 = 0;
    if (F.inert && F.inert[p]) { // see k_inert_flags: a sub-step leaves such a cell as it is while its state is all +0.0,
                                 // so only the last one (which also writes the velocities) is run, as fused_cell does
        bool zero = plus_zero(qold) && plus_zero(A.ChanM3Kin[p]) && plus_zero(A.ChanQ[p]);
        if (SPLIT && zero)
            zero = plus_zero(q2old) && plus_zero(m3_2) && plus_zero(A.CrossSection2Area[p]) && plus_zero(A.Sideflow1Chan[p]);
        if (zero) {
            s0 = nsteps - 1;
            // several model steps in the call: fused_cell runs the last sub-step of EVERY model step on such a cell and
            // stores sum + ChanQ with ChanQ = +0.0 -- the same store here, so that a -0.0 the caller left in the sums of
            // the earlier model steps ends up as +0.0 in both forms (any other value is unchanged by the addition)
            for (int m = F.msteps - 1; m < nsteps - 1; m += F.msteps) fused_sum(F, m)[p] = fused_sum(F, m)[p] + 0.0;
        }
    }
    double v = 0, q = 0, chanq = 0, s1 = 0, v2 = 0, q2 = 0;
    double side_m3 = fused_side(F, s0)[p];
    sum = fused_sum(F, s0)[p];
    // position of the sub-step inside its model step, counted along (s starts per lane -- inert cells run only the last
    // sub-step -- so `s % msteps` would be a vector-register division, ~25 instructions, twice per sub-step)
    int sm = s0 % F.msteps;
    for (int s = s0; s < nsteps; ++s) {
        double ups1, ups2;
        if (!DIST) {
            upstream_sum8_pairs2(F.hist1 + (long long)s * n, F.hist2 + (long long)s * n, SPLIT, u0, u1, kmax, ups1, ups2);
        } else {
            ups1 = ups_of(F.hist1, F.root1, s);
            ups2 = SPLIT ? ups_of(F.hist2, F.root2, s) : 0.0;
        }
        const bool first_of_step = sm == 0;
        if ((F.side_stride != 0 || first_of_step) && s > s0) side_m3 = fused_side(F, s)[p];
        if (first_of_step && s > s0) sum = fused_sum(F, s)[p]; // the next model step's sum (zeroed by the caller)
        // ---- sideflow (routing.py:512, 524 / 549-567) ----
        const double side = is_chan ? side_m3 * inv_len * A.InvDtRouting : 0.0;
        double s2 = 0.0;
        s1 = side;
        if (!SPLIT) {
            if (isnan(side)) s1 = 0.0;
        } else {
            const double tot = m3 + m3_2;
            const double ratio = (tot > 0) ? m3 / tot : 0.0;
            s1 = ((tot - start) > m3limit) ? ratio * side : side;
            if (fabs(side) < 1e-7) s1 = side;
            s2 = (side - s1) + q2start * inv_len;
        }
        // ---- main channel: router call + fix-up (routing.py:526-532 / 573-578); floodplains (routing.py:583-603) ----
        double qr, q2r = 0;
        if constexpr (ALL35) {
            double pw1, pw2, pr1, pr2;
            cone_pow_3_5_two<SPLIT>(qold, q2old, pw1, pw2);
            const double c = ups1 + (ap1 * pw1 + s1 * dxp);
            const double c2 = SPLIT ? ups2 + (ap2 * pw2 + s2 * dxp) : 0.0;
            cone_solve_two<SPLIT>(c, ap1, c2, ap2, F, qr, q2r);
            cone_pow_3_5_two<SPLIT>(qr, q2r, pr1, pr2);
            v = len * alpha1 * pr1;
            if (v < 0.0) v = 0.0;
            const double x = v * inv_len * inv_alpha1;
            double x2 = 0.0;
            if (SPLIT) {
                v2 = len * alpha2 * pr2;
                if ((v2 - start) < 0.0) v2 = start;
                x2 = v2 * inv_len * inv_alpha2;
            }
            cone_pow_5_3_two<SPLIT>(x, x2, q, q2);
            chanq = q;
            if (SPLIT) {
                chanq = q + q2 - qlimit;
                if (chanq < 0.0) chanq = 0.0;
            }
        } else {
            const double cst = ap1 * (s35 ? lf_pow_3_5(qold) : pow(qold, F.beta)) + s1 * dxp;
            const double c = ups1 + cst;
            qr = solve_any(c, ap1, s35, F);
            v = len * alpha1 * (b35 ? lf_pow_3_5(qr) : pow(qr, A.Beta));
            if (v < 0.0) v = 0.0;
            const double x = v * inv_len * inv_alpha1;
            q = b35 ? lf_pow_5_3(x) : pow(x, A.InvBeta);
            chanq = q;
            if (SPLIT) {
                const double cst2 = ap2 * (s35 ? lf_pow_3_5(q2old) : pow(q2old, F.beta)) + s2 * dxp;
                const double c2 = ups2 + cst2;
                q2r = solve_any(c2, ap2, s35, F);
                v2 = len * alpha2 * (b35 ? lf_pow_3_5(q2r) : pow(q2r, A.Beta));
                if ((v2 - start) < 0.0) v2 = start;
                const double x2 = v2 * inv_len * inv_alpha2;
                q2 = b35 ? lf_pow_5_3(x2) : pow(x2, A.InvBeta);
                chanq = q + q2 - qlimit;
                if (chanq < 0.0) chanq = 0.0;
            }
        }
        F.hist1[(long long)s * n + p] = cut ? 0.0 : qr;
        if (SPLIT) F.hist2[(long long)s * n + p] = cut ? 0.0 : q2r;
        if (DIST && slot >= 0) { // kept for the next phase / rank, as fused_cell does
            F.root1[slot * F.root_ss + (long long)s * F.root_st] = qr;
            if (SPLIT) F.root2[slot * F.root_ss + (long long)s * F.root_st] = q2r;
        }
        // the state of the next sub-step
        m3 = v;
        qold = q;
        sum = sum + chanq;
        const bool last_of_step = sm + 1 == F.msteps; // fused_last(F, s)
        if (last_of_step && s != nsteps - 1) fused_sum(F, s)[p] = sum; // a model step inside the call is complete
        sm = last_of_step ? 0 : sm + 1;
        if (SPLIT) {
            m3_2 = v2;
            q2old = q2;
        }
    }
    // ---- what the sub-step-by-sub-step sequence leaves behind ----
    A.ChanM3Kin[p] = v;
    A.ChanQKin[p] = q;
    A.ChanQ[p] = chanq;
    fused_sum(F, nsteps - 1)[p] = sum;
    if (SPLIT) {
        A.Sideflow1Chan[p] = s1;
        A.Chan2M3Kin[p] = v2;
        A.CrossSection2Area[p] = (v2 - start) * inv_len;
        A.Chan2QKin[p] = q2;
    }
    { // routing.py:693-703
        double area = v * inv_len;
        if (area < 0.01) area = 0.01;
        const double v1 = q / area, vv2 = 0.36 * pow(q, 0.24);
        double vel = (vv2 < v1) ? vv2 : v1;
        if (isnan(vv2)) vel = vv2;
        double sinu = sqrt(pix_area) * inv_len;
        if (sinu > 1) sinu = 1;
        vel *= sinu;
        A.FlowVelocity[p] = vel;
        A.TravelDistance[p] = vel * A.DtSec;
    }
}


// the state a level leaves behind (cells without a result -- beyond the cone, skipped -- store nothing)
template <bool SPLIT, bool STRUCT>
__device__ __forceinline__ void cone_store(const fused_args &F, const cone_out &O, int s)
{
    if (!O.valid) return;
    const lf_substep_args &A = F.S;
    const unsigned o8 = (unsigned)O.p * 8u;
    if (STRUCT) {
        const lf_inloop_args &I = F.I;
        if (I.QInM3Old) {
            cone_st(I.QInDt, o8, O.qin);
            cone_st(I.QinADDEDM3, o8, O.qin_added);
        }
        if (I.UpTrans) {
            cone_st(I.TransLossM3Dt, o8, O.loss);
            cone_st(I.TransCum, o8, O.trans_cum);
        }
        cone_st(I.SideflowChanM3, o8, O.side_m3);
    }
    const bool last = fused_last(F, s);
    cone_st(A.ChanM3Kin, o8, O.v);
    cone_st(A.ChanQKin, o8, O.q);
    cone_st(fused_sum(F, s), o8, O.sum);
    if (SPLIT) {
        cone_st(A.Chan2M3Kin, o8, O.v2);
        cone_st(A.Chan2QKin, o8, O.q2);
    }
    if (STRUCT || last) { // see fused_cell: intermediate values nobody reads are not stored
        cone_st(A.ChanQ, o8, O.chanq);
        if (SPLIT) {
            cone_st(A.Sideflow1Chan, o8, O.s1);
            cone_st(A.CrossSection2Area, o8, O.csa);
        }
    }
    if (last) {
        cone_st(A.FlowVelocity, o8, O.vel);
        cone_st(A.TravelDistance, o8, O.trav);
    }
}

// The cones of a (block, sub-step) in XCD-contiguous order (lf_blocks.h: lf_xcd_contiguous).  -DLF_XCD_REMAP=0: launch order
// (A/B builds).
#ifndef LF_XCD_REMAP
#define LF_XCD_REMAP 1
#endif
__device__ __forceinline__ int xcd_contiguous(int i, int n, unsigned linear_id)
{
    return LF_XCD_REMAP ? lf_xcd_contiguous(i, n, linear_id) : i;
}

// LDS barrier: the wavefronts of the workgroup have finished their LDS writes; global stores keep draining
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

#ifndef LF_CONES_WAVES
#define LF_CONES_WAVES 2
#endif
// CW: cells per level of a cone = threads of the workgroup (64: one wavefront per cone, no barrier between the levels)
template <bool SPLIT, bool ALL35, bool STRUCT, bool DIST = false, int CW = kBlock>
__global__ void __launch_bounds__(CW) __attribute__((amdgpu_waves_per_eu(LF_CONES_WAVES))) k_fused_cones(fused_args F)
{
    // (slot CW of every row holds 0.0: what an absent upstream cell reads -- no select behind the LDS read)
    __shared__ double x1[2][CW + 1], x2[SPLIT ? 2 : 1][SPLIT ? CW + 1 : 1];
    if (threadIdx.x < 2) {
        x1[threadIdx.x][CW] = 0.0;
        if (SPLIT) x2[threadIdx.x][CW] = 0.0;
    }
    int s, blk;
    if (F.packed) {
        int cnt = 0, start = 0;
        for (int q = 0; q < F.nsteps; ++q) {
            const bool ge = (int)blockIdx.x >= F.blk_start[q];
            cnt += ge ? 1 : 0;
            start = ge ? F.blk_start[q] : start;
        }
        s = cnt - 1;
        blk = (int)blockIdx.x - start;
    } else {
        s = blockIdx.y;
        blk = blockIdx.x;
    }
    const int bi = F.t - s;
    if (bi < 0 || bi >= F.fb_nblocks) return;
    const int b = F.fb_block0 + bi;
    // the plan is read through the constant address space: scalar loads, no vector-memory wait on the way (ld_table)
    const int row0 = ld_table(F.fb_row, b), ncones = ld_table(F.fb_row, b + 1) - row0 - 1;
    if (blk >= ncones) return;
    blk = xcd_contiguous(blk, ncones, blockIdx.x + blockIdx.y * gridDim.x);
    const int nl = ld_table(F.fb_level, b + 1) - ld_table(F.fb_level, b);
    const int *c0 = F.fb_cone + (size_t)ld_table(F.fb_off, b) + (size_t)blk * nl, *c1 = c0 + nl; // this cone / the next
    const int kmax = F.kmax;
    const long long par = (long long)(s & 1) * F.n;
    const int tid = threadIdx.x;
    const unsigned int dflags = derived_flags(F);
    const bool rc = (dflags & 1u) != 0u, dx_is_len = (dflags & 2u) != 0u;
    cone_out pend;
    pend.valid = false;
    pend.p = 0;
    int first_up = 0; // first position of the level above (LDS index 0)
    // One level of the cone: `cur` holds the loaded state of this thread's cell of level j, `nxt` receives that of level
    // j + 1.  The loop below calls it with the two register sets swapping roles (no copies).
    // The ~45 array pointers of fused_args do not fit the scalar registers next to the loop's other state: left to itself
    // the compiler loads them once and spills them to VGPR lanes (200 v_readlane per level).  Instead every level reads
    // what it needs from the kernel-argument segment again (scalar loads from the constant cache, a handful per level):
    // the pointer is laundered through an empty asm so that nothing is hoisted out of the level.
    typedef const fused_args __attribute__((address_space(4))) *kargs_t;
    const kargs_t K0 = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    auto level = [&](int j, const cone_cell &cur, cone_cell &nxt, int first) {
        kargs_t Kp = K0;
        asm volatile("" : "+s"(Kp));
        const fused_args &F = *(const fused_args *)Kp;
        const long long p = first + tid;
        if (j > 0) { // level j-1 of this cone is in LDS
            if (CW > 64)
                lds_barrier();
            else
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // one wavefront: its LDS operations complete in order
        }
        // Right behind the barrier: the state stores of level j-1 and the state loads of level j+1.  Both have the
        // arithmetic of level j to complete, so the wait at the end of this level finds them done (issued after the
        // arithmetic, the stores' acknowledgement would be waited for on every level).
        cone_store<SPLIT, STRUCT>(F, pend, s);
        pend.valid = false;
        // nothing of the next level's state depends on this launch (behind the last level: an empty range, nothing loaded)
        const int jn = j + 1 < nl ? j + 1 : j;
        const int nfirst = ld_table(c0, jn);
        const cone_range nrng = {nfirst, j + 1 < nl ? ld_table(c1, jn) - nfirst : 0};
        cone_load<SPLIT, STRUCT, DIST>(F, nrng, tid, s, nxt, rc, dx_is_len);
        if (!cone_skip<SPLIT>(cur)) {
            double ups1, ups2 = 0.0;
            // row-block partition: the upstream cells are the consecutive local run [base, base + count) -- a run of ghost
            // slots is not one, see k_fused_substeps_dist -- or come from the list: same-phase cells (level before: LDS,
            // or the parity buffers for the block's first level) and slab slots (earlier phases, other ranks)
            const int cu0 = DIST ? cur.base : cur.u0, cu1 = DIST ? cur.base + (cur.u1 - cur.u0) : cur.u1;
            if (DIST && (cur.base < 0 || (long long)cu1 > F.n)) {
                const double *y1 = &x1[(j - 1) & 1][0], *y2 = &x2[SPLIT ? (j - 1) & 1 : 0][0];
                const long long soff = (long long)s * F.root_st;
                double v1[8], v2[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    double a1 = 0.0, a2 = 0.0;
                    if (k < kmax && cur.u0 + k < cur.u1) {
                        const int e = F.d_ups_idx[cur.u0 + k];
                        if (e < 0) {
                            const long long at = (long long)(-(e + 1)) * F.root_ss + soff;
                            a1 = F.root1[at];
                            if (SPLIT) a2 = F.root2[at];
                        } else if (j == 0) {
                            a1 = F.qr1[par + e];
                            if (SPLIT) a2 = F.qr2[par + e];
                        } else {
                            a1 = y1[e - first_up];
                            if (SPLIT) a2 = y2[e - first_up];
                        }
                    }
                    v1[k] = a1;
                    v2[k] = a2;
                }
                ups1 = 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) ups1 += v1[k];
                if (SPLIT) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) ups2 += v2[k];
                }
            } else if (j == 0) { // from the block before (previous launch) through the parity buffers
                if (DIST) {
                    ups1 = upstream_sum8(F.qr1 + par, cu0, cu1, kmax);
                    if (SPLIT) ups2 = upstream_sum8(F.qr2 + par, cu0, cu1, kmax);
                } else {
                    upstream_sum8_pairs2(F.qr1 + par, F.qr2 + par, SPLIT, cu0, cu1, kmax, ups1, ups2);
                }
            } else { // from LDS, branch-free: absent neighbours read the slot that holds 0.0 (the sum as upstream_sum8)
                const double *y1 = &x1[(j - 1) & 1][0], *y2 = &x2[SPLIT ? (j - 1) & 1 : 0][0];
                const int base = cu0 - first_up;
                double v1[8], v2[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool have = k < kmax && cu0 + k < cu1;
                    const int idx = have ? base + k : CW;
                    v1[k] = y1[idx];
                    if (SPLIT) v2[k] = y2[idx];
                }
                ups1 = 0.0;
#pragma unroll
                for (int k = 0; k < 8; ++k) ups1 += v1[k];
                if (SPLIT) {
#pragma unroll
                    for (int k = 0; k < 8; ++k) ups2 += v2[k];
                }
            }
            double qr, q2r;
            cone_compute<SPLIT, ALL35, STRUCT>(F, cur, p, s, ups1, ups2, qr, q2r, pend);
            if (DIST && cur.slot >= 0) { // feeds a later phase or another rank: kept for every sub-step
                const long long at = (long long)cur.slot * F.root_ss + (long long)s * F.root_st;
                F.root1[at] = qr;
                if (SPLIT) F.root2[at] = q2r;
            }
            if (cur.cut_raw) qr = q2r = 0.0; // zero-length structure links: their router output is stored as 0
            if (j + 1 < nl) {
                x1[j & 1][tid] = qr;
                if (SPLIT) x2[j & 1][tid] = q2r;
            } else { // the block's last level: read by the next block in the next launch
                F.qr1[par + p] = qr;
                if (SPLIT) F.qr2[par + p] = q2r;
            }
        }
        // the loads of the next level (issued before this level's arithmetic) and the stores of the previous one: done
        // by now.  Stated explicitly so that the compiler does not wait for `nxt` behind the NEXT level's stores.
        __builtin_amdgcn_s_waitcnt(0x0F70);
        cone_fix(F, nxt, dx_is_len);
        if (rc) cone_derive<SPLIT>(F, nxt); // off the next level's chain: before its barrier
        first_up = first;
        return nfirst;
    };
    cone_cell ra = {}, rb = {}; // (zeros: what cone_load leaves alone stays 0 -- the inert and link flags above all)
    int first = ld_table(c0, 0);
    {
        const cone_range rng0 = {first, ld_table(c1, 0) - first};
        cone_load<SPLIT, STRUCT, DIST>(F, rng0, tid, s, ra, rc, dx_is_len);
    }
    __builtin_amdgcn_s_waitcnt(0x0F70); // vmcnt(0): inside the loop the current level's registers are always complete
    cone_fix(F, ra, dx_is_len);
    if (rc) cone_derive<SPLIT>(F, ra);
    for (int j = 0; j < nl; j += 2) {
        first = level(j, ra, rb, first);
        if (j + 1 < nl) first = level(j + 1, rb, ra, first);
    }
    cone_store<SPLIT, STRUCT>(F, pend, s);
}

// element of an array at a 32-bit BYTE offset: array pointer in scalar registers + one vector register (see cone_ld)
template <class T>
__device__ __forceinline__ T split_ld(const T *base, unsigned byte_offset) { return *(const T *)((const char *)base + byte_offset); }

// ---- the cone kernel of a model step with the work split between a chain wavefront and supply wavefronts (round 4) ----
// k_fused_cones makes ONE wavefront (or four, with a barrier per level) do everything a (cell, sub-step) needs: ~25 state
// loads, the sideflow split, two old-discharge terms, two closure solves, the Q -> V -> Q fix-ups, nine stores -- ~650
// instructions of mostly dependent fp64 arithmetic per level, during which the level below waits.  With ~2 wavefronts
// per SIMD (5000 cells per level x 24 sub-steps is all the parallelism a `deep` 5000^2 raster has) nothing hides those
// chains: VALU busy 29 % (profiles/pmc_r03a_fused_deep_5000.txt).  Only gather -> sum -> solve feeds the level below.
// So, as in k_sweep_cones_split (lf_sweep.h), per cone of at most 64 cells per level:
//   * the CHAIN wavefront runs the two routers of the cell (main channel, floodplains) from LDS records: gather the level
//     above, add, solve, put the router outputs back into LDS -- nothing else (cone_chain<true, NR = 2>);
//   * KC = 2 SUPPLY wavefronts, one per level of a chunk, work two chunks ahead and two chunks behind it:
//       phase ph:   request the state of chunk ph + 1                          (pre-loads)
//                   request what the fix-ups of chunk ph - 1 need again          (post-loads: 7 streams, L2 hits mostly)
//                   fix-ups and stores of chunk ph - 2 from its router outputs  (routing.py:526-532, 573-603)
//                   sideflow split, constant terms, gather addresses of chunk ph (routing.py:512-567) -> LDS
//     the two meet at one workgroup barrier per chunk.  Loads and stores sit behind no branch (lanes beyond the cone's
//     range load cell 0 and store beyond the end of a buffer resource; the outputs only the last sub-step keeps are
//     stored beyond the end on the others), so the compiler's wait counts stay exact.
// Arithmetic per cell = fused_cell's, operation by operation (cone_compute's beta = 3/5 forms): bit-identical.
// For: beta = 3/5 router and fix-ups, no inert-pixel test (a compact channel domain has none), the single-domain plan with
// cones of <= 64 cells; with structures in the loop (STRUCT: sideflow assembly in the supply wavefronts, sites between the
// launches as before) or without, not on a graph with links without them; every other case keeps k_fused_cones.
constexpr int kFusedKC = 2; // (chunks of 2 levels: 31 KB of LDS and three wavefronts per cone; 4: 62 KB and five -- slower at every size measured)

template <bool SPLIT, bool STRUCT = false, int KC = kFusedKC>
__global__ void __launch_bounds__(64 * (1 + KC)) k_fused_cones_split(fused_args F)
{
    constexpr int NR = SPLIT ? 2 : 1;
    __shared__ cone_lds<NR, KC> S;
    __shared__ double s1buf[3][KC][64]; // Sideflow1Chan of a cell from its constant terms to its stores (two chunks later)
    int s, blk;
    {
        int cnt = 0, start = 0;
        for (int q = 0; q < F.nsteps; ++q) {
            const bool ge = (int)blockIdx.x >= F.blk_start[q];
            cnt += ge ? 1 : 0;
            start = ge ? F.blk_start[q] : start;
        }
        s = cnt - 1;
        blk = (int)blockIdx.x - start;
    }
    const int bi = F.t - s;
    if (bi < 0 || bi >= F.fb_nblocks) return;
    const int b = F.fb_block0 + bi;
    const int row0 = ld_table(F.fb_row, b), ncones = ld_table(F.fb_row, b + 1) - row0 - 1;
    if (blk >= ncones) return;
    const int nl = ld_table(F.fb_level, b + 1) - ld_table(F.fb_level, b);
    const int *c0 = F.fb_cone + (size_t)ld_table(F.fb_off, b) + (size_t)blk * nl, *c1 = c0 + nl; // this cone / the next
    const int tid = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // 0 = chain, 1 .. KC = supply
    const int nch = (nl + KC - 1) / KC;
    if (wave == 0) {
        if (tid < 2 * KC * NR) S.xr[tid][64] = 0.0; // the zero slot of every row
        sweep_args_multi M;                          // (the chain's cold path: a outside the fast range, c beyond 1e30)
        for (int r = 0; r < kMaxMulti; ++r) {
            M.r[r].a = (r == 1 && SPLIT) ? F.a2 : F.a1;
            M.r[r].beta = F.beta;
            M.r[r].inv_beta = F.inv_beta;
            M.r[r].b_minus_1 = F.b_minus_1;
        }
        lds_barrier(); // chunk 0 is in LDS
        switch (F.kmax) {
        case 0:
        case 1: cone_chain<true, NR, KC, 1>(S, tid, nl, c0, M); break;
        case 2: cone_chain<true, NR, KC, 2>(S, tid, nl, c0, M); break;
        case 3: cone_chain<true, NR, KC, 3>(S, tid, nl, c0, M); break;
        case 4: cone_chain<true, NR, KC, 4>(S, tid, nl, c0, M); break;
        case 5: cone_chain<true, NR, KC, 5>(S, tid, nl, c0, M); break;
        case 6: cone_chain<true, NR, KC, 6>(S, tid, nl, c0, M); break;
        case 7: cone_chain<true, NR, KC, 7>(S, tid, nl, c0, M); break;
        default: cone_chain<true, NR, KC, 8>(S, tid, nl, c0, M); break;
        }
        return;
    }
    // ---- a supply wavefront: level sw of every chunk ----
    const int sw = wave - 1;
    const unsigned int dflags = derived_flags(F);
    const bool dx_is_len = (dflags & 2u) != 0u;
    const bool last = fused_last(F, s);
    const long long par = (long long)(s & 1) * F.n;
    const unsigned nbytes = (unsigned)F.n * 8u;
    typedef int v2i __attribute__((ext_vector_type(2)));
    typedef const fused_args __attribute__((address_space(4))) *kargs_t;
    const kargs_t K0 = (kargs_t)__builtin_amdgcn_kernarg_segment_ptr();
    auto lbound = [&](const int *t, int k) { // entry k of a cone's row of the plan, 0 outside the block
        const bool in = (unsigned)k < (unsigned)nl;
        const int v = ld_table(t, in ? k : 0);
        return in ? v : 0;
    };
    struct pre_t { // what the constant terms of a cell need, as loaded
        int u0, u1, p;
        double len, side_m3, qold, alpha1, m3, m3_2, start, m3limit, q2start, q2old, alpha2, inv_len, ap1, ap2, dxv;
        // STRUCT: the terms of the sideflow assembly (routing.py:462-478), as cone_load reads them
        double eva, wuse, qin_old, qdelta, qin_added_old, chanq_old, transcum, lakeout, resout, polder;
        unsigned char chan, uptrans;
        bool act;
    };
    struct post_t { // what its fix-ups need again two chunks later
        int p;
        double len, alpha1, inv_len, inv_alpha1, start, alpha2, inv_alpha2, qlimit, sum_old, pix_area;
        unsigned char cut; // STRUCT: zero-length structure link -- its router output is stored as 0
        bool act;
    };
    auto body = [&](auto rc_tag) {
        constexpr bool RC = decltype(rc_tag)::value; // the five derived statics are recomputed (fused_args::recompute)
        auto issue_pre = [&](int ph, pre_t &P) {
            kargs_t Kp = K0; // (the ~40 array pointers: re-read from the kernel-argument segment per phase, see k_fused_cones)
            asm volatile("" : "+s"(Kp));
            const fused_args &G = *(const fused_args *)Kp;
            const lf_substep_args &A = G.S;
            const int j = ph * KC + sw;
            const int p = lbound(c0, j) + tid;
            P.act = p < lbound(c1, j);
            P.p = p;
            const int pc = P.act ? p : 0;
            // (array pointer in scalar registers + this lane's 32-bit byte offset: no address arithmetic per array, as cone_ld)
            const unsigned o8 = (unsigned)pc * 8u, o4 = (unsigned)pc * 4u, o1 = (unsigned)pc;
            P.u0 = split_ld(G.ups_ptr, o4);
            P.u1 = split_ld(G.ups_ptr, o4 + 4u);
            P.len = split_ld(A.ChanLength, o8);
            P.chan = split_ld(A.IsChannelKinematic, o1);
            P.eva = P.wuse = P.qin_old = P.qdelta = P.qin_added_old = P.chanq_old = P.transcum = P.lakeout = P.resout = P.polder = 0.0;
            P.uptrans = 0;
            if (!STRUCT)
                P.side_m3 = split_ld(fused_side(G, s), o8);
            else { // (an option that is off: any valid stream instead of a branch around the load; its value is not used)
                const lf_inloop_args &I = G.I;
                const double *any = I.ToChanM3RunoffDt;
                P.side_m3 = split_ld(any, o8);
                P.eva = split_ld(I.EvaAddM3Dt ? I.EvaAddM3Dt : any, o8);
                P.wuse = split_ld(I.WUseAddM3Dt ? I.WUseAddM3Dt : any, o8);
                P.qin_old = split_ld(I.QInM3Old ? I.QInM3Old : any, o8);
                P.qdelta = split_ld(I.QInM3Old ? I.QDelta : any, o8);
                P.qin_added_old = split_ld(I.QInM3Old ? (const double *)I.QinADDEDM3 : any, o8);
                P.chanq_old = split_ld(A.ChanQ, o8);
                P.uptrans = split_ld(I.UpTrans ? I.UpTrans : A.IsChannelKinematic, o1);
                P.transcum = split_ld(I.UpTrans ? (const double *)I.TransCum : any, o8);
                P.lakeout = split_ld(I.QLakeOutM3Dt ? (const double *)I.QLakeOutM3Dt : any, o8);
                P.resout = split_ld(I.QResOutM3Dt ? (const double *)I.QResOutM3Dt : any, o8);
                P.polder = split_ld(I.ChannelToPolderM3Dt ? I.ChannelToPolderM3Dt : any, o8);
            }
            P.qold = split_ld(A.ChanQKin, o8);
            P.alpha1 = split_ld(A.ChannelAlpha, o8);
            P.dxv = split_ld(G.dx ? G.dx : A.ChanLength, o8); // (no per-pixel dx: any valid stream, the scalar is selected)
            P.inv_len = RC ? 0.0 : split_ld(A.InvChanLength, o8);
            P.ap1 = RC ? 0.0 : split_ld(G.a1, o8);
            P.m3 = P.m3_2 = P.start = P.m3limit = P.q2start = P.q2old = P.alpha2 = P.ap2 = 0.0;
            if (SPLIT) {
                P.m3 = split_ld(A.ChanM3Kin, o8);
                P.m3_2 = split_ld(A.Chan2M3Kin, o8);
                P.start = split_ld(A.Chan2M3Start, o8);
                P.m3limit = split_ld(A.M3Limit, o8);
                P.q2start = split_ld(A.Chan2QStart, o8);
                P.q2old = split_ld(A.Chan2QKin, o8);
                P.alpha2 = split_ld(A.ChannelAlpha2, o8);
                P.ap2 = RC ? 0.0 : split_ld(G.a2, o8);
            }
        };
        auto finish_pre = [&](int ph, const pre_t &P, auto first_chunk) {
            kargs_t Kp = K0;
            asm volatile("" : "+s"(Kp));
            const fused_args &G = *(const fused_args *)Kp;
            const lf_substep_args &A = G.S;
            const int ob = ph & 1, jj = sw, j = ph * KC + jj;
            cone_chunk_ops<NR, KC> &O = S.ops[ob];
            const bool block_top = decltype(first_chunk)::value && sw == 0; // the block's first level
            // ---- the reads of the level above ----
            const int above = jj > 0 ? (ob * KC + jj - 1) * NR : ((ob ^ 1) * KC + KC - 1) * NR;
            int cnt = P.act ? P.u1 - P.u0 : 0;
            if (STRUCT && !block_top) {
                // the range of a level's LAST cell runs to the end of the level above, over the zero-length structure links
                // parked there (lf_graph.cpp): their router output counts as 0, so they are left out of the gather (x + 0.0
                // == x; the chain wavefront keeps their true output in LDS for their own fix-ups)
                const int lv = ld_table(G.fb_level, b) + j;          // absolute level of this cell
                const bool in_block = (unsigned)j < (unsigned)nl;
                typedef const long long __attribute__((address_space(4))) *cll; // (scalar load: the table is never written)
                const int level_end = in_block ? (int)((cll)(unsigned long long)G.level_start)[lv + 1] : 0;
                const int parked = (in_block && j > 0) ? ld_table(G.level_nlinked, lv - 1) : 0;
                if (P.act && P.p + 1 == level_end) cnt -= parked;
            }
            const int ab = above * kConeRow + (((P.u0 - lbound(c0, j - 1)) & 0xff) << 3);
            cone_rec_ad a0, a1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a0.ad[k] = (k < cnt && !block_top) ? ab + 8 * k : 64 * 8;
                a1.ad[k] = (k + 4 < cnt && !block_top) ? ab + 8 * (k + 4) : 64 * 8;
            }
            O.ad[jj][0][tid] = a0;
            O.ad[jj][1][tid] = a1;
            // ---- sideflow (routing.py:512, 524 / 549-567), as cone_compute ----
            const double dxp = dx_is_len ? P.len : (G.dx ? P.dxv : G.dx_scalar);
            const double inv_len = RC ? 1.0 / P.len : P.inv_len;
            const double ap1 = RC ? P.alpha1 * dxp / G.dt : P.ap1;
            double side_m3 = P.side_m3;
            if (STRUCT) { // inflow.py:142-144, transmission.py:76-87, sideflow assembly routing.py:462-478 -- as cone_compute
                const lf_inloop_args &I = G.I;
                double qin = 0.0, qin_added = 0.0, loss = 0.0, trans_cum = 0.0;
                if (I.EvaAddM3Dt) side_m3 -= P.eva;
                if (I.WUseAddM3Dt) side_m3 -= P.wuse;
                if (I.QInM3Old) {
                    qin = (P.qin_old + (s + 1) * P.qdelta) * I.InvNoRoutSteps;
                    qin_added = (s < 1 ? 0.0 : P.qin_added_old) + qin;
                    side_m3 += qin;
                }
                if (I.UpTrans) {
                    const double qc = P.chanq_old;
                    double tout = qc;
                    if (P.uptrans) {
                        const double inner = lf_pow_scalar_exponent(qc, I.TransPower2) - I.TransSub;
                        tout = lf_pow_scalar_exponent(inner, I.TransPower1);
                    }
                    loss = (qc - tout) * I.DtRouting;
                    trans_cum = P.transcum + loss;
                    side_m3 -= loss;
                }
                if (I.QLakeOutM3Dt) side_m3 += P.lakeout;
                if (I.QResOutM3Dt) side_m3 += P.resout;
                if (I.ChannelToPolderM3Dt) side_m3 -= P.polder;
                // what the assembly leaves behind (an option that is off: stored beyond the end of a buffer, dropped)
                const unsigned off = P.act ? (unsigned)P.p * 8u : 0xffffffffu;
                auto put = [&](double *base, bool on, double val) {
                    __builtin_amdgcn_raw_buffer_store_b64(
                        __builtin_bit_cast(v2i, val),
                        __builtin_amdgcn_make_buffer_rsrc(on ? base : I.SideflowChanM3, 0, (int)nbytes, 0x00020000),
                        on ? off : 0xffffffffu, 0, 0);
                };
                put(I.QInDt, I.QInM3Old != nullptr, qin);
                put(I.QinADDEDM3, I.QInM3Old != nullptr, qin_added);
                put(I.TransLossM3Dt, I.UpTrans != nullptr, loss);
                put(I.TransCum, I.UpTrans != nullptr, trans_cum);
                put(I.SideflowChanM3, true, side_m3);
            }
            const double side = (P.chan != 0) ? side_m3 * inv_len * A.InvDtRouting : 0.0;
            double s1 = side, s2 = 0.0;
            if (!SPLIT) {
                if (isnan(side)) s1 = 0.0;
            } else {
                const double tot = P.m3 + P.m3_2;
                const double ratio = (tot > 0) ? P.m3 / tot : 0.0;
                s1 = ((tot - P.start) > P.m3limit) ? ratio * side : side;
                if (fabs(side) < 1e-7) s1 = side;
                s2 = (side - s1) + P.q2start * inv_len;
            }
            s1buf[(ph + 3) % 3][jj][tid] = s1;
            // ---- constant terms of the two routers (kinematic_wave_parallel.py:163,175) ----
            double cst[2], apv[2];
            cst[0] = ap1 * cone_pow_3_5<true>(P.qold, G.beta, true) + s1 * dxp;
            apv[0] = ap1;
            if (SPLIT) {
                const double ap2 = RC ? P.alpha2 * dxp / G.dt : P.ap2;
                cst[1] = ap2 * cone_pow_3_5<true>(P.q2old, G.beta, true) + s2 * dxp;
                apv[1] = ap2;
            }
            if (decltype(first_chunk)::value) {
                if (block_top) { // from the block before (previous launch) through the parity buffers
                    cst[0] = upstream_sum8(G.qr1 + par, P.u0, P.u0 + cnt, G.kmax) + cst[0];
                    if (SPLIT) cst[1] = upstream_sum8(G.qr2 + par, P.u0, P.u0 + cnt, G.kmax) + cst[1];
                }
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const bool fast_a = lf_fast_range(apv[r]);
                cone_rec_ca ca;
                ca.cst = cst[r];
                ca.ap = fast_a ? apv[r] : 1.0;
                cone_rec_fl fl;
                fl.af = (float)ca.ap;
                fl.laf = __builtin_amdgcn_logf(fl.af);
                fl.fast_a = fast_a;
                fl.pos = P.act ? P.p : 0;
                O.ca[jj][r][tid] = ca;
                O.fl[jj][r][tid] = fl;
            }
        };
        auto issue_post = [&](int ph, post_t &R) {
            kargs_t Kp = K0;
            asm volatile("" : "+s"(Kp));
            const fused_args &G = *(const fused_args *)Kp;
            const lf_substep_args &A = G.S;
            const int j = ph * KC + sw;
            const int p = lbound(c0, j) + tid;
            R.act = p < lbound(c1, j);
            R.p = p;
            const int pc = R.act ? p : 0;
            const unsigned o8 = (unsigned)pc * 8u, o1 = (unsigned)pc;
            R.len = split_ld(A.ChanLength, o8);
            R.alpha1 = split_ld(A.ChannelAlpha, o8);
            R.sum_old = split_ld(fused_sum(G, s), o8);
            R.inv_len = RC ? 0.0 : split_ld(A.InvChanLength, o8);
            R.inv_alpha1 = RC ? 0.0 : split_ld(A.InvChannelAlpha, o8);
            R.pix_area = split_ld(A.PixelArea, last ? o8 : 0u); // (read by the last sub-step only: one line on the others)
            R.cut = STRUCT ? split_ld(G.linked, o1) : 0;
            R.start = R.alpha2 = R.inv_alpha2 = R.qlimit = 0.0;
            if (SPLIT) {
                R.start = split_ld(A.Chan2M3Start, o8);
                R.alpha2 = split_ld(A.ChannelAlpha2, o8);
                R.qlimit = split_ld(A.QLimit, o8);
                R.inv_alpha2 = RC ? 0.0 : split_ld(A.InvChannelAlpha2, o8);
            }
        };
        auto finish_post = [&](int ph, const post_t &R) { // fix-ups and stores of chunk ph (routing.py:526-532, 573-603, 693-703)
            kargs_t Kp = K0;
            asm volatile("" : "+s"(Kp));
            const fused_args &G = *(const fused_args *)Kp;
            const lf_substep_args &A = G.S;
            const int ob = ph & 1, jj = sw, j = ph * KC + jj;
            const double qr = S.xr[(ob * KC + jj) * NR][tid];
            const double q2r = SPLIT ? S.xr[(ob * KC + jj) * NR + (NR - 1)][tid] : 0.0;
            const double s1 = s1buf[(ph + 3) % 3][jj][tid];
            const double inv_len = RC ? 1.0 / R.len : R.inv_len;
            const double inv_alpha1 = RC ? 1.0 / R.alpha1 : R.inv_alpha1;
            double v = R.len * R.alpha1 * cone_pow_3_5<true>(qr, A.Beta, true);
            if (v < 0.0) v = 0.0;
            const double x = v * inv_len * inv_alpha1;
            const double q = cone_pow_5_3<true>(x, A.InvBeta, true);
            double chanq = q, v2 = 0, q2 = 0;
            if (SPLIT) {
                const double inv_alpha2 = RC ? 1.0 / R.alpha2 : R.inv_alpha2;
                v2 = R.len * R.alpha2 * cone_pow_3_5<true>(q2r, A.Beta, true);
                if ((v2 - R.start) < 0.0) v2 = R.start;
                const double x2 = v2 * inv_len * inv_alpha2;
                q2 = cone_pow_5_3<true>(x2, A.InvBeta, true);
                chanq = q + q2 - R.qlimit;
                if (chanq < 0.0) chanq = 0.0;
            }
            double vel = 0.0;
            if (last) { // routing.py:693-703
                double area = v * inv_len;
                if (area < 0.01) area = 0.01;
                const double v1 = q / area, vv2 = 0.36 * cold_pow(q, 0.24);
                vel = (vv2 < v1) ? vv2 : v1;
                if (isnan(vv2)) vel = vv2;
                double sinu = sqrt(R.pix_area) * inv_len;
                if (sinu > 1) sinu = 1;
                vel *= sinu;
            }
            const unsigned off = R.act ? (unsigned)R.p * 8u : 0xffffffffu; // beyond the buffer: dropped
            const unsigned off_last = last ? off : 0xffffffffu;           // what only the last sub-step leaves behind ...
            const unsigned off_keep = STRUCT ? off : off_last;            // ... or every sub-step when structures read ChanQ
            const unsigned off_out = (j == nl - 1) ? off : 0xffffffffu;   // the block's last level: read by the next block
            auto put = [&](double *base, unsigned o, double val) {
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, val),
                                                      __builtin_amdgcn_make_buffer_rsrc(base, 0, (int)nbytes, 0x00020000), o, 0, 0);
            };
            put(G.qr1 + par, off_out, R.cut ? 0.0 : qr);
            put(A.ChanM3Kin, off, v);
            put(A.ChanQKin, off, q);
            put(A.ChanQ, off_keep, chanq);
            put(fused_sum(G, s), off, R.sum_old + chanq);
            if (SPLIT) {
                put(G.qr2 + par, off_out, R.cut ? 0.0 : q2r);
                put(A.Sideflow1Chan, off_keep, s1);
                put(A.Chan2M3Kin, off, v2);
                put(A.CrossSection2Area, off_keep, (v2 - R.start) * inv_len);
                put(A.Chan2QKin, off, q2);
            }
            put(A.FlowVelocity, off_last, vel);
            put(A.TravelDistance, off_last, vel * A.DtSec);
        };
        // phase ph: pre-loads of chunk ph + 1, post-loads of chunk ph - 1, fix-ups of chunk ph - 2, constant terms of chunk
        // ph; barriers behind the phases 0 .. nch.  Chunks of even / odd number use the register sets A / B.
        pre_t PA, PB;
        post_t RA, RB;
        RB.act = false; // (the first trip runs the fix-ups of a chunk -1: nothing of it is stored)
        RB.p = 0;
        RB.cut = 0;
        RB.len = RB.alpha1 = RB.inv_len = RB.inv_alpha1 = RB.start = RB.alpha2 = RB.inv_alpha2 = RB.qlimit = RB.sum_old =
            RB.pix_area = 1.0;
        issue_pre(0, PA);
        issue_pre(1, PB);
        finish_pre(0, PA, std::true_type());
        lds_barrier();
        for (int ph = 1; ph <= nch; ph += 2) {
            issue_pre(ph + 1, PA);
            issue_post(ph - 1, RA);
            finish_post(ph - 2, RB);
            finish_pre(ph, PB, std::false_type());
            lds_barrier();
            if (ph + 1 > nch) break;
            issue_pre(ph + 2, PB);
            issue_post(ph, RB);
            finish_post(ph - 1, RA);
            finish_pre(ph + 1, PA, std::false_type());
            lds_barrier();
        }
        if ((nch - 1) & 1)
            finish_post(nch - 1, RB);
        else
            finish_post(nch - 1, RA);
    };
    if (dflags & 1u)
        body(std::true_type());
    else
        body(std::false_type());
}

} // namespace
