// lf_sweep.h -- device code of the level sweep, shared by the single-GPU router (lf_router.hip) and the
// row-block distributed router (lf_dist.hip).  See lf_router.hip for the data layout.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "lf_common.h"
#include "lf_math.h"

namespace {


constexpr int kBlock = 256;
constexpr int kNarrowBlock = 1024;
constexpr int kNarrowMax = 1024; // levels up to this many cells are swept by the single-workgroup kernel

inline int blocks_for(int64_t n) { return (int)((n + kBlock - 1) / kBlock); }

// solve1Pixel, kinematic_wave_parallel_tools.py:59-82 (c = const_plus_ups_infl)
__device__ __forceinline__ double lf_solve_cell(double c, double a, double ba, double beta, double inv_beta,
                                                double b_minus_1)
{
    if (c <= LF_NEWTON_TOL) return 0.0;
    const double t = ba * pow(c, b_minus_1);
    double secant;
    if (t <= 1.0)
        secant = c / (1.0 + t);
    else
        secant = c / (1.0 + pow(t, inv_beta));
    const double other = pow((c - secant) / a, inv_beta);
    double q = (secant + other) / 2.0;
    double err = q + a * pow(q, beta) - c; // closureError, :89-92
    double prev = -1.0;
    int count = 0;
    while (fabs(err) > LF_NEWTON_TOL && q != prev && count < LF_MAX_ITERS) {
        prev = q;
        q -= err / (1.0 + ba * pow(q, b_minus_1));
        q = (LF_NEWTON_TOL > q) ? LF_NEWTON_TOL : q; // builtins.max(q, NEWTON_TOL)
        err = q + a * pow(q, beta) - c;
        ++count;
    }
    if (q == LF_NEWTON_TOL) q = 0.0;
    return q;
}

// The same out of line: inside the cone kernel's level loop the general path is taken by single lanes with extreme
// arguments or not at all when beta = 3/5 -- one copy to jump to instead of ~1500 instructions to jump over per level
__device__ __attribute__((noinline)) double lf_solve_cell_cold(double c, double a, double ba, double beta, double inv_beta,
                                                               double b_minus_1)
{
    return lf_solve_cell(c, a, ba, beta, inv_beta, b_minus_1);
}
__device__ __attribute__((noinline)) double lf_pow_cold(double x, double y) { return pow(x, y); }
// lf_pow_3_5 with the OCML fallback (arguments beyond the fast range) out of line
__device__ __forceinline__ double lf_pow_3_5_hot(double x)
{
    if (x == 0.0) return 0.0;
    if (lf_fast_range(x)) {
        const double r = lf_root5(x);
        return r * r * r;
    }
    return lf_pow_cold(x, 0.6);
}

// constant = a*Qold^beta + q*dx (kinematic_wave_parallel.py:163,175), gathered into sweep order
__global__ void __launch_bounds__(kBlock) k_prep(int n, const int *__restrict__ perm, const double *__restrict__ q_pix,
                                                 const double *__restrict__ lat_pix, const double *__restrict__ a,
                                                 const double *__restrict__ dx, double dx_scalar, double beta,
                                                 double *__restrict__ constant)
{
    const int p = blockIdx.x * kBlock + threadIdx.x;
    if (p >= n) return;
    const int pix = perm ? perm[p] : p; // perm == nullptr: vectors already in sweep order
    const double lateral = lat_pix[pix] * (dx ? dx[p] : dx_scalar);
    constant[p] = a[p] * pow(q_pix[pix], beta) + lateral;
}

// everything one sweep launch needs (passed by value)
struct sweep_args {
    const int *__restrict__ ups_ptr;
    const int *__restrict__ ups_idx;     // INDEXED only: upstream positions (may point into the ghost slots)
    const int *__restrict__ ups_base;    // INDEXED only: first upstream position if they are consecutive, else -1
    const int *__restrict__ perm;        // pixel-order I/O only
    const double *__restrict__ a;        // alpha*dx/dt, sweep order
    const double *__restrict__ constant; // general path: written by k_prep
    const double *__restrict__ lat;      // fused path: specific lateral inflow (pixel order, or sweep order if ORDERED)
    const double *__restrict__ dx;       // fused path: per-pixel dx (sweep order) or nullptr
    double dx_scalar;
    double beta, inv_beta, b_minus_1;
    int kmax;     // max in-degree of the graph (<= 8)
    double *qord; // discharge in sweep order: upstream values are read from it, the new value is written to it
    double *q_pix; // pixel-order I/O: caller's discharge vector (old value in, new value out); unused if ORDERED
    // INDEXED only (row-block partition, pipelined calls): the old discharge is read from here instead of qord when not
    // null -- successive calls ping-pong between two state vectors so that a call may start before the one before it has
    // finished its later phases (lf_dist_router_route_many)
    const double *qold_src;
    // STATICS kernels only (wide levels of an ordered beta = 3/5 call): what a cell reads of the router's static vectors as
    // ONE record instead of separate loads -- the level sweep is sensitive to the number of its memory instructions at the
    // same bytes.  STATICS = 1: (a, dx), 16 bytes, ups_ptr read as before.
    const double2 *__restrict__ adx;
    // STATICS = 3 (INDEXED: the row-block partition's level kernel, which reads four static streams -- a, dx, ups_ptr,
    // ups_base): (a, dx, first list entry | count << 28, ups_base), 24 bytes
    const struct lf_rec24 *__restrict__ rec24;
};
struct lf_rec24 {
    double a, dx;
    unsigned int up; // first entry of the cell's upstream list (28 bits) | number of upstream cells << 28
    int base;        // ups_base: first upstream position if they are consecutive, else -1
};

// One cell of the implicit sweep.
//   FUSED (beta == 3/5): the old-discharge term is computed here (kinematic_wave_parallel.py:163,175 folded
//     into the sweep) and the closure is solved as a quintic in Q^(1/5) (lf_math.h); otherwise `constant`
//     comes from k_prep and the reference's own Newton iteration runs.
//   ORDERED: discharge and lateral inflow are resident in sweep order (engine layout): qord[p] holds the old
//     discharge on entry and the new one on exit, every access is a coalesced stream.  Otherwise they are
//     gathered from / scattered to the caller's pixel-order vectors through perm.
//   INDEXED: upstream positions come from the index list ups_idx[ups_ptr[p] .. ups_ptr[p+1]) instead of being the
//     contiguous positions themselves (row-block partitions: upstream cells may sit in other phases or in the
//     ghost slots filled by the halo exchange).
template <bool FUSED, bool ORDERED, bool INDEXED = false, int STATICS = 0>
__device__ __forceinline__ void sweep_cell_range(int p, int u0, int u1, const sweep_args &A)
{
    static_assert(STATICS == 0 || (FUSED && ORDERED && INDEXED == (STATICS == 3)), "the records are the ordered beta = 3/5 router's");
    const int pix = ORDERED ? p : A.perm[p];
    double ap, dxp;
    int rec_base = 0;
    if (STATICS == 3) {
        const lf_rec24 R = A.rec24[p];
        ap = R.a;
        dxp = R.dx;
        u0 = (int)(R.up & 0x0fffffffu);
        u1 = u0 + (int)(R.up >> 28);
        rec_base = R.base;
    } else if (STATICS == 1) {
        const double2 sd = A.adx[p];
        ap = sd.x;
        dxp = sd.y;
    } else {
        ap = A.a[p];
        dxp = (FUSED && A.dx) ? A.dx[p] : A.dx_scalar;
    }
    // the cell's own two values first: requested with the statics above, before anything waits for the upstream range
    double lat_q = 0.0, qold = 0.0;
    if (FUSED) {
        lat_q = A.lat[pix];
        qold = ORDERED ? ((INDEXED && A.qold_src) ? A.qold_src[p] : A.qord[p]) : A.q_pix[pix];
    }
    // Upstream inflow, summed in ascending pixel id (kinematic_wave_parallel_tools.py:57-58).  A D8 cell has
    // at most 8 upstream neighbours: all candidate loads are issued at once (predicated) instead of a
    // dependent load per loop trip; missing ones contribute +0.0, which leaves the sum bit-identical.
    // The requests go out as soon as the upstream range is known, BEFORE the old-discharge term is worked out (its power
    // runs while they travel), and nothing may use a value before all of them are out: the empty asm behind the power is
    // where they are first needed (without it the compiler sinks `0.0 + v[0]` into the first load's branch and waits for
    // that load before it requests the others).
    double v[8];
    const int base = INDEXED ? (STATICS == 3 ? rec_base : A.ups_base[p]) : u0;
    // (the upstream loads sit behind branches -- a wavefront none of whose lanes has a fourth pair skips that request -- so
    // the compiler cannot count them and waits for ALL loads wherever it needs one: the cell's own values are therefore
    // taken in here, before the upstream requests go out, and the power below then runs with only those in flight)
    // (pixel-order vectors: the cell's own values are themselves a second trip, behind perm -- the upstream requests go out
    // with them and one wait takes all of it in)
    if (FUSED && ORDERED) asm volatile("" : "+v"(qold), "+v"(lat_q));
    if (!INDEXED) {
        // the contiguous upstream run two values per load: four 16-byte loads instead of eight 8-byte ones (the sweep is
        // sensitive to the number of its memory instructions).  The second value of a pair may lie one position behind the
        // run -- still inside the vector (a cell's upstream positions all precede its own) -- and is then discarded.
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            double2 t = make_double2(0.0, 0.0);
            if (2 * j < A.kmax && u0 + 2 * j < u1) t = *(const double2 *)(A.qord + u0 + 2 * j); // (8-byte aligned)
            v[2 * j] = t.x;
            v[2 * j + 1] = t.y;
        }
    } else if (base >= 0) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (k < A.kmax && u0 + k < u1) ? A.qord[base + k] : 0.0;
    } else { // ghost or cross-phase inflow: positions from the list (a second, dependent load)
        // (all list entries first, then all discharges: entry by entry the compiler waits for every load in flight before each
        // dependent one -- sixteen round trips in a row for a wavefront with one such lane)
        int e[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) e[k] = (k < A.kmax && u0 + k < u1) ? A.ups_idx[u0 + k] : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(e[k]));
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = (k < A.kmax && u0 + k < u1) ? A.qord[e[k]] : 0.0;
    }
    double cst;
    if (FUSED) {
        const double lateral = lat_q * dxp;
        cst = ap * lf_pow_3_5(qold) + lateral;
    } else {
        cst = A.constant[p];
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) asm volatile("" : "+v"(v[k]));
    if (!INDEXED) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[2 * j + 1] = (u0 + 2 * j + 1 < u1) ? v[2 * j + 1] : 0.0;
    }
    double ups = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k) ups += v[k];
    const double c = ups + cst;
    double q;
    if (FUSED && lf_fast_range(c) && lf_fast_range(ap)) {
        q = (c <= LF_NEWTON_TOL) ? 0.0 : lf_solve_3_5(c, ap);
    } else {
        q = lf_solve_cell(c, ap, A.beta * ap, A.beta, A.inv_beta, A.b_minus_1); // incl. alpha == 0 / NaN semantics
    }
    A.qord[p] = q;
    if (!ORDERED) A.q_pix[pix] = q;
}
template <bool FUSED, bool ORDERED, bool INDEXED = false, int STATICS = 0>
__device__ __forceinline__ void sweep_cell(int p, const sweep_args &A)
{
    if (STATICS == 3)
        sweep_cell_range<FUSED, ORDERED, INDEXED, STATICS>(p, 0, 0, A); // (the upstream range comes with the record)
    else
        sweep_cell_range<FUSED, ORDERED, INDEXED, STATICS>(p, A.ups_ptr[p], A.ups_ptr[p + 1], A);
}

// Read of a table that no kernel of this library ever writes (the cone plans of the level blocks) through the
// constant address space: with a wavefront-uniform index it becomes a scalar load (s_load) instead of a vector load that
// the compiler, fearing an alias with the discharge stores, would wait for together with every store in flight.
__device__ __forceinline__ int ld_table(const int *p, int i)
{
    typedef const int __attribute__((address_space(4))) *const_int_ptr;
    return ((const_int_ptr)(unsigned long long)p)[i];
}

// one wide level: one cell per lane
#ifndef LF_LEVEL_WAVES
#define LF_LEVEL_WAVES 8 /* <= 80 SGPRs / 64 VGPRs: 8 wavefronts per SIMD instead of the 7 that 96 SGPRs allow (+2.3 %) */
#endif
#ifndef LF_LEVEL_BLOCK
#define LF_LEVEL_BLOCK 256
#endif
constexpr int kLevelBlock = LF_LEVEL_BLOCK; // workgroup of the wide-level kernel
inline int level_blocks_for(int64_t n) { return (int)((n + kLevelBlock - 1) / kLevelBlock); }
#if LF_LEVEL_WAVES
#define LF_LEVEL_ATTR __attribute__((amdgpu_waves_per_eu(LF_LEVEL_WAVES)))
#else
#define LF_LEVEL_ATTR
#endif
template <bool FUSED, bool ORDERED, bool INDEXED = false, int STATICS = 0>
__global__ void __launch_bounds__(kLevelBlock) LF_LEVEL_ATTR k_level(int first, int count, sweep_args A)
{
    const int i = blockIdx.x * kLevelBlock + threadIdx.x;
    if (i >= count) return;
    sweep_cell<FUSED, ORDERED, INDEXED, STATICS>(first + i, A);
}

// the 24-byte records of k_level<.., INDEXED, STATICS = 3> (row-block partition; once per router and section)
__global__ void __launch_bounds__(kLevelBlock) k_static_records_indexed(long long n, const double *__restrict__ a,
                                                                        const double *__restrict__ dx, const int *__restrict__ ups_ptr,
                                                                        const int *__restrict__ ups_base, lf_rec24 *__restrict__ rec)
{
    const long long i = (long long)blockIdx.x * kLevelBlock + threadIdx.x;
    if (i >= n) return;
    lf_rec24 r;
    r.a = a[i];
    r.dx = dx[i];
    r.up = (unsigned)ups_ptr[i] | ((unsigned)(ups_ptr[i + 1] - ups_ptr[i]) << 28);
    r.base = ups_base[i];
    rec[i] = r;
}

// the (a, dx) records of k_level<.., STATICS = 1> from the router's static vectors (once per router and section)
__global__ void __launch_bounds__(kLevelBlock) k_static_records(long long n, const double *__restrict__ a, const double *__restrict__ dx,
                                                                double2 *__restrict__ adx)
{
    const long long i = (long long)blockIdx.x * kLevelBlock + threadIdx.x;
    if (i < n) adx[i] = make_double2(a[i], dx[i]);
}

// Several routers on ONE graph (surface_routing.py:151-153: the direct / other / forest overland routers differ only in
// alpha and in their vectors) swept together: blockIdx.y (wide levels) or blockIdx.x (narrow runs) picks the router, so a
// level costs one launch for all of them.
constexpr int kMaxMulti = 4;
struct sweep_args_multi {
    sweep_args r[kMaxMulti];
};

template <bool FUSED, bool ORDERED>
__global__ void __launch_bounds__(kBlock) LF_LEVEL_ATTR k_level_multi(int first, int count, sweep_args_multi M)
{
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= count) return;
    sweep_cell<FUSED, ORDERED, false>(first + i, M.r[blockIdx.y]);
}

template <bool FUSED, bool ORDERED>
__global__ void __launch_bounds__(kNarrowBlock) k_levels_narrow_multi(int k0, int k1, const long long *__restrict__ level_start,
                                                                      sweep_args_multi M)
{
    const sweep_args &A = M.r[blockIdx.x];
    for (int k = k0; k < k1; ++k) {
        const int first = (int)level_start[k], last = (int)level_start[k + 1];
        for (int p = first + (int)threadIdx.x; p < last; p += kNarrowBlock) sweep_cell<FUSED, ORDERED, false>(p, A);
        __threadfence_block();
        __syncthreads();
    }
}

// One BLOCK of consecutive levels of a router call, cone by cone (the plan of lf_router.hip: build_level_blocks).  A
// workgroup owns a chunk of the block's last level and the whole upstream cone above it -- level by level one contiguous
// range of at most CW cells, the cones tile every level -- so the levels of a block need no synchronisation between
// workgroups: the new discharges travel to the next level through LDS (two buffers by level parity; they are also
// stored, the block after this one and later calls read them).  One level of a cone is a dependent chain about as long
// as one trip to memory and nothing else runs on its SIMD, so the operands of a level are requested TWO levels ahead
// (three rotating register sets), the request holds loaded values only and sits behind no branch, the stores of a level
// leave one level later, and the rarely taken general-exponent paths live out of line.  NR
// routers of one graph (the overland routers of surface_routing.py:151-153) share the cone.  Arithmetic per cell =
// sweep_cell: bit-identical to the level sweep.
struct cone_plan_args {
    const int *__restrict__ cone; // starts of this block's cones, nl per cone, then the closing row (ends of the levels)
    int nl;                       // levels of the block
    int n_cells;                  // cells of the graph (k_sweep_cones_split: extent of the state vectors, < 2^29)
};

template <int CW>
__device__ __forceinline__ void cone_sync() // between the levels of a cone: workgroup barrier, or nothing for one wavefront
{
    if (CW > 64)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else
        asm volatile("" ::: "memory"); // its LDS operations complete in order
}

// CW: cells per level of a cone = threads of the workgroup (64: one wavefront, no barrier between the levels)
template <bool FUSED, bool ORDERED, int NR, int CW = kBlock>
__global__ void __launch_bounds__(CW) k_sweep_cones(cone_plan_args C, sweep_args_multi M)
{
    __shared__ double x[NR][2][CW];
    const int tid = threadIdx.x, nl = C.nl;
    const int *c0 = C.cone + (size_t)blockIdx.x * nl, *c1 = c0 + nl;
    struct cell { // the operands of one cell as loaded: no arithmetic before the level that solves it (a product here
                  // would make the loads wait where they are issued)
        int u0, u1, pix;
        double ap[NR], lat[NR], dx[NR], qold[NR];
        bool active;
    };
    auto load = [&](int p, bool active, cell &R) {
        R.active = active;
        const int pc = active ? p : 0; // lanes beyond the cone's range load cell 0: no branch around the loads
        R.u0 = M.r[0].ups_ptr[pc];
        R.u1 = M.r[0].ups_ptr[pc + 1];
        R.pix = ORDERED ? pc : M.r[0].perm[pc];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const sweep_args &A = M.r[r];
            R.ap[r] = A.a[pc];
            if (FUSED) {
                R.lat[r] = A.lat[R.pix];
                R.dx[r] = A.dx ? A.dx[pc] : A.dx_scalar;
                R.qold[r] = ORDERED ? A.qord[pc] : A.q_pix[R.pix];
            } else {
                R.lat[r] = A.constant[pc];
                R.dx[r] = 1.0;
                R.qold[r] = 0.0;
            }
        }
    };
    int first_up = 0;
    // the stores of a level are issued one level later, right behind the barrier (with the next loads): then everything
    // outstanding at the end of a level was issued before its arithmetic, and the wait there is short
    double pend_q[NR];
    int pend_p = 0, pend_pix = 0;
    bool pend = false;
    auto flush = [&]() {
        if (!pend) return;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            M.r[r].qord[pend_p] = pend_q[r];
            if (!ORDERED) M.r[r].q_pix[pend_pix] = pend_q[r];
        }
        pend = false;
    };
    // Software pipeline of depth 2: level j is solved from operands loaded while level j - 2 was solved, so a load has two
    // level times to arrive (one level of a cone takes about as long as one trip to memory: with depth 1 every level
    // waited for its own loads).  Three operand sets rotate; the waits are the compiler's (vmcnt counts the loads and
    // stores issued since, nothing is waited for that is not needed).
    auto level = [&](int j, const cell &cur, cell &nn, int first, int last, int fn2, int ln2) {
        const int p = first + tid;
        flush();
        load(fn2 + tid, fn2 + tid < ln2, nn); // nothing of them depends on this launch; beyond the block: bounds 0, 0
        if (j > 0) { // level j-1 of this cone is in LDS
            if (CW > 64)
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            else
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // one wavefront: its LDS operations complete in order
        }
        // every lane runs the level's arithmetic (a lane beyond the cone's range: on cell 0, nothing of it is kept), the
        // quintic path without a branch on sanitised arguments -- the loop over the levels is straight code but for the
        // block's first level and the out-of-line general path (single lanes with extreme arguments, or beta != 3/5)
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const sweep_args &A = M.r[r];
            double v[8];
            if (j == 0) { // from the block before (previous launch)
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (k < A.kmax && cur.u0 + k < cur.u1) ? A.qord[cur.u0 + k] : 0.0;
            } else {
                const double *y = &x[r][(j - 1) & 1][0];
                const int base = cur.u0 - first_up;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const bool have = k < A.kmax && cur.u0 + k < cur.u1;
                    const double t = y[have ? base + k : 0];
                    v[k] = have ? t : 0.0;
                }
            }
            double ups = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) ups += v[k];
            const double ap = cur.ap[r];
            double q;
            if (FUSED) {
                const double qo = cur.qold[r];
                const bool fq = lf_fast_range(qo);
                const double rt = lf_root5(fq ? qo : 1.0);
                double pw = fq ? rt * rt * rt : 0.0;                  // lf_pow_3_5: +-0 -> +0
                if (!fq && qo != 0.0) pw = lf_pow_cold(qo, 0.6);      // beyond the fast range, NaN
                const double c = ups + (ap * pw + cur.lat[r] * cur.dx[r]);
                const bool fc = lf_fast_range(c) && lf_fast_range(ap);
                const bool solve = fc && !(c <= LF_NEWTON_TOL);
                q = lf_solve_3_5(solve ? c : 1.0, solve ? ap : 1.0);
                q = solve ? q : 0.0;                                   // fast range and c <= NEWTON_TOL: 0
                if (!fc) q = lf_solve_cell_cold(c, ap, A.beta * ap, A.beta, A.inv_beta, A.b_minus_1);
            } else {
                const double c = ups + cur.lat[r];
                q = lf_solve_cell(c, ap, A.beta * ap, A.beta, A.inv_beta, A.b_minus_1);
            }
            pend_q[r] = q;
            if (j + 1 < nl) x[r][j & 1][tid] = q;
        }
        pend = cur.active;
        pend_p = p;
        pend_pix = cur.pix;
        first_up = first;
        (void)last;
    };
    auto bound = [&](const int *t, int k) { return k < nl ? ld_table(t, k) : 0; };
    cell r0, r1, r2;
    int f0 = ld_table(c0, 0), f1 = bound(c0, 1), f2 = bound(c0, 2), l0 = ld_table(c1, 0), l1 = bound(c1, 1), l2 = bound(c1, 2);
    load(f0 + tid, f0 + tid < l0, r0);
    load(f1 + tid, f1 + tid < l1, r1);
    // three levels per trip, no branch around a level or its loads (the wait counts stay exact): a level beyond the block
    // has the bounds 0, 0 -- no active cell, only its barrier
    for (int j = 0; j < nl; j += 3) {
        // bounds of the levels whose operands this round of three requests
        const int f3 = bound(c0, j + 3), l3 = bound(c1, j + 3), f4 = bound(c0, j + 4), l4 = bound(c1, j + 4);
        const int f5 = bound(c0, j + 5), l5 = bound(c1, j + 5);
        level(j, r0, r2, f0, l0, f2, l2);
        level(j + 1, r1, r0, f1, l1, f3, l3);
        level(j + 2, r2, r1, f2, l2, f4, l4);
        f0 = f3;
        f1 = f4;
        f2 = f5;
        l0 = l3;
        l1 = l4;
        l2 = l5;
    }
    flush();
}

// The cone sweep with the work split between TWO wavefronts of a 128-thread workgroup (round 4).  One level of a cone is
// a dependent chain that a single wavefront issues in order; on chain-bound networks three quarters of the SIMDs idle while
// that wavefront also works out everything that does NOT depend on the level above (operand addresses and loads, the
// old-discharge term a*Qold^0.6 + q*dx, the stores).  Here
//   * the SUPPLY wavefront (threads 64..127) runs a chunk of KC levels ahead: it requests every operand of the chunk at
//     once, works out the constant term of every cell (kinematic_wave_parallel.py:163,175) and parks it in LDS together
//     with a and the cell's upstream range as (first slot, count) in the level above; one chunk later it takes the chunk's
//     new discharges out of LDS and stores them;
//   * the CHAIN wavefront (threads 0..63) touches LDS and registers only: gather the level above -> sum in ascending
//     pixel id (kinematic_wave_parallel_tools.py:57-58) -> closure solve -> LDS.
// The two meet at one workgroup barrier per chunk; operand and result buffers alternate by chunk parity.  The upstream
// discharges of the block's first level are final in memory (the block before): the supply wavefront adds them up and
// hands the chain wavefront c = ups + constant with an empty range (0.0 + c == c for every c that is solved; c == -0.0
// gives 0 either way).  Arithmetic per cell = sweep_cell: bit-identical to the level sweep.
#ifndef LF_CONE_KC
#define LF_CONE_KC 8 /* levels per chunk and supply wavefronts of the few-cones shape (lf_router.hip: launch_split_shape) */
#endif
#ifndef LF_CONE_NS
#define LF_CONE_NS 4
#endif
constexpr int kConeSlots = 65;           // a level's 64 discharges + one slot holding 0.0 ("no such upstream cell" reads it)
constexpr int kConeRow = kConeSlots * 8; // bytes of a row

// What the supply wavefronts park for a chunk of KC levels, per lane in 16-byte records (one ds_read_b128 each).
struct cone_rec_ca {
    double cst; // a*Qold^beta + q*dx (first level of the block: + the inflow from the block before)
    double ap;  // a, 1.0 where a is outside the fast range (the cold path reads the true value from memory)
};
struct cone_rec_fl {
    float af, laf; // (float)a and log2 of it: the part of lf_solve_3_5's seed that does not depend on c
    int fast_a;    // a in the fast range
    int pos;       // the position the lane's operands were read from (cell 0 for a lane beyond the cone's range)
};
struct cone_rec_ad {
    int ad[4]; // byte offsets (in the result rows) of four reads of the level above: slot first + k, or the 0.0 in slot 64
};
template <int NR, int KC>
struct cone_chunk_ops {
    cone_rec_ca ca[KC][NR][64];
    cone_rec_fl fl[KC][NR][64];
    cone_rec_ad ad[KC][2][64]; // reads 0..3 and 4..7 of router 0 (router r: + r rows)
};
template <int NR, int KC>
struct cone_lds {
    double xr[2 * KC * NR][kConeSlots]; // row (chunk parity * KC + level of the chunk) * NR + router; first: small offsets
    cone_chunk_ops<NR, KC> ops[2];
};

// The chain wavefront's loop over the chunks, for a graph of at most KM upstream cells per cell: LDS and registers only.
template <bool FUSED, int NR, int KC, int KM>
__device__ __forceinline__ void cone_chain(cone_lds<NR, KC> &S, int tid, int nl, const int *c0, const sweep_args_multi &M)
{
    const int nch = (nl + KC - 1) / KC;
    const char *xb = (const char *)&S.xr[0][0];
    struct operands {
        cone_rec_ca ca[NR];
        cone_rec_fl fl[NR];
        cone_rec_ad a0, a1;
    };
    auto request = [&](const cone_chunk_ops<NR, KC> &O, int jj, operands &P) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            P.ca[r] = O.ca[jj][r][tid];
            P.fl[r] = O.fl[jj][r][tid];
        }
        P.a0 = O.ad[jj][0][tid];
        if (KM > 4) P.a1 = O.ad[jj][1][tid];
    };
    for (int ch = 0; ch < nch; ++ch) {
        const int ob = ch & 1, L = nl - ch * KC < KC ? nl - ch * KC : KC;
        const cone_chunk_ops<NR, KC> &O = S.ops[ob];
        // one level: `cur` holds its operands, those of the next level are requested into `nxt` behind the gather
        auto level = [&](int jj, const operands &cur, operands &nxt) {
            // the level above, all reads in flight together
            double t[NR][KM];
#pragma unroll
            for (int r = 0; r < NR; ++r)
#pragma unroll
                for (int k = 0; k < KM; ++k) // (router r: r rows on; slot 64 of EVERY row holds 0.0)
                    t[r][k] = *(const double *)(xb + (k < 4 ? cur.a0.ad[k & 3] : cur.a1.ad[k & 3]) + r * kConeRow);
            request(O, jj + 1 < L ? jj + 1 : jj, nxt); // (the chunk's last level: its own again)
            const int row = (ob * KC + jj) * NR;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                // ascending pixel id (kinematic_wave_parallel_tools.py:57-58); the reference starts from 0.0, and 0.0 + x
                // is x bit for bit unless x is -0.0, which no discharge written to LDS is (0.0, a positive root, or NaN)
                double ups = t[r][0];
#pragma unroll
                for (int k = 1; k < KM; ++k) ups += t[r][k];
                const double c = ups + cur.ca[r].cst;
                double q;
                if (FUSED) {
                    // c <= NEWTON_TOL: 0 on every path; else the quintic when c and a are in its range, else the general
                    // path (c beyond 1e30, NaN, a outside the fast range: kinematic_wave_parallel_tools.py:59-82)
                    const bool le = c <= LF_NEWTON_TOL;
                    const bool quintic = c <= LF_FAST_MAX && cur.fl[r].fast_a != 0;
#ifdef LF_EXP_NOSOLVE /* timing experiments only (tools/build_variant.sh): WRONG results, bounded values */
                    q = c * 1e-3 + 0.5 * cur.ca[r].cst;
#else
                    q = lf_solve_3_5_pre(c, cur.ca[r].ap, cur.fl[r].af, cur.fl[r].laf); // (lanes not selected: garbage)
#endif
                    q = (le || !quintic) ? 0.0 : q;
                    if (!le && !quintic) {
                        const sweep_args &A = M.r[r];
                        const double at = A.a[cur.fl[r].pos]; // (a lane beyond the cone's range: cell 0, like its other operands)
                        q = lf_solve_cell_cold(c, at, A.beta * at, A.beta, A.inv_beta, A.b_minus_1);
                    }
                } else {
                    const sweep_args &A = M.r[r];
                    q = lf_solve_cell(c, cur.ca[r].ap, A.beta * cur.ca[r].ap, A.beta, A.inv_beta, A.b_minus_1);
                }
                S.xr[row + r][tid] = q;
            }
        };
        operands pa, pb; // two sets, swapping roles level by level (no copies)
        request(O, 0, pa); // the chunk's first level; inside the chunk the operands are requested one level ahead
#ifdef LF_EXP_NOCHAIN /* timing experiments only: the supply side alone, WRONG results */
        for (int jj = 0; jj < L; ++jj) S.xr[(ob * KC + jj) * NR][tid] = 1.0;
        for (int jj = 0; jj < 0; jj += 2) {
#else
        for (int jj = 0; jj < L; jj += 2) {
#endif
            level(jj, pa, pb);
            if (jj + 1 >= L) break;
            level(jj + 1, pb, pa);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
}

// NS supply wavefronts share a chunk level by level (wavefront s: levels s, s + NS, ... of every chunk); each keeps the
// operands of its levels of the NEXT chunk in flight while it works out the current one (two register sets).
template <bool FUSED, bool ORDERED, int NR, int KC, int NS>
__global__ void __launch_bounds__(64 * (1 + NS)) k_sweep_cones_split(cone_plan_args C, sweep_args_multi M)
{
    __shared__ cone_lds<NR, KC> S;
    static_assert(2 * KC * NR <= 64, "one lane per row");
    static_assert(KC % NS == 0, "every supply wavefront takes KC / NS levels of a chunk");
    constexpr int KS = KC / NS;
    const int tid = threadIdx.x & 63, nl = C.nl;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); // 0 = chain, 1.. = supply (uniform: scalar table reads)
    const int nch = (nl + KC - 1) / KC;
    const int *c0 = C.cone + (size_t)blockIdx.x * nl, *c1 = c0 + nl;
    if (wave == 0) {
        if (tid < 2 * KC * NR) S.xr[tid][64] = 0.0; // the zero slot of every row
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // chunk 0 is in LDS
        switch (M.r[0].kmax) {
        case 0:
        case 1: cone_chain<FUSED, NR, KC, 1>(S, tid, nl, c0, M); break;
        case 2: cone_chain<FUSED, NR, KC, 2>(S, tid, nl, c0, M); break;
        case 3: cone_chain<FUSED, NR, KC, 3>(S, tid, nl, c0, M); break;
        case 4: cone_chain<FUSED, NR, KC, 4>(S, tid, nl, c0, M); break;
        case 5: cone_chain<FUSED, NR, KC, 5>(S, tid, nl, c0, M); break;
        case 6: cone_chain<FUSED, NR, KC, 6>(S, tid, nl, c0, M); break;
        case 7: cone_chain<FUSED, NR, KC, 7>(S, tid, nl, c0, M); break;
        default: cone_chain<FUSED, NR, KC, 8>(S, tid, nl, c0, M); break;
        }
        return;
    }
    // ---- a supply wavefront ----
    // Its loop over the chunks is straight code but for the out-of-line pow of an extreme old discharge: no branch around
    // a load or a store (lanes beyond the cone's range load cell 0 and store beyond the end of a buffer resource, which the
    // hardware drops; phases beyond the block work on empty levels), so the compiler's wait counts stay exact and the
    // operands requested during one phase are not waited for before the next.
    const int sw = wave - 1;
    typedef int v2i __attribute__((ext_vector_type(2)));
    __amdgpu_buffer_rsrc_t q_rsrc[NR], qpix_rsrc[NR];
    const double *dxp[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        q_rsrc[r] = __builtin_amdgcn_make_buffer_rsrc(M.r[r].qord, 0, (int)((unsigned)C.n_cells * 8u), 0x00020000);
        qpix_rsrc[r] = __builtin_amdgcn_make_buffer_rsrc(ORDERED ? M.r[r].qord : M.r[r].q_pix, 0, (int)((unsigned)C.n_cells * 8u), 0x00020000);
        dxp[r] = M.r[r].dx ? M.r[r].dx : M.r[r].a; // (no per-pixel dx: any valid stream, the scalar is selected)
    }
    auto lbound = [&](const int *t, int k) { // entry k of a cone's row of the plan, 0 outside the block: an unconditional
        const bool in = (unsigned)k < (unsigned)nl; // scalar load and a select, no branch in the load step
        const int v = ld_table(t, in ? k : 0);
        return in ? v : 0;
    };
    struct opset { // the operands of this wavefront's KS levels of a chunk, as loaded
        int u0[KS], u1[KS], pix[KS];
        double ap[KS][NR], lat[KS][NR], dx[KS][NR], qo[KS][NR];
        bool act[KS];
    };
    auto issue = [&](int ph, opset &P) { // independent loads, all in flight together
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const int j = ph * KC + sw + i * NS;
            const int p = lbound(c0, j) + tid;
            P.act[i] = p < lbound(c1, j);    // a level beyond the block: bounds 0, 0
            const int pc = P.act[i] ? p : 0; // lanes beyond the cone's range load cell 0: no branch around the loads
            P.u0[i] = M.r[0].ups_ptr[pc];
            P.u1[i] = M.r[0].ups_ptr[pc + 1];
            P.pix[i] = ORDERED ? pc : M.r[0].perm[pc];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const sweep_args &A = M.r[r];
                P.ap[i][r] = A.a[pc];
                if (FUSED) {
                    P.lat[i][r] = A.lat[P.pix[i]];
                    P.dx[i][r] = dxp[r][pc];
                    P.qo[i][r] = ORDERED ? A.qord[pc] : A.q_pix[P.pix[i]];
                } else {
                    P.lat[i][r] = A.constant[pc];
                    P.dx[i][r] = 1.0;
                    P.qo[i][r] = 0.0;
                }
            }
        }
    };
    auto store = [&](int ph) { // results of chunk ph - 2: out of LDS, into the state vector
        const int ob = ph & 1;
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const int jj = sw + i * NS, j = (ph - 2) * KC + jj;
            const int p = lbound(c0, j) + tid;
            const bool act = p < lbound(c1, j);
            const unsigned off = act ? (unsigned)p * 8u : 0xffffffffu; // beyond the buffer: dropped
            unsigned off_pix = off;
            if (!ORDERED) {
                const int pix = M.r[0].perm[act ? p : 0];
                off_pix = act ? (unsigned)pix * 8u : 0xffffffffu;
            }
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const double q = S.xr[(ob * KC + jj) * NR + r][tid];
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, q), q_rsrc[r], off, 0, 0);
                if (!ORDERED) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i, q), qpix_rsrc[r], off_pix, 0, 0);
            }
        }
    };
    auto finish = [&](int ph, const opset &P, auto first_chunk) { // constant terms of chunk ph
        const int ob = ph & 1;
        cone_chunk_ops<NR, KC> &O = S.ops[ob];
#pragma unroll
        for (int i = 0; i < KS; ++i) {
            const int jj = sw + i * NS, j = ph * KC + jj;
            const bool block_top = decltype(first_chunk)::value && i == 0 && sw == 0; // the block's first level
            // the reads of the level above: its rows are those of level jj - 1 of this chunk or the last of the chunk before
            const int above = jj > 0 ? (ob * KC + jj - 1) * NR : ((ob ^ 1) * KC + KC - 1) * NR;
            int cnt = P.act[i] ? P.u1[i] - P.u0[i] : 0;
            const int ab = above * kConeRow + (((P.u0[i] - lbound(c0, j - 1)) & 0xff) << 3);
            cone_rec_ad a0, a1;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a0.ad[k] = (k < cnt && !block_top) ? ab + 8 * k : 64 * 8;
                a1.ad[k] = (k + 4 < cnt && !block_top) ? ab + 8 * (k + 4) : 64 * 8;
            }
            O.ad[jj][0][tid] = a0;
            O.ad[jj][1][tid] = a1;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const sweep_args &A = M.r[r];
                double cst;
                if (FUSED) {
                    const double q = P.qo[i][r];
                    const bool fq = lf_fast_range(q);
                    const double rt = lf_root5(fq ? q : 1.0);
                    double pw = fq ? rt * rt * rt : 0.0;           // lf_pow_3_5: +-0 -> +0
                    if (!fq && q != 0.0) pw = lf_pow_cold(q, 0.6); // beyond the fast range, NaN
                    cst = P.ap[i][r] * pw + P.lat[i][r] * (A.dx ? P.dx[i][r] : A.dx_scalar);
                } else {
                    cst = P.lat[i][r];
                }
                if (decltype(first_chunk)::value && i == 0) {
                    if (block_top) { // its upstream cells are final in memory (the block before)
                        double v[8];
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = (k < A.kmax && k < cnt) ? A.qord[P.u0[i] + k] : 0.0;
                        double ups = 0.0;
#pragma unroll
                        for (int k = 0; k < 8; ++k) ups += v[k];
                        cst = ups + cst;
                    }
                }
                const bool fast_a = lf_fast_range(P.ap[i][r]);
                cone_rec_ca ca;
                ca.cst = cst;
                ca.ap = (fast_a || !FUSED) ? P.ap[i][r] : 1.0;
                cone_rec_fl fl;
                fl.af = (float)ca.ap;
                fl.laf = __builtin_amdgcn_logf(fl.af);
                fl.fast_a = fast_a;
                fl.pos = P.act[i] ? lbound(c0, j) + tid : 0;
                O.ca[jj][r][tid] = ca;
                O.fl[jj][r][tid] = fl;
            }
        }
    };
    // phase ph: request chunk ph + 1, store chunk ph - 2, work out chunk ph; barriers behind the phases 0 .. nch
    opset SA, SB;
    issue(0, SA);
    issue(1, SB);
    finish(0, SA, std::true_type());
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (int ph = 1; ph <= nch; ph += 2) {
        issue(ph + 1, SA);
        store(ph);
        finish(ph, SB, std::false_type());
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (ph + 1 > nch) break;
        issue(ph + 2, SB);
        store(ph + 1);
        finish(ph + 1, SA, std::false_type());
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    store(nch + 1);
}

// The same on the row-block partition (lf_dist.hip; one router, sweep-order vectors, one wavefront per cone).  A cell's
// upstream cells are the consecutive run starting at ups_base or come from the list ups_idx; those in the unit above of
// this cone are new this launch and come from LDS, all others -- ghost slots, earlier phases, the block before -- are
// final in the state vector.  Reading those inside the level would cost two dependent trips to memory on every level that
// has one such cell, so the prefetch has THREE stages over five rotating register sets:
//   A  four units ahead: upstream range, run start and the operands;
//   B  two units ahead:  the upstream positions (run start + k, or the list entries -- needs A);
//   C  one unit ahead:   the discharges of the positions that are not in the unit above (needs B; usually L2 hits);
// the level itself touches LDS and registers only.  qold_src: the old discharge from a second state vector (pipelined calls).
template <bool FUSED>
__global__ void __launch_bounds__(64) k_sweep_cones_dist(cone_plan_args C, sweep_args A)
{
    __shared__ double x[2][64];
    const int tid = threadIdx.x, nl = C.nl;
    const int *c0 = C.cone + (size_t)blockIdx.x * nl, *c1 = c0 + nl;
    const double *qold_from = A.qold_src ? A.qold_src : A.qord;
    const int kmax = A.kmax;
    struct cell {
        int u0, u1, base, p;
        double ap, lat, dx, qold;
        int e[8];
        double gq[8];
        bool active;
    };
    auto stage_a = [&](int first, int last, cell &R) {
        const int p = first + tid;
        R.active = p < last;
        const int pc = R.active ? p : 0; // a lane beyond the cone's range: cell 0, nothing of it is kept
        R.p = p;
        R.u0 = A.ups_ptr[pc];
        R.u1 = A.ups_ptr[pc + 1];
        R.base = A.ups_base[pc];
        R.ap = A.a[pc];
        if (FUSED) {
            R.lat = A.lat[pc];
            R.dx = A.dx ? A.dx[pc] : A.dx_scalar;
            R.qold = qold_from[pc];
        } else {
            R.lat = A.constant[pc];
            R.dx = 1.0;
            R.qold = 0.0;
        }
    };
    auto stage_b = [&](cell &R) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const bool have = k < kmax && R.u0 + k < R.u1;
            int e = R.base + k;
            if (R.base < 0) e = have ? A.ups_idx[R.u0 + k] : 0; // ghost or cross-phase inflow: from the list
            R.e[k] = have ? e : -1;
        }
    };
    // [fa, la): the cone's range in the unit above (empty for the block's first unit: everything comes from memory)
    auto stage_c = [&](cell &R, int fa, int la) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = R.e[k];
            const bool ext = e >= 0 && !(e >= fa && e < la);
            R.gq[k] = 0.0;
            if (ext) R.gq[k] = A.qord[e];
        }
    };
    double pend_q = 0.0;
    int pend_p = 0;
    bool pend = false;
    auto level = [&](int j, const cell &cur, cell &n1, cell &n2, cell &n4, int first, int last, int fa, int la, int f4,
                     int l4) {
        if (pend) A.qord[pend_p] = pend_q;
        stage_a(f4, l4, n4);
        stage_b(n2);
        stage_c(n1, first, last);
        if (j > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // one wavefront: its LDS operations complete in order
        const double *y = &x[(j - 1) & 1][0];
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int e = cur.e[k];
            const bool above = e >= fa && e < la;
            const double t = y[above ? e - fa : 0];
            v[k] = above ? t : cur.gq[k]; // (no upstream cell k: gq = 0)
        }
        double ups = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) ups += v[k];
        const double ap = cur.ap;
        double q;
        if (FUSED) {
            const double qo = cur.qold;
            const bool fq = lf_fast_range(qo);
            const double rt = lf_root5(fq ? qo : 1.0);
            double pw = fq ? rt * rt * rt : 0.0;
            if (!fq && qo != 0.0) pw = lf_pow_cold(qo, 0.6);
            const double c = ups + (ap * pw + cur.lat * cur.dx);
            const bool fc = lf_fast_range(c) && lf_fast_range(ap);
            const bool solve = fc && !(c <= LF_NEWTON_TOL);
            q = lf_solve_3_5(solve ? c : 1.0, solve ? ap : 1.0);
            q = solve ? q : 0.0;
            if (!fc) q = lf_solve_cell_cold(c, ap, A.beta * ap, A.beta, A.inv_beta, A.b_minus_1);
        } else {
            q = lf_solve_cell(ups + cur.lat, ap, A.beta * ap, A.beta, A.inv_beta, A.b_minus_1);
        }
        x[j & 1][tid] = q;
        pend = cur.active;
        pend_p = cur.p;
        pend_q = q;
    };
    auto bound = [&](const int *t, int k) { return k < nl ? ld_table(t, k) : 0; };
    cell s0, s1, s2, s3, s4;
    int f[5], l[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        f[i] = bound(c0, i);
        l[i] = bound(c1, i);
    }
    stage_a(f[0], l[0], s0);
    stage_a(f[1], l[1], s1);
    stage_a(f[2], l[2], s2);
    stage_a(f[3], l[3], s3);
    stage_b(s0);
    stage_b(s1);
    stage_c(s0, 0, 0);
    int fa = 0, la = 0; // the cone's range in the unit above the current one
    for (int j = 0; j < nl; j += 5) {
        int fn[5], ln[5]; // bounds of the units j + 5 .. j + 9
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            fn[i] = bound(c0, j + 5 + i);
            ln[i] = bound(c1, j + 5 + i);
        }
        // (the exits leave the trip straight code: a stage of three units -- the tail phases -- does not pay for five)
        level(j, s0, s1, s2, s4, f[0], l[0], fa, la, f[4], l[4]);
        if (j + 1 >= nl) break;
        level(j + 1, s1, s2, s3, s0, f[1], l[1], f[0], l[0], fn[0], ln[0]);
        if (j + 2 >= nl) break;
        level(j + 2, s2, s3, s4, s1, f[2], l[2], f[1], l[1], fn[1], ln[1]);
        if (j + 3 >= nl) break;
        level(j + 3, s3, s4, s0, s2, f[3], l[3], f[2], l[2], fn[2], ln[2]);
        if (j + 4 >= nl) break;
        level(j + 4, s4, s0, s1, s3, f[4], l[4], f[3], l[3], fn[3], ln[3]);
        fa = f[4];
        la = l[4];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            f[i] = fn[i];
            l[i] = ln[i];
        }
    }
    if (pend) A.qord[pend_p] = pend_q;
}

// a run of narrow levels [k0, k1): one workgroup, barrier between levels
template <bool FUSED, bool ORDERED, bool INDEXED = false>
__global__ void __launch_bounds__(kNarrowBlock) k_levels_narrow(int k0, int k1, const long long *__restrict__ level_start,
                                                                sweep_args A)
{
    for (int k = k0; k < k1; ++k) {
        const int first = (int)level_start[k], last = (int)level_start[k + 1];
        for (int p = first + (int)threadIdx.x; p < last; p += kNarrowBlock) sweep_cell<FUSED, ORDERED, INDEXED>(p, A);
        __threadfence_block();
        __syncthreads();
    }
}

} // namespace
