// lf_math.h -- device math for the kinematic-wave solve.
//
// The reference solves, per cell,   Q + a*Q^beta = C   (a = alpha*dx/dt, kinematic_wave_parallel_tools.py:89-92)
// by Newton-Raphson on Q with ~11 fp64 pow() per cell; on gfx950 an fp64 pow is ~190 VALU instructions
// (no fp64 transcendental unit), which makes the sweep ALU-bound at ~8 % of the HBM roofline.
//
// beta is a scalar (routing.py:66) and is 0.6 = 3/5 in every LISFLOOD setting (Manning).  For that value
// the substitution r = Q^(1/5) turns the closure into the POLYNOMIAL
//         g(r) = r^5 + a*r^3 - C = 0,        Q = r^5,
// which Newton solves with multiplies and one reciprocal per iteration -- no pow at all.  g is convex and
// increasing on r > 0, so Newton started at an upper bound of the root converges monotonically.  The
// result is the root of the same equation the reference iterates on, to ~1e-15 relative; the reference
// itself stops at |closure error| <= 1e-12 (absolute), so the two agree to rounding for ordinary
// discharges and to <= 1e-12 m3/s absolutely everywhere (tests: rtol 1e-9, atol 1e-12).
// Any other beta takes the general path (lf_solve_cell in lf_router.hip), which follows the reference's
// own iteration with OCML pow.
//
// x^(1/5) and x^(1/3) are built from a hardware fp32 log2/exp2 seed (v_log_f32 / v_exp_f32, ~1e-7) and
// fp64 Newton steps with v_rcp_f64 (the steps are self-correcting, so the raw ~2^-26 reciprocal is enough).
#pragma once
#include <hip/hip_runtime.h>

#define LF_FAST_MIN 1e-30 // fast paths handle LF_FAST_MIN <= x <= LF_FAST_MAX, everything else falls back
#define LF_FAST_MAX 1e30

__device__ __forceinline__ double lf_rcp(double x) { return __builtin_amdgcn_rcp(x); }

// x^(1/5), LF_FAST_MIN <= x <= LF_FAST_MAX; ~1 ulp
__device__ __forceinline__ double lf_root5(double x)
{
    const float xf = (float)x;
    double r = (double)__builtin_amdgcn_exp2f(0.2f * __builtin_amdgcn_logf(xf));
#pragma unroll
    for (int i = 0; i < 2; ++i) { // r <- r - (r^5 - x) / (5 r^4): 2e-7 -> 8e-14 -> rounding
        const double r2 = r * r, r4 = r2 * r2;
        r = fma(-fma(r4, r, -x), lf_rcp(5.0 * r4), r);
    }
    return r;
}

// x^(1/3), LF_FAST_MIN <= x <= LF_FAST_MAX; ~1 ulp
__device__ __forceinline__ double lf_cbrt(double x)
{
    const float xf = (float)x;
    double r = (double)__builtin_amdgcn_exp2f(0.33333334f * __builtin_amdgcn_logf(xf));
#pragma unroll
    for (int i = 0; i < 2; ++i) { // r <- r - (r^3 - x) / (3 r^2)
        const double r2 = r * r;
        r = fma(-fma(r2, r, -x), lf_rcp(3.0 * r2), r);
    }
    return r;
}

__device__ __forceinline__ bool lf_fast_range(double x) { return x >= LF_FAST_MIN && x <= LF_FAST_MAX; }

// x^0.6 for beta = 3/5; exact pow semantics outside the fast range (0, negatives, NaN, inf, extremes)
__device__ __forceinline__ double lf_pow_3_5(double x)
{
    if (x == 0.0) return 0.0; // pow(+-0, 0.6) = +0: dry cells are common and must not take the OCML path
    if (lf_fast_range(x)) {
        const double r = lf_root5(x);
        return r * r * r;
    }
    return pow(x, 0.6);
}

// x^(1/0.6) = x^(5/3) = x * cbrt(x)^2
__device__ __forceinline__ double lf_pow_5_3(double x)
{
    if (x == 0.0) return 0.0; // pow(+-0, 5/3) = +0
    if (lf_fast_range(x)) {
        const double r = lf_cbrt(x);
        return x * (r * r);
    }
    return pow(x, 1.0 / 0.6);
}

// Root of Q + a*Q^(3/5) = c for c > 1e-12, a > 0 (both finite, in the fast range): returns Q >= 0.
// Mirrors the reference's floor: a root at or below NEWTON_TOL = 1e-12 is reported as 0
// (kinematic_wave_parallel_tools.py:77,81-82).
__device__ __forceinline__ double lf_solve_3_5(double c, double a)
{
    // upper bounds of the root r*: r*^5 <= c and a r*^3 <= c
    const float cf = (float)c, af = (float)a;
    const float lc = __builtin_amdgcn_logf(cf);
    const float ra = __builtin_amdgcn_exp2f(0.2f * lc);
    const float rb = __builtin_amdgcn_exp2f(0.33333334f * (lc - __builtin_amdgcn_logf(af)));
    float rf = fminf(ra, rb) * 1.000001f;
    // fp32 Newton: from <= 15 % above the root to ~1e-7
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float r2 = rf * rf, r3 = r2 * rf;
        const float g = fmaf(r3, r2, fmaf(af, r3, -cf));
        const float gp = r2 * fmaf(5.0f, r2, 3.0f * af);
        rf = fmaf(-g, __builtin_amdgcn_rcpf(gp), rf);
    }
    double r = (double)rf;
    // fp64 Newton: 1e-7 -> 2e-14 -> rounding (third step is insurance for the worst start)
#ifndef LF_SOLVE_F64_STEPS
#define LF_SOLVE_F64_STEPS 3
#endif
#pragma unroll
    for (int i = 0; i < LF_SOLVE_F64_STEPS; ++i) {
        const double r2 = r * r, r3 = r2 * r;
        const double g = fma(r3, r2, fma(a, r3, -c));
        const double gp = r2 * fma(5.0, r2, 3.0 * a);
        r = fma(-g, lf_rcp(gp), r);
    }
    const double r2 = r * r;
    const double q = (r2 * r2) * r;
    return (q > 1e-12) ? q : 0.0;
}


// ------------------------------------------------------------------------------------------------
// x^y for x >= 0 and finite y > 0 with per-lane exponents (soil kernel: Xinanjiang, Van Genuchten).
// OCML's pow costs ~220 VALU instructions on gfx950, most of them double-double bookkeeping and special-case
// handling the soil kernel never needs (negative bases, integer-exponent tests, overflow).  This version is
// exp2(y * log2(x)) with an fdlibm-style log (argument reduced to [sqrt(1/2), sqrt(2)), 7-term minimax in
// s = f/(2+f)), the product y*e carried with its rounding error (fma), and a degree-13 polynomial for 2^r:
// ~70 instructions, relative error ~1e-15 for |y log2 x| <= 50 (measured against OCML in the gpu tests).
// x == 0 -> 0, x == 1 -> 1, NaN propagates, x == +inf -> +inf.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ double lf_pow_pos(double x, double y)
{
    // branch-free: the main path runs on a sanitised argument and the special cases are selected at the end, so
    // several inlined calls form ONE basic block and the scheduler interleaves their dependent chains
    const bool pos = x > 0.0;
    const double xs = pos ? x : 1.0;
    int e = __builtin_amdgcn_frexp_exp(xs);
    double m = __builtin_amdgcn_frexp_mant(xs); // [0.5, 1)
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m * 2.0 : m;
    e = lo ? e - 1 : e;
    const double f = m - 1.0;
    const double s = f / (2.0 + f);
    const double z = s * s, w = z * z;
    const double t1 = w * (3.999999999940941908e-01 + w * (2.222219843214978396e-01 + w * 1.531383769920937332e-01));
    const double t2 = z * (6.666666666666735130e-01 +
                           w * (2.857142874366239149e-01 + w * (1.818357216161805012e-01 + w * 1.479819860511658591e-01)));
    const double R = t2 + t1;
    const double hfsq = 0.5 * f * f;
    const double ln_m = f - (hfsq - s * (hfsq + R)); // log(m), |error| < 1 ulp
    const double l2m = ln_m * 1.44269504088896340736;  // log2(m), |l2m| <= 0.5
    const double ed = (double)e;
    const double p = y * ed;
    const double p_err = fma(y, ed, -p);
    const double q = y * l2m;
    const double n = rint(p + q);
    const double r = ((p - n) + q) + p_err; // |r| <= ~0.5
    const double u = r * 0.69314718055994530942;
    // e^u, |u| <= 0.36
    // degree-13 Taylor polynomial in Estrin form (dependent depth 5 instead of 14)
    const double u2 = u * u, u4 = u2 * u2, u8 = u4 * u4;
    const double a0 = fma(1.0, u, 1.0);
    const double a1 = fma(1.6666666666666666e-01, u, 0.5);
    const double a2 = fma(8.333333333333333e-03, u, 4.1666666666666664e-02);
    const double a3 = fma(1.984126984126984e-04, u, 1.388888888888889e-03);
    const double a4 = fma(2.7557319223985893e-06, u, 2.48015873015873e-05);
    const double a5 = fma(2.505210838544172e-08, u, 2.755731922398589e-07);
    const double a6 = fma(1.6059043836821613e-10, u, 2.08767569878681e-09);
    const double b0 = fma(a1, u2, a0), b1 = fma(a3, u2, a2), b2 = fma(a5, u2, a4);
    const double d0 = fma(b1, u4, b0), d1 = fma(a6, u4, b2);
    const double ex = fma(d1, u8, d0);
    const double nn = fmin(fmax(n, -2000.0), 2000.0);
    const double val = ldexp(ex, (int)nn);
    // 0 -> 0; negative base or NaN -> NaN, as OCML pow and numpy's ** for a non-integer exponent; a NaN exponent
    // propagates through val (y * ed)
    return pos ? val : ((x == 0.0) ? ((y != y) ? y : 0.0) : __builtin_nan(""));
}
