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

// N independent x^0.6 in lockstep (stage by stage for all arguments, so the N dependent chains interleave in the
// instruction stream); element by element the operations of lf_pow_3_5
template <int N>
__device__ __forceinline__ void lf_pow_3_5_n(const double (&x)[N], double (&out)[N])
{
    bool fast[N];
    double xs[N], r[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        fast[i] = lf_fast_range(x[i]);
        xs[i] = fast[i] ? x[i] : 1.0;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) r[i] = (double)__builtin_amdgcn_exp2f(0.2f * __builtin_amdgcn_logf((float)xs[i]));
#pragma unroll
    for (int it = 0; it < 2; ++it) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const double r2 = r[i] * r[i], r4 = r2 * r2;
            r[i] = fma(-fma(r4, r[i], -xs[i]), lf_rcp(5.0 * r4), r[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < N; ++i) out[i] = r[i] * r[i] * r[i];
#pragma unroll
    for (int i = 0; i < N; ++i)
        if (!fast[i]) out[i] = (x[i] == 0.0) ? 0.0 : pow(x[i], 0.6); // zero, or beyond the fast range (rare)
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

// Newton steps of lf_solve_3_5 behind its seed: 1 in fp32 + 2 in fp64 (rounds 1-3: 4 + 3 from a seed up to 19 % off).
// A relative error e becomes <= 2 e^2 per step (g''/(2 g') * r is in [1, 2]), so the seed's <= 0.2 % gives 8e-6 -> 1.3e-10
// -> 3e-20: rounding.  tools/solve_steps_check.py replays the scheme in numpy against a long-double root.
#ifndef LF_SOLVE_F32_STEPS
#define LF_SOLVE_F32_STEPS 1
#endif
#ifndef LF_SOLVE_F64_STEPS
#define LF_SOLVE_F64_STEPS 2
#endif

// Seed of the root of r^5 + a r^3 = c from cf = (float)c, af = (float)a, laf = log2(af).  ra = c^(1/5) and rb = (c/a)^(1/3)
// are both upper bounds of the root r*; with r = sqrt(a) y the equation is y^5 + y^3 = c a^(-5/2), a ONE-parameter family,
// so r* / min(ra, rb) is a function of d = log2(ra / rb) alone: 1 at both ends, 0.838 at its lowest.  It is fitted by
// 1 - x (A + B x), x = 2^(-p |d|) with one rate p per sign of d, to 0.2 % (max over the whole family:
// tools/solve_steps_check.py); the upper bound itself is up to 19 % off.
__device__ __forceinline__ float lf_solve_3_5_seed(float cf, float laf)
{
    const float lc = __builtin_amdgcn_logf(cf);
    const float la = 0.2f * lc, lb = 0.33333334f * (lc - laf);
    const float d = la - lb;
    const float x = __builtin_amdgcn_exp2f(fminf(-4.16617164f * d, 3.23546616f * d)); // 2^(-p |d|)
    const float corr = fmaf(-x, fmaf(-0.05679083f, x, 0.21762144f), 1.0f);
    return __builtin_amdgcn_exp2f(fminf(la, lb)) * corr;
}

// r >= LF_ROOT_FLOOR  <=>  fl(fl(fl(r r)^2) r) > 1e-12 (the product is monotone in r; the threshold is the smallest
// double whose fifth power, rounded as computed below, exceeds 1e-12 -- tests/test_host_cpu.py checks both neighbours)
#define LF_ROOT_FLOOR 0x1.04e74cc73ee88p-8

// the Newton steps and the floor of lf_solve_3_5, from the seed on
__device__ __forceinline__ double lf_solve_3_5_from(float rf, double c, double a, float cf, float af)
{
#pragma unroll
    for (int i = 0; i < LF_SOLVE_F32_STEPS; ++i) {
        const float r2 = rf * rf, r3 = r2 * rf;
        const float g = fmaf(r3, r2, fmaf(af, r3, -cf));
        const float gp = r2 * fmaf(5.0f, r2, 3.0f * af);
        rf = fmaf(-g, __builtin_amdgcn_rcpf(gp), rf);
    }
    double r = (double)rf;
#pragma unroll
    for (int i = 0; i < LF_SOLVE_F64_STEPS; ++i) {
        const double r2 = r * r, r3 = r2 * r;
        const double g = fma(r3, r2, fma(a, r3, -c));
        const double gp = r2 * fma(5.0, r2, 3.0 * a);
        r = fma(-g, lf_rcp(gp), r);
    }
    const double r2 = r * r;
    const double q = (r2 * r2) * r;
    return (r >= LF_ROOT_FLOOR) ? q : 0.0; // q > 1e-12, decided beside the products instead of behind them
}

// Root of Q + a*Q^(3/5) = c for c > 1e-12, a > 0 (both finite, in the fast range): returns Q >= 0.
// Mirrors the reference's floor: a root at or below NEWTON_TOL = 1e-12 is reported as 0
// (kinematic_wave_parallel_tools.py:77,81-82).
__device__ __forceinline__ double lf_solve_3_5(double c, double a)
{
    const float cf = (float)c, af = (float)a;
    return lf_solve_3_5_from(lf_solve_3_5_seed(cf, __builtin_amdgcn_logf(af)), c, a, cf, af);
}

// the same with the a-only part of the seed supplied (af = (float)a, laf = log2(af)): the same operations
__device__ __forceinline__ double lf_solve_3_5_pre(double c, double a, float af, float laf)
{
    const float cf = (float)c;
    return lf_solve_3_5_from(lf_solve_3_5_seed(cf, laf), c, a, cf, af);
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
#ifndef LF_RCP_SEED
#define LF_RCP_SEED(d) __builtin_amdgcn_rcp(d)
#define LF_FREXP_EXP(x) __builtin_amdgcn_frexp_exp(x)
#define LF_FREXP_MANT(x) __builtin_amdgcn_frexp_mant(x)
#endif
// N independent powers in LOCKSTEP: every stage of the algorithm is written for all N arguments before the next stage,
// so the N dependent chains are interleaved in the instruction stream (the scheduler keeps calls that follow one another
// in the source one after the other, each a ~90-instruction dependent chain: measured 45 % VALU busy at two wavefronts
// per SIMD).  Element by element the arithmetic is exactly that of one call.
template <int N>
__device__ __forceinline__ void lf_pow_pos_n(const double (&x)[N], const double (&y)[N], double (&out)[N])
{
    // branch-free: the main path runs on a sanitised argument and the special cases are selected at the end by plain
    // two-operand selects (no nested conditional expressions: the compiler turns those into divergent branches, which
    // cut the caller's loop body into basic blocks)
    bool pos[N];
    int e[N];
    double f[N], s[N], R[N], hfsq[N], l2m[N], p[N], p_err[N], n[N], u[N], ex[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        pos[i] = x[i] > 0.0;
        const double xs = pos[i] ? x[i] : 1.0;
        e[i] = LF_FREXP_EXP(xs);
        double m = LF_FREXP_MANT(xs); // [0.5, 1)
        const bool lo = m < 0.70710678118654752440;
        m = lo ? m * 2.0 : m;
        e[i] = lo ? e[i] - 1 : e[i];
        f[i] = m - 1.0;
    }
    // s = f / (2 + f), 2 + f in [1.70, 3.42): hardware reciprocal seed, two Newton steps, one correction of the
    // quotient (no scaling or fix-up needed in this range; <= 1 ulp)
    double d[N], rc[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        d[i] = 2.0 + f[i];
        rc[i] = LF_RCP_SEED(d[i]);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) rc[i] = fma(fma(-d[i], rc[i], 1.0), rc[i], rc[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) rc[i] = fma(fma(-d[i], rc[i], 1.0), rc[i], rc[i]);
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double s0 = f[i] * rc[i];
        s[i] = fma(fma(-d[i], s0, f[i]), rc[i], s0);
    }
    // fdlibm-style log(m): 7-term minimax in z = s^2, split in even and odd powers of w = z^2
    double z[N], w[N], t1[N], t2[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        z[i] = s[i] * s[i];
        w[i] = z[i] * z[i];
        hfsq[i] = 0.5 * f[i] * f[i];
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        t1[i] = fma(w[i], 1.531383769920937332e-01, 2.222219843214978396e-01);
        t2[i] = fma(w[i], 1.479819860511658591e-01, 1.818357216161805012e-01);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        t1[i] = fma(w[i], t1[i], 3.999999999940941908e-01);
        t2[i] = fma(w[i], t2[i], 2.857142874366239149e-01);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        t1[i] = w[i] * t1[i];
        t2[i] = z[i] * fma(w[i], t2[i], 6.666666666666735130e-01);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        R[i] = t2[i] + t1[i];
        const double ln_m = f[i] - fma(-s[i], hfsq[i] + R[i], hfsq[i]); // log(m), |error| < 1 ulp
        l2m[i] = ln_m * 1.44269504088896340736;                          // log2(m), |l2m| <= 0.5
    }
    // y * log2(x) = y * e + y * log2(m), the first product carried with its rounding error; n = nearest integer
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double ed = (double)e[i];
        p[i] = y[i] * ed;
        p_err[i] = fma(y[i], ed, -p[i]);
        const double q = y[i] * l2m[i];
        n[i] = rint(p[i] + q);
        const double r = ((p[i] - n[i]) + q) + p_err[i]; // |r| <= ~0.5
        u[i] = r * 0.69314718055994530942;
    }
    // e^u, |u| <= 0.36: degree-13 Taylor polynomial in Estrin form (dependent depth 5 instead of 14)
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double u1 = u[i];
        const double u2 = u1 * u1, u4 = u2 * u2, u8 = u4 * u4;
        const double a0 = fma(1.0, u1, 1.0);
        const double a1 = fma(1.6666666666666666e-01, u1, 0.5);
        const double a2 = fma(8.333333333333333e-03, u1, 4.1666666666666664e-02);
        const double a3 = fma(1.984126984126984e-04, u1, 1.388888888888889e-03);
        const double a4 = fma(2.7557319223985893e-06, u1, 2.48015873015873e-05);
        const double a5 = fma(2.505210838544172e-08, u1, 2.755731922398589e-07);
        const double a6 = fma(1.6059043836821613e-10, u1, 2.08767569878681e-09);
        const double b0 = fma(a1, u2, a0), b1 = fma(a3, u2, a2), b2 = fma(a5, u2, a4);
        const double d0 = fma(b1, u4, b0), d1 = fma(a6, u4, b2);
        ex[i] = fma(d1, u8, d0);
    }
#pragma unroll
    for (int i = 0; i < N; ++i) {
        const double nn = fmin(fmax(n[i], -2000.0), 2000.0);
        const double val = ldexp(ex[i], (int)nn);
        // 0 -> 0; negative base or NaN -> NaN, as OCML pow and numpy's ** for a non-integer exponent; a NaN exponent
        // propagates through val (y * ed) and through at_zero
        const double at_zero = (y[i] != y[i]) ? y[i] : 0.0;
        const double not_pos = (x[i] == 0.0) ? at_zero : __builtin_nan("");
        out[i] = pos[i] ? val : not_pos;
    }
}

__device__ __forceinline__ double lf_pow_pos(double x, double y)
{
    const double xs[1] = {x}, ys[1] = {y};
    double r[1];
    lf_pow_pos_n<1>(xs, ys, r);
    return r[0];
}

// x^y of the transmission loss (transmission.py:76-87: (Q^p2 - sub)^p1, both exponents scalars of the settings): lf_pow_pos
// for a finite positive exponent and a base that is not negative (0 -> 0, NaN -> NaN, as pow), OCML pow otherwise -- a
// NEGATIVE base must keep pow's semantics: with the settings' defaults TransPower1 = 2, TransSub = 0.3 the inner term is
// negative on every reach below 0.09 m3/s and pow(-0.3, 2.0) = 0.09, not NaN (integer exponents are defined there).  ~75
// instead of ~220 instructions per call on the common path; every kernel that computes the loss goes through here, so
// they stay bit-identical to one another.
// Out of line: inlined twice into the cone kernels it cost them 30 VGPRs (spills in k_fused_cones<STRUCT>, one wavefront
// per SIMD in k_fused_cones_split<STRUCT>).
// The settings' defaults are TransPower1 = 2 and TransPower2 = 1 / TransPower1 = 0.5 (cold.xml:1329, transmission.py:57):
// numpy itself evaluates `arr ** 2.0` as np.square and `arr ** 0.5` as np.sqrt (its scalar-exponent fast paths), so x * x
// and the IEEE square root are what the reference computes there -- one and ~15 instructions (y is wavefront-uniform).
static __device__ __attribute__((noinline)) double lf_pow_scalar_exponent(double x, double y)
{
    if (y == 2.0) return x * x;
    if (y == 0.5) return sqrt(x);
    if (y > 0.0 && y < 1e6 && !(x < 0.0)) return lf_pow_pos(x, y); // (y is a kernel argument; negative bases are the rare lanes)
    return pow(x, y);
}
