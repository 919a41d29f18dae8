"""Replay of lf_solve_3_5 (csrc/lf_math.h) in numpy: the fitted seed, N32 fp32 + N64 fp64 Newton steps, against a
long-double root -- over the one-parameter family y^5 + y^3 = k the solve depends on (r = sqrt(a) y) at several a, and
over random (c, a).  Prints the seed's largest relative error and the result's for a few step counts; `--fit` refits
the seed's four constants.  (fp32 log2/exp2/rcp here are correctly rounded; the hardware's are within 1 ulp: 1e-7,
against a seed tolerance of 1e-3.)"""
import sys

import numpy as np

f32 = np.float32
A, B, P1, P2 = f32(0.21762144), f32(-0.05679083), f32(4.16617164), f32(3.23546616)
rng = np.random.default_rng(0)


def seed(cf, af):
    lc, laf = np.log2(cf), np.log2(af)
    la, lb = f32(0.2) * lc, f32(0.33333334) * (lc - laf)
    d = la - lb
    x = np.exp2(np.minimum(-P1 * d, P2 * d))
    return (np.exp2(np.minimum(la, lb)) * (f32(1) - x * (A + B * x))).astype(f32)


def solve(c, a, n32, n64, perturb=0.0, rcp_bits=26):
    cf, af = c.astype(f32), a.astype(f32)
    rf = (seed(cf, af) * f32(1 + perturb)).astype(f32)
    for _ in range(n32):
        r2 = rf * rf
        r3 = r2 * rf
        g = r3 * r2 + (af * r3 - cf)
        rf = (rf - g / (r2 * (f32(5) * r2 + f32(3) * af))).astype(f32)
    r = rf.astype(np.float64)
    for _ in range(n64):
        r2 = r * r
        r3 = r2 * r
        g = (r3.astype(np.longdouble) * r2 + (a.astype(np.longdouble) * r3 - c)).astype(np.float64)  # the fma pair
        rc = 1.0 / (r2 * (5.0 * r2 + 3.0 * a))
        rc = rc * (1 + rng.uniform(-1, 1, rc.size) * 2.0 ** -rcp_bits)                          # v_rcp_f64: ~2^-26
        r = r - g * rc
    return r


def exact(c, a):
    r = solve(c, a, 3, 3).astype(np.longdouble)
    cl, al = c.astype(np.longdouble), a.astype(np.longdouble)
    for _ in range(4):
        r2 = r * r
        r3 = r2 * r
        r = r - (r3 * r2 + al * r3 - cl) / (r2 * (5 * r2 + 3 * al))
    return r


def report(tag, c, a):
    ex = exact(c, a)
    s = seed(c.astype(f32), a.astype(f32)).astype(np.float64)
    print("%s: seed off by at most %.4f %%" % (tag, 100 * np.abs(s / ex.astype(np.float64) - 1).max()))
    for n32, n64, pt in ((1, 2, 0.0), (1, 2, 0.002), (1, 2, -0.002), (0, 3, 0.0), (1, 1, 0.0)):
        err = np.abs((solve(c, a, n32, n64, pt).astype(np.longdouble) - ex) / ex).astype(np.float64)
        print("    %d fp32 + %d fp64 steps, seed scaled by %+.3f: max relative error of the root %.3e (%.1f ulp)" % (
            n32, n64, pt, err.max(), err.max() / 1.11e-16))


def fit():
    from scipy.optimize import minimize
    w = np.linspace(-25, 25, 400001)
    L = 3 * w + np.log2(1 + 4.0 ** w)
    la, lb = L / 5, L / 3
    d = la - lb
    phi = 2.0 ** (w - np.minimum(la, lb))

    def err(p):
        x = 2.0 ** np.minimum(-p[2] * d, p[3] * d)
        return np.abs((1 - x * (p[0] + p[1] * x)) / phi - 1).max()
    best = None
    for a0 in (0.1, 0.16, 0.2):
        for b0 in (-0.05, 0.0, 0.05):
            for p1 in (2, 3.5):
                for p2 in (2, 2.8):
                    r = minimize(err, [a0, b0, p1, p2], method="Nelder-Mead", options=dict(xatol=1e-9, fatol=1e-11, maxiter=8000))
                    if best is None or r.fun < best.fun:
                        best = r
    print("A, B, p1, p2 =", best.x, " max relative error", best.fun, " (lowest root / upper bound: %.4f)" % phi.min())


if __name__ == "__main__":
    if "--fit" in sys.argv:
        fit()
        sys.exit(0)
    w = np.linspace(-14, 14, 2_000_001)
    y = 2.0 ** w
    for a0 in (1.0, 3.7e-3, 41.0, 1e-9, 1e9):
        rr = np.sqrt(a0) * y
        c = rr ** 5 + a0 * rr ** 3
        ok = (c > 1e-12) & (c < 1e30)
        report("family at a = %g" % a0, c[ok], np.full(int(ok.sum()), a0))
    n = 2_000_000
    report("random c in [1e-11, 1e12], a in [1e-6, 1e8]", np.exp(rng.uniform(np.log(1e-11), np.log(1e12), n)),
           np.exp(rng.uniform(np.log(1e-6), np.log(1e8), n)))
