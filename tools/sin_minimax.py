"""The coefficients of sin_small2 (termdaw_amd/csrc/kernels.hip): the odd polynomial  r + r^3 (c3 + c5 r^2 + c7 r^4 + c9 r^6)
closest to sin(r) in the maximum norm on |r| <= pi/2 + 0.1 (the reduced argument of the magic-number reduction below 2e6 rad: its
quotient is rounded once, but fl(1 / pi) is short by 4e-8 of itself -- 0.026 half turns at 2e6 rad, |r| up to pi/2 + 0.08), by Lawson-weighted least squares on Chebyshev nodes; then the
same evaluated in f32 with fused steps against sin() in double.
    python tools/sin_minimax.py
Degree 9 in the maximum norm (9e-9) beats the degree-11 Taylor polynomial sin_any2 keeps (5.6e-8 at pi/2) with one multiply-add
less; in f32: max 1.3e-7 / RMS 2.1e-8 against 1.6e-7 / 6e-8 on the same range; the whole of sin_small2 over [0, 2e6] rad: max 1.2e-7, RMS 2.2e-8.  (sin_any2 -- arguments of any size -- rounds a
PRODUCT to the quotient, off by up to half a turn at 1e7 rad: there the Taylor polynomial, which degrades more gently outside
pi/2, stays: a render 161 s into a 12.5 kHz voice measured 1.3e-6 RMS with this one.)"""
import numpy as np


def fit(terms, R, n=6000, rounds=400):
    x = R * np.cos(np.pi * (np.arange(n) + 0.5) / n)
    x = x[x > 0]
    A = np.stack([x ** (2 * k + 3) for k in range(terms)], axis=1)
    b = np.sin(x) - x
    w = np.ones_like(x)
    for _ in range(rounds):
        c = np.linalg.lstsq(A * w[:, None], b * w, rcond=None)[0]
        e = np.abs(A @ c - b)
        w = w * (0.5 + e / e.max())
        w /= w.max()
    return c, np.abs(A @ c - b).max()


def fma(a, b, c):
    return np.float32(np.float64(a) * np.float64(b) + np.float64(c))


if __name__ == "__main__":
    R = np.pi / 2 + 0.1
    c, e = fit(4, R)
    cf = [np.float32(v) for v in c]
    print("double:", [repr(float(v)) for v in c], "max error %.3g" % e)
    print("f32:   ", [repr(float(v)) for v in cf])
    r = np.linspace(-R, R, 2000001).astype(np.float32)
    r2 = (r * r).astype(np.float32)
    p = np.full_like(r, cf[3])
    for k in (2, 1, 0):
        p = fma(p, r2, cf[k])
    q = fma(r, (r2 * p).astype(np.float32), r)
    err = np.abs(q.astype(np.float64) - np.sin(r.astype(np.float64)))
    print("f32 evaluation: max %.3g, RMS %.3g (|r| <= pi/2: max %.3g)" % (err.max(), np.sqrt((err ** 2).mean()), err[np.abs(r) <= np.pi / 2].max()))
