"""Generates the polynomial coefficients of the canonical fp32 activations.

TEST INFRASTRUCTURE (oracle/): not imported by the product path.

The HIP kernels and the C oracle evaluate tanh / exp / gelu with the same
sequence of IEEE-754 fp32 operations (fma, mul, add, div, rint, bit ops), so
the two agree bit for bit.  This script derives the coefficients (weighted
least squares on Chebyshev nodes, fp64) and prints them as C hex-float
literals; the values are pasted into oracle/lrnde_oracle.c and
localregneuralde.jl_amd/csrc/lrnde_math.hpp.
"""
import numpy as np

def cheb_nodes(a, b, n):
    k = np.arange(n)
    x = np.cos(np.pi * (2 * k + 1) / (2 * n))
    return 0.5 * (a + b) + 0.5 * (b - a) * x

def fit(f, a, b, deg, n=None, weight=None):
    """Chebyshev interpolation of degree `deg` on [a, b] (near-minimax), power basis out."""
    from numpy.polynomial import chebyshev as C, polynomial as P
    g = lambda z: f(0.5 * (a + b) + 0.5 * (b - a) * z)
    cz = C.chebinterpolate(g, deg)
    pz = C.cheb2poly(cz)                      # polynomial in z = (2x-(a+b))/(b-a)
    # substitute z = alpha*x + beta
    alpha, beta = 2.0 / (b - a), -(a + b) / (b - a)
    out = np.zeros(1)
    lin = np.array([beta, alpha])
    pw = np.ones(1)
    for c in pz:
        out = P.polyadd(out, c * pw)
        pw = P.polymul(pw, lin)
    return out

def hexf(v):
    return float(np.float32(v)).hex()

if __name__ == "__main__":
    # exp(r) = 1 + r + r^2 * P(r), r in [-ln2/2, ln2/2]
    L = np.log(2.0) / 2
    def g(r):
        r = np.where(np.abs(r) < 1e-9, 1e-9, r)
        return (np.expm1(r) - r) / (r * r)
    cexp = fit(g, -L * 1.0001, L * 1.0001, 5)
    r = np.linspace(-L, L, 200001)
    approx = 1 + r + r * r * np.polyval(cexp[::-1], r)
    print("exp: max rel err (fp64 eval) = %.3e" % np.max(np.abs(approx / np.exp(r) - 1)))
    print("EXP_P = {" + ", ".join(hexf(c) for c in cexp) + "}")
    print("      = ", [float(np.float32(c)) for c in cexp])

    # tanh(x) = x + x*s*Q(s), s = x^2, |x| <= T0
    T0 = 0.625
    def h(s):
        s = np.asarray(s, dtype=np.float64)
        ser = (-1/3 + s*(2/15 + s*(-17/315 + s*(62/2835 + s*(-1382/155925 + s*(21844/6081075
               + s*(-929569/638512875 + s*(6404582/10854718875))))))))
        ss = np.where(s < 0.04, 1.0, s)
        x = np.sqrt(ss)
        return np.where(s < 0.04, ser, (np.tanh(x) / x - 1.0) / ss)
    ctanh = fit(h, 0.0, T0 * T0 * 1.0001, 6)
    x = np.linspace(1e-6, T0, 200001)
    s = x * x
    approx = x + x * s * np.polyval(ctanh[::-1], s)
    print("tanh: max rel err (fp64 eval) = %.3e" % np.max(np.abs(approx / np.tanh(x) - 1)))
    print("TANH_Q = {" + ", ".join(hexf(c) for c in ctanh) + "}")
    print("       = ", [float(np.float32(c)) for c in ctanh])
    ln2 = np.log(2.0)
    hi = np.float32(0.693359375)  # 9 bits: n*hi exact for |n| < 2^15
    lo = np.float32(ln2 - float(hi))
    print("LN2_HI", hexf(hi), "LN2_LO", hexf(lo), "LOG2E", hexf(1 / ln2))
