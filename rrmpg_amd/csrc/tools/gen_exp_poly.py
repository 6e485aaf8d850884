#!/usr/bin/env python3
"""Coefficients of the two exponential polynomials of fastmath.h, by
interpolation in the Chebyshev nodes of the interval (near-minimax; what is
left after rounding the coefficients to double is dominated by that rounding):

  fastpow_tab_core:  2**x = 1 + x Q(x),       |x| <= 1/2,      Q of degree 10
                     (the Taylor polynomial of the same accuracy has degree 12)
  fast_tanh_parts:   expm1(r) = r + r^2 q(r),  |r| <= ln2 / 2,  q of degree 10
                     (Taylor: degree 11)

Prints the doubles (highest power first, as the Horner chains consume them)
and the approximation error in units of 2^-53 relative to the function.
Needs mpmath (present in the build image; not needed at run time).
"""
import mpmath as mp

mp.mp.dps = 60
LN2 = mp.log(2)


def cheb_fit(f, a, b, deg):
    n = deg + 1
    nodes = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (2 * k + 1) / (2 * n))
             for k in range(n)]
    A = mp.matrix(n, n)
    y = mp.matrix(n, 1)
    for i, x in enumerate(nodes):
        for j in range(n):
            A[i, j] = x ** j
        y[i] = f(x)
    return [float(c) for c in mp.lu_solve(A, y)]


def max_rel_err(c, build, a, b, target, K=8000):
    e = 0
    for k in range(K + 1):
        x = a + (b - a) * (mp.mpf(k) + mp.mpf("0.37")) / (K + 1)
        p = sum(mp.mpf(cj) * x ** j for j, cj in enumerate(c))
        e = max(e, abs((build(x, p) - target(x)) / target(x)))
    return float(e / mp.mpf(2) ** -53)


def exp2_q(x):
    if abs(x) < mp.mpf(10) ** -30:
        return LN2
    return mp.expm1(x * LN2) / x


def expm1_q(r):
    if abs(r) < mp.mpf(10) ** -25:
        return mp.mpf(1) / 2 + r / 6
    return (mp.expm1(r) - r) / r ** 2


def show(name, c, err):
    print("// %s: approximation error %.3f x 2^-53" % (name, err))
    for j in range(len(c) - 1, -1, -1):
        print("    %r,   // x^%d" % (c[j], j))


def main():
    h = mp.mpf(1) / 2
    c = cheb_fit(exp2_q, -h, h, 10)
    show("2**x = 1 + x Q(x), |x| <= 1/2", c,
         max_rel_err(c, lambda x, p: 1 + x * p, -h, h, lambda x: mp.mpf(2) ** x))
    a = LN2 / 2
    c = cheb_fit(expm1_q, -a, a, 10)
    show("expm1(r) = r + r^2 q(r), |r| <= ln2/2", c,
         max_rel_err(c, lambda r, p: r + r * r * p, -a, a, mp.expm1))


if __name__ == "__main__":
    main()
