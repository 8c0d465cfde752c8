#!/usr/bin/env python3
"""Minimax-ish (Chebyshev interpolation, then relative-error check) polynomial for 2^f on
[-1/2, 1/2]; prints Horner coefficients (highest first) and the max relative error of the
double-rounded polynomial.  usage: tools/exp2_poly.py [degree]"""
import sys
import mpmath as mp
mp.mp.dps = 60
deg = int(sys.argv[1]) if len(sys.argv) > 1 else 9
a = mp.mpf(1) / 2
# Chebyshev nodes of the first kind on [-a, a]; interpolate 2^x
n = deg + 1
xs = [a * mp.cos(mp.pi * (2 * k + 1) / (2 * n)) for k in range(n)]
ys = [mp.mpf(2) ** x for x in xs]
V = mp.matrix(n, n)
for i, x in enumerate(xs):
    for j in range(n):
        V[i, j] = x ** j
c = mp.lu_solve(V, mp.matrix(ys))
c = [mp.mpf(float(ci)) for ci in c]  # round to double
pass  # c[0] is left as interpolated: forcing 1 would triple the error
def p(x):
    r = mp.mpf(0)
    for ci in reversed(c):
        r = r * x + ci
    return r
worst = 0
for i in range(-4000, 4001):
    x = a * i / 4000
    e = abs(p(x) / (mp.mpf(2) ** x) - 1)
    worst = max(worst, e)
print("degree", deg, "max rel err", mp.nstr(worst, 5))
for ci in reversed(c):
    print("  %.20e," % float(ci))
