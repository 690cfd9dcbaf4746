"""Exact check (rational arithmetic) that q1 = RN(q0 + r y), y = RN(1/b), q0 = RN(a y), r = a - b q0 equals the IEEE float32 quotient RN(a/b):
the three-instruction division of csrc/sdf_collision.h:sdf_div.  usage: python scripts/experiments/markstein_division.py"""
import numpy as np
from fractions import Fraction
rng = np.random.RandomState(0)
def rn32(fr):
    # correctly rounded float32 of a Fraction (round-to-nearest-even), normal range assumed
    if fr == 0: return np.float32(0)
    s = 1 if fr > 0 else -1
    a = abs(fr)
    # find exponent e with 2^e <= a < 2^(e+1)
    e = a.numerator.bit_length() - a.denominator.bit_length()
    if Fraction(2)**e > a: e -= 1
    elif Fraction(2)**(e+1) <= a: e += 1
    q = a / Fraction(2)**(e-23)       # in [2^23, 2^24)
    n = q.numerator // q.denominator
    rem = q - n
    if rem > Fraction(1,2) or (rem == Fraction(1,2) and (n & 1)): n += 1
    return np.float32(s * float(Fraction(n) * Fraction(2)**(e-23)))
F = lambda x: Fraction(float(x))
bad = 0
N = 200000
for i in range(N):
    # a = v - c (differences of coordinates ~ +-0.3), b = scale (0.05..0.3); also a wide-range mix
    if i % 2 == 0:
        a = np.float32(rng.uniform(-0.4, 0.4)); b = np.float32(rng.uniform(0.03, 0.4))
    else:
        a = np.float32(rng.standard_normal() * 10.0 ** rng.uniform(-6, 6)); b = np.float32(abs(rng.standard_normal()) * 10.0 ** rng.uniform(-6, 6) + 1e-30)
    if a == 0: continue
    y = rn32(Fraction(1) / F(b))
    q0 = rn32(F(a) * F(y))
    r = rn32(F(a) - F(b) * F(q0))
    q1 = rn32(F(q0) + F(r) * F(y))
    ref = rn32(F(a) / F(b))
    if q1 != ref:
        bad += 1
        if bad < 5: print("mismatch", a, b, q1, ref)
print("checked", N, "mismatches", bad)
