"""div_by_shared (fsk_common.h): the Welford kernels divide every cell's delta by the iteration number with
q = RN(a * rb), rem = a - q * b (one FMA, exact), result = RN(q + rem * rb), rb = RN(1 / b) — three
full-rate instructions instead of a division per cell. The result must be the IEEE quotient bit for
bit (the reference divides, fastsk_kernel.cpp:108-143). Checked here in exact rational arithmetic for
the operands the kernels see: b an iteration number, a a difference of a count and a running mean."""
import random
from fractions import Fraction as F


def shared_divisor_quotient(a, b):
    rb = 1.0 / b
    q = a * rb
    rem = F(a) - F(b) * F(q)        # what the FMA computes before its rounding
    remd = float(rem)
    assert F(remd) == rem           # ... and that rounding is exact
    return float(F(q) + F(remd) * F(rb))   # int / int true division: correctly rounded


def test_quotient_matches_ieee_division():
    rng = random.Random(7)
    for _ in range(60000):
        b = float(rng.choice([rng.randint(1, 300), rng.randint(1, 2 ** 20), rng.randint(1, 2 ** 31)]))
        kind = rng.random()
        if kind < 0.3:
            a = float(rng.randint(-2 ** 32, 2 ** 32))
        elif kind < 0.6:
            a = rng.uniform(-1, 1) * 2.0 ** rng.randint(-40, 33)
        else:
            a = float(rng.randint(0, 5000)) - rng.uniform(0, 5000)
        assert shared_divisor_quotient(a, b) == a / b, (a, b)


def test_quotient_near_ties_and_exact_multiples():
    for b in range(1, 120):
        for k in range(1, 60):
            for eps in (0.0, 2.0 ** -40, -2.0 ** -40, 2.0 ** -52 * k, 0.5, 1 / 3):
                for a in (float(b * k) + eps, (k + 0.5) * b + eps, -(float(b * k) + eps)):
                    assert shared_divisor_quotient(a, float(b)) == a / b, (a, b)
    assert shared_divisor_quotient(0.0, 7.0) == 0.0
