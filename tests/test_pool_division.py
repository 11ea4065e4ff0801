"""The average pool's division (csrc/conv_body.h: avgpool_body).  The reference's kernel adds f / rf term by term
(libs/spconv/src/spconv/avgpool.cu:130); the HIP kernel computes that quotient as
    r = RN(1 / d);  q0 = RN(f * r);  e = fma(-q0, d, f);  q = fma(e, r, q0)
for 2^-100 <= |f| <= 2^100 (anything else takes the division instruction sequence itself).  Scaling f by a power of two scales
every intermediate by the same power as long as nothing leaves the normal range, so walking all 2^23 significands of one
binade, for every divisor of a 3^3 window, is the whole proof that q == RN(f / d) bit for bit.  Pure numpy: the float32
product and quotient are IEEE operations; the two FMAs are evaluated in float64, where the products are exact, and any case
in which the float64 sum could have hidden a double rounding is redone in rationals."""
from fractions import Fraction

import numpy as np
import pytest


def markstein_quotient(f, d):
    d32 = np.float32(d)
    r = np.float32(1.0) / d32
    q0 = f * r                                                  # float32 RN
    e = f.astype(np.float64) - q0.astype(np.float64) * np.float64(d)        # exact: 24 + 5 bits, then a short difference
    assert np.array_equal(e, e.astype(np.float32).astype(np.float64))      # ... and representable: the FMA returns it
    s = q0.astype(np.float64) + e * np.float64(r)               # e * r exact (<= 6 + 24 bits); the sum rounds to 53 bits
    q = s.astype(np.float32)
    # double rounding can only bite where the float64 sum sits exactly on a float32 midpoint: redo those exactly
    up = np.nextafter(q, np.float32(np.inf)).astype(np.float64)
    dn = np.nextafter(q, np.float32(-np.inf)).astype(np.float64)
    tie = (s == (q.astype(np.float64) + up) / 2) | (s == (q.astype(np.float64) + dn) / 2)
    for i in np.flatnonzero(tie):
        exact = Fraction(float(q0[i])) + Fraction(float(e[i])) * Fraction(float(r))
        cands = sorted({float(dn[i]), float(q[i]), float(up[i])}, key=lambda c: (abs(Fraction(c) - exact), int(np.float32(c).view(np.uint32)) & 1))
        q[i] = np.float32(cands[0])
    return q


@pytest.mark.parametrize("d", range(1, 28))
def test_three_instruction_quotient_is_the_ieee_quotient_for_every_significand(d):
    f = np.arange(1 << 23, 1 << 24, dtype=np.uint32).astype(np.float32)      # all significands of one binade, exactly
    want = f / np.float32(d)
    got = markstein_quotient(f, d)
    bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, "d=%d: %d significands differ, first f=%r: %r vs %r" % (d, bad.size, f[bad[:1]], got[bad[:1]], want[bad[:1]])


def test_quotient_at_the_guard_exponents_and_signs():
    rng = np.random.default_rng(5)
    m = rng.integers(1 << 23, 1 << 24, size=200000, dtype=np.uint32).astype(np.float32)
    for ex in (-100 - 23, 100 - 24, -23, 0):                    # |f| at 2^-100, just under 2^100, and ordinary values
        for sign in (1.0, -1.0):
            f = np.ldexp(m, ex).astype(np.float32) * np.float32(sign)
            assert np.all(np.isfinite(f))
            for d in (3, 7, 11, 13, 19, 23, 27):
                got, want = markstein_quotient(f, d), f / np.float32(d)
                assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (ex, sign, d)
    z = np.zeros(4, np.float32)
    assert np.array_equal(markstein_quotient(z, 5), z)
