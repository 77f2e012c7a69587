"""The kernels' closed-form B-spline (csrc/nid_bspline.h, compiled for the host
through the nid_bspline4_host hook of the C-ABI) must be BIT-identical to the
reference recursion as restated by the oracle (types_six_dof_expmap.cpp:738-800).
No GPU needed: this runs the host instantiation of the same header."""
import ctypes as C
import math

import numpy as np
import pytest


def _bits(x):
    return np.asarray(x, dtype=np.float64).view(np.uint64)


@pytest.mark.parametrize("nb", [4, 5, 6, 8, 10, 12, 14, 16])
def test_closed_form_equals_recursion_bitwise(capi, oracle, nb):
    S = nb - 3
    rng = np.random.default_rng(nb)
    us = np.concatenate([
        rng.uniform(0, S, 20000),
        np.arange(0, S, 1.0),                                   # knots incl. the u == 0 quirk
        np.nextafter(np.arange(0, S, 1.0), np.inf),              # just right of a knot
        np.nextafter(np.arange(1, S + 1, 1.0), -np.inf),         # just left of a knot
        np.array([254.999 * S / 255.0, 1e-300, 5e-324]),         # clamp value, tiny, denormal
        (rng.integers(0, 255 * 4, 4000) / 4.0) * S / 255.0,      # bilinear-like rational intensities
    ])
    us = us[(us >= 0) & (us < S)]
    lib = oracle.load()
    for u in us:
        j = int(math.floor(u))
        B, D = capi.bspline4_host(u, nb)
        Bo = [lib.nid_oracle_bspline(nb, j + k, 4, float(u)) for k in range(4)]
        Do = [lib.nid_oracle_bspline_der(nb, j + k, 4, float(u)) for k in range(4)]
        # -0.0 vs +0.0 cannot reach any sum; compare values, and bits for non-zeros
        for a, b in zip(list(B) + list(D), Bo + Do):
            if a == 0.0 and b == 0.0:
                continue
            assert _bits(a) == _bits(b), (nb, u, B, Bo, D, Do)


def test_div_small_is_correctly_rounded(capi):
    lib = capi.load()
    rng = np.random.default_rng(7)
    xs = np.concatenate([rng.uniform(-8, 8, 200000), rng.uniform(0, 1e-3, 50000),
                         np.ldexp(rng.uniform(1, 2, 50000), rng.integers(-60, 10, 50000))])
    for d in (1.0, 2.0, 3.0):
        got = np.array([lib.nid_div_small_host(float(x), d) for x in xs[:60000]])
        assert np.array_equal(_bits(got), _bits(xs[:60000] / d)), d


@pytest.mark.parametrize("nb", [4, 6, 8, 10, 14, 16])
def test_fast_mode_polynomial_table(capi, oracle, nb):
    """FAST math: the per-span polynomial form of the basis (built in long double from the knot
    vector) agrees with the reference recursion to a few ulp, including at knots and the u == 0 quirk."""
    S = nb - 3
    rng = np.random.default_rng(100 + nb)
    us = np.concatenate([rng.uniform(0, S, 5000), np.arange(0, S, 1.0), np.nextafter(np.arange(1, S + 1, 1.0), -np.inf)])
    us = us[(us >= 0) & (us < S)]
    lib = oracle.load()
    worst_b = worst_d = 0.0
    for u in us:
        j = int(math.floor(u))
        B, D = capi.bspline4_poly_host(u, nb)
        Bo = np.array([lib.nid_oracle_bspline(nb, j + k, 4, float(u)) for k in range(4)])
        Do = np.array([lib.nid_oracle_bspline_der(nb, j + k, 4, float(u)) for k in range(4)])
        worst_b = max(worst_b, np.abs(B - Bo).max())
        worst_d = max(worst_d, np.abs(D - Do).max())
    assert worst_b < 1e-15 and worst_d < 4e-15, (worst_b, worst_d)
    B, D = capi.bspline4_poly_host(0.0, nb)
    assert list(B) == [1.0, 0.0, 0.0, 0.0] and list(D) == [0.0, 0.0, 0.0, 0.0]


def test_log2_fast_host(capi):
    """FAST math: the entropy fold's short log2 (atanh series) against numpy's, on probabilities down to
    the reference's 1e-30 floor, near 1, at powers of two and across the sqrt(1/2) range split."""
    rng = np.random.default_rng(11)
    xs = np.concatenate([10.0 ** rng.uniform(-30, 0, 20000), 1.0 - 10.0 ** rng.uniform(-16, -1, 2000),
                         np.ldexp(1.0, np.arange(-100, 3)), np.sqrt(0.5) * (1 + rng.uniform(-1e-12, 1e-12, 200)),
                         rng.uniform(0.5, 2.0, 5000)])
    got = np.array([capi.log2_fast_host(x) for x in xs])
    ref = np.log2(xs.astype(np.longdouble)).astype(np.float64)
    err = np.abs(got - ref)
    assert np.all(err <= 2.0 ** -51 * np.maximum(1.0, np.abs(ref))), float(err.max())
    assert capi.log2_fast_host(1.0) == 0.0 and capi.log2_fast_host(0.25) == -2.0


def test_quotient_by_255_from_its_reciprocal(tmp_path):
    """csrc/nid_kernels.hip.h div_255 (round 6): RN(x / 255.0) as q0 = x * RN(1/255), q = fma(fma(-255, q0, x), RN(1/255), q0)
    -- the bin position (ic * S) / 255 of the reference (types_six_dof_expmap.cpp:574) without a division sequence.  IEEE
    arithmetic, so the identity is checked on the host: 10^7 random and structured x against the division, bit for bit."""
    import os, subprocess
    src = tmp_path / "div255.c"
    src.write_text(r"""
#include <math.h>
#include <stdio.h>
#include <string.h>
static double div_255(double x) { const double r = 1.0 / 255.0, q0 = x * r; return fma(fma(-255.0, q0, x), r, q0); }
static int same(double a, double b) { return memcmp(&a, &b, 8) == 0; }
int main(void) {
  unsigned long long s = 88172645463325252ull, bad = 0, n = 0;
  for (int i = 0; i < 10000000; i++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    double x = (double)(s >> 11) / 9007199254740992.0 * 255.0 * 13.0;
    if (i % 5 == 1) x = (double)(s % (255 * 13 * 4)) / 4.0;                       /* bilinear-like rationals */
    if (i % 5 == 2) x = nextafter(255.0 * (double)(s % 14), (s & 1) ? 0.0 : 1e9);  /* next to a multiple of 255 */
    if (i % 5 == 3) x = 254.999 * (double)(1 + s % 13);                           /* the clamp value times S */
    if (i % 5 == 4) x = ldexp(x, -(int)(s % 60));                                 /* small */
    n++;
    if (!same(x / 255.0, div_255(x))) { if (!bad) printf("first: %a\n", x); bad++; }
  }
  printf("%llu of %llu differ\n", bad, n);
  return bad != 0;
}
""")
    exe = tmp_path / "div255"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-mfma", "-o", str(exe), str(src), "-lm"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and r.stdout.strip().endswith("0 of 10000000 differ"), r.stdout
