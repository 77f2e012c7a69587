// nid_bspline.h -- closed-form (non-recursive) cubic B-spline basis values and
// derivatives on the clamped uniform knot vector knots[i] = clamp(i-3, 0, S),
// S = bin_num - 3, shared by the HIP kernels and a host unit-test hook.
//
// Replaces the exponential recursion of the reference
//   Bspline / BsplineDer          g2o/g2o/types/types_six_dof_expmap.cpp:738-800
//   Bspline0 / BsplineDer0        g2o/g2o/core/computeH.cu:27-89
// with the local Cox-de Boor triangle over the one non-zero order-1 interval.
// Every non-zero intermediate is computed by the SAME expression (same operand
// order, same roundings) as the recursion; the dropped terms are exact zeros,
// so the results are bit-identical to the recursion (tests/test_bspline_host.py
// checks that against the oracle).  Quirks kept: order-1
// intervals are right-closed, so at an interior knot u == m the non-zero
// interval is (m-1, m]; at u == 0 exactly the value is (1,0,0,0) and the
// derivative is (0,0,0,0).
//
// All knot differences on the non-zero path are 1, 2 or 3, so the divisions
// are done by `div_small`: q0 = x * RN(1/d); q = fma(fma(-d, q0, x), RN(1/d), q0)
// (Markstein correction) -- exactly RN(x / d) for d in {1,2,3}, 3 flops
// instead of an IEEE division sequence.
#pragma once

#if defined(__HIPCC__)
#define NID_HD __host__ __device__ __forceinline__
#else
#define NID_HD inline
#endif

#include <math.h>

namespace nid {

// log2 of a normal positive double for the FAST-mode entropy fold: x = 2^e m with m in
// [sqrt(1/2), sqrt(2)), log2(m) = (2/ln 2) atanh(s), s = (m-1)/(m+1), |s| <= 0.1716; ten terms of the
// odd series (truncation < 3e-17 relative to log2(m)).  Absolute error < 3e-16 for |log2 x| < 128
// (tests/test_bspline_host.py); ~35 dependent operations instead of the library's ~150.
NID_HD double log2_fast(double x) {
  int e;
  double m = frexp(x, &e);  // [0.5, 1)
  if (m < 0.70710678118654752440) { m += m; e -= 1; }
#if defined(__HIP_DEVICE_COMPILE__)
  const double d = m + 1.0;  // hardware reciprocal estimate + two Newton steps (~1 ulp)
  double r = __builtin_amdgcn_rcp(d);
  r = fma(r, fma(-d, r, 1.0), r);
  r = fma(r, fma(-d, r, 1.0), r);
  const double s = (m - 1.0) * r;
#else
  const double s = (m - 1.0) / (m + 1.0);
#endif
  const double z = s * s;
  // (2/ln2)/(2k+1), k = 9..0
  double q = 0.15186263588304877;
  q = fma(q, z, 0.16972882833987804);
  q = fma(q, z, 0.19235933878519512);
  q = fma(q, z, 0.22195308321368667);
  q = fma(q, z, 0.2623081892525388);
  q = fma(q, z, 0.3205988979753252);
  q = fma(q, z, 0.4121985831111324);
  q = fma(q, z, 0.5770780163555853);
  q = fma(q, z, 0.9617966939259756);
  q = fma(q, z, 2.8853900817779268);
  return fma(s, q, (double)e);
}

NID_HD double knot(int i, int S) {
  int k = i - 3;
  k = k < 0 ? 0 : k;
  k = k > S ? S : k;
  return (double)k;
}

// RN(1.0 / d), d in {1,2,3}
NID_HD double rcp_small(double d) {
  return (d == 1.0) ? 1.0 : ((d == 2.0) ? 0.5 : (1.0 / 3.0));
}
// RN(x / d) for d in {1.0, 2.0, 3.0}, r = RN(1/d)
NID_HD double div_small_r(double x, double d, double r) {
  const double q0 = x * r;
  const double rem = fma(-d, q0, x);
  return fma(rem, r, q0);
}
NID_HD double div_small(double x, double d) { return div_small_r(x, d, rcp_small(d)); }

// Reciprocal providers for bspline4: by selects (host / setup kernels) or from a
// per-span table of the six RN(1/d) a span needs (hot kernel; table in LDS).
// Table row layout (8 doubles per span jj = j-3): r10 r1m1 r20 r1m2 r2m1 r30 - -
struct RcpSelect {
  NID_HD double get(int, double d) const { return rcp_small(d); }
};
struct RcpTable {
  const double *row;
  NID_HD double get(int e, double) const { return row[e]; }
};
constexpr int kRcpRow = 8;
// denominator e (0..5) of span jj, as used by the table builder
NID_HD double span_denominator(int jj, int e, int S) {
  const int j = jj + 3;
  switch (e) {
    case 0: return knot(j + 1, S) - knot(j, S);
    case 1: return knot(j + 1, S) - knot(j - 1, S);
    case 2: return knot(j + 2, S) - knot(j, S);
    case 3: return knot(j + 1, S) - knot(j - 2, S);
    case 4: return knot(j + 2, S) - knot(j - 1, S);
    default: return knot(j + 3, S) - knot(j, S);
  }
}

// Values B[k] = Bspline(jc+k, 4, u) and derivatives D[k] = BsplineDer(jc+k, 4, u),
// k = 0..3, jc = floor(u), 0 <= u < S.
template <bool WANT_DER, typename Rcp>
NID_HD void bspline4_impl(double u, int jc, int S, const Rcp &rcp, double B[4], double D[4]);

template <bool WANT_DER>
NID_HD void bspline4(double u, int jc, int S, double B[4], double D[4]) {
  bspline4_impl<WANT_DER>(u, jc, S, RcpSelect(), B, D);
}

// hot-kernel form: `rtab` = table of kRcpRow doubles per span (see RcpTable)
template <bool WANT_DER>
NID_HD void bspline4_tab(double u, int jc, int S, const double *rtab, double B[4], double D[4]) {
  const bool on_knot = (u == (double)jc) && (u != 0.0);
  const int jj = on_knot ? jc - 1 : jc;
  RcpTable t{rtab + jj * kRcpRow};
  bspline4_impl<WANT_DER>(u, jc, S, t, B, D);
}

template <bool WANT_DER, typename Rcp>
NID_HD void bspline4_impl(double u, int jc, int S, const Rcp &rcp, double B[4], double D[4]) {
  // u == 0: index==0 closed interval + degenerate knots: value (1,0,0,0), derivative 0.
  // Handled by selects at the end so that the function is branch-free on the device.
  const bool zero = (u == 0.0);
  // the one order-1 function equal to 1: knots[j] < u <= knots[j+1]
  const bool on_knot = (u == (double)jc) && !zero;  // then jc >= 1
  const int j = on_knot ? jc + 2 : jc + 3;
  const double tm2 = knot(j - 2, S), tm1 = knot(j - 1, S), t0 = knot(j, S);
  const double tp1 = knot(j + 1, S), tp2 = knot(j + 2, S), tp3 = knot(j + 3, S);
  // order 2 (indices j-1, j)
  const double d10 = tp1 - t0;                       // == 1
  const double r10 = rcp.get(0, d10);
  const double b2a = div_small_r(tp1 - u, d10, r10);  // B(j-1,2) = c2(j-1,2)
  const double b2b = div_small_r(u - t0, d10, r10);   // B(j,2)   = c1(j,2)
  // order 3 (indices j-2, j-1, j)
  const double d1m1 = tp1 - tm1, d20 = tp2 - t0;
  const double r1m1 = rcp.get(1, d1m1), r20 = rcp.get(2, d20);
  const double c2_jm2_3 = div_small_r(tp1 - u, d1m1, r1m1);
  const double c1_jm1_3 = div_small_r(u - tm1, d1m1, r1m1);
  const double c2_jm1_3 = div_small_r(tp2 - u, d20, r20);
  const double c1_j_3 = div_small_r(u - t0, d20, r20);
  const double b3a = c2_jm2_3 * b2a;                       // B(j-2,3)
  const double b3b = c1_jm1_3 * b2a + c2_jm1_3 * b2b;      // B(j-1,3)
  const double b3c = c1_j_3 * b2b;                         // B(j,3)
  // order 4 (indices j-3 .. j)
  const double d1m2 = tp1 - tm2, d2m1 = tp2 - tm1, d30 = tp3 - t0;
  const double r1m2 = rcp.get(3, d1m2), r2m1 = rcp.get(4, d2m1), r30 = rcp.get(5, d30);
  const double c2_jm3_4 = div_small_r(tp1 - u, d1m2, r1m2);
  const double c1_jm2_4 = div_small_r(u - tm2, d1m2, r1m2);
  const double c2_jm2_4 = div_small_r(tp2 - u, d2m1, r2m1);
  const double c1_jm1_4 = div_small_r(u - tm1, d2m1, r2m1);
  const double c2_jm1_4 = div_small_r(tp3 - u, d30, r30);
  const double c1_j_4 = div_small_r(u - t0, d30, r30);
  const double n0 = c2_jm3_4 * b3a;
  const double n1 = c1_jm2_4 * b3a + c2_jm2_4 * b3b;
  const double n2 = c1_jm1_4 * b3b + c2_jm1_4 * b3c;
  const double n3 = c1_j_4 * b3c;
  double e0 = 0, e1 = 0, e2 = 0, e3 = 0;
  if (WANT_DER) {
    // order 2
    const double d2a = -r10;   // D(j-1,2) = c4(j-1,2)
    const double d2b = r10;    // D(j,2)   = c3(j,2)
    // order 3: ((c1*Da + c2*Db) + c3*Ba) + c4*Bb, zero terms dropped
    const double d3a = c2_jm2_3 * d2a + (-r1m1) * b2a;                                  // D(j-2,3)
    const double d3b = ((c1_jm1_3 * d2a + c2_jm1_3 * d2b) + r1m1 * b2a) + (-r20) * b2b;  // D(j-1,3)
    const double d3c = c1_j_3 * d2b + r20 * b2b;                                         // D(j,3)
    // order 4
    e0 = c2_jm3_4 * d3a + (-r1m2) * b3a;
    e1 = ((c1_jm2_4 * d3a + c2_jm2_4 * d3b) + r1m2 * b3a) + (-r2m1) * b3b;
    e2 = ((c1_jm1_4 * d3b + c2_jm1_4 * d3c) + r2m1 * b3b) + (-r30) * b3c;
    e3 = c1_j_4 * d3c + r30 * b3c;
  }
  // on a knot the evaluated indices jc..jc+3 are j-2..j+1 ; index j+1 is identically 0
  B[0] = zero ? 1.0 : (on_knot ? n1 : n0);
  B[1] = zero ? 0.0 : (on_knot ? n2 : n1);
  B[2] = zero ? 0.0 : (on_knot ? n3 : n2);
  B[3] = (zero || on_knot) ? 0.0 : n3;
  if (WANT_DER) {
    D[0] = zero ? 0.0 : (on_knot ? e1 : e0);
    D[1] = zero ? 0.0 : (on_knot ? e2 : e1);
    D[2] = zero ? 0.0 : (on_knot ? e3 : e2);
    D[3] = (zero || on_knot) ? 0.0 : e3;
  }
}

}  // namespace nid
