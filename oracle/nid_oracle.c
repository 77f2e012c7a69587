/*
 * nid_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's CPU NID path.  See nid_oracle.h for
 * the "PARITY UNPINNED" statement and who may load this library.
 *
 * Build: gcc -O3 -march=native -ffp-contract=off -fPIC -shared (oracle/Makefile).
 *   -ffp-contract=off because the reference's g2o is built without -march
 *   flags (build.sh:16 "Relase" -> no CMAKE_CXX_FLAGS_RELEASE), i.e. baseline
 *   x86-64 without FMA: every multiply and add is rounded on its own.
 *
 * Faithfulness notes: Q1 normalisation by the
 * initial-pose count, Q2 no depth-sign test / cols vs cols-1, Q3 (int)
 * truncation in the bilinear sampler, Q4 clamps, Q5 right-closed order-1
 * B-spline intervals, Q6 fx*x/z vs fx*(x/z), Q7 Jacobian reuses the state of
 * the preceding computeError, Q8 float Huber delta^2, Q9 (1+log2 p).
 */
#include "nid_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NID_MAX_BINS 32
#define NID_SIGMA 1e-30 /* types_six_dof_expmap.h:281 */

struct nid_oracle {
  int rows, cols, cell, nb, deg;
  int rb, cb, ncell;
  double fx, fy, cx, cy;
  int jac_bound_mode, xform_mode;
  double knots[NID_MAX_BINS + 4];
  /* per pixel, image order */
  double *pts;          /* 3N, NaN = invalid depth */
  double *I0;           /* N  (_measurement) */
  double *wr;           /* 4N (bs_value_ref_), zero until computeHref */
  double *ic;           /* N  (intensity_current_), zero-initialised: .cpp:652 */
  double *jgx, *jgy, *jpc, *jdw; int *jjc;  /* Jacobian-pass dump: gx, gy, bin position, 4 derivatives, span (NaN / -1: no contribution) */
  double *jabs;  /* per cell: the sum of the ABSOLUTE values of the terms the last linearizeOplus added up (nid_oracle_jac_abs_scale) */
  unsigned char *im1;   /* N  (image1_) */
  /* dump of the last evaluate */
  double *du, *dv, *dwc;
  int *djc;
  /* per cell */
  int *Nc;              /* obs.rows() - ob */
  int *active;
  double *Href;
  double *pc;           /* ncell * nb      (pro_current_) */
  double *pj;           /* ncell * nb * nb (pro_joint_)   */
  double *Hc, *Hj;
  long n_eval_cost, n_eval_jac;
};

/* ------------------------------------------------------------------------ */
/* B-spline: types_six_dof_expmap.cpp:738-764 (value), :766-800 (derivative) */
/* knot vector: types_six_dof_expmap.h:283-296, computeH.cu:99-112           */
/* Pixel order inside a cell.  The reference walks a cell's pixels row-major (NID_pose_estimation.cpp:411-427 builds
 * the edge's point set in that order; types_six_dof_expmap.cpp:549,397 loop over it).  Built with
 * -DNID_ORACLE_REVERSE_PIXELS (oracle/libnid_oracle_rev.so) the same arithmetic runs over the pixels in the
 * opposite order: every f64 sum gets a different rounding, nothing else changes.  Tests use the pair to MEASURE how
 * far the reference's own optimisation is reproducible on given data (tests/test_oracle_known_answers.py). */
#ifdef NID_ORACLE_REVERSE_PIXELS
#define NID_FOR_PIXEL_ROWS(r, lo, hi) for (int r = (hi) - 1; r >= (lo); r--)
#define NID_FOR_PIXEL_COLS(c, lo, hi) for (int c = (hi) - 1; c >= (lo); c--)
#else
#define NID_FOR_PIXEL_ROWS(r, lo, hi) for (int r = (lo); r < (hi); r++)
#define NID_FOR_PIXEL_COLS(c, lo, hi) for (int c = (lo); c < (hi); c++)
#endif

static void make_knots(int nb, double *knots) {
  int S = nb - 3;
  for (int i = 0; i < nb + 4; i++) {
    int k = i - 3;
    if (k < 0) k = 0;
    if (k > S) k = S;
    knots[i] = (double)k;
  }
}

static double bspline(const double *knots, int index, int order, double u) {
  double coef1, coef2;
  if (order == 1) {
    if (index == 0)
      if ((knots[index] <= u) && (u <= knots[index + 1])) return 1.0;
    if ((knots[index] < u) && (u <= knots[index + 1])) return 1.0;
    else return 0.0;
  } else {
    if (knots[index + order - 1] == knots[index]) {
      if (u == knots[index]) coef1 = 1; else coef1 = 0;
    } else
      coef1 = (u - knots[index]) / (knots[index + order - 1] - knots[index]);
    if (knots[index + order] == knots[index + 1]) {
      if (u == knots[index + order]) coef2 = 1; else coef2 = 0;
    } else
      coef2 = (knots[index + order] - u) / (knots[index + order] - knots[index + 1]);
    return (coef1 * bspline(knots, index, order - 1, u) +
            coef2 * bspline(knots, index + 1, order - 1, u));
  }
}

static double bspline_der(const double *knots, int index, int order, double u) {
  double coef1, coef2, coef3, coef4;
  if (order == 1) {
    return 0.0;
  } else {
    if (knots[index + order - 1] == knots[index]) {
      if (u == knots[index]) coef1 = 1; else coef1 = 0;
      coef3 = 0.0;
    } else {
      coef1 = (u - knots[index]) / (knots[index + order - 1] - knots[index]);
      coef3 = 1.0 / (knots[index + order - 1] - knots[index]);
    }
    if (knots[index + order] == knots[index + 1]) {
      if (u == knots[index + order]) coef2 = 1; else coef2 = 0;
      coef4 = 0.0;
    } else {
      coef2 = (knots[index + order] - u) / (knots[index + order] - knots[index + 1]);
      coef4 = -1.0 / (knots[index + order] - knots[index + 1]);
    }
    return (coef1 * bspline_der(knots, index, order - 1, u) +
            coef2 * bspline_der(knots, index + 1, order - 1, u) +
            coef3 * bspline(knots, index, order - 1, u) +
            coef4 * bspline(knots, index + 1, order - 1, u));
  }
}

double nid_oracle_bspline(int bin_num, int index, int order, double u) {
  double knots[NID_MAX_BINS + 4];
  make_knots(bin_num, knots);
  return bspline(knots, index, order, u);
}
double nid_oracle_bspline_der(int bin_num, int index, int order, double u) {
  double knots[NID_MAX_BINS + 4];
  make_knots(bin_num, knots);
  return bspline_der(knots, index, order, u);
}

/* ------------------------------------------------------------------------ */
/* bilinear sampler: types_six_dof_expmap.h:310-328 ((int) truncation, Q3)   */
/* linearizeOplus samples bil(u - 1, v) and bil(u, v - 1) (:434-435).  For u in (0, 1) the (int) truncation toward zero
 * makes that an extrapolation from columns 0 and 1; at u == 0.0 EXACTLY it is (int)(-1.0) = -1: the reference reads
 * im[-1], the element before the row (before the buffer in row 0) -- undefined behaviour, which this restatement
 * reproduces (AddressSanitizer reports it; the result changes from one instance to the next).  Built with
 * -DNID_ORACLE_MARGIN (oracle/libnid_oracle_margin.so) column -1 and row -1 hold the value the extrapolation tends to,
 * 2 I[0] - I[1]: the continuous completion of what the reference computes next to them, and what the HIP path's image
 * margin holds (k_im1_margins).  Tests use it to check the cells the reference leaves undefined. */
#ifdef NID_ORACLE_MARGIN
static double tap_at(const unsigned char *im, int cols, int r, int c) {
  if (r >= 0 && c >= 0) return im[r * cols + c];
  if (r >= 0) return 2.0 * im[r * cols] - im[r * cols + 1];
  if (c >= 0) return 2.0 * im[c] - im[cols + c];
  return 2.0 * (2.0 * im[0] - im[1]) - (2.0 * im[cols] - im[cols + 1]);
}
#define NID_TAP(r, c) tap_at(im, cols, (r), (c))
#else
#define NID_TAP(r, c) ((double)im[(r) * cols + (c)])
#endif
static double bilinear_u8(const unsigned char *im, int cols, double x, double y) {
  int ix = (int)x;
  int iy = (int)y;
  double dx = x - ix;
  double dy = y - iy;
  double dxdy = dx * dy;
  return (double)(dxdy * NID_TAP(iy + 1, ix + 1) +
                  (dy - dxdy) * NID_TAP(iy + 1, ix) +
                  (dx - dxdy) * NID_TAP(iy, ix + 1) +
                  (1 - dx - dy + dxdy) * NID_TAP(iy, ix));
}

/* The reference's own rounding noise, made measurable.  Built with -DNID_ORACLE_TWIN (oracle/libnid_oracle_twin.so,
 * together with the reversed pixel order) every bilinear sample -- the centre sample of the cost pass
 * (types_six_dof_expmap.cpp:567) and the four samples of the image gradient (:434-435) -- uses the mathematically
 * identical two-lerp association instead of the reference's four-term form.  Wherever the reference's result is
 * ROUNDING NOISE the twin returns another rounding of the same quantity: a constant or saturated patch, where the
 * four-term form returns the constant +- an ulp and its central differences are 1e-14-level noise; a lone sample whose
 * constant intensity sits on a knot of the B-spline, where that ulp decides between an exactly symmetric (zero)
 * derivative and 1e-13.  The centre sample is also taken one ulp up in u and v (the reference's u is itself a few
 * roundings away from the exact projection).  |J(oracle) - J(twin)| per cell is the measured width of the reference's own Jacobian; tests
 * allow for it instead of a chosen floor (tests/test_parity_gpu.py).  The reference's DECISIONS are not re-rounded: a
 * centre sample keeps the reference's value when either form puts it at a clamp (ic >= 255 -> 254.999, ic < 0 -> 0,
 * :572-575), so the histograms' discontinuities -- which the HIP path reproduces exactly -- stay where they are and
 * saturated cells get no allowance from it. */
#ifdef NID_ORACLE_TWIN
static double bilinear_grad(const unsigned char *im, int cols, double x, double y) {
  int ix = (int)x;
  int iy = (int)y;
  double dx = x - ix;
  double dy = y - iy;
  double i00 = NID_TAP(iy, ix), i01 = NID_TAP(iy, ix + 1);
  double i10 = NID_TAP(iy + 1, ix), i11 = NID_TAP(iy + 1, ix + 1);
  double top = i00 + dx * (i01 - i00);
  double bot = i10 + dx * (i11 - i10);
  return top + dy * (bot - top);
}
#else
#define bilinear_grad bilinear_u8
#endif

/* ------------------------------------------------------------------------ */
/* SE(3): Eigen closed forms used by se3quat.h (Eigen is un-vendored and
 * un-pinned in the reference: ulp-level parity unpinned)      */
static void quat_rotate(const double *q /*xyzw*/, const double *v, double *out) {
  /* Eigen QuaternionBase::_transformVector: uv = 2*(vec x v);
   * v + w*uv + vec x uv */
  double uvx = q[1] * v[2] - q[2] * v[1];
  double uvy = q[2] * v[0] - q[0] * v[2];
  double uvz = q[0] * v[1] - q[1] * v[0];
  uvx += uvx; uvy += uvy; uvz += uvz;
  double cx_ = q[1] * uvz - q[2] * uvy;
  double cy_ = q[2] * uvx - q[0] * uvz;
  double cz_ = q[0] * uvy - q[1] * uvx;
  out[0] = v[0] + q[3] * uvx + cx_;
  out[1] = v[1] + q[3] * uvy + cy_;
  out[2] = v[2] + q[3] * uvz + cz_;
}

void nid_oracle_se3_map(const double *p, const double *x, double *y) {
  /* se3quat.h:217-220: _r*xyz + _t */
  double r[3];
  quat_rotate(p, x, r);
  y[0] = r[0] + p[4];
  y[1] = r[1] + p[5];
  y[2] = r[2] + p[6];
}

static void quat_normalize_pos(double *q) {
  /* se3quat.h:280-285 normalizeRotation */
  if (q[3] < 0) { q[0] *= -1; q[1] *= -1; q[2] *= -1; q[3] *= -1; }
  double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
  q[0] /= n; q[1] /= n; q[2] /= n; q[3] /= n;
}

static void quat_from_R(const double *m /*row-major 3x3*/, double *q) {
  /* Eigen quaternionbase_assign_impl<Matrix3>::run */
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[2 * 3 + 1] - m[1 * 3 + 2]) * t;
    q[1] = (m[0 * 3 + 2] - m[2 * 3 + 0]) * t;
    q[2] = (m[1 * 3 + 0] - m[0 * 3 + 1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 3 + i]) i = 2;
    int j = (i + 1) % 3;
    int k = (j + 1) % 3;
    t = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
    q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
  }
}

static void quat_to_R(const double *q, double *R /*row-major*/) {
  /* Eigen QuaternionBase::toRotationMatrix */
  double tx = 2 * q[0], ty = 2 * q[1], tz = 2 * q[2];
  double twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
  double txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
  double tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

void nid_oracle_se3_from_Rt(const double *R, const double *t, double *p) {
  /* se3quat.h:58-60 */
  quat_from_R(R, p);
  quat_normalize_pos(p);
  p[4] = t[0]; p[5] = t[1]; p[6] = t[2];
}

void nid_oracle_se3_to_matrix(const double *p, double *M) {
  /* se3quat.h:270-278, column-major like Eigen .data() */
  double R[9];
  quat_to_R(p, R);
  for (int c = 0; c < 3; c++)
    for (int r = 0; r < 3; r++) M[c * 4 + r] = R[r * 3 + c];
  M[3] = M[7] = M[11] = 0.0;
  M[12] = p[4]; M[13] = p[5]; M[14] = p[6]; M[15] = 1.0;
}

void nid_oracle_se3_mul(const double *a, const double *b, double *out) {
  /* se3quat.h:106-112: t = a.t + a.r*b.t ; r = a.r*b.r ; normalize */
  double rt[3], q[4];
  quat_rotate(a, b + 4, rt);
  /* Eigen quaternion product (a*b), coefficient order xyzw */
  q[3] = a[3] * b[3] - a[0] * b[0] - a[1] * b[1] - a[2] * b[2];
  q[0] = a[3] * b[0] + a[0] * b[3] + a[1] * b[2] - a[2] * b[1];
  q[1] = a[3] * b[1] + a[1] * b[3] + a[2] * b[0] - a[0] * b[2];
  q[2] = a[3] * b[2] + a[2] * b[3] + a[0] * b[1] - a[1] * b[0];
  quat_normalize_pos(q);
  out[4] = a[4] + rt[0]; out[5] = a[5] + rt[1]; out[6] = a[6] + rt[2];
  out[0] = q[0]; out[1] = q[1]; out[2] = q[2]; out[3] = q[3];
}

static void mat3_mul(const double *A, const double *B, double *C) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
      C[i * 3 + j] = s;
    }
}

void nid_oracle_se3_exp(const double *upd, double *pose7) {
  /* se3quat.h:223-257: update = (omega, upsilon) */
  double om[3] = {upd[0], upd[1], upd[2]};
  double up[3] = {upd[3], upd[4], upd[5]};
  double theta = sqrt(om[0] * om[0] + om[1] * om[1] + om[2] * om[2]);
  double Om[9] = {0, -om[2], om[1], om[2], 0, -om[0], -om[1], om[0], 0}; /* se3_ops.hpp skew */
  double Om2[9];
  mat3_mul(Om, Om, Om2);
  double R[9], V[9];
  static const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) { R[i] = I3[i] + Om[i] + Om2[i]; V[i] = R[i]; }
  } else {
    double a = sin(theta) / theta;
    double b = (1 - cos(theta)) / (theta * theta);
    double c = (theta - sin(theta)) / (pow(theta, 3));
    for (int i = 0; i < 9; i++) {
      R[i] = I3[i] + a * Om[i] + b * Om2[i];
      V[i] = I3[i] + b * Om[i] + c * Om2[i];
    }
  }
  double t[3];
  for (int i = 0; i < 3; i++) t[i] = V[i * 3] * up[0] + V[i * 3 + 1] * up[1] + V[i * 3 + 2] * up[2];
  /* SE3Quat(Quaterniond(R), t): se3quat.h:62-64 */
  quat_from_R(R, pose7);
  quat_normalize_pos(pose7);
  pose7[4] = t[0]; pose7[5] = t[1]; pose7[6] = t[2];
}

/* ------------------------------------------------------------------------ */
/* dense LDLT 6x6 with diagonal pivoting: linear_solver_dense.h:105-113
 * (Eigen::LDLT; summation order inside Eigen is unpinned)                  */
int nid_oracle_ldlt6_solve(const double *H, const double *b, double *x) {
  enum { n = 6 };
  double A[n][n];
  int tr[n];
  int positive = 1;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) A[i][j] = H[i * n + j];
  for (int k = 0; k < n; k++) {
    int idx = k; double best = fabs(A[k][k]);
    for (int i = k + 1; i < n; i++) if (fabs(A[i][i]) > best) { best = fabs(A[i][i]); idx = i; }
    tr[k] = idx;
    if (idx != k) {
      for (int j = 0; j < k; j++) { double t = A[k][j]; A[k][j] = A[idx][j]; A[idx][j] = t; }
      for (int i = idx + 1; i < n; i++) { double t = A[i][k]; A[i][k] = A[i][idx]; A[i][idx] = t; }
      { double t = A[k][k]; A[k][k] = A[idx][idx]; A[idx][idx] = t; }
      for (int i = k + 1; i < idx; i++) { double t = A[i][k]; A[i][k] = A[idx][i]; A[idx][i] = t; }
    }
    if (k > 0) {
      double temp[n];
      for (int j = 0; j < k; j++) temp[j] = A[j][j] * A[k][j];
      double s = 0;
      for (int j = 0; j < k; j++) s += A[k][j] * temp[j];
      A[k][k] -= s;
      for (int i = k + 1; i < n; i++) {
        double s2 = 0;
        for (int j = 0; j < k; j++) s2 += A[i][j] * temp[j];
        A[i][k] -= s2;
      }
    }
    double akk = A[k][k];
    if (akk < 0) positive = 0;
    if (fabs(akk) > 0) for (int i = k + 1; i < n; i++) A[i][k] /= akk;
  }
  if (!positive) return 0; /* _cholesky.isPositive() false -> solve() returns false, x untouched */
  double y[n];
  for (int i = 0; i < n; i++) y[i] = b[i];
  for (int k = 0; k < n; k++) if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
  for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
  for (int i = 0; i < n; i++) { if (fabs(A[i][i]) > DBL_MIN) y[i] /= A[i][i]; else y[i] = 0; }
  for (int i = n - 1; i >= 0; i--) for (int j = i + 1; j < n; j++) y[i] -= A[j][i] * y[j];
  for (int k = n - 1; k >= 0; k--) if (tr[k] != k) { double t = y[k]; y[k] = y[tr[k]]; y[tr[k]] = t; }
  for (int i = 0; i < n; i++) x[i] = y[i];
  return 1;
}

/* ------------------------------------------------------------------------ */
nid_oracle *nid_oracle_create(int rows, int cols, int cell, int bin_num,
                              double fx, double fy, double cx, double cy) {
  if (bin_num < 4 || bin_num > NID_MAX_BINS || cell < 1 || rows < cell || cols < cell) return NULL;
  nid_oracle *o = (nid_oracle *)calloc(1, sizeof(*o));
  o->rows = rows; o->cols = cols; o->cell = cell; o->nb = bin_num; o->deg = 3;
  o->rb = rows / cell; o->cb = cols / cell; o->ncell = cell * cell;
  o->fx = fx; o->fy = fy; o->cx = cx; o->cy = cy;
  make_knots(bin_num, o->knots);
  size_t N = (size_t)rows * cols;
  o->pts = (double *)calloc(3 * N, sizeof(double));
  o->I0 = (double *)calloc(N, sizeof(double));
  o->wr = (double *)calloc(4 * N, sizeof(double));
  o->ic = (double *)calloc(N, sizeof(double));
  o->jgx = (double *)calloc(N, sizeof(double)); o->jgy = (double *)calloc(N, sizeof(double));
  o->jpc = (double *)calloc(N, sizeof(double)); o->jdw = (double *)calloc(4 * N, sizeof(double));
  o->jjc = (int *)calloc(N, sizeof(int));
  o->jabs = (double *)calloc((size_t)o->ncell, sizeof(double));
  o->im1 = (unsigned char *)calloc(N, 1);
  o->du = (double *)calloc(N, sizeof(double));
  o->dv = (double *)calloc(N, sizeof(double));
  o->dwc = (double *)calloc(4 * N, sizeof(double));
  o->djc = (int *)calloc(N, sizeof(int));
  o->Nc = (int *)calloc(o->ncell, sizeof(int));
  o->active = (int *)calloc(o->ncell, sizeof(int));
  o->Href = (double *)calloc(o->ncell, sizeof(double));
  o->pc = (double *)calloc((size_t)o->ncell * bin_num, sizeof(double));
  o->pj = (double *)calloc((size_t)o->ncell * bin_num * bin_num, sizeof(double));
  o->Hc = (double *)calloc(o->ncell, sizeof(double));
  o->Hj = (double *)calloc(o->ncell, sizeof(double));
  for (size_t i = 0; i < 3 * N; i++) o->pts[i] = NAN;
  return o;
}

void nid_oracle_destroy(nid_oracle *o) {
  if (!o) return;
  free(o->pts); free(o->I0); free(o->wr); free(o->ic); free(o->im1);
  free(o->jgx); free(o->jgy); free(o->jpc); free(o->jdw); free(o->jjc); free(o->jabs);
  free(o->du); free(o->dv); free(o->dwc); free(o->djc);
  free(o->Nc); free(o->active); free(o->Href); free(o->pc); free(o->pj);
  free(o->Hc); free(o->Hj);
  free(o);
}

void nid_oracle_set_options(nid_oracle *o, int jac_bound_mode, int xform_mode) {
  o->jac_bound_mode = jac_bound_mode;
  o->xform_mode = xform_mode;
}

/* CudaPoints3d.cu:5-32 == NID_pose_estimation.cpp:411-427 (same operation
 * order; the CPU routine skips where the kernel writes NaN) */
void nid_oracle_backproject(const double *depth, const double *T, double fx,
                            double fy, double cx, double cy, int rows, int cols,
                            double *pts) {
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) {
      int id = r * cols + c;
      double z = depth[id];
      if (z < 0.01 || z > 100) {
        pts[3 * id] = pts[3 * id + 1] = pts[3 * id + 2] = NAN;
        continue;
      }
      double x0 = z * (c - cx) / fx;
      double y0 = z * (r - cy) / fy;
      pts[3 * id]     = T[0] * x0 + T[4] * y0 + T[8] * z + T[12];
      pts[3 * id + 1] = T[1] * x0 + T[5] * y0 + T[9] * z + T[13];
      pts[3 * id + 2] = T[2] * x0 + T[6] * y0 + T[10] * z + T[14];
    }
}

void nid_oracle_set_reference(nid_oracle *o, const double *points3d, const unsigned char *im0) {
  size_t N = (size_t)o->rows * o->cols;
  memcpy(o->pts, points3d, 3 * N * sizeof(double));
  for (size_t i = 0; i < N; i++) o->I0[i] = (double)im0[i]; /* NID_pose_estimation.cpp:425 */
  memset(o->wr, 0, 4 * N * sizeof(double));                 /* .cpp:650 */
  memset(o->ic, 0, N * sizeof(double));                     /* .cpp:652 */
}

void nid_oracle_clear_intensity(nid_oracle *o) { memset(o->ic, 0, (size_t)o->rows * o->cols * sizeof(double)); }  /* .cpp:652 */

void nid_oracle_set_target(nid_oracle *o, const unsigned char *im1) {
  memcpy(o->im1, im1, (size_t)o->rows * o->cols);
}

/* world -> camera under the candidate pose */
typedef struct { double q[7]; double M[16]; int mode; } xform_t;

static void xform_init(const nid_oracle *o, const double *pose7, xform_t *x) {
  memcpy(x->q, pose7, 7 * sizeof(double));
  nid_oracle_se3_to_matrix(pose7, x->M);
  x->mode = o->xform_mode;
}

static inline void xform_apply(const xform_t *x, const double *p, double *out) {
  if (x->mode == NID_ORACLE_XFORM_QUAT) {
    nid_oracle_se3_map(x->q, p, out);
  } else { /* computeH.cu:152-154 */
    const double *M = x->M;
    out[0] = M[0] * p[0] + M[4] * p[1] + M[8] * p[2] + M[12];
    out[1] = M[1] * p[0] + M[5] * p[1] + M[9] * p[2] + M[13];
    out[2] = M[2] * p[0] + M[6] * p[1] + M[10] * p[2] + M[14];
  }
}

static inline int pixel_valid(const nid_oracle *o, int id) {
  const double *p = o->pts + 3 * (size_t)id;
  return !(isnan(p[0]) || isnan(p[1]) || isnan(p[2]));
}

/* types_six_dof_expmap.cpp:655-725 */
void nid_oracle_compute_href(nid_oracle *o, const double *pose7, int *bs_counter, double *Href) {
  xform_t xf; xform_init(o, pose7, &xf);
  const int nb = o->nb;
  for (int ci = 0; ci < o->cell; ci++)
    for (int cj = 0; cj < o->cell; cj++) {
      int c = cj + o->cell * ci;
      double pro_ref[NID_MAX_BINS] = {0};
      int n = 0, ob = 0;
      NID_FOR_PIXEL_ROWS(r, o->rb * ci, o->rb * (ci + 1))
        NID_FOR_PIXEL_COLS(cc, o->cb * cj, o->cb * (cj + 1)) {
          int id = r * o->cols + cc;
          if (!pixel_valid(o, id)) continue;
          n++;
          double p_c[3];
          xform_apply(&xf, o->pts + 3 * (size_t)id, p_c);
          double u = o->fx * p_c[0] / p_c[2] + o->cx;
          double v = o->fy * p_c[1] / p_c[2] + o->cy;
          if (u >= 0 && u + 3 <= o->cols && v >= 0 && v + 3 <= o->rows) {
            o->ic[id] = bilinear_u8(o->im1, o->cols, u, v);
          } else {
            ob++;
            continue;
          }
          double obs = o->I0[id];
          if (obs >= 255) obs = 254.999;
          if (obs < 0) obs = 0.0;
          double bin_pos_ref = obs * (nb - o->deg) / 255.0;
          int jr = (int)floor(bin_pos_ref);
          for (int k = 0; k < 4; k++) {
            double w = bspline(o->knots, jr + k, o->deg + 1, bin_pos_ref);
            o->wr[4 * (size_t)id + k] = w;
            pro_ref[jr + k] += w;
          }
        }
      o->Nc[c] = n - ob;
      o->Href[c] = 0.0;
      if (n - ob < 300) { /* setLevel(1); GPU-path convention: Href = NaN (CudaComputeHref.cu:206-209) */
        o->active[c] = 0;
        o->Href[c] = NAN;
      } else {
        o->active[c] = 1;
        for (int i = 0; i < nb; i++) pro_ref[i] /= (n - ob);
        for (int i = 0; i < nb; i++) {
          if (pro_ref[i] < NID_SIGMA) continue;
          o->Href[c] -= pro_ref[i] * log2(pro_ref[i]);
        }
      }
      if (bs_counter) bs_counter[c] = o->Nc[c];
      if (Href) Href[c] = o->Href[c];
    }
}

/* ClearPrevH + ComputeH: types_six_dof_expmap.cpp:727-736, 544-637 */
static void cell_compute_h(nid_oracle *o, const xform_t *xf, int ci, int cj) {
  const int nb = o->nb;
  int c = cj + o->cell * ci;
  double *pro_current = o->pc + (size_t)c * nb;
  double *pro_joint = o->pj + (size_t)c * nb * nb;
  memset(pro_current, 0, nb * sizeof(double));
  memset(pro_joint, 0, (size_t)nb * nb * sizeof(double));
  double H_current = 0.0, H_joint = 0.0;
  NID_FOR_PIXEL_ROWS(r, o->rb * ci, o->rb * (ci + 1))
    NID_FOR_PIXEL_COLS(cc, o->cb * cj, o->cb * (cj + 1)) {
      int id = r * o->cols + cc;
      o->du[id] = NAN; o->dv[id] = NAN; o->djc[id] = -1;
      if (!pixel_valid(o, id)) continue;
      double p_c[3];
      xform_apply(xf, o->pts + 3 * (size_t)id, p_c);
      double obs = o->I0[id];
      if (obs >= 255) obs = 254.999;
      if (obs < 0) obs = 0.0;
      double bin_pos_ref = obs * (nb - o->deg) / 255.0;
      int jr = (int)floor(bin_pos_ref);
      double u = o->fx * p_c[0] / p_c[2] + o->cx;
      double v = o->fy * p_c[1] / p_c[2] + o->cy;
      o->du[id] = u; o->dv[id] = v;
      if (u >= 0 && u + 3 <= o->cols && v >= 0 && v + 3 <= o->rows) {
        o->ic[id] = bilinear_u8(o->im1, o->cols, u, v);
#ifdef NID_ORACLE_TWIN
        {
          /* ... one ulp up in u and v (the reference's own u carries a few ulps of rounding from the pose transform
           * and the division; its two functions even associate the projection differently, Q6), as long as that names
           * the same pixel: the centre sample's sensitivity to the last bit of its position -- a sample on a steep edge
           * just below the 255 clamp moves its end-span weight, and a cell that hangs on it, by 1e-9 of itself */
          double u2 = nextafter(u, INFINITY), v2 = nextafter(v, INFINITY);
          if ((int)u2 != (int)u || (int)v2 != (int)v) { u2 = u; v2 = v; }
          double il = bilinear_grad(o->im1, o->cols, u2, v2);
          if (o->ic[id] < 255 && o->ic[id] >= 0 && il < 255 && il >= 0) o->ic[id] = il;
        }
#endif
      } else {
        continue;
      }
      if (o->ic[id] >= 255) o->ic[id] = 254.999;
      if (o->ic[id] < 0) o->ic[id] = 0.0;
      double bin_pos_current = o->ic[id] * (nb - 3.0) / 255.0;
      double bins_index_current = floor(bin_pos_current);
      int jc = (int)bins_index_current;
      o->djc[id] = jc;
      double wc[4];
      for (int k = 0; k < 4; k++) {
        wc[k] = bspline(o->knots, jc + k, o->deg + 1, bin_pos_current);
        o->dwc[4 * (size_t)id + k] = wc[k];
      }
      for (int k = 0; k < 4; k++) pro_current[jc + k] += wc[k];
      const double *wr = o->wr + 4 * (size_t)id;
      for (int m = 0; m < 4; m++)
        for (int n = 0; n < 4; n++) pro_joint[(jr + m) * nb + jc + n] += wr[m] * wc[n];
    }
  int Nc = o->Nc[c];
  if (Nc < 300) { o->Hc[c] = NAN; o->Hj[c] = NAN; return; }
  for (int i = 0; i < nb; i++) pro_current[i] /= Nc;
  for (int i = 0; i < nb * nb; i++) pro_joint[i] /= Nc;
  for (int i = 0; i < nb; i++) {
    if (pro_current[i] < NID_SIGMA) continue;
    H_current -= pro_current[i] * log2(pro_current[i]);
  }
  for (int i = 0; i < nb; i++)
    for (int j = 0; j < nb; j++) {
      double p = pro_joint[i * nb + j];
      if (p < NID_SIGMA) continue;
      H_joint -= p * log2(p);
    }
  o->Hc[c] = H_current;
  o->Hj[c] = H_joint;
}

/* linearizeOplus, CPU branch: types_six_dof_expmap.cpp:383-529 */
static void cell_linearize(nid_oracle *o, const xform_t *xf, int ci, int cj, double *J) {
  const int nb = o->nb;
  int c = cj + o->cell * ci;
  const double *pro_current = o->pc + (size_t)c * nb;
  const double *pro_joint = o->pj + (size_t)c * nb * nb;
  double d_sum_bs_pose[NID_MAX_BINS][6];
  static __thread double d_sum_joint_bs_pose[NID_MAX_BINS][NID_MAX_BINS][6];
  memset(d_sum_bs_pose, 0, sizeof(d_sum_bs_pose));
  memset(d_sum_joint_bs_pose, 0, sizeof(d_sum_joint_bs_pose));
  double d_mi_i = (nb - o->deg) / 255.0;
  const double fx = o->fx, fy = o->fy;
  const int jcols = (o->jac_bound_mode == NID_ORACLE_JACBOUND_CPU) ? o->cols - 1 : o->cols;
  NID_FOR_PIXEL_ROWS(r, o->rb * ci, o->rb * (ci + 1))
    NID_FOR_PIXEL_COLS(cc, o->cb * cj, o->cb * (cj + 1)) {
      int id = r * o->cols + cc;
      o->jgx[id] = NAN; o->jgy[id] = NAN; o->jpc[id] = NAN; o->jjc[id] = -1;
      for (int k = 0; k < 4; k++) o->jdw[4 * (size_t)id + k] = NAN;
      if (!pixel_valid(o, id)) continue;
      double p_c[3];
      xform_apply(xf, o->pts + 3 * (size_t)id, p_c);
      double obs = o->I0[id];
      if (obs >= 255) obs = 254.999;
      if (obs < 0) obs = 0.0;
      double bin_pos_ref = obs * (nb - o->deg) / 255.0;
      int jr = (int)floor(bin_pos_ref);
      double u_c = p_c[0] / p_c[2];
      double v_c = p_c[1] / p_c[2];
      double x = p_c[0], y = p_c[1];
      double invz = 1.0 / p_c[2];
      double invz_2 = invz * invz;
      double u = fx * u_c + o->cx;
      double v = fy * v_c + o->cy;
      double bin_pos_current = o->ic[id] * (nb - 3.0) / 255.0;
      int jc = (int)floor(bin_pos_current);
      double gx, gy, Ju[6], Jv[6];
      if (u >= 0 && u + 3 <= jcols && v >= 0 && v + 3 <= o->rows) {
        gx = (bilinear_grad(o->im1, o->cols, u + 1, v) - bilinear_grad(o->im1, o->cols, u - 1, v)) / 2;
        gy = (bilinear_grad(o->im1, o->cols, u, v + 1) - bilinear_grad(o->im1, o->cols, u, v - 1)) / 2;
        Ju[0] = -x * y * invz_2 * fx;
        Ju[1] = (1 + (x * x * invz_2)) * fx;
        Ju[2] = -y * invz * fx;
        Ju[3] = invz * fx;
        Ju[4] = 0;
        Ju[5] = -x * invz_2 * fx;
        Jv[0] = -(1 + y * y * invz_2) * fy;
        Jv[1] = x * y * invz_2 * fy;
        Jv[2] = x * invz * fy;
        Jv[3] = 0;
        Jv[4] = invz * fy;
        Jv[5] = -y * invz_2 * fy;
      } else {
        continue;
      }
      double d_i_pose[6];
      for (int n = 0; n < 6; n++) d_i_pose[n] = gx * Ju[n] + gy * Jv[n];
      double d_bs_mi[4];
      for (int m = 0; m < 4; m++) d_bs_mi[m] = bspline_der(o->knots, jc + m, o->deg + 1, bin_pos_current);
      o->jgx[id] = gx; o->jgy[id] = gy; o->jpc[id] = bin_pos_current; o->jjc[id] = jc;
      for (int m = 0; m < 4; m++) o->jdw[4 * (size_t)id + m] = d_bs_mi[m];
      for (int m = 0; m < 4; m++)
        for (int n = 0; n < 6; n++) d_sum_bs_pose[jc + m][n] += d_bs_mi[m] * d_mi_i * d_i_pose[n];
      const double *wr = o->wr + 4 * (size_t)id;
      for (int k = 0; k < 4; k++)
        for (int m = 0; m < 4; m++)
          for (int n = 0; n < 6; n++)
            d_sum_joint_bs_pose[jr + k][jc + m][n] += wr[k] * d_bs_mi[m] * d_mi_i * d_i_pose[n];
    }
  int Nc = o->Nc[c];
  for (int m = 0; m < nb; m++)
    for (int n = 0; n < 6; n++) d_sum_bs_pose[m][n] /= Nc;
  for (int m = 0; m < nb; m++)
    for (int n = 0; n < nb; n++)
      for (int k = 0; k < 6; k++) d_sum_joint_bs_pose[m][n][k] /= Nc;
  double d_hj_p[6], d_hl_p[6];
  for (int i = 0; i < 6; i++) {
    double tmp = 0.0;
    for (int m = 0; m < nb; m++)
      for (int n = 0; n < nb; n++) {
        if (pro_joint[m * nb + n] < NID_SIGMA) continue;
        tmp -= (1.0 + log2(pro_joint[m * nb + n])) * d_sum_joint_bs_pose[m][n][i];
      }
    d_hj_p[i] = tmp;
  }
  for (int i = 0; i < 6; i++) {
    d_hl_p[i] = 0.0;
    for (int j = 0; j < nb; j++) {
      if (pro_current[j] < NID_SIGMA) continue;
      d_hl_p[i] -= (1.0 + log2(pro_current[j])) * d_sum_bs_pose[j][i];
    }
  }
  double H_joint = o->Hj[c], H_current = o->Hc[c], H_ref = o->Href[c];
  double inv_square_hj = 1.0 / (H_joint * H_joint);
  for (int i = 0; i < 6; i++)
    J[i] = (d_hj_p[i] * (H_current + H_ref) - d_hl_p[i] * H_joint) * inv_square_hj;
  /* Test infrastructure, not part of the restated path: the CONDITION of that result -- the same expression with every
   * term's absolute value, i.e. the magnitude of what the reference adds up with alternating signs (the bins' derivative
   * sums add up to zero, :505-519, and J is a difference of two products, :521).  A cell whose J is many orders below it
   * is a cancellation residue: no re-association of these sums reproduces it to a relative bound. */
  {
    double worst = 0.0;
    for (int i = 0; i < 6; i++) {
      double aj = 0.0, al = 0.0;
      for (int m = 0; m < nb; m++)
        for (int n = 0; n < nb; n++)
          if (!(pro_joint[m * nb + n] < NID_SIGMA)) aj += fabs((1.0 + log2(pro_joint[m * nb + n])) * d_sum_joint_bs_pose[m][n][i]);
      for (int j = 0; j < nb; j++)
        if (!(pro_current[j] < NID_SIGMA)) al += fabs((1.0 + log2(pro_current[j])) * d_sum_bs_pose[j][i]);
      double t = (aj * fabs(H_current + H_ref) + al * fabs(H_joint)) * inv_square_hj;
      if (t > worst) worst = t;
    }
    o->jabs[c] = worst;
  }
}

void nid_oracle_evaluate(nid_oracle *o, const double *pose7, int want_jac,
                         double *Hc, double *Hj, double *err, double *J6) {
  xform_t xf; xform_init(o, pose7, &xf);
  if (want_jac) o->n_eval_jac++; else o->n_eval_cost++;
  for (int ci = 0; ci < o->cell; ci++)
    for (int cj = 0; cj < o->cell; cj++) {
      int c = cj + o->cell * ci;
      if (!o->active[c]) { /* level-1 edges are never evaluated (Q10) */
        if (Hc) Hc[c] = NAN;
        if (Hj) Hj[c] = NAN;
        if (err) err[c] = NAN;
        if (J6 && want_jac) for (int n = 0; n < 6; n++) J6[6 * c + n] = NAN;
        continue;
      }
      cell_compute_h(o, &xf, ci, cj);
      if (Hc) Hc[c] = o->Hc[c];
      if (Hj) Hj[c] = o->Hj[c];
      /* types_six_dof_expmap.h:227 */
      if (err) err[c] = (2 * o->Hj[c] - o->Href[c] - o->Hc[c]) / o->Hj[c];
      if (want_jac && J6) cell_linearize(o, &xf, ci, cj, J6 + 6 * c);
    }
}

void nid_oracle_dump_pixels(const nid_oracle *o, double *u, double *v, double *ic,
                            int *jc, double *wc4, double *wr4, int *jr) {
  size_t N = (size_t)o->rows * o->cols;
  if (u) memcpy(u, o->du, N * sizeof(double));
  if (v) memcpy(v, o->dv, N * sizeof(double));
  if (ic) memcpy(ic, o->ic, N * sizeof(double));
  if (jc) memcpy(jc, o->djc, N * sizeof(int));
  if (wc4) memcpy(wc4, o->dwc, 4 * N * sizeof(double));
  if (wr4) memcpy(wr4, o->wr, 4 * N * sizeof(double));
  if (jr)
    for (size_t i = 0; i < N; i++) {
      double obs = o->I0[i];
      if (obs >= 255) obs = 254.999;
      jr[i] = (int)floor(obs * (o->nb - o->deg) / 255.0);
    }
}

void nid_oracle_jac_abs_scale(const nid_oracle *o, double *per_cell) {
  memcpy(per_cell, o->jabs, (size_t)o->ncell * sizeof(double));
}

void nid_oracle_dump_jac(const nid_oracle *o, double *gx, double *gy, double *pc, int *jc, double *dw4) {
  size_t N = (size_t)o->rows * o->cols;
  if (gx) memcpy(gx, o->jgx, N * sizeof(double));
  if (gy) memcpy(gy, o->jgy, N * sizeof(double));
  if (pc) memcpy(pc, o->jpc, N * sizeof(double));
  if (jc) memcpy(jc, o->jjc, N * sizeof(int));
  if (dw4) memcpy(dw4, o->jdw, 4 * N * sizeof(double));
}

/* base_unary_edge.hpp:43-72, robust_kernel_impl.cpp:65-91 (float dsqr,
 * robust_kernel_impl.h:84), base_edge.h:58-61,96-102 */
static void huber(double e2, double delta, double *rho0, double *rho1) {
  float dsqr = (float)(delta * delta);
  if (e2 <= dsqr) { *rho0 = e2; *rho1 = 1.; }
  else {
    double sqrte = sqrt(e2);
    *rho0 = 2 * sqrte * delta - dsqr;
    *rho1 = delta / sqrte;
  }
}

void nid_oracle_normal_equations(const double *err, const double *J6, int cells,
                                 double delta, double *H36, double *b6,
                                 double *chi2, int *n_active) {
  double H[36] = {0}, b[6] = {0}, F = 0;
  int na = 0;
  for (int c = 0; c < cells; c++) {
    if (isnan(err[c])) continue;
    na++;
    double e = err[c];
    double e2 = e * (1.0 * e); /* _error.dot(information()*_error) */
    double r0, r1;
    huber(e2, delta, &r0, &r1);
    F += r0;
    if (J6) {
      const double *A = J6 + 6 * c;
      for (int i = 0; i < 6; i++) b[i] -= ((r1 * A[i]) * 1.0) * e;
      for (int i = 0; i < 6; i++)
        for (int j = 0; j < 6; j++) H[i * 6 + j] += (A[i] * (r1 * 1.0)) * A[j];
    }
  }
  if (H36) memcpy(H36, H, sizeof(H));
  if (b6) memcpy(b6, b, sizeof(b));
  if (chi2) *chi2 = F;
  if (n_active) *n_active = na;
}

/* optimization_algorithm_levenberg.cpp:61-225 inside
 * sparse_optimizer.cpp:356-450 (verbose branch re-evaluates the cost) */
int nid_oracle_lm(nid_oracle *o, double *pose7, int iterations, double delta,
                  nid_oracle_lm_rec *trace) {
  const int nc = o->ncell;
  double *err = (double *)malloc(nc * sizeof(double));
  double *J = (double *)malloc(6 * nc * sizeof(double));
  double lambda = -1., ni = 2.;
  int nBad = 0, done = 0;
  const double tau = 1e-5, goodUp = 2. / 3., goodLo = 1. / 3.;
  const int maxTrials = 10;
  for (int it = 0; it < iterations; it++) {
    double H[36], b[6], x[6] = {0}, currentChi, tempChi, iniChi;
    nid_oracle_evaluate(o, pose7, 1, NULL, NULL, err, J);
    nid_oracle_normal_equations(err, J, nc, delta, H, b, &currentChi, NULL);
    tempChi = currentChi; iniChi = currentChi;
    if (it == 0) { /* computeLambdaInit: :227-241 */
      double maxDiag = 0.;
      for (int j = 0; j < 6; j++) maxDiag = fmax(fabs(H[j * 6 + j]), maxDiag);
      lambda = tau * maxDiag; ni = 2; nBad = 0;
    }
    double rho = 0; int qmax = 0;
    do {
      double backup[7]; memcpy(backup, pose7, sizeof(backup)); /* push */
      double Hl[36]; memcpy(Hl, H, sizeof(Hl));
      for (int j = 0; j < 6; j++) Hl[j * 6 + j] += lambda;      /* setLambda */
      int ok2 = nid_oracle_ldlt6_solve(Hl, b, x);
      double upd[7], np[7];
      nid_oracle_se3_exp(x, upd);                               /* oplusImpl: types_six_dof_expmap.h:74-77 */
      nid_oracle_se3_mul(upd, pose7, np);
      memcpy(pose7, np, sizeof(np));
      nid_oracle_evaluate(o, pose7, 0, NULL, NULL, err, NULL);
      nid_oracle_normal_equations(err, NULL, nc, delta, NULL, NULL, &tempChi, NULL);
      if (!ok2) tempChi = DBL_MAX;
      rho = (currentChi - tempChi);
      double scale = 0.;                                        /* computeScale :243-250 */
      for (int j = 0; j < 6; j++) scale += x[j] * (lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && isfinite(tempChi)) {
        double alpha = 1. - pow((2 * rho - 1), 3);
        alpha = fmin(alpha, goodUp);
        double scaleFactor = fmax(goodLo, alpha);
        lambda *= scaleFactor; ni = 2; currentChi = tempChi;    /* discardTop */
      } else {
        lambda *= ni; ni *= 2;
        memcpy(pose7, backup, sizeof(backup));                  /* pop */
      }
      qmax++;
    } while (rho < 0 && qmax < maxTrials);
    int ok = 1;
    if (qmax == maxTrials || rho == 0) ok = 0;                  /* Terminate */
    else {
      if ((iniChi - currentChi) * 1e3 < iniChi) nBad++; else nBad = 0;
      if (nBad >= 3) ok = 0;
    }
    /* verbose branch: sparse_optimizer.cpp:404-441 */
    double chi;
    nid_oracle_evaluate(o, pose7, 0, NULL, NULL, err, NULL);
    nid_oracle_normal_equations(err, NULL, nc, delta, NULL, NULL, &chi, NULL);
    if (trace) {
      trace[it].iteration = it; trace[it].chi2 = chi; trace[it].lambda = lambda;
      trace[it].lm_trials = qmax; trace[it].rho = rho;
      memcpy(trace[it].pose7, pose7, 7 * sizeof(double));
    }
    done = it + 1;
    if (!ok) break;
  }
  free(err); free(J);
  return done;
}

long nid_oracle_eval_count(const nid_oracle *o, int with_jac) {
  return with_jac ? o->n_eval_jac : o->n_eval_cost;
}
