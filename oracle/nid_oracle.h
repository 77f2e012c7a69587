/*
 * nid_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's CPU NID path (the g2o edge
 * EdgeSE3ProjectIntensityOnlyPoseNID and the host pieces around it).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (libnid_hip.so) never links or calls it.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors
 * for this path and the reference itself cannot be built
 * in this image (needs nvcc + CUDA runtime, Eigen 3, OpenCV 3 -- none present;
 * see DESIGN.md "Oracle").  The restatement is pinned only by analytic known
 * answers (tests/test_oracle_known_answers.py) and by the B-spline table that
 * the survey obtained from the verbatim recursion.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * the reference checkout).
 */
#ifndef NID_ORACLE_H
#define NID_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nid_oracle nid_oracle;

/* pose7 = { qx, qy, qz, qw, tx, ty, tz }  (g2o::SE3Quat: _r, _t) */

/* ---- options ---------------------------------------------------------- */
#define NID_ORACLE_JACBOUND_CPU  0 /* u+3 <= cols-1, types_six_dof_expmap.cpp:433 */
#define NID_ORACLE_JACBOUND_CUDA 1 /* u+3 <= cols,   computeH.cu:164              */
#define NID_ORACLE_XFORM_QUAT    0 /* SE3Quat::map,  se3quat.h:217-220            */
#define NID_ORACLE_XFORM_MATRIX  1 /* 4x4 col-major, computeH.cu:152-154          */

nid_oracle *nid_oracle_create(int rows, int cols, int cell, int bin_num,
                              double fx, double fy, double cx, double cy);
void nid_oracle_destroy(nid_oracle *o);
void nid_oracle_set_options(nid_oracle *o, int jac_bound_mode, int xform_mode);

/* ---- setup ------------------------------------------------------------ */
/* CudaPoints3d.cu:5-32 / NID_pose_estimation.cpp:401-432.  depth in metres,
 * T_wc0 column-major 4x4, points3d AoS xyz (NaN = invalid depth). */
void nid_oracle_backproject(const double *depth, const double *T_wc0_colmajor,
                            double fx, double fy, double cx, double cy,
                            int rows, int cols, double *points3d);
void nid_oracle_set_reference(nid_oracle *o, const double *points3d,
                              const unsigned char *im0);
void nid_oracle_set_target(nid_oracle *o, const unsigned char *im1);

/* types_six_dof_expmap.cpp:655-725 (computeHref) for every cell.
 * bs_counter[c] = N_c, Href[c] = NaN when N_c < 300. */
void nid_oracle_compute_href(nid_oracle *o, const double *pose7,
                             int *bs_counter, double *Href);

/* ---- per-iteration path ------------------------------------------------ */
/* computeError (types_six_dof_expmap.h:220-228 + .cpp:544-637) and, when
 * want_jac, linearizeOplus (.cpp:381-529) for every cell at `pose7`.
 * Outputs are per cell; inactive cells get NaN.  Any output may be NULL. */
void nid_oracle_evaluate(nid_oracle *o, const double *pose7, int want_jac,
                         double *Hc, double *Hj, double *err, double *J6);

/* base_unary_edge.hpp:43-72 + robust_kernel_impl.cpp:65-91 over active cells
 * in cell-id order.  H36 row-major full 6x6, b6, chi2 = sum rho0. */
void nid_oracle_normal_equations(const double *err, const double *J6, int cells,
                                 double huber_delta, double *H36, double *b6,
                                 double *chi2, int *n_active);

/* per-pixel dump of the last evaluate (for bit-exact per-pixel parity tests):
 * arrays of length rows*cols in image order; pixels not visited hold NaN/-1 */
void nid_oracle_dump_pixels(const nid_oracle *o, double *u, double *v,
                            double *ic, int *jc, double *wc4, double *wr4,
                            int *jr);

/* per-pixel dump of the Jacobian pass of the last evaluate(want_jac): image gradient (gx, gy), bin position,
 * span and the four B-spline derivatives of every pixel that contributed (NaN / -1 elsewhere) */
void nid_oracle_dump_jac(const nid_oracle *o, double *gx, double *gy, double *pc, int *jc, double *dw4);
/* TEST INFRASTRUCTURE: intensity_current_ back to the zeros the edge starts with (types_six_dof_expmap.cpp:652) -- the
 * state in which a pixel that linearizeOplus takes but computeError did not write (Q6: the two project differently, an
 * ulp apart) contributes nothing (B-spline derivative identically 0 at 0, Q5).  The reference's Jacobian depends on what
 * EARLIER calls left in those slots; this pins the history the HIP path's value corresponds to. */
void nid_oracle_clear_intensity(nid_oracle *o);
/* per cell, of the last evaluate with the Jacobian: the sum of the absolute values of the terms linearizeOplus added up
 * (the condition of its result; test infrastructure) */
void nid_oracle_jac_abs_scale(const nid_oracle *o, double *per_cell);

/* ---- B-spline (types_six_dof_expmap.cpp:738-800) ----------------------- */
double nid_oracle_bspline(int bin_num, int index, int order, double u);
double nid_oracle_bspline_der(int bin_num, int index, int order, double u);

/* ---- SE(3) / solver helpers (se3quat.h, linear_solver_dense.h) --------- */
void nid_oracle_se3_from_Rt(const double *R_rowmajor9, const double *t3, double *pose7);
void nid_oracle_se3_exp(const double *upd6, double *pose7);            /* se3quat.h:223-257 */
void nid_oracle_se3_mul(const double *a7, const double *b7, double *out7); /* se3quat.h:106-112 */
void nid_oracle_se3_to_matrix(const double *pose7, double *M16_colmajor);  /* se3quat.h:270-278 */
void nid_oracle_se3_map(const double *pose7, const double *x3, double *y3); /* se3quat.h:217-220 */
int  nid_oracle_ldlt6_solve(const double *H36, const double *b6, double *x6); /* linear_solver_dense.h:105-113 */

/* ---- Levenberg-Marquardt (optimization_algorithm_levenberg.cpp:61-225,
 *      sparse_optimizer.cpp:356-450) ------------------------------------- */
typedef struct {
  int    iteration;
  double chi2;        /* activeRobustChi2 printed by the verbose branch */
  double lambda;
  int    lm_trials;   /* _levenbergIterations */
  double rho;         /* last rho of the inner loop */
  double pose7[7];    /* estimate after this outer iteration */
} nid_oracle_lm_rec;

/* returns number of outer iterations performed; pose7 updated in place */
int nid_oracle_lm(nid_oracle *o, double *pose7, int iterations,
                  double huber_delta, nid_oracle_lm_rec *trace);

/* evaluation counters (for the CPU-baseline timing) */
long nid_oracle_eval_count(const nid_oracle *o, int with_jac);

#ifdef __cplusplus
}
#endif
#endif
