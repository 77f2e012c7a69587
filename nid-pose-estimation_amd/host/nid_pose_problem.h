// nid_pose_problem.h -- C entry points of libnid_host.so: the reference driver's
// optimisation (NID_pose_estimation.cpp:163-366) on the g2o-shaped host API, and the
// small host-only routines (SE(3), LDLT, Huber) for CPU unit tests.
#ifndef NID_POSE_PROBLEM_H
#define NID_POSE_PROBLEM_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
  int32_t rows, cols, cell_num, bin_num, iterations;
  int32_t jac_bound_cuda;   /* 0 = CPU-edge bound (parity target), 1 = CUDA-kernel bound */
  int32_t fused;            /* 0 = per-edge walk like the reference, 1 = fused device normal equations,
                               2 = fused + the LM rejection chain evaluated in one batched launch,
                               3 = 2 + the first trials evaluated with their Jacobian (one launch per outer iteration),
                               4 = 1 + the first trial of every outer iteration evaluated with its Jacobian: nothing but
                                   single-pose evaluations, one per outer iteration in the steady state (the flow the
                                   resident evaluator -- nid_legacy_set_resident -- is made for) */
  int32_t strict_math;      /* 1 = NID_MATH_STRICT, 0 = NID_MATH_FAST */
  int32_t legacy_setup;     /* fused != 0 only.  0 (default): the pair is set up through the C-ABI in the driver's own formats
                               (nid_legacy_set_pair_u16: u16 depth + u8 images up, counts and Href back -- nothing else crosses
                               PCIe); 1: through Calculate3Dpoint / CudaComputeHref like the reference's main() (what fused == 0
                               always does: the per-edge flow calls CudaComputeH with the operators' arrays).  Same pose bits. */
  double fx, fy, cx, cy, depth_factor, huber_delta;
  const uint8_t *im0, *im1;        /* rows*cols */
  const uint16_t *depth_u16;       /* rows*cols, metres = value * depth_factor */
  const double *T_wc0_colmajor;    /* 16 */
} nid_pose_problem;

typedef struct {
  int32_t iteration, lm_trials;
  double chi2, lambda, rho;
  double pose7[7];
  double time_s;            /* wall time of the outer iteration (G2OBatchStatistics::timeIteration) */
} nid_host_lm_record;

/* returns the number of outer iterations done (or < 0); pose7 = {qx,qy,qz,qw,tx,ty,tz} in/out */
int nid_host_run_lm(const nid_pose_problem *pb, double *pose7_inout, nid_host_lm_record *trace, int max_trace,
                    char *log_buf, int log_cap);

/* PNG input of the reference's driver (host/nid_png.cpp; zlib only).  Return 0, or -1 cannot open, -2 not a
 * PNG / corrupt, -3 unsupported variant, -4 inflate failed, -5 buffer too small.  With out == NULL only the
 * size is returned.  swap_rb = 0 reproduces the driver's imread(UNCHANGED) + CV_RGB2GRAY (blue weighted as red). */
int nid_png_info(const char *path, int *rows, int *cols, int *channels, int *bit_depth);
int nid_png_read_gray_u8(const char *path, int swap_rb, int *rows, int *cols, uint8_t *out, size_t cap);
int nid_png_read_u16(const char *path, int *rows, int *cols, uint16_t *out, size_t cap);

/* The reference's second program (NID_standard_property.cpp:150-201) on the HIP library: plain-histogram NID of
 * every cell at pose7 (T_cw1), printed per cell in the reference's format into log_buf; returns 0 and
 * *final_nid = sqrt(sum nid^2), < 0 on failure.  pb->bin_num = number of hard bins (8 in the reference). */
int nid_host_standard_property(const nid_pose_problem *pb, const double *pose7, double *final_nid, char *log_buf,
                               int log_cap);

/* Coarse-to-fine schedule (own definition, see nid_pyramid.cpp): `levels`
 * pyramid levels, `pb->iterations` LM iterations on each from the coarsest to level 0 (= *pb).  trace holds
 * levels * max_trace_per_level records (coarsest level first), done_per_level[levels] the iterations done.
 * Returns the total number of outer iterations, < 0 on failure (-2: sizes not divisible). */
int nid_host_run_pyramid_lm(const nid_pose_problem *pb, int levels, double *pose7_inout, nid_host_lm_record *trace,
                            int max_trace_per_level, int *done_per_level, char *log_buf, int log_cap);
void nid_pyr_down_u8(const uint8_t *src, int rows, int cols, uint8_t *dst);
void nid_pyr_down_depth_u16(const uint16_t *src, int rows, int cols, double depth_factor, uint16_t *dst);

/* Multi-GPU (include/nid/nid_multi.h, include/nid/legacy_ops.h): every later nid_host_run_lm /
 * nid_host_run_pyramid_lm shards the cells of each frame pair (each pyramid level) over these devices of this
 * process (entries may repeat; reduce_rccl: sum the 6x6 blocks with RCCL instead of on the host) ... */
void nid_host_set_devices(const int32_t *devices, int n, int reduce_rccl);
/* nid_legacy_set_resident (include/nid/legacy_ops.h) for C callers */
void nid_host_set_resident(int on);
/* ... or runs as rank `rank` of `world` processes, one per GPU, summing with RCCL (id: nid_multi_comm_unique_id
 * of rank 0).  Every rank runs the same optimisation and takes the same decisions. */
void nid_host_set_rank(int device, int rank, int world, const uint8_t *rccl_id128);

/* wall time of the optimize() call of the last nid_host_run_lm (setup excluded) */
double nid_host_last_optimize_seconds(void);

void nid_host_se3_exp(const double *upd6, double *pose7);
void nid_host_se3_mul(const double *a7, const double *b7, double *out7);
void nid_host_se3_to_matrix(const double *pose7, double *M16_colmajor);
int nid_host_ldlt6_solve(const double *H36, const double *b6, double *x6);
void nid_host_minimal_vector(const double *pose7, double *v6);
void nid_host_huber(double e2, double delta, double *rho3);

/* C shims of Calculate3Dpoint / CudaComputeHref / g2o::CudaComputeH (include/nid/legacy_ops.h) */
void nid_legacy_call_Calculate3Dpoint(double *depth, double *pose_c2w, double *points_3d, double *intr, int rows,
                                      int cols);
void nid_legacy_call_CudaComputeHref(double *im0, double *points3d, double *pose, double *intr, int bin_num,
                                     int bs_degree, int cell_num, int rows, int cols, double *bs_value,
                                     int *bs_index, int *bs_counter, double *Href);
void nid_legacy_call_CudaComputeH(int calculate_der, double *im0, double *im1, double *points3d, int *bs_counter,
                                  double *bs_ref, int *bs_index_ref, double *pose, double *intr, int bin_num,
                                  int bs_degree, int cell_num, int rows, int cols, double *Href, double *Htarget,
                                  double *Hjoint, double *der);

#ifdef __cplusplus
}
#endif
#endif
