// nid_pose_problem.cpp -- the graph that NID_pose_estimation.cpp builds
// (NID_pose_estimation.cpp:163-366), assembled on the g2o-shaped host API, plus a
// C entry point so that tests can drive the C++ host stack through ctypes.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>
#include <vector>

#include "g2o_min/g2o_min.h"
#include "nid/legacy_ops.h"
#include "nid_pose_problem.h"

static double g_last_optimize_s = 0.0;
double nid_host_last_optimize_seconds(void) { return g_last_optimize_s; }

void nid_host_set_devices(const int32_t *devices, int n, int reduce_rccl) { nid_legacy_set_devices(devices, n, reduce_rccl); }
void nid_host_set_resident(int on) { nid_legacy_set_resident(on); }
void nid_host_set_rank(int device, int rank, int world, const uint8_t *rccl_id128) { nid_legacy_set_rank(device, rank, world, rccl_id128); }

int nid_host_run_lm(const nid_pose_problem *pb, double *pose7_inout, nid_host_lm_record *trace,
                    int max_trace, char *log_buf, int log_cap) {
  if (!pb || !pose7_inout) return -1;
  static const bool step_trace = getenv("NID_LEGACY_TRACE") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto stamp = [&](const char *name) {
    if (!step_trace) return;
    const auto n = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[nid trace] nid_host_run_lm: %s %.1f us\n", name, std::chrono::duration<double, std::micro>(n - t_last).count());
    t_last = n;
  };
  const int rows = pb->rows, cols = pb->cols, cell = pb->cell_num, bin_num = pb->bin_num, bs_degree = 3;
  const size_t N = (size_t)rows * cols;
  // (the operators' device state is keyed on geometry and content: a pair of the geometry of the last one reuses its
  // context -- creating and destroying one per pair was 18 ms of a 24 ms call, profiles/r04_pair_setup.txt)
  nid_legacy_set_jacobian_bound(pb->jac_bound_cuda ? 1 : 0);
  nid_legacy_set_math_mode(pb->strict_math ? 1 : 0);

  // NID_pose_estimation.cpp:229-251 -- buffers owned by the caller of the operators (kept from call to call: 30 MB of
  // fresh pages per pair were 0.4 ms of page faults)
  std::vector<double> intrinscis = {pb->fx, pb->fy, pb->cx, pb->cy, pb->depth_factor};
  static thread_local std::vector<double> bs_value, points_3d_all, im0_data, im1_data, depth;
  static thread_local std::vector<int> bin_index;
  // The fused flows read none of the operators' per-pixel arrays: the pair goes to the device in the driver's own formats
  // (round 6; NID_HOST_LEGACY_SETUP=1 or pb->legacy_setup: the reference main()'s route, as before)
  static const bool env_legacy_setup = getenv("NID_HOST_LEGACY_SETUP") != nullptr;
  const bool native = pb->fused != 0 && !pb->legacy_setup && !env_legacy_setup;
  std::vector<double> Href(cell * cell, 0.0);
  std::vector<int> bs_counter(cell * cell);
  std::vector<double> T_wc0(pb->T_wc0_colmajor, pb->T_wc0_colmajor + 16);
  if (!native) {
    bs_value.resize(4 * N); points_3d_all.resize(3 * N); im0_data.resize(N); im1_data.resize(N); depth.resize(N); bin_index.resize(N);
    for (size_t i = 0; i < N; i++) { im0_data[i] = (double)pb->im0[i]; im1_data[i] = (double)pb->im1[i]; }
    for (size_t i = 0; i < N; i++) depth[i] = (double)pb->depth_u16[i] * pb->depth_factor;  // convertTo(CV_64F, 1/5000), :106
    stamp("caller's buffers (u8 / u16 -> f64, allocations)");
  }

  g2o::SparseOptimizer optimizer;
  g2o::BlockSolver_6_X::LinearSolverType *linearSolver = new g2o::LinearSolverDense();
  g2o::BlockSolver_6_X *solver_ptr = new g2o::BlockSolver_6_X(linearSolver);
  g2o::OptimizationAlgorithmLevenberg *solver = new g2o::OptimizationAlgorithmLevenberg(solver_ptr);
  solver->setFusedNormalEquations(pb->fused != 0);
  solver->setSpeculativeTrials(pb->fused == 2 || pb->fused == 3);
  solver->setSpeculativeJacobian(pb->fused >= 3);
  optimizer.setAlgorithm(solver);
  optimizer.setVerbose(true);
  optimizer.setComputeBatchStatistics(true);
  std::ostringstream log;
  optimizer.setLogStream(&log);

  g2o::VertexSE3Expmap *vSE3 = new g2o::VertexSE3Expmap();
  vSE3->setEstimate(g2o::SE3Quat::fromPose7(pose7_inout));
  vSE3->setId(0);
  vSE3->setFixed(false);
  optimizer.addVertex(vSE3);

  // :253, :257
  stamp("optimizer, solver, vertex");
  g2o::Matrix4d M0 = vSE3->estimate().to_homogeneous_matrix();
  nid_multi *native_pair = nullptr;
  if (native) {
    // what :253 and :257 compute, on the device: 1.2 MB up at 640x480, the per-cell counts and Href back
    native_pair = nid_legacy_set_pair_u16(pb->depth_u16, pb->im0, pb->im1, T_wc0.data(), M0.data(), intrinscis.data(), bin_num,
                                          bs_degree, cell, rows, cols, bs_counter.data(), Href.data());
    stamp("nid_legacy_set_pair_u16 (uploads, back-projection, tiles, margins, reference stage)");
  } else {
    Calculate3Dpoint(depth.data(), T_wc0.data(), points_3d_all.data(), intrinscis.data(), rows, cols);
    stamp("Calculate3Dpoint");
    CudaComputeHref(im0_data.data(), points_3d_all.data(), M0.data(), intrinscis.data(), bin_num, bs_degree, cell,
                    rows, cols, bs_value.data(), bin_index.data(), bs_counter.data(), Href.data());
    stamp("CudaComputeHref");
  }
  if (native ? !native_pair : !nid_legacy_multi()) {  // the operators print and carry on like the reference's (computeH.cu:454-473); a run must not
    if (log_buf && log_cap > 0) std::snprintf(log_buf, (size_t)log_cap, "the NID operators could not set up their device state (see stderr)");
    nid_legacy_reset();
    return -3;
  }
  // :264-276
  optimizer.im0_ = im0_data.data(); optimizer.im1_ = im1_data.data(); optimizer.points3d_ = points_3d_all.data();
  optimizer.rows_ = rows; optimizer.cols_ = cols; optimizer.camera_intrincis_ = intrinscis.data();
  optimizer.bin_num_ = bin_num; optimizer.bs_degree_ = bs_degree; optimizer.cell_num_ = cell;
  optimizer.bs_counter_ = bs_counter.data(); optimizer.bs_value_ref_ = bs_value.data();
  optimizer.bs_index_ref_ = bin_index.data(); optimizer.Href_ = Href.data();
  optimizer.native_pair_ = native_pair;

  const double deltaNID = pb->huber_delta > 0 ? pb->huber_delta : std::sqrt(0.95);  // :279
  for (int i = 0; i < cell; i++)
    for (int j = 0; j < cell; j++) {  // :283-337 (use_gpu branch)
      g2o::EdgeSE3ProjectIntensityOnlyPoseNID *e = new g2o::EdgeSE3ProjectIntensityOnlyPoseNID();
      e->setVertex(0, optimizer.vertex(0));
      e->use_CPU_ = false;
      e->set_bspline_relates(bs_degree, bin_num);
      g2o::RobustKernelHuber *rk = new g2o::RobustKernelHuber;
      e->setRobustKernel(rk);
      rk->setDelta(deltaNID);
      e->setInformation(1.0);
      if (std::isnan(Href[j + cell * i])) e->setLevel(1);
      else e->set_href(Href[j + cell * i]);
      optimizer.addEdge(e);
    }

  optimizer.initializeOptimization(0);
  stamp("edges, initializeOptimization");
  const auto t_opt0 = std::chrono::steady_clock::now();
  const int done = optimizer.optimize(pb->iterations);  // :349-350
  g_last_optimize_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_opt0).count();
  stamp("optimize()");

  vSE3->estimate().toPose7(pose7_inout);
  const std::vector<g2o::IterationRecord> &tr = optimizer.trace();
  for (int k = 0; k < (int)tr.size() && k < max_trace; k++) {
    trace[k].iteration = tr[k].iteration; trace[k].chi2 = tr[k].chi2; trace[k].lambda = tr[k].lambda;
    trace[k].rho = tr[k].rho; trace[k].lm_trials = tr[k].levenbergIter;
    std::memcpy(trace[k].pose7, tr[k].pose7, sizeof(trace[k].pose7));
    trace[k].time_s = k < (int)optimizer.batchStatistics().size() ? optimizer.batchStatistics()[k].timeIteration : 0.0;
  }
  if (log_buf && log_cap > 0) {
    const std::string s = log.str();
    std::snprintf(log_buf, (size_t)log_cap, "%s", s.c_str());
  }
  nid_legacy_quiesce();  // (a resident kernel leaves the device; the context stays for the next pair)
  stamp("trace copy, nid_legacy_quiesce");
  return done;
}

void nid_host_se3_exp(const double *upd6, double *pose7) {
  g2o::Vector6d u;
  for (int i = 0; i < 6; i++) u[i] = upd6[i];
  g2o::SE3Quat::exp(u).toPose7(pose7);
}

void nid_host_se3_mul(const double *a7, const double *b7, double *out7) {
  (g2o::SE3Quat::fromPose7(a7) * g2o::SE3Quat::fromPose7(b7)).toPose7(out7);
}

void nid_host_se3_to_matrix(const double *pose7, double *M16) {
  g2o::Matrix4d M = g2o::SE3Quat::fromPose7(pose7).to_homogeneous_matrix();
  std::memcpy(M16, M.data(), 16 * sizeof(double));
}

int nid_host_ldlt6_solve(const double *H36, const double *b6, double *x6) {
  g2o::LinearSolverDense ls;
  return ls.solve(H36, x6, b6) ? 1 : 0;
}

void nid_host_minimal_vector(const double *pose7, double *v6) {
  g2o::Vector6d v = g2o::SE3Quat::fromPose7(pose7).toMinimalVector();
  for (int i = 0; i < 6; i++) v6[i] = v[i];
}

void nid_host_huber(double e2, double delta, double *rho3) {
  g2o::RobustKernelHuber rk;
  rk.setDelta(delta);
  rk.robustify(e2, rho3);
}

// ---- C shims of the three legacy C++ operators (tests drive them through ctypes) --------
void nid_legacy_call_Calculate3Dpoint(double *depth, double *pose_c2w, double *points_3d, double *intr, int rows,
                                      int cols) {
  Calculate3Dpoint(depth, pose_c2w, points_3d, intr, rows, cols);
}
void nid_legacy_call_CudaComputeHref(double *im0, double *points3d, double *pose, double *intr, int bin_num,
                                     int bs_degree, int cell_num, int rows, int cols, double *bs_value,
                                     int *bs_index, int *bs_counter, double *Href) {
  CudaComputeHref(im0, points3d, pose, intr, bin_num, bs_degree, cell_num, rows, cols, bs_value, bs_index, bs_counter,
                  Href);
}
void nid_legacy_call_CudaComputeH(int calculate_der, double *im0, double *im1, double *points3d, int *bs_counter,
                                  double *bs_ref, int *bs_index_ref, double *pose, double *intr, int bin_num,
                                  int bs_degree, int cell_num, int rows, int cols, double *Href, double *Htarget,
                                  double *Hjoint, double *der) {
  g2o::CudaComputeH(calculate_der != 0, im0, im1, points3d, bs_counter, bs_ref, bs_index_ref, pose, intr, bin_num,
                    bs_degree, cell_num, rows, cols, Href, nullptr, nullptr, Htarget, Hjoint, der);
}
