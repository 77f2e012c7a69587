// g2o_min.cpp -- implementation of include/g2o_min/g2o_min.h: the host side that
// stays on the host (north_star: "the g2o solver itself stays on the host").
// Every routine cites the reference code whose behaviour it restates.
#include "g2o_min/g2o_min.h"

#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "nid/legacy_ops.h"
#include "nid/nid_c.h"
#include "nid/nid_multi.h"

namespace g2o {

// ---------------------------------------------------------------------------- algebra
Matrix3d Matrix3d::operator*(const Matrix3d &o) const {
  Matrix3d r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) {
      double s = 0;
      for (int k = 0; k < 3; k++) s += m[i * 3 + k] * o.m[k * 3 + j];
      r.m[i * 3 + j] = s;
    }
  return r;
}
Vector3d Matrix3d::operator*(const Vector3d &x) const {
  Vector3d r;
  for (int i = 0; i < 3; i++) r.v[i] = m[i * 3] * x.v[0] + m[i * 3 + 1] * x.v[1] + m[i * 3 + 2] * x.v[2];
  return r;
}
Matrix3d Matrix3d::transpose() const {
  Matrix3d r;
  for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) r.m[i * 3 + j] = m[j * 3 + i];
  return r;
}

Quaterniond::Quaterniond(const Matrix3d &mat) {
  // Eigen quaternionbase_assign_impl<Matrix3>
  const double *m = mat.m;
  double q[4];
  double t = m[0] + m[4] + m[8];
  if (t > 0) {
    t = std::sqrt(t + 1.0);
    q[3] = 0.5 * t;
    t = 0.5 / t;
    q[0] = (m[2 * 3 + 1] - m[1 * 3 + 2]) * t;
    q[1] = (m[0 * 3 + 2] - m[2 * 3 + 0]) * t;
    q[2] = (m[1 * 3 + 0] - m[0 * 3 + 1]) * t;
  } else {
    int i = 0;
    if (m[4] > m[0]) i = 1;
    if (m[8] > m[i * 3 + i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    t = std::sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    q[i] = 0.5 * t;
    t = 0.5 / t;
    q[3] = (m[k * 3 + j] - m[j * 3 + k]) * t;
    q[j] = (m[j * 3 + i] + m[i * 3 + j]) * t;
    q[k] = (m[k * 3 + i] + m[i * 3 + k]) * t;
  }
  x_ = q[0]; y_ = q[1]; z_ = q[2]; w_ = q[3];
}

Matrix3d Quaterniond::toRotationMatrix() const {
  Matrix3d R;
  const double tx = 2 * x_, ty = 2 * y_, tz = 2 * z_;
  const double twx = tx * w_, twy = ty * w_, twz = tz * w_;
  const double txx = tx * x_, txy = ty * x_, txz = tz * x_;
  const double tyy = ty * y_, tyz = tz * y_, tzz = tz * z_;
  R.m[0] = 1 - (tyy + tzz); R.m[1] = txy - twz;       R.m[2] = txz + twy;
  R.m[3] = txy + twz;       R.m[4] = 1 - (txx + tzz); R.m[5] = tyz - twx;
  R.m[6] = txz - twy;       R.m[7] = tyz + twx;       R.m[8] = 1 - (txx + tyy);
  return R;
}

Vector3d Quaterniond::operator*(const Vector3d &v) const {
  // Eigen QuaternionBase::_transformVector
  double uvx = y_ * v[2] - z_ * v[1];
  double uvy = z_ * v[0] - x_ * v[2];
  double uvz = x_ * v[1] - y_ * v[0];
  uvx += uvx; uvy += uvy; uvz += uvz;
  const double cx = y_ * uvz - z_ * uvy;
  const double cy = z_ * uvx - x_ * uvz;
  const double cz = x_ * uvy - y_ * uvx;
  return Vector3d(v[0] + w_ * uvx + cx, v[1] + w_ * uvy + cy, v[2] + w_ * uvz + cz);
}

Quaterniond Quaterniond::operator*(const Quaterniond &b) const {
  const Quaterniond &a = *this;
  return Quaterniond(a.w_ * b.w_ - a.x_ * b.x_ - a.y_ * b.y_ - a.z_ * b.z_,
                     a.w_ * b.x_ + a.x_ * b.w_ + a.y_ * b.z_ - a.z_ * b.y_,
                     a.w_ * b.y_ + a.y_ * b.w_ + a.z_ * b.x_ - a.x_ * b.z_,
                     a.w_ * b.z_ + a.z_ * b.w_ + a.x_ * b.y_ - a.y_ * b.x_);
}

void Quaterniond::normalize() {
  const double n = std::sqrt(x_ * x_ + y_ * y_ + z_ * z_ + w_ * w_);
  x_ /= n; y_ /= n; z_ /= n; w_ /= n;
}

// ---------------------------------------------------------------------------- SE3Quat
void SE3Quat::normalizeRotation() {  // se3quat.h:280-285
  if (_r.w_ < 0) { _r.x_ *= -1; _r.y_ *= -1; _r.z_ *= -1; _r.w_ *= -1; }
  _r.normalize();
}

SE3Quat SE3Quat::operator*(const SE3Quat &tr2) const {  // se3quat.h:106-112
  SE3Quat result(*this);
  result._t = result._t + (_r * tr2._t);
  result._r = _r * tr2._r;
  result.normalizeRotation();
  return result;
}

static Matrix3d skew(const Vector3d &v) {  // se3_ops.hpp
  Matrix3d m;
  m(0, 1) = -v[2]; m(0, 2) = v[1]; m(1, 2) = -v[0];
  m(1, 0) = v[2];  m(2, 0) = -v[1]; m(2, 1) = v[0];
  return m;
}

SE3Quat SE3Quat::exp(const Vector6d &update) {  // se3quat.h:223-257
  Vector3d omega(update[0], update[1], update[2]);
  Vector3d upsilon(update[3], update[4], update[5]);
  const double theta = omega.norm();
  const Matrix3d Omega = skew(omega);
  const Matrix3d Omega2 = Omega * Omega;
  const Matrix3d I = Matrix3d::Identity();
  Matrix3d R, V;
  if (theta < 0.00001) {
    for (int i = 0; i < 9; i++) { R.m[i] = I.m[i] + Omega.m[i] + Omega2.m[i]; V.m[i] = R.m[i]; }
  } else {
    const double a = std::sin(theta) / theta;
    const double b = (1 - std::cos(theta)) / (theta * theta);
    const double c = (theta - std::sin(theta)) / (std::pow(theta, 3));
    for (int i = 0; i < 9; i++) {
      R.m[i] = I.m[i] + a * Omega.m[i] + b * Omega2.m[i];
      V.m[i] = I.m[i] + b * Omega.m[i] + c * Omega2.m[i];
    }
  }
  return SE3Quat(Quaterniond(R), V * upsilon);
}

Matrix4d SE3Quat::to_homogeneous_matrix() const {  // se3quat.h:270-278
  Matrix4d M;
  const Matrix3d R = _r.toRotationMatrix();
  for (int r = 0; r < 3; r++) for (int c = 0; c < 3; c++) M(r, c) = R(r, c);
  M(0, 3) = _t[0]; M(1, 3) = _t[1]; M(2, 3) = _t[2];
  M(3, 3) = 1.0;
  return M;
}

Vector6d SE3Quat::toMinimalVector() const {  // se3quat.h:155-164
  Vector6d v;
  v[0] = _t[0]; v[1] = _t[1]; v[2] = _t[2];
  v[3] = _r.x(); v[4] = _r.y(); v[5] = _r.z();
  return v;
}

void SE3Quat::toPose7(double *p) const {
  p[0] = _r.x(); p[1] = _r.y(); p[2] = _r.z(); p[3] = _r.w();
  p[4] = _t[0]; p[5] = _t[1]; p[6] = _t[2];
}

SE3Quat SE3Quat::fromPose7(const double *p) {
  SE3Quat s;
  s._r = Quaterniond(p[3], p[0], p[1], p[2]);
  s._t = Vector3d(p[4], p[5], p[6]);
  return s;  // taken verbatim: the caller supplies a unit quaternion with w >= 0
}

void VertexSE3Expmap::oplusImpl(const double *update_) {  // types_six_dof_expmap.h:74-77
  Vector6d update;
  for (int i = 0; i < 6; i++) update[i] = update_[i];
  setEstimate(SE3Quat::exp(update) * estimate());
}

// ---------------------------------------------------------------------------- Huber
void RobustKernelHuber::robustify(double e, double rho[3]) const {  // robust_kernel_impl.cpp:77-91
  if (e <= dsqr) {
    rho[0] = e; rho[1] = 1.; rho[2] = 0.;
  } else {
    const double sqrte = std::sqrt(e);
    rho[0] = 2 * sqrte * _delta - dsqr;
    rho[1] = _delta / sqrte;
    rho[2] = -0.5 * rho[1] / e;
  }
}

// ---------------------------------------------------------------------------- edge
EdgeSE3ProjectIntensityOnlyPoseNID::EdgeSE3ProjectIntensityOnlyPoseNID()
    : _vertex(nullptr), _rk(nullptr), _information(1.0), _error(0.0), _level(0), _id(0), _internalId(-1) {
  std::memset(_jacobianOplusXi, 0, sizeof(_jacobianOplusXi));
}
EdgeSE3ProjectIntensityOnlyPoseNID::~EdgeSE3ProjectIntensityOnlyPoseNID() { delete _rk; }

void EdgeSE3ProjectIntensityOnlyPoseNID::computeError() {  // types_six_dof_expmap.h:220-228
  // use_CPU_ would call ClearPrevH(); ComputeH(); -- not part of this build (checked in optimize())
  _error = (2 * H_joint_ - H_ref_ - H_current_) / H_joint_;
}

void EdgeSE3ProjectIntensityOnlyPoseNID::linearizeOplus() {  // types_six_dof_expmap.cpp:530-538
  _jacobianOplusXi[0] = j0_; _jacobianOplusXi[1] = j1_; _jacobianOplusXi[2] = j2_;
  _jacobianOplusXi[3] = j3_; _jacobianOplusXi[4] = j4_; _jacobianOplusXi[5] = j5_;
}

void EdgeSE3ProjectIntensityOnlyPoseNID::constructQuadraticForm() {  // base_unary_edge.hpp:43-72
  VertexSE3Expmap *from = _vertex;
  if (from->fixed()) return;
  const double *A = _jacobianOplusXi;
  const double omega = _information;
  if (_rk) {
    const double error = chi2();
    double rho[3];
    _rk->robustify(error, rho);
    const double weightedOmega = rho[1] * _information;  // base_edge.h:96-102
    for (int i = 0; i < 6; i++) from->b[i] -= ((rho[1] * A[i]) * omega) * _error;
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) from->H[i * 6 + j] += (A[i] * weightedOmega) * A[j];
  } else {
    for (int i = 0; i < 6; i++) from->b[i] -= (A[i] * omega) * _error;
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) from->H[i * 6 + j] += (A[i] * omega) * A[j];
  }
}

// ---------------------------------------------------------------------------- dense LDLT
bool LinearSolverDense::solve(const double *Hin, double *x, const double *b) const {
  // linear_solver_dense.h:105-113: Eigen::LDLT (diagonal pivoting), isPositive() or fail
  enum { n = 6 };
  double A[n][n];
  int tr[n];
  bool positive = true;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) A[i][j] = Hin[i * n + j];
  for (int k = 0; k < n; k++) {
    int idx = k; double best = std::fabs(A[k][k]);
    for (int i = k + 1; i < n; i++) if (std::fabs(A[i][i]) > best) { best = std::fabs(A[i][i]); idx = i; }
    tr[k] = idx;
    if (idx != k) {
      for (int j = 0; j < k; j++) std::swap(A[k][j], A[idx][j]);
      for (int i = idx + 1; i < n; i++) std::swap(A[i][k], A[i][idx]);
      std::swap(A[k][k], A[idx][idx]);
      for (int i = k + 1; i < idx; i++) std::swap(A[i][k], A[idx][i]);
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
    const double akk = A[k][k];
    if (akk < 0) positive = false;
    if (std::fabs(akk) > 0) for (int i = k + 1; i < n; i++) A[i][k] /= akk;
  }
  if (!positive) return false;
  double y[n];
  for (int i = 0; i < n; i++) y[i] = b[i];
  for (int k = 0; k < n; k++) if (tr[k] != k) std::swap(y[k], y[tr[k]]);
  for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) y[i] -= A[i][j] * y[j];
  for (int i = 0; i < n; i++) { if (std::fabs(A[i][i]) > DBL_MIN) y[i] /= A[i][i]; else y[i] = 0; }
  for (int i = n - 1; i >= 0; i--) for (int j = i + 1; j < n; j++) y[i] -= A[j][i] * y[j];
  for (int k = n - 1; k >= 0; k--) if (tr[k] != k) std::swap(y[k], y[tr[k]]);
  for (int i = 0; i < n; i++) x[i] = y[i];
  return true;
}

// ---------------------------------------------------------------------------- block solver
bool BlockSolver_6_X::buildSystem() {  // block_solver.hpp:503-570
  VertexSE3Expmap *v = _opt->_vertices[0];
  v->clearQuadraticForm();
  const std::vector<EdgeSE3ProjectIntensityOnlyPoseNID *> &edges = _opt->_activeEdges;
  // The reference's walk (`while (isnan(der[6*m])) m++`, :535) assumes that exactly the level-1 cells are NaN
  // and runs off the array otherwise (an ACTIVE cell can still be NaN: im0 and im1 both constant over it give
  // Hj = 0, err = 0/0).  Same walk, bounded: an overrun fails the solve instead of reading past the array.
  const int n_cells = _opt->cell_num_ * _opt->cell_num_;
  for (int k = 0, m = 0; k < (int)edges.size(); ++k, m++) {
    EdgeSE3ProjectIntensityOnlyPoseNID *e = edges[k];
    while (m < n_cells && std::isnan(der_[6 * m])) m++;  // NaN-skip walk, :535
    if (m >= n_cells) {
      std::cerr << "BlockSolver::buildSystem: NaN Jacobian in an active cell (edge " << k << "): cell walk overran\n";
      return false;
    }
    e->set_j(der_[6 * m], der_[6 * m + 1], der_[6 * m + 2], der_[6 * m + 3], der_[6 * m + 4], der_[6 * m + 5]);
    e->linearizeOplus();
    e->constructQuadraticForm();
  }
  std::memcpy(_H, v->H, sizeof(_H));
  std::memcpy(_b, v->b, sizeof(_b));
  return true;
}

bool BlockSolver_6_X::setLambda(double lambda, bool backup) {
  for (int i = 0; i < 6; i++) {
    if (backup) _diagBackup[i] = _H[i * 6 + i];
    _H[i * 6 + i] += lambda;
  }
  return true;
}

void BlockSolver_6_X::restoreDiagonal() {
  for (int i = 0; i < 6; i++) _H[i * 6 + i] = _diagBackup[i];
}

bool BlockSolver_6_X::solve() { return _ls->solve(_H, _x, _b); }

// ---------------------------------------------------------------------------- LM
OptimizationAlgorithmLevenberg::OptimizationAlgorithmLevenberg(BlockSolver_6_X *solver)
    : _solver(solver), _optimizer(nullptr), _currentLambda(-1.), _tau(1e-5), _goodStepLowerScale(1. / 3.),
      _goodStepUpperScale(2. / 3.), _ni(2.), _lastRho(0.), _maxTrialsAfterFailure(10), _levenbergIterations(0),
      _nBad(0), _fused(false) {}

double OptimizationAlgorithmLevenberg::computeLambdaInit() const {
  double maxDiagonal = 0.;
  const VertexSE3Expmap *v = _optimizer->_vertices[0];
  for (int j = 0; j < 6; ++j) maxDiagonal = std::max(std::fabs(v->hessian(j, j)), maxDiagonal);
  return _tau * maxDiagonal;
}

double OptimizationAlgorithmLevenberg::computeScale() const {
  double scale = 0.;
  for (size_t j = 0; j < _solver->vectorSize(); j++)
    scale += _solver->x()[j] * (_currentLambda * _solver->x()[j] + _solver->b()[j]);
  return scale;
}

// Fused fast path (not in the reference): H, b and the robust chi2 come straight from the device
// reduction (nid_normal_equations) -- one kernel and 256 B per evaluation instead of the per-cell
// arrays and the per-edge set_h/set_j walk.  Same LM control flow as solve() below.
OptimizationAlgorithmLevenberg::SolverResult OptimizationAlgorithmLevenberg::solveFused(int iteration) {
  SparseOptimizer *opt = _optimizer;
  VertexSE3Expmap *vm = opt->_vertices[0];
  // the shards the legacy operators run on (one context on one GPU, or the cells of the pair spread over several:
  // include/nid/nid_multi.h): H, b and chi2 arrive already summed over the whole image
  const auto t_start = std::chrono::steady_clock::now();
  nid_multi *ctx = opt->native_pair_ ? opt->native_pair_ : nid_legacy_multi();
  if (opt->native_pair_) {
    if (iteration == 0) _haveNext = false;
  } else if (!ctx || iteration == 0) {  // first use: the legacy operator's own upload of the frame-pair state
    ctx = nid_legacy_prepare(opt->im0_, opt->im1_, opt->points3d_, opt->bs_counter_, opt->bs_value_ref_,
                             opt->bs_index_ref_, opt->camera_intrincis_, opt->bin_num_, opt->bs_degree_, opt->cell_num_,
                             opt->rows_, opt->cols_, opt->Href_);
    if (!ctx) return Fail;
    _haveNext = false;
  }
  double delta = 1e300;
  if (!opt->_activeEdges.empty() && opt->_activeEdges[0]->robustKernel())
    delta = opt->_activeEdges[0]->robustKernel()->delta();
  double p7[7], H[36], b[6], currentChi = 0;
  int32_t na = 0;
  vm->estimate().toPose7(p7);
  const double t_setup = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t_start).count();
  // NID_LM_TRACE=1: wall-clock of the stages of every outer iteration on stderr (tools/latency_sweep.py)
  static const bool lm_trace = std::getenv("NID_LM_TRACE") != nullptr;
  auto us_since = [&](std::chrono::steady_clock::time_point t0) {
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  };
  if (_haveNext && std::memcmp(p7, _nextPose, sizeof(p7)) == 0) {
    // the accepted trial of the previous outer iteration was evaluated WITH its Jacobian (see below): same pose bits,
    // same launch shape, hence the bits a launch now would give
    std::memcpy(H, _nextH, sizeof(H));
    std::memcpy(b, _nextB, sizeof(b));
    currentChi = _nextChi;
  } else if (nid_multi_normal_equations(ctx, p7, 1, delta, H, b, &currentChi, &na) != NID_OK) {
    return Fail;
  }
  _haveNext = false;
  const double t_jac = us_since(t_start);
  std::memcpy(vm->H, H, sizeof(H));
  std::memcpy(vm->b, b, sizeof(b));
  _solver->setSystem(H, b);
  double tempChi = currentChi;
  const double iniChi = currentChi;
  if (iteration == 0) { _currentLambda = computeLambdaInit(); _ni = 2; _nBad = 0; }
  double rho = 0;
  int &qmax = _levenbergIterations;
  qmax = 0;
  if (_speculative) {
    // The trial chain of one outer iteration is known in advance IF every trial is rejected:
    // lambda_{k+1} = lambda_k * ni_k, ni_{k+1} = 2 ni_k (:196-197).  Solve all of them on the host
    // (6x6, microseconds), evaluate the candidate poses in ONE batched cost-only launch, then replay
    // the reference's accept/reject scan over the results.  Same poses, same chi2 bits, same
    // decisions as the sequential loop below -- only the number of launches changes.
    const SE3Quat T0 = vm->estimate();
    constexpr int kTrialBatch = 16;  // one launch whose per-pose records travel as kernel arguments
    double xprev[6], xlast[6] = {0, 0, 0, 0, 0, 0};
    bool first_slice = true, first_of_chain = true;
    // How many of them to evaluate at once: a steady iteration accepts its first or second trial, the first
    // iteration of a pair needs up to seven (lambda_0 = 1e-5 max diag is far too small for this cost).  A launch of
    // 3 trial poses costs ~15 us, one of 10 ~35 us (tools/latency_sweep.py), so the chain is evaluated in slices:
    // as many trials as the PREVIOUS outer iteration needed (at least _speculativeFirst), then the rest.
    // The first iteration's own count says nothing about the next one (lambda has been tuned by then).
    int slice = iteration == 0 ? _maxTrialsAfterFailure
                               : std::max(_speculativeFirst, std::min(_lastTrials, _maxTrialsAfterFailure));
    while (qmax < _maxTrialsAfterFailure) {
      const int nb = std::min(std::min(kTrialBatch, slice), _maxTrialsAfterFailure - qmax);
      slice = kTrialBatch;  // a second slice takes everything that is left
      double lam[kTrialBatch], nis[kTrialBatch], xs[kTrialBatch][6], poses[kTrialBatch * 7];
      bool oks[kTrialBatch];
      SE3Quat cand[kTrialBatch];
      double l = _currentLambda, n_i = _ni;
      for (int j = 0; j < 6; j++) xprev[j] = first_slice ? _solver->x()[j] : xlast[j];
      first_slice = false;
      for (int k = 0; k < nb; k++) {
        lam[k] = l; nis[k] = n_i;
        double Hl[36];
        std::memcpy(Hl, H, sizeof(Hl));
        for (int j = 0; j < 6; j++) Hl[j * 6 + j] += l;
        LinearSolverDense ls;
        for (int j = 0; j < 6; j++) xs[k][j] = (k ? xs[k - 1][j] : xprev[j]);  // failed solve leaves x untouched
        oks[k] = ls.solve(Hl, xs[k], b);
        Vector6d u;
        for (int j = 0; j < 6; j++) u[j] = xs[k][j];
        cand[k] = SE3Quat::exp(u) * T0;
        cand[k].toPose7(poses + 7 * k);
        l *= n_i; n_i *= 2;
      }
      const auto t_slice = std::chrono::steady_clock::now();
      const double t_prep = us_since(t_start);
      // The FIRST trial of the chain -- the one a steady iteration accepts -- is evaluated WITH the Jacobian phase,
      // concurrently with the cost-only launch of the others (nid_launch_chain): if it is accepted, the next outer
      // iteration's H and b are already on the host -- one round trip per outer iteration instead of two (the chi2
      // of a pose is the same bits with and without the Jacobian phase).
      const int n_jac = (_speculativeJacobian && first_of_chain) ? 1 : 0;
      first_of_chain = false;
      if (nid_multi_launch_chain(ctx, 0, nb, poses, n_jac, delta) != NID_OK) return Fail;
      double chis[kTrialBatch], H0[36], b0[6];
      for (int k = 0; k < nb; k++)
        if (nid_multi_wait(ctx, k, k < n_jac ? H0 : nullptr, k < n_jac ? b0 : nullptr, &chis[k], &na) != NID_OK) return Fail;
      if (lm_trace)
        std::fprintf(stderr, "[lm] it %d: setup until %.1f us, J until %.1f us, host solves until %.1f us, %d trial poses%s %.1f us\n", iteration,
                     t_setup, t_jac, t_prep, nb, n_jac ? " (first with J)" : "", us_since(t_slice));
      bool accepted = false;
      for (int k = 0; k < nb; k++) {
        tempChi = oks[k] ? chis[k] : std::numeric_limits<double>::max();
        rho = (currentChi - tempChi);
        double scale = 0.;
        for (int j = 0; j < 6; j++) scale += xs[k][j] * (lam[k] * xs[k][j] + b[j]);
        scale += 1e-3;
        rho /= scale;
        qmax++;
        if (rho > 0 && std::isfinite(tempChi)) {
          double alpha = 1. - std::pow((2 * rho - 1), 3);
          alpha = (std::min)(alpha, _goodStepUpperScale);
          const double scaleFactor = (std::max)(_goodStepLowerScale, alpha);
          _currentLambda = lam[k] * scaleFactor;
          _ni = 2;
          currentChi = tempChi;
          vm->setEstimate(cand[k]);
          accepted = true;
          if (k < n_jac) {
            _haveNext = true;
            std::memcpy(_nextPose, poses + 7 * k, sizeof(_nextPose));
            std::memcpy(_nextH, H0, sizeof(_nextH));
            std::memcpy(_nextB, b0, sizeof(_nextB));
            _nextChi = chis[k];
          }
          break;
        }
        _currentLambda = lam[k] * nis[k];
        _ni = nis[k] * 2;
        if (!(rho < 0)) break;  // rho == 0 (or NaN): the reference's do/while stops here
      }
      for (int j = 0; j < 6; j++) xlast[j] = xs[nb - 1][j];
      if (accepted || !(rho < 0)) break;
    }
    _lastRho = rho;
    _fusedChi = currentChi;
    _lastTrials = iteration == 0 ? 0 : qmax;
    if (qmax == _maxTrialsAfterFailure || rho == 0) return Terminate;
    if ((iniChi - currentChi) * 1e3 < iniChi) _nBad++; else _nBad = 0;
    if (_nBad >= 3) return Terminate;
    return OK;
  }
  do {
    opt->push();
    _solver->setLambda(_currentLambda, true);
    const bool ok2 = _solver->solve();
    opt->update(_solver->x());
    _solver->restoreDiagonal();
    opt->_vertices[0]->estimate().toPose7(p7);
    // The FIRST trial -- the one a steady iteration accepts -- is evaluated WITH its Jacobian if asked for
    // (setSpeculativeJacobian without speculative trials: fused 4): accepted, it IS the next outer iteration's
    // evaluation -- one single-pose round trip per outer iteration, which is what the resident evaluator answers
    // fastest (the chi2 of a pose is the same bits with and without the Jacobian phase).
    const bool with_jac = _speculativeJacobian && qmax == 0;
    double H0[36], b0[6];
    if (nid_multi_normal_equations(ctx, p7, with_jac ? 1 : 0, delta, with_jac ? H0 : nullptr, with_jac ? b0 : nullptr, &tempChi, &na) != NID_OK)
      return Fail;
    const double trialChi = tempChi;
    if (!ok2) tempChi = std::numeric_limits<double>::max();
    rho = (currentChi - tempChi);
    double scale = computeScale();
    scale += 1e-3;
    rho /= scale;
    if (rho > 0 && std::isfinite(tempChi)) {
      double alpha = 1. - std::pow((2 * rho - 1), 3);
      alpha = (std::min)(alpha, _goodStepUpperScale);
      const double scaleFactor = (std::max)(_goodStepLowerScale, alpha);
      _currentLambda *= scaleFactor;
      _ni = 2;
      currentChi = tempChi;
      opt->discardTop();
      if (with_jac) {
        _haveNext = true;
        std::memcpy(_nextPose, p7, sizeof(_nextPose));
        std::memcpy(_nextH, H0, sizeof(_nextH));
        std::memcpy(_nextB, b0, sizeof(_nextB));
        _nextChi = trialChi;
      }
    } else {
      _currentLambda *= _ni;
      _ni *= 2;
      opt->pop();
    }
    qmax++;
  } while (rho < 0 && qmax < _maxTrialsAfterFailure);
  if (lm_trace)
    std::fprintf(stderr, "[lm] it %d: setup until %.1f us, J until %.1f us%s, %d sequential trials%s, end %.1f us\n", iteration, t_setup, t_jac,
                 t_jac - t_setup < 2.0 ? " (carried over)" : "", qmax, _speculativeJacobian ? " (first with J)" : "", us_since(t_start));
  _lastRho = rho;
  _fusedChi = currentChi;
  if (qmax == _maxTrialsAfterFailure || rho == 0) return Terminate;
  if ((iniChi - currentChi) * 1e3 < iniChi) _nBad++; else _nBad = 0;
  if (_nBad >= 3) return Terminate;
  return OK;
}

OptimizationAlgorithmLevenberg::SolverResult OptimizationAlgorithmLevenberg::solve(int iteration) {
  if (_fused) return solveFused(iteration);
  SparseOptimizer *opt = _optimizer;
  VertexSE3Expmap *vm = opt->_vertices[0];
  const int cell_num_2 = opt->cell_num_ * opt->cell_num_;
  std::vector<double> Htarget(cell_num_2, 0.0), Hjoint(cell_num_2, 0.0), der(6 * cell_num_2, 0.0);

  // :98 -- cost + Jacobian at the current estimate
  Matrix4d M = vm->estimate().to_homogeneous_matrix();
  CudaComputeH(true, opt->im0_, opt->im1_, opt->points3d_, opt->bs_counter_, opt->bs_value_ref_, opt->bs_index_ref_,
               M.data(), opt->camera_intrincis_, opt->bin_num_, opt->bs_degree_, opt->cell_num_, opt->rows_,
               opt->cols_, opt->Href_, nullptr, nullptr, Htarget.data(), Hjoint.data(), der.data());
  opt->set_h_pointer(Htarget.data(), Hjoint.data());
  if (!opt->computeActiveErrors()) return Fail;
  double currentChi = opt->activeRobustChi2();
  double tempChi = currentChi;
  const double iniChi = currentChi;
  _solver->set_j_bs(der.data());
  if (!_solver->buildSystem()) return Fail;

  if (iteration == 0) {
    _currentLambda = computeLambdaInit();
    _ni = 2;
    _nBad = 0;
  }
  double rho = 0;
  int &qmax = _levenbergIterations;
  qmax = 0;
  do {
    opt->push();
    _solver->setLambda(_currentLambda, true);
    const bool ok2 = _solver->solve();
    opt->update(_solver->x());
    VertexSE3Expmap *vn = opt->_vertices[0];
    _solver->restoreDiagonal();

    std::fill(Htarget.begin(), Htarget.end(), 0.0);
    std::fill(Hjoint.begin(), Hjoint.end(), 0.0);
    Matrix4d Mn = vn->estimate().to_homogeneous_matrix();
    CudaComputeH(false, opt->im0_, opt->im1_, opt->points3d_, opt->bs_counter_, opt->bs_value_ref_,
                 opt->bs_index_ref_, Mn.data(), opt->camera_intrincis_, opt->bin_num_, opt->bs_degree_,
                 opt->cell_num_, opt->rows_, opt->cols_, opt->Href_, nullptr, nullptr, Htarget.data(),
                 Hjoint.data(), der.data());
    if (!opt->computeActiveErrors()) return Fail;
    tempChi = opt->activeRobustChi2();
    if (!ok2) tempChi = std::numeric_limits<double>::max();

    rho = (currentChi - tempChi);
    double scale = computeScale();
    scale += 1e-3;
    rho /= scale;

    if (rho > 0 && std::isfinite(tempChi)) {
      double alpha = 1. - std::pow((2 * rho - 1), 3);
      alpha = (std::min)(alpha, _goodStepUpperScale);
      const double scaleFactor = (std::max)(_goodStepLowerScale, alpha);
      _currentLambda *= scaleFactor;
      _ni = 2;
      currentChi = tempChi;
      opt->discardTop();
    } else {
      _currentLambda *= _ni;
      _ni *= 2;
      opt->pop();
    }
    qmax++;
  } while (rho < 0 && qmax < _maxTrialsAfterFailure);
  _lastRho = rho;

  if (qmax == _maxTrialsAfterFailure || rho == 0) return Terminate;
  if ((iniChi - currentChi) * 1e3 < iniChi) _nBad++; else _nBad = 0;
  if (_nBad >= 3) return Terminate;
  return OK;
}

// ---------------------------------------------------------------------------- optimizer
SparseOptimizer::SparseOptimizer()
    : _algorithm(nullptr), h_target_(nullptr), h_joint_(nullptr), _verbose(false), _log(&std::cerr) {}

SparseOptimizer::~SparseOptimizer() {
  delete _algorithm;
  for (size_t i = 0; i < _edges.size(); i++) delete _edges[i];
  for (size_t i = 0; i < _vertices.size(); i++) delete _vertices[i];
}

bool SparseOptimizer::addEdge(EdgeSE3ProjectIntensityOnlyPoseNID *e) {
  e->_internalId = (int)_edges.size();  // insertion order = cell id (optimizable_graph.cpp:281)
  _edges.push_back(e);
  return true;
}

VertexSE3Expmap *SparseOptimizer::vertex(int id) const {
  for (size_t i = 0; i < _vertices.size(); i++) if (_vertices[i]->id() == id) return _vertices[i];
  return nullptr;
}

bool SparseOptimizer::initializeOptimization(int level) {
  _activeEdges.clear();
  for (size_t i = 0; i < _edges.size(); i++) if (_edges[i]->level() == level) _activeEdges.push_back(_edges[i]);
  return !_vertices.empty();
}

bool SparseOptimizer::computeActiveErrors() {  // sparse_optimizer.cpp:61-90
  const int n_cells = cell_num_ * cell_num_;
  for (int k = 0, n = 0; k < (int)_activeEdges.size(); ++k, ++n) {
    EdgeSE3ProjectIntensityOnlyPoseNID *e = _activeEdges[k];
    while (n < n_cells && (std::isnan(h_target_[n]) || std::isnan(h_joint_[n]))) n++;  // bounded, see buildSystem
    if (n >= n_cells) {
      std::cerr << "SparseOptimizer::computeActiveErrors: NaN entropy in an active cell (edge " << k << "): cell walk overran\n";
      return false;
    }
    e->set_h(h_target_[n], h_joint_[n]);
    e->computeError();
  }
  return true;
}

double SparseOptimizer::activeRobustChi2() const {  // sparse_optimizer.cpp:102-116
  double rho[3];
  double chi = 0.0;
  for (size_t i = 0; i < _activeEdges.size(); i++) {
    const EdgeSE3ProjectIntensityOnlyPoseNID *e = _activeEdges[i];
    if (e->robustKernel()) {
      e->robustKernel()->robustify(e->chi2(), rho);
      chi += rho[0];
    } else
      chi += e->chi2();
  }
  return chi;
}

void SparseOptimizer::update(const double *upd) { _vertices[0]->oplusImpl(upd); }
void SparseOptimizer::push() { _vertices[0]->push(); }
void SparseOptimizer::pop() { _vertices[0]->pop(); }
void SparseOptimizer::discardTop() { _vertices[0]->discardTop(); }

int SparseOptimizer::optimize(int iterations) {  // sparse_optimizer.cpp:356-450
  if (_vertices.empty() || !_algorithm) {
    std::cerr << "SparseOptimizer::optimize: 0 vertices to optimize, maybe forgot to call initializeOptimization()\n";
    return -1;
  }
  for (size_t i = 0; i < _activeEdges.size(); i++)
    if (_activeEdges[i]->use_CPU_) {
      std::cerr << "SparseOptimizer::optimize: use_CPU_ edges are not available in this build -- the CPU NID edge "
                   "exists only as the test oracle; set use_gpu: 1\n";
      return -1;
    }
  int cjIterations = 0;
  bool ok = true;
  robustchi2_his_.assign(iterations, 0.0);
  _trace.clear();
  _batchStatistics.clear();
  OptimizationAlgorithmLevenberg::SolverResult result = OptimizationAlgorithmLevenberg::OK;
  for (int i = 0; i < iterations && ok; i++) {
    const auto t_it0 = std::chrono::steady_clock::now();
    struct StatScope {  // one G2OBatchStatistics entry per outer iteration, whichever branch ends it
      SparseOptimizer *o; int it; std::chrono::steady_clock::time_point t0;
      ~StatScope() {
        if (!o->_computeBatchStatistics) return;
        G2OBatchStatistics st;
        st.iteration = it; st.numVertices = (int)o->_vertices.size(); st.numEdges = (int)o->_activeEdges.size();
        st.levenbergIterations = o->_algorithm->levenbergIteration();
        st.chi2 = o->_trace.empty() ? 0.0 : o->_trace.back().chi2;
        st.hessianDimension = 6;
        st.timeIteration = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        o->_batchStatistics.push_back(st);
      }
    } stat_scope{this, i, t_it0};
    result = _algorithm->solve(i);
    ok = (result == OptimizationAlgorithmLevenberg::OK);
    if (verbose() && _algorithm->fused()) {
      // fused mode: the accepted pose's robust chi2 is already known; skip the extra evaluation
      const double chi = _algorithm->fusedChi();
      if (_log) {
        char buf[256];
        std::snprintf(buf, sizeof(buf), "iteration= %d\t chi2= %.6f\t edges= %d\t schur= 0\t lambda= %.6f\t levenbergIter= %d\n",
                      i, chi, (int)_activeEdges.size(), _algorithm->currentLambda(), _algorithm->levenbergIteration());
        (*_log) << buf;
      }
      IterationRecord r;
      r.iteration = i; r.chi2 = chi; r.lambda = _algorithm->currentLambda(); r.rho = _algorithm->lastRho();
      r.levenbergIter = _algorithm->levenbergIteration();
      _vertices[0]->estimate().toPose7(r.pose7);
      _trace.push_back(r);
      robustchi2_his_[i] = chi;
      ++cjIterations;
      continue;
    }
    if (verbose()) {
      // the verbose branch evaluates the cost once more at the accepted pose (:404-427)
      const int n2 = cell_num_ * cell_num_;
      std::vector<double> Htarget(n2, 0.0), Hjoint(n2, 0.0);
      Matrix4d M = _vertices[0]->estimate().to_homogeneous_matrix();
      CudaComputeH(false, im0_, im1_, points3d_, bs_counter_, bs_value_ref_, bs_index_ref_, M.data(),
                   camera_intrincis_, bin_num_, bs_degree_, cell_num_, rows_, cols_, Href_, nullptr, nullptr,
                   Htarget.data(), Hjoint.data(), nullptr);
      set_h_pointer(Htarget.data(), Hjoint.data());
      const bool walked = computeActiveErrors();
      set_h_pointer(nullptr, nullptr);
      if (!walked) return 0;
      const double chi = activeRobustChi2();
      if (_log) {
        char buf[256];
        std::snprintf(buf, sizeof(buf), "iteration= %d\t chi2= %.6f\t edges= %d\t schur= 0\t lambda= %.6f\t levenbergIter= %d\n",
                      i, chi, (int)_activeEdges.size(), _algorithm->currentLambda(), _algorithm->levenbergIteration());
        (*_log) << buf;
      }
      IterationRecord r;
      r.iteration = i; r.chi2 = chi; r.lambda = _algorithm->currentLambda(); r.rho = _algorithm->lastRho();
      r.levenbergIter = _algorithm->levenbergIteration();
      _vertices[0]->estimate().toPose7(r.pose7);
      _trace.push_back(r);
    }
    ++cjIterations;
    robustchi2_his_[i] = activeRobustChi2();
  }
  if (result == OptimizationAlgorithmLevenberg::Fail) return 0;
  return cjIterations;
}

}  // namespace g2o
