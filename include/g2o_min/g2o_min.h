// g2o_min.h -- Eigen-free, g2o-shaped host API for the NID pose problem.
//
// Mirrors, name for name, the slice of the (modified) g2o that
// NID_pose_estimation.cpp drives (reference paths relative to the checkout):
//   g2o::SE3Quat                           g2o/g2o/types/se3quat.h
//   g2o::VertexSE3Expmap                   g2o/g2o/types/types_six_dof_expmap.h:60-78
//   g2o::EdgeSE3ProjectIntensityOnlyPoseNID  types_six_dof_expmap.h:210-333 (GPU-mode members)
//   g2o::RobustKernelHuber                 g2o/g2o/core/robust_kernel_impl.{h,cpp}
//   g2o::LinearSolverDense / BlockSolver_6_X  solvers/linear_solver_dense.h, core/block_solver.h(pp)
//   g2o::OptimizationAlgorithmLevenberg    core/optimization_algorithm_levenberg.cpp
//   g2o::SparseOptimizer (+ public NID fields)  core/sparse_optimizer.{h,cpp}:299-313
// With one vertex and unary edges the generic hypergraph machinery of g2o
// degenerates to a 6x6 accumulate + solve, which is what this header keeps.
// The NID operator itself is reached through include/nid/legacy_ops.h
// (g2o::CudaComputeH etc.), i.e. through the HIP library -- there is no CPU
// compute path here: an edge with use_CPU_ == true makes optimize() fail.
#ifndef G2O_MIN_H
#define G2O_MIN_H

#include <cmath>
#include <cstddef>
#include <cstring>
#include <iosfwd>
#include <vector>

struct nid_multi;

namespace g2o {

// ---- tiny fixed-size algebra (column-major where Eigen's .data() matters) ----
struct Vector3d {
  double v[3];
  Vector3d() : v{0, 0, 0} {}
  Vector3d(double a, double b, double c) : v{a, b, c} {}
  double &operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
  double &operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  Vector3d operator+(const Vector3d &o) const { return Vector3d(v[0] + o.v[0], v[1] + o.v[1], v[2] + o.v[2]); }
  double norm() const { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }
};

struct Vector6d {
  double v[6];
  Vector6d() { std::memset(v, 0, sizeof(v)); }
  double &operator[](int i) { return v[i]; }
  double operator[](int i) const { return v[i]; }
  double &operator()(int i) { return v[i]; }
  double operator()(int i) const { return v[i]; }
  Vector6d operator-(const Vector6d &o) const { Vector6d r; for (int i = 0; i < 6; i++) r.v[i] = v[i] - o.v[i]; return r; }
  const double *data() const { return v; }
  double *data() { return v; }
};

struct Matrix3d {  // row-major access m(r,c)
  double m[9];
  Matrix3d() { std::memset(m, 0, sizeof(m)); }
  static Matrix3d Identity() { Matrix3d I; I.m[0] = I.m[4] = I.m[8] = 1; return I; }
  double &operator()(int r, int c) { return m[r * 3 + c]; }
  double operator()(int r, int c) const { return m[r * 3 + c]; }
  Matrix3d operator*(const Matrix3d &o) const;
  Vector3d operator*(const Vector3d &x) const;
  Matrix3d transpose() const;
};

struct Matrix4d {  // COLUMN-major storage, like Eigen: data()[c*4 + r]
  double d[16];
  Matrix4d() { std::memset(d, 0, sizeof(d)); }
  double &operator()(int r, int c) { return d[c * 4 + r]; }
  double operator()(int r, int c) const { return d[c * 4 + r]; }
  const double *data() const { return d; }
  double *data() { return d; }
};

struct Quaterniond {
  double x_, y_, z_, w_;
  Quaterniond() : x_(0), y_(0), z_(0), w_(1) {}
  Quaterniond(double w, double x, double y, double z) : x_(x), y_(y), z_(z), w_(w) {}  // Eigen argument order
  explicit Quaterniond(const Matrix3d &R);
  double x() const { return x_; } double y() const { return y_; } double z() const { return z_; } double w() const { return w_; }
  Matrix3d toRotationMatrix() const;
  Vector3d operator*(const Vector3d &v) const;      // Eigen _transformVector
  Quaterniond operator*(const Quaterniond &o) const;
  void normalize();
};

// ---- se3quat.h ------------------------------------------------------------------
class SE3Quat {
 public:
  SE3Quat() {}
  SE3Quat(const Matrix3d &R, const Vector3d &t) : _r(Quaterniond(R)), _t(t) { normalizeRotation(); }
  SE3Quat(const Quaterniond &q, const Vector3d &t) : _r(q), _t(t) { normalizeRotation(); }
  const Vector3d &translation() const { return _t; }
  const Quaterniond &rotation() const { return _r; }
  SE3Quat operator*(const SE3Quat &tr2) const;              // se3quat.h:106-112
  Vector3d map(const Vector3d &xyz) const { return _r * xyz + _t; }  // :217-220
  static SE3Quat exp(const Vector6d &update);               // :223-257
  Matrix4d to_homogeneous_matrix() const;                   // :270-278
  Vector6d toMinimalVector() const;                         // :155-164
  void normalizeRotation();                                 // :280-285
  // pose7 = {qx,qy,qz,qw,tx,ty,tz} as used by the C-ABI
  void toPose7(double *p) const;
  static SE3Quat fromPose7(const double *p);

 protected:
  Quaterniond _r;
  Vector3d _t;
};

// ---- vertex ------------------------------------------------------------------------
class VertexSE3Expmap {
 public:
  VertexSE3Expmap() : _id(0), _fixed(false) {}
  void setEstimate(const SE3Quat &e) { _estimate = e; }
  const SE3Quat &estimate() const { return _estimate; }
  void setId(int id) { _id = id; }
  int id() const { return _id; }
  void setFixed(bool f) { _fixed = f; }
  bool fixed() const { return _fixed; }
  void oplusImpl(const double *update);                     // types_six_dof_expmap.h:74-77
  void push() { _backup.push_back(_estimate); }
  void pop() { _estimate = _backup.back(); _backup.pop_back(); }
  void discardTop() { _backup.pop_back(); }
  // quadratic form of the single 6-dof vertex
  double H[36];  // row-major
  double b[6];
  void clearQuadraticForm() { std::memset(H, 0, sizeof(H)); std::memset(b, 0, sizeof(b)); }
  double hessian(int i, int j) const { return H[i * 6 + j]; }

 private:
  SE3Quat _estimate;
  std::vector<SE3Quat> _backup;
  int _id;
  bool _fixed;
};

// ---- robust kernel -----------------------------------------------------------------
class RobustKernelHuber {
 public:
  RobustKernelHuber() : _delta(1.0), dsqr(1.0f) {}
  void setDelta(double delta) { dsqr = (float)(delta * delta); _delta = delta; }  // robust_kernel_impl.cpp:65-69
  double delta() const { return _delta; }
  void robustify(double e2, double rho[3]) const;           // :77-91
 private:
  double _delta;
  float dsqr;  // robust_kernel_impl.h:84
};

// ---- the NID edge (GPU-mode members of types_six_dof_expmap.h:210-333) -----------------
class EdgeSE3ProjectIntensityOnlyPoseNID {
 public:
  EdgeSE3ProjectIntensityOnlyPoseNID();
  ~EdgeSE3ProjectIntensityOnlyPoseNID();
  void setVertex(int, VertexSE3Expmap *v) { _vertex = v; }
  VertexSE3Expmap *vertex(int) const { return _vertex; }
  void setRobustKernel(RobustKernelHuber *rk) { delete _rk; _rk = rk; }
  RobustKernelHuber *robustKernel() const { return _rk; }
  void setInformation(double info) { _information = info; }
  void setLevel(int l) { _level = l; }
  int level() const { return _level; }
  void setId(int id) { _id = id; }
  int internalId() const { return _internalId; }
  void set_bspline_relates(int bs_degree, int bin_num) { bs_degree_ = bs_degree; bin_num_ = bin_num; }
  void set_href(double Href) { H_ref_ = Href; }                       // :235-237
  void set_h(double Htarget, double Hjoint) { H_current_ = Htarget; H_joint_ = Hjoint; }  // :298-302
  void set_j(double j0, double j1, double j2, double j3, double j4, double j5) {           // :304-306
    j0_ = j0; j1_ = j1; j2_ = j2; j3_ = j3; j4_ = j4; j5_ = j5;
  }
  void computeError();        // :220-228 (GPU branch: no recomputation)
  void linearizeOplus();      // types_six_dof_expmap.cpp:530-538
  void constructQuadraticForm();  // base_unary_edge.hpp:43-72
  double chi2() const { return _error * (_information * _error); }   // base_edge.h:58-61
  double error() const { return _error; }
  const double *jacobianOplusXi() const { return _jacobianOplusXi; }

  bool use_CPU_ = false;  // must stay false: the CPU edge is not part of this build
  double fx_ = 0, fy_ = 0, cx_ = 0, cy_ = 0;
  double j0_ = 0, j1_ = 0, j2_ = 0, j3_ = 0, j4_ = 0, j5_ = 0;
  double H_current_ = 0.0, H_ref_ = 0.0, H_joint_ = 0.0;
  int bs_degree_ = 3, bin_num_ = 10;

 private:
  friend class SparseOptimizer;
  VertexSE3Expmap *_vertex;
  RobustKernelHuber *_rk;
  double _information, _error, _jacobianOplusXi[6];
  int _level, _id, _internalId;
};

// ---- solvers ------------------------------------------------------------------------
class LinearSolverDense {  // solvers/linear_solver_dense.h:65-118 (Eigen::LDLT on a dense 6x6)
 public:
  bool solve(const double *H36, double *x, const double *b) const;
};

class SparseOptimizer;

class BlockSolver_6_X {
 public:
  typedef LinearSolverDense LinearSolverType;
  struct PoseMatrixType {};
  explicit BlockSolver_6_X(LinearSolverType *ls) : _ls(ls), _opt(nullptr), der_(nullptr) {}
  ~BlockSolver_6_X() { delete _ls; }
  void set_j_bs(double *der) { der_ = der; }                 // block_solver.h:144
  bool buildSystem();                                        // block_solver.hpp:503-570
  bool setLambda(double lambda, bool backup);                // :574-599
  void restoreDiagonal();                                    // :602-614
  bool solve();                                              // :355-366
  const double *x() const { return _x; }
  const double *b() const { return _b; }
  size_t vectorSize() const { return 6; }
  void setOptimizer(SparseOptimizer *o) { _opt = o; }
  void setSystem(const double *H36, const double *b6) { std::memcpy(_H, H36, sizeof(_H)); std::memcpy(_b, b6, sizeof(_b)); }
 private:
  LinearSolverType *_ls;
  SparseOptimizer *_opt;
  double *der_;
  double _H[36], _b[6], _x[6], _diagBackup[6];
};

class OptimizationAlgorithmLevenberg {
 public:
  enum SolverResult { Terminate = 2, OK = 1, Fail = -1 };
  explicit OptimizationAlgorithmLevenberg(BlockSolver_6_X *solver);
  ~OptimizationAlgorithmLevenberg() { delete _solver; }
  SolverResult solve(int iteration);                         // optimization_algorithm_levenberg.cpp:61-225
  double currentLambda() const { return _currentLambda; }
  int levenbergIteration() const { return _levenbergIterations; }
  double lastRho() const { return _lastRho; }
  void setOptimizer(SparseOptimizer *o) { _optimizer = o; _solver->setOptimizer(o); }
  // fast path (not in the reference): take H, b, chi2 from the fused device reduction
  // (nid_normal_equations) instead of the per-edge set_h/set_j walk
  void setFusedNormalEquations(bool on) { _fused = on; }
  // with the fused path: evaluate the whole rejection chain of an outer iteration (up to 8
  // candidate poses) in one batched launch; decisions identical to the sequential trials
  void setSpeculativeTrials(bool on) { _speculative = on; }
  // with speculative trials: evaluate the first (short) slice of the chain with its Jacobian, so that the accepted
  // trial already carries the next iteration's H and b -- one launch per outer iteration
  void setSpeculativeJacobian(bool on) { _speculativeJacobian = on; }
  bool fused() const { return _fused; }
  double fusedChi() const { return _fusedChi; }
 private:
  SolverResult solveFused(int iteration);
  double computeLambdaInit() const;                          // :227-241
  double computeScale() const;                               // :243-250
  BlockSolver_6_X *_solver;
  SparseOptimizer *_optimizer;
  double _currentLambda, _tau, _goodStepLowerScale, _goodStepUpperScale, _ni, _lastRho;
  int _maxTrialsAfterFailure, _levenbergIterations, _nBad;
  bool _fused;
  bool _speculative = false;
  int _speculativeFirst = 3;   // trial poses of the first slice of an outer iteration's rejection chain (see solveFused)
  int _lastTrials = 10;        // trials the previous outer iteration needed (the first iteration of a pair: all of them)
  double _fusedChi = 0.0;
  bool _speculativeJacobian = false, _haveNext = false;
  double _nextPose[7], _nextH[36], _nextB[6], _nextChi = 0.0;
};

// the fields of batch_stats.h:39-78 that exist for a one-vertex dense problem (no Schur complement, no
// symbolic decomposition); filled per outer iteration when setComputeBatchStatistics(true)
// (sparse_optimizer.cpp:372-446, optimization_algorithm_levenberg.cpp:118-162)
struct G2OBatchStatistics {
  int iteration = 0, numVertices = 0, numEdges = 0, levenbergIterations = 0;
  double chi2 = 0.0, timeIteration = 0.0;  // seconds
  size_t hessianDimension = 0;
};
typedef std::vector<G2OBatchStatistics> BatchStatisticsContainer;

struct IterationRecord {  // one line of the verbose output (sparse_optimizer.cpp:434-440)
  int iteration;
  double chi2, lambda, rho;
  int levenbergIter;
  double pose7[7];
};

class SparseOptimizer {
 public:
  SparseOptimizer();
  ~SparseOptimizer();
  void setAlgorithm(OptimizationAlgorithmLevenberg *a) { _algorithm = a; a->setOptimizer(this); }
  void setVerbose(bool v) { _verbose = v; }
  bool verbose() const { return _verbose; }
  bool addVertex(VertexSE3Expmap *v) { _vertices.push_back(v); return true; }
  bool addEdge(EdgeSE3ProjectIntensityOnlyPoseNID *e);
  VertexSE3Expmap *vertex(int id) const;
  bool initializeOptimization(int level = 0);
  int optimize(int iterations);                               // sparse_optimizer.cpp:356-450
  bool computeActiveErrors();                                 // :61-90; false: the NaN-skip walk overran (see .cpp)
  double activeRobustChi2() const;                            // :102-116
  void set_h_pointer(double *h_target, double *h_joint) { h_target_ = h_target; h_joint_ = h_joint; }  // :648-651
  void update(const double *upd);                             // :453-466
  void push(); void pop(); void discardTop();
  const std::vector<EdgeSE3ProjectIntensityOnlyPoseNID *> &activeEdges() const { return _activeEdges; }
  const std::vector<VertexSE3Expmap *> &activeVertices() const { return _vertices; }
  const double *activeRobustChi2His() const { return robustchi2_his_.data(); }
  const std::vector<IterationRecord> &trace() const { return _trace; }
  void setComputeBatchStatistics(bool on) { _computeBatchStatistics = on; }
  const BatchStatisticsContainer &batchStatistics() const { return _batchStatistics; }
  void setLogStream(std::ostream *os) { _log = os; }

  // public NID fields, sparse_optimizer.h:299-313
  double *im0_ = nullptr, *im1_ = nullptr, *points3d_ = nullptr;
  int rows_ = 0, cols_ = 0;
  double *camera_intrincis_ = nullptr;
  int bin_num_ = 10, bs_degree_ = 3, cell_num_ = 16;
  int *bs_counter_ = nullptr;
  double *bs_value_ref_ = nullptr;
  int *bs_index_ref_ = nullptr;
  double *Href_ = nullptr;
  std::vector<double> robustchi2_his_;
  // (not in the reference) the frame pair is already on the device(s), set up through the C-ABI
  // (nid_legacy_set_pair_u16): the fused flows evaluate on these shards and never touch the array fields above
  nid_multi *native_pair_ = nullptr;

 private:
  friend class OptimizationAlgorithmLevenberg;
  friend class BlockSolver_6_X;
  OptimizationAlgorithmLevenberg *_algorithm;
  std::vector<VertexSE3Expmap *> _vertices;
  std::vector<EdgeSE3ProjectIntensityOnlyPoseNID *> _edges, _activeEdges;
  double *h_target_, *h_joint_;
  bool _verbose;
  std::ostream *_log;
  std::vector<IterationRecord> _trace;
  bool _computeBatchStatistics = false;
  BatchStatisticsContainer _batchStatistics;
};

}  // namespace g2o
#endif
