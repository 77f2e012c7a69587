// legacy_ops.cpp -- Calculate3Dpoint / CudaComputeHref / g2o::CudaComputeH on top
// of the C-ABI (include/nid/nid_c.h).  See include/nid/legacy_ops.h.
#include "nid/legacy_ops.h"

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nid/nid_c.h"

namespace {

struct LegacyState {
  nid_ctx *ctx = nullptr;
  int rows = 0, cols = 0, cell = 0, bins = 0;
  double intr[4] = {0, 0, 0, 0};
  // identity of what is resident on the device
  const double *im0 = nullptr, *im1 = nullptr, *points = nullptr, *bs_ref = nullptr, *href = nullptr;
  const int *bs_counter = nullptr;
  bool have_ref = false, have_target = false, have_href = false;
};

LegacyState g_state;
int g_device = 0;
int g_jac_bound = NID_JACBOUND_CPU;
int g_math_mode = NID_MATH_FAST;
long g_uploads = 0;

bool always_upload() {
  static const bool v = getenv("NID_LEGACY_ALWAYS_UPLOAD") != nullptr;
  return v;
}

void report(const char *where, int rc, nid_ctx *ctx) {
  // the reference prints CUDA errors and carries on (computeH.cu:454-473)
  std::fprintf(stderr, "[nid legacy] %s failed: %s (%d) %s\n", where, nid_status_string(rc), rc,
               ctx ? nid_last_error(ctx) : "");
}

nid_ctx *get_ctx(int rows, int cols, int cell, int bins, int deg, const double *intr) {
  LegacyState &S = g_state;
  const bool same = S.ctx && S.rows == rows && S.cols == cols && S.cell == cell && S.bins == bins &&
                    S.intr[0] == intr[0] && S.intr[1] == intr[1] && S.intr[2] == intr[2] && S.intr[3] == intr[3];
  if (same) return S.ctx;
  if (S.ctx) nid_destroy(S.ctx);
  S = LegacyState();
  nid_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.rows = rows; cfg.cols = cols; cfg.cell_num = cell; cfg.bin_num = bins; cfg.bs_degree = deg;
  cfg.device = g_device; cfg.cell_begin = 0; cfg.cell_end = 0;
  cfg.fx = intr[0]; cfg.fy = intr[1]; cfg.cx = intr[2]; cfg.cy = intr[3];
  nid_ctx *ctx = nullptr;
  int rc = nid_create(&cfg, &ctx);
  if (rc != NID_OK) { report("nid_create", rc, nullptr); return nullptr; }
  // the operator signatures carry a 4x4 matrix: computeH.cu:152-154 semantics for the transform
  nid_set_options(ctx, g_jac_bound, NID_XFORM_MATRIX);
  nid_set_math_mode(ctx, g_math_mode);
  // the operators are blocking, one pose (or one LM rejection chain) at a time: latency matters, not the
  // pipelined throughput the 128-thread default is tuned for.  256-thread workgroups: single-pose
  // cost+Jacobian 29.6 -> 18.6 us, cost-only 17.7 -> 14.1 us, a 10-pose cost-only chain 30.7 -> 27.8 us.
  nid_set_block_threads(ctx, 256);
  S.ctx = ctx; S.rows = rows; S.cols = cols; S.cell = cell; S.bins = bins;
  std::memcpy(S.intr, intr, sizeof(S.intr));
  return ctx;
}

bool to_u8(const double *im, size_t n, std::vector<uint8_t> *out) {
  out->resize(n);
  return nid_set_reference_image_f64(im, (int64_t)n, out->data()) == NID_OK;
}

int ensure_reference(LegacyState &S, const double *im0, const double *points3d) {
  if (S.have_ref && !always_upload() && S.im0 == im0 && S.points == points3d) return NID_OK;
  const size_t N = (size_t)S.rows * S.cols;
  std::vector<uint8_t> im;
  if (!to_u8(im0, N, &im)) return NID_ERR_UNSUPPORTED;
  int rc = nid_set_reference_points(S.ctx, points3d, im.data());
  if (rc != NID_OK) return rc;
  S.im0 = im0; S.points = points3d; S.have_ref = true; S.have_href = false;
  g_uploads++;
  return NID_OK;
}

}  // namespace

void Calculate3Dpoint(double *depth, double *pose_c2w, double *points_3d, double *camera_intrincis, int rows,
                      int cols) {
  int rc = nid_backproject(depth, pose_c2w, camera_intrincis[0], camera_intrincis[1], camera_intrincis[2],
                           camera_intrincis[3], rows, cols, g_device, points_3d);
  if (rc != NID_OK) report("Calculate3Dpoint", rc, nullptr);
}

void CudaComputeHref(double *im0, double *points3d, double *pose, double *camera_intrincis, int bin_num,
                     int bs_degree, int cell_num, int rows, int cols, double *bs_value, int *bs_index,
                     int *bs_counter, double *Href) {
  nid_ctx *ctx = get_ctx(rows, cols, cell_num, bin_num, bs_degree, camera_intrincis);
  if (!ctx) return;
  LegacyState &S = g_state;
  int rc = ensure_reference(S, im0, points3d);
  if (rc != NID_OK) { report("CudaComputeHref(reference upload)", rc, ctx); return; }
  const size_t N = (size_t)rows * cols;
  const int ncell = cell_num * cell_num;
  std::vector<int32_t> cnt(ncell), idx(bs_index ? N : 0);
  std::vector<double> href(ncell);
  rc = nid_compute_href_matrix(ctx, pose, cnt.data(), href.data(), bs_value, bs_index ? idx.data() : nullptr);
  if (rc != NID_OK) { report("CudaComputeHref", rc, ctx); return; }
  for (int c = 0; c < ncell; c++) {
    bs_counter[c] = cnt[c];
    // CudaComputeHref.cu:205-220: NaN when inactive, otherwise subtract onto the caller's value
    Href[c] = std::isnan(href[c]) ? NAN : Href[c] + href[c];
  }
  if (bs_value) {
    // legacy marker (CudaComputeHref.cu:82-87,126-130): NaN weights for pixels that are invalid or
    // out of frame at this pose.  In-frame weights sum to 1, so an all-zero row is exactly that set.
    for (size_t i = 0; i < N; i++) {
      double *w = bs_value + 4 * i;
      if (w[0] == 0.0 && w[1] == 0.0 && w[2] == 0.0 && w[3] == 0.0) w[0] = w[1] = w[2] = w[3] = NAN;
    }
  }
  if (bs_index) for (size_t i = 0; i < N; i++) bs_index[i] = idx[i];
  S.have_href = true;  // the device already holds these weights (CPU-edge convention: 0 instead of NaN)
  S.bs_ref = bs_value; S.bs_counter = bs_counter; S.href = Href;
}

namespace g2o {

void CudaComputeH(bool calculate_der, double *im0, double *im1, double *points3d, int *bs_counter, double *bs_ref,
                  int *bs_index_ref, double *pose, double *camera_intrincis, int bin_num, int bs_degree,
                  int cell_num, int rows, int cols, double *Href, double *pro_target, double *pro_joint,
                  double *Htarget, double *Hjoint, double *der) {
  (void)pro_target; (void)pro_joint;  // accepted, never read or written (computeH.cu:373-502)
  nid_ctx *ctx = get_ctx(rows, cols, cell_num, bin_num, bs_degree, camera_intrincis);
  if (!ctx) return;
  LegacyState &S = g_state;
  const size_t N = (size_t)rows * cols;
  const int ncell = cell_num * cell_num;
  int rc = ensure_reference(S, im0, points3d);
  if (rc != NID_OK) { report("CudaComputeH(reference upload)", rc, ctx); return; }
  if (!S.have_target || always_upload() || S.im1 != im1) {
    std::vector<uint8_t> im;
    if (!to_u8(im1, N, &im)) { report("CudaComputeH(im1 is not u8-valued)", NID_ERR_UNSUPPORTED, ctx); return; }
    rc = nid_set_target_u8(ctx, im.data());
    if (rc != NID_OK) { report("CudaComputeH(target upload)", rc, ctx); return; }
    S.im1 = im1; S.have_target = true;
    g_uploads++;
  }
  if (!S.have_href || always_upload() || S.bs_ref != bs_ref || S.bs_counter != bs_counter) {
    // Href is only handed over with calculate_der (computeH.cu:428-429); the kernels also need it to
    // know which cells are active, so the first call must carry it
    std::vector<double> href(ncell, 0.0);
    if (Href) for (int c = 0; c < ncell; c++) href[c] = Href[c];
    rc = nid_set_href_state(ctx, bs_counter, href.data(), bs_ref, bs_index_ref);
    if (rc != NID_OK) { report("CudaComputeH(href state upload)", rc, ctx); return; }
    S.bs_ref = bs_ref; S.bs_counter = bs_counter; S.href = Href; S.have_href = true;
    g_uploads++;
  }
  std::vector<double> ht(ncell), hj(ncell);
  rc = nid_evaluate_matrix(ctx, pose, calculate_der ? 1 : 0, ht.data(), hj.data(), nullptr,
                           calculate_der ? der : nullptr);
  if (rc != NID_OK) { report("CudaComputeH", rc, ctx); return; }
  for (int c = 0; c < ncell; c++) {
    // CalculateHKernel: NaN for bs_counter < 300, else `-=` onto the caller's (zeroed) value
    Htarget[c] = std::isnan(ht[c]) ? NAN : Htarget[c] + ht[c];
    Hjoint[c] = std::isnan(hj[c]) ? NAN : Hjoint[c] + hj[c];
  }
}

}  // namespace g2o

extern "C" {

void nid_legacy_set_jacobian_bound(int mode) {
  g_jac_bound = mode ? NID_JACBOUND_CUDA : NID_JACBOUND_CPU;
  if (g_state.ctx) nid_set_options(g_state.ctx, g_jac_bound, NID_XFORM_MATRIX);
}

void nid_legacy_set_math_mode(int mode) {
  g_math_mode = mode ? NID_MATH_STRICT : NID_MATH_FAST;
  if (g_state.ctx) nid_set_math_mode(g_state.ctx, g_math_mode);
}

void nid_legacy_set_device(int device) {
  if (device != g_device) nid_legacy_reset();
  g_device = device;
}

void nid_legacy_reset(void) {
  if (g_state.ctx) nid_destroy(g_state.ctx);
  g_state = LegacyState();
}

nid_ctx *nid_legacy_context(void) { return g_state.ctx; }

long nid_legacy_upload_count(void) { return g_uploads; }

}  // extern "C"
