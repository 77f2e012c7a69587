// nid_hip_stub.cpp -- TEST INFRASTRUCTURE: a stand-in for the entry points of libnid_hip.so that
// host/legacy_ops.cpp calls, without a GPU, so that the legacy operators' HOST logic -- what they read of the
// caller's buffers and WHEN -- runs under AddressSanitizer on the CPU (tests/test_host_cpu.py; the pattern of
// tests/cpp/rccl_stub.c).  The "device" is a private copy of what was uploaded; an "evaluation" takes ~20 us and
// returns per-cell checksums of that copy (so a test can see WHICH content was evaluated).  No NID arithmetic here.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "nid/nid_c.h"
#include "nid/nid_multi.h"

struct nid_multi {
  nid_config cfg;
  std::vector<double> points, bs;
  std::vector<uint8_t> im0, im1;
  std::vector<int32_t> cnt;
  std::vector<double> href;
};

namespace {
size_t npix(const nid_multi *m) { return (size_t)m->cfg.rows * m->cfg.cols; }
int ncell(const nid_multi *m) { return m->cfg.cell_num * m->cfg.cell_num; }
double cell_sum_u8(const nid_multi *m, const std::vector<uint8_t> &im, int c) {
  const int rb = m->cfg.rows / m->cfg.cell_num, cb = m->cfg.cols / m->cfg.cell_num, ci = c / m->cfg.cell_num, cj = c % m->cfg.cell_num;
  double s = 0;
  for (int r = ci * rb; r < (ci + 1) * rb; r++)
    for (int q = cj * cb; q < (cj + 1) * cb; q++) s += im[(size_t)r * m->cfg.cols + q];
  return s;
}
}  // namespace

extern "C" {

const char *nid_status_string(int) { return "stub"; }
int nid_set_reference_image_f64(const double *im, int64_t n, uint8_t *out) {
  for (int64_t i = 0; i < n; i++) {
    const double v = im[i];
    if (!(v >= 0.0 && v <= 255.0) || v != (double)(int)v) return NID_ERR_UNSUPPORTED;
    out[i] = (uint8_t)v;
  }
  return NID_OK;
}
int nid_backproject(const double *depth, const double *, double fx, double fy, double cx, double cy, int32_t rows, int32_t cols, int32_t,
                    double *points3d) {
  for (int r = 0; r < rows; r++)
    for (int c = 0; c < cols; c++) {
      const size_t i = (size_t)r * cols + c;
      const double z = depth[i];
      points3d[3 * i] = z * (c - cx) / fx; points3d[3 * i + 1] = z * (r - cy) / fy; points3d[3 * i + 2] = z;
    }
  return NID_OK;
}
int nid_backproject_release(void) { return NID_OK; }

int nid_multi_create(const nid_config *cfg, const int32_t *, int32_t, nid_multi **out) {
  nid_multi *m = new nid_multi;
  m->cfg = *cfg;
  *out = m;
  return NID_OK;
}
int nid_multi_create_rank(const nid_config *cfg, int32_t, int32_t, int32_t, nid_multi **out) { return nid_multi_create(cfg, nullptr, 1, out); }
int nid_multi_destroy(nid_multi *m) { delete m; return NID_OK; }
const char *nid_multi_last_error(const nid_multi *) { return ""; }
nid_ctx *nid_multi_shard(nid_multi *, int32_t) { return nullptr; }
int nid_multi_set_options(nid_multi *, int, int) { return NID_OK; }
int nid_multi_set_math_mode(nid_multi *, int) { return NID_OK; }
int nid_multi_set_href_nan_markers(nid_multi *, int) { return NID_OK; }
int nid_multi_set_launch_shape(nid_multi *, int, int) { return NID_OK; }
int nid_multi_set_resident(nid_multi *, int) { return NID_OK; }
int nid_multi_resident_pause(nid_multi *) { return NID_OK; }
int nid_comm_create_rank(const uint8_t *, int32_t, int32_t, int32_t, nid_comm **) { return NID_ERR_UNSUPPORTED; }
int nid_comm_create_local(const int32_t *, int32_t, nid_comm **) { return NID_ERR_UNSUPPORTED; }
int nid_comm_destroy(nid_comm *) { return NID_OK; }
int nid_multi_attach_comm(nid_multi *, nid_comm *) { return NID_ERR_UNSUPPORTED; }

int nid_multi_set_reference_points(nid_multi *m, const double *points3d, const uint8_t *im0) {
  m->points.assign(points3d, points3d + 3 * npix(m));
  m->im0.assign(im0, im0 + npix(m));
  return NID_OK;
}
int nid_multi_set_pair_u16(nid_multi *m, const uint16_t *, double, const uint8_t *im0, const uint8_t *im1, const double *, const double *,
                           const double *, int32_t *bs_counter, double *Href) {
  m->im0.assign(im0, im0 + npix(m));
  m->im1.assign(im1, im1 + npix(m));
  for (int c = 0; c < ncell(m); c++) { if (bs_counter) bs_counter[c] = 1000; if (Href) Href[c] = 1.0 + c; }
  return NID_OK;
}
int nid_multi_set_target_u8(nid_multi *m, const uint8_t *im1) { m->im1.assign(im1, im1 + npix(m)); return NID_OK; }
int nid_multi_compute_href_matrix(nid_multi *m, const double *, int32_t *bs_counter, double *Href, double *bs_value, int32_t *bs_index) {
  const size_t N = npix(m);
  for (int c = 0; c < ncell(m); c++) { bs_counter[c] = 1000; Href[c] = 1.0 + c; }
  if (bs_value) for (size_t i = 0; i < 4 * N; i++) bs_value[i] = (double)(m->im0[i / 4] + (i & 3));
  if (bs_index) for (size_t i = 0; i < N; i++) bs_index[i] = m->im0[i] & 7;
  m->bs.assign(bs_value ? bs_value : nullptr, bs_value ? bs_value + 4 * N : nullptr);
  return NID_OK;
}
int nid_multi_set_href_state(nid_multi *m, const int32_t *bs_counter, const double *Href, const double *bs_value, const int32_t *) {
  m->cnt.assign(bs_counter, bs_counter + ncell(m));
  m->href.assign(Href, Href + ncell(m));
  if (bs_value) m->bs.assign(bs_value, bs_value + 4 * npix(m));
  return NID_OK;
}
int nid_multi_evaluate_matrix(nid_multi *m, const double *pose, int want_jac, double *Ht, double *Hj, double *, double *der) {
  std::this_thread::sleep_for(std::chrono::microseconds(20));  // (the device's turn: the hash workers run meanwhile)
  if (m->im1.size() != npix(m) || m->im0.size() != npix(m)) return NID_ERR_STATE;
  for (int c = 0; c < ncell(m); c++) {
    Ht[c] = -cell_sum_u8(m, m->im1, c);                 // what is "resident" decides the answer
    Hj[c] = -(cell_sum_u8(m, m->im0, c) + pose[12]);
    if (want_jac && der) for (int k = 0; k < 6; k++) der[6 * c + k] = Ht[c] * (k + 1);
  }
  return NID_OK;
}

}  // extern "C"
