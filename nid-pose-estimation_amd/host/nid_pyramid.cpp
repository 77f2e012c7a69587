// nid_pyramid.cpp -- coarse-to-fine schedule around the reference's single-level optimisation
// (SURVEY.md section 8 row f1 / BASELINE.json configs[4]).  The reference has NO pyramid
// (NID_pose_estimation.cpp:339-351 only hints at a multi-round scheme with its unused
// its[4] / chi2threshold[4]); the definition below is this repository's own and is restated in
// numpy by oracle/oracle_py.py (pyr_down_u8 / pyr_down_depth_u16 / pyramid_lm), which is what
// the parity tests compare against.
//
// Level l+1 from level l (rows, cols even):
//   * images: 2x2 box mean of the u8 values, rounded half up: (a + b + c + d + 2) >> 2;
//   * depth:  mean of the VALID samples of the 2x2 block (valid as in CudaPoints3d.cu:12:
//             0.01 <= metres <= 100), rounded half up in u16 counts; 0 (invalid) if none is valid;
//   * intrinsics: fx/2, fy/2, cx' = (cx - 0.5)/2, cy' = (cy - 0.5)/2 (pixel j of level l+1 covers
//             pixels 2j, 2j+1 of level l, centre 2j + 0.5);
//   * cells:  cell_num >> 1, so a cell keeps its pixel count (A: 16/8/4 cells of 30x40 px) and the
//             reference's "< 300 in-frame pixels -> inactive" rule keeps its meaning.
// The optimisation runs `iterations` LM iterations per level from the coarsest level to level 0,
// each level starting from the previous level's pose; every level is the unchanged single-level
// problem (nid_host_run_lm), i.e. the same operators and the same HIP kernels.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "nid/nid_c.h"
#include "nid_pose_problem.h"

extern "C" {

void nid_pyr_down_u8(const uint8_t *src, int rows, int cols, uint8_t *dst) {
  const int r2 = rows / 2, c2 = cols / 2;
  for (int r = 0; r < r2; r++) {
    const uint8_t *a = src + (size_t)(2 * r) * cols, *b = a + cols;
    uint8_t *o = dst + (size_t)r * c2;
    for (int c = 0; c < c2; c++)
      o[c] = (uint8_t)((a[2 * c] + a[2 * c + 1] + b[2 * c] + b[2 * c + 1] + 2) >> 2);
  }
}

void nid_pyr_down_depth_u16(const uint16_t *src, int rows, int cols, double depth_factor, uint16_t *dst) {
  const int r2 = rows / 2, c2 = cols / 2;
  for (int r = 0; r < r2; r++) {
    const uint16_t *a = src + (size_t)(2 * r) * cols, *b = a + cols;
    uint16_t *o = dst + (size_t)r * c2;
    for (int c = 0; c < c2; c++) {
      const uint16_t v[4] = {a[2 * c], a[2 * c + 1], b[2 * c], b[2 * c + 1]};
      unsigned sum = 0, n = 0;
      for (int k = 0; k < 4; k++) {
        const double z = (double)v[k] * depth_factor;
        if (!(z < 0.01 || z > 100)) { sum += v[k]; n++; }
      }
      o[c] = n ? (uint16_t)((2 * sum + n) / (2 * n)) : (uint16_t)0;
    }
  }
}

int nid_host_run_pyramid_lm(const nid_pose_problem *pb, int levels, double *pose7_inout, nid_host_lm_record *trace,
                            int max_trace_per_level, int *done_per_level, char *log_buf, int log_cap) {
  if (!pb || !pose7_inout || levels < 1 || levels > 8) return -1;
  if ((pb->rows % (1 << (levels - 1))) || (pb->cols % (1 << (levels - 1))) || (pb->cell_num >> (levels - 1)) < 1 ||
      (pb->cell_num % (1 << (levels - 1))))
    return -2;
  struct Level {
    int rows, cols, cell;
    double fx, fy, cx, cy;
    std::vector<uint8_t> im0, im1;
    std::vector<uint16_t> depth;
  };
  std::vector<Level> L(levels);
  L[0].rows = pb->rows; L[0].cols = pb->cols; L[0].cell = pb->cell_num;
  L[0].fx = pb->fx; L[0].fy = pb->fy; L[0].cx = pb->cx; L[0].cy = pb->cy;
  const size_t N0 = (size_t)pb->rows * pb->cols;
  L[0].im0.assign(pb->im0, pb->im0 + N0);
  L[0].im1.assign(pb->im1, pb->im1 + N0);
  L[0].depth.assign(pb->depth_u16, pb->depth_u16 + N0);
  for (int l = 1; l < levels; l++) {
    const Level &p = L[l - 1];
    Level &q = L[l];
    q.rows = p.rows / 2; q.cols = p.cols / 2; q.cell = p.cell / 2;
    q.fx = p.fx / 2; q.fy = p.fy / 2; q.cx = (p.cx - 0.5) / 2; q.cy = (p.cy - 0.5) / 2;
    const size_t n = (size_t)q.rows * q.cols;
    q.im0.resize(n); q.im1.resize(n); q.depth.resize(n);
    nid_pyr_down_u8(p.im0.data(), p.rows, p.cols, q.im0.data());
    nid_pyr_down_u8(p.im1.data(), p.rows, p.cols, q.im1.data());
    nid_pyr_down_depth_u16(p.depth.data(), p.rows, p.cols, pb->depth_factor, q.depth.data());
  }
  std::string log;
  std::vector<char> lbuf(16384);
  int total = 0;
  for (int l = levels - 1; l >= 0; l--) {
    nid_pose_problem q = *pb;
    q.rows = L[l].rows; q.cols = L[l].cols; q.cell_num = L[l].cell;
    q.fx = L[l].fx; q.fy = L[l].fy; q.cx = L[l].cx; q.cy = L[l].cy;
    q.im0 = L[l].im0.data(); q.im1 = L[l].im1.data(); q.depth_u16 = L[l].depth.data();
    nid_host_lm_record *tr = trace ? trace + (size_t)(levels - 1 - l) * max_trace_per_level : nullptr;
    const int done = nid_host_run_lm(&q, pose7_inout, tr, tr ? max_trace_per_level : 0, lbuf.data(), (int)lbuf.size());
    char head[96];
    std::snprintf(head, sizeof(head), "---- pyramid level %d: %dx%d, %dx%d cells ----\n", l, q.cols, q.rows, q.cell_num,
                  q.cell_num);
    log += head;
    log += lbuf.data();
    if (done_per_level) done_per_level[levels - 1 - l] = done;
    if (done < 0) { total = done; break; }
    total += done;
  }
  if (log_buf && log_cap > 0) std::snprintf(log_buf, (size_t)log_cap, "%s", log.c_str());
  return total;
}

int nid_host_standard_property(const nid_pose_problem *pb, const double *pose7, double *final_nid, char *log_buf,
                               int log_cap) {
  if (!pb || !pose7) return -1;
  nid_config cfg;
  std::memset(&cfg, 0, sizeof(cfg));
  cfg.rows = pb->rows; cfg.cols = pb->cols; cfg.cell_num = pb->cell_num; cfg.bin_num = 8; cfg.bs_degree = 3;
  cfg.fx = pb->fx; cfg.fy = pb->fy; cfg.cx = pb->cx; cfg.cy = pb->cy;
  nid_ctx *ctx = nullptr;
  if (nid_create(&cfg, &ctx) != NID_OK) return -3;
  const size_t N = (size_t)pb->rows * pb->cols;
  std::vector<double> depth(N);
  for (size_t i = 0; i < N; i++) depth[i] = (double)pb->depth_u16[i] * pb->depth_factor;  // convertTo(CV_64F, 1/5000), :106
  int rc = nid_set_reference_depth(ctx, depth.data(), pb->im0, pb->T_wc0_colmajor);
  if (rc == NID_OK) rc = nid_set_target_u8(ctx, pb->im1);
  const int ncell = pb->cell_num * pb->cell_num;
  std::vector<double> hr(ncell), hc(ncell), hj(ncell), nid(ncell), mi(ncell);
  double total = 0.0;
  if (rc == NID_OK) rc = nid_plain_nid(ctx, pose7, pb->bin_num, hr.data(), hc.data(), hj.data(), nid.data(), mi.data(), nullptr, &total);
  std::string log;
  if (rc == NID_OK) {
    char line[256];
    for (int c = 0; c < ncell; c++) {
      if (nid[c] != nid[c]) continue;  // < 300 in-frame pixels: the reference prints nothing for the cell
      std::snprintf(line, sizeof(line), "Href, current, joint from standard method is %g,%g,%g, MI %g, NID %g\n", hr[c], hc[c],
                    hj[c], mi[c], nid[c]);  // :481
      log += line;
    }
    std::snprintf(line, sizeof(line), "final nid is %g\n", total);  // :201
    log += line;
  } else {
    log = std::string("nid_plain_nid failed: ") + nid_last_error(ctx) + "\n";
  }
  nid_destroy(ctx);
  if (final_nid) *final_nid = total;
  if (log_buf && log_cap > 0) std::snprintf(log_buf, (size_t)log_cap, "%s", log.c_str());
  return rc == NID_OK ? 0 : -4;
}

}  // extern "C"
