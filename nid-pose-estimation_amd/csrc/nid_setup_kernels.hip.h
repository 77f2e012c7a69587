// nid_setup_kernels.hip.h -- the once-per-pair kernels: back-projection + tiling, the target image's margins, the reference
// stage at the initial pose (k_href), the plain-histogram program (k_plain_nid), untiling of the per-pixel outputs.
// Streaming kernels, not on the iteration path.  Included by ONE translation unit (nid_capi.hip), which launches them.
#pragma once

#include "nid_kernels.hip.h"

namespace nid {

// ---------------------------------------------------------------------------
// Setup: back-projection + tiling.  Calculate3DpointKernel (CudaPoints3d.cu:5-32)
// == Get3dPointAndIntensity (NID_pose_estimation.cpp:401-432) folded into the
// cell-major tile writer.  One thread per tile slot.
__global__ void k_tile(Geometry g, const double *__restrict__ depth,
                       const double *__restrict__ points_in, const uint8_t *__restrict__ im0,
                       const double *__restrict__ Twc /*col-major 16*/, Tiles t,
                       double *__restrict__ points_out /*3N or null*/) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)g.nloc * g.pstride;
  if (gid >= total) return;
  const int cl = (int)(gid / g.pstride);
  const int s = (int)(gid % g.pstride);
  double X = NAN, Y = NAN, Z = NAN;
  int jr = -1;
  uint8_t i0 = 0;
  if (s < g.ps) {
    const int c = g.cell_begin + cl * g.cell_stride;
    const int ci = c / g.cell_num, cj = c % g.cell_num;
    const int r = ci * g.rb + s / g.cb;
    const int col = cj * g.cb + s % g.cb;
    const long id = (long)r * g.cols + col;
    i0 = im0[id];
    bool valid;
    if (depth) {
      const double z = depth[id];
      valid = !(z < 0.01 || z > 100);  // CudaPoints3d.cu:12
      if (valid) {
        const double x0 = z * (col - g.cx) / g.fx;
        const double y0 = z * (r - g.cy) / g.fy;
        X = Twc[0] * x0 + Twc[4] * y0 + Twc[8] * z + Twc[12];
        Y = Twc[1] * x0 + Twc[5] * y0 + Twc[9] * z + Twc[13];
        Z = Twc[2] * x0 + Twc[6] * y0 + Twc[10] * z + Twc[14];
      }
      if (points_out) {
        points_out[3 * id] = X; points_out[3 * id + 1] = Y; points_out[3 * id + 2] = Z;
      }
    } else {
      X = points_in[3 * id]; Y = points_in[3 * id + 1]; Z = points_in[3 * id + 2];
      valid = !(isnan(X) || isnan(Y) || isnan(Z));  // computeH.cu:145
      if (!valid) { X = NAN; Y = NAN; Z = NAN; }
    }
    if (valid) {
      double obs = (double)i0;  // types_six_dof_expmap.cpp:553-559
      if (obs >= 255) obs = 254.999;
      const double bin_pos_ref = obs * (double)g.S / 255.0;
      jr = (int)floor(bin_pos_ref);
    }
  }
  t.X[gid] = X; t.Y[gid] = Y; t.Z[gid] = Z;
  t.JR[gid] = (int8_t)jr;
  t.I0[gid] = i0;
}

// The driver's depth map as it comes off the disk: u16 at 1/5000 m (NID_pose_estimation.cpp:73,106:
// depth.convertTo(depth, CV_64F, depth_factor) -- one IEEE multiplication per pixel, no contraction) -> the f64 metres
// k_tile / k_backproject_plain read.  nid_set_pair_u16: 0.6 MB cross PCIe instead of 2.5 MB.
__global__ void k_depth_u16(long n, const uint16_t *__restrict__ src, double factor, double *__restrict__ dst) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id < n) dst[id] = (double)src[id] * factor;
}

// Target image for the evaluation kernel: int16 copy of the u8 image with the extrapolated top / left margin
// (see Win).  dst is (rows + 1) x stride; one thread per destination element of the first `cols + 1` columns.
__global__ void k_im1_margins(int rows, int cols, int stride, const uint8_t *__restrict__ src, int16_t *__restrict__ dst) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)(rows + 1) * (cols + 1)) return;
  const int r = (int)(gid / (cols + 1)) - 1, c = (int)(gid % (cols + 1)) - 1;
  auto at = [&](int rr, int cc) { return (int)src[(size_t)rr * cols + cc]; };
  auto col_m1 = [&](int rr) { return cols > 1 ? 2 * at(rr, 0) - at(rr, 1) : at(rr, 0); };
  int v;
  if (r >= 0 && c >= 0) v = at(r, c);
  else if (r >= 0) v = col_m1(r);
  else if (c >= 0) v = rows > 1 ? 2 * at(0, c) - at(1, c) : at(0, c);
  else v = rows > 1 ? 2 * col_m1(0) - col_m1(1) : col_m1(0);  // corner: never read with a non-zero weight
  dst[(size_t)(r + 1) * stride + (c + 1)] = (int16_t)v;
}

// depth pixels that belong to no cell (rows/cols not divisible by cell_num, Q11)
// still need Calculate3Dpoint's output when the caller asks for points3d.
__global__ void k_backproject_plain(Geometry g, const double *__restrict__ depth,
                                    const double *__restrict__ Twc, double *__restrict__ pts) {
  const long id = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= (long)g.rows * g.cols) return;
  const int r = (int)(id / g.cols), col = (int)(id % g.cols);
  const double z = depth[id];
  double X = NAN, Y = NAN, Z = NAN;
  if (!(z < 0.01 || z > 100)) {
    const double x0 = z * (col - g.cx) / g.fx;
    const double y0 = z * (r - g.cy) / g.fy;
    X = Twc[0] * x0 + Twc[4] * y0 + Twc[8] * z + Twc[12];
    Y = Twc[1] * x0 + Twc[5] * y0 + Twc[9] * z + Twc[13];
    Z = Twc[2] * x0 + Twc[6] * y0 + Twc[10] * z + Twc[14];
  }
  pts[3 * id] = X; pts[3 * id + 1] = Y; pts[3 * id + 2] = Z;
}

// The reference stage's per-pixel outputs in the caller's layout (CudaComputeHref's bs_value / bs_index, image order):
// one thread per tile slot writes its pixel's four weights (the sign of the first is the evaluation kernel's knot flag:
// stripped) and its bin index.  Pixels of no cell of this context are not written.
__global__ void k_untile_bs(Geometry g, Tiles t, double *__restrict__ bsv /*4N or null*/, int *__restrict__ bsi /*N or null*/, int nan_rows) {
  const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long)g.nloc * g.pstride) return;
  const int cl = (int)(gid / g.pstride), s = (int)(gid % g.pstride);
  if (s >= g.ps) return;
  const int c = g.cell_begin + cl * g.cell_stride;
  const long id = (long)((c / g.cell_num) * g.rb + s / g.cb) * g.cols + (c % g.cell_num) * g.cb + s % g.cb;
  if (bsv) {
    const double4 w = *reinterpret_cast<const double4 *>(t.W + 4 * gid);
    // nan_rows (nid_set_href_nan_markers): the legacy operators' convention -- a pixel that is invalid or out of frame at
    // this pose reads NaN, NaN, NaN, NaN (CudaComputeHref.cu:82-87, 126-130); in-frame weights sum to 1, so an all-zero
    // row is exactly that set
    const bool none = nan_rows && w.x == 0.0 && w.y == 0.0 && w.z == 0.0 && w.w == 0.0;
    *reinterpret_cast<double4 *>(bsv + 4 * id) = none ? make_double4(NAN, NAN, NAN, NAN) : make_double4(fabs(w.x), w.y, w.z, w.w);
  }
  if (bsi) bsi[id] = (int)t.JR[gid];
}

// ---------------------------------------------------------------------------
// Setup: reference stage at the initial pose -- computeHref
// (types_six_dof_expmap.cpp:655-725) / CalculateHrefKernel
// (CudaComputeHref.cu:33-135) for one cell per workgroup.
template <int NT>
__global__ __launch_bounds__(NT) void k_href(Geometry g, Pose pose, Tiles t, int *__restrict__ Nc,
                                             double *__restrict__ Href, double hist_scale,
                                             double hist_inv_scale) {
  __shared__ unsigned long long hist[kMaxBins * kHistCopies];
  __shared__ double red[2 * (NT / 64)];
  const int cl = blockIdx.x, tid = threadIdx.x;
  const int copy = tid & (kHistCopies - 1);
  for (int i = tid; i < g.nb * kHistCopies; i += NT) hist[i] = 0ull;
  __syncthreads();
  const size_t base = (size_t)cl * g.pstride;
  int count = 0;
  for (int s = tid; s < g.pstride; s += NT) {
    const size_t gi = base + s;
    const int jr = t.JR[gi];
    double w[4] = {0.0, 0.0, 0.0, 0.0};
    if (jr >= 0) {
      double qx, qy, qz;
      xform_point(pose, t.X[gi], t.Y[gi], t.Z[gi], qx, qy, qz);
      const double u = g.fx * qx / qz + g.cx;
      const double v = g.fy * qy / qz + g.cy;
      if (u >= 0 && u + 3 <= g.cols && v >= 0 && v + 3 <= g.rows) {
        count++;
        double obs = (double)t.I0[gi];
        if (obs >= 255) obs = 254.999;
        const double bin_pos_ref = obs * (double)g.S / 255.0;
        double d[4];
        bspline4<false>(bin_pos_ref, jr, g.S, w, d);
#pragma unroll
        for (int k = 0; k < 4; k++) atomicAdd(&hist[(jr + k) * kHistCopies + copy], fx_encode(w[k], hist_scale));
      }
    }
    // the evaluation kernel's "tiny non-zero reference weights" flag rides in the sign of the first weight (hist_add)
    {
      const double wmin = fmin(w[0], w[3]);
      if (wmin < kTinyW && wmin != 0.0) w[0] = -w[0];
    }
#pragma unroll
    for (int k = 0; k < 4; k++) t.W[4 * gi + k] = w[k];  // slot-major: see load_tile_w
  }
  double v2[2] = {(double)count, 0.0};
  block_sum<NT, 2>(v2, red, tid);
  const int n_c = (int)v2[0];
  __syncthreads();
  // entropy of the reference histogram
  double term = 0.0;
  if (tid < g.nb) {
    unsigned long long acc = 0;
    for (int c = 0; c < kHistCopies; c++) acc += hist[tid * kHistCopies + ((c + tid) & (kHistCopies - 1))];
    const double p = ((double)(long long)acc * hist_inv_scale) / (double)n_c;
    if (!(p < kSigma)) term = p * log2(p);
  }
  double v1[2] = {term, 0.0};
  block_sum<NT, 2>(v1, red, tid);
  if (tid == 0) {
    Nc[cl] = n_c;
    Href[cl] = (n_c < 300) ? NAN : (0.0 - v1[0]);  // CudaComputeHref.cu:206-209
  }
}

// ---------------------------------------------------------------------------
// Plain-histogram NID of one cell per workgroup: NID::ComputeHref + NID::ComputeH of the reference's
// second program (NID_standard_property.cpp:342-485) -- same warp and bilinear sample, HARD binning
// floor(I * bins / 255), no B-spline, no Jacobian.  Counts are integers (u32 LDS atomics), so the
// histograms are exact; the entropies are summed by one thread in the reference's bin order.
// out[cell*6 + {0..5}] = H_ref, H_current, H_joint, nid, MI, n_in.  Cells with fewer than 300 in-frame
// pixels get NaN for H_current / H_joint / nid / MI (the reference returns early there and then reads an
// uninitialised nid_: not reproduced).
template <int NT>
__global__ __launch_bounds__(NT) void k_plain_nid(Geometry g, Pose pose, Tiles t, const uint8_t *__restrict__ im1,
                                                  int bins, double *__restrict__ out) {
  __shared__ unsigned h_ref[kMaxPlainBins], h_cur[kMaxPlainBins], h_joint[kMaxPlainBins * kMaxPlainBins];
  __shared__ unsigned n_in_s;
  const int cl = blockIdx.x, tid = threadIdx.x;
  for (int i = tid; i < bins; i += NT) { h_ref[i] = 0u; h_cur[i] = 0u; }
  for (int i = tid; i < bins * bins; i += NT) h_joint[i] = 0u;
  if (tid == 0) n_in_s = 0u;
  __syncthreads();
  const size_t base = (size_t)cl * g.pstride;
  for (int s = tid; s < g.pstride; s += NT) {
    const size_t gi = base + s;
    if (t.JR[gi] < 0) continue;  // invalid depth / padding: Get3dPointAndIntensity skips the pixel (:226-227)
    double qx, qy, qz;
    xform_point(pose, t.X[gi], t.Y[gi], t.Z[gi], qx, qy, qz);
    const double u = g.fx * qx / qz + g.cx;  // :357-358
    const double v = g.fy * qy / qz + g.cy;
    if (!(u >= 0 && u + 3 <= g.cols && v >= 0 && v + 3 <= g.rows)) continue;  // ob++
    double i0 = (double)t.I0[gi];
    if (i0 >= 255) i0 = 254.999;  // :370-373
    const int br = (int)floor(i0 * bins / 255.0);
    double ic = bilinear_u8(im1, g.cols, u, v);
    if (ic >= 255) ic = 254.999;  // :432-435
    if (ic < 0) ic = 0.0;
    const int bc = (int)floor(ic * bins / 255.0);
    atomicAdd(&h_ref[br], 1u);
    atomicAdd(&h_cur[bc], 1u);
    atomicAdd(&h_joint[br * bins + bc], 1u);
    atomicAdd(&n_in_s, 1u);
  }
  __syncthreads();
  if (tid == 0) {
    const unsigned n_in = n_in_s;
    const double n = (double)n_in;
    double Href = 0.0, Hc = 0.0, Hj = 0.0;
    for (int i = 0; i < bins; i++) {
      const double p = (double)h_ref[i] / n;
      if (p < kSigma) continue;
      Href -= p * log2(p);
    }
    double nid = NAN, mi = NAN;
    if (n_in >= 300u) {
      for (int i = 0; i < bins; i++) {
        const double p = (double)h_cur[i] / n;
        if (p < kSigma) continue;
        Hc -= p * log2(p);
      }
      for (int i = 0; i < bins * bins; i++) {
        const double p = (double)h_joint[i] / n;
        if (p < kSigma) continue;
        Hj -= p * log2(p);
      }
      nid = (2 * Hj - Href - Hc) / Hj;  // :469-470
      mi = Href + Hc - Hj;
      if (Href == 0.0 && Hc == 0.0 && Hj == 0.0) { mi = 0.0; nid = 0.0; }  // :474-477
    } else {
      Hc = NAN; Hj = NAN;
    }
    double *o = out + (size_t)cl * 6;
    o[0] = Href; o[1] = Hc; o[2] = Hj; o[3] = nid; o[4] = mi; o[5] = n;
  }
}

}  // namespace nid
