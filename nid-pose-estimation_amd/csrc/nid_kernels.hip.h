// nid_kernels.hip.h -- hand-written gfx950 kernels of the NID path.
//
// Design (DESIGN.md has the long form):
//  * one workgroup per cell (a cell is rb x cb pixels, 30x40 = 1200 in both
//    BASELINE configs), NT threads, PPT pixels per thread kept in registers
//    across the two phases;
//  * operands are read from a cell-major, pixel-minor SoA tile built once per
//    frame pair, so every wave issues fully coalesced 512-B (f64) rows;
//  * phase 1 (cost): warp -> bilinear -> target B-spline weights -> per-cell
//    histograms in LDS.  Bins are accumulated with ds_add_u64 in 64-bit fixed
//    point (exact integer adds: order-independent, bitwise reproducible) into
//    NC lane-interleaved copies so that the 32 lanes of an LDS lane group
//    never collide on a bank;
//  * entropies + weight tables W = -(1 + log2 p) are formed once per cell;
//  * phase 2 (Jacobian): per pixel two scalars s = sum Wj*wr*dw, t = sum Wc*dw
//    replace the reference's 6*bin^2 derivative tensor (same algebra, only the
//    summation order differs), then a 12-value wavefront + LDS reduction;
//  * a second 1-workgroup kernel applies Huber and reduces to the 6x6 system.
//
// Reference semantics restated here (never copied): CalculateProKernel /
// CalculateHKernel / CalculateDerKernel g2o/g2o/core/computeH.cu:93-368 and the
// CPU edge g2o/g2o/types/types_six_dof_expmap.cpp:381-637 (the parity target).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nid_bspline.h"

namespace nid {

constexpr int kHistCopies = 32;   // NC: lane-interleaved histogram copies
constexpr int kMaxBins = 16;
constexpr int kCellOut = 10;      // Hc, Hj, err, J[6], Nc
constexpr int kReducedLen = 32;
constexpr double kSigma = 1e-30;  // types_six_dof_expmap.h:281

struct Geometry {
  int rows, cols, cell_num, rb, cb;
  int ps;        // pixels per cell = rb*cb
  int pstride;   // ps rounded up to 64
  int cell_begin, nloc;
  int nb, S;
  double fx, fy, cx, cy;
};

struct Pose {
  double q[7];   // qx qy qz qw tx ty tz
  double M[12];  // rows of the 3x4 [R|t]: M[4*r + c]
  int mode;      // NID_XFORM_*
};

struct Tiles {
  double *X, *Y, *Z;  // [nloc*pstride]; NaN where depth invalid / padding
  double *W;          // 4 planes of nloc*pstride reference weights
  int8_t *JR;         // reference bin index, -1 = invalid depth / padding
  uint8_t *I0;        // reference intensity
};

struct EvalParams {
  Geometry g;
  Pose pose;
  Tiles t;
  const uint8_t *im1;
  const int *Nc;        // [nloc]
  const double *Href;   // [nloc]
  double *cellout;      // [nloc*kCellOut]
  int jac_cols;         // cols or cols-1
  double hist_scale, hist_inv_scale;
  // optional per-pixel dump (image order), null when disabled
  double *dbg_u, *dbg_v, *dbg_ic, *dbg_wc;
  int *dbg_jc;
};

// ---------------------------------------------------------------------------
__device__ __forceinline__ void xform_point(const Pose &P, double x, double y, double z,
                                            double &ox, double &oy, double &oz) {
  if (P.mode == 0) {
    // Eigen QuaternionBase::_transformVector as used by SE3Quat::map (se3quat.h:217-220)
    const double qx = P.q[0], qy = P.q[1], qz = P.q[2], qw = P.q[3];
    double uvx = qy * z - qz * y;
    double uvy = qz * x - qx * z;
    double uvz = qx * y - qy * x;
    uvx += uvx; uvy += uvy; uvz += uvz;
    const double cxx = qy * uvz - qz * uvy;
    const double cyy = qz * uvx - qx * uvz;
    const double czz = qx * uvy - qy * uvx;
    ox = (x + qw * uvx + cxx) + P.q[4];
    oy = (y + qw * uvy + cyy) + P.q[5];
    oz = (z + qw * uvz + czz) + P.q[6];
  } else {
    // computeH.cu:152-154
    ox = P.M[0] * x + P.M[1] * y + P.M[2] * z + P.M[3];
    oy = P.M[4] * x + P.M[5] * y + P.M[6] * z + P.M[7];
    oz = P.M[8] * x + P.M[9] * y + P.M[10] * z + P.M[11];
  }
}

// types_six_dof_expmap.h:310-328: (int) truncation, left-to-right sum
__device__ __forceinline__ double bilinear_u8(const uint8_t *__restrict__ im, int cols, double x,
                                              double y) {
  const int ix = (int)x;
  const int iy = (int)y;
  const double dx = x - ix;
  const double dy = y - iy;
  const double dxdy = dx * dy;
  const uint8_t *p = im + (size_t)iy * cols + ix;
  const double i00 = (double)p[0], i01 = (double)p[1];
  const double i10 = (double)p[cols], i11 = (double)p[cols + 1];
  return dxdy * i11 + (dy - dxdy) * i10 + (dx - dxdy) * i01 + (1 - dx - dy + dxdy) * i00;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Deterministic block sum of NV values; result broadcast to every thread.
// `red` holds NV * (NT/64) doubles and must not be in use.
template <int NT, int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *red, int tid) {
  constexpr int NW = NT / 64;
#pragma unroll
  for (int k = 0; k < NV; k++) v[k] = wave_sum(v[k]);
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; k++) red[wave * NV + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; k++) {
    double s = red[k];
    for (int w = 1; w < NW; w++) s += red[w * NV + k];
    v[k] = s;
  }
}

__device__ __forceinline__ void hist_add(unsigned long long *hist, int bin, int copy, double w,
                                         double scale) {
  // 64-bit fixed point: exact, order-independent accumulation (ds_add_u64)
  const long long q = __double2ll_rn(w * scale);
  atomicAdd(&hist[bin * kHistCopies + copy], (unsigned long long)q);
}

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
    const int c = g.cell_begin + cl;
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
  const size_t plane = (size_t)g.nloc * g.pstride;
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
        for (int k = 0; k < 4; k++) hist_add(hist, jr + k, copy, w[k], hist_scale);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) t.W[k * plane + gi] = w[k];
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
// The hot kernel: cost (+ Jacobian) of one cell per workgroup.
template <int NT, int PPT, bool JAC>
__global__ __launch_bounds__(NT) void k_eval(EvalParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Geometry &g = P.g;
  const int nb = g.nb;
  const int nbins = nb * nb + nb;  // [0,nb): target histogram, then joint row-major [ref][target]
  constexpr int NW = NT / 64;
  // LDS carve (all 16-B aligned): hist | tab | red
  unsigned long long *hist = reinterpret_cast<unsigned long long *>(smem);
  double *tab = reinterpret_cast<double *>(smem + (size_t)nbins * kHistCopies * 8);
  double *red = tab + ((nbins + 1) & ~1);

  const int cl = blockIdx.x, tid = threadIdx.x;
  const int n_c = P.Nc[cl];
  const double href = P.Href[cl];
  double *out = P.cellout + (size_t)cl * kCellOut;
  if (n_c < 300 || isnan(href)) {  // level-1 edge: never evaluated (computeH.cu:271-275)
    if (tid < kCellOut) out[tid] = (tid == kCellOut - 1) ? (double)n_c : NAN;
    return;
  }

  for (int i = tid; i < nbins * kHistCopies; i += NT) hist[i] = 0ull;

  const int copy = tid & (kHistCopies - 1);
  const size_t base = (size_t)cl * g.pstride;
  const size_t plane = (size_t)g.nloc * g.pstride;

  // per-pixel state carried from phase 1 to phase 2
  double s_x[PPT], s_y[PPT], s_iz[PPT], s_u[PPT], s_v[PPT];
  double s_wr[PPT][4], s_dw[PPT][4];
  int s_jr[PPT], s_jc[PPT];
  unsigned s_flags = 0;  // bit i: pixel i contributes to the Jacobian

  // ---- loads first (all independent), then compute -------------------------
  double lx[PPT], ly[PPT], lz[PPT];
#pragma unroll
  for (int i = 0; i < PPT; i++) {
    const int s = i * NT + tid;
    s_jr[i] = -1;
    if (s < g.pstride) {
      const size_t gi = base + s;
      s_jr[i] = P.t.JR[gi];
      lx[i] = P.t.X[gi]; ly[i] = P.t.Y[gi]; lz[i] = P.t.Z[gi];
#pragma unroll
      for (int k = 0; k < 4; k++) s_wr[i][k] = P.t.W[k * plane + gi];
    }
  }
  __syncthreads();  // histogram zeroed

  // ---- phase 1: warp, sample, target weights, histograms -------------------
#pragma unroll
  for (int i = 0; i < PPT; i++) {
    s_jc[i] = -1;
    if (s_jr[i] < 0) continue;
    double qx, qy, qz;
    xform_point(P.pose, lx[i], ly[i], lz[i], qx, qy, qz);
    // types_six_dof_expmap.cpp:562-563: fx * x / z + cx
    const double u = g.fx * qx / qz + g.cx;
    const double v = g.fy * qy / qz + g.cy;
    const bool inb = (u >= 0 && u + 3 <= g.cols && v >= 0 && v + 3 <= g.rows);
    double ic = NAN, wc[4] = {NAN, NAN, NAN, NAN};
    int jc = -1;
    if (inb) {
      ic = bilinear_u8(P.im1, g.cols, u, v);
      if (ic >= 255) ic = 254.999;
      if (ic < 0) ic = 0.0;
      const double pc = ic * ((double)nb - 3.0) / 255.0;
      jc = (int)floor(pc);
      double dw[4];
      bspline4<JAC>(pc, jc, g.S, wc, dw);
#pragma unroll
      for (int k = 0; k < 4; k++) hist_add(hist, jc + k, copy, wc[k], P.hist_scale);
      const int jr = s_jr[i];
#pragma unroll
      for (int m = 0; m < 4; m++)
#pragma unroll
        for (int k = 0; k < 4; k++)
          hist_add(hist, nb + (jr + m) * nb + jc + k, copy, s_wr[i][m] * wc[k], P.hist_scale);
      if (JAC) {
        // linearizeOplus recomputes u as fx*(x/z)+cx (types_six_dof_expmap.cpp:407-422, Q6)
        const double iz = 1.0 / qz;
        const double uj = g.fx * (qx / qz) + g.cx;
        const double vj = g.fy * (qy / qz) + g.cy;
        const bool jin = (uj >= 0 && uj + 3 <= P.jac_cols && vj >= 0 && vj + 3 <= g.rows);
        if (jin) s_flags |= (1u << i);
        s_x[i] = qx; s_y[i] = qy; s_iz[i] = iz; s_u[i] = uj; s_v[i] = vj;
#pragma unroll
        for (int k = 0; k < 4; k++) s_dw[i][k] = dw[k];
      }
    }
    s_jc[i] = jc;
    if (P.dbg_u) {
      const int s = i * NT + tid;
      const int c = g.cell_begin + cl;
      const int r = (c / g.cell_num) * g.rb + s / g.cb;
      const int col = (c % g.cell_num) * g.cb + s % g.cb;
      const size_t id = (size_t)r * g.cols + col;
      P.dbg_u[id] = u; P.dbg_v[id] = v; P.dbg_ic[id] = ic; P.dbg_jc[id] = jc;
#pragma unroll
      for (int k = 0; k < 4; k++) P.dbg_wc[4 * id + k] = wc[k];
    }
  }
  __syncthreads();

  // ---- fold the copies, probabilities, entropies, weight tables ------------
  double ent[2] = {0.0, 0.0};  // sum p*log2(p): target, joint
  for (int b = tid; b < nbins; b += NT) {
    unsigned long long acc = 0;
#pragma unroll 8
    for (int c = 0; c < kHistCopies; c++) acc += hist[b * kHistCopies + ((c + b) & (kHistCopies - 1))];
    // CalculateHKernel: pro /= bs_counter (computeH.cu:277-291), N_c of the initial pose (Q1)
    const double p = ((double)(long long)acc * P.hist_inv_scale) / (double)n_c;
    double w = 0.0;
    if (!(p < kSigma)) {
      const double l = log2(p);
      w = -(1.0 + l);  // Q9: (1 + log2 p)
      if (b < nb) ent[0] += p * l; else ent[1] += p * l;
    }
    tab[b] = w;
  }
  block_sum<NT, 2>(ent, red, tid);  // contains the barrier that publishes tab[]
  const double Hc = 0.0 - ent[0];
  const double Hj = 0.0 - ent[1];

  if (!JAC) {
    if (tid == 0) {
      out[0] = Hc; out[1] = Hj;
      out[2] = (2 * Hj - href - Hc) / Hj;  // types_six_dof_expmap.h:227
      out[kCellOut - 1] = (double)n_c;
    }
    return;
  }

  // ---- phase 2: Jacobian ----------------------------------------------------
  const double kappa = (double)g.S / 255.0;  // d_mi_i, types_six_dof_expmap.cpp:393
  double acc[12];
#pragma unroll
  for (int n = 0; n < 12; n++) acc[n] = 0.0;
#pragma unroll
  for (int i = 0; i < PPT; i++) {
    if (!(s_flags & (1u << i))) continue;
    const double u = s_u[i], v = s_v[i];
    const double gx = (bilinear_u8(P.im1, g.cols, u + 1, v) - bilinear_u8(P.im1, g.cols, u - 1, v)) / 2;
    const double gy = (bilinear_u8(P.im1, g.cols, u, v + 1) - bilinear_u8(P.im1, g.cols, u, v - 1)) / 2;
    const double x = s_x[i], y = s_y[i], invz = s_iz[i];
    const double invz_2 = invz * invz;
    // types_six_dof_expmap.cpp:438-450
    double Ju[6], Jv[6];
    Ju[0] = -x * y * invz_2 * g.fx;
    Ju[1] = (1 + (x * x * invz_2)) * g.fx;
    Ju[2] = -y * invz * g.fx;
    Ju[3] = invz * g.fx;
    Ju[4] = 0;
    Ju[5] = -x * invz_2 * g.fx;
    Jv[0] = -(1 + y * y * invz_2) * g.fy;
    Jv[1] = x * y * invz_2 * g.fy;
    Jv[2] = x * invz * g.fy;
    Jv[3] = 0;
    Jv[4] = invz * g.fy;
    Jv[5] = -y * invz_2 * g.fy;
    const int jr = s_jr[i], jc = s_jc[i];
    // s = sum_{k,m} Wj[jr+k][jc+m] * (wr[k]*dw[m]*kappa) ; t = sum_m Wc[jc+m] * (dw[m]*kappa)
    double s = 0.0, tt = 0.0;
#pragma unroll
    for (int m = 0; m < 4; m++) tt += tab[jc + m] * (s_dw[i][m] * kappa);
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int m = 0; m < 4; m++)
        s += tab[nb + (jr + k) * nb + jc + m] * (s_wr[i][k] * s_dw[i][m] * kappa);
#pragma unroll
    for (int n = 0; n < 6; n++) {
      const double dI = gx * Ju[n] + gy * Jv[n];  // d_i_pose, :460
      acc[n] += s * dI;
      acc[6 + n] += tt * dI;
    }
  }
  __syncthreads();  // `red` is reused
  block_sum<NT, 12>(acc, red, tid);
  if (tid < 6) {
    // CalculateDerKernel tail (computeH.cu:358-366) == types_six_dof_expmap.cpp:521-528
    const double d_hj = acc[tid] / (double)n_c;
    const double d_hl = acc[6 + tid] / (double)n_c;
    const double inv_square_hj = 1.0 / (Hj * Hj);
    out[3 + tid] = (d_hj * (Hc + href) - d_hl * Hj) * inv_square_hj;
  }
  if (tid == 0) {
    out[0] = Hc; out[1] = Hj;
    out[2] = (2 * Hj - href - Hc) / Hj;
    out[kCellOut - 1] = (double)n_c;
  }
}

// ---------------------------------------------------------------------------
// Huber + per-cell quadratic form + reduction to the 6x6 normal equations:
// BaseUnaryEdge::constructQuadraticForm (base_unary_edge.hpp:43-72),
// RobustKernelHuber::robustify (robust_kernel_impl.cpp:77-91, float dsqr :84 of .h)
// reduced[0]=chi2, [1..6]=b, [7..27]=H upper triangle, [28]=n_active.
__global__ __launch_bounds__(256) void k_reduce(const double *__restrict__ cellout, int nloc,
                                                int have_jac, double delta, float dsqr,
                                                double *__restrict__ reduced) {
  __shared__ double red[29 * 4];
  const int tid = threadIdx.x;
  double v[29];
#pragma unroll
  for (int k = 0; k < 29; k++) v[k] = 0.0;
  for (int c = tid; c < nloc; c += 256) {
    const double *o = cellout + (size_t)c * kCellOut;
    const double e = o[2];
    if (isnan(e)) continue;
    const double e2 = e * e;
    double rho0, rho1;
    if (e2 <= dsqr) { rho0 = e2; rho1 = 1.0; }
    else {
      const double sqrte = sqrt(e2);
      rho0 = 2 * sqrte * delta - dsqr;
      rho1 = delta / sqrte;
    }
    v[0] += rho0;
    v[28] += 1.0;
    if (have_jac) {
      double J[6];
#pragma unroll
      for (int n = 0; n < 6; n++) J[n] = o[3 + n];
#pragma unroll
      for (int n = 0; n < 6; n++) v[1 + n] -= (rho1 * J[n]) * e;
      int idx = 7;
#pragma unroll
      for (int a = 0; a < 6; a++)
#pragma unroll
        for (int b = a; b < 6; b++) v[idx++] += (J[a] * rho1) * J[b];
    }
  }
  block_sum<256, 29>(v, red, tid);
  if (tid < 29) {
    double r = 0.0;
#pragma unroll
    for (int k = 0; k < 29; k++) if (k == tid) r = v[k];
    reduced[tid] = r;
  } else if (tid < kReducedLen) {
    reduced[tid] = 0.0;
  }
}

}  // namespace nid
