// nid_kernels.hip.h -- hand-written gfx950 kernels of the NID path: shared types and device helpers, eval_cell (one cell at one
// pose), the launched kernels k_eval2 / k_repair.  Siblings: nid_setup_kernels.hip.h (once per pair), nid_resident_kernels.hip.h
// (the kernels that stay on the device between requests).
//
// Design (DESIGN.md has the long form):
//  * one workgroup per cell and candidate pose (a cell is rb x cb pixels, 30x40 = 1200 in both
//    BASELINE configs), NT threads, a run-time loop over rounds of NT pixels; nothing per pixel is
//    kept from the cost phase to the Jacobian phase except one decision bit per round and lane
//    (the latency form of the 512 / 1024-thread shapes keeps a pixel's gradient and projection);
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
//    summation order differs), then six sums over the workgroup (through LDS, or DPP + LDS);
//  * each cell's Huber-weighted quadratic form is written write-through and the
//    workgroup that arrives last sums them (fixed order) into the 6x6 system.
//
// Reference semantics restated here (never copied): CalculateProKernel /
// CalculateHKernel / CalculateDerKernel g2o/g2o/core/computeH.cu:93-368 and the
// CPU edge g2o/g2o/types/types_six_dof_expmap.cpp:381-637 (the parity target).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <type_traits>

#include "nid_bspline.h"

// 1: clamped target samples are summed per reference bin and folded in with their constant weights (kClampBins)
#ifndef NID_CLAMP_BINS
#define NID_CLAMP_BINS 1
#endif
// 1: the Jacobian phase's second pass takes its rare lanes from raremask (EXPERIMENT: 122 VGPRs under the 128-register
// budget of the EXT kernels, i.e. four waves per SIMD); 0: it classifies again
#ifndef NID_JAC_SECOND_MASK
#define NID_JAC_SECOND_MASK 0
#endif
// 1: wave-uniform fine-level adds of the target histogram are added once per wave (hist_add); 0: one atomic per lane
#ifndef NID_LO_AGGREGATE
#define NID_LO_AGGREGATE 1
#endif

namespace nid {

// NC: lane-interleaved histogram copies.  The LDS services a 64-bit access in four groups of 16
// consecutive lanes (MI355X_MICROARCH.md, LDS table): 16 copies give every lane of a group its own
// address and its own pair of banks, so more copies buy nothing and cost zeroing + fold time.
constexpr int kHistCopies = 16;
// k_eval2's copies, by workgroup shape and -- round 6 -- by the bin count the kernel is specialised for: NID_HIST_COPIES_NB10
// copies in the 10-bin instantiations (`nb_spec` = the template's NB; 0 = the generic kernels: always kHistCopies)
#ifndef NID_HIST_COPIES_NB10
#define NID_HIST_COPIES_NB10 16
#endif
constexpr int eval_hist_copies(int nt, int nb_spec = 0) { return nt >= 128 ? (nb_spec == 10 ? NID_HIST_COPIES_NB10 : kHistCopies) : 8; }
constexpr int kMaxBins = 16;
constexpr int kMaxPlainBins = 32;  // plain-histogram mode (k_plain_nid)
constexpr int kCellOut = 10;      // Hc, Hj, err, J[6], Nc
constexpr int kReducedLen = 32;
constexpr double kSigma = 1e-30;  // types_six_dof_expmap.h:281

struct Geometry {
  int rows, cols, cell_num, rb, cb;
  int ps;        // pixels per cell = rb*cb
  int pstride;   // ps rounded up to 64
  int cell_begin, nloc, cell_stride;  // this context owns cells cell_begin + cl * cell_stride, cl = 0 .. nloc-1
  int nb, S;
  double fx, fy, cx, cy;
};

struct Pose {
  double q[7];   // qx qy qz qw tx ty tz
  double M[12];  // rows of the 3x4 [R|t]: M[4*r + c]
  int mode;      // NID_XFORM_*
};
constexpr unsigned kPoseQuatDwords = 14;  // q[7] leads the record

struct Tiles {
  double *X, *Y, *Z;  // [nloc*pstride]; NaN where depth invalid / padding
  double *W;          // [nloc*pstride][4] reference weights (slot-major: a slot's four weights are adjacent); the first
                      // one carries a flag in its sign (negative: see hist_add) -- its value is the magnitude
  int8_t *JR;         // reference bin index, -1 = invalid depth / padding
  uint8_t *I0;        // reference intensity
};

constexpr int kMaxBatch = 16;     // candidate poses of one launch whose arguments ride in the kernel arguments
constexpr int kMaxBatchExt = 256;  // ... or, beyond that, in a device-resident SlotArgs array (EvalParams::slots_ext)
// FAST math mode: per span 4 basis functions x (4 value + 3 derivative) polynomial coefficients in
// t = u - floor(u):  B_k = ((a3 t + a2) t + a1) t + a0,  B_k' = (d2 t + d1) t + d0
constexpr int kCoefRow = 28;

// what differs between the poses of one launch: the pose and where its results go
struct SlotArgs {
  Pose pose;
  double *cellout;                 // [nloc*kCellOut]
  double *quad;                    // [nloc*32] per-cell quadratic-form blocks (sc1 traffic)
  double *gpart;                   // [ngroups*32] group sums (second reduction level)
  unsigned *ticket;                // [0] top arrival counter, [1+g] group counters; zero between launches
  double *out_reduced;             // 32 doubles: device memory or pinned host memory
  unsigned long long *host_seq;    // pinned host word that receives launch_seq (or null)
  unsigned long long launch_seq;
  int cellout_host;                // cellout is mapped pinned host memory (the blocking per-cell calls): publish it at system scope
  int host_quad;                   // DIRECT launch (one pose, the host is waiting for it): no in-launch reduction.  1: `quad` is
                                   // mapped pinned host memory, [2][nloc][kDirectRec]: every cell's workgroup writes its two
                                   // records (residual, Jacobian) straight there and the HOST forms and sums the quadratic
                                   // forms; 2: only the per-cell outputs, to `cellout` (mapped pinned host memory too);
                                   // 3: GROUP-DIRECT -- the in-launch tail up to the group sums, which go to `out_reduced` =
                                   // mapped pinned host memory [ngroups][32] (the host adds the groups up)
};
// A cell's records of a DIRECT launch, one 64-byte line each, every line written by ONE store instruction (a line
// written in two instalments sat in a write combiner for ~30 us):
//   residual record [cl]:        err | 1.0 active / 0.0 level-1 edge | 0 x 6   -- sent when the cost phase is over
//   Jacobian record [nloc + cl]: J[6] | 0 | 0                                   -- cost + Jacobian launches only
constexpr int kDirectRec = 8;

struct EvalParams {
  Geometry g;
  Tiles t;
  const int16_t *im1s;  // target image as int16, (rows+1) x im1_stride, element (r, c) at [(r+1)*stride + c+1]; see Win
  int im1_stride;
  const int *Nc;        // [nloc]
  const double *Href;   // [nloc]
  int jac_cols;         // cols or cols-1
  // FAST border bounds, as unsigned range checks on the HIGH DWORD of u, v (a non-negative double orders like its bit
  // pattern; a negative one or a NaN falls outside every range): `inside` iff hi(u) - hu_lo <= hu_span, i.e. u in
  // [eps', cols-3-eps'] with eps' >= kBorderEps at the granularity of the high dword (2^-11 px near 640: a few more
  // samples take the exact decisions).  Round 4: the f64 bounds were twelve scalar registers inside both pixel loops, and
  // the kernels are short of those (what does not fit is parked in vector lanes and read back inside the loops).
  unsigned hu_lo, hu_span, hv_span, hj_span;  // u: [hu_lo, hu_lo + hu_span], v: [hu_lo, hu_lo + hv_span], linearizeOplus' u: hj_span
  double hist_dn, hist_inv_scale;  // 2^(s - 562), 2^-s: see fx_bits (2^(s - 1074), for the plain weights, is hist_dn * kWcPre)
  // fused Huber + 6x6 reduction
  double huber_delta;
  float huber_dsqr;
  int group_size;                  // cells per first-level group
  int batch;                       // poses in this launch (slot[0..batch-1])
  const double *ctab;              // FAST mode: per-span B-spline polynomial coefficients [S][kCoefRow], the value
                                   // coefficients (entries 0..3 of every 7-entry row) times kWcPre
  SlotArgs slot[kMaxBatch];        // per-pose arguments of launches of <= kMaxBatch poses (3.7 KB of kernel arguments)
  const SlotArgs *slots_ext;       // larger launches: the same records in device memory (copied in-stream); else null
  // optional per-pixel dump (image order), null when disabled
  double *dbg_u, *dbg_v, *dbg_ic, *dbg_wc;
  int *dbg_jc;
  int dbg_jac;  // 0: the dump describes the cost phase (u, v, ic, jc, wc[4]); 1: the Jacobian phase (gx, gy, pc, jc, dw[4])
  unsigned long long *repair_count;  // cells and poses that ran the repair pass (kLinFlagW), or null
  unsigned *repair_queue;            // [0] entries, [1] k_repair's exit ticket, [2 + 3 i ..] = pose << 16 | cell, the fold's repair set (two words): see k_repair
  // optional phase stamps (s_memtime) of wave 0 of every workgroup: [nloc][10] (8 phase stamps + s_memrealtime at start/end); diagnostic runs only
  long long *dbg_stamps;
};

// Stamp k: ordered AFTER the values passed as dependencies and BEFORE any later
// volatile asm / memory operation.  One asm statement per the guide.
__device__ __forceinline__ void nid_stamp(long long *buf, int cell, int k, double d0 = 0.0, double d1 = 0.0,
                                          double d2 = 0.0, double d3 = 0.0) {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : "v"(d0), "v"(d1), "v"(d2), "v"(d3) : "memory");
  if (threadIdx.x == 0) buf[(size_t)cell * 10 + k] = (long long)t;
  if (threadIdx.x == 0 && (k == 0 || k == 7)) buf[(size_t)cell * 10 + (k == 0 ? 8 : 9)] = (long long)wall_clock64();
}
#define NID_STAMP(k, ...)                                    \
  do {                                                       \
    if (DBG && P.dbg_stamps && pose_idx == 0) nid_stamp(P.dbg_stamps, cl, (k), ##__VA_ARGS__); \
  } while (0)

static_assert(sizeof(EvalParams) <= 4096, "kernel arguments are limited to 4 KiB: lower kMaxBatch");

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

// 4x4 target-image window around a warped pixel.  Every tap of the five bilinear samples a pixel needs --
// (u,v), (u+-1,v), (u,v+-1) -- lies inside it, so the image is read once per pixel and phase (4 unaligned
// 8-byte loads) and nothing else of the pixel touches global memory.
// The kernels read the target image as int16 with a one-pixel margin on the top and on the left that holds the
// LINEAR EXTRAPOLATION 2*I[0] - I[1] of the first two rows / columns (k_im1_margins).  Reason: the reference
// samples bil(u - 1, v) for the gradient, and for u < 1 its (int) truncation toward zero turns that sample into
// an extrapolation from columns 0 and 1 (types_six_dof_expmap.h:312-317, Q3); interpolating between the margin
// and column 0 is the same number, so the fixed-tap gradient below holds for the first row / column too and
// the window origin ((int)u - 1, (int)v - 1) needs no clamping.  A row is two dwords: taps 0,1 | taps 2,3.
typedef uint2 WRow;
struct Win {
  WRow r0, r1, r2, r3;
  int wx, wy;
};

// window loads as uniform 64-bit base + zero-extended 32-bit lane BYTE offset (the saddr + voffset form of global_load:
// no per-lane 64-bit address arithmetic, one VGPR per address instead of two); `elem` counts int16 elements
__device__ __forceinline__ WRow load_row_at(const int16_t *base, unsigned elem) {
  WRow v;
  __builtin_memcpy(&v, reinterpret_cast<const char *>(base) + (elem << 1), 8);
  return v;
}
__device__ __forceinline__ unsigned load_u32_at(const int16_t *base, unsigned elem) {
  unsigned v;
  __builtin_memcpy(&v, reinterpret_cast<const char *>(base) + (elem << 1), 4);
  return v;
}

__device__ __forceinline__ WRow win_row(const Win &w, int k) {
  WRow a, b;
  a.x = (k & 1) ? w.r1.x : w.r0.x; a.y = (k & 1) ? w.r1.y : w.r0.y;
  b.x = (k & 1) ? w.r3.x : w.r2.x; b.y = (k & 1) ? w.r3.y : w.r2.y;
  WRow r;
  r.x = (k & 2) ? b.x : a.x; r.y = (k & 2) ? b.y : a.y;
  return r;
}

// tap k (0..3, taken modulo 4) of a window row, sign-extended
__device__ __forceinline__ int tap_i(WRow row, int k) {
  const unsigned word = (k & 2) ? row.y : row.x;
  return __builtin_amdgcn_sbfe((int)word, (unsigned)(k & 1) * 16u, 16u);
}
__device__ __forceinline__ double win_tap(WRow row, int k) { return (double)tap_i(row, k); }

// the two window rows a sample at ordinate y interpolates between, and its dy
struct RowPair {
  WRow ra, rb;
  double dy;
};
__device__ __forceinline__ RowPair win_rows(const Win &w, double y) {
  const int iy = (int)y;
  RowPair r;
  r.dy = y - iy;
  int ky = iy - w.wy;
  ky = min(max(ky, 0), 3);
  r.ra = win_row(w, ky);
  r.rb = win_row(w, min(ky + 1, 3));
  return r;
}
// types_six_dof_expmap.h:310-328 on register taps (same operation order)
__device__ __forceinline__ double bilinear_rows(const RowPair &r, int wx, double x) {
  const int ix = (int)x;
  const double dx = x - ix;
  const double dy = r.dy;
  const double dxdy = dx * dy;
  int kx = ix - wx;
  kx = min(max(kx, 0), 3);
  const double i00 = win_tap(r.ra, kx), i01 = win_tap(r.ra, kx + 1);
  const double i10 = win_tap(r.rb, kx), i11 = win_tap(r.rb, kx + 1);
  return dxdy * i11 + (dy - dxdy) * i10 + (dx - dxdy) * i01 + (1 - dx - dy + dxdy) * i00;
}

// x - (double)(int)x for x >= 0 (an in-frame coordinate, a bin position): v_fract_f64 is that difference -- exactly, the
// subtraction of the integer part of a non-negative double is exact either way -- in ONE instruction instead of a
// conversion back and a subtraction (round 5; tools/ubench/valu_wallclock.hip has the rates).  Lanes whose x is negative
// or NaN get another value than the subtraction's; they hold no sample.
#ifndef NID_USE_FRACT
#define NID_USE_FRACT 1
#endif
__device__ __forceinline__ double frac_nonneg(double x, int ix) {
#if NID_USE_FRACT
  (void)ix;
  return __builtin_amdgcn_fract(x);
#else
  return x - (double)ix;
#endif
}

// ---- FAST math helpers (not rounding-identical to the reference; see k_eval) ----------
// bilinear sample as two lerps on register taps
__device__ __forceinline__ double bilinear_rows_fast(const RowPair &r, int wx, double x) {
  const int ix = (int)x;
  const double dx = x - ix;
  int kx = ix - wx;
  kx = min(max(kx, 0), 3);
  const double i00 = win_tap(r.ra, kx), i01 = win_tap(r.ra, kx + 1);
  const double i10 = win_tap(r.rb, kx), i11 = win_tap(r.rb, kx + 1);
  const double top = fma(dx, i01 - i00, i00);
  const double bot = fma(dx, i11 - i10, i10);
  return fma(r.dy, bot - top, top);
}


// TWICE the central-difference image gradient of the bilinear surface at (u, v) (the caller folds
// the 1/2 into its constants):
//   gx = bil(u+1,v) - bil(u-1,v), gy = bil(u,v+1) - bil(u,v-1)
// (types_six_dof_expmap.cpp:434-435).  Interior pixels (window origin = (ix-1, iy-1)) use the 12
// shared taps with exact integer differences; the first row/column falls back to the generic form,
// which also reproduces the (int)-truncation extrapolation of the reference there.
// Centre sample only (cost phase): interior pixels read the four fixed taps.
__device__ __forceinline__ double sample_fast_interior(const Win &w, double u, double v) {
  const int ix = (int)u, iy = (int)v;
  const double dx = u - ix, dy = v - iy;
  const int a11 = tap_i(w.r1, 1), a12 = tap_i(w.r1, 2), a21 = tap_i(w.r2, 1), a22 = tap_i(w.r2, 2);
  const double m1 = fma(dx, (double)(a12 - a11), (double)a11);
  const double m2 = fma(dx, (double)(a22 - a21), (double)a21);
  return fma(dy, m2 - m1, m1);
}
// Gradient by the 12 shared taps of the window at origin (ix-1, iy-1), exact integer differences; `ic` receives
// the centre sample bil(u,v) (same arithmetic as sample_fast_interior).  Valid for every in-frame sample: the
// first row / column of the image interpolates towards the extrapolated margin (see Win).
__device__ __forceinline__ void gradient_fast_interior(const Win &w, double u, double v, double &gx, double &gy, double &ic) {
  const int ix = (int)u, iy = (int)v;
  const double dx = u - ix, dy = v - iy;
  // gx: rows iy (r1), iy+1 (r2); L(ix+1) - L(ix-1) = (a2 - a0) + dx*((a3 - a2) - (a1 - a0))
  const int a10 = tap_i(w.r1, 0), a11 = tap_i(w.r1, 1), a12 = tap_i(w.r1, 2), a13 = tap_i(w.r1, 3);
  const int a20 = tap_i(w.r2, 0), a21 = tap_i(w.r2, 1), a22 = tap_i(w.r2, 2), a23 = tap_i(w.r2, 3);
  const double g1 = fma(dx, (double)((a13 - a12) - (a11 - a10)), (double)(a12 - a10));
  const double g2 = fma(dx, (double)((a23 - a22) - (a21 - a20)), (double)(a22 - a20));
  gx = fma(dy, g2 - g1, g1);
  // gy: M(j) = a(j,1) + dx*(a(j,2) - a(j,1)) on rows iy-1 .. iy+2;
  //     bil(v+1) - bil(v-1) = (M2 - M0) + dy*((M3 - M2) - (M1 - M0))
  const int a01 = tap_i(w.r0, 1), a02 = tap_i(w.r0, 2), a31 = tap_i(w.r3, 1), a32 = tap_i(w.r3, 2);
  const double m0 = fma(dx, (double)(a02 - a01), (double)a01);
  const double m1 = fma(dx, (double)(a12 - a11), (double)a11);
  const double m2 = fma(dx, (double)(a22 - a21), (double)a21);
  const double m3 = fma(dx, (double)(a32 - a31), (double)a31);
  gy = fma(dy, (m3 - m2) - (m1 - m0), m2 - m0);
  ic = fma(dy, m2 - m1, m1);
}
// B-spline values/derivatives from the per-span polynomial table (kCoefRow doubles per span):
// row layout [k][a0 a1 a2 a3 d0 d1 d2], k = 0..3.  t = u - jc in [0,1).  The reference's u == 0
// quirk (derivative identically 0 at exactly 0, Q5) is kept by a select.
// Experiment switches of the LDS read scheduling (tools/build_variant.py; profiles/r02_ablations_A.txt): left to itself
// the scheduler sends the table reads of a sample through one register quad, one dependent LDS round trip after the
// other.  NID_BS_BATCH (cost phase, 16 value coefficients; 2 or 4 rows per wait), NID_BSD_BATCH (derivative
// coefficients at once; the FAST Jacobian phase reads the contracted tables instead since round 5).
#ifndef NID_BS_BATCH
#define NID_BS_BATCH 0
#endif
#ifndef NID_BSD_BATCH
#define NID_BSD_BATCH 0
#endif
// FRAC: the caller's jc is (int)u unclamped (the main passes): t by frac_nonneg
template <bool WANT_DER, bool BATCH = false, bool FRAC = false>
__device__ __forceinline__ void bspline4_poly(double u, int jc, const double *ctab, double B[4], double D[4]) {
  const double t = FRAC ? frac_nonneg(u, jc) : u - (double)jc;
  const double *c = ctab + __mul24(jc, kCoefRow);
#if NID_BS_BATCH
  if (!WANT_DER && BATCH) {
    // the value coefficients requested together and waited for once per NID_BS_BATCH rows; the empty asm statements
    // make the loaded values "used" at that point, so the reads cannot sink below them
    double cf[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int p = 0; p < 4; p++) cf[k][p] = c[7 * k + p];
    if (NID_BS_BATCH == 4) {
      asm volatile("" : "+v"(cf[0][0]), "+v"(cf[0][1]), "+v"(cf[0][2]), "+v"(cf[0][3]), "+v"(cf[1][0]), "+v"(cf[1][1]), "+v"(cf[1][2]), "+v"(cf[1][3]),
                        "+v"(cf[2][0]), "+v"(cf[2][1]), "+v"(cf[2][2]), "+v"(cf[2][3]), "+v"(cf[3][0]), "+v"(cf[3][1]), "+v"(cf[3][2]), "+v"(cf[3][3]));
    } else {
#pragma unroll
      for (int k = 0; k < 4; k += 2)
        asm volatile("" : "+v"(cf[k][0]), "+v"(cf[k][1]), "+v"(cf[k][2]), "+v"(cf[k][3]), "+v"(cf[k + 1][0]), "+v"(cf[k + 1][1]), "+v"(cf[k + 1][2]), "+v"(cf[k + 1][3]));
    }
#pragma unroll
    for (int k = 0; k < 4; k++) B[k] = fma(fma(fma(cf[k][3], t, cf[k][2]), t, cf[k][1]), t, cf[k][0]);
    return;
  }
#endif
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double *ck = c + 7 * k;
    B[k] = fma(fma(fma(ck[3], t, ck[2]), t, ck[1]), t, ck[0]);
    if (WANT_DER) {
      const double d = fma(fma(ck[6], t, ck[5]), t, ck[4]);
      D[k] = (u == 0.0) ? 0.0 : d;
    }
  }
}

// The four VALUES with an error relative to each weight at BOTH ends of the span.  Horner in t = u - jc is exact to
// ~1e-16 ABSOLUTE: fine for a weight that vanishes at the span's left end (its constant coefficient is an exact 0), not
// for one that vanishes at the right end -- 3 (1 - t) at 1 - t = 4e-16 (a target sample an ulp below 255 on the last span)
// came out 15 % off, and a bin made of such weights alone enters the Jacobian through log2 of its mass (2.3e-3 of a
// one-sample cell's Jacobian: sweep seed 344412).  The clamped knot vector is symmetric: the basis on span jc at t is the
// basis on span S - 1 - jc at s = 1 - t in reverse order, and s = (jc + 1) - u is exact next to the knot -- so the right
// half of a span is evaluated from its mirror image's row.  Used where small weights are looked at one by one: the rare
// branch of hist_add and the constant weights of the clamped / near-saturated samples.
__device__ __forceinline__ void bspline4_vals_both_ends(double u, int jc, int S, const double *ctab, double B[4]) {
  const double t = u - (double)jc;
  const bool right = t > 0.5;
  const double x = right ? (double)(jc + 1) - u : t;
  const double *c = ctab + __mul24(right ? S - 1 - jc : jc, kCoefRow);
  double v[4];
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double *ck = c + 7 * k;
    v[k] = fma(fma(fma(ck[3], x, ck[2]), x, ck[1]), x, ck[0]);
  }
#pragma unroll
  for (int k = 0; k < 4; k++) B[k] = right ? v[3 - k] : v[k];
}

// derivative-only form of bspline4_poly (Jacobian phase of k_eval2).  The u == 0 quirk (Q5: all
// four derivatives identically 0) is applied by the caller, once, on the pixel's coefficient.
__device__ __forceinline__ void bspline4_poly_der(double u, int jc, const double *ctab, double D[4]) {
  const double t = u - (double)jc;
  const double *c = ctab + __mul24(jc, kCoefRow);
#if NID_BSD_BATCH
  {
    double cf[4][3];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
      for (int p = 0; p < 3; p++) cf[k][p] = c[7 * k + 4 + p];
    asm volatile("" : "+v"(cf[0][0]), "+v"(cf[0][1]), "+v"(cf[0][2]), "+v"(cf[1][0]), "+v"(cf[1][1]), "+v"(cf[1][2]),
                      "+v"(cf[2][0]), "+v"(cf[2][1]), "+v"(cf[2][2]), "+v"(cf[3][0]), "+v"(cf[3][1]), "+v"(cf[3][2]));
#pragma unroll
    for (int k = 0; k < 4; k++) D[k] = fma(fma(cf[k][2], t, cf[k][1]), t, cf[k][0]);
    return;
  }
#endif
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const double *ck = c + 7 * k;
    D[k] = fma(fma(ck[6], t, ck[5]), t, ck[4]);
  }
}

// 1/z to ~1 ulp from the hardware estimate and two Newton steps (FAST math only)
__device__ __forceinline__ double rcp_fast(double z) {
  double r = __builtin_amdgcn_rcp(z);
  double e = fma(-z, r, 1.0);
  r = fma(r, e, r);
  e = fma(-z, r, 1.0);
  return fma(r, e, r);
}

// ---- IEEE quotients that share their reciprocal (round 6) ----------------------------------------------------------------
// What the compiler emits for a / z in f64 (AMDGPU LowerFDIV64): v_div_scale of both operands, v_rcp_f64 of the denominator,
// two Newton steps on the reciprocal, q = a * r, ONE residual step fma(fma(-z, q, a), r, q) (v_div_fmas), v_div_fixup.  The
// scale and fix-up instructions act only on operands near the ends of the exponent range (a denormal or huge denominator, an
// exponent difference of 768 and more, a numerator below 2^-969, NaN / infinity / zero) and pass everything else through:
// the quotient IS the sequence below -- rcp_fast is its first half --, and the reciprocal can be shared by every quotient over
// the same z (u and v of a projection; 1/z, x/z, y/z of linearizeOplus): the same instructions on the same values, hence
// the IEEE quotient's bits (tools/ubench/div_shared.hip compares 2^28 quotients with the compiler's; STRICT's per-pixel
// u, v stay bit-exact against the oracle, tests/test_parity_gpu.py).  Operands outside [2^-100, 2^100] -- and a zero
// numerator, whose sign the short form can lose -- take the division itself (a divergent, practically never taken branch).
__device__ __forceinline__ bool exp_mid(double x) { return (((unsigned)__double2hiint(x) >> 20) & 0x7FFu) - 923u <= 200u; }
__device__ __forceinline__ double div_shared(double a, double z, double r, bool z_mid) {
  const double q = a * r;
  double res = fma(fma(-z, q, a), r, q);
  if (__builtin_expect(!(z_mid && exp_mid(a)), 0)) res = a / z;
  return res;
}
// RN(x / 255.0): the quotient by a CONSTANT from its correctly rounded reciprocal and one residual step (Markstein; exact for
// every x when the divisor's significand is not all ones -- tests/test_bspline_host.py compares 10^7 values with the division)
__device__ __forceinline__ double div_255(double x) { return div_small_r(x, 255.0, 1.0 / 255.0); }

// The reference's four-term bilinear sample (types_six_dof_expmap.h:310-328: (int) truncation, left-to-right sum) from the
// 2x2 cell of (u, v) alone: two unaligned dwords at the element offset of pixel ((int)u, (int)v) instead of the 4x4 window
// and its run-time tap selection.  For samples inside the frame (u, v >= 0: the sample's own pixel is never in the margin).
__device__ __forceinline__ double bilinear_cell_exact(const EvalParams &P, double u, double v) {
  const int ix = (int)u, iy = (int)v;
  const unsigned st = (unsigned)P.im1_stride;
  const unsigned po = __umul24((unsigned)iy, st) + (unsigned)ix + (st + 1u);  // (the buffer starts at pixel (-1, -1))
  const unsigned c1 = load_u32_at(P.im1s, po), c2 = load_u32_at(P.im1s, po + st);
  const double dx = u - ix, dy = v - iy;
  const double dxdy = dx * dy;
  const double i00 = (double)__builtin_amdgcn_sbfe((int)c1, 0u, 16u), i01 = (double)__builtin_amdgcn_sbfe((int)c1, 16u, 16u);
  const double i10 = (double)__builtin_amdgcn_sbfe((int)c2, 0u, 16u), i11 = (double)__builtin_amdgcn_sbfe((int)c2, 16u, 16u);
  return dxdy * i11 + (dy - dxdy) * i10 + (dx - dxdy) * i01 + (1 - dx - dy + dxdy) * i00;
}

// bilinear_u8 on the window: same arithmetic, taps from registers.  A tap index
// that leaves the window can only belong to a tap whose weight is exactly 0
// (x or y rounded up to an integer) -- any finite byte is then correct.
__device__ __forceinline__ double bilinear_w(const Win &w, double x, double y) {
  const int ix = (int)x;
  const int iy = (int)y;
  const double dx = x - ix;
  const double dy = y - iy;
  const double dxdy = dx * dy;
  int kx = ix - w.wx, ky = iy - w.wy;
  kx = min(max(kx, 0), 3);
  ky = min(max(ky, 0), 3);
  const WRow ra = win_row(w, ky), rb = win_row(w, min(ky + 1, 3));
  const double i00 = win_tap(ra, kx), i01 = win_tap(ra, kx + 1);
  const double i10 = win_tap(rb, kx), i11 = win_tap(rb, kx + 1);
  return dxdy * i11 + (dy - dxdy) * i10 + (dx - dxdy) * i01 + (1 - dx - dy + dxdy) * i00;
}

// Wave64 sum by DPP (VALU cross-lane moves, no LDS traffic): quad swaps, row
// (half-)mirror, then row_bcast15 / row_bcast31.  The total ends in lane 63.
// Lanes the move does not write (rows outside ROW_MASK) are left UNDEFINED (no `old` operand, so no v_mov 0 per
// half in front of every v_mov_dpp: 12 VALU instructions less per summed value): wave_sum_to_lane63 only promises
// lane 63, and no value that reaches lane 63 ever passes through such a lane (see there).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_mov_f64(double x) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, ROW_MASK, 0xf, false);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, ROW_MASK, 0xf, false);
  return __hiloint2double(hi, lo);
}
// (Steps 1-4 write every lane.  row_bcast15 writes rows 1, 3 from lane 15 of rows 0, 2; row_bcast31 writes rows 2, 3
// from lane 31: rows 0 and 2 hold garbage after step 5, lane 31 (row 1) and lane 63 (row 3) do not, and step 6
// reads lane 31 only.)
__device__ __forceinline__ double wave_sum_to_lane63(double v) {
  v += dpp_mov_f64<0xB1, 0xf>(v);   // quad_perm [1,0,3,2]
  v += dpp_mov_f64<0x4E, 0xf>(v);   // quad_perm [2,3,0,1]
  v += dpp_mov_f64<0x141, 0xf>(v);  // row_half_mirror
  v += dpp_mov_f64<0x140, 0xf>(v);  // row_mirror
  v += dpp_mov_f64<0x142, 0xa>(v);  // row_bcast15 -> rows 1,3
  v += dpp_mov_f64<0x143, 0xc>(v);  // row_bcast31 -> rows 2,3
  return v;
}

// A value every lane of the wave holds identically, moved to scalar registers (frees its VGPR pair)
__device__ __forceinline__ double wave_uniform(double x) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(x));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(x));
  return __hiloint2double(hi, lo);
}

// the value lane 63 holds, in scalar registers
__device__ __forceinline__ double wave_lane63(double x) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), 63);
  return __hiloint2double(hi, lo);
}

// OR of a 32-bit value over the wave's 64 lanes, in a scalar register (same DPP ladder as wave_sum_to_lane63)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned dpp_or_u32(unsigned x) {
  return x | (unsigned)__builtin_amdgcn_update_dpp(0, (int)x, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ unsigned wave_or_u32(unsigned v) {
  v = dpp_or_u32<0xB1, 0xf>(v);
  v = dpp_or_u32<0x4E, 0xf>(v);
  v = dpp_or_u32<0x141, 0xf>(v);
  v = dpp_or_u32<0x140, 0xf>(v);
  v = dpp_or_u32<0x142, 0xa>(v);
  v = dpp_or_u32<0x143, 0xc>(v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}

// the value lane `l` (a constant) holds, in scalar registers
template <int L>
__device__ __forceinline__ double wave_lane(double x) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), L);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), L);
  return __hiloint2double(hi, lo);
}

// Deterministic block sum of NV values; result broadcast to every thread.
// `red` holds NV * (NT/64) doubles and must not be in use.
template <int NT, int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double *red, int tid) {
  constexpr int NW = NT / 64;
#pragma unroll
  for (int k = 0; k < NV; k++) v[k] = wave_sum_to_lane63(v[k]);
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 63) {
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

// 64-bit fixed point by the 2^52 magic number: for 0 <= w*scale < 2^52 the low
// 52 bits of RN(w*scale + 2^52) ARE round-to-nearest-even(w*scale).
__device__ __forceinline__ unsigned long long fx_encode(double w, double scale) {
  const double t = fma(w, scale, 0x1p52);
  return (unsigned long long)__double_as_longlong(t) - 0x4330000000000000ull;
}

// Hot-kernel form: the bit pattern of a SUBNORMAL product.  A non-negative double below 2^-1021 has the bit pattern
// value / 2^-1074 (biased exponents 0 and 1 share one ulp), so with the operands pre-scaled so that the product is
// w * 2^(s - 1074) its bits ARE round-to-nearest-even(w * 2^s) as a 64-bit integer: one v_mul_f64 per atomic, no
// magic number in the top 12 bits -- hence no per-copy carry limit (the copies of a bin add up as plain 64-bit
// integers, the scale only has to keep a whole cell's bin below 2^63: s = 52 for cells of up to 2047 slots) and seven
// more fractional bits than RN(w * 2^45 + 2^52) had (round 2: one random case in ~3 000 missed the 1e-9 Jacobian
// bound by the quantum of mid-range products in the joint bins).  f64 subnormals are never flushed and cost nothing
// on gfx950 (tools/ubench/denorm_encode.hip checks the bits against the host's rounding and the issue rate).
// Both factors must be NORMAL numbers (a subnormal factor would have lost bits already): the scale is split --
// kWcPre = 2^-512 rides on the target weights (on the value polynomials of the FAST B-spline table: a power of two
// commutes with every rounding of the Horner steps), EvalParams::hist_dn = 2^(s - 562) on the reference weights and
// on the marginal adds.  Operands must be >= +0 (a negative product would carry the sign bit into the sum): target
// weights that could be negative by rounding are the tiny outer ones, which take the fine-level path below.
__device__ __forceinline__ unsigned long long fx_bits(double subnormal_product) {
  return (unsigned long long)__double_as_longlong(subnormal_product);
}
constexpr double kWcPre = 0x1p-512, kWcPreInv = 0x1p512;

// Fine histogram levels for SMALL target weights.  The copies above resolve 2^-52 per addend (fx_bits; 2^-45 in round 2):
// exact enough for every bin's entropy term, but not for the Jacobian's weight W = -(1 + log2 p) of a bin whose
// whole mass is made of small addends -- the reference keeps such masses in f64, takes W ~ 50..100 of
// p ~ 1e-15..1e-30 and multiplies it with that bin's derivative sums.  A sample within 2^-9 of a knot feeds its
// outer bins with t^3/6-like weights; on the two END spans of the clamped knot vector the basis functions next
// to the end are LINEAR / quadratic in t (weight 3t, derivative 3), so a weight of 1e-16 (a black or clamped
// saturated sample, or any sample at an integer position of an identity-like pose) carries an O(1) derivative:
// the bin mass must then be right to ~1e-9 RELATIVE.  Integer accumulation (order-independent, bitwise
// reproducible) with that dynamic range = several fixed-point levels: a weight w < 2^-8 of such a sample goes to
// level L = min(floor((-8 - e) / 24), 4), e = its binary exponent, scaled by 2^(59 + 24 L) -- and every joint addend
// wr[m] * w below 2^-8 of a sample whose target OR reference sample sits next to a knot to the level of THAT
// product's exponent: at least 27 bits of every addend survive, at most 2^51 per addend and 2^62 per bin.  Weights >= 2^-8 stay in the coarse copies
// (error of the Jacobian term there <= quantum * |dw| / w = 2.2e-16 * 3 * 256 = 2e-13).  kTinyW: the smaller
// outer weight of a sample (target or reference) -- always the CUBIC one, c * d^3 in the distance d to the knot, on end
// spans too -- is below it iff the sample sits within ~5.6e-4 of a knot.  What a sample OUTSIDE that distance loses in
// the coarse copies is bounded by quantum * |dw| / w ~ quantum * 3 / d for its cubic weight, quantum * 2 / d for a
// quadratic and quantum / d for a linear end-span weight: 2^-52 / 5.6e-4 = 4e-13 per addend.  Round 2 had 2^-45 / 2.8e-3 =
// 1e-11 (threshold 2^-28: chosen for THAT quantum); with the 2^-52 encode the threshold moved to 2^-35 -- a fifth of the
// samples take the rare branch of hist_add (one wave-round in eight instead of one in two on a smooth image): -1.1 %
// kernel time on the plain pair, -1.9 % on the flash pair; 2^-40 would give another 0.2 / 0.7 %.  The per-case Jacobian
// errors of a 3 000-case sweep are the same to three digits at 2^-28, 2^-35 and 2^-40 (profiles/r03_ablations_A.txt).
#ifndef NID_TINY_W_EXP
#define NID_TINY_W_EXP 35
#endif
constexpr double kTinyW = 1.0 / (double)(1ull << NID_TINY_W_EXP);
constexpr double kFineW = 0x1p-8;
// An addend below this is dropped: a bin is looked at only if its mass reaches kSigma * N_c >= 3e-28 (types_six_dof_expmap.h:281,
// the 300-pixel activity threshold), and 19 200 such addends -- the largest cell -- are 1e-9 of the smallest mass that
// counts.  What falls below it: the outermost weight of a sample an ulp away from a knot (a saturated target sample the
// reference's bilinear sum leaves at 255 - 3e-14: weight 1e-47) and its products -- one in seventeen of the saturated samples
// of a flash pair's hot spot, each of which used to cost five single-copy fine-level atomics on the same few addresses.  (Also: a negative
// weight by rounding, which the integer encodes cannot take.)
constexpr double kNegligibleW = 0x1p-136;
constexpr int kFineLevels = 5;
// REPAIR of the end spans' LINEAR-weight columns (round 4; sweep seed 407031).  On the two end spans of the clamped knot
// vector one basis function is linear in the distance to the end knot (weight 3 t, derivative 3: target column 1 on the
// first span, column nb - 2 on the last), so a sample an ulp away from the knot carries an O(1) derivative on a weight of
// 1e-15: the Jacobian then needs W = -(1 + log2 p) of every bin that weight lands in, i.e. the bin's WHOLE mass, to
// ~1e-9 relative -- also the part of it that ordinary samples added through the coarse copies, whose 2^-52 quantum is an
// ABSOLUTE resolution (one product of 9e-13 = 1.3e-8 * 7e-5, neither factor small enough for the fine levels, is 1.2e-4
// off in relative terms: 3e-8 of that cell's Jacobian).  No per-sample work in the main pass pays for that:
//  * the rare branches of the histogram update FLAG the bins (joint row / marginal, column 1 or nb - 2) that receive a
//    linear end-span weight below kLinFlagW (one LDS atomic OR inside code that is rare already; the near-saturated and
//    clamped groups flag theirs in the fold);
//  * the fold looks at the flagged bins only: one that holds coarse addends and less than kRepairMass in all is to be
//    REPAIRED -- the cost phase's pixel loops run once more (hist_add<REPAIR>), every addend that went to the coarse
//    copies of such a bin now goes to the fine level of its own exponent (27+ bits of every addend, still integer adds),
//    and the bin is folded again from its fine levels alone.
// What an unflagged or heavier bin loses is bounded by n * 2^-53 * 3 / kLinFlagW resp. n * 2^-53 / kRepairMass per unit of
// the flagged samples' reference weight: 1e-11 * n.  Cells that need a repair: one in ~45 000 random cases, and cells on
// the rim of a saturated patch (counted: EvalParams::repair_count, nid_debug_repair_count).
constexpr double kLinFlagW = 0x1p-16;
constexpr double kRepairMass = 0x1p-12;
constexpr double kRepairRel = 0x1p-12;
__device__ __forceinline__ int fine_level(double w) {
  const int e = __builtin_amdgcn_frexp_exp(w);  // w = m * 2^e, m in [0.5, 1): e <= -8 for w < 2^-8
  const int x = min(max(-8 - e, 0), 119);
  return (x * 43) >> 10;                        // floor(x / 24) for 0 <= x < 120
}
__device__ __forceinline__ double fine_scale(int level) {  // 2^(59 + 24 level)
  return __hiloint2double((1023 + 59 + 24 * level) << 20, 0);
}
__device__ __forceinline__ double fine_inv_scale(int level) {
  return __hiloint2double((1023 - 59 - 24 * level) << 20, 0);
}

// ---------------------------------------------------------------------------
// The hot kernel: cost (+ Jacobian) of one cell per workgroup, the cell's
// Huber-weighted quadratic form, and -- in the workgroup that finishes last --
// the sum over all cells (the 6x6 normal equations), written straight to the
// caller's result buffer (device memory or pinned host memory).
//
// Inter-workgroup hand-off (cdna_hip_programming.md Guideline 16, table row 1):
// the per-cell 32-double block is stored write-through (sc1, relaxed agent-scope
// atomics) by wave 0 only, wave 0 drains vmcnt, one lane takes a ticket with an
// agent-scope atomic add; the workgroup whose add returns nblocks-1 reads every
// block back with sc1 loads (no fence needed on that row) and reduces in a fixed
// order.  No dispatch-order or placement assumption; the ticket is reset by the
// last arriver and zeroed once at context creation.
__device__ __forceinline__ void store_sc1(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double load_sc1(const double *p) {
  return __longlong_as_double((long long)__hip_atomic_load(
      reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// Per-cell results of a DIRECT launch in host memory: plain system-scope stores (fine-grained host memory is uncached on the device: they
// go out over PCIe as they are issued; the host polls every WORD against a sentinel, so nothing has to be ordered)
__device__ __forceinline__ void store_sys(double *p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v),
                     __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// The bit pattern the host fills a DIRECT launch's result words with before the launch (a signalling-NaN payload no
// arithmetic produces: NaN results are the canonical quiet NaN); a word that still holds it has not arrived.
constexpr unsigned long long kHostSentinel = 0x7FF4DEADBEEF5A5Aull;

// 128- / 256-thread shapes: the six sums go through a [6][kXposeStride] array of doubles that reuses the histogram area
// (rows padded by 8 doubles so that the six row groups of a read do not share banks); 0 = the DPP form everywhere
#ifndef NID_XPOSE_SUM
#define NID_XPOSE_SUM 1
#endif
// the six Jacobian sums of every wave (DPP form: the 512- / 1024-thread shapes; the others need no such scratch)
constexpr int kRedDoubles(int nt) { return (nt <= 256 && NID_XPOSE_SUM) ? 0 : ((6 * (nt / 64) + 1) & ~1); }
// Clamped target samples (ic >= 255 -> 254.999, types_six_dof_expmap.cpp:572-573) all have the SAME four target weights:
// instead of 20 histogram adds each (15 of them fine-level adds), FAST math sums their reference weights per reference bin
// (nb bins) and counts them (one more bin), NC copies like the histograms; the fold adds c_k * sum to the bins they feed.
// The coarse sums have kClampCopies copies (same scale as the histograms); reference weights below 2^-8 go to fine levels of their own exponent like everywhere else (a joint bin
// fed only by clamped samples is c_k * this sum: it needs the sum to ~1e-9 relative even when it is one weight of 1e-8).
constexpr int kClampBins(int nb) { return (nb + 2) & ~1; }
constexpr int kFlagWords = 7;  // clamp_flag[2] | lin_flag[2] | repair_set[2] | grp_flag (eval_cell), 16-byte aligned start
constexpr int kFlagDoubles = 4;  // ... in a 32-byte slot between the tables and the histogram area
constexpr int kFlagPadWords = 6;  // (where the first six used to sit, behind the clamped samples' sums: kept as padding)
// (more copies would thin out the same-address conflicts of a saturated patch, but the workgroup's LDS request sits at the
// 15 360 bytes ten workgroups per CU allow: 8 or 16 copies cost the plain pair 4 %, profiles/r03_ablations_A.txt)
#ifndef NID_CLAMP_COPIES
#define NID_CLAMP_COPIES 4
#endif
#ifndef NID_NEAR_SAT_BINS
#define NID_NEAR_SAT_BINS 1
#endif
constexpr int kClampCopies = NID_CLAMP_COPIES;
// NEAR-SATURATED target samples: four equal taps of 255 leave the reference's four-term bilinear sum at 255 (clamped,
// above) or an ulp or two below it (one sample in seventeen of a saturated patch at the bench's poses).  Every sample at EXACTLY 255 - ulp has the same four
// target weights, bit for bit -- FAST math sums the reference weights of those in a second set of bins and folds them in
// with that sample's weights, like the clamped ones.  Only the exact value qualifies: the last span is an END span of the
// clamped knot vector, its small weights are linear / quadratic in 1 - t (1e-15, 1e-30) with O(1) derivatives, and a joint
// bin made of them alone enters the Jacobian through log2 of its mass -- a sample two ulps below 255 has TWICE that
// weight, so it takes the general path (a window of 2^-40 below 255 instead of the one value put 0.17 on a Jacobian of
// 6.7 in one cell of sweep seed 70874).  The Jacobian phase does not add to histograms and is unchanged.
constexpr double kNearSatIc = 255.0 - 0x1p-45;  // the largest double below 255
constexpr int kNearSatBinBytes(int nb) { return kClampBins(nb) * (kClampCopies + kFineLevels) * 8; }
// the two weight tables (tab, term: (nbins + 1) & ~1 doubles each) lend the bins their area when it is large enough
__host__ __device__ constexpr bool near_sat_aliased(int nb) { return 16 * ((nb * nb + nb + 1) & ~1) >= kNearSatBinBytes(nb); }
constexpr int kXposeStride(int nt) { return nt + 8; }
constexpr int kXposeDoubles(int nt) { return (nt <= 256 && NID_XPOSE_SUM) ? 6 * kXposeStride(nt) : 0; }
constexpr int kQuad = 32;  // per-cell block: rho0 | b[6] | H upper[21] | 1.0 | 0 0 0

// which (a,b) of the upper triangle a quad slot 7..27 holds
__device__ __forceinline__ void quad_ab(int slot, int &a, int &b) {
  int k = slot - 7, aa = 0, len = 6;
  while (k >= len) { k -= len; aa++; len--; }
  a = aa; b = aa + k;
}

// Two-level in-launch reduction of the per-cell blocks (cells -> groups of P.group_size cells -> total),
// each level a ticket + last-arriver sum, so the serial read of any one wave is
// <= max(group_size, ngroups) * 256 B.
// Run by ONE wave (wave 0 of the workgroup; the other waves have retired by then).
// 64 lanes: v = lane & 31 selects the element, half = lane >> 5 the even / odd blocks; eight sc1 loads in
// flight per lane, the two halves are combined with one cross-lane read.  Result in lanes 0..31.
__device__ __forceinline__ double sum_blocks_w0(const double *src, int count, int lane) {
  const int v = lane & 31, half = lane >> 5;
  double x[8];
  double s = 0.0;
  for (int c0 = half; c0 < count; c0 += 16) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int c = c0 + 2 * k;
      x[k] = (c < count) ? load_sc1(src + (size_t)c * kQuad + v) : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) s += x[k];
  }
  return s + __shfl_down(s, 32, 64);
}

// Called by wave 0 only, all 64 lanes active, after it has stored (sc1) this cell's 32-double block.
__device__ __forceinline__ void finish_and_reduce_w0(const EvalParams &P, const SlotArgs &SA, int cl, int lane) {
  const int gs = P.group_size;
  const int gq = cl / gs;
  const int ngroups = (P.g.nloc + gs - 1) / gs;
  const int gcount = min(gs, P.g.nloc - gq * gs);
  // level 0: drain the block's stores, take a group ticket.  Per-cell outputs that go straight to host memory
  // (cellout_host) are made visible at system scope first: the host reads them as soon as the LAST workgroup's
  // sequence word arrives, and that workgroup can only vouch for its own stores.
  if (SA.cellout_host) __threadfence_system();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned last = 0u;
  if (lane == 0) {
    const unsigned old = __hip_atomic_fetch_add(SA.ticket + 1 + gq, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (old == (unsigned)(gcount - 1)) ? 1u : 0u;
  }
  if (__builtin_amdgcn_readfirstlane(last) == 0u) return;
  double r = sum_blocks_w0(SA.quad + (size_t)gq * gs * kQuad, gcount, lane);
  if (SA.host_quad == 3) {
    // GROUP-DIRECT: the group's sum straight to its 256-byte block in pinned host memory (out_reduced: [ngroups][32]);
    // the host adds the groups up (wait_groups in nid_capi.hip).  No second ticket, no fence, no sequence word: every
    // word is polled against a sentinel.
    if (lane < 32) store_sys(SA.out_reduced + (size_t)gq * kQuad + lane, r);
    if (lane == 0) __hip_atomic_store(SA.ticket + 1 + gq, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // level 1: publish the group sum, take the top ticket
  if (lane < 32) store_sc1(SA.gpart + (size_t)gq * kQuad + lane, r);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  last = 0u;
  if (lane == 0) {
    __hip_atomic_store(SA.ticket + 1 + gq, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned old = __hip_atomic_fetch_add(SA.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (old == (unsigned)(ngroups - 1)) ? 1u : 0u;
  }
  if (__builtin_amdgcn_readfirstlane(last) == 0u) return;
  r = sum_blocks_w0(SA.gpart, ngroups, lane);
  if (lane < 32) SA.out_reduced[lane] = r;
  if (lane == 0) __hip_atomic_store(SA.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (SA.host_seq) {  // results live in pinned host memory: publish at system scope
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) {
      __threadfence_system();
      __hip_atomic_store(SA.host_seq, SA.launch_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

// Math modes of the evaluation kernel:
// STRICT = true : every rounding of the reference path is reproduced (bit-exact per-pixel
//                 intermediates; IEEE divisions, quaternion rotate, (int)-truncating bilinear form,
//                 recursion-identical B-splines, fx*x/z vs fx*(x/z)).
// STRICT = false: FAST math -- same algorithm, cheaper but not rounding-identical arithmetic
//                 (3x4 matrix with FMAs, one reciprocal per pixel, lerp bilinear on shared taps,
//                 per-span polynomial B-splines); per-pixel values differ by a few ulp, per-cell
//                 results stay inside the stated 1e-11 / 1e-9 tolerances (tests run both modes).
// NB > 0: bin count known at compile time (LDS offsets of the 20 histogram atomics and the 20
// weight-table reads fold into instruction immediates); NB == 0: read it from the geometry.
// DBG: per-pixel dump / phase stamps compiled in (diagnostic launches only).

// ---------------------------------------------------------------------------
// k_eval2: the hot kernel, organised for OCCUPANCY instead of per-thread ILP (an earlier
// variant that kept 5 pixels per thread in registers across both phases needed ~220 VGPRs,
// fit 2 workgroups per CU and measured 10 % slower).
// Nothing is carried in registers from the cost phase to the Jacobian phase: the Jacobian
// phase re-reads the pixel's tile entry (L2-resident) and its 4x4 window and recomputes
// the warp / sample / B-spline derivative with the identical instruction sequence (so the
// reference's Q7 "Jacobian reuses the cost pass's intensities" holds by construction).  The
// pixel loops are NOT unrolled: ~100 VGPRs instead of ~220, so 4-5 workgroups share a CU
// and the scheduler hides the load / LDS latencies that one wave per SIMD exposes.
// guard bands of the FAST-mode decision re-check (see exact_decisions)
// clamp guard: ic within 1/8 of 0 or of 255 (the clamp replaces ic >= 255 by 254.999), as ONE compare: an unsigned range
// check on the sample's high dword -- [kGuardLoHi, kGuardLoHi + kGuardSpanHi] lies inside [0.125, 254.875]; 32-bit
// literals in the instruction instead of two f64 constants in scalar registers.
// Why 1/8 and not the 1e-4 the clamp itself needs (rounds 2-4): the two END spans of the clamped knot vector carry a
// weight that is LINEAR in the distance to the end knot, w = 3 d with d = ic * S / 255 resp. (255 - ic) * S / 255, and a
// joint bin made of such a weight enters the Jacobian through log2 of its mass -- FAST math's u is an ulp or two from the
// reference's (x * (1/z) against x / z), on a steep edge (200 grey levels per pixel) that is 4e-12 in ic, i.e.
// 4e-12 / (255 - ic) RELATIVE in that weight: 7e-10 for a sample at 254.9947 (sweep seed 511576: 1.5e-9 of its cell's
// Jacobian, over the 1e-9 bound in FAST math only).  Samples that close to an end knot are decided by the second
// passes, with the reference's own (u, v, ic) (exact_decisions); at 1/8 from the knot the same ulp is 3e-11 of the weight.
// On a natural image those are the samples of black or saturated patches and their rims, which the 1e-4 band sent there
// already.
constexpr unsigned kGuardLoHi = 0x3FC00001u;                  // high dword of 0.125, plus one
constexpr unsigned kGuardSpanHi = 0x406FDBFFu - kGuardLoHi;    // ... up to the high dword of 254.875, minus one
__device__ __forceinline__ bool outside_clamp_guard(double ic) {
  return (unsigned)__double2hiint(ic) - kGuardLoHi > kGuardSpanHi;
}
constexpr unsigned kBorderEpsHi = 0x3EB00000u;                  // high dword of kBorderEps = 2^-20
// A sample the FAST main passes take lies inside the clamp guard: 0.125 < ic < 254.875, so the clamp ic >= 255 and
// the u == 0 quirk of the B-spline derivative (pc == 0) cannot apply to it; only the second passes (exact_decisions)
// carry those selects.
constexpr bool kMainPassClamps = false;
constexpr double kBorderEps = 0x1p-20;  // FAST u, v are within ~1e-12 of the reference's (|u| < 2^11): a wide margin
struct PixelFront {
  unsigned e0;  // FAST: element offset of the sample's pixel in the target image, iy * stride + ix (0 for a lane without a sample): see load_win_centre_e
  bool in, jin;
  bool redo;  // FAST: valid pixel outside the interior band -- decided by exact_decisions, not by FAST arithmetic
  int jr;
  double x, y, zq, u, v;  // zq = z (STRICT) or 1/z (FAST)
  Win w;
  double wr[4];
};

// f64 plane element at a 32-bit BYTE offset: uniform 64-bit base + zero-extended 32-bit lane offset
// is the addressing mode of global_load (saddr + voffset), so no per-lane 64-bit address arithmetic
__device__ __forceinline__ double ld_f64(const double *base, unsigned byte_off) {
  return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + byte_off);
}

// one pixel's tile entry, as loaded (the round loops fetch the NEXT round's entry before they work on
// the current one, so the L2 / Infinity-Cache latency of the tile stream overlaps the arithmetic)
struct TileIn {
  double x, y, z;
  double wr[4];
  int jr;
};

__device__ __forceinline__ void load_tile_xyz(const EvalParams &P, unsigned gi, TileIn &t) {
  t.jr = P.t.JR[gi];
  const unsigned bo = gi << 3;  // nloc * pstride * 8 < 2^32 (nid_create)
  t.x = ld_f64(P.t.X, bo); t.y = ld_f64(P.t.Y, bo); t.z = ld_f64(P.t.Z, bo);
}
typedef double f64x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void load_tile_w(const EvalParams &P, unsigned gi, unsigned plane, TileIn &t) {
  // the four reference weights of a slot are adjacent (32 B): two 16-byte loads on one uniform base and one 32-bit
  // lane offset, no per-plane address arithmetic (32 * nloc * pstride < 2^32, nid_create); a wave reads 2 KB in a row
  (void)plane;
  const f64x2 *w = reinterpret_cast<const f64x2 *>(reinterpret_cast<const char *>(P.t.W) + (gi << 5));
  const f64x2 a = w[0], b = w[1];
  t.wr[0] = a.x; t.wr[1] = a.y; t.wr[2] = b.x; t.wr[3] = b.y;
}
__device__ __forceinline__ void load_tile(const EvalParams &P, unsigned gi, unsigned plane, TileIn &t) {
  load_tile_xyz(P, gi, t);
  load_tile_w(P, gi, plane, t);
}

// the 4x4 window at origin (w.wx, w.wy) >= (-1, -1); the image buffer starts at pixel (-1, -1)
__device__ __forceinline__ void load_window(const EvalParams &P, Win &w) {
  const unsigned st = (unsigned)P.im1_stride;
  const unsigned po = __umul24((unsigned)(w.wy + 1), st) + (unsigned)(w.wx + 1);  // (rows, stride < 2^24: one v_mad_u32_u24)
  w.r0 = load_row_at(P.im1s, po);
  w.r1 = load_row_at(P.im1s, po + st);
  w.r2 = load_row_at(P.im1s, po + 2u * st);
  w.r3 = load_row_at(P.im1s, po + 3u * st);
}

// What the FAST main passes load instead of the whole window.  Cost phase: the 2x2 cell of the sample = taps 1, 2
// of window rows 1 and 2 (two unaligned dwords).  Jacobian phase: rows 1 and 2 whole, of rows 0 and 3 taps 1, 2.
struct WinC { unsigned c1, c2; };
struct WinJ { WRow r1, r2; unsigned c0, c3; };
// ... addressed by the element offset e0 = iy * stride + ix of the sample's own pixel (the window's origin is one up and one
// to the left, and the buffer starts at pixel (-1, -1): the two cancel): one multiply-add, ONE select for the lanes
// without a sample, the constant stride + 1.
__device__ __forceinline__ unsigned win_e0(const EvalParams &P, bool in, double u, double v) {
  const unsigned e = __umul24((unsigned)(int)v, (unsigned)P.im1_stride) + (unsigned)(int)u;
  return in ? e : 0u;
}
__device__ __forceinline__ void load_win_centre_e(const EvalParams &P, unsigned e0, WinC &w) {
  const unsigned st = (unsigned)P.im1_stride;
  const unsigned po = e0 + (st + 1u);  // row 1, tap 1 = the sample's own pixel
  w.c1 = load_u32_at(P.im1s, po);
  w.c2 = load_u32_at(P.im1s, po + st);
}
__device__ __forceinline__ void load_win_jac_e(const EvalParams &P, unsigned e0, WinJ &w) {
  const unsigned st = (unsigned)P.im1_stride;
  w.c0 = load_u32_at(P.im1s, e0 + 1u);
  w.r1 = load_row_at(P.im1s, e0 + st);
  w.r2 = load_row_at(P.im1s, e0 + 2u * st);
  w.c3 = load_u32_at(P.im1s, e0 + 3u * st + 1u);
}
__device__ __forceinline__ int lo16(unsigned w) { return __builtin_amdgcn_sbfe((int)w, 0u, 16u); }
__device__ __forceinline__ int hi16(unsigned w) { return __builtin_amdgcn_sbfe((int)w, 16u, 16u); }

// FAST centre sample from the 2x2 cell (two lerps on exact integer tap differences)
__device__ __forceinline__ double sample_fast_c(const WinC &w, double u, double v) {
  const int ix = (int)u, iy = (int)v;
  const double dx = frac_nonneg(u, ix), dy = frac_nonneg(v, iy);
  const int a11 = lo16(w.c1), a12 = hi16(w.c1), a21 = lo16(w.c2), a22 = hi16(w.c2);
  const double m1 = fma(dx, (double)(a12 - a11), (double)a11);
  const double m2 = fma(dx, (double)(a22 - a21), (double)a21);
  return fma(dy, m2 - m1, m1);
}
// FAST gradient (TWICE the central difference, see gradient_fast_interior) and centre sample from WinJ
__device__ __forceinline__ void gradient_fast_j(const WinJ &w, double u, double v, double &gx, double &gy, double &ic) {
  const int ix = (int)u, iy = (int)v;
  const double dx = frac_nonneg(u, ix), dy = frac_nonneg(v, iy);
  const int a10 = lo16(w.r1.x), a11 = hi16(w.r1.x), a12 = lo16(w.r1.y), a13 = hi16(w.r1.y);
  const int a20 = lo16(w.r2.x), a21 = hi16(w.r2.x), a22 = lo16(w.r2.y), a23 = hi16(w.r2.y);
  const double g1 = fma(dx, (double)((a13 - a12) - (a11 - a10)), (double)(a12 - a10));
  const double g2 = fma(dx, (double)((a23 - a22) - (a21 - a20)), (double)(a22 - a20));
  gx = fma(dy, g2 - g1, g1);
  const int a01 = lo16(w.c0), a02 = hi16(w.c0), a31 = lo16(w.c3), a32 = hi16(w.c3);
  const double m0 = fma(dx, (double)(a02 - a01), (double)a01);
  const double m1 = fma(dx, (double)(a12 - a11), (double)a11);
  const double m2 = fma(dx, (double)(a22 - a21), (double)a21);
  const double m3 = fma(dx, (double)(a32 - a31), (double)a31);
  gy = fma(dy, (m3 - m2) - (m1 - m0), m2 - m0);
  ic = fma(dy, m2 - m1, m1);
}

// FAST warp of one tile entry: f.jr, f.x, f.y, f.zq = 1/z, f.u, f.v -- pixel_front<false> without its border tests
// (the same operations on the same values: the two give the same bits)
// (t0, t1, t2: the matrix's translation column -- M[3], M[7], M[11] -- from VECTOR registers: an FMA takes one scalar
// operand, so fma(M[2], z, M[3]) costs a v_mov of M[3] per sample unless the caller parks the column in registers once)
__device__ __forceinline__ void warp_fast(const EvalParams &P, const SlotArgs &SA, const TileIn &t, PixelFront &f, double t0, double t1, double t2) {
  const Geometry &g = P.g;
  const double *M = SA.pose.M;
  const double lx = t.x, ly = t.y, lz = t.z;
  const double qx = fma(M[0], lx, fma(M[1], ly, fma(M[2], lz, t0)));
  const double qy = fma(M[4], lx, fma(M[5], ly, fma(M[6], lz, t1)));
  const double qz = fma(M[8], lx, fma(M[9], ly, fma(M[10], lz, t2)));
  const double iz = rcp_fast(qz);
  f.jr = t.jr;
  f.x = qx; f.y = qy; f.zq = iz;
  f.u = fma(g.fx * qx, iz, g.cx);
  f.v = fma(g.fy * qy, iz, g.cy);
}

// OPAQUE (FAST main passes): (u, v) pass through an empty asm statement right behind the warp, so that the scheduler
// cannot hoist the next round's tile loads above it (see cost_round) -- here, in front of everything that is derived from
// them: behind the border tests it made the sample's (int)u, (int)v a second pair of conversions of "other" values.
template <bool STRICT, bool OPAQUE = false>
__device__ __forceinline__ void pixel_front(const EvalParams &P, const SlotArgs &SA, const TileIn &t, PixelFront &f) {
  const Geometry &g = P.g;
  f.jr = t.jr;
  const double lx = t.x, ly = t.y, lz = t.z;
#pragma unroll
  for (int k = 0; k < 4; k++) f.wr[k] = t.wr[k];
  double qx, qy, qz, u, v;
  if (STRICT) {
    xform_point(SA.pose, lx, ly, lz, qx, qy, qz);
    const double rz = rcp_fast(qz);  // one reciprocal for both quotients: div_shared
    const bool zm = exp_mid(qz);
    u = div_shared(g.fx * qx, qz, rz, zm) + g.cx;  // types_six_dof_expmap.cpp:562-563
    v = div_shared(g.fy * qy, qz, rz, zm) + g.cy;
    f.zq = qz;
  } else {
    const double *M = SA.pose.M;
    qx = fma(M[0], lx, fma(M[1], ly, fma(M[2], lz, M[3])));
    qy = fma(M[4], lx, fma(M[5], ly, fma(M[6], lz, M[7])));
    qz = fma(M[8], lx, fma(M[9], ly, fma(M[10], lz, M[11])));
    const double iz = rcp_fast(qz);
    u = fma(g.fx * qx, iz, g.cx);
    v = fma(g.fy * qy, iz, g.cy);
    f.zq = iz;
    if (OPAQUE) asm volatile("" : "+v"(u), "+v"(v));
  }
  f.redo = false;
  f.x = qx; f.y = qy; f.u = u; f.v = v;
  if (STRICT) {
    f.in = (f.jr >= 0) && (u >= 0 && u + 3 <= g.cols && v >= 0 && v + 3 <= g.rows);
    f.jin = f.in;  // re-decided in the Jacobian phase from fx*(x/z)+cx (Q6)
    f.w.wx = f.in ? (int)u - 1 : 0;  // out-of-frame pixels load the window at (0,0); never used
    f.w.wy = f.in ? (int)v - 1 : 0;
  } else {
    // FAST u, v carry ~1e-12 of error, so FAST arithmetic decides a border test only when the value is at least
    // kBorderEps away from equality (five compares against precomputed bounds); a pixel whose FAST tests
    // leave any doubt is flagged (`redo`, classified in the rarely taken branch of the pixel loops: clearly out
    // of frame, clearly outside the Jacobian's narrower bound, or within kBorderEps of a border -> exact_decisions).
    // The window origin is NOT clamped: ((int)u - 1, (int)v - 1) may be -1 (the image buffer has zeroed
    // margins), so the fixed-tap sample applies to every in-frame pixel.
    // (plain `&`: five compares in a row; `&&` makes the compiler branch around the later ones)
    const unsigned hu = (unsigned)__double2hiint(u) - P.hu_lo, hv = (unsigned)__double2hiint(v) - P.hu_lo;
    const bool in = (f.jr >= 0) & (hu <= P.hu_span) & (hv <= P.hv_span);
    f.in = in;
    f.jin = in && (hu <= P.hj_span);
    f.redo = (f.jr >= 0) && !f.jin;
    f.w.wx = in ? (int)u - 1 : 0;
    f.w.wy = in ? (int)v - 1 : 0;
    f.e0 = win_e0(P, in, u, v);
  }
}

// FAST: what `redo` means for a pixel, from FAST values that are at least kBorderEps away from every border they
// are compared with.  Returns true when the pixel needs exact_decisions (a border within kBorderEps, or NaN).
__device__ __forceinline__ bool classify_redo(const EvalParams &P, const PixelFront &f) {
  // (conservative integer forms of u < -eps, u > cols-3+eps, ...: the upper bounds from the ranges' ends -- the high
  // dword of cols-3+eps is at most one above that of cols-3-eps --, so that no further scalar registers are needed;
  // a NaN is "clearly out" like in the reference, whose comparisons all fail)
  const int hu = __double2hiint(f.u), hv = __double2hiint(f.v);
  const unsigned neg_out = 0x80000000u + kBorderEpsHi;  // below -eps' (the sign bit set, magnitude beyond the band)
  const int u_top = (int)(P.hu_lo + P.hu_span + 2u), v_top = (int)(P.hu_lo + P.hv_span + 2u), j_top = (int)(P.hu_lo + P.hj_span + 2u);
  // (a pure function: `redo` lanes have jin == false already, and one that is clearly out of frame cannot have passed
  // pixel_front's range checks, so in == false too -- writing them again made every flag a phi of this branch)
  const bool clearly_out = ((unsigned)hu > neg_out) | ((unsigned)hv > neg_out) | (hu > u_top) | (hv > v_top);
  // clearly in frame for the cost, clearly outside linearizeOplus' narrower bound (cols-1, :433)
  const bool in_not_jin = f.in & (hu > j_top);
  return !(clearly_out | in_not_jin);
}

// ---- FAST math: decisions that are discontinuous in the arithmetic -----------------------------------
// FAST mode evaluates the SAME algorithm with cheaper arithmetic; every quantity it feeds into the
// histograms is a continuous function of (u, v, ic), so a few ulp of difference stay a few ulp -- except
// where the reference takes a DECISION on a value that sits within rounding of the decision threshold:
//   * the frame-border tests u >= 0, u+3 <= cols (cost) / cols-1 (Jacobian), v >= 0, v+3 <= rows
//     (types_six_dof_expmap.cpp:565, :433) -- an identity-like pose projects every pixel onto integer
//     coordinates +- an ulp, so whole rows / columns of pixels sit ON the border;
//   * the saturation clamp ic >= 255 -> 254.999 (:572-573): with all four taps at 255 the reference's
//     four-term bilinear sum lands on either side of 255.0 by its last rounding, a 1e-3 intensity jump;
//   * ic < 0 -> 0 (:574-575) and the u == 0 quirk of the B-spline derivative (Q5).
// FAST arithmetic therefore decides only the pixels that are safe from all of that (the interior band of
// pixel_front and the clamp guard, outside_clamp_guard); the others (PixelFront::redo, ic within 1/8 of 0 or 255) are
// decided here with the reference's own operation order -- xform_point in the configured mode, IEEE
// divisions, fx*x/z + cx, the (int)-truncating four-term bilinear form on the register window,
// fx*(x/z) + cx for the Jacobian's border test (Q6) -- so FAST takes bit for bit the decisions STRICT takes
// and the clamp sees the reference's intensity.  Everything else about the pixel (weights, gradient,
// d(u,v)/d(xi)) stays FAST.  Rare on natural images: the pixels that leave the frame or land within a
// pixel of its border, and every pixel of a saturated (flash) or black region.
// Returns with f.in / f.jin / f.u / f.v replaced and, for f.in, the whole window f.w loaded and ic valid.
// EXT launches keep only the matrix of the pose in scalar registers (see k_eval2); the quaternion is fetched here,
// where it is needed, from the launch's record array.
// FULLWIN: the whole 4x4 window around the reference's (u, v) is loaded into f.w (the Jacobian phase's gradient needs it);
// otherwise (cost phase) the sample is taken from its 2x2 cell alone and f.w is not touched.
template <bool EXT, bool FULLWIN = true>
__device__ __forceinline__ void exact_decisions(const EvalParams &P, const SlotArgs &SA, int pose_idx, const TileIn &t,
                                                PixelFront &f, double &ic) {
  const Geometry &g = P.g;
  const double lx = t.x, ly = t.y, lz = t.z;  // the second pass has the tile entry already
  double qx, qy, qz;
#ifdef NID_ABL_EXACT_MATRIX  // (ablation builds only: the FAST matrix transform in place of the configured one -- other bits)
  if (true) {
    const double *M = SA.pose.M;
    qx = fma(M[0], lx, fma(M[1], ly, fma(M[2], lz, M[3])));
    qy = fma(M[4], lx, fma(M[5], ly, fma(M[6], lz, M[7])));
    qz = fma(M[8], lx, fma(M[9], ly, fma(M[10], lz, M[11])));
  } else
#endif
  if (EXT && SA.pose.mode == 0) {
    Pose pq;
    // scalar loads through the constant address space, like the rest of the record (k_eval2)
    typedef const unsigned __attribute__((address_space(4))) *ConstDwords;
    ConstDwords src = (ConstDwords)(reinterpret_cast<uintptr_t>(&P.slots_ext[pose_idx].pose));
    unsigned *dst = reinterpret_cast<unsigned *>(&pq);
#pragma unroll
    for (unsigned i = 0; i < kPoseQuatDwords; i++) dst[i] = src[i];
    pq.mode = 0;
    xform_point(pq, lx, ly, lz, qx, qy, qz);
  } else {
    xform_point(SA.pose, lx, ly, lz, qx, qy, qz);
  }
  const double rz = rcp_fast(qz);  // one reciprocal for both quotients (div_shared: the IEEE quotients' bits)
  const bool zm = exp_mid(qz);
  const double u = div_shared(g.fx * qx, qz, rz, zm) + g.cx;  // types_six_dof_expmap.cpp:562-563
  const double v = div_shared(g.fy * qy, qz, rz, zm) + g.cy;
  const bool in = (f.jr >= 0) && (u >= 0 && u + 3 <= g.cols && v >= 0 && v + 3 <= g.rows);
  bool jin = false;
  if (in) {
    if (FULLWIN) {
      // the whole window around the REFERENCE's (u, v) -- which may truncate to the pixel next to FAST's
      f.w.wx = (int)u - 1; f.w.wy = (int)v - 1;
      load_window(P, f.w);
      ic = bilinear_rows(win_rows(f.w, v), f.w.wx, u);
    } else {
      ic = bilinear_cell_exact(P, u, v);  // (the same four taps in the same expression: the same bits)
    }
    if (ic < 0) ic = 0.0;                              // :574-575
    // linearizeOplus' own projection fx*(x/z)+cx (:407-422, Q6) differs from u by a rounding: it can decide a border
    // test differently only within an ulp of the border -- two more divisions, taken only by samples that close
    double uj = u, vj = v;
    if (
        fabs(u) < kBorderEps || fabs(v) < kBorderEps || fabs(u + 3 - (double)P.jac_cols) < kBorderEps || fabs(v + 3 - (double)g.rows) < kBorderEps) {
      uj = g.fx * div_shared(qx, qz, rz, zm) + g.cx;
      vj = g.fy * div_shared(qy, qz, rz, zm) + g.cy;
    }
    jin = (uj >= 0 && uj + 3 <= P.jac_cols && vj >= 0 && vj + 3 <= g.rows);
  }
  f.in = in; f.jin = jin; f.u = u; f.v = v;
}

// FAST: clamped centre sample -> bin position -> span; returns jc, pc.  CLAMP = false: the caller knows
// ic < 255 (the main passes only take samples inside the clamp guard, outside_clamp_guard; the clamp's
// compare and two selects then cost nothing)
// EXACT (the second passes, whose `ic` is the reference's own sample): the bin position with the reference's own
// rounding, (ic * S) / 255 (types_six_dof_expmap.cpp:574) -- a sample an ulp below 255 sits 1 - t = one or two ulps of
// S below the last knot, the end span's linear weight 3 (1 - t) is that distance, and a bin made of such weights alone
// enters the Jacobian through log2 of its mass: ic * (S / 255) is the neighbouring double for some S (S = 2: twice the
// weight, W off by 1 of 59 -- 2.3e-3 of a one-sample cell's Jacobian, sweep seed 344412).  The main passes only take
// samples at least 1/8 away from 0 and 255 (the clamp guard): there an ulp of S is 1e-13 of 1 - t and below.
template <bool CLAMP = true, bool EXACT = false>
__device__ __forceinline__ int fast_bin(double &ic, int S, double &pc) {
  if (CLAMP) { if (ic >= 255) ic = 254.999; }
  if (EXACT) {
    pc = div_255(ic * (double)S);
    return min((int)pc, S - 1);  // (a quotient that rounds up to S itself: t = 1 on the last span)
  }
  pc = ic * ((double)S / 255.0);
  return (int)pc;
}

// centre sample -> clamped intensity -> bin position -> B-spline weights (and derivatives)
template <bool STRICT, bool WANT_DER>
__device__ __forceinline__ int pixel_sample(const PixelFront &f, int nb, int S, const double *rtab, double &ic,
                                            double wc[4], double dw[4]) {
  int jc;
  if (STRICT) {
    const RowPair rp = win_rows(f.w, f.v);
    ic = bilinear_rows(rp, f.w.wx, f.u);
    if (ic >= 255) ic = 254.999;
    if (ic < 0) ic = 0.0;
    const double pc = div_255(ic * ((double)nb - 3.0));
    jc = (int)floor(pc);
    bspline4_tab<WANT_DER>(pc, jc, S, rtab, wc, dw);
  } else {  // (the FAST loops of k_eval2 inline this with the exact_decisions guard in front)
    ic = sample_fast_interior(f.w, f.u, f.v);
    double pc;
    jc = fast_bin(ic, S, pc);
    bspline4_poly<WANT_DER>(pc, jc, rtab, wc, dw);
  }
  return jc;
}

// Tuning switches (tools/build_variant.py; results in profiles/r02_ablations_A.txt).  NID_FAST_WAVES: occupancy target
// of the FAST kernels (5 waves/SIMD = 96 VGPRs; 6 spills inside the loops: 216.8 k).  The EXT kernels (the throughput
// launches) are compiled with a budget of 128 registers: the cost + Jacobian kernel still ends at 95 (five waves per
// SIMD), but the scheduler, no longer at its limit, orders the window loads and their first uses better -- the same
// instructions, 6 % less time per launch (profiles/r02_ablations_A.txt, "w4").
// LAT kernels: what the Jacobian phase needs of a pixel the cost phase has already worked out
struct LatPix {
  double x, y, iz, gx, gy, pc;
  double wr[4];
  int jr, jc;
  bool go;  // contributes to the Jacobian in the main pass: inside linearizeOplus' frame and decided by FAST arithmetic
};

#ifndef NID_FAST_WAVES
#define NID_FAST_WAVES 5
#endif
#ifndef NID_EXT_WAVES
#define NID_EXT_WAVES 4
#endif
// EXT: the launch has more than kMaxBatch poses and their records live in P.slots_ext (device memory); the
// workgroup's record is then pulled into scalar registers once (scalar loads through the constant address space), so
// that the pose matrix and the pointers are SGPR operands exactly as when they come from the kernel arguments.
// LAT > 0 (FAST math, launches of few poses: one workgroup per CU or less): the latency form of the pixel loops.
// A launch that cannot fill the chip is bound by the DEPENDENT memory round trips of one wave -- tile entry ->
// projection -> target window -> sample -- once per round and phase.  With NT >= 512 a cell is LAT <= 3 rounds,
// so the rounds are unrolled and staged: every round's tile loads are issued first, then every round's window
// loads, then the arithmetic; and what the Jacobian phase needs of a pixel (its point over z, the image gradient,
// the bin position, the reference weights: LatPix) stays in registers across the fold, so the Jacobian phase of
// the main pass touches no global memory at all.  One workgroup per CU leaves 128+ VGPRs per lane for that.
// Same operations on the same values in the same per-lane order as the loop form: bit-identical results (tested).
// BIG: cells of more than 32 * NT slots (more rounds per wave than a lane's 32-bit gomask has bits).
// what the resident kernel keeps of its cell from one request to the next
struct ResCell {
  int n_c;
  double href;
  bool fresh;  // first request of this kernel: the histograms have not been zeroed yet
};

// The body of the evaluation kernels: one cell at one pose by one workgroup.  RES: called from the resident kernel
// (k_resident) -- the B-spline table is in LDS already, the cell's count and reference entropy are in registers, and
// the histograms were zeroed behind the previous request (by the waves that had nothing left to do).
// REPAIR_INLINE: the repair pass (kLinFlagW) is part of this instantiation.  The loop-form kernels -- the throughput
// path -- are compiled WITHOUT it: merely carrying its code between the two phases cost them 7 % on data that never needs
// it (same-box A/B, profiles/r04_ablations_A.txt: the flags and the fold's detection cost nothing, the repair block's
// presence does -- the kernel sits at the limit of its scalar registers, the block's needs come on top of everything
// that lives across the fold, and the allocator parks the LONGEST lives, the pointers and bounds of both pixel loops,
// in vector lanes: v_readlane + hazard nops inside the hot loops.  A second copy of the body behind the first, an
// out-of-line function, arguments re-read per phase, a scalar-register diet: all measured, all slower than no repair
// code).  A workgroup of theirs whose fold finds a bin to repair publishes NOTHING: it puts (pose, cell) into the
// launch's repair queue and leaves; k_repair -- a kernel of its own, enqueued behind every such launch -- does the queued
// cells from the start with the instantiation that repairs inline and publishes them like any other (the in-launch
// reduction is "whoever arrives last", across kernels too; the DIRECT protocol's host polls every record).  Same
// algorithm either way: the same bits.  The latency-form, resident, diagnostic and big-cell kernels repair inline
// (other register budgets; the latency paths measured unchanged).
__device__ __forceinline__ constexpr bool repair_inline_default(int lat, bool res, bool dbg, bool big) { return lat > 0 || res || dbg || big; }

// The kernel's arguments ONCE MORE, through a pointer the optimiser cannot connect with the one it has been using: what is
// read through it is a NEW scalar load from the kernarg segment where it is needed, not a longer life of a register
// loaded at the head of the kernel (EvalParams is the FIRST argument of every kernel that runs eval_cell, i.e. it sits
// at the start of the kernarg segment).  For what only a cold block needs: the kernels sit at the limit of their scalar
// registers, and two more that live from the head of the kernel to the fold change what the allocator parks in vector
// lanes inside the pixel loops (measured: the repair queue's pointer alone, +8 % kernel time).
__device__ __forceinline__ const EvalParams &reread_args() {
  typedef const EvalParams __attribute__((address_space(4))) *ConstParams;
  uintptr_t a = reinterpret_cast<uintptr_t>(__builtin_amdgcn_kernarg_segment_ptr());
  asm volatile("" : "+s"(a));
  return *(const EvalParams *)(ConstParams)a;
}

// A READER'S MAP of eval_cell (one function on purpose: its phases share the LDS layout, the per-cell scalars in SGPRs and
// the register allocation -- out-of-line pieces and second copies were measured, see above; the pieces are lambdas, in this order):
//   LDS layout, per-cell scalars, level-1 edge exit ....... the head
//   zero_histograms
//   hist_add(jr, jc, wr, wc, ...) ......................... ONE sample's 20 addends: the ordinary path at its end (4 marginal
//       lo_add_marginal, fine_residual, add_c / add_j,        + 16 joint ds_add_u64), in front of it the rare branches -- clamped /
//       flag_linear                                            near-saturated groups, fine levels, REPAIR routing (kLinFlagW)
//   phase 1, cost: strict_cost_loop | cost_round<main / second pass> | repair_round, and the loops that drive them
//       (LAT: rounds unrolled and staged); then the barrier
//   fold: fold_bin (copies + fine levels + groups -> p, W, p log p), the repair decision (lin_word, repair_wanted),
//       entropies Hc / Hj in a shape-independent order, residual_and_huber; cost-only launches end here (tail: done,
//       deferred_to_repair, the DIRECT records or the in-launch reduction finish_and_reduce_w0)
//   phase 2, Jacobian: the contracted tables (FAST), jac_accumulate_fast | jac_accumulate (STRICT: the dw[4] form),
//       jac_round<main / second pass> | jac_round_masked (the cost phase's decisions, gomask), the LAT form from registers
//   block sum of the six accumulators (through LDS for 128 / 256 threads, DPP + LDS beyond), the cell's quadratic form, tail
#ifdef NID_CENSUS
static __device__ unsigned long long g_census[64];
#endif
template <int NT, bool JAC, bool STRICT, int NB, bool DBG, bool EXT, int LAT, bool BIG, bool RES, bool REPAIR_INLINE = repair_inline_default(LAT, RES, DBG, BIG),
          bool PRESET = false>
__device__ __forceinline__ bool eval_cell(const EvalParams &P, const SlotArgs &SA, const int cl, const int pose_idx, unsigned char *smem,
                                          const ResCell rc = ResCell{0, 0.0, true}, const uint2 preset = uint2{0u, 0u}) {
  static_assert(!PRESET || (REPAIR_INLINE && LAT == 0 && !RES), "PRESET: k_repair's instantiation");
  static_assert(LAT == 0 || (!STRICT && !EXT), "the latency form exists for FAST math launches of <= kMaxBatch poses (DBG: phase stamps only)");
  const int tid = threadIdx.x;
  constexpr int NC = eval_hist_copies(NT, NB);
  const Geometry &g = P.g;
  const int nb = NB > 0 ? NB : g.nb;
  const int nbins = nb * nb + nb;
  const int S = nb - 3;
  // LDS: [weight table | B-spline table | reduction scratch | histogram copies].  The two tables come
  // first so that their reads sit inside the 2040-byte immediate range of ds_read2_b64.
  double *tab = reinterpret_cast<double *>(smem);
  double *term = tab + ((nbins + 1) & ~1);  // p log2 p of every bin (entropy sums in a workgroup-shape independent order)
  double *rtab = term + ((nbins + 1) & ~1);
  // (the cell's flag words sit in front of the histogram area: the Jacobian block sum reuses that area, and the tails
  // still look at repair_set)
  unsigned *clamp_flag = reinterpret_cast<unsigned *>(rtab + S * kCoefRow);  // [0]: clamped samples seen, [1]: near-saturated ones
  double *red = rtab + S * kCoefRow + kFlagDoubles;  // (the tables: S rows of kCoefRow (FAST) or kRcpRow <= kCoefRow (STRICT) doubles)
  unsigned long long *hist = reinterpret_cast<unsigned long long *>(red + kRedDoubles(NT));
  unsigned long long *hist_lo = hist + nbins * NC;  // [kFineLevels][nbins], single copies (see kTinyW)
  unsigned long long *clampb = hist_lo + kFineLevels * nbins;                   // [kClampBins(nb)][kClampCopies]: see kClampBins
  unsigned long long *clamp_lo = clampb + kClampBins(nb) * kClampCopies;         // [kFineLevels][kClampBins(nb)], single copies
  double *rclamp = reinterpret_cast<double *>(clamp_lo + kFineLevels * kClampBins(nb));  // [kClampBins(nb)] folded sums, then the flags
  unsigned *flag_pad = reinterpret_cast<unsigned *>(rclamp + kClampBins(nb));            // (24 bytes, unused: keeps the alignment of what follows)
  unsigned *lin_flag = clamp_flag + 2;  // [0]: target column 1, [1]: column nb - 2 -- bit a: joint row a, bit 16: the marginal bin (kLinFlagW)
  unsigned *repair_set = clamp_flag + 4;  // the bins of those two columns that the fold wants repaired (same bits)
  unsigned *grp_flag = clamp_flag + 6;    // column nb - 2 again: the bins the clamped / near-saturated GROUPS' linear weight lands in (their
                                          // reference weights are known as sums per row: a bound relative to them, kRepairRel)
  // ... and the same for the NEAR-SATURATED samples (see kNearSatIc): folded sums behind the flags; the bins themselves
  // borrow the area of the two weight tables, which nobody touches before the fold (when they fit there: nb >= 5)
  double *rns = reinterpret_cast<double *>(flag_pad + kFlagPadWords);                     // [nb + 1]
  // (not in the resident kernel, which clears the bins for the next request while wave 0 still reads the tables)
  const bool ns_alias = !RES && near_sat_aliased(nb);
  unsigned long long *ns_own = reinterpret_cast<unsigned long long *>(rns + ((nb + 1) | 1));  // (rns sits 8 bytes past a 16-byte boundary: an odd count ends on one)
  unsigned long long *nsb = ns_alias ? reinterpret_cast<unsigned long long *>(tab) : ns_own;  // [kClampBins(nb)][kClampCopies]
  unsigned long long *ns_lo = nsb + kClampBins(nb) * kClampCopies;                       // [kFineLevels][kClampBins(nb)]
  unsigned char *lds_tail = reinterpret_cast<unsigned char *>(ns_alias ? reinterpret_cast<unsigned long long *>(rns + nb + 1)
                                                                        : ns_own + kNearSatBinBytes(nb) / 8);

  // DIRECT launches are launches of ONE pose: the kernels for more than kMaxBatch poses (EXT: the throughput path) are
  // compiled without that code (five registers of the hot kernel)
  const bool direct_launch = !EXT && (unsigned)(SA.host_quad - 1) < 2u;  // (3: GROUP-DIRECT keeps the in-launch tail, see finish_and_reduce_w0)
  const int n_c = RES ? rc.n_c : P.Nc[cl];
  const double href = RES ? rc.href : P.Href[cl];
  double *out = SA.cellout + (size_t)cl * kCellOut;
  const bool want_cellout = SA.cellout != nullptr;  // (the pipelined loops hand over none: ten stores per cell and pose nobody reads)
  double *quad = SA.quad + (size_t)cl * kQuad;
  if (n_c < 300 || isnan(href)) {  // level-1 edge: never evaluated (computeH.cu:271-275)
    if (tid >= 64) return false;
    if (direct_launch) {  // see the cost-only tail
      if (tid < kCellOut && SA.cellout_host) store_sys(out + tid, (tid == kCellOut - 1) ? (double)n_c : NAN);
      else if (!RES && tid < kCellOut && want_cellout) out[tid] = (tid == kCellOut - 1) ? (double)n_c : NAN;  // the slot's device block (nid_slot_buffers; the resident kernel's `cellout` is the per-cell calls' host buffer)
      if (tid < kDirectRec && SA.host_quad == 1) {  // both records: the host does not know yet that the cell is a level-1 edge
        store_sys(SA.quad + (size_t)cl * kDirectRec + tid, 0.0);
        if (JAC) store_sys(SA.quad + ((size_t)P.g.nloc + cl) * kDirectRec + tid, 0.0);
      }
      return false;
    }
    if (tid < kCellOut && want_cellout) out[tid] = (tid == kCellOut - 1) ? (double)n_c : NAN;
    if (tid < kQuad) store_sc1(quad + tid, 0.0);
    finish_and_reduce_w0(P, SA, cl, tid);
    return false;
  }
  NID_STAMP(0);
  const int copy = (NC & (NC - 1)) == 0 ? (tid & (NC - 1)) : (tid % NC);  // (a copy count that is no power of two: experiments, NID_HIST_COPIES_NB10)
  const unsigned base = (unsigned)cl * (unsigned)g.pstride;
  const unsigned plane = (unsigned)g.nloc * (unsigned)g.pstride;
  // pstride is a multiple of 64, so a wave is either entirely inside the tile or entirely past
  // it: the round loops run on a wave-uniform bound and the last round costs idle waves nothing
  const int wave_base = __builtin_amdgcn_readfirstlane(tid & ~63);
  const int lane = tid & 63;
  TileIn pre, prej;  // loop form of the FAST pixel loops: the next round's point and bin index (see cost_round)
  (void)pre; (void)prej;
  // the copies and the fine levels behind them, 16 bytes per store (nbins is even, hist is 16-byte aligned); threads
  // first .. NT-1 share the work
  auto zero_histograms = [&](int first) {
    uint4 *h4 = reinterpret_cast<uint4 *>(hist);
    const int n4 = (nbins * (NC + kFineLevels) + kClampBins(nb) * (kClampCopies + kFineLevels)) / 2;  // ... and the clamped samples' bins
    if (tid == first) {
#pragma unroll
      for (int i = 0; i < kFlagWords; i++) clamp_flag[i] = 0u;
    }
    // the near-saturated samples' bins (elsewhere in LDS, see nsb) ride behind: the slots n4 .. n4 + nns - 1 of the same
    // loop -- with 8 bins and 128 threads they fall into the idle threads of its last iteration
    uint4 *n4p = reinterpret_cast<uint4 *>(nsb);
    const int nns = (!STRICT && NID_CLAMP_BINS) ? kNearSatBinBytes(nb) / 16 : 0;
    if (NB > 0 && first == 0) {
#pragma unroll
      for (int i = 0; i < (n4 + nns + NT - 1) / NT; i++) {
        const int idx = i * NT + tid;
        if ((i + 1) * NT <= n4 || idx < n4) h4[idx] = make_uint4(0u, 0u, 0u, 0u);
        else if (idx < n4 + nns) n4p[idx - n4] = make_uint4(0u, 0u, 0u, 0u);
      }
    } else {
      for (int i = tid - first; i < n4 + nns; i += NT - first) {
        if (i < n4) h4[i] = make_uint4(0u, 0u, 0u, 0u);
        else n4p[i - n4] = make_uint4(0u, 0u, 0u, 0u);
      }
    }
  };
  // (RES: the previous request's tail has cleared the histograms -- except in the throughput shapes, whose Jacobian block
  // sum goes through the histogram area AFTER the fold (NID_XPOSE_SUM) while the other waves are already gone)
  constexpr bool kResRezero = RES && NT <= 256 && NID_XPOSE_SUM;
  // Latency form (launched): the cell's tile entries are REQUESTED before the histograms are zeroed and the B-spline table is
  // copied, and warped -- target windows requested -- before the barrier behind those: neither needs LDS, and a launch of
  // one pose per CU has nothing else to hide the two dependent round trips behind (round 6: the phase stamps of the
  // single-pose kernel, profiles/r06_latency_A.txt).  The same operations on the same values: the same bits.
  constexpr bool kLatEarly = LAT > 0 && !RES;
  constexpr int kLatN = LAT > 0 ? LAT : 1;
  TileIn lat_tin[kLatN];
  PixelFront lat_fr[kLatN];
  WinC lat_wc[kLatN];
  WinJ lat_wj[kLatN];
  (void)lat_tin; (void)lat_fr; (void)lat_wc; (void)lat_wj;
  if constexpr (kLatEarly) {
#pragma unroll
    for (int q = 0; q < LAT; q++)
      if (wave_base + q * NT < g.pstride) load_tile(P, base + (unsigned)(wave_base + q * NT + lane), plane, lat_tin[q]);
  }
  if (!RES || rc.fresh || kResRezero) zero_histograms(0);
  if (RES) {
    // (the resident kernel has loaded the table once)
  } else if (STRICT) {
    if (tid < S * 6) {
      const int jj = tid / 6, e = tid % 6;
      rtab[jj * kRcpRow + e] = 1.0 / span_denominator(jj, e, S);
    }
  } else {
    // (the host's table carries kWcPre on its value coefficients, see hist_add / fx_bits)
    for (int i = tid; i < S * kCoefRow; i += NT) rtab[i] = P.ctab[i];
  }
  if constexpr (kLatEarly) {
#pragma unroll
    for (int q = 0; q < LAT; q++)
      if (wave_base + q * NT < g.pstride) {
        pixel_front<false>(P, SA, lat_tin[q], lat_fr[q]);
        if (JAC) load_win_jac_e(P, lat_fr[q].e0, lat_wj[q]);
        else load_win_centre_e(P, lat_fr[q].e0, lat_wc[q]);
      }
  }
  if (!RES || rc.fresh || kResRezero) __syncthreads();  // (a later resident request: the waves have met at the kernel's own barriers since)
  NID_STAMP(1);

  // ---- phase 1: cost ---------------------------------------------------------------
  // One round = 64 consecutive tile slots per wave.  The histogram update of one in-frame sample:
  // FAST math: the value polynomials of the LDS copy of the B-spline table carry kWcPre (a power of two: every
  // Horner step is the unscaled step times the scale, bit for bit), so the target weights arrive scaled (`win`
  // = wc * kWcPre, PRESCALED); the rare branches below get the plain weights back by the inverse scale (exact).
  // The reference weights take hist_dn (four multiplications per sample), and every add is the bit pattern of a
  // subnormal product (fx_bits).
  // (kTinyW * kWcPre as a signed compare on the high dwords: a weight that is negative by rounding takes the rare branch
  // too; a 32-bit literal instead of an f64 constant in scalar registers)
  constexpr int tiny_scaled_hi = (int)((1023u - 512u - (unsigned)NID_TINY_W_EXP) << 20);
  // REPAIR (see kLinFlagW): the same routing of every addend, but nothing is added except what NORMAL mode sent to the
  // COARSE copies of a bin in the repair set -- to the fine level of its own exponent
  unsigned rep_col1 = 0u, rep_colz = 0u;  // repair_set[0], [1] (wave-uniform; loaded before the repair pass)
  bool repaired = false;                  // this cell ran the repair pass (wave-uniform)
  // PRESET (k_repair, round 5): the repair set is KNOWN -- the loop-form kernel that queued this cell found it in its fold and
  // handed it over with the queue entry --, so the cost passes route at once what the repair pass would move: an addend
  // that NORMAL mode sends to the coarse copies of a bin in the set goes to the bin's fine levels instead (MIXED in
  // hist_add: the very adds of REPAIR mode, made one traversal earlier; integer sums, so the same bits), the fold finds
  // those bins' copies empty -- which IS the refold without the copies -- and the third traversal of the pixels is gone.
  if constexpr (PRESET) {
    rep_col1 = (preset.x >> 31) ? (preset.x & 0x1FFFFu) : 0u;
    rep_colz = (preset.y >> 31) ? (preset.y & 0x1FFFFu) : 0u;
    repaired = true;
    if (tid == 0 && P.repair_count) atomicAdd(P.repair_count, 1ull);  // (diagnostics: nid_debug_repair_count)
  }
  auto hist_add = [&](int jr, int jc, const double (&wr_in)[4], const double (&win)[4], auto prescaled, double pcv, int group, auto repair_tag) {
    constexpr bool PRESCALED = decltype(prescaled)::value;
    constexpr bool REPAIR = decltype(repair_tag)::value;
    constexpr bool MIXED = PRESET && !REPAIR;  // NORMAL mode with the repair set known beforehand (see PRESET)
    if (group != 0) {  // FAST second passes only: 1 = clamped (kClampBins), 2 = near-saturated (kNearSatIc)
      if (REPAIR) return;  // (their sums are exact relative to themselves: coarse copies for weights >= 2^-8 only)
#ifdef NID_ABL_NO_GROUP_ADDS  // (ablation builds only: the groups' classification without their LDS atomics -- wrong results)
      clamp_flag[group - 1] = 1u;
      return;
#endif
      unsigned long long *hx = (group == 2 ? nsb : clampb) + ((unsigned)copy & (kClampCopies - 1));
      unsigned long long *hlo = group == 2 ? ns_lo : clamp_lo;
      {
        // the count: every lane of the wave in this group would add the same number to the same bin -- one lane adds the lot
        const unsigned long long in2 = __builtin_amdgcn_ballot_w64(group == 2);
        const unsigned long long act = group == 2 ? in2 : (__builtin_amdgcn_ballot_w64(true) & ~in2);
        if ((unsigned)__builtin_ctzll(act) == (unsigned)(tid & 63))
          atomicAdd(hx + nb * kClampCopies, (unsigned long long)__builtin_popcountll(act) * fx_bits(P.hist_dn * kWcPre));
      }
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const double w = m == 0 ? fabs(wr_in[0]) : wr_in[m];
        if (w >= kFineW) {
          atomicAdd(hx + (jr + m) * kClampCopies, fx_bits(w * (P.hist_dn * kWcPre)));
        } else if (w > kNegligibleW) {
          const int lv = fine_level(w);
          atomicAdd(hlo + lv * kClampBins(nb) + (unsigned)(jr + m), fx_encode(w, fine_scale(lv)));
        }
      }
      clamp_flag[group - 1] = 1u;
      return;
    }
    // fixed-point encode: the scales are powers of two, so wcs = wc * kWcPre and wrs = wr * hist_dn are exact and
    // wrs * wcs rounds wr * wc * 2^s once, to an integer (fx_bits)
    double wcs[4];
#pragma unroll
    for (int k = 0; k < 4; k++) wcs[k] = PRESCALED ? win[k] : win[k] * kWcPre;
    const unsigned hrow = (unsigned)(__mul24(jr, nb) + jc);
    unsigned long long *hc = hist + ((unsigned)jc * NC + (unsigned)copy);
    unsigned long long *hj = hist + (((unsigned)nb + hrow) * NC + (unsigned)copy);
    double wr[4] = {wr_in[0], wr_in[1], wr_in[2], wr_in[3]};
    // A fine-level add of the TARGET histogram.  The fine levels are single copies; the lanes of a saturated (clamped:
    // one constant intensity) or black region all add the SAME value to the SAME bin -- a 64-way serialised LDS atomic
    // per weight and round (the flash pair ran 1.8x slower than the plain pair mostly for this).  If every active
    // lane agrees on address and value, one lane adds count * value (integers: the same sum).
    auto lo_add_marginal = [&](unsigned idx, unsigned long long val) {
#if NID_LO_AGGREGATE
      const unsigned i0 = (unsigned)__builtin_amdgcn_readfirstlane((int)idx);
      const unsigned v0l = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)val);
      const unsigned v0h = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(val >> 32));
      const unsigned long long v0 = ((unsigned long long)v0h << 32) | v0l;
      const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
      const unsigned long long eq = __builtin_amdgcn_ballot_w64(idx == i0 && val == v0);
      if (eq == act) {
        if ((unsigned)__builtin_ctzll(act) == (unsigned)(tid & 63))
          atomicAdd(hist_lo + idx, val * (unsigned long long)__builtin_popcountll(act));
        return;
      }
#endif
      atomicAdd(hist_lo + idx, val);
    };
    // What the fine-level add of `w` at level lv LEFT OUT, one level down (round 5; the constructed cases of kind "edges",
    // profiles/r05_adversarial.txt class F).  A level keeps 27..51 bits of an addend, depending on where in the level's
    // 24 binary orders the addend sits; hundreds of EQUAL addends at the bottom of a level (a step edge sampled 3e-12 px
    // past an integer position: every sample's linear end-span weight is 4e-10) leave the bin's mass with 1e-10 of
    // relative error, W = -(1 + log2 p) with 1.4e-10 -- times an O(1) derivative, in a cell whose Jacobian is the residue
    // of a cancellation.  For the addends that carry such derivatives -- the LINEAR end-span columns below kLinFlagW and
    // everything the repair pass moves -- the rounding residual r = w * 2^(59 + 24 lv) - RN(...) (exact: one fma) goes,
    // times 2^24 and rounded again, to level lv + 1 as a SIGNED integer (the fold reads the levels as signed): 51+ bits
    // of every such addend.  Rare inside rare; the last level has no level below it (addends below 2^-104).
    auto fine_residual = [&](int lv, unsigned bin, double w, auto marginal_tag) {
      if (lv >= kFineLevels - 1) return;
      const double sc = fine_scale(lv);
      const double hi = fma(w, sc, 0x1p52) - 0x1p52;  // RN(w * sc): the integer fx_encode added
      const double r = fma(w, sc, -hi);
      const unsigned long long ri = (unsigned long long)__double2ll_rn(r * 0x1p24);
      if (decltype(marginal_tag)::value) lo_add_marginal((unsigned)(lv + 1) * nbins + bin, ri);
      else atomicAdd(hist_lo + (unsigned)(lv + 1) * nbins + bin, ri);
    };
    auto linear_end_small = [&](int k, double w) { return w < kLinFlagW && ((k == 1 && jc == 0) || (k == 2 && jc == S - 1)); };
    // a COARSE add to the marginal bin jc + k / to the joint bin (jr + m, jc + k); REPAIR: to the fine levels, if the bin
    // is in the repair set (`w`, `pr`: the plain weight / the reference's own product, rounded once like there)
    auto rep_set = [&](int col) -> unsigned { return col == 1 ? rep_col1 : (col == nb - 2 ? rep_colz : 0u); };
    auto add_c = [&](int k, double wcs_k, double w) {
      if constexpr (!REPAIR && !MIXED) { atomicAdd(hc + k * NC, fx_bits(wcs_k * P.hist_dn)); return; }
      if (MIXED && !((rep_set(jc + k) >> 16) & 1u)) { atomicAdd(hc + k * NC, fx_bits(wcs_k * P.hist_dn)); return; }
      if (((rep_set(jc + k) >> 16) & 1u) && w > kNegligibleW) {
        const int lv = fine_level(w);
        atomicAdd(hist_lo + lv * nbins + (unsigned)(jc + k), fx_encode(w, fine_scale(lv)));
        fine_residual(lv, (unsigned)(jc + k), w, std::false_type{});
      }
    };
    auto add_j = [&](int m, int k, double wr_m, double wcs_k, double w) {
      if constexpr (!REPAIR && !MIXED) { atomicAdd(hj + (m * nb + k) * NC, fx_bits((wr_m * P.hist_dn) * wcs_k)); return; }
      if (MIXED && !((rep_set(jc + k) >> (jr + m)) & 1u)) { atomicAdd(hj + (m * nb + k) * NC, fx_bits((wr_m * P.hist_dn) * wcs_k)); return; }
      if ((rep_set(jc + k) >> (jr + m)) & 1u) {
        const double pr = wr_m * w;
        if (pr > kNegligibleW) {
          const int lm = fine_level(pr);
          atomicAdd(hist_lo + lm * nbins + ((unsigned)nb + hrow + (unsigned)(m * nb + k)), fx_encode(pr, fine_scale(lm)));
          fine_residual(lm, (unsigned)nb + hrow + (unsigned)(m * nb + k), pr, std::false_type{});
        }
      }
    };
    // NORMAL mode, rare branches: a LINEAR end-span weight below kLinFlagW flags the bins it lands in (kLinFlagW)
    auto flag_linear = [&](int k, double w, const double (&wrm)[4]) {
      const bool last = k == 2 && jc == S - 1;
      if (w < kLinFlagW && ((k == 1 && jc == 0) || last)) {
        unsigned bits = 1u << 16;
#pragma unroll
        for (int m = 0; m < 4; m++) bits |= (wrm[m] * w > kNegligibleW) ? (1u << (jr + m)) : 0u;
        atomicOr(lin_flag + (last ? 1 : 0), bits);
      }
    };
    // Rare: the TARGET sample sits next to a knot (also: clamped saturated, black, integer-position samples) -- its
    // small weights, and their products with the reference weights, go to the fine level of their own exponent --
    // or the REFERENCE sample does with NON-ZERO tiny weights (a saturated reference pixel: I0 = 255 -> 254.999 has
    // outer weights of 1e-13 / 1e-8; exactly zero weights -- a reference sample ON a knot, or out of frame at the
    // initial pose -- lose nothing): then every product below 2^-8 goes to its own level.  (Why the reference side:
    // a joint bin can consist of one end-span target sample's 1e-15 weight with an O(1) derivative PLUS hundreds of
    // 1e-16 products of tiny reference weights; lost in the coarse quantum they shifted that bin's W by 2.8 and
    // one cell's Jacobian by 0.7 % -- 3 of 2 500 random cases, tools/random_parity_sweep.py.)
    // k_href leaves the "reference sample next to a knot with non-zero tiny weights" decision in the SIGN of the
    // first stored weight (min(wr[0], wr[3]) < kTinyW and != 0 <=> wr_in[0] < 0): one compare here instead of
    // five instructions per sample; |wr_in[0]| is a free source modifier of the products below
    const bool ref_tiny = wr_in[0] < 0.0;
    if (min(__double2hiint(wcs[0]), __double2hiint(wcs[3])) < tiny_scaled_hi || ref_tiny) {
      double wc[4];
      if (!STRICT && PRESCALED) {
        // FAST: the weights once more, each with an error relative to ITSELF (bspline4_vals_both_ends; the same values
        // for a sample in the left half of its span)
        double wv[4];
        bspline4_vals_both_ends(pcv, jc, S, rtab, wv);
#pragma unroll
        for (int k = 0; k < 4; k++) { wcs[k] = wv[k]; wc[k] = wv[k] * kWcPreInv; }
      } else {
#pragma unroll
        for (int k = 0; k < 4; k++) wc[k] = PRESCALED ? win[k] * kWcPreInv : win[k];
      }
      wr[0] = fabs(wr[0]);
      if (!ref_tiny) {
        // target side only (about one sample in 450 on a smooth image: one wave-round in eight gets here)
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (wc[k] < kFineW) {
            if (!REPAIR && wc[k] > kNegligibleW) {
              const int lv = fine_level(fabs(wc[k]));
              const bool lin = linear_end_small(k, wc[k]);
              lo_add_marginal(lv * nbins + (unsigned)(jc + k), fx_encode(wc[k], fine_scale(lv)));
              if (lin) fine_residual(lv, (unsigned)(jc + k), wc[k], std::true_type{});
#pragma unroll
              for (int m = 0; m < 4; m++) {
                const double pr = wr[m] * wc[k];  // the reference's own product, rounded once like there
                if (pr > kNegligibleW) {
                  const int lm = fine_level(fabs(pr));
                  atomicAdd(hist_lo + lm * nbins + ((unsigned)nb + hrow + (unsigned)(m * nb + k)), fx_encode(pr, fine_scale(lm)));
                  if (lin) fine_residual(lm, (unsigned)nb + hrow + (unsigned)(m * nb + k), pr, std::false_type{});
                }
              }
              flag_linear(k, wc[k], wr);
            }
          } else {
            add_c(k, wcs[k], wc[k]);
#pragma unroll
            for (int m = 0; m < 4; m++) add_j(m, k, wr[m], wcs[k], wc[k]);
          }
        }
        return;
      }
      // saturated reference pixel: every product below 2^-8 to its own level, whichever factor is small
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const bool small_c = wc[k] < kFineW;
        if (small_c) {
          if (!REPAIR && wc[k] > kNegligibleW) {
            const int lv = fine_level(fabs(wc[k]));
            lo_add_marginal(lv * nbins + (unsigned)(jc + k), fx_encode(wc[k], fine_scale(lv)));
            if (linear_end_small(k, wc[k])) fine_residual(lv, (unsigned)(jc + k), wc[k], std::true_type{});
            flag_linear(k, wc[k], wr);
          }
        } else {
          add_c(k, wcs[k], wc[k]);
        }
#pragma unroll
        for (int m = 0; m < 4; m++) {
          if (small_c || wr[m] < kFineW) {
            const double pr = wr[m] * wc[k];
            if (!REPAIR && pr > kNegligibleW) {
              const int lm = fine_level(fabs(pr));
              atomicAdd(hist_lo + lm * nbins + ((unsigned)nb + hrow + (unsigned)(m * nb + k)), fx_encode(pr, fine_scale(lm)));
              if (small_c && linear_end_small(k, wc[k])) fine_residual(lm, (unsigned)nb + hrow + (unsigned)(m * nb + k), pr, std::false_type{});
            }
          } else {
            add_j(m, k, wr[m], wcs[k], wc[k]);
          }
        }
      }
      return;
    }
    if constexpr (REPAIR) {  // an ordinary sample: all twenty addends went to the coarse copies
#pragma unroll
      for (int k = 0; k < 4; k++) {
        if (rep_set(jc + k) == 0u) continue;
        const double w = wcs[k] * kWcPreInv;
        add_c(k, wcs[k], w);
#pragma unroll
        for (int m = 0; m < 4; m++) add_j(m, k, wr[m], wcs[k], w);
      }
      return;
    }
    if constexpr (MIXED) {
      if ((rep_col1 | rep_colz) != 0u) {  // an ordinary sample: a column in the repair set takes its addends one by one
#pragma unroll
        for (int k = 0; k < 4; k++) {
          const double mk = wcs[k] * P.hist_dn;       // (the coarse forms of the ordinary path, below)
          const unsigned rs = rep_set(jc + k);
          const double w = wcs[k] * kWcPreInv;        // (the fine forms of REPAIR mode, above)
          if ((rs >> 16) & 1u) {
            if (w > kNegligibleW) {
              const int lv = fine_level(w);
              atomicAdd(hist_lo + lv * nbins + (unsigned)(jc + k), fx_encode(w, fine_scale(lv)));
              fine_residual(lv, (unsigned)(jc + k), w, std::false_type{});
            }
          } else {
            atomicAdd(hc + k * NC, fx_bits(mk));
          }
#pragma unroll
          for (int m = 0; m < 4; m++) {
            if ((rs >> (jr + m)) & 1u) {
              const double pr = wr[m] * w;
              if (pr > kNegligibleW) {
                const int lm = fine_level(pr);
                atomicAdd(hist_lo + lm * nbins + ((unsigned)nb + hrow + (unsigned)(m * nb + k)), fx_encode(pr, fine_scale(lm)));
                fine_residual(lm, (unsigned)nb + hrow + (unsigned)(m * nb + k), pr, std::false_type{});
              }
            } else {
              atomicAdd(hj + (m * nb + k) * NC, fx_bits(mk * wr[m]));
            }
          }
        }
        return;
      }
    }
    // The marginal addend of column k, wcs[k] * hist_dn = wc * 2^(s - 1074), is a subnormal double whose bit pattern IS
    // the integer I_k = RN(wc * 2^s) (fx_bits).  The joint addends of that column are taken from it: I_k * wr[m], again a
    // subnormal product, i.e. RN(I_k * wr[m]) as an integer -- within ONE quantum (2^-s) of wr * wc * 2^s instead of half
    // a quantum (I_k is off by at most 1/2, wr <= 1), and the four multiplications wr[m] * hist_dn per sample are gone
    // (round 5: 24 -> 20 multiplications for the 20 adds).  The coarse copies promise an ABSOLUTE quantum either way;
    // what needs more goes through the fine levels / the repair (above).  f64 subnormal OPERANDS cost nothing and are
    // not flushed (tools/ubench/denorm_encode.hip).
    double marg[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      marg[k] = wcs[k] * P.hist_dn;
      atomicAdd(hc + k * NC, fx_bits(marg[k]));
    }
#pragma unroll
    for (int m = 0; m < 4; m++)
#pragma unroll
      for (int k = 0; k < 4; k++)
        atomicAdd(hj + (m * nb + k) * NC, fx_bits(marg[k] * wr[m]));  // wr[0] >= 0 here (ref_tiny is false)
  };
  auto dump_pixel = [&](int s, const PixelFront &f, double ic, int jc, const double (&wc)[4]) {
    const int c = g.cell_begin + cl * g.cell_stride;
    const int r = (c / g.cell_num) * g.rb + s / g.cb;
    const int col = (c % g.cell_num) * g.cb + s % g.cb;
    const size_t id = (size_t)r * g.cols + col;
    P.dbg_u[id] = f.u; P.dbg_v[id] = f.v; P.dbg_ic[id] = ic; P.dbg_jc[id] = jc;
#pragma unroll
    for (int k = 0; k < 4; k++) P.dbg_wc[4 * id + k] = wc[k];
  };
  // FAST math runs every pixel loop as TWO passes.  The main pass does the samples FAST arithmetic may decide
  // (see exact_decisions) and records, per round, whether the wave met any other ("rare") sample; the second
  // pass revisits only those rounds and only those samples, with the reference's arithmetic.  The code of the
  // rare path thus sits outside the hot loop (its registers, its loads and its IEEE divisions cost the main
  // pass nothing: ten compares and a ballot), the lane -> sample assignment is the same in both passes (so
  // the sums stay run-to-run reproducible), and integer histogram adds do not care about the order.
  // rounds: bit min(r, 63) of a wave-uniform mask; a cell of more than 64 rounds per wave shares the last bit.
  unsigned long long rare_rounds = 0ull;

  // Loop form, cost + Jacobian: bit r of a lane's gomask = "the main pass of the cost phase took this lane's sample of
  // round r".  The Jacobian phase redoes the identical warp and sample, so its main pass takes exactly those samples
  // (minus the ones outside linearizeOplus' narrower frame) and revisits exactly the cost phase's rare rounds: five f64
  // compares, the clamp guard, the ballot and the mask bookkeeping per round are replaced by one bit test.
  // 32 rounds per wave cover cells of up to 32 * NT pixels; for larger cells the host launches the BIG instantiation,
  // which classifies again (jac_round).  (A run-time switch between the two loops costs spills around the cold one,
  // and a kernel with scratch measured 4 % slower: profiles/r02_ablations_A.txt.)
  unsigned gomask = 0u;
  (void)gomask;
  // raremask, likewise: bit r = "this lane's sample of round r is rare"; the wave's rare rounds are the OR over the
  // lanes, taken once after the loop (instead of a ballot and six scalar instructions per round).
  unsigned raremask = 0u;
  (void)raremask;
  constexpr bool use_lane_masks = !STRICT && LAT == 0 && !DBG && !BIG;
  constexpr bool use_gomask = JAC && use_lane_masks;
  LatPix lat[LAT > 0 ? LAT : 1];  // LAT + JAC: the cost phase's hand-over to the Jacobian phase
  (void)lat;
  // STRICT: the cost phase's pixel loop (REPAIR: once more, see kLinFlagW)
  auto strict_cost_loop = [&](auto repair_tag) {
    constexpr bool REPAIR = decltype(repair_tag)::value;
#pragma clang loop unroll(disable)
    for (int sb = wave_base; sb < g.pstride; sb += NT) {
      const int s = sb + lane;
      TileIn tin;
      PixelFront f;
      load_tile(P, base + (unsigned)s, plane, tin);
      pixel_front<true>(P, SA, tin, f);
      double ic = NAN, wc[4] = {NAN, NAN, NAN, NAN}, dw[4];
      int jc = -1;
      if (f.in) {
        // pixel_sample<true>'s arithmetic on the sample's 2x2 cell (round 6: no 4x4 window, no run-time tap selection; the
        // cost phase needs the centre sample only)
        ic = bilinear_cell_exact(P, f.u, f.v);
        if (ic >= 255) ic = 254.999;  // types_six_dof_expmap.cpp:572-575
        if (ic < 0) ic = 0.0;
        const double pc = div_255(ic * ((double)nb - 3.0));
        jc = (int)floor(pc);
        bspline4_tab<false>(pc, jc, S, rtab, wc, dw);
        hist_add(f.jr, jc, f.wr, wc, std::false_type{}, 0.0, 0, repair_tag);
      }
      if (!REPAIR && DBG && P.dbg_u && !P.dbg_jac && pose_idx == 0 && f.jr >= 0) dump_pixel(s, f, ic, jc, wc);
    }
  };
  // FAST: one round of the cost phase's pixel loops (main or second pass; REPAIR: once more, see kLinFlagW).
  // Main pass of the loop form: the NEXT round's point and bin index (7 registers) are fetched while this round is
  // worked on, and this round's reference weights are fetched behind its window loads: a round then exposes ONE
  // memory round trip (the window) instead of two.  Both phases; the Jacobian phase ends at 95 of its 96 VGPRs.
  // Measured: 1030 -> 1008 us per 256-pose launch (profiles/r02_ablations_A.txt).
  auto cost_round = [&](int sb, int r, const TileIn &cur, TileIn &nxt, auto second_pass) -> bool {
    if constexpr (STRICT) {
      return false;
    } else {
      constexpr bool SECOND = decltype(second_pass)::value;
      const int s = sb + lane;
      TileIn tin;
      PixelFront f;
      if (!SECOND) {
        tin = cur;
#pragma unroll
        for (int k = 0; k < 4; k++) tin.wr[k] = 0.0;
      } else if constexpr (use_lane_masks) {
        // (the loop below has requested this round's point while the previous rare round was worked on: second_pass_rounds)
        tin = cur;
        load_tile_w(P, base + (unsigned)s, plane, tin);
      } else {
        load_tile(P, base + (unsigned)s, plane, tin);
      }
      double ic = NAN;
      bool rare, go;
      if constexpr (SECOND && use_lane_masks) {
        // the main pass has left every lane's rare samples in raremask: no FAST front, no window, no sample -- straight
        // to the exact decisions of the lanes that need them
        f.jr = tin.jr;
#pragma unroll
        for (int k = 0; k < 4; k++) f.wr[k] = tin.wr[k];
        f.in = false; f.jin = false; f.redo = false;
        rare = ((raremask >> r) & 1u) != 0u;
        go = false;
      } else {
      // the next round's point is requested BEHIND this round's warp: x, y, z are dead by then, so the loads land in
      // the registers they come from and nothing is copied from round to round (pixel_front's empty asm keeps the
      // scheduler from hoisting the loads above the warp again)
      pixel_front<false, !SECOND>(P, SA, tin, f);
      if (!SECOND && sb + NT < g.pstride) load_tile_xyz(P, base + (unsigned)(s + NT), nxt);
      WinC wc2;
      load_win_centre_e(P, f.e0, wc2);
      if (!SECOND) load_tile_w(P, base + (unsigned)s, plane, tin);
      // fixed-tap sample (a convex combination of u8 taps: never negative), computed for every lane -- lanes
      // without a sample hold a harmless window
      ic = sample_fast_c(wc2, f.u, f.v);
      rare = f.in && outside_clamp_guard(ic);
      if (f.redo && classify_redo(P, f)) rare = true;  // (waves that meet the frame border)
      go = f.in && !rare;
      }
      if (use_lane_masks && !SECOND) {  // r < 32
        const unsigned bit = 1u << r;
        raremask |= rare ? bit : 0u;
        if (use_gomask) gomask |= go ? bit : 0u;  // for the Jacobian phase (jac_round_masked)
      }
      if (SECOND) {
        go = false;
#ifdef NID_ABL_COST2_NO_EXACT  // (ablation builds only: the second pass's loop, loads and mask tests alone)
        rare = false;
#endif
        if (rare) {
          exact_decisions<EXT, false>(P, SA, pose_idx, tin, f, ic);
          go = f.in;
        }
      }
      double wc[4] = {NAN, NAN, NAN, NAN}, dw[4];
      int jc = -1;
      if (go) {
        double pc;
        jc = fast_bin<SECOND || kMainPassClamps, SECOND>(ic, S, pc);
        bspline4_poly<false, JAC && !SECOND, !SECOND>(pc, jc, rtab, wc, dw);
        // (second pass: `ic` is the reference's own sample, evaluated exactly and clamped like there)
#ifdef NID_ABL_COST2_NO_HIST  // (ablation builds only: the second pass without its histogram updates -- wrong results)
        if (!SECOND)
#endif
        hist_add(f.jr, jc, SECOND ? f.wr : tin.wr, wc, std::true_type{}, pc,
                 (SECOND && NID_CLAMP_BINS) ? (ic == 254.999 ? 1 : ((NID_NEAR_SAT_BINS && ic == kNearSatIc) ? 2 : 0)) : 0, std::false_type{});
      } else {
        ic = NAN;
      }
      if (DBG && P.dbg_u && !P.dbg_jac && pose_idx == 0 && f.jr >= 0 && rare == SECOND) {
#pragma unroll
        for (int k = 0; k < 4; k++) wc[k] *= kWcPreInv;  // NaN stays NaN
        dump_pixel(s, f, ic, jc, wc);
      }
      if (use_lane_masks) return false;
      return !SECOND && __builtin_amdgcn_ballot_w64(rare) != 0ull;
    }
  };
  // FAST, REPAIR (see kLinFlagW; rare, so ONE pass and no prefetch): every sample once more with the arithmetic of the
  // pass that took it -- the same classification on the same values -- and hist_add in REPAIR mode
  auto repair_round = [&](int sb) {
    if constexpr (!STRICT) {
      TileIn tin;
      PixelFront f;
      load_tile(P, base + (unsigned)(sb + lane), plane, tin);
      pixel_front<false>(P, SA, tin, f);
      WinC wc2;
      load_win_centre_e(P, f.e0, wc2);
      double ic = sample_fast_c(wc2, f.u, f.v);
      bool rare = f.in && outside_clamp_guard(ic);
      if (f.redo && classify_redo(P, f)) rare = true;
      bool go = f.in && !rare;
      if (rare) {
        exact_decisions<EXT, false>(P, SA, pose_idx, tin, f, ic);
        go = f.in;
      }
      if (go) {
        double pc, wc[4], dw[4];
        int jc;
        if (rare) jc = fast_bin<true, true>(ic, S, pc);
        else jc = fast_bin<kMainPassClamps, false>(ic, S, pc);
        bspline4_poly<false, false>(pc, jc, rtab, wc, dw);
        hist_add(f.jr, jc, tin.wr, wc, std::true_type{}, pc,
                 (rare && NID_CLAMP_BINS) ? (ic == 254.999 ? 1 : ((NID_NEAR_SAT_BINS && ic == kNearSatIc) ? 2 : 0)) : 0, std::true_type{});
      }
    }
  };
  // the rounds of `todo` (a wave-uniform bit mask, bit r = round r), each with its point (x, y, z, jr) requested one round ahead
  auto second_pass_rounds = [&](unsigned todo, auto &&body) {
    int rr = __builtin_ctz(todo);
    TileIn nx;
    load_tile_xyz(P, base + (unsigned)(wave_base + rr * NT + lane), nx);
#pragma clang loop unroll(disable)
    for (;;) {
      todo &= todo - 1u;
      const int rn = todo ? __builtin_ctz(todo) : -1;
      const TileIn cur = nx;
      if (rn >= 0) load_tile_xyz(P, base + (unsigned)(wave_base + rn * NT + lane), nx);
      body(wave_base + rr * NT, rr, cur);
      if (rn < 0) break;
      rr = rn;
    }
  };
  if constexpr (STRICT) {
    strict_cost_loop(std::false_type{});
  } else {
    int r = 0;
    if constexpr (LAT > 0) {
      // staged main pass (see the template comment); the host launches this form only if LAT rounds cover the cell
      TileIn (&tin)[kLatN] = lat_tin;
      PixelFront (&fr)[kLatN] = lat_fr;
      WinC (&wcn)[kLatN] = lat_wc;
      WinJ (&wjn)[kLatN] = lat_wj;
      // RES: the cell's tile entries stay in LDS from the kernel's first request on (every thread re-reads what it
      // wrote itself: [round][field][thread] doubles, then the bin indices) -- an LDS read instead of an L2 round trip
      // at the head of every request
      double *tcache = reinterpret_cast<double *>((reinterpret_cast<uintptr_t>(lds_tail) + 15) & ~(uintptr_t)15);
      int *jcache = reinterpret_cast<int *>(tcache + LAT * 7 * NT);
      if constexpr (!kLatEarly) {
#pragma unroll
      for (int q = 0; q < LAT; q++)
        if (wave_base + q * NT < g.pstride) {
          if (RES && !rc.fresh) {
            const double *tc = tcache + q * 7 * NT + tid;
            tin[q].x = tc[0]; tin[q].y = tc[NT]; tin[q].z = tc[2 * NT];
#pragma unroll
            for (int k = 0; k < 4; k++) tin[q].wr[k] = tc[(3 + k) * NT];
            tin[q].jr = jcache[q * NT + tid];
          } else {
            load_tile(P, base + (unsigned)(wave_base + q * NT + lane), plane, tin[q]);
            if (RES) {
              double *tc = tcache + q * 7 * NT + tid;
              tc[0] = tin[q].x; tc[NT] = tin[q].y; tc[2 * NT] = tin[q].z;
#pragma unroll
              for (int k = 0; k < 4; k++) tc[(3 + k) * NT] = tin[q].wr[k];
              jcache[q * NT + tid] = tin[q].jr;
            }
          }
        }
#pragma unroll
      for (int q = 0; q < LAT; q++)
        if (wave_base + q * NT < g.pstride) {
          pixel_front<false>(P, SA, tin[q], fr[q]);
          if (JAC) load_win_jac_e(P, fr[q].e0, wjn[q]);
          else load_win_centre_e(P, fr[q].e0, wcn[q]);
        }
      }  // (!kLatEarly)
#pragma unroll
      for (int q = 0; q < LAT; q++) {
        lat[q].go = false;
        if (wave_base + q * NT < g.pstride) {
          PixelFront &f = fr[q];
          double ic, gx = 0.0, gy = 0.0;
          if (JAC) gradient_fast_j(wjn[q], f.u, f.v, gx, gy, ic);  // its centre sample IS sample_fast_c's
          else ic = sample_fast_c(wcn[q], f.u, f.v);
          bool rare = f.in && outside_clamp_guard(ic);
          if (f.redo && classify_redo(P, f)) rare = true;
          double wc[4], dw[4], pc = 0.0;
          int jc = -1;
          if (f.in && !rare) {
            jc = fast_bin<kMainPassClamps>(ic, S, pc);
            bspline4_poly<false, false, true>(pc, jc, rtab, wc, dw);
            hist_add(f.jr, jc, f.wr, wc, std::true_type{}, pc, 0, std::false_type{});
          }
          if (__builtin_amdgcn_ballot_w64(rare) != 0ull) rare_rounds |= 1ull << q;
          if (JAC) {
            LatPix &l = lat[q];
            l.x = f.x; l.y = f.y; l.iz = f.zq; l.gx = gx; l.gy = gy; l.pc = pc;
#pragma unroll
            for (int k = 0; k < 4; k++) l.wr[k] = k == 0 ? fabs(f.wr[0]) : f.wr[k];  // sign: k_href's knot flag
            l.jr = f.jr; l.jc = jc;
            l.go = f.jin && !rare;
          }
        }
      }
    } else {
      if (wave_base < g.pstride) load_tile_xyz(P, base + (unsigned)(wave_base + lane), pre);
#pragma clang loop unroll(disable)
      for (int sb = wave_base; sb < g.pstride; sb += NT, r++)
        if (cost_round(sb, r, pre, pre, std::false_type{})) rare_rounds |= 1ull << min(r, 63);
      if (use_lane_masks) rare_rounds = wave_or_u32(raremask);
    }
#ifndef NID_ABL_NO_COST_SECOND  // (ablation builds only, tools/build_variant.py: what the pass costs -- wrong results)
    if (rare_rounds != 0ull) {
      if constexpr (use_lane_masks) {
#ifdef NID_CENSUS  // (census builds only, tools/rare_census.py: how full the second pass's rounds are)
        {
          int c = __builtin_popcount(raremask);
          for (int o = 32; o; o >>= 1) c += __shfl_xor(c, o);
          const int R = __builtin_popcount((unsigned)rare_rounds);
          if (lane == 0) {
            atomicAdd(&g_census[0], 1ull);                                   // waves with a second pass
            atomicAdd(&g_census[1], (unsigned long long)R);                  // rounds they run now
            atomicAdd(&g_census[2], (unsigned long long)((c + 63) / 64));    // rounds of the same samples packed per wave
            atomicAdd(&g_census[3], (unsigned long long)c);                  // rare samples
            atomicAdd(&g_census[8 + min(R, 31)], 1ull);                      // histogram of R
            atomicAdd(&g_census[40 + min((c + 63) / 64, 15)], 1ull);         // histogram of the packed count
          }
        }
#endif
        // The second pass is a chain of DEPENDENT round trips per round -- tile entry -> exact projection -> target cell ->
        // histogram -- and on flash data every workgroup of a CU sits in it at the same time (the poses of one cell are
        // dispatched next to each other: nothing else to run meanwhile).  Round 6: the NEXT rare round's point is requested
        // before this one is worked on (the rounds are known: rare_rounds), like the main pass does -- one exposed round
        // trip less per round (profiles/r06_ablations_A.txt: the pass is latency-bound, instruction diets measure +-0).
        second_pass_rounds((unsigned)rare_rounds, [&](int sb, int rr, const TileIn &cur) { cost_round(sb, rr, cur, pre, std::true_type{}); });
      } else {
        r = 0;
#pragma clang loop unroll(disable)
        for (int sb = wave_base; sb < g.pstride; sb += NT, r++)
          if ((rare_rounds >> min(r, 63)) & 1ull) cost_round(sb, r, pre, pre, std::true_type{});
      }
    }
#endif
  }
  NID_STAMP(2);
  __syncthreads();

  // ---- fold the copies, probabilities, entropies, weight tables ----------------------
  // clamped samples first (flash data only: the flag is workgroup-uniform): their sums per reference bin and their count
  uint4 flagw = *reinterpret_cast<const uint4 *>(clamp_flag);  // (one LDS read: clamped | near-saturated | lin_flag[2])
  const bool any_sat = !STRICT && NID_CLAMP_BINS && (flagw.x | flagw.y) != 0u;
  const bool any_clamped = any_sat && flagw.x != 0u, any_ns = any_sat && flagw.y != 0u;
  double cw[4] = {0.0, 0.0, 0.0, 0.0};  // their four target weights, as the sample path computes them
  double nw[4] = {0.0, 0.0, 0.0, 0.0};  // ... and the near-saturated samples'
  int jc_cl = 0, jc_ns = 0;
  unsigned grp_colz = 0u;  // grp_flag (wave-uniform)
  if (any_sat) {
    double ic_cl = 254.999, pc_cl;
    jc_cl = fast_bin<false, true>(ic_cl, S, pc_cl);  // (as the second pass places a sample)
    bspline4_vals_both_ends(pc_cl, jc_cl, S, rtab, cw);
    double ic_ns = kNearSatIc, pc_ns;
    jc_ns = fast_bin<false, true>(ic_ns, S, pc_ns);
    bspline4_vals_both_ends(pc_ns, jc_ns, S, rtab, nw);
#pragma unroll
    for (int k = 0; k < 4; k++) { cw[k] *= kWcPreInv; nw[k] *= kWcPreInv; }  // rtab's value polynomials carry kWcPre
    // (both groups, 2 * (nb + 1) sums; a group without samples holds zeros.  The near-saturated bins may sit in the
    // weight tables' area: the barrier below separates these reads from the fold's table writes)
    for (int e2 = tid; e2 < 2 * (nb + 1); e2 += NT) {
      const bool ns = e2 > nb;
      const int e = ns ? e2 - (nb + 1) : e2;
      const unsigned long long *gb = ns ? nsb : clampb, *gl = ns ? ns_lo : clamp_lo;
      const uint4 *hv = reinterpret_cast<const uint4 *>(gb + (size_t)e * kClampCopies);
      unsigned long long acc_lo = 0;
      unsigned acc_hi = 0;
#pragma unroll
      for (int c = 0; c < kClampCopies / 2; c++) {
        const uint4 q = hv[c];
        acc_lo += q.x;
        acc_lo += q.z;
        acc_hi += q.y + q.w;
      }
      double sum = (double)(long long)(acc_lo + ((unsigned long long)acc_hi << 32)) * P.hist_inv_scale;
#pragma unroll
      for (int lv = 0; lv < kFineLevels; lv++)
        sum = fma((double)(long long)gl[lv * kClampBins(nb) + e], fine_inv_scale(lv), sum);
      (ns ? rns : rclamp)[e] = sum;
      // the groups' LINEAR end-span weight (last span, column nb - 2; the near-saturated samples': 3 ulps of S) flags the
      // bins it lands in like a sample of the rare branches does (flag_linear in hist_add; e == nb: the marginal bin)
      const double wlin = ns ? nw[2] : cw[2];
      if (sum != 0.0 && wlin < kLinFlagW && wlin > kNegligibleW && (ns ? jc_ns : jc_cl) == S - 1)
        atomicOr(grp_flag, e < nb ? (1u << e) : (1u << 16));
    }
    __syncthreads();
    flagw = *reinterpret_cast<const uint4 *>(clamp_flag);
    grp_colz = (unsigned)__builtin_amdgcn_readfirstlane((int)*grp_flag);
  }
  const unsigned lin_col1 = (unsigned)__builtin_amdgcn_readfirstlane((int)flagw.z);
  const unsigned lin_colz = (unsigned)__builtin_amdgcn_readfirstlane((int)flagw.w);
  // One bin: its mass from the copies (COARSE = false: without them, after a repair), the fine levels and the two
  // groups; p = mass / N_c, W = -(1 + log2 p), p log2 p into the tables.  Returns the copies' integer sum.
  auto fold_bin = [&](int b, auto coarse_tag, double &mass_out) -> unsigned long long {
    constexpr bool COARSE = decltype(coarse_tag)::value;
    unsigned long long acc = 0ull;
    if (COARSE) {
      const uint4 *hv = reinterpret_cast<const uint4 *>(hist + (size_t)b * NC);
      // low and high dwords summed apart: 32-bit adds for the high parts, one carry chain for the low parts, joined
      // once (the same integer as the 64-bit sum; a whole cell's bin stays below 2^63, see fx_bits)
      unsigned long long acc_lo = 0;
      unsigned acc_hi = 0;
#pragma unroll
      for (int c = 0; c < NC / 2; c++) {
        const uint4 q = hv[((NC / 2) & (NC / 2 - 1)) == 0 ? ((c + b) & (NC / 2 - 1)) : ((c + b) % (NC / 2))];
        acc_lo += q.x;
        acc_lo += q.z;
        acc_hi += q.y + q.w;
      }
      acc = acc_lo + ((unsigned long long)acc_hi << 32);
    }
    double mass = (double)(long long)acc * P.hist_inv_scale;
#pragma unroll
    for (int lv = 0; lv < kFineLevels; lv++) {  // small target weights (kTinyW); all zero leaves `mass` bit for bit
      const long long lo = (long long)hist_lo[lv * nbins + b];  // (0 * scale + mass = mass: no test needed)
      mass = fma((double)lo, fine_inv_scale(lv), mass);         // a power-of-two scale: the product is exact, one rounding
    }
    if (any_clamped) {
      const int col = b < nb ? b : (b - nb) % nb;
      const int k = col - jc_cl;
      if (k >= 0 && k < 4) {
        const double ck = k == 0 ? cw[0] : (k == 1 ? cw[1] : (k == 2 ? cw[2] : cw[3]));
        mass = fma(ck, b < nb ? rclamp[nb] : rclamp[(b - nb) / nb], mass);
      }
    }
    if (any_ns) {
      const int col = b < nb ? b : (b - nb) % nb;
      const int k = col - jc_ns;
      if (k >= 0 && k < 4) {
        const double nk = k == 0 ? nw[0] : (k == 1 ? nw[1] : (k == 2 ? nw[2] : nw[3]));
        mass = fma(nk, b < nb ? rns[nb] : rns[(b - nb) / nb], mass);
      }
    }
    const double p = mass / (double)n_c;  // Q1: N_c of the initial pose
    double w = 0.0, pl = 0.0;
    if (!(p < kSigma)) {
      const double l = STRICT ? log2(p) : log2_fast(p);
      w = -(1.0 + l);  // Q9
      pl = p * l;
    }
    tab[b] = w;
    term[b] = pl;
    mass_out = mass;
    return acc;
  };
  // which bit of lin_flag / repair_set a bin is, and in which of the two words (-1: not in a linear-weight column)
  auto lin_word = [&](int b, int &bit) -> int {
    const int col = b < nb ? b : (b - nb) % nb;
    bit = b < nb ? 16 : (b - nb) / nb;
    return col == 1 ? 0 : (col == nb - 2 ? 1 : -1);
  };
  double ent[2] = {0.0, 0.0};
  for (int b = tid; b < nbins; b += NT) {
    double mass;
    const unsigned long long acc = fold_bin(b, std::true_type{}, mass);
    if (__builtin_expect((lin_col1 | lin_colz | grp_colz) != 0u, 0)) {  // (workgroup-uniform, rare: some bin received a tiny linear end-span weight)
      // A flagged bin with less than kRepairMass is repaired -- if the COLUMN has coarse addends at all: a joint product
      // may have rounded to zero in the copies (1e-8 * 1e-9 against the 2^-52 quantum), a marginal weight of a coarse
      // sample never does (>= kTinyW), so the marginal bin's copies say whether any sample went that way (bit 31)
      int bit;
      const int wsel = lin_word(b, bit);
      if (wsel >= 0) {
        // flagged by a sample of the rare branches: kRepairMass.  Flagged by the groups only: their reference weights
        // in this row are known (the folded sums), and what the bin's W is multiplied with is 3 x that sum -- a row the
        // group barely touches tolerates a coarser mass: kRepairRel x the sum, at most kRepairMass
        double thr = 0.0;
        if (((wsel == 0 ? lin_col1 : lin_colz) >> bit) & 1u) thr = kRepairMass;
        else if (wsel == 1 && ((grp_colz >> bit) & 1u)) {
          const int e = bit == 16 ? nb : bit;
          const double wsum = (any_ns ? rns[e] : 0.0) + ((any_clamped && cw[2] < kLinFlagW) ? rclamp[e] : 0.0);
          thr = fmin(kRepairMass, kRepairRel * wsum);
        }
        unsigned bits = mass < thr ? (1u << bit) : 0u;
        if (b < nb && acc != 0ull) bits |= 1u << 31;
        if (bits) atomicOr(repair_set + wsel, bits);
      }
    }
  }
  __syncthreads();
  // !REPAIR_INLINE (the loop-form kernels): NOTHING happens here -- not even a branch: a conditional exit at this point
  // cost the throughput kernel 8 % (profiles/r04_ablations_A.txt).  The cell goes through its Jacobian phase with the
  // tables as they are and its TAIL looks at repair_set (deferred_to_repair): a cell with a bin to repair publishes
  // nothing and is queued for k_repair.
  auto repair_wanted = [&]() -> bool {  // (the fold's verdict, from LDS: wave-uniform)
    if constexpr (REPAIR_INLINE) {
      return false;
    } else {
      const uint2 rs = *reinterpret_cast<const uint2 *>(repair_set);
      unsigned a = (unsigned)__builtin_amdgcn_readfirstlane((int)rs.x), b = (unsigned)__builtin_amdgcn_readfirstlane((int)rs.y);
      a = (a >> 31) ? (a & 0x1FFFFu) : 0u;
      b = (b >> 31) ? (b & 0x1FFFFu) : 0u;
      return (a | b) != 0u;
    }
  };
  // what the waves that leave a tail early return: "this cell wants the repair pass and publishes nothing" (RES without
  // the inline repair: no such kernel is built since round 6's removal of the resident batch evaluator)
  auto done = [&]() -> bool {
    if constexpr (RES && !REPAIR_INLINE) return repair_wanted();
    else return false;
  };
  auto deferred_to_repair = [&]() -> bool {  // wave 0, in a tail
    if constexpr (REPAIR_INLINE) {
      return false;
    } else {
      if (__builtin_expect(!repair_wanted(), 1)) return false;
      if constexpr (RES) return true;  // (no queue, no k_repair behind a resident kernel: its caller redoes the cell)
      if (tid == 0) {
        unsigned *q = reread_args().repair_queue;  // (loaded here, not at the head of the kernel: see reread_args)
        const unsigned i = __hip_atomic_fetch_add(q, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (a launch pushes at most one entry per workgroup and k_repair empties the queue behind it: the bound only
        // holds against a queue that was never drained)
        if (i < (unsigned)P.g.nloc * (unsigned)kMaxBatchExt) {  // the entry: which cell and pose, and the repair set the fold found
          const uint2 rs = *reinterpret_cast<const uint2 *>(repair_set);
          q[2 + 3 * i] = ((unsigned)pose_idx << 16) | (unsigned)cl;
          q[3 + 3 * i] = rs.x;
          q[4 + 3 * i] = rs.y;
        }
      }
      return true;
    }
  };
  if constexpr (REPAIR_INLINE && !PRESET)
  if (__builtin_expect((lin_col1 | lin_colz | grp_colz) != 0u, 0)) {
    const uint2 rs = *reinterpret_cast<const uint2 *>(repair_set);
    rep_col1 = (unsigned)__builtin_amdgcn_readfirstlane((int)rs.x);
    rep_colz = (unsigned)__builtin_amdgcn_readfirstlane((int)rs.y);
    rep_col1 = (rep_col1 >> 31) ? (rep_col1 & 0x1FFFFu) : 0u;
    rep_colz = (rep_colz >> 31) ? (rep_colz & 0x1FFFFu) : 0u;
    if (__builtin_expect((rep_col1 | rep_colz) != 0u, 0)) {
      // REPAIR (see kLinFlagW): the pixel loops once more; what they sent to the coarse copies of a bin in the repair
      // set goes to its fine levels now, and the bin is folded again without the copies.  Same lane -> sample
      // assignment, integer adds: run-to-run reproducible like everything else.
      if (tid == 0 && P.repair_count) atomicAdd(P.repair_count, 1ull);  // (diagnostics: nid_debug_repair_count)
      repaired = true;
      if constexpr (STRICT) {
        strict_cost_loop(std::true_type{});
      } else {
#pragma clang loop unroll(disable)
        for (int sb = wave_base; sb < g.pstride; sb += NT) repair_round(sb);
      }
      __syncthreads();
      for (int b = tid; b < nbins; b += NT) {
        int bit;
        const int wsel = lin_word(b, bit);
        if (wsel >= 0 && (((wsel == 0 ? rep_col1 : rep_colz) >> bit) & 1u)) {
          double mass;
          fold_bin(b, std::false_type{}, mass);
        }
      }
      __syncthreads();
    }
  }
  // Entropies: every wave sums the bins' terms itself, in ONE order whatever the workgroup shape is (lane l takes
  // the joint bins l, l + 64, ...; then the DPP tree), so that Hc, Hj, err and chi2 of a pose are the same bits
  // for 128-, 256-, 512- and 1024-thread workgroups: cost-only launches may pick their shape by batch size.
  {
    const int lane_ = tid & 63;
    double e0 = (lane_ < nb) ? term[lane_] : 0.0, e1 = 0.0;
    for (int j = lane_; j < nb * nb; j += 64) e1 += term[nb + j];
    ent[0] = wave_lane63(wave_sum_to_lane63(e0));
    ent[1] = wave_lane63(wave_sum_to_lane63(e1));
  }
  NID_STAMP(3, ent[0], ent[1]);
  // block_sum leaves the same value in every lane: keep the per-cell scalars in SGPRs
  const double Hc = wave_uniform(0.0 - ent[0]);
  const double Hj = wave_uniform(0.0 - ent[1]);
  // residual and Huber weights: needed by the cell's tail only, i.e. by wave 0 after the other waves retired
  double err, rho0, rho1;
  auto residual_and_huber = [&]() {
    err = wave_uniform((2 * Hj - href - Hc) / Hj);  // types_six_dof_expmap.h:227
    const double e2 = err * err;                    // Huber, robust_kernel_impl.cpp:77-91 (float dsqr)
    rho0 = e2; rho1 = 1.0;
    if (!(e2 <= P.huber_dsqr)) {
      const double sqrte = sqrt(e2);
      rho0 = 2 * sqrte * P.huber_delta - P.huber_dsqr;
      rho1 = P.huber_delta / sqrte;
    }
    rho0 = wave_uniform(rho0);
    rho1 = wave_uniform(rho1);
  };
  if (!JAC) {
    if (tid >= 64) {  // the cell's tail is wave 0's business: the other waves free their slots now
      if (RES && !kResRezero) zero_histograms(64);  // ... in the resident kernel after clearing the histograms for the next request
      return done();
    }
    if (deferred_to_repair()) return true;
    if (direct_launch) {
      // DIRECT launch (one pose, the host is waiting for it): the cell's record (err, J[6], active) -- or, for the
      // per-cell calls, its outputs -- goes straight to pinned host memory, word by word, and the HOST forms the
      // Huber-weighted quadratic forms (the device's operations in the device's order: IEEE mul / sub / sqrt / div give
      // the same bits) and adds them up in the order sum_blocks_w0 uses.  No ticket, no device-scope round trip, no
      // fence: the launch is over for the host when the last cell's 64 bytes have crossed PCIe (round 2: 5 us of
      // reduction tail behind the last cell).
      err = wave_uniform((2 * Hj - href - Hc) / Hj);  // types_six_dof_expmap.h:227
      if (tid < kCellOut && SA.cellout_host) {
        const double o = tid == 0 ? Hc : (tid == 1 ? Hj : (tid == 2 ? err : (tid == kCellOut - 1 ? (double)n_c : NAN)));
        store_sys(out + tid, o);  // (cost-only: the Jacobian slots hold NaN)
      } else if (!RES && tid < kCellOut && want_cellout) {  // the slot's device block stays valid for callers that gather it (nid_slot_buffers)
        out[tid] = tid == 0 ? Hc : (tid == 1 ? Hj : (tid == 2 ? err : (tid == kCellOut - 1 ? (double)n_c : NAN)));
      }
      if (tid < kDirectRec && SA.host_quad == 1)
        store_sys(SA.quad + (size_t)cl * kDirectRec + tid, tid == 0 ? err : (tid == 1 ? 1.0 : 0.0));
      NID_STAMP(6);
      NID_STAMP(7);
      return false;
    }
    residual_and_huber();
    if (tid == 0 && want_cellout) { out[0] = Hc; out[1] = Hj; out[2] = err; out[kCellOut - 1] = (double)n_c; }
    if (tid < kQuad) store_sc1(quad + tid, tid == 0 ? rho0 : (tid == 28 ? 1.0 : 0.0));
    NID_STAMP(6);
    finish_and_reduce_w0(P, SA, cl, tid);
    NID_STAMP(7);
    return false;
  }

  // DIRECT launch: the residual goes to the host NOW -- while the Jacobian phase runs here, the host works out the
  // cells' Huber weights (a square root and a division each) and their chi2 sum; what is left to it behind the
  // Jacobians' arrival is multiplications and additions (wait_direct in nid_capi.hip)
  // (not for a cell that k_repair will do again: its record comes from there)
  if (direct_launch && SA.host_quad == 1 && tid < kDirectRec && !repair_wanted())
    store_sys(SA.quad + (size_t)cl * kDirectRec + tid, tid == 0 ? (2 * Hj - href - Hc) / Hj : (tid == 1 ? 1.0 : 0.0));

  // ---- phase 2: Jacobian (recompute, see header comment) -------------------------------
  // FAST math: the CONTRACTED weight tables.  A sample needs s = sum_k wr[k] * G[jr + k](pc) and t = G[nb](pc) with
  // G[a](pc) = sum_m W[a][jc + m] * B'_m(pc), and on span jc every B'_m is the quadratic d0 + d1 x + d2 x^2 of the table
  // (x = pc - jc): G[a] is itself a quadratic in x whose three coefficients A_p[a][jc] = sum_m W[a][jc + m] * d_pm[jc]
  // depend on the cell only.  They are formed here, ONCE per cell ((nb + 1) * S entries, a == nb: the marginal table), and
  // a sample evaluates four + one quadratics -- 14 fused multiply-adds and 15 table values instead of 8 (the derivatives)
  // + 24 (the contraction) and 32 (round 5; same algebra as types_six_dof_expmap.cpp:473-519, another association).
  // The table sits at the head of the histogram area, which is dead since the fold; the Jacobian block sum of the
  // throughput shapes reuses the same area behind the loops' barrier.
  double *jtab = reinterpret_cast<double *>(hist);  // [(nb + 1) * S][4]: A0, A1, A2, (pad)
  if constexpr (!STRICT) {
    for (int e = tid; e < (nb + 1) * S; e += NT) {
      const int a = e / S, sp = e - a * S;
      const double *wrow = a < nb ? tab + (nb + a * nb + sp) : tab + sp;
      const double *c = rtab + sp * kCoefRow + 4;
      double A0 = 0.0, A1 = 0.0, A2 = 0.0;
#pragma unroll
      for (int m = 0; m < 4; m++) {
        const double wm = wrow[m];
        A0 = fma(wm, c[7 * m], A0); A1 = fma(wm, c[7 * m + 1], A1); A2 = fma(wm, c[7 * m + 2], A2);
      }
      *reinterpret_cast<double4 *>(jtab + 4 * e) = make_double4(A0, A1, A2, 0.0);
    }
    __syncthreads();
  }
  const double cA = wave_uniform(Hc + href), cB = wave_uniform(Hj);
  double acc[6];
#pragma unroll
  for (int n = 0; n < 6; n++) acc[n] = 0.0;
  // FAST-mode constants: the gradient helper returns twice the gradient, so 1/2 rides on fx, fy
  const double cAx = wave_uniform(cA * (0.5 * g.fx)), cBx = wave_uniform(cB * (0.5 * g.fx));
  const double cAy = wave_uniform(cA * (0.5 * g.fy)), cBy = wave_uniform(cB * (0.5 * g.fy));
  // FAST: one sample's contribution to the six sums from the contracted tables (jtab).  d(u,v)/d(xi) with a = x/z, b = y/z
  // (types_six_dof_expmap.cpp:438-450 regrouped): Ju = fx [-ab, 1+a^2, -b, 1/z, 0, -a/z], Jv = fy [-(1+b^2), ab, a, 0, 1/z, -b/z];
  // with P = c_u gx, Q = c_v gy and R = P a + Q b the six terms are -(R b + Q), R a + P, Q a - P b, P/z, Q/z, -R/z.
  // acc[0] and acc[5] accumulate the NEGATED sums (fixed after the loop).
  auto jac_accumulate_fast = [&](const auto &f, double invz, double gx, double gy, double pc, int jc, auto q5_possible) {
    // (the second passes place a sample with the reference's own rounding and may clamp jc: fast_bin<.., EXACT>)
    const double x = decltype(q5_possible)::value ? pc - (double)jc : frac_nonneg(pc, jc);
    // jr * S + jc in ONE full-rate instruction (left to itself the compiler folds the constant S into a v_mad_u64_u32: quarter rate)
    int ent;
    if constexpr (NB > 0) asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(ent) : "v"(f.jr), "n"(NB - 3), "v"(jc));
    else asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(ent) : "v"(f.jr), "s"(S), "v"(jc));
    const double *A = jtab + 4 * ent;
    double ss = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const double *Ak = A + 4 * S * k;
      ss = fma(f.wr[k], fma(fma(Ak[2], x, Ak[1]), x, Ak[0]), ss);
    }
    const double *T = jtab + 4 * (nb * S + jc);
    const double tt = fma(fma(T[2], x, T[1]), x, T[0]);
    double cP = fma(ss, cAx, -(tt * cBx)), cQ = fma(ss, cAy, -(tt * cBy));
    if (decltype(q5_possible)::value) {
      if (pc == 0.0) { cP = 0.0; cQ = 0.0; }  // Q5: B-spline derivative identically 0 at u == 0 (second passes only, see kMainPassClamps)
    }
    const double Pg = cP * gx, Qg = cQ * gy;
    const double a = f.x * invz, b = f.y * invz;
    const double R = fma(Pg, a, Qg * b);
    acc[0] = fma(R, b, acc[0] + Qg);
    acc[1] = fma(R, a, acc[1] + Pg);
    acc[2] = fma(Qg, a, fma(-Pg, b, acc[2]));
    acc[3] = fma(Pg, invz, acc[3]);
    acc[4] = fma(Qg, invz, acc[4]);
    acc[5] = fma(R, invz, acc[5]);
  };
  // STRICT: the reference's derivative values dw[4] against the weight tables
  auto jac_accumulate = [&](const auto &f, double invz, double gx, double gy, double pc, int jc, const double (&dw)[4], auto q5_possible) {
    (void)pc; (void)q5_possible;
    const double *tj = tab + ((unsigned)nb + (unsigned)(__mul24(f.jr, nb) + jc));
    double tt = 0.0, ss = 0.0;
#pragma unroll
    for (int m = 0; m < 4; m++) tt = fma(tab[jc + m], dw[m], tt);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      double inner = 0.0;
#pragma unroll
      for (int m = 0; m < 4; m++) inner = fma(tj[k * nb + m], dw[m], inner);
      ss = fma(f.wr[k], inner, ss);
    }
    const double c = fma(ss, cA, -(tt * cB));
    const double cgx = c * gx, cgy = c * gy;
    const double x = f.x, y = f.y;
    const double invz_2 = invz * invz;
    // rows of d(u,v)/d(xi), types_six_dof_expmap.cpp:438-450 (omega first, then upsilon)
    const double ju0 = -x * y * invz_2 * g.fx, ju1 = (1 + (x * x * invz_2)) * g.fx;
    const double ju2 = -y * invz * g.fx, ju3 = invz * g.fx, ju5 = -x * invz_2 * g.fx;
    const double jv0 = -(1 + y * y * invz_2) * g.fy, jv1 = x * y * invz_2 * g.fy;
    const double jv2 = x * invz * g.fy, jv4 = invz * g.fy, jv5 = -y * invz_2 * g.fy;
    acc[0] = fma(cgx, ju0, fma(cgy, jv0, acc[0]));
    acc[1] = fma(cgx, ju1, fma(cgy, jv1, acc[1]));
    acc[2] = fma(cgx, ju2, fma(cgy, jv2, acc[2]));
    acc[3] = fma(cgx, ju3, acc[3]);
    acc[4] = fma(cgy, jv4, acc[4]);
    acc[5] = fma(cgx, ju5, fma(cgy, jv5, acc[5]));
  };
  // Jacobian-phase dump (dbg_jac): for every sample that contributes, the image gradient in the reference's
  // convention (central difference / 2), the bin position, the span and the four B-spline derivatives
  auto dump_jac = [&](int s, double gx, double gy, double pc, int jc, const double (&dw)[4]) {
    const int c = g.cell_begin + cl * g.cell_stride;
    const int r = (c / g.cell_num) * g.rb + s / g.cb;
    const int col = (c % g.cell_num) * g.cb + s % g.cb;
    const size_t id = (size_t)r * g.cols + col;
    P.dbg_u[id] = gx; P.dbg_v[id] = gy; P.dbg_ic[id] = pc; P.dbg_jc[id] = jc;
#pragma unroll
    for (int k = 0; k < 4; k++) P.dbg_wc[4 * id + k] = dw[k];
  };
  if constexpr (STRICT) {
#pragma clang loop unroll(disable)
    for (int sb = wave_base; sb < g.pstride; sb += NT) {
      const int s = sb + lane;
      TileIn tin;
      PixelFront f;
      load_tile(P, base + (unsigned)s, plane, tin);
      pixel_front<true>(P, SA, tin, f);
      f.wr[0] = fabs(f.wr[0]);  // the sign of the first reference weight is k_href's knot flag (hist_add)
      load_window(P, f.w);
      if (f.in) {
        // linearizeOplus: fx*(x/z)+cx (types_six_dof_expmap.cpp:407-422, Q6)
        const double rz = rcp_fast(f.zq);  // one reciprocal for the three quotients (div_shared: the IEEE quotients' bits)
        const bool zm = exp_mid(f.zq);
        const double invz = div_shared(1.0, f.zq, rz, zm);
        const double u = g.fx * div_shared(f.x, f.zq, rz, zm) + g.cx;
        const double v = g.fy * div_shared(f.y, f.zq, rz, zm) + g.cy;
        if (u >= 0 && u + 3 <= P.jac_cols && v >= 0 && v + 3 <= g.rows) {
          double ic, wc[4], dw[4];
          const int jc = pixel_sample<true, true>(f, nb, S, rtab, ic, wc, dw);  // from the cost pass's (u, v): Q7
          // The gradient is sampled around linearizeOplus' OWN (u, v) (Q6), which may truncate to the neighbouring
          // pixel when the cost pass's sits within an ulp of an integer (identity-like poses): its taps then lie
          // outside the window loaded around the cost pass's position -- load the one it needs.
          if ((int)u - 1 != f.w.wx || (int)v - 1 != f.w.wy) {
            f.w.wx = (int)u - 1; f.w.wy = (int)v - 1;
            load_window(P, f.w);
          }
          const RowPair r0 = win_rows(f.w, v);
          const double gx = (bilinear_rows(r0, f.w.wx, u + 1) - bilinear_rows(r0, f.w.wx, u - 1)) / 2;
          const RowPair rp = win_rows(f.w, v + 1), rm = win_rows(f.w, v - 1);
          const double gy = (bilinear_rows(rp, f.w.wx, u) - bilinear_rows(rm, f.w.wx, u)) / 2;
          jac_accumulate(f, invz, gx, gy, 1.0, jc, dw, std::false_type{});
          if (DBG && P.dbg_u && P.dbg_jac && pose_idx == 0) dump_jac(s, gx, gy, ic * ((double)nb - 3.0) / 255.0, jc, dw);
        }
      }
    }
  } else {
    // Two passes like the cost phase, with the same classification on the same values (gradient_fast_interior's
    // centre sample IS sample_fast_interior's): both phases take the same decisions and use the same
    // intensity (Q7).
    auto jac_round = [&](int sb, int r, const TileIn &cur, TileIn &nxt, auto second_pass) {
      constexpr bool SECOND = decltype(second_pass)::value;
      const int s = sb + lane;
      TileIn tin;
      PixelFront f;
      if (!SECOND) {
        tin = cur;
#pragma unroll
        for (int k = 0; k < 4; k++) tin.wr[k] = 0.0;
      } else if constexpr (use_lane_masks) {
        tin = cur;  // (requested one rare round ahead: second_pass_rounds)
        load_tile_w(P, base + (unsigned)s, plane, tin);
      } else {
        load_tile(P, base + (unsigned)s, plane, tin);
      }
      if constexpr (SECOND && use_lane_masks && NID_JAC_SECOND_MASK) {
        // the cost phase's rare samples are this phase's (raremask): FAST warp for d(u,v)/d(xi), then straight to the
        // exact decisions -- no window, no gradient, no classification for the lanes that are not rare
        warp_fast(P, SA, tin, f, SA.pose.M[3], SA.pose.M[7], SA.pose.M[11]);
#pragma unroll
        for (int k = 0; k < 4; k++) f.wr[k] = k == 0 ? fabs(tin.wr[0]) : tin.wr[k];
        f.in = false; f.jin = false; f.redo = false;
        if (((raremask >> r) & 1u) != 0u) {
          double ic, gx, gy, dummy;
          exact_decisions<EXT>(P, SA, pose_idx, tin, f, ic);
          if (f.jin) {
            gradient_fast_interior(f.w, f.u, f.v, gx, gy, dummy);
            double pc;
            const int jc = fast_bin<true, true>(ic, S, pc);
            jac_accumulate_fast(f, f.zq, gx, gy, pc, jc, std::true_type{});
          }
        }
        return false;
      }
      // (the next round's point behind this round's warp: see cost_round)
      pixel_front<false, !SECOND>(P, SA, tin, f);
      if (!SECOND && sb + NT < g.pstride) load_tile_xyz(P, base + (unsigned)(s + NT), nxt);
      WinJ wj;
      load_win_jac_e(P, f.e0, wj);
      if (!SECOND) {
        load_tile_w(P, base + (unsigned)s, plane, tin);
#pragma unroll
        for (int k = 0; k < 4; k++) f.wr[k] = tin.wr[k];
      }
      f.wr[0] = fabs(f.wr[0]);  // the sign of the first reference weight is k_href's knot flag (hist_add)
      double ic, gx, gy;
      gradient_fast_j(wj, f.u, f.v, gx, gy, ic);  // every lane, see the cost phase
      bool exact = f.in && outside_clamp_guard(ic);
      if (f.redo && classify_redo(P, f)) exact = true;
      bool go = f.jin && !exact;
      if (SECOND) {
        go = false;
        // A rare sample whose twelve gradient taps are all equal (the inside of a saturated or black patch: most rare
        // samples of a flash pair) has the gradient (0, 0): whatever the exact decisions say, it adds exact zeros to
        // the six sums -- skipped.  (Only if the reference's window is this window: (u, v) away from integer
        // coordinates by more than FAST arithmetic's error, or its truncation could pick the neighbouring pixel.)
        // (... and only if `wj` IS the sample's window: a sample the FAST front does not place in the frame -- within
        // the border band, which the integer range checks make 2^-11 px wide -- has the window of pixel (0, 0) in `wj`;
        // taken for the sample's, a flat image corner silently dropped such samples' contributions: up to 12 % of a
        // border cell's Jacobian in sweep seed 502812.  The f64 tests of rounds 2-3 had the same hole, 2^-20 px wide.)
        if (exact && f.in) {
          const unsigned t0 = wj.r1.x;
          const bool flat = (t0 == wj.r1.y) & (t0 == wj.r2.x) & (t0 == wj.r2.y) & (t0 == wj.c0) & (t0 == wj.c3) & ((t0 >> 16) == (t0 & 0xFFFFu));
          const double fu = f.u - floor(f.u), fv = f.v - floor(f.v);
          if (flat && fu > kBorderEps && fu < 1.0 - kBorderEps && fv > kBorderEps && fv < 1.0 - kBorderEps) exact = false;
        }
        if (exact) {
          exact_decisions<EXT, false>(P, SA, pose_idx, tin, f, ic);  // (u, v), the frame tests and ic from the sample's 2x2 cell
          go = f.jin;
          if (go) {  // the gradient again, on the window around the reference's (u, v) -- which may truncate to the pixel next to FAST's
            double dummy;
            f.w.wx = (int)f.u - 1; f.w.wy = (int)f.v - 1;
            load_window(P, f.w);
            gradient_fast_interior(f.w, f.u, f.v, gx, gy, dummy);
          }
        }
      }
      if (go) {
        double pc;
        const int jc = fast_bin<SECOND || kMainPassClamps, SECOND>(ic, S, pc);
        jac_accumulate_fast(f, f.zq, gx, gy, pc, jc, std::integral_constant<bool, SECOND || kMainPassClamps>{});
        if (DBG && P.dbg_u && P.dbg_jac && pose_idx == 0) {
          double dw[4], dq[4];
          bspline4_poly_der(pc, jc, rtab, dw);  // (the dump's derivative values: the sums take them through the contracted tables)
#pragma unroll
          for (int k = 0; k < 4; k++) dq[k] = (pc == 0.0) ? 0.0 : dw[k];  // Q5, applied inside jac_accumulate_fast
          dump_jac(s, 0.5 * gx, 0.5 * gy, pc, jc, dq);
        }
      }
      return !SECOND && __builtin_amdgcn_ballot_w64(exact) != 0ull;
    };
    double tcol0 = SA.pose.M[3], tcol1 = SA.pose.M[7], tcol2 = SA.pose.M[11];
    asm volatile("" : "+v"(tcol0), "+v"(tcol1), "+v"(tcol2));  // (vector registers, once: see warp_fast)
    auto jac_round_masked = [&](int sb, int r, const TileIn &cur, TileIn &nxt) {
      const int s = sb + lane;
      PixelFront f;
      warp_fast(P, SA, cur, f, tcol0, tcol1, tcol2);
      // (the whole warp before the branch on `go`: left alone the compiler sinks v's arithmetic into the branch, the
      // point then lives across the prefetch and is copied from round to round again)
      asm volatile("" : "+v"(f.u), "+v"(f.v));
      if (sb + NT < g.pstride) load_tile_xyz(P, base + (unsigned)(s + NT), nxt);
      const bool go = ((gomask >> r) & 1u) != 0u && (unsigned)__double2hiint(f.u) - P.hu_lo <= P.hj_span;
      WinJ wj;
      load_win_jac_e(P, win_e0(P, go, f.u, f.v), wj);
      TileIn tw;
      load_tile_w(P, base + (unsigned)s, plane, tw);
#pragma unroll
      for (int k = 0; k < 4; k++) f.wr[k] = tw.wr[k];
      f.wr[0] = fabs(f.wr[0]);  // the sign of the first reference weight is k_href's knot flag (hist_add)
      double ic, gx, gy;
      gradient_fast_j(wj, f.u, f.v, gx, gy, ic);
      if (go) {
        double pc;
        const int jc = fast_bin<false>(ic, S, pc);
        jac_accumulate_fast(f, f.zq, gx, gy, pc, jc, std::false_type{});
      }
    };
    unsigned long long rare2 = 0ull;
    int r = 0;
    if (LAT > 0 && !repaired) {
      // main pass from the registers the cost phase left (same values, same per-lane order as the loop form);
      // the rare samples are the cost phase's rare samples (same classification on the same values)
      if constexpr (LAT > 0) {
#pragma unroll
      for (int q = 0; q < LAT; q++)
        if (lat[q].go) {
          jac_accumulate_fast(lat[q], lat[q].iz, lat[q].gx, lat[q].gy, lat[q].pc, lat[q].jc, std::integral_constant<bool, kMainPassClamps>{});
        }
      }
      rare2 = rare_rounds;
    } else {
      // (LAT after a repair pass -- rare --: the loop form, which takes nothing from the cost phase's registers: they
      // would have to live across the repair pass, and a cell that ran it evaluates its Jacobian from memory like the
      // loop-form kernels do; the same operations on the same values: the same bits)
      if (wave_base < g.pstride) load_tile_xyz(P, base + (unsigned)(wave_base + lane), prej);
      if constexpr (use_gomask) {
        // the cost phase's decisions (gomask) instead of a second classification; the rounds with rare samples are
        // the cost phase's too (same classification on the same values)
        // (tried and dropped, profiles/r03_ablations_A.txt: skipping the rounds in which the cost phase's main pass took
        // no sample of the wave -- the branch cost the plain pair 3 % --; a per-lane mask from the cost phase's second
        // pass that spares this phase's second pass the rounds whose rare samples all sit in flat windows: +-0)
#pragma clang loop unroll(disable)
        for (int sb = wave_base; sb < g.pstride; sb += NT, r++) jac_round_masked(sb, r, prej, prej);
        rare2 = rare_rounds;
      } else {
#pragma clang loop unroll(disable)
        for (int sb = wave_base; sb < g.pstride; sb += NT, r++)
          if (jac_round(sb, r, prej, prej, std::false_type{})) rare2 |= 1ull << min(r, 63);
      }
    }
#ifndef NID_ABL_NO_JAC_SECOND  // (ablation builds only)
    if (rare2 != 0ull) {
      if constexpr (use_lane_masks) {
        second_pass_rounds((unsigned)rare2, [&](int sb, int rr, const TileIn &cur) { jac_round(sb, rr, cur, prej, std::true_type{}); });
      } else {
        r = 0;
#pragma clang loop unroll(disable)
        for (int sb = wave_base; sb < g.pstride; sb += NT, r++)
          if ((rare2 >> min(r, 63)) & 1ull) jac_round(sb, r, prej, prej, std::true_type{});
      }
    }
#endif
  }
  if (!STRICT) { acc[0] = -acc[0]; acc[5] = -acc[5]; }
  NID_STAMP(4, acc[0], acc[2], acc[3], acc[5]);
  __syncthreads();  // the fold's use of `red` is over
  // Block sum of the six accumulators: every wave reduces its own (DPP), lane 63 leaves the partials in LDS;
  // after the barrier waves 1..3 retire -- the cell's tail (quadratic form, hand-off, reductions) is
  // wave 0's business, and the retired waves' slots go to the next workgroup ~5 k cycles earlier.
  if constexpr (NT <= 256 && NID_XPOSE_SUM) {
    // Throughput shapes: the sums go through LDS instead of six 64-lane DPP trees per wave (108 VALU instructions per
    // wave, 250 per workgroup with the cross-wave part): every thread leaves its six accumulators in the (free by
    // now) histogram area, row k = accumulator k; after the barrier the other waves retire and wave 0 sums row k with
    // eight lanes -- lane 8k + j adds the elements j, j + 8, ... in ascending order (NT / 8 independent reads in
    // flight), three DPP steps inside the group of eight add the parts.  One fixed order per workgroup shape.
    double *xs = reinterpret_cast<double *>(hist);
#pragma unroll
    for (int k = 0; k < 6; k++) xs[k * kXposeStride(NT) + tid] = acc[k];
    __syncthreads();
    if (tid >= 64) return done();
    double part = 0.0;
    if (tid < 48) {
      const double *row = xs + (tid >> 3) * kXposeStride(NT) + (tid & 7);
      double v[NT / 8];
#pragma unroll
      for (int i = 0; i < NT / 8; i++) v[i] = row[i * 8];
#pragma unroll
      for (int i = 0; i < NT / 8; i++) part += v[i];
    }
    part += dpp_mov_f64<0xB1, 0xf>(part);   // quad_perm [1,0,3,2]
    part += dpp_mov_f64<0x4E, 0xf>(part);   // quad_perm [2,3,0,1]
    part += dpp_mov_f64<0x141, 0xf>(part);  // row_half_mirror: all eight lanes of a group hold the group's sum
    acc[0] = wave_lane<0>(part); acc[1] = wave_lane<8>(part); acc[2] = wave_lane<16>(part);
    acc[3] = wave_lane<24>(part); acc[4] = wave_lane<32>(part); acc[5] = wave_lane<40>(part);
  } else {
    constexpr int NW = NT / 64;
#pragma unroll
    for (int k = 0; k < 6; k++) acc[k] = wave_sum_to_lane63(acc[k]);
    if ((tid & 63) == 63) {
#pragma unroll
      for (int k = 0; k < 6; k++) red[(tid >> 6) * 6 + k] = acc[k];
    }
    __syncthreads();
    if (tid >= 64) {
      if (RES) zero_histograms(64);  // (see the cost-only tail; the histograms are dead since the fold)
      return done();
    }
    if constexpr (NW <= 4) {
#pragma unroll
      for (int k = 0; k < 6; k++) {
        double sum = red[k];
        for (int w = 1; w < NW; w++) sum += red[w * 6 + k];
        acc[k] = sum;
      }
    } else {
      // 8 / 16 waves: lane k < 6 sums column k (the same order, w ascending: the same bits), NW loads in flight
      // per lane instead of 6 NW; the six sums come back as scalars
      const int kk = tid < 6 ? tid : 0;
      double sum = red[kk];
#pragma unroll
      for (int w = 1; w < NW; w++) sum += red[w * 6 + kk];
      acc[0] = wave_lane<0>(sum); acc[1] = wave_lane<1>(sum); acc[2] = wave_lane<2>(sum);
      acc[3] = wave_lane<3>(sum); acc[4] = wave_lane<4>(sum); acc[5] = wave_lane<5>(sum);
    }
  }
  NID_STAMP(5, acc[0], acc[2], acc[3], acc[5]);
  if (deferred_to_repair()) return true;
  {
    const double kappa = (double)S / 255.0;  // d_mi_i (:393), 1/N_c (:488,494), 1/Hj^2 (:521)
    const double scale = (kappa / (double)n_c) * (1.0 / (Hj * Hj));
    double J[6];
#pragma unroll
    for (int n = 0; n < 6; n++) J[n] = acc[n] * scale;
    if (direct_launch) {  // see the cost-only tail
      err = wave_uniform((2 * Hj - href - Hc) / Hj);
      double o = tid == 0 ? Hc : (tid == 1 ? Hj : (tid == 2 ? err : (double)n_c));   // per-cell outputs
      double r = 0.0;                                                                 // the Jacobian record (the residual went ahead)
#pragma unroll
      for (int n = 0; n < 6; n++) { if (tid == 3 + n) o = J[n]; if (tid == n) r = J[n]; }
      if (tid < kCellOut && SA.cellout_host) store_sys(out + tid, o);
      else if (!RES && tid < kCellOut && want_cellout) out[tid] = o;  // the slot's device block (nid_slot_buffers)
      if (tid < kDirectRec && SA.host_quad == 1) store_sys(SA.quad + ((size_t)P.g.nloc + cl) * kDirectRec + tid, r);
      NID_STAMP(6);
      NID_STAMP(7);
      return false;
    }
    residual_and_huber();
    if (tid == 0 && want_cellout) {
      out[0] = Hc; out[1] = Hj; out[2] = err;
#pragma unroll
      for (int n = 0; n < 6; n++) out[3 + n] = J[n];
      out[kCellOut - 1] = (double)n_c;
    }
    if (tid < kQuad) {  // constructQuadraticForm (base_unary_edge.hpp:56-63)
      double val = 0.0;
      if (tid == 0) val = rho0;
      else if (tid == 28) val = 1.0;
      else if (tid < 7) {
        double Jn = 0.0;
#pragma unroll
        for (int n = 0; n < 6; n++) if (n == tid - 1) Jn = J[n];
        val = 0.0 - (rho1 * Jn) * err;
      } else if (tid < 28) {
        int a, b;
        quad_ab(tid, a, b);
        double Ja = 0.0, Jb = 0.0;
#pragma unroll
        for (int n = 0; n < 6; n++) { if (n == a) Ja = J[n]; if (n == b) Jb = J[n]; }
        val = (Ja * rho1) * Jb;
      }
      store_sc1(quad + tid, val);
    }
  }
  NID_STAMP(6);
  finish_and_reduce_w0(P, SA, cl, tid);
  NID_STAMP(7);
  return false;
}

#ifndef NID_EXT_VGPRS
#define NID_EXT_VGPRS 0
#endif
template <int NT, bool JAC, bool STRICT, int NB, bool DBG, bool EXT = false, int LAT = 0, bool BIG = false>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu((STRICT || DBG || LAT || NT >= 512) ? 4 : (EXT ? NID_EXT_WAVES : NID_FAST_WAVES))))
#if NID_EXT_VGPRS
__attribute__((amdgpu_num_vgpr((EXT && !STRICT && !DBG && LAT == 0 && NT <= 256) ? NID_EXT_VGPRS : 0)))
#endif
void k_eval2(EvalParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Geometry &g = P.g;
  // XCD-aware block -> (cell, pose) map: workgroups are dealt round-robin over the 8 XCDs, so
  // id % 8 fixes the XCD; all `batch` poses of a cell get the same id % 8 and consecutive slots
  // of that XCD's stream, i.e. they run together on one XCD and share the cell's tile rows in its
  // L2 (measured on the 1280x960 config: HBM/MALL traffic per launch 1.13 GB -> see DESIGN.md).
  // Speed only: nothing depends on the placement.
  const int bid = blockIdx.x;
  const int q = bid >> 3;
  // (Round 6 tried the opposite -- the workgroups an XCD runs side by side spread over 2 / 4 / 8 cells far apart in the image,
  // so that a saturated cell's latency-bound second passes overlap other cells' arithmetic: plain pair 790 -> 840-850 us per
  // 256 poses, flash pair 1047 -> 1280-1470 us, profiles/r06_ablations_A.txt.  The poses of ONE cell side by side it stays.)
  const int pose_idx = q % P.batch;
  const int cl = (q / P.batch) * 8 + (bid & 7);
  if (cl >= g.nloc) return;  // padding of the last group of 8 cells
  SlotArgs sa_ext;
  if (EXT) {
    static_assert(sizeof(SlotArgs) % 4 == 0, "SlotArgs is copied dword by dword");
    // The record array is written before the launch (in-stream copy) and never during it: read it through the CONSTANT
    // address space, i.e. with a handful of scalar loads (s_load_dwordx16 ...) instead of one vector load and one
    // v_readfirstlane per dword (84 instructions at the head of every wave).  pose_idx is uniform (from blockIdx).
    typedef const unsigned __attribute__((address_space(4))) *ConstDwords;
    ConstDwords src = (ConstDwords)(reinterpret_cast<uintptr_t>(P.slots_ext + pose_idx));
    unsigned *dst = reinterpret_cast<unsigned *>(&sa_ext);
    // FAST math transforms with the matrix; the quaternion (the first kPoseQuatDwords of the record) is only
    // needed by exact_decisions, which fetches it itself: 14 scalar registers less across the pixel loops
    static_assert(offsetof(SlotArgs, pose) == 0 && offsetof(Pose, q) == 0 && offsetof(Pose, M) == 4 * kPoseQuatDwords, "record layout");
#pragma unroll
    for (unsigned i = STRICT ? 0u : kPoseQuatDwords; i < sizeof(SlotArgs) / 4; i++) dst[i] = src[i];
  }
  const SlotArgs &SA = EXT ? sa_ext : P.slot[pose_idx];
  eval_cell<NT, JAC, STRICT, NB, DBG, EXT, LAT, BIG, false>(P, SA, cl, pose_idx, smem);
}

// The cells and poses a loop-form launch left in its repair queue (eval_cell's REPAIR_INLINE; rare: kLinFlagW), done from
// the start by the instantiation that repairs inline.  Enqueued behind EVERY loop-form launch on its stream, with the
// same arguments; a small grid whose workgroups stride over the queue -- an empty queue costs one load per workgroup.
// The last workgroup out clears the queue for the stream's next launch (arrival ticket; the stream keeps the next
// launch's kernels behind this one).  Registers instead of occupancy: no scratch segment.
template <int NT, bool JAC, bool STRICT, int NB, bool EXT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(1, 2))) void k_repair(EvalParams P) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned *q = P.repair_queue;
  const unsigned count = (unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  for (unsigned i = blockIdx.x; i < count; i += gridDim.x) {
    __syncthreads();  // every wave is back from the previous entry (its tail is wave 0's business)
    const unsigned e = (unsigned)__builtin_amdgcn_readfirstlane((int)q[2 + 3 * i]);
    const uint2 preset = {(unsigned)__builtin_amdgcn_readfirstlane((int)q[3 + 3 * i]), (unsigned)__builtin_amdgcn_readfirstlane((int)q[4 + 3 * i])};
    const int cl = (int)(e & 0xFFFFu), pose_idx = (int)(e >> 16);
    SlotArgs sa_ext;
    if (EXT) {
      typedef const unsigned __attribute__((address_space(4))) *ConstDwords;
      ConstDwords src = (ConstDwords)(reinterpret_cast<uintptr_t>(P.slots_ext + pose_idx));
      unsigned *dst = reinterpret_cast<unsigned *>(&sa_ext);
#pragma unroll
      for (unsigned k = STRICT ? 0u : kPoseQuatDwords; k < sizeof(SlotArgs) / 4; k++) dst[k] = src[k];
    }
    const SlotArgs &SA = EXT ? sa_ext : P.slot[pose_idx];
    eval_cell<NT, JAC, STRICT, NB, false, EXT, 0, false, false, true, true>(P, SA, cl, pose_idx, smem, ResCell{0, 0.0, true}, preset);
  }
  if (count == 0u && blockIdx.x != 0u) return;  // (nothing queued: one workgroup keeps the ticket's books)
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned arrivals = count == 0u ? 1u : gridDim.x;
    __threadfence();
    if (__hip_atomic_fetch_add(q + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == arrivals - 1u) {
      __hip_atomic_store(q, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(q + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

}  // namespace nid
