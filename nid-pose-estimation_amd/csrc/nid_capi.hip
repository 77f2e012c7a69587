// nid_capi.hip -- context management and the C-ABI of include/nid/nid_c.h.
//
// Host side of the operator boundary that replaces the reference's three
// CUDA host wrappers (Calculate3Dpoint CudaPoints3d.cu:35-73, CudaComputeHref
// CudaComputeHref.cu:139-222, g2o::CudaComputeH computeH.cu:373-502).  Unlike
// those, nothing is allocated, zeroed or re-uploaded per call: every buffer is
// persistent in the context, the pose travels as a kernel argument, and a call
// is [evaluation kernel -> reduction kernel -> small D2H].
#include <hip/hip_runtime.h>

#include <atomic>
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "nid/nid_c.h"
#include "nid_eval_launch.h"
#ifndef NID_REPAIR_NT
#define NID_REPAIR_NT 256  // (nid_eval_tu.inc)
#endif
#include "nid_setup_kernels.hip.h"     // k_tile, k_im1_margins, k_backproject_plain, k_href, k_plain_nid, k_untile_bs: this translation unit's
#include "nid_resident_kernels.hip.h"  // control words and record layouts of the resident evaluators (their kernels: nid_resident_tu.hip)

using namespace nid;

namespace {

struct Slot {
  double *cellout_dev = nullptr;
  double *reduced_dev = nullptr;   // device scratch target (evaluate path / slot_buffers)
  double *quad_dev = nullptr;      // per-cell quadratic forms
  unsigned *ticket_dev = nullptr;  // [0] top, [1+g] groups
  double *gpart_dev = nullptr;     // group sums
  double *cellout_host = nullptr;  // pinned, mapped: the blocking per-cell calls let the kernel write here directly
  double *cellout_host_devptr = nullptr;
  double *reduced_host = nullptr;  // pinned, mapped: [32 doubles][u64 sequence word]
  double *reduced_host_devptr = nullptr;
  double *quad_host = nullptr;     // pinned, mapped, created on first use: [nloc][kDirectRec] per-cell records of a DIRECT launch (wait_direct)
  double *quad_host_devptr = nullptr;
  double *groups_host = nullptr;   // pinned, mapped, created on first use: [ngroups][32] group sums of a GROUP-DIRECT launch (wait_groups)
  double *groups_host_devptr = nullptr;
  bool groups = false;             // the launch in flight is a GROUP-DIRECT one (nid_set_direct_results(ctx, 2))
  bool groups_dirty = false;
  bool direct = false;             // the launch in flight is a DIRECT one: the host forms and sums the cells' quadratic forms
  bool resident = false;           // ... it is a request to the resident kernel (its records arrive in ctx->res.rec_host)
  bool collected = false;          // its result is in reduced_host already (resident_quiesce): nid_wait only hands it over
  bool quad_dirty = false;         // a DIRECT launch wrote (or may still write) into quad_host and wait_direct has not consumed it
  bool direct_jac = false;         // ... it carries Jacobians
  double direct_delta = 0.0;       // ... its Huber delta
  unsigned long long seq = 0;      // sequence number of the last launch into this slot
  hipEvent_t done = nullptr, e0 = nullptr, e1 = nullptr;  // e0 / e1: timing events, created on first use
  bool pending = false;
  bool timed = false;
  bool external_target = false;
  int done_slot = 0;               // external targets: the slot whose `done` event covers this one's launch
};

}  // namespace

struct nid_ctx {
  nid_config cfg{};
  Geometry g{};
  int jac_bound = NID_JACBOUND_CPU;
  int xform = NID_XFORM_QUAT;
  int jac_threads = 0, cost_threads = 0;  // nid_set_launch_shape: 0 = default (128) / automatic (pick_threads)
  bool loop_form = false;                  // nid_set_loop_form (diagnostics)
  bool href_nan_rows = false;              // nid_set_href_nan_markers: bs_value rows of pixels without a sample read NaN (the legacy operators' convention)
  int direct_mode = 1;                     // nid_set_direct_results: 0 in-launch reduction, 1 DIRECT records, 2 GROUP-DIRECT (group sums on the device)
  bool direct_results = true;              // nid_set_direct_results: single-pose launches whose result the host waits for are DIRECT
  int seq_chunk = 0, seq_streams = 0;      // nid_set_short_sequence_policy: poses per launch / streams of a SHORT sequence (0 = the measured table)
  // the RESIDENT evaluator (nid_set_resident; k_resident in nid_kernels.hip.h)
  struct Resident {
    bool enabled = false;          // asked for
    bool running = false;          // the kernel is on the device
    int probed = 0;                // 0: not yet, 1: the mailbox is CPU-addressable, -1: it is not (resident launches unavailable)
    int nt = 0;                    // workgroup shape of the running kernel
    ResidentCtl *ctl = nullptr;    // fine-grained device memory, written by the CPU through the PCIe BAR
    hipStream_t stream = nullptr;
    unsigned long long seq = 0;
    std::chrono::steady_clock::time_point last_post{};
    double *rec_host = nullptr, *rec_devptr = nullptr;  // [nloc][kDirectRec], pinned + mapped
    int pending_slot = -1;         // the slot whose request is in flight (one at a time)
    // the request in flight, for the fallback (a kernel that had left: re-issued as an ordinary DIRECT launch)
    Pose pose{};
    bool jac = false, want_cellout = false;
    long served = 0, fallbacks = 0, starts = 0;
    int fallback_run = 0;          // consecutive requests that timed out (resident_fallback switches the mode off after a few)
    std::string why;               // probed < 0 / unfit_nt: which step said no
    int unfit_nt = 0;              // the launch shape a start was refused for (its workgroups do not all fit the device)
  } res;
  std::vector<double> direct_rho1;  // wait_direct: the cells' Huber weights between its two passes
  // own_stream: setup + blocking calls; aux_stream: odd slots of the pipelined path, so that
  // launch N+1 overlaps the reduction tail and the launch gap of launch N (separate
  // per-slot buffers make that safe).  An external stream (nid_set_stream) disables it.
  hipStream_t own_stream = nullptr, aux_stream = nullptr, stream = nullptr;
  bool external_stream = false;
  Tiles t{};
  // launches of more than kMaxBatch poses: per-pose argument records travel through a small ring of device
  // arrays (host fills the pinned mirror, one in-stream copy, then the kernel)
  static constexpr int kExtRing = 4;
  SlotArgs *ext_dev[kExtRing] = {nullptr, nullptr, nullptr, nullptr};
  SlotArgs *ext_host[kExtRing] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ext_done[kExtRing] = {nullptr, nullptr, nullptr, nullptr};
  bool ext_busy[kExtRing] = {false, false, false, false};
  int ext_next = 0;
  uint8_t *im1_dev = nullptr, *im0_dev = nullptr;
  int16_t *im1s_dev = nullptr;   // int16 copy with the extrapolated top / left margin the evaluation kernel samples (k_im1_margins)
  int im1_stride = 0;
  double *depth_dev = nullptr, *points_dev = nullptr, *Twc_dev = nullptr;
  uint16_t *depth16_dev = nullptr;   // nid_set_pair_u16: the depth map as uploaded, and the pinned block its inputs and
  uint8_t *pair_stage = nullptr;     // per-cell outputs travel through (created on first use)
  int *Nc_dev = nullptr;
  double *Href_dev = nullptr;
  double *dbg_u = nullptr, *dbg_v = nullptr, *dbg_ic = nullptr, *dbg_wc = nullptr;
  int *dbg_jc = nullptr;
  long long *dbg_stamps = nullptr;
  // compute_href's per-pixel outputs in image order (k_untile_bs) and their pinned staging, created on first use
  double *bsv_img_dev = nullptr, *bsv_stage = nullptr;
  int *bsi_img_dev = nullptr, *bsi_stage = nullptr;
  hipEvent_t bs_part_done[4] = {nullptr, nullptr, nullptr, nullptr};  // (kBsParts)
  unsigned long long *repair_count_dev = nullptr;  // EvalParams::repair_count (nid_debug_repair_count)
  unsigned *repair_queue_dev[2] = {nullptr, nullptr};  // EvalParams::repair_queue of launches on `stream` / on aux_stream (k_repair)
  bool dbg_enabled = false;
  int dbg_jac = 0;
  bool timing = false;
  bool have_ref = false, have_target = false, have_href = false, ref_from_depth = false;
  double hist_scale = 0, hist_inv_scale = 0;            // k_href: whole-cell sums below 2^62
  int group_size = 1, ngroups = 1;
  int math_mode = NID_MATH_FAST;
  double *ctab_dev = nullptr;
  Slot slots[NID_SLOTS];
  // nid_run_sequence: result buffers of the launches in flight (device + pinned host), their copy stream and events
  static constexpr int kSeqRing = 16;  // launches in flight: min(kSeqRing, NID_SLOTS / batch)
  double *seq_dev[kSeqRing] = {}, *seq_host[kSeqRing] = {};
  hipEvent_t seq_done[kSeqRing] = {}, seq_fence[kSeqRing] = {};
  size_t seq_cap = 0;
  hipStream_t copy_stream = nullptr;
  // what the slots' buffers are carved from (one allocation per kind)
  double *slab_cellout = nullptr, *slab_reduced = nullptr, *slab_quad = nullptr, *slab_gpart = nullptr, *slab_reduced_host = nullptr;
  unsigned *slab_ticket = nullptr;
  std::string last_error;
};

namespace {

void resident_retire(nid_ctx *ctx);  // every call that changes what a resident kernel has cached, or frees memory, retires it first
int launch_batch(nid_ctx *ctx, int first_slot, int n, const Pose *poses, int want_jac, double delta, double *reduced_dev_base = nullptr,
                 bool on_aux_stream = false, bool relaunch_ok = false, bool allow_direct = true);
int resident_quiesce(nid_ctx *ctx);  // ... and every ordinary evaluation launch (the resident workgroups hold most of every CU)

#define NID_HIP(ctx, expr)                                                            \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);          \
      return NID_ERR_HIP;                                                             \
    }                                                                                 \
  } while (0)

template <typename T>
int dev_alloc(nid_ctx *ctx, T **p, size_t n) {
  NID_HIP(ctx, hipMalloc(reinterpret_cast<void **>(p), n * sizeof(T)));
  return NID_OK;
}

void pose_from_pose7(const double *p, int mode, Pose *out) {
  // to_homogeneous_matrix (se3quat.h:270-278) = Eigen toRotationMatrix
  for (int i = 0; i < 7; i++) out->q[i] = p[i];
  const double x = p[0], y = p[1], z = p[2], w = p[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  double *M = out->M;
  M[0] = 1 - (tyy + tzz); M[1] = txy - twz;       M[2] = txz + twy;        M[3] = p[4];
  M[4] = txy + twz;       M[5] = 1 - (txx + tzz); M[6] = tyz - twx;        M[7] = p[5];
  M[8] = txz - twy;       M[9] = tyz + twx;       M[10] = 1 - (txx + tyy); M[11] = p[6];
  out->mode = mode;
}

void pose_from_matrix16(const double *m, Pose *out) {
  for (int i = 0; i < 7; i++) out->q[i] = 0.0;
  out->q[3] = 1.0;
  for (int r = 0; r < 3; r++)
    for (int c = 0; c < 4; c++) out->M[4 * r + c] = m[c * 4 + r];
  out->mode = NID_XFORM_MATRIX;
}

// FAST mode: polynomial form of the four cubic B-spline basis functions of every span of the
// clamped knot vector (types_six_dof_expmap.h:283-296), by running the Cox-de Boor recursion
// (types_six_dof_expmap.cpp:738-764) on polynomials in t = u - knots[j] in long double.
void build_coef_table(int S, std::vector<double> *out) {
  typedef long double ld;
  auto knot = [&](int i) -> ld { int k = i - 3; k = k < 0 ? 0 : k; k = k > S ? S : k; return (ld)k; };
  out->assign((size_t)S * kCoefRow, 0.0);
  for (int jj = 0; jj < S; jj++) {
    const int j = jj + 3;
    const ld t0 = knot(j);
    // basis[order][i - (j - order + 1)] as polynomials in t
    ld cur[4][4] = {{0}}, nxt[4][4];
    cur[0][0] = 1.0L;  // N_{j,1} = 1 on the span
    for (int order = 2; order <= 4; order++) {
      for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) nxt[a][b] = 0.0L;
      for (int n = 0; n < order; n++) {      // basis index i = j - order + 1 + n
        const int i = j - order + 1 + n;
        // term 1: (u - t_i)/(t_{i+order-1} - t_i) * B_{i,order-1}      (B_{i,order-1} = cur[n-1])
        if (n >= 1) {
          const ld d = knot(i + order - 1) - knot(i);
          if (d != 0.0L) {
            const ld a0 = (t0 - knot(i)) / d, a1 = 1.0L / d;
            for (int p = 0; p < 4; p++) {
              nxt[n][p] += a0 * cur[n - 1][p];
              if (p + 1 < 4) nxt[n][p + 1] += a1 * cur[n - 1][p];
            }
          }
        }
        // term 2: (t_{i+order} - u)/(t_{i+order} - t_{i+1}) * B_{i+1,order-1}  (= cur[n])
        if (n <= order - 2) {
          const ld d = knot(i + order) - knot(i + 1);
          if (d != 0.0L) {
            const ld a0 = (knot(i + order) - t0) / d, a1 = -1.0L / d;
            for (int p = 0; p < 4; p++) {
              nxt[n][p] += a0 * cur[n][p];
              if (p + 1 < 4) nxt[n][p + 1] += a1 * cur[n][p];
            }
          }
        }
      }
      for (int a = 0; a < 4; a++) for (int b = 0; b < 4; b++) cur[a][b] = nxt[a][b];
    }
    for (int k = 0; k < 4; k++) {  // basis index j-3+k = jc+k
      double *row = out->data() + (size_t)jj * kCoefRow + 7 * k;
      for (int p = 0; p < 4; p++) row[p] = (double)cur[k][p];
      row[4] = (double)cur[k][1];
      row[5] = (double)(2.0L * cur[k][2]);
      row[6] = (double)(3.0L * cur[k][3]);
    }
  }
}

// k_eval2 accumulates the bit patterns of subnormal products = RN(w * 2^s) as plain 64-bit integers (fx_bits in
// nid_kernels.hip.h): a single addend needs w * 2^s < 2^53 (s <= 52 for weights <= 1), a whole cell's bin -- at most
// pstride weights, each <= 1, over all copies -- must stay below 2^63
int eval_hist_shift(const Geometry &g) {
  int bits = 0;
  while ((1L << bits) < (long)g.pstride + 1) bits++;
  return std::min(52, 63 - bits);
}

void set_hist_params(EvalParams &P) {
  const int hs = eval_hist_shift(P.g);
  P.hist_dn = std::ldexp(1.0, hs - 562);    // times kWcPre = 2^-512 on the other factor: 2^(hs - 1074)
  P.hist_inv_scale = std::ldexp(1.0, -hs);
}

// nb_spec: the bin count the kernel is specialised for (its template's NB; 0 = a generic kernel): eval_hist_copies
size_t eval_lds_bytes(const Geometry &g, int nt, bool resident = false, int nb_spec = -1) {
  if (nb_spec < 0) nb_spec = (g.nb == 8 || g.nb == 10) ? g.nb : 0;
  const int nbins = g.nb * g.nb + g.nb;
  // copies + fine levels (the Jacobian block sum of the throughput shapes reuses the area: at least kXposeDoubles); tab + term
  // clamped samples: coarse copies, fine levels, folded sums, flags; near-saturated samples: folded sums, and their bins
  // unless the weight tables lend them their area (near_sat_aliased)
  const size_t clamp_bytes = (size_t)kClampBins(g.nb) * (kClampCopies + kFineLevels + 1) * 8 + (size_t)kFlagWords * 4 + (size_t)(g.nb + 1) * 8 +
                             ((near_sat_aliased(g.nb) && !resident) ? 0 : (size_t)kNearSatBinBytes(g.nb) + 8);
  const size_t hist_bytes = std::max((size_t)nbins * (eval_hist_copies(nt, nb_spec) + kFineLevels) * 8 + clamp_bytes, (size_t)kXposeDoubles(nt) * 8);
  return hist_bytes + 2 * (size_t)((nbins + 1) & ~1) * 8 + (size_t)g.S * kCoefRow * 8 + (size_t)kFlagDoubles * 8 + (size_t)kRedDoubles(nt) * 8;
}

constexpr int lat_rounds(int nt) { return nt == 512 ? 3 : (nt == 1024 ? 2 : 0); }

// Threads per workgroup (one workgroup per cell and pose) of a launch.
//  * Cost + Jacobian launches use the context's shape (nid_set_launch_shape; default 128): the six Jacobian sums
//    depend on the shape in their last bits, and a pose must give the same bits alone and in a batch.
//  * Cost-only launches (the trial poses of an LM iteration) give the same bits in every shape -- integer
//    histograms, entropy sums in a fixed order --, so unless the context pins their shape they take the one that
//    fills the chip: ~8 waves per SIMD at most, i.e. 1024 threads for a single pose of 256 cells (a cell's
//    1200 pixels in two rounds instead of ten), 128 from 16 poses on.  Measured: tools/latency_sweep.py.
int pick_threads(const nid_ctx *ctx, bool jac, int batch) {
  int nt = jac ? ctx->jac_threads : ctx->cost_threads;
  if (nt == 0) {
    if (jac) return 128;
    const long wg = (long)ctx->g.nloc * batch;
    nt = 1024;
    while (nt > 128 && wg * (nt / 64) > 8L * 1024) nt >>= 1;  // 1024 SIMDs
  }
  if (batch > kMaxBatch && nt > 256) nt = 256;  // the latency shapes exist for launches of <= kMaxBatch poses
  return nt;
}

int launch_eval2(nid_ctx *ctx, EvalParams &P, bool jac, hipStream_t stream, int batch, hipEvent_t eval_end = nullptr) {
  { int rc = resident_quiesce(ctx); if (rc) return rc; }
  P.batch = batch;
  // Workgroup shape of the throughput path: 128 threads.  Measured on MI355X (16 poses per launch, two launches in
  // flight): 640x480 / 8 bins 237k (256 threads) -> 260k (128) evaluations/s, 1280x960 63.8k -> 70.5k; one-wave
  // workgroups with 8 histogram copies measured 248k / 66.5k and were dropped.  Two waves per workgroup halve the
  // wave-time lost at the workgroup's barriers and in its serial phases (fold, block sums, tail), and ten workgroups
  // still fit a CU (LDS 16 KB, 96 VGPRs).  Launches of few poses are latency bound: see pick_threads.
  const bool dbg = ctx->dbg_enabled || ctx->dbg_stamps != nullptr;
  int nt = pick_threads(ctx, jac, batch);
  // the diagnostic instantiations exist for 128 and 256 threads; phase stamps alone also for the latency form
  const bool stamps_lat = dbg && !ctx->dbg_enabled && !ctx->loop_form && ctx->math_mode != NID_MATH_STRICT && nt >= 512 &&
                          P.g.pstride <= lat_rounds(nt) * nt && batch <= kMaxBatch;
  if (dbg && nt > 256 && !stamps_lat) nt = 256;
  set_hist_params(P);
  const bool strict = ctx->math_mode == NID_MATH_STRICT;
  // The kernels live in one translation unit per workgroup shape and kind (nid_eval_launch.h).  Families: the latency
  // form (512 / 1024 threads, FAST math, <= kMaxBatch poses, LAT rounds cover the cell) with or without phase stamps;
  // diagnostics (128 / 256 threads: they keep the workgroup shape, the Jacobian sums depend on it in their last bits);
  // cells of more than 32 * NT slots (test geometries: a single cell of 6 144 / 19 200 pixels: the FAST kernels' BIG
  // instantiation, generic bin count only); everything else the loop form.
  P.repair_queue = ctx->repair_queue_dev[(stream == ctx->aux_stream && !ctx->external_stream) ? 1 : 0];
  int family = kFamLoop;
  if (stamps_lat) family = kFamStampsLat;
  else if (dbg) family = kFamDbg;
  else if (!strict && P.g.pstride > 32 * nt) family = kFamBig;
  else if (nt >= 512) {
    static const bool no_lat = getenv("NID_NO_LAT") != nullptr;  // experiments: the loop form at the latency shapes
    if (!strict && !no_lat && !ctx->loop_form && P.g.pstride <= lat_rounds(nt) * nt) family = kFamLat;
  }
  if (family == kFamLat && batch > kMaxBatch) return NID_ERR_INVALID_ARG;
  // (the bin-specialised kernels -- the loop and latency families at 8 / 10 bins -- may keep fewer histogram copies than the
  // generic ones the other families run: eval_hist_copies)
  const int nb_spec = (family == kFamLoop || family == kFamLat) && (P.g.nb == 8 || P.g.nb == 10) ? P.g.nb : 0;
  size_t lds = eval_lds_bytes(P.g, nt, false, nb_spec);
  if (lds > 160 * 1024) return NID_ERR_UNSUPPORTED;
  static const char *pad_env = getenv("NID_OCCUPANCY_LDS_PAD");  // occupancy experiments: pad the LDS request
  if (pad_env) lds = std::min<size_t>(160 * 1024, lds + (size_t)atoi(pad_env));
  const size_t lds_repair = eval_lds_bytes(P.g, std::max(nt, NID_REPAIR_NT), false, nb_spec);  // k_repair's workgroup shape: nid_eval_tu.inc
  switch (nt) {
    case 128: (jac ? launch_eval_128_jac : launch_eval_128_cost)(P, family, strict, lds, lds_repair, stream, batch, eval_end); break;
    case 256: (jac ? launch_eval_256_jac : launch_eval_256_cost)(P, family, strict, lds, lds_repair, stream, batch, eval_end); break;
    case 512: (jac ? launch_eval_512_jac : launch_eval_512_cost)(P, family, strict, lds, lds_repair, stream, batch, eval_end); break;
    default: (jac ? launch_eval_1024_jac : launch_eval_1024_cost)(P, family, strict, lds, lds_repair, stream, batch, eval_end); break;
  }
  NID_HIP(ctx, hipGetLastError());
  return NID_OK;
}

int launch_eval(nid_ctx *ctx, EvalParams &P, bool jac, hipStream_t stream, int batch = 1, hipEvent_t eval_end = nullptr) {
  return launch_eval2(ctx, P, jac, stream, batch, eval_end);
}

void fill_common_params(nid_ctx *ctx, double delta, EvalParams *P) {
  P->huber_delta = delta;
  P->huber_dsqr = (float)(delta * delta);  // RobustKernelHuber::setDelta, float dsqr (robust_kernel_impl.h:84)
  P->group_size = ctx->group_size;
  P->ctab = ctx->ctab_dev;
  P->slots_ext = nullptr;  // launch_batch points it at a device array for more than kMaxBatch poses
  P->g = ctx->g;
  P->t = ctx->t;
  P->im1s = ctx->im1s_dev;
  P->im1_stride = ctx->im1_stride;
  P->Nc = ctx->Nc_dev;
  P->Href = ctx->Href_dev;
  P->jac_cols = (ctx->jac_bound == NID_JACBOUND_CPU) ? ctx->g.cols - 1 : ctx->g.cols;
  {
    // FAST border bounds as ranges of the coordinates' high dwords (EvalParams::hu_lo ...): strictly inside
    // [kBorderEps, bound - kBorderEps] at the high dword's granularity
    auto hi = [](double x) { uint64_t b; std::memcpy(&b, &x, 8); return (unsigned)(b >> 32); };
    const unsigned lo = hi(kBorderEps) + 1u;
    auto span = [&](double bound) { const unsigned top = hi(bound - kBorderEps); return top > lo + 1u ? top - 1u - lo : 0u; };
    P->hu_lo = lo;
    P->hu_span = span((double)ctx->g.cols - 3.0);
    P->hv_span = span((double)ctx->g.rows - 3.0);
    P->hj_span = span((double)P->jac_cols - 3.0);
  }
  P->hist_dn = P->hist_inv_scale = 0.0;  // set by launch_eval2
  if (ctx->dbg_enabled) {
    P->dbg_u = ctx->dbg_u; P->dbg_v = ctx->dbg_v; P->dbg_ic = ctx->dbg_ic;
    P->dbg_wc = ctx->dbg_wc; P->dbg_jc = ctx->dbg_jc;
    P->dbg_jac = ctx->dbg_jac;
  } else {
    P->dbg_u = P->dbg_v = P->dbg_ic = P->dbg_wc = nullptr;
    P->dbg_jc = nullptr;
  }
  P->dbg_stamps = ctx->dbg_stamps;
  P->repair_count = ctx->repair_count_dev;
  P->repair_queue = ctx->repair_queue_dev[0];  // (launch_eval2 picks the launch stream's)
}

void fill_slot_args(const Pose &pose, Slot &S, double *out_reduced, unsigned long long *host_seq, SlotArgs *A) {
  A->pose = pose;
  A->cellout = S.cellout_dev;
  A->quad = S.quad_dev;
  A->gpart = S.gpart_dev;
  A->ticket = S.ticket_dev;
  A->out_reduced = out_reduced;
  A->host_seq = host_seq;
  A->launch_seq = S.seq;
  A->cellout_host = 0;
  A->host_quad = 0;
}

// ---- DIRECT launches ---------------------------------------------------------------------------------------------
// One pose, the host is waiting for the result (a Gauss-Newton / LM loop is a chain of such launches).  Every cell's
// workgroup writes its record -- err, J[6], an active flag: 64 bytes -- straight to pinned host memory and the HOST
// forms the Huber-weighted quadratic forms (RobustKernelHuber::robustify robust_kernel_impl.cpp:77-91 with the float
// dsqr, constructQuadraticForm base_unary_edge.hpp:56-63: the kernel's operations in the kernel's order, all IEEE) and
// adds them up in the order the in-launch reduction uses (sum_blocks_w0 twice: per group the even-numbered cells in
// ascending order plus the odd-numbered ones, then the groups likewise), so the 6x6 system has the same bits either
// way.  The launch has no ticket, no device-scope round trip and no fence; for the host it is over when the last
// cell's 64 bytes have crossed PCIe.  No word is ordered against any other: the host pre-fills the records with a
// sentinel no arithmetic produces and polls every word; a consumed record is reset on the spot.  (Round 2 measured 5 us
// of reduction tail behind the last workgroup of a single-pose launch: seven dependent device-scope round trips; a
// first version of this path that shipped the cells' finished 32-double blocks spent 4.5 us reading 64 KB of freshly
// written host memory behind the last arrival.)
void fill_sentinel(double *p, size_t n) {
  unsigned long long *q = reinterpret_cast<unsigned long long *>(p);
  for (size_t i = 0; i < n; i++) q[i] = kHostSentinel;
}

int ensure_quad_host(nid_ctx *ctx, Slot &S) {
  if (S.quad_host && S.quad_dirty) {
    // the previous DIRECT launch into this buffer was never collected (its wait failed): let it finish, start clean
    NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
    NID_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));
    fill_sentinel(S.quad_host, (size_t)2 * ctx->g.nloc * kDirectRec);
    S.quad_dirty = false;
  }
  if (S.quad_host) return NID_OK;
  const size_t n = (size_t)2 * ctx->g.nloc * kDirectRec;
  if (hipHostMalloc(reinterpret_cast<void **>(&S.quad_host), n * sizeof(double), hipHostMallocMapped) != hipSuccess) {
    S.quad_host = nullptr;
    return NID_ERR_NOMEM;
  }
  fill_sentinel(S.quad_host, n);
  NID_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void **>(&S.quad_host_devptr), S.quad_host, 0));
  return NID_OK;
}

// may this launch be DIRECT?  (timed and diagnostic launches keep the in-launch reduction: they may be re-issued into
// the same buffers before the host has looked)
bool direct_ok(const nid_ctx *ctx) {
  return ctx->direct_results && !ctx->timing && !ctx->dbg_enabled && !ctx->dbg_stamps;
}

// GROUP-DIRECT (nid_set_direct_results(ctx, 2)): the whole per-cell tail stays on the device -- Huber kernel, quadratic
// form, and the first level of the reduction (every group's last workgroup sums its group's blocks) -- and the group
// sums go straight to pinned host memory, one 256-byte block per group; the host adds the <= 32 group blocks in
// sum_blocks_w0's order (the second level: even-numbered groups ascending plus the odd-numbered ones).  Against the
// in-launch reduction it saves the second ticket round trip, the re-read of the group sums and the system-scope fence;
// against DIRECT it keeps a15 on the GPU and costs the first ticket round trip.  Same bits as both.
int ensure_groups_host(nid_ctx *ctx, Slot &S) {
  const int ngroups = (ctx->g.nloc + ctx->group_size - 1) / ctx->group_size;
  const size_t n = (size_t)ngroups * kQuad;
  if (S.groups_host && S.groups_dirty) {
    NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
    NID_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));
    fill_sentinel(S.groups_host, n);
    S.groups_dirty = false;
  }
  if (S.groups_host) return NID_OK;
  if (hipHostMalloc(reinterpret_cast<void **>(&S.groups_host), n * sizeof(double), hipHostMallocMapped) != hipSuccess) {
    S.groups_host = nullptr;
    return NID_ERR_NOMEM;
  }
  fill_sentinel(S.groups_host, n);
  NID_HIP(ctx, hipHostGetDevicePointer(reinterpret_cast<void **>(&S.groups_host_devptr), S.groups_host, 0));
  return NID_OK;
}

// true when all `n` words at p have arrived
inline bool words_arrived(const double *p, int n) {
  const volatile unsigned long long *q = reinterpret_cast<const volatile unsigned long long *>(p);
  for (int i = 0; i < n; i++)
    if (q[i] == kHostSentinel) return false;
  return true;
}

// NID_DIRECT_TRACE=1 (experiments): per wait, microseconds from entering wait_direct to the first cell's arrival, to
// the last moment the host had to WAIT for a cell, and to the end (the difference of the last two = the host's own
// backlog behind the last arrival); averages printed every 2000 waits
struct DirectTrace { double first = 0, last_wait = 0, end = 0; long n = 0; };
static DirectTrace g_dtrace;

// spin until `arrived()`; after a long while let a device error surface through the runtime, then give up
template <typename F>
int spin_until(nid_ctx *ctx, F arrived, unsigned long &spins, bool &synced, bool &waited) {
  while (!arrived()) {
    waited = true;
    if (++spins > 50000000ul) {
      if (synced) { ctx->last_error = "a cell's results never arrived"; return NID_ERR_HIP; }
      NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
      NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
      NID_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));
      synced = true;
      spins = 0;
    }
  }
  return NID_OK;
}

// One cell's quadratic form, from its record, ADDED to acc[0..28] -- exactly the 32-double block k_eval2's tail forms
// (residual_and_huber + constructQuadraticForm: the same IEEE operations on the same values) and the same addition
// sum_blocks_w0 performs.  Terms that are +0.0 by construction (an inactive cell's block, the b and H entries of a
// cost-only block) are not added: an accumulator that starts at +0.0 never holds -0.0, so x + 0.0 is x, bit for bit.
inline void huber_weights(double err, double delta, float dsqr, double &rho0, double &rho1) {
  const double e2 = err * err;  // robust_kernel_impl.cpp:77-91 (float dsqr)
  rho0 = e2; rho1 = 1.0;
  if (!(e2 <= dsqr)) {
    const double sqrte = std::sqrt(e2);
    rho0 = 2 * sqrte * delta - dsqr;
    rho1 = delta / sqrte;
  }
}

// first half, from the record's early words (err, active flag): chi2 and the count; returns rho1 (0 for a level-1 edge)
inline double add_record_cost(const double *rec, double delta, float dsqr, double *acc) {
  if (rec[1] == 0.0) return 0.0;  // level-1 edge
  double rho0, rho1;
  huber_weights(rec[0], delta, dsqr, rho0, rho1);
  acc[0] += rho0;
  acc[28] += 1.0;
  return rho1;
}
// second half, from the Jacobian words: b and the upper triangle of H
inline void add_record_jac(const double *J, double err, double rho1, double *acc) {
  for (int n = 0; n < 6; n++) acc[1 + n] += 0.0 - (rho1 * J[n]) * err;
  int idx = 7;
  for (int a = 0; a < 6; a++)
    for (int b = a; b < 6; b++, idx++) acc[idx] += (J[a] * rho1) * J[b];
}

// the same sums with AVX-512 (nid_hostsum.cpp: host-only C++ next to this file; the same operations lane by lane)
extern "C" __attribute__((visibility("hidden"))) int nid_hostsum_have_avx512(void);
extern "C" __attribute__((visibility("hidden"))) void nid_hostsum_jac_avx512(const double *rec, double err, double rho1, double *acc);
extern "C" __attribute__((visibility("hidden"))) int nid_hostsum_take_jac_avx512(double *rec, unsigned long long sentinel, int active, double err,
                                                                                  double rho1, double *acc);

#ifndef NID_DIRECT_AHEAD
#define NID_DIRECT_AHEAD 24
#endif
#ifndef NID_DIRECT_HINT
#define NID_DIRECT_HINT 3
#endif
constexpr int kDirectAhead = NID_DIRECT_AHEAD;
constexpr std::chrono::milliseconds kResidentPatience(2);      // a resident request unanswered for this long: fallback
constexpr std::chrono::milliseconds kResidentHostIdle(50);     // the host retires a kernel it has not used for this long ...
constexpr long long kResidentIdleTicks = 20000000;             // ... the kernel leaves by itself after 200 ms (100 MHz ticks)

int resident_fallback(nid_ctx *ctx, Slot &S);

int wait_direct(nid_ctx *ctx, Slot &S) {
  const int nloc = ctx->g.nloc, gs = ctx->group_size;
  const float dsqr = (float)(S.direct_delta * S.direct_delta);  // RobustKernelHuber::setDelta (robust_kernel_impl.h:84)
  const bool jac = S.direct_jac;
  static const bool trace = getenv("NID_DIRECT_TRACE") != nullptr;
  static const bool vec512 = nid_hostsum_have_avx512() != 0 && getenv("NID_DIRECT_SCALAR") == nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  std::vector<double> &hub = ctx->direct_rho1;  // per cell: rho1, err, active
  if (jac && hub.size() < (size_t)3 * nloc) hub.resize((size_t)3 * nloc);
  for (;;) {  // (a second round only after a resident kernel had to be replaced by an ordinary launch)
    double *base = S.resident ? ctx->res.rec_host : S.quad_host;
    // Two cursors over the cells, each in cell order.  c1: residual records (a cost + Jacobian kernel sends them
    // BEFORE its Jacobian phase, so this work -- a square root and a division per cell -- is done while the device is
    // still busy) -> Huber weights, chi2 (entry 0), count (entry 28).  c2 <= c1: Jacobian records -> b, H (entries
    // 1..27).  Every entry has its own chain of additions, in sum_blocks_w0's order, whichever cursor performs it:
    // per group the even-numbered cells in ascending order plus the odd-numbered ones, then the groups likewise.
    alignas(64) double top[2][32] = {}, grp[2][32] = {};
    double g00[2] = {0.0, 0.0}, g28[2] = {0.0, 0.0};
    int c1 = 0, c2 = jac ? 0 : nloc;
    unsigned long spins = 0, patience = 0;
    bool synced = false, again = false;
    auto t_first = t0, t_lastwait = t0;
    bool got_first = false, waited = false;
    while (c1 < nloc || c2 < nloc) {
      if (c1 < nloc) {
        double *rec = base + (size_t)c1 * kDirectRec;
        if (words_arrived(rec, kDirectRec)) {
          // every record is a line the device has just written, i.e. a miss to memory: ask for the lines ahead
          if (c1 + kDirectAhead < nloc) __builtin_prefetch(rec + (size_t)kDirectAhead * kDirectRec, 0, NID_DIRECT_HINT);
          __atomic_thread_fence(__ATOMIC_ACQUIRE);  // (the words were read as volatile; keep the plain reads below behind them)
          const int i = c1 % gs, gq = c1 / gs;
          double acc[32];
          acc[0] = g00[i & 1]; acc[28] = g28[i & 1];
          const double r1 = add_record_cost(rec, S.direct_delta, dsqr, acc);
          g00[i & 1] = acc[0]; g28[i & 1] = acc[28];
          if (jac) { double *w = &hub[3 * (size_t)c1]; w[0] = r1; w[1] = rec[0]; w[2] = rec[1]; }
          fill_sentinel(rec, kDirectRec);
          c1++;
          if (i == gs - 1 || c1 == nloc) {  // the group is complete
            top[gq & 1][0] += g00[0] + g00[1];
            top[gq & 1][28] += g28[0] + g28[1];
            g00[0] = g00[1] = g28[0] = g28[1] = 0.0;
          }
          if (trace && !jac) {
            if (!got_first) { t_first = std::chrono::steady_clock::now(); got_first = true; }
            if (waited) t_lastwait = std::chrono::steady_clock::now();
          }
          waited = false;
          continue;
        }
      }
      if (c2 < c1) {
        double *rec = base + (size_t)(nloc + c2) * kDirectRec;
        const int i = c2 % gs, gq = c2 / gs;
        const double *w = &hub[3 * (size_t)c2];
        bool took;
        if (vec512) {  // arrival test, sums and re-arming on one load of the line (nid_hostsum.cpp)
          took = nid_hostsum_take_jac_avx512(rec, kHostSentinel, w[2] != 0.0, w[1], w[0], grp[i & 1]) != 0;
          if (took && c2 + kDirectAhead < nloc) __builtin_prefetch(rec + (size_t)kDirectAhead * kDirectRec, 0, NID_DIRECT_HINT);
        } else {
          took = words_arrived(rec, kDirectRec);
          if (took) {
            if (c2 + kDirectAhead < nloc) __builtin_prefetch(rec + (size_t)kDirectAhead * kDirectRec, 0, NID_DIRECT_HINT);
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            if (w[2] != 0.0) add_record_jac(rec, w[1], w[0], grp[i & 1]);
            fill_sentinel(rec, kDirectRec);
          }
        }
        if (took) {
          c2++;
          if (i == gs - 1 || c2 == nloc) {
            double *t = top[gq & 1];
            for (int v = 1; v < 28; v++) t[v] += grp[0][v] + grp[1][v];
            std::memset(grp, 0, sizeof(grp));
          }
          if (trace) {
            if (!got_first) { t_first = std::chrono::steady_clock::now(); got_first = true; }
            if (waited) t_lastwait = std::chrono::steady_clock::now();
          }
          waited = false;
          continue;
        }
      }
      // nothing to do yet
      waited = true;
      if (S.resident) {
        // a resident kernel answers within microseconds; one that has left (its idle limit, a device-wide wait
        // elsewhere) never will: after kResidentPatience the request is re-issued as an ordinary DIRECT launch
        if ((++patience & 1023ul) == 0 && std::chrono::steady_clock::now() - t0 > kResidentPatience) {
          int rc = resident_fallback(ctx, S);
          if (rc) return rc;
          again = true;
          break;
        }
      } else if (++spins > 50000000ul) {  // let a device error surface through the runtime, then give up
        if (synced) { ctx->last_error = "a cell's results never arrived"; return NID_ERR_HIP; }
        NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
        NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
        NID_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));
        synced = true;
        spins = 0;
      }
    }
    if (again) continue;
    for (int v = 0; v < kReducedLen; v++) S.reduced_host[v] = v < 29 ? top[0][v] + top[1][v] : 0.0;
    if (trace) {
      const auto t1 = std::chrono::steady_clock::now();
      auto us = [&](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double, std::micro>(t - t0).count(); };
      g_dtrace.first += us(t_first); g_dtrace.last_wait += us(t_lastwait); g_dtrace.end += us(t1);
      if (++g_dtrace.n % 2000 == 0) {
        fprintf(stderr, "[direct trace] %ld waits: first %s record %.2f us, last wait %.2f us, end %.2f us after entering the wait\n", g_dtrace.n,
                jac ? "Jacobian" : "residual", g_dtrace.first / g_dtrace.n, g_dtrace.last_wait / g_dtrace.n, g_dtrace.end / g_dtrace.n);
        g_dtrace = DirectTrace();
      }
    }
    break;
  }
  if (S.resident) { ctx->res.pending_slot = -1; ctx->res.served++; ctx->res.fallback_run = 0; S.resident = false; }
  else S.quad_dirty = false;
  return NID_OK;
}

int wait_groups(nid_ctx *ctx, Slot &S) {
  const int ngroups = (ctx->g.nloc + ctx->group_size - 1) / ctx->group_size;
  alignas(64) double top[2][32] = {};
  unsigned long spins = 0;
  bool synced = false, waited = false;
  for (int gq = 0; gq < ngroups; gq++) {
    double *blk = S.groups_host + (size_t)gq * kQuad;
    int rc = spin_until(ctx, [&] { return words_arrived(blk, kQuad); }, spins, synced, waited);
    if (rc) return rc;
    if (gq + 2 < ngroups) __builtin_prefetch(blk + 2 * kQuad, 0, 3);
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    double *t = top[gq & 1];
    for (int v = 0; v < kQuad; v++) t[v] += blk[v];
    fill_sentinel(blk, kQuad);
  }
  for (int v = 0; v < kReducedLen; v++) S.reduced_host[v] = top[0][v] + top[1][v];
  S.groups_dirty = false;
  return NID_OK;
}

// the per-cell calls: every cell's kCellOut outputs, word by word
int wait_direct_cellout(nid_ctx *ctx, Slot &S) {
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    unsigned long spins = 0;
    bool synced = false, waited = false, again = false;
    for (int cl = 0; cl < ctx->g.nloc && !again; cl++) {
      const double *co = S.cellout_host + (size_t)cl * kCellOut;
      if (S.resident) {
        unsigned long n = 0;
        while (!words_arrived(co, kCellOut)) {
          if ((++n & 1023ul) == 0 && std::chrono::steady_clock::now() - t0 > kResidentPatience) {
            int rc = resident_fallback(ctx, S);
            if (rc) return rc;
            again = true;
            break;
          }
        }
      } else {
        int rc = spin_until(ctx, [&] { return words_arrived(co, kCellOut); }, spins, synced, waited);
        if (rc) return rc;
      }
    }
    if (!again) break;
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);
  if (S.resident) { ctx->res.pending_slot = -1; ctx->res.served++; ctx->res.fallback_run = 0; S.resident = false; }
  return NID_OK;
}

void fill_eval_params(nid_ctx *ctx, const Pose &pose, Slot &S, double delta, double *out_reduced,
                      unsigned long long *host_seq, EvalParams *P) {
  fill_common_params(ctx, delta, P);
  fill_slot_args(pose, S, out_reduced, host_seq, &P->slot[0]);
}

int check_ready(nid_ctx *ctx) {
  if (!ctx) return NID_ERR_INVALID_ARG;
  if (!ctx->have_ref || !ctx->have_target || !ctx->have_href) {
    ctx->last_error = "reference / target / href state not set";
    return NID_ERR_STATE;
  }
  return NID_OK;
}

// timing events of a slot (nid_enable_timing, nid_time_launches): created when first needed
int timing_events(nid_ctx *ctx, Slot &S) {
  if (!S.e0) NID_HIP(ctx, hipEventCreate(&S.e0));
  if (!S.e1) NID_HIP(ctx, hipEventCreate(&S.e1));
  return NID_OK;
}

// ---- the RESIDENT evaluator (k_resident): host side -- nid_capi_resident.inc, this translation unit ---
#include "nid_capi_resident.inc"

int launch_slot(nid_ctx *ctx, int slot, const Pose &pose, int want_jac, double delta,
                void *reduced_target) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  if (slot < 0 || slot >= NID_SLOTS) return NID_ERR_INVALID_ARG;
  Slot &S = ctx->slots[slot];
  if (S.pending) {  // its result has not been collected: a second launch would silently replace it
    ctx->last_error = "slot still pending: nid_wait() it first";
    return NID_ERR_STATE;
  }
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  S.seq++;
  S.external_target = reduced_target != nullptr;
  double *target = S.external_target ? static_cast<double *>(reduced_target) : S.reduced_host_devptr;
  unsigned long long *host_seq =
      S.external_target ? nullptr : reinterpret_cast<unsigned long long *>(S.reduced_host_devptr + kReducedLen);
  EvalParams P{};
  fill_eval_params(ctx, pose, S, delta, target, host_seq, &P);
  S.direct = S.resident = S.groups = false;
  if (!S.external_target && direct_ok(ctx)) {
    S.direct = true;
    S.direct_jac = want_jac != 0;
    S.direct_delta = delta;
    if (ctx->direct_mode != 2 && resident_usable(ctx) && resident_post(ctx, slot, pose, want_jac != 0, false) == NID_OK) {
      S.resident = true;  // no launch at all: the resident kernel has the request
      S.timed = false;
      S.done_slot = slot;
      S.pending = true;
      return NID_OK;
    }
    if (ctx->direct_mode == 2) {  // GROUP-DIRECT: the in-launch tail up to the group sums, those straight to the host
      rc = ensure_groups_host(ctx, S);
      if (rc) return rc;
      P.slot[0].out_reduced = S.groups_host_devptr;
      P.slot[0].host_seq = nullptr;
      P.slot[0].host_quad = 3;
      S.direct = false;
      S.groups = true;
      S.groups_dirty = true;
    } else {
      rc = ensure_quad_host(ctx, S);
      if (rc) return rc;
      P.slot[0].quad = S.quad_host_devptr;
      P.slot[0].host_quad = 1;
      S.quad_dirty = true;
    }
  }
  if (ctx->dbg_enabled) {
    const size_t N = (size_t)ctx->g.rows * ctx->g.cols;
    NID_HIP(ctx, hipMemsetAsync(ctx->dbg_u, 0xFF, N * 8, ctx->stream));
    NID_HIP(ctx, hipMemsetAsync(ctx->dbg_v, 0xFF, N * 8, ctx->stream));
    NID_HIP(ctx, hipMemsetAsync(ctx->dbg_ic, 0xFF, N * 8, ctx->stream));
    NID_HIP(ctx, hipMemsetAsync(ctx->dbg_wc, 0xFF, N * 32, ctx->stream));
    NID_HIP(ctx, hipMemsetAsync(ctx->dbg_jc, 0xFF, N * 4, ctx->stream));
  }
  // single-pose launches stay on one in-order stream (alternating streams per single-pose launch measured
  // 2.2x slower); the batched pipeline of nid_run_sequence is the one that alternates
  hipStream_t st = ctx->stream;
  S.timed = ctx->timing;
  if (S.timed) { rc = timing_events(ctx, S); if (rc) return rc; }
  if (S.timed) NID_HIP(ctx, hipEventRecord(S.e0, st));
  rc = launch_eval(ctx, P, want_jac != 0, st, 1, S.timed ? S.e1 : nullptr);  // (timed: e1 right behind k_eval2, in front of k_repair)
  if (rc) return rc;
  if (S.external_target) NID_HIP(ctx, hipEventRecord(S.done, st));
  S.done_slot = slot;
  S.pending = true;
  return NID_OK;
}

// n candidate poses in ONE launch (grid.y = n): each pose uses its own slot (buffers, tickets,
// pinned result block), so the reduction tails and the launch cost overlap with other poses' work
// and two workgroups share a CU.  Results are collected per slot with nid_wait().
int launch_batch(nid_ctx *ctx, int first_slot, int n, const Pose *poses, int want_jac, double delta,
                 double *reduced_dev_base, bool on_aux_stream, bool relaunch_ok, bool allow_direct) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  if (n < 1 || n > kMaxBatchExt || first_slot < 0 || first_slot + n > NID_SLOTS) return NID_ERR_INVALID_ARG;
  if (ctx->dbg_enabled) return NID_ERR_STATE;  // the per-pixel dump describes one pose
  if (!relaunch_ok)  // (nid_time_launches re-issues the same launch on one in-order stream on purpose)
    for (int k = 0; k < n; k++)
      if (ctx->slots[first_slot + k].pending) {
        ctx->last_error = "slot still pending: nid_wait() it first";
        return NID_ERR_STATE;
      }
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  EvalParams P{};
  fill_common_params(ctx, delta, &P);
  hipStream_t st = (on_aux_stream && !ctx->external_stream) ? ctx->aux_stream : ctx->stream;
  // up to kMaxBatch poses: the per-pose records are kernel arguments; beyond: a device array from the ring
  int ring = -1;
  SlotArgs *recs = P.slot;
  if (n > kMaxBatch) {
    ring = ctx->ext_next;
    ctx->ext_next = (ctx->ext_next + 1) % nid_ctx::kExtRing;
    if (ctx->ext_busy[ring]) {  // the launch that last used this entry must have consumed its records
      NID_HIP(ctx, hipEventSynchronize(ctx->ext_done[ring]));
      ctx->ext_busy[ring] = false;
    }
    recs = ctx->ext_host[ring];
    P.slots_ext = ctx->ext_dev[ring];
  }
  // (one awaited pose in GROUP-DIRECT mode, nid_set_direct_results(ctx, 2): the same form nid_launch takes)
  if (n == 1 && !reduced_dev_base && !relaunch_ok && allow_direct && !on_aux_stream && ctx->direct_mode == 2 && direct_ok(ctx))
    return launch_slot(ctx, first_slot, poses[0], want_jac, delta, nullptr);
  const bool direct = n == 1 && !reduced_dev_base && !relaunch_ok && allow_direct && direct_ok(ctx);
  if (direct && !on_aux_stream && resident_usable(ctx) && resident_post(ctx, first_slot, poses[0], want_jac != 0, false) == NID_OK) {
    Slot &S = ctx->slots[first_slot];
    S.seq++;
    S.external_target = false;
    S.direct = S.resident = true;
    S.direct_jac = want_jac != 0;
    S.direct_delta = delta;
    S.timed = false;
    S.done_slot = first_slot;
    S.pending = true;
    return NID_OK;
  }
  if (direct) { rc = ensure_quad_host(ctx, ctx->slots[first_slot]); if (rc) return rc; }
  for (int k = 0; k < n; k++) {
    Slot &S = ctx->slots[first_slot + k];
    S.seq++;
    S.direct = S.resident = false;
    S.external_target = reduced_dev_base != nullptr;
    if (S.external_target) {  // caller-owned device buffer: pose k's block at base + k*32 (the pipelined loops, multi-GPU all-reduce)
      fill_slot_args(poses[k], S, reduced_dev_base + (size_t)k * kReducedLen, nullptr, &recs[k]);
      if (n > kMaxBatch) recs[k].cellout = nullptr;  // nobody reads the per-cell outputs of such a launch
    }
    else
      fill_slot_args(poses[k], S, S.reduced_host_devptr,
                     reinterpret_cast<unsigned long long *>(S.reduced_host_devptr + kReducedLen), &recs[k]);
    if (direct) {
      recs[k].quad = S.quad_host_devptr;
      recs[k].host_quad = 1;
      S.quad_dirty = true;
      S.direct = true;
      S.direct_jac = want_jac != 0;
      S.direct_delta = delta;
    }
  }
  if (ring >= 0)
    NID_HIP(ctx, hipMemcpyAsync(ctx->ext_dev[ring], recs, (size_t)n * sizeof(SlotArgs), hipMemcpyHostToDevice, st));
  Slot &S0 = ctx->slots[first_slot];
  S0.timed = ctx->timing;
  if (S0.timed) { rc = timing_events(ctx, S0); if (rc) return rc; }
  if (S0.timed) NID_HIP(ctx, hipEventRecord(S0.e0, st));
  rc = launch_eval(ctx, P, want_jac != 0, st, n, S0.timed ? S0.e1 : nullptr);  // (timed: e1 right behind k_eval2, in front of k_repair)
  if (rc) return rc;
  if (ring >= 0) {
    NID_HIP(ctx, hipEventRecord(ctx->ext_done[ring], st));
    ctx->ext_busy[ring] = true;
  }
  // caller-owned result buffer: the launch is over when its ONE event is (recorded on the batch's first slot; the
  // other slots point at it -- an event record per pose cost ~4 us each, 1 ms of host time per 256-pose launch)
  if (S0.external_target) NID_HIP(ctx, hipEventRecord(S0.done, st));
  for (int k = 0; k < n; k++) {
    Slot &S = ctx->slots[first_slot + k];
    S.pending = true;
    S.done_slot = first_slot;
    if (k) S.timed = false;
  }
  return NID_OK;
}

// SHORT sequences (round 4): n poses whose results the host collects slot by slot, n too small to fill a pipeline --
// the candidates of a Gauss-Newton / LM step, the driver's `--steps 20`.  One launch of n poses is not the fastest
// form: beyond kMaxBatch poses the per-pose records travel by an in-stream copy (a second enqueue and a dependency in
// front of the kernel), and a launch's workgroups start together and walk through their phases in step (load, atomics,
// fold, Jacobian), so that the chip's units take turns idling.  Launches of <= kMaxBatch poses carry their records in
// the kernel arguments, and two of them on the context's two streams run beside each other out of step.  The plan:
// poses per launch and whether consecutive launches alternate between the streams; from the measured table
// (profiles/r04_short_sequences.txt, tools/short_seq_sweep.py) unless nid_set_short_sequence_policy pins it.
// Results do not depend on the plan (a pose gives the same bits alone, in any launch, on either stream: tested).
struct SplitPlan { int chunk; bool two_streams; };
constexpr int kShortSequence = 64;  // longer sequences are one launch (<= NID_MAX_BATCH) or the pipeline (nid_run_sequence)

SplitPlan plan_split(const nid_ctx *ctx, int n, bool jac) {
  if (n > kShortSequence) return {n, false};
  if (ctx->seq_chunk > 0) return {std::min(ctx->seq_chunk, n), ctx->seq_streams != 1};
  if (ctx->external_stream) return {n, false};  // one stream: nothing to run beside
  (void)jac;
  // up to ten poses one launch fills the chip once (10 x 256 cells = one workgroup slot per cell and pose on 256 CUs): a second
  // launch on the other stream only adds its own latency (profiles/r05_short_sequences.txt: 8 poses 52.6 us against 57.1,
  // cost only 38.2 against 47.5; 10 poses 57.6 against 59.8)
  if (n <= 10) return {n, false};
  // as few launches of <= kMaxBatch poses as cover n, at least two, equally filled: 20 -> 10 + 10, 40 -> 14 + 13 + 13
  const int launches = std::max(2, (n + kMaxBatch - 1) / kMaxBatch);
  return {(n + launches - 1) / launches, true};
}

int launch_split(nid_ctx *ctx, int first_slot, int n, const Pose *poses, int want_jac, double delta) {
  if (!ctx) return NID_ERR_INVALID_ARG;
  if (n < 1 || n > kMaxBatchExt || first_slot < 0 || first_slot + n > NID_SLOTS) return NID_ERR_INVALID_ARG;
  const SplitPlan plan = plan_split(ctx, n, want_jac != 0);
  if (plan.chunk >= n || ctx->dbg_enabled || ctx->timing) return launch_batch(ctx, first_slot, n, poses, want_jac, delta);
  for (int k = 0; k < n; k++)  // all or nothing: no half-launched sequence
    if (ctx->slots[first_slot + k].pending) { ctx->last_error = "slot still pending: nid_wait() it first"; return NID_ERR_STATE; }
  int l = 0;
  for (int i = 0; i < n; i += plan.chunk, l++) {
    const int cnt = std::min(plan.chunk, n - i);
    int rc = launch_batch(ctx, first_slot + i, cnt, poses + i, want_jac, delta, nullptr, plan.two_streams && (l & 1), false, /*allow_direct=*/false);
    if (rc) {
      // all or nothing: the chunks that did go out are waited for and their slots released -- the caller gets the error
      // and no slot of this sequence is left pending (it cannot know which ones were)
      const std::string why = ctx->last_error;
      for (int k = 0; k < i; k++) {
        Slot &S = ctx->slots[first_slot + k];
        if (S.pending && wait_host_seq(ctx, S) == NID_OK) S.pending = false;
      }
      ctx->last_error = why;
      return rc;
    }
  }
  return NID_OK;
}

// spin on the sequence word the last workgroup of slot S's launch writes behind its results (system-scope release)
int wait_host_seq(nid_ctx *ctx, Slot &S) {
  volatile unsigned long long *seqw = reinterpret_cast<volatile unsigned long long *>(S.reduced_host + kReducedLen);
  unsigned long spins = 0;
  while (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != S.seq) {
    if (++spins > 20000000ul) {  // fall back to the runtime so that a device error surfaces
      NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
      NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
      NID_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));
      if (__atomic_load_n(seqw, __ATOMIC_ACQUIRE) != S.seq) {
        ctx->last_error = "result sequence word never arrived";
        return NID_ERR_HIP;
      }
      break;
    }
  }
  return NID_OK;
}

int evaluate_common(nid_ctx *ctx, const Pose &pose, int want_jac, double *Ht, double *Hj,
                    double *err, double *der) {
  int rc = check_ready(ctx);
  if (rc) return rc;
  Slot &S = ctx->slots[0];
  if (S.pending) {  // the blocking calls use slot 0's buffers
    ctx->last_error = "slot 0 has an uncollected launch: nid_wait(ctx, 0, ...) first";
    return NID_ERR_STATE;
  }
  // The kernel writes the per-cell outputs straight into mapped pinned host memory and the last workgroup posts the
  // slot's sequence word behind them (as for the fused 6x6 path): no copy, no stream synchronisation -- the call
  // returns ~6 us after the kernel ends instead of ~18.
  S.seq++;
  S.external_target = false;
  EvalParams P{};
  fill_eval_params(ctx, pose, S, std::sqrt(0.95), S.reduced_host_devptr,
                   reinterpret_cast<unsigned long long *>(S.reduced_host_devptr + kReducedLen), &P);
  P.slot[0].cellout = S.cellout_host_devptr;
  P.slot[0].cellout_host = 1;
  const bool direct = direct_ok(ctx);
  S.resident = false;
  if (direct) {  // the per-cell outputs straight to the host, word by word, and nothing else (wait_direct_cellout)
    fill_sentinel(S.cellout_host, (size_t)ctx->g.nloc * kCellOut);
    P.slot[0].host_quad = 2;
    if (resident_usable(ctx) && resident_post(ctx, 0, pose, want_jac != 0, true) == NID_OK) S.resident = true;
  }
  if (ctx->dbg_enabled) {
    const size_t N = (size_t)ctx->g.rows * ctx->g.cols;
    NID_HIP(ctx, hipMemsetAsync(ctx->dbg_u, 0xFF, N * 8, ctx->stream));
    NID_HIP(ctx, hipMemsetAsync(ctx->dbg_v, 0xFF, N * 8, ctx->stream));
    NID_HIP(ctx, hipMemsetAsync(ctx->dbg_ic, 0xFF, N * 8, ctx->stream));
    NID_HIP(ctx, hipMemsetAsync(ctx->dbg_wc, 0xFF, N * 32, ctx->stream));
    NID_HIP(ctx, hipMemsetAsync(ctx->dbg_jc, 0xFF, N * 4, ctx->stream));
  }
  if (!S.resident) {
    rc = launch_eval(ctx, P, want_jac != 0, ctx->stream);
    if (rc) return rc;
  }
  rc = direct ? wait_direct_cellout(ctx, S) : wait_host_seq(ctx, S);
  if (rc) return rc;
  if (ctx->dbg_enabled) NID_HIP(ctx, hipStreamSynchronize(ctx->stream));  // the dump is read back by the runtime
  for (int cl = 0; cl < ctx->g.nloc; cl++) {
    const double *o = S.cellout_host + (size_t)cl * kCellOut;
    const int c = ctx->g.cell_begin + cl * ctx->g.cell_stride;
    if (Ht) Ht[c] = o[0];
    if (Hj) Hj[c] = o[1];
    if (err) err[c] = o[2];
    if (der && want_jac) for (int n = 0; n < 6; n++) der[6 * c + n] = o[3 + n];
  }
  return NID_OK;
}

constexpr int kBsParts = 4;

int href_common(nid_ctx *ctx, const Pose &pose, int32_t *bs_counter, double *Href, double *bs_value,
                int32_t *bs_index) {
  if (!ctx) return NID_ERR_INVALID_ARG;
  if (!ctx->have_ref) { ctx->last_error = "reference not set"; return NID_ERR_STATE; }
  resident_retire(ctx);
  hipLaunchKernelGGL((k_href<256>), dim3(ctx->g.nloc), dim3(256), 0, ctx->stream, ctx->g, pose, ctx->t,
                     ctx->Nc_dev, ctx->Href_dev, ctx->hist_scale, ctx->hist_inv_scale);
  NID_HIP(ctx, hipGetLastError());
  const int nloc = ctx->g.nloc;
  std::vector<int> nc(nloc);
  std::vector<double> hr(nloc);
  NID_HIP(ctx, hipMemcpyAsync(nc.data(), ctx->Nc_dev, nloc * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  NID_HIP(ctx, hipMemcpyAsync(hr.data(), ctx->Href_dev, nloc * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
  for (int cl = 0; cl < nloc; cl++) {
    if (bs_counter) bs_counter[ctx->g.cell_begin + cl * ctx->g.cell_stride] = nc[cl];
    if (Href) Href[ctx->g.cell_begin + cl * ctx->g.cell_stride] = hr[cl];
  }
  if (bs_value || bs_index) {
    // The per-pixel outputs are put into the caller's layout ON THE DEVICE (k_untile_bs), come home through pinned
    // staging and are copied out row by row (whole, when this context owns every pixel): a pageable copy of the tile
    // plus a scalar scatter loop on the host was 1.4 ms of every frame pair (profiles/r04_pair_setup.txt).
    const Geometry &g = ctx->g;
    const size_t N = (size_t)g.rows * g.cols;
    if (bs_value && (!ctx->bsv_img_dev || !ctx->bsv_stage)) {  // (both or neither: a half-made pair from a failed call is completed)
      if (!ctx->bsv_img_dev) { int rc = dev_alloc(ctx, &ctx->bsv_img_dev, 4 * N); if (rc) return rc; }
      if (!ctx->bsv_stage && hipHostMalloc(reinterpret_cast<void **>(&ctx->bsv_stage), 4 * N * sizeof(double), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError(); ctx->bsv_stage = nullptr; return NID_ERR_NOMEM;
      }
    }
    if (bs_index && (!ctx->bsi_img_dev || !ctx->bsi_stage)) {
      if (!ctx->bsi_img_dev) { int rc = dev_alloc(ctx, &ctx->bsi_img_dev, N); if (rc) return rc; }
      if (!ctx->bsi_stage && hipHostMalloc(reinterpret_cast<void **>(&ctx->bsi_stage), N * sizeof(int), hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError(); ctx->bsi_stage = nullptr; return NID_ERR_NOMEM;
      }
    }
    const long total = (long)g.nloc * g.pstride;
    hipLaunchKernelGGL(k_untile_bs, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, g, ctx->t,
                       bs_value ? ctx->bsv_img_dev : nullptr, bs_index ? ctx->bsi_img_dev : nullptr, ctx->href_nan_rows ? 1 : 0);
    NID_HIP(ctx, hipGetLastError());
    const bool whole = g.cell_stride == 1 && g.nloc == g.cell_num * g.cell_num && g.rb * g.cell_num == g.rows && g.cb * g.cell_num == g.cols;
    if (whole && bs_value) {
      // bs_value (9.8 MB at 640x480) comes home in kBsParts pieces, each copied out to the caller while the next is
      // still crossing PCIe (the copy-out is as long as the transfer: one after the other they were 0.5 ms)
      for (int p = 0; p < kBsParts; p++) if (!ctx->bs_part_done[p]) NID_HIP(ctx, hipEventCreateWithFlags(&ctx->bs_part_done[p], hipEventDisableTiming));
      const size_t part = ((4 * N / kBsParts) + 7) & ~(size_t)7;
      for (int p = 0; p < kBsParts; p++) {
        const size_t lo = std::min(4 * N, part * p), hi = p == kBsParts - 1 ? 4 * N : std::min(4 * N, part * (p + 1));
        if (hi > lo) NID_HIP(ctx, hipMemcpyAsync(ctx->bsv_stage + lo, ctx->bsv_img_dev + lo, (hi - lo) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        NID_HIP(ctx, hipEventRecord(ctx->bs_part_done[p], ctx->stream));
      }
      if (bs_index) NID_HIP(ctx, hipMemcpyAsync(ctx->bsi_stage, ctx->bsi_img_dev, N * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
      for (int p = 0; p < kBsParts; p++) {
        const size_t lo = std::min(4 * N, part * p), hi = p == kBsParts - 1 ? 4 * N : std::min(4 * N, part * (p + 1));
        NID_HIP(ctx, hipEventSynchronize(ctx->bs_part_done[p]));
        if (hi > lo) std::memcpy(bs_value + lo, ctx->bsv_stage + lo, (hi - lo) * sizeof(double));
      }
      NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
      if (bs_index) std::memcpy(bs_index, ctx->bsi_stage, N * sizeof(int));
      ctx->have_href = true;
      return NID_OK;
    }
    if (bs_value) NID_HIP(ctx, hipMemcpyAsync(ctx->bsv_stage, ctx->bsv_img_dev, 4 * N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (bs_index) NID_HIP(ctx, hipMemcpyAsync(ctx->bsi_stage, ctx->bsi_img_dev, N * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (whole) {
      if (bs_value) std::memcpy(bs_value, ctx->bsv_stage, 4 * N * sizeof(double));
      if (bs_index) std::memcpy(bs_index, ctx->bsi_stage, N * sizeof(int));
    } else {
      for (int cl = 0; cl < g.nloc; cl++) {
        const int c = g.cell_begin + cl * g.cell_stride, ci = c / g.cell_num, cj = c % g.cell_num;
        for (int r = 0; r < g.rb; r++) {
          const size_t id = (size_t)(ci * g.rb + r) * g.cols + (size_t)cj * g.cb;
          if (bs_value) std::memcpy(bs_value + 4 * id, ctx->bsv_stage + 4 * id, 4 * (size_t)g.cb * sizeof(double));
          if (bs_index) std::memcpy(bs_index + id, ctx->bsi_stage + id, (size_t)g.cb * sizeof(int));
        }
      }
    }
  }
  ctx->have_href = true;
  return NID_OK;
}

int upload_tiles(nid_ctx *ctx, const double *depth_m, const double *points3d, const uint8_t *im0,
                 const double *Twc) {
  const Geometry &g = ctx->g;
  const size_t N = (size_t)g.rows * g.cols;
  resident_retire(ctx);
  NID_HIP(ctx, hipMemcpyAsync(ctx->im0_dev, im0, N, hipMemcpyHostToDevice, ctx->stream));
  const double *depth_dev = nullptr, *points_dev = nullptr;
  if (depth_m) {
    NID_HIP(ctx, hipMemcpyAsync(ctx->depth_dev, depth_m, N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    NID_HIP(ctx, hipMemcpyAsync(ctx->Twc_dev, Twc, 16 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    depth_dev = ctx->depth_dev;
  } else {
    if (!ctx->points_dev) { int rc = dev_alloc(ctx, &ctx->points_dev, 3 * N); if (rc) return rc; }
    NID_HIP(ctx, hipMemcpyAsync(ctx->points_dev, points3d, 3 * N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    points_dev = ctx->points_dev;
  }
  const long total = (long)g.nloc * g.pstride;
  hipLaunchKernelGGL(k_tile, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, g, depth_dev,
                     points_dev, ctx->im0_dev, ctx->Twc_dev, ctx->t, (double *)nullptr);
  NID_HIP(ctx, hipGetLastError());
  NID_HIP(ctx, hipStreamSynchronize(ctx->stream));  // host source buffers may be released by the caller
  ctx->have_ref = true;
  ctx->have_href = false;
  ctx->ref_from_depth = depth_m != nullptr;
  return NID_OK;
}

// One frame pair in the driver's own formats, handed over in one go (nid_set_pair_u16; the multi-GPU layer enqueues on
// every shard before it collects from any).  Everything on ctx->stream: three uploads from ONE pinned block (depth u16,
// im0, im1: 1.2 MB at 640x480 against 2.5 + 7.4 + 9.8 MB each way on the legacy operators' route), depth -> metres,
// back-projection + tiling, the target's margins, the reference stage at pose0, counts and Href back into the block.
struct PairStage { size_t depth, im0, im1, twc, nc, href, bytes; };
PairStage pair_stage_layout(const Geometry &g) {
  const size_t N = (size_t)g.rows * g.cols;
  auto up = [](size_t v) { return (v + 63) & ~(size_t)63; };
  PairStage L;
  L.depth = 0; L.im0 = up(2 * N); L.im1 = up(L.im0 + N); L.twc = up(L.im1 + N);
  L.nc = up(L.twc + 16 * sizeof(double)); L.href = up(L.nc + (size_t)g.nloc * sizeof(int));
  L.bytes = up(L.href + (size_t)g.nloc * sizeof(double));
  return L;
}

int pair_u16_enqueue(nid_ctx *ctx, const uint16_t *depth_u16, double depth_factor, const uint8_t *im0, const uint8_t *im1,
                     const double *Twc, const Pose &pose0) {
  const Geometry &g = ctx->g;
  const size_t N = (size_t)g.rows * g.cols;
  const PairStage L = pair_stage_layout(g);
  resident_retire(ctx);
  if (!ctx->depth16_dev) NID_HIP(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->depth16_dev), N * sizeof(uint16_t)));
  if (!ctx->pair_stage && hipHostMalloc(reinterpret_cast<void **>(&ctx->pair_stage), L.bytes, hipHostMallocDefault) != hipSuccess) {
    (void)hipGetLastError(); ctx->pair_stage = nullptr; return NID_ERR_NOMEM;
  }
  uint8_t *st = ctx->pair_stage;
  std::memcpy(st + L.depth, depth_u16, 2 * N);
  std::memcpy(st + L.im0, im0, N);
  std::memcpy(st + L.im1, im1, N);
  std::memcpy(st + L.twc, Twc, 16 * sizeof(double));
  NID_HIP(ctx, hipMemcpyAsync(ctx->depth16_dev, st + L.depth, 2 * N, hipMemcpyHostToDevice, ctx->stream));
  NID_HIP(ctx, hipMemcpyAsync(ctx->im0_dev, st + L.im0, N, hipMemcpyHostToDevice, ctx->stream));
  NID_HIP(ctx, hipMemcpyAsync(ctx->im1_dev, st + L.im1, N, hipMemcpyHostToDevice, ctx->stream));
  NID_HIP(ctx, hipMemcpyAsync(ctx->Twc_dev, st + L.twc, 16 * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
  hipLaunchKernelGGL(k_depth_u16, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream, (long)N, ctx->depth16_dev, depth_factor, ctx->depth_dev);
  const long total = (long)g.nloc * g.pstride;
  hipLaunchKernelGGL(k_tile, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, g, ctx->depth_dev,
                     (const double *)nullptr, ctx->im0_dev, ctx->Twc_dev, ctx->t, (double *)nullptr);
  const long mtotal = (long)(g.rows + 1) * (g.cols + 1);
  hipLaunchKernelGGL(k_im1_margins, dim3((unsigned)((mtotal + 255) / 256)), dim3(256), 0, ctx->stream, g.rows, g.cols,
                     ctx->im1_stride, ctx->im1_dev, ctx->im1s_dev);
  hipLaunchKernelGGL((k_href<256>), dim3(g.nloc), dim3(256), 0, ctx->stream, g, pose0, ctx->t, ctx->Nc_dev, ctx->Href_dev,
                     ctx->hist_scale, ctx->hist_inv_scale);
  NID_HIP(ctx, hipGetLastError());
  NID_HIP(ctx, hipMemcpyAsync(st + L.nc, ctx->Nc_dev, (size_t)g.nloc * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  NID_HIP(ctx, hipMemcpyAsync(st + L.href, ctx->Href_dev, (size_t)g.nloc * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  ctx->have_ref = ctx->have_target = ctx->have_href = false;  // until pair_u16_collect has seen the stream drain
  return NID_OK;
}

int pair_u16_collect(nid_ctx *ctx, int32_t *bs_counter, double *Href) {
  const Geometry &g = ctx->g;
  const PairStage L = pair_stage_layout(g);
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
  const int *nc = reinterpret_cast<const int *>(ctx->pair_stage + L.nc);
  const double *hr = reinterpret_cast<const double *>(ctx->pair_stage + L.href);
  for (int cl = 0; cl < g.nloc; cl++) {
    if (bs_counter) bs_counter[g.cell_begin + cl * g.cell_stride] = nc[cl];
    if (Href) Href[g.cell_begin + cl * g.cell_stride] = hr[cl];
  }
  ctx->have_ref = ctx->have_target = ctx->have_href = true;
  ctx->ref_from_depth = true;
  return NID_OK;
}

}  // namespace

// ===========================================================================
extern "C" {

int nid_abi_version(void) { return NID_ABI_VERSION; }

const char *nid_status_string(int s) {
  switch (s) {
    case NID_OK: return "ok";
    case NID_ERR_INVALID_ARG: return "invalid argument";
    case NID_ERR_NO_DEVICE: return "no HIP device";
    case NID_ERR_HIP: return "HIP runtime error";
    case NID_ERR_UNSUPPORTED: return "unsupported shape";
    case NID_ERR_STATE: return "call order: reference/target/href state missing";
    case NID_ERR_NOMEM: return "out of memory";
    default: return "unknown status";
  }
}

const char *nid_last_error(const nid_ctx *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int nid_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int nid_create(const nid_config *cfg, nid_ctx **out) { return nid_create_strided(cfg, 1, out); }

int nid_create_strided(const nid_config *cfg, int32_t cell_stride, nid_ctx **out) {
  if (!cfg || !out || cell_stride < 1) return NID_ERR_INVALID_ARG;
  *out = nullptr;
  if (cfg->bs_degree != 3 || cfg->bin_num < 4 || cfg->bin_num > kMaxBins || cfg->cell_num < 1 ||
      cfg->rows < cfg->cell_num || cfg->cols < cfg->cell_num)
    return NID_ERR_INVALID_ARG;
  const int ncell = cfg->cell_num * cfg->cell_num;
  int cb_ = cfg->cell_begin, ce_ = cfg->cell_end;
  if (cb_ == 0 && ce_ == 0) ce_ = ncell;
  if (cb_ < 0 || ce_ > ncell || cb_ >= ce_) return NID_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NID_ERR_NO_DEVICE;
  if (cfg->device < 0 || cfg->device >= ndev) return NID_ERR_INVALID_ARG;
  nid_ctx *ctx = new (std::nothrow) nid_ctx();
  if (!ctx) return NID_ERR_NOMEM;
  ctx->cfg = *cfg;
  Geometry &g = ctx->g;
  g.rows = cfg->rows; g.cols = cfg->cols; g.cell_num = cfg->cell_num;
  g.rb = cfg->rows / cfg->cell_num; g.cb = cfg->cols / cfg->cell_num;
  g.ps = g.rb * g.cb; g.pstride = (g.ps + 63) & ~63;
  g.cell_begin = cb_; g.cell_stride = cell_stride; g.nloc = (ce_ - cb_ + cell_stride - 1) / cell_stride;
  g.nb = cfg->bin_num; g.S = cfg->bin_num - 3;
  g.fx = cfg->fx; g.fy = cfg->fy; g.cx = cfg->cx; g.cy = cfg->cy;
  if (g.ps > (1 << 20)) { delete ctx; return NID_ERR_UNSUPPORTED; }  // fixed-point head-room of the histograms
  int bits = 0;
  while ((1 << bits) < g.ps + 1) bits++;
  // whole-cell sums stay below 2^62 and every single weight (<= 1) below the 2^52 of the magic-number encode
  const int hs = std::min(62 - bits, 51);
  ctx->hist_scale = std::ldexp(1.0, hs);
  ctx->hist_inv_scale = std::ldexp(1.0, -hs);
  // 32-bit byte offsets into the reference weights (load_tile_w): 4 x 8 B x nloc x pstride < 2^32
  if ((size_t)g.nloc * (size_t)g.pstride >= ((size_t)1 << 27)) { delete ctx; return NID_ERR_UNSUPPORTED; }
  if (g.nloc >= (1 << 16)) { delete ctx; return NID_ERR_UNSUPPORTED; }  // a repair-queue entry is pose << 16 | cell (k_repair)
  if (const char *bt = getenv("NID_BLOCK_THREADS")) {  // tuning: both kinds of launches
    const int v = atoi(bt);
    if (v == 128 || v == 256 || v == 512 || v == 1024) ctx->jac_threads = ctx->cost_threads = v;
  }
  auto fail = [&](int rc) { nid_destroy(ctx); return rc; };
  if (hipSetDevice(cfg->device) != hipSuccess) return fail(NID_ERR_NO_DEVICE);
  if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) return fail(NID_ERR_HIP);
  if (hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking) != hipSuccess) return fail(NID_ERR_HIP);
  ctx->stream = ctx->own_stream;
  const size_t N = (size_t)g.rows * g.cols;
  const size_t plane = (size_t)g.nloc * g.pstride;
  // two-level reduction geometry: ~sqrt(nloc) cells per group
  ctx->group_size = std::max(1, (int)std::ceil(std::sqrt((double)g.nloc)));
  ctx->ngroups = (g.nloc + ctx->group_size - 1) / ctx->group_size;
  int rc;
  if ((rc = dev_alloc(ctx, &ctx->t.X, plane))) return fail(rc);
  if ((rc = dev_alloc(ctx, &ctx->t.Y, plane))) return fail(rc);
  if ((rc = dev_alloc(ctx, &ctx->t.Z, plane))) return fail(rc);
  if ((rc = dev_alloc(ctx, &ctx->t.W, 4 * plane))) return fail(rc);
  if ((rc = dev_alloc(ctx, &ctx->t.JR, plane))) return fail(rc);
  if ((rc = dev_alloc(ctx, &ctx->t.I0, plane))) return fail(rc);
  if ((rc = dev_alloc(ctx, &ctx->im1_dev, N + 64))) return fail(rc);
  {
    // int16 image with margins: (rows + 1) rows of `stride` elements, plus slack for the windows of lanes
    // without a sample (origin (0,0): rows 0..3) on images of fewer than 4 rows
    ctx->im1_stride = (g.cols + 1 + 3) & ~3;
    const size_t n16 = (size_t)(g.rows + 5) * ctx->im1_stride + 64;
    if ((rc = dev_alloc(ctx, &ctx->im1s_dev, n16))) return fail(rc);
    if (hipMemset(ctx->im1s_dev, 0, n16 * sizeof(int16_t)) != hipSuccess) return fail(NID_ERR_HIP);
  }
  if ((rc = dev_alloc(ctx, &ctx->im0_dev, N))) return fail(rc);
  if ((rc = dev_alloc(ctx, &ctx->depth_dev, N))) return fail(rc);
  if ((rc = dev_alloc(ctx, &ctx->Twc_dev, 16))) return fail(rc);
  if ((rc = dev_alloc(ctx, &ctx->Nc_dev, g.nloc))) return fail(rc);
  if ((rc = dev_alloc(ctx, &ctx->Href_dev, g.nloc))) return fail(rc);
  if ((rc = dev_alloc(ctx, &ctx->repair_count_dev, 1))) return fail(rc);
  if (hipMemset(ctx->repair_count_dev, 0, sizeof(unsigned long long)) != hipSuccess) return fail(NID_ERR_HIP);
  for (int q = 0; q < 2; q++) {  // one queue per launch stream: [count | exit ticket | (pose << 16 | cell, repair set x 2) ...], see k_repair
    const size_t n = 2 + 3 * (size_t)g.nloc * kMaxBatchExt;
    if ((rc = dev_alloc(ctx, &ctx->repair_queue_dev[q], n))) return fail(rc);
    if (hipMemset(ctx->repair_queue_dev[q], 0, n * sizeof(unsigned)) != hipSuccess) return fail(NID_ERR_HIP);
  }
  {
    // the evaluation kernels read the table with kWcPre on its value coefficients (k_eval2's hist_add, fx_bits)
    std::vector<double> coef;
    build_coef_table(g.S, &coef);
    for (size_t i = 0; i < coef.size(); i++) if ((i % 7) < 4) coef[i] *= kWcPre;
    if ((rc = dev_alloc(ctx, &ctx->ctab_dev, coef.size()))) return fail(rc);
    if (hipMemcpy(ctx->ctab_dev, coef.data(), coef.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
      return fail(NID_ERR_HIP);
  }
  {
    // Per-slot buffers come out of ONE allocation per kind (NID_SLOTS x 5 hipMalloc + 2 hipHostMalloc calls used to be
    // most of the context's creation time); the slots hold pointers into the slabs.
    const size_t n_cellout = (size_t)g.nloc * kCellOut, n_quad = (size_t)g.nloc * kQuad;
    const size_t n_ticket = ((size_t)ctx->ngroups + 4 + 3) & ~(size_t)3, n_gpart = (size_t)ctx->ngroups * kQuad;
    const size_t n_rhost = kReducedLen + 2;  // [32 doubles][u64 sequence word][pad]
    if ((rc = dev_alloc(ctx, &ctx->slab_cellout, n_cellout * NID_SLOTS))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->slab_reduced, (size_t)kReducedLen * NID_SLOTS))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->slab_quad, n_quad * NID_SLOTS))) return fail(rc);
    if ((rc = dev_alloc(ctx, &ctx->slab_ticket, n_ticket * NID_SLOTS))) return fail(rc);
    if (hipMemset(ctx->slab_ticket, 0, n_ticket * NID_SLOTS * sizeof(unsigned)) != hipSuccess) return fail(NID_ERR_HIP);
    if ((rc = dev_alloc(ctx, &ctx->slab_gpart, n_gpart * NID_SLOTS))) return fail(rc);
    if (hipHostMalloc(reinterpret_cast<void **>(&ctx->slab_reduced_host), n_rhost * NID_SLOTS * sizeof(double),
                      hipHostMallocMapped) != hipSuccess) return fail(NID_ERR_NOMEM);
    std::memset(ctx->slab_reduced_host, 0, n_rhost * NID_SLOTS * sizeof(double));
    double *rhost_dev = nullptr;
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&rhost_dev), ctx->slab_reduced_host, 0) != hipSuccess) return fail(NID_ERR_HIP);
    for (int s = 0; s < NID_SLOTS; s++) {
      Slot &S = ctx->slots[s];
      S.cellout_dev = ctx->slab_cellout + n_cellout * s;
      S.reduced_dev = ctx->slab_reduced + (size_t)kReducedLen * s;
      S.quad_dev = ctx->slab_quad + n_quad * s;
      S.ticket_dev = ctx->slab_ticket + n_ticket * s;
      S.gpart_dev = ctx->slab_gpart + n_gpart * s;
      S.reduced_host = ctx->slab_reduced_host + n_rhost * s;
      S.reduced_host_devptr = rhost_dev + n_rhost * s;
      if (hipEventCreateWithFlags(&S.done, hipEventDisableTiming) != hipSuccess) return fail(NID_ERR_HIP);
    }
    // the blocking per-cell calls use slot 0 only: its per-cell outputs can go straight to mapped pinned memory
    Slot &S0 = ctx->slots[0];
    if (hipHostMalloc(reinterpret_cast<void **>(&S0.cellout_host), n_cellout * sizeof(double), hipHostMallocMapped) != hipSuccess)
      return fail(NID_ERR_NOMEM);
    if (hipHostGetDevicePointer(reinterpret_cast<void **>(&S0.cellout_host_devptr), S0.cellout_host, 0) != hipSuccess)
      return fail(NID_ERR_HIP);
  }
  for (int r = 0; r < nid_ctx::kExtRing; r++) {
    if ((rc = dev_alloc(ctx, &ctx->ext_dev[r], (size_t)kMaxBatchExt))) return fail(rc);
    if (hipHostMalloc(reinterpret_cast<void **>(&ctx->ext_host[r]), (size_t)kMaxBatchExt * sizeof(SlotArgs),
                      hipHostMallocDefault) != hipSuccess) return fail(NID_ERR_NOMEM);
    if (hipEventCreateWithFlags(&ctx->ext_done[r], hipEventDisableTiming) != hipSuccess) return fail(NID_ERR_HIP);
  }
  *out = ctx;
  return NID_OK;
}

int nid_destroy(nid_ctx *ctx) {
  if (!ctx) return NID_OK;
  (void)hipSetDevice(ctx->cfg.device);
  resident_retire(ctx);
  if (ctx->res.stream) (void)hipStreamDestroy(ctx->res.stream);
  if (ctx->res.ctl) (void)hipFree(ctx->res.ctl);
  if (ctx->res.rec_host) (void)hipHostFree(ctx->res.rec_host);
  if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
  (void)hipFree(ctx->t.X); (void)hipFree(ctx->t.Y); (void)hipFree(ctx->t.Z); (void)hipFree(ctx->t.W);
  (void)hipFree(ctx->t.JR); (void)hipFree(ctx->t.I0);
  (void)hipFree(ctx->im1_dev); (void)hipFree(ctx->im1s_dev); (void)hipFree(ctx->im0_dev); (void)hipFree(ctx->depth_dev);
  (void)hipFree(ctx->points_dev); (void)hipFree(ctx->Twc_dev); (void)hipFree(ctx->depth16_dev);
  if (ctx->pair_stage) (void)hipHostFree(ctx->pair_stage);
  (void)hipFree(ctx->Nc_dev); (void)hipFree(ctx->Href_dev); (void)hipFree(ctx->ctab_dev); (void)hipFree(ctx->repair_count_dev);
  (void)hipFree(ctx->bsv_img_dev); (void)hipFree(ctx->bsi_img_dev);
  if (ctx->bsv_stage) (void)hipHostFree(ctx->bsv_stage);
  if (ctx->bsi_stage) (void)hipHostFree(ctx->bsi_stage);
  for (hipEvent_t e : ctx->bs_part_done) if (e) (void)hipEventDestroy(e);
  (void)hipFree(ctx->repair_queue_dev[0]); (void)hipFree(ctx->repair_queue_dev[1]);
  (void)hipFree(ctx->dbg_u); (void)hipFree(ctx->dbg_v); (void)hipFree(ctx->dbg_ic);
  (void)hipFree(ctx->dbg_wc); (void)hipFree(ctx->dbg_jc); (void)hipFree(ctx->dbg_stamps);
  for (int r = 0; r < nid_ctx::kSeqRing; r++) {
    (void)hipFree(ctx->seq_dev[r]);
    if (ctx->seq_host[r]) (void)hipHostFree(ctx->seq_host[r]);
    if (ctx->seq_done[r]) (void)hipEventDestroy(ctx->seq_done[r]);
    if (ctx->seq_fence[r]) (void)hipEventDestroy(ctx->seq_fence[r]);
  }
  if (ctx->copy_stream) { (void)hipStreamSynchronize(ctx->copy_stream); (void)hipStreamDestroy(ctx->copy_stream); }
  (void)hipFree(ctx->slab_cellout); (void)hipFree(ctx->slab_reduced); (void)hipFree(ctx->slab_quad);
  (void)hipFree(ctx->slab_ticket); (void)hipFree(ctx->slab_gpart);
  if (ctx->slab_reduced_host) (void)hipHostFree(ctx->slab_reduced_host);
  for (int s = 0; s < NID_SLOTS; s++) {
    Slot &S = ctx->slots[s];
    if (S.cellout_host) (void)hipHostFree(S.cellout_host);
    if (S.quad_host) (void)hipHostFree(S.quad_host);
    if (S.groups_host) (void)hipHostFree(S.groups_host);
    if (S.done) (void)hipEventDestroy(S.done);
    if (S.e0) (void)hipEventDestroy(S.e0);
    if (S.e1) (void)hipEventDestroy(S.e1);
  }
  for (int r = 0; r < nid_ctx::kExtRing; r++) {
    (void)hipFree(ctx->ext_dev[r]);
    if (ctx->ext_host[r]) (void)hipHostFree(ctx->ext_host[r]);
    if (ctx->ext_done[r]) (void)hipEventDestroy(ctx->ext_done[r]);
  }
  if (ctx->aux_stream) { (void)hipStreamSynchronize(ctx->aux_stream); (void)hipStreamDestroy(ctx->aux_stream); }
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return NID_OK;
}

int nid_set_options(nid_ctx *ctx, int jac_bound_mode, int xform_mode) {
  if (!ctx || jac_bound_mode < 0 || jac_bound_mode > 1 || xform_mode < 0 || xform_mode > 1)
    return NID_ERR_INVALID_ARG;
  resident_retire(ctx);
  ctx->jac_bound = jac_bound_mode;
  ctx->xform = xform_mode;
  return NID_OK;
}

int nid_set_math_mode(nid_ctx *ctx, int mode) {
  if (!ctx || (mode != NID_MATH_STRICT && mode != NID_MATH_FAST)) return NID_ERR_INVALID_ARG;
  resident_retire(ctx);
  ctx->math_mode = mode;
  return NID_OK;
}

double nid_log2_fast_host(double x) { return log2_fast(x); }

void nid_bspline4_poly_host(double u, int bin_num, double *B4, double *D4) {
  // host evaluation of the FAST-mode polynomial table (unit tests)
  std::vector<double> coef;
  const int S = bin_num - 3;
  build_coef_table(S, &coef);
  const int jc = (int)u;
  const double t = u - (double)jc;
  const double *c = coef.data() + (size_t)jc * kCoefRow;
  for (int k = 0; k < 4; k++) {
    const double *ck = c + 7 * k;
    if (B4) B4[k] = std::fma(std::fma(std::fma(ck[3], t, ck[2]), t, ck[1]), t, ck[0]);
    if (D4) D4[k] = (u == 0.0) ? 0.0 : std::fma(std::fma(ck[6], t, ck[5]), t, ck[4]);
  }
}

int nid_set_stream(nid_ctx *ctx, void *hip_stream) {
  if (!ctx) return NID_ERR_INVALID_ARG;
  hipStream_t next = hip_stream ? static_cast<hipStream_t>(hip_stream) : ctx->own_stream;
  if (next == ctx->stream) return NID_OK;
  // Evaluation launches of a context run on `stream` and (own streams only) on aux_stream, and each of the two has ONE
  // repair queue (launch_eval2): a launch still in flight on the old stream and one on the new stream would share
  // queue 0 without being ordered -- k_repair of the one could read or reset the count while k_eval2 of the other
  // pushes, and a lost cell never publishes (its nid_wait would run into the time-out).  So: no switch while a launch
  // is uncollected, and the old streams are drained (k_repair behind a collected launch may still be running).
  for (int s = 0; s < NID_SLOTS; s++)
    if (ctx->slots[s].pending) { ctx->last_error = "nid_set_stream: a launch is still pending: nid_wait() it first"; return NID_ERR_STATE; }
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  resident_retire(ctx);
  if (ctx->stream) NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (ctx->aux_stream) NID_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));
  ctx->stream = next;
  ctx->external_stream = hip_stream != nullptr;
  return NID_OK;
}

static bool valid_threads(int t) { return t == 0 || t == 128 || t == 256 || t == 512 || t == 1024; }

int nid_set_launch_shape(nid_ctx *ctx, int jac_threads, int cost_threads) {
  if (!ctx) return NID_ERR_INVALID_ARG;
  if (!valid_threads(jac_threads) || !valid_threads(cost_threads)) return NID_ERR_UNSUPPORTED;
  if (jac_threads != ctx->jac_threads) resident_retire(ctx);
  ctx->jac_threads = jac_threads;
  ctx->cost_threads = cost_threads;
  return NID_OK;
}

int nid_set_block_threads(nid_ctx *ctx, int threads) {  // one shape for both kinds of launches (0 = defaults)
  if (!ctx || threads < 0) return NID_ERR_INVALID_ARG;
  return nid_set_launch_shape(ctx, threads, threads);
}

int nid_set_reference_depth(nid_ctx *ctx, const double *depth_m, const uint8_t *im0, const double *Twc) {
  if (!ctx || !depth_m || !im0 || !Twc) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  return upload_tiles(ctx, depth_m, nullptr, im0, Twc);
}

int nid_set_reference_points(nid_ctx *ctx, const double *points3d, const uint8_t *im0) {
  if (!ctx || !points3d || !im0) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  return upload_tiles(ctx, nullptr, points3d, im0, nullptr);
}

int nid_set_pair_u16(nid_ctx *ctx, const uint16_t *depth_u16, double depth_factor, const uint8_t *im0, const uint8_t *im1,
                     const double *Twc, const double *pose0_7, const double *pose0_colmajor16, int32_t *bs_counter, double *Href) {
  if (!ctx || !depth_u16 || !im0 || !im1 || !Twc || (pose0_7 == nullptr) == (pose0_colmajor16 == nullptr)) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  Pose p;
  if (pose0_7) pose_from_pose7(pose0_7, ctx->xform, &p); else pose_from_matrix16(pose0_colmajor16, &p);
  int rc = pair_u16_enqueue(ctx, depth_u16, depth_factor, im0, im1, Twc, p);
  return rc ? rc : pair_u16_collect(ctx, bs_counter, Href);
}

int nid_get_points3d(nid_ctx *ctx, double *points3d) {
  if (!ctx || !points3d) return NID_ERR_INVALID_ARG;
  if (!ctx->have_ref) return NID_ERR_STATE;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  const size_t N = (size_t)ctx->g.rows * ctx->g.cols;
  if (!ctx->points_dev) { int rc = dev_alloc(ctx, &ctx->points_dev, 3 * N); if (rc) return rc; }
  if (ctx->ref_from_depth) {
    hipLaunchKernelGGL(k_backproject_plain, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, ctx->stream,
                       ctx->g, ctx->depth_dev, ctx->Twc_dev, ctx->points_dev);
    NID_HIP(ctx, hipGetLastError());
  }
  NID_HIP(ctx, hipMemcpyAsync(points3d, ctx->points_dev, 3 * N * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
  NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
  return NID_OK;
}

namespace {
// nid_backproject's scratch, kept from call to call (one frame pair after the other, the same size): device buffers,
// pinned staging and a stream -- three hipMalloc + three hipFree (each a device-wide wait) and two pageable copies per
// call were most of its 0.8 ms (profiles/r04_pair_setup.txt).  The function has no context, so the scratch is the
// process's: callers are serialised by g_bp_mutex (the function was re-entrant before it kept a scratch), and
// nid_backproject_release() -- called by nid_legacy_reset -- gives the ~25 MB per 640x480 back.
constexpr int kBpParts = 4;
struct BackprojectScratch {
  int device = -1;
  size_t cap = 0;
  double *d_depth = nullptr, *d_T = nullptr, *d_pts = nullptr, *h_stage = nullptr;  // h_stage: [max(3 cap, cap + 16)] pinned
  hipStream_t stream = nullptr;
  hipEvent_t part_done[kBpParts] = {};
  bool make_events() {
    for (hipEvent_t &e : part_done)
      if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return false;
    return true;
  }
  void release() {
    if (device >= 0) (void)hipSetDevice(device);
    (void)hipFree(d_depth); (void)hipFree(d_T); (void)hipFree(d_pts);
    if (h_stage) (void)hipHostFree(h_stage);
    if (stream) (void)hipStreamDestroy(stream);
    for (hipEvent_t e : part_done) if (e) (void)hipEventDestroy(e);
    *this = BackprojectScratch();
  }
} g_bp;
std::mutex g_bp_mutex;
}  // namespace

int nid_backproject_release(void) {
  std::lock_guard<std::mutex> lock(g_bp_mutex);
  g_bp.release();
  return NID_OK;
}

int nid_backproject(const double *depth_m, const double *T_wc0, double fx, double fy, double cx, double cy,
                    int32_t rows, int32_t cols, int32_t device, double *points3d) {
  // context-free twin of Calculate3Dpoint (CudaPoints3d.cu:35-73)
  if (!depth_m || !T_wc0 || !points3d || rows < 1 || cols < 1) return NID_ERR_INVALID_ARG;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return NID_ERR_NO_DEVICE;
  if (device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) return NID_ERR_INVALID_ARG;
  const size_t N = (size_t)rows * cols;
  std::lock_guard<std::mutex> lock(g_bp_mutex);
  BackprojectScratch &B = g_bp;
  if (B.device != device || B.cap < N) {
    B.release();
    if (hipSetDevice(device) != hipSuccess) return NID_ERR_INVALID_ARG;
    if (hipMalloc(reinterpret_cast<void **>(&B.d_depth), N * 8) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&B.d_T), 16 * 8) != hipSuccess ||
        hipMalloc(reinterpret_cast<void **>(&B.d_pts), 3 * N * 8) != hipSuccess ||
        // (depth + the 4x4 pose go down through it: N + 16 doubles, more than 3 N for images of fewer than 8 pixels)
        hipHostMalloc(reinterpret_cast<void **>(&B.h_stage), std::max(3 * N, N + 16) * 8, hipHostMallocDefault) != hipSuccess ||
        hipStreamCreateWithFlags(&B.stream, hipStreamNonBlocking) != hipSuccess || !B.make_events()) {
      (void)hipGetLastError();
      B.release();
      return NID_ERR_NOMEM;
    }
    B.device = device; B.cap = N;
  }
  Geometry g{};
  g.rows = rows; g.cols = cols; g.fx = fx; g.fy = fy; g.cx = cx; g.cy = cy;
  // depth and pose go down through the pinned block (its first N + 16 doubles), the points come back into all of it
  std::memcpy(B.h_stage, depth_m, N * 8);
  std::memcpy(B.h_stage + N, T_wc0, 16 * 8);
  int rc = NID_OK;
  if (hipMemcpyAsync(B.d_depth, B.h_stage, N * 8, hipMemcpyHostToDevice, B.stream) != hipSuccess ||
      hipMemcpyAsync(B.d_T, B.h_stage + N, 16 * 8, hipMemcpyHostToDevice, B.stream) != hipSuccess) rc = NID_ERR_HIP;
  if (rc == NID_OK) {
    hipLaunchKernelGGL(k_backproject_plain, dim3((unsigned)((N + 255) / 256)), dim3(256), 0, B.stream, g, B.d_depth, B.d_T, B.d_pts);
    if (hipGetLastError() != hipSuccess) rc = NID_ERR_HIP;
    // the points (7.4 MB at 640x480) come home in pieces, each copied out to the caller while the next crosses PCIe
    const size_t total = 3 * N, part = ((total / kBpParts) + 7) & ~(size_t)7;
    for (int p = 0; p < kBpParts && rc == NID_OK; p++) {
      const size_t lo = std::min(total, part * p), hi = p == kBpParts - 1 ? total : std::min(total, part * (p + 1));
      if ((hi > lo && hipMemcpyAsync(B.h_stage + lo, B.d_pts + lo, (hi - lo) * 8, hipMemcpyDeviceToHost, B.stream) != hipSuccess) ||
          hipEventRecord(B.part_done[p], B.stream) != hipSuccess) rc = NID_ERR_HIP;
    }
    for (int p = 0; p < kBpParts && rc == NID_OK; p++) {
      const size_t lo = std::min(total, part * p), hi = p == kBpParts - 1 ? total : std::min(total, part * (p + 1));
      if (hipEventSynchronize(B.part_done[p]) != hipSuccess) { rc = NID_ERR_HIP; break; }
      if (hi > lo) std::memcpy(points3d + lo, B.h_stage + lo, (hi - lo) * 8);
    }
    if (rc != NID_OK) (void)hipStreamSynchronize(B.stream);
  }
  if (rc != NID_OK) (void)hipGetLastError();
  return rc;
}

int nid_set_target_u8(nid_ctx *ctx, const uint8_t *im1) {
  if (!ctx || !im1) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  resident_retire(ctx);
  const size_t N = (size_t)ctx->g.rows * ctx->g.cols;
  NID_HIP(ctx, hipMemcpyAsync(ctx->im1_dev, im1, N, hipMemcpyHostToDevice, ctx->stream));
  {
    const Geometry &g = ctx->g;
    const long total = (long)(g.rows + 1) * (g.cols + 1);
    hipLaunchKernelGGL(k_im1_margins, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, g.rows, g.cols,
                       ctx->im1_stride, ctx->im1_dev, ctx->im1s_dev);
    NID_HIP(ctx, hipGetLastError());
  }
  NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
  ctx->have_target = true;
  return NID_OK;
}

// branch-free so that it vectorises (this conversion sits on the first call of every frame pair: the
// reference's operators take the images as double arrays); compiled twice, the AVX2 body is picked at run time
#define NID_F64_TO_U8_BODY                                      \
  int bad = 0;                                                  \
  for (int64_t i = 0; i < n; i++) {                             \
    const double v = im[i];                                     \
    const bool in = (v >= 0.0) & (v <= 255.0); /* NaN: false */ \
    const int iv = (int)(in ? v : 0.0);                         \
    bad |= (int)!in | (int)((double)iv != v);                   \
    out[i] = (uint8_t)iv;                                       \
  }                                                             \
  return bad;
static int f64_to_u8_generic(const double *__restrict__ im, int64_t n, uint8_t *__restrict__ out) { NID_F64_TO_U8_BODY }
__attribute__((target("avx2"))) static int f64_to_u8_avx2(const double *__restrict__ im, int64_t n,
                                                          uint8_t *__restrict__ out) { NID_F64_TO_U8_BODY }
#undef NID_F64_TO_U8_BODY

int nid_set_reference_image_f64(const double *im, int64_t n, uint8_t *out) {
  if (!im || !out || n < 0) return NID_ERR_INVALID_ARG;
  const int bad = __builtin_cpu_supports("avx2") ? f64_to_u8_avx2(im, n, out) : f64_to_u8_generic(im, n, out);
  return bad ? NID_ERR_UNSUPPORTED : NID_OK;
}

int nid_set_target_f64(nid_ctx *ctx, const double *im1) {
  if (!ctx || !im1) return NID_ERR_INVALID_ARG;
  const size_t N = (size_t)ctx->g.rows * ctx->g.cols;
  std::vector<uint8_t> tmp(N);
  int rc = nid_set_reference_image_f64(im1, (int64_t)N, tmp.data());
  if (rc) { ctx->last_error = "f64 image is not u8-valued"; return rc; }
  return nid_set_target_u8(ctx, tmp.data());
}

int nid_compute_href(nid_ctx *ctx, const double *pose7, int32_t *bs_counter, double *Href,
                     double *bs_value, int32_t *bs_index) {
  if (!ctx || !pose7) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  Pose p; pose_from_pose7(pose7, ctx->xform, &p);
  return href_common(ctx, p, bs_counter, Href, bs_value, bs_index);
}

int nid_compute_href_matrix(nid_ctx *ctx, const double *pose16, int32_t *bs_counter, double *Href,
                            double *bs_value, int32_t *bs_index) {
  if (!ctx || !pose16) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  Pose p; pose_from_matrix16(pose16, &p);
  return href_common(ctx, p, bs_counter, Href, bs_value, bs_index);
}

int nid_plain_nid(nid_ctx *ctx, const double *pose7, int bins, double *Href, double *Hcur, double *Hjoint,
                  double *nid, double *mi, int32_t *n_in, double *total) {
  if (!ctx || !pose7 || bins < 1 || bins > kMaxPlainBins) return NID_ERR_INVALID_ARG;
  if (!ctx->have_ref || !ctx->have_target) { ctx->last_error = "reference / target not set"; return NID_ERR_STATE; }
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  resident_retire(ctx);  // (hipFree below waits for the whole device)
  Pose p; pose_from_pose7(pose7, NID_XFORM_MATRIX, &p);  // T_cw1 * pw as a 4x4 product (:353-354)
  const int nloc = ctx->g.nloc;
  double *out_dev = nullptr;
  NID_HIP(ctx, hipMalloc(&out_dev, (size_t)nloc * 6 * sizeof(double)));
  hipLaunchKernelGGL((k_plain_nid<256>), dim3(nloc), dim3(256), 0, ctx->stream, ctx->g, p, ctx->t, ctx->im1_dev, bins,
                     out_dev);
  std::vector<double> o((size_t)nloc * 6);
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) e = hipMemcpyAsync(o.data(), out_dev, o.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  (void)hipFree(out_dev);
  NID_HIP(ctx, e);
  double sum = 0.0;
  for (int cl = 0; cl < nloc; cl++) {
    const int c = ctx->g.cell_begin + cl * ctx->g.cell_stride;
    const double *r = o.data() + (size_t)cl * 6;
    if (Href) Href[c] = r[0];
    if (Hcur) Hcur[c] = r[1];
    if (Hjoint) Hjoint[c] = r[2];
    if (nid) nid[c] = r[3];
    if (mi) mi[c] = r[4];
    if (n_in) n_in[c] = (int32_t)r[5];
    if (!std::isnan(r[3])) sum += r[3] * r[3];  // total_nid += nid*nid (:196), cells in row-major order
  }
  if (total) *total = std::sqrt(sum);  // "final nid" (:201); with cell sharding: this rank's cells only
  return NID_OK;
}

int nid_set_href_state(nid_ctx *ctx, const int32_t *bs_counter, const double *Href,
                       const double *bs_value, const int32_t *bs_index) {
  (void)bs_index;  // the reference bin index is recomputed from im0 for every pixel (Appendix A.2)
  if (!ctx || !bs_counter || !Href || !bs_value) return NID_ERR_INVALID_ARG;
  if (!ctx->have_ref) return NID_ERR_STATE;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  resident_retire(ctx);
  const Geometry &g = ctx->g;
  const size_t plane = (size_t)g.nloc * g.pstride;
  std::vector<double> W(4 * plane, 0.0);
  for (int cl = 0; cl < g.nloc; cl++) {
    const int c = g.cell_begin + cl * g.cell_stride, ci = c / g.cell_num, cj = c % g.cell_num;
    for (int s = 0; s < g.ps; s++) {
      const size_t id = (size_t)(ci * g.rb + s / g.cb) * g.cols + cj * g.cb + s % g.cb;
      const size_t gi = (size_t)cl * g.pstride + s;
      for (int k = 0; k < 4; k++) {
        const double w = bs_value[4 * id + k];
        W[4 * gi + k] = std::isnan(w) ? 0.0 : w;  // CUDA-path NaN marker -> CPU-edge zero (D2)
      }
      // the evaluation kernel's flag for tiny non-zero outer weights, as k_href leaves it: the sign of the first weight
      const double wmin = std::fmin(W[4 * gi], W[4 * gi + 3]);
      if (wmin < nid::kTinyW && wmin != 0.0) W[4 * gi] = -W[4 * gi];
    }
  }
  NID_HIP(ctx, hipMemcpy(ctx->t.W, W.data(), 4 * plane * sizeof(double), hipMemcpyHostToDevice));
  std::vector<int> nc(g.nloc);
  std::vector<double> hr(g.nloc);
  for (int cl = 0; cl < g.nloc; cl++) { nc[cl] = bs_counter[g.cell_begin + cl * g.cell_stride]; hr[cl] = Href[g.cell_begin + cl * g.cell_stride]; }
  NID_HIP(ctx, hipMemcpy(ctx->Nc_dev, nc.data(), g.nloc * sizeof(int), hipMemcpyHostToDevice));
  NID_HIP(ctx, hipMemcpy(ctx->Href_dev, hr.data(), g.nloc * sizeof(double), hipMemcpyHostToDevice));
  ctx->have_href = true;
  return NID_OK;
}

int nid_evaluate(nid_ctx *ctx, const double *pose7, int want_jac, double *Ht, double *Hj, double *err,
                 double *der) {
  if (!ctx || !pose7) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  Pose p; pose_from_pose7(pose7, ctx->xform, &p);
  return evaluate_common(ctx, p, want_jac, Ht, Hj, err, der);
}

int nid_evaluate_matrix(nid_ctx *ctx, const double *pose16, int want_jac, double *Ht, double *Hj,
                        double *err, double *der) {
  if (!ctx || !pose16) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  Pose p; pose_from_matrix16(pose16, &p);
  return evaluate_common(ctx, p, want_jac, Ht, Hj, err, der);
}

int nid_unpack_reduced(const double *r, double *H36, double *b6, double *chi2, int32_t *n_active) {
  if (!r) return NID_ERR_INVALID_ARG;
  if (chi2) *chi2 = r[0];
  if (b6) for (int n = 0; n < 6; n++) b6[n] = r[1 + n];
  if (H36) {
    int idx = 7;
    for (int a = 0; a < 6; a++)
      for (int b = a; b < 6; b++) { H36[a * 6 + b] = r[idx]; H36[b * 6 + a] = r[idx]; idx++; }
  }
  if (n_active) *n_active = (int32_t)r[28];
  return NID_OK;
}

int nid_launch(nid_ctx *ctx, int slot, const double *pose7, int want_jac, double delta) {
  if (!ctx || !pose7) return NID_ERR_INVALID_ARG;
  Pose p; pose_from_pose7(pose7, ctx->xform, &p);
  return launch_slot(ctx, slot, p, want_jac, delta, nullptr);
}

int nid_launch_batch(nid_ctx *ctx, int first_slot, int n, const double *poses7, int want_jac, double delta) {
  if (!ctx || !poses7 || n < 1 || n > kMaxBatchExt) return NID_ERR_INVALID_ARG;
  Pose p[kMaxBatchExt];
  for (int k = 0; k < n; k++) pose_from_pose7(poses7 + 7 * k, ctx->xform, &p[k]);
  return launch_split(ctx, first_slot, n, p, want_jac, delta);  // (one launch, or a short sequence's several: plan_split)
}

int nid_set_short_sequence_policy(nid_ctx *ctx, int poses_per_launch, int streams) {
  if (!ctx || poses_per_launch < 0 || poses_per_launch > kMaxBatchExt || streams < 0 || streams > 2) return NID_ERR_INVALID_ARG;
  ctx->seq_chunk = poses_per_launch;
  ctx->seq_streams = streams;
  return NID_OK;
}

int nid_launch_chain(nid_ctx *ctx, int first_slot, int n, const double *poses7, int n_jac, double delta) {
  if (!ctx || !poses7 || n < 1 || n > kMaxBatchExt || n_jac < 0 || n_jac > n) return NID_ERR_INVALID_ARG;
  if (first_slot < 0 || first_slot + n > NID_SLOTS) return NID_ERR_INVALID_ARG;
  for (int k = 0; k < n; k++)  // all or nothing: no half-launched chain
    if (ctx->slots[first_slot + k].pending) { ctx->last_error = "slot still pending: nid_wait() it first"; return NID_ERR_STATE; }
  Pose p[kMaxBatchExt];
  for (int k = 0; k < n; k++) pose_from_pose7(poses7 + 7 * k, ctx->xform, &p[k]);
  int rc = NID_OK;
  if (n_jac > 0) rc = launch_batch(ctx, first_slot, n_jac, p, 1, delta);
  if (rc == NID_OK && n_jac < n) rc = launch_batch(ctx, first_slot + n_jac, n - n_jac, p + n_jac, 0, delta, nullptr, true);
  return rc;
}

int nid_launch_batch_to(nid_ctx *ctx, int first_slot, int n, const double *poses7, int want_jac, double delta,
                        void *reduced_dev) {
  if (!ctx || !poses7 || !reduced_dev || n < 1 || n > kMaxBatchExt) return NID_ERR_INVALID_ARG;
  Pose p[kMaxBatchExt];
  for (int k = 0; k < n; k++) pose_from_pose7(poses7 + 7 * k, ctx->xform, &p[k]);
  return launch_batch(ctx, first_slot, n, p, want_jac, delta, static_cast<double *>(reduced_dev));
}

}  // extern "C"

namespace {
// result buffers (device + pinned host), events and the copy stream of the pipelined loop, for launches of `batch` poses
int ensure_seq_ring(nid_ctx *ctx, int batch) {
  if (ctx->seq_cap >= (size_t)batch) return NID_OK;
  resident_retire(ctx);  // (hipFree waits for the whole device)
  for (int r = 0; r < nid_ctx::kSeqRing; r++) {
    (void)hipFree(ctx->seq_dev[r]); ctx->seq_dev[r] = nullptr;
    if (ctx->seq_host[r]) (void)hipHostFree(ctx->seq_host[r]);
    ctx->seq_host[r] = nullptr;
    int rc = dev_alloc(ctx, &ctx->seq_dev[r], (size_t)batch * kReducedLen);
    if (rc) return rc;
    if (hipHostMalloc(reinterpret_cast<void **>(&ctx->seq_host[r]), (size_t)batch * kReducedLen * sizeof(double),
                      hipHostMallocDefault) != hipSuccess) return NID_ERR_NOMEM;
    if (!ctx->seq_done[r]) NID_HIP(ctx, hipEventCreateWithFlags(&ctx->seq_done[r], hipEventDisableTiming));
    if (!ctx->seq_fence[r]) NID_HIP(ctx, hipEventCreateWithFlags(&ctx->seq_fence[r], hipEventDisableTiming));
  }
  if (!ctx->copy_stream) NID_HIP(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
  ctx->seq_cap = (size_t)batch;
  return NID_OK;
}
}  // namespace

extern "C" {

int nid_run_sequence(nid_ctx *ctx, const double *poses7, int n, int batch, int want_jac, double delta,
                     double *reduced_out) {
  // Host-side pipeline: `batch` poses per launch, up to kSeqRing launches in flight on the context's two streams.
  // A launch writes its result blocks to a DEVICE buffer and ONE copy brings them to pinned host memory behind it
  // (copy stream + event).  (The blocking calls let the kernel write each pose's block straight to host memory --
  // lowest latency; for a stream of launches that is 256 system-scope fences and PCIe writes per launch from
  // inside the kernel: this form measured 254 k -> 270-280 k evaluations/s on 640x480.)
  // (any batch <= NID_MAX_BATCH: ring entry r owns the slots r * batch .. r * batch + batch - 1, and depth * batch <= NID_SLOTS below)
  if (!ctx || !poses7 || n < 0 || batch < 1 || batch > kMaxBatchExt) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  for (int s = 0; s < NID_SLOTS; s++) if (ctx->slots[s].pending) return NID_ERR_STATE;
  if (n <= batch) {
    // Nothing to pipeline, latency is what counts -- the kernel writes every pose's block straight to pinned host
    // memory and the host spins on the sequence words (no copy, no event: 128 us instead of 193 us from the enqueue
    // to the last of 20 results); a SHORT sequence goes as several launches of <= kMaxBatch poses (plan_split)
    if (n == 0) return NID_OK;
    Pose q[kMaxBatchExt];
    for (int k = 0; k < n; k++) pose_from_pose7(poses7 + 7 * (size_t)k, ctx->xform, &q[k]);
    int rc = launch_split(ctx, 0, n, q, want_jac, delta);
    if (rc) return rc;
    for (int k = 0; k < n; k++) {
      rc = nid_wait(ctx, k, nullptr, nullptr, nullptr, nullptr);
      if (rc) return rc;
      if (reduced_out) std::memcpy(reduced_out + (size_t)k * kReducedLen, ctx->slots[k].reduced_host, kReducedLen * sizeof(double));
    }
    return NID_OK;
  }
  const int depth = std::min((int)nid_ctx::kSeqRing, NID_SLOTS / batch);
  { int rc = ensure_seq_ring(ctx, batch); if (rc) return rc; }
  // Consecutive launches alternate between the context's two streams: a launch's last workgroups leave most CUs
  // idle for a while and the next launch's workgroups fill that tail.  Launches are independent (own slots, own
  // result buffers), so no cross-stream ordering is needed.
  static const bool one_stream = getenv("NID_ONE_STREAM") != nullptr;
  Pose p[kMaxBatchExt];
  const int launches = (n + batch - 1) / batch;
  auto collect = [&](int l) -> int {  // launch l has landed in seq_host[l % depth]
    const int r = l % depth;
    NID_HIP(ctx, hipEventSynchronize(ctx->seq_done[r]));
    const int first = l * batch, cnt = std::min(batch, n - first);
    if (reduced_out) std::memcpy(reduced_out + (size_t)first * kReducedLen, ctx->seq_host[r], (size_t)cnt * kReducedLen * sizeof(double));
    for (int k = 0; k < cnt; k++) ctx->slots[r * batch + k].pending = false;
    return NID_OK;
  };
  for (int l = 0; l < launches; l++) {
    const int r = l % depth;
    if (l >= depth) { int rc = collect(l - depth); if (rc) return rc; }  // frees ring entry r and its slots
    const int first = l * batch, nb = std::min(batch, n - first);
    for (int k = 0; k < nb; k++) pose_from_pose7(poses7 + 7 * (size_t)(first + k), ctx->xform, &p[k]);
    const bool aux = !one_stream && (l & 1) && !ctx->external_stream;
    int rc = launch_batch(ctx, r * batch, nb, p, want_jac, delta, ctx->seq_dev[r], aux);
    if (rc) return rc;
    hipStream_t ls = aux ? ctx->aux_stream : ctx->stream;
    hipStream_t cs = ctx->external_stream ? ctx->stream : ctx->copy_stream;
    if (cs != ls) {
      NID_HIP(ctx, hipEventRecord(ctx->seq_fence[r], ls));
      NID_HIP(ctx, hipStreamWaitEvent(cs, ctx->seq_fence[r], 0));
    }
    NID_HIP(ctx, hipMemcpyAsync(ctx->seq_host[r], ctx->seq_dev[r], (size_t)nb * kReducedLen * sizeof(double), hipMemcpyDeviceToHost, cs));
    NID_HIP(ctx, hipEventRecord(ctx->seq_done[r], cs));
  }
  for (int l = std::max(0, launches - depth); l < launches; l++) { int rc = collect(l); if (rc) return rc; }
  return NID_OK;
}

int nid_run_chain(nid_ctx *ctx, const double *poses7, int n, int want_jac, double delta, double *reduced_out, double *seconds) {
  if (!ctx || !poses7 || n < 0) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; i++) {
    int rc = nid_launch(ctx, 0, poses7 + 7 * (size_t)i, want_jac, delta);
    if (rc) return rc;
    rc = nid_wait(ctx, 0, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
    if (reduced_out) std::memcpy(reduced_out + (size_t)i * kReducedLen, ctx->slots[0].reduced_host, kReducedLen * sizeof(double));
  }
  if (seconds) *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return NID_OK;
}

int nid_launch_to(nid_ctx *ctx, int slot, const double *pose7, int want_jac, double delta, void *reduced_dev) {
  if (!ctx || !pose7 || !reduced_dev) return NID_ERR_INVALID_ARG;
  Pose p; pose_from_pose7(pose7, ctx->xform, &p);
  return launch_slot(ctx, slot, p, want_jac, delta, reduced_dev);
}

int nid_wait(nid_ctx *ctx, int slot, double *H36, double *b6, double *chi2, int32_t *n_active) {
  if (!ctx || slot < 0 || slot >= NID_SLOTS) return NID_ERR_INVALID_ARG;
  Slot &S = ctx->slots[slot];
  if (!S.pending) return NID_ERR_STATE;
  // `pending` is cleared only once the result is known to have arrived: after an error the slot stays
  // pending (a retry is possible, a relaunch is refused)
  if (S.external_target) {
    NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
    NID_HIP(ctx, hipEventSynchronize(ctx->slots[S.done_slot].done));
    S.pending = false;
    return NID_OK;
  }
  // the last workgroup stores the 32 results, then the sequence word (system-scope release); DIRECT launches: the
  // host collects the cells' records and sums their quadratic forms
  if (S.collected) {
    S.collected = false;
  } else {
    int rc = S.direct ? wait_direct(ctx, S) : (S.groups ? wait_groups(ctx, S) : wait_host_seq(ctx, S));
    if (rc) return rc;
  }
  S.direct = false;
  S.groups = false;
  S.pending = false;
  return nid_unpack_reduced(S.reduced_host, H36, b6, chi2, n_active);
}

int nid_normal_equations(nid_ctx *ctx, const double *pose7, int want_jac, double delta, double *H36,
                         double *b6, double *chi2, int32_t *n_active) {
  if (!ctx || !pose7) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  int rc = nid_launch(ctx, 0, pose7, want_jac, delta);
  if (rc) return rc;
  return nid_wait(ctx, 0, H36, b6, chi2, n_active);
}

int nid_slot_buffers(nid_ctx *ctx, int slot, void **reduced_dev, void **cellout_dev) {
  if (!ctx || slot < 0 || slot >= NID_SLOTS) return NID_ERR_INVALID_ARG;
  if (reduced_dev) *reduced_dev = ctx->slots[slot].reduced_dev;
  if (cellout_dev) *cellout_dev = ctx->slots[slot].cellout_dev;
  return NID_OK;
}

int nid_debug_read_device(nid_ctx *ctx, const void *dev, void *host, size_t bytes) {
  if (!ctx || !dev || !host) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  resident_retire(ctx);  // (a blocking copy on the null stream would wait for the resident kernel to leave)
  NID_HIP(ctx, hipMemcpy(host, dev, bytes, hipMemcpyDeviceToHost));
  return NID_OK;
}

int nid_debug_enable_pixel_dump(nid_ctx *ctx, int enable) {
  if (!ctx) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  resident_retire(ctx);
  if (enable && !ctx->dbg_u) {
    const size_t N = (size_t)ctx->g.rows * ctx->g.cols;
    int rc;
    if ((rc = dev_alloc(ctx, &ctx->dbg_u, N))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->dbg_v, N))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->dbg_ic, N))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->dbg_wc, 4 * N))) return rc;
    if ((rc = dev_alloc(ctx, &ctx->dbg_jc, N))) return rc;
  }
  ctx->dbg_enabled = enable != 0;
  ctx->dbg_jac = enable == 2 ? 1 : 0;
  return NID_OK;
}

int nid_debug_get_pixel_dump(nid_ctx *ctx, double *u, double *v, double *ic, int32_t *jc, double *wc4) {
  if (!ctx || !ctx->dbg_u) return NID_ERR_STATE;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  const size_t N = (size_t)ctx->g.rows * ctx->g.cols;
  NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
  if (u) NID_HIP(ctx, hipMemcpy(u, ctx->dbg_u, N * 8, hipMemcpyDeviceToHost));
  if (v) NID_HIP(ctx, hipMemcpy(v, ctx->dbg_v, N * 8, hipMemcpyDeviceToHost));
  if (ic) NID_HIP(ctx, hipMemcpy(ic, ctx->dbg_ic, N * 8, hipMemcpyDeviceToHost));
  if (jc) NID_HIP(ctx, hipMemcpy(jc, ctx->dbg_jc, N * 4, hipMemcpyDeviceToHost));
  if (wc4) NID_HIP(ctx, hipMemcpy(wc4, ctx->dbg_wc, N * 32, hipMemcpyDeviceToHost));
  return NID_OK;
}

int nid_debug_repair_count(nid_ctx *ctx, int64_t *count, int reset) {
  if (!ctx || !count) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  resident_retire(ctx);
  NID_HIP(ctx, hipDeviceSynchronize());
  unsigned long long v = 0;
  NID_HIP(ctx, hipMemcpy(&v, ctx->repair_count_dev, sizeof(v), hipMemcpyDeviceToHost));
  *count = (int64_t)v;
  if (reset) NID_HIP(ctx, hipMemset(ctx->repair_count_dev, 0, sizeof(v)));
  return NID_OK;
}

int nid_debug_enable_stamps(nid_ctx *ctx, int enable) {
  if (!ctx) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  resident_retire(ctx);
  if (enable && !ctx->dbg_stamps) {
    int rc = dev_alloc(ctx, &ctx->dbg_stamps, (size_t)ctx->g.nloc * 10);
    if (rc) return rc;
    NID_HIP(ctx, hipMemset(ctx->dbg_stamps, 0, (size_t)ctx->g.nloc * 10 * sizeof(long long)));
  } else if (!enable && ctx->dbg_stamps) {
    NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(ctx->dbg_stamps);
    ctx->dbg_stamps = nullptr;
  }
  return NID_OK;
}

int nid_debug_get_stamps(nid_ctx *ctx, int64_t *stamps) {
  if (!ctx || !stamps || !ctx->dbg_stamps) return NID_ERR_STATE;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  NID_HIP(ctx, hipStreamSynchronize(ctx->stream));
  NID_HIP(ctx, hipMemcpy(stamps, ctx->dbg_stamps, (size_t)ctx->g.nloc * 10 * sizeof(long long), hipMemcpyDeviceToHost));
  return NID_OK;
}

void nid_bspline4_host(double u, int bin_num, double *B4, double *D4) {
  double B[4], D[4];
  nid::bspline4<true>(u, (int)std::floor(u), bin_num - 3, B, D);
  for (int k = 0; k < 4; k++) { if (B4) B4[k] = B[k]; if (D4) D4[k] = D[k]; }
}

double nid_div_small_host(double x, double d) { return nid::div_small(x, d); }

int nid_set_href_nan_markers(nid_ctx *ctx, int on) {
  if (!ctx) return NID_ERR_INVALID_ARG;
  ctx->href_nan_rows = on != 0;
  return NID_OK;
}

int nid_set_direct_results(nid_ctx *ctx, int on) {
  if (!ctx) return NID_ERR_INVALID_ARG;
  for (int s = 0; s < NID_SLOTS; s++) if (ctx->slots[s].pending) return NID_ERR_STATE;
  if (on < 0 || on > 2) return NID_ERR_INVALID_ARG;
  ctx->direct_results = on != 0;
  ctx->direct_mode = on;
  return NID_OK;
}

int nid_set_resident(nid_ctx *ctx, int on) {
  if (!ctx) return NID_ERR_INVALID_ARG;
  for (int s = 0; s < NID_SLOTS; s++) if (ctx->slots[s].pending) return NID_ERR_STATE;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  if (!on) {
    resident_retire(ctx);
    ctx->res.enabled = false;
    return NID_OK;
  }
  int rc = resident_probe(ctx);
  if (rc == NID_OK && ctx->res.unfit_nt != 0 && ctx->res.unfit_nt == ctx->jac_threads) rc = NID_ERR_UNSUPPORTED;
  if (rc) { ctx->last_error = "resident evaluator unavailable: " + (ctx->res.why.empty() ? std::string("mailbox setup failed") : ctx->res.why); return rc; }
  ctx->res.enabled = true;
  ctx->res.fallback_run = 0;  // (asked for again: the back-off of resident_fallback starts over)
  return NID_OK;
}

int nid_resident_pause(nid_ctx *ctx) {
  if (!ctx) return NID_ERR_INVALID_ARG;
  if (!ctx->res.running) return NID_OK;
  for (int s = 0; s < NID_SLOTS; s++) if (ctx->slots[s].pending) return NID_ERR_STATE;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  resident_retire(ctx);  // (the next request starts another)
  return NID_OK;
}

int nid_resident_stats(const nid_ctx *ctx, int64_t *served, int64_t *fallbacks, int64_t *starts) {
  if (!ctx) return NID_ERR_INVALID_ARG;
  if (served) *served = ctx->res.served;
  if (fallbacks) *fallbacks = ctx->res.fallbacks;
  if (starts) *starts = ctx->res.starts;
  return NID_OK;
}

int nid_set_loop_form(nid_ctx *ctx, int on) {
  if (!ctx) return NID_ERR_INVALID_ARG;
  if (on) resident_retire(ctx);
  ctx->loop_form = on != 0;
  return NID_OK;
}

int nid_enable_timing(nid_ctx *ctx, int enable) {
  if (!ctx) return NID_ERR_INVALID_ARG;
  ctx->timing = enable != 0;
  return NID_OK;
}

int nid_last_kernel_ms(nid_ctx *ctx, int slot, float *eval_ms, float *reduce_ms) {
  if (!ctx || slot < 0 || slot >= NID_SLOTS) return NID_ERR_INVALID_ARG;
  Slot &S = ctx->slots[slot];
  if (!S.timed) return NID_ERR_STATE;
  NID_HIP(ctx, hipEventSynchronize(S.e1));
  if (eval_ms) NID_HIP(ctx, hipEventElapsedTime(eval_ms, S.e0, S.e1));
  if (reduce_ms) *reduce_ms = 0.0f;  // the reduction is fused into the evaluation kernel
  return NID_OK;
}

int nid_time_launches(nid_ctx *ctx, int n, const double *poses7, int want_jac, double delta, int repeats,
                      float *ms_per_launch) {
  // `repeats` identical n-pose launches back to back on the context's stream between two events: the
  // per-launch duration a kernel trace reports (event pairs around ONE launch add ~5 us of marker and
  // dispatch latency to a ~50 us kernel).  Same-stream launches are serialised, so reusing the slots is safe.
  if (!ctx || !poses7 || !ms_per_launch || n < 1 || n > kMaxBatchExt || repeats < 1) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  for (int s = 0; s < NID_SLOTS; s++) if (ctx->slots[s].pending) return NID_ERR_STATE;
  Pose p[kMaxBatchExt];
  for (int k = 0; k < n; k++) pose_from_pose7(poses7 + 7 * k, ctx->xform, &p[k]);
  Slot &S0 = ctx->slots[0];
  { int rc = timing_events(ctx, S0); if (rc) return rc; }
  // (nid_enable_timing would make launch_batch record the slot's e0 / e1 around every repeat: off for the duration)
  struct TimingOff { nid_ctx *c; bool was; ~TimingOff() { c->timing = was; } } timing_off{ctx, ctx->timing};
  ctx->timing = false;
  // launches of more than kMaxBatch poses are what the pipelined loop issues: they write their result blocks to a
  // device buffer like there (nid_run_sequence); smaller ones to pinned host memory like the blocking calls
  double *target = nullptr;
  if (n > kMaxBatch) {
    int rc = ensure_seq_ring(ctx, n);
    if (rc) return rc;
    target = ctx->seq_dev[0];
  }
  NID_HIP(ctx, hipEventRecord(S0.e0, ctx->stream));
  for (int r = 0; r < repeats; r++) {
    // (never DIRECT: the launches are re-issued into the same buffers before the host has looked)
    int rc = launch_batch(ctx, 0, n, p, want_jac, delta, target, false, /*relaunch_ok=*/r > 0, /*allow_direct=*/false);
    if (rc) return rc;
  }
  NID_HIP(ctx, hipEventRecord(S0.e1, ctx->stream));
  for (int k = 0; k < n; k++) {
    int rc = nid_wait(ctx, k, nullptr, nullptr, nullptr, nullptr);
    if (rc) return rc;
  }
  NID_HIP(ctx, hipEventSynchronize(S0.e1));
  float ms = 0.f;
  NID_HIP(ctx, hipEventElapsedTime(&ms, S0.e0, S0.e1));
  *ms_per_launch = ms / (float)repeats;
  return NID_OK;
}


int nid_time_kernel(nid_ctx *ctx, int n, const double *poses7, int want_jac, double delta, int repeats, float *ms_kernel) {
  // The evaluation kernel ALONE: every repeat is one n-pose launch with an event right in front of k_eval2 (behind the in-stream
  // copy of the per-pose records) and one right behind it (in front of k_repair), awaited before the next -- what a kernel trace
  // reports for that kernel (nid_time_launches brackets whole launches back to back: copy + k_eval2 + k_repair + dispatch gaps).
  if (!ctx || !poses7 || !ms_kernel || n < 1 || n > kMaxBatchExt || repeats < 1) return NID_ERR_INVALID_ARG;
  NID_HIP(ctx, hipSetDevice(ctx->cfg.device));
  for (int s = 0; s < NID_SLOTS; s++) if (ctx->slots[s].pending) return NID_ERR_STATE;
  Pose p[kMaxBatchExt];
  for (int k = 0; k < n; k++) pose_from_pose7(poses7 + 7 * k, ctx->xform, &p[k]);
  struct TimingOn { nid_ctx *c; bool was; ~TimingOn() { c->timing = was; } } timing_on{ctx, ctx->timing};
  ctx->timing = true;
  double *target = nullptr;
  if (n > kMaxBatch) {
    int rc = ensure_seq_ring(ctx, n);
    if (rc) return rc;
    target = ctx->seq_dev[0];
  }
  double sum = 0.0;
  for (int r = 0; r < repeats; r++) {
    int rc = launch_batch(ctx, 0, n, p, want_jac, delta, target, false, /*relaunch_ok=*/false, /*allow_direct=*/false);
    if (rc) return rc;
    for (int k = 0; k < n; k++) {
      rc = nid_wait(ctx, k, nullptr, nullptr, nullptr, nullptr);
      if (rc) return rc;
    }
    Slot &S0 = ctx->slots[0];
    NID_HIP(ctx, hipEventSynchronize(S0.e1));
    float ms = 0.f;
    NID_HIP(ctx, hipEventElapsedTime(&ms, S0.e0, S0.e1));
    sum += ms;
  }
  *ms_kernel = (float)(sum / repeats);
  return NID_OK;
}

int64_t nid_contract_bytes(const nid_ctx *ctx) {
  if (!ctx) return 0;
  return (int64_t)68 * ctx->g.nloc * ctx->g.ps + (int64_t)64 * ctx->g.nloc;
}

}  // extern "C"

// multi-GPU layer (include/nid/nid_multi.h): same translation unit, it drives the shard contexts' internals
#include "nid_multi.inc"
