/*
 * nid_c.h -- C-ABI of the MI355X-native NID cost/Jacobian path (libnid_hip.so).
 *
 * Drop-in boundary for the three operator entry points of arpg/NID-Pose-Estimation
 * (paths relative to the reference checkout):
 *
 *   Calculate3Dpoint   CudaPoints3d.cuh:6      -> nid_set_reference_depth / nid_get_points3d
 *   CudaComputeHref    CudaComputeHref.cuh:5   -> nid_set_reference_* + nid_set_target_u8 + nid_compute_href
 *   g2o::CudaComputeH  g2o/g2o/core/computeH.cuh:8 -> nid_evaluate / nid_evaluate_matrix
 *   per-edge quadratic form + Huber, g2o/g2o/core/base_unary_edge.hpp:43-72,
 *   robust_kernel_impl.cpp:65-91              -> nid_normal_equations (fused on device)
 *
 * The legacy C++ signatures themselves are provided, unchanged, by
 * include/nid/legacy_ops.h on top of this ABI.
 *
 * Conventions (same as the reference operators):
 *  - every pointer is a caller-owned HOST buffer unless its name ends in _dev;
 *  - images are row-major; a pose matrix is COLUMN-major 4x4 (Eigen .data());
 *  - pose7 = {qx,qy,qz,qw,tx,ty,tz} of g2o::SE3Quat (world -> camera 1);
 *  - cell id = col_cell + cell_num * row_cell (computeH.cu:220);
 *  - per-cell outputs of inactive cells (N_c < 300 or NaN Href) are NaN
 *    (computeH.cu:271-275, 313-322);
 *  - all arithmetic is IEEE f64; histogram bins are accumulated in 64-bit
 *    fixed point (order-independent, bitwise reproducible), see DESIGN.md.
 *
 * Errors: every call returns NID_OK (0) or a negative nid_status; nothing
 * throws across the boundary.  A context is bound to one HIP device and is
 * not thread-safe; use one context per host thread (the reference is single
 * threaded: computeH.cu:378,449).
 */
#ifndef NID_C_H
#define NID_C_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NID_ABI_VERSION 4 /* 2: NID_SLOTS 128 -> 1024, NID_MAX_BATCH 64 -> 256, launch shapes, launch chains, strided cell sets, nid_multi.h; 3: nid_set_pair_u16 / nid_multi_set_pair_u16 added, nid_resident_batch_stats and nid_set_resident(ctx, 2) removed; 4: nid_time_kernel added, nid_last_kernel_ms's eval_ms ends in front of the repair kernel */

typedef enum {
  NID_OK = 0,
  NID_ERR_INVALID_ARG = -1,
  NID_ERR_NO_DEVICE = -2,    /* no HIP device / HIP runtime error at init */
  NID_ERR_HIP = -3,          /* HIP runtime error (see nid_last_error) */
  NID_ERR_UNSUPPORTED = -4,  /* shape outside what the kernels are built for */
  NID_ERR_STATE = -5,        /* call order: reference/target/href not set */
  NID_ERR_NOMEM = -6
} nid_status;

/* in-bound test of the Jacobian pass: SURVEY.md section 0.2 */
#define NID_JACBOUND_CPU  0 /* u+3 <= cols-1  types_six_dof_expmap.cpp:433 (parity target) */
#define NID_JACBOUND_CUDA 1 /* u+3 <= cols    computeH.cu:164 */
/* world->camera transform of a point */
#define NID_XFORM_QUAT   0  /* quaternion rotate + t, SE3Quat::map se3quat.h:217-220 */
#define NID_XFORM_MATRIX 1  /* 4x4 matrix rows, computeH.cu:152-154 */

/* arithmetic of the evaluation kernel */
#define NID_MATH_FAST   0  /* default: same algorithm, cheaper arithmetic; per-cell results inside the stated
                              tolerances (1e-11 entropies, 1e-9 Jacobian), per-pixel values within a few ulp */
#define NID_MATH_STRICT 1  /* every rounding of the reference path reproduced: bit-exact per-pixel values */

typedef struct nid_ctx nid_ctx;

typedef struct {
  int32_t rows, cols;     /* image size */
  int32_t cell_num;       /* cells per side (16 -> 256 cells) */
  int32_t bin_num;        /* B-spline bins, 4..16 (reference default 10) */
  int32_t bs_degree;      /* must be 3 */
  int32_t device;         /* HIP device ordinal */
  int32_t cell_begin;     /* this context owns cells [cell_begin, cell_end) ... */
  int32_t cell_end;       /* ... 0,0 = all cells (multi-GPU: one shard per rank) */
  double fx, fy, cx, cy;  /* camera_intrincis[0..3] (NID_pose_estimation.cpp:232-233) */
} nid_config;

/* ---- life cycle -------------------------------------------------------- */
int nid_abi_version(void);
const char *nid_status_string(int status);
const char *nid_last_error(const nid_ctx *ctx); /* text of the last HIP error */
int nid_device_count(void);                     /* 0 when no GPU; never initialises a context */

int nid_create(const nid_config *cfg, nid_ctx **out);
/* the same for a STRIDED cell set: the context owns cells cell_begin, cell_begin + stride, ... < cell_end
 * (interleaved shards of include/nid/nid_multi.h); stride 1 = nid_create */
int nid_create_strided(const nid_config *cfg, int32_t cell_stride, nid_ctx **out);
int nid_destroy(nid_ctx *ctx);
int nid_set_options(nid_ctx *ctx, int jac_bound_mode, int xform_mode);
int nid_set_math_mode(nid_ctx *ctx, int mode);
/* bs_value of nid_compute_href[_matrix]: by default a pixel without a sample (invalid depth, out of frame at the pose) has
 * four zero weights (the CPU edge's convention, types_six_dof_expmap.cpp:655-725); on != 0: four NaNs, the CUDA operator's
 * convention (CudaComputeHref.cu:82-87, 126-130) -- written on the device, so that the legacy wrapper need not pass over
 * the 9.8 MB again.  The device-resident weights are not affected. */
int nid_set_href_nan_markers(nid_ctx *ctx, int on);
/* run every kernel of this context on a caller-owned hipStream_t (e.g. the
 * current torch stream) instead of the context's own stream; NULL restores it.  NID_ERR_STATE while a launch
 * is uncollected (nid_wait it first); the streams used so far are drained before the switch, so that work of this
 * context never runs on two unordered streams at once.  With an external stream every launch goes to THAT stream
 * (the pipelined calls stop alternating with the context's second stream). */
int nid_set_stream(nid_ctx *ctx, void *hip_stream);
/* Threads per workgroup of the evaluation kernel (one workgroup per cell and pose): 128, 256, 512 or 1024.
 * jac_threads: cost + Jacobian launches; 0 = 128, the throughput shape.  The six Jacobian sums depend on the
 * shape in their last bits, so every launch of a context uses the same one.  A blocking caller that evaluates
 * one pose at a time (an LM loop) wants 1024: a cell's pixels in two rounds instead of ten.
 * cost_threads: cost-only launches; 0 = chosen per launch by its size (their results are the same bits in every
 * shape).  NID_ERR_UNSUPPORTED for any other value.
 * The 512 / 1024 shapes exist for launches of up to 16 poses (they are latency shapes).  A launch of MORE than 16
 * poses on a context set to 512 / 1024 runs with 256 threads: a cost + Jacobian evaluation in such a launch
 * (nid_launch_batch, nid_run_sequence) then differs from the same pose evaluated alone in the last bits of H and b
 * (chi2 and the active count are the same bits in every shape).  Callers that need identical bits across both keep
 * jac_threads at 0, 128 or 256. */
int nid_set_launch_shape(nid_ctx *ctx, int jac_threads, int cost_threads);
/* Diagnostics: FAST launches of the 512 / 1024 shapes run the latency form of the pixel loops (rounds unrolled and
 * staged, the Jacobian phase fed from registers) whenever its rounds cover a cell; on != 0 forces the loop form the
 * other shapes use.  Same bits either way (tests/test_parity_gpu.py::test_latency_form_equals_loop_form). */
int nid_set_loop_form(nid_ctx *ctx, int on);
/* DIRECT results (default on): a launch of ONE pose whose result the host waits for (nid_launch + nid_wait,
 * nid_normal_equations, nid_evaluate, nid_run_chain, the first pose of nid_launch_chain) lets every cell's workgroup
 * write a 64-byte record -- err, J[6], an active flag -- straight to pinned host memory; the HOST applies the Huber
 * kernel, forms each cell's quadratic form (the kernel's operations in the kernel's order, all IEEE) and adds them up
 * in the order the in-launch reduction uses: the same bits, without that reduction's device-scope round trips behind
 * the last cell.  (So in this mode the last step of the path -- robustify + J^T w J + the 256-way sum, ~2 us of host
 * work -- runs on the CPU; on == 0 keeps it on the device.)  The slot's cellout_dev block is written either way, its
 * reduced_dev block by neither (nid_slot_buffers).  on == 0: always the in-launch reduction.
 * on == 2, GROUP-DIRECT (nid_launch / nid_normal_equations / nid_run_chain): Huber, quadratic form and the first level
 * of the reduction stay on the device; every group's sum (256 bytes, at most 32 groups) goes straight to pinned host
 * memory and the host adds the groups up -- the whole of a15 on the GPU except the last <= 32 additions per entry, one
 * device-scope round trip instead of two behind the last cell.  Same bits.  The resident evaluator is not used in this
 * mode.  (Measured: profiles/r04_latency_A.txt; the default stays 1, which is faster.)
 * Timed and diagnostic launches always use the in-launch reduction.  NID_ERR_INVALID_ARG for any other value.
 * NID_ERR_STATE while a launch is pending. */
int nid_set_direct_results(nid_ctx *ctx, int on);
/* The RESIDENT evaluator (default off): DIRECT single-pose launches of a context in FAST math whose cost + Jacobian
 * shape is 512 threads (nid_set_launch_shape) and whose cells fit that shape's latency form (at most 1536 slots) -- or
 * 256 threads, see below -- are
 * not launched at all -- a kernel started once per frame pair keeps one workgroup per cell on the device; the host writes the pose
 * into a mailbox in device memory (through the PCIe BAR), every workgroup evaluates its cell and writes its record to
 * pinned host memory, and waits for the next request.  Same bits as the launched form; ~4 us less per dependent
 * evaluation (no runtime call, no packet, no dispatch of the grid).  The kernel is bounded: it leaves on an exit word
 * (every call that changes the frame pair, the options or the shapes, and nid_destroy, retire it), when the host has
 * not used it for 50 ms (the next request restarts it), and by itself after 200 ms without a request; a request that
 * finds it gone is re-issued as an ordinary launch.  One request in flight at a time; further single-pose launches
 * meanwhile are ordinary launches.  While it runs, its workgroups hold two waves per SIMD with up to 256 registers
 * each on every CU they sit on: every ordinary evaluation launch of the context retires it first (its request in
 * flight, if any, is collected before), so mixing the two costs a restart each time -- it pays for chains of
 * single-pose evaluations, which is what a Gauss-Newton / LM loop is.  A process-wide device synchronisation (hipDeviceSynchronize,
 * hipFree outside this library) waits until the kernel leaves -- at most its 200 ms.
 * One workgroup per cell has to be ON the device at once.  In the 512-thread shape a workgroup fills a CU: a context of
 * more cells than the device has CUs (1024 cells on 256 CUs: BASELINE configs[1] at 1280x960) is answered by ordinary
 * launches -- the first request finds that out, nid_resident_stats keeps saying served == 0, and later
 * nid_set_resident(ctx, 1) calls in that shape return NID_ERR_UNSUPPORTED with the reason in nid_last_error.
 * In the 256-thread shape (nid_set_launch_shape(ctx, 256, 0): the loop form of the kernels, at most 128 registers) four
 * resident workgroups share a CU: contexts of up to 4 x CUs cells are served.  Measured (profiles/r04_latency_B.txt): it
 * pays for cost-only requests (configs[1]: 23.3 us instead of 25.1; 640x480: 13.9 instead of 17.2) and is a wash for
 * cost + Jacobian ones (36.6 vs 35.4 us; 21.3 vs 22.2) -- four workgroups per CU move through the phases of an evaluation
 * in lockstep, which is what a single pose of 1024 cells costs, launched or not.
 * Sharing the device: the co-residency of one workgroup per cell is an assumption about the WHOLE device, so
 *  - one resident kernel per device and process: while one context's kernel is on a device, other contexts' requests
 *    on that device are answered by ordinary launches (the owner's pause / retire / destroy frees the place);
 *  - a request the kernel does not pick up within its deadline (another process's work holds CUs: some workgroups
 *    are not on the device) is re-issued as an ordinary launch and counted in `fallbacks`; after 3 such requests in
 *    a row, or once there are more than 8 and more than the requests served, the mode switches itself off for the
 *    context (reason in nid_last_error; nid_set_resident(ctx, 1) arms it again).
 * NID_ERR_UNSUPPORTED also if the device's memory is not CPU-addressable as a whole (no large PCIe BAR: the
 * hipDeviceAttributeIsLargeBar attribute; nothing is probed by touching the mapping);
 * NID_ERR_STATE while a launch is pending. */
int nid_set_resident(nid_ctx *ctx, int on);
/* Takes a running resident kernel off the device (it holds most of every CU) without disabling the mode: for a caller
 * that is done with this context for now and lets others use the GPU.  The next request starts another. */
int nid_resident_pause(nid_ctx *ctx);
/* requests served by the resident kernel, requests re-issued as ordinary launches, kernel starts */
int nid_resident_stats(const nid_ctx *ctx, int64_t *served, int64_t *fallbacks, int64_t *starts);
/* (Round 5's resident BATCH evaluator -- nid_set_resident(ctx, 2), a persistent grid answering requests of 2..64 poses --
 * measured slower than launches at every length (20 poses: 146 us against 88 us, profiles/r05_short_sequences.txt: a
 * persistent grid walks through the phases of an evaluation in lockstep) and was removed in round 6; `on` is 0 or 1.) */
/* both at once (0 = the defaults above) */
int nid_set_block_threads(nid_ctx *ctx, int threads);

/* ---- once per frame pair ------------------------------------------------ */
/* Calculate3Dpoint: depth in metres (f64, rows*cols), T_wc0 column-major.
 * Back-projects on the device into the cell-tiled layout. */
int nid_set_reference_depth(nid_ctx *ctx, const double *depth_m, const uint8_t *im0,
                            const double *T_wc0_colmajor16);
/* same, from an existing AoS point cloud (NaN xyz = invalid), as handed to
 * CudaComputeHref / CudaComputeH */
int nid_set_reference_points(nid_ctx *ctx, const double *points3d, const uint8_t *im0);
/* One frame pair in the formats the reference's driver reads from disk (NID_pose_estimation.cpp:84-113: u8 grey
 * images, u16 depth whose metres are value * depth_factor, :73,106), handed over in ONE call: what Calculate3Dpoint
 * (CudaPoints3d.cu:35-73), the reference upload, the target upload and CudaComputeHref (CudaComputeHref.cu:139-222) do
 * together, with 1.2 MB across PCIe at 640x480, one stream, one synchronisation, and only bs_counter / Href (may be
 * NULL) coming back.  The reference stage runs at pose0 -- given as pose7 OR as a column-major 4x4 (exactly one of the two
 * non-NULL; the matrix form is what the legacy operators take).  Same device state, bit for bit, as
 * nid_set_reference_depth(depth_u16 * depth_factor) + nid_set_target_u8 + nid_compute_href[_matrix]. */
int nid_set_pair_u16(nid_ctx *ctx, const uint16_t *depth_u16, double depth_factor, const uint8_t *im0, const uint8_t *im1,
                     const double *T_wc0_colmajor16, const double *pose0_7, const double *pose0_colmajor16,
                     int32_t *bs_counter, double *Href);
/* context-free Calculate3Dpoint (CudaPoints3d.cuh:6): depth f64 metres -> AoS world points, NaN = invalid */
int nid_backproject(const double *depth_m, const double *T_wc0_colmajor16, double fx, double fy, double cx,
                    double cy, int32_t rows, int32_t cols, int32_t device, double *points3d);
/* nid_backproject keeps device and pinned scratch (sized by its largest image) from call to call, one per process, its
 * callers serialised by a mutex; this frees it (nid_legacy_reset does) */
int nid_backproject_release(void);
/* copy the back-projected points back in Calculate3Dpoint's output layout */
int nid_get_points3d(nid_ctx *ctx, double *points3d);
int nid_set_target_u8(nid_ctx *ctx, const uint8_t *im1);
/* legacy f64 image carrying u8 values ((double)im.data[i], NID_pose_estimation.cpp:245-251);
 * NID_ERR_UNSUPPORTED if a value is not an integer in [0,255] */
int nid_set_target_f64(nid_ctx *ctx, const double *im1);
int nid_set_reference_image_f64(const double *im0, int64_t n, uint8_t *out_u8);

/* CudaComputeHref at the initial pose.  bs_counter[cells], Href[cells] (NaN =
 * inactive); bs_value (4 per pixel) and bs_index (1 per pixel) may be NULL;
 * when given they are filled in image order with the CPU-edge convention
 * (weights 0 for pixels out of frame at the initial pose, bs_index = -1 for
 * invalid depth). */
int nid_compute_href(nid_ctx *ctx, const double *pose7, int32_t *bs_counter, double *Href,
                     double *bs_value, int32_t *bs_index);
int nid_compute_href_matrix(nid_ctx *ctx, const double *pose_colmajor16, int32_t *bs_counter,
                            double *Href, double *bs_value, int32_t *bs_index);
/* install externally computed reference weights (the legacy CudaComputeH gets
 * bs_counter / bs_ref / bs_index_ref / Href from its caller every call) */
int nid_set_href_state(nid_ctx *ctx, const int32_t *bs_counter, const double *Href,
                       const double *bs_value, const int32_t *bs_index);

/* ---- per Gauss-Newton / LM iteration ------------------------------------ */
/* g2o::CudaComputeH twin: Htarget[cells], Hjoint[cells], err[cells] =
 * (2*Hj - Href - Ht)/Hj, der[6*cells] (only when want_jac).  Output pointers
 * may be NULL.  Only the context's own cell range is written.  Blocking. */
int nid_evaluate(nid_ctx *ctx, const double *pose7, int want_jac, double *Htarget,
                 double *Hjoint, double *err, double *der);
int nid_evaluate_matrix(nid_ctx *ctx, const double *pose_colmajor16, int want_jac,
                        double *Htarget, double *Hjoint, double *err, double *der);

/* Fused: evaluate + Huber-weighted reduction over the context's cells on the
 * device.  H36 row-major full 6x6, b6 (g2o sign: b = -sum rho1 J^T e),
 * chi2 = sum rho0, n_active.  Blocking. */
int nid_normal_equations(nid_ctx *ctx, const double *pose7, int want_jac, double huber_delta,
                         double *H36, double *b6, double *chi2, int32_t *n_active);

/* Non-blocking forms for pipelining and multi-GPU.  `slot` in [0, NID_SLOTS):
 * results of a launch stay in the slot until nid_wait() collects them. */
#define NID_SLOTS 1024
#define NID_MAX_BATCH 256 /* poses per launch (up to 16 travel as kernel arguments, more through a device array) */
#define NID_REDUCED_LEN 32 /* [0]=chi2 [1..6]=b [7..27]=H upper triangle row-major [28]=n_active */
int nid_launch(nid_ctx *ctx, int slot, const double *pose7, int want_jac, double huber_delta);
int nid_wait(nid_ctx *ctx, int slot, double *H36, double *b6, double *chi2, int32_t *n_active);
/* n <= NID_MAX_BATCH candidate poses (poses7 = n x 7) in ONE kernel launch, into slots
 * first_slot .. first_slot+n-1 (e.g. the trial steps of one LM iteration for several lambdas,
 * or consecutive candidates of a sampling optimiser); collect each with nid_wait() */
int nid_launch_batch(nid_ctx *ctx, int first_slot, int n, const double *poses7, int want_jac,
                     double huber_delta);
/* SHORT sequences: nid_launch_batch and nid_run_sequence split a sequence of <= 64 poses whose results the host
 * collects into launches of `poses_per_launch` poses, alternating between the context's two streams when `streams` is
 * 2 (1: one stream).  poses_per_launch = 0 (the default): the library's measured table (two or more launches of <= 16
 * poses -- their records ride in the kernel arguments -- beside each other).  Results do not depend on the policy. */
int nid_set_short_sequence_policy(nid_ctx *ctx, int poses_per_launch, int streams);
/* The rejection chain of a Levenberg-Marquardt iteration in ONE call: poses 0 .. n_jac-1 are evaluated WITH the
 * Jacobian phase (the likely accepted trials: their H and b are the next iteration's linear system), the rest cost
 * only, as two concurrent launches on the context's two streams (with a caller's stream: back to back).  Slot by slot
 * the results are the bits nid_launch_batch gives for the same pose and want_jac.  0 <= n_jac <= n. */
int nid_launch_chain(nid_ctx *ctx, int first_slot, int n, const double *poses7, int n_jac, double huber_delta);
/* same, with the n result blocks written to a caller-owned DEVICE buffer (n x NID_REDUCED_LEN doubles,
 * pose k at offset k*NID_REDUCED_LEN) so that ONE collective can sum the partial blocks of all n poses
 * across the cell shards of several GPUs; nid_wait(slot) then only waits for the launch */
int nid_launch_batch_to(nid_ctx *ctx, int first_slot, int n, const double *poses7, int want_jac,
                        double huber_delta, void *reduced_dev);
/* Host-side pipeline over a sequence of n candidate poses: `batch` poses per launch (batch
 * divides NID_SLOTS), min(16, NID_SLOTS/batch) launches in flight on the context's two streams, each
 * launch's result blocks written to a device buffer and brought home by one copy behind it;
 * reduced_out (n x NID_REDUCED_LEN, may be NULL) receives every pose's [chi2, b, H upper, n_active]
 * block.  Blocking. */
int nid_run_sequence(nid_ctx *ctx, const double *poses7, int n, int batch, int want_jac,
                     double huber_delta, double *reduced_out);
/* A DEPENDENT chain of n evaluations, the way a Gauss-Newton / LM loop issues them: one pose per launch, the host
 * waits for each result before it launches the next (slot 0).  reduced_out (n x NID_REDUCED_LEN) may be NULL;
 * *seconds (may be NULL) receives the wall time of the chain.  Blocking. */
int nid_run_chain(nid_ctx *ctx, const double *poses7, int n, int want_jac, double huber_delta, double *reduced_out,
                  double *seconds);
/* device address of slot's reduced block (NID_REDUCED_LEN doubles) and of its
 * per-cell block (cells_local x NID_CELL_OUT doubles: Hc,Hj,err,J[6], n_c) so
 * that a caller can run a collective on them (RCCL all-reduce / all-gather).  Who writes them:
 *  - cellout_dev: every LAUNCHED evaluation of the slot through nid_launch / nid_launch_to / nid_launch_batch(_to) /
 *    nid_launch_chain (DIRECT or not), complete when nid_wait(slot) returns.  Not the pipelined loops
 *    (nid_run_sequence / nid_run_chain hand the kernels no per-cell block), not nid_evaluate (its per-cell values go
 *    straight to the caller), and not a request the RESIDENT evaluator answered (nid_set_resident: nothing is
 *    launched; its records go to the host);
 *  - reduced_dev: nobody, unless the caller names it as the target of nid_launch_to / nid_launch_batch_to (it is a
 *    device block the context owns for that purpose: slot k's is good for one pose, a batch of n needs n blocks of
 *    the caller's).  Every other launch reports through pinned host memory (the in-launch reduction's last workgroup
 *    writes the block there; a DIRECT launch forms it on the host) and nid_wait hands that over. */
#define NID_CELL_OUT 10
int nid_slot_buffers(nid_ctx *ctx, int slot, void **reduced_dev, void **cellout_dev);
/* test / tool helper: blocking copy of `bytes` from a device address of this context's device */
int nid_debug_read_device(nid_ctx *ctx, const void *dev, void *host, size_t bytes);
/* launch variant that writes the reduced block into a caller-owned DEVICE
 * buffer (e.g. a torch tensor that torch.distributed all-reduces) */
int nid_launch_to(nid_ctx *ctx, int slot, const double *pose7, int want_jac, double huber_delta,
                  void *reduced_dev);
int nid_unpack_reduced(const double *reduced, double *H36, double *b6, double *chi2,
                       int32_t *n_active);

/* ---- introspection / tests ---------------------------------------------- */
/* per-pixel intermediates of the most recent nid_evaluate* in image order
 * (NaN / -1 where a pixel was not visited); any pointer may be NULL.
 * enable = 1: the cost phase -- u, v, ic, jc, wc4 as named;
 * enable = 2: the Jacobian phase of a want_jac evaluation, for every sample that contributes -- in the same
 *             arrays: u <- gx, v <- gy (central difference / 2), ic <- bin position, jc, wc4 <- the four
 *             B-spline derivatives (all zero at bin position 0, Q5) */
int nid_debug_enable_pixel_dump(nid_ctx *ctx, int enable);
int nid_debug_get_pixel_dump(nid_ctx *ctx, double *u, double *v, double *ic, int32_t *jc,
                             double *wc4);
/* diagnostic: s_memtime stamps of wave 0 of every workgroup at the phase
 * boundaries of the evaluation kernel, [cells_local][10]: 8 phase stamps, then
 * s_memrealtime (100 MHz) at kernel start and end of the same wave */
int nid_debug_enable_stamps(nid_ctx *ctx, int enable);
int nid_debug_get_stamps(nid_ctx *ctx, int64_t *stamps);
/* diagnostic: how many (cell, pose) evaluations of this context ran the REPAIR pass of the histogram fold (a bin of an
 * end span's linear-weight column rebuilt in the fine fixed-point levels: kLinFlagW in csrc/nid_kernels.hip.h) since
 * creation or the last reset.  Waits for the device. */
int nid_debug_repair_count(nid_ctx *ctx, int64_t *count, int reset);
/* host evaluation of the closed-form B-spline used by the kernels */
void nid_bspline4_host(double u, int bin_num, double *B4, double *D4);
/* host evaluation of the FAST-mode per-span polynomial B-spline table */
void nid_bspline4_poly_host(double u, int bin_num, double *B4, double *D4);
/* Plain-histogram NID per cell at pose7 (T_cw1): NID::ComputeHref + NID::ComputeH of the reference's second
 * program (NID_standard_property.cpp:342-485): hard binning floor(I*bins/255), no B-spline, no Jacobian.
 * Arrays are indexed by global cell id (only this context's cells are written), any may be NULL; cells
 * with fewer than 300 in-frame pixels get NaN in Hcur / Hjoint / nid / mi.  *total = sqrt(sum nid^2) over
 * this context's cells (the program's "final nid").  Needs the reference and the target, not the href state. */
int nid_plain_nid(nid_ctx *ctx, const double *pose7, int bins, double *Href, double *Hcur, double *Hjoint,
                  double *nid, double *mi, int32_t *n_in, double *total);
/* host twin of the FAST-mode log2 of the entropy fold */
double nid_log2_fast_host(double x);
/* host twin of the kernels' division-by-small-constant helper */
double nid_div_small_host(double x, double d);
/* timing of a slot's last launch, ms (hipEvent; launches issued while nid_enable_timing is on): eval_ms = the evaluation kernel,
 * events right in front of and right behind it (ABI 4: in front of the repair kernel that follows a loop-form launch) */
int nid_last_kernel_ms(nid_ctx *ctx, int slot, float *eval_ms, float *reduce_ms);
int nid_enable_timing(nid_ctx *ctx, int enable);
/* average duration (ms) of `repeats` identical n-pose launches issued back to back on the context's
 * stream, from one pair of HIP events around the whole group: what a launch costs with nothing beside it -- the in-stream copy
 * of the per-pose records, the evaluation kernel, the repair kernel behind it, the dispatch gaps */
int nid_time_launches(nid_ctx *ctx, int n, const double *poses7, int want_jac, double delta, int repeats,
                      float *ms_per_launch);
/* average duration (ms) of the evaluation kernel ALONE over `repeats` n-pose launches issued one at a time: one HIP event
 * right in front of the kernel (behind the in-stream copy of the per-pose records) and one right behind it (in front of the
 * repair kernel that follows every loop-form launch) -- the figure a kernel trace reports for that kernel; nid_time_launches
 * brackets whole launches back to back (copy + kernel + repair kernel + dispatch gaps: about 3 % more at 256 poses).
 * nid_enable_timing / nid_last_kernel_ms report the same interval per launch.  (ABI 4) */
int nid_time_kernel(nid_ctx *ctx, int n, const double *poses7, int want_jac, double delta, int repeats, float *ms_kernel);
/* algorithmic (contract) bytes of one evaluation over this context's cells:
 * 68 B per pixel + 64 B per cell */
int64_t nid_contract_bytes(const nid_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
