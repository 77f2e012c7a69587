// legacy_ops.h -- the reference's three operator entry points, same names, same
// argument lists, same C++ linkage, implemented on the HIP library:
//
//   void Calculate3Dpoint(...)     replaces CudaPoints3d.cuh:6    (CudaPoints3d.cu:35-73)
//   void CudaComputeHref(...)      replaces CudaComputeHref.cuh:5 (CudaComputeHref.cu:139-222)
//   void g2o::CudaComputeH(...)    replaces g2o/g2o/core/computeH.cuh:8 (computeH.cu:373-502)
//
// A maintainer of the reference swaps the three .cu files for legacy_ops.cpp and
// links libnid_hip.so; NID_pose_estimation.cpp and the g2o hooks
// (optimization_algorithm_levenberg.cpp:98,173; sparse_optimizer.cpp:423) call
// them unchanged.
//
// Conventions kept: column-major 4x4 poses; camera_intrincis =
// {fx, fy, cx, cy, depth_factor}; images row-major f64 holding u8 values;
// points3d AoS xyz with NaN = invalid; Htarget/Hjoint are in/out (the callee
// subtracts onto caller-zeroed storage, computeH.cu:285,298); der is untouched
// when calculate_der == false; pro_target / pro_joint are accepted and ignored.
// Errors are printed to stderr and execution continues, like the reference
// (computeH.cu:454-473) -- the signatures return void.
//
// Differences, all deliberate: no per-call allocation / memset / re-upload.  The
// frame-pair state is cached per (rows, cols, cell_num, bin_num) and re-uploaded
// when the CONTENT of a caller buffer has changed.
// THE CONTRACT: none beyond the reference's.  The caller owns every buffer; the operators read them only
// between their entry and their return (round 6), so a caller may free or rewrite them the moment a call
// is back -- NID_pose_estimation.cpp:388-395 frees everything right after the last CudaComputeH.  How the
// operators make sure the device holds what the buffers hold NOW (nid_legacy_set_verify_mode):
//  * NID_LEGACY_VERIFY_ROTATING (default): a call checks address, length and 64 samples of each big
//    buffer (a microsecond in all) and hands a few of the 128 slices of im0 / points3d / im1 / bs_ref -- 1/32
//    of every buffer on average, 0.7 MB at 640x480; 6 slices with the Jacobian, 3 without: what hides behind
//    the call's own evaluation -- to a small pool of worker threads (NID_LEGACY_HASH_THREADS, default 3; 0:
//    the caller hashes alone), evaluates on the device meanwhile, and JOINS the workers before it returns.
//    A buffer rewritten IN PLACE is found within 43 calls (35 in the LM's pattern; the reference's LM makes
//    40-60 per pair): that call says so on stderr, counts it (nid_legacy_stale_detections), uploads the new
//    content and evaluates again before it returns.  +3-5 us per call (profiles/r06_legacy_call_cost.txt).
//    nid_legacy_set_verify_slices(k) / NID_LEGACY_VERIFY_SLICES=k: k slices per call instead of 4.
//  * NID_LEGACY_VERIFY_EVERY_CALL (= all 128 slices on every call; NID_LEGACY_VERIFY_EVERY_CALL=1): every call
//    hashes all four buffers in full and evaluates what they hold now -- the reference's guarantee exactly
//    (it uploads everything on every call, computeH.cu:420-429), at 0.4 ms per call at 640x480.
//  * NID_LEGACY_VERIFY_TRUSTED (nid_legacy_set_trust_buffers(1), NID_LEGACY_TRUST_BUFFERS=1): the cheap
//    check only; full hashes when it fails, on every 128th call of a pair, and after
//    nid_legacy_invalidate(parts).  For callers that do not rewrite buffers in place, or say so.
// bs_counter / Href (1-2 KB) are hashed in full on every call in every mode; a new frame pair (new
// buffers, or CudaComputeHref) is always noticed at once.  NID_LEGACY_ALWAYS_UPLOAD=1 uploads everything
// on every call.  No thread of the library touches a caller buffer outside a call in any mode.
// Out-of-frame reference weights are NaN in the
// arrays handed back (as CudaComputeHref.cu:126-130 writes them) but are treated
// as 0 inside, the CPU edge's convention; the Jacobian in-frame
// test defaults to the CPU edge's `cols-1` (the parity target),
// nid_legacy_set_jacobian_bound(1) selects the CUDA kernel's `cols`.
#ifndef NID_LEGACY_OPS_H
#define NID_LEGACY_OPS_H

#include <stdint.h>

void Calculate3Dpoint(double *depth, double *pose_c2w, double *points_3d, double *camera_intrincis,
                      int rows, int cols);

void CudaComputeHref(double *im0, double *points3d, double *pose, double *camera_intrincis, int bin_num,
                     int bs_degree, int cell_num, int rows, int cols, double *bs_value, int *bs_index,
                     int *bs_counter, double *Href);

namespace g2o {
void CudaComputeH(bool calculate_der, double *im0, double *im1, double *points3d, int *bs_counter,
                  double *bs_ref, int *bs_index_ref, double *pose, double *camera_intrincis, int bin_num,
                  int bs_degree, int cell_num, int rows, int cols, double *Href, double *pro_target,
                  double *pro_joint, double *Htarget, double *Hjoint, double *der);
}

extern "C" {
struct nid_ctx;
struct nid_multi;
// 0 = CPU-edge bound (default), 1 = CUDA-kernel bound
void nid_legacy_set_jacobian_bound(int mode);
// 0 = FAST arithmetic (default), 1 = STRICT: every rounding of the reference reproduced (nid_c.h)
void nid_legacy_set_math_mode(int mode);
void nid_legacy_set_device(int device);
// threads per workgroup of the operators' launches (nid_set_launch_shape in nid_c.h); jac_threads -1 (default): 512 with
// up to 256 cells per shard, 256 beyond
void nid_legacy_set_launch_shape(int jac_threads, int cost_threads);
// 1: the operators' single-pose evaluations are answered by the resident evaluator (nid_set_resident in nid_c.h:
// a kernel that stays on the device between calls; FAST math, 512-thread shape, i.e. up to 256 cells per shard);
// silently the launched form wherever it does not apply.  Default 0; also switched on by NID_LEGACY_RESIDENT=1.
void nid_legacy_set_resident(int on);
// Multi-GPU (include/nid/nid_multi.h): the operators shard the cells of the frame pair over `n` devices of THIS
// process (entries may repeat); reduce_rccl != 0 sums the fused 6x6 blocks with RCCL instead of on the host ...
void nid_legacy_set_devices(const int32_t *devices, int n, int reduce_rccl);
// ... or this process is rank `rank` of `world` (one process per GPU); rccl_id128 = the ncclUniqueId of the job
// (nid_multi_comm_unique_id on rank 0, handed to every rank).  The communicator is created on first use and kept
// until the device set changes.
void nid_legacy_set_rank(int device, int rank, int world, const uint8_t *rccl_id128);
// drop every cached context and the device memory behind it (optional: nothing of the caller's is referenced between calls)
void nid_legacy_reset(void);
// done with the GPU for now, keep everything for the next frame pair: takes a running resident kernel off the device
// (the context, its buffers and its communicator stay: the next pair of the same geometry costs no context creation)
void nid_legacy_quiesce(void);
// how the operators make sure the device holds what the caller's buffers hold (THE CONTRACT above)
#define NID_LEGACY_SLICES 128        /* slices per big buffer (a content key holds one hash per slice) */
#define NID_LEGACY_SLICES_PER_CALL 4 /* ROTATING checks this many of them per call: 1/32 of every buffer */
enum { NID_LEGACY_VERIFY_ROTATING = 0, NID_LEGACY_VERIFY_EVERY_CALL = 1, NID_LEGACY_VERIFY_TRUSTED = 2 };
void nid_legacy_set_verify_mode(int mode);
void nid_legacy_set_verify_slices(int per_call);  // ROTATING: slices (of NID_LEGACY_SLICES) per call; < 1: NID_LEGACY_SLICES_PER_CALL
void nid_legacy_set_trust_buffers(int on);   // 1: NID_LEGACY_VERIFY_TRUSTED, 0: the default
// in-place changes the rotating verification has found and reported so far
long nid_legacy_stale_detections(void);
// The caller has changed, IN PLACE, the content of the buffers named by `parts` since its last call (needed with trusted
// buffers; spares the rotating mode its detection delay):
// the next CudaComputeH recomputes their full hashes (and uploads what differs) instead of trusting address + samples.
enum { NID_LEGACY_REFERENCE = 1 /* im0, points3d */, NID_LEGACY_TARGET = 2 /* im1 */, NID_LEGACY_HREF_STATE = 4 /* bs_ref */ };
void nid_legacy_invalidate(unsigned parts);
// the context the legacy calls are currently using (NULL before the first call); lets a host
// mix the legacy operators with the fused C-ABI entry points on the same device state
nid_ctx *nid_legacy_context(void);   // shard 0
nid_multi *nid_legacy_multi(void);
// What CudaComputeH does before its kernels, without the evaluation: the frame-pair state (images, points, Href
// state) on the device(s), uploaded only where its content changed.  Returns the shards (NULL on error, reported on
// stderr) for callers that continue on the nid_multi_* interface (g2o_min's fused LM).
nid_multi *nid_legacy_prepare(double *im0, double *im1, double *points3d, int *bs_counter, double *bs_ref,
                              int *bs_index_ref, double *camera_intrincis, int bin_num, int bs_degree, int cell_num,
                              int rows, int cols, double *Href);
// A whole frame pair in the formats the reference's driver reads (u16 depth, u8 images; NID_pose_estimation.cpp:84-113) on
// the operators' context -- nid_multi_set_pair_u16 (nid_multi.h): back-projection, tiling and the reference stage at
// pose0 on the device, bs_counter / Href (per cell, NaN = inactive; plain values, not subtracted onto anything) back --
// for hosts that continue on the nid_multi_* interface and read none of the operators' per-pixel arrays.
// camera_intrincis = {fx, fy, cx, cy, depth_factor}.  NULL on error (reported on stderr).
nid_multi *nid_legacy_set_pair_u16(const uint16_t *depth_u16, const uint8_t *im0, const uint8_t *im1, const double *T_wc0_colmajor16,
                                   const double *pose0_colmajor16, const double *camera_intrincis, int bin_num, int bs_degree,
                                   int cell_num, int rows, int cols, int32_t *bs_counter, double *Href);
// number of host->device uploads of frame-pair data done so far (tests: must not grow per call)
long nid_legacy_upload_count(void);
}

#endif
