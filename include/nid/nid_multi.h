/*
 * nid_multi.h -- the NID path over the GPUs of one node, behind the same C-ABI (libnid_hip.so).
 *
 * The reference has no multi-GPU path; north_star: "partition cells across the 8 GPUs of one node with an
 * RCCL all-reduce over xGMI of the per-cell 6x6/6x1 blocks; the g2o solver itself stays on the host"
 * (callers NID_pose_estimation.cpp:253-351,
 * g2o/g2o/core/optimization_algorithm_levenberg.cpp:98,173).
 *
 * Given the pose, every cell's (Hc, Hj, err, J) depends only on that cell's pixels, so cells shard:
 * shard k of K owns the contiguous cell range [k*cells/K, (k+1)*cells/K) (nid_multi_cell_range).  A
 * nid_multi owns one shard context (nid_ctx, include/nid/nid_c.h) per entry of a device list -- all of them
 * in ONE process (nid_multi_create; entries may repeat, e.g. {0,0} puts two shards on one GPU) -- or the one
 * shard of this process in a job of one process per GPU (nid_multi_create_rank).  The frame pair is replicated
 * where every shard needs it (target image) and split where it is owned (reference tile of the shard's cells).
 * An evaluation is one kernel launch per shard for all candidate poses plus ONE exchange of the partial
 * [chi2, b(6), H upper(21), n_active] blocks, 32 doubles per pose:
 *   NID_REDUCE_HOST  every shard's last workgroup writes its partial blocks to pinned host memory and the host
 *                    adds them in shard order (single process only): no collective at all, the lowest latency
 *                    when the solver sits on the host anyway, bitwise reproducible;
 *   NID_REDUCE_RCCL  the partial blocks stay on the device, ncclAllReduce(ncclDouble, ncclSum) over xGMI sums
 *                    them in-stream across the shards of this process (ncclCommInitAll: distinct devices) or
 *                    across the processes of the job (ncclCommInitRank with an id the ranks exchange), then
 *                    the result is copied to pinned host memory.  librccl is loaded on first use (from the directory of the
 *                    HIP runtime the process runs on; NID_RCCL_LIBRARY=<path> names another one).
 * Per-cell outputs (the legacy CudaComputeH contract) need no exchange inside a process -- each shard writes
 * its own cell range of the caller's arrays --; across processes they are summed by RCCL over a zero-filled
 * per-cell buffer (NaN = inactive survives the sum).
 *
 * Same conventions and error codes as nid_c.h; not thread safe; every rank of a multi-process job must issue
 * the same calls in the same order (every call that touches RCCL is collective).
 */
#ifndef NID_MULTI_H
#define NID_MULTI_H

#include "nid/nid_c.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nid_multi nid_multi;

#define NID_REDUCE_HOST 0
#define NID_REDUCE_RCCL 1
#define NID_REDUCE_HOOK 2 /* host sum inside the process, then a caller-supplied exchange across processes */
#define NID_MAX_SHARDS 64
#define NID_RCCL_ID_BYTES 128 /* sizeof(ncclUniqueId) */

/* [lo, hi) of shard `k` of `n` over `ncell` cells: k*ncell/n .. (k+1)*ncell/n.  NID_ERR_INVALID_ARG if n > ncell. */
int nid_multi_cell_range(int32_t k, int32_t n, int32_t ncell, int32_t *lo, int32_t *hi);

/* One process, n shards on devices[0..n-1] (cfg->device, cell_begin, cell_end are ignored). */
int nid_multi_create(const nid_config *cfg, const int32_t *devices, int32_t n, nid_multi **out);
/* One process per GPU: this process is rank `rank` of `world` and owns that shard on `device`.  Evaluations
 * need nid_multi_comm_init (world > 1). */
int nid_multi_create_rank(const nid_config *cfg, int32_t device, int32_t rank, int32_t world, nid_multi **out);
/* Which cells a shard owns.  CONTIGUOUS (the two calls above): shard k of K owns [k*cells/K, (k+1)*cells/K).
 * INTERLEAVED: shard k owns cells k, k + K, k + 2K, ... -- cells differ in cost (inactive ones cost nothing, border
 * ones less than interior ones) and cluster in the image, so contiguous ranges load the shards unequally (measured:
 * the last of 8 ranks holds every inactive cell of the test pair, profiles/r02_rank_rates.txt); interleaving spreads
 * them.  Results do not depend on the partition beyond the order of the partial sums.  `rank` / `world` as in
 * nid_multi_create_rank (world == 1: n shards in this process; world > 1: n must be 1). */
#define NID_PARTITION_CONTIGUOUS 0
#define NID_PARTITION_INTERLEAVED 1
/* The cells of shard `k` of `n` under `partition`: begin, begin + stride, ... < end (no device needed). */
int nid_multi_cell_partition(int32_t k, int32_t n, int32_t ncell, int32_t partition, int32_t *begin, int32_t *end, int32_t *stride);
int nid_multi_create_partitioned(const nid_config *cfg, const int32_t *devices, int32_t n, int32_t rank, int32_t world,
                                 int32_t partition, nid_multi **out);
int nid_multi_destroy(nid_multi *m);
const char *nid_multi_last_error(const nid_multi *m);
int nid_multi_shards(const nid_multi *m);          /* shards in THIS process */
nid_ctx *nid_multi_shard(nid_multi *m, int32_t k); /* borrowed */
int nid_multi_world(const nid_multi *m, int32_t *rank, int32_t *world);

/* ---- RCCL ---------------------------------------------------------------------------------------------- */
/* rank 0 calls this and hands the id to every rank (any side channel: MPI, a file, torch.distributed) */
int nid_multi_comm_unique_id(uint8_t id[NID_RCCL_ID_BYTES]);
/* A communicator that outlives the nid_multi objects it serves (one per process; e.g. every level of the
 * coarse-to-fine schedule creates its own nid_multi on the same devices).  create_rank is collective over the
 * processes of the job (ncclCommInitRank); create_local spans the distinct devices of one process
 * (ncclCommInitAll).  nid_multi_attach_comm borrows it and selects NID_REDUCE_RCCL. */
typedef struct nid_comm nid_comm;
int nid_comm_create_rank(const uint8_t id[NID_RCCL_ID_BYTES], int32_t rank, int32_t world, int32_t device, nid_comm **out);
int nid_comm_create_local(const int32_t *devices, int32_t n, nid_comm **out);
int nid_comm_destroy(nid_comm *c);
int nid_comm_ranks(const nid_comm *c, int32_t *nranks); /* ncclCommCount */
int nid_multi_attach_comm(nid_multi *m, nid_comm *c);
/* shorthands that create a communicator owned by the nid_multi: collective over the processes of the job ... */
int nid_multi_comm_init(nid_multi *m, const uint8_t id[NID_RCCL_ID_BYTES]);
/* ... or over the shards of this process (devices must be distinct) */
int nid_multi_comm_init_local(nid_multi *m);
int nid_multi_comm_ranks(const nid_multi *m, int32_t *nranks); /* ncclCommCount of the live communicator */
/* measurement: `repeats` exchanges of `blocks` 32-double blocks back to back on the comm stream(s), milliseconds per
 * exchange (0 without a live communicator: host sum / hook).  Collective: every rank of the job calls it. */
int nid_multi_time_exchange(nid_multi *m, int blocks, int repeats, float *ms_per_exchange);
/* Another transport instead of RCCL for a multi-process job (MPI, gloo, a test harness): after the shards of this
 * process have been summed on the host, `fn` must replace data[0..count) by its sum over all processes (0 = ok).
 * Collective like RCCL: every rank gets the same calls in the same order.  Selects NID_REDUCE_HOOK. */
typedef int (*nid_exchange_fn)(double *data, int64_t count, void *user);
int nid_multi_set_exchange_hook(nid_multi *m, nid_exchange_fn fn, void *user);
int nid_multi_set_reduce_mode(nid_multi *m, int mode);

/* ---- the nid_c.h calls, on every shard ------------------------------------------------------------------- */
int nid_multi_set_options(nid_multi *m, int jac_bound_mode, int xform_mode);
int nid_multi_set_math_mode(nid_multi *m, int mode);
int nid_multi_set_href_nan_markers(nid_multi *m, int on);
int nid_multi_set_block_threads(nid_multi *m, int threads);
int nid_multi_set_launch_shape(nid_multi *m, int jac_threads, int cost_threads);
/* nid_set_resident (nid_c.h) on every shard that has its device to itself among the shards of this process (two
 * resident kernels on one device would each be checked against the whole device); the other shards keep launching.
 * Returns the first shard's error. */
int nid_multi_set_resident(nid_multi *m, int on);
int nid_multi_resident_pause(nid_multi *m); /* nid_resident_pause on every shard */
int nid_multi_set_reference_depth(nid_multi *m, const double *depth_m, const uint8_t *im0, const double *T_wc0_colmajor16);
int nid_multi_set_reference_points(nid_multi *m, const double *points3d, const uint8_t *im0);
int nid_multi_set_target_u8(nid_multi *m, const uint8_t *im1);
/* nid_set_pair_u16 (nid_c.h) on every shard -- all of them enqueued before the first is waited for --, bs_counter /
 * Href for every cell of the image as with nid_multi_compute_href */
int nid_multi_set_pair_u16(nid_multi *m, const uint16_t *depth_u16, double depth_factor, const uint8_t *im0, const uint8_t *im1,
                           const double *T_wc0_colmajor16, const double *pose0_7, const double *pose0_colmajor16,
                           int32_t *bs_counter, double *Href);
/* bs_counter / Href: every cell of the image (summed across processes when a communicator is live);
 * bs_value / bs_index: the pixels of the cells this PROCESS owns */
int nid_multi_compute_href(nid_multi *m, const double *pose7, int32_t *bs_counter, double *Href, double *bs_value,
                           int32_t *bs_index);
int nid_multi_compute_href_matrix(nid_multi *m, const double *pose_colmajor16, int32_t *bs_counter, double *Href,
                                  double *bs_value, int32_t *bs_index);
int nid_multi_set_href_state(nid_multi *m, const int32_t *bs_counter, const double *Href, const double *bs_value,
                             const int32_t *bs_index);
/* per-cell outputs of every cell of the image (g2o::CudaComputeH twin), see the header comment */
int nid_multi_evaluate(nid_multi *m, const double *pose7, int want_jac, double *Htarget, double *Hjoint, double *err,
                       double *der);
int nid_multi_evaluate_matrix(nid_multi *m, const double *pose_colmajor16, int want_jac, double *Htarget,
                              double *Hjoint, double *err, double *der);
/* fused: the 6x6 system of the WHOLE image */
int nid_multi_normal_equations(nid_multi *m, const double *pose7, int want_jac, double huber_delta, double *H36,
                               double *b6, double *chi2, int32_t *n_active);
/* n <= NID_MAX_BATCH candidate poses in one launch per shard + one exchange; collect each with nid_multi_wait */
int nid_multi_launch_batch(nid_multi *m, int first_slot, int n, const double *poses7, int want_jac, double huber_delta);
/* nid_launch_chain on every shard: the first n_jac poses with the Jacobian phase, the rest cost only; one exchange */
int nid_multi_launch_chain(nid_multi *m, int first_slot, int n, const double *poses7, int n_jac, double huber_delta);
int nid_multi_wait(nid_multi *m, int slot, double *H36, double *b6, double *chi2, int32_t *n_active);
/* pipelined throughput loop (the bench's timed region): `batch` poses per launch, launches alternating between
 * each shard's two streams, one exchange per `group` launches, two groups in flight; reduced_out (n x
 * NID_REDUCED_LEN, may be NULL) receives every pose's summed block.  Blocking. */
int nid_multi_run_sequence(nid_multi *m, const double *poses7, int n, int batch, int group, int want_jac,
                           double huber_delta, double *reduced_out);
/* algorithmic (contract) bytes of one evaluation over the cells of THIS process */
int64_t nid_multi_contract_bytes(const nid_multi *m);

#ifdef __cplusplus
}
#endif
#endif
