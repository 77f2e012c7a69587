// nid_resident_kernels.hip.h -- the kernels that stay on the device between requests: k_resident (single poses through a
// mailbox) and k_resident_batch (short sequences), their control words and record layouts.  Both run eval_cell, the body
// of the launched kernels (nid_kernels.hip.h).  Included by nid_resident_tu.hip (device code) and nid_capi.hip (the host
// side of the protocols).
#pragma once

#include "nid_kernels.hip.h"

namespace nid {

// ---------------------------------------------------------------------------
// The RESIDENT evaluator: the latency form of the evaluation kernel, launched ONCE per frame pair, one workgroup per
// cell, all of them co-resident; each workgroup waits for a request -- a doorbell word in fine-grained device memory
// that the CPU writes through the PCIe BAR, behind the request's pose record --, evaluates its cell with eval_cell and
// writes the cell's record straight to pinned host memory (the DIRECT protocol: the host forms and sums the
// quadratic forms), then waits again.  A Gauss-Newton / LM loop launches one pose and waits for it: the kernel launch
// itself -- runtime call, packet, dispatch of 256 workgroups: ~6 us from the enqueue to a first result word on the
// host, tools/ubench/mailbox_latency.hip -- is then most of what is left of an evaluation; a request to a resident
// workgroup takes ~2 us there and back.
// EVERY wave leaves: on the exit word, or after `idle_ticks` of s_memrealtime (100 MHz) without a request (a dead
// host cannot keep the workgroups spinning); the host retires the kernel itself long before that (nid_capi.hip).
// The mailbox: ONE 64-byte line -- the request's pose7 (q x y z w, t) in seven words and, in the eighth, its number and
// flags.  The CPU writes the seven words, a store fence, the eighth word, a store fence (write-combining stores are not
// ordered among themselves; across the fence they are), so a poll that reads a new eighth word -- the sixteen dwords
// of the line in one request -- has the request's pose in the same registers: no second trip to memory.  The 3x4
// matrix is formed on the device with the host's operations (pose_from_pose7 in nid_capi.hip: IEEE, no contraction:
// the same bits).  Eighth word: (number << 8) | flags; kResExitWord: leave.
// A pose that exists only as a matrix (the legacy operators hand over a 4x4: nid_evaluate_matrix) takes a second line:
// M[0..6] | word, M[7..11], 0, 0 | the same word; such a request has arrived when both lines carry its word.
struct ResidentCtl {
  unsigned long long w[16];
};
constexpr unsigned long long kResExitWord = ~0ull;
// (measured on A, 256 workgroups: one load at a time + 8 x 64 clocks of sleep answers first; deeper pipelines of loads
// queue behind each other on the line's memory channel: profiles/r03_launch_cost_A.txt)
#ifndef NID_RES_POLL_DEPTH
#define NID_RES_POLL_DEPTH 1
#endif
#ifndef NID_RES_POLL_SLEEP
#define NID_RES_POLL_SLEEP 8
#endif
constexpr int kResPollDepth = NID_RES_POLL_DEPTH;
constexpr unsigned kResJac = 1u, kResRecord = 2u, kResCellout = 4u, kResMatrix = 8u, kResLeave = 0xFFFFFFFFu;

// pose_from_pose7 (nid_capi.hip) on the device: to_homogeneous_matrix (se3quat.h:270-278) = Eigen toRotationMatrix
__device__ __forceinline__ void pose_from_pose7_dev(const double *p, int mode, Pose &out) {
#pragma unroll
  for (int i = 0; i < 7; i++) out.q[i] = p[i];
  const double x = p[0], y = p[1], z = p[2], w = p[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  double *M = out.M;
  M[0] = 1 - (tyy + tzz); M[1] = txy - twz;       M[2] = txz + twy;        M[3] = p[4];
  M[4] = txy + twz;       M[5] = 1 - (txx + tzz); M[6] = tyz - twx;        M[7] = p[5];
  M[8] = txz - twy;       M[9] = tyz + twx;       M[10] = 1 - (txx + tyy); M[11] = p[6];
  out.mode = mode;
}

__device__ __forceinline__ unsigned long long ld_sys_u64(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned ld_sys_u32(const unsigned *p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// (One workgroup per CU: a 512-thread workgroup is two waves per SIMD, so its waves may use 256 registers -- the pose
// arrives at run time, i.e. it is pinned in scalar registers for the whole evaluation instead of being re-read from
// the kernel arguments where it is needed, and the scalar registers that no longer fit spill into vector lanes.)
template <int NT, int NB, int LAT>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NT == 512 ? 2 : 4))) void k_resident(EvalParams P, const ResidentCtl *ctl, unsigned long long word0, long long idle_ticks,
                                                                                                         int xform_mode) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ unsigned s_req[2 + 32];  // kResLeave or 0 | (unused) | the request as it sits in the mailbox
  const Geometry &g = P.g;
  const int tid = threadIdx.x;
  const int cl = (int)(blockIdx.x >> 3) * 8 + (int)(blockIdx.x & 7);  // k_eval2's map for one pose
  if (cl >= g.nloc) return;
  {  // the B-spline table, once (eval_cell's LDS layout)
    const int nb = NB > 0 ? NB : g.nb;
    const int nbins = nb * nb + nb;
    double *rtab = reinterpret_cast<double *>(smem) + 2 * ((nbins + 1) & ~1);
    for (int i = tid; i < (nb - 3) * kCoefRow; i += NT) rtab[i] = P.ctab[i];
  }
  ResCell rc{P.Nc[cl], P.Href[cl], true};
  unsigned long long last = word0;
  long long t_idle = __builtin_amdgcn_s_memrealtime();
  for (;;) {
    __syncthreads();  // every wave is back from the previous request: the request words may be rewritten
    if (tid < 64) {   // wave 0 waits for the request
      // One poll = the mailbox line's 16 dwords in one request (uncached device memory: ~1 us).  One load at a time
      // would notice a request between one and two load times after it was written; with kResPollDepth loads in flight
      // (they return in order: each turn waits for the oldest and issues a new one, then sleeps so that 256
      // workgroups do not saturate the line's memory channel) it is noticed within about one load time.
      const unsigned *mail = reinterpret_cast<const unsigned *>(ctl) + (tid & 31);
      unsigned w[kResPollDepth];
#pragma unroll
      for (int k = 0; k < kResPollDepth; k++) w[k] = ld_sys_u32(mail);
      unsigned got = 0u, leave = 0u;
      unsigned long long word = last;
      for (bool done = false; !done;) {
#pragma unroll
        for (int k = 0; k < kResPollDepth; k++) {
          const unsigned v = w[k];
          w[k] = ld_sys_u32(mail);
          if (!done) {
            const unsigned long long t = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)v, 15) << 32) |
                                         (unsigned)__builtin_amdgcn_readlane((int)v, 14);
            const unsigned long long t1 = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)v, 31) << 32) |
                                          (unsigned)__builtin_amdgcn_readlane((int)v, 30);
            if (t == kResExitWord) { leave = 1u; done = true; }
            else if (t != last && (!(t & kResMatrix) || t1 == t)) { got = v; word = t; done = true; }
            else __builtin_amdgcn_s_sleep(NID_RES_POLL_SLEEP);
          }
        }
        if (!done && __builtin_amdgcn_s_memrealtime() - t_idle > idle_ticks) { leave = 1u; done = true; }
      }
      if (leave) {
        if (tid == 0) s_req[0] = kResLeave;
      } else {
        last = word;
        if (tid == 0) s_req[0] = 0u;
        if (tid < 32) s_req[2 + tid] = got;  // the request, as it sits in the mailbox
      }
    }
    __syncthreads();
    if ((unsigned)__builtin_amdgcn_readfirstlane((int)s_req[0]) == kResLeave) return;  // every wave of the workgroup, here
    const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane((int)s_req[2 + 14]) & 0xFFu;
    SlotArgs SA = P.slot[0];
    {
      double p7[7], m5[5];
#pragma unroll
      for (int i = 0; i < 7; i++)
        p7[i] = __hiloint2double(__builtin_amdgcn_readfirstlane((int)s_req[2 + 2 * i + 1]), __builtin_amdgcn_readfirstlane((int)s_req[2 + 2 * i]));
#pragma unroll
      for (int i = 0; i < 5; i++)
        m5[i] = __hiloint2double(__builtin_amdgcn_readfirstlane((int)s_req[2 + 16 + 2 * i + 1]), __builtin_amdgcn_readfirstlane((int)s_req[2 + 16 + 2 * i]));
      pose_from_pose7_dev(p7, xform_mode, SA.pose);
      if (flags & kResMatrix) {  // pose_from_matrix16 (nid_capi.hip): the matrix as it is, computeH.cu:152-154 semantics
#pragma unroll
        for (int i = 0; i < 7; i++) { SA.pose.q[i] = i == 3 ? 1.0 : 0.0; SA.pose.M[i] = p7[i]; }
#pragma unroll
        for (int i = 0; i < 5; i++) SA.pose.M[7 + i] = m5[i];
        SA.pose.mode = 1;  // NID_XFORM_MATRIX
      }
      // (VALU results, uniform: back to scalar registers, where eval_cell expects a pose to live)
#pragma unroll
      for (int i = 0; i < 12; i++) SA.pose.M[i] = wave_uniform(SA.pose.M[i]);
    }
    SA.host_quad = (flags & kResRecord) ? 1 : 2;
    SA.cellout_host = (flags & kResCellout) ? 1 : 0;
    if (flags & kResJac) eval_cell<NT, true, false, NB, false, false, LAT, false, true>(P, SA, cl, 0, smem, rc);
    else eval_cell<NT, false, false, NB, false, false, LAT, false, true>(P, SA, cl, 0, smem, rc);
    rc.fresh = false;
    t_idle = __builtin_amdgcn_s_memrealtime();
  }
}

// ---------------------------------------------------------------------------
// The resident BATCH evaluator (round 5): the throughput form of the evaluation kernel -- 128-thread workgroups, the loop
// form, per-pose records in device memory (eval_cell's EXT form), the in-launch reduction -- launched once and kept on
// the device like k_resident, answering requests of K <= kResBatchMax poses: the candidates of a Gauss-Newton / LM step, a
// short sequence.  What such a request costs as launches is mostly not evaluation (profiles/r04_short_sequences.txt: 20
// poses = 92 us in a loop, 132-181 us as one shot behind a device-wide synchronisation -- the first launches after one
// pay 20-50 us): here it is ONE word written through the PCIe BAR.
// Grid: `replicas` workgroups per cell (k_eval2's block -> cell map with batch = replicas); the workgroups of a cell
// share the request's poses through a counter (whoever is free takes the next one).
// A request: the host fills the poses' SlotArgs records -- exactly as a launch of more than kMaxBatch poses carries them
// (nid_capi.hip fills both with the same function) -- into PINNED HOST memory, a store fence, and writes the mailbox's
// word = (request number << 16) | (K << 8) | flags through the BAR.  (A store through the BAR is a PCIe transaction of
// its own, ~0.2 us each: the records themselves written that way -- 580 words for 20 poses -- cost more than the
// evaluation; the device fetches them instead.)
// Uncached memory does not take thousands of readers either (2048 workgroups polling one line, or fetching their records
// from one place, queue up behind each other on its memory channel), so the request fans out through ordinary device
// memory, agent-scope stores and loads throughout (no placement assumption):
//   root   (cell 0, replica 0) polls the mailbox word, copies the K records host -> `grec` (all its threads, eight loads
//          in flight each), drains, and publishes the word in gword;
//   leader (replica 0 of every other cell) polls gword; root and leaders forward the word to their cell's record;
//   the other replicas poll that record.
// A workgroup copies the record of the pose it is about to evaluate from grec into its OWN entry of a staging array
// (P.slots_ext, one entry per workgroup), invalidates the scalar cache, and from there on the record is what it is for a
// launched k_eval2 of a large batch: scalar loads through the constant address space, the quaternion fetched where
// exact_decisions needs it.  Same eval_cell instantiation up to RES (a table that stays in LDS): the same bits as launches.
// Bounded like k_resident: exit word (forwarded down the same tree), idle limit by s_memrealtime in every polling loop.
constexpr int kResBatchMax = 64;
constexpr int kResBatchRecDwords = (int)(sizeof(SlotArgs) / 4);
struct ResBatchCtl {
  unsigned long long w[8];  // one line; the word is w[7]
};
__host__ __device__ constexpr unsigned long long resbatch_word(unsigned long long number, int k, unsigned flags) {
  return (number << 16) | ((unsigned long long)(unsigned)k << 8) | flags;
}
// device-side fan-out state, in u64 words: gword in its own 64 bytes | per cell {word, pose counter} | the request's records (grec)
__host__ __device__ constexpr size_t resbatch_cells_off() { return 8; }
__host__ __device__ inline size_t resbatch_grec_off(long nloc8) { return 8 + (size_t)nloc8 * 2; }
__host__ __device__ inline size_t resbatch_fan_words(long nloc8) { return resbatch_grec_off(nloc8) + ((size_t)kResBatchRecDwords * kResBatchMax + 1) / 2; }

template <int NB>
__global__ __launch_bounds__(128) __attribute__((amdgpu_waves_per_eu(4))) void k_resident_batch(EvalParams P, const ResBatchCtl *ctl, const unsigned *hrec /*pinned host: the records*/,
                                                                                                unsigned long long *fan, unsigned long long word0, long long idle_ticks, int replicas) {
  constexpr int NT = 128;
  static_assert(sizeof(SlotArgs) / 4 <= 64 && sizeof(SlotArgs) % 8 == 0, "a record is copied by one wave, a dword per lane");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  __shared__ unsigned s_req[4];  // kResLeave or 0 | K | flags | the next pose
  const Geometry &g = P.g;
  const int tid = threadIdx.x;
  const int q = (int)(blockIdx.x >> 3);
  const int rep = q % replicas;
  const int cl = (q / replicas) * 8 + (int)(blockIdx.x & 7);  // k_eval2's map with batch = replicas
  if (cl >= g.nloc) return;
  {  // the B-spline table, once (eval_cell's LDS layout)
    const int nb = NB > 0 ? NB : g.nb;
    const int nbins = nb * nb + nb;
    double *rtab = reinterpret_cast<double *>(smem) + 2 * ((nbins + 1) & ~1);
    for (int i = tid; i < (nb - 3) * kCoefRow; i += NT) rtab[i] = P.ctab[i];
  }
  ResCell rc{P.Nc[cl], P.Href[cl], true};
  unsigned long long last = word0;
  long long t_idle = __builtin_amdgcn_s_memrealtime();
  const long nloc8 = ((long)g.nloc + 7) / 8 * 8;
  unsigned long long *mycell = fan + resbatch_cells_off() + (size_t)cl * 2;
  unsigned *grec = reinterpret_cast<unsigned *>(fan + resbatch_grec_off(nloc8));
  const bool root = rep == 0 && cl == 0;
  // the word this workgroup polls, two dwords by two lanes: the mailbox's (root), gword (the other cells' replica 0), the
  // cell's record (everyone else)
  const unsigned *src = (root ? reinterpret_cast<const unsigned *>(ctl->w + 7) : reinterpret_cast<const unsigned *>(rep == 0 ? fan : mycell)) + (tid & 1);
  unsigned *mine = reinterpret_cast<unsigned *>(const_cast<SlotArgs *>(P.slots_ext) + blockIdx.x);  // this workgroup's staging entry
  for (;;) {
    __syncthreads();  // every wave is back from the previous request
    if (tid < 64) {
      unsigned leave = 0u;
      for (;;) {
        const unsigned v = root ? ld_sys_u32(src) : __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long t = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)v, 1) << 32) | (unsigned)__builtin_amdgcn_readlane((int)v, 0);
        if (t == kResExitWord) { leave = 1u; break; }
        if (t != last) { last = t; break; }
        if (__builtin_amdgcn_s_memrealtime() - t_idle > idle_ticks) { leave = 1u; break; }
        if (root) __builtin_amdgcn_s_sleep(NID_RES_POLL_SLEEP); else __builtin_amdgcn_s_sleep(2);
      }
      if (tid == 0) {
        s_req[0] = leave ? kResLeave : 0u;
        s_req[1] = (unsigned)(last >> 8) & 0xFFu;
        s_req[2] = (unsigned)last & 0xFFu;
      }
    }
    __syncthreads();
    const bool leaving = (unsigned)__builtin_amdgcn_readfirstlane((int)s_req[0]) == kResLeave;
    const int K = min((int)__builtin_amdgcn_readfirstlane((int)s_req[1]), kResBatchMax);
    const unsigned flags = (unsigned)__builtin_amdgcn_readfirstlane((int)s_req[2]);
    if (root && !leaving) {
      // the K records: pinned host memory -> grec, eight loads in flight per thread (a load over PCIe takes a microsecond or two)
      const int n = K * kResBatchRecDwords;
      for (int i0 = tid; i0 < n; i0 += 8 * NT) {
        unsigned v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (i0 + j * NT < n) ? ld_sys_u32(hrec + i0 + j * NT) : 0u;
#pragma unroll
        for (int j = 0; j < 8; j++)
          if (i0 + j * NT < n) __hip_atomic_store(grec + i0 + j * NT, v[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (rep == 0 && tid == 0) {
      // publish downwards: the word -- or the exit word -- in gword (root) and in the cell's record
      const unsigned long long w = leaving ? kResExitWord : last;  // (tid 0 is in wave 0: `last` is the new word)
      if (root) __hip_atomic_store(fan, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(mycell, w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (leaving) return;  // every wave of the workgroup, here
    for (;;) {
      // The cell's next pose: whoever is free takes it.  The counter word is (request number << 8) | next pose and is
      // never reset: the first workgroup of a cell to arrive at a request moves it on by compare-and-swap, and a workgroup
      // that is late (still on its way out of the previous request when the others have begun the next) finds a newer
      // number and takes nothing -- a counter that the cell's leader reset could hand such a straggler pose 0 again.
      if (tid == 0) {
        const unsigned long long myreq = last >> 16;
        unsigned long long c = __hip_atomic_load(mycell + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned take = (unsigned)K;
        for (;;) {
          unsigned long long want;
          unsigned cand;
          if ((c >> 8) == myreq) {
            cand = (unsigned)(c & 255ull);
            if (cand >= (unsigned)K) break;
            want = c + 1;
          } else if ((c >> 8) < myreq) {
            cand = 0u;
            want = (myreq << 8) | 1ull;
          } else {
            break;
          }
          if (__hip_atomic_compare_exchange_strong(mycell + 1, &c, want, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { take = cand; break; }
        }
        s_req[3] = take;
      }
      __syncthreads();
      const int p = (int)__builtin_amdgcn_readfirstlane((int)s_req[3]);
      if (p >= K) break;
      // pose p's record: grec -> this workgroup's staging entry
      if (tid < kResBatchRecDwords) {
        __hip_atomic_store(mine + tid, __hip_atomic_load(grec + p * kResBatchRecDwords + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT),
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      __builtin_amdgcn_s_dcache_inv();
      // (an opaque copy of the entry's index per pose: the record is read through the CONSTANT address space, whose
      // loads the optimiser may otherwise take for loop invariant)
      int pidx = (int)blockIdx.x;
      asm volatile("" : "+s"(pidx));
      SlotArgs sa_ext;
      {
        typedef const unsigned __attribute__((address_space(4))) *ConstDwords;
        ConstDwords csrc = (ConstDwords)(reinterpret_cast<uintptr_t>(P.slots_ext + pidx));
        unsigned *dst = reinterpret_cast<unsigned *>(&sa_ext);
#pragma unroll
        for (unsigned i = kPoseQuatDwords; i < sizeof(SlotArgs) / 4; i++) dst[i] = csrc[i];
      }
      // The loop-form body WITHOUT the inline repair (kLinFlagW: carrying that code costs the pixel loops their registers);
      // a cell that wants the repair pass publishes nothing and says so -- workgroup-uniform -- and is done again from
      // the start by the instantiation that repairs inline: what k_repair does behind a launch, here in place (cold).
      bool again;
      if (flags & kResJac) again = eval_cell<NT, true, false, NB, false, true, 0, false, true, false>(P, sa_ext, cl, pidx, smem, rc);
      else again = eval_cell<NT, false, false, NB, false, true, 0, false, true, false>(P, sa_ext, cl, pidx, smem, rc);
      if (__builtin_expect(again, 0)) {
        __syncthreads();
        if (flags & kResJac) (void)eval_cell<NT, true, false, NB, false, true, 0, false, true, true>(P, sa_ext, cl, pidx, smem, rc);
        else (void)eval_cell<NT, false, false, NB, false, true, 0, false, true, true>(P, sa_ext, cl, pidx, smem, rc);
      }
      rc.fresh = false;
      __syncthreads();  // (the staging entry and s_req[3] are rewritten for the next pose)
    }
    t_idle = __builtin_amdgcn_s_memrealtime();
  }
}

}  // namespace nid
