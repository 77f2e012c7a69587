// nid_resident_kernels.hip.h -- the kernel that stays on the device between requests: k_resident (single poses through a
// mailbox), its control words and record layouts.  It runs eval_cell, the body of the launched kernels (nid_kernels.hip.h).  Included by nid_resident_tu.hip (device code) and nid_capi.hip (the host
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

}  // namespace nid
