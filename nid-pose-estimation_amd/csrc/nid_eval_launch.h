// nid_eval_launch.h -- the evaluation kernels' launch entry points, one translation unit per workgroup shape and
// kernel kind (nid_eval_tu.inc, built in parallel by __graft_entry__.build(): the ~120 instantiations of k_eval2 /
// k_resident in one translation unit took five minutes to compile, split over eight they take one).
// Plain host functions: each translation unit owns the device code of the kernels it launches.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>

#include "nid_kernels.hip.h"

namespace nid {

struct ResidentCtl;  // nid_resident_kernels.hip.h

// which family of k_eval2 instantiations a launch takes (launch_eval2 in nid_capi.hip decides)
enum EvalFamily {
  kFamLoop = 0,       // bin-specialised (8 / 10) or generic; EXT when P.slots_ext is set (128 / 256 threads)
  kFamDbg = 1,        // per-pixel dump / phase stamps compiled in (128 / 256 threads, generic bin count)
  kFamBig = 2,        // cells of more than 32 * NT slots (FAST, generic bin count)
  kFamLat = 3,        // the latency form (512 / 1024 threads, FAST, <= kMaxBatch poses)
  kFamStampsLat = 4,  // ... with phase stamps (generic bin count)
};

// lds_repair: the dynamic LDS of k_repair's workgroup shape (repair_threads(NT), nid_eval_tu.inc)
// eval_end: null, or an event recorded right behind k_eval2 -- in front of k_repair -- (timed launches: nid_time_kernel)
#define NID_DECLARE_EVAL_TU(NT, KIND) \
  void launch_eval_##NT##_##KIND(const EvalParams &P, int family, bool strict, size_t lds, size_t lds_repair, hipStream_t s, int batch, hipEvent_t eval_end);
NID_DECLARE_EVAL_TU(128, jac) NID_DECLARE_EVAL_TU(128, cost)
NID_DECLARE_EVAL_TU(256, jac) NID_DECLARE_EVAL_TU(256, cost)
NID_DECLARE_EVAL_TU(512, jac) NID_DECLARE_EVAL_TU(512, cost)
NID_DECLARE_EVAL_TU(1024, jac) NID_DECLARE_EVAL_TU(1024, cost)
#undef NID_DECLARE_EVAL_TU

// k_resident<512, NB, 3> (nid_resident_tu.hip): sets the kernel's dynamic LDS limit and launches it
void launch_resident(const EvalParams &P, int nt, size_t lds, unsigned grid, hipStream_t s, const ResidentCtl *ctl,
                         unsigned long long word0, long long idle_ticks, int xform_mode);

}  // namespace nid
