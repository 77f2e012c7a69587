// the resident evaluator's kernels (k_resident<512, NB, 3>: one workgroup per CU, the latency form; k_resident<256, NB, 0>:
// four workgroups per CU, the loop form -- contexts of more cells than the device has CUs): see nid_eval_launch.h
#include "nid_eval_launch.h"
#include "nid_resident_kernels.hip.h"

namespace nid {

namespace {
template <int NT, int NB, int LAT>
void launch_res(const EvalParams &P, size_t lds, unsigned grid, hipStream_t s, const ResidentCtl *ctl, unsigned long long word0,
                long long idle_ticks, int xform_mode) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_resident<NT, NB, LAT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipGetLastError();
  hipLaunchKernelGGL((k_resident<NT, NB, LAT>), dim3(grid), dim3(NT), lds, s, P, ctl, word0, idle_ticks, xform_mode);
}
template <int NT, int LAT>
void launch_res_nb(const EvalParams &P, size_t lds, unsigned grid, hipStream_t s, const ResidentCtl *ctl, unsigned long long word0,
                   long long idle_ticks, int xform_mode) {
  if (P.g.nb == 8) launch_res<NT, 8, LAT>(P, lds, grid, s, ctl, word0, idle_ticks, xform_mode);
  else if (P.g.nb == 10) launch_res<NT, 10, LAT>(P, lds, grid, s, ctl, word0, idle_ticks, xform_mode);
  else launch_res<NT, 0, LAT>(P, lds, grid, s, ctl, word0, idle_ticks, xform_mode);
}
}  // namespace

void launch_resident(const EvalParams &P, int nt, size_t lds, unsigned grid, hipStream_t s, const ResidentCtl *ctl,
                     unsigned long long word0, long long idle_ticks, int xform_mode) {
  if (nt == 512) launch_res_nb<512, 3>(P, lds, grid, s, ctl, word0, idle_ticks, xform_mode);
  else launch_res_nb<256, 0>(P, lds, grid, s, ctl, word0, idle_ticks, xform_mode);
}

}  // namespace nid
