// the resident evaluator's kernels (k_resident<512, NB, 3>): see nid_eval_launch.h
#include "nid_eval_launch.h"

namespace nid {

namespace {
template <int NB>
void launch_res(const EvalParams &P, size_t lds, unsigned grid, hipStream_t s, const ResidentCtl *ctl, unsigned long long word0,
                long long idle_ticks, int xform_mode) {
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_resident<512, NB, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  (void)hipGetLastError();
  hipLaunchKernelGGL((k_resident<512, NB, 3>), dim3(grid), dim3(512), lds, s, P, ctl, word0, idle_ticks, xform_mode);
}
}  // namespace

void launch_resident_512(const EvalParams &P, size_t lds, unsigned grid, hipStream_t s, const ResidentCtl *ctl,
                         unsigned long long word0, long long idle_ticks, int xform_mode) {
  if (P.g.nb == 8) launch_res<8>(P, lds, grid, s, ctl, word0, idle_ticks, xform_mode);
  else if (P.g.nb == 10) launch_res<10>(P, lds, grid, s, ctl, word0, idle_ticks, xform_mode);
  else launch_res<0>(P, lds, grid, s, ctl, word0, idle_ticks, xform_mode);
}

}  // namespace nid
