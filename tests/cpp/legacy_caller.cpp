// legacy_caller.cpp -- a C++ translation unit that uses the three operators exactly the way the reference's main()
// does (NID_pose_estimation.cpp:229-276: buffers owned by the caller, im0 / im1 / points3d from
// cudaMallocManaged, Calculate3Dpoint -> CudaComputeHref -> g2o::CudaComputeH with zeroed Htarget / Hjoint), compiled
// by tests/test_host_gpu.py against include/nid/legacy_ops.h and the cuda_runtime.h forwarding header: the
// "unchanged caller" claim checked by a compiler, not by ctypes trampolines.
//   legacy_caller <dir>: reads <dir>/in.bin, writes <dir>/out.bin (layout below).
#include <cuda_runtime.h>  // include/nid/compat/cuda_runtime.h

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "nid/legacy_ops.h"

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  char path[1024];
  std::snprintf(path, sizeof(path), "%s/in.bin", argv[1]);
  FILE *f = std::fopen(path, "rb");
  if (!f) return 3;
  int hdr[4];  // rows, cols, cell, bin_num
  if (std::fread(hdr, sizeof(int), 4, f) != 4) return 4;
  const int rows = hdr[0], cols = hdr[1], cell = hdr[2], bin_num = hdr[3], bs_degree = 3;
  const size_t N = (size_t)rows * cols;
  double intrinscis[5], T_wc0[16], pose0[16], pose1[16];
  std::vector<double> depth(N);
  cudaFree(0);  // :57
  double *im0_data, *im1_data, *points_3d_all;  // :240-242
  cudaMallocManaged(&im0_data, N * sizeof(double));
  cudaMallocManaged(&im1_data, N * sizeof(double));
  cudaMallocManaged(&points_3d_all, 3 * N * sizeof(double));
  bool ok = std::fread(intrinscis, 8, 5, f) == 5 && std::fread(T_wc0, 8, 16, f) == 16 && std::fread(pose0, 8, 16, f) == 16 &&
            std::fread(pose1, 8, 16, f) == 16 && std::fread(depth.data(), 8, N, f) == N &&
            std::fread(im0_data, 8, N, f) == N && std::fread(im1_data, 8, N, f) == N;
  std::fclose(f);
  if (!ok) return 5;
  // :229-238 -- host buffers of the operators' outputs
  double *bs_value = (double *)malloc(4 * N * sizeof(double));
  int *bin_index = (int *)malloc(N * sizeof(int));
  int *bs_counter = (int *)malloc(cell * cell * sizeof(int));
  double *Href = (double *)calloc(cell * cell, sizeof(double));
  double *Htarget = (double *)calloc(cell * cell, sizeof(double));
  double *Hjoint = (double *)calloc(cell * cell, sizeof(double));
  double *der = (double *)malloc(6 * cell * cell * sizeof(double));

  Calculate3Dpoint(depth.data(), T_wc0, points_3d_all, intrinscis, rows, cols);                       // :253
  CudaComputeHref(im0_data, points_3d_all, pose0, intrinscis, bin_num, bs_degree, cell, rows, cols,  // :257
                  bs_value, bin_index, bs_counter, Href);
  g2o::CudaComputeH(true, im0_data, im1_data, points_3d_all, bs_counter, bs_value, bin_index, pose1, intrinscis, bin_num,
                    bs_degree, cell, rows, cols, Href, nullptr, nullptr, Htarget, Hjoint, der);  // levenberg.cpp:98

  std::snprintf(path, sizeof(path), "%s/out.bin", argv[1]);
  f = std::fopen(path, "wb");
  if (!f) return 6;
  std::fwrite(bs_counter, sizeof(int), cell * cell, f);
  std::fwrite(Href, 8, cell * cell, f);
  std::fwrite(Htarget, 8, cell * cell, f);
  std::fwrite(Hjoint, 8, cell * cell, f);
  std::fwrite(der, 8, 6 * cell * cell, f);
  std::fwrite(points_3d_all, 8, 3 * N, f);
  std::fclose(f);
  cudaFree(im0_data); cudaFree(im1_data); cudaFree(points_3d_all);  // :388-390
  free(bs_value); free(bin_index); free(bs_counter); free(Href); free(Htarget); free(Hjoint); free(der);
  return 0;
}
