// legacy_lm_caller.cpp -- the reference's WHOLE call pattern on the three operators, with the reference's buffer
// ownership: the buffers of NID_pose_estimation.cpp:229-242, Calculate3Dpoint (:253),
// CudaComputeHref (:257), then the LM's calls -- per outer iteration one CudaComputeH with calculate_der == true
// (optimization_algorithm_levenberg.cpp:98), cost-only trials (:173) and the verbose re-evaluation
// (sparse_optimizer.cpp:423), Htarget / Hjoint zeroed by the caller before every call (levenberg.cpp:87-88), a few
// milliseconds of host work between some of them -- ONE undeclared in-place change of an im1 pixel in the middle,
// and then the frees of :388-395 AT ONCE and return.  It calls nothing but the three operators: no nid_legacy_*.
// Compiled by the tests (a) against libnid_host.so / libnid_hip.so and run on the GPU, 50 times, with MALLOC_PERTURB_
// set; (b) together with host/legacy_ops.cpp and tests/cpp/nid_hip_stub.cpp under -fsanitize=address on the CPU: a
// read of a caller buffer after its call has returned is a heap-use-after-free there.
//   legacy_lm_caller <dir>: reads <dir>/in.bin, writes <dir>/out.bin (layouts below).
#include <cuda_runtime.h>  // include/nid/compat/cuda_runtime.h

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "nid/legacy_ops.h"

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  char path[1024];
  std::snprintf(path, sizeof(path), "%s/in.bin", argv[1]);
  FILE *f = std::fopen(path, "rb");
  if (!f) return 3;
  int hdr[8];  // rows, cols, cell, bin_num, ncalls, change_at (call index before which ONE im1 pixel changes), change_px, pause_every
  if (std::fread(hdr, sizeof(int), 8, f) != 8) return 4;
  const int rows = hdr[0], cols = hdr[1], cell = hdr[2], bin_num = hdr[3], ncalls = hdr[4], change_at = hdr[5], change_px = hdr[6],
            pause_every = hdr[7], bs_degree = 3;
  const size_t N = (size_t)rows * cols;
  const int ncell = cell * cell;
  double intrinscis[5], T_wc0[16], pose0[16];
  std::vector<double> poses((size_t)16 * ncalls), depth(N);
  cudaFree(0);  // :57
  double *im0_data, *im1_data, *points_3d_all;  // :240-242
  cudaMallocManaged(&im0_data, N * sizeof(double));
  cudaMallocManaged(&im1_data, N * sizeof(double));
  cudaMallocManaged(&points_3d_all, 3 * N * sizeof(double));
  bool ok = std::fread(intrinscis, 8, 5, f) == 5 && std::fread(T_wc0, 8, 16, f) == 16 && std::fread(pose0, 8, 16, f) == 16 &&
            std::fread(poses.data(), 8, poses.size(), f) == poses.size() && std::fread(depth.data(), 8, N, f) == N &&
            std::fread(im0_data, 8, N, f) == N && std::fread(im1_data, 8, N, f) == N;
  std::fclose(f);
  if (!ok) return 5;
  // :229-238 -- host buffers of the operators' outputs
  double *bs_value = (double *)malloc(4 * N * sizeof(double));
  int *bin_index = (int *)malloc(N * sizeof(int));
  int *bs_counter = (int *)malloc(ncell * sizeof(int));
  double *Href = (double *)calloc(ncell, sizeof(double));
  double *Htarget = (double *)malloc(ncell * sizeof(double));
  double *Hjoint = (double *)malloc(ncell * sizeof(double));
  double *der = (double *)calloc(6 * ncell, sizeof(double));
  std::vector<double> record((size_t)ncalls * 8 * ncell);  // per call: Htarget, Hjoint, der[6]

  Calculate3Dpoint(depth.data(), T_wc0, points_3d_all, intrinscis, rows, cols);                       // :253
  CudaComputeHref(im0_data, points_3d_all, pose0, intrinscis, bin_num, bs_degree, cell, rows, cols,  // :257
                  bs_value, bin_index, bs_counter, Href);
  for (int k = 0; k < ncalls; k++) {
    // one outer iteration = [der, cost, cost, verbose cost]: der on every fourth call
    const bool calculate_der = k % 4 == 0;
    if (k == change_at) im1_data[change_px] = 255.0 - im1_data[change_px];  // in place, undeclared, off the sampled indices
    if (pause_every > 0 && k % pause_every == pause_every - 1)
      std::this_thread::sleep_for(std::chrono::milliseconds(3));  // (the solver's host work: a background verification WOULD start)
    std::memset(Htarget, 0, ncell * sizeof(double));  // levenberg.cpp:87-88, 171-172
    std::memset(Hjoint, 0, ncell * sizeof(double));
    g2o::CudaComputeH(calculate_der, im0_data, im1_data, points_3d_all, bs_counter, bs_value, bin_index, &poses[(size_t)16 * k],
                      intrinscis, bin_num, bs_degree, cell, rows, cols, Href, nullptr, nullptr, Htarget, Hjoint, der);
    double *rec = &record[(size_t)k * 8 * ncell];
    std::memcpy(rec, Htarget, ncell * sizeof(double));
    std::memcpy(rec + ncell, Hjoint, ncell * sizeof(double));
    std::memcpy(rec + 2 * ncell, der, 6 * ncell * sizeof(double));
  }
  // :388-395 -- everything is freed right behind the last call
  cudaFree(im0_data); cudaFree(im1_data); cudaFree(points_3d_all);
  free(bs_value); free(bin_index); free(Href); free(Htarget); free(Hjoint); free(der);
  std::vector<int> counters(bs_counter, bs_counter + ncell);
  free(bs_counter);
  // (what a late reader of the freed buffers would run into: the allocator hands the memory out again, other content)
  for (int rep = 0; rep < 4; rep++) {
    void *again = malloc(4 * N * sizeof(double));
    std::memset(again, 0x5A, 4 * N * sizeof(double));
    std::this_thread::sleep_for(std::chrono::milliseconds(2));
    free(again);
  }
  std::snprintf(path, sizeof(path), "%s/out.bin", argv[1]);
  f = std::fopen(path, "wb");
  if (!f) return 6;
  std::fwrite(counters.data(), sizeof(int), ncell, f);
  std::fwrite(record.data(), 8, record.size(), f);
  std::fclose(f);
  return 0;
}
