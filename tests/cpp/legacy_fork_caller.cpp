// legacy_fork_caller.cpp -- fork() between two operator calls (round 5's hash pool held a lock across
// calls, so os.fork() / multiprocessing after a CudaComputeH hung for ever, and a child inherited a half-consumed job).
// A few calls, fork(), then BOTH processes keep calling: the child has no worker threads (they do not survive fork)
// and must hash alone; nobody may hang (alarm) or crash.  Built by tests/test_host_cpu.py with host/legacy_ops.cpp and
// tests/cpp/nid_hip_stub.cpp; no GPU.
#include <cstdio>
#include <cstdlib>
#include <sys/wait.h>
#include <unistd.h>
#include <vector>

#include "nid/legacy_ops.h"

int main() {
  alarm(30);  // a deadlock ends the test with SIGALRM instead of hanging it
  const int rows = 240, cols = 320, cell = 8, nb = 8, ncell = cell * cell;
  const size_t N = (size_t)rows * cols;
  std::vector<double> depth(N, 2.0), im0(N), im1(N), pts(3 * N), bsv(4 * N), href(ncell, 0.0), ht(ncell), hj(ncell), der(6 * ncell);
  std::vector<int> bsi(N), cnt(ncell);
  for (size_t i = 0; i < N; i++) { im0[i] = (double)(i * 7 % 256); im1[i] = (double)(i * 13 % 256); }
  double intr[5] = {240.6, -240.0, 159.5, 119.5, 1.0 / 5000}, T[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  Calculate3Dpoint(depth.data(), T, pts.data(), intr, rows, cols);
  CudaComputeHref(im0.data(), pts.data(), T, intr, nb, 3, cell, rows, cols, bsv.data(), bsi.data(), cnt.data(), href.data());
  auto call = [&](int k) {
    std::fill(ht.begin(), ht.end(), 0.0); std::fill(hj.begin(), hj.end(), 0.0);
    g2o::CudaComputeH(k % 2 == 0, im0.data(), im1.data(), pts.data(), cnt.data(), bsv.data(), bsi.data(), T, intr, nb, 3, cell, rows, cols,
                      href.data(), nullptr, nullptr, ht.data(), hj.data(), der.data());
    double s = 0;
    for (double v : ht) s += v;   // (the stub's per-cell checksums of the resident target)
    return s;
  };
  const double want = call(0);
  for (int k = 1; k < 5; k++) if (call(k) != want) return 10;
  const pid_t pid = fork();
  if (pid < 0) return 11;
  if (pid == 0) {
    for (int k = 0; k < 20; k++) if (call(k) != want) _exit(12);
    im1[N / 2 + 17] = 255.0 - im1[N / 2 + 17];   // an in-place change is still followed in the child (hashing alone)
    bool seen = false;
    for (int k = 0; k < NID_LEGACY_SLICES / NID_LEGACY_SLICES_PER_CALL + 1; k++) seen = seen || call(k) != want;
    _exit(seen ? 0 : 13);
  }
  for (int k = 0; k < 20; k++) if (call(k) != want) return 14;
  int status = 0;
  if (waitpid(pid, &status, 0) != pid) return 15;
  if (!WIFEXITED(status) || WEXITSTATUS(status) != 0) { std::fprintf(stderr, "child status %d\n", status); return 16; }
  std::puts("fork ok");
  return 0;
}
