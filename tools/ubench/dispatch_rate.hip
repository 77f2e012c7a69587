// dispatch_rate.hip -- how fast does one kernel launch get its workgroups running on gfx950?
// Every workgroup records the chip-wide 100 MHz clock (s_memrealtime) when its first wave starts, then spins for
// `hold_us` (so that no workgroup retires while others are still being dispatched), and records the clock again.
// Printed per (grid, block, dynamic LDS, kernel-argument bytes): start spread = last start - first start.
// Build: hipcc --offload-arch=gfx950 -O2 -o dispatch_rate dispatch_rate.hip ; run: ./dispatch_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

struct BigArgs { long long pad[480]; };  // 3840 B, the size of the evaluation kernel's argument block

template <typename A>
__global__ void k(long long *out, int hold_ticks, A args) {
  extern __shared__ unsigned char smem[];
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    smem[0] = (unsigned char)args.pad[0];
    out[2 * blockIdx.x] = t0;
    while (__builtin_amdgcn_s_memrealtime() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(1);
    out[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
}
struct NoArgs { long long pad[1]; };

template <typename A>
void run(const char *name, int grid, int block, size_t lds, long long *dev, std::vector<long long> &host) {
  A a{};
  double spread[7];
  for (int rep = 0; rep < 7; rep++) {
    hipLaunchKernelGGL((k<A>), dim3(grid), dim3(block), lds, 0, dev, 1500, a);
    hipDeviceSynchronize();
    hipMemcpy(host.data(), dev, sizeof(long long) * 2 * grid, hipMemcpyDeviceToHost);
    long long lo = host[0], hi = host[0];
    for (int i = 0; i < grid; i++) { lo = std::min(lo, host[2 * i]); hi = std::max(hi, host[2 * i]); }
    spread[rep] = (hi - lo) / 100.0;
  }
  std::sort(spread, spread + 7);
  printf("%-8s grid %5d block %5d lds %6zu B : start spread median %6.2f us (min %6.2f) = %6.1f ns per workgroup\n", name, grid,
         block, lds, spread[3], spread[0], spread[3] * 1e3 / grid);
}

int main() {
  long long *dev;
  hipMalloc(&dev, sizeof(long long) * 2 * 8192);
  std::vector<long long> host(2 * 8192);
  for (int grid : {64, 128, 256, 512, 1024})
    for (int block : {64, 128, 512, 1024}) {
      if ((long)grid * block > 256L * 2048) continue;  // everything must be resident at once (the workgroups spin)
      run<NoArgs>("small", grid, block, 0, dev, host);
    }
  for (int grid : {128, 256})
    for (int block : {128, 512}) {
      run<NoArgs>("lds16k", grid, block, 16384, dev, host);
      run<BigArgs>("bigargs", grid, block, 16384, dev, host);
    }
  hipFree(dev);
  return 0;
}
