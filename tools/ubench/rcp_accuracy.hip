// tools/ubench/rcp_accuracy.hip: relative error of v_rcp_f64 and of one / two Newton steps on it (FAST math's rcp_fast
// uses two).  hipcc --offload-arch=gfx950 -O2 -o rcp_accuracy rcp_accuracy.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const double *x, double *r0, double *r1, double *r2, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double z = x[i];
  double r = __builtin_amdgcn_rcp(z);
  r0[i] = r;
  double e = fma(-z, r, 1.0);
  r = fma(r, e, r);
  r1[i] = r;
  e = fma(-z, r, 1.0);
  r2[i] = fma(r, e, r);
}
int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), a(n), b(n), c(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x[i] = 0.25 + 7.75 * (double)(s >> 11) / 9007199254740992.0; }
  double *dx, *d0, *d1, *d2;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8); hipMalloc(&d2, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, d2, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost); hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; i++) {
    const long double t = 1.0L / (long double)x[i];
    m0 = fmax(m0, (double)fabsl(((long double)a[i] - t) / t));
    m1 = fmax(m1, (double)fabsl(((long double)b[i] - t) / t));
    m2 = fmax(m2, (double)fabsl(((long double)c[i] - t) / t));
  }
  printf("max relative error over %d values in [0.25, 8): v_rcp_f64 %.3e, + 1 Newton step %.3e, + 2 steps %.3e (2^-53 = 1.11e-16)\n", n, m0, m1, m2);
  return 0;
}
