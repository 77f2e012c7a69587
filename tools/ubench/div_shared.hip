// tools/ubench/div_shared.hip: the shared-reciprocal quotient of csrc/nid_kernels.hip.h (div_shared: rcp + two Newton steps
// + q = a r + ONE residual step, no v_div_scale / v_div_fixup) against the compiler's IEEE a / z on the device, bit for bit,
// over 2^28 operand pairs of the ranges the kernels see (z = depth-like 0.01 .. 100 of both signs, a = f * x up to 1e5, also
// tiny and huge magnitudes inside the guard's [2^-100, 2^100]); and x / 255 by the constant's reciprocal (div_255) against
// x / 255.0.  Also on the HOST's division (the oracle divides there).  Prints the mismatch counts: all must be 0.
//   hipcc --offload-arch=gfx950 -O2 -ffp-contract=off -o div_shared div_shared.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
__device__ __forceinline__ double rcp_fast(double z) {
  double r = __builtin_amdgcn_rcp(z);
  double e = fma(-z, r, 1.0);
  r = fma(r, e, r);
  e = fma(-z, r, 1.0);
  return fma(r, e, r);
}
__device__ __forceinline__ bool exp_mid(double x) { return (((unsigned)__double2hiint(x) >> 20) & 0x7FFu) - 923u <= 200u; }
__device__ __forceinline__ double div_shared(double a, double z, double r, bool z_mid) {
  const double q = a * r;
  double res = fma(fma(-z, q, a), r, q);
  if (!(z_mid && exp_mid(a))) res = a / z;
  return res;
}
__device__ __forceinline__ double div_255(double x) {
  const double r = 1.0 / 255.0, q0 = x * r;
  return fma(fma(-255.0, q0, x), r, q0);
}
__device__ __forceinline__ unsigned long long rng(unsigned long long &s) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
__device__ __forceinline__ double unit(unsigned long long &s) { return (double)(rng(s) >> 11) * (1.0 / 9007199254740992.0); }
__global__ void k(unsigned long long seed, int per_thread, unsigned long long *bad, double *sample) {
  unsigned long long s = seed + 0x9E3779B97F4A7C15ull * (blockIdx.x * (unsigned long long)blockDim.x + threadIdx.x + 1);
  unsigned long long bad_div = 0, bad_255 = 0, shorts = 0;
  for (int i = 0; i < per_thread; i++) {
    const int kind = (int)(rng(s) & 7);
    double z = 0.01 + 99.99 * unit(s);
    double a = 2e5 * (unit(s) - 0.5);
    if (kind == 1) { z = 0.3 + 6.0 * unit(s); a = 481.2 * (8.0 * (unit(s) - 0.5)); }           // the pair's own ranges
    if (kind == 2) z = -z;
    if (kind == 3) { z = ldexp(1.0 + unit(s), (int)(rng(s) % 190) - 95); a = ldexp(1.0 + unit(s), (int)(rng(s) % 190) - 95); }
    if (kind == 4) a = (double)(long long)(a);                                                   // integers: exact quotients happen
    if (kind == 5) { z = (double)(1 + (rng(s) % 4096)); a = z * (double)(rng(s) % 4096); }        // exact quotients
    if (kind == 6) a = ldexp(a, -110);                                                           // outside the guard: the fallback
    const double r = rcp_fast(z);
    const bool zm = exp_mid(z);
    const double want = a / z, got = div_shared(a, z, r, zm);
    if (zm && exp_mid(a)) shorts++;
    if (__double_as_longlong(want) != __double_as_longlong(got)) { if (!bad_div) { sample[0] = a; sample[1] = z; } bad_div++; }
    const double x = 255.0 * 16.0 * unit(s) * (kind == 7 ? 1e-6 : 1.0);
    if (__double_as_longlong(x / 255.0) != __double_as_longlong(div_255(x))) { if (!bad_255) sample[2] = x; bad_255++; }
  }
  atomicAdd(bad, bad_div); atomicAdd(bad + 1, bad_255); atomicAdd(bad + 2, shorts);
}
int main() {
  unsigned long long *bad; double *sample;
  hipMalloc(&bad, 24); hipMalloc(&sample, 24); hipMemset(bad, 0, 24); hipMemset(sample, 0, 24);
  const int blocks = 4096, threads = 256, per = 256;  // 2^28 pairs
  hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, 20261005ull, per, bad, sample);
  unsigned long long h[3]; double hs[3];
  hipMemcpy(h, bad, 24, hipMemcpyDeviceToHost); hipMemcpy(hs, sample, 24, hipMemcpyDeviceToHost);
  printf("%llu quotient pairs (%llu through the short form): %llu differ from the compiler's a / z%s; x / 255: %llu differ\n",
         (unsigned long long)blocks * threads * per, h[2], h[0], h[0] ? " (first: see below)" : "", h[1]);
  if (h[0]) printf("  first mismatch: a = %a, z = %a\n", hs[0], hs[1]);
  if (h[1]) printf("  first x / 255 mismatch: x = %a\n", hs[2]);
  return (h[0] || h[1]) ? 1 : 0;
}
