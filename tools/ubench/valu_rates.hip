// valu_rates.hip -- issue cost of the VALU / LDS instructions the NID kernels are made of, on gfx950.
// For each instruction: a loop of 8 independent copies x 256 iterations, timed with s_memtime, run
// with 1 and with 4 waves per SIMD (one workgroup of 256 / 1024 threads on one CU).
// Prints cycles per wave-instruction per SIMD (= elapsed cycles * waves_per_simd^-1 ... see below).
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_rates valu_rates.hip ; run: ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <string>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ void k(long long *out, int iters, double seed) {
  __shared__ unsigned long long lds[4096];
  double a[8], b = seed + threadIdx.x * 1e-3, c = 1.0000001;
  int ia[8]; unsigned long long la[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; ia[i] = threadIdx.x + i; la[i] = threadIdx.x * 8 + i; }
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  unsigned ldsaddr = (threadIdx.x & 1023) * 8;
  long long r0 = __builtin_amdgcn_s_memrealtime();
  long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int it = 0; it < iters; it++) {
#define X(i) \
    if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b)); \
    else if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(b)); \
    else if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c)); \
    else if (OP == 3) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(ia[i])); \
    else if (OP == 4) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(ia[i]) : "v"(a[i])); \
    else if (OP == 5) asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(a[i]) : "v"(ia[i])); \
    else if (OP == 6) asm volatile("v_cmp_le_f64 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc"); \
    else if (OP == 7) asm volatile("v_rcp_f64 %0, %1" : "=v"(a[i]) : "v"(b)); \
    else if (OP == 8) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7])); \
    else if (OP == 9) asm volatile("v_bfe_u32 %0, %1, 8, 8" : "=v"(ia[i]) : "v"(ia[(i + 1) & 7])); \
    else if (OP == 10) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]) : "vcc"); \
    else if (OP == 11) asm volatile("v_lshl_add_u64 %0, %0, 3, %1" : "+v"(la[i]) : "v"(la[(i + 1) & 7])); \
    else if (OP == 12) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7])); \
    else if (OP == 13) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(la[i]) : "v"(ia[i]), "v"(ia[(i + 1) & 7]) : "vcc"); \
    else if (OP == 14) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]), "v"(ia[(i + 2) & 7])); \
    else if (OP == 15) asm volatile("ds_add_u64 %0, %1 offset:0" :: "v"(ldsaddr), "v"(la[i]) : "memory"); \
    else if (OP == 16) asm volatile("ds_write_b64 %0, %1 offset:0" :: "v"(ldsaddr), "v"(la[i]) : "memory"); \
    else if (OP == 17) asm volatile("ds_read_b64 %0, %1 offset:0" : "=v"(la[i]) : "v"(ldsaddr) : "memory"); \
    else if (OP == 18) asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(*(__attribute__((ext_vector_type(4))) unsigned *)&la[i & 6]) : "v"(ldsaddr) : "memory"); \
    else if (OP == 19) asm volatile("ds_read_b128 %0, %1" : "=v"(*(__attribute__((ext_vector_type(4))) unsigned *)&la[i & 6]) : "v"(ldsaddr * 2) : "memory"); \
    else if (OP == 20) asm volatile("v_max_i32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7])); \
    else if (OP == 21) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]), "v"(ia[(i + 2) & 7])); \
    else if (OP == 22) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b)); \
    else if (OP == 23) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]) : "s20", "s21"); \
    else if (OP == 24) asm volatile("v_cmp_le_f64_e64 s[20:21], %0, %1" :: "v"(a[i]), "v"(b) : "s20", "s21"); \
    else if (OP == 25) asm volatile("v_cmp_le_f64_e64 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %3, s[20:21]" : "+v"(ia[i]) : "v"(a[i]), "v"(b), "v"(ia[(i + 1) & 7]) : "s20", "s21"); \
    else if (OP == 26) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(ia[i]) : "v"(ia[(i + 1) & 7])); \
    else if (OP == 27) asm volatile("v_readfirstlane_b32 s20, %0" :: "v"(ia[i]) : "s20"); \
    else if (OP == 28) asm volatile("v_and_b32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7])); \
    else if (OP == 29) asm volatile("s_and_saveexec_b64 s[20:21], vcc\n\ts_or_b64 exec, exec, s[20:21]" ::: "s20", "s21"); \
    else if (OP == 30) asm volatile("v_cmp_lt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]) : "vcc"); \
    else if (OP == 31) asm volatile("s_nop 0"); \
    else if (OP == 32) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(ia[i]) : "v"(ia[(i + 1) & 7]), "v"(ia[(i + 2) & 7]) : "vcc");
    REP8(X)
#undef X
    if (OP >= 15 && OP <= 19 && (it & 1)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long t1 = __builtin_amdgcn_s_memtime();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0; long long li = 0;
  for (int i = 0; i < 8; i++) { s += a[i]; li += ia[i] + (long long)la[i]; }
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[2] = r1 - r0; }
  if (s == 12345.678 && li == 77) out[1] = 1;
}

static const char *names[] = {"v_fma_f64", "v_add_f64", "v_mul_f64", "v_cvt_f64_i32", "v_cvt_i32_f64", "v_cvt_f64_u32",
                              "v_cmp_le_f64", "v_rcp_f64", "v_add_u32", "v_bfe_u32", "v_cndmask_b32", "v_lshl_add_u64",
                              "v_mul_lo_u32", "v_mad_u64_u32", "v_fma_f32", "ds_add_u64", "ds_write_b64", "ds_read_b64",
                              "ds_read2_b64", "ds_read_b128", "v_max_i32", "v_mad_i32_i24", "v_fmac_f64", "cndmask_e64_sgpr", "cmp_f64_e64_sgpr", "cmp_f64+cndmask", "v_mov_dpp", "readfirstlane", "v_and_b32", "saveexec+or", "cmp_i32+cndmask", "s_nop", "cndmask_3op"};

static int g_grid = 1;
template <int OP>
void run(long long *d) {
  const int iters = 4096;
  for (int threads : {256, 512, 1024}) {
    long long h[3];
    for (int rep = 0; rep < 3; rep++) {
      hipLaunchKernelGGL(k<OP>, dim3(g_grid), dim3(threads), 0, 0, d, iters, 1.5);
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    }
    const int wps = threads / 256;  // waves per SIMD
    // wave 0's elapsed cycles; its SIMD issued wps * iters * 8 of these instructions meanwhile
    printf("%-16s waves/SIMD %d: %7.2f ticks (%6.2f ns) per wave-instruction per SIMD (elapsed %lld ticks, %.2f us, %.3f GHz)\n",
           names[OP], wps, (double)h[0] / (iters * 8.0 * wps), (double)h[2] * 10.0 / (iters * 8.0 * wps), h[0], h[2] / 100.0,
           (double)h[0] / (h[2] * 10.0));
  }
}

template <int OP> struct Runner { static void go(long long *d) { run<OP>(d); Runner<OP + 1>::go(d); } };
template <> struct Runner<33> { static void go(long long *) {} };

int main(int argc, char **argv) {
  if (argc > 1) g_grid = atoi(argv[1]);  // 256*k blocks: k blocks per CU (block 0 is the one timed)
  long long *d;
  hipMalloc(&d, 64); hipMemset(d, 0, 64);
  Runner<0>::go(d);
  return 0;
}
