// valu_wallclock.hip -- how long a wave64 VALU instruction occupies a SIMD on gfx950, measured THREE ways at once so
// that profiles/r01_valu_rates.txt (2.0 s_memtime ticks per v_fma_f64 at 4 waves / SIMD) and the SQ counters
// (SQ_ACTIVE_INST_VALU x 4 / SQ_INSTS_VALU = 4.1) can be reconciled:
//   (a) s_memtime ticks of wave 0 of block 0 (what valu_rates.hip printed),
//   (b) s_memrealtime (100 MHz) of the same wave,
//   (c) the HIP-event wall clock of the whole launch -- every CU busy with the same work, the launch long enough
//       (>= 5 ms) for launch overhead to vanish.
// Every CU gets `bpc` blocks of 256 threads (one wave per SIMD each), i.e. bpc waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -o valu_wallclock valu_wallclock.hip ; run: ./valu_wallclock
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// OP 0 v_fma_f64 | 1 v_fma_f32 | 2 v_add_u32 | 3 v_mul_f64 | 4 s_add_u32 (SALU only) | 5 v_fma_f64 + s_add_u32 alternating
// (16 instructions per REP8) | 6 v_cvt_f64_i32 | 7 v_rcp_f64 | 8 v_cmp_le_f64 (vcc) | 9 v_cndmask_b32 (vcc)
template <int OP>
__global__ __launch_bounds__(256) void k(long long *out, int iters, double seed) {
  double a[8], b = seed + threadIdx.x * 1e-3, c = 1.0000001;
  int ia[8];
  for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; ia[i] = threadIdx.x + i; }
  long long r0 = __builtin_amdgcn_s_memrealtime();
  long long t0 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int it = 0; it < iters; it++) {
#define X(i) \
    if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(b)); \
    else if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]), "v"(ia[(i + 2) & 7])); \
    else if (OP == 2) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7])); \
    else if (OP == 3) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c)); \
    else if (OP == 4) asm volatile("s_add_u32 s20, s20, 1" ::: "s20", "scc"); \
    else if (OP == 5) asm volatile("v_fma_f64 %0, %0, %1, %2\n\ts_add_u32 s20, s20, 1" : "+v"(a[i]) : "v"(c), "v"(b) : "s20", "scc"); \
    else if (OP == 6) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[i]) : "v"(ia[i])); \
    else if (OP == 7) asm volatile("v_rcp_f64 %0, %1" : "=v"(a[i]) : "v"(b)); \
    else if (OP == 8) asm volatile("v_cmp_le_f64 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc"); \
    else if (OP == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(ia[i]) : "v"(ia[(i + 1) & 7]) : "vcc");
    REP8(X)
#undef X
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0; long long li = 0;
  for (int i = 0; i < 8; i++) { s += a[i]; li += ia[i]; }
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[2] = r1 - r0; }
  if (s == 12345.678 && li == 77) out[1] = 1;
}

static const char *names[] = {"v_fma_f64", "v_fma_f32", "v_add_u32", "v_mul_f64", "s_add_u32", "v_fma_f64+s_add", "v_cvt_f64_i32", "v_rcp_f64", "v_cmp_le_f64", "v_cndmask_b32"};

template <int OP>
void run(long long *d, int cus) {
  const int iters = 1 << 17;  // 2^20 instructions per wave
  const double per_wave = (double)iters * 8.0;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int bpc : {1, 2, 4, 5, 8}) {
    long long h[3];
    float ms = 0.f;
    for (int rep = 0; rep < 2; rep++) {  // the second launch is the one reported (clocks up)
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k<OP>, dim3(cus * bpc), dim3(256), 0, 0, d, iters, 1.5);
      hipEventRecord(e1, 0);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    }
    const double per_simd = per_wave * bpc;  // wave-instructions one SIMD executed (pairs count once in OP 5)
    const double wall_ns = ms * 1e6;
    printf("%-16s waves/SIMD %d: wave0 %6.2f s_memtime ticks per instr; in-kernel clock %5.3f GHz (ticks / s_memrealtime x 100 MHz); "
           "per SIMD: %5.2f ticks, %5.3f ns by s_memrealtime, %5.3f ns by HIP events (launch %.2f ms) = %4.2f cycles at the in-kernel clock\n",
           names[OP], bpc, (double)h[0] / per_wave, (double)h[0] / (h[2] * 10.0), (double)h[0] / per_simd, (double)h[2] * 10.0 / per_simd,
           wall_ns / per_simd, ms, wall_ns / per_simd * ((double)h[0] / (h[2] * 10.0)));
  }
}

template <int OP> struct Runner { static void go(long long *d, int c) { run<OP>(d, c); Runner<OP + 1>::go(d, c); } };
template <> struct Runner<10> { static void go(long long *, int) {} };

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  int wall_khz = 0;
  hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
  printf("# %s: %d CUs, clockRate %d kHz, wall clock rate %d kHz; blocks of 256 threads, bpc per CU (= waves per SIMD), 2^20 instructions per wave\n",
         p.gcnArchName, p.multiProcessorCount, p.clockRate, wall_khz);
  long long *d;
  hipMalloc(&d, 64); hipMemset(d, 0, 64);
  Runner<0>::go(d, p.multiProcessorCount);
  return 0;
}
