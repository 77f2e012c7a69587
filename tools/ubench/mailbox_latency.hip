// mailbox_latency.hip -- how fast can a RESIDENT kernel be handed a request and answer it?
// A one-workgroup kernel polls a "doorbell" word and answers each new value by writing it to a response word in
// pinned host memory; the host measures doorbell-write -> response-visible round trips.  Doorbell placements:
//   host   : pinned, mapped host memory, polled by the kernel with system-scope loads (a PCIe read per poll)
//   device : fine-grained DEVICE memory (hipExtMallocWithFlags) written by the CPU through the PCIe BAR -- if the
//            platform lets the CPU dereference it (probed under a SIGSEGV / SIGBUS handler: a fault is the answer "no")
// plus, for comparison, the ordinary route: launch an empty kernel that writes the response word (enqueue -> word).
// Every kernel here is BOUNDED: it leaves after `max_requests` answers or `lifetime` ticks of s_memrealtime (100 MHz),
// whichever comes first, so a dead host cannot leave it spinning.
// Build: hipcc --offload-arch=gfx950 -O2 -o mailbox_latency mailbox_latency.hip ; run: ./mailbox_latency
#include <hip/hip_runtime.h>
#include <setjmp.h>
#include <signal.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

__global__ void k_resident(const unsigned long long *doorbell, unsigned long long *response, int system_scope,
                           int max_requests, long long lifetime) {
  if (threadIdx.x != 0) return;
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long last = 0;
  int served = 0;
  while (served < max_requests) {
    unsigned long long v;
    if (system_scope) v = __hip_atomic_load(doorbell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    else v = __hip_atomic_load(doorbell, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v != last) {
      last = v;
      __hip_atomic_store(response, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      served++;
    } else {
      __builtin_amdgcn_s_sleep(2);
    }
    if (__builtin_amdgcn_s_memrealtime() - t0 > lifetime) break;
  }
}

__global__ void k_answer(unsigned long long *response, unsigned long long v) {
  if (threadIdx.x == 0) __hip_atomic_store(response, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

static sigjmp_buf g_probe;
static void on_fault(int) { siglongjmp(g_probe, 1); }

static double now_us() {
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static void report(const char *name, std::vector<double> &t) {
  std::sort(t.begin(), t.end());
  printf("%-46s n=%zu  min %.2f  median %.2f  p90 %.2f  max %.2f us\n", name, t.size(), t.front(), t[t.size() / 2],
         t[t.size() * 9 / 10], t.back());
}

static bool run_resident(const char *name, volatile unsigned long long *doorbell_host_view,
                         unsigned long long *doorbell_dev, int system_scope, unsigned long long *resp, unsigned long long *resp_dev,
                         hipStream_t st) {
  const int n = 2000;
  *doorbell_host_view = 0;
  *resp = 0;
  hipLaunchKernelGGL(k_resident, dim3(1), dim3(64), 0, st, doorbell_dev, resp_dev, system_scope, n, 200000000LL /* 2 s */);
  CK(hipGetLastError());
  usleep(2000);
  std::vector<double> t;
  bool ok = true;
  for (int i = 1; i <= n && ok; i++) {
    const double t0 = now_us();
    *doorbell_host_view = (unsigned long long)i;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    while (__atomic_load_n(resp, __ATOMIC_ACQUIRE) != (unsigned long long)i) {
      if (now_us() - t0 > 200000.0) { printf("%s: request %d unanswered after 200 ms\n", name, i); ok = false; break; }
    }
    t.push_back(now_us() - t0);
  }
  CK(hipStreamSynchronize(st));  // the kernel leaves by itself (served == n, or its lifetime)
  if (ok) report(name, t);
  return ok;
}

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned long long *host = nullptr, *host_dev = nullptr;
  CK(hipHostMalloc((void **)&host, 4096, hipHostMallocMapped));
  CK(hipHostGetDevicePointer((void **)&host_dev, host, 0));
  unsigned long long *resp = host + 64, *resp_dev = host_dev + 64;  // its own cache line
  // 0. the ordinary route
  {
    std::vector<double> t;
    for (int i = 1; i <= 2000; i++) {
      *resp = 0;
      const double t0 = now_us();
      hipLaunchKernelGGL(k_answer, dim3(1), dim3(64), 0, st, resp_dev, (unsigned long long)i);
      while (__atomic_load_n(resp, __ATOMIC_ACQUIRE) != (unsigned long long)i) {}
      t.push_back(now_us() - t0);
    }
    CK(hipStreamSynchronize(st));
    report("launch of an empty kernel -> word on the host", t);
  }
  // 1. doorbell in host memory
  run_resident("resident kernel, doorbell in pinned host memory", host, host_dev, 1, resp, resp_dev, st);
  // 2. doorbell in fine-grained device memory, written by the CPU
  unsigned long long *fg = nullptr;
  hipError_t e = hipExtMallocWithFlags((void **)&fg, 4096, hipDeviceMallocFinegrained);
  if (e != hipSuccess) {
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
  } else {
    CK(hipMemset(fg, 0, 4096));
    CK(hipDeviceSynchronize());
    fflush(stdout);
    struct sigaction sa = {}, old_segv, old_bus;
    sa.sa_handler = on_fault;
    sigaction(SIGSEGV, &sa, &old_segv);
    sigaction(SIGBUS, &sa, &old_bus);
    bool addressable = false;
    if (sigsetjmp(g_probe, 1) == 0) {
      volatile unsigned long long *p = fg;
      *p = 1;
      addressable = (*p == 1);
    }
    sigaction(SIGSEGV, &old_segv, nullptr);
    sigaction(SIGBUS, &old_bus, nullptr);
    if (addressable) {
      printf("fine-grained device memory IS CPU-addressable here\n");
      CK(hipMemset(fg, 0, 4096));
      CK(hipDeviceSynchronize());
      run_resident("resident kernel, doorbell in device memory (BAR)", fg, fg, 0, resp, resp_dev, st);
      run_resident("  ... same, polled at system scope", fg, fg, 1, resp, resp_dev, st);
    } else {
      printf("fine-grained device memory is NOT CPU-addressable here (the store faulted)\n");
    }
  }
  return 0;
}
