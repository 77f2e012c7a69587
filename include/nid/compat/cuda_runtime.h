// cuda_runtime.h -- forwarding header for the three CUDA runtime calls the reference's main() makes around the
// operators (NID_pose_estimation.cpp:57 cudaFree(0); :240-242 cudaMallocManaged of im0 / im1 / points3d;
// :388-390 cudaFree), so that the file compiles unchanged against include/nid/legacy_ops.h.  The operators of
// this library take HOST pointers (they stage what they need themselves), so "managed" memory is plain memory.
// Put include/nid/compat on the include path ONLY for that translation unit.
#pragma once
#include <cstddef>
#include <cstdlib>

typedef int cudaError_t;
enum { cudaSuccess = 0, cudaErrorMemoryAllocation = 2 };

inline cudaError_t cudaFree(void *p) { std::free(p); return cudaSuccess; }
template <class T>
inline cudaError_t cudaMallocManaged(T **p, size_t bytes) {
  *p = static_cast<T *>(std::malloc(bytes));
  return *p ? cudaSuccess : cudaErrorMemoryAllocation;
}
inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
inline const char *cudaGetErrorString(cudaError_t e) { return e == cudaSuccess ? "no error" : "out of memory"; }
