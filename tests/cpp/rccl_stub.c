/* rccl_stub.c -- a librccl whose ncclCommInitRank never returns (tests/test_multi_gpu.py: bench.py --gpus 2 must end
 * with an error line and a non-zero exit code instead of hanging until the driver's timeout).  Exports exactly the
 * symbols csrc/nid_multi.inc resolves.  Built by the test with gcc -shared -fPIC. */
#include <string.h>
#include <unistd.h>

typedef struct { char internal[128]; } ncclUniqueId;
typedef void *ncclComm_t;
typedef int ncclResult_t;

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) { memset(id, 7, sizeof(*id)); return 0; }
ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
  (void)comm; (void)nranks; (void)id; (void)rank;
  for (;;) pause();
  return 0;
}
ncclResult_t ncclCommInitAll(ncclComm_t *comm, int ndev, const int *devlist) { (void)comm; (void)ndev; (void)devlist; return 1; }
ncclResult_t ncclCommDestroy(ncclComm_t comm) { (void)comm; return 0; }
ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) { (void)comm; *count = 0; return 0; }
ncclResult_t ncclAllReduce(const void *s, void *r, size_t n, int dt, int op, ncclComm_t c, void *st) {
  (void)s; (void)r; (void)n; (void)dt; (void)op; (void)c; (void)st;
  return 1;
}
ncclResult_t ncclGroupStart(void) { return 0; }
ncclResult_t ncclGroupEnd(void) { return 0; }
const char *ncclGetErrorString(ncclResult_t r) { (void)r; return "rccl stub"; }
