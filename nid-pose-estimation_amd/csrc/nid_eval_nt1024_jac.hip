// one translation unit of evaluation kernels: see nid_eval_tu.inc
#define NID_TU_NT 1024
#define NID_TU_JAC 1
#include "nid_eval_tu.inc"
