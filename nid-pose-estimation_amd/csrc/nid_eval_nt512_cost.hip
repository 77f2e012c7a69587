// one translation unit of evaluation kernels: see nid_eval_tu.inc
#define NID_TU_NT 512
#define NID_TU_JAC 0
#include "nid_eval_tu.inc"
