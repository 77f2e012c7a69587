// nid_hostsum.cpp -- host side of the DIRECT results (nid_capi.hip, wait_direct): one cell's Jacobian record added to the
// 27 running sums b[6], H upper[21] of the normal equations, with AVX-512 where the host CPU has it.
// The sums are what the in-launch reduction forms (sum_blocks_w0 in nid_kernels.hip.h): per entry (J[a] * rho1) * J[b],
// resp. 0.0 - (rho1 * J[n]) * err, IEEE mul / sub / add in that order -- the vector lanes do exactly the scalar
// operations, so the two paths give the same bits (tests/test_parity_gpu.py::test_direct_results_equal_in_launch_reduction
// runs on whatever the box's CPU takes; NID_DIRECT_SCALAR=1 forces the scalar loop).
// Why: behind the last record's arrival the host still has all 256 cells' quadratic forms to add -- 2.4 us of scalar
// arithmetic on config A (NID_DIRECT_TRACE), a sixth of a dependent evaluation; the cells finish within 0.4 us of each
// other, so this work cannot hide behind the device.
// Plain C++ (host only): built into libnid_hip.so next to nid_capi.hip.
#include <immintrin.h>

// Every product and sum below is an operation of its own: a compiler that fuses the multiplication into the addition
// (fp-contract, on by default in GNU mode once the target has FMA) would give other bits than the scalar loop and the
// device.  The library is built with -ffp-contract=off; this file says so itself as well.
#if defined(__clang__)
#pragma clang fp contract(off)
#elif defined(__GNUC__)
#pragma GCC optimize("fp-contract=off")
#endif

namespace {

struct Tables {
  alignas(64) long long a[4][8], b[4][8];
  Tables() {
    // acc layout (kReducedLen prefix): [0] chi2 | [1..6] b | [7..27] H upper, row-major | [28] count | [29..31] unused
    // source vectors: Jr = rho1 * [J0..J5, 0, 0]; Jb = [J0..J5, err, 0]: index 7 reads 0.0 in both
    for (int k = 0; k < 4; k++)
      for (int l = 0; l < 8; l++) { a[k][l] = 7; b[k][l] = 7; }
    for (int n = 0; n < 6; n++) { a[0][1 + n] = n; b[0][1 + n] = 6; }   // (rho1 * J[n]) * err
    int v = 7;
    for (int p = 0; p < 6; p++)
      for (int q = p; q < 6; q++, v++) { a[v / 8][v % 8] = p; b[v / 8][v % 8] = q; }  // (J[p] * rho1) * J[q]
  }
};
const Tables g_tab;

}  // namespace

extern "C" {

__attribute__((visibility("hidden"))) int nid_hostsum_have_avx512(void) { return __builtin_cpu_supports("avx512f") ? 1 : 0; }

// acc: 32 doubles, 64-byte aligned; rec: the cell's record [J0..J5, 0, 0], 64-byte aligned
__attribute__((visibility("hidden"), target("avx512f")))
void nid_hostsum_jac_avx512(const double *rec, double err, double rho1, double *acc) {
  const __m512d Jz = _mm512_load_pd(rec);
  const __m512d Jr = _mm512_mul_pd(Jz, _mm512_set1_pd(rho1));
  const __m512d Jb = _mm512_mask_mov_pd(Jz, 0x40, _mm512_set1_pd(err));
  const __m512d zero = _mm512_setzero_pd();
#pragma GCC unroll 4
  for (int k = 0; k < 4; k++) {
    const __m512d A = _mm512_permutexvar_pd(_mm512_load_si512(g_tab.a[k]), Jr);
    const __m512d B = _mm512_permutexvar_pd(_mm512_load_si512(g_tab.b[k]), Jb);
    __m512d P = _mm512_mul_pd(A, B);
    if (k == 0) P = _mm512_mask_sub_pd(P, 0x7E, zero, P);  // b entries: 0.0 - (rho1 * J[n]) * err
    _mm512_store_pd(acc + 8 * k, _mm512_add_pd(_mm512_load_pd(acc + 8 * k), P));
  }
}

// The same with the record's arrival test and its re-arming folded in: ONE 64-byte load of the record -- if any of its
// eight words still holds `sentinel` the record has not (entirely) arrived and nothing happens (returns 0); otherwise the
// loaded words are the record (every word is written once, so a snapshot without a sentinel is the whole record), the
// sums take it (`active` == 0: a level-1 edge, nothing to add) and the line is filled with the sentinel again.
__attribute__((visibility("hidden"), target("avx512f")))
int nid_hostsum_take_jac_avx512(double *rec, unsigned long long sentinel, int active, double err, double rho1, double *acc) {
  const __m512i raw = _mm512_load_si512(rec);
  const __m512i sv = _mm512_set1_epi64((long long)sentinel);
  if (_mm512_cmpeq_epi64_mask(raw, sv) != 0) return 0;
  _mm512_store_si512(rec, sv);
  if (!active) return 1;
  const __m512d Jz = _mm512_castsi512_pd(raw);
  const __m512d Jr = _mm512_mul_pd(Jz, _mm512_set1_pd(rho1));
  const __m512d Jb = _mm512_mask_mov_pd(Jz, 0x40, _mm512_set1_pd(err));
  const __m512d zero = _mm512_setzero_pd();
#pragma GCC unroll 4
  for (int k = 0; k < 4; k++) {
    const __m512d A = _mm512_permutexvar_pd(_mm512_load_si512(g_tab.a[k]), Jr);
    const __m512d B = _mm512_permutexvar_pd(_mm512_load_si512(g_tab.b[k]), Jb);
    __m512d P = _mm512_mul_pd(A, B);
    if (k == 0) P = _mm512_mask_sub_pd(P, 0x7E, zero, P);
    _mm512_store_pd(acc + 8 * k, _mm512_add_pd(_mm512_load_pd(acc + 8 * k), P));
  }
  return 1;
}

}  // extern "C"
