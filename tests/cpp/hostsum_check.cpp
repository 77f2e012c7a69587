// tests/cpp/hostsum_check.cpp -- csrc/nid_hostsum.cpp (the AVX-512 sums of the DIRECT results) against the scalar loop of
// csrc/nid_capi.hip (add_record_jac), bit for bit, on random records: plain values, zeros, signed zeros, huge and tiny
// magnitudes, inactive cells, records that have not arrived.  Built and run by tests/test_host_cpu.py (skipped on a CPU
// without AVX-512).  Exit code 0 = identical, 1 = mismatch, 2 = no AVX-512 here.
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>

extern "C" int nid_hostsum_have_avx512(void);
extern "C" void nid_hostsum_jac_avx512(const double *rec, double err, double rho1, double *acc);
extern "C" int nid_hostsum_take_jac_avx512(double *rec, unsigned long long sentinel, int active, double err, double rho1, double *acc);

// the scalar form, as in nid_capi.hip
static void add_record_jac(const double *J, double err, double rho1, double *acc) {
  for (int n = 0; n < 6; n++) acc[1 + n] += 0.0 - (rho1 * J[n]) * err;
  int idx = 7;
  for (int a = 0; a < 6; a++)
    for (int b = a; b < 6; b++, idx++) acc[idx] += (J[a] * rho1) * J[b];
}

int main() {
  if (!nid_hostsum_have_avx512()) { std::puts("no AVX-512 on this CPU"); return 2; }
  const unsigned long long sentinel = 0x7FF4DEADBEEF5A5Aull;
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> uni(-1.0, 1.0);
  alignas(64) double acc_s[32] = {}, acc_v[32] = {}, acc_t[32] = {}, rec[8], rec2[8];
  long bad = 0;
  for (int it = 0; it < 200000; it++) {
    const int kind = it % 7;
    for (int n = 0; n < 6; n++) {
      double x = uni(rng);
      if (kind == 1) x = std::ldexp(x, (int)(rng() % 600) - 300);
      if (kind == 2 && n % 2) x = 0.0;
      if (kind == 3 && n % 3 == 0) x = -0.0;
      rec[n] = x;
    }
    rec[6] = 0.0; rec[7] = 0.0;
    double err = uni(rng) * (kind == 4 ? 1e6 : 1.0), rho1 = kind == 5 ? 1.0 : 0.5 + 0.5 * uni(rng) * uni(rng);
    if (it % 5000 == 0) std::memset(acc_s, 0, sizeof(acc_s)), std::memset(acc_v, 0, sizeof(acc_v)), std::memset(acc_t, 0, sizeof(acc_t));
    add_record_jac(rec, err, rho1, acc_s);
    nid_hostsum_jac_avx512(rec, err, rho1, acc_v);
    std::memcpy(rec2, rec, sizeof(rec));
    const int active = kind != 6;
    if (!active) {  // an inactive cell's record is consumed and adds nothing
      double keep[32];
      std::memcpy(keep, acc_t, sizeof(keep));
      if (nid_hostsum_take_jac_avx512(rec2, sentinel, 0, err, rho1, acc_t) != 1 || std::memcmp(keep, acc_t, sizeof(keep)) != 0) bad++;
      add_record_jac(rec, err, rho1, acc_t);  // keep the three accumulators in step
    } else if (nid_hostsum_take_jac_avx512(rec2, sentinel, 1, err, rho1, acc_t) != 1) {
      bad++;
    }
    for (int w = 0; w < 8; w++) {  // ... and re-armed
      unsigned long long bits;
      std::memcpy(&bits, &rec2[w], 8);
      if (bits != sentinel) bad++;
    }
    // a record with one word still missing is left alone
    std::memcpy(rec2, rec, sizeof(rec));
    std::memcpy(&rec2[it % 8], &sentinel, 8);
    double keep[32];
    std::memcpy(keep, acc_t, sizeof(keep));
    if (nid_hostsum_take_jac_avx512(rec2, sentinel, 1, err, rho1, acc_t) != 0 || std::memcmp(keep, acc_t, sizeof(keep)) != 0) bad++;
    for (int v = 1; v < 28; v++) {
      if (std::memcmp(&acc_s[v], &acc_v[v], 8) != 0 || std::memcmp(&acc_s[v], &acc_t[v], 8) != 0) {
        if (bad < 5) std::printf("mismatch it %d entry %d: scalar %a vector %a take %a\n", it, v, acc_s[v], acc_v[v], acc_t[v]);
        bad++;
      }
    }
  }
  std::printf("%ld mismatches in 200000 records\n", bad);
  return bad ? 1 : 0;
}
