// tests/cpp/png_fuzz.cpp ITERATIONS WORKDIR SEED.png...: mutation fuzz of host/nid_png.cpp, built with ASan + UBSan by
// tests/test_host_cpu.py::test_png_reader_survives_mutated_files -- valid PNGs with flipped bits / bytes, truncations, rewritten IHDR
// fields and chunk lengths, inserted runs; three times out of four the chunk CRCs are re-made so that the parser BEHIND the CRC
// check is reached too.  Every mutant goes through nid_png_info and, when accepted, the two readers (once with a short caller
// buffer).  Prints the counts; a finding is a sanitizer report.  (200 000 mutants in round 6: none.)
#include <zlib.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <random>
#include <string>
extern "C" {
int nid_png_info(const char *path, int *rows, int *cols, int *channels, int *bit_depth);
int nid_png_read_gray_u8(const char *path, int swap_rb, int *rows, int *cols, uint8_t *out, size_t cap);
int nid_png_read_u16(const char *path, int *rows, int *cols, uint16_t *out, size_t cap);
}
static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }
static void fix_crcs(std::vector<uint8_t> &f) {
  size_t pos = 8;
  while (pos + 12 <= f.size()) {
    const uint32_t len = be32(&f[pos]);
    if ((uint64_t)pos + 12 + len > f.size()) break;
    const uint32_t c = (uint32_t)crc32(crc32(0L, Z_NULL, 0), &f[pos + 4], len + 4);
    f[pos + 8 + len] = c >> 24; f[pos + 9 + len] = c >> 16; f[pos + 10 + len] = c >> 8; f[pos + 11 + len] = c;
    pos += 12 + len;
  }
}
int main(int argc, char **argv) {
  const int iters = atoi(argv[1]);
  const std::string tmp = std::string(argv[2]) + "/mutant.png";
  std::vector<std::vector<uint8_t>> seeds;
  for (int a = 3; a < argc; a++) {
    FILE *fp = fopen(argv[a], "rb"); std::vector<uint8_t> v; uint8_t b[4096]; size_t n;
    while ((n = fread(b, 1, sizeof b, fp)) > 0) v.insert(v.end(), b, b + n);
    fclose(fp); seeds.push_back(v);
  }
  std::mt19937_64 rng(12345);
  std::vector<uint8_t> out8(1 << 22); std::vector<uint16_t> out16(1 << 22);
  long ok = 0, refused = 0;
  for (int it = 0; it < iters; it++) {
    std::vector<uint8_t> f = seeds[rng() % seeds.size()];
    const int kind = rng() % 6;
    const int nm = 1 + rng() % 4;
    for (int m = 0; m < nm; m++) {
      if (kind == 0 && f.size() > 40) f.resize(8 + rng() % (f.size() - 8));                 // truncate
      else if (kind == 1) f[rng() % f.size()] ^= (uint8_t)(1u << (rng() % 8));                 // bit flip
      else if (kind == 2) f[rng() % f.size()] = (uint8_t)rng();                                 // byte
      else if (kind == 3 && f.size() > 33) f[16 + rng() % 13] = (uint8_t)rng();               // IHDR fields
      else if (kind == 4 && f.size() > 60) { size_t a = 33 + rng() % (f.size() - 40); f[a] = (uint8_t)rng(); f[a + 1] = (uint8_t)rng(); }  // chunk lengths / payload
      else if (kind == 5) { size_t a = rng() % f.size(), n = rng() % 64; f.insert(f.begin() + a, n, (uint8_t)rng()); }    // insert
    }
    if (rng() % 4) fix_crcs(f);
    FILE *fp = fopen(tmp.c_str(), "wb"); fwrite(f.data(), 1, f.size(), fp); fclose(fp);
    int r, c, ch, bd;
    int rc = nid_png_info(tmp.c_str(), &r, &c, &ch, &bd);
    if (rc == 0) {
      ok++;
      if ((size_t)r * c <= out8.size()) {
        nid_png_read_gray_u8(tmp.c_str(), (int)(rng() & 1), &r, &c, out8.data(), out8.size());
        nid_png_read_u16(tmp.c_str(), &r, &c, out16.data(), out16.size());
        nid_png_read_gray_u8(tmp.c_str(), 0, &r, &c, out8.data(), 10);  // short caller buffer
      }
    } else refused++;
  }
  printf("%d mutants: %ld decoded, %ld refused\n", iters, ok, refused);
  return 0;
}
