// nid_png.cpp -- the wire formats of the reference's driver:
// PNG decoding (this image has zlib but neither libpng nor OpenCV) and the driver's colour -> grey
// conversion.  NID_pose_estimation.cpp:84-113 reads <dir>/<type>/<id>.png with imread(UNCHANGED) --
// OpenCV delivers B,G,R channel order -- and converts with cvtColor(CV_RGB2GRAY), i.e. it weights the
// FIRST channel (true blue) with the red coefficient.  That quirk is kept (swap_rb = 0 selects it);
// OpenCV's 8-bit path is fixed point: (c0*4899 + c1*9617 + c2*1868 + 8192) >> 14.  OpenCV is not
// available here, so this conversion is unpinned (restated from the published algorithm).
// Depth images are 16-bit grey PNGs (value / 5000 = metres).
//
// Supported: non-interlaced PNG, bit depth 8 or 16, colour types 0 (grey), 2 (RGB), 3 (palette, 8 bit),
// 4 (grey+alpha), 6 (RGBA).  Everything else is refused with an error code.
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "nid_pose_problem.h"

namespace {

uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

struct Png {
  int rows = 0, cols = 0, channels = 0, depth = 0;  // depth: bits per sample (8 | 16)
  std::vector<uint8_t> px;                          // rows * cols * channels samples; 16-bit ones big-endian
};

int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// 0 ok; -1 cannot open; -2 not a PNG / corrupt; -3 unsupported variant; -4 inflate failed
// Sizes come from the file: they are capped (kMaxPixels, and what the IDAT stream can possibly inflate to)
// BEFORE anything is allocated, chunk CRCs are checked, and the extern "C" entry points catch allocation
// failures -- nothing throws across the boundary.
constexpr uint64_t kMaxPixels = 64ull << 20;  // 64 MPix: far beyond any camera frame this driver is fed

uint32_t crc32_png(const uint8_t *p, size_t n) { return (uint32_t)crc32(crc32(0L, Z_NULL, 0), p, (uInt)n); }

int decode_png_impl(const char *path, Png *out) {
  FILE *f = std::fopen(path, "rb");
  if (!f) return -1;
  std::vector<uint8_t> file;
  uint8_t buf[65536];
  size_t n;
  while ((n = std::fread(buf, 1, sizeof(buf), f)) > 0) file.insert(file.end(), buf, buf + n);
  std::fclose(f);
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
  if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8)) return -2;
  size_t pos = 8;
  int colour = -1, interlace = 0;
  std::vector<uint8_t> idat, plte;
  bool have_ihdr = false, end = false;
  while (!end && pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const uint8_t *type = &file[pos + 4], *data = &file[pos + 8];
    if (pos + 12 + (size_t)len > file.size()) return -2;
    if (crc32_png(type, 4 + (size_t)len) != be32(data + len)) return -2;  // chunk CRC covers type + data
    if (!std::memcmp(type, "IHDR", 4)) {
      if (len != 13) return -2;
      out->cols = (int)be32(data); out->rows = (int)be32(data + 4);
      out->depth = data[8]; colour = data[9]; interlace = data[12];
      if (data[10] != 0 || data[11] != 0) return -2;
      have_ihdr = true;
    } else if (!std::memcmp(type, "PLTE", 4)) {
      plte.assign(data, data + len);
    } else if (!std::memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!std::memcmp(type, "IEND", 4)) {
      end = true;
    }
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || idat.empty() || out->rows <= 0 || out->cols <= 0) return -2;
  if ((uint64_t)out->rows * (uint64_t)out->cols > kMaxPixels) return -3;
  if (interlace != 0 || (out->depth != 8 && out->depth != 16)) return -3;
  int ch;
  switch (colour) {
    case 0: ch = 1; break;
    case 2: ch = 3; break;
    case 3: ch = 1; if (out->depth != 8 || plte.size() < 3) return -3; break;
    case 4: ch = 2; break;
    case 6: ch = 4; break;
    default: return -3;
  }
  const size_t bps = (size_t)out->depth / 8, bpp = bps * ch, stride = bpp * out->cols;
  // deflate cannot expand by more than ~1032:1: a header that promises more than the IDAT stream can hold is corrupt
  if ((uint64_t)out->rows * (stride + 1) > (uint64_t)idat.size() * 1032ull + 1024ull) return -2;
  std::vector<uint8_t> raw((size_t)out->rows * (stride + 1));
  uLongf rawlen = (uLongf)raw.size();
  if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return -4;
  std::vector<uint8_t> img((size_t)out->rows * stride);
  for (int r = 0; r < out->rows; r++) {
    const uint8_t *src = &raw[(size_t)r * (stride + 1)];
    uint8_t *cur = &img[(size_t)r * stride];
    const uint8_t *up = r ? cur - stride : nullptr;
    const int ft = src[0];
    if (ft > 4) return -2;
    for (size_t i = 0; i < stride; i++) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
      int pred = 0;
      if (ft == 1) pred = a;
      else if (ft == 2) pred = b;
      else if (ft == 3) pred = (a + b) >> 1;
      else if (ft == 4) pred = paeth(a, b, c);
      cur[i] = (uint8_t)(src[1 + i] + pred);
    }
  }
  if (colour == 3) {  // expand the palette to RGB
    out->channels = 3;
    out->px.resize((size_t)out->rows * out->cols * 3);
    for (size_t i = 0; i < (size_t)out->rows * out->cols; i++) {
      const size_t k = (size_t)img[i] * 3;
      if (k + 2 >= plte.size()) return -2;
      out->px[3 * i] = plte[k]; out->px[3 * i + 1] = plte[k + 1]; out->px[3 * i + 2] = plte[k + 2];
    }
  } else {
    out->channels = ch;
    out->px.swap(img);
  }
  return 0;
}

int decode_png(const char *path, Png *out) {
  try {
    return decode_png_impl(path, out);
  } catch (const std::bad_alloc &) {
    return -4;
  } catch (...) {
    return -2;
  }
}

}  // namespace

extern "C" {

int nid_png_info(const char *path, int *rows, int *cols, int *channels, int *bit_depth) {
  Png p;
  const int rc = decode_png(path, &p);
  if (rc) return rc;
  if (rows) *rows = p.rows;
  if (cols) *cols = p.cols;
  if (channels) *channels = p.channels;
  if (bit_depth) *bit_depth = p.depth;
  return 0;
}

int nid_png_read_gray_u8(const char *path, int swap_rb, int *rows, int *cols, uint8_t *out, size_t cap) {
  Png p;
  const int rc = decode_png(path, &p);
  if (rc) return rc;
  if (p.depth != 8) return -3;
  const size_t N = (size_t)p.rows * p.cols;
  if (rows) *rows = p.rows;
  if (cols) *cols = p.cols;
  if (!out) return 0;
  if (cap < N) return -5;
  for (size_t i = 0; i < N; i++) {
    const uint8_t *s = &p.px[i * p.channels];
    if (p.channels <= 2) { out[i] = s[0]; continue; }  // already grey (the reference's cvtColor would reject it)
    // file order is R,G,B; imread(UNCHANGED) hands OpenCV B,G,R and CV_RGB2GRAY weights channel 0 as "R":
    // reference quirk (swap_rb == 0): 0.299*B + 0.587*G + 0.114*R; swap_rb != 0: the conventional luma
    const int c0 = swap_rb ? s[0] : s[2], c1 = s[1], c2 = swap_rb ? s[2] : s[0];
    out[i] = (uint8_t)((c0 * 4899 + c1 * 9617 + c2 * 1868 + 8192) >> 14);
  }
  return 0;
}

int nid_png_read_u16(const char *path, int *rows, int *cols, uint16_t *out, size_t cap) {
  Png p;
  const int rc = decode_png(path, &p);
  if (rc) return rc;
  if (p.channels != 1) return -3;
  const size_t N = (size_t)p.rows * p.cols;
  if (rows) *rows = p.rows;
  if (cols) *cols = p.cols;
  if (!out) return 0;
  if (cap < N) return -5;
  for (size_t i = 0; i < N; i++)
    out[i] = p.depth == 16 ? (uint16_t)((p.px[2 * i] << 8) | p.px[2 * i + 1]) : (uint16_t)p.px[i];
  return 0;
}

}  // extern "C"
