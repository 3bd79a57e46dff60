// io/hdr_writer.hpp -- Radiance RGBE (.hdr) writer, replacing stbi_write_hdr
// (reference: src/main.cu:257; deps/stb is an empty, unpinned submodule).  Writes flat
// (non-RLE) scanlines, which every reader accepts.  Linear values, no tone mapping.
// Parity with stb's exact bytes is unpinned (no stb in the image, no reference test covers it).
#pragma once
#include <cmath>
#include <cstdio>

#include "container/image_buffer.hpp"

namespace rt {

inline void float_to_rgbe(const Vec3f &c, unsigned char out[4])
{
  float m = c.x > c.y ? (c.x > c.z ? c.x : c.z) : (c.y > c.z ? c.y : c.z);
  if (!(m >= 1e-32f)) { out[0] = out[1] = out[2] = out[3] = 0; return; }
  int e;
  float scale = std::frexp(m, &e) * 256.0f / m;
  out[0] = static_cast<unsigned char>(c.x * scale);
  out[1] = static_cast<unsigned char>(c.y * scale);
  out[2] = static_cast<unsigned char>(c.z * scale);
  out[3] = static_cast<unsigned char>(e + 128);
}

// top scanline first, i.e. call after ImageBuffer::flip_vertical()
inline bool write_hdr(const char *path, const ImageBuffer &img)
{
  std::FILE *f = std::fopen(path, "wb");
  if (!f) return false;
  std::fprintf(f, "#?RADIANCE\nFORMAT=32-bit_rle_rgbe\n\n-Y %zu +X %zu\n", img.height(), img.width());
  unsigned char px[4];
  for (std::size_t i = 0; i < img.width() * img.height(); ++i) {
    Vec3f c = img.buffer()[i];
    if (c.x < 0) c.x = 0;
    if (c.y < 0) c.y = 0;
    if (c.z < 0) c.z = 0;
    float_to_rgbe(c, px);
    std::fwrite(px, 1, 4, f);
  }
  return std::fclose(f) == 0;
}

} // namespace rt
