// pipeline/multisample_mesh.hpp -- rt::MultisampleMesh, the fixed 11x11 sub-pixel pattern
// (reference: include/pipeline/multisample_mesh.cuh:10-45).  The GPU regenerates the same
// offsets from (norm_x, norm_y) with the same expressions; this host copy is for tooling.
#pragma once
#include <array>
#include <cstddef>

#include "math/vec.hpp"

namespace rt {

class MultisampleMesh {
public:
  static constexpr std::size_t WIDTH = 11;
  static constexpr std::size_t HEIGHT = 11;
  using Samples = std::array<Vec2f, WIDTH * HEIGHT>;

  MultisampleMesh(float norm_x, float norm_y)
  {
    constexpr auto DIV_H = float(HEIGHT + 2);
    constexpr auto DIV_W = float(WIDTH + 2);
    for (std::size_t r = 0u; r < HEIGHT; ++r) {
      float y = (-1.0f + 2.0f * float(r + 1) / float(DIV_H)) / norm_y;
      for (std::size_t c = 0u; c < WIDTH; ++c) {
        float x = (-1.0f + 2.0f * float(c + 1) / float(DIV_W)) / norm_x;
        offsets_[r * WIDTH + c] = Vec2f{x, y};
      }
    }
  }

  Samples generate_samples(Vec2f pixel) const
  {
    Samples s;
    for (std::size_t i = 0; i < WIDTH * HEIGHT; ++i) s[i] = pixel + offsets_[i];
    return s;
  }

private:
  Samples offsets_;
};

} // namespace rt
