// container/image_buffer.hpp -- rt::ImageBuffer (reference: include/container/alloc.cuh:13-106).
// Linear float RGB, row-major, row 0 = v 0 (the reference's pre-flip order).  Host memory:
// the device-side framebuffer lives behind the C-ABI.  Also carries the hit/ID and t buffers
// the reference keeps only in locals (src/main.cu:61-62).
#pragma once
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

#include "math/vec.hpp"

namespace rt {

class ImageBuffer {
public:
  ImageBuffer(std::size_t width, std::size_t height)
    : width_(width), height_(height), rgb_(width * height), id_(width * height, -1), t_(width * height, 0.0f)
  {}
  std::size_t width() const { return width_; }
  std::size_t height() const { return height_; }
  Vec3f *buffer() { return rgb_.data(); }
  const Vec3f *buffer() const { return rgb_.data(); }
  std::int32_t *ids() { return id_.data(); }
  const std::int32_t *ids() const { return id_.data(); }
  float *ts() { return t_.data(); }
  const float *ts() const { return t_.data(); }

  // vertical flip done correctly (the reference indexes row height-r, one past the end for
  // r = 0: src/main.cu:249-255)
  void flip_vertical()
  {
    for (std::size_t r = 0; r < height_ / 2; ++r)
      for (std::size_t c = 0; c < width_; ++c) {
        std::size_t lhs = r * width_ + c, rhs = (height_ - 1 - r) * width_ + c;
        std::swap(rgb_[lhs], rgb_[rhs]);
        std::swap(id_[lhs], id_[rhs]);
        std::swap(t_[lhs], t_[rhs]);
      }
  }

private:
  std::size_t width_, height_;
  std::vector<Vec3f> rgb_;
  std::vector<std::int32_t> id_;
  std::vector<float> t_;
};

static_assert(sizeof(Vec3f) == 12, "Vec3f must be 3 packed floats");

} // namespace rt
