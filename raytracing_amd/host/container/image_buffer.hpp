// container/image_buffer.hpp -- rt::ImageBuffer (reference: include/container/alloc.cuh:13-106).
// Linear float RGB, row-major, row 0 = v 0 (the reference's pre-flip order).  The reference keeps the image in
// managed memory (cudaMallocManaged, alloc.cuh:30); here it is page-locked host memory from the back end
// (mrt_host_alloc), so that a rendered frame arrives at the speed of the link -- plain memory when no GPU
// runtime answers (host-only tools and tests).  Move-only, like the reference's.  Also carries the hit/ID and
// t buffers the reference keeps only in locals (src/main.cu:61-62).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <new>
#include <utility>

#include "math/vec.hpp"
#include "mrt.h"

namespace rt {

class ImageBuffer {
public:
  ImageBuffer(std::size_t width, std::size_t height) : width_(width), height_(height)
  {
    const std::size_t n = width * height;
    const std::size_t bytes = n * (sizeof(Vec3f) + sizeof(std::int32_t) + sizeof(float)) + 64;
    void *p = nullptr;
    if (mrt_host_alloc(bytes, &p) == MRT_OK && p) {
      pinned_ = true;
    } else {
      p = std::malloc(bytes);
      if (!p) throw std::bad_alloc();
    }
    base_ = p;
    rgb_ = static_cast<Vec3f *>(p);
    id_ = reinterpret_cast<std::int32_t *>(rgb_ + n);
    t_ = reinterpret_cast<float *>(id_ + n);
    for (std::size_t i = 0; i < n; ++i) { rgb_[i] = Vec3f{}; id_[i] = -1; t_[i] = 0.0f; }
  }
  ImageBuffer(const ImageBuffer &) = delete;
  ImageBuffer &operator=(const ImageBuffer &) = delete;
  ImageBuffer(ImageBuffer &&o) noexcept { steal(o); }
  ImageBuffer &operator=(ImageBuffer &&o) noexcept
  {
    if (this != &o) { release(); steal(o); }
    return *this;
  }
  ~ImageBuffer() { release(); }

  std::size_t width() const { return width_; }
  std::size_t height() const { return height_; }
  bool page_locked() const { return pinned_; }
  Vec3f *buffer() { return rgb_; }
  const Vec3f *buffer() const { return rgb_; }
  std::int32_t *ids() { return id_; }
  const std::int32_t *ids() const { return id_; }
  float *ts() { return t_; }
  const float *ts() const { return t_; }

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
  void release()
  {
    if (!base_) return;
    if (pinned_) mrt_host_free(base_); else std::free(base_);
    base_ = nullptr;
  }
  void steal(ImageBuffer &o)
  {
    width_ = o.width_; height_ = o.height_; base_ = o.base_; pinned_ = o.pinned_;
    rgb_ = o.rgb_; id_ = o.id_; t_ = o.t_;
    o.base_ = nullptr; o.rgb_ = nullptr; o.id_ = nullptr; o.t_ = nullptr; o.width_ = o.height_ = 0;
  }

  std::size_t width_ = 0, height_ = 0;
  void *base_ = nullptr;
  bool pinned_ = false;
  Vec3f *rgb_ = nullptr;
  std::int32_t *id_ = nullptr;
  float *t_ = nullptr;
};

static_assert(sizeof(Vec3f) == 12, "Vec3f must be 3 packed floats");

} // namespace rt
