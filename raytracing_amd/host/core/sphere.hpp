// core/sphere.hpp -- rt::Sphere (reference: include/core/sphere.cuh:15-118).
// Host-side description only: the hit test runs on the GPU (csrc/mrt_kernels.hip);
// rt::hit() in pipeline/intersect.hpp is the host entry to it.
#pragma once
#include "core/transform.hpp"
#include "math/vec.hpp"

namespace rt {

class Sphere {
public:
  Sphere() = default;
  Sphere(const Vec3f &c, float r) : center_(c), radius_(r) {}
  const Vec3f &center() const { return center_; }
  float radius() const { return radius_; }
  void apply_transform(const Transform &t) { center_ = t(center_); }

private:
  Vec3f center_;
  float radius_ = 0.0f;
};

} // namespace rt
