// core/sphere.hpp -- rt::Sphere (reference: include/core/sphere.cuh:15-118).
// Host-side description only: the hit test runs on the GPU (csrc/mrt_kernels.hip);
// rt::hit() in pipeline/intersect.hpp is the host entry to it.
#pragma once
#include "core/ray.hpp"
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
  // sphere.cuh:26-30: a strictly positive discriminant (a tangent ray, which hit() accepts, says no).  Not on the
  // render path; evaluated on the host with the hit test's fp32 operations in their order (sphere.cuh:89-114).
  bool can_hit(const Ray &ray) const
  {
    const Vec3f oc = ray.origin() - center_;
    const float a = ray.direction().dot(ray.direction());
    const float b = oc.dot(ray.direction());
    const float c = oc.dot(oc) - radius_ * radius_;
    return b * b - a * c > 0.0f;
  }

private:
  Vec3f center_;
  float radius_ = 0.0f;
};

} // namespace rt
