// core/ray.hpp -- rt::Ray (reference: include/core/ray.cuh:7-30)
#pragma once
#include "math/vec.hpp"

namespace rt {

class Ray {
public:
  Ray(const Vec3f &o, const Vec3f &d) : origin_(o), direction_(d) {}
  const Vec3f &origin() const { return origin_; }
  const Vec3f &direction() const { return direction_; }
  Vec3f &origin() { return origin_; }
  Vec3f &direction() { return direction_; }
  Vec3f point_at_param(float t) const { return origin_ + t * direction_; }

private:
  Vec3f origin_, direction_;
};

} // namespace rt
