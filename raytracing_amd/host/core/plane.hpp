// core/plane.hpp -- rt::Plane (reference: include/core/plane.cuh:11-39).
// The reference's Plane::hit is a stub that never hits (:25-29); the type is kept so
// that the geometry variant keeps its index order Sphere=0, Plane=1, Triangle=2.
#pragma once
#include "core/ray.hpp"
#include "core/transform.hpp"
#include "math/vec.hpp"

namespace rt {

class Plane {
public:
  Plane(float intercept, Vec3f normal) : intercept_(intercept), normal_(normal) {}
  float intercept() const { return intercept_; }
  const Vec3f &normal() const { return normal_; }
  void apply_transform(const Transform &) const {}
  bool can_hit(const Ray &) const { return false; } // plane.cuh:20-23: "TODO implement" in the reference

private:
  float intercept_;
  Vec3f normal_;
};

} // namespace rt
