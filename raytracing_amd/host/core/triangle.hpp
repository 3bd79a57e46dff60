// core/triangle.hpp -- rt::Triangle (reference: include/core/triangle.cuh:13-131).
//
// As in the reference, the constructor runs on the HOST and leaves behind the
// precomputed form the hit test consumes: plane normal + intercept and three edge
// discriminators.  Reference quirks kept on purpose:
//  - the edge-plane normal is dir x point_inside, with point_inside used as a
//    position vector (triangle.cuh:39-40);
//  - the side test compares a value that is mathematically 0 against -1e-6 (:23).
// Unlike the reference, a_, b_, c_ are actually stored, so apply_transform works.
#pragma once
#include "core/ray.hpp"
#include "core/transform.hpp"
#include "math/vec.hpp"

namespace rt {

namespace internal {

class LineDiscriminator {
public:
  LineDiscriminator(Vec3f line_start, Vec3f line_end, Vec3f point_inside)
  {
    Vec3f dir = (line_end - line_start).normalized();
    normal_ = dir.cross(point_inside);
    intercept_ = normal_.dot(line_start);
    float side = (normal_.dot(point_inside) - intercept_) > -1e-6f ? 1.0f : -1.0f;
    normal_ *= side;
    intercept_ *= side;
  }
  const Vec3f &normal() const { return normal_; }
  float intercept() const { return intercept_; }

private:
  Vec3f normal_;
  float intercept_;
};

} // namespace internal

class Triangle {
public:
  Triangle(Vec3f a, Vec3f b, Vec3f c)
    : plane_normal_(((a - b).cross(c - a)).normalized()), plane_intercept_(), disc_ab_to_c_(a, b, c),
      disc_bc_to_a_(b, c, a), disc_ca_to_b_(c, a, b), a_(a), b_(b), c_(c)
  {
    plane_intercept_ = plane_normal_.dot(a);
  }

  void apply_transform(const Transform &t) { *this = Triangle(t(a_), t(b_), t(c_)); }
  // triangle.cuh:68-75: the ray is not parallel to the plane.  Not on the render path.
  bool can_hit(const Ray &ray) const { return ray.direction().dot(plane_normal_) != 0.0f; }

  // the 16 floats the GPU consumes, in the stream order of mrt_scene_desc.tri
  void precomputed(float out[16]) const
  {
    const internal::LineDiscriminator *d[3] = {&disc_ab_to_c_, &disc_bc_to_a_, &disc_ca_to_b_};
    out[0] = plane_normal_.x; out[1] = plane_normal_.y; out[2] = plane_normal_.z; out[3] = plane_intercept_;
    for (int i = 0; i < 3; ++i) {
      out[4 + 4 * i + 0] = d[i]->normal().x;
      out[4 + 4 * i + 1] = d[i]->normal().y;
      out[4 + 4 * i + 2] = d[i]->normal().z;
      out[4 + 4 * i + 3] = d[i]->intercept();
    }
  }
  const Vec3f &a() const { return a_; }
  const Vec3f &b() const { return b_; }
  const Vec3f &c() const { return c_; }

private:
  Vec3f plane_normal_;
  float plane_intercept_;
  internal::LineDiscriminator disc_ab_to_c_, disc_bc_to_a_, disc_ca_to_b_;
  Vec3f a_, b_, c_;
};

} // namespace rt
