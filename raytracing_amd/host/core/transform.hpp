// core/transform.hpp -- rt::Transform (reference: include/core/transform.cuh:6-21)
#pragma once
#include "math/vec.hpp"

namespace rt {

class Transform {
public:
  Transform(Mat3f r, Vec3f t) : r_(r), t_(t) {}
  Vec3f operator()(const Vec3f &v) const { return r_ * v + t_; }

private:
  Mat3f r_;
  Vec3f t_;
};

} // namespace rt
