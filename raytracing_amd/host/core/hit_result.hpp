// core/hit_result.hpp -- rt::HitResult (reference: include/core/hit_result.cuh:6-10)
#pragma once
#include "math/vec.hpp"

namespace rt {

struct HitResult {
  float t;
  Vec3f point;
  Vec3f normal;
};

} // namespace rt
