// pipeline/viewport.hpp -- rt::Viewport, the look-at camera
// (reference: include/pipeline/viewport.cuh:6-48).  The constructor runs on the host as in
// the reference; compute_ray runs on the GPU from the four vectors kept here.
#pragma once
#include <cmath>

#include "core/ray.hpp"
#include "mrt.h"

namespace rt {

class Viewport {
public:
  Viewport(const Vec3f &look_from, const Vec3f &look_at, const Vec3f &up, float vertical_fov, float aspect)
    : origin_(look_from)
  {
    // float * double / float -> double, stored to float (viewport.cuh:19); tan on a float argument
    float theta = static_cast<float>(vertical_fov * M_PI / 180.0f);
    float half_height = std::tan(theta / 2);
    float half_width = aspect * half_height;
    Vec3f w = (look_from - look_at).normalized();
    Vec3f u = up.cross(w).normalized();
    Vec3f v = w.cross(u);
    up_left_ = origin_ - half_width * u - half_height * v - w;
    horizontal_dir_ = 2.0f * half_width * u;
    vertical_dir_ = 2.0f * half_height * v;
  }

  // host mirror of the device function, for tooling; same expression order (viewport.cuh:32)
  Ray compute_ray(float h_percentage, float v_percentage) const
  {
    Vec3f dir = up_left_ + h_percentage * horizontal_dir_ + v_percentage * vertical_dir_ - origin_;
    return Ray(origin_, dir);
  }

  mrt_camera to_abi() const
  {
    mrt_camera c;
    for (int i = 0; i < 3; ++i) {
      c.origin[i] = origin_[i];
      c.horizontal[i] = horizontal_dir_[i];
      c.vertical[i] = vertical_dir_[i];
      c.up_left[i] = up_left_[i];
    }
    return c;
  }

private:
  Vec3f origin_, horizontal_dir_, vertical_dir_, up_left_;
};

} // namespace rt
