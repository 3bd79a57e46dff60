// core/light.hpp -- rt::DirectionalLight / rt::PointLight (reference: include/core/light.cuh:36-86).
// ray_from, obstructed_by and light_color() run on the GPU.  SphericalLight / PanelLight /
// CylindricalLight / DomeLight are unfinished stubs in the reference (:88-132) and are not mirrored.
#pragma once
#include <variant>

#include "math/vec.hpp"

namespace rt {

class PointLight {
public:
  PointLight(Vec3f pos, Vec3f color) : pos_(pos), color_(color) {}
  Vec3f &pos() { return pos_; }
  const Vec3f &pos() const { return pos_; }
  const Vec3f &color() const { return color_; }

private:
  Vec3f pos_, color_;
};

class DirectionalLight {
public:
  DirectionalLight() = default;
  // the constructor normalises the direction (light.cuh:64-67)
  DirectionalLight(Vec3f dir, Vec3f color) : dir_(dir.normalized()), color_(color) {}
  Vec3f &dir() { return dir_; }
  const Vec3f &dir() const { return dir_; }
  const Vec3f &color() const { return color_; }

private:
  Vec3f dir_, color_;
};

// the reference's light_color() is typed on DirectionalLight only (light.cuh:139); the
// render path here takes either kind, in list order
using Light = std::variant<DirectionalLight, PointLight>;

} // namespace rt
