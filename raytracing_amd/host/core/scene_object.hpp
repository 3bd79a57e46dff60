// core/scene_object.hpp -- rt::SceneObject (reference: include/core/scene_object.cuh:12-94).
// std::variant replaces the reference's device-safe Variant (container/variant.cuh);
// index order Sphere=0, Plane=1, Triangle=2 / Diffuse=0, Metallic=1 is kept because the
// C-ABI's kind codes are defined on it.
#pragma once
#include <utility>
#include <variant>

#include "core/material.hpp"
#include "core/plane.hpp"
#include "core/sphere.hpp"
#include "core/transform.hpp"
#include "core/triangle.hpp"

namespace rt {

using GeometryVariant = std::variant<Sphere, Plane, Triangle>;
using MaterialVariant = std::variant<DiffuseMaterial, MetallicMaterial>;

class SceneObject {
public:
  SceneObject(GeometryVariant geometry, MaterialVariant material)
    : geometry_(std::move(geometry)), material_(std::move(material))
  {}

  const GeometryVariant &geometry() const { return geometry_; }
  const MaterialVariant &material() const { return material_; }

  // scene_object.cuh:29-33
  bool can_hit(const Ray &r) const
  {
    return std::visit([&r](const auto &g) { return g.can_hit(r); }, geometry_);
  }

  void apply_transform(const Transform &t)
  {
    std::visit([&t](auto &g) { g.apply_transform(t); }, geometry_);
  }

private:
  GeometryVariant geometry_;
  MaterialVariant material_;
};

} // namespace rt
