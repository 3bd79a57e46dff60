// core/material.hpp -- rt::DiffuseMaterial / rt::MetallicMaterial and the colour
// factories (reference: include/core/material.cuh:42-196).  Host-side description:
// light_bounce and the colour function run on the GPU.
#pragma once
#include "math/vec.hpp"

namespace rt {

class DiffuseMaterial {
public:
  explicit DiffuseMaterial(Vec3f color) : color_(color) {}
  const Vec3f &color() const { return color_; }

private:
  Vec3f color_;
};

class MetallicMaterial {
public:
  explicit MetallicMaterial(Vec3f color) : color_(color) {}
  const Vec3f &color() const { return color_; }

private:
  Vec3f color_;
};

// diffuse colours are scaled by 1/16 (material.cuh:81-114)
inline DiffuseMaterial diffuse_red() { return DiffuseMaterial{Vec3f{1.0f / 16.0f, 0.0f, 0.0f}}; }
inline DiffuseMaterial diffuse_green() { return DiffuseMaterial{Vec3f{0.0f, 1.0f / 16.0f, 0.0f}}; }
inline DiffuseMaterial diffuse_blue() { return DiffuseMaterial{Vec3f{0.0f, 0.0f, 1.0f / 16.0f}}; }
inline DiffuseMaterial diffuse_yellow() { return DiffuseMaterial{Vec3f{1.0f / 16.0f, 1.0f / 16.0f, 0.0f}}; }
inline DiffuseMaterial diffuse_purple() { return DiffuseMaterial{Vec3f{1.0f / 16.0f, 0.0f, 1.0f / 16.0f}}; }
inline DiffuseMaterial diffuse_white() { return DiffuseMaterial{Vec3f{1.0f / 16.0f, 1.0f / 16.0f, 1.0f / 16.0f}}; }
inline DiffuseMaterial diffuse_black() { return DiffuseMaterial{Vec3f{0.001f, 0.001f, 0.001f}}; }

// metallic colours are full scale (material.cuh:163-196)
inline MetallicMaterial metallic_red() { return MetallicMaterial{Vec3f{1.0f, 0.0f, 0.0f}}; }
inline MetallicMaterial metallic_green() { return MetallicMaterial{Vec3f{0.0f, 1.0f, 0.0f}}; }
inline MetallicMaterial metallic_blue() { return MetallicMaterial{Vec3f{0.0f, 0.0f, 1.0f}}; }
inline MetallicMaterial metallic_yellow() { return MetallicMaterial{Vec3f{1.0f, 1.0f, 0.0f}}; }
inline MetallicMaterial metallic_purple() { return MetallicMaterial{Vec3f{1.0f, 0.0f, 1.0f}}; }
inline MetallicMaterial metallic_white() { return MetallicMaterial{Vec3f{1.0f, 1.0f, 1.0f}}; }
inline MetallicMaterial metallic_black() { return MetallicMaterial{Vec3f{0.001f, 0.001f, 0.001f}}; }

} // namespace rt
