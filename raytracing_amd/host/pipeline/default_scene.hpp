// pipeline/default_scene.hpp -- the scene, light and camera hard-coded in the reference's
// main() (src/main.cu:34-41, :183-237), as a reusable host function.
#pragma once
#include "container/vector.hpp"
#include "core/light.hpp"
#include "core/scene_object.hpp"
#include "pipeline/viewport.hpp"

namespace rt {
namespace config {

// src/main.cu:34-41 (aspect 2.0 there; pass W/H for other frame shapes)
inline Viewport make_viewport(float aspect = 2.0f)
{
  return Viewport(Vec3f(0.0f, 4.0f, 6.0f), Vec3f(0.0f, 0.0f, -20.0f), Vec3f(0.0f, 1.0f, 0.0f), 20.0f, aspect);
}

struct Scene {
  Vector<SceneObject> objects;
  Vector<Light> lights;
};

inline Scene make_default_scene()
{
  auto line_start = Vec3f(-2.25f, -1.0f, -10.f);
  auto line_dir = Vec3f(2.25f, 1.0f, -3.0f);
  auto line_up = Vec3f(0.0f, 1.0f, 0.0f);
  auto line_dir_normed = line_dir.normalized();
  line_up = (line_up - line_up.dot(line_dir_normed) * line_dir_normed).normalized();
  Vec3f line_horizontal = line_up.cross(line_dir_normed);

  // the reference writes 0.6 (a double literal); Eigen converts it to float first
  Vec3f t1_center = line_start + 0.6f * line_dir;
  Vec3f t1_a = t1_center + 0.5f * line_up;
  Vec3f t1_b = t1_center + 0.5f * line_horizontal - 0.5f / 1.5f * line_up;
  Vec3f t1_c = t1_center - 0.5f * line_horizontal - 0.5f / 1.5f * line_up;

  Vec3f t2_center = line_start + 1.1f * line_dir;
  Vec3f t2_a = t2_center + 0.75f * line_up;
  Vec3f t2_b = t2_center + 0.75f * line_horizontal - 0.75f / 1.5f * line_up;
  Vec3f t2_c = t2_center - 0.75f * line_horizontal - 0.75f / 1.5f * line_up;

  Vec3f mirror_ul{-4.5f, 2.0f, -21.0f}, mirror_dl{-4.5f, -2.0f, -21.0f};
  Vec3f mirror_ur{4.5f, 2.0f, -21.0f}, mirror_dr{4.5f, -2.0f, -21.0f};
  Vec3f frame_ul{-4.6f, 2.1f, -21.1f}, frame_dl{-4.6f, -2.1f, -21.1f};
  Vec3f frame_ur{4.6f, 2.1f, -21.1f}, frame_dr{4.6f, -2.1f, -21.1f};

  Scene s;
  s.objects.push_back(SceneObject(Sphere(line_start + 0.0f * line_dir, 0.25f), diffuse_red()));
  s.objects.push_back(SceneObject(Triangle(t1_a, t1_b, t1_c), diffuse_yellow()));
  s.objects.push_back(SceneObject(Sphere(line_start + 0.8f * line_dir, 0.5f), diffuse_green()));
  s.objects.push_back(SceneObject(Triangle(t2_a, t2_b, t2_c), diffuse_yellow()));
  s.objects.push_back(SceneObject(Sphere(line_start + 2.0f * line_dir, 1.0f), diffuse_blue()));
  s.objects.push_back(SceneObject(Triangle(mirror_ul, mirror_ur, mirror_dl), metallic_white()));
  s.objects.push_back(SceneObject(Triangle(mirror_ur, mirror_dr, mirror_dl), metallic_white()));
  s.objects.push_back(SceneObject(Triangle(frame_ul, frame_ur, frame_dl), diffuse_black()));
  s.objects.push_back(SceneObject(Triangle(frame_ur, frame_dr, frame_dl), diffuse_black()));
  s.objects.push_back(SceneObject(Sphere(Vec3f(-4.5f, 0.0f, -13.0f), 1.0f), metallic_white()));

  s.lights.push_back(Light(DirectionalLight{line_dir, Vec3f(16.0f, 16.0f, 16.0f)}));
  return s;
}

} // namespace config
} // namespace rt
