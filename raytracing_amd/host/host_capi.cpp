// host_capi.cpp -- C exports of the C++ host layer for the Python harness (tests, bench):
// the host-side precompute the reference also does on the host (Triangle and Viewport
// constructors, the default scene), so that Python can hand libmrt.so exactly what the
// C++ application would.  No rendering or intersection happens here.
#include <cstdint>
#include <cstring>

#include "pipeline/default_scene.hpp"
#include "pipeline/multisample_mesh.hpp"
#include "pipeline/render.hpp"

extern "C" {

// rt::Viewport(look_from, look_at, up, vfov, aspect) -> origin, horizontal, vertical, up_left
void rth_viewport(const float from[3], const float at[3], const float up[3], float vfov_deg, float aspect,
                  float out12[12])
{
  rt::Viewport v(rt::Vec3f(from[0], from[1], from[2]), rt::Vec3f(at[0], at[1], at[2]), rt::Vec3f(up[0], up[1], up[2]),
                 vfov_deg, aspect);
  mrt_camera c = v.to_abi();
  std::memcpy(out12 + 0, c.origin, 12);
  std::memcpy(out12 + 3, c.horizontal, 12);
  std::memcpy(out12 + 6, c.vertical, 12);
  std::memcpy(out12 + 9, c.up_left, 12);
}

// rt::Triangle constructor, batched: verts [n*9] (a,b,c) -> 16 streams of n floats
void rth_triangle_precompute(const float *verts, int64_t n, float *streams)
{
  for (int64_t i = 0; i < n; ++i) {
    const float *p = verts + 9 * i;
    rt::Triangle t(rt::Vec3f(p[0], p[1], p[2]), rt::Vec3f(p[3], p[4], p[5]), rt::Vec3f(p[6], p[7], p[8]));
    float q[16];
    t.precomputed(q);
    for (int s = 0; s < 16; ++s) streams[s * n + i] = q[s];
  }
}

// rt::SceneObject::can_hit over a sphere / a triangle (core/scene_object.cuh:29-33)
int rth_sphere_can_hit(const float c[3], float r, const float ro[3], const float rd[3])
{
  rt::SceneObject o(rt::Sphere(rt::Vec3f(c[0], c[1], c[2]), r), rt::diffuse_white());
  return o.can_hit(rt::Ray(rt::Vec3f(ro[0], ro[1], ro[2]), rt::Vec3f(rd[0], rd[1], rd[2]))) ? 1 : 0;
}
int rth_triangle_can_hit(const float v[9], const float ro[3], const float rd[3])
{
  rt::SceneObject o(rt::Triangle(rt::Vec3f(v[0], v[1], v[2]), rt::Vec3f(v[3], v[4], v[5]), rt::Vec3f(v[6], v[7], v[8])),
                    rt::diffuse_white());
  return o.can_hit(rt::Ray(rt::Vec3f(ro[0], ro[1], ro[2]), rt::Vec3f(rd[0], rd[1], rd[2]))) ? 1 : 0;
}

// The default scene with rt::Transform (rotation r [9, row-major], translation t [3]) applied to every object
// through SceneObject::apply_transform (core/transform.cuh:6-21, sphere.cuh:75-78, triangle.cuh:105-113), flattened
void *rth_default_scene_transformed_create(const float r[9], const float t[3])
{
  auto s = rt::config::make_default_scene();
  rt::Mat3f m;
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) m.m[i][j] = r[3 * i + j];
  const rt::Transform tf(m, rt::Vec3f(t[0], t[1], t[2]));
  for (std::size_t i = 0; i < s.objects.size(); ++i) s.objects[i].apply_transform(tf);
  return new rt::FlatScene(rt::flatten(rt::make_view(s.objects), rt::make_view(s.lights)));
}

// rt::DirectionalLight constructor (normalises): dir[3] -> unit dir
void rth_directional_light_dir(const float dir[3], float out[3])
{
  rt::DirectionalLight l(rt::Vec3f(dir[0], dir[1], dir[2]), rt::Vec3f(0, 0, 0));
  out[0] = l.dir().x; out[1] = l.dir().y; out[2] = l.dir().z;
}

// default scene (src/main.cu:183-237), flattened
void *rth_default_scene_create()
{
  auto s = rt::config::make_default_scene();
  return new rt::FlatScene(rt::flatten(rt::make_view(s.objects), rt::make_view(s.lights)));
}
void rth_flat_scene_destroy(void *h) { delete static_cast<rt::FlatScene *>(h); }
void rth_flat_scene_desc(void *h, mrt_scene_desc *out) { *out = static_cast<rt::FlatScene *>(h)->desc(); }

// rt::MultisampleMesh offsets: out [121*2]
void rth_msmesh_offsets(float norm_x, float norm_y, float *out)
{
  rt::MultisampleMesh m(norm_x, norm_y);
  auto s = m.generate_samples(rt::Vec2f(0.0f, 0.0f));
  for (std::size_t i = 0; i < s.size(); ++i) { out[2 * i] = s[i].x; out[2 * i + 1] = s[i].y; }
}

} // extern "C"
