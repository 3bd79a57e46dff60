// test/host_cpu_tests.cpp -- the parts of the C++ host layer that need no GPU: Transform / apply_transform
// (reference: core/transform.cuh, core/sphere.cuh:75-78, core/triangle.cuh:105-113), Interval modes
// (core/interval.cuh:25-42), make_work_partition (cuda_utils/work_partition.cuh:12-23), ImageBuffer ownership and
// flip (container/alloc.cuh:39-59, src/main.cu:249-255), scene flattening.  Run by tests/test_host_layer.py.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <utility>

#include "core/interval.hpp"
#include "core/scene_object.hpp"
#include "pipeline/default_scene.hpp"
#include "pipeline/render.hpp"

static int g_failed = 0, g_checks = 0;
#define EXPECT_TRUE(x) do { ++g_checks; if (!(x)) { ++g_failed; std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); } } while (0)
#define EXPECT_EQ(a, b) do { ++g_checks; if (!((a) == (b))) { ++g_failed; std::printf("  FAILED %s:%d: %s == %s\n", __FILE__, __LINE__, #a, #b); } } while (0)

using rt::Vec3f;

static bool same16(const rt::Triangle &a, const rt::Triangle &b)
{
  float qa[16], qb[16];
  a.precomputed(qa); b.precomputed(qb);
  return std::memcmp(qa, qb, sizeof qa) == 0;
}

int main()
{
  // Transform: r * v + t, applied to the geometry of a scene object
  rt::Mat3f rot; // 90 degrees about y: (x, y, z) -> (z, y, -x)
  rot.m[0][0] = 0; rot.m[0][2] = 1; rot.m[2][0] = -1; rot.m[2][2] = 0;
  const rt::Transform tf(rot, Vec3f{1.0f, 2.0f, 3.0f});
  const Vec3f p = tf(Vec3f{1.0f, 0.0f, 0.0f});
  EXPECT_TRUE(p.x == 1.0f && p.y == 2.0f && p.z == 2.0f);

  rt::Sphere s(Vec3f{0.0f, 0.0f, -5.0f}, 1.5f);
  s.apply_transform(tf);
  EXPECT_TRUE(s.center().x == -4.0f && s.center().y == 2.0f && s.center().z == 3.0f && s.radius() == 1.5f);

  const Vec3f a{0.0f, 1.0f, -1.0f}, b{-1.0f, -1.0f, -1.5f}, c{1.0f, -1.0f, -2.0f};
  rt::Triangle t(a, b, c);
  t.apply_transform(tf); // the reference cannot do this: its a_, b_, c_ are never stored (core/triangle.cuh:58-66)
  EXPECT_TRUE(same16(t, rt::Triangle(tf(a), tf(b), tf(c))));
  EXPECT_TRUE(!same16(t, rt::Triangle(a, b, c)));

  rt::SceneObject obj(rt::Sphere(Vec3f{0, 0, -5}, 1.0f), rt::diffuse_red());
  obj.apply_transform(tf);
  EXPECT_TRUE(std::get<rt::Sphere>(obj.geometry()).center().x == -4.0f);

  // can_hit (sphere.cuh:26-30, triangle.cuh:68-75, plane.cuh:20-23, scene_object.cuh:29-33): the reference's
  // sphere test rays -- a head-on ray can hit, a grazing one (discriminant exactly 0) cannot, although hit() accepts it
  const rt::Sphere s5(Vec3f{0.0f, 0.0f, -5.0f}, 1.0f);
  EXPECT_TRUE(s5.can_hit(rt::Ray(Vec3f{0, 0, 0}, Vec3f{0, 0, -1})));
  EXPECT_TRUE(!s5.can_hit(rt::Ray(Vec3f{-1, 0, 0}, Vec3f{0, 0, -1})));
  EXPECT_TRUE(!s5.can_hit(rt::Ray(Vec3f{3, 0, 0}, Vec3f{0, 0, -1})));
  const rt::Triangle tz(Vec3f{0, 1, -1}, Vec3f{-1, -1, -1}, Vec3f{1, -1, -1});
  EXPECT_TRUE(tz.can_hit(rt::Ray(Vec3f{0, 0, 0}, Vec3f{0, 0, -1})) && !tz.can_hit(rt::Ray(Vec3f{0, 0, 0}, Vec3f{1, 0, 0})));
  EXPECT_TRUE(!rt::Plane(0.0f, Vec3f{0, 1, 0}).can_hit(rt::Ray(Vec3f{0, 1, 0}, Vec3f{0, -1, 0})));
  EXPECT_TRUE(rt::SceneObject(tz, rt::diffuse_red()).can_hit(rt::Ray(Vec3f{0, 0, 0}, Vec3f{0, 0.5f, -1})));
  EXPECT_TRUE(!rt::SceneObject(rt::Plane(0.0f, Vec3f{0, 1, 0}), rt::diffuse_red()).can_hit(rt::Ray(Vec3f{0, 1, 0}, Vec3f{0, -1, 0})));

  // Interval: four modes
  EXPECT_TRUE(rt::Interval(0.0f, 1.0f).contains(0.0f) && rt::Interval(0.0f, 1.0f).contains(1.0f));
  EXPECT_TRUE(!rt::Interval(0.0f, 1.0f, false, true).contains(0.0f) && rt::Interval(0.0f, 1.0f, false, true).contains(1.0f));
  EXPECT_TRUE(rt::Interval(0.0f, 1.0f, true, false).contains(0.0f) && !rt::Interval(0.0f, 1.0f, true, false).contains(1.0f));
  EXPECT_TRUE(!rt::Interval(0.0f, 1.0f).contains(NAN) && rt::Interval(0.01f, INFINITY).contains(INFINITY));
  EXPECT_TRUE(rt::Interval(0.0f, 1.0f).is_closed() && !rt::Interval(0.0f, 1.0f, false, false).is_closed());

  // make_work_partition
  auto wp = rt::make_work_partition(2048u * 1024u, 128u);
  EXPECT_TRUE(!wp.is_uneven_partition && wp.elements_per_thread == 16384u);
  wp = rt::make_work_partition(1000u, 128u);
  EXPECT_TRUE(wp.is_uneven_partition && wp.elements_per_thread == 8u && wp.elements_in_last_thread == 104u);

  // ImageBuffer: move-only owner; the flip exchanges rows r and height-1-r for all three buffers
  rt::ImageBuffer img(4, 3);
  for (std::size_t i = 0; i < 12; ++i) { img.buffer()[i] = Vec3f{float(i), 0, 0}; img.ids()[i] = int(i); img.ts()[i] = float(i); }
  rt::ImageBuffer moved(std::move(img));
  EXPECT_TRUE(img.buffer() == nullptr && moved.width() == 4 && moved.height() == 3);
  moved.flip_vertical();
  EXPECT_TRUE(moved.buffer()[0].x == 8.0f && moved.ids()[0] == 8 && moved.ts()[11] == 3.0f && moved.ids()[5] == 5);
  rt::ImageBuffer other(2, 2);
  other = std::move(moved);
  EXPECT_TRUE(other.width() == 4 && moved.buffer() == nullptr);

  // flattening of the default scene: object numbers, type partition, materials, light
  auto scene = rt::config::make_default_scene();
  rt::FlatScene f = rt::flatten(rt::make_view(scene.objects), rt::make_view(scene.lights));
  EXPECT_EQ(f.mat_kind.size(), 10u);
  EXPECT_EQ(f.sphere_obj.size(), 4u);
  EXPECT_EQ(f.tri_obj.size(), 6u);
  EXPECT_TRUE(f.sphere_obj[0] == 0 && f.sphere_obj[3] == 9 && f.tri_obj[0] == 1 && f.tri_obj[5] == 8);
  EXPECT_TRUE(f.mat_kind[5] == MRT_MAT_METALLIC && f.mat_kind[7] == MRT_MAT_DIFFUSE && f.mat_rgb[0] == 1.0f / 16.0f);
  EXPECT_EQ(f.light_kind.size(), 1u);

  std::printf("%d checks, %d failed\n", g_checks, g_failed);
  return g_failed ? 1 : 0;
}
