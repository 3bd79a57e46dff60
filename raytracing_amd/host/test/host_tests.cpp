// test/host_tests.cpp -- the reference's unit tests against this API, on the GPU, plus the full-frame anchors.
//
// The reference's tests (test/spheretest.cu, test/triangletest.cu; test/raytest.cu repeats the triangle file)
// construct a primitive and a ray on the host, run Sphere::hit / Triangle::hit in a <<<1,1>>> kernel with the
// interval [-1, 100] and compare t and the normal with EXPECT_NEAR(..., 1e-6).  Here the same cases (names and
// numbers: tests/golden/reference_known_answers.json cites each one) go through rt::hit(), i.e. mrt_intersect.
// The last test renders the default scene through rt::Renderer and checks the object-ID histogram and ray counts
// of tests/golden/survey_anchors.json (640x480).  No test framework in this image: a few macros below.
#include <cmath>
#include <cstdio>
#include <vector>

#include "pipeline/default_scene.hpp"
#include "pipeline/intersect.hpp"
#include "pipeline/render.hpp"

static int g_failed = 0, g_checks = 0;
#define EXPECT_TRUE(x) do { ++g_checks; if (!(x)) { ++g_failed; std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #x); } } while (0)
#define EXPECT_NEAR(a, b, eps) do { ++g_checks; const double a_ = (a), b_ = (b); if (!(std::fabs(a_ - b_) <= (eps))) { ++g_failed; \
  std::printf("  FAILED %s:%d: %s = %.9g, expected %.9g\n", __FILE__, __LINE__, #a, a_, b_); } } while (0)
#define EXPECT_EQ(a, b) do { ++g_checks; const long long a_ = (long long)(a), b_ = (long long)(b); if (a_ != b_) { ++g_failed; \
  std::printf("  FAILED %s:%d: %s = %lld, expected %lld\n", __FILE__, __LINE__, #a, a_, b_); } } while (0)

static constexpr float EPS = 1.e-6f;
using rt::Vec3f;

struct SphereCase { const char *name; Vec3f center; float radius; Vec3f origin, dir; float t; bool normal_from_hit; Vec3f normal; };
struct TriangleCase { const char *name; Vec3f a, b, c, origin, dir; float t; Vec3f normal; };

static void test_sphere(rt::Renderer &gpu)
{
  const SphereCase cases[] = { // test/spheretest.cu:16-125
    {"BasicIntersectZAlignedSphereNear", {0, 0, -2}, 1, {0, 0, 0}, {0, 0, -1}, 1, false, {0, 0, 1}},
    {"BasicIntersectZAlignedSphereFar", {0, 0, -5}, 1, {0, 0, 0}, {0, 0, -1}, 4, false, {0, 0, 1}},
    {"BasicIntersectZAlignedSphereLeft", {0, 0, -5}, 1, {-1, 0, 0}, {0, 0, -1}, 5, false, {-1, 0, 0}},
    {"BasicIntersectZAlignedSphereRight", {0, 0, -5}, 1, {1, 0, 0}, {0, 0, -1}, 5, false, {1, 0, 0}},
    {"BasicIntersectZAlignedSphereUp", {0, 0, -5}, 1, {0, 1, 0}, {0, 0, -1}, 5, false, {0, 1, 0}},
    {"BasicIntersectZAlignedSphereDown", {0, 0, -5}, 1, {0, -1, 0}, {0, 0, -1}, 5, false, {0, -1, 0}},
    {"BasicIntersectNonZAlignedSphere", {1, 1, -5}, 1, {0, 0, 0}, {1, 1, -4}, 1, true, {0, 0, 0}},
  };
  for (const SphereCase &c : cases) {
    std::printf("[ RUN      ] TestSphere.%s\n", c.name);
    rt::Sphere s(c.center, c.radius);
    rt::Ray r(c.origin, c.dir);
    auto hit = rt::hit(gpu.ctx(), s, {r}, rt::Interval(-1.0f, 100.0f))[0];
    EXPECT_TRUE(hit.has_value());
    if (!hit) continue;
    EXPECT_NEAR(hit->t, c.t, EPS);
    Vec3f n = c.normal;
    if (c.normal_from_hit) n = (r.point_at_param(hit->t) - c.center).normalized(); // as the reference test computes it
    EXPECT_NEAR(hit->normal.x, n.x, EPS);
    EXPECT_NEAR(hit->normal.y, n.y, EPS);
    EXPECT_NEAR(hit->normal.z, n.z, EPS);
  }
}

static void test_triangle(rt::Renderer &gpu)
{
  const float q = 1.0f / std::sqrt(2.0f);
  const TriangleCase cases[] = { // test/triangletest.cu:17-90
    {"BasicIntersectZAlignedUpTriangle", {0, 1, -1}, {-1, -1, -1}, {1, -1, -1}, {0, 0, 0}, {0, 0, -1}, 1, {0, 0, 1}},
    {"BasicIntersectZAlignedDownTriangle", {0, -1, -1}, {1, 1, -1}, {-1, 1, -1}, {0, 0, 0}, {0, 0, -1}, 1, {0, 0, 1}},
    {"BasicIntersectAtRayOrigin", {0, -1, 0}, {1, 1, -1}, {-1, 1, 1}, {0, 0, 0}, {0, 0, -1}, 0, {q, 0, q}},
    {"BasicIntersectNonZAligned", {0, -1, -3}, {1, 1, -4}, {-1, 1, -2}, {0, 0.5f, 0}, {0, 0, -1}, 3, {q, 0, q}},
  };
  for (const TriangleCase &c : cases) {
    std::printf("[ RUN      ] TestTriangle.%s\n", c.name);
    rt::Triangle t(c.a, c.b, c.c);
    rt::Ray r(c.origin, c.dir);
    auto hit = rt::hit(gpu.ctx(), t, {r}, rt::Interval(-1.0f, 100.0f))[0];
    EXPECT_TRUE(hit.has_value());
    if (!hit) continue;
    EXPECT_NEAR(hit->t, c.t, EPS);
    EXPECT_NEAR(hit->normal.x, c.normal.x, EPS);
    EXPECT_NEAR(hit->normal.y, c.normal.y, EPS);
    EXPECT_NEAR(hit->normal.z, c.normal.z, EPS);
  }
  std::printf("[ RUN      ] TestTriangle.MissesOutsideAndBehind\n");
  rt::Triangle t(Vec3f{0, 1, -1}, Vec3f{-1, -1, -1}, Vec3f{1, -1, -1});
  auto hits = rt::hit(gpu.ctx(), t, {rt::Ray(Vec3f{3, 0, 0}, Vec3f{0, 0, -1}), rt::Ray(Vec3f{0, 0, 0}, Vec3f{0, 0, 1})},
                      rt::Interval(0.01f, 100.0f));
  EXPECT_TRUE(!hits[0].has_value());
  EXPECT_TRUE(!hits[1].has_value());
  std::printf("[ RUN      ] TestTriangle.IntervalModes\n"); // core/interval.cuh:14-42: t = 1 is a bound of the interval
  const rt::Ray down(Vec3f{0, 0, 0}, Vec3f{0, 0, -1});
  EXPECT_TRUE(rt::hit(gpu.ctx(), t, {down}, rt::Interval(1.0f, 100.0f))[0].has_value());
  EXPECT_TRUE(!rt::hit(gpu.ctx(), t, {down}, rt::Interval(1.0f, 100.0f, false, true))[0].has_value());
  EXPECT_TRUE(rt::hit(gpu.ctx(), t, {down}, rt::Interval(-1.0f, 1.0f))[0].has_value());
  EXPECT_TRUE(!rt::hit(gpu.ctx(), t, {down}, rt::Interval(-1.0f, 1.0f, true, false))[0].has_value());
  EXPECT_TRUE(rt::hit(gpu.ctx(), t, {down}, rt::Interval(-1.0f, 100.0f, false, false))[0].has_value());
}

static void test_default_scene_frame(rt::Renderer &gpu)
{
  std::printf("[ RUN      ] TestRender.DefaultSceneAnchors640x480\n");
  const std::size_t W = 640, H = 480;
  rt::ImageBuffer img(W, H);
  auto viewport = rt::config::make_viewport(2.0f); // the survey's anchors use the reference's aspect 2.0 at every size
  auto scene = rt::config::make_default_scene();
  gpu.upload(rt::make_view(scene.objects), rt::make_view(scene.lights));
  const mrt_stats st = gpu.render(viewport, img, 1, 0);
  const long long want[10] = {900, 1021, 2619, 1927, 7932, 29054, 19341, 2071, 1412, 10642}; // survey_anchors.json
  long long hist[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, miss = 0;
  for (std::size_t i = 0; i < W * H; ++i) {
    const int id = img.ids()[i];
    if (id < 0) ++miss; else if (id < 10) ++hist[id];
  }
  for (int k = 0; k < 10; ++k) EXPECT_EQ(hist[k], want[k]);
  EXPECT_EQ(miss, 230281);
  EXPECT_EQ(st.hit_pixels, 76919);
  EXPECT_EQ(st.shadow_rays, 73418);
  EXPECT_EQ(st.occluded, 7104);
  EXPECT_EQ(st.primary_rays + st.shadow_rays, 380618);

  std::printf("[ RUN      ] TestRender.BounceChainsLeaveThePrimaryHitAlone\n");
  rt::ImageBuffer deep(W, H);
  gpu.render(viewport, deep, 1, 8); // config::MAX_RECURSION_DEPTH = 8 (src/main.cu:26)
  long long diff_id = 0, diff_rgb = 0;
  for (std::size_t i = 0; i < W * H; ++i) {
    diff_id += deep.ids()[i] != img.ids()[i];
    diff_rgb += deep.buffer()[i].x != img.buffer()[i].x;
  }
  EXPECT_EQ(diff_id, 0);
  EXPECT_TRUE(diff_rgb > 1000);
}

// rt::MultiGpuRenderer over the GPUs this box has (one here): the same frame as rt::Renderer, byte for byte; and
// the ROCm installation's RCCL opened, a communicator made and destroyed, from a process without torch
static void test_multi_gpu_renderer(rt::Renderer &gpu)
{
  std::printf("[ RUN      ] TestRender.MultiGpuRendererMatchesRenderer\n");
  const std::size_t W = 320, H = 200;
  auto viewport = rt::config::make_viewport(float(W) / float(H));
  auto scene = rt::config::make_default_scene();
  rt::ImageBuffer one(W, H), many(W, H);
  gpu.upload(rt::make_view(scene.objects), rt::make_view(scene.lights));
  const mrt_stats s1 = gpu.render(viewport, one, 1, 0);
  int ngpu = 0;
  checkRtErrors(mrt_device_count(&ngpu));
  rt::MultiGpuRenderer group(ngpu, 16);
  group.upload(rt::make_view(scene.objects), rt::make_view(scene.lights));
  const mrt_stats sn = group.render(viewport, many, 1, 0);
  long long diff = 0;
  for (std::size_t i = 0; i < W * H; ++i)
    diff += (one.ids()[i] != many.ids()[i]) + (one.ts()[i] != many.ts()[i]) + (one.buffer()[i].x != many.buffer()[i].x) +
            (one.buffer()[i].y != many.buffer()[i].y) + (one.buffer()[i].z != many.buffer()[i].z);
  EXPECT_EQ(diff, 0);
  EXPECT_EQ(sn.shadow_rays, s1.shadow_rays);
  EXPECT_EQ(sn.hit_pixels, s1.hit_pixels);

  std::printf("[ RUN      ] TestRender.LocalCommunicatorOfOne\n");
  mrt_ctx *c = gpu.ctx();
  mrt_comm *comm = nullptr;
  EXPECT_EQ(mrt_comm_create_local(&c, 1, &comm), 0);
  EXPECT_TRUE(comm != nullptr);
  mrt_camera cam = viewport.to_abi();
  mrt_render_params p{};
  p.width = int(W); p.height = int(H); p.spp = 1; p.max_depth = 0;
  p.part = rt::make_strip_partition(0, 1, 8);
  rt::ImageBuffer again(W, H);
  EXPECT_EQ(mrt_render_multi(&c, &comm, 1, &cam, &p, &again.buffer()[0].x, again.ids(), again.ts(), nullptr), 0);
  long long d2 = 0;
  for (std::size_t i = 0; i < W * H; ++i) d2 += (one.ids()[i] != again.ids()[i]) + (one.buffer()[i].y != again.buffer()[i].y);
  EXPECT_EQ(d2, 0);
  EXPECT_EQ(mrt_comm_destroy(comm), 0);
}

int main()
{
  rt::Renderer gpu(0);
  test_sphere(gpu);
  test_triangle(gpu);
  test_default_scene_frame(gpu);
  test_multi_gpu_renderer(gpu);
  std::printf("%d checks, %d failed\n", g_checks, g_failed);
  return g_failed ? 1 : 0;
}
