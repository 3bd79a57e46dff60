// pipeline/intersect.hpp -- host entry to the GPU hit tests: what the reference's test
// kernels hit_sphere / hit_triangle <<<1,1>>> do (test/spheretest.cu:6-12,
// test/triangletest.cu:6-12), batched over rays.
#pragma once
#include <cstdint>
#include <optional>
#include <vector>

#include "core/hit_result.hpp"
#include "core/interval.hpp"
#include "core/plane.hpp"
#include "core/ray.hpp"
#include "core/sphere.hpp"
#include "core/triangle.hpp"
#include "hip_utils/check_error.hpp"
#include "mrt.h"

namespace rt {

namespace internal {
inline std::vector<std::optional<HitResult>> intersect(mrt_ctx *ctx, int kind, const float *prim,
                                                       const std::vector<Ray> &rays, const Interval &I)
{
  const std::size_t n = rays.size();
  std::vector<float> r(6 * n), t(n), point(3 * n), normal(3 * n);
  std::vector<std::uint8_t> hit(n);
  for (std::size_t i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) { r[6 * i + k] = rays[i].origin()[k]; r[6 * i + 3 + k] = rays[i].direction()[k]; }
  checkRtErrors(mrt_intersect_interval(ctx, kind, prim, r.data(), I.lower(), I.upper(), I.lower_closed() ? 1 : 0,
                                       I.upper_closed() ? 1 : 0, static_cast<std::int64_t>(n), hit.data(), t.data(),
                                       point.data(), normal.data()));
  std::vector<std::optional<HitResult>> out(n);
  for (std::size_t i = 0; i < n; ++i)
    if (hit[i])
      out[i] = HitResult{t[i], Vec3f{point[3 * i], point[3 * i + 1], point[3 * i + 2]},
                         Vec3f{normal[3 * i], normal[3 * i + 1], normal[3 * i + 2]}};
  return out;
}
} // namespace internal

inline std::vector<std::optional<HitResult>> hit(mrt_ctx *ctx, const Sphere &s, const std::vector<Ray> &rays,
                                                 const Interval &I)
{
  const float prim[4] = {s.center().x, s.center().y, s.center().z, s.radius()};
  return internal::intersect(ctx, MRT_GEOM_SPHERE, prim, rays, I);
}

inline std::vector<std::optional<HitResult>> hit(mrt_ctx *ctx, const Triangle &t, const std::vector<Ray> &rays,
                                                 const Interval &I)
{
  float prim[16];
  t.precomputed(prim);
  return internal::intersect(ctx, MRT_GEOM_TRIANGLE, prim, rays, I);
}

inline std::vector<std::optional<HitResult>> hit(mrt_ctx *ctx, const Plane &p, const std::vector<Ray> &rays,
                                                 const Interval &I)
{
  const float prim[4] = {p.normal().x, p.normal().y, p.normal().z, p.intercept()};
  return internal::intersect(ctx, MRT_GEOM_PLANE, prim, rays, I);
}

} // namespace rt
