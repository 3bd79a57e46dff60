// mrt_device.h -- device-side arithmetic helpers shared by the render kernels.
// One evaluation order for everything (DESIGN.md "Arithmetic contract"); build with -ffp-contract=off.
#pragma once
#include <hip/hip_runtime.h>

namespace mrt {

#define MRT_INF __builtin_inff()

__device__ __forceinline__ float dot3(float ax, float ay, float az, float bx, float by, float bz)
{
  return ax * bx + (ay * by + az * bz);
}

// core/interval.cuh:28-29
__device__ __forceinline__ bool contains(float lo, float hi, float f) { return (lo <= f) && (f <= hi); }

// v.normalized(): z = v.v; z > 0 ? v / sqrt(z) : v
__device__ __forceinline__ void normalize3(float &x, float &y, float &z)
{
  float n2 = dot3(x, y, z, x, y, z);
  if (n2 > 0.0f) {
    float n = sqrtf(n2);
    x = x / n; y = y / n; z = z / n;
  }
}

// Exact tail of Sphere::hit (core/sphere.cuh:37-69) once a, b and the discriminant are known.
__device__ __forceinline__ bool sphere_roots(float a, float b, float disc, float lo, float hi, float &t_out)
{
  if (disc > 0.0f) {
    float sq = sqrtf(disc);
    float t = (-b - sq) / a;
    if (contains(lo, hi, t)) { t_out = t; return true; }
    t = (-b + sq) / a;
    if (contains(lo, hi, t)) { t_out = t; return true; }
  } else if (disc == 0.0f) {
    float t = (-b) / a;
    if (contains(lo, hi, t)) { t_out = t; return true; }
  }
  return false;
}

} // namespace mrt
