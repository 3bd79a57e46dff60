// mrt_cull_common.h -- bound tests shared by the culling kernels (mrt_cull.hip, mrt_cull2.hip).
// None of this arithmetic has to be exact: it only decides which primitives get the exact test, and every
// bound is inflated far beyond what fp32 rounding of the exact tests can move (margins below).
#pragma once
#include "mrt_device.h"

namespace mrt {

// ---- margins ----------------------------------------------------------------------------------------
// Exact sphere test: disc = b*b - a*c >= 0 in fp32.  With L = |o - c| the rounding error of disc is below
// 10 eps a L^2 = 6e-7 a L^2, so disc >= 0 implies (true distance ray-centre)^2 <= r^2 + 6e-7 L^2.
// The cull radius uses 4e-6 L^2 (6x that), then 0.1 % and 1e-6 on top for the cull arithmetic itself.
#define CULL_DISC_SLACK 4e-6f
#define CULL_REL 1.001f
#define CULL_ABS 1e-6f
// cancellation in |w|^2 - proj^2 and the ~1e-6 relative error of the approximate (v_rsq/v_sqrt/v_rcp)
// arithmetic the bound tests use: lower bound of the perpendicular distance.  The bound tests never
// need exact arithmetic -- only the exact tests behind them do.
#define CULL_PERP_SLACK 1e-5f

__device__ __forceinline__ float wave_max(float x)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off, 64));
  return x;
}
__device__ __forceinline__ float wave_min(float x)
{
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) x = fminf(x, __shfl_xor(x, off, 64));
  return x;
}
// single-instruction approximations (1 ulp), for bound tests only
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ void fast_normalize(float &x, float &y, float &z)
{
  const float s = fast_rsq(dot3(x, y, z, x, y, z));
  x *= s; y *= s; z *= s;
}
__device__ __forceinline__ float bcast(float x, int lane)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}

// effective radius of a sphere of squared radius rr seen from distance <= Lmax
__device__ __forceinline__ float sphere_cull_radius(float rr, float Lmax2)
{
  return fast_sqrt(rr + CULL_DISC_SLACK * Lmax2) * CULL_REL + CULL_ABS;
}

// Cone {x : angle(x - apex, axis) <= theta}: true when the ball (w = centre - apex, radius reff) may
// touch it.  dist(ball centre, cone) >= perp*cos - proj*sin (distance to the supporting half-space).
__device__ __forceinline__ bool ball_may_touch_cone(float wx, float wy, float wz, float reff, float ax, float ay,
                                                    float az, float cosT, float sinT)
{
  const float L2 = dot3(wx, wy, wz, wx, wy, wz);
  const float proj = dot3(wx, wy, wz, ax, ay, az);
  const float perp = fast_sqrt(fmaxf(L2 - proj * proj - CULL_PERP_SLACK * L2, 0.0f));
  return !(perp * cosT - proj * sinT > reff);
}

// Ray bundle {m + e + t*D : |e| <= rho, t >= 0} (directional light), D ~ unit: true when the ball may touch it
__device__ __forceinline__ bool ball_may_touch_sweep(float wx, float wy, float wz, float reach, float Dx, float Dy,
                                                     float Dz, float DD)
{
  const float L2 = dot3(wx, wy, wz, wx, wy, wz);
  const float proj = dot3(wx, wy, wz, Dx, Dy, Dz);
  // squared distance to the half-line; behind the origin the distance is |w|
  const float d2 = (proj > 0.0f) ? (L2 - (proj * proj) * fast_rcp(DD)) : L2;
  const float dist = fast_sqrt(fmaxf(d2 - CULL_PERP_SLACK * L2, 0.0f));
  return !(dist > reach);
}

// Segment bundle: every shadow ray of the tile runs from a point within rho of m to the light position
// m + e (point light), so it stays within rho of the segment [m, m+e].
__device__ __forceinline__ bool ball_may_touch_segment(float wx, float wy, float wz, float reach, float ex, float ey,
                                                       float ez, float ee)
{
  const float L2 = dot3(wx, wy, wz, wx, wy, wz);
  float s = dot3(wx, wy, wz, ex, ey, ez) * fast_rcp(ee); // ee == 0 gives NaN/inf: clamped below, NaN -> pass
  s = fminf(fmaxf(s, 0.0f), 1.0f);
  const float qx = wx - s * ex, qy = wy - s * ey, qz = wz - s * ez;
  const float d2 = dot3(qx, qy, qz, qx, qy, qz);
  const float dist = fast_sqrt(fmaxf(d2 - CULL_PERP_SLACK * (L2 + ee), 0.0f));
  return !(dist > reach);
}

struct Light {
  int kind;
  float vx, vy, vz;
};

// Shadow-ray bundle of one light over one tile, and the per-lane ray.
struct Bundle {
  // tile-uniform
  float mx, my, mz, rho;   // hit points of the tile lie within rho of m
  float Dx, Dy, Dz, DD;    // directional: ray direction and D.D;  point: e = pos - m and e.e
  bool point;
};

// Does the ball (centre c, squared radius rr for spheres / radius rb for triangle bounds) possibly touch
// one of the tile's shadow rays?
__device__ __forceinline__ bool bundle_may_touch(const Bundle &B, float cx, float cy, float cz, float reff)
{
  const float wx = cx - B.mx, wy = cy - B.my, wz = cz - B.mz;
  const float reach = (reff + B.rho) * CULL_REL + CULL_ABS;
  return B.point ? ball_may_touch_segment(wx, wy, wz, reach, B.Dx, B.Dy, B.Dz, B.DD)
                 : ball_may_touch_sweep(wx, wy, wz, reach, B.Dx, B.Dy, B.Dz, B.DD);
}

} // namespace mrt
