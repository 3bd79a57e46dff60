// mrt_cull_common.h -- bound tests shared by the culling kernels (mrt_cull.hip, mrt_cull2.hip).
// None of this arithmetic has to be exact: it only decides which primitives get the exact test, and every
// bound is inflated far beyond what fp32 rounding of the exact tests can move (margins below).
#pragma once
#include "mrt_device.h"

namespace mrt {

// ---- margins ----------------------------------------------------------------------------------------
// Exact sphere test: disc = b*b - a*c >= 0 in fp32.  With L = |o - c| and u = 2^-24, standard forward error
// analysis of the reference's expression order gives |fl(disc) - disc| <= 16 u a L^2 = 9.5e-7 a L^2, so
// fl(disc) >= 0 implies (true distance ray-centre)^2 <= r^2 + 9.5e-7 L^2.  The cull radius uses 2e-6 L^2,
// then 0.1 % and 1e-6 on top for the arithmetic of the bound tests themselves.
#define CULL_DISC_SLACK 2e-6f
#define CULL_DISC_SLACK_SQRT 1.5e-3f /* >= sqrt(CULL_DISC_SLACK) */
#define CULL_REL 1.001f
#define CULL_ABS 1e-6f
// The bound tests use single-instruction approximations (v_rsq/v_sqrt/v_rcp, 1 ulp) and plain fp32; their
// distances are computed from explicit perpendicular vectors (no cancelling differences of squares), so
// they are off by at most a few 1e-7 of the lengths involved.  Every comparison gets 4e-6 of the 1-norm
// of those lengths as slack.  Only the exact tests behind the bounds need exact arithmetic.
#define CULL_LEN_SLACK 4e-6f

__device__ __forceinline__ float bcast(float x, int lane)
{
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}
// Reductions over the wave with DPP row operations (no LDS traffic); the result is wave-uniform.
#define MRT_DPP_STEP(OP, ctrl, row_mask)                                                                      \
  x = OP(x, __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(x), __float_as_int(x), ctrl, row_mask, \
                                                       0xf, false)))
#define MRT_DPP_REDUCE(OP)                                                                          \
  MRT_DPP_STEP(OP, 0xB1, 0xf);  /* quad_perm [1,0,3,2] */                                            \
  MRT_DPP_STEP(OP, 0x4E, 0xf);  /* quad_perm [2,3,0,1] */                                            \
  MRT_DPP_STEP(OP, 0x141, 0xf); /* row_half_mirror */                                                \
  MRT_DPP_STEP(OP, 0x140, 0xf); /* row_mirror: every lane of a row of 16 holds the row's result */   \
  MRT_DPP_STEP(OP, 0x142, 0xa); /* row_bcast:15 into rows 1 and 3 */                                 \
  MRT_DPP_STEP(OP, 0x143, 0xc); /* row_bcast:31 into rows 2 and 3: lane 63 holds the wave's result */
__device__ __forceinline__ float wave_max(float x)
{
  MRT_DPP_REDUCE(fmaxf)
  return bcast(x, 63);
}
__device__ __forceinline__ float wave_min(float x)
{
  MRT_DPP_REDUCE(fminf)
  return bcast(x, 63);
}
#undef MRT_DPP_REDUCE
#undef MRT_DPP_STEP
// single-instruction approximations (1 ulp), for bound tests only
__device__ __forceinline__ float fast_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float fast_rsq(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ void fast_normalize(float &x, float &y, float &z)
{
  const float s = fast_rsq(dot3(x, y, z, x, y, z));
  x *= s; y *= s; z *= s;
}

// effective radius of a sphere of squared radius rr seen from distance <= Lmax
__device__ __forceinline__ float sphere_cull_radius(float rr, float Lmax2)
{
  return __builtin_fmaf(fast_sqrt(__builtin_fmaf(CULL_DISC_SLACK, Lmax2, rr)), CULL_REL, CULL_ABS);
}

__device__ __forceinline__ float norm1(float x, float y, float z) { return fabsf(x) + fabsf(y) + fabsf(z); }

// Bound arithmetic may fuse (the build contracts nothing by itself: -ffp-contract=off is part of the exact tests'
// contract): a dot product as one multiply and two FMAs.  tools/ubench_mix.hip: a v_fma_f32 costs a SIMD 1.5 v_mul_f32,
// the multiply-add pair it replaces 2.  Only for bounds -- the exact tests keep dot3().
__device__ __forceinline__ float bdot3(float ax, float ay, float az, float bx, float by, float bz)
{
  return __builtin_fmaf(ax, bx, __builtin_fmaf(ay, by, az * bz));
}

// Cone {x : angle(x - apex, axis) <= theta}, |axis| = 1: true when the ball (w = centre - apex, radius reff)
// may touch it.  dist(ball centre, cone) >= perp*cos - proj*sin (distance to the supporting half-space).
__device__ __forceinline__ bool ball_may_touch_cone(float wx, float wy, float wz, float reff, float ax, float ay,
                                                    float az, float cosT, float sinT)
{
  const float proj = bdot3(wx, wy, wz, ax, ay, az);
  const float qx = __builtin_fmaf(-proj, ax, wx), qy = __builtin_fmaf(-proj, ay, wy), qz = __builtin_fmaf(-proj, az, wz);
  const float perp = fast_sqrt(bdot3(qx, qy, qz, qx, qy, qz));
  return !(__builtin_fmaf(perp, cosT, -(proj * sinT)) > __builtin_fmaf(CULL_LEN_SLACK, norm1(wx, wy, wz), reff));
}

// Ray bundle {m + e + t*D : |e| <= rho, t >= 0} (directional light): true when the ball at w = centre - m may
// come within `reach` of the half-line from m along D (DD = D.D)
__device__ __forceinline__ bool ball_may_touch_sweep(float wx, float wy, float wz, float reach, float Dx, float Dy,
                                                     float Dz, float DD)
{
  const float s = fmaxf(bdot3(wx, wy, wz, Dx, Dy, Dz) * fast_rcp(DD), 0.0f); // behind the origin: distance to m
  const float qx = __builtin_fmaf(-s, Dx, wx), qy = __builtin_fmaf(-s, Dy, wy), qz = __builtin_fmaf(-s, Dz, wz);
  const float dist = fast_sqrt(bdot3(qx, qy, qz, qx, qy, qz));
  return !(dist > __builtin_fmaf(CULL_LEN_SLACK, __builtin_fmaf(s, norm1(Dx, Dy, Dz), norm1(wx, wy, wz)), reach));
}

// Segment bundle: every shadow ray of the tile runs from a point within rho of m to the light position
// m + e (point light), so it stays within rho of the segment [m, m+e].
__device__ __forceinline__ bool ball_may_touch_segment(float wx, float wy, float wz, float reach, float ex, float ey,
                                                       float ez, float ee)
{
  float s = bdot3(wx, wy, wz, ex, ey, ez) * fast_rcp(ee); // ee == 0 gives NaN/inf: clamped below, NaN -> pass
  s = fminf(fmaxf(s, 0.0f), 1.0f);
  const float qx = __builtin_fmaf(-s, ex, wx), qy = __builtin_fmaf(-s, ey, wy), qz = __builtin_fmaf(-s, ez, wz);
  const float dist = fast_sqrt(bdot3(qx, qy, qz, qx, qy, qz));
  return !(dist > __builtin_fmaf(CULL_LEN_SLACK, norm1(wx, wy, wz) + norm1(ex, ey, ez), reach));
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
  // Optional slab |n.(x - m)| <= sw that holds every shadow ray of the tile: n is the unit normal of the plane
  // through m spanned by the tile's view axis and the light direction (resp. the direction to a point light).
  // The tile's hit points lie along the view axis, so they -- and with them their shadow rays, which move
  // inside that plane -- stay within a small sw of it even when they are spread over a large depth range
  // (where the ball (m, rho) is huge).  sw = +inf: no slab.
  float snx = 0.0f, sny = 0.0f, snz = 0.0f, sw = MRT_INF;
};

// Build the slab of B for the current light (B.D set): a = unit view axis of the tile, (ex, ey, ez) = this lane's
// hit point minus m, have = this lane has a hit point.  Wave-uniform result.
__device__ __forceinline__ void bundle_make_slab(Bundle &B, float ax, float ay, float az, float ex, float ey, float ez,
                                                 bool have)
{
  float nx = ay * B.Dz - az * B.Dy, ny = az * B.Dx - ax * B.Dz, nz = ax * B.Dy - ay * B.Dx;
  const float nn = bdot3(nx, ny, nz, nx, ny, nz);
  B.snx = 0.0f; B.sny = 0.0f; B.snz = 0.0f; B.sw = MRT_INF;
  // the angle between axis and direction must be at least ~3 degrees: below that the normal is ill-conditioned
  // (its rounding error is 6e-8 / sin) and the slab is not used
  if (nn > 2.5e-3f * B.DD) {
    const float inv = fast_rsq(nn);
    nx *= inv; ny *= inv; nz *= inv;
    const float w = wave_max(have ? fabsf(bdot3(nx, ny, nz, ex, ey, ez)) : 0.0f);
    B.snx = nx; B.sny = ny; B.snz = nz;
    // a point light's own offset from the plane is pure rounding: |n.(L - m)| <= 1.2e-6 |L - m|
    B.sw = w * CULL_REL + CULL_ABS + (B.point ? 1e-5f * norm1(B.Dx, B.Dy, B.Dz) : 0.0f);
  }
}

// Does the ball (centre c, squared radius rr for spheres / radius rb for triangle bounds) possibly touch
// one of the tile's shadow rays?
__device__ __forceinline__ bool bundle_may_touch(const Bundle &B, float cx, float cy, float cz, float reff)
{
  const float wx = cx - B.mx, wy = cy - B.my, wz = cz - B.mz;
  // slab: a shadow ray reaches at most |c - m| + reff along its direction before it can touch the ball, and drifts
  // off the (rounded) plane by at most 1.2e-6 of that
  if (fabsf(bdot3(B.snx, B.sny, B.snz, wx, wy, wz)) > __builtin_fmaf(1e-5f, norm1(wx, wy, wz) + reff, __builtin_fmaf(reff, CULL_REL, B.sw))) return false;
  const float reach = __builtin_fmaf(reff + B.rho, CULL_REL, CULL_ABS);
  return B.point ? ball_may_touch_segment(wx, wy, wz, reach, B.Dx, B.Dy, B.Dz, B.DD)
                 : ball_may_touch_sweep(wx, wy, wz, reach, B.Dx, B.Dy, B.Dz, B.DD);
}

// ---- primitive-side hierarchy walk ---------------------------------------------------------------------
// Primitives are stored in kd order; runs of 64 primitives have a bounding ball ("cluster"), runs of 64
// clusters another one ("super-cluster").  A wave walks the hierarchy with lane-parallel tests at every
// level: 64 super-clusters per step, then the 64 clusters of each surviving super-cluster, then
// body(first primitive of the cluster) for each surviving cluster.  A level with a single entry is not
// tested (tiny scenes: C2).  body returns false to stop the walk (shadow early-out).
// n: balls at this level (the streams' stride); limit <= n: only balls below it are visited
template <class Test>
__device__ __forceinline__ unsigned long long level_mask(const float *__restrict__ balls, int n, int begin, int lane,
                                                         Test test, int limit)
{
  if (n == 1) return limit > 0 ? 1ull : 0ull;
  const int i = begin + lane;
  bool pass = false;
  if (i < min(limit, begin + 64)) pass = test(balls[i], balls[n + i], balls[2 * n + i], balls[3 * n + i]);
  return __ballot(pass);
}

// nc_limit (optional, <= nc): the walk covers clusters 0 .. nc_limit - 1 only
template <bool HIER, class Test, class Body>
__device__ __forceinline__ void walk_hierarchy(const float *__restrict__ sclu, int nsc, const float *__restrict__ clu,
                                               int nc, int lane, Test test, Body body, int nc_limit = -1)
{
  if (!HIER) { // at most one cluster per primitive type (C2): nothing to walk
    if (nc > 0) body(0);
    return;
  }
  if (nc_limit < 0) nc_limit = nc;
  const int nsc_limit = (nc_limit + 63) >> 6;
  for (int sb = 0; sb < nsc_limit; sb += 64) {
    unsigned long long smask = level_mask(sclu, nsc, sb, lane, test, nsc_limit);
    while (smask) {
      const int cb = (sb + __builtin_ctzll(smask)) * 64;
      smask &= smask - 1;
      unsigned long long cmask = level_mask(clu, nc, cb, lane, test, nc_limit);
      while (cmask) {
        const int base = (cb + __builtin_ctzll(cmask)) * 64;
        cmask &= cmask - 1;
        if (!body(base)) return;
      }
    }
  }
}

// effective radius of a ball of radius R containing spheres seen from distance <= Lmax:
// a member sphere's cull radius is sqrt(r^2 + s L^2) <= r + sqrt(s) L
__device__ __forceinline__ float cluster_cull_radius(float R, float Lmax)
{
  return __builtin_fmaf(R, CULL_REL, __builtin_fmaf(CULL_DISC_SLACK_SQRT * CULL_REL, Lmax, CULL_ABS));
}

} // namespace mrt
