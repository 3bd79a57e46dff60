// mrt_cull.hip -- the default render kernel: per-tile conservative culling in front of the exact tests.
//
// The reference tests every ray against every object (src/main.cu:67-77, core/light.cuh:153-158).  The
// results of that loop depend only on the primitives a ray can actually hit, so this kernel first asks,
// once per 8x8 pixel tile (one wave64), which primitives can be hit by ANY ray of the tile:
//
//   * primitives are stored in kd order (mrt_api.cpp) with bounding balls for every 64 primitives (cluster) and every
//     64 clusters (super-cluster), built at upload (mrt_api.cpp).  At each level the 64 lanes each take a
//     DIFFERENT ball (coalesced SoA loads: 64 balls per instruction) and test it against a bound of the
//     tile's ray bundle
//       - primary rays:   a circular cone from the camera origin around the tile's four corner rays
//       - shadow rays:    the swept sphere of the tile's hit points along the light direction
//                         (directional light) or towards the light position (point light);
//   * __ballot turns the 64 verdicts into a wave-uniform mask; the wave walks the set bits in ascending
//     primitive order and runs the EXACT reference test (same fp32 operations, same order as
//     mrt_kernels.hip and the oracle) for all 64 pixels against that primitive, whose record is
//     broadcast from the culling lane's registers (v_readlane) or fetched with scalar loads.
//
// Culling never changes a result, only the number of exact tests: every bound is inflated by far more
// than the fp32 rounding of the exact test can move a verdict (see the margins below), a NaN or an
// infinite bound always passes, and tests/test_gpu_parity.py checks the buffers bit for bit against the
// brute-force kernel (MRT_CULL=0) and the oracle.  There is no LDS and no barrier in this kernel: every
// wave is independent.

#include "mrt_internal.h"
#include "mrt_device.h"
#include "mrt_cull_common.h"

#include <cmath>
#include <cstdlib>

namespace mrt {

// ---- primitive-side hierarchy walk ---------------------------------------------------------------------
// Primitives are stored in kd order; runs of 64 primitives have a bounding ball ("cluster"), runs of 64
// clusters another one ("super-cluster").  A wave walks the hierarchy with lane-parallel tests at every
// level: 64 super-clusters per step, then the 64 clusters of each surviving super-cluster, then
// body(first primitive of the cluster) for each surviving cluster.  A level with a single entry is not
// tested (tiny scenes: C2).  body returns false to stop the walk (shadow early-out).
template <class Test>
__device__ __forceinline__ unsigned long long level_mask(const float *__restrict__ balls, int n, int begin, int lane,
                                                         Test test)
{
  if (n == 1) return 1ull;
  const int i = begin + lane;
  bool pass = false;
  if (i < min(n, begin + 64)) pass = test(balls[i], balls[n + i], balls[2 * n + i], balls[3 * n + i]);
  return __ballot(pass);
}

template <bool HIER, class Test, class Body>
__device__ __forceinline__ void walk_hierarchy(const float *__restrict__ sclu, int nsc, const float *__restrict__ clu,
                                               int nc, int lane, Test test, Body body)
{
  if (!HIER) { // at most one cluster per primitive type (C2): nothing to walk
    if (nc > 0) body(0);
    return;
  }
  for (int sb = 0; sb < nsc; sb += 64) {
    unsigned long long smask = level_mask(sclu, nsc, sb, lane, test);
    while (smask) {
      const int cb = (sb + __builtin_ctzll(smask)) * 64;
      smask &= smask - 1;
      unsigned long long cmask = level_mask(clu, nc, cb, lane, test);
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
  return R * CULL_REL + CULL_DISC_SLACK_SQRT * CULL_REL * Lmax + CULL_ABS;
}

template <bool HIER>
__global__ __launch_bounds__(256) void render_cull_kernel(RenderArgs A)
{
  const DevScene &sc = A.sc;
  const DevCamera &cam = A.cam;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tx = lane & 7, ty = lane >> 3;
  const int ns = sc.ns, nt = sc.nt;

  // 2-D grid: blockIdx.y = tile row, blockIdx.x*4 + wave = tile column (both wave-uniform, so the strip
  // arithmetic below runs on the scalar unit)
  const int tile_x = blockIdx.x * 4 + wave;
  const int tile_y = blockIdx.y;
  if (tile_x * 8 >= A.W) return; // waves are independent: no barrier anywhere below
  const int col = tile_x * 8 + tx;
  const int lrow = tile_y * 8 + ty;
  // strips are multiples of 8 rows, so a tile never straddles two strips
  const int lrow0 = tile_y * 8;
  const int strip = lrow0 / A.strip_rows;
  const int grow0 = (strip * A.nranks + A.rank) * A.strip_rows + (lrow0 - strip * A.strip_rows);
  const int grow = grow0 + ty;
  const bool valid = (lrow < A.local_rows) && (grow < A.H) && (col < A.W);

  // ---- cone around the tile's rays, padded by one pixel on every side (which also covers the sub-pixel
  //      offsets of the 121-spp pattern, within (-11/13, +9/13) pixel).  d(u,v) is affine in (u,v), so every
  //      ray of the padded tile is the centre ray plus an offset of norm <= cone_ext (launcher): the angle
  //      to the centre ray is at most asin(cone_ext / |centre ray|). ----
  float cax, cay, caz, cosT, sinT;
  {
    const float rw = fast_rcp(float(A.W)), rh = fast_rcp(float(A.H)); // bound arithmetic: approximate is fine
    const float um = (float(tile_x * 8) + 3.5f) * rw, vm = (float(grow0) + 3.5f) * rh;
    cax = ((cam.ux + um * cam.hx) + vm * cam.vx) - cam.ox;
    cay = ((cam.uy + um * cam.hy) + vm * cam.vy) - cam.oy;
    caz = ((cam.uz + um * cam.hz) + vm * cam.vz) - cam.oz;
    const float inv = fast_rsq(dot3(cax, cay, caz, cax, cay, caz));
    cax *= inv; cay *= inv; caz *= inv;
    sinT = A.cone_ext * inv * (1.0f + 1e-4f) + 1e-6f;
    cosT = fast_sqrt(fmaxf(1.0f - sinT * sinT, 0.0f)) * (1.0f - 1e-5f) - 1e-6f;
    // a cone of 90 degrees or more (or a NaN camera) cannot cull: cos = -1 makes every ball pass
    if (!(sinT < 0.99f) || !(cosT > 0.0f)) { cosT = -1.0f; sinT = 0.0f; }
  }
  // hierarchy tests against the cone: a ball of spheres / a ball of triangle bounds
  auto cone_vs_sphere_ball = [&](float x, float y, float z, float R) {
    const float wx = x - cam.ox, wy = y - cam.oy, wz = z - cam.oz;
    const float L = fast_sqrt(dot3(wx, wy, wz, wx, wy, wz)) * CULL_REL + R;
    return ball_may_touch_cone(wx, wy, wz, cluster_cull_radius(R, L), cax, cay, caz, cosT, sinT);
  };
  auto cone_vs_tri_ball = [&](float x, float y, float z, float R) {
    return ball_may_touch_cone(x - cam.ox, y - cam.oy, z - cam.oz, R * CULL_REL + CULL_ABS, cax, cay, caz, cosT, sinT);
  };

  float accr = 0.0f, accg = 0.0f, accb = 0.0f;
  int out_id = -1;
  float out_t = MRT_INF;
  unsigned int n_shadow = 0, n_hit = 0, n_occl = 0;
  unsigned int n_exact = 0, n_clusters = 0; // wave-uniform diagnostics (lane 0 reports them)
  const int nsamp = A.spp;
  const float DIVW = 13.0f, DIVH = 13.0f;
  const float v = float(grow) / float(A.H); // src/main.cu:137

  for (int sidx = 0; sidx < nsamp; ++sidx) {
    float offx = 0.0f, offy = 0.0f;
    if (nsamp > 1) { // pipeline/multisample_mesh.cuh:23-30
      int r = sidx / 11, c = sidx - r * 11;
      offy = (-1.0f + 2.0f * float(r + 1) / DIVH) / float(A.H);
      offx = (-1.0f + 2.0f * float(c + 1) / DIVW) / float(A.W);
    }
    // ---- primary ray: pipeline/viewport.cuh:30-34 ----
    float su = float(col) / float(A.W), sv = v; // src/main.cu:136
    if (nsamp > 1) { su = su + offx; sv = v + offy; }
    const float dx = ((cam.ux + su * cam.hx) + sv * cam.vx) - cam.ox;
    const float dy = ((cam.uy + su * cam.hy) + sv * cam.vy) - cam.oy;
    const float dz = ((cam.uz + su * cam.hz) + sv * cam.vz) - cam.oz;
    const float aa = dot3(dx, dy, dz, dx, dy, dz);
    float best_t = MRT_INF;
    int best_obj = 0x7fffffff, best_prim = -1;

    // ---- closest hit over spheres: src/main.cu:60-77, core/sphere.cuh:32-73, interval [0.01, inf] ----
    walk_hierarchy<HIER>(sc.sph_sclu, sc.nscs, sc.sph_clu, sc.ncs, lane, cone_vs_sphere_ball, [&](int base) {
      const int i = base + lane;
      float cx = 0.0f, cy = 0.0f, cz = 0.0f, r = 0.0f;
      bool pass = false;
      if (i < ns) {
        cx = sc.sph_cx[i]; cy = sc.sph_cy[i]; cz = sc.sph_cz[i]; r = sc.sph_r[i];
        const float wx = cx - cam.ox, wy = cy - cam.oy, wz = cz - cam.oz;
        const float L2 = dot3(wx, wy, wz, wx, wy, wz);
        pass = ball_may_touch_cone(wx, wy, wz, sphere_cull_radius(r * r, L2), cax, cay, caz, cosT, sinT);
      }
      unsigned long long mask = __ballot(pass);
      n_clusters++; n_exact += __popcll(mask);
      while (mask) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1;
        const float scx = bcast(cx, j), scy = bcast(cy, j), scz = bcast(cz, j), sr = bcast(r, j);
        const float ocx = cam.ox - scx, ocy = cam.oy - scy, ocz = cam.oz - scz;
        const float b = dot3(ocx, ocy, ocz, dx, dy, dz);
        const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - sr * sr;
        const float disc = b * b - aa * c;
        if (disc >= 0.0f) {
          float t;
          if (sphere_roots(aa, b, disc, 0.01f, MRT_INF, t)) {
            const int obj = sc.sph_obj[base + j];
            if (t < best_t || (t == best_t && t < MRT_INF && obj < best_obj)) {
              best_t = t; best_obj = obj; best_prim = base + j;
            }
          }
        }
      }
      return true;
    });

    // ---- closest hit over triangles: core/triangle.cuh:77-103 ----
    walk_hierarchy<HIER>(sc.tri_sclu, sc.nsct, sc.tri_clu, sc.nct, lane, cone_vs_tri_ball, [&](int base) {
      const int i = base + lane;
      bool pass = false;
      if (i < nt) pass = cone_vs_tri_ball(sc.tri_bx[i], sc.tri_by[i], sc.tri_bz[i], sc.tri_br[i]);
      unsigned long long mask = __ballot(pass);
      n_clusters++; n_exact += __popcll(mask);
      while (mask) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1;
        const float *q = sc.tri_rec + 16 * size_t(base + j); // wave-uniform address: one s_load_dwordx16
        const float pnx = q[0], pny = q[1], pnz = q[2], kp = q[3];
        const float rdn = dot3(dx, dy, dz, pnx, pny, pnz);
        const float t = (kp - dot3(cam.ox, cam.oy, cam.oz, pnx, pny, pnz)) / rdn;
        const float px = cam.ox + t * dx, py = cam.oy + t * dy, pz = cam.oz + t * dz;
        const float e0 = dot3(q[4], q[5], q[6], px, py, pz) - q[7];
        const float e1 = dot3(q[8], q[9], q[10], px, py, pz) - q[11];
        const float e2 = dot3(q[12], q[13], q[14], px, py, pz) - q[15];
        const bool hit = (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (0.01f <= t) & (t <= MRT_INF);
        if (hit) {
          const int obj = sc.tri_obj[base + j];
          if (t < best_t || (t == best_t && t < MRT_INF && obj < best_obj)) {
            best_t = t; best_obj = obj; best_prim = ns + base + j;
          }
        }
      }
      return true;
    });

    // ---- hit record of the winner (core/sphere.cuh:45-47, core/triangle.cuh:96-97) ----
    const bool have_hit = best_prim >= 0;
    float hpx = 0.0f, hpy = 0.0f, hpz = 0.0f, hnx = 0.0f, hny = 0.0f, hnz = 0.0f;
    if (have_hit) {
      hpx = cam.ox + best_t * dx; hpy = cam.oy + best_t * dy; hpz = cam.oz + best_t * dz;
      if (best_prim < ns) {
        float nx = hpx - sc.sph_cx[best_prim], ny = hpy - sc.sph_cy[best_prim], nz = hpz - sc.sph_cz[best_prim];
        normalize3(nx, ny, nz);
        hnx = nx; hny = ny; hnz = nz;
      } else {
        const float *q = sc.tri_rec + 16 * size_t(best_prim - ns);
        const float pnx = q[0], pny = q[1], pnz = q[2];
        const bool front = dot3(dx, dy, dz, pnx, pny, pnz) < 0;
        hnx = front ? pnx : -pnx; hny = front ? pny : -pny; hnz = front ? pnz : -pnz;
      }
      if (valid) n_hit++;
    }

    // ---- shadow rays + direct light: core/light.cuh:135-178 ----
    float lr = 0.0f, lg = 0.0f, lb = 0.0f;
    bool have_light = false;
    if (__any(have_hit)) {
      // ball around the tile's hit points (any subset of them is covered too)
      Bundle B;
      {
        if (ns + nt > 64) {
          // many primitives: a tight ball pays for itself (centre of the hit points' box, 7 reductions)
          const float big = 3.0e38f;
          const float lox = wave_min(have_hit ? hpx : big), hix = wave_max(have_hit ? hpx : -big);
          const float loy = wave_min(have_hit ? hpy : big), hiy = wave_max(have_hit ? hpy : -big);
          const float loz = wave_min(have_hit ? hpz : big), hiz = wave_max(have_hit ? hpz : -big);
          B.mx = 0.5f * (lox + hix); B.my = 0.5f * (loy + hiy); B.mz = 0.5f * (loz + hiz);
        } else {
          // few primitives: centre = the first hit lane's point (1 reduction)
          const int first = __builtin_ctzll(__ballot(have_hit));
          B.mx = bcast(hpx, first); B.my = bcast(hpy, first); B.mz = bcast(hpz, first);
        }
        const float ex = hpx - B.mx, ey = hpy - B.my, ez = hpz - B.mz;
        const float e2 = have_hit ? dot3(ex, ey, ez, ex, ey, ez) : 0.0f;
        B.rho = fast_sqrt(wave_max(e2)) * CULL_REL + CULL_ABS;
      }
      auto bundle_vs_sphere_ball = [&](float x, float y, float z, float R) {
        const float wx = x - B.mx, wy = y - B.my, wz = z - B.mz;
        const float Lm = fast_sqrt(dot3(wx, wy, wz, wx, wy, wz)) * CULL_REL + B.rho + R;
        return bundle_may_touch(B, x, y, z, cluster_cull_radius(R, Lm));
      };
      auto bundle_vs_tri_ball = [&](float x, float y, float z, float R) { return bundle_may_touch(B, x, y, z, R); };

      for (int li = 0; li < sc.nl; ++li) {
        const int lkind = sc.light_kind[li];
        const float lvx = sc.light_vec[3 * li], lvy = sc.light_vec[3 * li + 1], lvz = sc.light_vec[3 * li + 2];
        // per-lane shadow ray and the light's "hit record" (ray_from: core/light.cuh:73-81 / :47-53)
        float sdx, sdy, sdz, tl, lnx, lny, lnz;
        if (lkind == 0) {
          sdx = -lvx; sdy = -lvy; sdz = -lvz; tl = MRT_INF;
          lnx = lvx; lny = lvy; lnz = lvz;
          B.point = false;
          B.Dx = sdx; B.Dy = sdy; B.Dz = sdz;
        } else {
          sdx = lvx - hpx; sdy = lvy - hpy; sdz = lvz - hpz; tl = 1.0f;
          lnx = hpx - lvx; lny = hpy - lvy; lnz = hpz - lvz;
          normalize3(lnx, lny, lnz);
          B.point = true;
          B.Dx = lvx - B.mx; B.Dy = lvy - B.my; B.Dz = lvz - B.mz;
        }
        B.DD = dot3(B.Dx, B.Dy, B.Dz, B.Dx, B.Dy, B.Dz);
        bundle_make_slab(B, cax, cay, caz, hpx - B.mx, hpy - B.my, hpz - B.mz, have_hit);
        const float sa = dot3(sdx, sdy, sdz, sdx, sdy, sdz);
        // the light must not shine through the surface, core/light.cuh:148
        const bool need = have_hit && !(dot3(sdx, sdy, sdz, hnx, hny, hnz) < 0);
        bool open = need;

        // any-hit over spheres, interval [1e-3, t_light] (core/light.cuh:28-33)
        if (__any(open))
          walk_hierarchy<HIER>(sc.sph_sclu, sc.nscs, sc.sph_clu, sc.ncs, lane, bundle_vs_sphere_ball, [&](int base) {
            const int i = base + lane;
            float cx = 0.0f, cy = 0.0f, cz = 0.0f, r = 0.0f;
            bool pass = false;
            if (i < ns) {
              cx = sc.sph_cx[i]; cy = sc.sph_cy[i]; cz = sc.sph_cz[i]; r = sc.sph_r[i];
              const float wx = cx - B.mx, wy = cy - B.my, wz = cz - B.mz;
              const float Lm = fast_sqrt(dot3(wx, wy, wz, wx, wy, wz)) * CULL_REL + B.rho; // >= |hit point - centre|
              pass = bundle_may_touch(B, cx, cy, cz, sphere_cull_radius(r * r, Lm * Lm));
            }
            unsigned long long mask = __ballot(pass);
            n_clusters++; n_exact += __popcll(mask);
            while (mask) {
              const int j = __builtin_ctzll(mask);
              mask &= mask - 1;
              const float scx = bcast(cx, j), scy = bcast(cy, j), scz = bcast(cz, j), sr = bcast(r, j);
              const float ocx = hpx - scx, ocy = hpy - scy, ocz = hpz - scz;
              const float b = dot3(ocx, ocy, ocz, sdx, sdy, sdz);
              const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - sr * sr;
              const float disc = b * b - sa * c;
              if (open && disc >= 0.0f) {
                float t;
                if (sphere_roots(sa, b, disc, 1e-3f, tl, t)) open = false;
              }
              if (!__any(open)) return false; // ballot early-out
            }
            return true;
          });
        // any-hit over triangles
        if (__any(open))
          walk_hierarchy<HIER>(sc.tri_sclu, sc.nsct, sc.tri_clu, sc.nct, lane, bundle_vs_tri_ball, [&](int base) {
            const int i = base + lane;
            bool pass = false;
            if (i < nt) pass = bundle_may_touch(B, sc.tri_bx[i], sc.tri_by[i], sc.tri_bz[i], sc.tri_br[i]);
            unsigned long long mask = __ballot(pass);
            n_clusters++; n_exact += __popcll(mask);
            while (mask) {
              const int j = __builtin_ctzll(mask);
              mask &= mask - 1;
              const float *q = sc.tri_rec + 16 * size_t(base + j);
              const float pnx = q[0], pny = q[1], pnz = q[2], kp = q[3];
              const float rdn = dot3(sdx, sdy, sdz, pnx, pny, pnz);
              const float t = (kp - dot3(hpx, hpy, hpz, pnx, pny, pnz)) / rdn;
              const float px = hpx + t * sdx, py = hpy + t * sdy, pz = hpz + t * sdz;
              const float e0 = dot3(q[4], q[5], q[6], px, py, pz) - q[7];
              const float e1 = dot3(q[8], q[9], q[10], px, py, pz) - q[11];
              const float e2 = dot3(q[12], q[13], q[14], px, py, pz) - q[15];
              const bool h = (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (1e-3f <= t) & (t <= tl);
              if (open && h) open = false;
              if (!__any(open)) return false;
            }
            return true;
          });

        if (need && valid) { n_shadow++; if (!open) n_occl++; }
        if (open) {
          // light_bounce with the LIGHT's normal, as the reference passes it (core/light.cuh:164-168)
          const float lcr = sc.light_rgb[3 * li], lcg = sc.light_rgb[3 * li + 1], lcb = sc.light_rgb[3 * li + 2];
          const float dn = dot3(sdx, sdy, sdz, lnx, lny, lnz);
          float cr, cg, cb;
          if (sc.mat_kind[best_obj] == 0) { // DiffuseMaterial::light_bounce, core/material.cuh:63-75
            const float scale = -dn;
            if (scale > 0.0f) { cr = scale * lcr; cg = scale * lcg; cb = scale * lcb; }
            else { cr = 0.0f; cg = 0.0f; cb = 0.0f; }
          } else {                          // MetallicMaterial::light_bounce, core/material.cuh:134-158
            const float s2 = 2.0f * dn;
            const float rx = sdx - s2 * lnx, ry = sdy - s2 * lny, rz = sdz - s2 * lnz;
            if (dot3(rx, ry, rz, dx, dy, dz) < -0.99f) { cr = lcr; cg = lcg; cb = lcb; }
            else { cr = 0.001f; cg = 0.001f; cb = 0.001f; }
          }
          if (have_light) { lr += cr; lg += cg; lb += cb; }
          else { lr = cr; lg = cg; lb = cb; have_light = true; }
        }
      }
    }

    // ---- combine: src/main.cu:91-104 with MAX_RECURSION_DEPTH = 0 ----
    float cr, cg, cb;
    if (have_hit) {
      const float *m = sc.mat_rgb + 3 * best_obj;
      const float mr = m[0], mg = m[1], mb = m[2];
      const float inc = 0.01f; // what the depth-1 call returns, src/main.cu:51-53
      if (have_light) { cr = inc * mr + lr * mr; cg = inc * mg + lg * mg; cb = inc * mb + lb * mb; }
      else { cr = inc * mr; cg = inc * mg; cb = inc * mb; }
    } else {
      float nx = dx, ny = dy, nz = dz;
      normalize3(nx, ny, nz);
      const float y = 0.5f * (ny + 1.0f);
      const float w0 = 1.0f - y;
      cr = w0 * 1.0f + y * 0.5f; cg = w0 * 1.0f + y * 0.7f; cb = w0 * 1.0f + y * 1.0f;
    }
    accr += cr; accg += cg; accb += cb; // src/main.cu:151
    out_id = have_hit ? best_obj : -1;
    out_t = best_t;
  }

  if (valid) { // src/main.cu:155
    const float fn = float(nsamp);
    const size_t pix = size_t(lrow) * A.W + col;
    A.rgb[3 * pix + 0] = accr / fn;
    A.rgb[3 * pix + 1] = accg / fn;
    A.rgb[3 * pix + 2] = accb / fn;
    if (A.id) A.id[pix] = out_id;
    if (A.t) A.t[pix] = out_t;
  }

  if (A.stats) {
    unsigned int v4[4] = {(valid ? 1u : 0u) * (unsigned)nsamp, n_shadow, n_hit, n_occl};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned int x = v4[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0 && x) atomicAdd(&A.stats[k], (unsigned long long)x);
    }
    if (lane == 0) {
      atomicAdd(&A.stats[4], (unsigned long long)n_exact);
      atomicAdd(&A.stats[5], (unsigned long long)n_clusters);
    }
  }
}

hipError_t launch_render_cull(const RenderArgs &in, hipStream_t stream)
{
  RenderArgs A = in;
  // largest offset of a ray of the padded tile (pixels -1..8, i.e. +-4.5 around the centre 3.5) from the
  // tile's centre ray: max |+-4.5/W hor +- 4.5/H ver|, in double, rounded up
  {
    const double hx[3] = {4.5 / A.W * A.cam.hx, 4.5 / A.W * A.cam.hy, 4.5 / A.W * A.cam.hz};
    const double hy[3] = {4.5 / A.H * A.cam.vx, 4.5 / A.H * A.cam.vy, 4.5 / A.H * A.cam.vz};
    double p = 0, m = 0;
    for (int k = 0; k < 3; ++k) { p += (hx[k] + hy[k]) * (hx[k] + hy[k]); m += (hx[k] - hy[k]) * (hx[k] - hy[k]); }
    A.cone_ext = float(std::sqrt(p > m ? p : m) * 1.001);
  }
  const int tiles_x = (A.W + 7) / 8, tiles_y = (A.local_rows + 7) / 8;
  if (tiles_x == 0 || tiles_y == 0) return hipSuccess;
  const dim3 grid((tiles_x + 3) / 4, tiles_y);
  if (A.sc.ncs > 1 || A.sc.nct > 1) hipLaunchKernelGGL(render_cull_kernel<true>, grid, dim3(256), 0, stream, A);
  else hipLaunchKernelGGL(render_cull_kernel<false>, grid, dim3(256), 0, stream, A);
  return hipGetLastError();
}

} // namespace mrt
