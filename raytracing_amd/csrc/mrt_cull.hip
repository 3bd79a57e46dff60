// mrt_cull.hip -- the default render kernel: per-tile conservative culling in front of the exact tests.
//
// The reference tests every ray against every object (src/main.cu:67-77, core/light.cuh:153-158).  The
// results of that loop depend only on the primitives a ray can actually hit, so this kernel first asks,
// once per 8x8 pixel tile (one wave64), which primitives can be hit by ANY ray of the tile:
//
//   * primitives are stored in kd order (mrt_scene_build.h) with bounding balls for every 64 primitives (cluster) and every
//     64 clusters (super-cluster), built at upload (mrt_scene_build.h).  At each level the 64 lanes each take a
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
#include <type_traits>

namespace mrt {

template <int P> struct CVec;
template <> struct CVec<1> { using f = float; using i = int; };
template <> struct CVec<2> { typedef float f __attribute__((ext_vector_type(2))); typedef int i __attribute__((ext_vector_type(2))); };
template <> struct CVec<4> { typedef float f __attribute__((ext_vector_type(4))); typedef int i __attribute__((ext_vector_type(4))); };

// HIER: walk the cluster hierarchy (false: at most one cluster per primitive type).  P: horizontally adjacent
// pixels per lane, i.e. a wave renders 8P x 8 pixels and pays for the walk once per tile.  MULTI: more than one
// sample per pixel (false: no accumulation across samples, fewer registers).
template <bool HIER, int P, bool MULTI>
__global__ __launch_bounds__(256) void render_cull_kernel(RenderArgs A)
{
  const DevScene &sc = A.sc;
  const DevCamera &cam = A.cam;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tx = lane & 7, ty = lane >> 3;
  const int ns = sc.ns, nt = sc.nt;
  constexpr int TW = 8 * P;

  // 2-D grid: blockIdx.y = tile row, blockIdx.x*4 + wave = tile column (both wave-uniform, so the strip
  // arithmetic below runs on the scalar unit)
  const int tile_x = blockIdx.x * (blockDim.x >> 6) + wave;
  const int tile_y = blockIdx.y;
  if (tile_x * TW >= A.W) return; // waves are independent: no barrier anywhere below
  const int col0 = tile_x * TW + tx * P;
  const int lrow = tile_y * 8 + ty;
  // strips are multiples of 8 rows, so a tile never straddles two strips
  const int lrow0 = tile_y * 8;
  const int strip = lrow0 / A.strip_rows;
  const int grow0 = (strip * A.nranks + A.rank) * A.strip_rows + (lrow0 - strip * A.strip_rows);
  const int grow = grow0 + ty;
  const bool valid_row = (lrow < A.local_rows) && (grow < A.H);
  bool valid[P];
#pragma unroll
  for (int p = 0; p < P; ++p) valid[p] = valid_row && (col0 + p < A.W);

  // ---- cone around the tile's rays, padded by one pixel on every side (which also covers the sub-pixel
  //      offsets of the 121-spp pattern, within (-11/13, +9/13) pixel).  d(u,v) is affine in (u,v), so every
  //      ray of the padded tile is the centre ray plus an offset of norm <= cone_ext (launcher): the angle
  //      to the centre ray is at most asin(cone_ext / |centre ray|). ----
  float cax, cay, caz, cosT, sinT;
  {
    const float rw = fast_rcp(float(A.W)), rh = fast_rcp(float(A.H)); // bound arithmetic: approximate is fine
    const float um = (float(tile_x * TW) + 0.5f * float(TW - 1)) * rw, vm = (float(grow0) + 3.5f) * rh;
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

  float accr[P], accg[P], accb[P], out_t[P];
  int out_id[P];
#pragma unroll
  for (int p = 0; p < P; ++p) { accr[p] = 0.0f; accg[p] = 0.0f; accb[p] = 0.0f; out_id[p] = -1; out_t[p] = MRT_INF; }
  unsigned int n_shadow = 0, n_hit = 0, n_occl = 0;
  unsigned int n_exact = 0, n_clusters = 0; // wave-uniform diagnostics (lane 0 reports them)
  const int nsamp = MULTI ? A.spp : 1;
  const float DIVW = 13.0f, DIVH = 13.0f;
  const float v = float(grow) / float(A.H); // src/main.cu:137

  for (int sidx = 0; sidx < nsamp; ++sidx) {
    float offx = 0.0f, offy = 0.0f;
    if (nsamp > 1) { // pipeline/multisample_mesh.cuh:23-30
      int r = sidx / 11, c = sidx - r * 11;
      offy = (-1.0f + 2.0f * float(r + 1) / DIVH) / float(A.H);
      offx = (-1.0f + 2.0f * float(c + 1) / DIVW) / float(A.W);
    }
    // ---- primary rays: pipeline/viewport.cuh:30-34 ----
    float dx[P], dy[P], dz[P], aa[P], best_t[P];
    int best_obj[P], best_prim[P];
    const float sv = nsamp > 1 ? v + offy : v;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float su = float(col0 + p) / float(A.W); // src/main.cu:136
      if (nsamp > 1) su = su + offx;
      dx[p] = ((cam.ux + su * cam.hx) + sv * cam.vx) - cam.ox;
      dy[p] = ((cam.uy + su * cam.hy) + sv * cam.vy) - cam.oy;
      dz[p] = ((cam.uz + su * cam.hz) + sv * cam.vz) - cam.oz;
      aa[p] = dot3(dx[p], dy[p], dz[p], dx[p], dy[p], dz[p]);
      best_t[p] = MRT_INF; best_obj[p] = 0x7fffffff; best_prim[p] = -1;
    }

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
        const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - sr * sr;
        const int obj = sc.sph_obj[base + j];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float b = dot3(ocx, ocy, ocz, dx[p], dy[p], dz[p]);
          const float disc = b * b - aa[p] * c;
          if (disc >= 0.0f) {
            float t;
            if (sphere_roots(aa[p], b, disc, 0.01f, MRT_INF, t)) {
              if (t < best_t[p] || (t == best_t[p] && t < MRT_INF && obj < best_obj[p])) {
                best_t[p] = t; best_obj[p] = obj; best_prim[p] = base + j;
              }
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
        const float num = kp - dot3(cam.ox, cam.oy, cam.oz, pnx, pny, pnz);
        const int obj = sc.tri_obj[base + j];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float rdn = dot3(dx[p], dy[p], dz[p], pnx, pny, pnz);
          const float t = num / rdn;
          // the three edge tests only matter in lanes whose t is in range and could still win; no such lane
          // in the wave (the triangle lies behind what the tile already hit, or behind the camera): skip them
          const bool cand = (rdn != 0.0f) & (0.01f <= t) & (t <= best_t[p]);
          if (!__any(cand)) continue;
          const float px = cam.ox + t * dx[p], py = cam.oy + t * dy[p], pz = cam.oz + t * dz[p];
          const float e0 = dot3(q[4], q[5], q[6], px, py, pz) - q[7];
          const float e1 = dot3(q[8], q[9], q[10], px, py, pz) - q[11];
          const float e2 = dot3(q[12], q[13], q[14], px, py, pz) - q[15];
          const bool hit = (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (0.01f <= t) & (t <= MRT_INF);
          if (hit) {
            if (t < best_t[p] || (t == best_t[p] && t < MRT_INF && obj < best_obj[p])) {
              best_t[p] = t; best_obj[p] = obj; best_prim[p] = ns + base + j;
            }
          }
        }
      }
      return true;
    });

    // ---- hit record of the winner (core/sphere.cuh:45-47, core/triangle.cuh:96-97) ----
    bool have_hit[P], any_hit_lane = false;
    float hpx[P], hpy[P], hpz[P], hnx[P], hny[P], hnz[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      have_hit[p] = best_prim[p] >= 0;
      any_hit_lane |= have_hit[p];
      hpx[p] = 0.0f; hpy[p] = 0.0f; hpz[p] = 0.0f; hnx[p] = 0.0f; hny[p] = 0.0f; hnz[p] = 0.0f;
      if (have_hit[p]) {
        hpx[p] = cam.ox + best_t[p] * dx[p]; hpy[p] = cam.oy + best_t[p] * dy[p]; hpz[p] = cam.oz + best_t[p] * dz[p];
        if (best_prim[p] < ns) {
          float nx = hpx[p] - sc.sph_cx[best_prim[p]], ny = hpy[p] - sc.sph_cy[best_prim[p]], nz = hpz[p] - sc.sph_cz[best_prim[p]];
          normalize3(nx, ny, nz);
          hnx[p] = nx; hny[p] = ny; hnz[p] = nz;
        } else {
          const float *q = sc.tri_rec + 16 * size_t(best_prim[p] - ns);
          const float pnx = q[0], pny = q[1], pnz = q[2];
          const bool front = dot3(dx[p], dy[p], dz[p], pnx, pny, pnz) < 0;
          hnx[p] = front ? pnx : -pnx; hny[p] = front ? pny : -pny; hnz[p] = front ? pnz : -pnz;
        }
        if (valid[p]) n_hit++;
      }
    }

    // ---- shadow rays + direct light: core/light.cuh:135-178 ----
    float lr[P], lg[P], lb[P];
    bool have_light[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { lr[p] = 0.0f; lg[p] = 0.0f; lb[p] = 0.0f; have_light[p] = false; }
    if (__any(any_hit_lane)) {
      // ball around the tile's hit points (any subset of them is covered too)
      Bundle B;
      {
        if (ns + nt > 64) {
          // many primitives: a tight ball pays for itself (centre of the hit points' box, 6 reductions)
          const float big = 3.0e38f;
          float lox = big, loy = big, loz = big, hix = -big, hiy = -big, hiz = -big;
#pragma unroll
          for (int p = 0; p < P; ++p)
            if (have_hit[p]) {
              lox = fminf(lox, hpx[p]); hix = fmaxf(hix, hpx[p]);
              loy = fminf(loy, hpy[p]); hiy = fmaxf(hiy, hpy[p]);
              loz = fminf(loz, hpz[p]); hiz = fmaxf(hiz, hpz[p]);
            }
          lox = wave_min(lox); hix = wave_max(hix);
          loy = wave_min(loy); hiy = wave_max(hiy);
          loz = wave_min(loz); hiz = wave_max(hiz);
          B.mx = 0.5f * (lox + hix); B.my = 0.5f * (loy + hiy); B.mz = 0.5f * (loz + hiz);
        } else {
          // few primitives: centre = the first hit point found (1 reduction)
          float mx = 0.0f, my = 0.0f, mz = 0.0f;
          bool found = false;
#pragma unroll
          for (int p = 0; p < P; ++p) {
            const unsigned long long hm = __ballot(have_hit[p]);
            if (!found && hm) {
              const int first = __builtin_ctzll(hm);
              mx = bcast(hpx[p], first); my = bcast(hpy[p], first); mz = bcast(hpz[p], first);
              found = true;
            }
          }
          B.mx = mx; B.my = my; B.mz = mz;
        }
        float e2 = 0.0f;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float ex = hpx[p] - B.mx, ey = hpy[p] - B.my, ez = hpz[p] - B.mz;
          e2 = fmaxf(e2, have_hit[p] ? dot3(ex, ey, ez, ex, ey, ez) : 0.0f);
        }
        B.rho = fast_sqrt(wave_max(e2)) * CULL_REL + CULL_ABS;
      }
      auto bundle_vs_sphere_ball = [&](float x, float y, float z, float R) {
        const float wx = x - B.mx, wy = y - B.my, wz = z - B.mz;
        const float Lm = fast_sqrt(dot3(wx, wy, wz, wx, wy, wz)) * CULL_REL + B.rho + R;
        return bundle_may_touch(B, x, y, z, cluster_cull_radius(R, Lm));
      };
      auto bundle_vs_tri_ball = [&](float x, float y, float z, float R) { return bundle_may_touch(B, x, y, z, R); };

      // One light.  POINT is a compile-time switch: for a directional light the shadow ray direction, its
      // squared length and the light's normal are wave-uniform scalars, for a point light per-pixel registers.
      auto shade_light = [&](auto point_tag, int li, float lvx, float lvy, float lvz) {
        constexpr bool POINT = decltype(point_tag)::value;
        constexpr int Q = POINT ? P : 1;
        auto ix = [](int p) { return POINT ? p : 0; };
        // per-lane shadow ray and the light's "hit record" (ray_from: core/light.cuh:73-81 / :47-53)
        float sdx[Q], sdy[Q], sdz[Q], lnx[Q], lny[Q], lnz[Q], sa[Q];
        float tl;
        if (!POINT) {
          tl = MRT_INF;
          sdx[0] = -lvx; sdy[0] = -lvy; sdz[0] = -lvz;
          lnx[0] = lvx; lny[0] = lvy; lnz[0] = lvz;
          B.point = false;
          B.Dx = -lvx; B.Dy = -lvy; B.Dz = -lvz;
        } else {
          tl = 1.0f;
#pragma unroll
          for (int p = 0; p < Q; ++p) {
            sdx[p] = lvx - hpx[p]; sdy[p] = lvy - hpy[p]; sdz[p] = lvz - hpz[p];
            float nx = hpx[p] - lvx, ny = hpy[p] - lvy, nz = hpz[p] - lvz;
            normalize3(nx, ny, nz);
            lnx[p] = nx; lny[p] = ny; lnz[p] = nz;
          }
          B.point = true;
          B.Dx = lvx - B.mx; B.Dy = lvy - B.my; B.Dz = lvz - B.mz;
        }
        B.DD = dot3(B.Dx, B.Dy, B.Dz, B.Dx, B.Dy, B.Dz);
        {
          // slab of the bundle for this light (bundle_make_slab, here with P hit points per lane)
          float nx = cay * B.Dz - caz * B.Dy, ny = caz * B.Dx - cax * B.Dz, nz = cax * B.Dy - cay * B.Dx;
          const float nn = dot3(nx, ny, nz, nx, ny, nz);
          B.snx = 0.0f; B.sny = 0.0f; B.snz = 0.0f; B.sw = MRT_INF;
          if (nn > 2.5e-3f * B.DD) {
            const float inv = fast_rsq(nn);
            nx *= inv; ny *= inv; nz *= inv;
            float dmax = 0.0f;
#pragma unroll
            for (int p = 0; p < P; ++p)
              dmax = fmaxf(dmax, have_hit[p] ? fabsf(dot3(nx, ny, nz, hpx[p] - B.mx, hpy[p] - B.my, hpz[p] - B.mz)) : 0.0f);
            B.snx = nx; B.sny = ny; B.snz = nz;
            B.sw = wave_max(dmax) * CULL_REL + CULL_ABS + (B.point ? 1e-5f * norm1(B.Dx, B.Dy, B.Dz) : 0.0f);
          }
        }
#pragma unroll
        for (int q = 0; q < Q; ++q) sa[q] = dot3(sdx[q], sdy[q], sdz[q], sdx[q], sdy[q], sdz[q]);
        bool need[P], open[P], any_open = false;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          // the light must not shine through the surface, core/light.cuh:148
          need[p] = have_hit[p] && !(dot3(sdx[ix(p)], sdy[ix(p)], sdz[ix(p)], hnx[p], hny[p], hnz[p]) < 0);
          open[p] = need[p];
          any_open |= open[p];
        }

        // any-hit over spheres, interval [1e-3, t_light] (core/light.cuh:28-33)
        if (__any(any_open))
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
              bool still = false;
#pragma unroll
              for (int p = 0; p < P; ++p) {
                const float ocx = hpx[p] - scx, ocy = hpy[p] - scy, ocz = hpz[p] - scz;
                const float b = dot3(ocx, ocy, ocz, sdx[ix(p)], sdy[ix(p)], sdz[ix(p)]);
                const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - sr * sr;
                const float disc = b * b - sa[ix(p)] * c;
                if (open[p] && disc >= 0.0f) {
                  float t;
                  if (sphere_roots(sa[ix(p)], b, disc, 1e-3f, tl, t)) open[p] = false;
                }
                still |= open[p];
              }
              if (!__any(still)) return false; // ballot early-out
            }
            return true;
          });
        any_open = false;
#pragma unroll
        for (int p = 0; p < P; ++p) any_open |= open[p];
        // any-hit over triangles
        if (__any(any_open))
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
              bool still = false;
#pragma unroll
              for (int p = 0; p < P; ++p) {
                const float rdn = dot3(sdx[ix(p)], sdy[ix(p)], sdz[ix(p)], pnx, pny, pnz);
                const float t = (kp - dot3(hpx[p], hpy[p], hpz[p], pnx, pny, pnz)) / rdn;
                if (!__any(open[p] & (rdn != 0.0f) & (1e-3f <= t) & (t <= tl))) { still |= open[p]; continue; }
                const float px = hpx[p] + t * sdx[ix(p)], py = hpy[p] + t * sdy[ix(p)], pz = hpz[p] + t * sdz[ix(p)];
                const float e0 = dot3(q[4], q[5], q[6], px, py, pz) - q[7];
                const float e1 = dot3(q[8], q[9], q[10], px, py, pz) - q[11];
                const float e2 = dot3(q[12], q[13], q[14], px, py, pz) - q[15];
                const bool h = (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (1e-3f <= t) & (t <= tl);
                if (open[p] && h) open[p] = false;
                still |= open[p];
              }
              if (!__any(still)) return false;
            }
            return true;
          });

        const float lcr = sc.light_rgb[3 * li], lcg = sc.light_rgb[3 * li + 1], lcb = sc.light_rgb[3 * li + 2];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if (need[p] && valid[p]) { n_shadow++; if (!open[p]) n_occl++; }
          if (open[p]) {
            // light_bounce with the LIGHT's normal, as the reference passes it (core/light.cuh:164-168)
            const float dn = dot3(sdx[ix(p)], sdy[ix(p)], sdz[ix(p)], lnx[ix(p)], lny[ix(p)], lnz[ix(p)]);
            float cr, cg, cb;
            if (sc.mat_kind[best_obj[p]] == 0) { // DiffuseMaterial::light_bounce, core/material.cuh:63-75
              const float scale = -dn;
              if (scale > 0.0f) { cr = scale * lcr; cg = scale * lcg; cb = scale * lcb; }
              else { cr = 0.0f; cg = 0.0f; cb = 0.0f; }
            } else {                              // MetallicMaterial::light_bounce, core/material.cuh:134-158
              const float s2 = 2.0f * dn;
              const float rx = sdx[ix(p)] - s2 * lnx[ix(p)], ry = sdy[ix(p)] - s2 * lny[ix(p)], rz = sdz[ix(p)] - s2 * lnz[ix(p)];
              if (dot3(rx, ry, rz, dx[p], dy[p], dz[p]) < -0.99f) { cr = lcr; cg = lcg; cb = lcb; }
              else { cr = 0.001f; cg = 0.001f; cb = 0.001f; }
            }
            if (have_light[p]) { lr[p] += cr; lg[p] += cg; lb[p] += cb; }
            else { lr[p] = cr; lg[p] = cg; lb[p] = cb; have_light[p] = true; }
          }
        }
      };
      for (int li = 0; li < sc.nl; ++li) {
        const float lvx = sc.light_vec[3 * li], lvy = sc.light_vec[3 * li + 1], lvz = sc.light_vec[3 * li + 2];
        if (sc.light_kind[li] == 0) shade_light(std::false_type{}, li, lvx, lvy, lvz);
        else shade_light(std::true_type{}, li, lvx, lvy, lvz);
      }
    }

    // ---- combine: src/main.cu:91-104 with MAX_RECURSION_DEPTH = 0 ----
#pragma unroll
    for (int p = 0; p < P; ++p) {
      float cr, cg, cb;
      if (have_hit[p]) {
        const float *m = sc.mat_rgb + 3 * best_obj[p];
        const float mr = m[0], mg = m[1], mb = m[2];
        const float inc = 0.01f; // what the depth-1 call returns, src/main.cu:51-53
        if (have_light[p]) { cr = inc * mr + lr[p] * mr; cg = inc * mg + lg[p] * mg; cb = inc * mb + lb[p] * mb; }
        else { cr = inc * mr; cg = inc * mg; cb = inc * mb; }
      } else {
        float nx = dx[p], ny = dy[p], nz = dz[p];
        normalize3(nx, ny, nz);
        const float y = 0.5f * (ny + 1.0f);
        const float w0 = 1.0f - y;
        cr = w0 * 1.0f + y * 0.5f; cg = w0 * 1.0f + y * 0.7f; cb = w0 * 1.0f + y * 1.0f;
      }
      accr[p] += cr; accg[p] += cg; accb[p] += cb; // src/main.cu:151
      out_id[p] = have_hit[p] ? best_obj[p] : -1;
      out_t[p] = best_t[p];
    }
  }

  // ---- framebuffer: src/main.cu:155 (x / 1.0f == x, so the division is skipped for one sample) ----
  if (nsamp > 1) {
    const float fn = float(nsamp);
#pragma unroll
    for (int p = 0; p < P; ++p) { accr[p] = accr[p] / fn; accg[p] = accg[p] / fn; accb[p] = accb[p] / fn; }
  }
  const size_t pix0 = size_t(A.frame_layout ? grow : lrow) * A.W + col0;
  if (P > 1 && A.wide && valid_row && col0 + P <= A.W) {
    // the lane's 3P floats / P ints / P floats are contiguous and 4P-byte aligned (launcher checks W and pointers)
    using VF = typename CVec<P>::f;
    using VI = typename CVec<P>::i;
    float o[3 * P];
#pragma unroll
    for (int p = 0; p < P; ++p) { o[3 * p] = accr[p]; o[3 * p + 1] = accg[p]; o[3 * p + 2] = accb[p]; }
    VF *dst = reinterpret_cast<VF *>(A.rgb + 3 * pix0);
#pragma unroll
    for (int k = 0; k < 3; ++k) dst[k] = *reinterpret_cast<const VF *>(&o[P * k]);
    if (A.id) *reinterpret_cast<VI *>(A.id + pix0) = *reinterpret_cast<const VI *>(&out_id[0]);
    if (A.t) *reinterpret_cast<VF *>(A.t + pix0) = *reinterpret_cast<const VF *>(&out_t[0]);
  } else {
#pragma unroll
    for (int p = 0; p < P; ++p)
      if (valid[p]) {
        A.rgb[3 * (pix0 + p) + 0] = accr[p];
        A.rgb[3 * (pix0 + p) + 1] = accg[p];
        A.rgb[3 * (pix0 + p) + 2] = accb[p];
        if (A.id) A.id[pix0 + p] = out_id[p];
        if (A.t) A.t[pix0 + p] = out_t[p];
      }
  }

  if (A.stats) {
    // counters are spread over stats_slots copies: same-address atomics cost ~12 ns each, one after the other
    unsigned long long *st = A.stats + 8u * ((blockIdx.x + 7u * blockIdx.y) & (unsigned)(A.stats_slots - 1));
    unsigned int nvalid = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) nvalid += valid[p] ? 1u : 0u;
    unsigned int v4[4] = {nvalid * (unsigned)(MULTI ? A.spp : 1), n_shadow, n_hit, n_occl};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned int x = v4[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0 && x) atomicAdd(&st[k], (unsigned long long)x);
    }
    if (lane == 0) {
      atomicAdd(&st[4], (unsigned long long)n_exact);
      atomicAdd(&st[5], (unsigned long long)n_clusters);
    }
  }
}

template <bool HIER, int P>
static hipError_t launch_cull(RenderArgs &A, hipStream_t stream)
{
  // largest offset of a ray of the padded tile (pixels -1..8P horizontally, -1..8 vertically) from the tile's
  // centre ray, in double, rounded up
  {
    const double hw = (4.0 * P + 0.5) / A.W, hh = 4.5 / A.H;
    const double hx[3] = {hw * A.cam.hx, hw * A.cam.hy, hw * A.cam.hz};
    const double hy[3] = {hh * A.cam.vx, hh * A.cam.vy, hh * A.cam.vz};
    double p = 0, m = 0;
    for (int k = 0; k < 3; ++k) { p += (hx[k] + hy[k]) * (hx[k] + hy[k]); m += (hx[k] - hy[k]) * (hx[k] - hy[k]); }
    A.cone_ext = float(std::sqrt(p > m ? p : m) * 1.001) + 2.0f * camera_rounding(A.cam); // the ray's and the centre's
  }
  const uintptr_t align = 4 * P - 1;
  A.wide = (A.W % P == 0) && ((uintptr_t)A.rgb & align) == 0 && ((uintptr_t)A.id & align) == 0 &&
           ((uintptr_t)A.t & align) == 0;
  const int tiles_x = (A.W + 8 * P - 1) / (8 * P), tiles_y = (A.local_rows + 7) / 8;
  if (tiles_x == 0 || tiles_y == 0) return hipSuccess;
  // waves per workgroup: waves share nothing; 2 measured best on C3 and C5 (413 / 1561 us against 444 / 1625 with 4)
  static const int wpw = getenv("MRT_CULL_WPW") ? atoi(getenv("MRT_CULL_WPW")) : 2;
  const dim3 grid((tiles_x + wpw - 1) / wpw, tiles_y);
  if (A.spp > 1) hipLaunchKernelGGL((render_cull_kernel<HIER, P, true>), grid, dim3(64 * wpw), 0, stream, A);
  else hipLaunchKernelGGL((render_cull_kernel<HIER, P, false>), grid, dim3(64 * wpw), 0, stream, A);
  return hipGetLastError();
}

hipError_t launch_render_cull(const RenderArgs &in, hipStream_t stream)
{
  RenderArgs A = in;
  // pixels per lane.  Measured: with a hierarchy to walk, one pixel per lane (72 registers, 7 waves
  // per SIMD) beats two on every large config but C3, where the two are even; without one, two win.
  // MRT_CULL_P = 1 | 2 overrides (ablation).
  static const int forced = [] {
    const char *e = getenv("MRT_CULL_P");
    const int x = e ? atoi(e) : 0;
    return (x == 1 || x == 2) ? x : 0;
  }();
  const bool hier = A.sc.ncs > 1 || A.sc.nct > 1;
  const int P = forced ? forced : (hier ? 1 : 2);
  if (hier) return P == 1 ? launch_cull<true, 1>(A, stream) : launch_cull<true, 2>(A, stream);
  return P == 1 ? launch_cull<false, 1>(A, stream) : launch_cull<false, 2>(A, stream);
}

} // namespace mrt
