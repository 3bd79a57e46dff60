// mrt_small.hip -- render kernel for scenes of at most 64 primitives (the reference's own scene has 10).
//
// Same results as the other render kernels (same fp32 operations in the same order for everything that
// reaches a buffer); what differs is how the work is laid over a wave64:
//
//   * a wave renders a tile of 8P x 8 pixels, each lane P horizontally adjacent pixels (a lane's outputs are
//     contiguous: 3P + P + P floats, vector stores).  Everything uniform over the tile -- bound tests, record
//     fetches, loop control -- is paid once per 8P x 8 pixels;
//   * lane k also OWNS primitive k (spheres first, then triangles): it keeps the ray-independent terms of the
//     exact tests (o - c, |o - c|^2 - r^2, k - n.o) in registers and tests the primitive's bound against the
//     tile; the survivors' terms are broadcast with v_readlane -- no memory access in the sphere loops, one
//     64-byte scalar load per surviving triangle;
//   * spheres are bounded with the cone / swept-ball tests of mrt_cull_common.h, triangles with SCREEN-SPACE
//     EDGE FUNCTIONS.  For a triangle (plane n, k; edge planes m_i, c_i) and a ray o + t d accepted by the
//     exact test (t > 0), e_i(p) = (w_i . d) / (n . d) with w_i = (m_i.o - c_i) n + (k - n.o) m_i, and
//     sign(n.d) = sign(k - n.o); w_i . d is affine in the pixel coordinates, so its maximum over the tile is
//     its value at the tile centre plus a constant.  If that maximum is below a margin for one edge, no ray
//     of the tile can pass that edge test and the triangle is skipped.  The functions depend on the camera:
//     the launcher computes them in double for every render (small_primary_functions) and they travel in the
//     kernel arguments.  The same construction bounds shadow rays: e_i at the point where the shadow ray of
//     hit point p meets the plane is affine in p, for directional and for point lights, and is bounded over
//     the ball around the tile's hit points; these functions depend on scene and lights only and are built
//     at upload (mrt_scene_build.h, shadow_functions);
//   * u = col / W and v = row / H come from two tables computed once per frame size with the same IEEE
//     division (uv_table_kernel) instead of 1 + P divisions per lane.
//
// Culling never changes a result: every comparison carries a margin of 1e-4 of the magnitudes involved
// (the exact tests round at 6e-8), NaN never culls, and tests/test_gpu_parity.py compares the buffers bit
// for bit with the brute-force kernel and the oracle.
//
// Reference semantics: src/main.cu:60-104,122-157 (render / color_ray with MAX_RECURSION_DEPTH = 0),
// core/sphere.cuh:32-73, core/triangle.cuh:77-103, core/light.cuh:135-178, core/material.cuh:63-75,134-158.

#include "mrt_internal.h"
#include "mrt_device.h"
#include "mrt_cull_common.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace mrt {

__global__ void uv_table_kernel(float *utab, int nu, int W, float *vtab, int nv, int H)
{
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nu) utab[i] = float(i) / float(W); // src/main.cu:136
  if (i < nv) vtab[i] = float(i) / float(H); // src/main.cu:137
}

hipError_t launch_uv_table(float *utab, int nu, int W, float *vtab, int nv, int H, hipStream_t stream)
{
  const int n = nu > nv ? nu : nv;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(uv_table_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, utab, nu, W, vtab, nv, H);
  return hipGetLastError();
}

constexpr int SH_STRIDE = 3 * 7 + 2; // floats per (light, triangle): 3 edges x (gx, gy, gz, off, gn, a, b) + sign guard

template <int P> struct VecOf;
typedef float nfloat2 __attribute__((ext_vector_type(2)));
typedef float nfloat4 __attribute__((ext_vector_type(4)));
typedef int nint2 __attribute__((ext_vector_type(2)));
typedef int nint4 __attribute__((ext_vector_type(4)));
template <> struct VecOf<1> { using f = float; using i = int; using nf = float; using ni = int; };
template <> struct VecOf<2> { using f = float2; using i = int2; using nf = nfloat2; using ni = nint2; };
template <> struct VecOf<4> { using f = float4; using i = int4; using nf = nfloat4; using ni = nint4; };

__device__ __forceinline__ float abs1(float x, float y, float z) { return fabsf(x) + fabsf(y) + fabsf(z); }

// SmallArgs (mrt_internal.h): the common kernel arguments plus prm[field][lane], read straight from the argument
// segment, one memory round trip after launch.

template <int P, bool MULTI>
__device__ __forceinline__ void render_small_body(const SmallArgs &S)
{
  const RenderArgs &A = S.A;
  const DevScene &sc = A.sc;
  const DevCamera &cam = A.cam;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tx = lane & 7, ty = lane >> 3;
  const int ns = sc.ns, nt = sc.nt; // ns + nt <= 64 (launcher)
  constexpr int TW = 8 * P;         // tile width in pixels

  const int tile_x = blockIdx.x * (blockDim.x >> 6) + wave;
  const int tile_y = blockIdx.y;
  if (tile_x * TW >= A.W) return; // waves are independent: no barrier anywhere below
  const int col0 = tile_x * TW + tx * P;
  const int lrow = tile_y * 8 + ty;
  const int lrow0 = tile_y * 8;
  const int strip = lrow0 / A.strip_rows; // strips are multiples of 8 rows: a tile never straddles two
  const int grow0 = (strip * A.nranks + A.rank) * A.strip_rows + (lrow0 - strip * A.strip_rows);
  const int grow = grow0 + ty;
  const bool valid_row = (lrow < A.local_rows) && (grow < A.H);

  // ---- cone around the tile's rays, padded by one pixel on every side (covers the 121-spp offsets): every
  //      ray of the padded tile is the centre ray plus an offset of norm <= cone_ext (launcher) ----
  float cax, cay, caz, cosT, sinT;
  const float rw = fast_rcp(float(A.W)), rh = fast_rcp(float(A.H)); // bound arithmetic: approximate is fine
  const float um = (float(tile_x * TW) + 0.5f * float(TW - 1)) * rw, vm = (float(grow0) + 3.5f) * rh;
  {
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

  // ---- the primitive this lane owns: bound, and the ray-independent terms of the exact primary tests ----
  const bool own_s = lane < ns, own_t = (lane >= ns) && (lane < ns + nt);
  float bx = 0.0f, by = 0.0f, bz = 0.0f, brad = 0.0f; // sphere centre + radius / triangle bound ball
  float g0 = 0.0f, g1 = 0.0f, g2 = 0.0f;               // sphere centre / triangle plane normal
  float e0 = 0.0f, e1 = 0.0f, e2 = 0.0f, e3 = 0.0f;    // sphere: o - c, |o - c|^2 - r^2;  triangle: e0 = k - n.o
  float rr = 0.0f;                                     // sphere: r^2
  int pobj = 0;
  bool pass_primary = false;
  if (own_s) {
    pass_primary = ball_may_touch_cone(S.prm[0][lane], S.prm[1][lane], S.prm[2][lane], S.prm[3][lane], cax, cay, caz,
                                       cosT, sinT);
    e0 = S.prm[4][lane]; e1 = S.prm[5][lane]; e2 = S.prm[6][lane]; e3 = S.prm[7][lane];
    pobj = __float_as_int(S.prm[8][lane]);
    // for the shadow rays and the hit record (not needed before the first exact test)
    bx = sc.sph_cx[lane]; by = sc.sph_cy[lane]; bz = sc.sph_cz[lane]; brad = sc.sph_r[lane];
    g0 = bx; g1 = by; g2 = bz;
    rr = brad * brad;
  } else if (own_t) {
    const int k = lane - ns;
    bool cull = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float f = __builtin_fmaf(um, S.prm[3 * i + 1][lane], __builtin_fmaf(vm, S.prm[3 * i + 2][lane], S.prm[3 * i][lane]));
      cull |= (f < 0.0f); // NaN compares false: never culls
    }
    pass_primary = !cull;
    e0 = S.prm[12][lane];
    pobj = __float_as_int(S.prm[13][lane]);
    bx = sc.tri_bx[k]; by = sc.tri_by[k]; bz = sc.tri_bz[k]; brad = sc.tri_br[k];
    g0 = sc.tri[k]; g1 = sc.tri[nt + k]; g2 = sc.tri[2 * nt + k];
  }
  const unsigned long long smask_all = ns >= 64 ? ~0ull : ((1ull << ns) - 1ull);
  const unsigned long long primary_mask = __ballot(pass_primary);

  float accr[P], accg[P], accb[P];
  int out_id[P];
  float out_t[P];
  bool valid[P];
#pragma unroll
  for (int p = 0; p < P; ++p) {
    accr[p] = 0.0f; accg[p] = 0.0f; accb[p] = 0.0f;
    out_id[p] = -1; out_t[p] = MRT_INF;
    valid[p] = valid_row && (col0 + p < A.W);
  }
  unsigned int n_shadow = 0, n_hit = 0, n_occl = 0;
  unsigned int n_exact = 0, n_clusters = 0; // wave-uniform diagnostics (lane 0 reports them)
  const int nsamp = MULTI ? A.spp : 1; // MULTI = false: one sample, no accumulation across samples
  const float DIVW = 13.0f, DIVH = 13.0f;

  // u, v of the lane's pixels: tables hold float(i) / float(W) and float(i) / float(H) (uv_table_kernel)
  float u0[P];
  {
    const typename VecOf<P>::f uu = *reinterpret_cast<const typename VecOf<P>::f *>(A.utab + col0);
    const float *up = reinterpret_cast<const float *>(&uu);
#pragma unroll
    for (int p = 0; p < P; ++p) u0[p] = up[p];
  }
  const float v = A.vtab[grow < A.H ? grow : A.H - 1]; // rows beyond the frame are never written

  for (int sidx = 0; sidx < nsamp; ++sidx) {
    float offx = 0.0f, offy = 0.0f;
    if (nsamp > 1) { // pipeline/multisample_mesh.cuh:23-30
      int r = sidx / 11, c = sidx - r * 11;
      offy = (-1.0f + 2.0f * float(r + 1) / DIVH) / float(A.H);
      offx = (-1.0f + 2.0f * float(c + 1) / DIVW) / float(A.W);
    }
    // ---- primary rays: pipeline/viewport.cuh:30-34 ----
    float dx[P], dy[P], dz[P], aa[P], best_t[P];
    int best_obj[P], best_k[P];
    const float sv = nsamp > 1 ? v + offy : v;
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const float su = nsamp > 1 ? u0[p] + offx : u0[p];
      dx[p] = ((cam.ux + su * cam.hx) + sv * cam.vx) - cam.ox;
      dy[p] = ((cam.uy + su * cam.hy) + sv * cam.vy) - cam.oy;
      dz[p] = ((cam.uz + su * cam.hz) + sv * cam.vz) - cam.oz;
      aa[p] = dot3(dx[p], dy[p], dz[p], dx[p], dy[p], dz[p]);
      best_t[p] = MRT_INF; best_obj[p] = 0x7fffffff; best_k[p] = -1;
    }

    // ---- closest hit over the surviving spheres: src/main.cu:60-77, core/sphere.cuh:32-73, [0.01, inf] ----
    {
      unsigned long long mask = primary_mask & smask_all;
      n_clusters++; n_exact += __popcll(primary_mask);
      while (mask) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1;
        const float ocx = bcast(e0, j), ocy = bcast(e1, j), ocz = bcast(e2, j), c = bcast(e3, j);
        const int obj = __builtin_amdgcn_readlane(pobj, j);
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float b = dot3(ocx, ocy, ocz, dx[p], dy[p], dz[p]);
          const float disc = b * b - aa[p] * c;
          if (disc >= 0.0f) {
            float t;
            if (sphere_roots(aa[p], b, disc, 0.01f, MRT_INF, t)) {
              if (t < best_t[p] || (t == best_t[p] && t < MRT_INF && obj < best_obj[p])) {
                best_t[p] = t; best_obj[p] = obj; best_k[p] = j;
              }
            }
          }
        }
      }
    }
    // ---- ... and triangles: core/triangle.cuh:77-103 ----
    {
      unsigned long long mask = ns >= 64 ? 0ull : (primary_mask >> ns);
      while (mask) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1;
        const float *q = sc.tri_rec + 16 * j; // wave-uniform address: one s_load_dwordx16
        const float pnx = q[0], pny = q[1], pnz = q[2];
        const float num = bcast(e0, ns + j);
        const int obj = __builtin_amdgcn_readlane(pobj, ns + j);
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float rdn = dot3(dx[p], dy[p], dz[p], pnx, pny, pnz);
          const float t = num / rdn;
          // the edge tests only matter in lanes whose t is in range and could still win: none in the wave, skip
          if (!__any((rdn != 0.0f) & (0.01f <= t) & (t <= best_t[p]))) continue;
          const float px = cam.ox + t * dx[p], py = cam.oy + t * dy[p], pz = cam.oz + t * dz[p];
          const float d0 = dot3(q[4], q[5], q[6], px, py, pz) - q[7];
          const float d1 = dot3(q[8], q[9], q[10], px, py, pz) - q[11];
          const float d2 = dot3(q[12], q[13], q[14], px, py, pz) - q[15];
          const bool hit = (rdn != 0.0f) & (d0 > -1e-6f) & (d1 > -1e-6f) & (d2 > -1e-6f) & (0.01f <= t) & (t <= MRT_INF);
          if (hit) {
            if (t < best_t[p] || (t == best_t[p] && t < MRT_INF && obj < best_obj[p])) {
              best_t[p] = t; best_obj[p] = obj; best_k[p] = ns + j;
            }
          }
        }
      }
    }

    // ---- hit records of the winners (core/sphere.cuh:45-47, core/triangle.cuh:96-97); the winner's centre /
    //      plane normal comes from the owning lane's registers ----
    bool have_hit[P], any_hit_lane = false;
    float hpx[P], hpy[P], hpz[P], hnx[P], hny[P], hnz[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
      have_hit[p] = best_k[p] >= 0;
      any_hit_lane |= have_hit[p];
      hpx[p] = 0.0f; hpy[p] = 0.0f; hpz[p] = 0.0f; hnx[p] = 0.0f; hny[p] = 0.0f; hnz[p] = 0.0f;
    }
    const bool tile_hit = __any(any_hit_lane);
    if (tile_hit) {
#pragma unroll
      for (int p = 0; p < P; ++p) {
        const int k = have_hit[p] ? best_k[p] : lane;
        const float w0 = __shfl(g0, k, 64), w1 = __shfl(g1, k, 64), w2 = __shfl(g2, k, 64);
        if (have_hit[p]) {
          hpx[p] = cam.ox + best_t[p] * dx[p]; hpy[p] = cam.oy + best_t[p] * dy[p]; hpz[p] = cam.oz + best_t[p] * dz[p];
          if (best_k[p] < ns) {
            float nx = hpx[p] - w0, ny = hpy[p] - w1, nz = hpz[p] - w2;
            normalize3(nx, ny, nz);
            hnx[p] = nx; hny[p] = ny; hnz[p] = nz;
          } else {
            const bool front = dot3(dx[p], dy[p], dz[p], w0, w1, w2) < 0;
            hnx[p] = front ? w0 : -w0; hny[p] = front ? w1 : -w1; hnz[p] = front ? w2 : -w2;
          }
          if (valid[p]) n_hit++;
        }
      }
    }

    // ---- shadow rays + direct light: core/light.cuh:135-178 ----
    float lr[P], lg[P], lb[P];
    bool have_light[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { lr[p] = 0.0f; lg[p] = 0.0f; lb[p] = 0.0f; have_light[p] = false; }
    if (tile_hit) {
      // ball around the tile's hit points: centre = the first hit point found, radius = the farthest one
      Bundle B;
      {
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
        float e2max = 0.0f;
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float ex = hpx[p] - mx, ey = hpy[p] - my, ez = hpz[p] - mz;
          e2max = fmaxf(e2max, have_hit[p] ? dot3(ex, ey, ez, ex, ey, ez) : 0.0f);
        }
        B.rho = fast_sqrt(wave_max(e2max)) * CULL_REL + CULL_ABS;
      }
      // distance bound from the bundle centre for the sphere this lane owns (>= |hit point - centre|)
      float Lm2 = 0.0f;
      if (own_s) {
        const float wx = bx - B.mx, wy = by - B.my, wz = bz - B.mz;
        const float Lm = fast_sqrt(dot3(wx, wy, wz, wx, wy, wz)) * CULL_REL + B.rho;
        Lm2 = Lm * Lm;
      }
      const float reff_shadow = own_s ? sphere_cull_radius(rr, Lm2) : brad;
      const float Pmax = abs1(B.mx, B.my, B.mz) + B.rho; // >= |hit point|_inf

      // One light.  POINT is a compile-time switch: for a directional light the shadow ray direction, its
      // squared length and the light's normal are wave-uniform scalars, for a point light per-pixel registers.
      auto shade_light = [&](auto point_tag, int li, float lvx, float lvy, float lvz) {
        constexpr bool POINT = decltype(point_tag)::value;
        constexpr int Q = POINT ? P : 1;
        auto ix = [](int p) { return POINT ? p : 0; };
        // shadow ray and the light's "hit record" (ray_from: core/light.cuh:73-81 / :47-53)
        float sdx[Q], sdy[Q], sdz[Q], lnx[Q], lny[Q], lnz[Q], sa[Q];
        float tl;
        bool need[P], open[P], any_open = false;
        if (!POINT) {
          tl = MRT_INF;
          B.point = false;
          B.Dx = -lvx; B.Dy = -lvy; B.Dz = -lvz;
          sdx[0] = -lvx; sdy[0] = -lvy; sdz[0] = -lvz;
          lnx[0] = lvx; lny[0] = lvy; lnz[0] = lvz;
        } else {
          tl = 1.0f;
          B.point = true;
          B.Dx = lvx - B.mx; B.Dy = lvy - B.my; B.Dz = lvz - B.mz;
#pragma unroll
          for (int p = 0; p < Q; ++p) {
            sdx[p] = lvx - hpx[p]; sdy[p] = lvy - hpy[p]; sdz[p] = lvz - hpz[p];
            float nx = hpx[p] - lvx, ny = hpy[p] - lvy, nz = hpz[p] - lvz;
            normalize3(nx, ny, nz);
            lnx[p] = nx; lny[p] = ny; lnz[p] = nz;
          }
        }
        B.DD = dot3(B.Dx, B.Dy, B.Dz, B.Dx, B.Dy, B.Dz);
#pragma unroll
        for (int q = 0; q < Q; ++q) sa[q] = dot3(sdx[q], sdy[q], sdz[q], sdx[q], sdy[q], sdz[q]);
#pragma unroll
        for (int p = 0; p < P; ++p) {
          // the light must not shine through the surface, core/light.cuh:148
          need[p] = have_hit[p] && !(dot3(sdx[ix(p)], sdy[ix(p)], sdz[ix(p)], hnx[p], hny[p], hnz[p]) < 0);
          open[p] = need[p];
          any_open |= open[p];
        }
        if (__any(any_open)) {
          // which primitives can touch one of the tile's shadow rays (interval [1e-3, t_light], core/light.cuh:28-33)
          bool pass = (own_s || own_t) && bundle_may_touch(B, bx, by, bz, reff_shadow);
          if (own_t && pass && li < sc.nls) {
            // shadow functions of (light li, this triangle), built at upload: [light][23][triangle]
            const float *fn = sc.tri_shfn + (size_t)li * SH_STRIDE * nt + (lane - ns);
            auto F = [&](int f) { return fn[(size_t)f * nt]; };
            const float scale = POINT ? abs1(lvx, lvy, lvz) + Pmax : 1.0f;
            bool cull = false;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              const float h = __builtin_fmaf(F(7 * i), B.mx, __builtin_fmaf(F(7 * i + 1), B.my,
                              __builtin_fmaf(F(7 * i + 2), B.mz, F(7 * i + 3))));
              const float margin = scale * __builtin_fmaf(F(7 * i + 6), Pmax, F(7 * i + 5));
              cull |= (__builtin_fmaf(B.rho, F(7 * i + 4), h) + margin < 0.0f); // NaN never culls
            }
            // point light (almost) in the triangle's plane: the sign argument of the functions fails
            const bool sign_ok = F(21) > F(22) * Pmax;
            if (cull && sign_ok) pass = false;
          }
          const unsigned long long shadow_mask = __ballot(pass);
          n_clusters++; n_exact += __popcll(shadow_mask);
          unsigned long long mask = shadow_mask & smask_all;
          bool go = true;
          while (mask && go) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float scx = bcast(g0, j), scy = bcast(g1, j), scz = bcast(g2, j), srr = bcast(rr, j);
            bool still = false;
#pragma unroll
            for (int p = 0; p < P; ++p) {
              const float ocx = hpx[p] - scx, ocy = hpy[p] - scy, ocz = hpz[p] - scz;
              const float b = dot3(ocx, ocy, ocz, sdx[ix(p)], sdy[ix(p)], sdz[ix(p)]);
              const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - srr;
              const float disc = b * b - sa[ix(p)] * c;
              if (open[p] && disc >= 0.0f) {
                float t;
                if (sphere_roots(sa[ix(p)], b, disc, 1e-3f, tl, t)) open[p] = false;
              }
              still |= open[p];
            }
            go = __any(still); // every shadow ray of the tile is blocked: stop
          }
          mask = (go && ns < 64) ? (shadow_mask >> ns) : 0ull;
          while (mask && go) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const float *q = sc.tri_rec + 16 * j;
            const float pnx = q[0], pny = q[1], pnz = q[2], kp = q[3];
            bool still = false;
#pragma unroll
            for (int p = 0; p < P; ++p) {
              const float rdn = dot3(sdx[ix(p)], sdy[ix(p)], sdz[ix(p)], pnx, pny, pnz);
              const float t = (kp - dot3(hpx[p], hpy[p], hpz[p], pnx, pny, pnz)) / rdn;
              if (!__any(open[p] & (rdn != 0.0f) & (1e-3f <= t) & (t <= tl))) { still |= open[p]; continue; }
              const float px = hpx[p] + t * sdx[ix(p)], py = hpy[p] + t * sdy[ix(p)], pz = hpz[p] + t * sdz[ix(p)];
              const float d0 = dot3(q[4], q[5], q[6], px, py, pz) - q[7];
              const float d1 = dot3(q[8], q[9], q[10], px, py, pz) - q[11];
              const float d2 = dot3(q[12], q[13], q[14], px, py, pz) - q[15];
              const bool h = (rdn != 0.0f) & (d0 > -1e-6f) & (d1 > -1e-6f) & (d2 > -1e-6f) & (1e-3f <= t) & (t <= tl);
              if (open[p] && h) open[p] = false;
              still |= open[p];
            }
            go = __any(still);
          }
        }

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
  if (A.wide && valid_row && col0 + P <= A.W) {
    // the lane's 3P floats / P ints / P floats are contiguous and 4P-byte aligned (launcher checks W and pointers)
    using VF = typename VecOf<P>::f;
    float o[3 * P];
#pragma unroll
    for (int p = 0; p < P; ++p) { o[3 * p] = accr[p]; o[3 * p + 1] = accg[p]; o[3 * p + 2] = accb[p]; }
    VF *dst = reinterpret_cast<VF *>(A.rgb + 3 * pix0);
    // the frame is written once and not read again by this kernel: streaming (non-temporal) stores
    using NF = typename VecOf<P>::nf;
    using NI = typename VecOf<P>::ni;
#pragma unroll
    for (int k = 0; k < 3; ++k) __builtin_nontemporal_store(*reinterpret_cast<const NF *>(&o[P * k]), reinterpret_cast<NF *>(dst) + k);
    if (A.id) __builtin_nontemporal_store(*reinterpret_cast<const NI *>(&out_id[0]), reinterpret_cast<NI *>(A.id + pix0));
    if (A.t) __builtin_nontemporal_store(*reinterpret_cast<const NF *>(&out_t[0]), reinterpret_cast<NF *>(A.t + pix0));
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
    unsigned int v4[4] = {nvalid * (unsigned)nsamp, n_shadow, n_hit, n_occl};
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

// Two pixels per lane and one sample fit 96 registers (five waves per SIMD) without spilling when asked to;
// left alone the compiler takes 100 and loses a wave (measured: 23.2 -> 22.0 us on the default frame).
template <int P, bool MULTI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(P == 2 && !MULTI ? 5 : 1, 8)))
void render_small_kernel(SmallArgs S) { render_small_body<P, MULTI>(S); }

// ---- primary-ray functions (host, double; per render) ---------------------------------------------------
// For triangle j (16 floats at tri16 + 16 j) and an accepted ray, sign(n.d) = sign(num) = s with
// num = k - n.o as the kernels compute it in fp32, and e_i(p) |n.d| = s w_i . d.  d(u, v) = base + u h + v vv
// is affine, so over the padded tile around (um, vm):  max s w_i . d = s w_i . d(um, vm) + |Wh| hu + |Wv| hv.
// Stored: W0' = s w_i . base + that extent + margin, Wh, Wv.  Margin: e_i > -1e-6 is the exact test, and the
// fp32 rounding of the exact test and of the function's evaluation stay below 1e-4 of the magnitudes in K.
static void small_primary_functions(const RenderArgs &A, const SmallHost &hs, double hu, double hv, double dmax,
                                    float (*prm)[64])
{
  const int ns = hs.ns, nt = hs.nt;
  const float *tri16 = hs.tri16;
  // spheres: the terms of the exact test (fp32, the kernels' operation order) and the cone-test inputs
  for (int i = 0; i < ns; ++i) {
    const float cx = hs.sph[4 * i], cy = hs.sph[4 * i + 1], cz = hs.sph[4 * i + 2], r = hs.sph[4 * i + 3];
    const float e0 = A.cam.ox - cx, e1 = A.cam.oy - cy, e2 = A.cam.oz - cz, rr = r * r;
    const float cc = (e0 * e0 + (e1 * e1 + e2 * e2)) - rr;
    const float wx = cx - A.cam.ox, wy = cy - A.cam.oy, wz = cz - A.cam.oz;
    const double L2 = (double)wx * wx + (double)wy * wy + (double)wz * wz;
    prm[0][i] = wx; prm[1][i] = wy; prm[2][i] = wz;
    prm[3][i] = (float)((std::sqrt((double)rr + 2e-6 * L2) * 1.001 + 1e-6) * (1.0 + 1e-6)); // sphere_cull_radius, rounded up
    prm[4][i] = e0; prm[5][i] = e1; prm[6][i] = e2; prm[7][i] = cc;
    std::memcpy(&prm[8][i], &hs.sph_obj[i], 4);
  }
  const double o[3] = {A.cam.ox, A.cam.oy, A.cam.oz};
  const double h[3] = {A.cam.hx, A.cam.hy, A.cam.hz}, vv[3] = {A.cam.vx, A.cam.vy, A.cam.vz};
  const double base[3] = {(double)A.cam.ux - o[0], (double)A.cam.uy - o[1], (double)A.cam.uz - o[2]};
  auto dot = [](const double *a, const double *b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; };
  auto n1f = [](const double *a) { return std::fabs(a[0]) + std::fabs(a[1]) + std::fabs(a[2]); };
  const double o1 = n1f(o);
  const double cam_delta = 2.0 * camera_rounding(A.cam);
  for (int j = 0; j < nt; ++j) {
    const float *q = tri16 + 16 * j;
    float (*pri)[64] = prm;
    const int col = ns + j; // the lane that owns triangle j
    const double n[3] = {q[0], q[1], q[2]}, kp = q[3];
    // the sign the kernel sees: same fp32 expression as the exact test's numerator
    const float numf = q[3] - (A.cam.ox * q[0] + (A.cam.oy * q[1] + A.cam.oz * q[2]));
    const double s = numf >= 0.0f ? 1.0 : -1.0;
    const double num = kp - dot(n, o), nn1 = n1f(n);
    for (int i = 0; i < 4; ++i) {
      double w[3], K;
      if (i < 3) {
        const double m[3] = {q[4 + 4 * i], q[5 + 4 * i], q[6 + 4 * i]}, c = q[7 + 4 * i];
        const double Ai = dot(m, o) - c, m1 = n1f(m);
        for (int a = 0; a < 3; ++a) w[a] = s * (Ai * n[a] + num * m[a]);
        K = 1e-6 * nn1 + 1e-4 * ((m1 * o1 + std::fabs(c)) * nn1 + (std::fabs(num) + std::fabs((double)numf)) * m1);
      } else { // plane side: s (n . d) < 0 on the whole tile means t < 0 for every ray
        for (int a = 0; a < 3; ++a) w[a] = s * n[a];
        K = 1e-5 * nn1;
      }
      const double Wh = dot(w, h), Wv = dot(w, vv), W0 = dot(w, base);
      const double ext = std::fabs(Wh) * hu + std::fabs(Wv) * hv;
      // + the rounding of the ray direction itself (camera_rounding), seen through w
      const double W0p = W0 + ext + dmax * K + n1f(w) * cam_delta + 1e-6 * (std::fabs(W0) + ext); // last term: float rounding below
      pri[3 * i + 0][col] = (float)W0p;
      pri[3 * i + 1][col] = (float)Wh;
      pri[3 * i + 2][col] = (float)Wv;
    }
    prm[12][col] = numf;
    std::memcpy(&prm[13][col], &hs.tri_obj[j], 4);
  }
}

// Camera-dependent kernel arguments of a small scene for tiles of 8P x 8 pixels: cone extent and prm.
void small_prepare_args(const RenderArgs &in, const SmallHost &hs, int P, SmallArgs &S)
{
  S.A = in;
  RenderArgs &A = S.A;
  // the padded tile (pixels -1..8P horizontally, -1..8 vertically) around its centre, in u / v units
  const double hw = (4.0 * P + 0.5) / A.W, hh = 4.5 / A.H;
  double dmax = 0;
  {
    // largest offset of a ray of the padded tile from the tile's centre ray, in double, rounded up
    const double hx[3] = {hw * A.cam.hx, hw * A.cam.hy, hw * A.cam.hz};
    const double hy[3] = {hh * A.cam.vx, hh * A.cam.vy, hh * A.cam.vz};
    double p = 0, m = 0;
    for (int k = 0; k < 3; ++k) { p += (hx[k] + hy[k]) * (hx[k] + hy[k]); m += (hx[k] - hy[k]) * (hx[k] - hy[k]); }
    A.cone_ext = float(std::sqrt(p > m ? p : m) * 1.001) + 2.0f * camera_rounding(A.cam); // the ray's and the centre's
    // largest 1-norm of a ray direction over the padded frame: |d|_1 is convex in (u, v), so at a corner
    for (int c = 0; c < 4; ++c) {
      const double u = (c & 1) ? 1.0 + hw : -hw, v = (c & 2) ? 1.0 + hh : -hh;
      const double d1 = std::fabs(A.cam.ux + u * A.cam.hx + v * A.cam.vx - A.cam.ox) +
                        std::fabs(A.cam.uy + u * A.cam.hy + v * A.cam.vy - A.cam.oy) +
                        std::fabs(A.cam.uz + u * A.cam.hz + v * A.cam.vz - A.cam.oz);
      if (!(d1 <= dmax)) dmax = d1; // a NaN camera gives NaN functions, which never cull
    }
    dmax *= 1.001;
  }
  small_primary_functions(A, hs, hw * 1.0001, hh * 1.0001, dmax, S.prm);
}

template <int P>
static hipError_t launch_small(const RenderArgs &in, const SmallHost &hs, hipStream_t stream)
{
  SmallArgs S; // 3.5 KB of kernel arguments
  small_prepare_args(in, hs, P, S);
  RenderArgs &A = S.A;
  // vector stores need every lane's first pixel on a 4P-byte boundary in all three buffers
  const uintptr_t align = 4 * P - 1;
  A.wide = (A.W % P == 0) && ((uintptr_t)A.rgb & align) == 0 && ((uintptr_t)A.id & align) == 0 &&
           ((uintptr_t)A.t & align) == 0;
  const int tiles_x = (A.W + 8 * P - 1) / (8 * P), tiles_y = (A.local_rows + 7) / 8;
  if (tiles_x == 0 || tiles_y == 0) return hipSuccess;
  // one wave per workgroup: waves share nothing, and single-wave workgroups fill freed wave slots sooner than
  // groups of four (measured 19.4 against 19.9 us on the default frame)
  static const int wpw = getenv("MRT_SMALL_WPW") ? atoi(getenv("MRT_SMALL_WPW")) : 1;
  const dim3 grid((tiles_x + wpw - 1) / wpw, tiles_y);
  if (A.spp > 1) hipLaunchKernelGGL((render_small_kernel<P, true>), grid, dim3(64 * wpw), 0, stream, S);
  else hipLaunchKernelGGL((render_small_kernel<P, false>), grid, dim3(64 * wpw), 0, stream, S);
  return hipGetLastError();
}

// Pixels per lane.  Measured on the default scene: 2 for one sample per pixel (21.5 us against 23.3
// with 1 and 27 with 4 at 1080p), 1 for the 121-sample mode (2.20 ms against 2.34 at the reference's 2048x1024:
// the multi-sample instantiation needs 124 registers at 2 pixels per lane).  MRT_SMALL_P = 1 | 2 | 4 overrides.
static int small_pixels_per_lane(int spp)
{
  static const int forced = [] {
    const char *e = getenv("MRT_SMALL_P");
    const int x = e ? atoi(e) : 0;
    return (x == 1 || x == 2 || x == 4) ? x : 0;
  }();
  return forced ? forced : (spp > 1 ? 1 : 2);
}

hipError_t launch_render_small(const RenderArgs &A, const SmallHost &hs, hipStream_t stream)
{
  if (A.sc.ns + A.sc.nt > 64 || !A.utab || !A.vtab || hs.ns != A.sc.ns || hs.nt != A.sc.nt) return hipErrorInvalidValue;
  switch (small_pixels_per_lane(A.spp)) {
  case 1: return launch_small<1>(A, hs, stream);
  case 4: return launch_small<4>(A, hs, stream);
  default: return launch_small<2>(A, hs, stream);
  }
}

} // namespace mrt
