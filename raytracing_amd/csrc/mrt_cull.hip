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

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace mrt {

#ifdef MRT_TIMELINE
// Diagnostic build only (make variant DEFS=-DMRT_TIMELINE; tools/timeline_cull.py): per wave the 100 MHz real-time
// clock at entry and exit, the shader clock it lived, and how much it had to do.
enum { TLC_WORDS = 4, TLC_MAX_WAVES = 1 << 18 };
__device__ unsigned long long g_timeline_cull[TLC_WORDS * TLC_MAX_WAVES];
extern "C" int mrt_diag_timeline_cull(unsigned long long *host_out, size_t words)
{
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_timeline_cull), words * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

template <int P> struct CVec;
template <> struct CVec<1> { using f = float; using i = int; };
template <> struct CVec<2> { typedef float f __attribute__((ext_vector_type(2))); typedef int i __attribute__((ext_vector_type(2))); };
template <> struct CVec<4> { typedef float f __attribute__((ext_vector_type(4))); typedef int i __attribute__((ext_vector_type(4))); };

// HIER: walk the cluster hierarchy (false: at most one cluster per primitive type).  P: horizontally adjacent
// pixels per lane, i.e. a wave renders 8P x 8 pixels and pays for the walk once per tile.  MULTI: more than one
// sample per pixel (false: no accumulation across samples, fewer registers).
//
// NW: waves per tile.  1: every wave has a tile of its own and never meets another wave.  NW > 1: the workgroup's NW
// waves are a TEAM on one tile -- all of them hold the tile's 64 rays and walk the upper levels of the hierarchy, but
// the leaf clusters that survive (bound tests of 64 primitives + exact tests: where the time goes) are dealt round
// robin, member w taking every NW-th.  Closest hit is an order-independent minimum over (t, object) and occlusion an
// order-independent "any", so the members' partial results merge exactly: through LDS, once after the primary walks
// and once per light.  Member 0 shades and stores.  This is for frames whose time is one tile's time: a tile of the
// 64-light config takes 1.3 ms on one wave while the whole frame is 0.6 ms of work per wave slot (tools/timeline_cull.py).
//
// LSPLIT (teams only): the team deals LIGHTS instead of shadow leaves -- member w takes the lights w, w + NW, ... and
// walks the scene for each of them on its own.  With many lights over a small scene a light's walk is short (two
// exact tests per light and tile in the 64-light config) and what a tile costs is the per-light set-up, 64 times:
// dealing leaves would repeat that set-up in every member.  The reference adds the lights' contributions to a pixel
// in light order (core/light.cuh:171-173; fp32 addition does not associate), so a member does not add up its own
// lights: it leaves each light's contribution in LDS, and member 0 folds them in order, LCH = 16 lights at a time.
#ifndef MRT_REFINE_MD
#define MRT_REFINE_MD 1
#endif
#ifndef MRT_REFINE_RIGOROUS
#define MRT_REFINE_RIGOROUS 1
#endif
#ifndef MRT_REFINE_U
#define MRT_REFINE_U 3e-7f
#endif
#ifndef MRT_REFINE_EPS
#define MRT_REFINE_EPS 1e-4f // relative margin of the leaf refinement's edge functions (the exact test rounds at 6e-8)
#endif
enum { LOOSE_BLOCK_SHIFT = 6, LOOSE_BLOCK = 1 << LOOSE_BLOCK_SHIFT }; // the loose triangles' lists: one per block of 64 x 64 pixels
// the tiles' bins (bin_primitives_kernel): counters start BIN_HEAD words into bin_cnt (word 0: the near list overflowed;
// 1, 2: spheres / triangles in the near list), two per tile (spheres, triangles); bin_near holds BIN_NEAR_CAP of each
enum { BIN_HEAD = 4, BIN_NEAR_CAP = 512 };
// A long near list makes the lists a loss: every tile tests every entry (300 000 spheres at 4K: the leaf clusters next to
// the camera are seen on more than BIN_MAX_RECT tiles each, 430 of them end up in the near list, and the frame took 5.8 ms
// with the lists against 1.6 ms walking).  From this many batches of a kind on, the tiles walk.
#ifndef MRT_BIN_NEAR_OK
#define MRT_BIN_NEAR_OK 32
#endif
#ifndef MRT_CULL_TEAM_WAVES
#define MRT_CULL_TEAM_WAVES 1 // 7 fits the team kernels into 72 registers only by spilling to scratch (C4 3 % faster, 0.3 GB of extra writes per frame): left alone
#endif
#ifndef MRT_CULL_LSPLIT_WAVES
#define MRT_CULL_LSPLIT_WAVES 7 // measured on the 64-light config with the pass's results in LDS (tm_pre): 7 waves (72 registers, 5 spilled outside the light loop) 1.042 ms, 6 waves (74, none spilled) 1.088 ms, 8 waves (64, 15 spilled) see DESIGN.md; before, with the results in pair lanes' registers, 6 waves spilled 23 of them inside the loop: 1.163 ms
#endif
// The cone around a tile's rays (tile column tile_x of TW-pixel-wide tiles, frame rows grow0 .. grow0 + 7), padded by one
// pixel on every side: unit axis, half-angle, and the centre ray itself.  ONE definition, because two kernels must agree on
// it to the bit: the tile kernel culls with it, and the binning kernel decides with it which tiles' lists a primitive
// enters (bin_primitives_kernel).
struct TileCone { float cax, cay, caz, cosT, sinT, dcx, dcy, dcz; };
__device__ __forceinline__ TileCone make_tile_cone(const DevCamera &cam, int W, int H, float cone_ext, int tile_x, int TW, int grow0)
{
  TileCone t;
  const float rw = fast_rcp(float(W)), rh = fast_rcp(float(H)); // bound arithmetic: approximate is fine
  const float um = (float(tile_x * TW) + 0.5f * float(TW - 1)) * rw, vm = (float(grow0) + 3.5f) * rh;
  t.dcx = __builtin_fmaf(vm, cam.vx, __builtin_fmaf(um, cam.hx, cam.ux)) - cam.ox;
  t.dcy = __builtin_fmaf(vm, cam.vy, __builtin_fmaf(um, cam.hy, cam.uy)) - cam.oy;
  t.dcz = __builtin_fmaf(vm, cam.vz, __builtin_fmaf(um, cam.hz, cam.uz)) - cam.oz;
  const float inv = fast_rsq(bdot3(t.dcx, t.dcy, t.dcz, t.dcx, t.dcy, t.dcz));
  t.cax = t.dcx * inv; t.cay = t.dcy * inv; t.caz = t.dcz * inv;
  t.sinT = __builtin_fmaf(cone_ext * inv, 1.0f + 1e-4f, 1e-6f);
  t.cosT = __builtin_fmaf(fast_sqrt(fmaxf(__builtin_fmaf(-t.sinT, t.sinT, 1.0f), 0.0f)), 1.0f - 1e-5f, -1e-6f);
  // a cone of 90 degrees or more (or a NaN camera) cannot cull: cos = -1 makes every ball pass
  if (!(t.sinT < 0.99f) || !(t.cosT > 0.0f)) { t.cosT = -1.0f; t.sinT = 0.0f; }
  return t;
}

// ---- refinement of a triangle against a pyramid of primary rays (a tile, or a block of tiles) -------------------
// What the pyramid looks like to the tests below: a cone around it (unit axis, half-angle) for the plane test, its
// centre ray and half extents along the camera's horizontal / vertical for the screen-space edge functions.
struct ConeBound {
  float cax, cay, caz, cosT, sinT;      // cone: unit axis, half-angle (cosT = -1: cannot cull)
  float dcx, dcy, dcz;                  // centre ray, not normalised
  float ehx, ehy, ehz, evx, evy, evz;   // half the padded region's extent along the camera's horizontal / vertical
  float cam_round;                      // rounding of a ray's direction and of the centre ray's (camera_rounding x 2)
  float dlen_max;                       // max |d| over the region's rays, rounded up
  float reach_max;                      // depth bound: nothing farther than this (along a ray, in units of length) can win; +inf: none
};

// Can a ray of the pyramid be accepted by the exact test of triangle i (core/triangle.cuh:77-103)?  false only when
// provably not; NaN never culls.  Lane-parallel: every lane its own triangle.  AOS: the lanes' triangles are scattered
// (a list), so each lane reads its 64-byte record (one cache line) instead of 16 strided streams (16 lines).
template <bool AOS = false>
__device__ __forceinline__ bool triangle_may_be_hit(const DevScene &sc, const DevCamera &cam, const ConeBound &C, int i)
{
  const int nt = sc.nt;
  float4 rec4[4];
  if constexpr (AOS) {
    const float4 *r4 = reinterpret_cast<const float4 *>(sc.tri_rec + 16 * size_t(i));
#pragma unroll
    for (int k = 0; k < 4; ++k) rec4[k] = r4[k];
  }
  [[maybe_unused]] auto field = [&](int k) { // float k of triangle i's 16
    const float4 &v = rec4[k >> 2];
    return (k & 3) == 0 ? v.x : (k & 3) == 1 ? v.y : (k & 3) == 2 ? v.z : v.w;
  };
    // Many candidates in one leaf (the usual leaf has one to three): worth a closer look at each, lane-parallel,
    // before the wave spends a division and up to three edge tests on every one of them for all its pixels.
    //
    // (1) The triangle's PLANE against the tile's cone: t = (k - n.o) / (n.d), and over the cone's unit
    // directions n.u stays within c cos(theta) -+ s sin(theta), c = n.axis, s = |n - c axis|.  If that range
    // keeps one sign, either every ray of the tile meets the plane behind the camera (the exact test wants
    // t >= 0.01), or at t >= |k - n.o| / (max |n.u| max |d|): beyond the tile's depth bound it cannot win.
    // (2) The EDGE planes in screen space (as the small-scene kernel does, mrt_small.hip): with the sign of n.d
    // fixed, e_i(p) |n.d| = +-w_i.d, w_i = (m_i.o - c_i) n + (k - n.o) m_i, which is affine in the pixel
    // coordinates: its maximum over the padded tile is its value at the tile's centre plus a constant.  Below
    // the exact test's -1e-6 (and a margin of 1e-4 of every magnitude that enters either computation; the
    // exact test rounds at 6e-8): no ray of the tile passes edge i.
    // This is what deals with triangles that have no finite bound -- the reference's edge planes accept an
    // unbounded wedge when a triangle's plane passes near the world origin: 223 of the 1M-triangle terrain,
    // which every tile used to test -- and with the long slivers along a horizon.
    const size_t snt = size_t(nt);
    const float pnx = AOS ? field(0) : sc.tri[i], pny = AOS ? field(1) : sc.tri[snt + i], pnz = AOS ? field(2) : sc.tri[2 * snt + i],
                kp = AOS ? field(3) : sc.tri[3 * snt + i];
    const float num = kp - dot3(cam.ox, cam.oy, cam.oz, pnx, pny, pnz);
    const float c = bdot3(pnx, pny, pnz, C.cax, C.cay, C.caz);
    const float qx = __builtin_fmaf(-c, C.cax, pnx), qy = __builtin_fmaf(-c, C.cay, pny), qz = __builtin_fmaf(-c, C.caz, pnz);
    const float sp = fast_sqrt(bdot3(qx, qy, qz, qx, qy, qz));
    const float pn1 = norm1(pnx, pny, pnz);
    const float slack = 2e-5f * pn1;
    const float ccos = c * C.cosT;
    const float lo = __builtin_fmaf(-sp, C.sinT, ccos) - slack, hi = __builtin_fmaf(sp, C.sinT, ccos) + slack;
    if (C.cosT > 0.0f && (lo > 0.0f || hi < 0.0f)) { // (C.cosT = -1: the cone cannot cull)
      // max |n.u| over the cone: cos(phi - theta) for a normal at angle phi >= theta from the axis -- but a normal
      // INSIDE the cone (phi < theta: a wall seen head-on, or any normal once rounding has widened the cone far
      // from the world origin) is some ray's direction, and the maximum is |n|
      const bool inside = sp * C.cosT <= fabsf(c) * C.sinT * 1.01f + slack;
      const float big = inside ? pn1 : fmaxf(fabsf(lo), fabsf(hi));
      const bool behind = (lo > 0.0f) ? (num < 0.0f) : (num > 0.0f);
      const float o1 = norm1(cam.ox, cam.oy, cam.oz);
      if (behind && fabsf(num) > 1e-4f * (fabsf(kp) + o1 * pn1)) return false;
      else if (fabsf(num) * (1.0f - 1e-4f) > C.reach_max * big) return false;
      {
        const float sgn = lo > 0.0f ? 1.0f : -1.0f;
        const float dl = C.dlen_max * 1.01f; // the padded tile's rays are a little longer than the pixels'
        [[maybe_unused]] const float N1 = fabsf(kp) + pn1 * o1;
        [[maybe_unused]] const float inv_lo = fast_rcp(fminf(fabsf(lo), fabsf(hi))) * 1.01f; // 1 / min |n.u| over the cone
#pragma unroll
        for (int e = 0; e < 3; ++e) {
          const float mx = AOS ? field(4 + 4 * e) : sc.tri[(4 + 4 * e) * snt + i], my = AOS ? field(5 + 4 * e) : sc.tri[(5 + 4 * e) * snt + i],
                      mz = AOS ? field(6 + 4 * e) : sc.tri[(6 + 4 * e) * snt + i], ce = AOS ? field(7 + 4 * e) : sc.tri[(7 + 4 * e) * snt + i];
          const float a = bdot3(mx, my, mz, cam.ox, cam.oy, cam.oz) - ce;
          const float wx = __builtin_fmaf(a, pnx, num * mx), wy = __builtin_fmaf(a, pny, num * my), wz = __builtin_fmaf(a, pnz, num * mz);
          const float top = sgn * bdot3(wx, wy, wz, C.dcx, C.dcy, C.dcz) + fabsf(bdot3(wx, wy, wz, C.ehx, C.ehy, C.ehz)) +
                            fabsf(bdot3(wx, wy, wz, C.evx, C.evy, C.evz));
          const float m1 = norm1(mx, my, mz);
          // How far can the exact test's fp32 e_i |n.d| be from w_i.d as computed here?  First-order rounding
          // analysis of both (u = 2^-24 per operation, 1-norms for the sums):
          //   num = k - n.o        off by 4u N1,  N1 = |k| + |n|1 |o|1
          //   t = num / (n.d)      relative 4u N1 / |num| + 3u |n|1 |d| / |n.d| + u  -- the middle term is what a
          //                        grazing ray amplifies: |n.d| >= lo_abs |d| over the cone bounds it
          //   p = o + t d, e = m.p - c   move by |m.d| |dt| + 5u (|m|1 |o|1 + |m.d| |t|) + u |c|
          // times |n.d| <= big |d|, plus this block's own a, w and w.d (4u each of their magnitudes):
          //   u dl [ m1 (8 N1 + 7 |num| + 3.5 |n|1 |num| / lo_abs) + (4 |n|1 + 5 big)(m1 |o|1 + |c|) + 4 |w|1 ]
          // taken five times over (REFINE_U = 3e-7); then the rounding of the rays themselves (coarse far from the
          // world origin: camera_rounding) through |w|, and the exact test's own -1e-6.  (Until round 2 this was a
          // flat 1e-4 of the magnitudes: three times as many candidates on the 1M-triangle terrain.)
#if MRT_REFINE_RIGOROUS
          const float w1 = norm1(wx, wy, wz);
#if MRT_REFINE_MD
          // the amplified term moves p ALONG THE RAY, so it reaches e_i through |m.d|, not |m| |d|: bound that over
          // the padded tile (centre value + extents) instead of by norms
          const float md = (fabsf(bdot3(mx, my, mz, C.dcx, C.dcy, C.dcz)) + fabsf(bdot3(mx, my, mz, C.ehx, C.ehy, C.ehz)) +
                            fabsf(bdot3(mx, my, mz, C.evx, C.evy, C.evz))) * 1.001f;
          const float anum = fabsf(num);
          const float err = __builtin_fmaf(MRT_REFINE_U,
                                __builtin_fmaf(dl * m1, __builtin_fmaf(7.0f, anum, 8.0f * N1),
                                    __builtin_fmaf(md * 3.5f * pn1, anum * inv_lo,
                                        dl * __builtin_fmaf(__builtin_fmaf(5.0f, big, 4.0f * pn1), __builtin_fmaf(m1, o1, fabsf(ce)), 4.0f * w1))),
                                __builtin_fmaf(w1, C.cam_round, 1e-6f * big * dl));
#else
          const float err = MRT_REFINE_U * dl * (m1 * (8.0f * N1 + fabsf(num) * (7.0f + 3.5f * pn1 * inv_lo)) +
                                                 (4.0f * pn1 + 5.0f * big) * (m1 * o1 + fabsf(ce)) + 4.0f * w1) +
                            w1 * C.cam_round + 1e-6f * big * dl;
#endif
#else
          const float err = MRT_REFINE_EPS * ((m1 * o1 + fabsf(ce)) * pn1 + fabsf(num) * m1) * (2.0f * dl) +
                            norm1(wx, wy, wz) * C.cam_round + 1e-6f * big * dl;
#endif
          if (top + err < 0.0f) return false; // NaN compares false: never culls
        }
      }
    }
  return true;
}

// ---- triangle set-up: the edge functions of every triangle at the frame's pixels, once per frame ------------------
// PAIRS (render_cull_kernel, frames of ONE ray per pixel).  The refinement above asks whether any ray of a tile's
// pyramid can be accepted; the same three functions sgn w_e.d, affine in a pixel's column and row, also say WHICH pixels
// can -- if they are sharp enough to tell which of a tile's rays a sliver on a horizon lies between.  In fp32 and per
// tile they are not: w_e.d cancels to 1e-5 of its terms on a grazing terrain and a pixel column is 3e-6 of them.  So a
// kernel of its own works them out once per frame and triangle (lane = triangle), in DOUBLE from the fp32 record and
// the camera (exact for the purpose), relative to a reference pixel next to the triangle so that fp32 carries them
// from there:
//   per edge e:  value at the reference pixel, step per column, step per row, error bound
//   per triangle: the reference pixel (column, row), the nearest an accepted hit can be, valid
// and a tile evaluates value + (column - ref) step + (row - ref) step in two FMAs.  Three things keep the bound small:
//  * the sign of n.d is the NUMERATOR's: the exact test wants t = (k - n.o) / (n.d) >= 0.01, so on every ray it accepts
//    n.d has the sign of k - n.o -- it need not be the same over a whole tile (on a hill's silhouette it is not);
//  * the directions of accepted rays come from the triangle's BOUNDING BALL (the hit point lies in the accepted region,
//    mrt_scene_build.h tri_bound): |m.d|, |n.d| and the |t| |d| that multiplies the rounding of n.d are bounded over
//    the rays through the ball, not over a tile's cone -- where min |n.d| is zero on a silhouette;
//  * a pixel's own fp32 ray ((ul + u h) + v vv) - o, u = col / W, v = row / H rounds seven times (u, u h, the sum, v,
//    v vv, the sum, the difference), in component k by at most 2^-24 (3 |ul_k| + 5 |h_k| + 4 |v_k| + |o_k|) =: r_k, and
//    enters as sum_k |w_k| r_k (taken 1.5 times) instead of |w|1 times a bound over the whole camera.
// What is left is the exact test's own rounding in e, times |n.d| <= big |d| (first order, u = 2^-24, five times over;
// N1 = |k| + |n|1 |o|1, t (n.d) = num):
//   t = fl(num / n.d), num off by 4u N1, n.d by 3u |n|1 |d|   ->  |dt| |n.d| <= 4u N1 + 3u |n|1 |t| |d| + u |num|, which
//                                                                 moves p ALONG the ray: e by |m.d| |dt|
//   p = fl(o + fl(t d)), e = fl(fl(m.p) - c)                  ->  5u |m|1 |t| |d| + 4u |m|1 |o|1 + 2u |c|
//   u [ |m.d| (8 N1 + 2 |num| + 3.5 |n|1 |t| |d|) + |d| (5 |m|1 |num| + 5 big (|m|1 |o|1 + |c|)) ] + 1e-6 big |d|
// A triangle whose plane holds the camera (to rounding: no sign to go by) is not valid (.w = 0): the tile refines it as
// above and tests it for all pixels.  One without a finite bound (the loose triangles: .w = 2) gets its values, steps and
// the rays' part of the bound here and the exact test's part from the tile, over the tile's cone (setup_bound_at_tile).
struct SetupCam {
  double ul[3], h[3], v[3], o[3];   // the camera, ray(col, row) = ul + (col / W) h + (row / H) v - o
  double inv[3][3];                 // rows: the dual basis of (h, v, ul - o) -- a direction's (alpha, beta, gamma)
  double rW, rH;
  float r[3];                       // rounding of a pixel's ray per component (above), taken 1.5 times
  float dl;                         // max |ray| over the frame, rounded up
  int W, H;
};
enum { SETUP_FLOATS = 16 };
__global__ __launch_bounds__(256) void tri_setup_kernel(DevScene sc, SetupCam C, float4 *__restrict__ out)
{
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= sc.nt) return;
  const float4 *r4 = reinterpret_cast<const float4 *>(sc.tri_rec + 16 * size_t(i));
  const float4 q0 = r4[0];
  const float ox = float(C.o[0]), oy = float(C.o[1]), oz = float(C.o[2]);
  const float num = q0.w - dot3(ox, oy, oz, q0.x, q0.y, q0.z); // the exact test's own numerator
  const float anum = fabsf(num), pn1 = norm1(q0.x, q0.y, q0.z), o1 = norm1(ox, oy, oz);
  const float N1 = fabsf(q0.w) + pn1 * o1;
  const float R = sc.tri_br[i];
  float4 *w4 = out + 4 * size_t(i);
  const bool valid = anum > 1e-4f * N1; // (NaN: not valid)
  const bool ball = R < 3.0e38f;         // false: no finite bound -- the tile works out the part of the bound that wants one
  if (!valid) {
    w4[0] = make_float4(0, 0, 0, 0); w4[1] = w4[0]; w4[2] = w4[0]; w4[3] = make_float4(0, 0, 0, 0); // .w == 0: not valid
    return;
  }
  const double sgn = num > 0.0f ? 1.0 : -1.0;
  // the ball: direction to its centre, the angle it is seen under, the farthest an accepted hit can be
  const double bx = double(sc.tri_bx[i]) - C.o[0], by = double(sc.tri_by[i]) - C.o[1], bz = double(sc.tri_bz[i]) - C.o[2];
  const double L = sqrt(bx * bx + by * by + bz * bz);
  const double Rp = ball ? double(R) * 1.01 + 4e-7 * (L + double(o1)) + 1e-30 : 0.0; // (a hit point is itself rounded)
  const double sinb = ball && L > Rp ? Rp / L : 1.0;
  const double far = (L + Rp) * 1.01;
  // reference pixel: where the ball's centre is seen, or the frame's middle when it is behind the camera
  double col = 0.5 * C.W, row = 0.5 * C.H;
  {
    const double al = C.inv[0][0] * bx + C.inv[0][1] * by + C.inv[0][2] * bz, be = C.inv[1][0] * bx + C.inv[1][1] * by + C.inv[1][2] * bz,
                 ga = C.inv[2][0] * bx + C.inv[2][1] * by + C.inv[2][2] * bz;
    if (ga > 0.0) {
      const double c0 = al / ga * C.W, r0 = be / ga * C.H;
      if (fabs(c0) < 1e6 && fabs(r0) < 1e6) { col = rint(c0); row = rint(r0); }
    }
  }
  const double u = col * C.rW, v = row * C.rH;
  const double dx = __builtin_fma(v, C.v[0], __builtin_fma(u, C.h[0], C.ul[0])) - C.o[0],
               dy = __builtin_fma(v, C.v[1], __builtin_fma(u, C.h[1], C.ul[1])) - C.o[1],
               dz = __builtin_fma(v, C.v[2], __builtin_fma(u, C.h[2], C.ul[2])) - C.o[2];
  const double nx = q0.x, ny = q0.y, nz = q0.z;
  const double num64 = double(q0.w) - (C.o[0] * nx + C.o[1] * ny + C.o[2] * nz);
  const double n2 = sqrt(nx * nx + ny * ny + nz * nz);
  const double invL = L > 0.0 ? 1.0 / L : 0.0;
  // max |n.u| over the unit directions through the ball
  const double big = fmin(n2, fabs(nx * bx + ny * by + nz * bz) * invL + n2 * sinb) * 1.001;
  const float dl = C.dl;
  float tn = float(fabs(num64) / (big > 0.0 ? big : 1e-300) * (1.0 - 1e-6)); // |t| |d| of an accepted ray is at least this
  float4 o4[3];
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const float4 qe = r4[1 + e];
    const double mx = qe.x, my = qe.y, mz = qe.z;
    const double a = (mx * C.o[0] + my * C.o[1] + mz * C.o[2]) - double(qe.w);
    const double wx = a * nx + num64 * mx, wy = a * ny + num64 * my, wz = a * nz + num64 * mz;
    const double S = sgn * (wx * dx + wy * dy + wz * dz);
    const double Hs = sgn * (wx * C.h[0] + wy * C.h[1] + wz * C.h[2]) * C.rW, Vs = sgn * (wx * C.v[0] + wy * C.v[1] + wz * C.v[2]) * C.rH;
    const double m2 = sqrt(mx * mx + my * my + mz * mz), m1 = fabs(mx) + fabs(my) + fabs(mz);
    const double md = double(dl) * fmin(m2, fabs(mx * bx + my * by + mz * bz) * invL + m2 * sinb) * 1.001; // |m.d| on accepted rays
    const double E = 1.02 * (fabs(wx) * C.r[0] + fabs(wy) * C.r[1] + fabs(wz) * C.r[2]);
    const double own = double(MRT_REFINE_U) * (md * (8.0 * N1 + 2.0 * anum + 3.5 * pn1 * far) +
                                               double(dl) * (5.0 * m1 * anum + 5.0 * big * (m1 * o1 + fabs(double(qe.w))))) +
                       1e-6 * big * dl;
    const double err = (ball ? E + own : E) * 1.0001;
    o4[e] = make_float4(float(S), float(Hs), float(Vs), float(err) * (1.0f + 1e-6f) + 1e-37f);
  }
  w4[0] = o4[0]; w4[1] = o4[1]; w4[2] = o4[2];
  w4[3] = make_float4(float(col), float(row), ball ? tn : 0.0f, ball ? 1.0f : 2.0f);
}

// The exact test's part of the bound for a triangle without a bounding ball, over a tile's cone instead (the terms of
// tri_setup_kernel with |m.d|, max |n.u| and |t| |d| = |num| / min |n.u| taken over the tile).  own[e] per edge; false:
// no ray of the tile is accepted (the plane tests of triangle_may_be_hit (1)); own = +inf when n.d changes sign over
// the tile (nothing to go by).
__device__ __forceinline__ bool setup_bound_at_tile(const DevScene &sc, const DevCamera &cam, const ConeBound &C, int i, float own[3])
{
  const float4 *r4 = reinterpret_cast<const float4 *>(sc.tri_rec + 16 * size_t(i));
  const float4 q0 = r4[0];
  const float num = q0.w - dot3(cam.ox, cam.oy, cam.oz, q0.x, q0.y, q0.z);
  const float anum = fabsf(num), pn1 = norm1(q0.x, q0.y, q0.z), o1 = norm1(cam.ox, cam.oy, cam.oz);
  const float N1 = fabsf(q0.w) + pn1 * o1;
  const float c = bdot3(q0.x, q0.y, q0.z, C.cax, C.cay, C.caz);
  const float qx = __builtin_fmaf(-c, C.cax, q0.x), qy = __builtin_fmaf(-c, C.cay, q0.y), qz = __builtin_fmaf(-c, C.caz, q0.z);
  const float sp = fast_sqrt(bdot3(qx, qy, qz, qx, qy, qz));
  const float slack = 2e-5f * pn1;
  const float ccos = c * C.cosT;
  const float lo = __builtin_fmaf(-sp, C.sinT, ccos) - slack, hi = __builtin_fmaf(sp, C.sinT, ccos) + slack;
  own[0] = own[1] = own[2] = MRT_INF;
  if (!(C.cosT > 0.0f && (lo > 0.0f || hi < 0.0f))) return true;
  const bool inside = sp * C.cosT <= fabsf(c) * C.sinT * 1.01f + slack;
  const float big = inside ? pn1 : fmaxf(fabsf(lo), fabsf(hi));
  const bool behind = (lo > 0.0f) ? (num < 0.0f) : (num > 0.0f);
  if (behind && anum > 1e-4f * N1) return false;
  if (anum * (1.0f - 1e-4f) > C.reach_max * big) return false;
  const float dl = C.dlen_max * 1.01f;
  const float far = anum * fast_rcp(fminf(fabsf(lo), fabsf(hi))) * 1.01f;
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const float4 qe = r4[1 + e];
    const float m1 = norm1(qe.x, qe.y, qe.z);
    const float md = (fabsf(bdot3(qe.x, qe.y, qe.z, C.dcx, C.dcy, C.dcz)) + fabsf(bdot3(qe.x, qe.y, qe.z, C.ehx, C.ehy, C.ehz)) +
                      fabsf(bdot3(qe.x, qe.y, qe.z, C.evx, C.evy, C.evz))) * 1.001f;
    own[e] = __builtin_fmaf(MRT_REFINE_U,
                 __builtin_fmaf(md, __builtin_fmaf(3.5f * pn1, far, __builtin_fmaf(8.0f, N1, 2.0f * anum)),
                     dl * __builtin_fmaf(5.0f * big, __builtin_fmaf(m1, o1, fabsf(qe.w)), 5.0f * m1 * anum)),
                 1e-6f * big * dl) * 1.0001f;
  }
  return true;
}

template <bool HIER, int P, bool MULTI, int NW, bool LSPLIT = false>
#ifndef MRT_CULL_SINGLE_WAVES
#define MRT_CULL_SINGLE_WAVES 1 // (experiment: 8 forces the one-wave-per-tile, one-sample kernel into 64 registers)
#endif
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NW > 1 && !MULTI ? (LSPLIT ? MRT_CULL_LSPLIT_WAVES : MRT_CULL_TEAM_WAVES) : (!MULTI && P == 1 && HIER ? MRT_CULL_SINGLE_WAVES : 1), 8)))
void render_cull_kernel(RenderArgs A)
{
  const DevScene &sc = A.sc;
  const DevCamera &cam = A.cam;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tx = lane & 7, ty = lane >> 3;
  const int ns = sc.ns, nt = sc.nt;
  constexpr int TW = 8 * P;
  static_assert(NW == 1 || P == 1, "teams render one pixel per lane");
  constexpr int NWS = NW > 1 ? NW : 1;
  __shared__ float tm_t[NW > 1 ? NWS * 64 : 1];            // members' closest hits, [member][lane]
  __shared__ int tm_obj[NW > 1 ? NWS * 64 : 1], tm_prim[NW > 1 ? NWS * 64 : 1];
  __shared__ unsigned long long tm_open[NW > 1 ? 2 * NWS : 1]; // members' still-unoccluded masks, [light parity][member]
  static_assert(!LSPLIT || NW > 1, "lights are dealt among the members of a team");
#ifndef MRT_CULL_LCH
#define MRT_CULL_LCH 16
#endif
  constexpr int LCH = MRT_CULL_LCH;                            // lights per fold
  constexpr int PRE_LPP = 16;                                  // lights per precompute pass (LDS slots per member)
  __shared__ float tm_c[LSPLIT ? 3 * LCH * 64 : 1];            // a light's contribution, [channel][light % LCH][lane]
  __shared__ unsigned long long tm_lit[LSPLIT ? LCH : 1];      // lanes the light contributes to
  // what a member's precompute_pass leaves for its next lights, [member][light of the pass][field] (the member's own
  // wave writes and reads it: no barrier).  In pair lanes' registers these 15 values were spilled to scratch and
  // reloaded for every light (96 B per lane, 0.37 GB of scratch writes per C5 frame)
  __shared__ float tm_pre[LSPLIT ? NWS * PRE_LPP * 16 : 1];
  const int member = NW > 1 ? wave : 0;
  // PAIRS (triangle_batch below): the edge functions of a batch's survivors at the tile's pixels, [wave][survivor][12],
  // and the pixels' lists of triangles still to be tested, [wave][entry][pixel]
  constexpr bool PAIRS = HIER && P == 1 && !MULTI && NW > 1 && !LSPLIT; // (a wave on its own: 25 more registers, a wave per SIMD less -- 10k spheres 0.32 -> 0.38 ms)
  constexpr int PK_WAVES = NWS;
#ifndef MRT_PK_CAP
#define MRT_PK_CAP 4    // entries of a pixel's list
#endif
#ifndef MRT_PK_SLOTS
#define MRT_PK_SLOTS 16 // survivors of a batch that hand their functions over at a time
#endif
#ifndef MRT_PK_FAT
#define MRT_PK_FAT 32   // a survivor that more pixels than this cannot exclude is tested the usual way
#endif
  constexpr int PK_CAP = MRT_PK_CAP;
  constexpr int PK_SLOTS = MRT_PK_SLOTS;
  __shared__ float4 pk_fn[PAIRS ? PK_WAVES * PK_SLOTS * 3 : 1];
  __shared__ uint32_t pk_list[PAIRS ? PK_WAVES * PK_CAP * 64 : 1];
  constexpr int PK_FAT = MRT_PK_FAT;
  // (the set-up lives behind RenderArgs.utab, which only the small-scene kernel reads: the arguments are at their 4 KB limit)
  [[maybe_unused]] const float4 *tri_setup = PAIRS ? reinterpret_cast<const float4 *>(A.utab) : nullptr;
  [[maybe_unused]] const bool pairs_on = PAIRS && tri_setup != nullptr; // no set-up (no scratch: the allocation-free entry point): the usual way
  [[maybe_unused]] const int wslot = wave;
  [[maybe_unused]] int pk_n = 0;              // triangles in this pixel's list
  [[maybe_unused]] const float pk_a = float(tx) - 3.5f, pk_b = float(ty) - 3.5f; // the pixel's column / row from the tile's centre ray

  // 2-D grid: blockIdx.y = tile row, blockIdx.x*4 + wave = tile column (both wave-uniform, so the strip
  // arithmetic below runs on the scalar unit); a team's tile column is its workgroup's index
  int tile_x = NW > 1 ? int(blockIdx.x) : int(blockIdx.x * (blockDim.x >> 6) + wave);
  int tile_y = blockIdx.y;
  if (A.tile_map) {
    // XCD-aware order (launcher: 1-D grid).  Workgroups go to the eight XCDs round robin, each with an L2 of its own: in
    // row order the eight neighbours of a tile -- which open the same leaves -- land on eight different L2s, and every
    // L2 ends up fetching the whole visible scene.  Here workgroup w serves XCD w % 8, and an XCD takes the frame in
    // blocks of 8 x 8 tiles dealt to the XCDs in turn: a block's tiles run side by side on ONE L2, and the blocks of
    // every row of blocks are spread over all eight (sky rows and geometry rows alike, so the XCDs stay balanced).
    const int bs = 1 << A.tile_map;                              // tiles along a block's side (launcher: 8)
    const int wg_tiles = NW > 1 ? 1 : int(blockDim.x >> 6);     // tiles of a workgroup, side by side
    const int upr = bs / wg_tiles;                               // workgroups per row of a block
    const int txn = (A.W + TW - 1) / TW, tyn = (A.local_rows + 7) >> 3;
    const int bxn = (txn + bs - 1) >> A.tile_map, nb = bxn * ((tyn + bs - 1) >> A.tile_map);
    const int w = blockIdx.x, x = w & 7, k = w >> 3;
    int b = x + 8 * (k / (bs * upr));
    const int t = k % (bs * upr);
    if (b >= nb) return;
    if (A.block_order) b = int(A.block_order[b]); // heaviest blocks first (block_order_kernel)
    tile_y = (b / bxn) * bs + t / upr;
    tile_x = (b % bxn) * bs + (t % upr) * wg_tiles + (NW > 1 ? 0 : wave);
    if (tile_y >= tyn) return;
  }
  if (tile_x * TW >= A.W) return; // NW == 1: waves are independent; NW > 1: the whole team leaves
  int leaf_k = 0; // leaf clusters met so far in the current walk (the same sequence in every member)
  [[maybe_unused]] int open_set = 0; // which set of tm_open the next merged light uses (the same in every member)
  [[maybe_unused]] bool solo = false; // a team's tile that member 0 renders alone (below: SOLO TILES)
#define MRT_TEAM_TAKES_LEAF() (NW == 1 || solo || (leaf_k++ % NWS) == member)
#define MRT_TEAM_TAKES_SHADOW_LEAF() (NW == 1 || LSPLIT || solo || (leaf_k++ % NWS) == member)
#ifdef MRT_TIMELINE
  const unsigned tl_idx = tile_y * ((A.W + TW - 1) / TW) + tile_x;
  const unsigned long long tl_rt0 = __builtin_amdgcn_s_memrealtime(), tl_c0 = __builtin_amdgcn_s_memtime();
#endif
  const int col0 = tile_x * TW + tx * P;
  const int lrow = tile_y * 8 + ty;
  // strips are multiples of 8 rows, so a tile never straddles two strips
  const int lrow0 = tile_y * 8;
  const int strip = lrow0 / A.strip_rows;
  const int grow0 = global_strip(strip, A.rank, A.nranks) * A.strip_rows + (lrow0 - strip * A.strip_rows);
  const int grow = grow0 + ty;
  const bool valid_row = (lrow < A.local_rows) && (grow < A.H);
  bool valid[P];
#pragma unroll
  for (int p = 0; p < P; ++p) valid[p] = valid_row && (col0 + p < A.W);

  // ---- the tile's lists (bin_primitives_kernel; "PRIMARY RAYS FROM BINS" below), looked up before anything else so that
  //      the members of a team that have nothing to do on this tile leave early ----
  const int gtile = (grow0 >> 3) * ((A.W + TW - 1) / TW) + tile_x; // the tile's number in the whole frame
  bool binned_s = false, binned_t = false;
  if constexpr (HIER && P == 1) {
    if (A.bin_cnt && A.bin_cnt[0] == 0u) { // (word 0: the frame's near list overflowed -- nobody uses the bins)
      const uint32_t cs = A.bin_cnt[BIN_HEAD + 2 * gtile], ct = A.bin_cnt[BIN_HEAD + 2 * gtile + 1];
      // (... nor a kind whose near list is long, in batches of 64 lanes: a leaf cluster is one, 64 primitives are one)
      const uint32_t near_ok = A.bin_clusters ? (uint32_t)MRT_BIN_NEAR_OK : 64u * (uint32_t)MRT_BIN_NEAR_OK;
      binned_s = cs <= (uint32_t)A.bin_cap && A.bin_cnt[1] <= near_ok;
      binned_t = ct <= (uint32_t)A.bin_cap && A.bin_cnt[2] <= near_ok;
    }
  }
  // SOLO TILES.  A team is there for the tiles that are long jobs; most tiles are not (1M triangles at 4K: the median
  // tile has three leaves and three exact tests, the 99th percentile 30 and 236), and on those four waves each pay
  // for the tile's set-up, its rays, the merges and their barriers.  With lists the amount of work is known before
  // it starts: a tile whose lists hold at most solo_max batches is rendered by member 0 alone -- as the one-wave
  // kernel would, no merges, no barriers -- and the other members leave at once.  The counts are the same in every
  // member, so is the decision.
  if constexpr (NW > 1 && !LSPLIT && HIER && P == 1) {
    if (binned_s && binned_t && A.bin_clusters && A.solo_max > 0) {
      uint32_t jobs = A.bin_cnt[BIN_HEAD + 2 * gtile] + A.bin_cnt[BIN_HEAD + 2 * gtile + 1] + A.bin_cnt[1] + A.bin_cnt[2];
      if (A.loose_cnt != nullptr) {
        const uint32_t lc = A.loose_cnt[(grow0 >> LOOSE_BLOCK_SHIFT) * A.loose_bx + ((tile_x * TW) >> LOOSE_BLOCK_SHIFT)];
        jobs += lc > (uint32_t)A.loose_cap ? 1u << 20 : (lc + 63u) >> 6;
      } else {
        jobs += (uint32_t)(sc.nct - A.bin_nct_regular); // the loose clusters are walked
      }
      solo = jobs <= (uint32_t)A.solo_max;
      if (solo && member != 0) return;
    }
  }

  // ---- cone around the tile's rays, padded by one pixel on every side (which also covers the sub-pixel
  //      offsets of the 121-spp pattern, within (-11/13, +9/13) pixel).  d(u,v) is affine in (u,v), so every
  //      ray of the padded tile is the centre ray plus an offset of norm <= cone_ext (launcher): the angle
  //      to the centre ray is at most asin(cone_ext / |centre ray|). ----
  float cax, cay, caz, cosT, sinT;
  float dcx, dcy, dcz, cam_round;          // the tile's centre ray, not normalised; rounding of a ray's direction
  float ehx, ehy, ehz, evx, evy, evz;      // half the padded tile's extent along the camera's horizontal / vertical
  constexpr int TRI_REFINE_MIN = 4;        // candidates in a leaf from which the per-triangle refinement runs
  {
    const TileCone tc = make_tile_cone(cam, A.W, A.H, A.cone_ext, tile_x, TW, grow0);
    cax = tc.cax; cay = tc.cay; caz = tc.caz; cosT = tc.cosT; sinT = tc.sinT;
    dcx = tc.dcx; dcy = tc.dcy; dcz = tc.dcz;
    const float rw = fast_rcp(float(A.W)), rh = fast_rcp(float(A.H)); // bound arithmetic: approximate is fine
    cam_round = 2.0f * camera_rounding(cam); // a ray's and the centre ray's
    const float du = (0.5f * float(TW) + 0.51f) * rw * 1.001f, dv = 4.51f * rh * 1.001f; // pixels -1 .. 8P, -1 .. 8
    ehx = du * cam.hx; ehy = du * cam.hy; ehz = du * cam.hz;
    evx = dv * cam.vx; evy = dv * cam.vy; evz = dv * cam.vz;
  }
  // hierarchy tests against the cone: a ball of spheres / a ball of triangle bounds
  auto cone_vs_sphere_ball = [&](float x, float y, float z, float R) {
    const float wx = x - cam.ox, wy = y - cam.oy, wz = z - cam.oz;
    const float L = __builtin_fmaf(fast_sqrt(bdot3(wx, wy, wz, wx, wy, wz)), CULL_REL, R);
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

    // ---- depth bound of the tile: once every pixel of the tile has a hit, nothing farther than the farthest of
    //      them can win anywhere in the tile (the closest-hit rule takes t < best, or t == best with a lower object
    //      number).  A hit at parameter t lies t |d| from the camera, so a primitive whose bound starts farther than
    //      tile_tmax * max |d| away is skipped, as are its cluster and super-cluster.  Wave-uniform, refreshed
    //      after every leaf that moved a best_t; +inf while a pixel (inside the frame) has no hit.  Margin 1e-4
    //      (relative) against a rounding of 1e-6 in the exact tests' t and hit points. ----
    // (max |d| over the tile: every ray of the padded tile is the centre ray plus an offset of norm <= cone_ext, so
    // |centre ray| + cone_ext bounds it without a reduction over the lanes)
    const float dlen_max = (fast_sqrt(bdot3(dcx, dcy, dcz, dcx, dcy, dcz)) + A.cone_ext) * (1.0f + 1e-5f);
    float reach_max = MRT_INF; // tile_tmax * max |d| * (1 + 1e-4)
    auto refresh_reach = [&]() {
      float m = 0.0f;
#pragma unroll
      for (int p = 0; p < P; ++p) m = fmaxf(m, valid[p] ? best_t[p] : 0.0f);
      reach_max = wave_max(m) * dlen_max * (1.0f + 1e-4f) + 1e-6f; // +inf stays +inf; NaN never culls (compares false)
    };
    auto sphere_ball_in_reach = [&](float x, float y, float z, float R) {
      if (!cone_vs_sphere_ball(x, y, z, R)) return false;
      const float wx = x - cam.ox, wy = y - cam.oy, wz = z - cam.oz;
      const float L = fast_sqrt(bdot3(wx, wy, wz, wx, wy, wz));
      return !(__builtin_fmaf(L, 1.0f - 1e-5f, -cluster_cull_radius(R, __builtin_fmaf(L, CULL_REL, R))) > reach_max);
    };
    // (Triangles get no such test on their bounding balls: far from the world origin the exact test's
    // t = (k - n.o) / (n.d) cancels catastrophically in k - n.o and is off by 1e-7 (|k| + |n||o|) / |n.d|, without
    // bound for grazing rays, so a t computed NEARER than the ball can win.  Found by tools/stress_parity.py with the
    // scene moved 1e5 units.  Their depth test is the plane test of the leaf refinement below, which starts from
    // the very same fp32 k - n.o.)
    auto tri_ball_in_reach = [&](float x, float y, float z, float R) { return cone_vs_tri_ball(x, y, z, R); };

    // ---- closest hit over spheres: src/main.cu:60-77, core/sphere.cuh:32-73, interval [0.01, inf] ----
    leaf_k = 0;
    // one batch of up to 64 candidate spheres, lane = sphere i (have: the lane has one).  binned: the cone test was made
    // when the sphere entered this tile's list; base >= 0: the lanes' spheres are base + lane (a leaf of the hierarchy)
    auto sphere_batch = [&](int i, bool have, int base, bool binned) {
      float cx = 0.0f, cy = 0.0f, cz = 0.0f, r = 0.0f;
      bool pass = false;
      if (have) {
        cx = sc.sph_cx[i]; cy = sc.sph_cy[i]; cz = sc.sph_cz[i]; r = sc.sph_r[i];
        const float wx = cx - cam.ox, wy = cy - cam.oy, wz = cz - cam.oz;
        const float L2 = bdot3(wx, wy, wz, wx, wy, wz);
        const float reff = sphere_cull_radius(r * r, L2);
        pass = (binned || ball_may_touch_cone(wx, wy, wz, reff, cax, cay, caz, cosT, sinT)) &&
               !(__builtin_fmaf(fast_sqrt(L2), 1.0f - 1e-5f, -reff) > reach_max);
      }
      unsigned long long mask = __ballot(pass);
      n_clusters++; n_exact += __popcll(mask);
      bool moved = false;
      while (mask) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1;
        const float scx = bcast(cx, j), scy = bcast(cy, j), scz = bcast(cz, j), sr = bcast(r, j);
        const float ocx = cam.ox - scx, ocy = cam.oy - scy, ocz = cam.oz - scz;
        const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - sr * sr;
        const int sj = base >= 0 ? base + j : __builtin_amdgcn_readlane(i, j); // lane j's sphere
        const int obj = sc.sph_obj[sj];
#pragma unroll
        for (int p = 0; p < P; ++p) {
          const float b = dot3(ocx, ocy, ocz, dx[p], dy[p], dz[p]);
          const float disc = b * b - aa[p] * c;
          if (disc >= 0.0f) {
            float t;
            if (sphere_roots(aa[p], b, disc, 0.01f, MRT_INF, t)) {
              if (t < best_t[p] || (t == best_t[p] && t < MRT_INF && obj < best_obj[p])) {
                best_t[p] = t; best_obj[p] = obj; best_prim[p] = sj; moved = true;
              }
            }
          }
        }
      }
      if (NW == 1 && __any(moved)) refresh_reach(); // (a team member only knows its own hits: it keeps +inf)
    };
    // PRIMARY RAYS FROM BINS.  The walk below asks, per tile, which primitives its cone may touch -- every tile walks the
    // same upper levels, and a team repeats them in each member.  bin_primitives_kernel asks the other way round, once per
    // frame: lane = primitive, which tiles' cones may touch its ball? -- and leaves a list per tile (the same cone, the
    // same test: the list holds exactly the primitives the walk's leaves would have passed).  A tile whose lists fit
    // takes its candidates from them; one whose lists overflowed (a horizon: thousands of sub-pixel triangles) walks.
    leaf_k = 0;
    if (binned_s) {
      const uint32_t cs = A.bin_cnt[BIN_HEAD + 2 * gtile], cn = A.bin_cnt[1];
      const uint32_t *list = A.bin_list + size_t(2 * gtile) * A.bin_cap;
      if (A.bin_clusters) {
        // the lists hold leaf clusters (scenes of more primitives than tiles): the walk's leaves without the walk --
        // which a team would otherwise repeat in each of its members
        auto leaves = [&](const uint32_t *l, uint32_t n) {
          for (uint32_t k0 = 0; k0 < n; k0 += 64) {
            const int mine = k0 + lane < n ? int(l[k0 + lane]) : 0;
            const int m = int(min(n - k0, 64u));
            for (int k = 0; k < m; ++k) {
              if (!MRT_TEAM_TAKES_LEAF()) continue;
              const int base = __builtin_amdgcn_readlane(mine, k) * 64;
              sphere_batch(base + lane, base + lane < ns, base, false);
            }
          }
        };
        leaves(list, cs);
        leaves(A.bin_near, cn);
      } else {
      for (uint32_t k0 = 0; k0 < cs; k0 += 64) {
        if (!MRT_TEAM_TAKES_LEAF()) continue;
        const bool have = k0 + lane < cs;
        sphere_batch(have ? int(list[k0 + lane]) : 0, have, -1, true);
      }
      for (uint32_t k0 = 0; k0 < cn; k0 += 64) { // spheres around or behind the camera: every tile's business
        if (!MRT_TEAM_TAKES_LEAF()) continue;
        const bool have = k0 + lane < cn;
        sphere_batch(have ? int(A.bin_near[k0 + lane]) : 0, have, -1, false);
      }
      }
    } else {
    walk_hierarchy<HIER>(sc.sph_sclu, sc.nscs, sc.sph_clu, sc.ncs, lane, sphere_ball_in_reach, [&](int base) {
      if (!MRT_TEAM_TAKES_LEAF()) return true;
      sphere_batch(base + lane, base + lane < ns, base, false);
      return true;
    });
    }

    // ---- closest hit over triangles: core/triangle.cuh:77-103 ----
    // one batch of up to 64 candidate triangles, lane = triangle i (pass: the lane has one): refinement when the
    // batch is crowded, then the exact test of every survivor for all pixels
    // (base >= 0: lane's triangle is base + lane, a leaf of the hierarchy; base < 0: the lanes' triangles come from a list)
    // PAIRS.  On a terrain seen at a grazing angle a tile's candidates are slivers that each cover a few of its 64 pixels
    // (1M triangles at 4K: 79 survive the refinement per tile, 43 of the tile's pixels hit anything), and the exact
    // test of ONE triangle for all 64 pixels spends 60 instructions on four useful lanes.  So the pixels keep lists:
    // the refinement hands back its edge functions at the tile's pixels (affine in column and row, with the error
    // bound that makes them conservative), every pixel evaluates them for each survivor -- six FMAs -- and notes the
    // triangles it cannot exclude; then every lane runs the exact test of ITS OWN next triangle, 64 different
    // (pixel, triangle) pairs per pass.  The closest-hit rule is an order-independent minimum over (t, object), so the
    // result is the same to the bit.  A team member notes pairs until a pixel's list is full or its leaves are dealt
    // with.  (Teams only: a wave on its own would want the hits after every batch for its depth bound, and the 25
    // registers this costs take a wave per SIMD from the scenes that have no triangles at all.)
    // the exact test of triangle tj (wave-uniform) for all pixels of the tile; true in the lanes it moved
    auto exact_pass = [&](int tj) __attribute__((always_inline)) {
      bool moved = false;
      const float *q = sc.tri_rec + 16 * size_t(tj);   // wave-uniform address: one s_load_dwordx16
      const float pnx = q[0], pny = q[1], pnz = q[2], kp = q[3];
      const float num = kp - dot3(cam.ox, cam.oy, cam.oz, pnx, pny, pnz);
      const int obj = sc.tri_obj[tj];
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
            best_t[p] = t; best_obj[p] = obj; best_prim[p] = ns + tj; moved = true;
          }
        }
      }
      return moved;
    };
    auto pk_flush = [&]() __attribute__((always_inline)) {
      if constexpr (PAIRS) {
        for (int s = 0; s < PK_CAP; ++s) {
          const bool act = s < pk_n;
          if (!__any(act)) break;
          if (act) {
            const int tj = int(pk_list[(wslot * PK_CAP + s) * 64 + lane]);
            const float4 *r4 = reinterpret_cast<const float4 *>(sc.tri_rec + 16 * size_t(tj)); // the lane's own record
            const float4 q0 = r4[0];
            const float num = q0.w - dot3(cam.ox, cam.oy, cam.oz, q0.x, q0.y, q0.z);
            const float rdn = dot3(dx[0], dy[0], dz[0], q0.x, q0.y, q0.z);
            const float t = num / rdn;
            if ((rdn != 0.0f) & (0.01f <= t) & (t <= best_t[0])) { // (a lane whose t cannot win skips the edges)
              const float4 q1 = r4[1], q2 = r4[2], q3 = r4[3];
              const float px = cam.ox + t * dx[0], py = cam.oy + t * dy[0], pz = cam.oz + t * dz[0];
              const float e0 = dot3(q1.x, q1.y, q1.z, px, py, pz) - q1.w;
              const float e1 = dot3(q2.x, q2.y, q2.z, px, py, pz) - q2.w;
              const float e2 = dot3(q3.x, q3.y, q3.z, px, py, pz) - q3.w;
              if ((e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (t <= MRT_INF)) {
                const int obj = sc.tri_obj[tj];
                if (t < best_t[0] || (t == best_t[0] && t < MRT_INF && obj < best_obj[0])) {
                  best_t[0] = t; best_obj[0] = obj; best_prim[0] = ns + tj;
                }
              }
            }
          }
        }
        pk_n = 0;
      }
    };
    auto triangle_batch = [&](int i, bool pass, int base) __attribute__((always_inline)) {
      bool moved = false;
      ConeBound CB;
      CB.cax = cax; CB.cay = cay; CB.caz = caz; CB.cosT = cosT; CB.sinT = sinT;
      CB.dcx = dcx; CB.dcy = dcy; CB.dcz = dcz;
      CB.ehx = ehx; CB.ehy = ehy; CB.ehz = ehz; CB.evx = evx; CB.evy = evy; CB.evz = evz;
      CB.cam_round = cam_round; CB.dlen_max = dlen_max; CB.reach_max = reach_max;
      if (PAIRS && pairs_on) {
        // the lane's triangle: its three functions at this tile (value at the tile's middle + error bound, step per
        // column, step per row); functions that hold everywhere when it has no set-up
        float F[9] = {3.0e38f, 0.0f, 0.0f, 3.0e38f, 0.0f, 0.0f, 3.0e38f, 0.0f, 0.0f};
        const int ti = base >= 0 ? base + lane : i;
        bool plain = false; // the lane's triangle has no set-up
        if (pass) {
          const float4 *su = tri_setup + 4 * size_t(ti);
          const float4 s3 = su[3];
          plain = s3.w == 0.0f;
          float own[3] = {0.0f, 0.0f, 0.0f};
          if (s3.w == 2.0f) pass = setup_bound_at_tile(sc, cam, CB, ti, own); // no bounding ball: the bound's second part from the tile
          if (!plain && pass) {
            if (s3.z * (1.0f - 1e-4f) > reach_max) pass = false; // beyond the tile's depth bound (NaN: never culls)
            // the tile's middle (column 3.5, row 3.5 of its pixels: the functions are affine) from the reference pixel
            const float ddx = (float(tile_x * TW) + 3.5f) - s3.x, ddy = (float(grow0) + 3.5f) - s3.y;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
              const float4 q = su[e];
              const float S = __builtin_fmaf(ddx, q.y, __builtin_fmaf(ddy, q.z, q.x));
              const float span = 3.5f * (fabsf(q.y) + fabsf(q.z)); // the pixels lie within 3.5 columns and rows of the middle
              // fp32 from the reference pixel to here (4e-7 of the terms) and from here to a pixel (2e-6: sixteen times what it can be)
              const float g = (q.w + own[e]) + __builtin_fmaf(4e-7f, fabsf(q.x) + fabsf(ddx * q.y) + fabsf(ddy * q.z), 2e-6f * (fabsf(S) + span));
              if (__builtin_fmaf(span, 1.0001f, S) + g < 0.0f) pass = false; // no pixel of the tile passes edge e
              F[3 * e] = S + g; F[3 * e + 1] = q.y; F[3 * e + 2] = q.z;
            }
          }
        }
        if (__any(plain)) { // (the loose triangles, mostly: they come in batches of their own)
          if (__popcll(__ballot(pass && plain)) >= TRI_REFINE_MIN && pass && plain)
            pass = base >= 0 ? triangle_may_be_hit<false>(sc, cam, CB, i) : triangle_may_be_hit<true>(sc, cam, CB, i);
        }
        const unsigned long long mask = __ballot(pass);
        const int total = __popcll(mask);
        n_clusters++; n_exact += total;
        // the survivors hand their functions and numbers to the pixels through LDS, PK_SLOTS of them at a time
        // ([wave][slot][12 floats]: a team's tile that one member renders alone still holds the whole workgroup's LDS,
        // and with 64 slots per wave that capped a CU at six tiles)
        const int rank = __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u)); // survivors below this lane
        for (int c0 = 0; c0 < total; c0 += PK_SLOTS) {
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // (the previous slots have been read)
          if (pass && rank >= c0 && rank < c0 + PK_SLOTS) {
            float4 *w = pk_fn + (wslot * PK_SLOTS + (rank - c0)) * 3;
            w[0] = make_float4(F[0], F[1], F[2], F[3]);
            w[1] = make_float4(F[4], F[5], F[6], F[7]);
            w[2] = make_float4(F[8], __int_as_float(ti), 0.0f, 0.0f);
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
          const int m = min(PK_SLOTS, total - c0);
          for (int k = 0; k < m; ++k) {
            const float4 *r = pk_fn + (wslot * PK_SLOTS + k) * 3; // one survivor: the same address in every lane
            const float4 f0 = r[0], f1 = r[1], f2 = r[2];
            const float e0 = __builtin_fmaf(pk_a, f0.y, __builtin_fmaf(pk_b, f0.z, f0.x));
            const float e1 = __builtin_fmaf(pk_a, f1.x, __builtin_fmaf(pk_b, f1.y, f0.w));
            const float e2 = __builtin_fmaf(pk_a, f1.w, __builtin_fmaf(pk_b, f2.x, f1.z));
            const bool covered = !(fminf(fminf(e0, e1), e2) < 0.0f); // (NaN: never excludes)
            const int ncov = __popcll(__ballot(covered));
            if (ncov > PK_FAT) {
              // most of the tile: one test for all pixels costs what one more entry in every list would, without the gathers
              moved |= exact_pass(__builtin_amdgcn_readfirstlane(__float_as_int(f2.y)));
            } else if (covered) {
              pk_list[(wslot * PK_CAP + pk_n) * 64 + lane] = (uint32_t)__float_as_int(f2.y);
              pk_n++;
            }
            if (__any(pk_n >= PK_CAP)) pk_flush(); // a full list
          }
        }
        if (NW == 1 && __any(moved)) refresh_reach();
        return;
      }
      if (HIER && __popcll(__ballot(pass)) >= TRI_REFINE_MIN && pass)
        pass = base >= 0 ? triangle_may_be_hit<false>(sc, cam, CB, i) : triangle_may_be_hit<true>(sc, cam, CB, i);
      unsigned long long mask = __ballot(pass);
      n_clusters++; n_exact += __popcll(mask);
      while (mask) {
        const int j = __builtin_ctzll(mask);
        mask &= mask - 1;
        moved |= exact_pass(base >= 0 ? base + j : __builtin_amdgcn_readlane(i, j)); // lane j's triangle
      }
      if (NW == 1 && __any(moved)) refresh_reach();
    };
    // The LOOSE triangles -- the tail of the triangle order (mrt_scene_build.h): long wedges and unbounded regions that
    // the reference's edge planes accept, whose bounding balls every tile's cone touches -- come from a list per
    // 64 x 64-pixel block instead (bin_loose_kernel, once per frame): the tile refines only what its block kept.
    const bool use_lists = HIER && A.loose_cnt != nullptr; // wave-uniform
    if (binned_t) {
      // the regular triangles from the tile's list (the loose ones follow from their block's list, or all of them)
      const uint32_t ct = A.bin_cnt[BIN_HEAD + 2 * gtile + 1], cn = A.bin_cnt[2];
      const uint32_t *list = A.bin_list + size_t(2 * gtile + 1) * A.bin_cap;
      if (A.bin_clusters) {
        auto leaves = [&](const uint32_t *l, uint32_t n) {
          for (uint32_t k0 = 0; k0 < n; k0 += 64) {
            const int mine = k0 + lane < n ? int(l[k0 + lane]) : 0;
            const int m = int(min(n - k0, 64u));
            for (int k = 0; k < m; ++k) {
              if (!MRT_TEAM_TAKES_LEAF()) continue;
              const int base = __builtin_amdgcn_readlane(mine, k) * 64, i = base + lane;
              triangle_batch(i, i < nt && tri_ball_in_reach(sc.tri_bx[i], sc.tri_by[i], sc.tri_bz[i], sc.tri_br[i]), base);
            }
          }
        };
        leaves(list, ct);
        leaves(A.bin_near + BIN_NEAR_CAP, cn);
      } else {
      for (uint32_t k0 = 0; k0 < ct; k0 += 64) {
        if (!MRT_TEAM_TAKES_LEAF()) continue;
        const bool have = k0 + lane < ct;
        triangle_batch(have ? int(list[k0 + lane]) : 0, have, -1);
      }
      for (uint32_t k0 = 0; k0 < cn; k0 += 64) { // triangles whose bound reaches round the camera
        if (!MRT_TEAM_TAKES_LEAF()) continue;
        const bool have = k0 + lane < cn;
        const int i = have ? int(A.bin_near[BIN_NEAR_CAP + k0 + lane]) : 0;
        triangle_batch(i, have && tri_ball_in_reach(sc.tri_bx[i], sc.tri_by[i], sc.tri_bz[i], sc.tri_br[i]), -1);
      }
      }
      if (!use_lists) // no block lists for the loose triangles: their clusters are walked
        for (int base = A.bin_nct_regular * 64; base < nt; base += 64) {
          if (!MRT_TEAM_TAKES_LEAF()) continue;
          const int i = base + lane;
          triangle_batch(i, i < nt && tri_ball_in_reach(sc.tri_bx[i], sc.tri_by[i], sc.tri_bz[i], sc.tri_br[i]), base);
        }
    } else {
    walk_hierarchy<HIER>(sc.tri_sclu, sc.nsct, sc.tri_clu, sc.nct, lane, tri_ball_in_reach, [&](int base) {
      if (!MRT_TEAM_TAKES_LEAF()) return true;
      const int i = base + lane;
      bool pass = false;
      if (i < nt) pass = tri_ball_in_reach(sc.tri_bx[i], sc.tri_by[i], sc.tri_bz[i], sc.tri_br[i]);
      triangle_batch(i, pass, base);
      return true;
    }, use_lists ? A.loose_nct_walk : -1);
    }
    if constexpr (HIER) {
      if (use_lists) {
        const int blk = (grow0 >> LOOSE_BLOCK_SHIFT) * A.loose_bx + ((tile_x * TW) >> LOOSE_BLOCK_SHIFT);
        const uint32_t cnt = A.loose_cnt[blk];
        const int first = A.loose_nct_walk * 64; // the first listed triangle
        if (cnt > (uint32_t)A.loose_cap) {
          // the block's list overflowed: its tiles take every loose triangle, batch by batch, as they used to
          for (int base = first; base < nt; base += 64) {
            if (!MRT_TEAM_TAKES_LEAF()) continue;
            triangle_batch(base + lane, base + lane < nt, base);
          }
        } else {
          const uint16_t *list = A.loose_list + size_t(blk) * A.loose_cap;
          for (uint32_t k0 = 0; k0 < cnt; k0 += 64) {
            if (!MRT_TEAM_TAKES_LEAF()) continue;
            const bool have = k0 + lane < cnt;
            triangle_batch(have ? first + int(list[k0 + lane]) : first, have, -1);
          }
        }
      }
    }

    pk_flush(); // (PAIRS: the pairs still noted)
    if (NW > 1 && !solo) {
      // the team's closest hit: the minimum of the members' (t, object) under the same rule as inside a member
      tm_t[member * 64 + lane] = best_t[0]; tm_obj[member * 64 + lane] = best_obj[0]; tm_prim[member * 64 + lane] = best_prim[0];
      __syncthreads();
      best_t[0] = MRT_INF; best_obj[0] = 0x7fffffff; best_prim[0] = -1;
#pragma unroll
      for (int w = 0; w < NWS; ++w) {
        const float t = tm_t[w * 64 + lane];
        const int obj = tm_obj[w * 64 + lane], prim = tm_prim[w * 64 + lane];
        if (prim >= 0 && (t < best_t[0] || (t == best_t[0] && t < MRT_INF && obj < best_obj[0]))) {
          best_t[0] = t; best_obj[0] = obj; best_prim[0] = prim;
        }
      }
      if (MULTI) __syncthreads(); // the next sample writes these slots again
    }

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
        if (valid[p] && member == 0) n_hit++;
      }
    }

    // ---- shadow rays + direct light: core/light.cuh:135-178 ----
    float lr[P], lg[P], lb[P];
    bool have_light[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { lr[p] = 0.0f; lg[p] = 0.0f; lb[p] = 0.0f; have_light[p] = false; }
    if (__any(any_hit_lane)) {
      bool diffuse_obj[P]; // the hit object's material kind: loaded once, used by every light
#pragma unroll
      for (int p = 0; p < P; ++p) diffuse_obj[p] = have_hit[p] && sc.mat_kind[best_obj[p]] == 0;
      // ball around the tile's hit points (any subset of them is covered too)
      Bundle B;
      {
        if (ns + nt > 64) {
          // many primitives: a tight ball pays for itself.  The hit points lie along the tile's rays between the nearest and
          // the farthest hit: centre = the tile's centre ray at the middle of that range (2 reductions; the centre of the
          // hit points' box took 6 for a ball of the same size, and a reduction is 13 vector instructions)
          const float big = 3.0e38f;
          float tlo = big, thi = -big;
#pragma unroll
          for (int p = 0; p < P; ++p)
            if (have_hit[p]) { tlo = fminf(tlo, best_t[p]); thi = fmaxf(thi, best_t[p]); }
          const float tmid = 0.5f * (wave_min(tlo) + wave_max(thi));
          B.mx = __builtin_fmaf(tmid, dcx, cam.ox); B.my = __builtin_fmaf(tmid, dcy, cam.oy); B.mz = __builtin_fmaf(tmid, dcz, cam.oz);
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
          e2 = fmaxf(e2, have_hit[p] ? bdot3(ex, ey, ez, ex, ey, ez) : 0.0f);
        }
        B.rho = fast_sqrt(wave_max(e2)) * CULL_REL + CULL_ABS;
      }
      auto bundle_vs_sphere_ball = [&](float x, float y, float z, float R) {
        const float wx = x - B.mx, wy = y - B.my, wz = z - B.mz;
        const float Lm = __builtin_fmaf(fast_sqrt(bdot3(wx, wy, wz, wx, wy, wz)), CULL_REL, B.rho + R);
        return bundle_may_touch(B, x, y, z, cluster_cull_radius(R, Lm));
      };
      auto bundle_vs_tri_ball = [&](float x, float y, float z, float R) { return bundle_may_touch(B, x, y, z, R); };

      // One light.  POINT is a compile-time switch: for a directional light the shadow ray direction, its
      // squared length and the light's normal are wave-uniform scalars, for a point light per-pixel registers.
      // ---- lights dealt among a team (LSPLIT), scenes of at most PRE_MAX_CLUSTERS clusters: what a light's shadow walk
      //      needs before its first leaf -- the bundle towards the light, its slab, the verdict on every cluster -- is
      //      computed for a member's lights of a fold chunk AT ONCE, lane = (light, cluster) pair, instead of once per
      //      light with most lanes idle (64 lights over 8 clusters: 128 set-ups and 512 ball tests per tile, a third
      //      of the kernel's instructions, become 16 lane-parallel passes).  The light loop fetches its light's values
      //      from the pair lanes with v_readlane.  The slab's width is bounded through the hit points' extents in the
      //      tile's own frame (view axis, two perpendiculars: three reductions per tile) instead of a reduction per
      //      light: |n.e| <= |n.a| Ea + |n.b1| E1 + |n.b2| E2 for every hit point e -- as conservative as before. ----
      // Scenes of at most 64 clusters (launcher: larger ones under many lights get teams that deal leaves).  A pass covers
      // 64 / (clusters rounded up to a power of two) lights and is computed when the light loop reaches its first light.
      const int ncl = sc.ncs + sc.nct;
      constexpr bool pre_lights = LSPLIT;
      int pre_shift = 0; // log2 of the lanes per light (clusters rounded up to a power of two)
      float fb1x = 0.0f, fb1y = 0.0f, fb1z = 0.0f, fb2x = 0.0f, fb2y = 0.0f, fb2z = 0.0f, fEa = 0.0f, fE1 = 0.0f, fE2 = 0.0f;
      if (pre_lights) {
        while ((1 << pre_shift) < ncl) ++pre_shift;
        // the frame: b1 = axis x (the coordinate axis least parallel to it), b2 = axis x b1
        const bool usex = fabsf(cax) < 0.7f;
        fb1x = usex ? 0.0f : -caz; fb1y = usex ? caz : 0.0f; fb1z = usex ? -cay : cax; // a x (1,0,0) = (0, az, -ay); a x (0,1,0) = (-az, 0, ax)
        fast_normalize(fb1x, fb1y, fb1z);
        fb2x = cay * fb1z - caz * fb1y; fb2y = caz * fb1x - cax * fb1z; fb2z = cax * fb1y - cay * fb1x;
        // wave-uniform values: keep them in scalar registers across the light loop
        fb1x = bcast(fb1x, 0); fb1y = bcast(fb1y, 0); fb1z = bcast(fb1z, 0);
        fb2x = bcast(fb2x, 0); fb2y = bcast(fb2y, 0); fb2z = bcast(fb2z, 0);
        float ea = 0.0f, e1 = 0.0f, e2 = 0.0f;
#pragma unroll
        for (int p = 0; p < P; ++p)
          if (have_hit[p]) {
            const float ex = hpx[p] - B.mx, ey = hpy[p] - B.my, ez = hpz[p] - B.mz;
            ea = fmaxf(ea, fabsf(bdot3(cax, cay, caz, ex, ey, ez)));
            e1 = fmaxf(e1, fabsf(bdot3(fb1x, fb1y, fb1z, ex, ey, ez)));
            e2 = fmaxf(e2, fabsf(bdot3(fb2x, fb2y, fb2z, ex, ey, ez)));
          }
        fEa = wave_max(ea); fE1 = wave_max(e1); fE2 = wave_max(e2);
      }
      // Which member takes which light: round robin.  (Member 0 also folds every light's contribution into the pixels;
      // dealing it half a share -- one of seven lights in a row, the others two each -- was slower, 1.17 against 1.04 ms
      // on 64 lights: a chunk of 16 lights ends at a barrier and takes as long as the member with most lights in it.)
      auto light_owner = [](int li) { return li % NWS; };
      auto light_place = [](int li) { return li / NWS; };
      auto member_light = [](int m, int n) { return m + NWS * n; };
      // lights per pass: what fits the wave's lanes, at most PRE_LPP (the LDS slots of a member)
      const int pre_lpp = (64 >> pre_shift) < PRE_LPP ? (64 >> pre_shift) : PRE_LPP;
      float *const my_pre = tm_pre + (LSPLIT ? member * PRE_LPP * 16 : 0);
      unsigned long long pre_mask = 0; // bit = lane = pair: the cluster may be touched by the light's bundle
      auto precompute_pass = [&](int li0) {
        // this member's lights from its li0-th on, j = 0 .. pre_lpp - 1
        const int j = lane >> pre_shift, c = lane & ((1 << pre_shift) - 1);
        const int li = member_light(member, li0 + j);
        const bool okl = j < pre_lpp && li < sc.nl;
        Bundle Bl = B;
        Bl.point = false; Bl.Dx = 0.0f; Bl.Dy = 0.0f; Bl.Dz = 0.0f;
        float plx = 0.0f, ply = 0.0f, plz = 0.0f, plr = 0.0f, plg = 0.0f, plb = 0.0f; // the pair's light: position / direction, colour
        int plk = 0;                                                                     // ... and kind
        if (okl) {
          const float lvx = sc.light_vec[3 * li], lvy = sc.light_vec[3 * li + 1], lvz = sc.light_vec[3 * li + 2];
          plx = lvx; ply = lvy; plz = lvz; plk = sc.light_kind[li];
          plr = sc.light_rgb[3 * li]; plg = sc.light_rgb[3 * li + 1]; plb = sc.light_rgb[3 * li + 2];
          Bl.point = plk != 0;
          if (!Bl.point) { Bl.Dx = -lvx; Bl.Dy = -lvy; Bl.Dz = -lvz; }
          else { Bl.Dx = lvx - B.mx; Bl.Dy = lvy - B.my; Bl.Dz = lvz - B.mz; }
        }
        Bl.DD = bdot3(Bl.Dx, Bl.Dy, Bl.Dz, Bl.Dx, Bl.Dy, Bl.Dz);
        {
          float nx = __builtin_fmaf(cay, Bl.Dz, -(caz * Bl.Dy)), ny = __builtin_fmaf(caz, Bl.Dx, -(cax * Bl.Dz)), nz = __builtin_fmaf(cax, Bl.Dy, -(cay * Bl.Dx));
          const float nn = bdot3(nx, ny, nz, nx, ny, nz);
          Bl.snx = 0.0f; Bl.sny = 0.0f; Bl.snz = 0.0f; Bl.sw = MRT_INF;
          if (nn > 2.5e-3f * Bl.DD) { // as bundle_make_slab: below ~3 degrees between axis and light the normal is ill-conditioned
            const float inv = fast_rsq(nn);
            nx *= inv; ny *= inv; nz *= inv;
            const float w = __builtin_fmaf(fabsf(bdot3(nx, ny, nz, cax, cay, caz)), fEa, __builtin_fmaf(fabsf(bdot3(nx, ny, nz, fb1x, fb1y, fb1z)), fE1,
                                 fabsf(bdot3(nx, ny, nz, fb2x, fb2y, fb2z)) * fE2));
            Bl.snx = nx; Bl.sny = ny; Bl.snz = nz;
            Bl.sw = __builtin_fmaf(w, CULL_REL * 1.001f, CULL_ABS) + (Bl.point ? 1e-5f * norm1(Bl.Dx, Bl.Dy, Bl.Dz) : 0.0f);
          }
        }
        bool pass = false;
        // (the cluster's number through an opaque copy: otherwise the eight 64-bit per-lane addresses below are hoisted
        // out of the light loop and kept -- in scratch -- across it)
        unsigned cu = (unsigned)c;
        asm volatile("" : "+v"(cu));
        if (okl && c < ncl) {
          if (c < sc.ncs) {
            const unsigned n = (unsigned)sc.ncs;
            const float x = sc.sph_clu[cu], y = sc.sph_clu[n + cu], z = sc.sph_clu[2u * n + cu], R = sc.sph_clu[3u * n + cu];
            const float wx = x - B.mx, wy = y - B.my, wz = z - B.mz;
            const float Lm = __builtin_fmaf(fast_sqrt(bdot3(wx, wy, wz, wx, wy, wz)), CULL_REL, B.rho + R);
            pass = bundle_may_touch(Bl, x, y, z, cluster_cull_radius(R, Lm));
          } else {
            const unsigned k = cu - (unsigned)sc.ncs, n = (unsigned)sc.nct;
            pass = bundle_may_touch(Bl, sc.tri_clu[k], sc.tri_clu[n + k], sc.tri_clu[2u * n + k], sc.tri_clu[3u * n + k]);
          }
        }
        pre_mask = __ballot(pass);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // (the previous pass's slots have been read)
        if (okl && c == 0) { // the light's first pair lane leaves what every pair of the light computed
          float4 *q = reinterpret_cast<float4 *>(my_pre + j * 16);
          q[0] = make_float4(Bl.Dx, Bl.Dy, Bl.Dz, Bl.DD);
          q[1] = make_float4(Bl.snx, Bl.sny, Bl.snz, Bl.sw);
          q[2] = make_float4(plx, ply, plz, __int_as_float(plk));
          q[3] = make_float4(plr, plg, plb, 0.0f);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      };
      auto shade_light = [&](auto point_tag, int li, float lvx, float lvy, float lvz, int pj) {
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
        unsigned long long pre_bits = 0; // clusters the light's bundle may touch (spheres first), when pre-computed
        if constexpr (LSPLIT) {
          const int slot = pj & (pre_lpp - 1);
          const float4 *q = reinterpret_cast<const float4 *>(my_pre + slot * 16); // (one address for the wave: broadcast reads)
          const float4 q0 = q[0], q1 = q[1];
          B.Dx = bcast(q0.x, 0); B.Dy = bcast(q0.y, 0); B.Dz = bcast(q0.z, 0); B.DD = bcast(q0.w, 0);
          B.snx = bcast(q1.x, 0); B.sny = bcast(q1.y, 0); B.snz = bcast(q1.z, 0); B.sw = bcast(q1.w, 0);
          pre_bits = (pre_mask >> (slot << pre_shift)) & (ncl >= 64 ? ~0ull : (1ull << ncl) - 1ull);
#if defined(MRT_ABL_LIGHTS) && MRT_ABL_LIGHTS == 1 // timing only (wrong image): no shadow leaves at all
          pre_bits = 0;
#endif
        } else {
        B.DD = bdot3(B.Dx, B.Dy, B.Dz, B.Dx, B.Dy, B.Dz);
        {
          // slab of the bundle for this light (bundle_make_slab, here with P hit points per lane)
          float nx = cay * B.Dz - caz * B.Dy, ny = caz * B.Dx - cax * B.Dz, nz = cax * B.Dy - cay * B.Dx;
          const float nn = bdot3(nx, ny, nz, nx, ny, nz);
          B.snx = 0.0f; B.sny = 0.0f; B.snz = 0.0f; B.sw = MRT_INF;
          if (nn > 2.5e-3f * B.DD) {
            const float inv = fast_rsq(nn);
            nx *= inv; ny *= inv; nz *= inv;
            float dmax = 0.0f;
#pragma unroll
            for (int p = 0; p < P; ++p)
              dmax = fmaxf(dmax, have_hit[p] ? fabsf(bdot3(nx, ny, nz, hpx[p] - B.mx, hpy[p] - B.my, hpz[p] - B.mz)) : 0.0f);
            B.snx = nx; B.sny = ny; B.snz = nz;
            B.sw = wave_max(dmax) * CULL_REL + CULL_ABS + (B.point ? 1e-5f * norm1(B.Dx, B.Dy, B.Dz) : 0.0f);
          }
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
        const bool team_need = __any(any_open); // from the merged hit records: the same in every member of a team
        leaf_k = 0;
        auto shadow_sphere_leaf = [&](int base) {
            if (!MRT_TEAM_TAKES_SHADOW_LEAF()) return true;
            const int i = base + lane;
            float cx = 0.0f, cy = 0.0f, cz = 0.0f, r = 0.0f;
            bool pass = false;
            if (i < ns) {
              cx = sc.sph_cx[i]; cy = sc.sph_cy[i]; cz = sc.sph_cz[i]; r = sc.sph_r[i];
              const float wx = cx - B.mx, wy = cy - B.my, wz = cz - B.mz;
              const float Lm = __builtin_fmaf(fast_sqrt(bdot3(wx, wy, wz, wx, wy, wz)), CULL_REL, B.rho); // >= |hit point - centre|
              pass = bundle_may_touch(B, cx, cy, cz, sphere_cull_radius(r * r, Lm * Lm));
            }
            unsigned long long mask = __ballot(pass);
            n_clusters++; n_exact += __popcll(mask);
#if defined(MRT_ABL_LIGHTS) && MRT_ABL_LIGHTS == 2 // timing only: leaf bounds, no exact tests
            if (LSPLIT) mask = 0;
#endif
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
          };
        if (team_need) {
          if constexpr (LSPLIT) { // the clusters this light's bundle may touch were found with the pass's other lights
            unsigned long long cm = pre_bits & (sc.ncs >= 64 ? ~0ull : (1ull << sc.ncs) - 1ull);
            while (cm) {
              const int c = __builtin_ctzll(cm);
              cm &= cm - 1;
              if (!shadow_sphere_leaf(c * 64)) break;
            }
          } else {
            walk_hierarchy<HIER>(sc.sph_sclu, sc.nscs, sc.sph_clu, sc.ncs, lane, bundle_vs_sphere_ball, shadow_sphere_leaf);
          }
        }
        any_open = false;
#pragma unroll
        for (int p = 0; p < P; ++p) any_open |= open[p];
        // any-hit over triangles (a member whose sphere walk ended early has met fewer leaves: count afresh)
        leaf_k = 0;
        auto shadow_triangle_leaf = [&](int base) {
            if (!MRT_TEAM_TAKES_SHADOW_LEAF()) return true;
            const int i = base + lane;
            bool pass = false;
            if (i < nt) pass = bundle_may_touch(B, sc.tri_bx[i], sc.tri_by[i], sc.tri_bz[i], sc.tri_br[i]);
            unsigned long long mask = __ballot(pass);
            n_clusters++; n_exact += __popcll(mask);
#if defined(MRT_ABL_LIGHTS) && MRT_ABL_LIGHTS == 2
            if (LSPLIT) mask = 0;
#endif
            while (mask) {
              const int j = __builtin_ctzll(mask);
              mask &= mask - 1;
              const float *q = sc.tri_rec + 16 * size_t(base + j);
              const float pnx = q[0], pny = q[1], pnz = q[2], kp = q[3];
              bool still = false;
#pragma unroll
              for (int p = 0; p < P; ++p) {
                const float rdn = dot3(sdx[ix(p)], sdy[ix(p)], sdz[ix(p)], pnx, pny, pnz);
                const float snum = kp - dot3(hpx[p], hpy[p], hpz[p], pnx, pny, pnz);
                const float t = snum / rdn;
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
          };
        if (__any(any_open)) {
          if constexpr (LSPLIT) {
            unsigned long long cm = sc.ncs >= 64 ? 0ull : pre_bits >> sc.ncs;
            while (cm) {
              const int c = __builtin_ctzll(cm);
              cm &= cm - 1;
              if (!shadow_triangle_leaf(c * 64)) break;
            }
          } else {
            walk_hierarchy<HIER>(sc.tri_sclu, sc.nsct, sc.tri_clu, sc.nct, lane, bundle_vs_tri_ball, shadow_triangle_leaf);
          }
        }

        if constexpr (NW > 1 && !LSPLIT) {
          if (team_need && !solo) {
            // a shadow ray is free only if no member found an occluder: AND of the members' masks.  The two sets of
            // slots alternate between the lights that GET HERE (team_need is the same in every member), so one barrier
            // per such light is enough: whoever writes a set again has passed the barrier of the light in between,
            // which every reader of the set reaches only after its reads.  (Alternating by the light's number would
            // let a light that skips the barrier put two users of one set back to back.)
            const unsigned long long mine = __ballot(open[0]);
            if (lane == 0) tm_open[open_set * NWS + member] = mine;
            __syncthreads();
            unsigned long long all = ~0ull;
#pragma unroll
            for (int w = 0; w < NWS; ++w) all &= tm_open[open_set * NWS + w];
            open[0] = (all >> lane) & 1ull;
            open_set ^= 1;
          }
          if (member != 0) return; // member 0 shades
        }
        float lcr, lcg, lcb;
        if constexpr (LSPLIT) {
          const float4 q3 = reinterpret_cast<const float4 *>(my_pre + (pj & (pre_lpp - 1)) * 16)[3];
          lcr = bcast(q3.x, 0); lcg = bcast(q3.y, 0); lcb = bcast(q3.z, 0);
        }
        else { lcr = sc.light_rgb[3 * li]; lcg = sc.light_rgb[3 * li + 1]; lcb = sc.light_rgb[3 * li + 2]; }
#pragma unroll
        for (int p = 0; p < P; ++p) {
          if (need[p] && valid[p]) { n_shadow++; if (!open[p]) n_occl++; }
          if (open[p]) {
            // light_bounce with the LIGHT's normal, as the reference passes it (core/light.cuh:164-168)
            const float dn = dot3(sdx[ix(p)], sdy[ix(p)], sdz[ix(p)], lnx[ix(p)], lny[ix(p)], lnz[ix(p)]);
            float cr, cg, cb;
            if (diffuse_obj[p]) { // DiffuseMaterial::light_bounce, core/material.cuh:63-75
              const float scale = -dn;
              if (scale > 0.0f) { cr = scale * lcr; cg = scale * lcg; cb = scale * lcb; }
              else { cr = 0.0f; cg = 0.0f; cb = 0.0f; }
            } else {                              // MetallicMaterial::light_bounce, core/material.cuh:134-158
              const float s2 = 2.0f * dn;
              const float rx = sdx[ix(p)] - s2 * lnx[ix(p)], ry = sdy[ix(p)] - s2 * lny[ix(p)], rz = sdz[ix(p)] - s2 * lnz[ix(p)];
              if (dot3(rx, ry, rz, dx[p], dy[p], dz[p]) < -0.99f) { cr = lcr; cg = lcg; cb = lcb; }
              else { cr = 0.001f; cg = 0.001f; cb = 0.001f; }
            }
            if constexpr (LSPLIT) {
              tm_c[(0 * LCH + (li & (LCH - 1))) * 64 + lane] = cr;
              tm_c[(1 * LCH + (li & (LCH - 1))) * 64 + lane] = cg;
              tm_c[(2 * LCH + (li & (LCH - 1))) * 64 + lane] = cb;
            } else {
              if (have_light[p]) { lr[p] += cr; lg[p] += cg; lb[p] += cb; }
              else { lr[p] = cr; lg[p] = cg; lb[p] = cb; have_light[p] = true; }
            }
          }
        }
        if constexpr (LSPLIT) {
          const unsigned long long lit = __ballot(open[0]);
          if (lane == 0) tm_lit[li & (LCH - 1)] = lit;
        }
      };
      for (int li = 0; li < sc.nl; ++li) {
        if (!LSPLIT || light_owner(li) == member) {
          const int pj = LSPLIT ? light_place(li) : -1; // the light's place among this member's lights
          float lvx, lvy, lvz;
          int lk;
          if constexpr (LSPLIT) {
            if ((pj & (pre_lpp - 1)) == 0) precompute_pass(pj); // the pass that starts with this light
            const float4 q2 = reinterpret_cast<const float4 *>(my_pre + (pj & (pre_lpp - 1)) * 16)[2]; // the light itself
            lvx = bcast(q2.x, 0); lvy = bcast(q2.y, 0); lvz = bcast(q2.z, 0);
            lk = __builtin_amdgcn_readfirstlane(__float_as_int(q2.w));
          } else {
            lvx = sc.light_vec[3 * li]; lvy = sc.light_vec[3 * li + 1]; lvz = sc.light_vec[3 * li + 2];
            lk = sc.light_kind[li];
          }
          if (lk == 0) shade_light(std::false_type{}, li, lvx, lvy, lvz, pj);
          else shade_light(std::true_type{}, li, lvx, lvy, lvz, pj);
        }
        if constexpr (LSPLIT) {
          if ((li & (LCH - 1)) == LCH - 1 || li == sc.nl - 1) {
            __syncthreads(); // the chunk's contributions are in LDS
            if (member == 0) {
              // result += light_bounce(...), in light order: core/light.cuh:171-173.  Four lights at a time, their slots
              // read without a branch (a light that left nothing for a lane has its bit clear: the slot's old content is
              // read and dropped) so that the twelve LDS reads are in flight together instead of one light after another
              static_assert(LCH % 4 == 0, "the fold takes four lights at a time");
              for (int k0 = li & ~(LCH - 1); k0 <= li; k0 += 4) {
                float c[4][3];
                bool lit[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                  const int kk = (k0 + u) & (LCH - 1);
                  lit[u] = k0 + u <= li && ((tm_lit[kk] >> lane) & 1ull);
                  c[u][0] = tm_c[(0 * LCH + kk) * 64 + lane]; c[u][1] = tm_c[(1 * LCH + kk) * 64 + lane]; c[u][2] = tm_c[(2 * LCH + kk) * 64 + lane];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                  const float nr = have_light[0] ? lr[0] + c[u][0] : c[u][0], ng = have_light[0] ? lg[0] + c[u][1] : c[u][1],
                              nb = have_light[0] ? lb[0] + c[u][2] : c[u][2];
                  lr[0] = lit[u] ? nr : lr[0]; lg[0] = lit[u] ? ng : lg[0]; lb[0] = lit[u] ? nb : lb[0];
                  have_light[0] = have_light[0] || lit[u];
                }
              }
            }
            __syncthreads(); // the slots are free for the next chunk
          }
        }
      }
    }

    // ---- combine: src/main.cu:91-104 with MAX_RECURSION_DEPTH = 0 ----
    if (NW > 1 && member != 0) continue;
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
  if (NW > 1 && member != 0) {
    // the other members only report the work they did
  } else if (P > 1 && A.wide && valid_row && col0 + P <= A.W) {
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

#ifdef MRT_TIMELINE
  if (tl_idx < TLC_MAX_WAVES && lane == 0 && member == 0) {
    unsigned long long *tlp = g_timeline_cull + (size_t)TLC_WORDS * tl_idx;
    tlp[0] = tl_rt0; tlp[1] = __builtin_amdgcn_s_memrealtime(); tlp[2] = __builtin_amdgcn_s_memtime() - tl_c0;
    tlp[3] = (unsigned long long)n_exact | ((unsigned long long)n_clusters << 32);
  }
#endif
  if (A.stats) {
    // counters are spread over stats_slots copies: same-address atomics cost ~12 ns each, one after the other
    unsigned long long *st = A.stats + 8u * ((blockIdx.x + 7u * blockIdx.y) & (unsigned)(A.stats_slots - 1));
    unsigned int nvalid = 0;
#pragma unroll
    for (int p = 0; p < P; ++p) nvalid += valid[p] ? 1u : 0u;
    unsigned int v4[4] = {nvalid * (unsigned)(MULTI ? A.spp : 1), n_shadow, n_hit, n_occl};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // a team's pixels and hits are counted by member 0; shadow rays by whoever traced them (member 0, or with
      // the lights dealt out every member for its own lights)
      if (NW > 1 && member != 0 && !(LSPLIT && (k == 1 || k == 3))) continue;
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

// ---- the loose triangles' lists: which of them can a ray of each 64 x 64-pixel block meet? ---------------------
// One wave per (block, 64 loose triangles), lane = triangle: the refinement the tiles use (triangle_may_be_hit: plane
// against the block's cone, edge planes in screen space over the padded block), without a depth bound.  A block's
// pyramid holds the pyramids of its 64 tiles, padding included, so whatever a tile could not exclude its block cannot
// exclude either.  Survivors are appended to the block's list (one atomic per wave); a list that outgrows its
// capacity only keeps counting, and the block's tiles then take every loose triangle as before.
__global__ __launch_bounds__(256) void bin_loose_kernel(RenderArgs A, float block_cone_ext, uint16_t *__restrict__ list,
                                                        uint32_t *__restrict__ cnt)
{
  const DevScene &sc = A.sc;
  const DevCamera &cam = A.cam;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int blk = blockIdx.x, by = blk / A.loose_bx, bx = blk - by * A.loose_bx;
  const int first = A.loose_nct_walk * 64;
  const int i = first + (int(blockIdx.y) * 4 + wave) * 64 + lane;
  if (first + (int(blockIdx.y) * 4 + wave) * 64 >= sc.nt) return;
  ConeBound C;
  {
    const float rw = fast_rcp(float(A.W)), rh = fast_rcp(float(A.H));
    const float um = (float(bx * LOOSE_BLOCK) + 0.5f * float(LOOSE_BLOCK - 1)) * rw, vm = (float(by * LOOSE_BLOCK) + 0.5f * float(LOOSE_BLOCK - 1)) * rh;
    C.dcx = ((cam.ux + um * cam.hx) + vm * cam.vx) - cam.ox;
    C.dcy = ((cam.uy + um * cam.hy) + vm * cam.vy) - cam.oy;
    C.dcz = ((cam.uz + um * cam.hz) + vm * cam.vz) - cam.oz;
    C.cam_round = 2.0f * camera_rounding(cam);
    const float du = (0.5f * float(LOOSE_BLOCK) + 0.51f) * rw * 1.001f, dv = (0.5f * float(LOOSE_BLOCK) + 0.51f) * rh * 1.001f; // pixels -1 .. 64
    C.ehx = du * cam.hx; C.ehy = du * cam.hy; C.ehz = du * cam.hz;
    C.evx = dv * cam.vx; C.evy = dv * cam.vy; C.evz = dv * cam.vz;
    const float dd = bdot3(C.dcx, C.dcy, C.dcz, C.dcx, C.dcy, C.dcz);
    const float inv = fast_rsq(dd);
    C.cax = C.dcx * inv; C.cay = C.dcy * inv; C.caz = C.dcz * inv;
    C.sinT = block_cone_ext * inv * (1.0f + 1e-4f) + 1e-6f;
    C.cosT = fast_sqrt(fmaxf(1.0f - C.sinT * C.sinT, 0.0f)) * (1.0f - 1e-5f) - 1e-6f;
    if (!(C.sinT < 0.99f) || !(C.cosT > 0.0f)) { C.cosT = -1.0f; C.sinT = 0.0f; } // cannot cull
    // no ray of the padded block is longer than the centre ray plus the two half extents
    C.dlen_max = (fast_sqrt(dd) + fast_sqrt(bdot3(C.ehx, C.ehy, C.ehz, C.ehx, C.ehy, C.ehz)) +
                  fast_sqrt(bdot3(C.evx, C.evy, C.evz, C.evx, C.evy, C.evz))) * (1.0f + 1e-4f);
    C.reach_max = MRT_INF;
  }
  const bool pass = i < sc.nt && triangle_may_be_hit(sc, cam, C, i);
  const unsigned long long mask = __ballot(pass);
  if (!mask) return;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(&cnt[blk], (uint32_t)__popcll(mask));
  base = __builtin_amdgcn_readfirstlane(base);
  if (pass) {
    const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (pos < (uint32_t)A.loose_cap) list[size_t(blk) * A.loose_cap + pos] = (uint16_t)(i - first);
  }
}

// entries per block: lists longer than this send the block's tiles back to the loose clusters themselves
#ifndef MRT_LOOSE_CAP
#define MRT_LOOSE_CAP 2048
#endif
// Lists pay when there are loose clusters to skip, few enough loose triangles for 16-bit entries, and frames of
// more than a few blocks.  MRT_LOOSE_LISTS=0 switches them off (ablation).
static bool loose_lists_apply(const RenderArgs &A)
{
  static const bool on = !(getenv("MRT_LOOSE_LISTS") && atoi(getenv("MRT_LOOSE_LISTS")) == 0);
  static const long least = getenv("MRT_LOOSE_MIN") ? atol(getenv("MRT_LOOSE_MIN")) : 256; // (stress runs: 1)
  const long nloose = (long)A.sc.nt - (long)A.sc.nct_regular * 64;
  if (opt_lists(A) == 1) return false; // MRT_OPT_LISTS: every tile walks
  return on && A.sc.nct > 1 && A.sc.nct_regular < A.sc.nct && nloose >= (opt_list_cap(A) > 0 ? 1 : least) && nloose <= 65536 && A.spp == 1;
}
static size_t loose_bytes(const RenderArgs &A)
{
  if (!loose_lists_apply(A)) return 0;
  const size_t blocks = size_t((A.W + LOOSE_BLOCK - 1) / LOOSE_BLOCK) * size_t((A.H + LOOSE_BLOCK - 1) / LOOSE_BLOCK);
  return (((blocks * 4 + 255) / 256) * 256 + blocks * MRT_LOOSE_CAP * 2 + 255) / 256 * 256;
}

// ---- the tiles' bins: which tiles can a primitive's bound be seen from? ---------------------------------------------
// lane = primitive (spheres first, then the regular triangles).  Its bounding ball, inflated exactly as the tile kernel
// inflates it, projects onto a rectangle of the image (closed form: the two tangent planes through the camera that are
// parallel to the image's vertical give the extent in u, and likewise in v); over that rectangle's tiles, plus one tile
// all round -- a tile's cone is the circle around its padded square and reaches up to four pixels past it -- the lane
// runs the tile kernel's own test, the tile's own cone (make_tile_cone) against the ball, and appends itself to the
// lists of the tiles that pass.  So a tile's list holds exactly the primitives its walk would have handed to the exact
// tests.  Balls that reach round or behind the camera (no rectangle), balls without a finite bound and rectangles of
// more than BIN_MAX_RECT tiles go to the frame's near list, which every tile takes; if that overflows, word 0 of the
// counters says so and the frame is rendered by the walk.  Lists that outgrow bin_cap keep counting: their tiles walk.
struct BinCamera {
  float hx, hy, hz, vx, vy, vz, nx, ny, nz; // unit vectors along the image's horizontal and vertical, and into the scene
  float bh, bv, bn;                         // the image's corner (u = v = 0) in that frame; bn > 0
  float lh, lv;                             // lengths of the horizontal and vertical edges
};
enum { BIN_MAX_RECT = 4096, BIN_DIRECT = 4, BIN_QUEUE_CAP = 1 << 18 };
// the ball of item g (spheres first, then triangles; A.bin_clusters: leaf clusters of spheres, then of triangles) as the
// tile kernel tests it against a cone: centre - origin, radius
__device__ __forceinline__ int bin_items0(const RenderArgs &A) { return A.bin_clusters ? A.sc.ncs : A.sc.ns; }
__device__ __forceinline__ int bin_items1(const RenderArgs &A) { return A.bin_clusters ? A.bin_nct_regular : min(A.sc.nt, A.bin_nct_regular * 64); }
__device__ __forceinline__ void bin_ball(const RenderArgs &A, int g, float &wx, float &wy, float &wz, float &reff)
{
  const DevScene &sc = A.sc;
  const DevCamera &cam = A.cam;
  if (A.bin_clusters) {
    if (g < sc.ncs) {
      const float R = sc.sph_clu[3 * sc.ncs + g];
      wx = sc.sph_clu[g] - cam.ox; wy = sc.sph_clu[sc.ncs + g] - cam.oy; wz = sc.sph_clu[2 * sc.ncs + g] - cam.oz;
      reff = cluster_cull_radius(R, fast_sqrt(bdot3(wx, wy, wz, wx, wy, wz)) * CULL_REL + R); // as cone_vs_sphere_ball
    } else {
      const int k = g - sc.ncs;
      wx = sc.tri_clu[k] - cam.ox; wy = sc.tri_clu[sc.nct + k] - cam.oy; wz = sc.tri_clu[2 * sc.nct + k] - cam.oz;
      reff = sc.tri_clu[3 * sc.nct + k] * CULL_REL + CULL_ABS;                              // as cone_vs_tri_ball
    }
  } else if (g < sc.ns) {
    const float r = sc.sph_r[g];
    wx = sc.sph_cx[g] - cam.ox; wy = sc.sph_cy[g] - cam.oy; wz = sc.sph_cz[g] - cam.oz;
    reff = sphere_cull_radius(r * r, bdot3(wx, wy, wz, wx, wy, wz)); // as sphere_batch
  } else {
    const int i = g - sc.ns;
    wx = sc.tri_bx[i] - cam.ox; wy = sc.tri_by[i] - cam.oy; wz = sc.tri_bz[i] - cam.oz;
    reff = sc.tri_br[i] * CULL_REL + CULL_ABS;                      // as cone_vs_tri_ball
  }
}
// primitive g against tile (tx, ty): the tile kernel's own test; a pass appends g to the tile's list of its type
__device__ __forceinline__ void bin_try_tile(const RenderArgs &A, int TW, int tiles_x, int g, float wx, float wy, float wz, float reff,
                                             int tx, int ty, uint32_t *__restrict__ cnt, uint32_t *__restrict__ list)
{
  // a rank of a partitioned frame only lists the tiles of its own strips (strips are multiples of 8 rows: a tile row
  // belongs to one strip)
  if (A.nranks > 1 && !owns_strip((ty * 8) / A.strip_rows, A.rank, A.nranks)) return;
  const TileCone t = make_tile_cone(A.cam, A.W, A.H, A.cone_ext, tx, TW, ty * 8);
  if (!ball_may_touch_cone(wx, wy, wz, reff, t.cax, t.cay, t.caz, t.cosT, t.sinT)) return;
  const int n0 = bin_items0(A);
  const bool tri = g >= n0;
  const int slot = 2 * (ty * tiles_x + tx) + (tri ? 1 : 0);
  const uint32_t pos = atomicAdd(&cnt[BIN_HEAD + slot], 1u);
  if (pos < (uint32_t)A.bin_cap) list[size_t(slot) * A.bin_cap + pos] = (uint32_t)(tri ? g - n0 : g);
}

// Pass 1, lane = primitive: the rectangle; small ones (up to BIN_DIRECT tiles: the million sub-pixel triangles of a
// terrain) are binned on the spot, larger ones are cut into blocks of 8 x 8 tiles and queued for pass 2 -- a handful of
// large spheres would otherwise keep a few lanes looping over thousands of tiles while the GPU idles.
__global__ __launch_bounds__(256) void bin_primitives_kernel(RenderArgs A, BinCamera C, int TW, uint32_t *__restrict__ cnt,
                                                             uint32_t *__restrict__ list, uint32_t *__restrict__ near_list,
                                                             uint2 *__restrict__ queue)
{
  const int g = blockIdx.x * 256 + threadIdx.x;
  const int ns = bin_items0(A), ntr = bin_items1(A); // (items of the first and of the second kind)
  const bool live = g < ns + ntr; // (lanes past the last primitive stay: the wave shares the larger rectangles out)
  const bool tri = g >= ns;
  float wx = 0.0f, wy = 0.0f, wz = 0.0f, reff = 0.0f;
  if (live) bin_ball(A, g, wx, wy, wz, reff);
  const int tiles_x = (A.W + TW - 1) / TW, tiles_y = (A.H + 7) >> 3;
  int tx0 = 0, tx1 = -1, ty0 = 0, ty1 = -1;
  bool near = live;
  if (live) {
    const float X = bdot3(wx, wy, wz, C.hx, C.hy, C.hz), Y = bdot3(wx, wy, wz, C.vx, C.vy, C.vz), Z = bdot3(wx, wy, wz, C.nx, C.ny, C.nz);
    const float L1 = norm1(wx, wy, wz);
    const float R = reff * 1.02f + 1e-5f * L1 + 1e-6f; // the test's slack (CULL_LEN_SLACK of |w|_1) and this block's own rounding
    if (Z > R * 1.01f + 1e-5f * L1 && R < 3.0e37f) {   // the ball lies in front of the camera plane, clear of it (NaN: near)
      const float den = Z * Z - R * R;                 // > 0
      const float sx = R * fast_sqrt(fmaxf(X * X + den, 0.0f)), sy = R * fast_sqrt(fmaxf(Y * Y + den, 0.0f));
      const float iden = fast_rcp(den);
      // tangents of the extreme directions, then u = (x bn - bh) / lh and v likewise
      const float xlo = (X * Z - sx) * iden, xhi = (X * Z + sx) * iden, ylo = (Y * Z - sy) * iden, yhi = (Y * Z + sy) * iden;
      const float ulo = (xlo * C.bn - C.bh) / C.lh, uhi = (xhi * C.bn - C.bh) / C.lh;
      const float vlo = (ylo * C.bn - C.bv) / C.lv, vhi = (yhi * C.bn - C.bv) / C.lv;
      // pixels, padded by what a tile's cone reaches beyond the tile (it is the circle around the padded square of 10 x 10
      // pixels, radius 7.1, the tile itself reaches 4 from its centre: 3.1), a pixel and 1e-4 of the frame
      const float pw = 1e-4f * float(A.W) + 4.2f, ph = 1e-4f * float(A.H) + 4.2f;
      const float c0 = ulo * float(A.W) - pw, c1 = uhi * float(A.W) + pw, r0 = vlo * float(A.H) - ph, r1 = vhi * float(A.H) + ph;
      if (c0 == c0 && c1 == c1 && r0 == r0 && r1 == r1) { // no NaN
        const float big = 1.0e7f;
        tx0 = int(floorf(fminf(fmaxf(c0, -big), big) / float(TW))); tx1 = int(floorf(fminf(fmaxf(c1, -big), big) / float(TW)));
        ty0 = int(floorf(fminf(fmaxf(r0, -big), big) * 0.125f));   ty1 = int(floorf(fminf(fmaxf(r1, -big), big) * 0.125f));
        tx0 = max(tx0, 0); ty0 = max(ty0, 0); tx1 = min(tx1, tiles_x - 1); ty1 = min(ty1, tiles_y - 1);
        near = (long)(tx1 - tx0 + 1) * (long)(ty1 - ty0 + 1) > BIN_MAX_RECT; // (an empty rectangle: off the frame, nothing to do)
      }
    }
  }
  if (near) {
    const uint32_t pos = atomicAdd(&cnt[tri ? 2 : 1], 1u);
    if (pos < BIN_NEAR_CAP) near_list[(tri ? BIN_NEAR_CAP : 0) + pos] = (uint32_t)(tri ? g - ns : g);
    else cnt[0] = 1u; // the frame falls back to the walk
  }
  const int rw = tx1 - tx0 + 1, rh = ty1 - ty0 + 1;
  const int ntile = (near || rw <= 0 || rh <= 0) ? 0 : rw * rh;
  if (ntile > 0 && ntile <= BIN_DIRECT) { // a few tiles: on the spot
    for (int ty = ty0; ty <= ty1; ++ty)
      for (int tx = tx0; tx <= tx1; ++tx) bin_try_tile(A, TW, tiles_x, g, wx, wy, wz, reff, tx, ty, cnt, list);
  }
  // (sharing a lane's rectangle out over its own wave's 64 lanes instead of queueing it was slower on 10k spheres, 0.40
  // against 0.36 ms: 157 waves is all that pass 1 has there, pass 2 spreads the same work over the chip)
  if (ntile > BIN_DIRECT) { // larger: cut into blocks of 8 x 8 tiles for pass 2
    const int bx0 = tx0 >> 3, bx1 = tx1 >> 3, by0 = ty0 >> 3, by1 = ty1 >> 3;
    const uint32_t nb = uint32_t((bx1 - bx0 + 1) * (by1 - by0 + 1));
    const uint32_t pos = atomicAdd(&cnt[3], nb);
    if (pos + nb > BIN_QUEUE_CAP) { cnt[0] = 1u; }
    else {
      uint32_t k = pos;
      for (int by = by0; by <= by1; ++by)
        for (int bx = bx0; bx <= bx1; ++bx) queue[k++] = make_uint2((uint32_t)g, (uint32_t)bx | ((uint32_t)by << 16));
    }
  }
}

// Pass 2, wave = (primitive, block of 8 x 8 tiles) from the queue, lane = tile.  A fixed grid of waves strides over the
// queue (its length is only known on the device).
__global__ __launch_bounds__(256) void bin_blocks_kernel(RenderArgs A, int TW, uint32_t *__restrict__ cnt, uint32_t *__restrict__ list,
                                                         const uint2 *__restrict__ queue)
{
  const int lane = threadIdx.x & 63;
  const uint32_t nq = min(cnt[3], (uint32_t)BIN_QUEUE_CAP), nwaves = gridDim.x * 4u;
  const int tiles_x = (A.W + TW - 1) / TW, tiles_y = (A.H + 7) >> 3;
  for (uint32_t q = blockIdx.x * 4u + (threadIdx.x >> 6); q < nq; q += nwaves) {
    const uint2 e = queue[q];
    const int g = int(e.x), tx = int(e.y & 0xffffu) * 8 + (lane & 7), ty = int(e.y >> 16) * 8 + (lane >> 3);
    float wx, wy, wz, reff;
    bin_ball(A, g, wx, wy, wz, reff);
    if (tx < tiles_x && ty < tiles_y) bin_try_tile(A, TW, tiles_x, g, wx, wy, wz, reff, tx, ty, cnt, list);
  }
}

// ---- start order of the blocks of tiles: heaviest first ---------------------------------------------------------
// With lists the work of a tile is known before the frame starts.  On a terrain the few tiles along the horizon are
// jobs of a millisecond (1 800 exact tests against a median of 35); in frame order they start two thirds into the frame
// and the GPU drains for 0.2 ms behind the last of them.  block_weight_kernel (wave = block of 2^map x 2^map tiles as
// the tile kernel numbers them, lane = tile) takes the largest list of each block, block_order_kernel (one workgroup)
// sorts the blocks by it, descending, with a counting sort; the tile kernel looks its block up in the result.
// Results do not depend on the order.
__global__ __launch_bounds__(256) void block_weight_kernel(RenderArgs A, int TW, int map, uint32_t *__restrict__ weight)
{
  const int lane = threadIdx.x & 63, bs = 1 << map;
  const int txn = (A.W + TW - 1) / TW, tyn = (A.local_rows + 7) >> 3;
  const int bxn = (txn + bs - 1) >> map, nb = bxn * ((tyn + bs - 1) >> map);
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= nb) return;
  uint32_t w = 0;
  for (int t = lane; t < bs * bs; t += 64) {
    const int tile_x = (b % bxn) * bs + (t & (bs - 1)), tile_y = (b / bxn) * bs + (t >> map);
    if (tile_x < txn && tile_y < tyn) {
      const int lrow0 = tile_y * 8, strip = lrow0 / A.strip_rows;
      const int grow0 = global_strip(strip, A.rank, A.nranks) * A.strip_rows + (lrow0 - strip * A.strip_rows);
      if (grow0 < A.H) {
        const int gtile = (grow0 >> 3) * txn + tile_x;
        w = max(w, A.bin_cnt[BIN_HEAD + 2 * gtile] + A.bin_cnt[BIN_HEAD + 2 * gtile + 1]);
      }
    }
  }
  w = (uint32_t)wave_max((float)min(w, 1023u));
  if (lane == 0) weight[b] = w;
}
__global__ __launch_bounds__(1024) void block_order_kernel(int nb, const uint32_t *__restrict__ weight, uint32_t *__restrict__ order)
{
  __shared__ uint32_t hist[1024], start[1024];
  for (int i = threadIdx.x; i < 1024; i += 1024) hist[i] = 0;
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += 1024) atomicAdd(&hist[1023u - min(weight[b], 1023u)], 1u); // key 0 = heaviest
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t acc = 0;
    for (int i = 0; i < 1024; ++i) { start[i] = acc; acc += hist[i]; }
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += 1024) order[atomicAdd(&start[1023u - min(weight[b], 1023u)], 1u)] = (uint32_t)b;
}

// entries per tile and primitive type: two batches
#ifndef MRT_BIN_CAP
#define MRT_BIN_CAP 128
#endif
// Bins apply to scenes with a hierarchy to walk, one pixel per lane, and a camera whose image is a rectangle (horizontal
// and vertical edges perpendicular: what rt::Viewport makes; anything else walks).  MRT_BINS=0 switches them off.
// -> 0: no bins; 1: lists of primitives; 2: lists of leaf clusters.
static int bins_apply(const RenderArgs &A, BinCamera *out)
{
  // MRT_BINS = 0: off; 1: lists of primitives whatever the scene; 2: lists of clusters whatever the scene (measurements)
  static const int env = getenv("MRT_BINS") ? atoi(getenv("MRT_BINS")) : -1;
  // Below a few thousand primitives the walk is a handful of steps per tile and not worth two more launches.  With
  // fewer primitives than tiles the lists hold primitives (10k spheres at 4K 0.437 -> 0.355 ms).  With more -- 1M
  // triangles at 4K: 12 M appends, and a tile spends its time on exact tests that lists do not change: 2.45 -> 2.69 ms
  // with lists of primitives -- they hold leaf clusters, a hundredth as many items: what is saved is the upper levels
  // of the walk, which a team repeats in every member.
  const long nprim = (long)A.sc.ns + A.sc.nt, ntile = (long)((A.W + 7) / 8) * ((A.H + 7) / 8);
  const int pinned = opt_lists(A) >= 2 ? opt_lists(A) - 1 : 0; // MRT_OPT_LISTS 2 | 3: whatever the scene (tests, stress runs)
  if (opt_lists(A) == 1 || (!pinned && (env == 0 || nprim < 2048))) return 0;
  const int mode = pinned ? pinned : env == 1 ? 1 : env == 2 ? 2 : (nprim > ntile ? 2 : 1);
  const double h[3] = {A.cam.hx, A.cam.hy, A.cam.hz}, v[3] = {A.cam.vx, A.cam.vy, A.cam.vz};
  const double b[3] = {(double)A.cam.ux - A.cam.ox, (double)A.cam.uy - A.cam.oy, (double)A.cam.uz - A.cam.oz};
  auto dot = [](const double *x, const double *y) { return x[0] * y[0] + x[1] * y[1] + x[2] * y[2]; };
  const double lh = std::sqrt(dot(h, h)), lv = std::sqrt(dot(v, v));
  if (!(lh > 1e-30) || !(lv > 1e-30) || !std::isfinite(lh) || !std::isfinite(lv)) return 0;
  const double hu[3] = {h[0] / lh, h[1] / lh, h[2] / lh}, vu[3] = {v[0] / lv, v[1] / lv, v[2] / lv};
  if (!(std::fabs(dot(hu, vu)) < 1e-4)) return 0;
  double n[3] = {hu[1] * vu[2] - hu[2] * vu[1], hu[2] * vu[0] - hu[0] * vu[2], hu[0] * vu[1] - hu[1] * vu[0]};
  double bn = dot(b, n);
  if (bn < 0) { bn = -bn; n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
  if (!(bn > 1e-6 * (std::fabs(dot(b, hu)) + std::fabs(dot(b, vu)) + lh + lv)) || !std::isfinite(bn)) return 0;
  if (out) {
    out->hx = (float)hu[0]; out->hy = (float)hu[1]; out->hz = (float)hu[2];
    out->vx = (float)vu[0]; out->vy = (float)vu[1]; out->vz = (float)vu[2];
    out->nx = (float)n[0]; out->ny = (float)n[1]; out->nz = (float)n[2];
    out->bh = (float)dot(b, hu); out->bv = (float)dot(b, vu); out->bn = (float)bn; out->lh = (float)lh; out->lv = (float)lv;
  }
  return mode;
}
static size_t bins_bytes(const RenderArgs &A)
{
  if (!bins_apply(A, nullptr)) return 0;
  const size_t tiles = size_t((A.W + 7) / 8) * size_t((A.H + 7) / 8);
  // (+ the blocks' weights and start order: at most one block per tile)
  return ((BIN_HEAD + 2 * tiles) * 4 + 255) / 256 * 256 + 2 * BIN_NEAR_CAP * 4 + (size_t)BIN_QUEUE_CAP * 8 + 2 * tiles * MRT_BIN_CAP * 4 + 2 * tiles * 4;
}

// Waves per tile for a scene (render_cull_kernel's NW and LSPLIT).  A team pays for itself when single tiles are long
// jobs: many lights (every hit tile walks the scene once per light) or a very large scene (tiles along a horizon
// open hundreds of clusters).  Measured (tools/timeline_cull.py): 64 point lights over 512 primitives
// 1.60 -> 1.17 ms dealing lights (2.9 ms dealing leaves), 1M triangles 5.4 -> 4.4 ms dealing leaves; 10k spheres under
// one light are faster without teams (0.41 against 0.94 ms).
// -> 1: one wave per tile; 4: teams dealing leaves; 5: teams dealing lights.  MRT_OPT_CULL_TEAM (or environment
// MRT_CULL_TEAM, read once) = 1 | 4 | 5 pins it; 0 = by the scene: lights from 8 lights on; leaves for scenes of many
// TRIANGLES PER TILE -- from 64 000 triangles and 2.5 per tile of the frame on.  (Until the end of round 3: from 100 000
// primitives of either kind.  Measured then, tools/exp_team_threshold.py, one wave against teams in ms: 100k spheres
// 0.33 / 0.63 at 1080p and 0.76 / 1.98 at 4K, 300k spheres 0.40 / 0.92 at 1080p -- spheres never want teams; terrains at
// 4K: 122k triangles 0.74 / 1.37, 250k 0.89 / 1.39, 360k 1.00 / 0.98, 490k 1.16 / 1.12, 1M 2.92 / 1.41; at 1080p: 40k 0.23 /
// 0.21, 90k 0.32 / 0.23, 122k 0.41 / 0.40, 250k 0.53 / 0.32, 360k 0.68 / 0.35, 490k 0.87 / 0.42.  A team is for frames
// whose time is a few long tiles; 130 000 tiles of a 4K frame hide them behind each other, 32 000 do not.)
static int team_mode(const RenderArgs &A)
{
  static const int env = [] {
    const char *e = getenv("MRT_CULL_TEAM");
    const int x = e ? atoi(e) : 0;
    return (x == 1 || x == 4 || x == 5) ? x : 0;
  }();
  const int forced = opt_team(A) ? opt_team(A) : env;
  if (forced == 5 && A.sc.ncs + A.sc.nct > 64) return 4; // see below
  if (forced) return forced;
  // (teams dealing lights set the lights' shadow walks up lane-parallel over (light, cluster) pairs: at most 64 clusters)
  if (A.sc.nl >= 8 && A.sc.ncs + A.sc.nct <= 64) return 5;
  const long tiles = (long)((A.W + 7) / 8) * ((A.H + 7) / 8); // of the whole frame, whatever this rank renders of it
  const bool dense = A.sc.nt >= 64000 && 2L * A.sc.nt >= 5L * tiles;
  return (dense || A.sc.nl >= 8) ? 4 : 1;
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
  if constexpr (HIER && P == 1) {
    BinCamera C;
    const int bmode = A.bin_cnt ? bins_apply(A, &C) : 0;
    if (bmode) {
      A.bin_clusters = bmode == 2;
      const size_t tiles = size_t((A.W + 7) / 8) * size_t((A.H + 7) / 8);
      hipError_t e = hipMemsetAsync((void *)A.bin_cnt, 0, (BIN_HEAD + 2 * tiles) * 4, stream);
      if (e != hipSuccess) return e;
      const long nprim = A.bin_clusters ? (long)A.sc.ncs + A.bin_nct_regular
                                        : (long)A.sc.ns + std::min<long>(A.sc.nt, (long)A.bin_nct_regular * 64);
      uint2 *queue = (uint2 *)(A.bin_near + 2 * BIN_NEAR_CAP);
      if (nprim > 0) {
        hipLaunchKernelGGL(bin_primitives_kernel, dim3((unsigned)((nprim + 255) / 256)), dim3(256), 0, stream, A, C, 8,
                           (uint32_t *)A.bin_cnt, (uint32_t *)A.bin_list, (uint32_t *)A.bin_near, queue);
        hipLaunchKernelGGL(bin_blocks_kernel, dim3(2048), dim3(256), 0, stream, A, 8, (uint32_t *)A.bin_cnt, (uint32_t *)A.bin_list,
                           (const uint2 *)queue);
      }
      if ((e = hipGetLastError()) != hipSuccess) return e;
    } else {
      A.bin_cnt = nullptr;
    }
  } else {
    A.bin_cnt = nullptr;
  }
  // XCD-aware tile order for scenes with a hierarchy (their leaves are what neighbouring tiles share); MRT_CULL_XCD=0: rows
  // (MRT_CULL_XCD = n: blocks of 2^n x 2^n tiles, whatever the scene; 0: rows.)  By default for scenes that do not fit an
  // L2 (from 100 000 primitives on): the 1M-triangle terrain 2.61 -> 2.41 ms; scenes that do fit lose 1-3 % to the
  // coarser interleaving of cheap and expensive tiles (10k spheres 421 -> 432 us, 64 lights 1141 -> 1155 us).
  static const int xcd = getenv("MRT_CULL_XCD") ? atoi(getenv("MRT_CULL_XCD")) : -1;
  A.tile_map = !HIER ? 0 : xcd >= 0 ? (xcd > 6 ? 6 : xcd) : ((long)A.sc.ns + A.sc.nt >= 100000 ? 3 : 0);
  A.block_order = nullptr;
  if constexpr (HIER && P == 1) {
    static const bool order_on = !(getenv("MRT_CULL_ORDER") && atoi(getenv("MRT_CULL_ORDER")) == 0);
    if (A.bin_cnt && A.tile_map && order_on) { // the blocks start heaviest first (block_weight_kernel)
      const size_t tiles = size_t((A.W + 7) / 8) * size_t((A.H + 7) / 8);
      const int bs = 1 << A.tile_map;
      const int nb = ((tiles_x + bs - 1) / bs) * ((tiles_y + bs - 1) / bs);
      uint32_t *weight = (uint32_t *)A.bin_list + 2 * tiles * (size_t)A.bin_cap, *order = weight + tiles;
      hipLaunchKernelGGL(block_weight_kernel, dim3((unsigned)((nb + 3) / 4)), dim3(256), 0, stream, A, 8, A.tile_map, weight);
      hipLaunchKernelGGL(block_order_kernel, dim3(1), dim3(1024), 0, stream, nb, (const uint32_t *)weight, order);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
      A.block_order = order;
    }
  }
  // waves per workgroup: waves share nothing; 2 measured best on C3 and C5 (413 / 1561 us against 444 / 1625 with 4)
  static const int wpw = getenv("MRT_CULL_WPW") ? atoi(getenv("MRT_CULL_WPW")) : 2;
  if constexpr (HIER && P == 1) {
    const int team = team_mode(A);
    if (team != 1) { // one workgroup of four waves per tile
      dim3 tgrid(tiles_x, tiles_y);
      if (A.tile_map) {
        const unsigned bs = 1u << A.tile_map;
        tgrid = dim3(8u * bs * bs * (unsigned)((((tiles_x + bs - 1) / bs) * ((tiles_y + bs - 1) / bs) + 7) / 8));
      }
      if (team == 5) {                  // the members deal the lights
        if (A.spp > 1) hipLaunchKernelGGL((render_cull_kernel<true, 1, true, 4, true>), tgrid, dim3(256), 0, stream, A);
        else hipLaunchKernelGGL((render_cull_kernel<true, 1, false, 4, true>), tgrid, dim3(256), 0, stream, A);
      } else {                          // ... or the leaves of every walk
        if (A.spp > 1) hipLaunchKernelGGL((render_cull_kernel<true, 1, true, 4, false>), tgrid, dim3(256), 0, stream, A);
        else hipLaunchKernelGGL((render_cull_kernel<true, 1, false, 4, false>), tgrid, dim3(256), 0, stream, A);
      }
      return hipGetLastError();
    }
  }
  dim3 grid((tiles_x + wpw - 1) / wpw, tiles_y);
  if (A.tile_map) {
    const unsigned bs = 1u << A.tile_map;
    if (wpw == 1 || wpw == 2 || wpw == 4 || wpw == 8)
      grid = dim3(8u * (bs * bs / (unsigned)wpw) * (unsigned)((((tiles_x + bs - 1) / bs) * ((tiles_y + bs - 1) / bs) + 7) / 8));
    else A.tile_map = 0;
  }
  if (A.spp > 1) hipLaunchKernelGGL((render_cull_kernel<HIER, P, true, 1>), grid, dim3(64 * wpw), 0, stream, A);
  else hipLaunchKernelGGL((render_cull_kernel<HIER, P, false, 1>), grid, dim3(64 * wpw), 0, stream, A);
  return hipGetLastError();
}

// the triangles' set-up (tri_setup_kernel): for the kernel that keeps lists of (pixel, triangle) pairs -- a hierarchy, one
// sample per pixel, teams dealing leaves.  MRT_PAIRS=0 in the environment: off (every survivor of the
// refinement is tested for all pixels of its tile, as before round 3's last change)
static bool setup_applies(const RenderArgs &A)
{
  static const bool on = !(getenv("MRT_PAIRS") && atoi(getenv("MRT_PAIRS")) == 0);
  const bool hier = A.sc.ncs > 1 || A.sc.nct > 1;
  return on && hier && A.sc.nt > 0 && A.spp == 1 && team_mode(A) == 4;
}
static size_t setup_bytes(const RenderArgs &A) { return setup_applies(A) ? size_t(A.sc.nt) * SETUP_FLOATS * 4 : 0; }
static bool make_setup_cam(const RenderArgs &A, SetupCam &C)
{
  const DevCamera &c = A.cam;
  const double ul[3] = {c.ux, c.uy, c.uz}, h[3] = {c.hx, c.hy, c.hz}, v[3] = {c.vx, c.vy, c.vz}, o[3] = {c.ox, c.oy, c.oz};
  double g[3], dlmax = 0.0;
  for (int k = 0; k < 3; ++k) { C.ul[k] = ul[k]; C.h[k] = h[k]; C.v[k] = v[k]; C.o[k] = o[k]; g[k] = ul[k] - o[k]; }
  // inverse of the matrix with columns (h, v, g): rows = cross products over the determinant
  const double cx[3][3] = {{v[1] * g[2] - v[2] * g[1], v[2] * g[0] - v[0] * g[2], v[0] * g[1] - v[1] * g[0]},
                           {g[1] * h[2] - g[2] * h[1], g[2] * h[0] - g[0] * h[2], g[0] * h[1] - g[1] * h[0]},
                           {h[1] * v[2] - h[2] * v[1], h[2] * v[0] - h[0] * v[2], h[0] * v[1] - h[1] * v[0]}};
  const double det = h[0] * cx[0][0] + h[1] * cx[0][1] + h[2] * cx[0][2];
  const bool ok = std::isfinite(det) && std::fabs(det) > 1e-300;
  for (int r = 0; r < 3; ++r)
    for (int k = 0; k < 3; ++k) C.inv[r][k] = ok ? cx[r][k] / det : 0.0; // (a degenerate camera: every reference pixel is the frame's middle)
  C.W = A.W; C.H = A.H; C.rW = 1.0 / A.W; C.rH = 1.0 / A.H;
  for (int k = 0; k < 3; ++k)
    C.r[k] = float(1.5 * 5.9604645e-8 * (3.0 * std::fabs(ul[k]) + 5.0 * std::fabs(h[k]) + 4.0 * std::fabs(v[k]) + std::fabs(o[k])) * 1.0001);
  for (int a = 0; a < 2; ++a)
    for (int b = 0; b < 2; ++b) {
      double d2 = 0.0;
      for (int k = 0; k < 3; ++k) { const double d = g[k] + a * h[k] + b * v[k]; d2 += d * d; }
      dlmax = std::max(dlmax, std::sqrt(d2));
    }
  C.dl = float(dlmax * 1.001) + 2.0f * camera_rounding(A.cam); // (|ray| is convex over the frame: a corner has the longest)
  return std::isfinite(C.dl);
}

size_t loose_scratch_bytes(const RenderArgs &A) { return (loose_bytes(A) + bins_bytes(A) + 255) / 256 * 256 + setup_bytes(A); }

hipError_t launch_render_cull(const RenderArgs &in, hipStream_t stream) { return launch_render_cull(in, nullptr, stream); }

hipError_t launch_render_cull(const RenderArgs &in, void *loose_scratch, hipStream_t stream)
{
  RenderArgs A = in;
  A.loose_list = nullptr; A.loose_cnt = nullptr;
  A.bin_cnt = nullptr; A.bin_list = nullptr; A.bin_near = nullptr;
  A.utab = nullptr; // (the cull kernels read the triangles' set-up from here, or null: none)
  if (loose_scratch && setup_bytes(A)) {
    SetupCam C;
    if (make_setup_cam(A, C)) {
      float4 *su = (float4 *)((char *)loose_scratch + (loose_bytes(A) + bins_bytes(A) + 255) / 256 * 256);
      hipLaunchKernelGGL(tri_setup_kernel, dim3((unsigned)((A.sc.nt + 255) / 256)), dim3(256), 0, stream, A.sc, C, su);
      hipError_t e = hipGetLastError();
      if (e != hipSuccess) return e;
      A.utab = (const float *)su;
    }
  }
  if (loose_scratch && bins_bytes(A)) { // the tiles' bins sit behind the loose triangles' lists; filled by launch_cull<true, 1>
    const size_t tiles = size_t((A.W + 7) / 8) * size_t((A.H + 7) / 8);
    char *b = (char *)loose_scratch + loose_bytes(A);
    A.bin_cnt = (const uint32_t *)b;
    A.bin_near = (const uint32_t *)(b + ((BIN_HEAD + 2 * tiles) * 4 + 255) / 256 * 256);
    A.bin_list = A.bin_near + 2 * BIN_NEAR_CAP + 2 * (size_t)BIN_QUEUE_CAP; // (the queue of the two binning passes sits in between)
    A.bin_cap = opt_list_cap(A) > 0 && opt_list_cap(A) < MRT_BIN_CAP ? opt_list_cap(A) : MRT_BIN_CAP;
    static const int solo_max = getenv("MRT_CULL_SOLO") ? atoi(getenv("MRT_CULL_SOLO")) : 48;
    A.solo_max = solo_max;
    A.bin_nct_regular = A.sc.nct_regular;
  }
  if (loose_scratch && loose_lists_apply(A)) {
    const int bx = (A.W + LOOSE_BLOCK - 1) / LOOSE_BLOCK, by = (A.H + LOOSE_BLOCK - 1) / LOOSE_BLOCK;
    const size_t blocks = size_t(bx) * by;
    uint32_t *cnt = (uint32_t *)loose_scratch;
    uint16_t *list = (uint16_t *)((char *)loose_scratch + ((blocks * 4 + 255) / 256) * 256);
    // (MRT_LOOSE_CAP in the environment shortens the lists: stress runs use it to reach the overflow path)
    static const int cap_env = getenv("MRT_LOOSE_CAP") ? atoi(getenv("MRT_LOOSE_CAP")) : MRT_LOOSE_CAP;
    const int cap = opt_list_cap(A) > 0 ? opt_list_cap(A) : cap_env;
    A.loose_bx = bx; A.loose_cap = cap < 1 ? 1 : (cap > MRT_LOOSE_CAP ? MRT_LOOSE_CAP : cap);
    A.loose_nct_walk = A.sc.nct_regular; A.loose_nsct_walk = (A.sc.nct_regular + 63) / 64;
    // the padded block's largest offset from its centre ray (as launch_cull does for a tile), in double, rounded up
    const double hw = (0.5 * LOOSE_BLOCK + 0.5) / A.W, hh = (0.5 * LOOSE_BLOCK + 0.5) / A.H;
    const double hx[3] = {hw * A.cam.hx, hw * A.cam.hy, hw * A.cam.hz}, hy[3] = {hh * A.cam.vx, hh * A.cam.vy, hh * A.cam.vz};
    double pp = 0, mm = 0;
    for (int k = 0; k < 3; ++k) { pp += (hx[k] + hy[k]) * (hx[k] + hy[k]); mm += (hx[k] - hy[k]) * (hx[k] - hy[k]); }
    const float block_ext = float(std::sqrt(pp > mm ? pp : mm) * 1.001) + 2.0f * camera_rounding(A.cam);
    hipError_t e = hipMemsetAsync(cnt, 0, blocks * 4, stream);
    if (e != hipSuccess) return e;
    const long nloose = (long)A.sc.nt - (long)A.sc.nct_regular * 64;
    hipLaunchKernelGGL(bin_loose_kernel, dim3((unsigned)blocks, (unsigned)((nloose + 255) / 256)), dim3(256), 0, stream, A, block_ext, list, cnt);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    A.loose_list = list; A.loose_cnt = cnt;
  }
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
