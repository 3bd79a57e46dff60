// mrt_bounce.hip -- bounce chains: config::MAX_RECURSION_DEPTH > 0 (src/main.cu:26, :79-98).
// MetallicMaterial::bounce mirrors the ray (core/material.cuh:126-132, :151-158); DiffuseMaterial::bounce sends
// it along normal + a random vector of (0,1]^3 drawn from the pixel's XORWOW stream (core/material.cuh:52-61,
// cuda_utils/curand_state.cuh:85-109; streams: mrt_rng.hip).  The stream belongs to the pixel, not the sample:
// the 121 samples of a pixel draw from it one after the other (src/main.cu:135, :149-153), so the sample loop
// below must stay sequential per lane.
//
// The reference recurses: color_ray(depth) = C (.) color_ray(depth + 1) [+ C (.) light].  Here the chain is
// walked iteratively (the reference's TODO.md asks for exactly that): every level pushes its object colour and
// its direct light on a small per-lane stack, the chain ends with the sky colour, or with the constant 0.01
// that the reference returns beyond the depth limit (src/main.cu:51-53), and the stack is folded from the last
// level back with the reference's operations in the reference's order -- bit-comparable with the oracle.
//
// One wave64 per 8x8 tile, same strip partition as the other render kernels.  Primary rays of a tile are culled
// per tile as in the depth-0 kernels; bounce rays of neighbouring pixels diverge, so further down a chain a ray
// is tested against every primitive (scenes of at most 64 primitives: wave-uniform loop index, scalar loads) or
// walks the primitive hierarchy on its own (larger scenes).

#include "mrt_internal.h"
#include "mrt_device.h"
#include "mrt_cull_common.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace mrt {

constexpr int MAX_LEVELS = 9; // depths 0..8

// ---- every lane runs its own (sample, level) state machine ----------------------------------------------------
// Walking a sample's chain level by level for the whole wave (the first version of this kernel) makes a wave
// finish a sample only when its longest chain has ended: with diffuse bounces some lane nearly always has a long
// one, and the wave pays max-over-lanes levels for every one of 121 samples.  But
// every step of every chain is the same work -- closest hit over all primitives, shadow rays over all primitives,
// one bounce -- whatever sample and level a lane is at.  So here a lane that ends a chain folds it, adds it to its
// pixel and starts its next sample at once; the wave runs until the lane with the most steps IN TOTAL is done
// (sum over samples instead of a maximum per sample).  Per lane the order of samples, levels and random draws is
// the reference's, so results do not change.  The level stack is indexed per lane now: it lives in registers and
// is written and folded through unrolled selects (9 levels x 6 floats).
// Tiles whose padded cone of primary rays (all 121 sample positions included) misses every primitive bound take a
// second shortcut: every chain is the sky colour of its primary ray; nothing is tested and nothing is drawn.
// ---- per-ray walk of the primitive hierarchy (scenes of more than 64 primitives) ---------------------------
// Bounce rays of neighbouring pixels go everywhere, so there is no tile to cull against; what is left is the ray
// itself.  Every lane walks the cluster hierarchy of mrt_scene_build.h for ITS ray: super-cluster ball, cluster ball
// (per-lane loads), then the 64 primitives of a cluster it has to open.  The wave moves in lock step -- all
// lanes look for their next cluster, then all lanes test 64 primitives of their own cluster -- so a lane costs
// what the lane with the most clusters costs, not what the union of all lanes' clusters costs.  The closest-hit
// rule is an order-independent minimum over (t, object), and occlusion an order-independent "any", so visiting
// fewer primitives in another order changes nothing; the ball tests carry the margins of mrt_cull_common.h.

// May the ray from the origin along the unit axis a (within tdist of the origin) touch the ball at w = centre -
// origin with effective radius reff?  NaN and infinite bounds pass.  The "behind" and "beyond" verdicts rest on the
// hit point the exact test computes, p = fl(o + t d) with the COMPUTED t in [tmin, tmax]: it lies ahead of the
// origin and within tmax |d| of it up to its own rounding, 1.2e-7 of |o| + |p - o| -- hence oslack >= 1e-6 |o|_1.
__device__ __forceinline__ bool ray_may_touch_ball(float wx, float wy, float wz, float reff, float ax, float ay,
                                                   float az, float tdist, float oslack)
{
  const float proj = dot3(wx, wy, wz, ax, ay, az);
  const float qx = wx - proj * ax, qy = wy - proj * ay, qz = wz - proj * az;
  const float perp = fast_sqrt(dot3(qx, qy, qz, qx, qy, qz));
  const float reach = reff + CULL_LEN_SLACK * norm1(wx, wy, wz);
  if (perp > reach) return false;                   // the line misses the ball
  const float along = reach + oslack + 1e-6f * reff;
  if (proj < -along) return false;                  // the ball lies behind the origin: t < 0
  if (proj - along > tdist) return false;           // ... or beyond the end of the segment
  return true;
}

// leaf(first primitive of the lane's cluster, lane takes part) -> lane wants to go on
template <bool SPHERES, class Leaf>
__device__ __forceinline__ void walk_per_lane(const float *__restrict__ sclu, int nsc, const float *__restrict__ clu,
                                              int nc, bool want, float ox, float oy, float oz, float ax, float ay,
                                              float az, float tdist, Leaf leaf)
{
  const float oslack = 1e-6f * norm1(ox, oy, oz);
  auto touch = [&](const float *__restrict__ b, int n, int i) {
    const float wx = b[i] - ox, wy = b[n + i] - oy, wz = b[2 * n + i] - oz, R = b[3 * n + i];
    float reff;
    if (SPHERES) { // member spheres: sqrt(r^2 + s L^2) <= r + sqrt(s) L, L <= |w| + R (cluster_cull_radius)
      const float L = fast_sqrt(dot3(wx, wy, wz, wx, wy, wz)) * CULL_REL + R;
      reff = R * CULL_REL + CULL_DISC_SLACK_SQRT * CULL_REL * L + CULL_ABS;
    } else {
      reff = R * CULL_REL + CULL_ABS;
    }
    return ray_may_touch_ball(wx, wy, wz, reff, ax, ay, az, tdist, oslack);
  };
  int c = -1;
  bool on = want && nc > 0;
  while (true) {
    bool found = false;
    while (on && !found) { // every lane looks for its next cluster
      ++c;
      if (c >= nc) { on = false; break; }
      if (nsc > 1 && (c & 63) == 0 && !touch(sclu, nsc, c >> 6)) { c += 63; continue; }
      found = nc == 1 || touch(clu, nc, c);
    }
    if (!__any(on)) break;
    on = leaf(c * 64, on) && on;
  }
}

__device__ __forceinline__ const RenderArgs &args_of(const RenderArgs &a) { return a; }
__device__ __forceinline__ const RenderArgs &args_of(const SmallArgs &s) { return s.A; }
__device__ __forceinline__ float prm_of(const RenderArgs &, int, int) { return 0.0f; }
__device__ __forceinline__ float prm_of(const SmallArgs &s, int f, int lane) { return s.prm[f][lane]; }

#ifdef MRT_TIMELINE
// Diagnostic build only (make variant DEFS=-DMRT_TIMELINE; tools/timeline_cull.py REF): per tile the 100 MHz real-time
// clock at entry and exit of its wave, the shader clock it lived, chain steps | loop iterations.
enum { TLB_WORDS = 16, TLB_MAX_WAVES = 1 << 17 }; // 4..7: cycles in sky + select | shading | bounce | fold + loop head; 8..13: finer stamps
__device__ unsigned long long g_timeline_bounce[TLB_WORDS * TLB_MAX_WAVES];
extern "C" int mrt_diag_timeline_bounce(unsigned long long *host_out, size_t words)
{
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_timeline_bounce), words * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

template <bool MULTI, bool SMALL, int WPB, class Args>
__device__ __forceinline__ void render_bounce_body(const Args &S)
{
  const RenderArgs &A = args_of(S);
  const DevScene &sc = A.sc;
  const DevCamera &cam = A.cam;
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int tx = lane & 7, ty = lane >> 3;
  const int ns = sc.ns, nt = sc.nt;

  const int tiles_x = (A.W + 7) >> 3, tiles_y = (A.local_rows + 7) >> 3;
  const int widx = blockIdx.x * WPB + wave; // WPB waves per workgroup, each with a tile of its own
  if (widx >= tiles_x * tiles_y) return; // waves are independent: no barrier below
  const int tile = A.tile_order ? int(A.tile_order[widx]) : widx; // heaviest tiles first (launcher), or row by row
  // (The frame ends when its slowest wave does, and with many samples per pixel that is ONE wave's serial work: 121
  // samples x 9 levels of a pixel between two mirrors are ~1050 loop iterations of ~12 000 cycles each, whether the wave
  // shares its SIMD or not -- tools/timeline_cull.py REF: every wave has started after 2.5 ms, the last ends at 5.5.
  // Raising the priority of the heaviest tiles' waves changes nothing, nor does a build without register spills.)
  const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
#ifdef MRT_TIMELINE
  const unsigned long long tl_rt0 = __builtin_amdgcn_s_memrealtime(), tl_c0 = __builtin_amdgcn_s_memtime();
  unsigned long long tl_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#endif
  const int col = tile_x * 8 + tx;
  const int lrow = tile_y * 8 + ty;
  const int lrow0 = tile_y * 8;
  const int strip = lrow0 / A.strip_rows;
  const int grow0 = global_strip(strip, A.rank, A.nranks) * A.strip_rows + (lrow0 - strip * A.strip_rows);
  const int grow = grow0 + ty;
  const bool valid = (lrow < A.local_rows) && (grow < A.H) && (col < A.W);

  const int nsamp = MULTI ? A.spp : 1;
  const float DIVW = 13.0f, DIVH = 13.0f;
  const float u0 = float(col) / float(A.W); // src/main.cu:136
  const float v0 = float(grow) / float(A.H); // src/main.cu:137
  float ox = cam.ox, oy = cam.oy, oz = cam.oz, dx = 0.0f, dy = 0.0f, dz = 0.0f; // current ray of the lane's chain
  // sample offsets of the 11 x 11 mesh (pipeline/multisample_mesh.cuh:23-30): four IEEE divisions per sample if
  // computed on the spot; instead every wave tabulates its 22 values once (LDS, one slice per wave) and lanes
  // look theirs up -- lanes sit at different samples, and a table read does not care which lanes are active
  __shared__ float offtab[WPB][32];
  if (MULTI) {
    if (lane < 11) offtab[wave][lane] = (-1.0f + 2.0f * float(lane + 1) / DIVW) / float(A.W);
    else if (lane >= 16 && lane < 27) offtab[wave][lane] = (-1.0f + 2.0f * float(lane - 16 + 1) / DIVH) / float(A.H);
    __builtin_amdgcn_wave_barrier(); // the table is written and read by this wave only
  }
  auto primary_ray = [&](int sidx) { // pipeline/viewport.cuh:30-34
    float su = u0, sv = v0;
    if (MULTI) {
      const int r = sidx / 11, c = sidx - r * 11;
      su = u0 + offtab[wave][c]; sv = v0 + offtab[wave][16 + r];
    }
    ox = cam.ox; oy = cam.oy; oz = cam.oz;
    dx = ((cam.ux + su * cam.hx) + sv * cam.vx) - cam.ox;
    dy = ((cam.uy + su * cam.hy) + sv * cam.vy) - cam.oy;
    dz = ((cam.uz + su * cam.hz) + sv * cam.vz) - cam.oz;
  };
  auto sky = [&](float &r, float &g, float &b) { // background, src/main.cu:101-104
    float nx = dx, ny = dy, nz = dz;
    normalize3(nx, ny, nz);
    const float y = 0.5f * (ny + 1.0f);
    const float w0 = 1.0f - y;
    r = w0 * 1.0f + y * 0.5f; g = w0 * 1.0f + y * 0.7f; b = w0 * 1.0f + y * 1.0f;
  };

  float accr = 0.0f, accg = 0.0f, accb = 0.0f;
  int out_id = -1;
  float out_t = MRT_INF;
  unsigned int n_shadow = 0, n_hit = 0, n_occl = 0;
  unsigned int n_steps = 0, n_iter = 0; // wave-uniform diagnostics: lane steps and wave iterations of the chain loop

  // ---- can any primary ray of the tile hit anything?  Cone around the tile's rays, padded by one pixel on
  //      every side (covers the sub-pixel sample offsets), against every primitive bound; as in mrt_cull.hip ----
  bool tile_has_candidates;
  unsigned long long primary_mask = ~0ull; // SMALL: primitives (bit = lane) a primary ray of the tile may hit
  float cax, cay, caz, cosT, sinT;         // the tile's cone of primary rays: unit axis, half-angle
  {
    const float rw = fast_rcp(float(A.W)), rh = fast_rcp(float(A.H));
    const float um = (float(tile_x * 8) + 3.5f) * rw, vm = (float(grow0) + 3.5f) * rh;
    cax = ((cam.ux + um * cam.hx) + vm * cam.vx) - cam.ox;
    cay = ((cam.uy + um * cam.hy) + vm * cam.vy) - cam.oy;
    caz = ((cam.uz + um * cam.hz) + vm * cam.vz) - cam.oz;
    const float inv = fast_rsq(dot3(cax, cay, caz, cax, cay, caz));
    cax *= inv; cay *= inv; caz *= inv;
    sinT = A.cone_ext * inv * (1.0f + 1e-4f) + 1e-6f;
    cosT = fast_sqrt(fmaxf(1.0f - sinT * sinT, 0.0f)) * (1.0f - 1e-5f) - 1e-6f;
    if (!(sinT < 0.99f) || !(cosT > 0.0f)) { cosT = -1.0f; sinT = 0.0f; } // cannot cull: every ball passes
    bool pass = false;
    if (SMALL) { // lane = primitive (spheres first): cone test / screen-space edge functions, as in mrt_small.hip
      if (lane < ns) {
        pass = ball_may_touch_cone(prm_of(S, 9, lane), prm_of(S, 10, lane), prm_of(S, 11, lane), prm_of(S, 12, lane), cax,
                                   cay, caz, cosT, sinT);
      } else if (lane < ns + nt) {
        bool cull = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float f = __builtin_fmaf(um, prm_of(S, 3 * i + 1, lane), __builtin_fmaf(vm, prm_of(S, 3 * i + 2, lane), prm_of(S, 3 * i, lane)));
          cull |= (f < 0.0f); // NaN compares false: never culls
        }
        pass = !cull;
      }
      primary_mask = __ballot(pass);
    } else {
      for (int base = 0; base < ns && !__any(pass); base += 64) {
        const int i = base + lane;
        if (i < ns) {
          const float r = sc.sph_r[i];
          const float wx = sc.sph_cx[i] - cam.ox, wy = sc.sph_cy[i] - cam.oy, wz = sc.sph_cz[i] - cam.oz;
          pass = ball_may_touch_cone(wx, wy, wz, sphere_cull_radius(r * r, dot3(wx, wy, wz, wx, wy, wz)), cax, cay, caz,
                                     cosT, sinT);
        }
      }
      for (int base = 0; base < nt && !__any(pass); base += 64) {
        const int i = base + lane;
        if (i < nt)
          pass = ball_may_touch_cone(sc.tri_bx[i] - cam.ox, sc.tri_by[i] - cam.oy, sc.tri_bz[i] - cam.oz,
                                     sc.tri_br[i] * CULL_REL + CULL_ABS, cax, cay, caz, cosT, sinT);
      }
    }
    tile_has_candidates = __any(pass);
  }

  // SMALL: lane j keeps the bound of primitive j (spheres first) for the shadow-ray culling below
  float pbx = 0.0f, pby = 0.0f, pbz = 0.0f, pbr = 0.0f; // sphere: centre, radius; triangle: bounding ball
  const bool own_s = SMALL && lane < ns, own_t = SMALL && lane >= ns && lane < ns + nt;
  if (tile_has_candidates) {
    if (own_s) { pbx = sc.sph_cx[lane]; pby = sc.sph_cy[lane]; pbz = sc.sph_cz[lane]; pbr = sc.sph_r[lane]; }
    else if (own_t) { const int k = lane - ns; pbx = sc.tri_bx[k]; pby = sc.tri_by[k]; pbz = sc.tri_bz[k]; pbr = sc.tri_br[k]; }
  }
  // SMALL: every wave keeps its own copy of the scene in LDS (at most 64 records of 20 words: spheres cx, cy, cz,
  // r, object; triangles the 16 floats + object), written once by the owning lanes and read back with one address
  // for the whole wave (broadcast).  The loops below then wait ~100 cycles of LDS latency per primitive instead of
  // a scalar-cache round trip, and nothing competes for scalar registers.  Larger scenes: scalar loads.
  __shared__ float scene_lds[SMALL ? WPB : 1][SMALL ? 64 * 20 : 1];
  if (SMALL && tile_has_candidates) {
    float *rec = &scene_lds[wave][lane * 20];
    if (own_s) {
      rec[0] = pbx; rec[1] = pby; rec[2] = pbz; rec[3] = pbr;
      rec[16] = __int_as_float(sc.sph_obj[lane]);
    } else if (own_t) {
      const float4 *q4 = reinterpret_cast<const float4 *>(sc.tri_rec + 16 * size_t(lane - ns));
#pragma unroll
      for (int k = 0; k < 4; ++k) reinterpret_cast<float4 *>(rec)[k] = q4[k];
      rec[16] = __int_as_float(sc.tri_obj[lane - ns]);
    }
    __builtin_amdgcn_wave_barrier(); // written and read by this wave only
  }
  auto sphere_of = [&](int i, float &cx, float &cy, float &cz, float &r, int &obj) {
    if (SMALL) {
      const float *rec = &scene_lds[wave][i * 20];
      const float4 a = *reinterpret_cast<const float4 *>(rec);
      cx = a.x; cy = a.y; cz = a.z; r = a.w;
      obj = __float_as_int(rec[16]);
    } else {
      cx = sc.sph_cx[i]; cy = sc.sph_cy[i]; cz = sc.sph_cz[i]; r = sc.sph_r[i];
      obj = sc.sph_obj[i];
    }
  };
  auto triangle_of = [&](int i, float (&q)[16], int &obj) {
    if (SMALL) {
      const float *rec = &scene_lds[wave][(ns + i) * 20];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float4 a = reinterpret_cast<const float4 *>(rec)[k];
        q[4 * k] = a.x; q[4 * k + 1] = a.y; q[4 * k + 2] = a.z; q[4 * k + 3] = a.w;
      }
      obj = __float_as_int(rec[16]);
    } else {
      const float *m = sc.tri_rec + 16 * size_t(i);
#pragma unroll
      for (int k = 0; k < 16; ++k) q[k] = m[k];
      obj = sc.tri_obj[i];
    }
  };

  if (!tile_has_candidates) {
    for (int sidx = 0; sidx < nsamp; ++sidx) {
      primary_ray(sidx);
      float r, g, b;
      sky(r, g, b);
      accr += r; accg += g; accb += b; // src/main.cu:151
    }
  } else {
    // src/main.cu:133-135: a private copy of the pixel's state; never written back
    const bool use_rng = A.rng != nullptr;
    uint32_t g0 = 0, g1 = 0, g2 = 0, g3 = 0, g4 = 0, gd = A.rng_d;
    if (use_rng && valid) {
      const size_t pi = size_t(grow) * A.W + col, rs = A.rng_stride;
      g0 = A.rng[pi]; g1 = A.rng[rs + pi]; g2 = A.rng[2 * rs + pi]; g3 = A.rng[3 * rs + pi]; g4 = A.rng[4 * rs + pi];
    }
    auto uniform = [&]() -> float { // curand_uniform: xorwow draw, then x * 2^-32 + 2^-33
      const uint32_t tt = g0 ^ (g0 >> 2);
      g0 = g1; g1 = g2; g2 = g3; g3 = g4;
      g4 = (g4 ^ (g4 << 4)) ^ (tt ^ (tt << 1));
      gd += 362437u;
      return float(g4 + gd) * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
    };

    float stC[MAX_LEVELS][3], stL[MAX_LEVELS][3]; // per level: object colour, direct light (registers: static indices only)
    unsigned int stH = 0;                         // bit k: a light contributed at level k
    int sidx = 0, depth = 0;
    bool work = valid; // lanes outside the frame have nothing to do
    if (work) primary_ray(0);

    // Two kinds of iteration.  Primary: every lane that is at level 0 takes its step; with SMALL only the tile's
    // candidate primitives are tested (the primary rays of all samples lie in the tile's padded cone).  Lanes
    // further down a chain wait.  Deep: when no lane is at level 0, or A.s_cap lanes are waiting (8: measured
    // best on the reference's frame, 7.8 ms against 9.2 with "only when no lane is at level 0" and 8.2 with 16),
    // the waiting lanes take one step against every primitive.  Waiting fills the expensive iterations: most
    // chains end at level 0 or 1.
    const int deep_threshold = A.s_cap > 0 ? A.s_cap : 64;
#ifdef MRT_TIMELINE
    unsigned long long tl_last = __builtin_amdgcn_s_memtime();
#define TLB_MARK(k) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); tl_acc[k] += now_ - tl_last; tl_last = now_; } while (0)
#else
#define TLB_MARK(k) do { } while (0)
#endif
    while (true) {
      TLB_MARK(3);
      const unsigned long long wmask = __ballot(work);
      if (!wmask) break;
      const unsigned long long m0 = __ballot(work && depth == 0);
      const bool primary_iter = m0 != 0 && __popcll(wmask & ~m0) < deep_threshold;
      const bool act = work && ((depth == 0) == primary_iter);
      n_steps += __popcll(__ballot(act)); n_iter++;
      // ---- closest hit: src/main.cu:60-77, interval [0.01, inf] ----
      const float aa = dot3(dx, dy, dz, dx, dy, dz);
      float best_t = MRT_INF;
      int best_obj = 0x7fffffff, best_prim = -1;
      auto hit_sphere = [&](int i) { // core/sphere.cuh:32-73
        float cx, cy, cz, r;
        int obj;
        sphere_of(i, cx, cy, cz, r, obj);
        const float ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
        const float b = dot3(ocx, ocy, ocz, dx, dy, dz);
        const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - r * r;
        const float disc = b * b - aa * c;
        if (disc >= 0.0f) {
          float t;
          if (sphere_roots(aa, b, disc, 0.01f, MRT_INF, t)) {
            if (t < best_t || (t == best_t && t < MRT_INF && obj < best_obj)) { best_t = t; best_obj = obj; best_prim = i; }
          }
        }
      };
      auto hit_triangle = [&](int i) { // core/triangle.cuh:77-103
        float q[16];
        int obj;
        triangle_of(i, q, obj);
        const float pnx = q[0], pny = q[1], pnz = q[2], kp = q[3];
        const float rdn = dot3(dx, dy, dz, pnx, pny, pnz);
        const float t = (kp - dot3(ox, oy, oz, pnx, pny, pnz)) / rdn;
        // the edge tests only matter in lanes whose t is in range and could still win: none in the wave, skip
        if (!__any(act & (rdn != 0.0f) & (0.01f <= t) & (t <= best_t))) return;
        const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
        const float e0 = dot3(q[4], q[5], q[6], px, py, pz) - q[7];
        const float e1 = dot3(q[8], q[9], q[10], px, py, pz) - q[11];
        const float e2 = dot3(q[12], q[13], q[14], px, py, pz) - q[15];
        const bool hit = (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (0.01f <= t) & (t <= MRT_INF);
        if (hit) {
          if (t < best_t || (t == best_t && t < MRT_INF && obj < best_obj)) { best_t = t; best_obj = obj; best_prim = ns + i; }
        }
      };
      TLB_MARK(4);
      if (SMALL && primary_iter) {
        unsigned long long mask = primary_mask;
        while (mask) {
          const int j = __builtin_ctzll(mask);
          mask &= mask - 1;
          if (j < ns) hit_sphere(j); else hit_triangle(j - ns);
        }
      } else if (SMALL) {
        for (int i = 0; i < ns; ++i) hit_sphere(i);
        for (int i = 0; i < nt; ++i) hit_triangle(i);
      } else if (primary_iter) {
        // large scene, all active lanes at level 0: their rays lie in the tile's cone, so the wave culls the
        // hierarchy once for all of them (lane-parallel ball tests, as in mrt_cull.hip) and runs the exact test
        // of a surviving primitive for every lane
        walk_hierarchy<true>(sc.sph_sclu, sc.nscs, sc.sph_clu, sc.ncs, lane,
                             [&](float x, float y, float z, float R) {
                               const float wx = x - cam.ox, wy = y - cam.oy, wz = z - cam.oz;
                               const float L = fast_sqrt(dot3(wx, wy, wz, wx, wy, wz)) * CULL_REL + R;
                               return ball_may_touch_cone(wx, wy, wz, cluster_cull_radius(R, L), cax, cay, caz, cosT, sinT);
                             },
                             [&](int base) {
          const int i = base + lane;
          bool pass = false;
          if (i < ns) {
            const float r = sc.sph_r[i];
            const float wx = sc.sph_cx[i] - cam.ox, wy = sc.sph_cy[i] - cam.oy, wz = sc.sph_cz[i] - cam.oz;
            pass = ball_may_touch_cone(wx, wy, wz, sphere_cull_radius(r * r, dot3(wx, wy, wz, wx, wy, wz)), cax, cay, caz,
                                       cosT, sinT);
          }
          unsigned long long mask = __ballot(pass);
          while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            hit_sphere(base + j);
          }
          return true;
        });
        walk_hierarchy<true>(sc.tri_sclu, sc.nsct, sc.tri_clu, sc.nct, lane,
                             [&](float x, float y, float z, float R) {
                               return ball_may_touch_cone(x - cam.ox, y - cam.oy, z - cam.oz, R * CULL_REL + CULL_ABS, cax, cay,
                                                          caz, cosT, sinT);
                             },
                             [&](int base) {
          const int i = base + lane;
          bool pass = false;
          if (i < nt)
            pass = ball_may_touch_cone(sc.tri_bx[i] - cam.ox, sc.tri_by[i] - cam.oy, sc.tri_bz[i] - cam.oz,
                                       sc.tri_br[i] * CULL_REL + CULL_ABS, cax, cay, caz, cosT, sinT);
          unsigned long long mask = __ballot(pass);
          while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            hit_triangle(base + j);
          }
          return true;
        });
      } else {
        float ax = dx, ay = dy, az = dz;
        fast_normalize(ax, ay, az);
        walk_per_lane<true>(sc.sph_sclu, sc.nscs, sc.sph_clu, sc.ncs, act, ox, oy, oz, ax, ay, az, MRT_INF,
                            [&](int base, bool on) {
          for (int p = 0; p < 64; ++p) { // core/sphere.cuh:32-73, the lane's own sphere
            const bool ok = on && base + p < ns;
            const int i = ok ? base + p : 0;
            const float cx = sc.sph_cx[i], cy = sc.sph_cy[i], cz = sc.sph_cz[i], r = sc.sph_r[i];
            const int obj = sc.sph_obj[i];
            const float ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
            const float b = dot3(ocx, ocy, ocz, dx, dy, dz);
            const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - r * r;
            const float disc = b * b - aa * c;
            if (ok && disc >= 0.0f) {
              float t;
              if (sphere_roots(aa, b, disc, 0.01f, MRT_INF, t)) {
                if (t < best_t || (t == best_t && t < MRT_INF && obj < best_obj)) { best_t = t; best_obj = obj; best_prim = i; }
              }
            }
          }
          return true;
        });
        walk_per_lane<false>(sc.tri_sclu, sc.nsct, sc.tri_clu, sc.nct, act, ox, oy, oz, ax, ay, az, MRT_INF,
                             [&](int base, bool on) {
          for (int p = 0; p < 64; ++p) { // core/triangle.cuh:77-103, the lane's own triangle
            const bool ok = on && base + p < nt;
            const int i = ok ? base + p : 0;
            const float4 *q4 = reinterpret_cast<const float4 *>(sc.tri_rec + 16 * size_t(i));
            const float4 qa = q4[0], qb = q4[1], qc = q4[2], qd = q4[3];
            const int obj = sc.tri_obj[i];
            const float rdn = dot3(dx, dy, dz, qa.x, qa.y, qa.z);
            const float t = (qa.w - dot3(ox, oy, oz, qa.x, qa.y, qa.z)) / rdn;
            const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
            const float e0 = dot3(qb.x, qb.y, qb.z, px, py, pz) - qb.w;
            const float e1 = dot3(qc.x, qc.y, qc.z, px, py, pz) - qc.w;
            const float e2 = dot3(qd.x, qd.y, qd.z, px, py, pz) - qd.w;
            const bool hit = ok & (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (0.01f <= t) & (t <= MRT_INF);
            if (hit) {
              if (t < best_t || (t == best_t && t < MRT_INF && obj < best_obj)) { best_t = t; best_obj = obj; best_prim = ns + i; }
            }
          }
          return true;
        });
      }
      TLB_MARK(5);
      const bool have_hit = act && best_prim >= 0;
      if (act && depth == 0) { // the ID / t buffers describe the primary hit (of the last sample)
        out_id = have_hit ? best_obj : -1;
        out_t = best_t;
        if (have_hit) n_hit++;
      }
      bool finish = false;
      int levels = 0;
      float inr = 0.0f, ing = 0.0f, inb = 0.0f; // what the end of the chain returns
      if (act && !have_hit) {                   // the chain ends in the sky
        sky(inr, ing, inb);
        levels = depth;
        finish = true;
      }

      TLB_MARK(0);
      // ---- hit record (core/sphere.cuh:45-47, core/triangle.cuh:96-97) ----
      float hpx = 0.0f, hpy = 0.0f, hpz = 0.0f, hnx = 0.0f, hny = 0.0f, hnz = 0.0f;
      if (have_hit) {
        hpx = ox + best_t * dx; hpy = oy + best_t * dy; hpz = oz + best_t * dz;
        float g0, g1, g2; // sphere centre / triangle plane normal of the lane's own hit
        if (best_prim < ns) {
          g0 = sc.sph_cx[best_prim]; g1 = sc.sph_cy[best_prim]; g2 = sc.sph_cz[best_prim];
        } else {
          const float *q = sc.tri_rec + 16 * size_t(best_prim - ns);
          g0 = q[0]; g1 = q[1]; g2 = q[2];
        }
        if (best_prim < ns) {
          float nx = hpx - g0, ny = hpy - g1, nz = hpz - g2;
          normalize3(nx, ny, nz);
          hnx = nx; hny = ny; hnz = nz;
        } else {
          const float pnx = g0, pny = g1, pnz = g2;
          const bool front = dot3(dx, dy, dz, pnx, pny, pnz) < 0;
          hnx = front ? pnx : -pnx; hny = front ? pny : -pny; hnz = front ? pnz : -pnz;
        }
      }

      TLB_MARK(6);
      // ---- shadow rays + direct light at this level: core/light.cuh:135-178 ----
      float lr = 0.0f, lg = 0.0f, lb = 0.0f;
      bool have_light = false;
      if (__any(have_hit)) {
        const bool diffuse_obj = have_hit && sc.mat_kind[best_obj] == 0;
        // SMALL: ball around the hit points of this iteration (centre = the first one); with a light it spans the
        // bundle of shadow rays that each primitive's bound is tested against (mrt_cull_common.h), one primitive
        // per lane.  Valid for any set of hit points; tight in primary iterations, loose in deep ones.
        // Large scenes, primary iterations: the same bundle (centre = centre of the hit points' box), with the slab
        // of mrt_cull.hip per light, against the hierarchy.
        Bundle B;
        const bool tile_shadows = SMALL || primary_iter;
        float bex = 0.0f, bey = 0.0f, bez = 0.0f; // this lane's hit point relative to the bundle's centre
        if (tile_shadows) {
          if (SMALL) {
            const int first = __builtin_ctzll(__ballot(have_hit));
            B.mx = bcast(hpx, first); B.my = bcast(hpy, first); B.mz = bcast(hpz, first);
          } else {
            const float big = 3.0e38f;
            B.mx = 0.5f * (wave_min(have_hit ? hpx : big) + wave_max(have_hit ? hpx : -big));
            B.my = 0.5f * (wave_min(have_hit ? hpy : big) + wave_max(have_hit ? hpy : -big));
            B.mz = 0.5f * (wave_min(have_hit ? hpz : big) + wave_max(have_hit ? hpz : -big));
          }
          bex = hpx - B.mx; bey = hpy - B.my; bez = hpz - B.mz;
          B.rho = fast_sqrt(wave_max(have_hit ? dot3(bex, bey, bez, bex, bey, bez) : 0.0f)) * CULL_REL + CULL_ABS;
        }
        for (int li = 0; li < sc.nl; ++li) {
          const int lkind = sc.light_kind[li];
          const float lvx = sc.light_vec[3 * li], lvy = sc.light_vec[3 * li + 1], lvz = sc.light_vec[3 * li + 2];
          float sdx, sdy, sdz, tl, lnx, lny, lnz; // ray_from: core/light.cuh:73-81 / :47-53
          if (lkind == 0) {
            sdx = -lvx; sdy = -lvy; sdz = -lvz; tl = MRT_INF;
            lnx = lvx; lny = lvy; lnz = lvz;
          } else {
            sdx = lvx - hpx; sdy = lvy - hpy; sdz = lvz - hpz; tl = 1.0f;
            lnx = hpx - lvx; lny = hpy - lvy; lnz = hpz - lvz;
            normalize3(lnx, lny, lnz);
          }
          const float sa = dot3(sdx, sdy, sdz, sdx, sdy, sdz);
          const bool need = have_hit && !(dot3(sdx, sdy, sdz, hnx, hny, hnz) < 0); // core/light.cuh:148
          bool open = need;
          // any-hit over every object, interval [1e-3, t_light] (core/light.cuh:28-33, :153-158)
          auto shadow_sphere = [&](int i) {
            float cx, cy, cz, r;
            int obj_unused;
            sphere_of(i, cx, cy, cz, r, obj_unused);
            const float ocx = hpx - cx, ocy = hpy - cy, ocz = hpz - cz;
            const float b = dot3(ocx, ocy, ocz, sdx, sdy, sdz);
            const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - r * r;
            const float disc = b * b - sa * c;
            if (open && disc >= 0.0f) {
              float t;
              if (sphere_roots(sa, b, disc, 1e-3f, tl, t)) open = false;
            }
          };
          auto shadow_triangle = [&](int i) {
            float q[16];
            int obj_unused;
            triangle_of(i, q, obj_unused);
            const float pnx = q[0], pny = q[1], pnz = q[2], kp = q[3];
            const float rdn = dot3(sdx, sdy, sdz, pnx, pny, pnz);
            const float t = (kp - dot3(hpx, hpy, hpz, pnx, pny, pnz)) / rdn;
            if (!__any(open & (rdn != 0.0f) & (1e-3f <= t) & (t <= tl))) return;
            const float px = hpx + t * sdx, py = hpy + t * sdy, pz = hpz + t * sdz;
            const float e0 = dot3(q[4], q[5], q[6], px, py, pz) - q[7];
            const float e1 = dot3(q[8], q[9], q[10], px, py, pz) - q[11];
            const float e2 = dot3(q[12], q[13], q[14], px, py, pz) - q[15];
            const bool h = (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (1e-3f <= t) & (t <= tl);
            if (open && h) open = false;
          };
          TLB_MARK(7);
          if (SMALL) {
            unsigned long long mask = 0;
            if (__any(open)) {
              B.point = lkind != 0;
              if (lkind == 0) { B.Dx = -lvx; B.Dy = -lvy; B.Dz = -lvz; }
              else { B.Dx = lvx - B.mx; B.Dy = lvy - B.my; B.Dz = lvz - B.mz; }
              B.DD = dot3(B.Dx, B.Dy, B.Dz, B.Dx, B.Dy, B.Dz);
              bool pass = false;
              if (own_s) {
                const float wx = pbx - B.mx, wy = pby - B.my, wz = pbz - B.mz;
                const float Lm = fast_sqrt(dot3(wx, wy, wz, wx, wy, wz)) * CULL_REL + B.rho; // >= |hit point - centre|
                pass = bundle_may_touch(B, pbx, pby, pbz, sphere_cull_radius(pbr * pbr, Lm * Lm));
              } else if (own_t) {
                pass = bundle_may_touch(B, pbx, pby, pbz, pbr);
              }
              mask = __ballot(pass);
            }
            while (mask && __any(open)) {
              const int j = __builtin_ctzll(mask);
              mask &= mask - 1;
              if (j < ns) shadow_sphere(j); else shadow_triangle(j - ns);
            }
          } else if (primary_iter) {
            if (__any(open)) {
              B.point = lkind != 0;
              if (lkind == 0) { B.Dx = -lvx; B.Dy = -lvy; B.Dz = -lvz; }
              else { B.Dx = lvx - B.mx; B.Dy = lvy - B.my; B.Dz = lvz - B.mz; }
              B.DD = dot3(B.Dx, B.Dy, B.Dz, B.Dx, B.Dy, B.Dz);
              bundle_make_slab(B, cax, cay, caz, bex, bey, bez, have_hit);
              walk_hierarchy<true>(sc.sph_sclu, sc.nscs, sc.sph_clu, sc.ncs, lane,
                                   [&](float x, float y, float z, float R) {
                                     const float wx = x - B.mx, wy = y - B.my, wz = z - B.mz;
                                     const float Lm = fast_sqrt(dot3(wx, wy, wz, wx, wy, wz)) * CULL_REL + B.rho + R;
                                     return bundle_may_touch(B, x, y, z, cluster_cull_radius(R, Lm));
                                   },
                                   [&](int base) {
                const int i = base + lane;
                bool pass = false;
                if (i < ns) {
                  const float cx = sc.sph_cx[i], cy = sc.sph_cy[i], cz = sc.sph_cz[i], r = sc.sph_r[i];
                  const float wx = cx - B.mx, wy = cy - B.my, wz = cz - B.mz;
                  const float Lm = fast_sqrt(dot3(wx, wy, wz, wx, wy, wz)) * CULL_REL + B.rho; // >= |hit point - centre|
                  pass = bundle_may_touch(B, cx, cy, cz, sphere_cull_radius(r * r, Lm * Lm));
                }
                unsigned long long mask = __ballot(pass);
                while (mask && __any(open)) {
                  const int j = __builtin_ctzll(mask);
                  mask &= mask - 1;
                  shadow_sphere(base + j);
                }
                return __any(open) != 0;
              });
            }
            if (__any(open))
              walk_hierarchy<true>(sc.tri_sclu, sc.nsct, sc.tri_clu, sc.nct, lane,
                                   [&](float x, float y, float z, float R) { return bundle_may_touch(B, x, y, z, R); },
                                   [&](int base) {
                const int i = base + lane;
                bool pass = false;
                if (i < nt) pass = bundle_may_touch(B, sc.tri_bx[i], sc.tri_by[i], sc.tri_bz[i], sc.tri_br[i]);
                unsigned long long mask = __ballot(pass);
                while (mask && __any(open)) {
                  const int j = __builtin_ctzll(mask);
                  mask &= mask - 1;
                  shadow_triangle(base + j);
                }
                return __any(open) != 0;
              });
          } else {
            float ax = sdx, ay = sdy, az = sdz;
            fast_normalize(ax, ay, az);
            // a point light's shadow ray ends at the light (t <= 1): nothing beyond |sd| (+ 0.1 %) matters
            const float tdist = lkind == 0 ? MRT_INF : fast_sqrt(sa) * 1.001f;
            walk_per_lane<true>(sc.sph_sclu, sc.nscs, sc.sph_clu, sc.ncs, open, hpx, hpy, hpz, ax, ay, az, tdist,
                                [&](int base, bool on) {
              for (int p = 0; p < 64; ++p) {
                const bool ok = on && base + p < ns;
                const int i = ok ? base + p : 0;
                const float cx = sc.sph_cx[i], cy = sc.sph_cy[i], cz = sc.sph_cz[i], r = sc.sph_r[i];
                const float ocx = hpx - cx, ocy = hpy - cy, ocz = hpz - cz;
                const float b = dot3(ocx, ocy, ocz, sdx, sdy, sdz);
                const float c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - r * r;
                const float disc = b * b - sa * c;
                if (ok && open && disc >= 0.0f) {
                  float t;
                  if (sphere_roots(sa, b, disc, 1e-3f, tl, t)) open = false;
                }
              }
              return open;
            });
            walk_per_lane<false>(sc.tri_sclu, sc.nsct, sc.tri_clu, sc.nct, open, hpx, hpy, hpz, ax, ay, az, tdist,
                                 [&](int base, bool on) {
              for (int p = 0; p < 64; ++p) {
                const bool ok = on && base + p < nt;
                const int i = ok ? base + p : 0;
                const float4 *q4 = reinterpret_cast<const float4 *>(sc.tri_rec + 16 * size_t(i));
                const float4 qa = q4[0], qb = q4[1], qc = q4[2], qd = q4[3];
                const float rdn = dot3(sdx, sdy, sdz, qa.x, qa.y, qa.z);
                const float t = (qa.w - dot3(hpx, hpy, hpz, qa.x, qa.y, qa.z)) / rdn;
                const float px = hpx + t * sdx, py = hpy + t * sdy, pz = hpz + t * sdz;
                const float e0 = dot3(qb.x, qb.y, qb.z, px, py, pz) - qb.w;
                const float e1 = dot3(qc.x, qc.y, qc.z, px, py, pz) - qc.w;
                const float e2 = dot3(qd.x, qd.y, qd.z, px, py, pz) - qd.w;
                const bool h = ok & (rdn != 0.0f) & (e0 > -1e-6f) & (e1 > -1e-6f) & (e2 > -1e-6f) & (1e-3f <= t) & (t <= tl);
                if (open && h) open = false;
              }
              return open;
            });
          }
          TLB_MARK(8);
          if (need) { n_shadow++; if (!open) n_occl++; }
          if (open) {
            // light_bounce with the LIGHT's normal, as the reference passes it (core/light.cuh:164-168);
            // the viewing ray is the current ray of the chain
            const float lcr = sc.light_rgb[3 * li], lcg = sc.light_rgb[3 * li + 1], lcb = sc.light_rgb[3 * li + 2];
            const float dn = dot3(sdx, sdy, sdz, lnx, lny, lnz);
            float cr, cg, cb;
            if (diffuse_obj) { // DiffuseMaterial::light_bounce, core/material.cuh:63-75
              const float scale = -dn;
              if (scale > 0.0f) { cr = scale * lcr; cg = scale * lcg; cb = scale * lcb; }
              else { cr = 0.0f; cg = 0.0f; cb = 0.0f; }
            } else {           // MetallicMaterial::light_bounce, core/material.cuh:134-158
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

      TLB_MARK(1);
      // ---- push the level; continue with the bounced ray or end the chain ----
      if (have_hit) {
        const float *m = sc.mat_rgb + 3 * best_obj;
        const float m0 = m[0], m1 = m[1], m2 = m[2];
#pragma unroll
        for (int k = 0; k < MAX_LEVELS; ++k)
          if (k == depth) {
            stC[k][0] = m0; stC[k][1] = m1; stC[k][2] = m2;
            stL[k][0] = lr; stL[k][1] = lg; stL[k][2] = lb;
          }
        stH = have_light ? (stH | (1u << depth)) : (stH & ~(1u << depth));
        // the bounce (src/main.cu:80-83) happens at every level, the last one included: a diffuse surface
        // advances the pixel's stream even when its bounce ray is never followed
        float bdx, bdy, bdz;
        if (use_rng && sc.mat_kind[best_obj] == 0) { // DiffuseMaterial::bounce, core/material.cuh:52-61
          float rx = uniform(), ry = uniform(), rz = uniform();
          // random_in_unit_sphere (cuda_utils/curand_state.cuh:93-109) tests the vector against the unit ball
          // around (0.5, 0.5, 0.5) and returns it UNcentred; in (0,1]^3 the test cannot fail (<= 0.75)
          for (;;) {
            const float cx = rx - 0.5f, cy = ry - 0.5f, cz = rz - 0.5f;
            if (dot3(cx, cy, cz, cx, cy, cz) <= 1.0f) break;
            rx = uniform(); ry = uniform(); rz = uniform();
          }
          bdx = hnx + rx; bdy = hny + ry; bdz = hnz + rz;
        } else { // MetallicMaterial::bounce, core/material.cuh:126-132, :151-158
          const float s2 = 2.0f * dot3(dx, dy, dz, hnx, hny, hnz);
          bdx = dx - s2 * hnx; bdy = dy - s2 * hny; bdz = dz - s2 * hnz;
        }
        TLB_MARK(9);
        if (depth + 1 > A.max_depth) { // what the recursive call returns beyond the limit, src/main.cu:51-53
          inr = 0.01f; ing = 0.01f; inb = 0.01f;
          levels = depth + 1;
          finish = true;
        } else {
          ox = hpx; oy = hpy; oz = hpz;
          dx = bdx; dy = bdy; dz = bdz;
          depth++;
        }
      }

      TLB_MARK(2);
      // ---- a chain ended: fold it (src/main.cu:91-98, innermost level first), add it to the pixel, next sample ----
      if (__any(finish)) {
        float cr = inr, cg = ing, cb = inb;
#pragma unroll
        for (int k = MAX_LEVELS - 1; k >= 0; --k) {
          if (!__any(finish && k < levels)) continue; // most chains are short: skip the levels no lane has
          if (k < levels) {
            const float mr = stC[k][0], mg = stC[k][1], mb = stC[k][2];
            if (stH & (1u << k)) { cr = cr * mr + stL[k][0] * mr; cg = cg * mg + stL[k][1] * mg; cb = cb * mb + stL[k][2] * mb; }
            else { cr = cr * mr; cg = cg * mg; cb = cb * mb; }
          }
        }
        if (finish) {
          accr += cr; accg += cg; accb += cb; // src/main.cu:151
          sidx++;
          depth = 0;
          work = sidx < nsamp;
          if (work) primary_ray(sidx);
        }
      }
    }
  }

  if (valid) { // src/main.cu:155
    const float fn = float(nsamp);
    const size_t pix = size_t(A.frame_layout ? grow : lrow) * A.W + col;
    A.rgb[3 * pix + 0] = accr / fn;
    A.rgb[3 * pix + 1] = accg / fn;
    A.rgb[3 * pix + 2] = accb / fn;
    if (A.id) A.id[pix] = out_id;
    if (A.t) A.t[pix] = out_t;
  }

#ifdef MRT_TIMELINE
  if (lane == 0 && tile < TLB_MAX_WAVES) {
    unsigned long long *tlp = g_timeline_bounce + (size_t)TLB_WORDS * tile;
    // the tile's predicted weight rides in the top byte of the cycle count (the weights sit right below the order)
    const unsigned long long tl_w = A.tile_order ? (unsigned long long)(A.tile_order - (size_t)tiles_x * tiles_y)[tile] : 0ull;
    tlp[0] = tl_rt0; tlp[1] = __builtin_amdgcn_s_memrealtime(); tlp[2] = (__builtin_amdgcn_s_memtime() - tl_c0) | (tl_w << 56);
    tlp[3] = (unsigned long long)n_steps | ((unsigned long long)n_iter << 32);
    for (int k = 0; k < 12; ++k) tlp[4 + k] = tl_acc[k];
  }
#endif
  if (A.stats) {
    // counters are spread over stats_slots copies: same-address atomics cost ~12 ns each, one after the other
    unsigned long long *st = A.stats + 8u * (blockIdx.x & (unsigned)(A.stats_slots - 1));
    unsigned int v4[4] = {(valid ? 1u : 0u) * (unsigned)nsamp, n_shadow, n_hit, n_occl};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      unsigned int x = v4[k];
      for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
      if (lane == 0 && x) atomicAdd(&st[k], (unsigned long long)x);
    }
    if (lane == 0 && n_iter) { // reported as exact_tests / clusters_opened: chain steps of all lanes / 64 x loop iterations
      atomicAdd(&st[4], (unsigned long long)n_steps);
      atomicAdd(&st[5], (unsigned long long)n_iter * 64ull);
    }
  }
}

// ---- tile order: heaviest first -------------------------------------------------------------------------------
// A pixel's samples share one random stream, so a pixel is a sequential job, and the frame cannot end before its
// longest job: 121 samples x 9 levels where a mirror faces a metallic sphere.  Waves that start such tiles late
// leave the GPU idle behind them (7.3 ms in row order, 5.2 ms with the tiles sorted by their measured cost).
// Cost is predicted here before the frame: 0 for a tile whose primary rays can hit nothing, else 1 + the largest sum,
// over the tile's 64 pixels, of the chain lengths of WEIGH_SAMPLES trial samples traced with approximate arithmetic:
// samples spread over the 11 x 11 mesh, a diffuse surface bouncing along normal + a hashed vector of (0,1)^3 -- a
// stand-in for the pixel's random stream; this only orders the work, results do not depend on it.  (The first
// predictor traced ONE sample through the pixel centre with every diffuse bounce along normal + (0.5, 0.5, 0.5): good
// enough while the frame waited for one wave anyway, but it put a few dozen tiles whose pixels reach the space between
// the mirrors only through some of their random bounces -- 900-1000 loop iterations -- into the big bucket of 240-
// iteration tiles, where they started at 2.5 ms of a frame that could have ended at 3.4.)  Two small kernels turn the
// weights into an order, heaviest bucket first: bounce_count_kernel sizes the buckets, bounce_order_kernel hands
// out positions.
enum { WEIGH_SAMPLES = 4, ORDER_BUCKETS = 1 + WEIGH_SAMPLES * MAX_LEVELS + 1 };


__global__ __launch_bounds__(256) void bounce_weigh_kernel(SmallArgs S, uint32_t *__restrict__ weight)
{
  const RenderArgs &A = S.A;
  const DevScene &sc = A.sc;
  const DevCamera &cam = A.cam;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tx = lane & 7, ty = lane >> 3;
  const int ns = sc.ns, nt = sc.nt;
  const int tiles_x = (A.W + 7) >> 3, tiles_y = (A.local_rows + 7) >> 3;
  const int tile = blockIdx.x * 4 + wave; // one wave per tile, one lane per pixel, as the render kernel
  if (tile >= tiles_x * tiles_y) return;
  const int tile_y = tile / tiles_x, tile_x = tile - tile_y * tiles_x;
  const int lrow0 = tile_y * 8;
  const int strip = lrow0 / A.strip_rows;
  const int grow0 = global_strip(strip, A.rank, A.nranks) * A.strip_rows + (lrow0 - strip * A.strip_rows);
  // candidates: the tile's cone against the spheres, the edge functions against the triangles (lane = primitive)
  bool pass = false;
  {
    const float um = (float(tile_x * 8) + 3.5f) / float(A.W), vm = (float(grow0) + 3.5f) / float(A.H);
    float ax = ((cam.ux + um * cam.hx) + vm * cam.vx) - cam.ox;
    float ay = ((cam.uy + um * cam.hy) + vm * cam.vy) - cam.oy;
    float az = ((cam.uz + um * cam.hz) + vm * cam.vz) - cam.oz;
    const float inv = fast_rsq(dot3(ax, ay, az, ax, ay, az));
    ax *= inv; ay *= inv; az *= inv;
    float sinT = A.cone_ext * inv * (1.0f + 1e-4f) + 1e-6f;
    float cosT = fast_sqrt(fmaxf(1.0f - sinT * sinT, 0.0f)) * (1.0f - 1e-5f) - 1e-6f;
    if (!(sinT < 0.99f) || !(cosT > 0.0f)) { cosT = -1.0f; sinT = 0.0f; }
    if (lane < ns) {
      pass = ball_may_touch_cone(S.prm[9][lane], S.prm[10][lane], S.prm[11][lane], S.prm[12][lane], ax, ay, az, cosT, sinT);
    } else if (lane < ns + nt) {
      bool cull = false;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        cull |= __builtin_fmaf(um, S.prm[3 * i + 1][lane], __builtin_fmaf(vm, S.prm[3 * i + 2][lane], S.prm[3 * i][lane])) < 0.0f;
      pass = !cull;
    }
  }
  uint32_t w = 0;
  if (__any(pass)) {
    uint32_t steps = 0; // chain steps of this pixel over the trial samples
    const int nsamp = A.spp > 1 ? WEIGH_SAMPLES : 1;
    for (int s = 0; s < nsamp; ++s) {
      // sample positions: mesh cells (1,1), (9,3), (3,9), (7,7) of the 11 x 11 pattern (pipeline/multisample_mesh.cuh)
      const int mc = s == 0 ? 1 : s == 1 ? 9 : s == 2 ? 3 : 7, mr = s == 0 ? 1 : s == 1 ? 3 : s == 2 ? 9 : 7;
      float u = float(tile_x * 8 + tx) / float(A.W), v = float(grow0 + ty) / float(A.H);
      if (A.spp > 1) { u += (-1.0f + 2.0f * float(mc + 1) / 13.0f) / float(A.W); v += (-1.0f + 2.0f * float(mr + 1) / 13.0f) / float(A.H); }
      float ox = cam.ox, oy = cam.oy, oz = cam.oz;
      float dx = ((cam.ux + u * cam.hx) + v * cam.vx) - cam.ox;
      float dy = ((cam.uy + u * cam.hy) + v * cam.vy) - cam.oy;
      float dz = ((cam.uz + u * cam.hz) + v * cam.vz) - cam.oz;
      uint32_t h = (uint32_t(tile) * 64u + uint32_t(lane)) * 2654435761u + uint32_t(s) * 0x9E3779B9u; // bounce hash state
      bool alive = true;
      for (int level = 0; level <= A.max_depth && __any(alive); ++level) {
        float best = MRT_INF, nx = 0.0f, ny = 0.0f, nz = 0.0f;
        int obj = -1;
        const float aa = dot3(dx, dy, dz, dx, dy, dz);
        for (int i = 0; i < ns; ++i) {
          const float cx = sc.sph_cx[i], cy = sc.sph_cy[i], cz = sc.sph_cz[i], r = sc.sph_r[i];
          const float ocx = ox - cx, ocy = oy - cy, ocz = oz - cz;
          const float b = dot3(ocx, ocy, ocz, dx, dy, dz), c = dot3(ocx, ocy, ocz, ocx, ocy, ocz) - r * r;
          const float disc = b * b - aa * c;
          if (disc >= 0.0f) {
            const float sq = fast_sqrt(disc), ia = fast_rcp(aa);
            float t = (-b - sq) * ia;
            if (!(t >= 0.01f)) t = (-b + sq) * ia;
            if (t >= 0.01f && t < best) {
              best = t; obj = sc.sph_obj[i];
              const float ir = fast_rcp(r);
              nx = (ox + t * dx - cx) * ir; ny = (oy + t * dy - cy) * ir; nz = (oz + t * dz - cz) * ir;
            }
          }
        }
        for (int i = 0; i < nt; ++i) {
          const float *q = sc.tri_rec + 16 * size_t(i);
          const float rdn = dot3(dx, dy, dz, q[0], q[1], q[2]);
          const float t = (q[3] - dot3(ox, oy, oz, q[0], q[1], q[2])) * fast_rcp(rdn);
          const float px = ox + t * dx, py = oy + t * dy, pz = oz + t * dz;
          const bool hit = rdn != 0.0f && t >= 0.01f && t < best && dot3(q[4], q[5], q[6], px, py, pz) - q[7] > -1e-6f &&
                           dot3(q[8], q[9], q[10], px, py, pz) - q[11] > -1e-6f &&
                           dot3(q[12], q[13], q[14], px, py, pz) - q[15] > -1e-6f;
          if (hit) {
            best = t; obj = sc.tri_obj[i];
            const float sgn = rdn < 0.0f ? 1.0f : -1.0f;
            nx = sgn * q[0]; ny = sgn * q[1]; nz = sgn * q[2];
          }
        }
        if (alive && obj < 0) alive = false; // the sky ends the chain
        if (alive) {
          ++steps;
          ox += best * dx; oy += best * dy; oz += best * dz;
          if (sc.mat_kind[obj] == 0) { // diffuse: normal + a vector of (0,1)^3
            h = h * 1664525u + 1013904223u; const float rx = float(h >> 8) * (1.0f / 16777216.0f);
            h = h * 1664525u + 1013904223u; const float ry = float(h >> 8) * (1.0f / 16777216.0f);
            h = h * 1664525u + 1013904223u; const float rz = float(h >> 8) * (1.0f / 16777216.0f);
            dx = nx + rx; dy = ny + ry; dz = nz + rz;
          } else {
            const float s2 = 2.0f * dot3(dx, dy, dz, nx, ny, nz);
            dx -= s2 * nx; dy -= s2 * ny; dz -= s2 * nz;
          }
        }
      }
    }
    w = 1u + (uint32_t)wave_max(float(steps)); // the tile costs what its longest pixel costs
    if (w > ORDER_BUCKETS - 1) w = ORDER_BUCKETS - 1;
  }
  if (lane == 0) weight[tile] = w;
}

// one lane per tile: bucket sizes, one atomic per wave and bucket
__global__ __launch_bounds__(256) void bounce_count_kernel(const uint32_t *__restrict__ weight, uint32_t *__restrict__ count,
                                                           int ntiles)
{
  const int tile = blockIdx.x * 256 + threadIdx.x;
  const bool in_range = tile < ntiles;
  const uint32_t w = in_range ? weight[tile] : 0u;
  const int lane = threadIdx.x & 63;
  for (uint32_t b = 0; b < ORDER_BUCKETS; ++b) {
    const unsigned long long m = __ballot(in_range && w == b);
    if (m && lane == __builtin_ctzll(m)) atomicAdd(&count[b], (uint32_t)__popcll(m));
  }
}

__global__ __launch_bounds__(256) void bounce_order_kernel(const uint32_t *__restrict__ weight,
                                                           const uint32_t *__restrict__ count,
                                                           uint32_t *__restrict__ cursor, uint32_t *__restrict__ order,
                                                           int ntiles)
{
  const int tile = blockIdx.x * 256 + threadIdx.x;
  const bool in_range = tile < ntiles;
  const uint32_t w = in_range ? weight[tile] : 0u;
  uint32_t first = 0; // position of the tile's bucket: heavier buckets come before it
  for (uint32_t b = ORDER_BUCKETS - 1; b > w; --b) first += count[b];
  const int lane = threadIdx.x & 63;
  for (uint32_t b = 0; b < ORDER_BUCKETS; ++b) {
    const unsigned long long m = __ballot(in_range && w == b);
    if (!m) continue;
    const int leader = __builtin_ctzll(m);
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(&cursor[b], (uint32_t)__popcll(m));
    base = __shfl(base, leader, 64);
    if (in_range && w == b) order[first + base + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)tile;
  }
}

// three waves per SIMD (at most 168 registers): left alone the compiler takes 172 and loses a wave
template <bool MULTI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 8)))
void render_bounce_kernel(RenderArgs A) { render_bounce_body<MULTI, false, 4>(A); }
// four waves per SIMD (128 registers, ~30 spilled: 5.85 ms on the reference's frame against 6.05 with the 157 the
// compiler takes by itself; five waves spill far too much: 10 ms)
// One wave per workgroup: the waves share nothing, and a tile of the reference's frame lives between 12 us (sky) and
// more than a millisecond -- a workgroup of four keeps its four wave slots until the slowest is done (measured on the
// reference's frame: 1.6 of 4 slots per SIMD in use on average with four waves per workgroup).
#ifndef MRT_BOUNCE_SMALL_WAVES
#define MRT_BOUNCE_SMALL_WAVES 4
#endif
#ifndef MRT_BOUNCE_WPB
#define MRT_BOUNCE_WPB 1
#endif
template <bool MULTI>
__global__ __launch_bounds__(64 * MRT_BOUNCE_WPB) __attribute__((amdgpu_waves_per_eu(MRT_BOUNCE_SMALL_WAVES, 8)))
void render_bounce_small_kernel(SmallArgs S) { render_bounce_body<MULTI, true, MRT_BOUNCE_WPB>(S); }

hipError_t launch_render_bounce(const RenderArgs &in, const SmallHost *hs, uint32_t *scratch, hipStream_t stream)
{
  if (in.max_depth < 1 || in.max_depth > MAX_LEVELS - 1) return hipErrorInvalidValue;
  const int tiles_x = (in.W + 7) / 8, tiles_y = (in.local_rows + 7) / 8;
  if (tiles_x == 0 || tiles_y == 0) return hipSuccess;
  const int ntiles = tiles_x * tiles_y;
  const dim3 grid((ntiles + 3) / 4), block(256);
  // MRT_BOUNCE_DEEP=n starts a deep iteration as soon as n lanes wait (default 8; 64 = only when no lane is at
  // level 0).  hs == null (a large scene, or MRT_OPT_CULL = 0) leaves the small-scene culling data out: every
  // ray then walks the hierarchy.
  static const int deep = getenv("MRT_BOUNCE_DEEP") ? atoi(getenv("MRT_BOUNCE_DEEP")) : 8;
  if (hs) {
    if (hs->ns != in.sc.ns || hs->nt != in.sc.nt || hs->ns + hs->nt > 64) return hipErrorInvalidValue;
    SmallArgs S;
    small_prepare_args(in, *hs, 1, S);
    S.A.s_cap = deep;
    // many samples per pixel: order the tiles by predicted cost (MRT_BOUNCE_ORDER=0: row by row)
    static const bool ordered = !(getenv("MRT_BOUNCE_ORDER") && atoi(getenv("MRT_BOUNCE_ORDER")) == 0);
    if (scratch && in.spp > 1 && ordered) {
      static_assert(2 * ORDER_BUCKETS <= BOUNCE_SCRATCH_HEAD, "bucket counters and cursors sit in the head of the scratch");
      uint32_t *count = scratch, *cursor = scratch + ORDER_BUCKETS, *weight = scratch + BOUNCE_SCRATCH_HEAD, *order = weight + ntiles;
      hipError_t e = hipMemsetAsync(scratch, 0, BOUNCE_SCRATCH_HEAD * sizeof(uint32_t), stream);
      if (e != hipSuccess) return e;
      const dim3 pg((ntiles + 255) / 256);
      hipLaunchKernelGGL(bounce_weigh_kernel, grid, block, 0, stream, S, weight);
      hipLaunchKernelGGL(bounce_count_kernel, pg, block, 0, stream, weight, count, ntiles);
      hipLaunchKernelGGL(bounce_order_kernel, pg, block, 0, stream, weight, count, cursor, order, ntiles);
      if ((e = hipGetLastError()) != hipSuccess) return e;
      S.A.tile_order = order;
    }
    // the two-wave pipeline per tile (mrt_bounce_pipe.hip); MRT_BOUNCE_PIPE=0: the one-wave kernel it replaced
    // (also MRT_OPT_CULL_TEAM = 1, "one wave per tile": the tests run both)
    static const bool pipe = !(getenv("MRT_BOUNCE_PIPE") && atoi(getenv("MRT_BOUNCE_PIPE")) == 0);
    if (pipe && opt_team(in) != 1) return launch_render_bounce_pipe(S, stream);
    const dim3 sgrid((ntiles + MRT_BOUNCE_WPB - 1) / MRT_BOUNCE_WPB), sblock(64 * MRT_BOUNCE_WPB);
    if (in.spp > 1) hipLaunchKernelGGL(render_bounce_small_kernel<true>, sgrid, sblock, 0, stream, S);
    else hipLaunchKernelGGL(render_bounce_small_kernel<false>, sgrid, sblock, 0, stream, S);
    return hipGetLastError();
  }
  RenderArgs A = in;
  A.s_cap = deep;
  { // largest offset of a ray of the padded tile (pixels -1..8 both ways) from the tile's centre ray, rounded up
    const double hw = 4.5 / A.W, hh = 4.5 / A.H;
    const double hx[3] = {hw * A.cam.hx, hw * A.cam.hy, hw * A.cam.hz};
    const double hy[3] = {hh * A.cam.vx, hh * A.cam.vy, hh * A.cam.vz};
    double p = 0, m = 0;
    for (int k = 0; k < 3; ++k) { p += (hx[k] + hy[k]) * (hx[k] + hy[k]); m += (hx[k] - hy[k]) * (hx[k] - hy[k]); }
    A.cone_ext = float(std::sqrt(p > m ? p : m) * 1.001) + 2.0f * camera_rounding(A.cam); // the ray's and the centre's
  }
  if (A.spp > 1) hipLaunchKernelGGL(render_bounce_kernel<true>, grid, block, 0, stream, A);
  else hipLaunchKernelGGL(render_bounce_kernel<false>, grid, block, 0, stream, A);
  return hipGetLastError();
}

} // namespace mrt
